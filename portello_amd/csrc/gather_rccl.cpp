// gather_rccl.cpp -- the record gather of SURVEY.md 8(e) behind the C ABI (include/portello_liftover.h, plo_gather_*; VERDICT r5 missing #4):
// what a host that is not Python binds for INTEGRATION.md section 6 route (b).  One process per GPU; the reference's sink is one locked
// writer behind all workers (src/read_alignment_scanner.rs:24, :483) -- here rank `root` receives every rank's result arrays:
//   1. sizes: ncclAllGather of {n_items, n_cigar} per rank (16 bytes each);
//   2. payload: ONE group of ncclSend (peers) / ncclRecv (root) per result array, straight out of / into device memory -- xGMI is point
//      to point, every peer's link runs into the root at the same time, a ring would be per-link bound for no benefit.
// RCCL is bound by name at the first call (dlopen librccl.so.1): a process that never gathers needs no RCCL, and a box without it gets
// PLO_ERR_NO_DEVICE with a message instead of a load error of the whole library.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include <mutex>
#include <string>
#include <vector>

#include "../../include/portello_liftover.h"

extern "C" void *plo_ctx_stream(plo_ctx *c);
extern "C" int plo_ctx_device(plo_ctx *c);

namespace {

struct UniqueId {
    char internal[128];
};
typedef void *comm_t;
typedef int (*fn_get_unique_id)(UniqueId *);
typedef int (*fn_comm_init_rank)(comm_t *, int, UniqueId, int);
typedef int (*fn_comm_destroy)(comm_t);
typedef int (*fn_all_gather)(const void *, void *, size_t, int, comm_t, hipStream_t);
typedef int (*fn_send)(const void *, size_t, int, int, comm_t, hipStream_t);
typedef int (*fn_recv)(void *, size_t, int, int, comm_t, hipStream_t);
typedef int (*fn_group)(void);
typedef const char *(*fn_err)(int);
constexpr int NCCL_UINT8 = 1, NCCL_UINT64 = 5;  // ncclDataType_t (rccl.h)

struct Rccl {
    void *lib = nullptr;
    fn_get_unique_id get_unique_id = nullptr;
    fn_comm_init_rank comm_init_rank = nullptr;
    fn_comm_destroy comm_destroy = nullptr;
    fn_all_gather all_gather = nullptr;
    fn_send send = nullptr;
    fn_recv recv = nullptr;
    fn_group group_start = nullptr, group_end = nullptr;
    fn_err err_string = nullptr;
    std::string why;
};
Rccl g_rccl;
std::once_flag g_once;
thread_local std::string g_err;

const Rccl &rccl() {
    std::call_once(g_once, [] {
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            g_rccl.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (g_rccl.lib) break;
        }
        if (!g_rccl.lib) {
            g_rccl.why = std::string("librccl.so.1 not found: ") + (dlerror() ? dlerror() : "");
            return;
        }
        auto sym = [&](const char *n) {
            void *p = dlsym(g_rccl.lib, n);
            if (!p && g_rccl.why.empty()) g_rccl.why = std::string("RCCL symbol missing: ") + n;
            return p;
        };
        g_rccl.get_unique_id = (fn_get_unique_id)sym("ncclGetUniqueId");
        g_rccl.comm_init_rank = (fn_comm_init_rank)sym("ncclCommInitRank");
        g_rccl.comm_destroy = (fn_comm_destroy)sym("ncclCommDestroy");
        g_rccl.all_gather = (fn_all_gather)sym("ncclAllGather");
        g_rccl.send = (fn_send)sym("ncclSend");
        g_rccl.recv = (fn_recv)sym("ncclRecv");
        g_rccl.group_start = (fn_group)sym("ncclGroupStart");
        g_rccl.group_end = (fn_group)sym("ncclGroupEnd");
        g_rccl.err_string = (fn_err)sym("ncclGetErrorString");
    });
    return g_rccl;
}

// the arrays of a plo_batch_out that travel, in this order (a pair of ranks matches its messages in posting order)
struct Field {
    size_t offset;    // of the pointer inside plo_batch_out
    size_t elem;      // bytes per element
    bool per_cigar;   // n_cigar elements (else n_items)
};
const Field kFields[] = {
    {offsetof(plo_batch_out, item_seg), 4, false},        {offsetof(plo_batch_out, item_cseg), 4, false},
    {offsetof(plo_batch_out, item_status), 1, false},     {offsetof(plo_batch_out, item_need_flipped), 1, false},
    {offsetof(plo_batch_out, item_mapq), 1, false},       {offsetof(plo_batch_out, item_chrom_index), 4, false},
    {offsetof(plo_batch_out, item_ref_pos), 8, false},    {offsetof(plo_batch_out, item_cigar_off), 8, false},
    {offsetof(plo_batch_out, item_cigar_len), 4, false},  {offsetof(plo_batch_out, cigar), 4, true},
};
constexpr int N_FIELDS = (int)(sizeof(kFields) / sizeof(kFields[0]));

void *&field_ptr(plo_batch_out &o, int f) { return *(void **)((char *)&o + kFields[f].offset); }
const void *field_ptr(const plo_batch_out &o, int f) { return *(void *const *)((const char *)&o + kFields[f].offset); }

struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    hipError_t ensure(size_t bytes) {
        if (bytes <= cap) return hipSuccess;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        const size_t want = bytes + bytes / 4 + 256;
        hipError_t e = hipMalloc(&p, want);
        if (e == hipSuccess) cap = want;
        return e;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
};

}  // namespace

struct plo_gather {
    comm_t comm = nullptr;
    int rank = 0, world = 1, device = 0;
    DevBuf sizes_send, sizes_recv;                 // {n_items, n_cigar} of this rank / of every rank
    std::vector<unsigned long long> sizes_host;    // [2 * world], after plo_gather_records on every rank
    std::vector<std::vector<DevBuf>> recv;         // root: [world][N_FIELDS]
    hipStream_t last_stream = nullptr;
    std::string err;
};

static plo_status gfail(plo_gather *g, plo_status st, const std::string &msg) {
    if (g) g->err = msg;
    g_err = msg;
    return st;
}
static std::string nccl_err(int rc) {
    const Rccl &r = rccl();
    return std::string(r.err_string ? r.err_string(rc) : "RCCL error") + " (" + std::to_string(rc) + ")";
}

extern "C" {

const char *plo_gather_last_error(const plo_gather *g) { return g ? g->err.c_str() : g_err.c_str(); }

plo_status plo_gather_unique_id(uint8_t id[PLO_GATHER_ID_BYTES]) {
    static_assert(PLO_GATHER_ID_BYTES == sizeof(UniqueId), "ncclUniqueId");
    if (!id) return PLO_ERR_INVALID_ARG;
    const Rccl &r = rccl();
    if (!r.get_unique_id) return gfail(nullptr, PLO_ERR_NO_DEVICE, "RCCL is not available: " + r.why);
    UniqueId u;
    const int rc = r.get_unique_id(&u);
    if (rc != 0) return gfail(nullptr, PLO_ERR_INTERNAL, "ncclGetUniqueId: " + nccl_err(rc));
    memcpy(id, u.internal, sizeof(u.internal));
    return PLO_OK;
}

plo_status plo_gather_create(const uint8_t id[PLO_GATHER_ID_BYTES], int rank, int world, int device, plo_gather **out) {
    if (!id || !out || world < 1 || rank < 0 || rank >= world) return PLO_ERR_INVALID_ARG;
    *out = nullptr;
    const Rccl &r = rccl();
    if (!r.comm_init_rank || !r.all_gather || !r.send || !r.recv || !r.group_start || !r.group_end || !r.comm_destroy)
        return gfail(nullptr, PLO_ERR_NO_DEVICE, "RCCL is not available: " + r.why);
    if (hipSetDevice(device) != hipSuccess) return gfail(nullptr, PLO_ERR_NO_DEVICE, "plo_gather_create: hipSetDevice failed");
    plo_gather *g = new plo_gather();
    g->rank = rank;
    g->world = world;
    g->device = device;
    UniqueId u;
    memcpy(u.internal, id, sizeof(u.internal));
    const int rc = r.comm_init_rank(&g->comm, world, u, rank);
    if (rc != 0) {
        const std::string msg = "ncclCommInitRank: " + nccl_err(rc);
        delete g;
        return gfail(nullptr, PLO_ERR_INTERNAL, msg);
    }
    if (g->sizes_send.ensure(16) != hipSuccess || g->sizes_recv.ensure(16 * (size_t)world) != hipSuccess) {
        plo_gather_destroy(g);
        return gfail(nullptr, PLO_ERR_OUT_OF_MEMORY, "plo_gather_create: out of device memory");
    }
    g->sizes_host.assign(2 * (size_t)world, 0ull);
    g->recv.assign((size_t)world, std::vector<DevBuf>(N_FIELDS));
    *out = g;
    return PLO_OK;
}

void plo_gather_destroy(plo_gather *g) {
    if (!g) return;
    (void)hipSetDevice(g->device);
    if (g->last_stream) (void)hipStreamSynchronize(g->last_stream);
    const Rccl &r = rccl();
    if (g->comm && r.comm_destroy) (void)r.comm_destroy(g->comm);
    g->sizes_send.release();
    g->sizes_recv.release();
    for (auto &v : g->recv)
        for (auto &b : v) b.release();
    delete g;
}

plo_status plo_gather_records(plo_gather *g, plo_ctx *ctx, const plo_batch_out *out, int root, plo_batch_out *gathered) {
    if (!g || !ctx || !out || root < 0 || root >= g->world || (g->rank == root && !gathered)) return PLO_ERR_INVALID_ARG;
    const Rccl &r = rccl();
    if (plo_ctx_device(ctx) != g->device) return gfail(g, PLO_ERR_INVALID_ARG, "plo_gather_records: the context lives on another device than the communicator");
    if (hipSetDevice(g->device) != hipSuccess) return gfail(g, PLO_ERR_NO_DEVICE, "hipSetDevice failed");
    hipStream_t st = (hipStream_t)plo_ctx_stream(ctx);  // the exchange is ordered behind the context's kernels (the compaction)
    g->last_stream = st;
    // 1. sizes
    const unsigned long long mine[2] = {(unsigned long long)out->n_items, (unsigned long long)out->n_cigar};
    if (hipMemcpyAsync(g->sizes_send.p, mine, 16, hipMemcpyHostToDevice, st) != hipSuccess) return gfail(g, PLO_ERR_INTERNAL, "size upload failed");
    int rc = r.all_gather(g->sizes_send.p, g->sizes_recv.p, 2, NCCL_UINT64, g->comm, st);
    if (rc != 0) return gfail(g, PLO_ERR_INTERNAL, "ncclAllGather: " + nccl_err(rc));
    if (hipMemcpyAsync(g->sizes_host.data(), g->sizes_recv.p, 16 * (size_t)g->world, hipMemcpyDeviceToHost, st) != hipSuccess ||
        hipStreamSynchronize(st) != hipSuccess)  // (mine[] and sizes_host must be through before they are looked at / go out of scope)
        return gfail(g, PLO_ERR_INTERNAL, "size exchange failed");
    // 2. payload: one group
    if (g->rank == root) {
        for (int p = 0; p < g->world; ++p) {
            plo_batch_out &o = gathered[p];
            memset(&o, 0, sizeof(o));
            o.n_items = (uint32_t)g->sizes_host[2 * (size_t)p];
            o.n_cigar = g->sizes_host[2 * (size_t)p + 1];
            for (int f = 0; f < N_FIELDS; ++f) {
                if (p == root) {
                    field_ptr(o, f) = const_cast<void *>(field_ptr(*out, f));  // the root's own arrays: not copied
                    continue;
                }
                const size_t bytes = (kFields[f].per_cigar ? (size_t)o.n_cigar : (size_t)o.n_items) * kFields[f].elem;
                if (g->recv[(size_t)p][(size_t)f].ensure(bytes ? bytes : 1) != hipSuccess) return gfail(g, PLO_ERR_OUT_OF_MEMORY, "plo_gather_records: out of device memory");
                field_ptr(o, f) = g->recv[(size_t)p][(size_t)f].p;
            }
        }
    }
    if (g->world > 1) {
        if ((rc = r.group_start()) != 0) return gfail(g, PLO_ERR_INTERNAL, "ncclGroupStart: " + nccl_err(rc));
        int bad = 0;
        if (g->rank == root) {
            for (int p = 0; p < g->world && !bad; ++p) {
                if (p == root) continue;
                for (int f = 0; f < N_FIELDS && !bad; ++f) {
                    const size_t bytes = (kFields[f].per_cigar ? (size_t)gathered[p].n_cigar : (size_t)gathered[p].n_items) * kFields[f].elem;
                    if (bytes) bad = r.recv(field_ptr(gathered[p], f), bytes, NCCL_UINT8, p, g->comm, st);
                }
            }
        } else {
            for (int f = 0; f < N_FIELDS && !bad; ++f) {
                const size_t bytes = (kFields[f].per_cigar ? (size_t)out->n_cigar : (size_t)out->n_items) * kFields[f].elem;
                if (bytes) bad = r.send(field_ptr(*out, f), bytes, NCCL_UINT8, root, g->comm, st);
            }
        }
        rc = r.group_end();
        if (bad) return gfail(g, PLO_ERR_INTERNAL, "ncclSend / ncclRecv: " + nccl_err(bad));
        if (rc != 0) return gfail(g, PLO_ERR_INTERNAL, "ncclGroupEnd: " + nccl_err(rc));
    }
    return PLO_OK;
}

plo_status plo_gather_wait(plo_gather *g) {
    if (!g) return PLO_ERR_INVALID_ARG;
    if (hipSetDevice(g->device) != hipSuccess) return gfail(g, PLO_ERR_NO_DEVICE, "hipSetDevice failed");
    if (g->last_stream && hipStreamSynchronize(g->last_stream) != hipSuccess) return gfail(g, PLO_ERR_INTERNAL, "the record exchange failed on the device");
    return PLO_OK;
}

}  // extern "C"
