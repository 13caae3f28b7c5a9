// bam_internal.hpp -- shared internals of the host-side BAM code (bam_host.cpp, phase1.cpp): BGZF input, record access, aux
// walking, CIGAR helpers.  Everything sits in an unnamed namespace (one private copy per translation unit).
#pragma once
#include <errno.h>
#include <dlfcn.h>
#include <fcntl.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <mutex>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "../../include/portello_bam.h"

void plo_bam_set_error(const std::string &msg);
// engine.hip: all BGZF blocks of a chunk inflated on the device (one thread per block); < 0 = not done, inflate on the host
extern "C" uint32_t plo_internal_bgzf_slots(void);  // blocks the device inflates at once (one per resident wave); 0 without a device
extern "C" void plo_internal_bgzf_acquire(void);
extern "C" void plo_internal_bgzf_release(void);
extern "C" int plo_internal_bgzf_set_device(int dev);
extern "C" int plo_internal_bgzf_begin(int slot, const uint8_t *comp, size_t comp_bytes, const void *blks, uint32_t n, uint8_t *out, size_t out_bytes,
                                       const uint32_t *crcs);
extern "C" int plo_internal_bgzf_wait(int slot);
extern "C" int plo_internal_bgzf_inflate(const uint8_t *comp, size_t comp_bytes, const void *blks, uint32_t n, uint8_t *out, size_t out_bytes);  // bam_host.cpp: the message plo_bam_last_error() returns (per thread)

namespace {

plo_status fail(plo_status st, const std::string &msg) {
    plo_bam_set_error(msg);
    return st;
}

// fork-join over [0, n): `threads` workers take indices from a shared counter
template <class F>
void parallel_for(size_t n, int threads, F fn) {
    if (threads <= 1 || n <= 1) {
        for (size_t i = 0; i < n; ++i) fn(i);
        return;
    }
    std::atomic<size_t> next{0};
    auto body = [&]() {
        for (;;) {
            size_t i = next.fetch_add(1);
            if (i >= n) break;
            fn(i);
        }
    };
    std::vector<std::thread> th;
    size_t nt = std::min<size_t>((size_t)threads, n);
    for (size_t t = 1; t < nt; ++t) th.emplace_back(body);
    body();
    for (auto &t : th) t.join();
}
// the same over contiguous ranges (per-thread state: one z_stream per range)
template <class F>
void parallel_ranges(size_t n, int threads, F fn) {
    size_t nt = std::max<size_t>(1, std::min<size_t>((size_t)std::max(1, threads), n));
    if (nt <= 1) {
        fn(0, n);
        return;
    }
    std::vector<std::thread> th;
    for (size_t t = 0; t < nt; ++t) {
        size_t lo = n * t / nt, hi = n * (t + 1) / nt;
        if (t + 1 < nt) th.emplace_back(fn, lo, hi);
        else fn(lo, hi);
    }
    for (auto &t : th) t.join();
}

// Large blocks of pageable host memory are RECYCLED process-wide: a window's record bytes and its output records are a few hundred MB each,
// and taking them from malloc means an mmap, a page fault and a kernel-side clear per 4 KB page on first touch and an munmap at the end
// -- about 40 ms per block on the GPU box, in the reader's and the writer's threads (the writer's "busy" time was more than half
// frees).  Blocks of at least BIG_BLOCK bytes come from / return to a small free list (first fit within 2 x the size asked for).
namespace bigpool {
constexpr size_t BIG_BLOCK = (size_t)1 << 20;
constexpr size_t KEEP_BYTES = (size_t)6 << 30;  // most the list keeps (the rest is freed)
struct Blk {
    uint8_t *p;
    size_t cap;
};
inline std::mutex &mu() {
    static std::mutex m;
    return m;
}
inline std::vector<Blk> &list() {
    static std::vector<Blk> l;
    return l;
}
inline size_t &kept() {
    static size_t k = 0;
    return k;
}
inline uint8_t *take(size_t want, size_t &cap) {
    if (want >= BIG_BLOCK) {
        std::lock_guard<std::mutex> g(mu());
        auto &l = list();
        size_t best = l.size();
        for (size_t i = 0; i < l.size(); ++i)
            if (l[i].cap >= want && l[i].cap <= 2 * want && (best == l.size() || l[i].cap < l[best].cap)) best = i;
        if (best < l.size()) {
            Blk b = l[best];
            l.erase(l.begin() + (long)best);
            kept() -= b.cap;
            cap = b.cap;
            return b.p;
        }
    }
    cap = want >= BIG_BLOCK ? (want + (want >> 3) + 4095) & ~(size_t)4095 : want;  // an eighth of headroom: windows of a run differ a little
    return (uint8_t *)malloc(cap ? cap : 1);
}
inline void give(uint8_t *p, size_t cap) {
    if (!p) return;
    if (cap >= BIG_BLOCK) {
        std::lock_guard<std::mutex> g(mu());
        if (kept() + cap <= KEEP_BYTES) {
            list().push_back({p, cap});
            kept() += cap;
            return;
        }
    }
    free(p);
}
}  // namespace bigpool

// growable byte buffer that does not zero what it allocates (the inflated stream and the output records are written once, in
// parallel, right after the allocation)
struct RawBuf {
    uint8_t *p = nullptr;
    size_t n = 0, cap = 0;
    bool pinned = false;  // page-locked (plo_host_alloc): the destination of device-to-host copies
    RawBuf() = default;
    RawBuf(const RawBuf &) = delete;
    RawBuf &operator=(const RawBuf &) = delete;
    ~RawBuf() {
        if (pinned) plo_host_free(p);
        else bigpool::give(p, cap);
    }
    uint8_t *data() { return p; }
    const uint8_t *data() const { return p; }
    size_t size() const { return n; }
    bool resize(size_t want) {  // keeps the first min(n, want) bytes
        if (want > cap) {
            size_t nc = std::max(want, cap + cap / 2);
            if (pinned) {
                void *q = nullptr;
                if (plo_host_alloc(nc ? nc : 1, &q) != PLO_OK || !q) return false;
                if (n) memcpy(q, p, std::min(n, want));
                plo_host_free(p);
                p = (uint8_t *)q;
            } else {
                size_t got = 0;
                uint8_t *q = bigpool::take(nc ? nc : 1, got);
                if (!q) return false;
                if (n) memcpy(q, p, std::min(n, want));
                bigpool::give(p, cap);
                p = q;
                nc = got;
            }
            cap = nc;
        }
        n = want;
        return true;
    }
};
void parallel_copy(uint8_t *dst, const uint8_t *src, size_t n, int threads);
// n bytes of the file behind `fd` from offset `off` straight into dst (page-locked staging of the device inflate): positional reads, no
// page faults on a file mapping and no second copy; false when the file ends early
bool parallel_pread(int fd, uint8_t *dst, size_t off, size_t n, int threads);

inline uint16_t rd16(const uint8_t *p) { return (uint16_t)(p[0] | (p[1] << 8)); }
inline uint32_t rd32(const uint8_t *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
inline int32_t rdi32(const uint8_t *p) { return (int32_t)rd32(p); }
inline void wr16(uint8_t *p, uint16_t v) {
    p[0] = (uint8_t)v;
    p[1] = (uint8_t)(v >> 8);
}
inline void wr32(uint8_t *p, uint32_t v) {
    p[0] = (uint8_t)v;
    p[1] = (uint8_t)(v >> 8);
    p[2] = (uint8_t)(v >> 16);
    p[3] = (uint8_t)(v >> 24);
}

// DEFLATE engine: zlib always works; libdeflate (whole-buffer API, 2-3x faster inflate, CRC with carry-less multiplies) is
// used when its shared library is on the machine.  Only the runtime is present in this image (no header), so the five entry
// points are bound by name -- their signatures are libdeflate's stable public API.
struct LibDeflate {
    void *(*alloc_decompressor)() = nullptr;
    int (*deflate_decompress)(void *, const void *, size_t, void *, size_t, size_t *) = nullptr;
    void (*free_decompressor)(void *) = nullptr;
    void *(*alloc_compressor)(int) = nullptr;
    size_t (*deflate_compress)(void *, const void *, size_t, void *, size_t) = nullptr;
    void (*free_compressor)(void *) = nullptr;
    uint32_t (*crc32)(uint32_t, const void *, size_t) = nullptr;
    bool ok = false;
    LibDeflate() {
        if (getenv("PLO_NO_LIBDEFLATE")) return;
        void *h = nullptr;
        for (const char *n : {"libdeflate.so.0", "libdeflate.so", "libdeflate.so.1"})
            if ((h = dlopen(n, RTLD_NOW | RTLD_LOCAL))) break;
        if (!h) return;
        alloc_decompressor = (void *(*)())dlsym(h, "libdeflate_alloc_decompressor");
        deflate_decompress = (int (*)(void *, const void *, size_t, void *, size_t, size_t *))dlsym(h, "libdeflate_deflate_decompress");
        free_decompressor = (void (*)(void *))dlsym(h, "libdeflate_free_decompressor");
        alloc_compressor = (void *(*)(int))dlsym(h, "libdeflate_alloc_compressor");
        deflate_compress = (size_t (*)(void *, const void *, size_t, void *, size_t))dlsym(h, "libdeflate_deflate_compress");
        free_compressor = (void (*)(void *))dlsym(h, "libdeflate_free_compressor");
        crc32 = (uint32_t (*)(uint32_t, const void *, size_t))dlsym(h, "libdeflate_crc32");
        ok = alloc_decompressor && deflate_decompress && free_decompressor && alloc_compressor && deflate_compress && free_compressor && crc32;
    }
};
const LibDeflate &libdeflate() {
    static const LibDeflate *l = new LibDeflate();
    return *l;
}
inline uint32_t fast_crc32(const uint8_t *p, size_t n) {
    const LibDeflate &ld = libdeflate();
    return ld.ok ? ld.crc32(0, p, n) : (uint32_t)crc32(0L, p, (uInt)n);
}

// ---------------------------------------------------------------------------------------------------------------------
// BGZF input: the file is mapped; blocks are located by their BSIZE fields and inflated in parallel, a chunk at a time
// ---------------------------------------------------------------------------------------------------------------------
struct BgzfIn {
    int fd = -1;
    const uint8_t *map = nullptr;
    size_t size = 0, cpos = 0;
    RawBuf buf;  // inflated bytes not yet consumed: [bpos, buf.size())
    RawBuf cstage3[3];  // device inflate: page-locked copies of three groups' compressed bytes (two on the device, the third being staged)
    uint32_t stage_base = 0;  // staging buffer of a refill's group g: (stage_base + g) % 3
    // TEST HOOK (tests/test_bam.py, CPU): PLO_BGZF_TEST_HOST_SLOTS=n runs the device path's bookkeeping -- groups of n blocks, three staging
    // buffers, refill preparation -- with zlib standing in for the device AT WAIT TIME (a group is inflated from its staging buffer when
    // finish() is called, so a buffer that is reused too early shows up as wrong bytes); PLO_BGZF_TEST_CHUNK_BYTES = the refill size.
    // Never set outside the tests: the page-locked buffers and the kernels are what the device path is for.
    int test_slots = -1;  // -1 undecided, 0 off
    size_t test_chunk = 0;
    int test_mode() {
        if (test_slots < 0) {
            const char *e = getenv("PLO_BGZF_TEST_HOST_SLOTS");
            test_slots = e ? std::max(0, atoi(e)) : 0;
            const char *c2 = getenv("PLO_BGZF_TEST_CHUNK_BYTES");
            test_chunk = c2 ? (size_t)std::max(1L, atol(c2)) : 0;
        }
        return test_slots;
    }
    // The NEXT refill, prepared while the last groups of this one are on the device and the host would only wait (device inflate): its block
    // headers walked (a page fault per block on the file mapping) and its first one or two groups staged.  Used by the next fill() when the
    // stream is still where the walk began (restart_at drops it).  PLO_BGZF_NO_PREFETCH=1 switches it off.
    struct NextBlk {
        size_t coff, clen, urel, ulen, fpos;
        uint32_t crc;
    };
    std::vector<NextBlk> nx_blks;
    size_t nx_pos0 = 0, nx_pos_end = 0;
    uint32_t nx_staged = 0;
    bool nx_valid = false;
    size_t bpos = 0;
    int threads = 1;
    bool eof = false;
    uint32_t dev_slots = 0;
    int dev_id = 0;  // HIP device of the device inflate
    int device = -1;  // -1 undecided (environment), -2 undecided (requested), 0 host inflate, 1 blocks are inflated on the GPU
    static constexpr size_t CHUNK = 256u << 20;
    // Inflated bytes a refill adds.  On the device a refill's groups of blocks run two at a time and the pair in flight drains at the end of
    // every refill: the first refill stays at CHUNK (the first window is out after one group's latency), the later ones take `later_chunk`
    // (PLO_BGZF_CHUNK_MB), so that the drain is paid once per gigabyte instead of once per two groups.
    size_t later_chunk = 0, n_fills = 0;
    size_t chunk_now() {
        if (!later_chunk) {
            const char *e = getenv("PLO_BGZF_CHUNK_MB");
            const long mb = e ? atol(e) : 0;
            later_chunk = mb >= 64 && mb <= 8192 ? (size_t)mb << 20 : (device == 1 ? DEVICE_CHUNK : CHUNK);
        }
        if (test_mode() && test_chunk) return test_chunk;
        return n_fills == 0 ? CHUNK : later_chunk;
    }
    static constexpr size_t DEVICE_CHUNK = 256u << 20;
    // A PART of the file (plo_bam_open_range): records whose first byte lies in a BGZF block that starts in [range_lo, range_end) of the
    // compressed file.  `blkmap`: where in `buf` every block of the current contents starts and where that block starts in the file.
    size_t range_end = (size_t)-1;
    bool ranged = false;
    std::vector<std::pair<size_t, size_t>> blkmap;  // (offset in buf, offset of the block in the file), ascending
    // file offset of the BGZF block that holds byte `at` of the unconsumed stream (ranged readers only)
    size_t block_file_off(size_t at) const {
        const size_t u = bpos + at;
        size_t lo = 0, hi = blkmap.size();
        while (lo + 1 < hi) {
            const size_t m = (lo + hi) / 2;
            if (blkmap[m].first <= u) lo = m;
            else hi = m;
        }
        return blkmap.empty() ? 0 : blkmap[lo].second;
    }
    // a BGZF block header at `h` (at least 18 readable bytes)?  returns the block's size or 0
    static uint32_t bgzf_block_at(const uint8_t *h, size_t left) {
        if (left < 28 || h[0] != 0x1f || h[1] != 0x8b || h[2] != 8 || !(h[3] & 4)) return 0;
        const uint32_t xlen = rd16(h + 10);
        if (12 + (size_t)xlen > left) return 0;
        uint32_t bsize = 0;
        for (uint32_t x = 0; x + 4 <= xlen;) {
            const uint8_t *e = h + 12 + x;
            const uint32_t slen = rd16(e + 2);
            if (e[0] == 'B' && e[1] == 'C' && slen == 2 && x + 6 <= xlen) bsize = (uint32_t)rd16(e + 4) + 1;
            x += 4 + slen;
        }
        if (bsize < 12 + xlen + 8 || bsize > left) return 0;
        return bsize;
    }
    // The first BGZF block that starts at or after file offset `from`: the magic and a BC field are not proof (compressed data can hold
    // the same bytes), a chain of three blocks that follow each other -- or end the file -- is.  Returns `size` when there is none.
    size_t seek_block(size_t from) const {
        for (size_t p = from; p + 28 <= size; ++p) {
            if (map[p] != 0x1f || map[p + 1] != 0x8b) continue;
            size_t q = p;
            int ok = 0;
            while (ok < 3 && q < size) {
                const uint32_t b = bgzf_block_at(map + q, size - q);
                if (!b) break;
                q += b;
                ++ok;
            }
            if (ok == 3 || (ok > 0 && q == size)) return p;
        }
        return size;
    }
    // continue the stream at the block that starts at file offset `at` (everything buffered is dropped)
    void restart_at(size_t at) {
        buf.resize(0);
        bpos = 0;
        blkmap.clear();
        cpos = at;
        eof = cpos >= size;
        nx_valid = false;
        nx_staged = 0;
    }
    struct DevBlk {  // engine.hip's BgzfBlk
        unsigned long long coff, uoff;
        uint32_t clen, ulen;
    };

    struct Blk {
        size_t coff, clen, uoff, ulen;
        uint32_t crc;
    };

    plo_status open(const char *path, int nt) {
        threads = std::max(1, nt);
        fd = ::open(path, O_RDONLY);
        if (fd < 0) return fail(PLO_ERR_IO, std::string("cannot open ") + path);
        struct stat st;
        if (fstat(fd, &st) != 0) return fail(PLO_ERR_IO, "fstat failed");
        size = (size_t)st.st_size;
        if (size) {
            void *p = mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0);
            if (p == MAP_FAILED) return fail(PLO_ERR_IO, "mmap failed");
            map = (const uint8_t *)p;
            madvise(p, size, MADV_SEQUENTIAL);
        }
        return PLO_OK;
    }
    void close() {
        if (map) munmap((void *)map, size);
        if (fd >= 0) ::close(fd);
        map = nullptr;
        fd = -1;
    }
    size_t avail() const { return buf.size() - bpos; }

    // makes at least `want` bytes available (fewer only at the end of the file)
    plo_status fill(size_t want) {
        if (avail() >= want || eof) return PLO_OK;
        const bool dbgf = getenv("PLO_DEBUG_READER") != nullptr;  // where a refill's time goes: tail move / header walk / buffer / staging / waits
        auto clk = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
        const double tf0 = dbgf ? clk() : 0;
        double tf_stage = 0, tf_wait = 0;
        if (bpos) {
            // (the unconsumed tail to the front: with several threads when it does not overlap its new place -- the usual case, a window
            // has just been cut off the front -- it used to be a single-threaded memmove of up to a refill's 256 MB)
            const size_t left = buf.size() - bpos;
            if (left <= bpos && left > ((size_t)8 << 20)) parallel_copy(buf.data(), buf.data() + bpos, left, threads);
            else memmove(buf.data(), buf.data() + bpos, left);
            buf.resize(buf.size() - bpos);
            if (ranged) {  // the blocks' places move with the bytes; blocks consumed whole are forgotten (the one that holds byte 0 stays)
                size_t keep = 0;
                while (keep + 1 < blkmap.size() && blkmap[keep + 1].first <= bpos) ++keep;
                blkmap.erase(blkmap.begin(), blkmap.begin() + (long)keep);
                for (auto &e : blkmap) e.first = e.first > bpos ? e.first - bpos : 0;
            }
            bpos = 0;
        }
        const double tf1 = dbgf ? clk() : 0;
        const size_t cpos0 = cpos;
        if (device < 0) {
            // opt-in (PLO_BGZF_DEVICE=1): measured on MI355X the kernel inflates 20 GB/s, three times what 16 host cores do with
            // libdeflate -- but staging the compressed bytes in page-locked memory and checking the CRCs afterwards cost those cores
            // nearly as much as inflating: pipelined with the device, a refill takes as long as on the host
            const char *e = getenv("PLO_BGZF_DEVICE");  // the environment has the last word; -2: asked for through plo_bam_set_device_inflate
            device = e ? (atoi(e) != 0 ? 1 : 0) : (device == -2 ? 1 : 0);
            if (device && test_mode()) {
                // (test hook: no page-locked memory, no device)
            } else if (device) {  // page-locked stream buffer; without a usable device the allocation fails and the host path stays
                void *q = nullptr;
                const size_t cap0 = std::max<size_t>(CHUNK + CHUNK / 2, buf.n);
                if (buf.pinned) {
                    // (already page-locked)
                } else if (plo_host_alloc(cap0, &q) == PLO_OK && q) {
                    if (buf.n) memcpy(q, buf.p, buf.n);  // (switched on after the header was read)
                    free(buf.p);
                    buf.p = (uint8_t *)q;
                    buf.cap = cap0;
                    buf.pinned = true;
                } else {
                    device = 0;
                }
            }
        }
        std::vector<Blk> blks;
        size_t u = buf.size();
        // always a whole chunk of NEW data: a window larger than one chunk must not degenerate into one small refill (thread
        // start-up, or a device round trip) per record
        const size_t target = std::max(want, u + chunk_now());
        ++n_fills;
        // device inflate: whole rounds of resident waves -- a group takes one block's decode time (~7 ms) however few blocks it has, so
        // the refill goes on to the next multiple of the wave slots (17 blocks left over used to cost a round of their own)
        if (device == 1 && !dev_slots) dev_slots = test_mode() ? (uint32_t)test_mode() : std::max<uint32_t>(plo_internal_bgzf_slots(), 64u);
        const size_t round_to = device == 1 ? dev_slots : 1;
        uint32_t pre_staged = 0;  // leading groups of this refill whose compressed bytes are in their staging buffers already
        if (nx_valid && device == 1 && nx_pos0 == cpos && !nx_blks.empty()) {
            for (const NextBlk &nb : nx_blks) {
                Blk b;
                b.coff = nb.coff;
                b.clen = nb.clen;
                b.crc = nb.crc;
                b.ulen = nb.ulen;
                b.uoff = u + nb.urel;
                if (ranged && b.ulen) blkmap.emplace_back(b.uoff, nb.fpos);
                blks.push_back(b);
            }
            u += nx_blks.back().urel + nx_blks.back().ulen;
            cpos = nx_pos_end;
            pre_staged = nx_staged;
        }
        nx_valid = false;
        nx_staged = 0;
        while (cpos < size && (u < target || blks.size() % round_to != 0)) {
            if (size - cpos < 28) return fail(PLO_ERR_IO, "truncated BGZF block header");
            const uint8_t *h = map + cpos;
            if (h[0] != 0x1f || h[1] != 0x8b || h[2] != 8 || !(h[3] & 4)) return fail(PLO_ERR_IO, "not a BGZF block");
            uint32_t xlen = rd16(h + 10), bsize = 0;
            if (12 + (size_t)xlen > size - cpos) return fail(PLO_ERR_IO, "truncated BGZF extra field");
            for (uint32_t x = 0; x + 4 <= xlen;) {
                const uint8_t *e = h + 12 + x;
                uint32_t slen = rd16(e + 2);
                if (e[0] == 'B' && e[1] == 'C' && slen == 2 && x + 6 <= xlen) bsize = (uint32_t)rd16(e + 4) + 1;
                x += 4 + slen;
            }
            if (bsize < 12 + xlen + 8 || bsize > size - cpos) return fail(PLO_ERR_IO, "corrupt or truncated BGZF block");
            Blk b;
            b.coff = cpos + 12 + xlen;
            b.clen = bsize - 12 - xlen - 8;
            b.crc = rd32(h + bsize - 8);
            b.ulen = rd32(h + bsize - 4);
            b.uoff = u;
            if (b.ulen > 65536) return fail(PLO_ERR_IO, "BGZF block larger than 64 KiB");
            if (ranged && b.ulen) blkmap.emplace_back(u, cpos);
            u += b.ulen;
            cpos += bsize;
            blks.push_back(b);
        }
        const double tf2 = dbgf ? clk() : 0;
        if (cpos >= size) eof = true;
        if (cpos > cpos0) madvise((void *)(map + (cpos0 & ~(size_t)4095)), cpos - (cpos0 & ~(size_t)4095), MADV_WILLNEED);
        if (!buf.resize(u)) return fail(PLO_ERR_OUT_OF_MEMORY, "out of host memory for the inflated BAM stream");
        const double tf3 = dbgf ? clk() : 0;
        std::atomic<int> bad{0};
        if (device == 1 && !blks.empty()) {
            // On the GPU, in groups of one block per resident wave (a group of exactly that many has no second, nearly empty round),
            // two groups in flight: while the device inflates group g the host stages the compressed bytes of group g + 1 in
            // page-locked memory (a large host-to-device copy straight from the file mapping would make the runtime pin file-backed
            // pages, far slower than this parallel copy) and checks the CRCs of group g - 1 (libdeflate's CRC runs at memory speed).
            if (!dev_slots) dev_slots = test_mode() ? (uint32_t)test_mode() : std::max<uint32_t>(plo_internal_bgzf_slots(), 64u);
            const bool test = test_mode() != 0;
            const size_t ng = (blks.size() + dev_slots - 1) / dev_slots;
            std::vector<DevBlk> db[2];
            std::vector<uint32_t> dcrc[2];
            // PLO_BGZF_HOST_CRC=1: the CRCs are checked on the host as in round 4 (the device's check is then skipped)
            const bool host_crc = getenv("PLO_BGZF_HOST_CRC") != nullptr;
            int rc = 0;
            // stage(g): the group's compressed bytes into page-locked buffer g % 3 -- host work only, done one group AHEAD of the device slots
            // (while groups g - 2 and g - 1 are on the device; round 5 staged inside begin(), between two waits: the device idled for it);
            // enqueue(g): upload, kernels and download of a staged group on slot g & 1
            auto stage = [&](size_t g) -> int {
                const size_t lo = g * dev_slots, hi = std::min(blks.size(), lo + dev_slots);
                const size_t c0 = blks[lo].coff;
                const size_t cbytes = blks[hi - 1].coff + blks[hi - 1].clen - c0;
                if (g < pre_staged) return 0;  // (staged while the last refill's final groups were on the device)
                RawBuf &cs = cstage3[(stage_base + g) % 3];
                cs.pinned = !test;
                if (!cs.resize(cbytes + 16)) return -101;
                // compressed bytes: positional reads straight into the page-locked stage (round 4 copied them out of the file mapping)
                const double ts0 = dbgf ? clk() : 0;
                if (!parallel_pread(fd, cs.data(), c0, cbytes, threads)) parallel_copy(cs.data(), map + c0, cbytes, threads);
                if (dbgf) tf_stage += clk() - ts0;
                return 0;
            };
            auto enqueue = [&](size_t g) -> int {
                const size_t lo = g * dev_slots, hi = std::min(blks.size(), lo + dev_slots);
                const size_t c0 = blks[lo].coff, u0 = blks[lo].uoff;
                std::vector<DevBlk> &d = db[g & 1];
                d.resize(hi - lo);
                std::vector<uint32_t> &dc = dcrc[g & 1];
                dc.resize(hi - lo);
                for (size_t i = lo; i < hi; ++i) {
                    d[i - lo] = DevBlk{blks[i].coff - c0, blks[i].uoff - u0, (uint32_t)blks[i].clen, (uint32_t)blks[i].ulen};
                    dc[i - lo] = blks[i].crc;
                }
                const size_t cbytes = blks[hi - 1].coff + blks[hi - 1].clen - c0, ubytes = blks[hi - 1].uoff + blks[hi - 1].ulen - u0;
                if (test) return 0;  // (the stand-in inflates at wait time)
                return plo_internal_bgzf_begin((int)(g & 1), cstage3[(stage_base + g) % 3].data(), cbytes, d.data(), (uint32_t)d.size(), buf.data() + u0, ubytes, host_crc ? nullptr : dc.data());
            };
            auto finish = [&](size_t g) -> int {
                const double tw0 = dbgf ? clk() : 0;
                if (test) {  // zlib from the group's STAGING buffer, as the device would read it until now
                    const size_t lo = g * dev_slots, hi = std::min(blks.size(), lo + dev_slots);
                    const uint8_t *cs = cstage3[(stage_base + g) % 3].data();
                    const size_t c0 = blks[lo].coff;
                    for (size_t i = lo; i < hi; ++i) {
                        const Blk &b = blks[i];
                        if (!b.ulen) continue;
                        z_stream zs;
                        memset(&zs, 0, sizeof(zs));
                        if (inflateInit2(&zs, -15) != Z_OK) return -200;
                        zs.next_in = (Bytef *)(cs + (b.coff - c0));
                        zs.avail_in = (uInt)b.clen;
                        zs.next_out = buf.data() + b.uoff;
                        zs.avail_out = (uInt)b.ulen;
                        const int zr = inflate(&zs, Z_FINISH);
                        const bool good = zr == Z_STREAM_END && zs.avail_out == 0 && (uint32_t)crc32(0L, buf.data() + b.uoff, (uInt)b.ulen) == b.crc;
                        inflateEnd(&zs);
                        if (!good) return -200;
                    }
                    return 0;
                }
                int r = plo_internal_bgzf_wait((int)(g & 1));
                if (dbgf) tf_wait += clk() - tw0;
                if (r) return r;
                if (!host_crc) return 0;  // (k_bgzf_crc has checked every block: a mismatch came back as the group's status)
                const size_t lo = g * dev_slots, hi = std::min(blks.size(), lo + dev_slots);
                parallel_for(hi - lo, threads, [&](size_t i) {
                    const Blk &b = blks[lo + i];
                    if (b.ulen && fast_crc32(buf.data() + b.uoff, b.ulen) != b.crc) bad = 1;
                });
                return 0;
            };
            if (!test) {
                plo_internal_bgzf_acquire();
                rc = plo_internal_bgzf_set_device(dev_id);
            }
            // the next refill's headers and first groups, in the time the host would wait for this refill's last groups
            double tf_pre = 0;
            auto prestage_upto = [&](size_t kmax) {
                if (cpos >= size || getenv("PLO_BGZF_NO_PREFETCH")) return;
                const double tp0 = dbgf ? clk() : 0;
                if (!nx_valid) {
                    nx_blks.clear();
                    nx_pos0 = cpos;
                    nx_staged = 0;
                    size_t pos = cpos, ur = 0;
                    const size_t chunk = chunk_now();
                    bool ok = true;
                    while (pos < size && (ur < chunk || nx_blks.size() % dev_slots != 0)) {
                        if (size - pos < 28) { ok = false; break; }
                        const uint8_t *h = map + pos;
                        if (h[0] != 0x1f || h[1] != 0x8b || h[2] != 8 || !(h[3] & 4)) { ok = false; break; }
                        const uint32_t xlen = rd16(h + 10);
                        uint32_t bsize = 0;
                        if (12 + (size_t)xlen > size - pos) { ok = false; break; }
                        for (uint32_t x = 0; x + 4 <= xlen;) {
                            const uint8_t *e = h + 12 + x;
                            const uint32_t slen = rd16(e + 2);
                            if (e[0] == 'B' && e[1] == 'C' && slen == 2 && x + 6 <= xlen) bsize = (uint32_t)rd16(e + 4) + 1;
                            x += 4 + slen;
                        }
                        if (bsize < 12 + xlen + 8 || bsize > size - pos) { ok = false; break; }
                        NextBlk nb;
                        nb.coff = pos + 12 + xlen;
                        nb.clen = bsize - 12 - xlen - 8;
                        nb.crc = rd32(h + bsize - 8);
                        nb.ulen = rd32(h + bsize - 4);
                        nb.urel = ur;
                        nb.fpos = pos;
                        if (nb.ulen > 65536) { ok = false; break; }
                        ur += nb.ulen;
                        pos += bsize;
                        nx_blks.push_back(nb);
                    }
                    if (!ok || nx_blks.empty()) {  // (an irregular block: the next fill() walks the headers itself and reports it)
                        nx_blks.clear();
                        if (dbgf) tf_pre += clk() - tp0;
                        return;
                    }
                    nx_pos_end = pos;
                    nx_valid = true;
                }
                while (nx_staged <= kmax) {
                    const size_t lo = (size_t)nx_staged * dev_slots;
                    if (lo >= nx_blks.size()) break;
                    const size_t hi = std::min(nx_blks.size(), lo + dev_slots);
                    if (hi - lo < dev_slots && nx_pos_end < size) break;  // (only groups that are final: full, or cut by the end of the file)
                    const size_t c0 = nx_blks[lo].coff, cbytes = nx_blks[hi - 1].coff + nx_blks[hi - 1].clen - c0;
                    RawBuf &cs = cstage3[(stage_base + ng + nx_staged) % 3];
                    cs.pinned = !test;
                    if (!cs.resize(cbytes + 16)) break;
                    if (!parallel_pread(fd, cs.data(), c0, cbytes, threads)) parallel_copy(cs.data(), map + c0, cbytes, threads);
                    ++nx_staged;
                }
                if (dbgf) tf_pre += clk() - tp0;
            };
            // groups g and g + 1 on the device, g + 2 staged while they run: enqueue(g + 2) follows finish(g) at once
            if (rc == 0) rc = stage(0);
            if (rc == 0) rc = enqueue(0);
            if (rc == 0 && ng > 1) rc = stage(1);
            if (rc == 0 && ng > 1) rc = enqueue(1);
            for (size_t g = 0; g < ng && rc == 0; ++g) {
                int rs = 0;
                if (g + 2 < ng) rs = stage(g + 2);
                else prestage_upto(std::min<size_t>(1, g + 2 - ng));  // (buffer of next group k: free once this refill's group ng + k - 3 is done, k <= g + 2 - ng)
                rc = finish(g);
                if (rc == 0) rc = rs;
                if (rc == 0 && g + 2 < ng) rc = enqueue(g + 2);
            }
            if (rc != 0 && !test) {  // leave nothing in flight
                (void)plo_internal_bgzf_wait(0);
                (void)plo_internal_bgzf_wait(1);
            }
            if (!test) plo_internal_bgzf_release();
            if (rc != 0) {  // (what was prepared lies in buffers counted from this refill's groups: dropped with it)
                nx_valid = false;
                nx_staged = 0;
            }
            stage_base = (uint32_t)((stage_base + ng) % 3);
            if (dbgf)
                fprintf(stderr, "[plo] refill: %zu blocks in %zu groups, %.3f s: tail move %.4f, header walk %.4f, buffer %.4f, staging %.4f, waits %.4f, rest %.4f (next refill prepared in %.4f, %u groups staged)\n", blks.size(), ng,
                        clk() - tf0, tf1 - tf0, tf2 - tf1, tf3 - tf2, tf_stage, tf_wait, (clk() - tf3) - tf_stage - tf_wait, tf_pre, nx_staged);
            if (rc == 0) {
                if (bad) return fail(PLO_ERR_IO, "BGZF block CRC mismatch after device inflate");
                return PLO_OK;
            }
            bad = 0;
            if (rc <= -100 && rc > -200) {  // no usable device: stay on the host from now on
                device = 0;
                if (getenv("PLO_DEBUG_READER")) fprintf(stderr, "[plo] reader: device inflate unavailable on device %d (code %d), inflating on the host\n", dev_id, rc);
            }
            // a block the device rejects is inflated again on the host, which reports what is wrong with it
        }
        parallel_ranges(blks.size(), threads, [&](size_t lo, size_t hi) {
            const LibDeflate &ld = libdeflate();
            if (ld.ok) {
                void *d = ld.alloc_decompressor();
                if (!d) {
                    bad = 1;
                    return;
                }
                for (size_t i = lo; i < hi; ++i) {
                    const Blk &b = blks[i];
                    if (b.ulen == 0) continue;
                    size_t got = 0;
                    int rc = ld.deflate_decompress(d, map + b.coff, b.clen, buf.data() + b.uoff, b.ulen, &got);
                    if (rc != 0 || got != b.ulen || ld.crc32(0, buf.data() + b.uoff, b.ulen) != b.crc) bad = 1;
                }
                ld.free_decompressor(d);
                return;
            }
            z_stream zs;
            memset(&zs, 0, sizeof(zs));
            if (inflateInit2(&zs, -15) != Z_OK) {
                bad = 1;
                return;
            }
            for (size_t i = lo; i < hi; ++i) {
                const Blk &b = blks[i];
                if (b.ulen == 0) continue;
                inflateReset(&zs);
                zs.next_in = (Bytef *)(map + b.coff);
                zs.avail_in = (uInt)b.clen;
                zs.next_out = buf.data() + b.uoff;
                zs.avail_out = (uInt)b.ulen;
                int rc = inflate(&zs, Z_FINISH);
                if (rc != Z_STREAM_END || zs.avail_out != 0 || (uint32_t)crc32(0L, buf.data() + b.uoff, (uInt)b.ulen) != b.crc) bad = 1;
            }
            inflateEnd(&zs);
        });
        if (bad) return fail(PLO_ERR_IO, "BGZF block failed to inflate (corrupt data or CRC mismatch)");
        return PLO_OK;
    }
    plo_status read(void *dst, size_t n) {
        plo_status st = fill(n);
        if (st != PLO_OK) return st;
        if (avail() < n) return fail(PLO_ERR_IO, "unexpected end of BAM stream");
        memcpy(dst, buf.data() + bpos, n);
        bpos += n;
        return PLO_OK;
    }
};

bool parallel_pread(int fd, uint8_t *dst, size_t off, size_t n, int threads) {
    const size_t piece = 4u << 20;
    const size_t np = (n + piece - 1) / piece;
    std::atomic<int> bad{0};
    parallel_for(np, std::min<int>(threads, 16), [&](size_t i) {
        size_t o = i * piece, left = std::min(piece, n - o);
        while (left) {
            const ssize_t k = pread(fd, dst + o, left, (off_t)(off + o));
            if (k < 0 && errno == EINTR) continue;
            if (k <= 0) {
                bad = 1;
                return;
            }
            o += (size_t)k;
            left -= (size_t)k;
        }
    });
    return bad == 0;
}
void parallel_copy(uint8_t *dst, const uint8_t *src, size_t n, int threads) {
    const size_t piece = 4u << 20;
    const size_t np = (n + piece - 1) / piece;
    parallel_for(np, std::min<int>(threads, 16), [&](size_t i) {
        size_t o = i * piece;
        memcpy(dst + o, src + o, std::min(piece, n - o));
    });
}

// Page-locked when the engine's allocator has a device, plain memory otherwise.  Pinning costs far more than the copy it
// speeds up, so page-locked blocks go back to a process-wide pool when a window is freed and the next window reuses them.
struct PinPool {
    std::mutex mu;
    std::vector<std::pair<void *, size_t>> blocks;
    void *take(size_t want, size_t &cap) {
        std::lock_guard<std::mutex> g(mu);
        size_t best = blocks.size();
        for (size_t i = 0; i < blocks.size(); ++i)
            if (blocks[i].second >= want && blocks[i].second <= 4 * want + (1u << 20) && (best == blocks.size() || blocks[i].second < blocks[best].second)) best = i;
        if (best == blocks.size()) return nullptr;
        void *p = blocks[best].first;
        cap = blocks[best].second;
        blocks.erase(blocks.begin() + (ptrdiff_t)best);
        return p;
    }
    void give(void *p, size_t cap) {
        std::lock_guard<std::mutex> g(mu);
        if (blocks.size() >= 256) {  // bounded: drop the smallest
            size_t k = 0;
            for (size_t i = 1; i < blocks.size(); ++i)
                if (blocks[i].second < blocks[k].second) k = i;
            if (blocks[k].second >= cap) {
                plo_host_free(p);
                return;
            }
            plo_host_free(blocks[k].first);
            blocks.erase(blocks.begin() + (ptrdiff_t)k);
        }
        blocks.emplace_back(p, cap);
    }
};
PinPool &pin_pool() {
    static PinPool *p = new PinPool();  // never destroyed: the HIP runtime may be gone by the time static destructors run
    return *p;
}
struct HostBuf {
    void *p = nullptr;
    size_t cap = 0;
    bool pinned = false;
    void *ensure(size_t bytes) {
        if (bytes <= cap) return p;
        release();
        size_t want = bytes + bytes / 8 + 64;
        size_t got = 0;
        void *q = pin_pool().take(want, got);
        if (q) {
            pinned = true;
            want = got;
        } else if (plo_host_alloc(want, &q) == PLO_OK && q) {
            pinned = true;
        } else {
            q = malloc(want);
            pinned = false;
        }
        p = q;
        cap = q ? want : 0;
        return p;
    }
    void release() {
        if (p) {
            if (pinned) pin_pool().give(p, cap);
            else free(p);
        }
        p = nullptr;
        cap = 0;
    }
    ~HostBuf() { release(); }
    template <class T>
    T *as() const {
        return (T *)p;
    }
};

// ---- record access (BAM specification 4.2; all offsets after the block_size word) ------------------------------------
struct Rec {
    const uint8_t *p;  // first byte after block_size
    uint32_t len;      // block_size
    int32_t tid() const { return rdi32(p); }
    int32_t pos() const { return rdi32(p + 4); }
    uint32_t l_qname() const { return p[8]; }
    uint8_t mapq() const { return p[9]; }
    uint16_t bin() const { return rd16(p + 10); }
    uint32_t n_cigar() const { return rd16(p + 12); }
    uint16_t flag() const { return rd16(p + 14); }
    uint32_t l_seq() const { return rd32(p + 16); }
    const uint8_t *qname() const { return p + 32; }
    const uint8_t *cigar() const { return qname() + l_qname(); }
    const uint8_t *seq() const { return cigar() + 4 * (size_t)n_cigar(); }
    const uint8_t *qual() const { return seq() + (l_seq() + 1) / 2; }
    const uint8_t *aux() const { return qual() + l_seq(); }
    const uint8_t *end() const { return p + len; }
    bool layout_ok() const { return len >= 32 && (size_t)(aux() - p) <= len; }
};

// length of the aux field starting at a (tag, type, value), 0 if malformed / beyond e
size_t aux_field_len(const uint8_t *a, const uint8_t *e) {
    if (e - a < 3) return 0;
    size_t n = 0;
    switch (a[2]) {
        case 'A': case 'c': case 'C': n = 1; break;
        case 's': case 'S': n = 2; break;
        case 'i': case 'I': case 'f': n = 4; break;
        case 'd': n = 8; break;
        case 'Z': case 'H': {
            const uint8_t *z = (const uint8_t *)memchr(a + 3, 0, (size_t)(e - a - 3));
            if (!z) return 0;
            n = (size_t)(z - (a + 3)) + 1;
            break;
        }
        case 'B': {
            if (e - a < 8) return 0;
            size_t es;
            switch (a[3]) {
                case 'c': case 'C': es = 1; break;
                case 's': case 'S': es = 2; break;
                case 'i': case 'I': case 'f': es = 4; break;
                default: return 0;
            }
            n = 5 + es * (size_t)rd32(a + 4);
            break;
        }
        default: return 0;
    }
    return (3 + n <= (size_t)(e - a)) ? 3 + n : 0;
}
// first field with the given tag (bam_aux_get), nullptr if absent
const uint8_t *aux_find(const uint8_t *a, const uint8_t *e, const char tag[2], size_t *flen) {
    while (a < e) {
        size_t n = aux_field_len(a, e);
        if (!n) return nullptr;
        if (a[0] == (uint8_t)tag[0] && a[1] == (uint8_t)tag[1]) {
            if (flen) *flen = n;
            return a;
        }
        a += n;
    }
    return nullptr;
}

// ---- CIGAR helpers (lib/rust-vc-utils/src/bam_utils/cigar/mod.rs) --------------------------------------------------------
inline bool op_is_match(uint32_t c) {  // :22-24
    uint32_t t = c & 15u;
    return t == 0 || t == 7 || t == 8;
}
inline uint64_t op_read_len(uint32_t c) {  // get_cigarseg_read_offset, ignore_hard_clip = false (:26-39)
    return ((0x1B3u >> (c & 15u)) & 1u) ? (uint64_t)(c >> 4) : 0;  // M I S H = X
}
inline int64_t op_ref_len(uint32_t c) {  // :41-47
    return ((0x18Du >> (c & 15u)) & 1u) ? (int64_t)(c >> 4) : 0;  // M D N = X
}
// get_read_clip_positions(cigar, false) (:85-118)
void read_clip_positions(const uint32_t *cig, size_t n, uint64_t &start, uint64_t &end, uint64_t &size) {
    uint64_t left = 0, right = 0, read_pos = 0;
    bool left_clip = true;
    for (size_t i = 0; i < n; ++i) {
        uint32_t t = cig[i] & 15u;
        if (t == 4 || t == 5) {
            if (left_clip) left += cig[i] >> 4;
            else right += cig[i] >> 4;
        } else {
            left_clip = false;
        }
        read_pos += op_read_len(cig[i]);
    }
    start = left;
    end = read_pos - right;
    size = read_pos;
}

struct SaSeg {
    uint32_t contig;
    int64_t pos;
    bool fwd;
    uint8_t mapq;
    std::vector<uint32_t> cigar;
    uint64_t so_start, so_end;
    bool primary;
};

bool parse_uint(const char *s, const char *e, uint64_t &v) {
    if (s == e) return false;
    v = 0;
    for (; s < e; ++s) {
        if (*s < '0' || *s > '9') return false;
        v = v * 10 + (uint64_t)(*s - '0');
        if (v > (1ull << 62)) return false;
    }
    return true;
}
bool parse_int(const char *s, const char *e, int64_t &v) {  // Rust's str::parse::<i64>: optional sign, digits
    bool neg = false;
    if (s < e && (*s == '-' || *s == '+')) {
        neg = *s == '-';
        ++s;
    }
    uint64_t u;
    if (!parse_uint(s, e, u)) return false;
    v = neg ? -(int64_t)u : (int64_t)u;
    return true;
}
// CigarString::try_from(&[u8]) of rust-htslib: <digits><op> repeated, ops MIDNSHP=X
bool parse_cigar_text(const char *s, const char *e, std::vector<uint32_t> &out) {
    out.clear();
    while (s < e) {
        const char *d = s;
        while (d < e && *d >= '0' && *d <= '9') ++d;
        uint64_t len;
        if (d == s || d == e || !parse_uint(s, d, len) || len > 0x0fffffffull) return false;
        const char *ops = "MIDNSHP=X";
        const char *o = strchr(ops, *d);
        if (!o || !*d) return false;
        out.push_back((uint32_t)(len << 4) | (uint32_t)(o - ops));
        s = d + 1;
    }
    return true;
}

// The CIGAR htslib hands to the reference for ANY record after bam_read1: a long one (more than 65535 ops) is stored in the
// CG:B,I tag behind the placeholder <l_seq>S<ref_len>N and restored on read.
inline void real_cigar(const Rec &rec, std::vector<uint32_t> &cigar) {
    const uint32_t nc = rec.n_cigar();
    cigar.resize(nc);
    for (uint32_t i = 0; i < nc; ++i) cigar[i] = rd32(rec.cigar() + 4 * (size_t)i);
    if (nc == 2 && (cigar[0] & 15u) == 4 && (cigar[0] >> 4) == rec.l_seq() && (cigar[1] & 15u) == 3) {
        size_t fl = 0;
        const uint8_t *cg = aux_find(rec.aux(), rec.end(), "CG", &fl);
        if (cg && cg[2] == 'B' && cg[3] == 'I') {
            uint32_t n = rd32(cg + 4);
            cigar.resize(n);
            for (uint32_t i = 0; i < n; ++i) cigar[i] = rd32(cg + 8 + 4 * (size_t)i);
        }
    }
}

// get_seq_order_read_split_segments (split_read.rs:56-155) of one primary record.
plo_status split_segments(const std::unordered_map<std::string, uint32_t> &label_to_index, const Rec &rec, std::vector<SaSeg> &out, std::vector<uint32_t> &primary_cigar,
                          std::string &err) {
    out.clear();
    real_cigar(rec, primary_cigar);
    const bool fwd = !(rec.flag() & 0x10);
    uint64_t rs, re, rsize;
    read_clip_positions(primary_cigar.data(), primary_cigar.size(), rs, re, rsize);
    SaSeg p;
    p.contig = (uint32_t)rec.tid();
    p.pos = rec.pos();
    p.fwd = fwd;
    p.mapq = rec.mapq();
    p.primary = true;
    p.so_start = fwd ? rs : rsize - re;  // get_seq_order_read_pos :78-89
    p.so_end = fwd ? re : rsize - rs;
    out.push_back(std::move(p));
    size_t fl = 0;
    const uint8_t *sa = aux_find(rec.aux(), rec.end(), "SA", &fl);
    if (sa) {
        if (sa[2] != 'Z') {
            err = "SA aux tag is not a string";  // unexpected_aux_val_err (aux/mod.rs:80-82)
            return PLO_ERR_DATA;
        }
        const char *s = (const char *)sa + 3, *e = (const char *)sa + fl - 1;
        uint32_t seg_index = 0;
        while (s < e) {  // split_terminator(';') (sa_tag_parser.rs:55-59)
            const char *q = (const char *)memchr(s, ';', (size_t)(e - s));
            const char *se = q ? q : e;
            // split_terminator(',') (:26): fields between commas, a trailing empty field dropped; exactly six (:27-31)
            const char *f[6], *fe[6];
            int nf = 0;
            bool too_many = false;
            for (const char *c = s, *start = s;; ++c) {
                if (c == se || *c == ',') {
                    if (!(c == se && start == se && se > s)) {  // not the empty field after a trailing comma
                        if (c == se && start == se && se == s) break;  // empty segment: no fields at all
                        if (nf == 6) {
                            too_many = true;
                            break;
                        }
                        f[nf] = start;
                        fe[nf] = c;
                        ++nf;
                    }
                    if (c == se) break;
                    start = c + 1;
                }
            }
            if (too_many || nf != 6) {
                err = "Unexpected segment in bam SA tag: " + std::string(s, se);
                return PLO_ERR_DATA;
            }
            SaSeg g;
            std::string rname(f[0], fe[0]);
            int64_t pos1, nm;
            uint64_t mq;
            if (!parse_int(f[1], fe[1], pos1) || !parse_cigar_text(f[3], fe[3], g.cigar) || !parse_uint(f[4], fe[4], mq) || mq > 255 ||
                !parse_int(f[5], fe[5], nm) || nm < INT32_MIN || nm > INT32_MAX) {
                err = "malformed SA segment: " + std::string(s, se);
                return PLO_ERR_DATA;
            }
            g.pos = pos1 - 1;
            g.fwd = (fe[2] - f[2] == 1 && f[2][0] == '+');
            g.mapq = (uint8_t)mq;
            g.primary = false;
            bool aligned = false;
            for (uint32_t c : g.cigar) aligned |= op_is_match(c);
            if (!aligned) {  // :112-115
                err = "Bam record split segment id unaligned in read " + std::string((const char *)rec.qname());
                return PLO_ERR_DATA;
            }
            uint64_t s0, e0, sz;
            read_clip_positions(g.cigar.data(), g.cigar.size(), s0, e0, sz);
            if (sz != rsize) {  // assert_eq!(primary_read_size, read_size) :118
                err = "SA segment read length differs from the primary record's in read " + std::string((const char *)rec.qname());
                return PLO_ERR_DATA;
            }
            g.so_start = g.fwd ? s0 : sz - e0;
            g.so_end = g.fwd ? e0 : sz - s0;
            auto it = label_to_index.find(rname);
            if (it == label_to_index.end()) {  // :121-130
                err = "In read '" + std::string((const char *)rec.qname()) + "', the SA aux tag describes a split read mapped to " + rname +
                      ", which is not found in the input header";
                return PLO_ERR_DATA;
            }
            g.contig = it->second;
            out.push_back(std::move(g));
            ++seg_index;
            s = q ? q + 1 : e;
        }
        (void)seg_index;
        std::stable_sort(out.begin(), out.end(), [](const SaSeg &a, const SaSeg &b) { return a.so_start < b.so_start; });  // :141
    }
    for (const SaSeg &g : out)
        if (g.so_start >= g.so_end) {  // :146-152
            err = "Can't parse consistent split read information from SA tag format in read: " + std::string((const char *)rec.qname());
            return PLO_ERR_DATA;
        }
    return PLO_OK;
}


}  // namespace
