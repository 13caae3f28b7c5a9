// bam_host.cpp -- host side of the path (include/portello_bam.h): BGZF/BAM input, batch construction from records
// (get_seq_order_read_split_segments), BAM record bytes of the lifted alignments, BGZF output.  Plain C++17 + zlib;
// no GPU code.  Citations are relative to /root/reference; rust-htslib / htslib semantics are restated from their
// published behaviour (third party, absent from the reference tree).
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
#include <mutex>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "../../include/portello_bam.h"

namespace {

thread_local std::string g_bam_err;

plo_status fail(plo_status st, const std::string &msg) {
    g_bam_err = msg;
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

// growable byte buffer that does not zero what it allocates (the inflated stream and the output records are written once, in
// parallel, right after the allocation)
struct RawBuf {
    uint8_t *p = nullptr;
    size_t n = 0, cap = 0;
    RawBuf() = default;
    RawBuf(const RawBuf &) = delete;
    RawBuf &operator=(const RawBuf &) = delete;
    ~RawBuf() { free(p); }
    uint8_t *data() { return p; }
    const uint8_t *data() const { return p; }
    size_t size() const { return n; }
    bool resize(size_t want) {  // keeps the first min(n, want) bytes
        if (want > cap) {
            size_t nc = std::max(want, cap + cap / 2);
            uint8_t *q = (uint8_t *)realloc(p, nc ? nc : 1);
            if (!q) return false;
            p = q;
            cap = nc;
        }
        n = want;
        return true;
    }
};
void parallel_copy(uint8_t *dst, const uint8_t *src, size_t n, int threads);

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
    size_t bpos = 0;
    int threads = 1;
    bool eof = false;
    static constexpr size_t CHUNK = 256u << 20;

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
        if (bpos) {
            memmove(buf.data(), buf.data() + bpos, buf.size() - bpos);
            buf.resize(buf.size() - bpos);
            bpos = 0;
        }
        const size_t cpos0 = cpos;
        std::vector<Blk> blks;
        size_t u = buf.size();
        const size_t target = std::max(want, CHUNK);
        while (cpos < size && u < target) {
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
            u += b.ulen;
            cpos += bsize;
            blks.push_back(b);
        }
        if (cpos >= size) eof = true;
        if (cpos > cpos0) madvise((void *)(map + (cpos0 & ~(size_t)4095)), cpos - (cpos0 & ~(size_t)4095), MADV_WILLNEED);
        if (!buf.resize(u)) return fail(PLO_ERR_OUT_OF_MEMORY, "out of host memory for the inflated BAM stream");
        std::atomic<int> bad{0};
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

}  // namespace

// ---------------------------------------------------------------------------------------------------------------------
// objects
// ---------------------------------------------------------------------------------------------------------------------
struct plo_bam_reader {
    BgzfIn in;
    std::string text;
    std::vector<std::string> names;
    std::vector<const char *> name_ptrs;
    std::vector<uint32_t> lens;
    std::unordered_map<std::string, uint32_t> label_to_index;  // ChromList::label_to_index (chrom_list.rs:21-24)
    int threads = 1;
};

struct plo_bam_window {
    const plo_bam_reader *reader = nullptr;
    int threads = 1;
    RawBuf raw;                     // the window's stretch of the BAM stream (all records)
    std::vector<uint64_t> rec_at;   // [n] offset (of the block_size word) of every primary record inside raw
    std::vector<uint8_t> unmapped;
    uint32_t n_unmapped = 0;
    // batch (plo_batch_in) arrays
    HostBuf b_rev, b_len, b_soff, b_seq, b_seg_read, b_seg_contig, b_seg_pos, b_seg_fwd, b_coff, b_cigar, b_flags, b_qual, b_qoff;
    std::vector<uint32_t> read_seg_off;  // [n + 1] first segment of every read
    // output records
    RawBuf out;
    std::vector<uint64_t> out_off;
    uint32_t n_records() const { return (uint32_t)rec_at.size(); }
    Rec record(uint32_t i) const {
        const uint8_t *p = raw.data() + rec_at[i];
        return Rec{p + 4, rd32(p)};
    }
};

extern "C" {

const char *plo_bam_last_error(void) { return g_bam_err.c_str(); }

plo_status plo_bam_open(const char *path, int n_threads, plo_bam_reader **out) {
    if (!path || !out) return PLO_ERR_INVALID_ARG;
    *out = nullptr;
    plo_bam_reader *r = new plo_bam_reader();
    r->threads = std::max(1, n_threads);
    plo_status st = r->in.open(path, r->threads);
    auto bail = [&](plo_status s) {
        r->in.close();
        delete r;
        return s;
    };
    if (st != PLO_OK) return bail(st);
    uint8_t hd[8];
    if ((st = r->in.read(hd, 8)) != PLO_OK) return bail(st);
    if (memcmp(hd, "BAM\1", 4) != 0) return bail(fail(PLO_ERR_IO, "not a BAM file (bad magic)"));
    uint32_t l_text = rd32(hd + 4);
    r->text.resize(l_text);
    if (l_text && (st = r->in.read(&r->text[0], l_text)) != PLO_OK) return bail(st);
    uint8_t w[4];
    if ((st = r->in.read(w, 4)) != PLO_OK) return bail(st);
    uint32_t n_ref = rd32(w);
    for (uint32_t i = 0; i < n_ref; ++i) {
        if ((st = r->in.read(w, 4)) != PLO_OK) return bail(st);
        uint32_t l_name = rd32(w);
        std::string name(l_name, '\0');
        if (l_name && (st = r->in.read(&name[0], l_name)) != PLO_OK) return bail(st);
        while (!name.empty() && name.back() == '\0') name.pop_back();
        if ((st = r->in.read(w, 4)) != PLO_OK) return bail(st);
        if (r->label_to_index.count(name)) return bail(fail(PLO_ERR_DATA, "duplicate reference name in BAM header: " + name));  // chrom_list.rs:47
        r->label_to_index[name] = i;
        r->names.push_back(name);
        r->lens.push_back(rd32(w));
    }
    for (auto &n : r->names) r->name_ptrs.push_back(n.c_str());
    *out = r;
    return PLO_OK;
}

void plo_bam_close(plo_bam_reader *r) {
    if (!r) return;
    r->in.close();
    delete r;
}

plo_status plo_bam_header(const plo_bam_reader *r, const char **text, uint32_t *l_text, uint32_t *n_ref, const char *const **ref_names,
                          const uint32_t **ref_lens) {
    if (!r) return PLO_ERR_INVALID_ARG;
    if (text) *text = r->text.c_str();
    if (l_text) *l_text = (uint32_t)r->text.size();
    if (n_ref) *n_ref = (uint32_t)r->names.size();
    if (ref_names) *ref_names = r->name_ptrs.data();
    if (ref_lens) *ref_lens = r->lens.data();
    return PLO_OK;
}

plo_status plo_bam_read_window(plo_bam_reader *r, uint32_t max_records, plo_bam_window **out) {
    if (!r || !out || !max_records) return PLO_ERR_INVALID_ARG;
    *out = nullptr;
    plo_bam_window *w = new plo_bam_window();
    w->reader = r;
    w->threads = r->threads;
    plo_status st = PLO_OK;
    // walk the records of the inflated stream (no copies), then take the whole stretch with one parallel copy
    size_t at = 0;  // offset from r->in.bpos
    std::vector<uint64_t> unm_at;
    while (w->rec_at.size() < max_records) {
        if (r->in.avail() < at + 4 && (st = r->in.fill(at + 4)) != PLO_OK) break;
        if (r->in.avail() == at) break;  // end of file
        if (r->in.avail() < at + 4) {
            st = fail(PLO_ERR_IO, "truncated BAM record");
            break;
        }
        uint32_t bs = rd32(r->in.buf.data() + r->in.bpos + at);
        if (bs < 32) {
            st = fail(PLO_ERR_IO, "BAM record shorter than its fixed fields");
            break;
        }
        if (r->in.avail() < at + 4 + (size_t)bs && (st = r->in.fill(at + 4 + (size_t)bs)) != PLO_OK) break;
        if (r->in.avail() < at + 4 + (size_t)bs) {
            st = fail(PLO_ERR_IO, "truncated BAM record");
            break;
        }
        const uint8_t *p = r->in.buf.data() + r->in.bpos + at;
        Rec rec{p + 4, bs};
        if (!rec.layout_ok()) {
            st = fail(PLO_ERR_IO, "BAM record fields exceed its block_size");
            break;
        }
        uint16_t flag = rec.flag();
        if (flag & 0x4) unm_at.push_back(at);                 // scan_unmapped_reads :551-555
        else if (!(flag & 0x800)) w->rec_at.push_back(at);    // :404 supplementary records are reached through the primary's SA tag
        at += 4 + (size_t)bs;
    }
    if (st == PLO_OK && !w->raw.resize(at)) st = fail(PLO_ERR_OUT_OF_MEMORY, "out of host memory for a window of records");
    if (st != PLO_OK) {
        delete w;
        return st;
    }
    parallel_copy(w->raw.data(), r->in.buf.data() + r->in.bpos, at, r->threads);
    r->in.bpos += at;
    for (uint64_t u : unm_at) {
        const uint8_t *p = w->raw.data() + u;
        w->unmapped.insert(w->unmapped.end(), p, p + 4 + rd32(p));
        ++w->n_unmapped;
    }
    *out = w;
    return PLO_OK;
}

void plo_bam_window_free(plo_bam_window *w) { delete w; }
uint32_t plo_bam_window_n_records(const plo_bam_window *w) { return w ? w->n_records() : 0; }
void plo_bam_window_unmapped(const plo_bam_window *w, const uint8_t **bytes, uint64_t *n_bytes, uint32_t *n_records) {
    if (bytes) *bytes = w ? w->unmapped.data() : nullptr;
    if (n_bytes) *n_bytes = w ? w->unmapped.size() : 0;
    if (n_records) *n_records = w ? w->n_unmapped : 0;
}

}  // extern "C"

namespace {

// get_seq_order_read_split_segments (split_read.rs:56-155) of one primary record.  Long CIGARs stored in the CG tag
// (more than 65535 ops) are what htslib hands to the reference after bam_read1: the real CIGAR.
plo_status split_segments(const plo_bam_reader *rd, const Rec &rec, std::vector<SaSeg> &out, std::vector<uint32_t> &primary_cigar,
                          std::string &err) {
    out.clear();
    const uint32_t nc = rec.n_cigar();
    primary_cigar.resize(nc);
    for (uint32_t i = 0; i < nc; ++i) primary_cigar[i] = rd32(rec.cigar() + 4 * (size_t)i);
    if (nc == 2 && (primary_cigar[0] & 15u) == 4 && (primary_cigar[0] >> 4) == rec.l_seq() && (primary_cigar[1] & 15u) == 3) {
        size_t fl = 0;
        const uint8_t *cg = aux_find(rec.aux(), rec.end(), "CG", &fl);
        if (cg && cg[2] == 'B' && cg[3] == 'I') {
            uint32_t n = rd32(cg + 4);
            primary_cigar.resize(n);
            for (uint32_t i = 0; i < n; ++i) primary_cigar[i] = rd32(cg + 8 + 4 * (size_t)i);
        }
    }
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
            auto it = rd->label_to_index.find(rname);
            if (it == rd->label_to_index.end()) {  // :121-130
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

extern "C" plo_status plo_bam_window_batch(plo_bam_window *w, plo_batch_in *batch, plo_finish_in *fin) {
    if (!w || !batch) return PLO_ERR_INVALID_ARG;
    memset(batch, 0, sizeof(*batch));
    const uint32_t n = w->n_records();
    // pass 1 (parallel): segments of every read; sizes
    std::vector<std::vector<SaSeg>> segs(n);
    std::vector<std::vector<uint32_t>> pcig(n);
    std::vector<uint32_t> n_seg(n + 1, 0);
    std::vector<uint64_t> n_ops(n + 1, 0), n_seqb(n + 1, 0), n_qual(n + 1, 0);
    std::atomic<int> bad{0};
    std::string first_err;
    std::atomic<uint32_t> err_rec{UINT32_MAX};
    std::vector<std::string> errs(n);
    parallel_for(n, w->threads, [&](size_t i) {
        Rec rec = w->record((uint32_t)i);
        std::string err;
        plo_status st = split_segments(w->reader, rec, segs[i], pcig[i], err);
        if (st != PLO_OK) {
            errs[i] = err;
            bad = 1;
            uint32_t cur = err_rec.load();
            while ((uint32_t)i < cur && !err_rec.compare_exchange_weak(cur, (uint32_t)i)) {
            }
            return;
        }
        uint64_t ops = 0;
        for (const SaSeg &g : segs[i]) ops += g.primary ? pcig[i].size() : g.cigar.size();
        n_seg[i + 1] = (uint32_t)segs[i].size();
        n_ops[i + 1] = ops;
        n_seqb[i + 1] = (rec.l_seq() + 1) / 2;
        n_qual[i + 1] = rec.l_seq();
    });
    if (bad) return fail(PLO_ERR_DATA, errs[err_rec.load()]);
    for (uint32_t i = 0; i < n; ++i) {
        n_seg[i + 1] += n_seg[i];
        n_ops[i + 1] += n_ops[i];
        n_seqb[i + 1] += n_seqb[i];
        n_qual[i + 1] += n_qual[i];
    }
    if (n_ops[n] > 0x7fffffffull) return fail(PLO_ERR_RANGE, "window carries more than 2^31 CIGAR ops; read fewer records per window");
    const uint32_t ns = n_seg[n];
    w->read_seg_off.assign(n_seg.begin(), n_seg.end());
    uint8_t *rev = (uint8_t *)w->b_rev.ensure(std::max<size_t>(n, 1));
    uint32_t *rlen = (uint32_t *)w->b_len.ensure(std::max<size_t>(n, 1) * 4);
    uint64_t *soff = (uint64_t *)w->b_soff.ensure(std::max<size_t>(n, 1) * 8);
    uint8_t *seq = (uint8_t *)w->b_seq.ensure(std::max<uint64_t>(n_seqb[n], 16));
    uint32_t *seg_read = (uint32_t *)w->b_seg_read.ensure(std::max<size_t>(ns, 1) * 4);
    uint32_t *seg_contig = (uint32_t *)w->b_seg_contig.ensure(std::max<size_t>(ns, 1) * 4);
    int64_t *seg_pos = (int64_t *)w->b_seg_pos.ensure(std::max<size_t>(ns, 1) * 8);
    uint8_t *seg_fwd = (uint8_t *)w->b_seg_fwd.ensure(std::max<size_t>(ns, 1));
    uint32_t *coff = (uint32_t *)w->b_coff.ensure(((size_t)ns + 1) * 4);
    uint32_t *cigar = (uint32_t *)w->b_cigar.ensure(std::max<uint64_t>(n_ops[n], 1) * 4);
    uint16_t *flags = (uint16_t *)w->b_flags.ensure(std::max<size_t>(n, 1) * 2);
    uint8_t *qual = fin ? (uint8_t *)w->b_qual.ensure(std::max<uint64_t>(n_qual[n], 16)) : nullptr;
    uint64_t *qoff = fin ? (uint64_t *)w->b_qoff.ensure(std::max<size_t>(n, 1) * 8) : nullptr;
    if (!rev || !rlen || !soff || !seq || !seg_read || !seg_contig || !seg_pos || !seg_fwd || !coff || !cigar || !flags || (fin && (!qual || !qoff)))
        return fail(PLO_ERR_OUT_OF_MEMORY, "out of host memory for the window's batch");
    // pass 2 (parallel): fill
    parallel_for(n, w->threads, [&](size_t i) {
        Rec rec = w->record((uint32_t)i);
        rev[i] = (rec.flag() & 0x10) ? 1 : 0;
        rlen[i] = rec.l_seq();
        soff[i] = n_seqb[i];
        flags[i] = rec.flag();
        memcpy(seq + n_seqb[i], rec.seq(), (size_t)(n_seqb[i + 1] - n_seqb[i]));
        if (fin) {
            qoff[i] = n_qual[i];
            memcpy(qual + n_qual[i], rec.qual(), rec.l_seq());
        }
        uint32_t s = n_seg[i];
        uint64_t o = n_ops[i];
        for (const SaSeg &g : segs[i]) {
            const std::vector<uint32_t> &cg = g.primary ? pcig[i] : g.cigar;
            seg_read[s] = (uint32_t)i;
            seg_contig[s] = g.contig;
            seg_pos[s] = g.pos;
            seg_fwd[s] = g.fwd ? 1 : 0;
            coff[s] = (uint32_t)o;
            if (!cg.empty()) memcpy(cigar + o, cg.data(), cg.size() * 4);
            o += cg.size();
            ++s;
        }
    });
    coff[ns] = (uint32_t)n_ops[n];
    batch->n_reads = n;
    batch->read_is_reverse = rev;
    batch->read_seq_len = rlen;
    batch->read_seq_off = soff;
    batch->seq = seq;
    batch->seq_bytes = n_seqb[n];
    batch->seq_fmt = PLO_SEQ_BAM4;
    batch->n_segs = ns;
    batch->seg_read = seg_read;
    batch->seg_contig = seg_contig;
    batch->seg_pos = seg_pos;
    batch->seg_is_fwd_strand = seg_fwd;
    batch->seg_cigar_off = coff;
    batch->cigar = cigar;
    if (fin) {
        fin->read_flags = flags;
        fin->qual = qual;
        fin->read_qual_off = qoff;
        fin->qual_bytes = n_qual[n];
    }
    return PLO_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// output records
// ---------------------------------------------------------------------------------------------------------------------
namespace {

// hts_reg2bin(begin, end, 14, 5) (lib/rust-vc-utils/src/bam_utils/util.rs:10-35)
uint16_t reg2bin(uint64_t begin, uint64_t end) {
    --end;
    int l = 5, s = 14;
    uint64_t t = ((1ull << 15) - 1) / 7;
    while (l > 0) {
        if (begin >> s == end >> s) return (uint16_t)(t + (begin >> s));
        --l;
        s += 3;
        t -= 1ull << (l * 3);
    }
    return 0;
}

size_t decimal_len(uint64_t v) {
    size_t n = 1;
    while (v >= 10) {
        v /= 10;
        ++n;
    }
    return n;
}
uint8_t *put_decimal(uint8_t *p, uint64_t v) {
    size_t n = decimal_len(v);
    for (size_t k = n; k-- > 0;) {
        p[k] = (uint8_t)('0' + v % 10);
        v /= 10;
    }
    return p + n;
}

// 4-bit reverse complement of a packed sequence, as decode -> rev_comp_in_place -> encode does it
// (src/read_alignment_scanner.rs:125-133; comp_base lib/rust-vc-utils/src/seq_util.rs:1-15): A<->T, C<->G, N stays,
// every other code (=, IUPAC ambiguity) becomes N; an odd length leaves the last low nibble 0
void revcomp_packed(const uint8_t *src, uint32_t n, uint8_t *dst) {
    static const uint8_t comp[16] = {15, 8, 4, 15, 2, 15, 15, 15, 1, 15, 15, 15, 15, 15, 15, 15};
    struct Tables {
        uint8_t swap[256];  // both nibbles complemented and exchanged (even length: byte k of the output = swap[byte nb-1-k])
        uint8_t hi[256];    // complemented high nibble, in the high position
        uint8_t lo[256];    // complemented low nibble, in the low position
        Tables() {
            for (int x = 0; x < 256; ++x) {
                swap[x] = (uint8_t)((comp[x & 15] << 4) | comp[x >> 4]);
                hi[x] = (uint8_t)(comp[x >> 4] << 4);
                lo[x] = comp[x & 15];
            }
        }
    };
    static const Tables T;
    const uint32_t nb = (n + 1) / 2;
    if (!(n & 1)) {
        for (uint32_t b = 0; b < nb; ++b) dst[b] = T.swap[src[nb - 1 - b]];
        return;
    }
    // odd length: output byte b = comp(high nibble of source byte m) | comp(low nibble of source byte m - 1), m = nb - 1 - b;
    // the last output byte keeps a zero low nibble
    for (uint32_t b = 0; b + 1 < nb; ++b) {
        const uint32_t m = nb - 1 - b;
        dst[b] = (uint8_t)(T.hi[src[m]] | T.lo[src[m - 1]]);
    }
    dst[nb - 1] = T.hi[src[0]];
}

struct AuxPlan {  // aux bytes of the clone after remove_aux_if_found NM, SA, PS, ZM (first occurrence of each, :105-118)
    const uint8_t *cut[5];
    size_t cut_len[5];
    int n_cut = 0;
    size_t kept = 0;
};
// A record stored with more than 65535 CIGAR ops carries the placeholder <l_seq>S<n>N and the real CIGAR in CG:B,I; htslib's
// bam_read1 restores the CIGAR and deletes that field before the reference sees the record.
bool has_cg_cigar(const Rec &rec, const uint8_t **field, size_t *flen) {
    if (rec.n_cigar() != 2) return false;
    uint32_t c0 = rd32(rec.cigar()), c1 = rd32(rec.cigar() + 4);
    if ((c0 & 15u) != 4 || (c0 >> 4) != rec.l_seq() || (c1 & 15u) != 3) return false;
    size_t fl = 0;
    const uint8_t *cg = aux_find(rec.aux(), rec.end(), "CG", &fl);
    if (!cg || cg[2] != 'B' || cg[3] != 'I') return false;
    if (field) *field = cg;
    if (flen) *flen = fl;
    return true;
}
AuxPlan plan_aux(const Rec &rec) {
    AuxPlan a;
    const uint8_t *b = rec.aux(), *e = rec.end();
    static const char tags[4][3] = {"NM", "SA", "PS", "ZM"};
    for (int k = 0; k < 4; ++k) {
        size_t fl = 0;
        const uint8_t *f = aux_find(b, e, tags[k], &fl);
        if (f) {
            a.cut[a.n_cut] = f;
            a.cut_len[a.n_cut] = fl;
            ++a.n_cut;
        }
    }
    {
        const uint8_t *f = nullptr;
        size_t fl = 0;
        if (has_cg_cigar(rec, &f, &fl)) {
            a.cut[a.n_cut] = f;
            a.cut_len[a.n_cut] = fl;
            ++a.n_cut;
        }
    }
    // sort the cuts by address (at most five)
    for (int i = 1; i < a.n_cut; ++i)
        for (int j = i; j > 0 && a.cut[j] < a.cut[j - 1]; --j) {
            std::swap(a.cut[j], a.cut[j - 1]);
            std::swap(a.cut_len[j], a.cut_len[j - 1]);
        }
    a.kept = (size_t)(e - b);
    for (int i = 0; i < a.n_cut; ++i) a.kept -= a.cut_len[i];
    return a;
}
uint8_t *copy_aux(const Rec &rec, const AuxPlan &a, uint8_t *dst) {
    const uint8_t *b = rec.aux(), *e = rec.end();
    for (int i = 0; i < a.n_cut; ++i) {
        memcpy(dst, b, (size_t)(a.cut[i] - b));
        dst += a.cut[i] - b;
        b = a.cut[i] + a.cut_len[i];
    }
    memcpy(dst, b, (size_t)(e - b));
    return dst + (e - b);
}

// text length of a CIGAR as rust-htslib's Display writes it ("{len}{op}")
size_t cigar_text_len(const uint32_t *c, uint32_t n) {
    size_t l = 0;
    for (uint32_t i = 0; i < n; ++i) l += decimal_len(c[i] >> 4) + 1;
    return l;
}
uint8_t *put_cigar_text(uint8_t *p, const uint32_t *c, uint32_t n) {
    for (uint32_t i = 0; i < n; ++i) {
        p = put_decimal(p, c[i] >> 4);
        *p++ = (uint8_t)"MIDNSHP=X"[c[i] & 15u];
    }
    return p;
}

}  // namespace

extern "C" plo_status plo_records_build(plo_bam_window *w, const plo_batch_out *lift, const plo_records_params *pr, plo_record_buf *out) {
    if (!w || !lift || !pr || !out || !pr->index || !pr->contig_names || !pr->ref_names) return PLO_ERR_INVALID_ARG;
    memset(out, 0, sizeof(*out));
    const uint32_t n = w->n_records();
    if (w->read_seg_off.size() != (size_t)n + 1) return fail(PLO_ERR_INVALID_ARG, "plo_records_build: call plo_bam_window_batch on the window first");
    const uint32_t ni = lift->n_items;
    const uint32_t ns = w->read_seg_off[n];
    const uint32_t *seg_contig = w->b_seg_contig.as<uint32_t>();
    const plo_index_desc *ix = pr->index;
    // items of every read: item_seg is non-decreasing (items ordered by read segment), so reads own contiguous item ranges
    std::vector<uint32_t> seg_item_lo(ns + 1, 0);
    {
        uint32_t it = 0;
        for (uint32_t s = 0; s < ns; ++s) {
            seg_item_lo[s] = it;
            while (it < ni && lift->item_seg[it] == s) ++it;
            if (it < ni && lift->item_seg[it] < s) return fail(PLO_ERR_INVALID_ARG, "plo_records_build: items are not ordered by read segment");
        }
        seg_item_lo[ns] = it;
        if (it != ni) return fail(PLO_ERR_INVALID_ARG, "plo_records_build: item_seg beyond the window's segments");
    }
    for (uint32_t i = 0; i < ni; ++i)
        if (lift->item_status[i] == PLO_ITEM_LEN_MISMATCH || lift->item_status[i] == PLO_ITEM_PANIC)
            return fail(PLO_ERR_DATA, "an item ended LEN_MISMATCH / PANIC: the reference aborts here (src/read_alignment_scanner.rs:207-229)");

    struct ItemInfo {
        uint32_t flag;
        uint32_t sa_len;  // length of this record's get_sa_tag_segment text
        uint32_t ps_len;
    };
    std::vector<ItemInfo> info(ni);
    std::vector<uint64_t> read_bytes(n + 1, 0);
    std::vector<uint32_t> read_nrec(n + 1, 0);
    const int threads = std::max(1, pr->n_threads > 0 ? pr->n_threads : w->threads);

    auto item_lifted = [&](uint32_t i) { return lift->item_status[i] == PLO_ITEM_LIFTED; };
    // serialised size of a lifted record, SA tag excluded
    auto lifted_size = [&](const Rec &rec, const AuxPlan &ap, uint32_t i) -> uint64_t {
        const uint32_t nc = lift->item_cigar_len[i];
        uint64_t sz = 4 + 32 + rec.l_qname() + (uint64_t)(rec.l_seq() + 1) / 2 + rec.l_seq() + ap.kept;
        sz += nc <= 0xffff ? 4ull * nc : 8 + 8 + 4ull * nc;  // bam_write1: fake CIGAR + CG:B,I
        sz += 3 + info[i].ps_len + 1;  // PS:Z
        sz += 4;                       // ZM:C
        return sz;
    };
    // pass 1: flags, primary, text lengths, sizes
    parallel_for(n, threads, [&](size_t r) {
        Rec rec = w->record((uint32_t)r);
        const uint32_t lo = seg_item_lo[w->read_seg_off[r]], hi = seg_item_lo[w->read_seg_off[r + 1]];
        uint32_t n_lift = 0, prim = UINT32_MAX;
        for (uint32_t i = lo; i < hi; ++i) {
            if (!item_lifted(i)) continue;
            ++n_lift;
            if (prim == UINT32_MAX || lift->item_mapq[prim] < lift->item_mapq[i]) prim = i;  // :338-345 first maximum wins
        }
        AuxPlan ap = plan_aux(rec);
        uint64_t bytes = 0;
        if (n_lift == 0) {
            if (!pr->is_target_region) {  // unmapped copy :321-334
                bytes = 4 + 32 + rec.l_qname() + (uint64_t)(rec.l_seq() + 1) / 2 + rec.l_seq() + ap.kept;
                read_nrec[r + 1] = 1;
            }
        } else {
            uint64_t sa_total = 0;
            for (uint32_t i = lo; i < hi; ++i) {
                if (!item_lifted(i)) continue;
                uint32_t fl = rec.flag();
                if (lift->item_need_flipped[i]) fl ^= 0x10;  // :126
                fl |= 0x800;                                 // :282
                if (i == prim) fl &= ~0x800u;                // :346
                info[i].flag = fl;
                const uint32_t seg = lift->item_seg[i];
                const uint32_t contig = seg_contig[seg];
                info[i].ps_len = (uint32_t)(strlen(pr->contig_names[contig]) + 6 + decimal_len(lift->item_cseg[i]) + 1);  // "{}_split{}{+|-}"
                // "{chrom},{pos+1},{strand},{cigar},{mapq},0;" (:292-301)
                const uint32_t *cg = lift->cigar + lift->item_cigar_off[i];
                info[i].sa_len = (uint32_t)(strlen(pr->ref_names[lift->item_chrom_index[i]]) + 1 + decimal_len((uint64_t)(lift->item_ref_pos[i] + 1)) +
                                            1 + 1 + 1 + cigar_text_len(cg, lift->item_cigar_len[i]) + 1 + decimal_len(lift->item_mapq[i]) + 3);
                sa_total += info[i].sa_len;
            }
            for (uint32_t i = lo; i < hi; ++i) {
                if (!item_lifted(i)) continue;
                bytes += lifted_size(rec, ap, i);
                if (n_lift > 1) bytes += 3 + (sa_total - info[i].sa_len) + 1;  // SA:Z of the other records (:352-364)
            }
            read_nrec[r + 1] = n_lift;
        }
        read_bytes[r + 1] = bytes;
    });
    for (uint32_t r = 0; r < n; ++r) {
        read_bytes[r + 1] += read_bytes[r];
        read_nrec[r + 1] += read_nrec[r];
    }
    if (!w->out.resize(read_bytes[n])) return fail(PLO_ERR_OUT_OF_MEMORY, "out of host memory for the output records");
    w->out_off.assign((size_t)read_nrec[n] + 1, 0);
    std::atomic<uint32_t> n_lifted{0}, n_unm{0};
    // pass 2: bytes
    parallel_for(n, threads, [&](size_t r) {
        Rec rec = w->record((uint32_t)r);
        const uint32_t lo = seg_item_lo[w->read_seg_off[r]], hi = seg_item_lo[w->read_seg_off[r + 1]];
        uint8_t *p = w->out.data() + read_bytes[r];
        uint32_t k = read_nrec[r];
        AuxPlan ap = plan_aux(rec);
        const uint32_t l_seq = rec.l_seq(), seqb = (l_seq + 1) / 2;
        auto put_seq_qual = [&](uint8_t *q, bool flip) -> uint8_t * {
            if (flip) {
                revcomp_packed(rec.seq(), l_seq, q);
                q += seqb;
                std::reverse_copy(rec.qual(), rec.qual() + l_seq, q);
                return q + l_seq;
            }
            memcpy(q, rec.seq(), (size_t)seqb + l_seq);
            return q + seqb + l_seq;
        };
        if (read_nrec[r + 1] - read_nrec[r] == 0) return;
        uint32_t n_lift = 0;
        for (uint32_t i = lo; i < hi; ++i) n_lift += item_lifted(i) ? 1 : 0;
        if (n_lift == 0) {  // unmapped copy :321-334
            w->out_off[k] = (uint64_t)(p - w->out.data());
            uint8_t *b = p + 4;
            uint32_t fl = rec.flag();
            fl |= 0x4;
            fl &= ~0x800u;
            const bool flip = (fl & 0x10) != 0;
            if (flip) fl ^= 0x10;
            wr32(b, (uint32_t)-1);
            wr32(b + 4, (uint32_t)-1);
            b[8] = (uint8_t)rec.l_qname();
            b[9] = 255;
            wr16(b + 10, rec.bin());
            wr16(b + 12, 0);
            wr16(b + 14, (uint16_t)fl);
            wr32(b + 16, l_seq);
            memcpy(b + 20, rec.p + 20, 12);  // mate reference, mate position, template length: untouched
            uint8_t *q = b + 32;
            memcpy(q, rec.qname(), rec.l_qname());
            q += rec.l_qname();
            q = put_seq_qual(q, flip);
            q = copy_aux(rec, ap, q);
            wr32(p, (uint32_t)(q - b));
            n_unm.fetch_add(1);
            return;
        }
        n_lifted.fetch_add(n_lift);
        for (uint32_t i = lo; i < hi; ++i) {
            if (!item_lifted(i)) continue;
            w->out_off[k++] = (uint64_t)(p - w->out.data());
            uint8_t *b = p + 4;
            const uint32_t nc = lift->item_cigar_len[i];
            const uint32_t *cg = lift->cigar + lift->item_cigar_off[i];
            const int64_t pos = lift->item_ref_pos[i];
            int64_t ref_len = 0;
            for (uint32_t c = 0; c < nc; ++c) ref_len += op_ref_len(cg[c]);
            wr32(b, lift->item_chrom_index[i]);
            wr32(b + 4, (uint32_t)(int32_t)pos);
            b[8] = (uint8_t)rec.l_qname();
            b[9] = lift->item_mapq[i];
            wr16(b + 10, reg2bin((uint64_t)pos, (uint64_t)(pos + ref_len)));  // :278-279
            wr16(b + 12, (uint16_t)(nc <= 0xffff ? nc : 2));
            wr16(b + 14, (uint16_t)info[i].flag);
            wr32(b + 16, l_seq);
            memcpy(b + 20, rec.p + 20, 12);
            uint8_t *q = b + 32;
            memcpy(q, rec.qname(), rec.l_qname());
            q += rec.l_qname();
            if (nc <= 0xffff) {
                for (uint32_t c = 0; c < nc; ++c) wr32(q + 4 * (size_t)c, cg[c]);
                q += 4 * (size_t)nc;
            } else {  // bam_write1: <l_seq>S<ref_len>N, the real CIGAR goes into CG:B,I after the other tags
                wr32(q, (l_seq << 4) | 4u);
                wr32(q + 4, ((uint32_t)ref_len << 4) | 3u);
                q += 8;
            }
            q = put_seq_qual(q, lift->item_need_flipped[i] != 0);
            q = copy_aux(rec, ap, q);
            // PS:Z "{contig}_split{cseg}{+|-}" (:254-265)
            const uint32_t seg = lift->item_seg[i];
            const uint32_t contig = seg_contig[seg];
            const bool cfwd = ix->seg_is_fwd_strand[ix->contig_seg_off[contig] + lift->item_cseg[i]] != 0;
            *q++ = 'P';
            *q++ = 'S';
            *q++ = 'Z';
            size_t cl = strlen(pr->contig_names[contig]);
            memcpy(q, pr->contig_names[contig], cl);
            q += cl;
            memcpy(q, "_split", 6);
            q += 6;
            q = put_decimal(q, lift->item_cseg[i]);
            *q++ = cfwd ? '+' : '-';
            *q++ = 0;
            // ZM:C original MAPQ (:266-268)
            *q++ = 'Z';
            *q++ = 'M';
            *q++ = 'C';
            *q++ = rec.mapq();
            if (n_lift > 1) {  // SA:Z: the segments of the read's other records, in record order (:352-364)
                *q++ = 'S';
                *q++ = 'A';
                *q++ = 'Z';
                for (uint32_t j = lo; j < hi; ++j) {
                    if (j == i || !item_lifted(j)) continue;
                    const char *cn = pr->ref_names[lift->item_chrom_index[j]];
                    size_t l = strlen(cn);
                    memcpy(q, cn, l);
                    q += l;
                    *q++ = ',';
                    q = put_decimal(q, (uint64_t)(lift->item_ref_pos[j] + 1));
                    *q++ = ',';
                    *q++ = (info[j].flag & 0x10) ? '-' : '+';
                    *q++ = ',';
                    q = put_cigar_text(q, lift->cigar + lift->item_cigar_off[j], lift->item_cigar_len[j]);
                    *q++ = ',';
                    q = put_decimal(q, lift->item_mapq[j]);
                    *q++ = ',';
                    *q++ = '0';
                    *q++ = ';';
                }
                *q++ = 0;
            }
            if (nc > 0xffff) {
                memcpy(q, "CGBI", 4);
                wr32(q + 4, nc);
                q += 8;
                for (uint32_t c = 0; c < nc; ++c) wr32(q + 4 * (size_t)c, cg[c]);
                q += 4 * (size_t)nc;
            }
            wr32(p, (uint32_t)(q - b));
            p = q;
        }
    });
    w->out_off[read_nrec[n]] = read_bytes[n];
    out->bytes = w->out.data();
    out->n_bytes = w->out.size();
    out->n_records = read_nrec[n];
    out->record_off = w->out_off.data();
    out->n_lifted = n_lifted.load();
    out->n_unmapped_copies = n_unm.load();
    return PLO_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// output: header text + BGZF writer
// ---------------------------------------------------------------------------------------------------------------------
extern "C" char *plo_bam_output_header(uint32_t n_ref, const char *const *ref_names, const uint32_t *ref_lens, const char *program_name,
                                       const char *program_version, const char *cmdline) {
    std::string t = "@HD\tVN:1.6\tSO:unsorted\n";
    for (uint32_t i = 0; i < n_ref; ++i) t += std::string("@SQ\tSN:") + ref_names[i] + "\tLN:" + std::to_string(ref_lens[i]) + "\n";
    std::string pn = program_name ? program_name : "portello", pv = program_version ? program_version : "";
    t += "@PG\tPN:" + pn + "\tID:" + pn + "-" + pv + "\tVN:" + pv + "\tCL:" + (cmdline ? cmdline : "") + "\n";
    char *r = (char *)malloc(t.size() + 1);
    if (r) memcpy(r, t.c_str(), t.size() + 1);
    return r;
}
extern "C" void plo_bam_free_text(char *text) { free(text); }

struct plo_bam_writer {
    int fd = -1;
    int level = 0, threads = 1;
    bool seekable = false;
    uint64_t file_off = 0;
    std::vector<uint8_t> pend;  // tail of the stream that does not fill a block yet (< BLOCK bytes)
    RawBuf scratch;
    static constexpr size_t BLOCK = 0xff00;  // htslib's BGZF_BLOCK_SIZE
    plo_status emit(const uint8_t *src, size_t n);  // n bytes -> ceil(n / BLOCK) BGZF blocks, written out
    plo_status put(const uint8_t *src, size_t n);
};

// Blocks are built in parallel, each in its own slot of a scratch buffer, and written with positional writes from several
// threads when the output is a regular file (page-cache copies scale with the writers), in order otherwise (pipe / stdout).
plo_status plo_bam_writer::emit(const uint8_t *src, size_t n) {
    const size_t nblk = (n + BLOCK - 1) / BLOCK;
    if (!nblk) return PLO_OK;
    const size_t slot = level == 0 ? 18 + 5 + BLOCK + 8 : 18 + BLOCK + 1024 + 8;
    if (!scratch.resize(nblk * slot)) return fail(PLO_ERR_OUT_OF_MEMORY, "out of host memory for BGZF output blocks");
    std::vector<uint32_t> olen(nblk, 0);
    std::atomic<int> bad{0};
    uint8_t *outb = scratch.data();
    parallel_ranges(nblk, threads, [&](size_t lo, size_t hi) {
        const LibDeflate &ld = libdeflate();
        void *lc = (level > 0 && ld.ok) ? ld.alloc_compressor(level) : nullptr;
        z_stream zs;
        memset(&zs, 0, sizeof(zs));
        if (level > 0 && !lc && deflateInit2(&zs, level, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) {
            bad = 1;
            return;
        }
        for (size_t b = lo; b < hi; ++b) {
            const uint8_t *in = src + b * BLOCK;
            const size_t len = std::min(BLOCK, n - b * BLOCK);
            uint8_t *o = outb + b * slot;
            static const uint8_t hdr[16] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0};
            memcpy(o, hdr, 16);
            size_t clen = 0;
            if (level == 0) {
                o[18] = 1;  // final stored block
                wr16(o + 19, (uint16_t)len);
                wr16(o + 21, (uint16_t)~len);
                memcpy(o + 23, in, len);
                clen = 5 + len;
            } else if (lc) {
                clen = ld.deflate_compress(lc, in, len, o + 18, slot - 18 - 8);
                if (clen == 0 || 18 + clen + 8 > 65536) {
                    bad = 1;
                    break;
                }
            } else {
                deflateReset(&zs);
                zs.next_in = (Bytef *)in;
                zs.avail_in = (uInt)len;
                zs.next_out = o + 18;
                zs.avail_out = (uInt)(slot - 18 - 8);
                int rc = deflate(&zs, Z_FINISH);
                clen = zs.total_out;
                if (rc != Z_STREAM_END || 18 + clen + 8 > 65536) {
                    bad = 1;
                    break;
                }
            }
            wr16(o + 16, (uint16_t)(18 + clen + 8 - 1));
            wr32(o + 18 + clen, fast_crc32(in, len));
            wr32(o + 18 + clen + 4, (uint32_t)len);
            olen[b] = (uint32_t)(18 + clen + 8);
        }
        if (lc) ld.free_compressor(lc);
        else if (level > 0) deflateEnd(&zs);
    });
    if (bad) return fail(PLO_ERR_IO, "BGZF block compression failed");
    std::vector<uint64_t> at(nblk + 1, 0);
    for (size_t b = 0; b < nblk; ++b) at[b + 1] = at[b] + olen[b];
    if (seekable) {
        // runs of blocks that are contiguous in the scratch buffer (level 0: all full blocks) go out with one pwrite each
        const size_t group = 128;
        const size_t ng = (nblk + group - 1) / group;
        parallel_for(ng, std::min(threads, 16), [&](size_t g) {
            size_t b = g * group, e = std::min(nblk, b + group);
            while (b < e) {
                size_t r = b + 1;
                while (r < e && olen[r - 1] == slot) ++r;  // block r starts right behind block r - 1
                const uint8_t *p = outb + b * slot;
                size_t left = (size_t)(at[r] - at[b]);
                uint64_t off = file_off + at[b];
                while (left) {
                    ssize_t k = pwrite(fd, p, left, (off_t)off);
                    if (k <= 0) {
                        bad = 1;
                        return;
                    }
                    p += k;
                    off += (uint64_t)k;
                    left -= (size_t)k;
                }
                b = r;
            }
        });
        if (bad) return fail(PLO_ERR_IO, "write failed");
    } else {
        for (size_t b = 0; b < nblk; ++b) {
            const uint8_t *p = outb + b * slot;
            size_t left = olen[b];
            while (left) {
                ssize_t k = ::write(fd, p, left);
                if (k <= 0) return fail(PLO_ERR_IO, "write failed");
                p += k;
                left -= (size_t)k;
            }
        }
    }
    file_off += at[nblk];
    return PLO_OK;
}

plo_status plo_bam_writer::put(const uint8_t *src, size_t n) {
    if (!pend.empty()) {  // complete the open block first
        size_t take = std::min(n, BLOCK - pend.size());
        pend.insert(pend.end(), src, src + take);
        src += take;
        n -= take;
        if (pend.size() < BLOCK) return PLO_OK;
        plo_status st = emit(pend.data(), pend.size());
        pend.clear();
        if (st != PLO_OK) return st;
    }
    const size_t full = n / BLOCK * BLOCK;
    if (full) {
        plo_status st = emit(src, full);  // straight from the caller's bytes
        if (st != PLO_OK) return st;
    }
    pend.insert(pend.end(), src + full, src + n);
    return PLO_OK;
}

extern "C" plo_status plo_bam_writer_open(const char *path, const char *header_text, uint32_t n_ref, const char *const *ref_names,
                                          const uint32_t *ref_lens, int level, int n_threads, plo_bam_writer **out) {
    if (!path || !out || (n_ref && (!ref_names || !ref_lens))) return PLO_ERR_INVALID_ARG;
    *out = nullptr;
    plo_bam_writer *w = new plo_bam_writer();
    w->level = std::min(9, std::max(0, level));
    w->threads = std::max(1, n_threads);
    w->fd = strcmp(path, "-") == 0 ? dup(1) : ::open(path, O_WRONLY | O_CREAT | O_TRUNC, 0644);
    if (w->fd < 0) {
        delete w;
        return fail(PLO_ERR_IO, std::string("cannot create ") + path);
    }
    w->seekable = lseek(w->fd, 0, SEEK_CUR) != (off_t)-1 && strcmp(path, "-") != 0;
    size_t lt = header_text ? strlen(header_text) : 0;
    std::vector<uint8_t> p;
    p.insert(p.end(), {'B', 'A', 'M', 1});
    uint8_t b4[4];
    wr32(b4, (uint32_t)lt);
    p.insert(p.end(), b4, b4 + 4);
    if (lt) p.insert(p.end(), (const uint8_t *)header_text, (const uint8_t *)header_text + lt);
    wr32(b4, n_ref);
    p.insert(p.end(), b4, b4 + 4);
    for (uint32_t i = 0; i < n_ref; ++i) {
        size_t l = strlen(ref_names[i]) + 1;
        wr32(b4, (uint32_t)l);
        p.insert(p.end(), b4, b4 + 4);
        p.insert(p.end(), (const uint8_t *)ref_names[i], (const uint8_t *)ref_names[i] + l);
        wr32(b4, ref_lens[i]);
        p.insert(p.end(), b4, b4 + 4);
    }
    plo_status st = w->emit(p.data(), p.size());  // the header ends its own block(s), as htslib's bam_hdr_write + bgzf_flush do
    if (st != PLO_OK) {
        ::close(w->fd);
        delete w;
        return st;
    }
    *out = w;
    return PLO_OK;
}

extern "C" plo_status plo_bam_write(plo_bam_writer *w, const uint8_t *bytes, uint64_t n) {
    if (!w || (n && !bytes)) return PLO_ERR_INVALID_ARG;
    return w->put(bytes, (size_t)n);
}

extern "C" plo_status plo_bam_writer_close(plo_bam_writer *w) {
    if (!w) return PLO_ERR_INVALID_ARG;
    plo_status st = w->pend.empty() ? PLO_OK : w->emit(w->pend.data(), w->pend.size());
    static const uint8_t eof_block[28] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0, 0x1b, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (st == PLO_OK) {
        ssize_t k = w->seekable ? pwrite(w->fd, eof_block, 28, (off_t)w->file_off) : ::write(w->fd, eof_block, 28);
        if (k != 28) st = fail(PLO_ERR_IO, "write failed");
    }
    ::close(w->fd);
    delete w;
    return st;
}
