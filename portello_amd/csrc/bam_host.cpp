// bam_host.cpp -- host side of the path (include/portello_bam.h): BGZF/BAM input, batch construction from records
// (get_seq_order_read_split_segments), BAM record bytes of the lifted alignments, BGZF output.  Plain C++17 + zlib;
// no GPU code.  Citations are relative to /root/reference; rust-htslib / htslib semantics are restated from their
// published behaviour (third party, absent from the reference tree).
#include <errno.h>
#include <chrono>
#include <sys/uio.h>

#include "bam_internal.hpp"

namespace {
thread_local std::string g_bam_err;
}
void plo_bam_set_error(const std::string &msg) { g_bam_err = msg; }

// ---------------------------------------------------------------------------------------------------------------------
// objects
// ---------------------------------------------------------------------------------------------------------------------
struct plo_bam_reader {
    uint64_t hdr_bytes = 0;  // inflated size of the BAM header (magic .. last reference length)
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
    bool eof = false;           // the stream ended while this window was collected
    int batch_kind = 0;         // 0: no batch built yet, 1: plo_bam_window_batch (dense bases), 2: plo_bam_window_batch_sparse
    // batch (plo_batch_in) arrays
    HostBuf b_rev, b_len, b_soff, b_seq, b_seg_read, b_seg_contig, b_seg_pos, b_seg_fwd, b_coff, b_cigar, b_flags, b_qual, b_qoff, b_full_off;
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

plo_status plo_bam_open(const char *path, int n_threads, plo_bam_reader **out) { return plo_bam_open_device(path, n_threads, -2, out); }
plo_status plo_bam_open_device(const char *path, int n_threads, int device, plo_bam_reader **out) {
    if (!path || !out) return PLO_ERR_INVALID_ARG;
    *out = nullptr;
    plo_bam_reader *r = new plo_bam_reader();
    r->threads = std::max(1, n_threads);
    if (device >= 0) {  // BGZF blocks on that GPU from the first refill on (plo_bam_set_device_inflate)
        r->in.device = -2;
        r->in.dev_id = device;
    } else if (device == -1) {
        r->in.device = 0;
    }
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
    r->hdr_bytes = 12 + (uint64_t)l_text;  // inflated bytes of the header as the file has them (plo_bam_open_range skips exactly these)
    for (uint32_t i = 0; i < n_ref; ++i) {
        if ((st = r->in.read(w, 4)) != PLO_OK) return bail(st);
        uint32_t l_name = rd32(w);
        r->hdr_bytes += 8 + (uint64_t)l_name;
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

// One PART of a BAM file for one rank / worker (the reference gives every worker an IndexedReader and fetches its region,
// src/worker_thread_data.rs:21-30, src/read_alignment_scanner.rs:382; here the split needs no index): the compressed file is cut at
// size x part / n_parts, a part owns the records whose first byte lies in a BGZF block that STARTS inside its stretch, and reads on
// past its end to finish the last of them.  A part behind the first finds its first block (BgzfIn::seek_block) and then its first
// record: the offset in the inflated stream from which a chain of records parses, eight times in a row (or to the end of the data).
// The test asks for what plo_bam_read_window itself insists on (block_size >= 32, fixed fields + name + CIGAR + bases + qualities within
// block_size) and for what every BAM record has by the format's definition (reference ids inside the header's list, a NUL-terminated
// name, CIGAR op codes 0 .. 8) -- and for nothing else (a test stricter than the reader's -- printable names, positions >= -1, a
// size bound -- let a record the reader lifts break every chain through it, and the records in front of the start found behind it were
// read by no part).
static bool plausible_record(const uint8_t *p, size_t left, uint32_t n_ref, size_t *len) {
    if (left < 4 + 32) return false;
    const uint32_t bs = rd32(p);
    if (bs < 32 || 4 + (size_t)bs > left) return false;
    Rec rec{p + 4, bs};
    const int32_t tid = rec.tid(), mtid = (int32_t)rd32(p + 4 + 20);
    if (tid < -1 || tid >= (int32_t)n_ref || mtid < -1 || mtid >= (int32_t)n_ref) return false;
    const uint32_t l_name = rec.l_qname();
    if (l_name < 1 || !rec.layout_ok()) return false;
    const uint8_t *name = p + 4 + 32;
    if (name[l_name - 1] != 0) return false;
    const uint32_t n_cig = rec.n_cigar();
    const uint8_t *cg = name + l_name;
    for (uint32_t i = 0; i < n_cig; ++i)
        if ((rd32(cg + 4 * i) & 15u) > 8u) return false;
    *len = 4 + (size_t)bs;
    return true;
}

plo_status plo_bam_open_range(const char *path, int n_threads, int device, uint32_t part, uint32_t n_parts, plo_bam_reader **out) {
    if (!out || !n_parts || part >= n_parts) return PLO_ERR_INVALID_ARG;
    plo_status st = plo_bam_open_device(path, n_threads, device, out);
    if (st != PLO_OK) return st;
    plo_bam_reader *r = *out;
    BgzfIn &in = r->in;
    auto bail = [&](plo_status s) {
        plo_bam_close(r);
        *out = nullptr;
        return s;
    };
    const size_t lo = (size_t)((unsigned __int128)in.size * part / n_parts), hi = (size_t)((unsigned __int128)in.size * (part + 1) / n_parts);
    // Every BGZF block has one owner: the part whose stretch [lo, hi) holds the block's first byte.  The records start in the block that
    // holds the first byte behind the header; parts in front of that block's owner have nothing, the owner starts there, the parts behind
    // it look for their first block and their first record.
    in.ranged = true;
    in.range_end = part + 1 == n_parts ? (size_t)-1 : hi;
    const size_t hdr_bytes = (size_t)r->hdr_bytes;  // (from the l_name fields as read: a name without its NUL, or padded, must not move the start)
    size_t c = 0, u = 0;  // walk the blocks from the file's start until the inflated offset passes the header
    while (c < in.size) {
        const uint32_t bsz = BgzfIn::bgzf_block_at(in.map + c, in.size - c);
        if (!bsz) return bail(fail(PLO_ERR_IO, "not a BGZF block"));
        const uint32_t ulen = rd32(in.map + c + bsz - 4);
        if (u + ulen > hdr_bytes) break;
        u += ulen;
        c += bsz;
    }
    const size_t first_rec_block = c, skip = hdr_bytes - u;
    if (first_rec_block >= in.size || hi <= first_rec_block) {  // no records at all, or this part lies in front of them
        in.restart_at(in.size);
        return PLO_OK;
    }
    if (lo <= first_rec_block) {  // the owner of the first records' block
        in.restart_at(first_rec_block);
        if ((st = in.fill(skip + 4)) != PLO_OK) return bail(st);
        in.bpos += std::min(skip, in.avail());
        return PLO_OK;
    }
    const size_t blk = in.seek_block(lo);
    in.restart_at(blk);
    if (blk >= in.size || blk >= hi) {  // no block starts inside this part's stretch
        in.restart_at(in.size);
        return PLO_OK;
    }
    // the first record: a chain of eight plausible records (or plausible records to the end of what there is)
    const uint32_t n_ref = (uint32_t)r->names.size();
    size_t have = 0;
    for (size_t want = (size_t)4 << 20;; want *= 4) {
        if ((st = in.fill(want)) != PLO_OK) return bail(st);
        have = in.avail();
        const uint8_t *b = in.buf.data() + in.bpos;
        for (size_t p = 0; p + 36 <= have; ++p) {
            size_t q = p, len = 0;
            int ok = 0;
            bool ran_out = false;
            while (ok < 8) {
                if (q == have && in.eof) break;
                if (!plausible_record(b + q, have - q, n_ref, &len)) {
                    ran_out = q + 36 > have || (q + 4 <= have && 4 + (size_t)rd32(b + q) > have - q && rd32(b + q) >= 32);
                    break;
                }
                q += len;
                ++ok;
            }
            if (ok == 8 || (ok > 0 && q == have && in.eof)) {
                // records that start in a block of the NEXT part's stretch are not this part's: read_window checks every record
                in.bpos += p;
                return PLO_OK;
            }
            if (ran_out && !in.eof && ok > 0) break;  // a chain cut by the end of the buffered data: buffer more
        }
        if (in.eof || want > ((size_t)1 << 32)) break;
    }
    // No record boundary behind this part's first block.  The tail of one record that began in an earlier part's block is all there is
    // when little data is left; megabytes without a boundary are not a record's tail -- the part would come back empty and the parts'
    // union incomplete with status 0, so that is an error.
    if (have > ((size_t)64 << 20) || !in.eof)
        return bail(fail(PLO_ERR_DATA, "plo_bam_open_range: no BAM record boundary found behind the part's first BGZF block"));
    in.restart_at(in.size);
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
    const bool dbg = getenv("PLO_DEBUG_READER") != nullptr;
    const auto t_begin = std::chrono::steady_clock::now();
    double t_fill = 0;
    auto timed_fill = [&](size_t want) {
        if (!dbg) return r->in.fill(want);
        const auto a = std::chrono::steady_clock::now();
        plo_status s2 = r->in.fill(want);
        t_fill += std::chrono::duration<double>(std::chrono::steady_clock::now() - a).count();
        return s2;
    };
    // walk the records of the inflated stream (no copies), then take the whole stretch with one parallel copy
    size_t at = 0;  // offset from r->in.bpos
    std::vector<uint64_t> unm_at;
    // A window also ends after 4 x max_records unmapped records or 1 GB (or more, below) of records: the tail of unmapped reads of a sorted
    // read->contig BAM (often gigabytes) then comes in windows of bounded size instead of one
    // (the byte bound follows the window asked for -- 64 KB per record, at least 1 GB -- so that it cuts the unmapped tail and runs of
    // oversized records, not an ordinary window of HiFi reads: 60 000 records of 15 kb reads are 1.9 GB)
    const size_t unm_cap = 4 * (size_t)max_records + 1024;
    const size_t byte_cap = std::max<size_t>((size_t)1 << 30, std::min<size_t>((size_t)8 << 30, (size_t)max_records << 16));
    while (w->rec_at.size() < max_records) {
        if (unm_at.size() >= unm_cap || (at >= byte_cap && w->rec_at.size() + unm_at.size() > 0)) break;
        if (r->in.avail() < at + 4 && (st = timed_fill(at + 4)) != PLO_OK) break;
        if (r->in.avail() == at) {  // end of file
            w->eof = true;
            break;
        }
        if (r->in.avail() < at + 4) {
            st = fail(PLO_ERR_IO, "truncated BAM record");
            break;
        }
        if (r->in.ranged && r->in.block_file_off(at) >= r->in.range_end) {  // the record starts in the next part's stretch
            w->eof = true;
            break;
        }
        uint32_t bs = rd32(r->in.buf.data() + r->in.bpos + at);
        if (bs < 32) {
            st = fail(PLO_ERR_IO, "BAM record shorter than its fixed fields");
            break;
        }
        if (r->in.avail() < at + 4 + (size_t)bs && (st = timed_fill(at + 4 + (size_t)bs)) != PLO_OK) break;
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
        if ((flag & 0x4) && rec.tid() >= 0) {
            // an unmapped record placed on a contig (its mate's position): the reference's window loop asserts
            // !record.is_unmapped() (src/read_alignment_scanner.rs:396) -- its unmapped fetch (:544) only sees tid = -1 records
            st = fail(PLO_ERR_DATA, "unmapped record placed on a contig (flag 0x4 with a reference id): the reference aborts on it (read_alignment_scanner.rs:396)");
            break;
        }
        if (flag & 0x4) unm_at.push_back(at);                 // scan_unmapped_reads :551-555
        else if (!(flag & 0x800)) w->rec_at.push_back(at);    // :404 supplementary records are reached through the primary's SA tag
        at += 4 + (size_t)bs;
    }
    if (st == PLO_OK && !w->raw.resize(at)) st = fail(PLO_ERR_OUT_OF_MEMORY, "out of host memory for a window of records");
    if (st != PLO_OK) {
        delete w;
        return st;
    }
    const auto t_walk = std::chrono::steady_clock::now();
    parallel_copy(w->raw.data(), r->in.buf.data() + r->in.bpos, at, r->threads);
    r->in.bpos += at;
    if (dbg) {
        const auto t_end = std::chrono::steady_clock::now();
        fprintf(stderr, "[plo] read_window: %zu records, %.1f MB: inflate (fill) %.3f s, record walk %.3f s, copy %.3f s\n", w->rec_at.size(), at / 1e6,
                t_fill, std::chrono::duration<double>(t_walk - t_begin).count() - t_fill, std::chrono::duration<double>(t_end - t_walk).count());
    }
    for (uint64_t u : unm_at) {
        const uint8_t *p = w->raw.data() + u;
        w->unmapped.insert(w->unmapped.end(), p, p + 4 + rd32(p));
        ++w->n_unmapped;
    }
    *out = w;
    return PLO_OK;
}

void plo_bam_set_device_inflate(plo_bam_reader *r, int device) {
    if (!r) return;
    r->in.device = device >= 0 ? -2 : 0;  // decided at the next refill (PLO_BGZF_DEVICE in the environment overrides)
    r->in.dev_id = device >= 0 ? device : 0;
}
void plo_bam_window_free(plo_bam_window *w) { delete w; }
uint32_t plo_bam_window_n_records(const plo_bam_window *w) { return w ? w->n_records() : 0; }
int plo_bam_window_eof(const plo_bam_window *w) { return (w && w->eof) ? 1 : 0; }
plo_status plo_bam_window_record(const plo_bam_window *w, uint32_t i, const uint8_t **bytes, uint32_t *n_bytes) {
    if (!w || !bytes || !n_bytes || i >= w->n_records()) return PLO_ERR_INVALID_ARG;
    const uint8_t *p = w->raw.data() + w->rec_at[i];
    *bytes = p;
    *n_bytes = 4 + rd32(p);
    return PLO_OK;
}
void plo_bam_window_unmapped(const plo_bam_window *w, const uint8_t **bytes, uint64_t *n_bytes, uint32_t *n_records) {
    if (bytes) *bytes = w ? w->unmapped.data() : nullptr;
    if (n_bytes) *n_bytes = w ? w->unmapped.size() : 0;
    if (n_records) *n_records = w ? w->n_unmapped : 0;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------------------------------
// PLO_SEQ_BAM4_SPARSE: only the bases around the indels of the read->contig CIGARs travel to the device
// ---------------------------------------------------------------------------------------------------------------------
namespace {

inline uint32_t sparse_hdr_bytes(uint32_t len) { return ((((len + 1023u) >> 10) * 8u) + 15u) & ~15u; }
inline uint32_t sparse_words(uint32_t len) { return (len + 1023u) >> 10; }

// Granule mask of one read: every I / D op of every segment's CIGAR marks its read interval, widened by `margin` bases on both
// sides, in the coordinates of the stored sequence (a segment on the other strand than the record counts from the far end:
// `changes`, src/read_alignment_scanner.rs:153-157).  These are the places the homology probes of left_shift_indels and the
// cluster trimming of simplify_alignment_indels start from; probes that run further are caught on the device.
// `seg_contig` + `seg_pos` + `ixd` (optional): a read segment that touches no reverse-mapped contig segment (the overlap rule of
// get_contig_split_segments_from_read_mapping, src/read_alignment_scanner.rs:80-103) never goes through the left shift (its homology
// probes are what the margins and the deletions' flanks are for); the only bases the path can compare there are the inserted ones of a
// complex indel cluster (simplify_alignment_indels.rs:55-85) -- its insertions' own intervals (+ 16 bases) are marked, nothing else.
void sparse_mark(uint32_t len, bool read_rev, uint32_t s0, uint32_t s1, const uint8_t *seg_fwd, const uint32_t *coff, const uint32_t *cigar,
                 uint32_t margin, uint32_t *mask, const uint32_t *seg_contig = nullptr, const int64_t *seg_pos = nullptr, const plo_index_desc *ixd = nullptr) {
    const uint32_t nw = sparse_words(len);
    for (uint32_t k = 0; k < nw; ++k) mask[k] = 0;
    if (!len) return;
    for (uint32_t s = s0; s < s1; ++s) {
        const bool changes = read_rev == (seg_fwd[s] != 0);
        bool ins_only = false;
        if (ixd && seg_contig && seg_pos && seg_contig[s] < ixd->n_contigs) {
            int64_t ref_len = 0;
            for (uint32_t o = coff[s]; o < coff[s + 1]; ++o)
                if ((0x18Du >> (cigar[o] & 15u)) & 1u) ref_len += (int64_t)(cigar[o] >> 4);
            const int64_t r_start = seg_pos[s], r_end = r_start + ref_len;
            ins_only = true;
            for (uint32_t g = ixd->contig_seg_off[seg_contig[s]]; g < ixd->contig_seg_off[seg_contig[s] + 1]; ++g)
                if (r_end >= ixd->seg_seq_order_start[g] && r_start < ixd->seg_seq_order_end[g] && !ixd->seg_is_fwd_strand[g]) ins_only = false;
        }
        const uint32_t margin_s = ins_only ? std::min(margin, 16u) : margin;  // (the comparisons read 16-base windows)
        int64_t q = 0;
        for (uint32_t o = coff[s]; o < coff[s + 1]; ++o) {
            const uint32_t t = cigar[o] & 15u, L = cigar[o] >> 4;
            if ((t == 1 || t == 2) && !(ins_only && t == 2)) {
                int64_t a = q, b = q + (t == 1 ? (int64_t)L : 0);
                if (changes) {
                    const int64_t a2 = (int64_t)len - b;
                    b = (int64_t)len - a;
                    a = a2;
                }
                a = std::max<int64_t>(0, a - (int64_t)margin_s);
                b = std::min<int64_t>((int64_t)len, b + (int64_t)margin_s);
                if (b > a)
                    for (int64_t g = a >> 5; g <= (b - 1) >> 5; ++g) mask[g >> 5] |= 1u << (g & 31);
            }
            if ((0x1B3u >> t) & 1u) q += L;
        }
    }
}
uint32_t sparse_granules(uint32_t len, const uint32_t *mask) {
    uint32_t n = 0;
    for (uint32_t k = 0; k < sparse_words(len); ++k) n += (uint32_t)__builtin_popcount(mask[k]);
    return n;
}
// header + present granules of one read at dst (sparse_hdr_bytes(len) + 16 * granules bytes)
void sparse_write(uint32_t len, const uint8_t *dense, const uint32_t *mask, uint8_t *dst) {
    const uint32_t nw = sparse_words(len), hb = sparse_hdr_bytes(len);
    const size_t nbytes = ((size_t)len + 1) / 2;
    uint32_t *hdr = (uint32_t *)dst;
    uint8_t *out = dst + hb;
    uint32_t rank = 0;
    for (uint32_t k = 0; k < hb / 4; ++k) hdr[k] = 0;
    for (uint32_t k = 0; k < nw; ++k) {
        hdr[2 * k] = mask[k];
        hdr[2 * k + 1] = rank;
        for (uint32_t m = mask[k]; m; m &= m - 1) {
            const size_t g = (size_t)k * 32 + (size_t)__builtin_ctz(m), at = g * 16;
            const size_t nb = at < nbytes ? std::min<size_t>(16, nbytes - at) : 0;
            memcpy(out + (size_t)rank * 16, dense + at, nb);
            if (nb < 16) memset(out + (size_t)rank * 16 + nb, 0, 16 - nb);
            ++rank;
        }
    }
}

// the reads' segment ranges (segments grouped by read, as get_seq_order_read_split_segments yields them)
bool read_seg_ranges(uint32_t n_reads, uint32_t n_segs, const uint32_t *seg_read, std::vector<uint32_t> &first) {
    first.assign((size_t)n_reads + 1, 0);
    for (uint32_t s = 0; s < n_segs; ++s) {
        if (seg_read[s] >= n_reads || (s && seg_read[s] < seg_read[s - 1])) return false;
        ++first[seg_read[s] + 1];
    }
    for (uint32_t i = 0; i < n_reads; ++i) first[i + 1] += first[i];
    return true;
}

constexpr uint64_t SPARSE_SLACK = 32;  // zero bytes behind the last granule: the wide window loads of the probes end inside the buffer

}  // namespace

extern "C" uint64_t plo_sparse_seq_bound(const plo_batch_in *dense) {
    if (!dense) return 0;
    uint64_t b = SPARSE_SLACK;
    for (uint32_t i = 0; i < dense->n_reads; ++i) {
        const uint32_t len = dense->read_seq_len[i];
        b += sparse_hdr_bytes(len) + 16ull * (((uint64_t)len + 31) >> 5);
    }
    return b;
}

extern "C" plo_status plo_sparse_seq_pack(const plo_batch_in *dense, uint32_t margin, int n_threads, uint8_t *out, uint64_t out_cap,
                                          uint64_t *out_read_off, plo_batch_in *sparse) {
    if (!dense || !out || !out_read_off || !sparse) return PLO_ERR_INVALID_ARG;
    if (dense->seq_fmt != PLO_SEQ_BAM4) return fail(PLO_ERR_INVALID_ARG, "plo_sparse_seq_pack: the batch must carry dense BAM 4-bit bases");
    const uint32_t n = dense->n_reads;
    std::vector<uint32_t> first;
    if (!read_seg_ranges(n, dense->n_segs, dense->seg_read, first))
        return fail(PLO_ERR_INVALID_ARG, "plo_sparse_seq_pack: seg_read must be non-decreasing and below n_reads");
    if (n && (!dense->read_seq_len || !dense->read_seq_off || !dense->read_is_reverse || !dense->seq))
        return fail(PLO_ERR_INVALID_ARG, "plo_sparse_seq_pack: NULL read array");
    if (dense->n_segs && (!dense->seg_is_fwd_strand || !dense->seg_cigar_off || !dense->cigar))
        return fail(PLO_ERR_INVALID_ARG, "plo_sparse_seq_pack: NULL segment array");
    for (uint32_t i = 0; i < n; ++i) {
        const uint64_t nb = ((uint64_t)dense->read_seq_len[i] + 1) / 2, off = dense->read_seq_off[i];
        if (off > dense->seq_bytes || nb > dense->seq_bytes - off) return fail(PLO_ERR_INVALID_ARG, "plo_sparse_seq_pack: a read's bases lie outside `seq`");
    }
    for (uint32_t s = 0; s < dense->n_segs; ++s)
        if (dense->seg_cigar_off[s + 1] < dense->seg_cigar_off[s]) return fail(PLO_ERR_INVALID_ARG, "plo_sparse_seq_pack: seg_cigar_off is not monotonic");
    std::vector<uint64_t> woff((size_t)n + 1, 0), boff((size_t)n + 1, 0);
    for (uint32_t i = 0; i < n; ++i) woff[i + 1] = woff[i] + sparse_words(dense->read_seq_len[i]);
    std::vector<uint32_t> masks(woff[n] + 1);
    const int th = std::max(1, n_threads);
    parallel_ranges(n, th, [&](size_t lo, size_t hi) {
        for (size_t i = lo; i < hi; ++i) {
            const uint32_t len = dense->read_seq_len[i];
            sparse_mark(len, dense->read_is_reverse[i] != 0, first[i], first[i + 1], dense->seg_is_fwd_strand, dense->seg_cigar_off, dense->cigar,
                        margin, masks.data() + woff[i]);
            boff[i + 1] = sparse_hdr_bytes(len) + 16ull * sparse_granules(len, masks.data() + woff[i]);
        }
    });
    for (uint32_t i = 0; i < n; ++i) boff[i + 1] += boff[i];
    const uint64_t total = boff[n] + SPARSE_SLACK;
    if (total > out_cap) return fail(PLO_ERR_INVALID_ARG, "plo_sparse_seq_pack: output buffer smaller than plo_sparse_seq_bound");
    parallel_ranges(n, th, [&](size_t lo, size_t hi) {
        for (size_t i = lo; i < hi; ++i) {
            out_read_off[i] = boff[i];
            sparse_write(dense->read_seq_len[i], dense->seq + dense->read_seq_off[i], masks.data() + woff[i], out + boff[i]);
        }
    });
    memset(out + boff[n], 0, SPARSE_SLACK);
    *sparse = *dense;
    sparse->seq = out;
    sparse->seq_bytes = total;
    sparse->seq_fmt = PLO_SEQ_BAM4_SPARSE;
    sparse->read_seq_off = out_read_off;
    sparse->seq_full = dense->seq;
    sparse->read_seq_full_off = dense->read_seq_off;
    return PLO_OK;
}

static plo_status window_batch(plo_bam_window *w, plo_batch_in *batch, plo_finish_in *fin, int sparse_margin, const plo_index_desc *ixd = nullptr);
extern "C" plo_status plo_bam_window_batch(plo_bam_window *w, plo_batch_in *batch, plo_finish_in *fin) { return window_batch(w, batch, fin, -1); }
extern "C" plo_status plo_bam_window_batch_sparse(plo_bam_window *w, uint32_t margin, plo_batch_in *batch, plo_finish_in *fin) {
    return window_batch(w, batch, fin, (int)std::min<uint32_t>(margin, 1u << 20));
}
extern "C" plo_status plo_bam_window_batch_sparse_strand(plo_bam_window *w, uint32_t margin, const plo_index_desc *index, plo_batch_in *batch, plo_finish_in *fin) {
    if (index && (!index->contig_seg_off || !index->seg_seq_order_start || !index->seg_seq_order_end || !index->seg_is_fwd_strand))
        return fail(PLO_ERR_INVALID_ARG, "plo_bam_window_batch_sparse_strand: the index description lacks its segment arrays");
    return window_batch(w, batch, fin, (int)std::min<uint32_t>(margin, 1u << 20), index);
}

static plo_status window_batch(plo_bam_window *w, plo_batch_in *batch, plo_finish_in *fin, int sparse_margin, const plo_index_desc *ixd) {
    if (!w || !batch) return PLO_ERR_INVALID_ARG;
    memset(batch, 0, sizeof(*batch));
    const bool sparse = sparse_margin >= 0;
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
        plo_status st = split_segments(w->reader->label_to_index, rec, segs[i], pcig[i], err);
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
    uint8_t *seq = sparse ? nullptr : (uint8_t *)w->b_seq.ensure(std::max<uint64_t>(n_seqb[n], 16));
    uint64_t *full_off = sparse ? (uint64_t *)w->b_full_off.ensure(std::max<size_t>(n, 1) * 8) : soff;
    uint32_t *seg_read = (uint32_t *)w->b_seg_read.ensure(std::max<size_t>(ns, 1) * 4);
    uint32_t *seg_contig = (uint32_t *)w->b_seg_contig.ensure(std::max<size_t>(ns, 1) * 4);
    int64_t *seg_pos = (int64_t *)w->b_seg_pos.ensure(std::max<size_t>(ns, 1) * 8);
    uint8_t *seg_fwd = (uint8_t *)w->b_seg_fwd.ensure(std::max<size_t>(ns, 1));
    uint32_t *coff = (uint32_t *)w->b_coff.ensure(((size_t)ns + 1) * 4);
    uint32_t *cigar = (uint32_t *)w->b_cigar.ensure(std::max<uint64_t>(n_ops[n], 1) * 4);
    uint16_t *flags = (uint16_t *)w->b_flags.ensure(std::max<size_t>(n, 1) * 2);
    uint8_t *qual = fin ? (uint8_t *)w->b_qual.ensure(std::max<uint64_t>(n_qual[n], 16)) : nullptr;
    uint64_t *qoff = fin ? (uint64_t *)w->b_qoff.ensure(std::max<size_t>(n, 1) * 8) : nullptr;
    if (!rev || !rlen || !soff || !full_off || (!sparse && !seq) || !seg_read || !seg_contig || !seg_pos || !seg_fwd || !coff || !cigar || !flags || (fin && (!qual || !qoff)))
        return fail(PLO_ERR_OUT_OF_MEMORY, "out of host memory for the window's batch");
    // pass 2 (parallel): fill
    parallel_for(n, w->threads, [&](size_t i) {
        Rec rec = w->record((uint32_t)i);
        rev[i] = (rec.flag() & 0x10) ? 1 : 0;
        rlen[i] = rec.l_seq();
        flags[i] = rec.flag();
        if (sparse) {  // the complete bases stay where they are, inside the window's copy of the records
            full_off[i] = (uint64_t)(rec.seq() - w->raw.data());
        } else {
            soff[i] = n_seqb[i];
            memcpy(seq + n_seqb[i], rec.seq(), (size_t)(n_seqb[i + 1] - n_seqb[i]));
        }
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
    uint64_t seq_bytes = n_seqb[n];
    if (sparse) {
        // pass 3 (parallel): granule masks from the CIGARs just written, then headers + granules straight from the records
        std::vector<uint64_t> woff((size_t)n + 1, 0), boff((size_t)n + 1, 0);
        for (uint32_t i = 0; i < n; ++i) woff[i + 1] = woff[i] + sparse_words(rlen[i]);
        std::vector<uint32_t> masks(woff[n] + 1);
        parallel_ranges(n, w->threads, [&](size_t lo, size_t hi) {
            for (size_t i = lo; i < hi; ++i) {
                sparse_mark(rlen[i], rev[i] != 0, n_seg[i], n_seg[i + 1], seg_fwd, coff, cigar, (uint32_t)sparse_margin, masks.data() + woff[i], seg_contig, seg_pos, ixd);
                boff[i + 1] = sparse_hdr_bytes(rlen[i]) + 16ull * sparse_granules(rlen[i], masks.data() + woff[i]);
            }
        });
        for (uint32_t i = 0; i < n; ++i) boff[i + 1] += boff[i];
        seq_bytes = boff[n] + SPARSE_SLACK;
        seq = (uint8_t *)w->b_seq.ensure(seq_bytes);
        if (!seq) return fail(PLO_ERR_OUT_OF_MEMORY, "out of host memory for the window's batch");
        parallel_ranges(n, w->threads, [&](size_t lo, size_t hi) {
            for (size_t i = lo; i < hi; ++i) {
                soff[i] = boff[i];
                sparse_write(rlen[i], w->raw.data() + full_off[i], masks.data() + woff[i], seq + boff[i]);
            }
        });
        memset(seq + boff[n], 0, SPARSE_SLACK);
        batch->seq_full = w->raw.data();
        batch->read_seq_full_off = full_off;
    }
    batch->n_reads = n;
    batch->read_is_reverse = rev;
    batch->read_seq_len = rlen;
    batch->read_seq_off = soff;
    batch->seq = seq;
    batch->seq_bytes = seq_bytes;
    batch->seq_fmt = sparse ? PLO_SEQ_BAM4_SPARSE : PLO_SEQ_BAM4;
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
    w->batch_kind = sparse ? 2 : 1;
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

// `fin` / `sa` (host copies of plo_finish_batch_dev's / plo_sa_segments_dev's results, or null): flags, bin, primary record,
// reversed bases and qualities and the SA segments come from there and are only copied into place; else they are made here.
static plo_status records_build(plo_bam_window *w, const plo_batch_out *lift, const plo_finish_out *fin, const plo_sa_out *sa, const plo_records_params *pr,
                                plo_record_buf *out) {
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
        else if (lift->item_status[i] == PLO_ITEM_NEED_BASES)
            return fail(PLO_ERR_INVALID_ARG, "plo_records_build: an item is PLO_ITEM_NEED_BASES (sparse bases lifted without seq_full): it has no result yet");

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
    std::atomic<bool> fin_bad{false};
    parallel_for(n, threads, [&](size_t r) {
        Rec rec = w->record((uint32_t)r);
        const uint32_t lo = seg_item_lo[w->read_seg_off[r]], hi = seg_item_lo[w->read_seg_off[r + 1]];
        uint32_t n_lift = 0, prim = UINT32_MAX;
        if (fin) {
            // the finished arrays must describe this very result: the record counts (the output is laid out by them) and every
            // offset into the reversed bases / qualities / SA text are checked before anything is copied from there
            for (uint32_t i = lo; i < hi; ++i) n_lift += item_lifted(i) ? 1 : 0;
            const uint64_t sb = (uint64_t)(rec.l_seq() + 1) / 2, qb = rec.l_seq();
            bool ok = n_lift == fin->read_n_lifted[r];
            if (n_lift == 0)
                ok = ok && (fin->read_seq_off[r] == PLO_NO_FLIP ||
                            (fin->read_seq_off[r] <= fin->rev_seq_bytes && sb <= fin->rev_seq_bytes - fin->read_seq_off[r] &&
                             fin->read_qual_off[r] <= fin->rev_qual_bytes && qb <= fin->rev_qual_bytes - fin->read_qual_off[r]));
            for (uint32_t i = lo; i < hi && ok; ++i) {
                if (!item_lifted(i)) continue;
                ok = fin->item_seq_off[i] == PLO_NO_FLIP ||
                     (fin->item_seq_off[i] <= fin->rev_seq_bytes && sb <= fin->rev_seq_bytes - fin->item_seq_off[i] &&
                      fin->item_qual_off[i] <= fin->rev_qual_bytes && qb <= fin->rev_qual_bytes - fin->item_qual_off[i]);
                // a record has flipped bases exactly when the lift says it needs them (a stale array would give silently wrong bases)
                // (a read without bases has nothing to flip: the finishing kernel leaves PLO_NO_FLIP there whatever the lift says)
                ok = ok && (rec.l_seq() == 0 || (fin->item_seq_off[i] != PLO_NO_FLIP) == (lift->item_need_flipped[i] != 0));
                if (ok && sa) ok = sa->item_sa_off[i] <= sa->item_sa_off[i + 1] && sa->item_sa_off[i + 1] <= sa->sa_bytes;
            }
            if (!ok) {
                fin_bad.store(true);
                return;
            }
        } else {
            for (uint32_t i = lo; i < hi; ++i) {
                if (!item_lifted(i)) continue;
                ++n_lift;
                if (prim == UINT32_MAX || lift->item_mapq[prim] < lift->item_mapq[i]) prim = i;  // :338-345 first maximum wins
            }
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
                info[i].flag = fin ? fin->item_flag[i] : fl;
                const uint32_t seg = lift->item_seg[i];
                const uint32_t contig = seg_contig[seg];
                info[i].ps_len = (uint32_t)(strlen(pr->contig_names[contig]) + 6 + decimal_len(lift->item_cseg[i]) + 1);  // "{}_split{}{+|-}"
                // "{chrom},{pos+1},{strand},{cigar},{mapq},0;" (:292-301)
                const uint32_t *cg = lift->cigar + lift->item_cigar_off[i];
                if (sa)
                    info[i].sa_len = sa->item_sa_off[i + 1] - sa->item_sa_off[i];
                else if (n_lift > 1)
                    info[i].sa_len = (uint32_t)(strlen(pr->ref_names[lift->item_chrom_index[i]]) + 1 + decimal_len((uint64_t)(lift->item_ref_pos[i] + 1)) +
                                                1 + 1 + 1 + cigar_text_len(cg, lift->item_cigar_len[i]) + 1 + decimal_len(lift->item_mapq[i]) + 3);
                else
                    info[i].sa_len = 0;
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
    if (fin_bad.load())
        return fail(PLO_ERR_INVALID_ARG, "plo_records_build_finished: the finished arrays do not belong to this lift result (record counts or offsets into the reversed "
                                         "bases / qualities / SA text disagree)");
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
        // (fin: the flipped bases and qualities were made on the device, k_revcomp)
        auto put_flipped = [&](uint8_t *q, uint64_t seq_off, uint64_t qual_off) -> uint8_t * {
            memcpy(q, fin->rev_seq + seq_off, seqb);
            memcpy(q + seqb, fin->rev_qual + qual_off, l_seq);
            return q + seqb + l_seq;
        };
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
            if (fin) fl = fin->read_unmapped_flag[r];
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
            q = fin && fin->read_seq_off[r] != PLO_NO_FLIP ? put_flipped(q, fin->read_seq_off[r], fin->read_qual_off[r]) : put_seq_qual(q, fin ? false : flip);
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
            if (fin)
                ref_len = fin->item_ref_end[i] - pos;
            else
                for (uint32_t c = 0; c < nc; ++c) ref_len += op_ref_len(cg[c]);
            wr32(b, lift->item_chrom_index[i]);
            wr32(b + 4, (uint32_t)(int32_t)pos);
            b[8] = (uint8_t)rec.l_qname();
            b[9] = lift->item_mapq[i];
            wr16(b + 10, fin ? fin->item_bin[i] : reg2bin((uint64_t)pos, (uint64_t)(pos + ref_len)));  // :278-279
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
            q = fin && fin->item_seq_off[i] != PLO_NO_FLIP ? put_flipped(q, fin->item_seq_off[i], fin->item_qual_off[i])
                                                           : put_seq_qual(q, fin ? false : lift->item_need_flipped[i] != 0);
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
                    if (sa) {  // (k_sa_text's segment of record j)
                        memcpy(q, sa->sa_text + sa->item_sa_off[j], info[j].sa_len);
                        q += info[j].sa_len;
                        continue;
                    }
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

extern "C" plo_status plo_records_build(plo_bam_window *w, const plo_batch_out *lift, const plo_records_params *pr, plo_record_buf *out) {
    return records_build(w, lift, nullptr, nullptr, pr, out);
}

extern "C" plo_status plo_records_build_finished(plo_bam_window *w, const plo_batch_out *lift, const plo_finish_out *fin, const plo_sa_out *sa,
                                                 const plo_records_params *pr, plo_record_buf *out) {
    if (!fin || !fin->item_flag || !fin->item_bin || !fin->item_ref_end || !fin->item_seq_off || !fin->item_qual_off || !fin->read_n_lifted ||
        !fin->read_unmapped_flag || !fin->read_seq_off || !fin->read_qual_off || ((fin->rev_seq_bytes || fin->rev_qual_bytes) && (!fin->rev_seq || !fin->rev_qual)))
        return fail(PLO_ERR_INVALID_ARG, "plo_records_build_finished: incomplete plo_finish_out (host copies of every array are needed)");
    if (sa && (sa->n_items != lift->n_items || !sa->item_sa_off || (sa->sa_bytes && !sa->sa_text)))
        return fail(PLO_ERR_INVALID_ARG, "plo_records_build_finished: plo_sa_out does not belong to this result");
    // the finished arrays carry their extents (API version 4): arrays of another (smaller) batch are refused before they are indexed
    if (!w || !lift || fin->n_items != lift->n_items || fin->n_reads != w->n_records())
        return fail(PLO_ERR_INVALID_ARG, "plo_records_build_finished: the finished arrays do not belong to this lift result (plo_finish_out::n_items / n_reads "
                                         "differ from the result's items / the window's reads)");
    if (w->batch_kind == 2)
        return fail(PLO_ERR_INVALID_ARG, "plo_records_build_finished: the window's batch was built with sparse bases (plo_bam_window_batch_sparse); records "
                                         "finished on the device need the dense batch (plo_bam_window_batch)");
    return records_build(w, lift, fin, sa, pr, out);
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
    // The file's blocks are reserved ahead of the writes (FALLOC_FL_KEEP_SIZE, a gigabyte or more at a time; what is left over is given back
    // when the writer closes; PLO_BGZF_FALLOCATE=0 switches it off, a filesystem without fallocate does so by itself).  Buffered writes into ONE file are serialised by the inode's lock whatever the number of
    // threads (tools/write_bench.cpp on the GPU box: 9.5 GB/s into one file, 61-126 GB/s into one file per thread); reserved blocks shorten
    // the time under the lock by the allocation (10.0-10.8 GB/s there).
    // Files below 64 MB reserve nothing (on a memory-backed filesystem a reservation is real, zero-filled memory); from there on the step
    // grows with the file, a quarter of a gigabyte to two (PLO_BGZF_FALLOCATE=2: from the first byte, for the tests).
    uint64_t reserved = 0;
    int falloc = -1;  // -1 undecided
    void reserve(uint64_t upto) {
        if (!seekable || falloc == 0) return;
        if (falloc < 0) {
            const char *e = getenv("PLO_BGZF_FALLOCATE");
            falloc = e ? std::max(0, atoi(e)) : 1;
            if (!falloc) return;
        }
        if (upto <= reserved || (falloc == 1 && upto < ((uint64_t)64 << 20))) return;
        const uint64_t to = upto + std::min<uint64_t>((uint64_t)2 << 30, std::max<uint64_t>((uint64_t)256 << 20, upto));
        if (fallocate(fd, FALLOC_FL_KEEP_SIZE, (off_t)reserved, (off_t)(to - reserved)) != 0) {
            falloc = 0;  // (a filesystem without it: nothing is lost)
            return;
        }
        reserved = to;
    }
    plo_status put(const uint8_t *src, size_t n);
};

// Blocks are built in parallel, each in its own slot of a scratch buffer, and written with positional writes from several
// threads when the output is a regular file (page-cache copies scale with the writers), in order otherwise (pipe / stdout).
plo_status plo_bam_writer::emit(const uint8_t *src, size_t n) {
    const size_t nblk = (n + BLOCK - 1) / BLOCK;
    if (!nblk) return PLO_OK;
    reserve(file_off + (uint64_t)nblk * (18 + 5 + BLOCK + 8 + 1024));
    if (level == 0 && seekable && !getenv("PLO_BGZF_COPY_BLOCKS")) {
        // Stored blocks into a regular file: nothing is copied in user space -- every block goes out as three pieces of one gather write
        // (its 23 header bytes, its <= 65 280 data bytes where they lie in the caller's buffer, its 8 trailer bytes); round 4 built the
        // blocks in a scratch buffer first, a second pass over the 23.6 kB a read's record weighs
        std::vector<uint8_t> hf(nblk * 32);
        parallel_ranges(nblk, threads, [&](size_t lo, size_t hi) {
            for (size_t b = lo; b < hi; ++b) {
                const uint8_t *in = src + b * BLOCK;
                const size_t len = std::min(BLOCK, n - b * BLOCK);
                uint8_t *o = hf.data() + b * 32;
                static const uint8_t hdr[16] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0};
                memcpy(o, hdr, 16);
                wr16(o + 16, (uint16_t)(18 + 5 + len + 8 - 1));
                o[18] = 1;  // final stored block
                wr16(o + 19, (uint16_t)len);
                wr16(o + 21, (uint16_t)~len);
                wr32(o + 23, fast_crc32(in, len));
                wr32(o + 27, (uint32_t)len);
            }
        });
        std::atomic<int> wbad{0};
        const size_t group = 256;  // (3 pieces per block: 768 of the 1 024 an iovec array may hold)
        const size_t ng = (nblk + group - 1) / group;
        parallel_for(ng, std::min(threads, 16), [&](size_t g) {
            const size_t b0 = g * group, b1 = std::min(nblk, b0 + group);
            std::vector<struct iovec> iov;
            iov.reserve(3 * (b1 - b0));
            for (size_t b = b0; b < b1; ++b) {
                const size_t len = std::min(BLOCK, n - b * BLOCK);
                iov.push_back({hf.data() + b * 32, 23});
                iov.push_back({(void *)(src + b * BLOCK), len});
                iov.push_back({hf.data() + b * 32 + 23, 8});
            }
            uint64_t off = file_off + (uint64_t)b0 * (18 + 5 + BLOCK + 8);  // (every block in front of b0 is a full one)
            size_t first = 0;
            while (first < iov.size()) {
                const int cnt = (int)std::min<size_t>(iov.size() - first, 1024);
                ssize_t k = pwritev(fd, iov.data() + first, cnt, (off_t)off);
                if (k < 0 && errno == EINTR) continue;  // (a signal -- a watchdog, a profiler's timer -- is not a failed write)
                if (k <= 0) {
                    wbad = 1;
                    return;
                }
                off += (uint64_t)k;
                while (k > 0 && first < iov.size()) {  // pieces written whole are done; a piece written in part continues
                    if ((size_t)k >= iov[first].iov_len) {
                        k -= (ssize_t)iov[first].iov_len;
                        ++first;
                    } else {
                        iov[first].iov_base = (uint8_t *)iov[first].iov_base + k;
                        iov[first].iov_len -= (size_t)k;
                        k = 0;
                    }
                }
            }
        });
        if (wbad) return fail(PLO_ERR_IO, "write failed");
        file_off += (uint64_t)(nblk - 1) * (18 + 5 + BLOCK + 8) + (18 + 5 + std::min(BLOCK, n - (nblk - 1) * BLOCK) + 8);
        return PLO_OK;
    }
    const size_t slot = level == 0 ? 18 + 5 + BLOCK + 8 : 18 + BLOCK + 1024 + 8;
    if (!scratch.resize(nblk * slot)) return fail(PLO_ERR_OUT_OF_MEMORY, "out of host memory for BGZF output blocks");
    std::vector<uint32_t> olen(nblk, 0);
    std::atomic<int> bad{0};
    uint8_t *outb = scratch.data();
    const bool dbg = getenv("PLO_DEBUG_WRITER") != nullptr;
    const auto t0 = std::chrono::steady_clock::now();
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
    const auto t1 = std::chrono::steady_clock::now();
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
                    if (k < 0 && errno == EINTR) continue;
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
                if (k < 0 && errno == EINTR) continue;
                if (k <= 0) return fail(PLO_ERR_IO, "write failed");
                p += k;
                left -= (size_t)k;
            }
        }
    }
    file_off += at[nblk];
    if (dbg)
        fprintf(stderr, "[plo] bgzf write: %.1f MB in %zu blocks: build %.3f s, write %.3f s\n", n / 1e6, nblk, std::chrono::duration<double>(t1 - t0).count(),
                std::chrono::duration<double>(std::chrono::steady_clock::now() - t1).count());
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
    uint64_t end = w->file_off;  // bytes actually written: the EOF block counts only when it went out
    if (st == PLO_OK) {
        ssize_t k;
        do k = w->seekable ? pwrite(w->fd, eof_block, 28, (off_t)w->file_off) : ::write(w->fd, eof_block, 28);
        while (k < 0 && errno == EINTR);
        if (k != 28) st = fail(PLO_ERR_IO, "write failed");
        else end += 28;
    }
    if (w->seekable && w->reserved > end && ftruncate(w->fd, (off_t)end) != 0 && st == PLO_OK)  // reserved blocks behind the end go back
        st = fail(PLO_ERR_IO, "ftruncate failed");
    ::close(w->fd);
    delete w;
    return st;
}
