// finish_core.hpp -- record finishing: the arithmetic of /root/reference/src/read_alignment_scanner.rs:245-284 and
// :310-346 (flags, alignment end, bin, primary selection, unmapped copy) and reverse_alignment_seq_and_qual (:125-133).
// The sequence/quality reversal is the HBM-streaming step of the path (~22.5 kB per flipped 15 kb read).
#pragma once
#include <plo_wave.hpp>
#include <stdint.h>

#include "lift_core.hpp"
#include "lift_types.hpp"

namespace plo {

struct DevFinish {
    // inputs
    const uint16_t *read_flags;
    const uint8_t *qual;
    const uint64_t *read_qual_off;
    uint64_t qual_bytes, seq_bytes;
    // per item
    uint16_t *item_flag, *item_bin;
    int64_t *item_ref_end;
    uint8_t *item_is_primary;
    uint64_t *item_seq_off, *item_qual_off;
    uint32_t *item_read;
    // per read
    uint32_t *read_n_lifted, *read_primary_item;
    uint16_t *read_unmapped_flag;
    uint64_t *read_seq_off, *read_qual_off_out;
    // entries = items then reads: sizes in 16-byte units and their exclusive scans
    uint32_t *su, *qu;
    const uint32_t *soff, *qoff;
    uint32_t *fflag;         // 1 per entry that needs the reversal (scanned into frank)
    const uint32_t *frank;
    uint32_t *flist;         // compact list of those entries
    uint8_t *rev_seq, *rev_qual;
    unsigned *n_fault;       // items that ended LEN_MISMATCH / PANIC (the reference aborts: the batch must not be written)
};

// hts_reg2bin / bam_reg2bin (lib/rust-vc-utils/src/bam_utils/util.rs:10-35)
PLO_DEV uint16_t bam_reg2bin(unsigned long long begin, unsigned long long end) {
    end = end - 1;
    unsigned l = 5, s = 14;
    unsigned long long t = ((1ull << 15) - 1) / 7;
    while (l > 0) {
        if ((begin >> s) == (end >> s)) return (uint16_t)(t + (begin >> s));
        l -= 1;
        s += 3;
        t -= 1ull << (l * 3);
    }
    return 0;
}

PLO_DEV uint32_t units16(unsigned long long bytes) { return (uint32_t)((bytes + 15) >> 4); }
PLO_DEV unsigned long long seq_bytes_of(int fmt, uint32_t len) { return fmt == PLO_SEQ_BAM4 ? ((unsigned long long)len + 1) / 2 : len; }

// one lifted record (:245-284): flags, get_alignment_end, bam_reg2bin; supplementary until the primary is chosen (:282)
PLO_DEV void finish_item(const DevBatch &bt, const DevWork &wk, const DevFinish &f, uint32_t i) {
    uint32_t r = bt.seg_read[wk.item_seg[i]];
    f.item_read[i] = r;
    f.item_is_primary[i] = 0;
    uint32_t su = 0, qu = 0;
    uint16_t flag = 0, bin = 0;
    long long e = 0;
    if (wk.status[i] == PLO_ITEM_LIFTED) {
        flag = f.read_flags[r];
        bool flipped = wk.flip[i] != 0;
        if (flipped) flag ^= 0x10;  // reverse_alignment_seq_and_qual :126
        const uint32_t *cg = wk.out_cigar + wk.cig_off[i];
        uint32_t n = wk.cig_len[i];
        e = wk.pos[i];
        for (uint32_t k = 0; k < n; ++k) {
            uint32_t c = cg[k];
            if ((0x18D >> (c & 15u)) & 1) e += (long long)(c >> 4);
        }
        bin = bam_reg2bin((unsigned long long)wk.pos[i], (unsigned long long)e);
        flag |= 0x800;
        if (flipped) {
            uint32_t len = bt.read_seq_len[r];
            su = units16(seq_bytes_of(bt.seq_fmt, len));
            qu = units16(len);
        }
    }
    f.item_flag[i] = flag;
    f.item_bin[i] = bin;
    f.item_ref_end[i] = e;
    f.su[i] = su;
    f.qu[i] = qu;
    f.fflag[i] = su ? 1u : 0u;
}

PLO_DEV uint32_t u32_lower_bound(const uint32_t *a, uint32_t n, uint32_t x) {
    uint32_t lo = 0, hi = n;
    while (lo < hi) {
        uint32_t mid = lo + ((hi - lo) >> 1);
        if (a[mid] < x) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// ---- SA tag text (get_sa_tag_segment, src/read_alignment_scanner.rs:292-301) ------------------------------------------
// One segment "{chrom},{pos+1},{strand},{cigar},{mapq},0;" per lifted record of a read with at least two lifted records
// (:352-364 give every record the segments of the read's *other* records, in record order; a read with one record gets
// no SA tag).  The CIGAR is written as rust-htslib's Display does: "{len}{op}" per op, nothing for an empty CIGAR.
struct DevSa {
    const uint32_t *chrom_name_off;  // [n_chroms + 1]
    const uint8_t *chrom_names;      // concatenated chromosome labels
    const uint16_t *item_flag;
    const uint32_t *item_read;
    const uint32_t *read_n_lifted;
    uint32_t *len;        // [n_items] bytes of the item's segment (0 = none)
    const uint32_t *off;  // exclusive scan of len
    uint8_t *text;
};
PLO_DEV uint32_t dec_digits(unsigned long long v) {
    uint32_t n = 1;
    while (v >= 10) {
        v /= 10;
        ++n;
    }
    return n;
}
PLO_DEV uint8_t *put_dec(uint8_t *p, unsigned long long v) {
    uint32_t n = dec_digits(v);
    for (uint32_t k = n; k > 0; --k) {
        p[k - 1] = (uint8_t)('0' + (v % 10));
        v /= 10;
    }
    return p + n;
}
PLO_DEV bool sa_wanted(const DevWork &wk, const DevSa &sa, uint32_t i) {
    return wk.status[i] == PLO_ITEM_LIFTED && sa.read_n_lifted[sa.item_read[i]] >= 2;
}
PLO_DEV uint32_t sa_item_len(const DevWork &wk, const DevSa &sa, uint32_t i) {
    if (!sa_wanted(wk, sa, i)) return 0;
    uint32_t chrom = wk.chrom[i];
    uint32_t n = sa.chrom_name_off[chrom + 1] - sa.chrom_name_off[chrom];
    n += dec_digits((unsigned long long)(wk.pos[i] + 1)) + dec_digits(wk.mapq[i]) + 8;  // five commas, strand, "0;"
    const uint32_t *cg = wk.out_cigar + wk.cig_off[i];
    for (uint32_t k = 0; k < wk.cig_len[i]; ++k) n += dec_digits(cg[k] >> 4) + 1;
    return n;
}
PLO_DEV void sa_item_emit(const DevWork &wk, const DevSa &sa, uint32_t i) {
    if (!sa_wanted(wk, sa, i)) return;
    uint8_t *p = sa.text + sa.off[i];
    uint32_t chrom = wk.chrom[i];
    for (uint32_t k = sa.chrom_name_off[chrom]; k < sa.chrom_name_off[chrom + 1]; ++k) *p++ = sa.chrom_names[k];
    *p++ = ',';
    p = put_dec(p, (unsigned long long)(wk.pos[i] + 1));
    *p++ = ',';
    *p++ = (sa.item_flag[i] & 0x10) ? '-' : '+';
    *p++ = ',';
    const uint32_t *cg = wk.out_cigar + wk.cig_off[i];
    for (uint32_t k = 0; k < wk.cig_len[i]; ++k) {
        p = put_dec(p, cg[k] >> 4);
        *p++ = (uint8_t)"MIDNSHP=XB??????"[cg[k] & 15u];
    }
    *p++ = ',';
    p = put_dec(p, wk.mapq[i]);
    *p++ = ',';
    *p++ = '0';
    *p++ = ';';
}

// finish_remapped_alignment_set (:310-366): primary = highest MAPQ, first wins (:338-346); no record -> unmapped copy (:317-335)
PLO_DEV void finish_read(const DevBatch &bt, const DevWork &wk, const DevFinish &f, uint32_t r) {
    uint32_t n = wk.n_items;
    uint32_t i0 = u32_lower_bound(f.item_read, n, r), i1 = u32_lower_bound(f.item_read, n, r + 1);
    uint32_t nl = 0, best = 0xffffffffu;
    for (uint32_t i = i0; i < i1; ++i) {
        if (wk.status[i] == PLO_ITEM_LEN_MISMATCH || wk.status[i] == PLO_ITEM_PANIC || wk.status[i] == PLO_ITEM_NEED_BASES) wv::atomic_add_global(f.n_fault, 1u);
        if (wk.status[i] != PLO_ITEM_LIFTED) continue;
        ++nl;
        if (best == 0xffffffffu || wk.mapq[best] < wk.mapq[i]) best = i;
    }
    uint32_t su = 0, qu = 0;
    uint16_t uf = 0;
    if (nl > 0) {
        f.item_is_primary[best] = 1;
        f.item_flag[best] = (uint16_t)(f.item_flag[best] & ~0x800);
    } else {
        uf = f.read_flags[r];
        uf |= 0x4;
        uf = (uint16_t)(uf & ~0x800);
        if (uf & 0x10) {
            uf ^= 0x10;
            uint32_t len = bt.read_seq_len[r];
            su = units16(seq_bytes_of(bt.seq_fmt, len));
            qu = units16(len);
        }
    }
    f.read_n_lifted[r] = nl;
    f.read_primary_item[r] = best;
    f.read_unmapped_flag[r] = uf;
    f.su[n + r] = su;
    f.qu[n + r] = qu;
    f.fflag[n + r] = su ? 1u : 0u;
}

// byte `idx` of a buffer, 0 outside it
PLO_DEV unsigned ld_byte(const uint8_t *buf, unsigned long long total, long long idx) {
    return (idx >= 0 && (unsigned long long)idx < total) ? buf[idx] : 0u;
}
// 32 bits starting at byte `idx` (little endian): two aligned dword loads + a funnel shift on the fast path
PLO_DEV unsigned ld_window32(const uint8_t *buf, unsigned long long total, long long idx) {
    unsigned long long addr = (unsigned long long)(uintptr_t)buf + (unsigned long long)idx;
    unsigned sh = (unsigned)(addr & 3ull);
    long long a = idx - (long long)sh;  // index of the aligned dword containing byte idx (buffer base 4-byte aligned or not)
    if (idx >= 4 && (unsigned long long)idx + 8 <= total) {
        const uint32_t *p = (const uint32_t *)(buf + a);
        unsigned lo = p[0], hi = p[1];
        unsigned long long w = ((unsigned long long)hi << 32) | lo;
        return (unsigned)(w >> (8 * sh));
    }
    return ld_byte(buf, total, idx) | (ld_byte(buf, total, idx + 1) << 8) | (ld_byte(buf, total, idx + 2) << 16) |
           (ld_byte(buf, total, idx + 3) << 24);
}
// 64 bits starting at byte `idx`
PLO_DEV unsigned long long ld_window64(const uint8_t *buf, unsigned long long total, long long idx) {
    unsigned long long addr = (unsigned long long)(uintptr_t)buf + (unsigned long long)idx;
    unsigned sh = (unsigned)(addr & 3ull);
    long long a = idx - (long long)sh;
    if (idx >= 4 && (unsigned long long)idx + 12 <= total) {
        const uint32_t *p = (const uint32_t *)(buf + a);
        unsigned w0 = p[0], w1 = p[1], w2 = p[2];
        unsigned long long lo = ((unsigned long long)w1 << 32) | w0;
        if (sh == 0) return lo;
        return (lo >> (8 * sh)) | ((unsigned long long)w2 << (64 - 8 * sh));
    }
    unsigned long long r = 0;
    for (int k = 0; k < 8; ++k) r |= (unsigned long long)ld_byte(buf, total, idx + k) << (8 * k);
    return r;
}

// Complement in the BAM 4-bit alphabet as the reference computes it: decode ("=ACMGRSVTWYHKDBN"), comp_base
// (lib/rust-vc-utils/src/seq_util.rs:1-15: everything but A C G T N becomes N), re-encode (record.set()):
// A(1)<->T(8), C(2)<->G(4), everything else -> N(15).  16 x 4-bit table in one 64-bit constant.
PLO_DEV unsigned comp_nibble(unsigned n) { return (unsigned)((0xFFFFFFF1FFF2F48Full >> (4 * n)) & 15ull); }

// comp_nibble on the eight nibbles of a word at once: a one-hot nibble (A C G T) is bit-reversed, anything else -> 15
PLO_DEV unsigned comp8(unsigned x) {
    unsigned r = ((x & 0x55555555u) << 1) | ((x >> 1) & 0x55555555u);
    r = ((r & 0x33333333u) << 2) | ((r >> 2) & 0x33333333u);                                       // per-nibble bit reversal
    unsigned p = x - ((x >> 1) & 0x77777777u) - ((x >> 2) & 0x33333333u) - ((x >> 3) & 0x11111111u);  // per-nibble popcount
    unsigned z = p ^ 0x11111111u;                                                                   // 0 where popcount == 1
    unsigned nz = (z | (z >> 1) | (z >> 2) | (z >> 3)) & 0x11111111u;
    unsigned m = nz * 15u;  // 0xF in every nibble that is not one-hot
    return (r & ~m) | m;
}
PLO_DEV unsigned nibswap32(unsigned x) { return ((x & 0x0f0f0f0fu) << 4) | ((x >> 4) & 0x0f0f0f0fu); }

struct U4 {
    unsigned x, y, z, w;
};

// reverse_alignment_seq_and_qual (:125-133) of one record by `nthreads` cooperating threads (thread `tid`).
// dst_seq / dst_qual are 16-byte aligned; every thread produces aligned 16-byte chunks from (unaligned) source windows
// read as aligned dwords + funnel shifts.
PLO_DEV void revcomp_record(const DevBatch &bt, const DevFinish &f, uint32_t read, uint8_t *dst_seq, uint8_t *dst_qual, int tid,
                            int nthreads) {
    const long long L = (long long)bt.read_seq_len[read];
    // ---- qualities: dst[i] = src[L-1-i] ----
    {
        const long long q0 = (long long)f.read_qual_off[read];
        const long long nc = (L + 15) >> 4;
        // chunks [c_lo, c_hi) take the branch-free fast path (all three conditions of the `if` below hold)
        long long c_hi = L >= 16 ? ((L - 16) >> 4) + 1 : 0;
        if (q0 + L - 20 < 0) c_hi = 0; else c_hi = wv::imin((int)c_hi, (int)(((q0 + L - 20) >> 4) + 1));
        long long over = q0 + L + 8 - (long long)f.qual_bytes;
        long long c_lo = over > 0 ? ((over + 15) >> 4) : 0;
        if (c_lo > c_hi) c_lo = c_hi;
        // four chunks per thread per trip: all loads are issued before the first store (the compiler cannot hoist
        // them itself, source and destination may alias as far as it knows)
        for (long long cb = c_lo + tid; cb < c_hi; cb += 4ll * nthreads) {
            unsigned w[4][5], shv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                long long c = cb + (long long)u * nthreads;
                long long g0 = q0 + (L - 16 - 16 * (c < c_hi ? c : cb));
                shv[u] = (unsigned)(((unsigned long long)(uintptr_t)f.qual + (unsigned long long)g0) & 3ull);
                const uint32_t *p = (const uint32_t *)(f.qual + (g0 - (long long)shv[u]));
#pragma unroll
                for (int k = 0; k < 5; ++k) w[u][k] = p[k];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                long long c = cb + (long long)u * nthreads;
                if (c >= c_hi) break;
                unsigned sh = shv[u];
                U4 o;
                o.x = __builtin_bswap32((unsigned)((((unsigned long long)w[u][4] << 32) | w[u][3]) >> (8 * sh)));
                o.y = __builtin_bswap32((unsigned)((((unsigned long long)w[u][3] << 32) | w[u][2]) >> (8 * sh)));
                o.z = __builtin_bswap32((unsigned)((((unsigned long long)w[u][2] << 32) | w[u][1]) >> (8 * sh)));
                o.w = __builtin_bswap32((unsigned)((((unsigned long long)w[u][1] << 32) | w[u][0]) >> (8 * sh)));
                *(U4 *)(dst_qual + 16 * c) = o;
            }
        }
        for (long long c = tid; c < nc; c += nthreads) {
            if (c >= c_lo && c < c_hi) continue;  // edge chunks only
            long long s0 = L - 16 - 16 * c;  // source bytes s0 .. s0+15 (s0 < 0 only for the last chunk)
            long long g0 = q0 + s0;
            U4 o;
            if (s0 >= 0 && g0 >= 4 && (unsigned long long)g0 + 24 <= f.qual_bytes) {
                unsigned sh = (unsigned)(((unsigned long long)(uintptr_t)f.qual + (unsigned long long)g0) & 3ull);
                const uint32_t *p = (const uint32_t *)(f.qual + (g0 - (long long)sh));
                unsigned w0 = p[0], w1 = p[1], w2 = p[2], w3 = p[3], w4 = p[4];
                unsigned s_0 = (unsigned)((((unsigned long long)w1 << 32) | w0) >> (8 * sh));
                unsigned s_1 = (unsigned)((((unsigned long long)w2 << 32) | w1) >> (8 * sh));
                unsigned s_2 = (unsigned)((((unsigned long long)w3 << 32) | w2) >> (8 * sh));
                unsigned s_3 = (unsigned)((((unsigned long long)w4 << 32) | w3) >> (8 * sh));
                o.x = __builtin_bswap32(s_3);
                o.y = __builtin_bswap32(s_2);
                o.z = __builtin_bswap32(s_1);
                o.w = __builtin_bswap32(s_0);
            } else {
                unsigned v[4] = {0, 0, 0, 0};
                for (int k = 0; k < 16; ++k) {
                    long long i = 16 * c + k;  // destination byte
                    if (i < L) v[k >> 2] |= ld_byte(f.qual, f.qual_bytes, q0 + (L - 1 - i)) << (8 * (k & 3));
                }
                o.x = v[0];
                o.y = v[1];
                o.z = v[2];
                o.w = v[3];
            }
            *(U4 *)(dst_qual + 16 * c) = o;
        }
    }
    // ---- bases ----
    const long long b0 = (long long)bt.read_seq_off[read];
    if (bt.seq_fmt != PLO_SEQ_BAM4) {  // ASCII: rev_comp_in_place byte-wise
        const long long nd = (L + 3) >> 2;
        uint32_t *d = (uint32_t *)dst_seq;
        for (long long w = tid; w < nd; w += nthreads) {
            unsigned x = 0;
            for (int k = 0; k < 4; ++k) {
                long long i = 4 * w + k;
                if (i < L) x |= (unsigned)comp_base((int)ld_byte(bt.seq, f.seq_bytes, b0 + (L - 1 - i))) << (8 * k);
            }
            d[w] = x;
        }
        return;
    }
    // 4-bit: destination base i (even i = high nibble of byte i/2) = comp(source base L-1-i).
    // With the nibbles of a source window swapped inside every byte the bases become a linear little-endian nibble
    // stream; an 8-base destination word is then bswap32 of the (nibble-shifted) source word, complemented.
    const long long nb = (L + 1) >> 1;    // destination bytes
    const long long nc = (nb + 15) >> 4;  // destination 16-byte chunks (32 bases each)
    long long c_hi = L >= 32 ? ((L - 32) >> 5) + 1 : 0;
    {
        // g0 = b0 + ((L - 32 - 32c) >> 1) >= 4  <=>  c <= (2*b0 + L - 32 - 8) / 32 (floor)
        long long t = 2 * b0 + L - 40;
        if (t < 0) c_hi = 0; else c_hi = wv::imin((int)c_hi, (int)((t >> 5) + 1));
    }
    long long c_lo = 0;
    {
        long long over = b0 + ((L - 32) >> 1) + 28 - (long long)f.seq_bytes;  // chunk 0
        if (over > 0) c_lo = (over + 15) >> 4;
        if (c_lo > c_hi) c_lo = c_hi;
    }
    for (long long cb = c_lo + tid; cb < c_hi; cb += 2ll * nthreads) {
      unsigned wv2[2][6], shv[2], rv[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
          long long c = cb + (long long)u * nthreads;
          long long jlo = L - 32 - 32 * (c < c_hi ? c : cb);
          long long g0 = b0 + (jlo >> 1);
          rv[u] = (unsigned)(jlo & 1);
          shv[u] = (unsigned)(((unsigned long long)(uintptr_t)bt.seq + (unsigned long long)g0) & 3ull);
          const uint32_t *p = (const uint32_t *)(bt.seq + (g0 - (long long)shv[u]));
#pragma unroll
          for (int k = 0; k < 6; ++k) wv2[u][k] = p[k];
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        long long c = cb + (long long)u * nthreads;
        if (c >= c_hi) break;
        unsigned r = rv[u], sh = shv[u];
        unsigned w0 = wv2[u][0], w1 = wv2[u][1], w2 = wv2[u][2], w3 = wv2[u][3], w4 = wv2[u][4], w5 = wv2[u][5];
        unsigned s_0 = nibswap32((unsigned)((((unsigned long long)w1 << 32) | w0) >> (8 * sh)));
        unsigned s_1 = nibswap32((unsigned)((((unsigned long long)w2 << 32) | w1) >> (8 * sh)));
        unsigned s_2 = nibswap32((unsigned)((((unsigned long long)w3 << 32) | w2) >> (8 * sh)));
        unsigned s_3 = nibswap32((unsigned)((((unsigned long long)w4 << 32) | w3) >> (8 * sh)));
        unsigned s_4 = nibswap32((unsigned)((((unsigned long long)w5 << 32) | w4) >> (8 * sh)));
        U4 o;
        o.x = comp8(__builtin_bswap32((unsigned)((((unsigned long long)s_4 << 32) | s_3) >> (4 * r))));
        o.y = comp8(__builtin_bswap32((unsigned)((((unsigned long long)s_3 << 32) | s_2) >> (4 * r))));
        o.z = comp8(__builtin_bswap32((unsigned)((((unsigned long long)s_2 << 32) | s_1) >> (4 * r))));
        o.w = comp8(__builtin_bswap32((unsigned)((((unsigned long long)s_1 << 32) | s_0) >> (4 * r))));
        *(U4 *)(dst_seq + 16 * c) = o;
      }
    }
    for (long long c = tid; c < nc; c += nthreads) {
        if (c >= c_lo && c < c_hi) continue;  // edge chunks only
        long long jlo = L - 32 - 32 * c;  // lowest source base of the chunk (negative only for the last chunk)
        U4 o;
        long long q = jlo >> 1;  // for jlo >= 0
        long long g0 = b0 + q;
        if (jlo >= 0 && g0 >= 4 && (unsigned long long)g0 + 28 <= f.seq_bytes) {
            unsigned r = (unsigned)(jlo & 1);
            unsigned sh = (unsigned)(((unsigned long long)(uintptr_t)bt.seq + (unsigned long long)g0) & 3ull);
            const uint32_t *p = (const uint32_t *)(bt.seq + (g0 - (long long)sh));
            unsigned w0 = p[0], w1 = p[1], w2 = p[2], w3 = p[3], w4 = p[4], w5 = p[5];
            // source bytes q .. q+19 as five dwords, nibbles swapped inside the bytes
            unsigned s_0 = nibswap32((unsigned)((((unsigned long long)w1 << 32) | w0) >> (8 * sh)));
            unsigned s_1 = nibswap32((unsigned)((((unsigned long long)w2 << 32) | w1) >> (8 * sh)));
            unsigned s_2 = nibswap32((unsigned)((((unsigned long long)w3 << 32) | w2) >> (8 * sh)));
            unsigned s_3 = nibswap32((unsigned)((((unsigned long long)w4 << 32) | w3) >> (8 * sh)));
            unsigned s_4 = nibswap32((unsigned)((((unsigned long long)w5 << 32) | w4) >> (8 * sh)));
            // 8 bases starting at base r + 8k of the window
            unsigned x0 = (unsigned)((((unsigned long long)s_1 << 32) | s_0) >> (4 * r));
            unsigned x1 = (unsigned)((((unsigned long long)s_2 << 32) | s_1) >> (4 * r));
            unsigned x2 = (unsigned)((((unsigned long long)s_3 << 32) | s_2) >> (4 * r));
            unsigned x3 = (unsigned)((((unsigned long long)s_4 << 32) | s_3) >> (4 * r));
            o.x = comp8(__builtin_bswap32(x3));
            o.y = comp8(__builtin_bswap32(x2));
            o.z = comp8(__builtin_bswap32(x1));
            o.w = comp8(__builtin_bswap32(x0));
        } else {
            unsigned v[4] = {0, 0, 0, 0};
            for (int k = 0; k < 32; ++k) {
                long long i = 32 * c + k;  // destination base
                if (i < L) {
                    long long j = L - 1 - i;
                    unsigned byte = ld_byte(bt.seq, f.seq_bytes, b0 + (j >> 1));
                    unsigned nib = (j & 1) ? (byte & 15u) : (byte >> 4);
                    v[k >> 3] |= comp_nibble(nib) << (8 * ((k & 7) >> 1) + ((k & 1) ? 0 : 4));
                }
            }
            o.x = v[0];
            o.y = v[1];
            o.z = v[2];
            o.w = v[3];
        }
        *(U4 *)(dst_seq + 16 * c) = o;
    }
}

}  // namespace plo
