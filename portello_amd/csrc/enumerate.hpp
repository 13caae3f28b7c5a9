// enumerate.hpp -- item enumeration: which contig split segments does a read split segment touch?
// Restates get_contig_split_segments_from_read_mapping (/root/reference/src/read_alignment_scanner.rs:80-103) with
// IntRange::intersect_range (lib/rust-vc-utils/src/int_range.rs:56-58; note the `>=`: left adjacency counts).
#pragma once
#include <plo_wave.hpp>

#include "lift_types.hpp"

namespace plo {

// 32 readable bytes in global memory: where the lanes without a valid address point their (unconditional) loads.  A load inside a
// lane-divergent branch is waited for at the end of that branch (the merge needs the value); an unconditional one only where its value
// is used, so that independent loads are in flight together.
#ifdef PLO_EMULATOR
alignas(16) static const uint32_t plo_safe_words[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#else
alignas(16) static __device__ const uint32_t plo_safe_words[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif

// End of a level of loads: an empty volatile statement that takes the loaded values as operands.  Every load of the level is issued in front
// of it and waited for there -- once -- instead of being sunk by the compiler to the branch that uses it and waited for one by one.
#ifdef PLO_EMULATOR
#define PLO_LEVEL_END(...)
#else
#define PLO_LV_(x) "v"(x)
#define PLO_LV1(a) PLO_LV_(a)
#define PLO_LV2(a, ...) PLO_LV_(a), PLO_LV1(__VA_ARGS__)
#define PLO_LV3(a, ...) PLO_LV_(a), PLO_LV2(__VA_ARGS__)
#define PLO_LV4(a, ...) PLO_LV_(a), PLO_LV3(__VA_ARGS__)
#define PLO_LV5(a, ...) PLO_LV_(a), PLO_LV4(__VA_ARGS__)
#define PLO_LV6(a, ...) PLO_LV_(a), PLO_LV5(__VA_ARGS__)
#define PLO_LV7(a, ...) PLO_LV_(a), PLO_LV6(__VA_ARGS__)
#define PLO_LV8(a, ...) PLO_LV_(a), PLO_LV7(__VA_ARGS__)
#define PLO_LV9(a, ...) PLO_LV_(a), PLO_LV8(__VA_ARGS__)
#define PLO_LV_N(_1, _2, _3, _4, _5, _6, _7, _8, _9, n, ...) PLO_LV##n
#define PLO_LEVEL_END(...) asm volatile("" ::PLO_LV_N(__VA_ARGS__, 9, 8, 7, 6, 5, 4, 3, 2, 1)(__VA_ARGS__))
#endif

// get_cigar_ref_offset (lib/rust-vc-utils/src/bam_utils/cigar/mod.rs:174-180)
PLO_DEV long long segment_ref_len(const DevBatch &bt, uint32_t seg) {
    uint32_t c0 = bt.seg_cigar_off[seg], c1 = bt.seg_cigar_off[seg + 1];
    long long r = 0;
    for (uint32_t i = c0; i < c1; ++i) {
        uint32_t c = bt.cigar[i];
        if ((0x18D >> (c & 15u)) & 1) r += (long long)(c >> 4);
    }
    return r;
}

// get_cigar_read_offset(cigar, false) (lib/rust-vc-utils/src/bam_utils/cigar/mod.rs:164-170), saturated at 2^32 - 1
PLO_DEV uint32_t segment_read_len_sat(const DevBatch &bt, uint32_t seg) {
    uint32_t c0 = bt.seg_cigar_off[seg], c1 = bt.seg_cigar_off[seg + 1];
    unsigned long long r = 0;
    for (uint32_t i = c0; i < c1; ++i) {
        uint32_t c = bt.cigar[i];
        if ((0x1B3 >> (c & 15u)) & 1) r += (unsigned long long)(c >> 4);  // M I S H = X
    }
    return r > 0xfffffffeull ? 0xffffffffu : (uint32_t)r;
}

// ops of the segment's CIGAR once neighbouring alignment-match ops (M = X) are merged into one
PLO_DEV uint32_t segment_n_merged(const DevBatch &bt, uint32_t seg) {
    uint32_t c0 = bt.seg_cigar_off[seg], c1 = bt.seg_cigar_off[seg + 1];
    uint32_t n = 0;
    bool prev = false;
    for (uint32_t i = c0; i < c1; ++i) {
        const bool m = ((0x181u >> (bt.cigar[i] & 15u)) & 1u) != 0u;
        n += (m && prev) ? 0u : 1u;
        prev = m;
    }
    return n;
}

// The LDS region of an item in the lane-per-item kernel (lane_core.hpp): its ops (`n`: merged count when a stage that merges
// follows, the raw count otherwise), room for what the liftover may add -- one more piece per key of the block map inside the item's
// span (entries w0+1 .. w1-1, and w0 itself when no block holds the item's start) and one jump deletion per block entered that
// way -- and a little slack for the shift stage.
constexpr int LANE_SLACK = 2;
#ifndef PLO_LANE_GAP_HALVES
#define PLO_LANE_GAP_HALVES 4  // the allowance per block-map key of the item's span, in halves of an op (4: the bound, two ops per key)
#endif
#ifndef PLO_LANE_GAP_CONST
#define PLO_LANE_GAP_CONST 0
#endif
PLO_DEV int lane_region_gap(int w0, int w1) { return (PLO_LANE_GAP_HALVES * (w1 > w0 ? w1 - w0 : 0) + 1) / 2 + PLO_LANE_GAP_CONST; }
PLO_DEV int lane_region_dwords(int n, int w0, int w1) { return n + lane_region_gap(w0, w1) + LANE_SLACK; }

// block-map searches (ReadToRefTreeMap::get_ref_range, read_to_ref_map.rs:74-85)
// Eight-way: seven independent probes per level, so that a map of a few thousand blocks costs four or five memory round trips
// instead of a dozen dependent ones (the descriptor kernel is a chain of dependent loads per item, nothing else).
PLO_DEV void kv_upper_bound_narrow(const KV *kv, int &lo, int &hi, int x) {  // narrows [lo,hi) to at most eight entries around the answer
    // invariant: keys below lo are <= x, keys from hi on are > x
    while (hi - lo > 8) {
        const int step = (hi - lo) >> 3;
        int k[7];
#pragma unroll
        for (int j = 0; j < 7; ++j) k[j] = kv[lo + (j + 1) * step].key;
        int c = 0;
#pragma unroll
        for (int j = 0; j < 7; ++j) c += k[j] <= x ? 1 : 0;  // sorted keys: the probes with key <= x are the first c
        const int nlo = c > 0 ? lo + c * step + 1 : lo;
        hi = c < 7 ? lo + (c + 1) * step : hi;
        lo = nlo;
    }
}
PLO_DEV int kv_upper_bound_tail(const KV *kv, int lo, int hi, int x) {  // hi - lo <= 8
    int k[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) k[j] = (lo + j < hi ? kv + (lo + j) : (const KV *)plo_safe_words)->key;  // (eight loads, one round trip)
    int c = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) c += ((lo + j < hi) & (k[j] <= x)) ? 1 : 0;
    return lo + c;
}
PLO_DEV int kv_upper_bound(const KV *kv, int lo, int hi, int x) {  // first index in [lo,hi) with key > x, else hi
    kv_upper_bound_narrow(kv, lo, hi, x);
    return kv_upper_bound_tail(kv, lo, hi, x);
}
PLO_DEV int kv_lower_bound(const KV *kv, int lo, int hi, int x) {  // first index in [lo,hi) with key >= x, else hi
    while (lo < hi) {
        int mid = (int)(((unsigned)lo + (unsigned)hi) >> 1);
        if (kv[mid].key < x)
            lo = mid + 1;
        else
            hi = mid;
    }
    return lo;
}

// same result as kv_lower_bound when the answer is expected a few entries after `lo` (the end of an item's window):
// probes lo, lo+1, lo+3, lo+7, ... before bisecting the bracket
PLO_DEV int kv_lower_bound_near(const KV *kv, int lo, int hi, int x) {
    {   // the next eight entries at once: nearly always enough (a read crosses a few blocks)
        int k[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) k[j] = (lo + j < hi ? kv + (lo + j) : (const KV *)plo_safe_words)->key;
        int c = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) c += ((lo + j < hi) & (k[j] < x)) ? 1 : 0;
        if (c < 8 || lo + 8 >= hi) return lo + c;
        lo += 8;
    }
    int step = 1;
    while (lo < hi) {
        int probe = lo + step - 1;
        if (probe >= hi) break;
        if (kv[probe].key < x) {
            lo = probe + 1;
            step <<= 1;
        } else {
            hi = probe;
            break;
        }
    }
    return kv_lower_bound(kv, lo, hi, x);
}

// Block map of one contig split segment: get_read_segment_to_ref_pos_tree_map
// (lib/rust-vc-utils/src/bam_utils/read_to_ref_map.rs:101-137, ignore_hard_clip = false as at
// src/contig_alignment_scanner/mod.rs:98-102).  Every flush of a match run inserts {start -> Some(ref), end -> None};
// keys arrive in non-decreasing order, so BTreeMap::insert's overwrite can only hit the entry appended last (a
// deletion in the contig->ref alignment: the previous block's None is replaced by the next block's Some).
// out == nullptr: count only.  Returns the number of entries, or -1 if a coordinate leaves the 31-bit range / an op
// code is invalid.
PLO_HD int build_segment_map(const uint32_t *cigar, uint32_t n, long long ref_pos, KV *out) {
    long long read_pos = 0, match_len = 0;
    int cnt = 0;
    int last_key = -1;
    bool bad = false;
    for (uint32_t i = 0; i <= n; ++i) {
        uint32_t c = i < n ? cigar[i] : 4u;  // a trailing pseudo soft clip of length 0 performs the final flush (:134)
        int t = (int)(c & 15u);
        long long len = (long long)(c >> 4);
        if (t > 8) bad = true;
        bool is_m = (t == OP_M || t == OP_EQ || t == OP_X);
        if (is_m) {
            match_len += len;
        } else if (match_len > 0) {  // update_map (:105-113)
            if (read_pos > 0x7ffffff0LL || ref_pos > 0x7ffffff0LL || ref_pos - match_len < 0) {
                bad = true;
            } else {
                int k0 = (int)(read_pos - match_len), v0 = (int)(ref_pos - match_len), k1 = (int)read_pos;
                if (cnt > 0 && last_key == k0) {
                    if (out) out[cnt - 1].val = v0;
                } else {
                    if (out) {
                        out[cnt].key = k0;
                        out[cnt].val = v0;
                    }
                    ++cnt;
                }
                if (out) {
                    out[cnt].key = k1;
                    out[cnt].val = NONE32;
                }
                ++cnt;
                last_key = k1;
            }
            match_len = 0;
        }
        if ((0x1B3 >> t) & 1) read_pos += len;  // M I S H = X
        if ((0x18D >> t) & 1) ref_pos += len;   // M D N = X
    }
    return bad ? -1 : cnt;
}

// The same block map built by ONE WAVE (index creation, k_map_build): 64 ops per step, lane i takes op base + i.  The sequential state of
// build_segment_map becomes scans: the read / reference positions in front of every op are prefix sums (64-bit, as two 16-bit halves per
// op); the match run that a non-match op flushes starts behind the nearest non-match op before it (a max-scan over lane indices; carried
// over the step's border as the positions at which the open run starts); a flush appends {k0 -> Some(v0)}, {k1 -> None} -- or, when the
// previous flush's k1 equals k0 (nothing of the read was consumed between the two: deletions), overwrites that None and appends one
// entry -- so the entries' places are a prefix sum of 1 or 2 per flush.  Wave-uniform call; returns what build_segment_map returns.
PLO_DEV long long wave_scan_add64(long long x) {  // inclusive; 0 <= x < 2^28 per lane (CIGAR op lengths)
    const long long lo = (long long)wv::scan_add((int)(x & 0xffff)), hi = (long long)wv::scan_add((int)(x >> 16));
    return (hi << 16) + lo;
}
PLO_DEV long long wave_bcast64(long long v, int src) { return wv::shfl(v, src); }
PLO_DEV int build_segment_map_wave(const uint32_t *cigar, uint32_t n, long long ref_pos0, KV *out) {
    const int lane = wv::lane();
    long long read_base = 0, ref_base = ref_pos0;          // positions in front of the step's first op (wave-uniform)
    long long run_read = 0, run_ref = ref_pos0;            // ... at which the open match run starts (behind the last non-match op so far)
    long long last_key = -1;                               // k1 of the last flush (-1: none yet)
    int cnt = 0;
    bool bad = false;
    for (uint32_t base = 0; base <= n; base += 64u) {
        const uint32_t i = base + (uint32_t)lane;
        const bool have = i <= n;                          // (op n: the trailing pseudo soft clip of length 0 that performs the final flush, :134)
        const uint32_t c = i < n ? cigar[i] : 4u;
        const int t = (int)(c & 15u);
        const long long len = have ? (long long)(c >> 4) : 0;
        bad = bad | (have & (t > 8));
        const bool is_m = (t == OP_M) | (t == OP_EQ) | (t == OP_X);
        const long long rl = ((0x1B3 >> t) & 1) ? len : 0, fl = ((0x18D >> t) & 1) ? len : 0;  // M I S H = X | M D N = X
        const long long r_inc = wave_scan_add64(rl), f_inc = wave_scan_add64(fl);
        const long long r_before = read_base + r_inc - rl, f_before = ref_base + f_inc - fl;
        // the nearest non-match op in front of this one, inside the step
        const int nm = (have & !is_m) ? lane : -1;
        const int nm_inc = wv::scan_max(nm);
        const int prev_nm = wv::shfl_up1(nm_inc, -1);      // (exclusive)
        // positions behind that op = in front of op prev_nm + 1: the run's start
        const long long r_after = r_before + rl, f_after = f_before + fl;
        const long long rs_l = wave_bcast64(r_after, prev_nm < 0 ? 0 : prev_nm), fs_l = wave_bcast64(f_after, prev_nm < 0 ? 0 : prev_nm);
        const long long rs = prev_nm < 0 ? run_read : rs_l, fs = prev_nm < 0 ? run_ref : fs_l;
        const long long match_len = r_before - rs;         // (match ops consume read and reference alike)
        const bool flush = have & !is_m & (match_len > 0);  // update_map (:105-113)
        const bool range_bad = flush & ((r_before > 0x7ffffff0LL) | (f_before > 0x7ffffff0LL) | (fs < 0));
        bad = bad | range_bad;
        const bool ev = flush & !range_bad;
        const long long k0 = rs, v0 = fs, k1 = r_before;
        // the previous flush: inside the step, or the carried one
        const int fl_lane = ev ? lane : -1;
        const int fl_inc = wv::scan_max(fl_lane);
        const int prev_fl = wv::shfl_up1(fl_inc, -1);
        const long long pk_l = wave_bcast64(k1, prev_fl < 0 ? 0 : prev_fl);
        const long long prev_key = prev_fl < 0 ? last_key : pk_l;
        const bool merge = ev & (prev_key >= 0) & (prev_key == k0);
        const int add = ev ? (merge ? 1 : 2) : 0;
        const int inc = wv::scan_add(add);
        const int at = cnt + inc - add;                    // first entry of this flush
        if (out && ev) {
            if (!merge) {
                out[at].key = (int)k0;
                out[at].val = (int)v0;
            }
            out[at + add - 1].key = (int)k1;
            out[at + add - 1].val = NONE32;
        }
        wv::sync();  // (the None of a flush is written before the next flush of the step replaces it)
        if (out && merge) out[at - 1].val = (int)v0;
        // carry
        const int last_nm = wv::bcast_last(nm_inc), last_fl = wv::bcast_last(fl_inc);
        const long long tr = wave_bcast64(r_after, 63), tf = wave_bcast64(f_after, 63);
        const long long nr = wave_bcast64(r_after, last_nm < 0 ? 0 : last_nm), nf = wave_bcast64(f_after, last_nm < 0 ? 0 : last_nm);
        run_read = last_nm < 0 ? run_read : nr;
        run_ref = last_nm < 0 ? run_ref : nf;
        const long long lk = wave_bcast64(k1, last_fl < 0 ? 0 : last_fl);
        last_key = last_fl < 0 ? last_key : lk;
        read_base = tr;
        ref_base = tf;
        cnt += wv::bcast_last(inc);
    }
    return wv::ballot(bad) != 0ull ? -1 : cnt;
}

// Upper bound of the elements an item needs in any stage: pieces <= ops + blocks, raw lifted ops <= pieces + one jump
// deletion per block.  Tiles are cut and large items are routed by this weight, so that a tile of `window` weight fits the
// LDS slice whatever the density of the contig's block map.
PLO_DEV int item_weight(int n_in, int w0, int w1, int kv1) {
    int nblk = (w1 + 1 < kv1 ? w1 + 1 : kv1) - w0;
    return n_in + 2 * (nblk > 0 ? nblk : 0);
}
enum { LEVEL_TILE = 0, LEVEL_RETRY = 1, LEVEL_LAST = 2, LEVEL_MID = 3 };

// The descriptor code is a chain of dependent loads per item and nothing else, so it is written level by level: every load of a level
// is issued before the first use of any of them, and every store comes after the last load (stores and loads may alias for all the
// compiler knows, and the memory counter is in order: a load behind a store waits for the store).

// What an item needs through its read segment: the segment's own fields (level 0), then its read's and its contig's (level 1)
struct SegInfo {
    uint32_t contig, read, in_off, n_in, n_m, read_len, g0, g1, seq_len;
    int contig_len;
    long long pos;
    uint64_t seq_off, revseq;
    bool seg_fwd, read_rev;
};
PLO_DEV void seg_info_level0(SegInfo &s, const DevBatch &bt, const DevWork &wk, uint32_t stages, uint32_t seg) {
    s.contig = bt.seg_contig[seg];
    s.read = bt.seg_read[seg];
    s.in_off = bt.seg_cigar_off[seg];
    s.n_in = bt.seg_cigar_off[seg + 1];
    s.pos = (long long)bt.seg_pos[seg];
    s.n_m = wk.seg_nm ? wk.seg_nm[seg] : 0u;
    s.read_len = wk.seg_readlen ? wk.seg_readlen[seg] : 0u;
    const uint32_t sf = (stages & PLO_STAGE_STRAND) ? (uint32_t)bt.seg_is_fwd[seg] : 0u;
    PLO_LEVEL_END(s.contig, s.read, s.in_off, s.n_in, s.pos, s.n_m, s.read_len, sf);
    s.seg_fwd = sf != 0u;
    s.n_in -= s.in_off;
}
PLO_DEV void seg_info_level1(SegInfo &s, const DevIndex &ix, const DevBatch &bt, const DevWork &wk, uint32_t stages, uint32_t seg) {
    s.g0 = ix.contig_seg_off[s.contig];
    s.g1 = ix.contig_seg_off[s.contig + 1];
    s.contig_len = ix.contig_len[s.contig];
    s.revseq = ix.contig_revseq ? (uint64_t)(uintptr_t)ix.contig_revseq[s.contig] : 0ull;
    s.seq_len = bt.read_seq_len[s.read];
    s.seq_off = bt.read_seq_off[s.read];
    const uint32_t rr = (stages & PLO_STAGE_STRAND) ? (uint32_t)bt.read_is_reverse[s.read] : 0u;
    PLO_LEVEL_END(s.g0, s.g1, s.contig_len, s.revseq, s.seq_len, s.seq_off, rr);
    s.read_rev = rr != 0u;
    if (!wk.seg_nm) s.n_m = segment_n_merged(bt, seg);
    if (!wk.seg_readlen) s.read_len = segment_read_len_sat(bt, seg);
}
// ... and through its contig split segment (level 2: one index, seven arrays)
struct CsegInfo {
    int start, end;
    uint32_t kv0, kv1, chrom;
    uint8_t is_fwd, mapq;
};
PLO_DEV CsegInfo cseg_info(const DevIndex &ix, uint32_t gseg) {
    CsegInfo c;
    c.start = ix.cs_start[gseg];
    c.end = ix.cs_end[gseg];
    c.kv0 = ix.cs_kv_off[gseg];
    c.kv1 = ix.cs_kv_off[gseg + 1];
    c.chrom = ix.cs_chrom[gseg];
    const uint32_t f = ix.cs_is_fwd[gseg], q = ix.cs_mapq[gseg];
    PLO_LEVEL_END(c.start, c.end, c.kv0, c.kv1, c.chrom, f, q);
    c.is_fwd = (uint8_t)f;
    c.mapq = (uint8_t)q;
    return c;
}

// Resolves everything the tile kernel needs to know about item i = (read segment seg, contig segment cseg):
// the caller glue of get_liftover_alignment_for_read_and_contig_segment (src/read_alignment_scanner.rs:146-176) --
// need_flipped (:153-157), rev_pos (:164-166) -- plus the window of the block map the item can touch.
PLO_DEV void build_item_desc(const DevIndex &ix, const DevWork &wk, uint32_t stages, uint32_t i, uint32_t seg, uint32_t cseg, long long ref_len,
                             const SegInfo &s, const CsegInfo &c) {
    const bool contig_fwd = c.is_fwd != 0;
    bool flip = false, rev = false;
    if (stages & PLO_STAGE_STRAND) {
        const bool changes = s.read_rev == s.seg_fwd;
        flip = (!contig_fwd) != changes;
        rev = !contig_fwd;
    }
    const long long pos1 = rev ? (long long)s.contig_len - (s.pos + ref_len) : s.pos;
    const int kv0 = (int)c.kv0, kv1 = (int)c.kv1;
    // every contig position the item can touch lies in [pos1, pos1 + ref_len]
    const long long lo = pos1 < -0x7fffffffLL ? -0x7fffffffLL : pos1;
    const long long hi = pos1 + ref_len > 0x7fffffffLL ? 0x7fffffffLL : pos1 + ref_len;
    int slo = kv0, shi = kv1;
    kv_upper_bound_narrow(ix.kv, slo, shi, (int)lo);
    // (the chromosome's two fields ride with the last level of the search)
    const uint64_t chrom_ref = (uint64_t)(uintptr_t)ix.chrom_seq[c.chrom];
    const int chrom_ref_len = ix.chrom_len[c.chrom];
    const int ub = kv_upper_bound_tail(ix.kv, slo, shi, (int)lo);
    PLO_LEVEL_END(chrom_ref, chrom_ref_len, ub);
    const int w0 = ub - 1 > kv0 ? ub - 1 : kv0;
    const int w1 = kv_lower_bound_near(ix.kv, w0, kv1, (int)hi);
    const bool do_shift = (stages & PLO_STAGE_LSHIFT) && (!(stages & PLO_STAGE_STRAND) || !contig_fwd);
    const bool merges = do_shift || (stages & PLO_STAGE_LIFTOVER);  // (the lane kernel's LOAD merges match runs for these stages)
    const int region = lane_region_dwords((int)(merges ? s.n_m : s.n_in), w0, w1);
    // ---- stores ----
    wk.item_seg[i] = seg;
    wk.item_cseg[i] = cseg;
    wk.item_nin[i] = (uint32_t)item_weight((int)s.n_in, w0, w1, kv1);  // tiling weight
    wk.item_cls[i] = (do_shift ? 1u : 0u) | ((wk.lane_max_w < 0 || region > wk.lane_max_w) ? 2u : 0u);
    if (wk.item_region) wk.item_region[i] = (uint32_t)region;
    wk.d.n_m[i] = s.n_m;
    wk.d.in_off[i] = s.in_off;
    wk.d.n_in[i] = s.n_in;
    wk.d.pos1[i] = (int)pos1;
    wk.d.w0[i] = (uint32_t)w0;
    wk.d.w1[i] = (uint32_t)w1;
    wk.d.kv0[i] = (uint32_t)kv0;
    wk.d.kv1[i] = (uint32_t)kv1;
    wk.d.flags[i] = (rev ? (uint32_t)ITF_REV : 0u) | (flip ? (uint32_t)ITF_FLIP : 0u) | (contig_fwd ? (uint32_t)ITF_CONTIG_FWD : 0u);
    wk.d.contig[i] = s.contig;
    wk.d.seq_len[i] = s.seq_len;
    wk.d.seq_off[i] = s.seq_off;
    wk.d.shift_ref[i] = s.revseq;
    wk.d.shift_ref_len[i] = s.contig_len;
    wk.d.chrom_ref[i] = chrom_ref;
    wk.d.chrom_ref_len[i] = chrom_ref_len;
    // The length check of src/read_alignment_scanner.rs:204-229 compares seq_len with the read bases the LIFTED CIGAR consumes.
    // Neither left_shift_indels nor liftover_read_alignment changes that number: the shift re-emits the same match / insertion
    // bases; the liftover turns every read-consuming piece into M, I or S of the same length (:102-123), copies I / S / H ops
    // (:157-160), and its edge clean-up turns insertions into clips of the same length.  So the check is made on the input CIGAR.
    wk.d.read_len[i] = s.read_len;
    // outputs that do not depend on the CIGAR pipeline
    wk.flip[i] = (uint8_t)flip;
    wk.mapq[i] = c.mapq;
    wk.chrom[i] = c.chrom;
}
// (an item of an explicit list: the three levels one after the other)
PLO_DEV void build_item_desc(const DevIndex &ix, const DevBatch &bt, const DevWork &wk, uint32_t stages, uint32_t i, uint32_t seg,
                             uint32_t cseg, long long ref_len) {
    SegInfo s;
    seg_info_level0(s, bt, wk, stages, seg);
    seg_info_level1(s, ix, bt, wk, stages, seg);
    build_item_desc(ix, wk, stages, i, seg, cseg, ref_len, s, cseg_info(ix, s.g0 + cseg));
}

// The items of one read segment, in contig-segment order, resolved at out_off (`s`: both levels loaded; the batch has been validated
// and the segment has items).  Returns their number.
PLO_DEV uint32_t emit_segment_items(const DevIndex &ix, const DevWork &wk, uint32_t stages, uint32_t seg, uint32_t out_off, long long ref_len,
                                    const SegInfo &s) {
    const long long r_start = s.pos, r_end = r_start + ref_len;
    uint32_t n = 0;
    for (uint32_t g = s.g0; g < s.g1; ++g) {
        const CsegInfo c = cseg_info(ix, g);
        // segment_range.intersect_range(&read_range): other.end >= self.start && other.start < self.end
        if ((r_end >= (long long)c.start) & (r_start < (long long)c.end)) {
            build_item_desc(ix, wk, stages, out_off + n, seg, g - s.g0, ref_len, s, c);
            ++n;
        }
    }
    return n;
}

// Counts (and, when wk != nullptr, resolves at out_off) the items of one read segment, in contig-segment order.
// `ref_len_cache`: per-segment reference spans; written by the counting pass (wk == nullptr), read by the emit pass.
PLO_DEV uint32_t enumerate_segment(const DevIndex &ix, const DevBatch &bt, uint32_t seg, const DevWork *wk, uint32_t stages,
                                   uint32_t out_off, int *ref_len_cache = nullptr, bool have_ref_len = false) {
    uint32_t contig = bt.seg_contig[seg];
    if (contig >= ix.n_contigs) return 0;
    uint32_t g0 = ix.contig_seg_off[contig], g1 = ix.contig_seg_off[contig + 1];
    if (g0 == g1) return 0;  // contig never seen in the asm->ref BAM (contig_alignment_scanner/mod.rs:364-367)
    long long ref_len;
    if ((wk || have_ref_len) && ref_len_cache) {
        ref_len = ref_len_cache[seg];
    } else {
        ref_len = segment_ref_len(bt, seg);
        if (ref_len_cache) ref_len_cache[seg] = (int)ref_len;
    }
    if (wk) {
        SegInfo s;
        seg_info_level0(s, bt, *wk, stages, seg);
        seg_info_level1(s, ix, bt, *wk, stages, seg);
        return emit_segment_items(ix, *wk, stages, seg, out_off, ref_len, s);
    }
    long long r_start = (long long)bt.seg_pos[seg];
    long long r_end = r_start + ref_len;
    uint32_t n = 0;
    for (uint32_t g = g0; g < g1; ++g)
        // segment_range.intersect_range(&read_range): other.end >= self.start && other.start < self.end
        if (r_end >= (long long)ix.cs_start[g] && r_start < (long long)ix.cs_end[g]) ++n;
    return n;
}

// Groups of the lane-per-item kernel cut by LDS budget: `reg[0 .. n)` = region dwords of the items of one sorted window, in order.  A
// group takes consecutive items while there are fewer than 64 of them and their regions fit `cap` dwords (at least one item: an item no
// slice holds goes to the retry list inside the kernel).  Returns the number of groups; group k = items [start[k], start[k + 1]).
// (A window sorted by weight has its long items together: cut into fixed 64s their group needs more LDS than the slice has and runs in
// several rounds -- cut by budget it simply has fewer lanes.)
template <class Reg>
PLO_HD uint32_t lane_groups_cut(const Reg *reg, uint32_t n, uint32_t cap, uint32_t *start, uint32_t max_groups) {
    uint32_t ng = 0, i = 0;
    while (i < n && ng < max_groups) {
        start[ng++] = i;
        uint32_t j = i, tot = 0;
        while (j < n && j - i < 64u && tot + (uint32_t)reg[j] <= cap) tot += (uint32_t)reg[j++];
        i = j > i ? j : i + 1;
    }
    start[ng] = n;  // (max_groups reached with items left over cannot happen for max_groups >= n: callers size it by the window)
    return ng;
}

// position of item i in class order, given the exclusive counts r0,r1,r2 of class-0/1/2 items before it and the class
// totals n0,n1,n2
PLO_DEV uint32_t class_order_pos(uint32_t i, uint32_t cls, uint32_t r0, uint32_t r1, uint32_t r2, uint32_t n0, uint32_t n1,
                                 uint32_t n2) {
    switch (cls) {
        case 0: return r0;
        case 1: return n0 + r1;
        case 2: return n0 + n1 + r2;
        default: return n0 + n1 + n2 + (i - r0 - r1 - r2);
    }
}

// first i in [0,n) with a[i] >= x
PLO_DEV uint32_t prefix_lower_bound(const uint32_t *a, uint32_t n, unsigned long long x) {
    uint32_t lo = 0, hi = n;
    while (lo < hi) {
        uint32_t mid = lo + ((hi - lo) >> 1);
        if ((unsigned long long)a[mid] < x)
            lo = mid + 1;
        else
            hi = mid;
    }
    return lo;
}

}  // namespace plo
