// lift_core.hpp -- the liftover tile algorithm, one wavefront per tile.
//
// A *tile* is up to 64 consecutive items (one item = one (read segment x contig segment) pair, i.e. one call of
// get_liftover_alignment_for_read_and_contig_segment, /root/reference/src/read_alignment_scanner.rs:136-288).
// The CIGAR ops of all items of the tile are laid out as ONE flattened op stream in LDS (item id per element), so
// every lane has an op to work on whatever the per-item CIGAR length is.  Lane t additionally owns the scalars
// of item t (position, block-map window, flags); elements fetch them with a cross-lane read (wv::shfl).
// The sequential state of the reference loops is recast as wave-level scans over the flattened stream:
//   * positions            : segmented exclusive sums   (add-scan + max-scan of the value at segment heads)
//   * "last X before me"   : max-scan over indices      (previous mapped piece, previous alive op, previous event)
//   * left-shift carry     : scan of min-plus function composition
//   * compaction           : add-scan of emission counts
// Chunks of 64 elements are processed with a running carry, so the same code serves small tiles (LDS) and
// single large items (wave-private global scratch).
//
// All citations are relative to /root/reference.
#pragma once
#include <type_traits>
#include <plo_wave.hpp>
#include <stdint.h>

#include "enumerate.hpp"
#include "lift_types.hpp"

namespace plo {

// Working storage of one tile: LDS for the tile kernel, wave-private global scratch for the large-item kernel.
struct TileMem {
    uint32_t *A, *B;    // ping-pong op arrays
    uint8_t *idA, *idB; // item id of each element (low 6 bits) + 2 flag bits
    int *T0, *T1, *T2, *T3, *T4;  // per-element temporaries
    int *itc;  // [64] per-item counters / scratch written by arbitrary lanes
    int *itf;  // [64] first match index  (clean-up)
    int *itl;  // [64] last match index   (clean-up)
    int *its;  // [64] leading-deletion shift (clean-up) / ref2_start (liftover)
    int *itp;  // [64] panic flags
    int *itq;  // [64] first piece of the item (liftover)
    int *K, *V;  // [capk] block-map entries (key, val) of the tile's items, staged once per tile
    int cap;
    int capk;
};
constexpr int tile_capk(int cap) { return cap / 2; }
// capk < 0: room for cap / 2 staged block-map entries (a tile of many short items); the workgroup-per-item kernel stages
// the window of ONE item and passes a fixed number (longer windows are read from global memory)
constexpr size_t tile_mem_bytes(int cap, int capk = -1) {
    return (size_t)cap * (4 + 4 + 1 + 1 + 5 * 4) + 6 * 64 * 4 + (size_t)(capk < 0 ? tile_capk(cap) : capk) * 8;
}

PLO_DEV TileMem carve_tile_mem(unsigned char *base, int cap, int capk = -1) {
    TileMem m;
    m.A = (uint32_t *)base;
    m.B = m.A + cap;
    m.T0 = (int *)(m.B + cap);
    m.T1 = m.T0 + cap;
    m.T2 = m.T1 + cap;
    m.T3 = m.T2 + cap;
    m.T4 = m.T3 + cap;
    m.itc = m.T4 + cap;
    m.itf = m.itc + 64;
    m.itl = m.itf + 64;
    m.its = m.itl + 64;
    m.itp = m.its + 64;
    m.itq = m.itp + 64;
    m.capk = capk < 0 ? tile_capk(cap) : capk;
    m.K = m.itq + 64;
    m.V = m.K + m.capk;
    m.idA = (uint8_t *)(m.V + m.capk);
    m.idB = m.idA + cap;
    m.cap = cap;
    return m;
}

// ---- op helpers (lib/rust-vc-utils/src/bam_utils/cigar/mod.rs:22-47, ignore_hard_clip = false) ----------------
PLO_DEV int op_type(uint32_t c) { return (int)(c & 15u); }
PLO_DEV int op_len(uint32_t c) { return (int)(c >> 4); }
PLO_DEV uint32_t mk_op(int type, int len) { return ((uint32_t)len << 4) | (uint32_t)type; }
PLO_DEV bool is_match(int t) { return t == OP_M || t == OP_EQ || t == OP_X; }
PLO_DEV bool ref_consuming(int t) { return (0x18D >> t) & 1; }   // M D N = X
PLO_DEV bool read_consuming(int t) { return (0x1B3 >> t) & 1; }  // M I S H = X
PLO_DEV bool is_indel(int t) { return t == OP_I || t == OP_D; }

// ---- sequence access --------------------------------------------------------------------------------------------
// comp_base, lib/rust-vc-utils/src/seq_util.rs:1-15
PLO_DEV int comp_base(int b) {
    // branch-free (a switch is lowered to exec-mask branches, ~140 instructions per inlined call): fold the case bit
    // away, map A<->T and C<->G, keep N/n, everything else is 'N'
    const int u = b & ~0x20, lower = b & 0x20;
    int r = 'N';
    r = (u == 'A') ? 'T' : r;
    r = (u == 'T') ? 'A' : r;
    r = (u == 'C') ? 'G' : r;
    r = (u == 'G') ? 'C' : r;
    const bool known = (r != 'N') || (u == 'N');
    return known ? (r | lower) : 'N';
}
// The read as the reference sees it after `record.seq().as_bytes()` (+ rev_comp_in_place when need_flipped),
// src/read_alignment_scanner.rs:170-173,238-241 -- never materialised: decoded / complemented per probe.
struct ReadSeq {
    const uint8_t *p;
    int len;
    int fmt;
    int flip;
    int lo, hi;  // bytes p[lo .. hi) belong to the batch's seq buffer (bounds for the wide-window loads), clamped to +-2^30
    int data;    // PLO_SEQ_BAM4_SPARSE: byte offset of the first granule behind the read's header
    bool miss;   // PLO_SEQ_BAM4_SPARSE: a probe needed bases the batch does not carry
};
// PLO_SEQ_BAM4_SPARSE (include/portello_liftover.h): the read's bases in granules of 32 (16 bytes of BAM 4-bit packing), only some
// of them present.  Header at p: one {u32 mask, u32 rank} pair per 32 granules (mask bit k = granule 32 b + k is present, rank =
// present granules before granule 32 b), padded to a multiple of 16 bytes; the present granules follow in ascending order.
// Stored bases [j0, j1] (at most two adjacent granules): byte offset of base j0's byte relative to p, or -1 when a granule is
// absent or the header points outside the buffer (garbage headers cannot make a probe leave the batch's buffer).
PLO_DEV int sparse_locate(const ReadSeq &r, int j0, int j1) {
    const int g0 = j0 >> 5, g1 = j1 >> 5;
    const PLO_GLOBAL uint32_t *hdr = (const PLO_GLOBAL uint32_t *)r.p;
    const uint32_t m0 = hdr[2 * (g0 >> 5)], r0 = hdr[2 * (g0 >> 5) + 1], m1 = hdr[2 * (g1 >> 5)];
    if (!((m0 >> (g0 & 31)) & 1u) || !((m1 >> (g1 & 31)) & 1u)) return -1;
    const unsigned rank = r0 + (unsigned)__builtin_popcount(m0 & ((1u << (g0 & 31)) - 1u));
    if (rank > 0x3ffffffu) return -1;
    const long long off = (long long)r.data + (long long)rank * 16 + ((j0 >> 1) & 15);
    if (off + (g1 - g0) * 16 + 20 > (long long)r.hi) return -1;
    return (int)off;
}
PLO_DEV int read_base(ReadSeq &r, int i) {
    int j = r.flip ? (r.len - 1 - i) : i;
    int c;
    if (r.fmt != PLO_SEQ_ASCII) {
        int at = j >> 1;
        if (r.fmt == PLO_SEQ_BAM4_SPARSE) {
            at = sparse_locate(r, j, j);
            if (at < 0) {
                r.miss = true;
                return 'N';
            }
        }
        int b = r.p[at];
        int nib = (j & 1) ? (b & 15) : (b >> 4);
        // "=ACMGRSVTWYHKDBN"
        const unsigned long long lo = 0x565352474d43413dull;  // = A C M G R S V
        const unsigned long long hi = 0x4e42444b48595754ull;  // T W Y H K D B N
        c = (int)(((nib & 8) ? hi : lo) >> ((nib & 7) * 8)) & 0xff;
    } else {
        c = r.p[j];
    }
    return r.flip ? comp_base(c) : c;
}
// SP: the kernel variant for batches with sparse bases.  The dense variants never see PLO_SEQ_BAM4_SPARSE (the format is folded to
// one of the two dense ones, so the compiler drops the granule look-ups from them: they cost the tile kernel registers otherwise).
template <bool SP>
PLO_DEV ReadSeq item_read_seq(const DevBatch &bt, unsigned long long seq_off, int seq_len, int flip) {
    ReadSeq r;
    r.p = bt.seq + seq_off;
    r.len = seq_len;
    r.fmt = SP ? (int)PLO_SEQ_BAM4_SPARSE : (bt.seq_fmt == PLO_SEQ_ASCII ? (int)PLO_SEQ_ASCII : (int)PLO_SEQ_BAM4);
    r.flip = flip;
    const unsigned long long before = seq_off, after = bt.seq_bytes - seq_off;
    r.lo = -(int)(before < (1ull << 30) ? before : (1ull << 30));
    r.hi = (int)(after < (1ull << 30) ? after : (1ull << 30));
    r.data = (int)sparse_header_bytes((uint32_t)seq_len);
    r.miss = false;
    return r;
}

// ---- cooperation of several waves on one tile ----------------------------------------------------------------------
// Coop<1>: the tile belongs to one wave (k_lift_tiles, k_lift_retry, k_lift_big).  Coop<NW>, NW > 1: the NW waves of a
// workgroup work on ONE item (k_lift_mid): a pass walks the element stream NW x 64 elements at a time, wave w taking
// elements [64 w, 64 w + 64) of every step; the scan carries cross the waves through a few LDS words and one workgroup
// barrier per scan; the item's scalars live in lane 0 of EVERY wave (replicated, kept identical) and reach the elements as
// scalar broadcasts (v_readfirstlane) instead of cross-lane reads.  Control flow must then be uniform over the whole
// workgroup wherever a scan, co.sync() or co.any() is called.
template <int NW>
struct Coop {
    static constexpr int NWAVES = NW;
    static constexpr int STEP = 64 * NW;
    static constexpr int XCH_INTS = 2 * 4 * NW + 4;  // two exchange buffers of 4 x NW words + 3 rotating flags
    int w = 0;           // index of the wave inside the workgroup
    int *xch = nullptr;  // LDS, XCH_INTS words, zeroed before the first use
    unsigned k = 0, kf = 0;

    PLO_DEV int lo() const { return NW > 1 ? 64 * w : 0; }  // first element of the wave in a step
    PLO_DEV bool lead() const { return NW == 1 || w == 0; }  // the wave that speaks for the item (global counters, outputs)
    PLO_DEV void sync() const {
        if constexpr (NW > 1) wv::block_sync();
        else wv::sync();
    }
    // scalar `v` of item `id`: lane id of the wave holds it (NW == 1); lane 0 of every wave holds the one item's (NW > 1)
    template <class T>
    PLO_DEV T item(T v, int id) const {
        if constexpr (NW > 1) return wv::bcast_first(v);
        else return wv::shfl(v, id);
    }
    PLO_DEV int item(bool v, int id) const { return item((int)v, id); }
    PLO_DEV int elem_id(const uint8_t *idX, int e, bool valid) const {
        if constexpr (NW > 1) return 0;
        else return valid ? (idX[e] & 63) : 0;
    }
    // true when `p` holds in any lane of the tile's wave(s)
    PLO_DEV bool any(bool p) {
        if constexpr (NW == 1) {
            return wv::ballot(p) != 0ull;
        } else {
            // three flags in rotation: call c uses flag c % 3, the lead wave clears the flag of call c + 1 before the barrier
            // of call c (its last readers, of call c - 2, have all passed the barrier of call c - 1)
            int *f = xch + 2 * 4 * NW;
            const unsigned c = kf++;
            const bool mine = wv::ballot(p) != 0ull;
            if (wv::lane() == 0) {
                if (w == 0) f[(c + 1) % 3] = 0;
                if (mine) f[c % 3] = 1;
            }
            wv::block_sync();
            return f[c % 3] != 0;
        }
    }
    // Exchange buffers alternate: a wave can write the buffer of call k + 2 only after passing the barrier of call k + 1,
    // which every wave reaches only after reading the buffer of call k.
    PLO_DEV int *slot() {
        int *s_ = xch + (k & 1u) * 4 * NW;
        ++k;
        return s_;
    }
    // One exchange (one barrier) for NA sums and NM maxima: per value, the total of the waves before this one (`pre`) and of
    // all waves of the step (`all`).  Every wave leaves its totals in the buffer; after the barrier lane l < NW reads wave l's
    // and a DPP scan over those lanes gives both numbers with two scalar reads.
    template <int NA, int NM>
    PLO_DEV void exch(const int *ta, const int *tm, int *prea, int *alla, int *prem, int *allm) {
        static_assert(NA + NM <= 4, "an exchange buffer holds four words per wave");
        int *s_ = slot();
        const int l = wv::lane();
        if (l == 0) {
#pragma unroll
            for (int i = 0; i < NA; ++i) s_[i * NW + w] = ta[i];
#pragma unroll
            for (int i = 0; i < NM; ++i) s_[(NA + i) * NW + w] = tm[i];
        }
        wv::block_sync();
        const bool in = l < NW;
        int va[NA > 0 ? NA : 1], vm[NM > 0 ? NM : 1];
#pragma unroll
        for (int i = 0; i < NA; ++i) va[i] = in ? s_[i * NW + l] : 0;
#pragma unroll
        for (int i = 0; i < NM; ++i) vm[i] = in ? s_[(NA + i) * NW + l] : (int)0x80000000;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            int inc = wv::scan_add(va[i]);
            alla[i] = wv::read_lane(inc, NW - 1);
            int p_ = wv::read_lane(inc, w - 1);
            prea[i] = w ? p_ : 0;
        }
#pragma unroll
        for (int i = 0; i < NM; ++i) {
            int inc = wv::scan_max(vm[i]);
            allm[i] = wv::read_lane(inc, NW - 1);
            int p_ = wv::read_lane(inc, w - 1);
            prem[i] = w ? p_ : (int)0x80000000;
        }
    }
    PLO_DEV void exch_add(int tot, int &pre, int &all) { exch<1, 0>(&tot, nullptr, &pre, &all, nullptr, nullptr); }
    PLO_DEV void exch_max(int tot, int &pre, int &all) { exch<0, 1>(nullptr, &tot, nullptr, nullptr, &pre, &all); }
    // min-plus functions: `pre` = carry composed with the waves before this one, `all` = carry composed with all waves
    PLO_DEV void exch_minplus(wv::MinPlus t, wv::MinPlus carry, wv::MinPlus &pre, wv::MinPlus &all) {
        int *s_ = slot();
        if (wv::lane() == 0) {
            s_[w] = t.a;
            s_[NW + w] = t.b;
            s_[2 * NW + w] = t.s;
        }
        wv::block_sync();
        wv::MinPlus acc = carry;
        pre = carry;
        for (int j = 0; j < NW; ++j) {
            if (j == w) pre = acc;
            wv::MinPlus c = {s_[j], s_[NW + j], s_[2 * NW + j]};
            acc = wv::mp_compose(acc, c);
        }
        all = acc;
    }
    // value of the previous element: lane - 1, or the last lane of the previous wave / of the previous step (`carry`)
    PLO_DEV int shift_up1(int v, int &carry) {
        if constexpr (NW == 1) {
            int r = wv::shfl_up1(v, carry);
            carry = wv::bcast_last(v);
            return r;
        } else {
            int last = wv::bcast_last(v);
            int *s_ = slot();
            if (wv::lane() == 0) s_[w] = last;
            wv::block_sync();
            int first = w ? s_[w - 1] : carry;
            carry = s_[NW - 1];
            return wv::shfl_up1(v, first);
        }
    }
    // a value of the lead wave, handed to every wave
    PLO_DEV unsigned long long from_lead(unsigned long long v) {
        if constexpr (NW == 1) {
            return v;
        } else {
            int *s_ = slot();
            if (w == 0 && wv::lane() == 0) {
                s_[0] = (int)(unsigned)(v & 0xffffffffull);
                s_[1] = (int)(unsigned)(v >> 32);
            }
            wv::block_sync();
            return ((unsigned long long)(unsigned)s_[1] << 32) | (unsigned)s_[0];
        }
    }
};
// elements of a pass: `base` runs over the chunks of 64 this wave handles; every wave makes the same number of steps
#define PLO_CHUNKS(base, n) for (int base = co.lo(); base - co.lo() < (n); base += Coop<NW>::STEP)

// ---- chunked scans with a running carry ---------------------------------------------------------------------------
template <int NW>
struct AddScanT {
    Coop<NW> &co;
    int carry = 0;
    PLO_DEV explicit AddScanT(Coop<NW> &c) : co(c) {}
    PLO_DEV int incl(int x) {
        int inc = wv::scan_add(x);
        int tot = wv::bcast_last(inc);
        int pre = 0, all = tot;
        if constexpr (NW > 1) co.exch_add(tot, pre, all);
        int r = carry + pre + inc;
        carry += all;
        return r;
    }
    PLO_DEV int excl(int x) { return incl(x) - x; }
};
template <int NW>
struct MaxScanT {
    Coop<NW> &co;
    int carry;
    int prev_carry;
    PLO_DEV MaxScanT(Coop<NW> &c, int init) : co(c), carry(init), prev_carry(init) {}
    PLO_DEV int incl(int x) {
        int m = wv::scan_max(x);
        int tot = wv::bcast_last(m);
        int pre = (int)0x80000000, all = tot;
        if constexpr (NW > 1) co.exch_max(tot, pre, all);
        prev_carry = wv::imax(carry, pre);  // inclusive value just before this wave's first lane
        carry = wv::imax(carry, all);
        return wv::imax(m, prev_carry);
    }
    // exclusive value belonging to the last incl() call
    PLO_DEV int excl_of(int incl_value) { return wv::shfl_up1(incl_value, prev_carry); }
};
// segmented exclusive sum of non-negative values: plain prefix minus the prefix at the segment head, the latter
// propagated by a max-scan (prefixes are non-decreasing).  One item per workgroup: its only head is element 0.
template <int NW>
struct SegSumT {
    AddScanT<NW> a;
    MaxScanT<NW> m;
    PLO_DEV explicit SegSumT(Coop<NW> &c) : a(c), m(c, 0) {}
    PLO_DEV int excl(int x, bool head) {
        int p = a.excl(x);
        if constexpr (NW > 1) {
            return p;
        } else {
            int base = m.incl(head ? p : 0);
            return p - base;
        }
    }
};
// NA add-scans and NM max-scans of the same step with ONE exchange between the waves (Coop<NW>, NW > 1); for a single wave
// they are simply independent scans.
template <int NW, int NA, int NM>
struct MultiScanT {
    Coop<NW> &co;
    int ca[NA > 0 ? NA : 1];
    int cm[NM > 0 ? NM : 1], pm[NM > 0 ? NM : 1];
    PLO_DEV MultiScanT(Coop<NW> &c, int max_init) : co(c) {
#pragma unroll
        for (int i = 0; i < NA; ++i) ca[i] = 0;
#pragma unroll
        for (int i = 0; i < NM; ++i) cm[i] = pm[i] = max_init;
    }
    // xa / xm: the lane's inputs; ra / rm: inclusive results
    PLO_DEV void incl(const int *xa, const int *xm, int *ra, int *rm) {
        int ia[NA > 0 ? NA : 1], im[NM > 0 ? NM : 1], ta[NA > 0 ? NA : 1], tm[NM > 0 ? NM : 1];
        int prea[NA > 0 ? NA : 1], alla[NA > 0 ? NA : 1], prem[NM > 0 ? NM : 1], allm[NM > 0 ? NM : 1];
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            ia[i] = wv::scan_add(xa[i]);
            ta[i] = wv::bcast_last(ia[i]);
            prea[i] = 0;
            alla[i] = ta[i];
        }
#pragma unroll
        for (int i = 0; i < NM; ++i) {
            im[i] = wv::scan_max(xm[i]);
            tm[i] = wv::bcast_last(im[i]);
            prem[i] = (int)0x80000000;
            allm[i] = tm[i];
        }
        if constexpr (NW > 1) co.template exch<NA, NM>(ta, tm, prea, alla, prem, allm);
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            ra[i] = ca[i] + prea[i] + ia[i];
            ca[i] += alla[i];
        }
#pragma unroll
        for (int i = 0; i < NM; ++i) {
            pm[i] = wv::imax(cm[i], prem[i]);
            cm[i] = wv::imax(cm[i], allm[i]);
            rm[i] = wv::imax(im[i], pm[i]);
        }
    }
    PLO_DEV int excl_of(int k, int incl_value) { return wv::shfl_up1(incl_value, pm[k]); }
};

using AddScan = AddScanT<1>;
using MaxScan = MaxScanT<1>;

// appends one op of item `id` at Y[p] when `on` (p advances): the emission passes write up to four ops per element this way
PLO_DEV void put_op(uint32_t *Y, uint8_t *idY, int &p, bool on, uint32_t v, int id) {
    if (on) {
        Y[p] = v;
        idY[p] = (uint8_t)id;
    }
    p += on ? 1 : 0;
}

// Per-item counts/starts of an output array from the inclusive emission prefix E[] stored per *input* element
// (input items are contiguous: item t = [s, s+c)).
PLO_DEV void finish_counts(const int *E, int s, int c, int &ns, int &nc) {
    int before = (s > 0) ? E[s - 1] : 0;
    if (c > 0) {
        ns = before;
        nc = E[s + c - 1] - before;
    } else {
        ns = before;
        nc = 0;
    }
}

// -------------------------------------------------------------------------------------------------------------------
// clean_up_cigar_edge_indels + compress_cigar  (lib/rust-vc-utils/src/bam_utils/cigar/mod.rs:265-291, 204-228)
// X (n elements, items [s,s+c) per lane) -> Y.  Items with active == false are copied verbatim.
// Returns the leading-deletion shift of the lane's item in `shift`, and in `indel_pairs` (wave-uniform) whether some
// active item's output has two neighbouring I/D ops, i.e. an indel cluster of more than one op.
// -------------------------------------------------------------------------------------------------------------------
template <int NW>
PLO_DEV void cleanup_compress(Coop<NW> &co, TileMem &m, uint32_t *X, uint8_t *idX, uint32_t *Y, uint8_t *idY, int n, int &s, int &c,
                              bool active, int &shift, int &n_out, bool &indel_pairs, bool edges_known = false) {
    const int lane = wv::lane();
    if (!edges_known) {
        m.itf[lane] = IMAX;
        m.itl[lane] = -1;
    }
    m.its[lane] = 0;
    co.sync();
    // pass 1: first / last alignment-match element of every item (the edges are everything outside them); a producer that
    // knows where it puts its match ops publishes m.itf / m.itl itself (edges_known) and only the run sums are zeroed
    if (edges_known) {
        PLO_CHUNKS(base, n) {
            int e = base + lane;
            if (e < n) Y[e] = 0;
        }
    } else if constexpr (NW > 1) {  // one item: its first / last match are a minimum and a maximum, no scan (no exchange)
        const int i_act = co.item((int)active, 0);
        int fmin = IMAX, lmax = -1;
        PLO_CHUNKS(base, n) {
            int e = base + lane;
            if (e < n) {
                if (i_act && is_match(op_type(X[e]))) {
                    fmin = wv::imin(fmin, e);
                    lmax = e;  // a lane's elements come in increasing order
                }
                Y[e] = 0;
            }
        }
        fmin = -wv::reduce_max(-fmin);
        lmax = wv::reduce_max(lmax);
        if (lane == 0) {
            wv::atomic_min(&m.itf[0], fmin);
            wv::atomic_max(&m.itl[0], lmax);
        }
    } else {
        MaxScanT<NW> lastm(co, -1);
        PLO_CHUNKS(base, n) {
            int e = base + lane;
            bool valid = e < n;
            int id = co.elem_id(idX, e, valid);
            int i_s = co.item(s, id);
            int i_c = co.item(c, id);
            int i_act = co.item((int)active, id);
            bool ism = valid && i_act && is_match(op_type(X[e]));
            int li = lastm.incl(ism ? e : -1);
            int le = lastm.excl_of(li);
            if (ism && le < i_s) m.itf[id] = e;                        // first match of the item: single writer
            if (valid && e == i_s + i_c - 1) m.itl[id] = (li >= i_s) ? li : -1;  // last element publishes the last match
            if (valid) Y[e] = 0;  // run sums of pass 2 start from zero
        }
    }
    co.sync();
    // pass 2: edge I -> S(len), edge D -> S(0) (+ leading D lengths summed into the position shift), applied on the fly;
    // then drop zero-length ops and merge equal neighbours (head flags + run sums)
    {
        MaxScanT<NW> lasta(co, -1);
        AddScanT<NW> heads(co);
        bool pairs = false;
        int shift_acc = 0;  // NW > 1: per-lane sum of the one item, one LDS atomic per wave at the end
        PLO_CHUNKS(base, n) {
            int e = base + lane;
            bool valid = e < n;
            int id = co.elem_id(idX, e, valid);
            int i_s = co.item(s, id);
            int i_act = co.item((int)active, id);
            uint32_t cc = valid ? X[e] : 0;
            int t = op_type(cc), L = op_len(cc);
            int f = IMAX, l = -1;
            if (valid && i_act) {
                f = m.itf[id];
                l = m.itl[id];
                if (e < f || e > l) {  // clean_up_cigar_edge_indels (:265-291)
                    if (t == OP_D) {
                        if constexpr (NW > 1) shift_acc += e < f ? L : 0;
                        else if (e < f) wv::atomic_add(&m.its[id], L);
                        t = OP_S;
                        L = 0;
                    } else if (t == OP_I) {
                        t = OP_S;
                    }
                }
            }
            bool alive = valid && (!i_act || L > 0);
            int ai = lasta.incl(alive ? e : -1);
            int pa = lasta.excl_of(ai);
            bool head = false;
            if (alive) {
                if (!i_act) {
                    head = true;
                } else {
                    int pt = -1;
                    if (pa >= i_s) {  // the previous surviving op of the item, after its own edge clean-up (a surviving edge op
                                      // is an insertion turned soft clip; edge deletions have length 0 and do not survive)
                        pt = op_type(X[pa]);
                        if ((pa < f || pa > l) && pt == OP_I) pt = OP_S;
                    }
                    head = pt != t;
                    pairs |= head && is_indel(t) && pt >= 0 && is_indel(pt);
                }
            }
            int hi = heads.incl(head ? 1 : 0);
            if (alive) {
                int r = hi - 1;
                // Pad is absent from the summing pattern (:210-212): a Pad following a Pad adds nothing
                uint32_t add = (uint32_t)((t == OP_P && !head) ? 0 : L) << 4;
                if (head) {
                    add |= (uint32_t)t;
                    idY[r] = (uint8_t)id;
                }
                wv::atomic_add(&Y[r], add);
            }
            if (valid) m.T0[e] = hi;
        }
        n_out = heads.carry;
        if constexpr (NW > 1) {
            shift_acc = wv::reduce_add(shift_acc);
            if (lane == 0 && shift_acc) wv::atomic_add(&m.its[0], shift_acc);
        }
        indel_pairs = co.any(pairs);
    }
    co.sync();
    int ns, nc;
    finish_counts(m.T0, s, c, ns, nc);
    shift = m.its[lane];
    s = ns;
    c = nc;
    co.sync();
}

// ---- sequence comparison ------------------------------------------------------------------------------------------
// The byte probes of the homology / trimming loops are random accesses into HBM-resident sequences.  They are made 16
// bases at a time: both 16-base windows are fetched as aligned dwords (8 loads, ONE memory round trip), the read side is
// decoded 4 bases per v_perm_b32 (the 16-entry "=ACMGRSVTWYHKDBN" table, or its complement for a flipped read, split into
// two 8-byte halves), and the XOR of the two windows gives the match run with one clz / ctz.

PLO_DEV unsigned comp4(unsigned w) {  // comp_base on the four ASCII bytes of w
    return (unsigned)comp_base((int)(w & 0xffu)) | ((unsigned)comp_base((int)((w >> 8) & 0xffu)) << 8) |
           ((unsigned)comp_base((int)((w >> 16) & 0xffu)) << 16) | ((unsigned)comp_base((int)(w >> 24)) << 24);
}

// X byte t (little-endian over X[0..3]) = ref[r0 + t] ^ read_base(rd, q0 + t), t = 0..15.
// Returns false when a window does not lie inside its buffer: the caller then compares byte-wise.
PLO_DEV bool xor_window16(const uint8_t *ref, int ref_len, int r0, ReadSeq &rd, int q0, unsigned X[4]) {
    // all positions are below 2^31 (BAM coordinates): 32-bit index arithmetic throughout
    if (r0 < 0 || q0 < 0 || q0 > rd.len - 16) return false;
    const int rsh = (int)(((unsigned)(uintptr_t)ref + (unsigned)r0) & 3u);
    if (r0 - rsh < 0 || r0 - rsh > ref_len - 20) return false;
    // read bases q0 .. q0+15 are the stored positions jmin .. jmin+15 (in reverse order when flipped)
    const int jmin = rd.flip ? rd.len - q0 - 16 : q0;
    const bool bam4 = rd.fmt != PLO_SEQ_ASCII;
    int b0 = bam4 ? (jmin >> 1) : jmin;  // first byte of the read window
    if (rd.fmt == PLO_SEQ_BAM4_SPARSE) {
        b0 = sparse_locate(rd, jmin, jmin + 15);
        if (b0 < 0) {  // the batch does not carry these bases: the item is reported PLO_ITEM_NEED_BASES, the probe result is not used
            rd.miss = true;
            X[0] = X[1] = X[2] = X[3] = 0xffffffffu;
            return true;
        }
    }
    const int qsh = (int)(((unsigned)(uintptr_t)rd.p + (unsigned)b0) & 3u);
    const int qwords = bam4 ? 3 : 5;
    if (b0 - qsh < rd.lo || b0 - qsh + 4 * qwords > rd.hi) return false;
    const PLO_GLOBAL uint32_t *pr = (const PLO_GLOBAL uint32_t *)(ref + (r0 - rsh));
    const PLO_GLOBAL uint32_t *pq = (const PLO_GLOBAL uint32_t *)(rd.p + (b0 - qsh));
    unsigned wr[5], wq[5];
#pragma unroll
    for (int u = 0; u < 5; ++u) wr[u] = pr[u];
#pragma unroll
    for (int u = 0; u < 5; ++u) wq[u] = u < qwords ? pq[u] : 0u;
    unsigned D[4];  // stored bases jmin .. jmin+15 as ASCII (complemented when flipped), base jmin+t at byte t
    if (bam4) {
        // nibble s = par + t of the byte stream: byte s>>1, high nibble when s is even
        unsigned S[5];
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            unsigned Q = wv::align_bytes(wq[q + 1], wq[q], (unsigned)qsh);
            unsigned H = (Q >> 4) & 0x0f0f0f0fu, L = Q & 0x0f0f0f0fu;
            S[2 * q] = wv::perm_bytes(L, H, 0x05010400u);  // nibbles 8q .. 8q+3, one per byte
            if (2 * q + 1 < 5) S[2 * q + 1] = wv::perm_bytes(L, H, 0x07030602u);
        }
        const unsigned par = (unsigned)(jmin & 1);
        // "=ACMGRSVTWYHKDBN" and its comp_base image "NTGNCNNNANNNNNNN", 8 entries per 64-bit half
        const unsigned long long lo = rd.flip ? 0x4e4e4e434e47544eull : 0x565352474d43413dull;
        const unsigned long long hi = rd.flip ? 0x4e4e4e4e4e4e4e41ull : 0x4e42444b48595754ull;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            unsigned T = wv::align_bytes(S[v + 1], S[v], par);
            unsigned idx = T & 0x07070707u;
            unsigned dl = wv::perm_bytes((unsigned)(lo >> 32), (unsigned)lo, idx);
            unsigned dh = wv::perm_bytes((unsigned)(hi >> 32), (unsigned)hi, idx);
            unsigned mk = ((T >> 3) & 0x01010101u) * 0xffu;
            D[v] = (dh & mk) | (dl & ~mk);
        }
    } else {
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            D[v] = wv::align_bytes(wq[v + 1], wq[v], (unsigned)qsh);
            if (rd.flip) D[v] = comp4(D[v]);
        }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        unsigned R = wv::align_bytes(wr[u + 1], wr[u], (unsigned)rsh);
        unsigned B = rd.flip ? wv::perm_bytes(0u, D[3 - u], 0x00010203u) : D[u];
        X[u] = R ^ B;
    }
    return true;
}
PLO_DEV int zero_bytes_from_top(const unsigned X[4]) {  // t = 15, 14, ...
    if (X[3]) return wv::clz32(X[3]) >> 3;
    if (X[2]) return 4 + (wv::clz32(X[2]) >> 3);
    if (X[1]) return 8 + (wv::clz32(X[1]) >> 3);
    if (X[0]) return 12 + (wv::clz32(X[0]) >> 3);
    return 16;
}
PLO_DEV int zero_bytes_from_bottom(const unsigned X[4]) {  // t = 0, 1, ...
    if (X[0]) return wv::ctz32(X[0]) >> 3;
    if (X[1]) return 4 + (wv::ctz32(X[1]) >> 3);
    if (X[2]) return 8 + (wv::ctz32(X[2]) >> 3);
    if (X[3]) return 12 + (wv::ctz32(X[3]) >> 3);
    return 16;
}

// number of k in [0, maxk) with ref[re-1-k] == read[qe-1-k], stopping at the first mismatch.  All indices are valid
// (checked by the callers); `probes` counts the compared base pairs like the reference's loop would.
PLO_DEV int match_run_back(const uint8_t *ref, int ref_len, int re, ReadSeq &rd, int qe, int maxk, int &probes) {
    int k = 0;
    while (k < maxk) {
        unsigned X[4];
        if (!xor_window16(ref, ref_len, re - k - 16, rd, qe - k - 16, X)) break;
        int n = wv::imin(16, maxk - k);
        int m = wv::imin(zero_bytes_from_top(X), n);
        probes += wv::imin(m + 1, n);
        k += m;
        if (m < n) return k;
    }
    while (k < maxk) {  // window outside a buffer (first / last bases of a sequence): 8 independent byte probes a round
        int n = wv::imin(8, maxk - k);
        int a[8], b[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            int jj = j < n ? j : 0;
            a[j] = ((const PLO_GLOBAL uint8_t *)ref)[re - 1 - k - jj];
            b[j] = read_base(rd, qe - 1 - k - jj);
        }
        int adv = n;
#pragma unroll
        for (int j = 7; j >= 0; --j)
            if (j < n && a[j] != b[j]) adv = j;
        probes += wv::imin(adv + 1, n);
        k += adv;
        if (adv < n) break;
    }
    return k;
}
// number of k in [0, maxk) with ref[rs+k] == read[qs+k], stopping at the first mismatch
PLO_DEV int match_run_fwd(const uint8_t *ref, int ref_len, int rs, ReadSeq &rd, int qs, int maxk, int &probes) {
    int k = 0;
    while (k < maxk) {
        unsigned X[4];
        if (!xor_window16(ref, ref_len, rs + k, rd, qs + k, X)) break;
        int n = wv::imin(16, maxk - k);
        int m = wv::imin(zero_bytes_from_bottom(X), n);
        probes += wv::imin(m + 1, n);
        k += m;
        if (m < n) return k;
    }
    while (k < maxk) {
        int n = wv::imin(8, maxk - k);
        int a[8], b[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            int jj = j < n ? j : 0;
            a[j] = ((const PLO_GLOBAL uint8_t *)ref)[rs + k + jj];
            b[j] = read_base(rd, qs + k + jj);
        }
        int adv = n;
#pragma unroll
        for (int j = 7; j >= 0; --j)
            if (j < n && a[j] != b[j]) adv = j;
        probes += wv::imin(adv + 1, n);
        k += adv;
        if (adv < n) break;
    }
    return k;
}

// left homology of get_indel_breakend_homology_info (lib/rust-vc-utils/src/indel_breakend_homology.rs:32-47),
// capped at `bound` (the result is only used as min(match_run, h), cigar_indel_shifter.rs:132-133).
PLO_DEV int left_homology(const uint8_t *ref, int ref_len, int rs, int del, ReadSeq &rd, int qs, int ins, int bound,
                          bool &panic, int &probes) {
    int re = rs + del, qe = qs + ins;
    int max_left = wv::imin(rs, qs);  // max_left_offset (:32)
    int maxk = wv::imin(max_left, bound);
    if (max_left > 0) {
        // the first probe has the largest indices; out of bounds = slice-index panic in the reference (:38-39)
        if (re - 1 >= ref_len || qe - 1 >= rd.len) {
            panic = true;
            return 0;
        }
    }
    return match_run_back(ref, ref_len, re, rd, qe, maxk, probes);
}

// State a (persistent) wave carries from tile to tile.  All fields except the per-lane statistics are wave-uniform.
// Output CIGARs are bump-allocated in slabs: one device-scope atomic per SLAB_OPS ops instead of one per tile -- with
// hundreds of thousands of tiles a single hot counter (~88 atomics/us per address) would otherwise bound the kernel.
struct WaveCtx {
    unsigned long long slab_base = 0;  // next free op of the wave's current slab
    unsigned long long slab_left = 0;  // ops left in it
    // per lane, 32 bits: a wave's share of one launch stays far below 2^32 bytes / ops per lane (a batch is at most 2^31 ops in all)
    unsigned algo_bytes = 0;
    unsigned in_ops = 0;
    unsigned out_ops = 0;
    // lane-per-item kernels: lanes at work and trips of the liftover loop and of the shift stage's event rounds (wave-uniform; from the
    // loops' own ballots, scalar instructions only) -> plo_timing::lane_utilisation
    unsigned long long u_act = 0, u_trips = 0;
#ifdef PLO_PHASE_TIMING
    long long tph[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};  // shader cycles per pipeline phase, flushed once per wave
    long long t_begin = wv::realtime();  // the wave's life on the constant 100 MHz clock (plo_ctx_wave_clocks)
    // lane kernel: groups lifted, the longest one's ticks, the last two groups' ticks, when the last one began
    long long g_n = 0, g_max = 0, g_last = 0, g_prev = 0, g_last_begin = 0;
#endif
};
constexpr unsigned long long SLAB_OPS = 16384;
#ifdef PLO_PHASE_TIMING
constexpr int STAT_WORDS = 16;  // (timing builds: + the wave's life and its groups' durations, plo_ctx_wave_clocks)
#else
constexpr int STAT_WORDS = 8;  // 64-bit words per statistics slot of a wave (five in use)
#endif

// A retiring wave leaves its statistics in its own slot (plain stores): 3 atomics per wave on one cache line used to cost a
// fixed ~75 us per launch (3 072 waves x 3 atomics at ~88 per microsecond and address).  k_sum_stats adds the slots up.
PLO_DEV void wave_ctx_flush(const DevWork &wk, WaveCtx &ctx, uint32_t slot) {
    unsigned lo = (unsigned)wv::reduce_add((int)(unsigned)(ctx.algo_bytes & 0xffffffu));
    unsigned hi = (unsigned)wv::reduce_add((int)(unsigned)(ctx.algo_bytes >> 24));
    unsigned nlo = (unsigned)wv::reduce_add((int)(unsigned)(ctx.in_ops & 0xffffffu));
    unsigned nhi = (unsigned)wv::reduce_add((int)(unsigned)(ctx.in_ops >> 24));
    unsigned olo = (unsigned)wv::reduce_add((int)(unsigned)(ctx.out_ops & 0xffffffu));
    unsigned ohi = (unsigned)wv::reduce_add((int)(unsigned)(ctx.out_ops >> 24));
    if (wv::lane() == 0) {
        unsigned long long *w = wk.wave_stats + (size_t)(wk.stat_base + slot) * STAT_WORDS;
        w[0] = (unsigned long long)lo + ((unsigned long long)hi << 24);
        w[1] = (unsigned long long)nlo + ((unsigned long long)nhi << 24);
        w[2] = (unsigned long long)olo + ((unsigned long long)ohi << 24);
        w[3] = ctx.u_act;
        w[4] = ctx.u_trips;
    }
    ctx.u_act = 0;
    ctx.u_trips = 0;
    ctx.algo_bytes = 0;
    ctx.in_ops = 0;
    ctx.out_ops = 0;
#ifdef PLO_PHASE_TIMING
    if (wv::lane() == 0) {
        for (int k = 0; k < 12; ++k) wv::atomic_add_global(&wk.counters[CNT_PHASE0 + k], (unsigned long long)ctx.tph[k]);
        unsigned long long *w = wk.wave_stats + (size_t)(wk.stat_base + slot) * STAT_WORDS;  // (k_sum_stats leaves these three alone)
        w[5] = (unsigned long long)ctx.t_begin;
        w[6] = (unsigned long long)wv::realtime();
        w[7] = (unsigned long long)wv::hw_id();
        w[8] = (unsigned long long)ctx.g_n;
        w[9] = (unsigned long long)ctx.g_max;
        w[10] = (unsigned long long)ctx.g_last;
        w[11] = (unsigned long long)ctx.g_prev;
        w[12] = (unsigned long long)ctx.g_last_begin;
    }
#endif
}

// -------------------------------------------------------------------------------------------------------------------
// The tile pipeline
// -------------------------------------------------------------------------------------------------------------------
// What a persistent wave fetches about a tile one iteration ahead (lift_tiles_persistent): lane t <-> item t of the tile's
// first 64 items -- the item index and the descriptor fields needed to issue the CIGAR / block-map gathers.
struct TilePre {
    uint32_t g = 0, in_off = 0, n_in = 0, w0 = 0, w1 = 0, kv1 = 0, fl = 0;
};

template <int NW, bool SP = false>
PLO_DEV void lift_tile(Coop<NW> &co, const DevIndex &ix, const DevBatch &bt, const DevWork &wk, uint32_t stages, uint32_t item_begin,
                       int nit, TileMem m, const uint32_t *list, int level, int big_thresh, WaveCtx &ctx, const TilePre *pre = nullptr) {
    // NW > 1 (k_lift_mid): nit == 1, every wave of the workgroup runs this function on the same item; lane 0 of every wave
    // carries the item's scalars (kept identical), the lead wave alone touches global counters and the per-item outputs
    // level: LEVEL_TILE (shared tile; a tile that overflows its capacity re-queues its items on the retry list),
    // LEVEL_RETRY (few items, larger capacity; overflow -> large-item list), LEVEL_LAST (one item, global scratch)
    const bool last_resort = level == LEVEL_LAST;
    const int lane = wv::lane();
#ifdef PLO_PHASE_TIMING
    long long tlast = wv::clock();
#define PLO_T(k)                      \
    {                                 \
        long long now_ = wv::clock(); \
        ctx.tph[k] += now_ - tlast;   \
        tlast = now_;                 \
    }
#else
#define PLO_T(k)
#endif
    bool has = lane < nit;
    // `list` maps positions to item indices: the class-order permutation (tiles), the retry list, or the large-item list
    const uint32_t g = pre ? pre->g : (has ? list[item_begin + (uint32_t)lane] : 0u);

    // ---- item descriptors: lane t <-> item t, resolved by build_item_desc (enumerate.hpp): one level of coalesced loads
    int n_in = 0, in_off = 0, pos1 = 0, kv0 = 0, kv1 = 0, W0 = 0, W1 = 0, seq_len = 0;
    int shift_ref_len = 0;
    bool len_bad = false;  // LENGTH CHECK, decided from the descriptors (see there)
    unsigned long long seq_off = 0, shift_ref = 0;
    bool rev = false, do_shift = false, flip = false;
    if (has) {
        uint32_t fl;
        if (pre) {  // fetched while the previous tile was being processed
            in_off = (int)pre->in_off;
            n_in = (int)pre->n_in;
            W0 = (int)pre->w0;
            W1 = (int)pre->w1;
            kv1 = (int)pre->kv1;
            fl = pre->fl;
        } else {
            in_off = (int)wk.d.in_off[g];
            n_in = (int)wk.d.n_in[g];
            W0 = (int)wk.d.w0[g];
            W1 = (int)wk.d.w1[g];
            kv1 = (int)wk.d.kv1[g];
            fl = wk.d.flags[g];
        }
        pos1 = wk.d.pos1[g];
        kv0 = (int)wk.d.kv0[g];
        seq_len = (int)wk.d.seq_len[g];
        {
            const uint32_t read_len_in = wk.d.read_len[g];
            len_bad = read_len_in == 0xffffffffu || (uint32_t)seq_len != read_len_in;
        }
        seq_off = wk.d.seq_off[g];
        shift_ref = wk.d.shift_ref[g];
        shift_ref_len = wk.d.shift_ref_len[g];
        rev = (fl & ITF_REV) != 0;
        flip = (fl & ITF_FLIP) != 0;
        bool contig_fwd = (fl & ITF_CONTIG_FWD) != 0;
        do_shift = (stages & PLO_STAGE_LSHIFT) && (!(stages & PLO_STAGE_STRAND) || !contig_fwd);
    }
    // Items too heavy for this level are handed on: from a shared tile to the workgroup-per-item kernel (big_list), from
    // there to the one-wave-per-item kernel that works in global scratch (huge_list)
    uint32_t *const next_list = level == LEVEL_MID ? wk.huge_list : wk.big_list;
    const int next_cnt = level == LEVEL_MID ? CNT_NHUGE : CNT_NBIG;
    {
        bool defer = has && !last_resort && item_weight(n_in, W0, W1, kv1) > big_thresh;
        unsigned long long dm = wv::ballot(defer);
        if (dm != 0ull) {
            int nd = __builtin_popcountll(dm);
            int slot = 0;
            if (lane == 0 && co.lead()) slot = (int)wv::atomic_add_global(&wk.counters[next_cnt], (unsigned long long)nd);
            slot = wv::bcast_first(slot);
            if (defer) {
                int rank = __builtin_popcountll(dm & ((1ull << lane) - 1ull));
                if (co.lead()) {
                    next_list[slot + rank] = g;
                    wk.status[g] = (uint8_t)ITEM_NEED_BIG;
                }
                has = false;
                n_in = 0;
            }
        }
    }
    int status = PLO_ITEM_LIFTED;
    bool alive = has;  // still flowing through the pipeline
    if (has && do_shift && shift_ref == 0ull) {
        status = PLO_ITEM_PANIC;  // rev_contig_seq.unwrap() on None (:174)
        alive = false;
        do_shift = false;
    }
    bool overflow = false;
    unsigned algo_bytes = 0;

    // Block-map windows of the tile's items -> LDS (m.K / m.V), flattened: entry j of the tile belongs to the item whose
    // [kb, kb + nk) contains j.  Only the source indices are computed here (into the idle ping-pong array); the loads go out
    // together with the CIGAR gather below, so that both cost one memory round trip per tile.
    const int nk_all = (has && (stages & PLO_STAGE_LIFTOVER)) ? (wv::imin(W1 + 1, kv1) - W0) : 0;
    const int inck = wv::scan_add(nk_all);
    const int kb = inck - nk_all;
    const int nkT = wv::bcast_last(inck);
    const bool staged = nkT <= m.capk;
    int *const srcK = (int *)m.B;
    if constexpr (NW > 1) {  // one item: entry j of the staged window is entry W0 + j of the map, no owner search
        if (staged && nkT > 0) {
            const int i_w0 = co.item(W0, 0);
            PLO_CHUNKS(base, nkT) {
                int j = base + lane;
                if (j < nkT) srcK[j] = i_w0 + j;
            }
            co.sync();
        }
    } else if (staged && nkT > 0) {
        PLO_CHUNKS(base, nkT) {
            int j = base + lane;
            if (j < nkT) m.T3[j] = 0;
        }
        co.sync();
        if (nk_all > 0) m.T3[kb] = lane + 1;
        co.sync();
        MaxScanT<NW> owner(co, 0);
        PLO_CHUNKS(base, nkT) {
            int j = base + lane;
            bool valid = j < nkT;
            int id = owner.incl(valid ? m.T3[j] : 0) - 1;
            if (id < 0) id = 0;
            int i_w0 = co.item(W0, id), i_kb = co.item(kb, id);
            if (valid) srcK[j] = i_w0 + (j - i_kb);
        }
        co.sync();
    }
    PLO_T(0)
    // ---- LOAD: flattened op stream of the tile (reversed for reverse-mapped contig segments, :167) --------------
    int cA = has ? n_in : 0;
    int incA = wv::scan_add(cA);
    int sA = incA - cA;
    int nA = wv::bcast_last(incA);
    if (nA > m.cap) overflow = true;
    m.itp[lane] = 0;
    if (!overflow) {
        if constexpr (NW > 1) {  // one item: every element is its
            const int i_off = co.item(in_off, 0), i_n = co.item(n_in, 0), i_rev = co.item((int)rev, 0);
            PLO_CHUNKS(base, nA) {
                int e = base + lane;
                if (e < nA) {
                    m.T0[e] = i_off + (i_rev ? (i_n - 1 - e) : e);
                    m.idA[e] = 0;
                    m.T3[e] = 0;
                    m.T4[e] = 0;
                }
            }
        } else {
        PLO_CHUNKS(base, nA) {
            int e = base + lane;
            if (e < nA) m.T3[e] = 0;
        }
        co.sync();
        if (has && cA > 0) m.T3[sA] = lane + 1;
        co.sync();
        MaxScanT<NW> owner(co, 0);
        PLO_CHUNKS(base, nA) {
            int e = base + lane;
            bool valid = e < nA;
            int id = owner.incl(valid ? m.T3[e] : 0) - 1;
            if (id < 0) id = 0;
            int i_off = co.item(in_off, id), i_n = co.item(n_in, id), i_s = co.item(sA, id);
            int i_rev = co.item((int)rev, id);
            if (valid) {
                int k = e - i_s;
                m.T0[e] = i_off + (i_rev ? (i_n - 1 - k) : k);  // source op of the element
                m.idA[e] = (uint8_t)id;
                m.T3[e] = 0;
                m.T4[e] = 0;
            }
        }
        }
        // the gather carries no cross-lane dependency: the loads of four chunks are in flight together (indices first,
        // then all loads, then all stores -- written out because the compiler must assume that m.A aliases m.T0 and
        // would otherwise finish every chunk's store before the next chunk's index read)
        const int nkS = staged ? nkT : 0;  // block-map entries to stage
        if constexpr (NW > 1) co.sync();  // the gather below cuts the elements differently (256 per wave and step)
        for (int base = 4 * co.lo(); base - 4 * co.lo() < nA || base - 4 * co.lo() < nkS; base += 4 * Coop<NW>::STEP) {
            int src[4], ksrc[4];
            uint32_t v[4];
            KV kvv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                int e = base + 64 * u + lane;
                src[u] = e < nA ? m.T0[e] : -1;
                ksrc[u] = e < nkS ? srcK[e] : -1;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                v[u] = src[u] >= 0 ? bt.cigar[src[u]] : 0u;
                kvv[u].key = 0;
                kvv[u].val = 0;
                if (ksrc[u] >= 0) kvv[u] = ix.kv[ksrc[u]];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                int e = base + 64 * u + lane;
                if (e < nA) m.A[e] = v[u];
                if (e < nkS) {
                    m.K[e] = kvv[u].key;
                    m.V[e] = kvv[u].val;
                }
            }
        }
        co.sync();
    }

    PLO_T(1)
    // ---- LEFT SHIFT (left_shift_indels.rs:17-39 + cigar_indel_shifter.rs:10-165), items with do_shift ---------------
    if (!overflow && wv::ballot(has && do_shift) != 0ull) {
        // pass A: classes, heads, cluster sums, positions.  The list of cluster heads lives in the idle ping-pong array
        // (m.K / m.V hold the staged block maps).
        int nH = 0;
        int *const heads_list = (int *)m.B;
        {
            SegSumT<NW> sr(co), sq(co), sm(co);
            MaxScanT<NW> pnz(co, -1), heads(co, -1);
            AddScanT<NW> hcount(co);
            MultiScanT<NW, 3, 1> grp1(co, -1);  // NW > 1: the scans of a step that do not depend on each other share one exchange
            MultiScanT<NW, 1, 1> grp2(co, -1);
            PLO_CHUNKS(base, nA) {
                int e = base + lane;
                bool valid = e < nA;
                uint32_t c = valid ? m.A[e] : 0;
                int id = co.elem_id(m.idA, e, valid);
                int i_do = co.item((int)do_shift, id), i_s = co.item(sA, id), i_pos = co.item(pos1, id);
                int t = op_type(c), L = op_len(c);
                bool on = valid && i_do;
                bool isC = on && is_indel(t) && L > 0;       // cluster member      (:73-85: len > 0 only)
                bool isZ = on && is_indel(t) && L == 0;      // invisible to the builder
                bool isM = on && is_match(t);
                bool isO = on && !is_indel(t) && !isM;
                bool ihead = valid && e == i_s;
                int R, Q, PM, pp;
                if constexpr (NW > 1) {
                    const int xa[3] = {(on && ref_consuming(t)) ? L : 0, (on && read_consuming(t)) ? L : 0, isM ? L : 0};
                    const int xm[1] = {(valid && !isZ) ? e : -1};
                    int ra[3], rm[1];
                    grp1.incl(xa, xm, ra, rm);
                    R = ra[0] - xa[0];
                    Q = ra[1] - xa[1];
                    PM = ra[2] - xa[2];
                    pp = grp1.excl_of(0, rm[0]);
                } else {
                    R = sr.excl((on && ref_consuming(t)) ? L : 0, ihead);
                    Q = sq.excl((on && read_consuming(t)) ? L : 0, ihead);
                    PM = sm.excl(isM ? L : 0, ihead);
                    int pi = pnz.incl((valid && !isZ) ? e : -1);
                    pp = pnz.excl_of(pi);
                }
                bool prevC = false;
                if (on && pp >= i_s) {
                    uint32_t pc = m.A[pp];
                    prevC = is_indel(op_type(pc)) && op_len(pc) > 0;
                }
                bool chead = isC && !prevC;
                bool event = chead || isO;
                int hidx, hrank;
                if constexpr (NW > 1) {
                    const int xa[1] = {chead ? 1 : 0}, xm[1] = {chead ? e : -1};
                    int ra[1], rm[1];
                    grp2.incl(xa, xm, ra, rm);
                    hidx = rm[0];
                    hrank = ra[0] - xa[0];
                } else {
                    hidx = heads.incl(chead ? e : -1);
                    hrank = hcount.excl(chead ? 1 : 0);
                }
                if (chead) heads_list[hrank] = e;  // compact list of cluster heads (hrank < nA <= cap)
                if (valid) {
                    m.T0[e] = i_pos + R;  // indel_block_ref_start
                    m.T1[e] = Q;          // indel_block_read_start
                    m.T2[e] = PM;         // match bases of the item before this element
                    m.idA[e] = (uint8_t)(id | (chead ? 64 : 0) | (event ? 128 : 0));
                    if (isC) wv::atomic_add(t == OP_D ? &m.T3[hidx] : &m.T4[hidx], L);
                }
            }
            nH = NW > 1 ? grp2.ca[0] : hcount.carry;
        }
        overflow = co.any(overflow);
        co.sync();
        PLO_T(1)
        // pass H: one lane per indel cluster (all clusters of the tile at once): left breakend homology.  This is
        // the only part of the shift that touches the sequences, so the HBM round trips are paid once per tile.
        if (!overflow) PLO_CHUNKS(base, nH) {
            int hl = base + lane;
            bool valid = hl < nH;
            int e = valid ? heads_list[hl] : 0;
            int id = co.elem_id(m.idA, e, true);
            int i_flip = co.item((int)flip, id), i_slen = co.item(seq_len, id), i_rlen = co.item(shift_ref_len, id);
            unsigned long long i_soff = co.item(seq_off, id), i_ref = co.item(shift_ref, id);
            if (valid) {
                bool panic = false;
                int probes = 0;
                ReadSeq rd = item_read_seq<SP>(bt, i_soff, i_slen, i_flip);
                int h = left_homology((const uint8_t *)(uintptr_t)i_ref, i_rlen, m.T0[e], m.T3[e], rd, m.T1[e], m.T4[e], m.T2[e],
                                      panic, probes);
                algo_bytes += 2u * (unsigned)probes;
                if (panic) wv::atomic_or(&m.itp[id], 1);
                if (rd.miss) {
                    wv::atomic_or(&m.itp[id], 2);
                    h = 0;
                }
                m.T1[e] = h;  // the read start is not needed any more: the head's slot carries the homology to pass B
            }
        }
        co.sync();
        PLO_T(10)
        // pass B: homology, carried match run (min-plus scan), emission
        int nB = 0;
        {
            MaxScanT<NW> ev(co, -1);
            AddScanT<NW> emit(co);
            wv::MinPlus carryF = {0, IMAX, 0};
            PLO_CHUNKS(base, nA) {
                int e = base + lane;
                bool valid = e < nA;
                uint32_t c = valid ? m.A[e] : 0;
                int idf = valid ? m.idA[e] : 0;
                int id = idf & 63;
                bool chead = (idf & 64) != 0, event = (idf & 128) != 0;
                int i_do = co.item((int)do_shift, id), i_s = co.item(sA, id), i_c = co.item(cA, id);
                int t = op_type(c), L = op_len(c);
                bool on = valid && i_do;
                int li = ev.incl((on && event) ? e : -1);
                int pe = ev.excl_of(li);
                bool have_prev = pe >= i_s;
                int pm = valid ? m.T2[e] : 0;
                int m_e = pm - ((on && have_prev) ? m.T2[pe] : 0);
                wv::MinPlus f = {0, IMAX, 0};
                int del = 0, ins = 0, h = 0;
                if (on && chead) {
                    del = m.T3[e];
                    ins = m.T4[e];
                }
                if (on && chead) {
                    h = m.T1[e];  // left homology from pass H
                    f.a = m_e;
                    f.b = h;
                    f.s = have_prev ? 0 : 1;
                } else if (on && event) {  // add_other: flush, match run restarts at 0 (:155-165)
                    f.a = m_e;
                    f.b = 0;
                    f.s = have_prev ? 0 : 1;
                }
                wv::MinPlus F = wv::scan_minplus(f);
                wv::MinPlus Fpre = carryF, Fall = carryF;  // carried function before this wave's first lane / after the step
                if constexpr (NW > 1) co.exch_minplus(wv::bcast_last(F), carryF, Fpre, Fall);
                F = wv::mp_compose(Fpre, F);
                int r_after = wv::imin(F.a, F.b);
                int r_excl = wv::shfl_up1(r_after, wv::imin(Fpre.a, Fpre.b));  // match run carried out of the previous element
                if constexpr (NW > 1) carryF = Fall;
                else carryF = wv::bcast_last(F);
                int r_before = have_prev ? r_excl : 0;
                int x = r_before + m_e;
                // Up to four ops per element, as flags + values (no indexed local array: that would live in scratch):
                //   copy (item not shifted) | cluster head: M(x-cs) I(ins) D(del), "nImD" order of the left shift
                //   (:132-147) | other event: M(x) op | last element: final add_other(None) M(x_end) (:54-60)
                const bool cp = valid && !i_do, ch = on && chead, ev = on && !chead && event;
                const int cs = wv::imin(x, h);  // actual_shift_len (:132)
                const bool e0 = cp || (ch && x - cs > 0) || (ev && x > 0);
                const uint32_t v0 = cp ? c : mk_op(OP_M, ch ? x - cs : x);
                const bool e1 = (ch && ins > 0) || ev;
                const uint32_t v1 = ch ? mk_op(OP_I, ins) : c;
                const bool e2 = ch && del > 0;
                const uint32_t v2 = mk_op(OP_D, del);
                int x_end = 0;
                if (on && e == i_s + i_c - 1) {
                    bool have_le = li >= i_s;
                    int pm_incl = pm + (is_match(t) ? L : 0);
                    int m_end = pm_incl - (have_le ? m.T2[li] : 0);
                    x_end = (have_le ? r_after : 0) + m_end;
                }
                const bool e3 = x_end > 0;
                const uint32_t v3 = mk_op(OP_M, x_end);
                const int ne = (int)e0 + (int)e1 + (int)e2 + (int)e3;
                int ei = emit.incl(ne);
                int p = ei - ne;
                if (p + ne > m.cap) {
                    overflow = true;
                } else {
                    put_op(m.B, m.idB, p, e0, v0, id);
                    put_op(m.B, m.idB, p, e1, v1, id);
                    put_op(m.B, m.idB, p, e2, v2, id);
                    put_op(m.B, m.idB, p, e3, v3, id);
                }
                if (valid) m.T0[e] = ei;
            }
            nB = emit.carry;
        }
        overflow = co.any(overflow);
        co.sync();
        PLO_T(11)
        if (!overflow) {
            int sB, cB;
            finish_counts(m.T0, sA, cA, sB, cB);
            co.sync();
            int shift = 0, nOut = 0;
            bool pairs_;
            cleanup_compress(co, m, m.B, m.idB, m.A, m.idA, nB, sB, cB, has && do_shift, shift, nOut, pairs_);
            sA = sB;
            cA = cB;
            nA = nOut;
            pos1 += shift;
            if (has && m.itp[lane]) {  // absent bases (sparse batches) come first: what the probes saw then is not the read
                status = (m.itp[lane] & 2) ? PLO_ITEM_NEED_BASES : PLO_ITEM_PANIC;
                alive = false;
            }
        }
    }

    PLO_T(2)
    // the lifted CIGARs of the tile have an indel cluster of more than one op (known after the liftover's clean-up)
    bool lifted_pairs = true;
    // ---- LIFTOVER (src/liftover_read_alignment.rs:35-223) ------------------------------------------------------------------
    if (!overflow && (stages & PLO_STAGE_LIFTOVER)) {
        // per item: the window [W0, W1) of the block map that can intersect the item was located by build_item_desc;
        // stage those entries (plus the one after, whose key bounds the last block) in LDS once per tile
        if (alive) {
            int nb = kv1 - kv0, lg = 0;
            while ((1 << lg) < nb) ++lg;
            if (co.lead()) algo_bytes += 16u * (unsigned)(W1 - W0) + 8u * (unsigned)lg;
        } else {
            W1 = W0;
        }
        // pass A: per op, first block f and number of (op x block) pieces.  Passes A and B are instantiated once for block
        // maps staged in LDS and once for maps read from global memory (a run-time select between the two pointers would
        // turn every probe into a flat load behind a branch).
        int P = 0;
        if constexpr (NW == 1) {  // piece range [itq, itc) of every item, published by pass A (items without ops: empty)
            m.itq[lane] = 0;
            m.itc[lane] = 0;
            co.sync();
        }
        auto pass_a = [&](auto staged_c) {
            constexpr bool STAGED = decltype(staged_c)::value;
            SegSumT<NW> sr(co);
            AddScanT<NW> pieces(co);
            PLO_CHUNKS(base, nA) {
                int e = base + lane;
                bool valid = e < nA;
                uint32_t c = valid ? m.A[e] : 0;
                int id = co.elem_id(m.idA, e, valid);
                int i_alive = co.item((int)alive, id), i_s = co.item(sA, id), i_pos = co.item(pos1, id);
                int i_c = 0;
                if constexpr (NW == 1) i_c = co.item(cA, id);
                int i_w0 = co.item(W0, id), i_w1 = co.item(W1, id), i_kb = co.item(kb, id);
                int t = op_type(c), L = op_len(c);
                bool on = valid && i_alive;
                bool rc = ref_consuming(t);
                int s = i_pos + sr.excl((on && rc) ? L : 0, valid && e == i_s);
                int cnt = 0, f = 0;
                if (on) {
                    if (rc) {
                        if (L > 0) {
                            // window-relative block indices: f = greatest key <= s (or -1: no block, only possible when
                            // the window starts at the map's first entry), l = greatest key < s+L
                            int nw = i_w1 - i_w0, lo = 0, hi = nw;
                            while (lo < hi) {  // upper bound of s
                                int mid = (lo + hi) >> 1;
                                int key = STAGED ? m.K[i_kb + mid] : ix.kv[i_w0 + mid].key;
                                if (key <= s) lo = mid + 1; else hi = mid;
                            }
                            f = lo - 1;
                            hi = nw;  // lower bound of s+L (>= the upper bound of s)
                            while (lo < hi) {
                                int mid = (lo + hi) >> 1;
                                int key = STAGED ? m.K[i_kb + mid] : ix.kv[i_w0 + mid].key;
                                if (key < s + L) lo = mid + 1; else hi = mid;
                            }
                            cnt = (lo - 1) - f + 1;
                        }
                    } else if (t == OP_I || t == OP_S || t == OP_H) {
                        cnt = 1;  // :157-160 copied through; Pad (:213) emits nothing
                    }
                }
                int b = pieces.excl(cnt);
                if (valid) {
                    m.T0[e] = s;
                    m.T1[e] = f;
                    m.T2[e] = cnt > 0 ? b : -1;
                    if constexpr (NW == 1) {
                        if (e == i_s) m.itq[id] = b;
                        if (e == i_s + i_c - 1) m.itc[id] = b + cnt;
                    }
                }
            }
            P = pieces.carry;
        };
        if (staged) pass_a(std::true_type{});
        else pass_a(std::false_type{});
        PLO_T(3)
        if (P > m.cap) overflow = true;
        if (!overflow) {
            PLO_CHUNKS(base, P) {
                int j = base + lane;
                if (j < P) m.T3[j] = 0;
            }
            int ps_t = 0, pe_t = 0;  // NW == 1: the item's pieces are [ps_t, pe_t)
            if constexpr (NW == 1) {
                co.sync();
                ps_t = m.itq[lane];
                pe_t = m.itc[lane];
            }
            m.its[lane] = NONE32;  // ref2_start_pos = None
            if constexpr (NW > 1) m.itc[lane] = 0;
            m.itf[lane] = IMAX;    // first / last match op of the item's lifted CIGAR, published by pass B for the clean-up
            m.itl[lane] = -1;
            co.sync();
            PLO_CHUNKS(base, nA) {
                int e = base + lane;
                if (e < nA && m.T2[e] >= 0) m.T3[m.T2[e]] = e + 1;
            }
            co.sync();
            // pass B: one lane per piece (update_ref2_cigar_segment, :35-133)
            int nB = 0;
            auto pass_b = [&](auto staged_c) {
                constexpr bool STAGED = decltype(staged_c)::value;
                MaxScanT<NW> owner(co, 0), lastmap(co, -1), lastfm(co, -1);
                AddScanT<NW> emit(co);
                MultiScanT<NW, 0, 2> grp(co, NONE32);
                int last_fm = -1;  // NW > 1: the lane's last match piece position (one reduction instead of an atomic per piece)
                PLO_CHUNKS(base, P) {
                    int j = base + lane;
                    bool valid = j < P;
                    int i = owner.incl(valid ? m.T3[j] : 0) - 1;
                    if (i < 0) i = 0;
                    uint32_t c = valid ? m.A[i] : 0;
                    int id = co.elem_id(m.idA, i, valid);
                    int i_w0 = co.item(W0, id), i_kv1 = co.item(kv1, id), i_kb = co.item(kb, id);
                    int t = op_type(c), L = op_len(c);
                    bool piece = valid && ref_consuming(t);
                    bool ism = is_match(t);
                    bool mapped = false, before = false, fm = false;
                    int plen = 0, val = NONE32, endval = 0, startval = 0;
                    if (piece) {
                        int s = m.T0[i], f = m.T1[i], tt = j - m.T2[i];
                        int b = f + tt;  // window-relative block index, -1 = before the first block of the map
                        before = b < 0;
                        int bkey = 0;
                        if (!before) {
                            if (STAGED) {
                                bkey = m.K[i_kb + b];
                                val = m.V[i_kb + b];
                            } else {
                                KV kvb = ix.kv[i_w0 + b];
                                bkey = kvb.key;
                                val = kvb.val;
                            }
                        }
                        int pstart = (tt == 0) ? s : bkey;
                        int pend = s + L;
                        if (i_w0 + b + 1 < i_kv1) {
                            int kn = STAGED ? m.K[i_kb + b + 1] : ix.kv[i_w0 + b + 1].key;
                            if (kn < pend) pend = kn;
                        }
                        plen = pend - pstart;
                        mapped = !before && val != NONE32;
                        fm = mapped && ism;
                        if (mapped) {
                            endval = val + (pend - bkey);      // :98-100
                            startval = val + (pstart - bkey);  // :84-88
                        }
                    }
                    // at most two ops per piece, as flags + values (no indexed local array: that would live in scratch)
                    bool e0 = false, started = false, started_before = false;  // e0: the jump deletion of :91-96
                    uint32_t v0 = 0;
                    int i_ps = 0, i_pe = 0, fi_last = -1;
                    if constexpr (NW > 1) {
                        // One item: "the previous mapped piece" needs no owner test, and its end (ref2_end_pos) is the running
                        // maximum of the ends -- reference positions never decrease along a block map built from a CIGAR -- so
                        // it comes out of the scan itself instead of an LDS hand-off between the waves.
                        const int xm[2] = {mapped ? endval : NONE32, fm ? j : NONE32};
                        int rm[2];
                        grp.incl(nullptr, xm, nullptr, rm);
                        const int prev_end = grp.excl_of(0, rm[0]);
                        const int fi = rm[1], fe = grp.excl_of(1, rm[1]);
                        if (mapped) {
                            started = fi >= 0;
                            started_before = fe >= 0;
                            if (fm && !started_before) m.its[id] = startval;
                            if (prev_end != NONE32) {  // :91-96
                                int d = val - prev_end;
                                e0 = d > 0 && started;
                                v0 = mk_op(OP_D, d);
                            }
                        }
                    } else {
                    // The pieces of an item are consecutive ([i_ps, i_pe), published by pass A): "the previous mapped piece /
                    // the first mapped match piece belongs to my item" is a comparison with i_ps, no look-up of that piece.
                    i_ps = co.item(ps_t, id);
                    i_pe = co.item(pe_t, id);
                    if (valid) m.T4[j] = endval;
                    co.sync();
                    int mi = lastmap.incl(mapped ? j : -1);
                    int pmap = lastmap.excl_of(mi);
                    fi_last = lastfm.incl(fm ? j : -1);
                    int fe = lastfm.excl_of(fi_last);
                    if (mapped) {
                        started = fi_last >= i_ps;  // ref2_start_pos.is_some(), after :84-88
                        started_before = fe >= i_ps;
                        if (fm && !started_before) m.its[id] = startval;
                        if (pmap >= i_ps) {  // :91-96
                            int d = val - m.T4[pmap];
                            e0 = d > 0 && started;
                            v0 = mk_op(OP_D, d);
                        }
                    }
                    }
                    // copied op | mapped piece (:102-109) | insertion over an unmapped block (:111-115) | soft clip before the
                    // first block (:117-123)
                    const bool e1 = (valid && !piece) || (piece && (mapped ? (ism || started) : ism));
                    const uint32_t v1 = !piece ? c
                                        : mk_op(mapped ? (t == OP_D ? OP_D : (t == OP_N ? OP_N : OP_M)) : (before ? OP_S : OP_I), plen);
                    const int ne = (int)e0 + (int)e1;
                    int ei = emit.incl(ne);
                    int p = ei - ne;
                    if (p + ne > m.cap) {
                        overflow = true;
                    } else {
                        if (fm) {  // the only match ops of the output are the mapped match pieces: position p (+1 after a jump D)
                            int pm = p + (e0 ? 1 : 0);
                            if (!started_before) m.itf[id] = pm;  // single writer, like m.its
                            if constexpr (NW > 1) last_fm = pm;  // positions grow with j
                        }
                        put_op(m.B, m.idB, p, e0, v0, id);
                        put_op(m.B, m.idB, p, e1, v1, id);
                    }
                    if constexpr (NW == 1) {
                        // no per-item atomics: the inclusive emission prefix of every piece stays in T3 (the owner mark of the
                        // piece has been consumed above); item op counts and the last match op are differences / look-ups of it
                        // after the pass.  The item's last piece leaves the index of its last mapped match piece.
                        if (valid) m.T3[j] = ei;
                        if (valid && j == i_pe - 1) m.itl[id] = fi_last >= i_ps ? fi_last : -1;
                    }
                }
                nB = emit.carry;
                if constexpr (NW > 1) {  // one item: its op count is the total, its last match piece one maximum per wave
                    int lm = wv::reduce_max(last_fm);
                    if (lane == 0) {
                        if (lm >= 0) wv::atomic_max(&m.itl[0], lm);
                        m.itc[0] = nB;
                    }
                }
            };
            if (staged) pass_b(std::true_type{});
            else pass_b(std::false_type{});
            overflow = co.any(overflow);
            co.sync();
            PLO_T(4)
            if (!overflow) {
                int cB;
                if constexpr (NW == 1) {
                    cB = pe_t > ps_t ? m.T3[pe_t - 1] - (ps_t > 0 ? m.T3[ps_t - 1] : 0) : 0;
                    const int lf = m.itl[lane];  // piece index of the item's last mapped match piece: its match op is the last op it emitted
                    const int lpos = lf >= 0 ? m.T3[lf] - 1 : -1;
                    co.sync();
                    m.itl[lane] = lpos;
                } else {
                    cB = m.itc[lane];
                }
                int incB = wv::scan_add(cB);
                int sB = incB - cB;
                int r2s = m.its[lane];
                co.sync();
                if (alive && r2s == NONE32) {  // :218 ref2_start_pos.map(...) on None
                    status = PLO_ITEM_NO_LIFTOVER;
                    alive = false;
                }
                int shift = 0, nOut = 0;
                cleanup_compress(co, m, m.B, m.idB, m.A, m.idA, nB, sB, cB, alive, shift, nOut, lifted_pairs, /*edges_known=*/true);  // :219-220
                sA = sB;
                cA = cB;
                nA = nOut;
                pos1 = r2s + shift;  // :221
            }
        }
    }

    PLO_T(5)
    // ---- LENGTH CHECK (src/read_alignment_scanner.rs:204-229) ------------------------------------------------------------------
    bool simp = alive;
    if (!overflow && (stages & PLO_STAGE_LENCHECK)) {
        // The read bases consumed by the CIGAR at this point equal those of the input CIGAR (build_item_desc explains why the
        // shift and the liftover keep that number), which the enumerate pass has summed: no pass over the ops here.
        if (alive && len_bad) {
            status = PLO_ITEM_LEN_MISMATCH;
            simp = false;
        }
    }

    PLO_T(6)
    // ---- SIMPLIFY (src/simplify_alignment_indels.rs:5-156) ------------------------------------------------------------------
    // On a CIGAR that the liftover stage has just cleaned and compressed (no zero-length ops, both steps idempotent),
    // simplify_alignment_indels is the identity unless some cluster has more than one op: end_indel re-emits a single
    // I or D unchanged (:41-44) and :153-155 change nothing.  Such tiles skip the stage.
    if (!overflow && (stages & PLO_STAGE_SIMPLIFY) && !((stages & PLO_STAGE_LIFTOVER) && !lifted_pairs)) {
        PLO_CHUNKS(base, nA) {  // cluster sums start from zero
            int e = base + lane;
            if (e < nA) {
                m.T3[e] = 0;
                m.T4[e] = 0;
            }
        }
        co.sync();
        // pass A: clusters = maximal runs of I/D ops; sums at the cluster head; compact list of cluster heads
        int nH = 0;
        int *const heads_list = (int *)m.B;
        bool changes = false;  // some cluster is not a single I/D op of non-zero length
        {
            SegSumT<NW> sr(co), sq(co);
            MaxScanT<NW> heads(co, -1);
            AddScanT<NW> hcount(co);
            int carry_c = 0;
            MultiScanT<NW, 2, 1> grp1(co, -1);
            MultiScanT<NW, 1, 1> grp2(co, -1);
            PLO_CHUNKS(base, nA) {
                int e = base + lane;
                bool valid = e < nA;
                uint32_t c = valid ? m.A[e] : 0;
                int id = co.elem_id(m.idA, e, valid);
                int i_on = co.item((int)simp, id), i_s = co.item(sA, id), i_pos = co.item(pos1, id);
                int t = op_type(c), L = op_len(c);
                bool on = valid && i_on;
                bool isC = on && is_indel(t);
                bool ihead = valid && e == i_s;
                int R, Q, prevC;
                if constexpr (NW > 1) {
                    // "the previous element is a cluster member" as a max-scan (index of the last member before this element),
                    // so that it shares the exchange of the two position sums
                    const int xa[2] = {(on && ref_consuming(t)) ? L : 0, (on && read_consuming(t)) ? L : 0};
                    const int xm[1] = {isC ? e : -1};
                    int ra[2], rm[1];
                    grp1.incl(xa, xm, ra, rm);
                    R = ra[0] - xa[0];
                    Q = ra[1] - xa[1];
                    const int lastC = grp1.excl_of(0, rm[0]);  // (a cross-lane read: outside the short-circuit below)
                    prevC = (e > 0 && lastC == e - 1) ? 1 : 0;
                } else {
                    R = sr.excl((on && ref_consuming(t)) ? L : 0, ihead);
                    Q = sq.excl((on && read_consuming(t)) ? L : 0, ihead);
                    prevC = co.shift_up1((int)isC, carry_c);
                }
                bool chead = isC && !(prevC && !ihead);
                changes |= isC && (!chead || L == 0);
                int hidx, hrank;
                if constexpr (NW > 1) {
                    const int xa[1] = {chead ? 1 : 0}, xm[1] = {chead ? e : -1};
                    int ra[1], rm[1];
                    grp2.incl(xa, xm, ra, rm);
                    hidx = rm[0];
                    hrank = ra[0] - xa[0];
                } else {
                    hidx = heads.incl(chead ? e : -1);
                    hrank = hcount.excl(chead ? 1 : 0);
                }
                if (chead) heads_list[hrank] = e;
                if (valid) {
                    m.T0[e] = i_pos + R;  // block_ref_start
                    m.T1[e] = Q;          // block_read_start
                    m.idA[e] = (uint8_t)(id | (chead ? 64 : 0));
                    if (isC) wv::atomic_add(t == OP_D ? &m.T3[hidx] : &m.T4[hidx], L);
                }
            }
            nH = NW > 1 ? grp2.ca[0] : hcount.carry;
        }
        overflow = co.any(overflow);
        co.sync();
        // (same test on the clusters themselves: complex clusters of items that take no part do not count)
        const bool identity = (stages & PLO_STAGE_LIFTOVER) && !co.any(changes);
        if (!identity) {
        // The items' reference chromosome and read bases are fetched here, not with the other descriptor fields: few tiles get
        // this far (the lifted CIGARs of most have no multi-op cluster), and the tile kernel is short of registers.
        unsigned long long chrom_ref = 0, seq_off_s = 0;
        int chrom_ref_len = 0, seq_len_s = 0;
        if (has && !overflow) {
            chrom_ref = wk.d.chrom_ref[g];
            chrom_ref_len = wk.d.chrom_ref_len[g];
            seq_off_s = wk.d.seq_off[g];
            seq_len_s = (int)wk.d.seq_len[g];
        }
        // pass H: one lane per cluster; only complex clusters (both I and D, not 1/1) look at the sequences
        // (CigarBlockInfo::end_indel :49-105).  Results overwrite the head's slots: T0 pre, T1 post, T3 del, T4 ins.
        if (!overflow) PLO_CHUNKS(base, nH) {
            int hl = base + lane;
            bool valid = hl < nH;
            int e = valid ? heads_list[hl] : 0;
            int id = co.elem_id(m.idA, e, true);
            int i_flip = co.item((int)flip, id), i_slen = co.item(seq_len_s, id), i_rlen = co.item(chrom_ref_len, id);
            unsigned long long i_soff = co.item(seq_off_s, id), i_ref = co.item(chrom_ref, id);
            if (valid) {
                int del = m.T3[e], ins = m.T4[e];
                int complex_done = 0;
                if (del > 0 && ins > 0 && !(del == 1 && ins == 1)) {
                    int rs0 = m.T0[e], qs0 = m.T1[e];
                    ReadSeq rd = item_read_seq<SP>(bt, i_soff, i_slen, i_flip);
                    const uint8_t *ref = (const uint8_t *)(uintptr_t)i_ref;
                    if (rs0 < 0 || rs0 + del - 1 >= i_rlen || qs0 + ins - 1 >= rd.len) {
                        wv::atomic_or(&m.itp[id], 1);  // slice index out of bounds: the reference panics (:58-60)
                    } else {
                        int pre = 0, post = 0, cmp = 0;
                        // :55-68 trailing bases shared by the inserted and the deleted sequence, then :71-85 leading ones
                        post = match_run_back(ref, i_rlen, rs0 + del, rd, qs0 + ins, wv::imin(del, ins), cmp);
                        del -= post;
                        ins -= post;
                        pre = match_run_fwd(ref, i_rlen, rs0, rd, qs0, wv::imin(del, ins), cmp);
                        del -= pre;
                        ins -= pre;
                        if (del == 1 && ins == 1) {  // :88-92
                            del = 0;
                            ins = 0;
                            ++post;
                        }
                        algo_bytes += 2u * (unsigned)cmp;
                        if (rd.miss) wv::atomic_or(&m.itp[id], 2);
                        m.T0[e] = pre;
                        m.T1[e] = post;
                        m.T3[e] = del;
                        m.T4[e] = ins;
                        complex_done = 1;
                    }
                }
                m.T2[e] = complex_done;
            }
        }
        co.sync();
        // pass B: emission
        int nB = 0;
        {
            AddScanT<NW> emit(co);
            PLO_CHUNKS(base, nA) {
                int e = base + lane;
                bool valid = e < nA;
                uint32_t c = valid ? m.A[e] : 0;
                int idf = valid ? m.idA[e] : 0;
                int id = idf & 63;
                bool chead = (idf & 64) != 0;
                int i_on = co.item((int)simp, id);
                int t = op_type(c);
                bool on = valid && i_on;
                // up to four ops per element, as flags + values (no indexed local array: that would live in scratch)
                const bool cp = valid && !(on && is_indel(t));  // :144-147 everything outside a cluster is copied
                const bool ch = on && is_indel(t) && chead;
                int del = 0, ins = 0, pre = 0, post = 0, was_complex = 0;
                if (ch) {
                    del = m.T3[e];
                    ins = m.T4[e];
                    was_complex = m.T2[e];
                    if (was_complex) {
                        pre = m.T0[e];
                        post = m.T1[e];
                    }
                }
                // complex cluster: M(pre) I(ins) D(del) M(post) (:101-104); simple: I | D | 1I1D -> M(1) (:41-48); a complex
                // cluster that would have panicked has both sizes > 0 and emits nothing (the item is reported PANIC)
                const bool one_one = ch && !was_complex && del == 1 && ins == 1;
                const bool both = del > 0 && ins > 0;
                const bool e0 = cp || (ch && (was_complex ? pre > 0 : one_one));
                const uint32_t v0 = cp ? c : mk_op(OP_M, was_complex ? pre : 1);
                const bool e1 = ch && ins > 0 && (was_complex || !both);
                const uint32_t v1 = mk_op(OP_I, ins);
                const bool e2 = ch && del > 0 && (was_complex || !both);
                const uint32_t v2 = mk_op(OP_D, del);
                const bool e3 = ch && was_complex && post > 0;
                const uint32_t v3 = mk_op(OP_M, post);
                const int ne = (int)e0 + (int)e1 + (int)e2 + (int)e3;
                int ei = emit.incl(ne);
                int p = ei - ne;
                if (p + ne > m.cap) {
                    overflow = true;
                } else {
                    put_op(m.B, m.idB, p, e0, v0, id);
                    put_op(m.B, m.idB, p, e1, v1, id);
                    put_op(m.B, m.idB, p, e2, v2, id);
                    put_op(m.B, m.idB, p, e3, v3, id);
                }
                if (valid) m.T2[e] = ei;
            }
            nB = emit.carry;
        }
        overflow = co.any(overflow);
        co.sync();
        PLO_T(7)
        if (!overflow) {
            int sB, cB;
            finish_counts(m.T2, sA, cA, sB, cB);
            co.sync();
            int shift = 0, nOut = 0;
            bool pairs_;
            cleanup_compress(co, m, m.B, m.idB, m.A, m.idA, nB, sB, cB, simp, shift, nOut, pairs_);  // :153-154
            sA = sB;
            cA = cB;
            nA = nOut;
            if (simp) pos1 += shift;  // :155
            if (has && m.itp[lane]) {  // absent bases (sparse batches) come first: what the probes saw then is not the read
                status = (m.itp[lane] & 2) ? PLO_ITEM_NEED_BASES : PLO_ITEM_PANIC;
                alive = false;
            }
        }
        }  // !identity
    }

    PLO_T(8)
    // ---- OUTPUT --------------------------------------------------------------------------------------------------------------
    overflow = co.any(overflow);
    if (overflow) {
        if (last_resort) {
            if (has) {
                wk.status[g] = PLO_ITEM_PANIC;
                wk.cig_len[g] = 0;
                wk.cig_off[g] = 0;
                wk.pos[g] = -1;
            }
            if (lane == 0) wv::atomic_add_global(&wk.counters[CNT_ERROR], 1ull);
        } else if (co.lead()) {
            // capacity exceeded: re-queue every item -- of a shared tile on the retry list (few items per wave, larger
            // capacity), of a retry group on the workgroup-per-item list, of a workgroup on the global-scratch list
            const bool to_next = level != LEVEL_TILE;
            unsigned long long hm = wv::ballot(has);
            int slot = 0;
            if (lane == 0)
                slot = (int)wv::atomic_add_global(&wk.counters[to_next ? next_cnt : CNT_NRETRY], (unsigned long long)__builtin_popcountll(hm));
            slot = wv::bcast_first(slot);
            if (has) {
                int rank = __builtin_popcountll(hm & ((1ull << lane) - 1ull));
                (to_next ? next_list : wk.retry_list)[slot + rank] = g;
                wk.status[g] = (uint8_t)ITEM_NEED_BIG;
            }
        }
        return;
    }
    bool emit_cigar = has && (status == PLO_ITEM_LIFTED || status == PLO_ITEM_LEN_MISMATCH);
    int oc = emit_cigar ? cA : 0;
    int inco = wv::scan_add(oc);
    int oS = inco - oc;
    int total = wv::bcast_last(inco);
    unsigned long long gbase = 0;
    if (co.lead()) {  // the slab state of a workgroup is the lead wave's
        if ((unsigned long long)total > ctx.slab_left) {  // wave-uniform: reserve a new slab
            unsigned long long want = (unsigned long long)total > SLAB_OPS ? (unsigned long long)total : SLAB_OPS;
            unsigned long long nb = 0;
            if (lane == 0) nb = wv::atomic_add_global(&wk.counters[CNT_CIGAR], want) + wk.slab_offset;
            ctx.slab_base = wv::bcast_first(nb);
            ctx.slab_left = want;
        }
        gbase = ctx.slab_base;
        ctx.slab_base += (unsigned long long)total;
        ctx.slab_left -= (unsigned long long)total;
    }
    gbase = co.from_lead(gbase);
    bool fits = gbase + (unsigned long long)total <= wk.out_cap;
    if (!fits && lane == 0 && co.lead()) wv::atomic_add_global(&wk.counters[CNT_OVERFLOW], 1ull);
    PLO_CHUNKS(base, nA) {
        int e = base + lane;
        bool valid = e < nA;
        int id = co.elem_id(m.idA, e, valid);
        int i_em = co.item((int)emit_cigar, id), i_s = co.item(sA, id), i_o = co.item(oS, id);
        if (valid && i_em && fits) wk.out_cigar[gbase + (unsigned long long)(i_o + (e - i_s))] = m.A[e];
    }
    if (has && co.lead()) {
        if (status == PLO_ITEM_NEED_BASES) wk.miss_list[wv::atomic_add_global(&wk.counters[CNT_NMISS], 1ull)] = g;  // rare
        wk.status[g] = (uint8_t)status;
        wk.pos[g] = emit_cigar ? (int64_t)pos1 : (int64_t)-1;
        wk.cig_off[g] = emit_cigar ? gbase + (unsigned long long)oS : 0ull;
        wk.cig_len[g] = (uint32_t)oc;
        algo_bytes += 40u + 4u * (unsigned)n_in + 24u + 4u * (unsigned)oc;
    }
    // statistics stay in registers until the wave retires (wave_ctx_flush)
    ctx.algo_bytes += algo_bytes;
    ctx.in_ops += (has && co.lead()) ? (unsigned)n_in : 0u;
    ctx.out_ops += co.lead() ? (unsigned)oc : 0u;
    PLO_T(9)
#undef PLO_T
}

// -------------------------------------------------------------------------------------------------------------------
// Tile assignment (k_tile_bounds): the tiled items, in class order, form one flattened stream of item weights that is cut
// into windows of `window`; every item goes to the window its first unit falls in.  A window with more than 64 items is
// processed in several passes.
//
// Persistent wave: tiles first, first + stride, ...  The dependent chain tile bounds -> item list -> descriptors -> gather
// would cost four memory round trips per tile; the first three are software-pipelined across tiles (bounds three tiles
// ahead, item indices two, descriptor fields one), so that only the gather itself is waited for.
// -------------------------------------------------------------------------------------------------------------------
template <bool SP = false>
PLO_DEV void lift_tiles_persistent(const DevIndex &ix, const DevBatch &bt, const DevWork &wk, uint32_t stages, uint32_t first,
                                   uint32_t stride, uint32_t n_tiles, int big_thresh, TileMem m, WaveCtx &ctx) {
    const uint32_t lane = (uint32_t)wv::lane();
    Coop<1> co;
    auto bounds = [&](uint32_t t, uint32_t &lo, uint32_t &hi) {  // per lane, same value in every lane
        lo = 0;
        hi = 0;
        if (t < n_tiles) {
            lo = wk.tile_lo[t];  // written by k_tile_bounds
            hi = wk.tile_lo[t + 1];
        }
    };
    auto load_g = [&](uint32_t lo, uint32_t hi) -> uint32_t { return (lo + lane < hi) ? wk.perm[lo + lane] : 0u; };
    auto load_desc = [&](uint32_t lo, uint32_t hi, uint32_t g, TilePre &p) {
        p.g = g;
        if (lo + lane < hi) {
            p.in_off = wk.d.in_off[g];
            p.n_in = wk.d.n_in[g];
            p.w0 = wk.d.w0[g];
            p.w1 = wk.d.w1[g];
            p.kv1 = wk.d.kv1[g];
            p.fl = wk.d.flags[g];
        }
    };
    uint32_t t = first;
    uint32_t lo0, hi0, lo1, hi1, lo2, hi2;
    bounds(t, lo0, hi0);
    bounds(t + stride, lo1, hi1);
    bounds(t + 2 * stride, lo2, hi2);
    lo1 = (uint32_t)wv::bcast_first((int)lo1);
    hi1 = (uint32_t)wv::bcast_first((int)hi1);
    lo2 = (uint32_t)wv::bcast_first((int)lo2);
    hi2 = (uint32_t)wv::bcast_first((int)hi2);
    uint32_t g1 = load_g(lo1, hi1);
    TilePre d0;
    load_desc(lo0, hi0, load_g(lo0, hi0), d0);
    for (; t < n_tiles; t += stride) {
        // next stages of the pipeline: issued now, consumed one iteration later
        TilePre d1;
        load_desc(lo1, hi1, g1, d1);
        uint32_t g2 = load_g(lo2, hi2);
        uint32_t lo3, hi3;
        bounds(t + 3 * stride, lo3, hi3);
        const uint32_t lo = (uint32_t)wv::bcast_first((int)lo0), hi = (uint32_t)wv::bcast_first((int)hi0);
        for (uint32_t b = lo; b < hi; b += 64) {
            int nit = (int)((hi - b) < 64u ? (hi - b) : 64u);
            lift_tile<1, SP>(co, ix, bt, wk, stages, b, nit, m, wk.perm, LEVEL_TILE, big_thresh, ctx, b == lo ? &d0 : nullptr);
            wv::sync();
        }
        lo0 = lo1;
        hi0 = hi1;
        lo1 = lo2;
        hi1 = hi2;
        lo2 = (uint32_t)wv::bcast_first((int)lo3);  // (the same value in every lane: kept as scalars from here on)
        hi2 = (uint32_t)wv::bcast_first((int)hi3);
        g1 = g2;
        d0 = d1;
    }
}

}  // namespace plo
