// lane_stream.hpp -- heavy items, STREAMED: the lane-per-item stages of lane_core.hpp chained through LDS rings.
//
// k_lift_lanes_g (lane_core.hpp, WIN) runs shift -> liftover -> simplify one after the other over a lane's whole CIGAR: every stage
// writes its CIGAR to a region in global memory and the next one reads it back (5.5 x the algorithmic traffic on the stress profile),
// every window refill is a global round trip for all 64 lanes, and a wave runs for as long as its longest item.  Here the stages are
// STEPS of one loop and a lane's ops flow through four small rings in LDS:
//
//      batch CIGAR --R--> [IN] --A: left shift--> [Q1] --B: liftover--> [Q2] --C: simplify--> [Q3] --F--> output CIGAR
//                            \________________(forward-mapped contig segments: no A)_____/
//
//   * a stage's writer (clean_up_cigar_edge_indels + compress_cigar as a streaming writer, LaneOut of lane_core.hpp) RELEASES an op to
//     the next stage only when it is final: the trailing-edge rule of clean_up_cigar_edge_indels (cigar/mod.rs:265-291) rewrites the
//     ops behind the LAST alignment match, so ops are released up to the last match written and the tail behind it stays in the ring
//     until another match follows or the stage ends (then the rule is applied to it in the ring).  A consumer starts after the first
//     release, i.e. when its producer's leading-edge position shift (:277-283) is final;
//   * every trip of the loop runs ONE step for the lanes that are ready for it -- the step most lanes are ready for (ballots and
//     population counts: scalar work) -- so the expensive bodies execute with most lanes active whatever the lanes' positions;
//   * a lane that has finished its item takes the wave's next one (S) while the others carry on: a wave's time is the sum of its
//     items' ops / 64, not groups x longest item;
//   * the only global traffic is the input CIGAR (R: 8 ops per lane and refill), the probes, the output (F: 8 ops per flush, straight
//     into the item's slot of the output buffer, allocated up front from its region bound) and the per-item descriptors / results.
// An item whose unreleased tail outgrows a ring (no step can make progress for the lane) or whose output outgrows its slot goes to
// the retry list -> wave-cooperative code, like every item a lane kernel cannot hold.
//
// The stage code is lane_core.hpp's, statement for statement (same reference lines), with ring reads / writes in place of region
// indices.  Stage set: STRAND | LSHIFT | LIFTOVER | LENCHECK | SIMPLIFY (PLO_STAGES_ALL) only; subsets take k_lift_lanes_g.
// All citations are relative to /root/reference.
#pragma once
#include "lane_core.hpp"

namespace plo {

// Ring of N ops per lane (N a power of two): op e lives in b[(e & (N - 1)) * 64] (b: the lane's word of element 0; bank = lane).
// Writer: lane_push()'s logic (leading edge, zero-length filter, run merging) with `rel`: ops [0, rel) are final.
template <int N>
struct LaneRing {
    uint32_t *b = nullptr;
    int no = 0;        // ops written
    int rel = 0;       // ops released to the consumer
    int rk = 0;        // ops consumed
    uint32_t acc = 0;  // the open run (starts as Match(0), cigar/mod.rs:206)
    int lead_shift = 0;
    bool seen_m = false, pairs = false, ovf = false;
};
template <int N>
PLO_DEV int ring_room(const LaneRing<N> &r) { return N - (r.no - r.rk); }
template <int N>
PLO_DEV void ring_reset(LaneRing<N> &r, bool on) {
    r.no = on ? 0 : r.no;
    r.rel = on ? 0 : r.rel;
    r.rk = on ? 0 : r.rk;
    r.acc = on ? 0u : r.acc;
    r.lead_shift = on ? 0 : r.lead_shift;
    r.seen_m = r.seen_m & !on;
    r.pairs = r.pairs & !on;
    r.ovf = r.ovf & !on;
}
// lane_push (lane_core.hpp) into a ring: straight-line; a flushed alignment-match op releases everything up to and including itself
template <bool PAD, int N>
PLO_DEV void ring_push(LaneRing<N> &o, bool on, int t, int L) {
    const bool lead = on & !o.seen_m;
    const bool drop_d = lead & (t == OP_D);
    o.lead_shift += drop_d ? L : 0;
    t = (lead & (t == OP_I)) ? (int)OP_S : t;
    o.seen_m = o.seen_m | (on & b_is_match(t));
    const bool live = on & !drop_d & (L > 0);
    const int at = (int)(o.acc & 15u);
    const bool same = live & (t == at);
    const bool flush = live & !same & (o.acc >= 16u);
    const bool ok = (o.no - o.rk) < N;
    if (flush & ok) o.b[(o.no & (N - 1)) * 64] = o.acc;
    o.ovf = o.ovf | (flush & !ok);
    o.pairs = o.pairs | (flush & b_is_indel(at) & b_is_indel(t));
    o.no += flush ? 1 : 0;
    o.rel = (flush & b_is_match(at)) ? o.no : o.rel;
    const uint32_t add = (PAD && t == OP_P) ? 0u : ((uint32_t)L << 4);
    o.acc = same ? o.acc + add : (live ? mk_op(t, L) : o.acc);
}
// End of the stage's output (wave-uniform call; `on`: the lanes whose stage ends): the open run goes out, the trailing-edge rule
// (I -> S, D dropped, merged again; lane_out_finish of lane_core.hpp) is applied to the unreleased tail -- the ops behind the last
// match written; without any match everything went through the leading rule already, which maps the same ops the same way -- and
// everything is released.
template <int N>
PLO_DEV void ring_finish(LaneRing<N> &o, bool on) {
    {
        const bool flush = on & (o.acc >= 16u);
        const bool ok = (o.no - o.rk) < N;
        if (flush & ok) o.b[(o.no & (N - 1)) * 64] = o.acc;
        o.ovf = o.ovf | (flush & !ok);
        o.no += flush ? 1 : 0;
        o.rel = (flush & b_is_match((int)(o.acc & 15u))) ? o.no : o.rel;
        o.acc = on ? 0u : o.acc;
    }
    const bool fix = on & !o.ovf & o.seen_m & (o.rel < o.no);
    int i = o.rel, w = o.rel;
    uint32_t run = 0;  // open run of the rewritten tail (0: none)
    while (wv::ballot(fix & (i < o.no)) != 0ull) {
        const bool act = fix & (i < o.no);
        const uint32_t c = o.b[((act ? i : 0) & (N - 1)) * 64];
        int t = op_type(c);
        const int L = op_len(c);
        i += act ? 1 : 0;
        const bool keep = act & (t != OP_D);  // a trailing D becomes S(0), which compress_cigar drops
        t = (t == OP_I) ? (int)OP_S : t;
        const bool same = keep & (run >= 16u) & (t == (int)(run & 15u));
        const bool flush = keep & !same & (run >= 16u);
        if (flush) o.b[(w & (N - 1)) * 64] = run;
        w += flush ? 1 : 0;
        const uint32_t add = (t == OP_P) ? 0u : ((uint32_t)L << 4);
        run = same ? run + add : (keep ? mk_op(t, L) : run);
    }
    {
        const bool flush = fix & (run >= 16u);
        if (flush) o.b[(w & (N - 1)) * 64] = run;
        w += flush ? 1 : 0;
        o.no = fix ? w : o.no;
        o.rel = on ? o.no : o.rel;
    }
}

constexpr int STREAM_A_PUSH = 5;  // most ops one step of a stage can flush into its ring: shift event M I D + M other,
constexpr int STREAM_B_PUSH = 2;  // liftover: gap deletion + piece,
constexpr int STREAM_C_PUSH = 5;  // simplify: M I D M + the copied op
constexpr int STREAM_REFILL = 8;  // ops per refill of IN / per flush of Q3 (two 16-byte accesses per lane)
constexpr int STREAM_A_MIN_IN = 4;  // a shift step is worth starting with this many ops of input in the ring (or the input's end)
PLO_DEV constexpr int stream_lds_dwords(int ni, int n1, int n2, int n3) { return 64 * (ni + n1 + n2 + n3) + LANE_KVS_DWORDS; }
// slot of the output buffer an item is given up front: its region bound (enumerate.hpp lane_region_dwords) + one flush
PLO_DEV int stream_out_alloc(int n_m, int w0, int w1) { return lane_region_dwords(n_m, w0, w1) + STREAM_REFILL; }

// One persistent wave over the class-order positions [b0, e0) and then [b1, e1) (its share of the two heavy classes).
template <bool SP, int NI, int N1, int N2, int N3>
PLO_DEV void lane_stream(const DevIndex &ix, const DevBatch &bt, const DevWork &wk, uint32_t b0, uint32_t e0, uint32_t b1, uint32_t e1, uint32_t *lds,
                         WaveCtx &ctx) {
    static_assert(NI >= STREAM_REFILL + STREAM_A_MIN_IN && (NI & (NI - 1)) == 0, "IN ring");
    static_assert(N1 > STREAM_A_PUSH && N2 > STREAM_B_PUSH && N3 >= STREAM_C_PUSH + STREAM_REFILL, "rings");
    const int lane = wv::lane();
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    uint32_t *const in_b = lds + lane;
    uint32_t *const kvs = lds + 64 * (NI + N1 + N2 + N3);
    const uint8_t *const safe = (const uint8_t *)plo_safe_words;
    const int n_cig_all = (int)bt.seg_cigar_off[bt.n_segs];  // ops in the batch's CIGAR buffer
    uint32_t q_next = b0, q_end = e0;  // the wave's queue (wave-uniform)
    bool second = false;
    if (q_next >= q_end) {
        q_next = b1;
        q_end = e1;
        second = true;
    }

    // ---- per-lane item state ------------------------------------------------------------------------------------------------
    bool live = false, dead = false, ovf = false;
    uint32_t g = 0;
    int status = PLO_ITEM_LIFTED;
    unsigned algo = 0;
    int n_in = 0, in_off = 0, pos1 = 0, kv0 = 0, kv1 = 0, W0 = 0, W1 = 0, seq_len = 0, shift_ref_len = 0, chrom_ref_len = 0, alloc = 0;
    bool len_bad = false, rev = false, do_shift = false;
    unsigned long long shift_ref = 0, chrom_ref = 0, out_base = 0;
    bool fits = true;
    ReadSeq rd = item_read_seq<SP>(bt, 0ull, 0, 0);
    int in_hi = 0;  // input ops loaded into IN (walking order)
    // A: left shift
    int a_k = 0, ref_head = 0, read_head = 0, match = 0, del = 0, ins = 0, blk_ref = 0, blk_read = 0, p_match = 0, p_ins = 0, p_del = 0, msince = 0, probes = 0;
    bool a_fin = false, in_blk = false, pend = false, panic = false;
    LaneProbe pr;
    LaneRing<N1> q1;
    q1.b = lds + 64 * NI + lane;
    // B: liftover
    int b_k = 0, t_op = 0, seg_start = 0, seg_end = 0, block_pos = 0, r2s = 0, r2e = 0, kb = 0, vb = NONE32, kn = IMAX, vn = NONE32, kf = IMAX, vf = NONE32, ni = 0;
    bool b_fin = false, b_started = false, in_op = false, ism_op = false, bvalid = false, has_start = false, has_end = false, kv_lds = false;
    LaneRing<N2> q2;
    q2.b = lds + 64 * (NI + N1) + lane;
    int kvs_base = 0, kvs_cnt = 0;
    // C: simplify
    int c_ref_head = 0, c_read_head = 0, c_del = 0, c_ins = 0, c_blk_ref = 0, c_blk_read = 0, cmp = 0;
    bool c_fin = false, c_started = false, c_in_blk = false, spanic = false, zero_m = false, c_miss = false, passthru = false;
    LaneRing<N3> q3;  // consumer: the flush to the output buffer (rk = ops flushed)
    q3.b = lds + 64 * (NI + N1 + N2) + lane;

    // entry `idx` of the block map for the lanes `on`: from the staged copy, or (items outside it) from global memory
    auto kv_fetch = [&](bool on, int idx, int &key, int &val) {
        const bool l = on & kv_lds & ((unsigned)(idx - kvs_base) < (unsigned)kvs_cnt);
        const uint32_t *q = kvs + 2 * (l ? idx - kvs_base : 0);
        const int lk = (int)q[0], lv = (int)q[1];
        key = l ? lk : key;
        val = l ? lv : val;
        const bool gl = on & !l;
        if (wv::ballot(gl) != 0ull) {
            if (gl) {
                const KV e = ix.kv[idx];
                key = e.key;
                val = e.val;
            }
        }
    };

    // ---- S: lanes without an item take the wave's next ones ------------------------------------------------------------------
    auto step_start = [&]() {
        const unsigned long long want = wv::ballot(!live);
        const uint32_t left = q_end - q_next;
        const uint32_t rank = (uint32_t)__builtin_popcountll(want & lt_mask);
        const bool mine = !live & (rank < left);
        const uint32_t n_take = (uint32_t)__builtin_popcountll(want) < left ? (uint32_t)__builtin_popcountll(want) : left;
        const uint32_t at = q_next + rank;
        q_next += n_take;
        if (q_next >= q_end && !second) {
            q_next = b1;
            q_end = e1;
            second = true;
        }
        int n_m = 0;
        bool flip = false;
        unsigned long long seq_off = 0;
        if (mine) {
            g = wk.perm[at];
            in_off = (int)wk.d.in_off[g];
            n_in = (int)wk.d.n_in[g];
            n_m = (int)wk.d.n_m[g];
            W0 = (int)wk.d.w0[g];
            W1 = (int)wk.d.w1[g];
            kv0 = (int)wk.d.kv0[g];
            kv1 = (int)wk.d.kv1[g];
            const uint32_t fl = wk.d.flags[g];
            pos1 = wk.d.pos1[g];
            seq_len = (int)wk.d.seq_len[g];
            const uint32_t read_len_in = wk.d.read_len[g];
            len_bad = read_len_in == 0xffffffffu || (uint32_t)seq_len != read_len_in;  // LENGTH CHECK, see lift_tile
            seq_off = wk.d.seq_off[g];
            shift_ref = wk.d.shift_ref[g];
            shift_ref_len = wk.d.shift_ref_len[g];
            chrom_ref = wk.d.chrom_ref[g];
            chrom_ref_len = wk.d.chrom_ref_len[g];
            rev = (fl & ITF_REV) != 0;
            flip = (fl & ITF_FLIP) != 0;
            do_shift = (fl & ITF_CONTIG_FWD) == 0;  // (all stages: the shift runs for reverse-mapped contig segments)
        }
        // the item's slot of the output buffer, allocated now: slabs as in lane_tile (one device-scope atomic per slab)
        const int want_ops = mine ? stream_out_alloc(n_m, W0, W1) : 0;
        const int inc = wv::scan_add(want_ops);
        const int total = wv::bcast_last(inc);
        if ((unsigned long long)total > ctx.slab_left) {  // wave-uniform: reserve a new slab
            const unsigned long long wnt = (unsigned long long)total > SLAB_OPS ? (unsigned long long)total : SLAB_OPS;
            unsigned long long nb = 0;
            if (lane == 0) nb = wv::atomic_add_global(&wk.counters[CNT_CIGAR], wnt) + wk.slab_offset;
            ctx.slab_base = wv::bcast_first(nb);
            ctx.slab_left = wnt;
        }
        const unsigned long long gbase = ctx.slab_base;
        ctx.slab_base += (unsigned long long)total;
        ctx.slab_left -= (unsigned long long)total;
        const bool fit = gbase + (unsigned long long)total <= wk.out_cap;
        if (!fit && total > 0 && lane == 0) wv::atomic_add_global(&wk.counters[CNT_OVERFLOW], 1ull);
        if (mine) {
            ReadSeq nrd = item_read_seq<SP>(bt, seq_off, seq_len, flip ? 1 : 0);
            rd = nrd;
        }
        out_base = mine ? gbase + (unsigned long long)(inc - want_ops) : out_base;
        alloc = mine ? want_ops : alloc;
        fits = mine ? fit : fits;
        live = live | mine;
        ovf = ovf & !mine;
        status = mine ? (int)PLO_ITEM_LIFTED : status;
        // rev_contig_seq.unwrap() on None (src/read_alignment_scanner.rs:174); an output buffer too small: the host runs the batch again
        const bool nosref = mine & do_shift & (shift_ref == 0ull);
        status = nosref ? (int)PLO_ITEM_PANIC : status;
        dead = mine ? (nosref | !fit) : dead;
        algo = mine ? 0u : algo;
        if (mine) {
            int nb = kv1 - kv0, lg = 0;
            while ((1 << lg) < nb) ++lg;
            algo = 16u * (unsigned)(W1 - W0) + 8u * (unsigned)lg;
        }
        in_hi = mine ? 0 : in_hi;
        // A
        a_k = mine ? 0 : a_k;
        a_fin = mine ? !do_shift : a_fin;
        ref_head = mine ? pos1 : ref_head;
        read_head = mine ? 0 : read_head;
        match = mine ? 0 : match;
        del = mine ? 0 : del;
        ins = mine ? 0 : ins;
        in_blk = in_blk & !mine;
        pend = pend & !mine;
        panic = panic & !mine;
        msince = mine ? 0 : msince;
        probes = mine ? 0 : probes;
        ring_reset(q1, mine);
        // B
        b_k = mine ? 0 : b_k;
        b_fin = b_fin & !mine;
        b_started = b_started & !mine;
        in_op = in_op & !mine;
        ism_op = ism_op & !mine;
        bvalid = bvalid & !mine;
        has_start = has_start & !mine;
        has_end = has_end & !mine;
        kb = mine ? 0 : kb;
        vb = mine ? NONE32 : vb;
        kn = mine ? IMAX : kn;
        vn = mine ? NONE32 : vn;
        kf = mine ? IMAX : kf;
        vf = mine ? NONE32 : vf;
        ni = mine ? W0 + 2 : ni;
        ring_reset(q2, mine);
        // C
        c_fin = c_fin & !mine;
        c_started = c_started & !mine;
        c_in_blk = c_in_blk & !mine;
        spanic = spanic & !mine;
        zero_m = zero_m & !mine;
        c_miss = c_miss & !mine;
        passthru = mine ? len_bad : passthru;
        status = (mine & len_bad & !dead) ? (int)PLO_ITEM_LEN_MISMATCH : status;  // src/read_alignment_scanner.rs:204-229, decided from the descriptors
        c_read_head = mine ? 0 : c_read_head;
        c_del = mine ? 0 : c_del;
        c_ins = mine ? 0 : c_ins;
        cmp = mine ? 0 : cmp;
        ring_reset(q3, mine);
        // the block-map entries the live items' cursors read -> LDS (lane_tile: one coalesced load; here whenever items start)
        {
            const bool use = live & !dead;
            const int need_hi = wv::imin(kv1, W1 + 2);
            kvs_base = -wv::reduce_max(use ? -W0 : -IMAX);
            const int top = wv::reduce_max(use ? need_hi : 0);
            kvs_cnt = wv::imax(0, wv::imin(top - kvs_base, LANE_KVS));
            kv_lds = use & (need_hi <= kvs_base + kvs_cnt);
            KV e0 = {0, 0}, e1 = {0, 0};
            if (lane < kvs_cnt) e0 = ix.kv[kvs_base + lane];
            if (lane + 64 < kvs_cnt) e1 = ix.kv[kvs_base + lane + 64];
            wv::sync();
            kvs[2 * lane] = (uint32_t)e0.key;
            kvs[2 * lane + 1] = (uint32_t)e0.val;
            kvs[2 * (lane + 64)] = (uint32_t)e1.key;
            kvs[2 * (lane + 64) + 1] = (uint32_t)e1.val;
            wv::sync();
        }
        // the liftover's cursor: the first two entries of the item's window
        kv_fetch(mine & !dead & (W0 < kv1), W0, kn, vn);
        kv_fetch(mine & !dead & (W0 + 1 < kv1), W0 + 1, kf, vf);
    };

    // ---- E: finished items leave their results -------------------------------------------------------------------------------
    auto step_end = [&](bool em) {
        {   // items a ring or the output slot could not hold: the wave-cooperative code takes them (retry list)
            const bool re = em & ovf;
            const unsigned long long om = wv::ballot(re);
            if (om != 0ull) {
                int slot = 0;
                if (lane == 0) slot = (int)wv::atomic_add_global(&wk.counters[CNT_NRETRY], (unsigned long long)__builtin_popcountll(om));
                slot = wv::bcast_first(slot);
                if (re) {
                    wk.retry_list[slot + __builtin_popcountll(om & lt_mask)] = g;
                    wk.status[g] = (uint8_t)ITEM_NEED_BIG;
                }
            }
        }
        const bool done = em & !ovf;
        const bool emit_cigar = done & ((status == PLO_ITEM_LIFTED) | (status == PLO_ITEM_LEN_MISMATCH));
        const int oc = emit_cigar ? q3.no : 0;
        // :221 ref2_start_pos + the liftover's leading-edge shift; simplify_alignment_indels.rs:155 + the simplify stage's
        const int pos = r2s + q2.lead_shift + (passthru ? 0 : q3.lead_shift);
        if (done) {
            if (status == PLO_ITEM_NEED_BASES) wk.miss_list[wv::atomic_add_global(&wk.counters[CNT_NMISS], 1ull)] = g;  // rare
            wk.status[g] = (uint8_t)status;
            wk.pos[g] = emit_cigar ? (int64_t)pos : (int64_t)-1;
            wk.cig_off[g] = emit_cigar ? out_base : 0ull;
            wk.cig_len[g] = (uint32_t)oc;
            ctx.algo_bytes += algo + 2u * (unsigned)probes + 2u * (unsigned)cmp + 40u + 4u * (unsigned)n_in + 24u + 4u * (unsigned)oc;
            ctx.in_ops += (unsigned)n_in;
            ctx.out_ops += (unsigned)oc;
        }
        live = live & !em;
        dead = dead & !em;
    };

    // ---- R: eight more input ops -> IN (win_fill_input of lane_core.hpp; reversed on the fly for reverse-mapped contig segments) ----
    auto step_refill = [&](bool rm) {
        const uint32_t *const src = bt.cigar + in_off;
        const int idx = in_hi;
        uint32_t a[STREAM_REFILL];
        const uint32_t *qa[STREAM_REFILL / 4];
        bool edge = false;
#pragma unroll
        for (int q = 0; q < STREAM_REFILL / 4; ++q) {
            const int kq = idx + 4 * q;
            const bool want = rm & (kq < n_in);
            const int gi = rev ? n_in - 4 - kq : kq;
            const bool inside = (gi + in_off >= 0) & (gi + in_off + 4 <= n_cig_all);
            edge = edge | (want & !inside);
            qa[q] = (want & inside) ? src + gi : (const uint32_t *)plo_safe_words;
        }
        if (wv::ballot(edge) == 0ull) {
            Ops4 v[STREAM_REFILL / 4];
#pragma unroll
            for (int q = 0; q < STREAM_REFILL / 4; ++q) v[q] = *(const PLO_GLOBAL Ops4 *)qa[q];
#pragma unroll
            for (int q = 0; q < STREAM_REFILL / 4; ++q) {
                a[4 * q] = rev ? v[q].w : v[q].x;
                a[4 * q + 1] = rev ? v[q].z : v[q].y;
                a[4 * q + 2] = rev ? v[q].y : v[q].z;
                a[4 * q + 3] = rev ? v[q].x : v[q].w;
            }
        } else {
#pragma unroll
            for (int j = 0; j < STREAM_REFILL; ++j) {
                a[j] = 0u;
                if (rm && idx + j < n_in) a[j] = src[rev ? n_in - 1 - idx - j : idx + j];
            }
        }
        if (rm) {
#pragma unroll
            for (int j = 0; j < STREAM_REFILL; ++j) in_b[((idx + j) & (NI - 1)) * 64] = a[j];
            in_hi = wv::imin(idx + STREAM_REFILL, n_in);
        }
    };

    // ---- F: eight ops of Q3 -> the item's slot of the output buffer (the ops behind `rel` in the chunk are not final: they are
    // written again by the next flush, which starts at the new `rk`) ----
    auto step_flush = [&](bool fm) {
        const bool room_ok = q3.rk + STREAM_REFILL <= alloc;
        ovf = ovf | (fm & !room_ok);  // the output outgrew its slot (the region bound): retry list
        const bool go = fm & room_ok;
        if (go) {
            uint32_t *const dst = wk.out_cigar + out_base + (unsigned long long)q3.rk;
#pragma unroll
            for (int q = 0; q < STREAM_REFILL / 4; ++q) {
                Ops4 v;
                v.x = q3.b[((q3.rk + 4 * q) & (N3 - 1)) * 64];
                v.y = q3.b[((q3.rk + 4 * q + 1) & (N3 - 1)) * 64];
                v.z = q3.b[((q3.rk + 4 * q + 2) & (N3 - 1)) * 64];
                v.w = q3.b[((q3.rk + 4 * q + 3) & (N3 - 1)) * 64];
                *(PLO_GLOBAL Ops4 *)(dst + 4 * q) = v;
            }
            q3.rk = wv::imin(q3.rk + STREAM_REFILL, q3.rel);
        }
    };

    // ---- A: one round of the LEFT SHIFT (left_shift_indels.rs:17-39 + cigar_indel_shifter.rs:10-165), lane_tile's event-aligned walk:
    // every lane scans to its next event (or to the end of the input loaded so far), then all run the event code together ----
    auto step_shift = [&](bool am) {
        const uint8_t *const sref = (const uint8_t *)(uintptr_t)shift_ref;
        PLO_MARK("STREAM SHIFT STEP BEGIN");
        // the probes of the clusters that ended at the lanes' last events go out now: their round trips run under this round's scan
        if (wv::ballot(am & pend) != 0ull) lane_probe_load(pr, am & pend, sref, shift_ref_len, rd, safe);
        bool stop = !am, got = false, ev_other = false, ev_end = false;
        int ev_t = 0, ev_L = 0;
        while (wv::ballot(!stop) != 0ull) {
            const bool act = !stop;
            const bool have = a_k < n_in;
            const bool stall = act & have & (a_k >= in_hi);  // the op is not in the ring yet (R)
            const bool okop = act & have & !stall;
            const uint32_t c = in_b[((okop ? a_k : 0) & (NI - 1)) * 64];
            const int t = op_type(c), L = op_len(c);
            const bool indel = okop & b_is_indel(t);
            const bool ism = okop & b_is_match(t);
            const bool other = okop & !indel & !ism;
            const bool atend = act & !have;
            const bool ev = act & !stall & ((in_blk & (ism | other | atend)) | other | atend);
            const bool take = act & !ev & !stall;
            const bool memb = take & indel & (L > 0);  // add_del / add_ins (:73-85, len > 0 only)
            const bool open = memb & !in_blk;
            blk_ref = open ? ref_head : blk_ref;
            blk_read = open ? read_head : blk_read;
            in_blk = in_blk | memb;
            del += (memb & (t == OP_D)) ? L : 0;
            ins += (memb & (t == OP_I)) ? L : 0;
            const bool tm = take & ism;  // add_match (:150-153)
            msince += (tm & pend) ? L : 0;
            match += (tm & !pend) ? L : 0;
            read_head += (take & b_read_cons(t)) ? L : 0;
            ref_head += (take & b_ref_cons(t)) ? L : 0;
            a_k += take ? 1 : 0;
            ev_t = ev ? t : ev_t;
            ev_L = ev ? L : ev_L;
            ev_other = ev ? other : ev_other;
            ev_end = ev ? atend : ev_end;
            got = got | ev;
            stop = stop | ev | stall;
        }
        // the event
        const bool evl = am & got;
        const bool endc = evl & in_blk;  // end_indel (:101-148)
        const bool flushing = evl & (ev_other | ev_end);
        auto resolve = [&](bool on) {  // end_indel's emission for the pending cluster (:132-147)
            int h = lane_probe_finish(pr, on, sref, shift_ref_len, rd, probes);
            h = rd.miss ? 0 : h;
            const int sh = wv::imin(p_match, h);  // actual_shift_len (:132)
            ring_push<false>(q1, on & (p_match - sh > 0), OP_M, p_match - sh);
            ring_push<false>(q1, on & (p_ins > 0), OP_I, p_ins);
            ring_push<false>(q1, on & (p_del > 0), OP_D, p_del);
            match = on ? sh + msince : match;
            msince = on ? 0 : msince;
            pend = pend & !on;
        };
#pragma nounroll
        for (int pass = 0; pass < 2; ++pass) {
            const bool res = (pass == 0 ? evl : flushing) & pend;
            if (wv::ballot(res) != 0ull) {
                if (pass == 1) lane_probe_load(pr, res, sref, shift_ref_len, rd, safe);
                resolve(res);
            }
            if (pass == 1) break;
            if (wv::ballot(endc) != 0ull) {
                // (lane_probe_arm writes the parameters of every lane; lanes that sit this step out may have a cluster pending)
                const int re0 = pr.re, qe0 = pr.qe, mk0 = pr.maxk;
                lane_probe_arm(pr, endc, shift_ref_len, blk_ref, del, rd, blk_read, ins, match, panic);
                pr.re = endc ? pr.re : re0;
                pr.qe = endc ? pr.qe : qe0;
                pr.maxk = endc ? pr.maxk : mk0;
                p_match = endc ? match : p_match;
                p_ins = endc ? ins : p_ins;
                p_del = endc ? del : p_del;
                ins = endc ? 0 : ins;
                del = endc ? 0 : del;
                in_blk = in_blk & !endc;
                pend = pend | endc;
            }
            if (wv::ballot(flushing & pend) == 0ull) break;
        }
        if (wv::ballot(flushing) != 0ull) {  // add_other (:155-165); at the end: get_cigar()'s add_other(None) (:54-60)
            ring_push<false>(q1, flushing & (match > 0), OP_M, match);
            match = flushing ? 0 : match;
            const bool oth = flushing & ev_other;
            ring_push<true>(q1, oth, ev_t, ev_L);
            read_head += (oth & b_read_cons(ev_t)) ? ev_L : 0;
            ref_head += (oth & b_ref_cons(ev_t)) ? ev_L : 0;
            a_k += oth ? 1 : 0;
            const bool fin_now = flushing & ev_end;
            if (wv::ballot(fin_now) != 0ull) {
                ring_finish(q1, fin_now);  // :35-38 clean_up_cigar_edge_indels + compress
                a_fin = a_fin | fin_now;
                // absent bases (sparse batches) come first: what the probes saw then is not the read
                const bool bad = fin_now & (panic | rd.miss);
                status = bad ? (rd.miss ? (int)PLO_ITEM_NEED_BASES : (int)PLO_ITEM_PANIC) : status;
                dead = dead | bad;
            }
        }
        ovf = ovf | (am & q1.ovf);
        PLO_MARK("STREAM SHIFT STEP END");
    };

    // ---- B: one (op x block) piece of the LIFTOVER (src/liftover_read_alignment.rs:35-223), lane_tile's flat loop body ----
    auto step_liftover = [&](bool bm, bool b_avail) {
        PLO_MARK("STREAM LIFTOVER STEP BEGIN");
        const bool fetch = bm & !in_op & b_avail;
        const bool b_end = bm & !in_op & !b_avail;  // (ready without an op to fetch: the producer has ended)
        // the first op: the producer's leading-edge shift is final (a release came, or its end)
        const bool first = fetch & !b_started;
        seg_start = first ? wrap_add(pos1, do_shift ? q1.lead_shift : 0) : seg_start;  // left_shift_indels.rs:38 pos + ref_pos_shift
        b_started = b_started | first;
        const uint32_t *const rp = do_shift ? q1.b + (((fetch ? q1.rk : 0) & (N1 - 1)) * 64) : in_b + (((fetch ? b_k : 0) & (NI - 1)) * 64);
        const uint32_t c = *rp;
        q1.rk += (fetch & do_shift) ? 1 : 0;
        b_k += (fetch & !do_shift) ? 1 : 0;
        const int tf = op_type(c), Lf = op_len(c);
        const bool copy = fetch & (((0x32u >> tf) & 1u) != 0u);  // I S H: :157-160 copied through; Pad (:213) emits nothing
        const bool start = fetch & b_ref_cons(tf) & (Lf > 0);
        t_op = start ? tf : t_op;
        ism_op = start ? b_is_match(tf) : ism_op;
        seg_end = start ? seg_start + Lf : seg_end;
        block_pos = start ? seg_start : block_pos;
        in_op = in_op | start;
        // the piece starts in the next block (get_ref_range walks on, read_to_ref_map.rs:79-84)
        const bool adv = bm & in_op & (kn <= block_pos);
        int fk = IMAX, fv = NONE32;
        kv_fetch(adv & (ni < kv1), ni, fk, fv);  // (used at the end of the step)
        kb = adv ? kn : kb;
        vb = adv ? vn : vb;
        bvalid = bvalid | adv;
        kn = adv ? kf : kn;
        vn = adv ? vf : vn;
        ni += adv ? 1 : 0;
        // (kn <= block_pos still: the shift stage moved the start past another key; the walk goes on next step)
        const bool piece = bm & in_op & (kn > block_pos);
        const int pend_ = wv::imin(seg_end, kn);  // :62-67
        const int plen = pend_ - block_pos;
        const bool mapped = bvalid & (vb != NONE32);
        const bool mp = piece & mapped;
        const bool set_start = mp & ism_op & !has_start;  // :84-88
        r2s = set_start ? wrap_add(vb, block_pos - kb) : r2s;
        has_start = has_start | set_start;
        const int d = wrap_add(vb, -r2e);  // :91-96 (wrapping: vb is NONE32 where the piece is not mapped, and then unused)
        const bool e0 = mp & has_end & (d > 0) & has_start;
        has_end = has_end | mp;
        r2e = mp ? wrap_add(vb, pend_ - kb) : r2e;  // :98-100
        // :102-109 mapped piece | :111-115 insertion over an unmapped block | :117-123 soft clip before the first block
        const bool e1p = piece & (mapped ? (ism_op | has_start) : ism_op);
        const int t1p = mapped ? (t_op == OP_D ? (int)OP_D : (t_op == OP_N ? (int)OP_N : (int)OP_M)) : (bvalid ? (int)OP_I : (int)OP_S);
        block_pos = piece ? pend_ : block_pos;
        const bool done = piece & (pend_ >= seg_end);
        in_op = in_op & !done;
        seg_start = done ? seg_end : seg_start;
        ring_push<false>(q2, e0, OP_D, d);
        ring_push<false>(q2, copy | e1p, copy ? tf : t1p, copy ? Lf : plen);
        kf = adv ? fk : kf;  // the entry after next (kv_fetch above)
        vf = adv ? fv : vf;
        if (wv::ballot(b_end) != 0ull) {
            ring_finish(q2, b_end);  // :219-220
            b_fin = b_fin | b_end;
            const bool nolift = b_end & !has_start;  // :218 ref2_start_pos.map(...) on None
            status = nolift ? (int)PLO_ITEM_NO_LIFTOVER : status;
            dead = dead | nolift;
        }
        ovf = ovf | (bm & q2.ovf);
        PLO_MARK("STREAM LIFTOVER STEP END");
    };

    // ---- C: one op of SIMPLIFY (src/simplify_alignment_indels.rs:5-156), lane_tile's loop body; items that fail the length check
    // (src/read_alignment_scanner.rs:204-229) keep the liftover's CIGAR: their ops pass through ----
    auto step_simplify = [&](bool cm, bool c_avail) {
        PLO_MARK("STREAM SIMPLIFY STEP BEGIN");
        const uint8_t *const cref = (const uint8_t *)(uintptr_t)chrom_ref;
        const bool a_miss = rd.miss;  // (the shift stage's flag; this stage's absent bases are c_miss)
        rd.miss = false;
        const bool valid = cm & c_avail;
        const bool atend = cm & !c_avail;
        const bool first = valid & !c_started;
        c_ref_head = first ? wrap_add(r2s, q2.lead_shift) : c_ref_head;  // :221 the liftover's position
        c_started = c_started | first;
        const uint32_t c = q2.b[((valid ? q2.rk : 0) & (N2 - 1)) * 64];
        q2.rk += valid ? 1 : 0;
        const int t = op_type(c), L = op_len(c);
        const bool raw = valid & passthru;
        if (raw) q3.b[(q3.no & (N3 - 1)) * 64] = c;
        q3.no += raw ? 1 : 0;
        q3.rel = raw ? q3.no : q3.rel;
        const bool sv = valid & !passthru, se = atend & !passthru;
        const bool indel = sv & b_is_indel(t);
        const bool endc = c_in_blk & !indel & (sv | se);
        if (wv::ballot(endc) != 0ull) {  // CigarBlockInfo::end_indel (:35-111)
            // :41-44 one kind only (nothing for 0 / 0); :45-48 1 / 1 -> M(1); else the base comparisons
            const bool single = endc & ((c_del == 0) | (c_ins == 0));
            const bool one_one = endc & (c_del == 1) & (c_ins == 1);
            const bool cplx = endc & !single & !one_one;
            int pre = one_one ? 1 : 0, post = 0;
            if (wv::ballot(cplx) != 0ull) {
                if (cplx) {
                    if (c_blk_ref < 0 || c_blk_ref + c_del - 1 >= chrom_ref_len || c_blk_read + c_ins - 1 >= rd.len) {
                        spanic = true;  // slice index out of bounds: the reference panics (:58-60)
                        c_del = 0;
                        c_ins = 0;
                    } else {
                        // :55-68 trailing bases shared by the inserted and the deleted sequence, then :71-85 leading ones
                        post = match_run_back(cref, chrom_ref_len, c_blk_ref + c_del, rd, c_blk_read + c_ins, wv::imin(c_del, c_ins), cmp);
                        c_del -= post;
                        c_ins -= post;
                        pre = match_run_fwd(cref, chrom_ref_len, c_blk_ref, rd, c_blk_read, wv::imin(c_del, c_ins), cmp);
                        c_del -= pre;
                        c_ins -= pre;
                        if (c_del == 1 && c_ins == 1) {  // :88-92
                            c_del = 0;
                            c_ins = 0;
                            ++post;
                        }
                    }
                }
            }
            // :101-104 M(pre) I D M(post).  A cluster of one kind (most) emits that one op: the M ops and the second kind only
            // where some lane has them
            const bool emit_id = endc & !one_one;
            const bool both = wv::ballot(endc & !single) != 0ull;
            if (both) ring_push<false>(q3, endc & (pre > 0), OP_M, pre);
            {
                const bool first_d = emit_id & (c_ins == 0);  // (then the one op is the D, if anything)
                ring_push<false>(q3, emit_id & ((first_d ? c_del : c_ins) > 0), first_d ? (int)OP_D : (int)OP_I, first_d ? c_del : c_ins);
                if (both) {
                    ring_push<false>(q3, emit_id & !first_d & (c_del > 0), OP_D, c_del);
                    ring_push<false>(q3, endc & (post > 0), OP_M, post);
                }
            }
            c_del = endc ? 0 : c_del;
            c_ins = endc ? 0 : c_ins;
            c_in_blk = c_in_blk & !endc;
        }
        const bool open = indel & !c_in_blk;  // _add_indel (:16-22)
        c_blk_ref = open ? c_ref_head : c_blk_ref;
        c_blk_read = open ? c_read_head : c_blk_read;
        c_in_blk = c_in_blk | indel;
        c_del += (indel & (t == OP_D)) ? L : 0;
        c_ins += (indel & (t == OP_I)) ? L : 0;
        const bool cp = sv & !indel;
        zero_m = zero_m | (cp & b_is_match(t) & (L == 0));  // an edge mark the writer would not see (LaneOut, lane_core.hpp)
        ring_push<true>(q3, cp, t, L);  // :144-147
        c_read_head += (sv & b_read_cons(t)) ? L : 0;
        c_ref_head += (sv & b_ref_cons(t)) ? L : 0;
        if (wv::ballot(atend) != 0ull) {
            ring_finish(q3, se);  // :153-154
            c_fin = c_fin | atend;
        }
        c_miss = c_miss | (cm & rd.miss);
        rd.miss = a_miss;
        {
            const bool bad = cm & (c_miss | spanic);
            status = bad ? (c_miss ? (int)PLO_ITEM_NEED_BASES : (int)PLO_ITEM_PANIC) : status;
            dead = dead | bad;
        }
        ovf = ovf | (cm & (q3.ovf | zero_m));
        PLO_MARK("STREAM SIMPLIFY STEP END");
    };

    // ---- the loop: one step per trip, the one most lanes are ready for -------------------------------------------------------
    for (;;) {
        const bool run = live & !dead & !ovf;
        // A: room for an event's ops, and input to scan (or its end)
        const bool a_rdy = run & !a_fin & (ring_room(q1) >= STREAM_A_PUSH) & ((in_hi - a_k >= STREAM_A_MIN_IN) | (in_hi >= n_in));
        // B: an op being cut into pieces, an op to fetch, or the producer's end
        const bool b_avail = do_shift ? (q1.rk < q1.rel) : (b_k < in_hi);
        const bool b_src_end = do_shift ? (a_fin & (q1.rk >= q1.no)) : (b_k >= n_in);
        const bool b_rdy = run & !b_fin & (ring_room(q2) >= STREAM_B_PUSH) & (in_op | b_avail | b_src_end);
        // C
        const bool c_avail = q2.rk < q2.rel;
        const bool c_src_end = b_fin & (q2.rk >= q2.no);
        const bool c_rdy = run & !c_fin & (ring_room(q3) >= STREAM_C_PUSH) & (c_avail | c_src_end);
        // R: room for eight more input ops
        const int in_lo = do_shift ? a_k : b_k;
        const bool r_rdy = run & (in_hi < n_in) & (in_hi - in_lo <= NI - STREAM_REFILL);
        const bool r_urgent = r_rdy & (in_hi - in_lo < STREAM_A_MIN_IN);
        // F: a chunk of released ops; or what there is when the writer is out of room or has ended
        const int f_have = q3.rel - q3.rk;
        const bool f_rdy = run & ((f_have >= STREAM_REFILL) | ((f_have > 0) & ((ring_room(q3) < STREAM_C_PUSH) | c_fin)));
        const bool f_urgent = f_rdy & ((ring_room(q3) < STREAM_C_PUSH) | c_fin);
        // E: killed, overflowed, or all flushed
        const bool e_rdy = live & (dead | ovf | (c_fin & (q3.rk >= q3.no)));
        // a lane that can do nothing at all is stuck behind a ring's unreleased tail: retry list
        const bool stuck = run & !(a_rdy | b_rdy | c_rdy | r_rdy | f_rdy | e_rdy);
        const unsigned long long mE = wv::ballot(e_rdy | stuck);
        if (mE != 0ull) {
            ovf = ovf | stuck;
            step_end(e_rdy | stuck);
            continue;
        }
        if (q_next < q_end && wv::ballot(!live) != 0ull) {
            step_start();
            continue;
        }
        const unsigned long long mA = wv::ballot(a_rdy), mB = wv::ballot(b_rdy), mC = wv::ballot(c_rdy), mR = wv::ballot(r_rdy), mF = wv::ballot(f_rdy);
        const int nA = __builtin_popcountll(mA), nB = __builtin_popcountll(mB), nC = __builtin_popcountll(mC);
        const int nR = __builtin_popcountll(mR), nF = __builtin_popcountll(mF);
        if ((mA | mB | mC | mR | mF) == 0ull) break;  // no lane has an item, the queue is empty
        const bool uR = wv::ballot(r_urgent) != 0ull, uF = wv::ballot(f_urgent) != 0ull;
        if (nR > 0 && (uR || nR >= 24 || (nA | nB | nC) == 0)) {
            step_refill(r_rdy);
            continue;
        }
        if (nF > 0 && (uF || nF >= 24 || (nA | nB | nC) == 0)) {
            step_flush(f_rdy);
            continue;
        }
        if (nA >= nB && nA >= nC) {
            ctx.u_act += (unsigned)nA;
            ctx.u_trips += 1;
            step_shift(a_rdy);
        } else if (nB >= nC) {
            ctx.u_act += (unsigned)nB;
            ctx.u_trips += 1;
            step_liftover(b_rdy, b_avail);
        } else {
            ctx.u_act += (unsigned)nC;
            ctx.u_trips += 1;
            step_simplify(c_rdy, c_avail);
        }
    }
}

// Persistent wave `first` of `n_waves` over the heavy classes: positions [lo, mid) (forward-mapped contig segments) and [mid, hi)
// (reverse-mapped: the shift stage) of the class order; every wave takes the same share of either class, contiguous -- its items are
// neighbours on a contig and share block-map and reference lines.
template <bool SP, int NI, int N1, int N2, int N3>
PLO_DEV void lane_stream_persistent(const DevIndex &ix, const DevBatch &bt, const DevWork &wk, uint32_t first, uint32_t n_waves, uint32_t lo, uint32_t mid,
                                    uint32_t hi, uint32_t *lds, WaveCtx &ctx) {
    const unsigned long long n0 = mid - lo, n1 = hi - mid;
    const uint32_t b0 = lo + (uint32_t)(n0 * first / n_waves), e0 = lo + (uint32_t)(n0 * (first + 1ull) / n_waves);
    const uint32_t b1 = mid + (uint32_t)(n1 * first / n_waves), e1 = mid + (uint32_t)(n1 * (first + 1ull) / n_waves);
    lane_stream<SP, NI, N1, N2, N3>(ix, bt, wk, b0, e0, b1, e1, lds, ctx);
}

}  // namespace plo
