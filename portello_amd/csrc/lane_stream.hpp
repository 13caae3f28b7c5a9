// lane_stream.hpp -- heavy items, STREAMED: the lane-per-item stages of lane_core.hpp as a PIPELINE OF WAVES chained through LDS rings.
//
// k_lift_lanes_g (lane_core.hpp, WIN) runs shift -> liftover -> simplify one after the other over a lane's whole CIGAR: every stage
// writes its CIGAR to a region in global memory and the next one reads it back (5.5 x the algorithmic traffic on the stress profile),
// every window refill is a global round trip for all 64 lanes, a wave runs for as long as its longest item, and an item's latency is
// the SUM of its stages.  Here a workgroup is a TEAM of waves that work on the same 64 item slots at the same time, one stage per wave:
//
//   reverse-mapped contig segments   batch CIGAR -> [IN] -> A: left shift -> [Q1] -> B: liftover -> [Q2] -> C: simplify -> [Q3] -> output
//   forward-mapped                   batch CIGAR -> [IN] ---------------------------> B: liftover -> [Q2] -> C: simplify -> [Q3] -> output
//
//   * lane l of every wave of the team works on the item in slot l; the ops flow from wave to wave through RINGS in LDS (element e of
//     lane l: word (e mod N) * 64 + l of the ring, bank = lane), with a released count written by the producer and a consumed count
//     written by the consumer -- plain LDS words: the LDS executes a wave's operations in order, so the data written before a count is
//     there when the other wave sees the count; no barriers, the waves drift as their stages' costs dictate (up to a ring's depth);
//   * a stage's writer (clean_up_cigar_edge_indels + compress_cigar as a streaming writer, LaneOut of lane_core.hpp) RELEASES an op to
//     the next stage only when it is final: the trailing-edge rule of clean_up_cigar_edge_indels (cigar/mod.rs:265-291) rewrites the ops
//     behind the LAST alignment match, so ops are released up to the last match written and the tail behind it stays in the ring until
//     another match follows or the item ends (then the rule is applied to it in the ring).  An item's header (its position in the
//     class order, the producer's leading-edge position shift :277-283, the liftover's ref2_start_pos) travels IN the ring in front of
//     its ops and is released with the first of them, i.e. when those values are final; an end-of-item marker carries the status;
//   * a lane of the HEAD wave that has finished its item takes the team's next one while the other lanes carry on, and the item's
//     header starts it in the waves behind: a team's time is the sum of its items' ops / 64 at the pace of the slowest STAGE, not
//     groups x longest item x all stages;
//   * every wave holds ONE stage's state in registers (the three together spill), so more waves fit a SIMD;
//   * the only global traffic is the input CIGAR (16 ops per lane and refill, asked for at the top of a step and stored into [IN] at
//     its end), the probes, the output (an aligned 64-byte line of 16 ops per flush, straight into the item's slot of the output buffer,
//     allocated when the item reaches the last wave from its region bound) and the per-item descriptors / results: 2.9 x the
//     algorithmic bytes on the stress profile (k_lift_lanes_g: 5.5 x).
// An item whose unreleased tail outgrows a ring (its producer cannot make progress though the consumer has taken all there is) or
// whose output outgrows its slot is handed to the retry list -> wave-cooperative code, like every item a lane kernel cannot hold.
//
// The stage code is lane_core.hpp's, statement for statement (same reference lines), with ring reads / writes in place of region
// indices.  Stage set: STRAND | LSHIFT | LIFTOVER | LENCHECK | SIMPLIFY (PLO_STAGES_ALL) only; subsets take k_lift_lanes_g.
// (Measured and dropped, round 5: the same stages as steps of ONE wave's loop, the step most lanes are ready for chosen every trip --
// 24 ms against 9.4 ms on the stress profile: the state of three stages does not fit the registers (19 spills reloaded inside the
// loops, flags as bytes in VGPRs) and half the lanes sit out every step.)
// All citations are relative to /root/reference.
#pragma once
#include "lane_core.hpp"

namespace plo {

#ifdef PLO_EMULATOR
PLO_DEV void pipe_idle(int) {}
PLO_DEV void pipe_order() {}
#else
#ifndef PLO_PIPE_SLEEP
#define PLO_PIPE_SLEEP 4
#endif
// nothing to do this trip: leave the issue slots to the SIMD's other waves -- for longer when it was the same the trips before (a wave
// behind a slower stage waits for thousands of cycles: every look costs the busy waves some hundred issue slots)
PLO_DEV void pipe_idle(int streak) {
    if (streak < 3) __builtin_amdgcn_s_sleep(PLO_PIPE_SLEEP);
    else __builtin_amdgcn_s_sleep(8 * PLO_PIPE_SLEEP);
}
// Between a ring's words and the count that publishes them (producer), and between reading a count and the words it covers (consumer).
// DS operations of a wave execute in order and a workgroup shares one CU's LDS, so what is needed is that the COMPILER keeps the order; the
// workgroup-scope fences say so inside the HIP memory model -- on gfx950, outside threadgroup-split mode, they lower to a wait
// for the wave's outstanding LDS operations and nothing else.
PLO_DEV void pipe_order() {
#ifdef PLO_PIPE_ASM_ORDER
    asm volatile("" ::: "memory");
#else
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
#endif
}
#endif

constexpr uint32_t PIPE_PAIR = 0x4Fu;  // (a marker, see below) an indel cluster of more than one op begins behind this word

// Ring of N ops per lane (N a power of two): op e lives in b[(e & (N - 1)) * 64] (b: the lane's word of element 0; bank = lane).
// Producer side: lane_push()'s logic (leading edge, zero-length filter, run merging) with `rel`: entries [0, rel) are final; `rk`: what
// the consumer has taken (its own count, or the last look at the consumer's published one).
template <int N>
struct LaneRing {
    uint32_t *b = nullptr;
    int no = 0;        // entries written
    int rel = 0;       // entries released to the consumer
    int rk = 0;        // entries consumed
    int base = 0;      // first op of the current item (its header and the items before it lie below)
    uint32_t acc = 0;  // the open run (starts as Match(0), cigar/mod.rs:206)
    int lead_shift = 0;
    bool seen_m = false, ovf = false;
    bool in_pair = false;  // MARK (ring_push): the indel cluster being written has its PIPE_PAIR marker already
};
template <int N>
PLO_DEV int ring_room(const LaneRing<N> &r) { return N - (r.no - r.rk); }
// a new item's writer state (the ring's counts run on)
template <int N>
PLO_DEV void ring_new_item(LaneRing<N> &r, bool on) {
    r.acc = on ? 0u : r.acc;
    r.lead_shift = on ? 0 : r.lead_shift;
    r.seen_m = r.seen_m & !on;
    r.ovf = r.ovf & !on;
    r.in_pair = r.in_pair & !on;
}
// one raw word (headers, markers): the open run must have been flushed
template <int N>
PLO_DEV void ring_put(LaneRing<N> &o, bool on, uint32_t v) {
    if (on) o.b[(o.no & (N - 1)) * 64] = v;
    o.no += on ? 1 : 0;
}
// lane_push (lane_core.hpp) into a ring: straight-line; a flushed alignment-match op releases everything up to and including itself
// MARK (the liftover's ring, round 6): an indel cluster of MORE THAN ONE op -- the only place the simplify stage can change anything
// (simplify_alignment_indels.rs:41-48: a cluster of one kind is emitted as it is) -- gets a PIPE_PAIR marker in front of its first op; it is
// written when the cluster's first op is flushed with another indel behind it (both still unreleased: a release ends at a match).
template <bool PAD, int N, bool MARK = false>
PLO_DEV void ring_push(LaneRing<N> &o, bool on, int t, int L) {
    const bool lead = on & !o.seen_m;
    const bool drop_d = lead & (t == OP_D);
    o.lead_shift += drop_d ? L : 0;
    t = (lead & (t == OP_I)) ? (int)OP_S : t;
    o.seen_m = o.seen_m | (on & b_is_match(t));
    const bool live = on & !drop_d & (L > 0);
    const int at = (int)(o.acc & 15u);
    const bool same = live & (t == at);
    // (the callers leave room for a step's pushes; a push without room would overwrite what the consumer has not read: the op is lost,
    // the ring stays as it is -- every entry the consumer sees is a whole one -- and the item goes to the retry list)
    const bool wflush = live & !same & (o.acc >= 16u);
    bool pair = false;
    if constexpr (MARK) {
        pair = wflush & b_is_indel(at) & b_is_indel(t) & !o.in_pair;
        o.in_pair = (o.in_pair | pair) & !(live & !b_is_indel(t));
    }
    const bool ok = (o.no - o.rk) < N - (pair ? 1 : 0);
    o.ovf = o.ovf | (wflush & !ok);
    const bool flush = wflush & ok;
    if constexpr (MARK) {
        if (flush & pair) o.b[(o.no & (N - 1)) * 64] = PIPE_PAIR;
        o.no += (flush & pair) ? 1 : 0;
    }
    if (flush) o.b[(o.no & (N - 1)) * 64] = o.acc;
    o.no += flush ? 1 : 0;
    o.rel = (flush & b_is_match(at)) ? o.no : o.rel;
    const uint32_t add = (PAD && t == OP_P) ? 0u : ((uint32_t)L << 4);
    o.acc = same ? o.acc + add : (live ? mk_op(t, L) : o.acc);
}
// End of the item's ops (wave-uniform call; `on`: the lanes whose item ends): the open run goes out, the trailing-edge rule (I -> S,
// D dropped, merged again; lane_out_finish of lane_core.hpp) is applied to the item's unreleased tail -- the ops behind the last match
// written; without any match everything went through the leading rule already, which maps the same ops the same way.  Nothing is
// released here: the caller appends the end marker and releases all.
template <int N>
PLO_DEV void ring_finish(LaneRing<N> &o, bool on) {
    {
        const bool ok = (o.no - o.rk) < N;
        const bool wflush = on & (o.acc >= 16u);
        o.ovf = o.ovf | (wflush & !ok);
        const bool flush = wflush & ok;
        if (flush) o.b[(o.no & (N - 1)) * 64] = o.acc;
        o.no += flush ? 1 : 0;
        o.rel = (flush & b_is_match((int)(o.acc & 15u))) ? o.no : o.rel;
        o.acc = on ? 0u : o.acc;
    }
    const int from = wv::imax(o.rel, o.base);
    const bool fix = on & !o.ovf & o.seen_m & (from < o.no);
    int i = from, w = from;
    uint32_t run = 0;  // open run of the rewritten tail (0: none)
    while (wv::ballot(fix & (i < o.no)) != 0ull) {
        const bool act = fix & (i < o.no);
        const uint32_t c = o.b[((act ? i : 0) & (N - 1)) * 64];
        int t = op_type(c);
        const int L = op_len(c);
        i += act ? 1 : 0;
        // a trailing D becomes S(0), which compress_cigar drops; a PIPE_PAIR marker goes with the cluster it marked (no indel is left behind
        // the last match)
        const bool keep = act & (t != OP_D) & (t != 15);
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
    }
}

// ---- what travels in the rings beside ops: markers (low nibble 15; CIGAR op codes end at 8) ----
constexpr uint32_t PIPE_SOI = 0x1Fu;   // start of an item; raw words follow: Q1 {class-order position, A's lead shift}, Q2 {position, B's lead shift, r2s}
constexpr uint32_t PIPE_EOI = 0x2Fu;   // end of the item; bits 8..15: status (0: the stage has no objection)
constexpr uint32_t PIPE_TERM = 0x3Fu;  // the lane's last word
constexpr int PIPE_ST_OVF = 0xFE;      // EOI status: retry list
constexpr int PIPE_H1 = 3, PIPE_H2 = 4;  // entries of an item's header in Q1 / Q2 (SOI included)
PLO_DEV bool pipe_is_marker(uint32_t c) { return (c & 15u) == 15u; }
constexpr int STREAM_A_PUSH = 8;    // most ops one step of a stage can flush into its ring: shift event M I D, again for a cluster right in front of a flushing op, M other;
constexpr int STREAM_B_PUSH = 1;    // liftover: a gap deletion or a piece (+ one PIPE_PAIR marker where the ring has them: stream_b_push),
constexpr int STREAM_C_PUSH = 5;    // simplify: M I D M + the copied op
constexpr int STREAM_END_PUSH = 2;  // an item's end: the open run + the end marker
// The liftover's ring carries PIPE_PAIR markers, and the simplify wave copies unmarked stretches through, when the ring has 16 entries or
// more (the production geometry); the 8-entry ring of the tests' tightest geometry has no room for a marker beside a step's two ops, a
// header and an end, and keeps every op on the step-by-step path.
PLO_DEV constexpr bool stream_marks(int n2) { return n2 >= 16; }
// (one op per step, a marker in front of it at most -- and never less than an item's end needs: the step that takes the input's end marker
// writes nothing, and the end must find its room behind it or the lane waits for a consumer that has nothing released to take)
PLO_DEV constexpr int stream_b_push(int n2) { return (1 + (stream_marks(n2) ? 1 : 0)) > STREAM_END_PUSH ? (1 + (stream_marks(n2) ? 1 : 0)) : STREAM_END_PUSH; }

#ifndef PLO_PIPE_BULK
#define PLO_PIPE_BULK 8
#endif
constexpr int STREAM_BULK = PLO_PIPE_BULK;  // ops per lane the simplify wave copies through per trip where it cannot change them (0: every op through its step)
#ifndef PLO_PIPE_REFILL
#define PLO_PIPE_REFILL 16
#endif
constexpr int STREAM_REFILL = PLO_PIPE_REFILL;  // ops per refill of IN (16-byte loads): 16 = a 64-byte stretch per lane (8: FETCH_SIZE +10 %, a line's second half
                                                // is often fetched again)
#ifndef PLO_PIPE_FLUSH
#define PLO_PIPE_FLUSH 16
#endif
constexpr int STREAM_FLUSH = PLO_PIPE_FLUSH;  // ops per flush of Q3: 16 = one aligned 64-byte line per lane (the items' slots start on lines)
constexpr int STREAM_A_MIN_IN = 4;  // a shift step is worth starting with this many ops of input in the ring (or the input's end)
#ifndef PLO_PIPE_BURST
#define PLO_PIPE_BURST 4
#endif
constexpr int PIPE_BURST = PLO_PIPE_BURST;  // steps per trip of a stage's loop (while most of the trip's lanes stay ready): the trip's bookkeeping --
                                            // the other waves' counts, item starts and ends, the published counts -- is paid once
constexpr int STREAM_IN_LOW = 6;    // a lane with fewer input ops left in IN asks for a refill; all lanes with room take part in it
// A stage runs its step when at least half of its lanes with an item are ready for it -- or it has found nothing better to do twice
// in a row (lanes wait for the others to be fed rather than run the step's few hundred instructions for a handful of lanes)
#ifndef PLO_PIPE_WORTH4
#define PLO_PIPE_WORTH4 2  // quarters of the lanes with an item
#endif
PLO_DEV bool pipe_worth(unsigned long long ready, unsigned long long live, int streak) {
    return ready != 0ull && (4 * __builtin_popcountll(ready) >= PLO_PIPE_WORTH4 * __builtin_popcountll(live) || streak >= 2);
}
constexpr int PIPE_CTL = 4;         // control words per lane: Q1 released, Q1 consumed, Q2 released, Q2 consumed
constexpr int PIPE_WAVES = 3;
PLO_DEV constexpr int stream_lds_dwords(int ni, int n1, int n2, int n3) { return 64 * (ni + n1 + n2 + n3 + PIPE_CTL) + LANE_KVS_DWORDS; }
// slot of the output buffer an item is given when it reaches the last wave: its region bound (enumerate.hpp lane_region_dwords) + one flush
PLO_DEV int stream_out_alloc(int n_m, int w0, int w1) { return (lane_region_dwords(n_m, w0, w1) + STREAM_FLUSH + STREAM_FLUSH - 1) / STREAM_FLUSH * STREAM_FLUSH; }

// the team's LDS
template <int NI, int N1, int N2, int N3>
struct PipeMem {
    uint32_t *in, *q1, *q2, *q3, *kvs;
    volatile uint32_t *q1_rel, *q1_rk, *q2_rel, *q2_rk;
    PLO_DEV PipeMem(uint32_t *lds, int lane) {
        in = lds + lane;
        q1 = lds + 64 * NI + lane;
        q2 = lds + 64 * (NI + N1) + lane;
        q3 = lds + 64 * (NI + N1 + N2) + lane;
        uint32_t *ctl = lds + 64 * (NI + N1 + N2 + N3);
        q1_rel = ctl + lane;
        q1_rk = ctl + 64 + lane;
        q2_rel = ctl + 128 + lane;
        q2_rk = ctl + 192 + lane;
        kvs = ctl + 64 * PIPE_CTL;
    }
};

// ---- the head wave's input: the team's queue of items and the IN ring ------------------------------------------------------------
struct PipeQueue {
    uint32_t next, end;  // class-order positions (wave-uniform)
};
// lanes `want` take the next items of the queue (wave-uniform call); returns which got one and its class-order position
PLO_DEV bool pipe_take(PipeQueue &q, bool want, uint32_t &at) {
    const unsigned long long m = wv::ballot(want);
    const uint32_t left = q.end - q.next;
    const uint32_t rank = (uint32_t)__builtin_popcountll(m & ((1ull << wv::lane()) - 1ull));
    const bool mine = want & (rank < left);
    at = q.next + rank;
    const uint32_t n = (uint32_t)__builtin_popcountll(m);
    q.next += n < left ? n : left;
    return mine;
}
// eight more input ops of the lanes `rm` (win_fill_input of lane_core.hpp; reversed on the fly for reverse-mapped contig segments):
// the loads now (pipe_refill_issue, at the top of a step), the stores into IN at the step's end (pipe_refill_commit)
struct PipeRefill {
    uint32_t a[STREAM_REFILL];
    int idx;
    bool on;
};
PLO_DEV void pipe_refill_issue(PipeRefill &r, bool rm, const DevBatch &bt, int n_cig_all, int in_off, int n_in, bool rev, int in_hi) {
    const uint32_t *const src = bt.cigar + in_off;
    r.idx = in_hi;
    r.on = rm;
    const uint32_t *qa[STREAM_REFILL / 4];
    bool edge = false;
#pragma unroll
    for (int q = 0; q < STREAM_REFILL / 4; ++q) {
        const int kq = in_hi + 4 * q;
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
            r.a[4 * q] = rev ? v[q].w : v[q].x;
            r.a[4 * q + 1] = rev ? v[q].z : v[q].y;
            r.a[4 * q + 2] = rev ? v[q].y : v[q].z;
            r.a[4 * q + 3] = rev ? v[q].x : v[q].w;
        }
    } else {
#pragma unroll
        for (int j = 0; j < STREAM_REFILL; ++j) {
            r.a[j] = 0u;
            if (rm && in_hi + j < n_in) r.a[j] = src[rev ? n_in - 1 - in_hi - j : in_hi + j];
        }
    }
}
template <int NI>
PLO_DEV void pipe_refill_commit(const PipeRefill &r, uint32_t *in_b, int n_in, int &in_hi) {
    if (r.on) {
#pragma unroll
        for (int j = 0; j < STREAM_REFILL; ++j) in_b[((r.idx + j) & (NI - 1)) * 64] = r.a[j];
        in_hi = wv::imin(r.idx + STREAM_REFILL, n_in);
    }
}

// ====================================================================================================================================
// A: the LEFT SHIFT (left_shift_indels.rs:17-39 + cigar_indel_shifter.rs:10-165), head wave of a team of the reverse class.
// lane_tile's event-aligned walk: every lane scans to its next event (or to the end of the input loaded so far), then all run the event
// code together; a cluster's homology probe goes out at the top of the lane's next step.
// ====================================================================================================================================
template <bool SP, int NI, int N1, int N2, int N3>
PLO_DEV void pipe_stage_shift(const DevBatch &bt, const DevWork &wk, PipeQueue q, PipeMem<NI, N1, N2, N3> &pm, WaveCtx &ctx) {
    const uint8_t *const safe = (const uint8_t *)plo_safe_words;
    const int n_cig_all = (int)bt.seg_cigar_off[bt.n_segs];
    // per lane: the item
    bool live = false, termd = false, finp = false;  // finp: the item's ops are through, its end (open run, marker) waits for room
    int n_in = 0, in_off = 0, pos1 = 0, shift_ref_len = 0, in_hi = 0, hdr = 0;
    bool rev = false, hdr_open = false;
    unsigned long long shift_ref = 0;
    ReadSeq rd = item_read_seq<SP>(bt, 0ull, 0, 0);
    int a_k = 0, ref_head = 0, read_head = 0, match = 0, del = 0, ins = 0, blk_ref = 0, blk_read = 0, p_match = 0, p_ins = 0, p_del = 0, msince = 0, probes = 0;
    bool in_blk = false, pend = false, panic = false, nosref = false;
    int p_re = 0, p_qe = 0, p_maxk = 0;  // the pending cluster's probe (lane_probe_arm)
    LaneRing<N1> q1;
    q1.b = pm.q1;
    int streak = 0;  // trips without a step (wave-uniform)
    for (;;) {
        q1.rk = (int)*pm.q1_rk;
        pipe_order();
        // ---- lanes without an item: the team's next ones (header into Q1), or the lane's last word ----
        {
            const bool want = !live & !finp & !termd & (ring_room(q1) >= PIPE_H1);
            if (wv::ballot(want) != 0ull) {
                uint32_t at = 0;
                const bool mine = pipe_take(q, want, at);
                const bool last = want & !mine;  // the queue is empty
                ring_put(q1, last, PIPE_TERM);
                q1.rel = last ? q1.no : q1.rel;
                termd = termd | last;
                unsigned long long seq_off = 0;
                int seq_len = 0;
                bool flip = false;
                if (mine) {
                    const uint32_t g = wk.perm[at];
                    in_off = (int)wk.d.in_off[g];
                    n_in = (int)wk.d.n_in[g];
                    const uint32_t fl = wk.d.flags[g];
                    pos1 = wk.d.pos1[g];
                    seq_len = (int)wk.d.seq_len[g];
                    seq_off = wk.d.seq_off[g];
                    shift_ref = wk.d.shift_ref[g];
                    shift_ref_len = wk.d.shift_ref_len[g];
                    rev = (fl & ITF_REV) != 0;
                    flip = (fl & ITF_FLIP) != 0;
                    rd = item_read_seq<SP>(bt, seq_off, seq_len, flip ? 1 : 0);
                }
                ring_new_item(q1, mine);
                hdr = mine ? q1.no : hdr;
                ring_put(q1, mine, PIPE_SOI);
                ring_put(q1, mine, at);
                ring_put(q1, mine, 0u);  // (the lead shift: filled in when the header is released)
                q1.base = mine ? q1.no : q1.base;
                hdr_open = hdr_open | mine;
                live = live | mine;
                nosref = mine ? (shift_ref == 0ull) : nosref;  // rev_contig_seq.unwrap() on None (src/read_alignment_scanner.rs:174)
                in_hi = mine ? 0 : in_hi;
                a_k = mine ? 0 : a_k;
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
                // (an item without its reverse contig sequence ends at once: nothing of it is read)
                finp = finp | (mine & nosref);
                live = live & !(mine & nosref);
            }
        }
        // ---- an item's end: open run, trailing edge, header (if not out yet), end marker ----
        {
            const bool fe = finp & (ring_room(q1) >= STREAM_END_PUSH);
            if (wv::ballot(fe) != 0ull) {
                ring_finish(q1, fe);  // :35-38 clean_up_cigar_edge_indels + compress
                // absent bases (sparse batches) come first: what the probes saw then is not the read
                const int st = q1.ovf ? PIPE_ST_OVF : (nosref ? (int)PLO_ITEM_PANIC : (rd.miss ? (int)PLO_ITEM_NEED_BASES : (panic ? (int)PLO_ITEM_PANIC : 0)));
                if (fe & hdr_open) q1.b[((hdr + 2) & (N1 - 1)) * 64] = (uint32_t)q1.lead_shift;
                hdr_open = hdr_open & !fe;
                ring_put(q1, fe, PIPE_EOI | ((uint32_t)st << 8));
                q1.rel = fe ? q1.no : q1.rel;
                if (fe) ctx.algo_bytes += 2u * (unsigned)probes;
                finp = finp & !fe;
            }
        }
        // ---- the step ----
        const bool r_rdy = live & (in_hi < n_in) & (in_hi - a_k <= NI - STREAM_REFILL);
        const bool room_ok = ring_room(q1) >= STREAM_A_PUSH;
        const bool a_rdy = live & room_ok & ((in_hi - a_k >= STREAM_A_MIN_IN) | (in_hi >= n_in));
        // no room though the consumer has taken all there is: the ring is full of the item's unreleased tail -> retry list
        const bool stuck = live & !room_ok & (q1.rk == q1.rel);
        if (wv::ballot(stuck) != 0ull) {
            // the tail is dropped (the header stays: the waves behind must see the item to its end)
            q1.no = stuck ? wv::imax(q1.rel, q1.base) : q1.no;
            q1.acc = stuck ? 0u : q1.acc;
            q1.ovf = q1.ovf | stuck;
            finp = finp | stuck;
            live = live & !stuck;
        }
        // (a refill when some lane runs low; all lanes with room take part)
        const unsigned long long mR = wv::ballot(r_rdy & (in_hi - a_k < STREAM_IN_LOW)) != 0ull ? wv::ballot(r_rdy) : 0ull;
        const unsigned long long mA = pipe_worth(wv::ballot(a_rdy), wv::ballot(live), streak) ? wv::ballot(a_rdy) : 0ull;
        if ((mA | mR) != 0ull) {
            PipeRefill rf;
            pipe_refill_issue(rf, r_rdy & (mR != 0ull), bt, n_cig_all, in_off, n_in, rev, in_hi);
            if (mA != 0ull) {
                ctx.u_act += (unsigned)__builtin_popcountll(mA);
                ctx.u_trips += 1;
                const bool am = a_rdy;
                const uint8_t *const sref = (const uint8_t *)(uintptr_t)shift_ref;
                PLO_MARK("PIPE SHIFT STEP BEGIN");
                // the probes of the clusters that ended at the lanes' last events go out now: their round trips run under this step's scan
                LaneProbe pr;
                pr.re = p_re;
                pr.qe = p_qe;
                pr.maxk = p_maxk;
                if (wv::ballot(am & pend) != 0ull) lane_probe_load(pr, am & pend, sref, shift_ref_len, rd, safe);
                bool stop = !am, got = false, ev_other = false, ev_end = false;
                int ev_t = 0, ev_L = 0;
                while (wv::ballot(!stop) != 0ull) {
                    const bool act = !stop;
                    const bool have = a_k < n_in;
                    const bool stall = act & have & (a_k >= in_hi);  // the op is not in the ring yet
                    const bool okop = act & have & !stall;
                    const uint32_t c = pm.in[((okop ? a_k : 0) & (NI - 1)) * 64];
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
                p_re = pr.re;
                p_qe = pr.qe;
                p_maxk = pr.maxk;
                if (wv::ballot(flushing) != 0ull) {  // add_other (:155-165); at the end: get_cigar()'s add_other(None) (:54-60)
                    ring_push<false>(q1, flushing & (match > 0), OP_M, match);
                    match = flushing ? 0 : match;
                    const bool oth = flushing & ev_other;
                    ring_push<true>(q1, oth, ev_t, ev_L);
                    read_head += (oth & b_read_cons(ev_t)) ? ev_L : 0;
                    ref_head += (oth & b_ref_cons(ev_t)) ? ev_L : 0;
                    a_k += oth ? 1 : 0;
                    const bool fin_now = flushing & ev_end;
                    finp = finp | fin_now;
                    live = live & !fin_now;
                }
                PLO_MARK("PIPE SHIFT STEP END");
            }
            pipe_refill_commit<NI>(rf, pm.in, n_in, in_hi);
        }
        // ---- what the step released: the header first (its lead shift is final: a match has been written), then the count ----
        {
            const bool rel_hdr = hdr_open & (q1.rel > hdr);
            if (rel_hdr) q1.b[((hdr + 2) & (N1 - 1)) * 64] = (uint32_t)q1.lead_shift;
            hdr_open = hdr_open & !rel_hdr;
            pipe_order();
            *pm.q1_rel = (uint32_t)q1.rel;
        }
        if (wv::ballot(!termd) == 0ull) break;
        streak = (mA | mR) != 0ull ? 0 : streak + 1;
        if ((mA | mR) == 0ull) pipe_idle(streak);
    }
}

// ====================================================================================================================================
// B: the LIFTOVER (src/liftover_read_alignment.rs:35-223), lane_tile's flat loop: one (op x block) piece or one copied op per step.
// HEAD: the team's first wave (forward class): items from the queue, ops from IN.  Else: items and ops from Q1.
// ====================================================================================================================================
template <bool SP, bool HEAD, int NI, int N1, int N2, int N3>
PLO_DEV void pipe_stage_liftover(const DevIndex &ix, const DevBatch &bt, const DevWork &wk, PipeQueue q, PipeMem<NI, N1, N2, N3> &pm, WaveCtx &ctx) {
    const int lane = wv::lane();
    const int n_cig_all = (int)bt.seg_cigar_off[bt.n_segs];
    bool live = false, termd = false, finp = false, drop = false;  // drop: Q2 could not hold the item's tail: its ops are skipped to its end
    int up_st = 0;                                                  // the status the item's end marker brought from the wave in front
    // HEAD: the input
    int n_in = 0, in_off = 0, in_hi = 0, b_k = 0;
    bool rev = false;
    // else: Q1's consumer side
    int rk1 = 0, rel1 = 0;
    int pos1 = 0, kv1 = 0, W0 = 0, W1 = 0, hdr = 0;
    bool hdr_open = false;
    int t_op = 0, seg_start = 0, seg_end = 0, block_pos = 0, r2s = 0, r2e = 0, kb = 0, vb = NONE32, kn = IMAX, vn = NONE32, kf = IMAX, vf = NONE32, ni = 0;
    bool in_op = false, ism_op = false, bvalid = false, has_start = false, has_end = false, kv_lds = false;
    int kvs_base = 0, kvs_cnt = 0;
    LaneRing<N2> q2;
    q2.b = pm.q2;
    int streak = 0;  // trips without a step (wave-uniform)
    // entry `idx` of the block map for the lanes `on`: from the staged copy, or (items outside it) from global memory
    auto kv_fetch = [&](bool on, int idx, int &key, int &val) {
        const bool l = on & kv_lds & ((unsigned)(idx - kvs_base) < (unsigned)kvs_cnt);
        const uint32_t *p = pm.kvs + 2 * (l ? idx - kvs_base : 0);
        const int lk = (int)p[0], lv = (int)p[1];
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
    for (;;) {
        q2.rk = (int)*pm.q2_rk;
        if constexpr (!HEAD) rel1 = (int)*pm.q1_rel;
        pipe_order();  // (the ring's words are read after the count that releases them)
        // ---- lanes between items: the next item's header (from the queue / from Q1), or the lane's last word ----
        {
            bool want = !live & !finp & !termd & (ring_room(q2) >= PIPE_H2);
            uint32_t first = 0;
            if constexpr (!HEAD) {
                first = pm.q1[((want ? rk1 : 0) & (N1 - 1)) * 64];
                want = want & (rel1 > rk1);
            }
            if (wv::ballot(want) != 0ull) {
                uint32_t at = 0;
                bool mine, last;
                int lead1 = 0;
                if constexpr (HEAD) {
                    mine = pipe_take(q, want, at);
                    last = want & !mine;
                } else {
                    // (a released header is released whole: SOI, position, lead shift)
                    last = want & (first == PIPE_TERM);
                    mine = want & !last;
                    at = pm.q1[((mine ? rk1 + 1 : 0) & (N1 - 1)) * 64];
                    lead1 = (int)pm.q1[((mine ? rk1 + 2 : 0) & (N1 - 1)) * 64];
                    rk1 += mine ? PIPE_H1 : (last ? 1 : 0);
                }
                ring_put(q2, last, PIPE_TERM);
                q2.rel = last ? q2.no : q2.rel;
                termd = termd | last;
                int kv0 = 0;
                if (mine) {
                    const uint32_t g = wk.perm[at];
                    W0 = (int)wk.d.w0[g];
                    W1 = (int)wk.d.w1[g];
                    kv0 = (int)wk.d.kv0[g];
                    kv1 = (int)wk.d.kv1[g];
                    pos1 = wk.d.pos1[g];
                    if constexpr (HEAD) {
                        in_off = (int)wk.d.in_off[g];
                        n_in = (int)wk.d.n_in[g];
                        rev = (wk.d.flags[g] & ITF_REV) != 0;
                    }
                    int nb = kv1 - kv0, lg = 0;
                    while ((1 << lg) < nb) ++lg;
                    ctx.algo_bytes += 16u * (unsigned)(W1 - W0) + 8u * (unsigned)lg;
                }
                ring_new_item(q2, mine);
                hdr = mine ? q2.no : hdr;
                ring_put(q2, mine, PIPE_SOI);
                ring_put(q2, mine, at);
                ring_put(q2, mine, 0u);  // (the lead shift and ref2_start_pos: filled in when the header is released)
                ring_put(q2, mine, 0u);
                q2.base = mine ? q2.no : q2.base;
                hdr_open = hdr_open | mine;
                live = live | mine;
                drop = drop & !mine;
                up_st = mine ? 0 : up_st;
                in_hi = mine ? 0 : in_hi;
                b_k = mine ? 0 : b_k;
                seg_start = mine ? wrap_add(pos1, lead1) : seg_start;  // left_shift_indels.rs:38 pos + ref_pos_shift
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
                // the block-map entries the live items' cursors read -> LDS (lane_tile: one coalesced load; here whenever items start)
                {
                    const int need_hi = wv::imin(kv1, W1 + 2);
                    kvs_base = -wv::reduce_max(live ? -W0 : -IMAX);
                    const int top = wv::reduce_max(live ? need_hi : 0);
                    kvs_cnt = wv::imax(0, wv::imin(top - kvs_base, LANE_KVS));
                    kv_lds = live & (need_hi <= kvs_base + kvs_cnt);
                    KV e0 = {0, 0}, e1 = {0, 0};
                    if (lane < kvs_cnt) e0 = ix.kv[kvs_base + lane];
                    if (lane + 64 < kvs_cnt) e1 = ix.kv[kvs_base + lane + 64];
                    wv::sync();
                    pm.kvs[2 * lane] = (uint32_t)e0.key;
                    pm.kvs[2 * lane + 1] = (uint32_t)e0.val;
                    pm.kvs[2 * (lane + 64)] = (uint32_t)e1.key;
                    pm.kvs[2 * (lane + 64) + 1] = (uint32_t)e1.val;
                    wv::sync();
                }
                // the cursor: the first two entries of the item's window
                kv_fetch(mine & (W0 < kv1), W0, kn, vn);
                kv_fetch(mine & (W0 + 1 < kv1), W0 + 1, kf, vf);
            }
        }
        // ---- an item's end: open run, trailing edge, header (if not out yet), end marker ----
        {
            const bool fe = finp & (ring_room(q2) >= STREAM_END_PUSH);
            if (wv::ballot(fe) != 0ull) {
                ring_finish(q2, fe);  // :219-220
                // the wave in front first (its objection came earlier in the reference's order), then :218 ref2_start_pos.map(...) on None
                const int st = (q2.ovf | (up_st == PIPE_ST_OVF)) ? PIPE_ST_OVF : (up_st != 0 ? up_st : (has_start ? 0 : (int)PLO_ITEM_NO_LIFTOVER));
                if (fe & hdr_open) {
                    q2.b[((hdr + 2) & (N2 - 1)) * 64] = (uint32_t)q2.lead_shift;
                    q2.b[((hdr + 3) & (N2 - 1)) * 64] = (uint32_t)r2s;
                }
                hdr_open = hdr_open & !fe;
                ring_put(q2, fe, PIPE_EOI | ((uint32_t)st << 8));
                q2.rel = fe ? q2.no : q2.rel;
                finp = finp & !fe;
            }
        }
        // ---- the step ----
        bool r_rdy = false, b_avail;
        if constexpr (HEAD) {
            r_rdy = live & (in_hi < n_in) & (in_hi - b_k <= NI - STREAM_REFILL);
            b_avail = (b_k < in_hi) | (b_k >= n_in);  // an op to fetch, or the input's end
        } else {
            b_avail = rk1 < rel1;
        }
        const bool room_ok = ring_room(q2) >= stream_b_push(N2);
        const bool b_rdy = live & (drop | room_ok) & (in_op | b_avail);
        const bool stuck = live & !drop & !room_ok & (q2.rk == q2.rel);
        if (wv::ballot(stuck) != 0ull) {
            q2.no = stuck ? wv::imax(q2.rel, q2.base) : q2.no;
            q2.in_pair = q2.in_pair & !stuck;
            q2.acc = stuck ? 0u : q2.acc;
            q2.ovf = q2.ovf | stuck;
            drop = drop | stuck;
            in_op = in_op & !stuck;
        }
        const unsigned long long mR = wv::ballot(r_rdy & (in_hi - b_k < STREAM_IN_LOW)) != 0ull ? wv::ballot(r_rdy) : 0ull;
        const unsigned long long mB = pipe_worth(wv::ballot(b_rdy), wv::ballot(live), streak) ? wv::ballot(b_rdy) : 0ull;
        if ((mB | mR) != 0ull) {
            PipeRefill rf;
            if constexpr (HEAD) pipe_refill_issue(rf, r_rdy & (mR != 0ull), bt, n_cig_all, in_off, n_in, rev, in_hi);
            if (mB != 0ull) {
              bool bm = b_rdy;
              for (int it = 0;; ++it) {
                if (it > 0) {  // another step of the burst: the lanes that are ready by this wave's own counts
                    bool av;
                    if constexpr (HEAD) av = (b_k < in_hi) | (b_k >= n_in);
                    else av = rk1 < rel1;
                    bm = live & (drop | (ring_room(q2) >= stream_b_push(N2))) & (in_op | av);
                    const unsigned long long m2 = wv::ballot(bm);
                    if (2 * __builtin_popcountll(m2) < __builtin_popcountll(mB) || m2 == 0ull) break;
                }
                ctx.u_act += (unsigned)__builtin_popcountll(wv::ballot(bm));
                ctx.u_trips += 1;
                PLO_MARK("PIPE LIFTOVER STEP BEGIN");
                const bool fetch = bm & !in_op;
                uint32_t c;
                bool b_end;
                if constexpr (HEAD) {
                    b_end = fetch & (b_k >= n_in);
                    c = pm.in[(((fetch & !b_end) ? b_k : 0) & (NI - 1)) * 64];
                    b_k += (fetch & !b_end) ? 1 : 0;
                } else {
                    c = pm.q1[((fetch ? rk1 : 0) & (N1 - 1)) * 64];
                    rk1 += fetch ? 1 : 0;
                    b_end = fetch & pipe_is_marker(c);  // (inside an item the only marker is its end)
                    up_st = b_end ? (int)((c >> 8) & 0xffu) : up_st;
                }
                const bool fop = fetch & !b_end & !drop;
                const int tf = op_type(c), Lf = op_len(c);
                const bool copy = fop & (((0x32u >> tf) & 1u) != 0u);  // I S H: :157-160 copied through; Pad (:213) emits nothing
                const bool start = fop & b_ref_cons(tf) & (Lf > 0);
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
                // ONE writer call per step, as in lane_tile (round 6): a piece that enters its block behind a jump of the reference takes two
                // steps -- the deletion D(d) first, which moves ref2_end_pos up to the block's start, then the piece itself with d == 0
                const bool e0 = mp & has_end & (d > 0) & has_start;
                has_end = has_end | mp;
                const bool go = piece & !e0;
                r2e = mp ? wrap_add(vb, e0 ? 0 : pend_ - kb) : r2e;  // :98-100
                // :102-109 mapped piece | :111-115 insertion over an unmapped block | :117-123 soft clip before the first block
                const bool e1p = go & (mapped ? (ism_op | has_start) : ism_op);
                const int t1p = mapped ? (t_op == OP_D ? (int)OP_D : (t_op == OP_N ? (int)OP_N : (int)OP_M)) : (bvalid ? (int)OP_I : (int)OP_S);
                block_pos = go ? pend_ : block_pos;
                const bool done = go & (pend_ >= seg_end);
                in_op = in_op & !done;
                seg_start = done ? seg_end : seg_start;
                ring_push<false, N2, stream_marks(N2)>(q2, e0 | copy | e1p, e0 ? (int)OP_D : (copy ? tf : t1p), e0 ? d : (copy ? Lf : plen));
                kf = adv ? fk : kf;  // the entry after next (kv_fetch above)
                vf = adv ? fv : vf;
                finp = finp | b_end;
                live = live & !b_end;
                PLO_MARK("PIPE LIFTOVER STEP END");
                if (it + 1 >= PIPE_BURST) break;
              }
            }
            if constexpr (HEAD) pipe_refill_commit<NI>(rf, pm.in, n_in, in_hi);
        }
        // ---- what the step released / consumed ----
        {
            const bool rel_hdr = hdr_open & (q2.rel > hdr);
            if (rel_hdr) {
                q2.b[((hdr + 2) & (N2 - 1)) * 64] = (uint32_t)q2.lead_shift;
                q2.b[((hdr + 3) & (N2 - 1)) * 64] = (uint32_t)r2s;
            }
            hdr_open = hdr_open & !rel_hdr;
            pipe_order();
            *pm.q2_rel = (uint32_t)q2.rel;
            if constexpr (!HEAD) *pm.q1_rk = (uint32_t)rk1;
        }
        if (wv::ballot(!termd) == 0ull) break;
        streak = (mB | mR) != 0ull ? 0 : streak + 1;
        if ((mB | mR) == 0ull) pipe_idle(streak);
    }
}

// ====================================================================================================================================
// C: SIMPLIFY (src/simplify_alignment_indels.rs:5-156), lane_tile's loop body: one op per step; the team's last wave: it gives the item
// its slot of the output buffer, flushes Q3 into it and leaves the item's results.  Items that fail the length check
// (src/read_alignment_scanner.rs:204-229) keep the liftover's CIGAR: their ops pass through.
// ====================================================================================================================================
template <bool SP, int NI, int N1, int N2, int N3>
PLO_DEV void pipe_stage_simplify(const DevBatch &bt, const DevWork &wk, PipeMem<NI, N1, N2, N3> &pm, WaveCtx &ctx) {
    const int lane = wv::lane();
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    bool live = false, termd = false, flushing = false;  // flushing: the item's ops are through, the rest of Q3 goes out, then its results
    uint32_t g = 0;
    int rk2 = 0, rel2 = 0;
    int n_in = 0, chrom_ref_len = 0, alloc = 0, pos0 = 0, status = 0, up_st = 0;
    bool passthru = false, ovf = false, fits = true;
    unsigned long long chrom_ref = 0, out_base = 0;
    ReadSeq rd = item_read_seq<SP>(bt, 0ull, 0, 0);
    int c_ref_head = 0, c_read_head = 0, c_del = 0, c_ins = 0, c_blk_ref = 0, c_blk_read = 0, cmp = 0;
    bool c_in_blk = false, spanic = false, zero_m = false;
    bool pair_open = false;  // a PIPE_PAIR marker has been taken and its cluster is not through yet: no copying ahead
    LaneRing<N3> q3;  // (both sides in this wave)
    q3.b = pm.q3;
    int flushed = 0;  // ops of the item written out; q3.rk = the start of the chunk that holds the next one (its ops go out again with it)
    int streak = 0;  // trips without a step (wave-uniform)
    for (;;) {
        rel2 = (int)*pm.q2_rel;
        pipe_order();
        // ---- lanes between items: the next item's header, or the lane's last word ----
        {
            const uint32_t first = pm.q2[((rk2) & (N2 - 1)) * 64];
            const bool want = !live & !flushing & !termd & (rel2 > rk2);
            if (wv::ballot(want) != 0ull) {
                const bool last = want & (first == PIPE_TERM);
                const bool mine = want & !last;
                rk2 += last ? 1 : 0;
                termd = termd | last;
                // (a released header is released whole: SOI, position, lead shift, ref2_start_pos)
                const uint32_t at = pm.q2[((mine ? rk2 + 1 : 0) & (N2 - 1)) * 64];
                const int lead2 = (int)pm.q2[((mine ? rk2 + 2 : 0) & (N2 - 1)) * 64];
                const int r2s = (int)pm.q2[((mine ? rk2 + 3 : 0) & (N2 - 1)) * 64];
                rk2 += mine ? PIPE_H2 : 0;
                int n_m = 0, w0 = 0, w1 = 0, seq_len = 0;
                bool flip = false;
                unsigned long long seq_off = 0;
                if (mine) {
                    g = wk.perm[at];
                    n_in = (int)wk.d.n_in[g];
                    n_m = (int)wk.d.n_m[g];
                    w0 = (int)wk.d.w0[g];
                    w1 = (int)wk.d.w1[g];
                    seq_len = (int)wk.d.seq_len[g];
                    const uint32_t read_len_in = wk.d.read_len[g];
                    passthru = read_len_in == 0xffffffffu || (uint32_t)seq_len != read_len_in;  // LENGTH CHECK, see lift_tile
                    seq_off = wk.d.seq_off[g];
                    chrom_ref = wk.d.chrom_ref[g];
                    chrom_ref_len = wk.d.chrom_ref_len[g];
                    flip = (wk.d.flags[g] & ITF_FLIP) != 0;
                    rd = item_read_seq<SP>(bt, seq_off, seq_len, flip ? 1 : 0);
                }
                // the item's slot of the output buffer: slabs as in lane_tile (one device-scope atomic per slab)
                const int want_ops = mine ? stream_out_alloc(n_m, w0, w1) : 0;
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
                if (!fit && total > 0 && lane == 0) wv::atomic_add_global(&wk.counters[CNT_OVERFLOW], 1ull);  // (the host runs the batch again)
                out_base = mine ? gbase + (unsigned long long)(inc - want_ops) : out_base;
                alloc = mine ? want_ops : alloc;
                fits = mine ? fit : fits;
                live = live | mine;
                ovf = ovf & !mine;
                up_st = mine ? 0 : up_st;
                status = mine ? (int)PLO_ITEM_LIFTED : status;
                pos0 = mine ? wrap_add(r2s, lead2) : pos0;  // :221 ref2_start_pos + the liftover's leading-edge shift
                c_ref_head = mine ? pos0 : c_ref_head;
                c_read_head = mine ? 0 : c_read_head;
                c_del = mine ? 0 : c_del;
                c_ins = mine ? 0 : c_ins;
                cmp = mine ? 0 : cmp;
                c_in_blk = c_in_blk & !mine;
                pair_open = pair_open & !mine;
                spanic = spanic & !mine;
                zero_m = zero_m & !mine;
                ring_new_item(q3, mine);
                // (Q3 holds one item at a time: its counts start over)
                q3.no = mine ? 0 : q3.no;
                q3.rel = mine ? 0 : q3.rel;
                q3.rk = mine ? 0 : q3.rk;
                q3.base = mine ? 0 : q3.base;
                flushed = mine ? 0 : flushed;
            }
        }
        // ---- the flush: eight ops of Q3 -> the item's slot (the ops behind `rel` in the chunk are not final: they are written again
        // by the next flush, which starts at the new `rk`) ----
        const int f_have = q3.rel - flushed;
        const bool q3_tight = ring_room(q3) < STREAM_C_PUSH;
        const bool f_rdy = (live | flushing) & !ovf & ((f_have >= STREAM_FLUSH) | ((f_have > 0) & (q3_tight | flushing)));
        const unsigned long long mF = wv::ballot(f_rdy);
        if (mF != 0ull) {
            // (whole chunks from a chunk's start: the lane's stores are aligned 64-byte lines; ops already out are written again with the rest)
            const int f0 = flushed & ~(STREAM_FLUSH - 1);
            const bool room_ok = f0 + STREAM_FLUSH <= alloc;
            ovf = ovf | (f_rdy & !room_ok);  // the output outgrew its slot (the region bound): retry list
            const bool go = f_rdy & room_ok & fits;
            if (go) {
                uint32_t *const dst = wk.out_cigar + out_base + (unsigned long long)f0;
#pragma unroll
                for (int qd = 0; qd < STREAM_FLUSH / 4; ++qd) {
                    Ops4 v;
                    v.x = q3.b[((f0 + 4 * qd) & (N3 - 1)) * 64];
                    v.y = q3.b[((f0 + 4 * qd + 1) & (N3 - 1)) * 64];
                    v.z = q3.b[((f0 + 4 * qd + 2) & (N3 - 1)) * 64];
                    v.w = q3.b[((f0 + 4 * qd + 3) & (N3 - 1)) * 64];
                    *(PLO_GLOBAL Ops4 *)(dst + 4 * qd) = v;
                }
            }
            flushed = (f_rdy & room_ok) ? wv::imin(f0 + STREAM_FLUSH, q3.rel) : flushed;
            q3.rk = flushed & ~(STREAM_FLUSH - 1);
        }
        // ---- an item's results, once all of it is out (or it has none) ----
        {
            const bool em = flushing & (ovf | (flushed >= q3.rel));
            if (wv::ballot(em) != 0ull) {
                const bool re = em & ovf;  // items a ring or the output slot could not hold: the wave-cooperative code takes them
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
                const bool done = em & !ovf;
                const bool emit_cigar = done & ((status == PLO_ITEM_LIFTED) | (status == PLO_ITEM_LEN_MISMATCH));
                const int oc = emit_cigar ? q3.no : 0;
                const int pos = wrap_add(pos0, passthru ? 0 : q3.lead_shift);  // simplify_alignment_indels.rs:155
                if (done) {
                    if (status == PLO_ITEM_NEED_BASES) wk.miss_list[wv::atomic_add_global(&wk.counters[CNT_NMISS], 1ull)] = g;  // rare
                    wk.status[g] = (uint8_t)status;
                    wk.pos[g] = emit_cigar ? (int64_t)pos : (int64_t)-1;
                    wk.cig_off[g] = emit_cigar ? out_base : 0ull;
                    wk.cig_len[g] = (uint32_t)oc;
                    ctx.algo_bytes += 2u * (unsigned)cmp + 40u + 4u * (unsigned)n_in + 24u + 4u * (unsigned)oc;
                    ctx.in_ops += (unsigned)n_in;
                    ctx.out_ops += (unsigned)oc;
                }
                flushing = flushing & !em;
            }
        }
        // ---- the stretches the stage cannot change (round 6, DESIGN.md 11.3 of round 5; off line 99.4 % of the stress profile's ops): up to
        // STREAM_BULK ops per lane go from Q2 to Q3 as they are -- no cluster state machine, no writer: the liftover's ring holds a cleaned,
        // compressed CIGAR, and on such a CIGAR the stage is the identity except at indel clusters of more than one op
        // (simplify_alignment_indels.rs:41-48), which the liftover has marked (PIPE_PAIR).  An op is taken while the entry BEHIND it is a
        // released op too: the op in front of a marker -- a match the cluster's M(pre) may merge into -- and the one in front of the item's end
        // are left to the step below, as is everything until the first match (the leading-edge rule) and while a cluster is open.
        unsigned long long mK = 0ull;
        if constexpr (STREAM_BULK > 0 && stream_marks(N2)) {
            const int avail = rel2 - rk2;
            const bool bk = live & !passthru & !ovf & q3.seen_m & !c_in_blk & !pair_open & (avail >= 3) & (ring_room(q3) >= STREAM_BULK + 1);
            const unsigned long long mb = wv::ballot(bk);
            if (mb != 0ull && 4 * __builtin_popcountll(mb) >= __builtin_popcountll(wv::ballot(live))) {
                mK = mb;
                uint32_t cw[STREAM_BULK + 1];
#pragma unroll
                for (int j = 0; j <= STREAM_BULK; ++j) cw[j] = pm.q2[((bk ? rk2 + j : 0) & (N2 - 1)) * 64];
                // nb: the longest run of ops from the front whose successor is a released op
                int nb = 0;
                bool run = bk;
#pragma unroll
                for (int j = 0; j < STREAM_BULK; ++j) {
                    run = run & (j + 1 < avail) & !pipe_is_marker(cw[j]) & !pipe_is_marker(cw[j + 1]);
                    nb += run ? 1 : 0;
                }
                const bool any = nb > 0;
                {   // the writer's open run goes out first (what follows it in a compressed CIGAR is of another kind)
                    const bool fl = any & (q3.acc >= 16u);
                    if (fl) q3.b[(q3.no & (N3 - 1)) * 64] = q3.acc;
                    q3.no += fl ? 1 : 0;
                    q3.rel = (fl & b_is_match((int)(q3.acc & 15u))) ? q3.no : q3.rel;
                    q3.acc = any ? 0u : q3.acc;
                }
#pragma unroll
                for (int j = 0; j < STREAM_BULK; ++j) {
                    const bool on = j < nb;
                    const uint32_t c = cw[j];
                    const int t = op_type(c), L = op_len(c);
                    if (on) q3.b[(q3.no & (N3 - 1)) * 64] = c;
                    q3.no += on ? 1 : 0;
                    q3.rel = (on & b_is_match(t)) ? q3.no : q3.rel;
                    c_read_head += (on & b_read_cons(t)) ? L : 0;
                    c_ref_head += (on & b_ref_cons(t)) ? L : 0;
                }
                rk2 += nb;
            }
        }
        // ---- the step ----
        const bool c_avail = rk2 < rel2;
        const bool room_ok = ring_room(q3) >= STREAM_C_PUSH;
        // (an item that has overflowed is read to its end without writing)
        const bool c_rdy = live & c_avail & (ovf | room_ok);
        // no room, nothing to flush: Q3 is full of the item's unreleased tail -> retry list
        ovf = ovf | (live & !room_ok & (q3.rel <= flushed));
        const unsigned long long mC = pipe_worth(wv::ballot(c_rdy), wv::ballot(live), streak) ? wv::ballot(c_rdy) : 0ull;
        if (mC != 0ull) {
          bool cm = c_rdy;
          for (int it = 0;; ++it) {
            if (it > 0) {  // another step of the burst
                cm = live & (rk2 < rel2) & (ovf | (ring_room(q3) >= STREAM_C_PUSH));
                const unsigned long long m2 = wv::ballot(cm);
                if (2 * __builtin_popcountll(m2) < __builtin_popcountll(mC) || m2 == 0ull) break;
            }
            ctx.u_act += (unsigned)__builtin_popcountll(wv::ballot(cm));
            ctx.u_trips += 1;
            PLO_MARK("PIPE SIMPLIFY STEP BEGIN");
            const uint8_t *const cref = (const uint8_t *)(uintptr_t)chrom_ref;
            const uint32_t c = pm.q2[((cm ? rk2 : 0) & (N2 - 1)) * 64];
            rk2 += cm ? 1 : 0;
            const bool is_pair = cm & (c == PIPE_PAIR);  // (the liftover's mark in front of a cluster of several ops: nothing to do for it here)
            const bool atend = cm & pipe_is_marker(c) & !is_pair;  // (the only other marker inside an item is its end)
            up_st = atend ? (int)((c >> 8) & 0xffu) : up_st;
            const bool valid = cm & !atend & !is_pair & !ovf;
            pair_open = pair_open | is_pair;
            const int t = op_type(c), L = op_len(c);
            const bool raw = valid & passthru;
            if (raw) q3.b[(q3.no & (N3 - 1)) * 64] = c;
            q3.no += raw ? 1 : 0;
            q3.rel = raw ? q3.no : q3.rel;
            // the end of an item the waves in front object to (or that has no CIGAR to simplify) emits nothing
            const bool sv = valid & !passthru, se = atend & !passthru & !ovf & (up_st == 0);
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
#ifndef PLO_EXP_CMP_NOMEM  // (timing experiment: what do the comparisons' round trips cost this wave?  results are wrong)
                            // :55-68 trailing bases shared by the inserted and the deleted sequence, then :71-85 leading ones
                            post = match_run_back(cref, chrom_ref_len, c_blk_ref + c_del, rd, c_blk_read + c_ins, wv::imin(c_del, c_ins), cmp);
                            c_del -= post;
                            c_ins -= post;
                            pre = match_run_fwd(cref, chrom_ref_len, c_blk_ref, rd, c_blk_read, wv::imin(c_del, c_ins), cmp);
                            c_del -= pre;
                            c_ins -= pre;
#endif
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
                pair_open = pair_open & !endc;
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
                q3.rel = atend ? q3.no : q3.rel;
                // statuses in the reference's order: what the waves in front object to, the length check, this stage
                const bool mine_bad = rd.miss | spanic;
                const int mine_st = passthru ? (int)PLO_ITEM_LEN_MISMATCH : (mine_bad ? (rd.miss ? (int)PLO_ITEM_NEED_BASES : (int)PLO_ITEM_PANIC) : (int)PLO_ITEM_LIFTED);
                status = atend ? ((up_st != 0 && up_st != PIPE_ST_OVF) ? up_st : mine_st) : status;
                ovf = ovf | (atend & ((up_st == PIPE_ST_OVF) | q3.ovf | zero_m));
                flushing = flushing | atend;
                live = live & !atend;
            }
            PLO_MARK("PIPE SIMPLIFY STEP END");
            if (it + 1 >= PIPE_BURST) break;
          }
        }
        pipe_order();
        *pm.q2_rk = (uint32_t)rk2;
        if (wv::ballot(!termd | flushing) == 0ull) break;
        streak = (mC | mF | mK) != 0ull ? 0 : streak + 1;
        if ((mC | mF | mK) == 0ull) pipe_idle(streak);
    }
}

// One team (workgroup of PIPE_WAVES waves) over the class-order positions [b, e) of one heavy class: `has_shift`: the reverse class
// (waves A, B, C); else the forward class (waves B, C; the third wave leaves at once).
template <bool SP, int NI, int N1, int N2, int N3>
PLO_DEV void pipe_team(const DevIndex &ix, const DevBatch &bt, const DevWork &wk, uint32_t b, uint32_t e, bool has_shift, uint32_t *lds, WaveCtx &ctx) {
    static_assert((NI & (NI - 1)) == 0 && (N1 & (N1 - 1)) == 0 && (N2 & (N2 - 1)) == 0 && (N3 & (N3 - 1)) == 0, "ring sizes are powers of two");
    static_assert(NI >= STREAM_REFILL + STREAM_A_MIN_IN, "IN: a refill fits while a step's worth of input is left");
    static_assert(N1 >= PIPE_H1 + STREAM_A_PUSH && N2 >= PIPE_H2 + stream_b_push(N2) + STREAM_END_PUSH && N3 >= STREAM_C_PUSH + STREAM_FLUSH + 4, "a fresh item's first step fits behind its header");
    const int role = wv::wave_id(), lane = wv::lane();
    PipeMem<NI, N1, N2, N3> pm(lds, lane);
    if (role == 0) {
        *pm.q1_rel = 0u;
        *pm.q1_rk = 0u;
        *pm.q2_rel = 0u;
        *pm.q2_rk = 0u;
    }
    wv::block_sync();
    PipeQueue q = {b, e};
    if (has_shift) {
        if (role == 0) pipe_stage_shift<SP, NI, N1, N2, N3>(bt, wk, q, pm, ctx);
        else if (role == 1) pipe_stage_liftover<SP, false, NI, N1, N2, N3>(ix, bt, wk, q, pm, ctx);
        else pipe_stage_simplify<SP, NI, N1, N2, N3>(bt, wk, pm, ctx);
    } else {
        if (role == 0) pipe_stage_liftover<SP, true, NI, N1, N2, N3>(ix, bt, wk, q, pm, ctx);
        else if (role == 1) pipe_stage_simplify<SP, NI, N1, N2, N3>(bt, wk, pm, ctx);
    }
}

// Team `t` of the launch: teams [0, t0) share the forward class [lo, mid) equally, teams [t0, t0 + t1) the reverse class [mid, hi) --
// contiguous shares: a team's items are neighbours on a contig and share block-map and reference lines.
PLO_DEV void pipe_team_span(uint32_t t, uint32_t t0, uint32_t t1, uint32_t lo, uint32_t mid, uint32_t hi, uint32_t &b, uint32_t &e, bool &has_shift) {
    has_shift = t >= t0;
    const unsigned long long n = has_shift ? hi - mid : mid - lo, k = has_shift ? t - t0 : t, nt = has_shift ? t1 : t0;
    const uint32_t base = has_shift ? mid : lo;
    b = base + (uint32_t)(n * k / (nt ? nt : 1ull));
    e = base + (uint32_t)(n * (k + 1ull) / (nt ? nt : 1ull));
}

}  // namespace plo
