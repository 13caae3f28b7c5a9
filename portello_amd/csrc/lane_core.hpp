// lane_core.hpp -- short-CIGAR fast path: one LANE per item, 64 items per wave.
//
// HiFi read->contig CIGARs have ~30 ops; for them the scan formulation of lift_core.hpp spends ~35 wave-instructions
// per op on cross-lane machinery.  Here every lane walks its own item sequentially (the reference's loops, unchanged)
// over lane-interleaved LDS arrays (element k of lane l at [k*64 + l]: the 64 lanes of a wave touch 64 consecutive
// banks when they are at the same k).  Cross-lane work is limited to output allocation (one add-scan per wave).
// Items whose intermediates do not fit LANE_CAP ops are re-queued for the wave-cooperative tile path.
//
// All citations are relative to /root/reference.
#pragma once
#include <plo_wave.hpp>
#include <stdint.h>

#include "lift_core.hpp"

namespace plo {

constexpr int LANE_CAP = 64;  // ops per item per buffer
constexpr int LANE_KV = 8;    // block-map entries staged per item

struct LaneMem {
    uint32_t *X, *Y;  // [LANE_CAP * 64] ping-pong op arrays, lane-interleaved
    int *K, *V;       // [LANE_KV * 64] staged block-map window
};
constexpr size_t lane_mem_bytes() { return (size_t)64 * (2 * LANE_CAP * 4 + 2 * LANE_KV * 4); }
PLO_DEV LaneMem carve_lane_mem(unsigned char *base) {
    LaneMem m;
    m.X = (uint32_t *)base;
    m.Y = m.X + LANE_CAP * 64;
    m.K = (int *)(m.Y + LANE_CAP * 64);
    m.V = m.K + LANE_KV * 64;
    return m;
}

struct LaneBuf {  // one lane's view of an interleaved array
    uint32_t *p;
    PLO_DEV uint32_t get(int k) const { return p[k * 64]; }
    PLO_DEV void set(int k, uint32_t v) const { p[k * 64] = v; }
};

// clean_up_cigar_edge_indels + compress_cigar (lib/rust-vc-utils/src/bam_utils/cigar/mod.rs:265-291, 204-228),
// sequential, `out` may alias `in` (the write index never passes the read index).  Returns the new length.
PLO_DEV int lane_cleanup_compress(LaneBuf in, int n, LaneBuf out, int &shift) {
    int first = n, last = -1;
    for (int k = 0; k < n; ++k)
        if (is_match(op_type(in.get(k)))) {
            if (first == n) first = k;
            last = k;
        }
    shift = 0;
    int no = 0;
    uint32_t acc = mk_op(OP_M, 0);  // last_elem = Cigar::Match(0) (:206)
    for (int k = 0; k < n; ++k) {
        uint32_t c = in.get(k);
        int t = op_type(c);
        if (k < first || k > last) {  // edges (:278-288)
            if (t == OP_D) {
                if (k < first) shift += op_len(c);
                c = mk_op(OP_S, 0);
            } else if (t == OP_I) {
                c = mk_op(OP_S, op_len(c));
            }
            t = op_type(c);
        }
        int L = op_len(c);
        if (L == 0) continue;  // .filter(|x| !x.is_empty())
        if (t == op_type(acc)) {
            if (t != OP_P) acc = mk_op(t, op_len(acc) + L);  // Pad is absent from the summing pattern (:210-212)
        } else {
            if (op_len(acc) != 0) out.set(no++, acc);
            acc = c;
        }
    }
    if (op_len(acc) != 0) out.set(no++, acc);
    return no;
}

PLO_DEV void lift_lanes(const DevIndex &ix, const DevBatch &bt, const DevWork &wk, uint32_t stages, uint32_t item_begin, int nit,
                        LaneMem m, WaveCtx &ctx) {
    const int lane = wv::lane();
    const bool has = lane < nit;
    const uint32_t g = has ? wk.perm[item_begin + (uint32_t)lane] : 0u;
    LaneBuf X{m.X + lane}, Y{m.Y + lane};
    int *Kl = m.K + lane, *Vl = m.V + lane;

    int n_in = 0, in_off = 0, pos1 = 0, kv1 = 0, W0 = 0, W1 = 0, seq_len = 0, shift_ref_len = 0, chrom_ref_len = 0;
    unsigned long long seq_off = 0, shift_ref = 0, chrom_ref = 0;
    bool rev = false, do_shift = false, flip = false;
    if (has) {
        in_off = (int)wk.d.in_off[g];
        n_in = (int)wk.d.n_in[g];
        pos1 = wk.d.pos1[g];
        W0 = (int)wk.d.w0[g];
        W1 = (int)wk.d.w1[g];
        kv1 = (int)wk.d.kv1[g];
        uint32_t fl = wk.d.flags[g];
        seq_len = (int)wk.d.seq_len[g];
        seq_off = wk.d.seq_off[g];
        shift_ref = wk.d.shift_ref[g];
        shift_ref_len = wk.d.shift_ref_len[g];
        chrom_ref = wk.d.chrom_ref[g];
        chrom_ref_len = wk.d.chrom_ref_len[g];
        rev = (fl & ITF_REV) != 0;
        flip = (fl & ITF_FLIP) != 0;
        do_shift = (stages & PLO_STAGE_LSHIFT) && (!(stages & PLO_STAGE_STRAND) || !(fl & ITF_CONTIG_FWD));
    }
    const int kv0 = has ? (int)wk.d.kv0[g] : 0;
    int status = PLO_ITEM_LIFTED;
    bool alive = has, ovf = false, panic = false;
    unsigned long long algo = 0;
    if (has && n_in > LANE_CAP) ovf = true;
    if (has && do_shift && shift_ref == 0ull) {  // rev_contig_seq.unwrap() on None (read_alignment_scanner.rs:174)
        status = PLO_ITEM_PANIC;
        alive = false;
        do_shift = false;
    }
    // ---- stage the block-map window and the input CIGAR (reversed for reverse-mapped contig segments, :167) ----
    const int nk = (has && (stages & PLO_STAGE_LIFTOVER)) ? (wv::imin(W1 + 1, kv1) - W0) : 0;
    for (int k = 0; k < LANE_KV; ++k)
        if (k < nk) {
            KV e = ix.kv[W0 + k];
            Kl[k * 64] = e.key;
            Vl[k * 64] = e.val;
        }
    int n = (has && !ovf) ? n_in : 0;
    for (int k0 = 0; k0 < n; k0 += 8) {  // 8 independent loads in flight per lane, then the LDS stores
        uint32_t r[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            int k = k0 + j;
            r[j] = k < n ? bt.cigar[in_off + (rev ? (n - 1 - k) : k)] : 0u;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (k0 + j < n) X.set(k0 + j, r[j]);
    }
    ReadSeq rd = item_read_seq(bt, seq_off, seq_len, flip);
    LaneBuf cur = X, oth = Y;  // `cur` holds the item's current CIGAR (n ops)

#define LANE_PUSH(buf, cnt, val)              \
    {                                         \
        if ((cnt) < LANE_CAP) (buf).set((cnt), (val)); \
        else ovf = true;                      \
        ++(cnt);                              \
    }

    // ---- left_shift_indels (shift_indels/left_shift_indels.rs:17-39, cigar_indel_shifter.rs:10-165) ----------------
    if (alive && !ovf && do_shift) {
        const uint8_t *ref = (const uint8_t *)(uintptr_t)shift_ref;
        int ref_head = pos1, read_head = 0, match = 0, del = 0, ins = 0, blk_ref = 0, blk_read = 0, no = 0;
        bool in_blk = false;
        for (int k = 0; k <= n; ++k) {
            uint32_t c = k < n ? cur.get(k) : 0u;
            int t = op_type(c), L = op_len(c);
            bool indel = k < n && is_indel(t);
            if (indel) {
                if (L > 0) {  // add_del / add_ins (:73-85)
                    if (!in_blk) {
                        in_blk = true;
                        blk_ref = ref_head;
                        blk_read = read_head;
                    }
                    if (t == OP_D) del += L; else ins += L;
                }
            } else {
                if (in_blk) {  // end_indel (:101-148)
                    in_blk = false;
                    int probes = 0;
                    int h = left_homology(ref, shift_ref_len, blk_ref, del, rd, blk_read, ins, match, panic, probes);
                    algo += 2ull * (unsigned)probes;
                    int sh = wv::imin(match, h);
                    if (match - sh > 0) LANE_PUSH(oth, no, mk_op(OP_M, match - sh))
                    match = sh;
                    if (ins > 0) LANE_PUSH(oth, no, mk_op(OP_I, ins))
                    if (del > 0) LANE_PUSH(oth, no, mk_op(OP_D, del))
                    ins = 0;
                    del = 0;
                }
                if (k < n && is_match(t)) {
                    match += L;  // add_match (:150-153)
                } else {         // add_other (:155-165); k == n is get_cigar()'s add_other(None)
                    if (match > 0) LANE_PUSH(oth, no, mk_op(OP_M, match))
                    match = 0;
                    if (k < n) LANE_PUSH(oth, no, c)
                }
            }
            if (k < n) {
                if (read_consuming(t)) read_head += L;
                if (ref_consuming(t)) ref_head += L;
            }
        }
        if (panic) {
            status = PLO_ITEM_PANIC;
            alive = false;
        } else if (!ovf) {
            int shift = 0;
            n = lane_cleanup_compress(oth, no, oth, shift);
            pos1 += shift;
            LaneBuf t2 = cur;
            cur = oth;
            oth = t2;
        }
    }

    // ---- liftover_read_alignment (src/liftover_read_alignment.rs:35-223) ----------------------------------------------
    if (alive && !ovf && (stages & PLO_STAGE_LIFTOVER)) {
        int nb = kv1 - kv0, lg = 0;
        while ((1 << lg) < nb) ++lg;
        algo += 16ull * (unsigned)(W1 - W0) + 8ull * (unsigned)lg;
        const int nw = W1 - W0;  // blocks of the window; entry nw (if staged) only bounds the last block
        auto key = [&](int j) { return j < LANE_KV ? Kl[j * 64] : ix.kv[W0 + j].key; };
        auto val = [&](int j) { return j < LANE_KV ? Vl[j * 64] : ix.kv[W0 + j].val; };
        bool has_start = false, has_end = false;
        int r2s = 0, r2e = 0, no = 0;
        int seg_start = pos1;
        int b = -1;  // window-relative index of the block containing seg_start (-1: before the first block of the map)
        for (int k = 0; k < n; ++k) {
            uint32_t c = cur.get(k);
            int t = op_type(c), L = op_len(c);
            if (t == OP_I || t == OP_S || t == OP_H) {
                LANE_PUSH(oth, no, c)  // :157-160
            } else if (ref_consuming(t)) {
                int seg_end = seg_start + L;
                if (L > 0) {
                    while (b + 1 < nw && key(b + 1) <= seg_start) ++b;  // greatest key <= seg_start (get_ref_range :79-82)
                    int block_pos = seg_start;
                    bool ism = is_match(t);
                    for (;;) {
                        int pend = seg_end;
                        if (W0 + b + 1 < kv1) {
                            int kn = key(b + 1);
                            if (kn < pend) pend = kn;
                        }
                        int plen = pend - block_pos;
                        if (b < 0) {  // :117-123 (W0 == kv0 here: no block at or before the op)
                            if (ism) LANE_PUSH(oth, no, mk_op(OP_S, plen))
                        } else {
                            int bv = val(b);
                            if (bv == NONE32) {  // :111-115
                                if (ism) LANE_PUSH(oth, no, mk_op(OP_I, plen))
                            } else {
                                int bk = key(b);
                                if (ism && !has_start) {  // :84-88
                                    has_start = true;
                                    r2s = bv + (block_pos - bk);
                                }
                                if (has_end) {  // :91-96
                                    int d = bv - r2e;
                                    if (d > 0 && has_start) LANE_PUSH(oth, no, mk_op(OP_D, d))
                                }
                                has_end = true;
                                r2e = bv + (pend - bk);  // :98-100
                                if (ism || has_start) LANE_PUSH(oth, no, mk_op(t == OP_D ? OP_D : (t == OP_N ? OP_N : OP_M), plen))
                            }
                        }
                        block_pos = pend;
                        if (pend >= seg_end) break;
                        ++b;
                    }
                }
                seg_start = seg_end;
            }  // Pad: nothing (:213)
        }
        if (!has_start) {  // :218
            status = PLO_ITEM_NO_LIFTOVER;
            alive = false;
        } else if (!ovf) {
            int shift = 0;
            n = lane_cleanup_compress(oth, no, oth, shift);  // :219-220
            pos1 = r2s + shift;                              // :221
            LaneBuf t2 = cur;
            cur = oth;
            oth = t2;
        }
    }

    // ---- length check (src/read_alignment_scanner.rs:204-229) ----------------------------------------------------------
    bool simp = alive && !ovf;
    if (alive && !ovf && (stages & PLO_STAGE_LENCHECK)) {
        int rl = 0;
        for (int k = 0; k < n; ++k) {
            uint32_t c = cur.get(k);
            if (read_consuming(op_type(c))) rl += op_len(c);
        }
        if (rl != seq_len) {
            status = PLO_ITEM_LEN_MISMATCH;
            simp = false;
        }
    }

    // ---- simplify_alignment_indels (src/simplify_alignment_indels.rs:5-156) ----------------------------------------------
    if (simp && (stages & PLO_STAGE_SIMPLIFY)) {
        const uint8_t *ref = (const uint8_t *)(uintptr_t)chrom_ref;
        int ref_head = pos1, read_head = 0, del = 0, ins = 0, blk_ref = 0, blk_read = 0, no = 0;
        bool in_blk = false;
        for (int k = 0; k <= n; ++k) {
            uint32_t c = k < n ? cur.get(k) : 0u;
            int t = op_type(c), L = op_len(c);
            if (k < n && is_indel(t)) {
                if (!in_blk) {  // _add_indel (:16-22)
                    in_blk = true;
                    blk_ref = ref_head;
                    blk_read = read_head;
                }
                if (t == OP_D) del += L; else ins += L;
            } else {
                if (in_blk) {  // end_indel (:35-111)
                    in_blk = false;
                    if (del == 0 && ins == 0) {
                    } else if (del == 0) {
                        LANE_PUSH(oth, no, mk_op(OP_I, ins))
                    } else if (ins == 0) {
                        LANE_PUSH(oth, no, mk_op(OP_D, del))
                    } else if (del == 1 && ins == 1) {
                        LANE_PUSH(oth, no, mk_op(OP_M, 1))
                    } else if (blk_ref < 0 || blk_ref + del - 1 >= chrom_ref_len || blk_read + ins - 1 >= rd.len) {
                        panic = true;  // slice index out of bounds (:58-60)
                    } else {
                        int pre = 0, post = 0, cmp = 0;
                        while (del > 0 && ins > 0) {  // :55-68
                            ++cmp;
                            if (ref[blk_ref + del - 1] != read_base(rd, blk_read + ins - 1)) break;
                            --del;
                            --ins;
                            ++post;
                        }
                        while (del > 0 && ins > 0) {  // :71-85
                            ++cmp;
                            if (ref[blk_ref + pre] != read_base(rd, blk_read + pre)) break;
                            --del;
                            --ins;
                            ++pre;
                        }
                        if (del == 1 && ins == 1) {  // :88-92
                            del = 0;
                            ins = 0;
                            ++post;
                        }
                        algo += 2ull * (unsigned)cmp;
                        if (pre > 0) LANE_PUSH(oth, no, mk_op(OP_M, pre))  // :101-104
                        if (ins > 0) LANE_PUSH(oth, no, mk_op(OP_I, ins))
                        if (del > 0) LANE_PUSH(oth, no, mk_op(OP_D, del))
                        if (post > 0) LANE_PUSH(oth, no, mk_op(OP_M, post))
                    }
                    del = 0;
                    ins = 0;
                }
                if (k < n) LANE_PUSH(oth, no, c)  // :144-147
            }
            if (k < n) {
                if (read_consuming(t)) read_head += L;
                if (ref_consuming(t)) ref_head += L;
            }
        }
        if (panic) {
            status = PLO_ITEM_PANIC;
            alive = false;
        } else if (!ovf) {
            int shift = 0;
            n = lane_cleanup_compress(oth, no, oth, shift);  // :153-154
            pos1 += shift;                                   // :155
            LaneBuf t2 = cur;
            cur = oth;
            oth = t2;
        }
    }
#undef LANE_PUSH

    // ---- output ---------------------------------------------------------------------------------------------------------------
    // items whose intermediates did not fit LANE_CAP go to the wave-cooperative tile path (retry list)
    {
        unsigned long long om = wv::ballot(has && ovf);
        if (om != 0ull) {
            int slot = 0;
            if (lane == 0) slot = (int)wv::atomic_add_global(&wk.counters[CNT_NRETRY], (unsigned long long)__builtin_popcountll(om));
            slot = wv::bcast_first(slot);
            if (has && ovf) wk.retry_list[slot + __builtin_popcountll(om & ((1ull << lane) - 1ull))] = g;
        }
    }
    bool emit_cigar = has && !ovf && (status == PLO_ITEM_LIFTED || status == PLO_ITEM_LEN_MISMATCH);
    int oc = emit_cigar ? n : 0;
    int inco = wv::scan_add(oc);
    int oS = inco - oc;
    int total = wv::bcast_last(inco);
    if ((unsigned long long)total > ctx.slab_left) {
        unsigned long long want = (unsigned long long)total > SLAB_OPS ? (unsigned long long)total : SLAB_OPS;
        unsigned long long nb = 0;
        if (lane == 0) nb = wv::atomic_add_global(&wk.counters[CNT_CIGAR], want) + wk.slab_offset;
        unsigned lo = (unsigned)wv::bcast_first((int)(unsigned)(nb & 0xffffffffull));
        unsigned hi = (unsigned)wv::bcast_first((int)(unsigned)(nb >> 32));
        ctx.slab_base = ((unsigned long long)hi << 32) | lo;
        ctx.slab_left = want;
    }
    const unsigned long long gbase = ctx.slab_base;
    ctx.slab_base += (unsigned long long)total;
    ctx.slab_left -= (unsigned long long)total;
    bool fits = gbase + (unsigned long long)total <= wk.out_cap;
    if (!fits && lane == 0) wv::atomic_add_global(&wk.counters[CNT_OVERFLOW], 1ull);
    if (fits)
        for (int k = 0; k < oc; ++k) wk.out_cigar[gbase + (unsigned long long)(oS + k)] = cur.get(k);
    if (has && !ovf) {
        wk.status[g] = (uint8_t)status;
        wk.pos[g] = emit_cigar ? (int64_t)pos1 : (int64_t)-1;
        wk.cig_off[g] = emit_cigar ? gbase + (unsigned long long)oS : 0ull;
        wk.cig_len[g] = (uint32_t)oc;
        ctx.algo_bytes += algo + 40ull + 4ull * (unsigned)n_in + 24ull + 4ull * (unsigned)oc;
        ctx.in_ops += (unsigned long long)n_in;
        ctx.out_ops += (unsigned long long)oc;
    }
}

}  // namespace plo
