// engine.hip -- gfx950 kernels + the C ABI of include/portello_liftover.h.
//
// Kernel pipeline of one batch (all on the context's stream):
//   k_seg_count          two lanes per read split segment, 16 ops per lane and step: batch validation, reference span, read length, op count
//                        with match runs merged, how many contig segments it touches (a8)
//   scan                 item offsets, n_items (host sync: buffer sizes)
//   k_item_emit          thread per segment: resolved item descriptors (strand glue a9, block-map window a3), item class + weight
//   k_cls_hist / k_cls_scan / k_permute2   class order (strand x light / heavy): groups and tiles are strand-homogeneous; the host reads
//                        the class counts and the weight sum / maximum while k_permute2 runs
//   k_chunk_sort         light items by weight inside windows of 128 positions of the class order (groups of similar length)
//   k_lift_lanes         DOMINANT KERNEL on HiFi batches (lane_core.hpp): persistent waves, 64 light items per wave (32 / 16 / 8 when the
//                        batch has few), one LANE per item, the whole shift / liftover / length check / simplify pipeline in place in a
//                        ~180-byte LDS region per item
//   k_lift_lanes_g       heavy items of batches with many of them (dominant on the stress workload): the same lane-per-item code, the
//                        regions in wave-private global scratch behind per-lane LDS windows (lane_core.hpp, LaneWin)
//   (heavy items of smaller batches)   scan of the weights, k_max_u32 (weight histogram -> tile geometry, host sync), k_tile_bounds,
//   k_lift_tiles         persistent waves, one wave per tile of items, the pipeline as wave scans over a flattened op stream in LDS
//   k_lift_retry         items whose LDS region / slice overflowed (lanes or tiles): one item per wave, larger slice
//   k_lift_mid           items heavier than the routing threshold: one WORKGROUP per item (8 or 16 waves share the item's op
//                        stream in LDS, scan carries cross the waves through LDS: Coop<NW>, lift_core.hpp)
//   k_lift_big           items too heavy even for that: one wave per item, wave-private global scratch
//   k_sum_stats          per-wave statistic slots -> batch counters (after the lift kernels)
//   k_compact_cigar      dense re-packing of the slab-allocated output CIGARs (plo_compact_output_dev)
//   k_finish_* / k_revcomp / k_sa_*   record finishing and SA text (plo_finish_batch_dev, plo_sa_segments_dev)
//   k_map_build          block maps of the contig segments, once per index (plo_index_create)
// There is no CPU path: every entry point fails with PLO_ERR_NO_DEVICE / PLO_ERR_HIP when the device is unusable.
#include <hip/hip_runtime.h>

#include <plo_wave.hpp>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <chrono>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "enumerate.hpp"
#include "finish_core.hpp"
#include "index_pack.hpp"
#include "inflate.hpp"
#include "lift_core.hpp"
#include "lane_core.hpp"
#include "lane_stream.hpp"

using namespace plo;

// ---------------------------------------------------------------------------------------------------------------------
// kernels
// ---------------------------------------------------------------------------------------------------------------------

// SEG_LANES lanes per read segment: the reference span (get_cigar_ref_offset) is a strided partial sum per lane + an
// xor-shuffle reduction, so that one load instruction covers SEG_LANES * 4 contiguous bytes of every segment's CIGAR;
// lane 0 of the group then does the overlap test against the contig's segments.
#ifndef PLO_SEG_LANES
#define PLO_SEG_LANES 2  // measured on MI355X (wgs30x enumerate pass): round 1, strided dword loads: 8 lanes 0.64 ms, 16 lanes 1.52 ms, 32 lanes 0.99 ms;
                         // round 2: 8 lanes 0.539 ms, 4 lanes 0.485 ms, 2 lanes 0.486 ms; round 3, every lane 32 / SEG_LANES consecutive ops per step
                         // with 16-byte loads: 8 lanes 0.497 ms, 4 lanes 0.432 ms, 2 lanes 0.402 ms, 1 lane 0.411 ms
#endif
constexpr uint32_t SEG_LANES = PLO_SEG_LANES, SEG_UNROLL = 32 / SEG_LANES < 2 ? 2 : 32 / SEG_LANES;
// It is also the boundary check of the device path (the kernels index with what the batch says): bit 0 of *err = an index
// outside its array (PLO_ERR_INVALID_ARG), bit 1 = a coordinate outside the 31-bit BAM range, an op code above 8 or a CIGAR
// spanning more than 2^30 bases (PLO_ERR_RANGE) -- what the reference's types rule out by construction.
enum { VERR_INDEX = 1u, VERR_RANGE = 2u, VERR_CAP = 4u };
__global__ __launch_bounds__(256) void k_seg_count(DevIndex ix, DevBatch bt, uint32_t *seg_cnt, int *seg_reflen, uint32_t *seg_readlen,
                                                   uint32_t *seg_nm, uint32_t *err) {
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t s = t / SEG_LANES, sub = t % SEG_LANES;
    const bool live = s < bt.n_segs;
    uint32_t bad = 0;
    // Two dependent chains, issued side by side so that the kernel is three memory round trips deep instead of five:
    //   CIGAR offsets -> ops (reference span)   and   contig -> its segment range -> segment intervals (overlap test,
    //   one contig segment per lane of the group)
    uint32_t c0 = 0, c1 = 0, contig = 0, rd = 0;
    long long r_start = 0;
    if (live) {  // level 0: the segment's own fields
        c0 = bt.seg_cigar_off[s];
        c1 = bt.seg_cigar_off[s + 1];
        contig = bt.seg_contig[s];
        r_start = (long long)bt.seg_pos[s];
        rd = bt.seg_read[s];
        if (c1 < c0 || c1 > 0x7fffffffu) {  // (ops are indexed with int: more than 2^31 - 1 of them cannot be addressed)
            bad |= c1 < c0 ? VERR_INDEX : VERR_RANGE;
            c1 = c0;
        }
        if (r_start < 0 || r_start > 0x7ffffff0LL) bad |= VERR_RANGE;
        if (contig >= ix.n_contigs) bad |= VERR_INDEX;
        if (rd >= bt.n_reads) bad |= VERR_INDEX;
    }
    // level 1, all of it unconditional (lanes without a valid index read plo_safe_words): the read's length and offset, the contig's segment
    // range -- and, in the loop below, the ops
    const bool rd_ok = live && sub == 0 && rd < bt.n_reads, ct_ok = live && contig < ix.n_contigs;
    const unsigned long long len = *(rd_ok ? bt.read_seq_len + rd : (const uint32_t *)plo_safe_words);
    const unsigned long long off = *(rd_ok ? bt.read_seq_off + rd : (const uint64_t *)plo_safe_words);
    const uint32_t *const gp = ct_ok ? ix.contig_seg_off + contig : (const uint32_t *)plo_safe_words;
    const uint32_t g0 = gp[0], g1 = gp[1];  // g0 == g1: contig never seen in the asm->ref BAM (contig_alignment_scanner/mod.rs:364-367), or no contig
    long long part = 0;
    unsigned long long rpart = 0;  // read bases the CIGAR consumes (M I S H = X): what the length check compares with seq_len
    // Every lane of the group takes eight CONSECUTIVE ops per step (two 16-byte loads; the group's four lanes 128 contiguous bytes), so
    // that "is the previous op an alignment match too" -- the merged op count the lane-per-item kernel sizes its LDS regions by --
    // needs one shuffle per step: the last op of the lane below, or of the group's last lane in the previous step.
    uint32_t pairs = 0, carry = 0;
    constexpr uint32_t OPL = 32 / SEG_LANES;  // (a multiple of 4: 16-byte loads)
    const uint32_t n_ops_all = bt.n_segs ? bt.seg_cigar_off[bt.n_segs] : 0u;
    for (uint32_t base = c0; base < c1; base += SEG_LANES * OPL) {
        const uint32_t my = base + OPL * sub;
        uint32_t c[OPL];
        if (my < c1 && my + OPL <= n_ops_all) {  // (ops behind the segment's end belong to the next segment: masked by `have` below)
#pragma unroll
            for (uint32_t q = 0; q < OPL / 4; ++q) {
                const Ops4 a = *(const Ops4 *)(bt.cigar + my + 4 * q);
                c[4 * q] = a.x, c[4 * q + 1] = a.y, c[4 * q + 2] = a.z, c[4 * q + 3] = a.w;
            }
        } else {
#pragma unroll
            for (uint32_t u = 0; u < OPL; ++u) c[u] = (my + u < c1) ? bt.cigar[my + u] : 0xfu;  // (15: no op code, counted nowhere below)
        }
        uint32_t mprev = 0, mfirst = 0;
#pragma unroll
        for (uint32_t u = 0; u < OPL; ++u) {
            const uint32_t t = c[u] & 15u;
            const bool have = my + u < c1;
            if (have && ((0x18D >> t) & 1)) part += (long long)(c[u] >> 4);
            if (have && ((0x1B3 >> t) & 1)) rpart += (unsigned long long)(c[u] >> 4);
            if (have && t > 8u) bad |= VERR_RANGE;
            const uint32_t m = (have && ((0x181u >> t) & 1u)) ? 1u : 0u;
            if (u == 0) mfirst = m;
            else pairs += m & mprev;
            mprev = m;
        }
        const uint32_t below = (uint32_t)__shfl_up((int)mprev, 1, 64);
        pairs += mfirst & (sub ? below : carry);
        carry = (uint32_t)__shfl((int)mprev, (int)((threadIdx.x & 63u) | (SEG_LANES - 1)), 64);
    }
    // the first interval of every lane is fetched before the reduction needs the ops
    const uint32_t gl = g0 + sub;
    long long cs = 0, ce = 0;
    if (gl < g1) {
        cs = (long long)ix.cs_start[gl];
        ce = (long long)ix.cs_end[gl];
    }
#pragma unroll
    for (uint32_t d = 1; d < SEG_LANES; d <<= 1) {
        part += __shfl_xor(part, (int)d, 64);
        rpart += (unsigned long long)__shfl_xor((long long)rpart, (int)d, 64);
        pairs += (uint32_t)__shfl_xor((int)pairs, (int)d, 64);
    }
    if (part > 0x3fffffffLL) bad |= VERR_RANGE;
    if (rd_ok) {
        // (sparse bases: the header must lie inside the buffer; where it points is checked by every probe)
        const unsigned long long need = bt.seq_fmt == PLO_SEQ_BAM4          ? (len + 1) / 2
                                        : bt.seq_fmt == PLO_SEQ_BAM4_SPARSE ? (unsigned long long)sparse_header_bytes((uint32_t)len)
                                                                            : len;
        if (off > bt.seq_bytes || need > bt.seq_bytes - off) bad |= VERR_INDEX;
        if (bt.seq_fmt == PLO_SEQ_BAM4_SPARSE && (off & 15ull)) bad |= VERR_INDEX;
    }
    if (bad) atomicOr(err, bad);
    const long long r_end = r_start + part;
    // segment_range.intersect_range(&read_range): other.end >= self.start && other.start < self.end (int_range.rs:56-58)
    uint32_t n = (gl < g1 && r_end >= cs && r_start < ce) ? 1u : 0u;
    for (uint32_t g = gl + SEG_LANES; g < g1; g += SEG_LANES)  // contigs with more segments than lanes
        if (r_end >= (long long)ix.cs_start[g] && r_start < (long long)ix.cs_end[g]) ++n;
#pragma unroll
    for (uint32_t d = 1; d < SEG_LANES; d <<= 1) {
        n += (uint32_t)__shfl_xor((int)n, (int)d, 64);
        bad |= (uint32_t)__shfl_xor((int)bad, (int)d, 64);
    }
    if (!live || sub != 0) return;
    if (bad) n = 0;  // (a segment that fails the checks has no items: the one-round-trip path launches the lift kernels before the host sees the flag)
    seg_reflen[s] = (int)part;
    seg_readlen[s] = rpart > 0xfffffffeull ? 0xffffffffu : (uint32_t)rpart;
    seg_nm[s] = (c1 - c0) - pairs;
    seg_cnt[s] = n;
}

// thread per read segment: resolve the descriptors of its items (build_item_desc) at their scanned offsets
// `item_cap` (the one-round-trip path, liftover_fast): the item arrays hold that many items; a batch with more raises VERR_CAP and writes
// none beyond them (the host then takes the path that asks for the count first)
__global__ void k_item_emit(DevIndex ix, DevBatch bt, DevWork wk, uint32_t stages, const uint32_t *seg_off, int *seg_reflen, uint32_t item_cap, uint32_t *err) {
    uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= bt.n_segs) return;
    // (level 0 of the descriptor loads goes out with the offsets: enumerate.hpp)
    const uint32_t o0 = seg_off[s], o1 = seg_off[s + 1];
    if (o1 > item_cap) {
        if (o1 > o0) atomicOr(err, (uint32_t)VERR_CAP);
        return;
    }
    const int ref_len = seg_reflen[s];
    SegInfo si;
    seg_info_level0(si, bt, wk, stages, s);
    if (o0 == o1) return;  // (no items -- among them the segments whose contig the index does not know: nothing below is indexed by it)
    seg_info_level1(si, ix, bt, wk, stages, s);
    emit_segment_items(ix, wk, stages, s, o0, (long long)ref_len, si);
}

// Block maps of the contig segments, built on the device at index creation: ONE WAVE per contig split segment walks its contig->reference
// CIGAR 64 ops at a time (build_segment_map_wave, enumerate.hpp: prefix sums for the positions, a max-scan for the start of every match
// run, a prefix sum of 1 or 2 entries per flush); count pass, host prefix sum (a few thousand segments), emit pass.  (Round 4: one THREAD
// per segment, 1 806 threads walking ~50 k ops each -- 37.8 ms per pass on wgs30x.)
__global__ __launch_bounds__(64) void k_map_build(const uint32_t *seg_cigar, const uint32_t *seg_cigar_off, const int64_t *seg_pos, uint32_t n_segments,
                                                  const uint32_t *kv_off, KV *kv, int *counts) {
    for (uint32_t g = blockIdx.x; g < n_segments; g += gridDim.x) {
        const uint32_t c0 = seg_cigar_off[g], c1 = seg_cigar_off[g + 1];
        const int cnt = build_segment_map_wave(seg_cigar + c0, c1 - c0, (long long)seg_pos[g], kv ? kv + kv_off[g] : nullptr);
        if (counts && threadIdx.x == 0) counts[g] = cnt;
    }
}

// explicit item list: thread per item
__global__ void k_item_desc(DevIndex ix, DevBatch bt, DevWork wk, uint32_t stages, const uint32_t *in_seg, const uint32_t *in_cseg,
                            uint32_t *err) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= wk.n_items) return;
    uint32_t seg = in_seg[i];
    {   // the pair must exist: a read segment of the batch and a contig segment of ITS contig
        bool ok = seg < bt.n_segs;
        if (ok) {
            uint32_t contig = bt.seg_contig[seg];
            ok = contig < ix.n_contigs && in_cseg[i] < ix.contig_seg_off[contig + 1] - ix.contig_seg_off[contig];
        }
        if (!ok) {  // the host discards the batch before any lift kernel runs; the scans in between only need a class and a weight
            atomicOr(err, (uint32_t)VERR_INDEX);
            wk.item_cls[i] = 2u;
            wk.item_nin[i] = 0u;
            return;
        }
    }
    build_item_desc(ix, bt, wk, stages, i, seg, in_cseg[i], segment_ref_len(bt, seg));
}

// ---- device-wide exclusive scan of uint32 (three launches; out has n+1 entries, out[n] = total) -------------------
constexpr int SCAN_THREADS = 256;
constexpr int SCAN_PER_THREAD = 8;
constexpr int SCAN_BLOCK = SCAN_THREADS * SCAN_PER_THREAD;

__device__ __forceinline__ unsigned block_scan_incl(unsigned v, unsigned *wave_tot /*[4]*/, unsigned &block_total) {
    int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    unsigned inc = (unsigned)wv::scan_add((int)v);
    if (lane == 63) wave_tot[w] = inc;
    __syncthreads();
    unsigned base = 0;
    for (int k = 0; k < w; ++k) base += wave_tot[k];
    block_total = wave_tot[0] + wave_tot[1] + wave_tot[2] + wave_tot[3];
    __syncthreads();
    return inc + base;
}

// Class order (lift_types.hpp DevWork) in three launches: per block of CLS_BLOCK items the counts of classes 0, 1, 2 (k_cls_hist,
// which also reduces the weights the host sizes buffers by), an exclusive scan of the block counts (k_cls_scan, one workgroup),
// and the permutation itself from block offset + rank inside the block (k_permute2).
constexpr uint32_t CLS_THREADS = 256, CLS_PER = 8, CLS_BLOCK = CLS_THREADS * CLS_PER;
// partial: [6][nb] per block: items of class 0 / 1 / 2, heaviest item, sum of the weights (exact, 64 bits as two words: the batch-size
// guard and the output sizing rest on it); totals (k_cls_scan): [0..2] class counts, [3] heaviest item, [4..5] 64-bit sum of the weights
// the CLS_PER = 8 consecutive values of a thread: two 16-byte loads (the arrays come from hipMalloc and base is a multiple of 8), element by element
// with `fill` behind the end only in the array's last block
__device__ __forceinline__ void load8(const uint32_t *a, uint32_t base, uint32_t n, uint32_t fill, uint32_t (&v)[8]) {
    if (base + 8u <= n) {
        const uint4 x = *(const uint4 *)(a + base), y = *(const uint4 *)(a + base + 4);
        v[0] = x.x, v[1] = x.y, v[2] = x.z, v[3] = x.w, v[4] = y.x, v[5] = y.y, v[6] = y.z, v[7] = y.w;
    } else {
#pragma unroll
        for (uint32_t k = 0; k < 8; ++k) v[k] = base + k < n ? a[base + k] : fill;
    }
}
// (`n_dev` != NULL: the item count is read from device memory -- the one-round-trip path launches with the arrays' capacity `n`; a count above
// it means k_item_emit raised VERR_CAP and the arrays hold a mixture of this batch's and the last one's descriptors: the class totals then
// come out zero, so that k_permute2, k_chunk_sort_w and the lift kernels touch nothing before the host sees the flag and takes the other path)
__global__ __launch_bounds__(CLS_THREADS) void k_cls_hist(const uint32_t *item_cls, const uint32_t *item_w, uint32_t n, uint32_t nb, uint32_t *partial, const uint32_t *n_dev) {
    __shared__ uint32_t acc[6];
    if (n_dev) n = *n_dev <= n ? *n_dev : 0u;  // (more items than the arrays hold, VERR_CAP: k_item_emit left stale descriptors -- nothing is classified, nothing lifted)
    if (threadIdx.x < 6) acc[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t base = blockIdx.x * CLS_BLOCK + threadIdx.x * CLS_PER;
    uint32_t c0 = 0, c1 = 0, c2 = 0, mw = 0, sl = 0, sh = 0;  // sl / sh: sums of the weights' low / high 16 bits (a block's fit 32 bits each)
    uint32_t cv[CLS_PER], wv_[CLS_PER];
    load8(item_cls, base, n, 3u, cv);
    load8(item_w, base, n, 0u, wv_);
#pragma unroll
    for (uint32_t k = 0; k < CLS_PER; ++k) {
        const uint32_t c = cv[k], w = wv_[k];  // (behind the end: class 3 counted nowhere, weight 0)
        c0 += c == 0;
        c1 += c == 1;
        c2 += c == 2;
        mw = w > mw ? w : mw;
        sl += w & 0xffffu;
        sh += w >> 16;
    }
    // (counts of a wave fit 16 bits: 64 threads x 8 items)
    const int p01 = wv::reduce_add((int)(c0 | (c1 << 16))), p2 = wv::reduce_add((int)c2), pm = wv::reduce_max((int)(mw & 0x7fffffffu));
    const int psl = wv::reduce_add((int)sl), psh = wv::reduce_add((int)sh);
    if ((threadIdx.x & 63u) == 0) {
        atomicAdd(&acc[0], (uint32_t)p01 & 0xffffu);
        atomicAdd(&acc[1], (uint32_t)p01 >> 16);
        atomicAdd(&acc[2], (uint32_t)p2);
        atomicMax(&acc[3], (uint32_t)pm);
        atomicAdd(&acc[4], (uint32_t)psl);
        atomicAdd(&acc[5], (uint32_t)psh);
    }
    __syncthreads();
    if (threadIdx.x < 6) partial[threadIdx.x * nb + blockIdx.x] = acc[threadIdx.x];
}
__global__ __launch_bounds__(SCAN_THREADS) void k_cls_scan(uint32_t *partial, uint32_t nb, uint32_t *totals) {
    __shared__ unsigned wt[4];
    // one pass over the blocks, SCAN_THREADS at a time: the six rows of a step are loaded together (one round trip per step, not one per
    // row and step), the three class rows scanned, the other three reduced
    unsigned carry[3] = {0, 0, 0};
    unsigned mx = 0;
    unsigned long long sum = 0;
    for (uint32_t b0 = 0; b0 < nb; b0 += SCAN_THREADS) {
        const uint32_t i = b0 + threadIdx.x;
        const bool ok = i < nb;
        const uint32_t j = ok ? i : 0u;  // (nb > 0: the kernel is launched for batches with items)
        unsigned v[6];
#pragma unroll
        for (uint32_t c = 0; c < 6; ++c) v[c] = partial[c * nb + j];
#pragma unroll
        for (uint32_t c = 0; c < 3; ++c) {
            const unsigned x = ok ? v[c] : 0u;
            unsigned tot;
            const unsigned inc = block_scan_incl(x, wt, tot);
            if (ok) partial[c * nb + i] = carry[c] + inc - x;
            carry[c] += tot;
        }
        if (ok) {
            mx = v[3] > mx ? v[3] : mx;
            sum += (unsigned long long)v[4] + ((unsigned long long)v[5] << 16);
        }
    }
    // heaviest item and the sum of the weights
    __shared__ unsigned long long ssum;
    __shared__ unsigned smax;
    if (threadIdx.x == 0) {
        ssum = 0;
        smax = 0;
    }
    __syncthreads();
    mx = (unsigned)wv::reduce_max((int)(mx & 0x7fffffffu));  // (weights are below 2^31: checked sums of CIGAR ops and block-map entries)
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) sum += (unsigned long long)__shfl_xor((long long)sum, d, 64);
    if ((threadIdx.x & 63u) == 0) {
        atomicMax(&smax, mx);
        atomicAdd(&ssum, sum);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        totals[0] = carry[0];
        totals[1] = carry[1];
        totals[2] = carry[2];
        totals[3] = smax;
        totals[4] = (uint32_t)(ssum & 0xffffffffull);
        totals[5] = (uint32_t)(ssum >> 32);
    }
}
__global__ __launch_bounds__(CLS_THREADS) void k_permute2(const uint32_t *item_cls, const uint32_t *item_nin, const uint32_t *partial, const uint32_t *totals,
                                                          uint32_t n, uint32_t nb, uint32_t *perm, uint32_t *nin_p, uint32_t huge_w, const uint32_t *n_dev) {
    __shared__ unsigned wt[4];
    if (n_dev) n = *n_dev <= n ? *n_dev : 0u;
    const uint32_t base = blockIdx.x * CLS_BLOCK + threadIdx.x * CLS_PER;
    // every load in front of the scans and of the stores (a load behind a store waits for the store: the memory counter is in order)
    uint32_t cls[CLS_PER], wgt[CLS_PER];
    load8(item_cls, base, n, 3u, cls);
    load8(item_nin, base, n, 0u, wgt);
    const uint32_t p0 = partial[0 * nb + blockIdx.x], p1 = partial[1 * nb + blockIdx.x], p2 = partial[2 * nb + blockIdx.x];
    const uint32_t n0 = totals[0], n1 = totals[1], n2 = totals[2];
    uint32_t c0 = 0, c1 = 0, c2 = 0;
#pragma unroll
    for (uint32_t k = 0; k < CLS_PER; ++k) {
        c0 += cls[k] == 0;
        c1 += cls[k] == 1;
        c2 += cls[k] == 2;
    }
    unsigned tot;
    const unsigned i01 = block_scan_incl(c0 | (c1 << 16), wt, tot);  // (a block holds 2 048 items: each count fits 16 bits)
    const unsigned i2 = block_scan_incl(c2, wt, tot);
    uint32_t r0 = p0 + (i01 & 0xffffu) - c0;
    uint32_t r1 = p1 + (i01 >> 16) - c1;
    uint32_t r2 = p2 + i2 - c2;
#pragma unroll
    for (uint32_t k = 0; k < CLS_PER; ++k) {
        const uint32_t i = base + k;
        if (i >= n) break;
        const uint32_t c = cls[k];
        const uint32_t j = class_order_pos(i, c, r0, r1, r2, n0, n1, n2);
        perm[j] = i;
        // only the large items are tiled; items no geometry can hold (heavier than huge_w) take no room in the weight stream: they
        // ride along in their neighbours' tiles, 64 per pass, and are handed to the large-item kernel there
        const uint32_t w = wgt[k];
        nin_p[j] = (c >= 2 && w <= huge_w) ? w : 0u;
        r0 += c == 0;
        r1 += c == 1;
        r2 += c == 2;
    }
}

// k_cls_scan and k_permute2 in one launch (the one-round-trip path: every launch of an enumerate pass costs ~5 us of dispatch latency, as
// much as these kernels' work on a 50 k-read window): every block adds up k_cls_hist's raw rows itself -- the blocks in front of it for its
// offsets, all of them for the class totals -- and block 0 leaves the totals ([0..5] as k_cls_scan, [6] = the item count as the scan left it).
__global__ __launch_bounds__(CLS_THREADS) void k_permute2_s(const uint32_t *item_cls, const uint32_t *item_nin, const uint32_t *partial, uint32_t *totals, uint32_t n,
                                                            uint32_t nb, uint32_t *perm, uint32_t *nin_p, uint32_t huge_w, const uint32_t *n_dev) {
    __shared__ unsigned wt[4];
    __shared__ uint32_t red[6];
    __shared__ unsigned long long rsum;
    __shared__ unsigned rmax;
    const uint32_t n_raw = *n_dev;
    n = n_raw <= n ? n_raw : 0u;
    const uint32_t base = blockIdx.x * CLS_BLOCK + threadIdx.x * CLS_PER;
    uint32_t cls[CLS_PER], wgt[CLS_PER];
    load8(item_cls, base, n, 3u, cls);
    load8(item_nin, base, n, 0u, wgt);
    if (threadIdx.x < 6) red[threadIdx.x] = 0;
    if (threadIdx.x == 0) {
        rsum = 0;
        rmax = 0;
    }
    __syncthreads();
    {
        uint32_t t0 = 0, t1 = 0, t2 = 0, q0 = 0, q1 = 0, q2 = 0;
        unsigned mx = 0;
        unsigned long long sum = 0;
        for (uint32_t b = threadIdx.x; b < nb; b += CLS_THREADS) {
            const uint32_t v0 = partial[b], v1 = partial[nb + b], v2 = partial[2 * nb + b];
            t0 += v0, t1 += v1, t2 += v2;
            if (b < blockIdx.x) q0 += v0, q1 += v1, q2 += v2;
            if (blockIdx.x == 0) {
                const unsigned m = partial[3 * nb + b];
                mx = m > mx ? m : mx;
                sum += (unsigned long long)partial[4 * nb + b] + ((unsigned long long)partial[5 * nb + b] << 16);
            }
        }
        const uint32_t r[6] = {(uint32_t)wv::reduce_add((int)t0), (uint32_t)wv::reduce_add((int)t1), (uint32_t)wv::reduce_add((int)t2),
                               (uint32_t)wv::reduce_add((int)q0), (uint32_t)wv::reduce_add((int)q1), (uint32_t)wv::reduce_add((int)q2)};
        if (blockIdx.x == 0) {
            mx = (unsigned)wv::reduce_max((int)(mx & 0x7fffffffu));
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) sum += (unsigned long long)__shfl_xor((long long)sum, d, 64);
        }
        if ((threadIdx.x & 63u) == 0) {
#pragma unroll
            for (int k = 0; k < 6; ++k)
                if (r[k]) atomicAdd(&red[k], r[k]);
            if (blockIdx.x == 0) {
                atomicMax(&rmax, mx);
                atomicAdd(&rsum, sum);
            }
        }
    }
    __syncthreads();
    const uint32_t n0 = red[0], n1 = red[1], n2 = red[2], p0 = red[3], p1 = red[4], p2 = red[5];
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        totals[0] = n0;
        totals[1] = n1;
        totals[2] = n2;
        totals[3] = rmax;
        totals[4] = (uint32_t)(rsum & 0xffffffffull);
        totals[5] = (uint32_t)(rsum >> 32);
        totals[6] = n_raw;
    }
    uint32_t c0 = 0, c1 = 0, c2 = 0;
#pragma unroll
    for (uint32_t k = 0; k < CLS_PER; ++k) {
        c0 += cls[k] == 0;
        c1 += cls[k] == 1;
        c2 += cls[k] == 2;
    }
    unsigned tot;
    const unsigned i01 = block_scan_incl(c0 | (c1 << 16), wt, tot);
    const unsigned i2 = block_scan_incl(c2, wt, tot);
    uint32_t r0 = p0 + (i01 & 0xffffu) - c0;
    uint32_t r1 = p1 + (i01 >> 16) - c1;
    uint32_t r2 = p2 + i2 - c2;
#pragma unroll
    for (uint32_t k = 0; k < CLS_PER; ++k) {
        const uint32_t i = base + k;
        if (i >= n) break;
        const uint32_t c = cls[k];
        const uint32_t j = class_order_pos(i, c, r0, r1, r2, n0, n1, n2);
        perm[j] = i;
        const uint32_t w = wgt[k];
        nin_p[j] = (c >= 2 && w <= huge_w) ? w : 0u;
        r0 += c == 0;
        r1 += c == 1;
        r2 += c == 2;
    }
}

// The lane-per-item kernel runs a group of 64 items for as long as its longest item takes -- in batch order about twice the average.
// Inside windows of LANE_SORT_WINDOW consecutive positions of the class order (two groups: the reads of a window still share their
// descriptor / CIGAR / block-map cache lines) the items are therefore sorted by weight: a group of the longer and a group of the
// shorter ones.  One workgroup per window, counting sort in LDS.
constexpr uint32_t LANE_SORT_WINDOW = 128;
constexpr uint32_t LANE_SORT_MAX_CHUNK = 2048, LANE_SORT_THREADS = 256, LANE_SORT_PER = LANE_SORT_MAX_CHUNK / LANE_SORT_THREADS;
// `glist` != NULL: the sorted window is also cut into the lane kernel's groups by LDS budget (lane_groups_cut, enumerate.hpp: consecutive
// items while there are fewer than 64 and their regions fit `cap` dwords) and the groups are appended to the list ({first position, count};
// *n_groups zeroed by the host).  A window sorted by weight has its long items together; as fixed groups of 64 those need more LDS than a
// wave's slice and run in several rounds -- which is what made windows larger than 128 slower although they level the lanes better.
// Sort key: what the group's loops run for -- the merged op count n_m (= / X ops, which the tiling weight counts, are merged away before any
// loop sees them) plus the block-map entries of the item's window (the liftover loop's extra pieces); in the class with the shift stage n_m
// counts twice (its rounds and scans follow the indel clusters, half of n_m, and cost twice a liftover iteration).  Off line, windows of 128,
// against the tiling weight: shift rounds per group 1.67 -> 1.52 x the mean, liftover iterations 1.59 -> 1.49 (forward class) / 1.56.
__global__ __launch_bounds__(LANE_SORT_THREADS) void k_chunk_sort(uint32_t *perm, const uint32_t *n_m, const uint32_t *w0, const uint32_t *w1, uint32_t n0, uint32_t n1,
                                                                  uint32_t chunk, const uint32_t *region, uint32_t cap, uint32_t *glist, uint32_t *n_groups, uint32_t glist_cap) {
    __shared__ uint32_t hist[256];
    __shared__ uint32_t wsum[4];
    __shared__ uint16_t reg[LANE_SORT_MAX_CHUNK], nxt[LANE_SORT_MAX_CHUNK];
    __shared__ uint32_t pre[LANE_SORT_MAX_CHUNK + 1];
    __shared__ uint32_t starts[LANE_SORT_MAX_CHUNK + 1], g_base, g_cnt;
    const uint32_t c0 = (n0 + chunk - 1) / chunk;
    uint32_t lo, hi;
    if (blockIdx.x < c0) {  // chunks do not straddle the two lane classes
        lo = blockIdx.x * chunk;
        hi = lo + chunk < n0 ? lo + chunk : n0;
    } else {
        lo = n0 + (blockIdx.x - c0) * chunk;
        hi = lo + chunk < n0 + n1 ? lo + chunk : n0 + n1;
    }
    hist[threadIdx.x] = 0;
    __syncthreads();
    uint32_t g[LANE_SORT_PER], k[LANE_SORT_PER];
#pragma unroll
    for (uint32_t j = 0; j < LANE_SORT_PER; ++j) {
        const uint32_t p = lo + j * LANE_SORT_THREADS + threadIdx.x;
        g[j] = p < hi ? perm[p] : 0u;
    }
#pragma unroll
    for (uint32_t j = 0; j < LANE_SORT_PER; ++j) {
        const uint32_t p = lo + j * LANE_SORT_THREADS + threadIdx.x;
        uint32_t w = 0u;
        if (p < hi) w = (blockIdx.x < c0 ? n_m[g[j]] : 2u * n_m[g[j]]) + (w1[g[j]] - w0[g[j]]);
        k[j] = w < 255u ? w : 255u;
        if (p < hi) atomicAdd(&hist[k[j]], 1u);
    }
    __syncthreads();
    // exclusive prefix of the 256 bins: thread t owns bin t
    const uint32_t mine = hist[threadIdx.x];
    const uint32_t inc = (uint32_t)wv::scan_add((int)mine);
    if ((threadIdx.x & 63u) == 63u) wsum[threadIdx.x >> 6] = inc;
    __syncthreads();
    uint32_t before = 0;
    for (uint32_t w = 0; w < (threadIdx.x >> 6); ++w) before += wsum[w];
    hist[threadIdx.x] = before + inc - mine;
    __syncthreads();
#pragma unroll
    for (uint32_t j = 0; j < LANE_SORT_PER; ++j) {
        const uint32_t p = lo + j * LANE_SORT_THREADS + threadIdx.x;
        if (p < hi) {
            const uint32_t pos = atomicAdd(&hist[k[j]], 1u);
            perm[lo + pos] = g[j];
            if (glist) reg[pos] = (uint16_t)(region[g[j]] < 0xffffu ? region[g[j]] : 0xffffu);
        }
    }
    if (!glist) return;
    const uint32_t n = hi - lo;
    __syncthreads();
    {   // exclusive prefix of the regions in sorted order: thread t owns positions [t * PER, t * PER + PER)
        uint32_t loc[LANE_SORT_PER], tot = 0;
#pragma unroll
        for (uint32_t j = 0; j < LANE_SORT_PER; ++j) {
            const uint32_t i = threadIdx.x * LANE_SORT_PER + j;
            loc[j] = tot;
            tot += i < n ? (uint32_t)reg[i] : 0u;
        }
        const uint32_t inc2 = (uint32_t)wv::scan_add((int)tot);
        __syncthreads();  // (wsum is reused)
        if ((threadIdx.x & 63u) == 63u) wsum[threadIdx.x >> 6] = inc2;
        __syncthreads();
        uint32_t base = inc2 - tot;
        for (uint32_t w = 0; w < (threadIdx.x >> 6); ++w) base += wsum[w];
#pragma unroll
        for (uint32_t j = 0; j < LANE_SORT_PER; ++j) {
            const uint32_t i = threadIdx.x * LANE_SORT_PER + j;
            if (i < n) pre[i] = base + loc[j];
        }
        if (threadIdx.x == LANE_SORT_THREADS - 1) pre[n] = base + tot;  // (positions beyond n add nothing)
    }
    __syncthreads();
    // the group that starts at position i ends at the largest j <= i + 64 with pre[j] - pre[i] <= cap (at least one item)
#pragma unroll
    for (uint32_t jj = 0; jj < LANE_SORT_PER; ++jj) {
        const uint32_t i = jj * LANE_SORT_THREADS + threadIdx.x;
        if (i < n) {
            uint32_t a = i, b = i + 64u < n ? i + 64u : n;  // invariant: a fits, everything above b does not (or is out of range)
            while (a < b) {
                const uint32_t m = (a + b + 1u) >> 1;
                if (pre[m] - pre[i] <= cap) a = m;
                else b = m - 1u;
            }
            nxt[i] = (uint16_t)(a > i ? a : i + 1u);
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t ng = 0;
        for (uint32_t i = 0; i < n; i = nxt[i]) starts[ng++] = i;
        starts[ng] = n;
        g_cnt = ng;
        g_base = atomicAdd(n_groups, ng);
    }
    __syncthreads();
    // (the host sizes the list for the most groups the windows can have, a group being cut at 64 items or at the slice's capacity; a
    // group beyond it would be a sizing bug -- it is not written, and the batch's count then exceeds the capacity, which the host checks)
    for (uint32_t q = threadIdx.x; q < g_cnt; q += LANE_SORT_THREADS) {
        if (g_base + q >= glist_cap) continue;
        glist[2 * (g_base + q)] = lo + starts[q];
        glist[2 * (g_base + q) + 1] = starts[q + 1] - starts[q];
    }
}

// The same sort for the default geometry -- windows of at most 128 positions, fixed groups: one WAVE per window (two items per lane, four bins
// per lane), four windows per workgroup, no workgroup barrier and 4 KB of LDS (k_chunk_sort: a workgroup and 25 KB per window).
__global__ __launch_bounds__(256) void k_chunk_sort_w(uint32_t *perm, const uint32_t *n_m, const uint32_t *w0, const uint32_t *w1, uint32_t n0, uint32_t n1,
                                                      uint32_t chunk, const uint32_t *totals_dev) {
    __shared__ uint32_t hist_all[4][256];
    if (totals_dev) {  // (the class counts from device memory: k_cls_scan's totals)
        n0 = totals_dev[0];
        n1 = totals_dev[1];
    }
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    uint32_t *const hist = hist_all[wave];
    const uint32_t c0 = (n0 + chunk - 1) / chunk, c1 = (n1 + chunk - 1) / chunk;
    const uint32_t win = blockIdx.x * 4u + wave;
    if (win >= c0 + c1) return;
    uint32_t lo, hi;
    if (win < c0) {  // windows do not straddle the two lane classes
        lo = win * chunk;
        hi = lo + chunk < n0 ? lo + chunk : n0;
    } else {
        lo = n0 + (win - c0) * chunk;
        hi = lo + chunk < n0 + n1 ? lo + chunk : n0 + n1;
    }
#pragma unroll
    for (uint32_t q = 0; q < 4; ++q) hist[lane + 64u * q] = 0;
    uint32_t g[2], k[2];
#pragma unroll
    for (uint32_t j = 0; j < 2; ++j) {
        const uint32_t p = lo + j * 64u + lane;
        g[j] = p < hi ? perm[p] : 0u;
    }
    uint32_t a[2], b0[2], b1[2];
#pragma unroll
    for (uint32_t j = 0; j < 2; ++j) {  // (six gathers in flight)
        a[j] = n_m[g[j]];
        b0[j] = w0[g[j]];
        b1[j] = w1[g[j]];
    }
    wv::sync();
#pragma unroll
    for (uint32_t j = 0; j < 2; ++j) {
        const uint32_t p = lo + j * 64u + lane;
        const uint32_t w = (win < c0 ? a[j] : 2u * a[j]) + (b1[j] - b0[j]);
        k[j] = w < 255u ? w : 255u;
        if (p < hi) atomicAdd(&hist[k[j]], 1u);
    }
    wv::sync();
    {   // exclusive prefix of the 256 bins: lane t owns bins 4 t .. 4 t + 3
        uint32_t h[4], tot = 0;
#pragma unroll
        for (uint32_t q = 0; q < 4; ++q) {
            h[q] = hist[4u * lane + q];
            tot += h[q];
        }
        uint32_t base = (uint32_t)wv::scan_add((int)tot) - tot;
#pragma unroll
        for (uint32_t q = 0; q < 4; ++q) {
            hist[4u * lane + q] = base;
            base += h[q];
        }
    }
    wv::sync();
#pragma unroll
    for (uint32_t j = 0; j < 2; ++j) {
        const uint32_t p = lo + j * 64u + lane;
        if (p < hi) perm[lo + atomicAdd(&hist[k[j]], 1u)] = g[j];
    }
}

// thread per tile: first class-order position (>= n_small) whose exclusive op prefix reaches the tile's window
__global__ void k_tile_bounds(const uint32_t *op_prefix, uint32_t n_items, uint32_t n_tiles, int window, uint32_t n_small, uint32_t *tile_lo) {
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t > n_tiles) return;
    uint32_t lo = prefix_lower_bound(op_prefix, n_items, (unsigned long long)t * (unsigned)window);
    tile_lo[t] = lo > n_small ? lo : n_small;
}

// maximum (out[0]) and 64-bit sum (out[2..3]) of a uint32 array, one atomic pair per wave, plus a histogram of the values
// in bins of WHIST_STEP (out[WHIST_AT + b], last bin = everything above); out must be zeroed
constexpr int WHIST_STEP = 32, WHIST_BINS = 40, WHIST_AT = 8;
// out[4..6] = *t0, *t1, *t2 (scan totals the host wants in the same copy)
__global__ __launch_bounds__(256) void k_max_u32(const uint32_t *in, uint32_t n, uint32_t *out, const uint32_t *t0, const uint32_t *t1,
                                                 const uint32_t *t2) {
    __shared__ uint32_t hist[WHIST_BINS];
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        out[4] = *t0;
        out[5] = *t1;
        out[6] = *t2;
    }
    if (threadIdx.x < WHIST_BINS) hist[threadIdx.x] = 0;
    __syncthreads();
    uint32_t m = 0;
    unsigned long long sum = 0;
    // (the weights of a batch crowd into a few bins: one LDS atomic per distinct bin of a wave's 64 items, not one per item)
    const uint32_t n_up = (n + 63u) & ~63u;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_up; i += gridDim.x * blockDim.x) {
        const bool live = i < n;
        uint32_t v = live ? in[i] : 0u;
        m = v > m ? v : m;
        sum += v;
        uint32_t b = v / WHIST_STEP;
        b = b < WHIST_BINS - 1 ? b : WHIST_BINS - 1;
        unsigned long long todo = __ballot(live);
        while (todo) {
            const int leader = __ffsll((long long)todo) - 1;
            const uint32_t lb = (uint32_t)__shfl((int)b, leader, 64);
            const unsigned long long same = __ballot(live && b == lb);
            if ((int)(threadIdx.x & 63) == leader) atomicAdd(&hist[lb], (uint32_t)__popcll(same));
            todo &= ~same;
        }
    }
    __syncthreads();
    if (threadIdx.x < WHIST_BINS && hist[threadIdx.x]) atomicAdd(out + WHIST_AT + threadIdx.x, hist[threadIdx.x]);
    int r = wv::reduce_max((int)(m & 0x7fffffffu));
    unsigned lo = (unsigned)wv::reduce_add((int)(unsigned)(sum & 0xffffffull));
    unsigned hi = (unsigned)wv::reduce_add((int)(unsigned)(sum >> 24));
    // one global atomic per block and counter (a hot address takes ~88 atomics per microsecond)
    __shared__ unsigned long long bsum;
    __shared__ unsigned bmax;
    if (threadIdx.x == 0) {
        bsum = 0;
        bmax = 0;
    }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) {
        if (r > 0) atomicMax(&bmax, (unsigned)r);
        atomicAdd(&bsum, (unsigned long long)lo + ((unsigned long long)hi << 24));
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        if (bmax) atomicMax(out, bmax);
        atomicAdd((unsigned long long *)(out + 2), bsum);
    }
}

__global__ __launch_bounds__(SCAN_THREADS) void k_scan_sums(const uint32_t *in, uint32_t n, uint32_t *partial) {
    __shared__ unsigned wt[4];
    uint32_t base = blockIdx.x * SCAN_BLOCK + threadIdx.x * SCAN_PER_THREAD;
    static_assert(SCAN_PER_THREAD == 8, "load8");
    uint32_t v[SCAN_PER_THREAD];
    load8(in, base, n, 0u, v);
    unsigned s = 0;
#pragma unroll
    for (int k = 0; k < SCAN_PER_THREAD; ++k) s += v[k];
    unsigned tot;
    block_scan_incl(s, wt, tot);
    if (threadIdx.x == 0) partial[blockIdx.x] = tot;
}

__global__ __launch_bounds__(SCAN_THREADS) void k_scan_partials(uint32_t *partial, uint32_t nb, uint32_t *out, uint32_t n) {
    __shared__ unsigned wt[4];
    unsigned carry = 0;
    for (uint32_t b0 = 0; b0 < nb; b0 += SCAN_THREADS) {
        uint32_t i = b0 + threadIdx.x;
        unsigned v = i < nb ? partial[i] : 0;
        unsigned tot;
        unsigned inc = block_scan_incl(v, wt, tot);
        if (i < nb) partial[i] = carry + inc - v;
        carry += tot;
    }
    if (threadIdx.x == 0) out[n] = carry;
}

__global__ __launch_bounds__(SCAN_THREADS) void k_scan_apply(const uint32_t *in, uint32_t n, const uint32_t *partial,
                                                             uint32_t *out) {
    __shared__ unsigned wt[4];
    uint32_t base = blockIdx.x * SCAN_BLOCK + threadIdx.x * SCAN_PER_THREAD;
    uint32_t v[SCAN_PER_THREAD];
    load8(in, base, n, 0u, v);
    const unsigned before = partial[blockIdx.x];
    unsigned s = 0;
#pragma unroll
    for (int k = 0; k < SCAN_PER_THREAD; ++k) s += v[k];
    unsigned tot;
    unsigned inc = block_scan_incl(s, wt, tot);
    unsigned run = before + inc - s;
    if (base + SCAN_PER_THREAD <= n) {  // (two 16-byte stores; out comes from hipMalloc and base is a multiple of 8)
        uint4 x, y;
        x.x = run, x.y = x.x + v[0], x.z = x.y + v[1], x.w = x.z + v[2];
        y.x = x.w + v[3], y.y = y.x + v[4], y.z = y.y + v[5], y.w = y.z + v[6];
        *(uint4 *)(out + base) = x;
        *(uint4 *)(out + base + 4) = y;
    } else {
        for (int k = 0; k < SCAN_PER_THREAD; ++k) {
            if (base + k < n) out[base + k] = run;
            run += v[k];
        }
    }
}

// (Round 6 also tried ONE WORKGROUP for a window's ~52 k values -- 16 waves, every wave its stretch as coalesced rows, one exchange through LDS:
// 0.239-0.242 ms per 50 k-read call against 0.236-0.239 with the three launches, EXPERIMENTS 6.12; removed.  Back-to-back launches of small
// kernels overlap their dispatch; a command is only dear where the stream has to drain: event records, the copy back.)
// The exclusive scan in ONE launch (decoupled look-back), an experiment of the one-round-trip path that stays switched off (PLO_SCAN_CHAIN=1):
// two launches fewer, but the chain of look-backs costs more than their ~5 us each -- wgs30x 2 M reads (1 006 tiles): enumerate pass 0.273 ms
// against 0.261, step 1.512 against 1.490 ms; 50 k-read window (25 tiles): 0.249-0.258 against 0.250 ms (tools/exp_scan.sh).
// How it works: tiles of SCAN_BLOCK values take their number from a ticket, so a tile only ever waits for tiles that started before
// it; a tile publishes {1, its sum}, looks back over the words of the tiles in front of it -- 64 at a time, one per lane of its first wave,
// adding sums until a word {2, inclusive prefix} turns up -- and publishes {2, its inclusive prefix}.  State and value travel in one 64-bit
// word (no ordering between two stores to rely on).  `state` (a word per tile) and `ticket` are zero at launch.  out[n] = the total.
__global__ __launch_bounds__(SCAN_THREADS) void k_scan_chain(const uint32_t *in, uint32_t n, uint32_t *out, unsigned long long *state, uint32_t *ticket) {
    __shared__ unsigned wt[4];
    __shared__ uint32_t sh_tile, sh_before;
    if (threadIdx.x == 0) sh_tile = atomicAdd(ticket, 1u);
    __syncthreads();
    const uint32_t tile = sh_tile;
    const uint32_t base = tile * SCAN_BLOCK + threadIdx.x * SCAN_PER_THREAD;
    uint32_t v[SCAN_PER_THREAD];
    load8(in, base, n, 0u, v);
    unsigned s = 0;
#pragma unroll
    for (int k = 0; k < SCAN_PER_THREAD; ++k) s += v[k];
    unsigned tot;
    const unsigned inc = block_scan_incl(s, wt, tot);
    if (threadIdx.x < 64u) {
        const int lane = (int)threadIdx.x;
        if (lane == 0) __hip_atomic_store(state + tile, ((tile ? 1ull : 2ull) << 32) | (unsigned long long)tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned before = 0;
        if (tile) {
            int j0 = (int)tile - 1;
            for (;;) {
                const int j = j0 - lane;
                unsigned long long w = 2ull << 32;  // (in front of tile 0: a prefix of zero)
                if (j >= 0) {
                    for (;;) {
                        w = __hip_atomic_load(state + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if ((w >> 32) != 0) break;
                        __builtin_amdgcn_s_sleep(1);
                    }
                }
                const unsigned long long pm = __ballot((w >> 32) == 2ull);
                const int first = pm ? __builtin_ctzll(pm) : 63;
                before += (unsigned)wv::reduce_add((int)(lane <= first ? (unsigned)w : 0u));
                if (pm) break;
                j0 -= 64;
            }
            if (lane == 0) __hip_atomic_store(state + tile, (2ull << 32) | (unsigned long long)(before + tot), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (lane == 0) {
            sh_before = before;
            if (tile + 1u == gridDim.x) out[n] = before + tot;
        }
    }
    __syncthreads();
    unsigned run = sh_before + inc - s;
    if (base + SCAN_PER_THREAD <= n) {
        uint4 x, y;
        x.x = run, x.y = x.x + v[0], x.z = x.y + v[1], x.w = x.z + v[2];
        y.x = x.w + v[3], y.y = y.x + v[4], y.z = y.y + v[5], y.w = y.z + v[6];
        *(uint4 *)(out + base) = x;
        *(uint4 *)(out + base + 4) = y;
    } else {
        for (int k = 0; k < SCAN_PER_THREAD; ++k) {
            if (base + k < n) out[base + k] = run;
            run += v[k];
        }
    }
}

// ---- the dominant kernel: one wave per tile ---------------------------------------------------------------------
constexpr uint32_t STAT_SLOTS = 1u << 16;  // >= waves of any lift launch (CUs x 16 blocks x 4 waves at most)
constexpr int TILE_WAVES = 4;  // waves per workgroup; every wave works on its own tile with its own LDS slice

// Persistent: the grid is sized to the resident capacity of the chip and every wave strides over the tiles, keeping its
// output slab and statistics in registers (WaveCtx).
// Register budget: 3 waves per SIMD (<= 168 VGPRs) is what the LDS slices allow as well (12 waves per CU); without the
// attribute the allocator may drift past 168 and the resident waves drop to 2 per SIMD.
#ifndef PLO_TILE_WPE
#define PLO_TILE_WPE 3
#endif
#define PLO_TILE_OCC __attribute__((amdgpu_waves_per_eu(PLO_TILE_WPE, PLO_TILE_WPE)))
// (every lift kernel exists twice: for dense read bases and, `_sp`, for PLO_SEQ_BAM4_SPARSE batches, whose probes look granules up)
// CAPC > 0: the slice capacity as a compile-time constant.  The arrays of the slice then sit at constant offsets from one base and
// the LDS instructions carry them as immediates (one address register per element instead of one per array and access): 166 -> 137
// VGPRs at cap 320, and with cap 256 the kernel fits 128 VGPRs, i.e. 16 waves per CU -- which is also exactly what 16 slices of
// tile_mem_bytes(256) = 10 240 B leave of the 160 KB of LDS.
template <bool SP, int CAPC>
PLO_DEV void lift_tiles_kernel(const DevIndex &ix, const DevBatch &bt, const DevWork &wk, uint32_t stages, uint32_t n_tiles, int window,
                               int big_thresh, int cap, uint32_t lds_per_wave) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int w = threadIdx.x >> 6;
    // XCD-aware placement: workgroup b runs on XCD b % 8 (observed dispatch order); consecutive wave ids -- hence
    // neighbouring tiles, i.e. reads of the same contig region with the same block-map / reference lines -- share an L2.
    uint32_t nb = gridDim.x;
    uint32_t per = nb >> 3;
    uint32_t b = blockIdx.x;
    uint32_t tb = (per > 0 && (nb & 7u) == 0) ? (b & 7u) * per + (b >> 3) : b;
    const uint32_t tw = blockDim.x >> 6;
    const uint32_t wave = tb * tw + (uint32_t)w, n_waves = nb * tw;
    if constexpr (CAPC > 0) {
        cap = CAPC;
        lds_per_wave = (uint32_t)((tile_mem_bytes(CAPC) + 15) & ~(size_t)15);
    }
    TileMem m = carve_tile_mem(smem + (size_t)w * lds_per_wave, cap);
    WaveCtx ctx;
    (void)window;
    if (wk.slab_pre) {  // first slab by wave id: 3 072 waves reserving theirs at the same moment would serialise on one counter
        ctx.slab_base = (unsigned long long)wave * SLAB_OPS;
        ctx.slab_left = SLAB_OPS;
    }
    lift_tiles_persistent<SP>(ix, bt, wk, stages, wave, n_waves, n_tiles, big_thresh, m, ctx);
    wave_ctx_flush(wk, ctx, wave);
}
__global__ __launch_bounds__(TILE_WAVES * 64) PLO_TILE_OCC void k_lift_tiles(DevIndex ix, DevBatch bt, DevWork wk, uint32_t stages,
                                                               uint32_t n_tiles, int window, int big_thresh, int cap,
                                                               uint32_t lds_per_wave) {
    lift_tiles_kernel<false, 0>(ix, bt, wk, stages, n_tiles, window, big_thresh, cap, lds_per_wave);
}
constexpr int TILE_CAP_SMALL = 256;  // the geometry of batches without heavy items (HiFi reads on a clean assembly): see lift_tiles_kernel
__global__ __launch_bounds__(TILE_WAVES * 64) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_lift_tiles_c256(DevIndex ix, DevBatch bt, DevWork wk, uint32_t stages,
                                                               uint32_t n_tiles, int window, int big_thresh, int cap,
                                                               uint32_t lds_per_wave) {
    lift_tiles_kernel<false, TILE_CAP_SMALL>(ix, bt, wk, stages, n_tiles, window, big_thresh, cap, lds_per_wave);
}
__global__ __launch_bounds__(TILE_WAVES * 64) PLO_TILE_OCC void k_lift_tiles_sp(DevIndex ix, DevBatch bt, DevWork wk, uint32_t stages,
                                                                  uint32_t n_tiles, int window, int big_thresh, int cap,
                                                                  uint32_t lds_per_wave) {
    lift_tiles_kernel<true, 0>(ix, bt, wk, stages, n_tiles, window, big_thresh, cap, lds_per_wave);
}

// ---- the lane-per-item kernel (lane_core.hpp): DOMINANT on HiFi batches -------------------------------------------------
// Persistent waves over groups of 64 items of the two lane classes (light items without / with the shift stage), one lane per
// item, every wave with its own LDS slice of capw dwords.
constexpr int LANE_WAVES = 4;
#ifndef PLO_LANE_WPE
#define PLO_LANE_WPE 3
#endif
#ifndef PLO_LANE_WPE_MIN
#define PLO_LANE_WPE_MIN PLO_LANE_WPE
#endif
#ifndef PLO_LANE_G_WPE
#define PLO_LANE_G_WPE 2  // k_lift_lanes_g: the windows' bookkeeping on top of 168 registers would spill; 12 KB of LDS per wave anyway
#endif
constexpr int LANE_G_WAVES = 4;
template <bool SP, bool STATS, bool H16>
PLO_DEV void lift_lanes_kernel(const DevIndex &ix, const DevBatch &bt, const DevWork &wk, uint32_t stages, uint32_t n0, uint32_t n1, uint32_t gs, int capw) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int w = threadIdx.x >> 6;
    // XCD-aware placement as in lift_tiles_kernel: neighbouring groups -- reads over the same stretch of a contig -- share an L2
    const uint32_t nb = gridDim.x, per = nb >> 3, b = blockIdx.x;
    const uint32_t tb = (per > 0 && (nb & 7u) == 0) ? (b & 7u) * per + (b >> 3) : b;
    const uint32_t tw = blockDim.x >> 6;
    const uint32_t wave = tb * tw + (uint32_t)w, n_waves = nb * tw;
    WaveCtx ctx;
    if (wk.slab_pre) {
        ctx.slab_base = (unsigned long long)wave * SLAB_OPS;
        ctx.slab_left = SLAB_OPS;
    }
    // (the ticket counter of the context's next launch: not in use during this one)
    if (wk.lane_ticket_next && b == 0u && threadIdx.x == 0u) *wk.lane_ticket_next = 0u;
    const int kvs_n = wk.lane_kvs ? (int)wk.lane_kvs : LANE_KVS;
    lane_tiles_persistent<SP, false, STATS, H16>(ix, bt, wk, stages, wave, n_waves, n0, n1, gs, (uint32_t *)smem + (size_t)w * (size_t)(capw + 2 * kvs_n), capw, ctx, 0u, kvs_n);
    wave_ctx_flush(wk, ctx, wave);
}
#define PLO_LANE_KERNEL(name, SP_, STATS_, H16_)                                                                                                             \
    __global__ __launch_bounds__(LANE_WAVES * 64) __attribute__((amdgpu_waves_per_eu(PLO_LANE_WPE_MIN, PLO_LANE_WPE))) void name(                             \
        DevIndex ix, DevBatch bt, DevWork wk, uint32_t stages, uint32_t n0, uint32_t n1, uint32_t gs, int capw, const uint32_t *totals_dev) {                   \
        if (totals_dev) { /* (the one-round-trip path: the class counts are on the device only) */                                                             \
            n0 = totals_dev[0];                                                                                                                                 \
            n1 = totals_dev[1];                                                                                                                                 \
        }                                                                                                                                                       \
        lift_lanes_kernel<SP_, STATS_, H16_>(ix, bt, wk, stages, n0, n1, gs, capw);                                                                             \
    }
PLO_LANE_KERNEL(k_lift_lanes, false, false, false)        // production: no statistics counters in the loops
PLO_LANE_KERNEL(k_lift_lanes_stats, false, true, false)   // plo_ctx_set_stats / PLO_LANE_STATS=1: algo_bytes and lane_utilisation counted
PLO_LANE_KERNEL(k_lift_lanes_sp, true, false, false)      // sparse read bases (PLO_SEQ_BAM4_SPARSE)
// 16-bit ops in the regions (lane_core.hpp, H16; PLO_LANE_H16=1, stage sets with the liftover): measured in round 6, 5-10 % slower than the 32-bit
// regions at every geometry tried (EXPERIMENTS 6.4) -- the conversions and the chunk tests cost more instructions than the halved LDS buys
PLO_LANE_KERNEL(k_lift_lanes16, false, false, true)
PLO_LANE_KERNEL(k_lift_lanes16_stats, false, true, true)
#undef PLO_LANE_KERNEL

// The same lane-per-item code for HEAVY items (too heavy for an LDS region: indel-dense or very long CIGARs), every lane's region in
// wave-private global scratch and reached through per-lane LDS windows (lane_core.hpp, LaneWin); `per` items per wave.
// A wave runs as long as its longest item, at the pace of a lone dependent instruction chain (about 3.5 us per op of the item on
// the stress workload, 2.8 us with a SIMD to itself): the kernel's time hardly depends on the number of items until every
// resident wave has its 64, and the workgroup-per-item kernel is the faster one for fewer than some 70 k heavy items
// (lane_heavy_min).  Measured and dropped: the heavy classes sorted by weight (k_chunk_sort's idea without its windows), long
// groups paired with short ones on a SIMD -- 7 % slower at 100 k reads, 11 % at 250 k: neighbours in the batch share reference and
// block-map lines, and a group of items from all over the genome gives that up.  Nor did ordering whole GROUPS by the length of their
// longest item (items left where they are), long ones paired with short ones on a SIMD in workgroups of 8 waves: 9.68 against 9.85 ms
// at 100 k reads, and 8-wave workgroups cost 20 % at 250 k (19.7 against 16.3 ms with two 4-wave workgroups per CU).
template <bool SP>
PLO_DEV void lift_lanes_g_kernel(const DevIndex &ix, const DevBatch &bt, const DevWork &wk, uint32_t stages, uint32_t lo, uint32_t mid, uint32_t hi,
                                 uint32_t per, uint32_t *scratch, int stride) {
    const uint32_t k = threadIdx.x >> 6, n_waves = gridDim.x * LANE_G_WAVES;
    const uint32_t wave = blockIdx.x * LANE_G_WAVES + k;
    __shared__ uint32_t windows[LANE_G_WAVES][64 * LANE_WIN_DWORDS + LANE_KVS_DWORDS];
    WaveCtx ctx;
    lane_heavy_persistent<SP>(ix, bt, wk, stages, wave, n_waves, lo, mid, hi, per, windows[k], scratch + (size_t)wave * (size_t)per * (size_t)stride, stride, ctx);
    wave_ctx_flush(wk, ctx, wave);
}
__global__ __launch_bounds__(LANE_G_WAVES * 64) __attribute__((amdgpu_waves_per_eu(PLO_LANE_G_WPE, PLO_LANE_G_WPE))) void k_lift_lanes_g(DevIndex ix, DevBatch bt, DevWork wk, uint32_t stages, uint32_t lo,
                                                                                                     uint32_t mid, uint32_t hi, uint32_t per, uint32_t *scratch, int stride) {
    lift_lanes_g_kernel<false>(ix, bt, wk, stages, lo, mid, hi, per, scratch, stride);
}
// The same kernel at three waves per SIMD (168 VGPRs, 144 B of spills per lane; 3 x 53 KB of LDS windows are exactly a CU's): slower for a lone
// wave -- 10.0 against 9.5 ms at 100 k stress reads, where every resident wave has one group and the kernel takes as long as its longest item
// -- and faster once the waves have several groups each and the kernel is bound by issue throughput: 25.7 against 29.3 ms at 500 k reads.
// Chosen by the number of heavy items (liftover_core).
__global__ __launch_bounds__(LANE_G_WAVES * 64) __attribute__((amdgpu_waves_per_eu(3, 3))) void k_lift_lanes_g_w3(DevIndex ix, DevBatch bt, DevWork wk, uint32_t stages, uint32_t lo,
                                                                                                     uint32_t mid, uint32_t hi, uint32_t per, uint32_t *scratch, int stride) {
    lift_lanes_g_kernel<false>(ix, bt, wk, stages, lo, mid, hi, per, scratch, stride);
}
__global__ __launch_bounds__(LANE_G_WAVES * 64) __attribute__((amdgpu_waves_per_eu(PLO_LANE_G_WPE, PLO_LANE_G_WPE))) void k_lift_lanes_g_sp(DevIndex ix, DevBatch bt, DevWork wk, uint32_t stages, uint32_t lo,
                                                                                                        uint32_t mid, uint32_t hi, uint32_t per, uint32_t *scratch, int stride) {
    lift_lanes_g_kernel<true>(ix, bt, wk, stages, lo, mid, hi, per, scratch, stride);
}

// HEAVY items, streamed (lane_stream.hpp): the same stage code as a PIPELINE OF WAVES -- a workgroup is a team of three waves (left shift,
// liftover, simplify; two for the forward class) working on the same 64 item slots, the ops flowing from wave to wave through rings in
// LDS; a lane of the head wave takes the team's next item when it has finished one.  Taken by batches that run all stages.
// Teams [0, t0): the forward class, [t0, t0 + t1): the reverse class; more teams than the chip holds are started as others retire.
#ifndef PLO_PIPE_WPE
#define PLO_PIPE_WPE 4
#endif
#ifndef PLO_PIPE_NI
#define PLO_PIPE_NI 32  // entries of the IN ring per lane (>= refill + 4)
#endif
template <bool SP, int NI, int N1, int N2, int N3>
PLO_DEV void lift_stream_kernel(const DevIndex &ix, const DevBatch &bt, const DevWork &wk, uint32_t lo, uint32_t mid, uint32_t hi, uint32_t t0, uint32_t t1, uint32_t xcd) {
    __shared__ uint32_t rings[stream_lds_dwords(NI, N1, N2, N3)];
    uint32_t b = 0, e = 0;
    bool has_shift = false;
    // XCD-aware placement (as lift_lanes_kernel): workgroup w runs on XCD w % 8; teams that are neighbours in the class order -- reads over the
    // same stretch of a contig, the same reference and block-map lines -- get workgroups of one XCD, i.e. one L2 (PLO_PIPE_XCD=1; off by default: measured slower)
    uint32_t team = blockIdx.x;
    if (xcd) {
        const uint32_t nb = gridDim.x, per = nb >> 3;
        if (team < 8u * per) team = (team & 7u) * per + (team >> 3);
    }
    pipe_team_span(team, t0, t1, lo, mid, hi, b, e, has_shift);
    WaveCtx ctx;
    pipe_team<SP, NI, N1, N2, N3>(ix, bt, wk, b, e, has_shift, rings, ctx);
    wave_ctx_flush(wk, ctx, blockIdx.x * PIPE_WAVES + (uint32_t)wv::wave_id());
}
__global__ __launch_bounds__(PIPE_WAVES * 64) __attribute__((amdgpu_waves_per_eu(PLO_PIPE_WPE, PLO_PIPE_WPE))) void k_lift_stream(DevIndex ix, DevBatch bt, DevWork wk, uint32_t lo, uint32_t mid, uint32_t hi, uint32_t t0, uint32_t t1, uint32_t xcd) {
    lift_stream_kernel<false, PLO_PIPE_NI, 16, 16, 32>(ix, bt, wk, lo, mid, hi, t0, t1, xcd);
}
__global__ __launch_bounds__(PIPE_WAVES * 64) __attribute__((amdgpu_waves_per_eu(PLO_PIPE_WPE, PLO_PIPE_WPE))) void k_lift_stream_sp(DevIndex ix, DevBatch bt, DevWork wk, uint32_t lo, uint32_t mid, uint32_t hi, uint32_t t0, uint32_t t1, uint32_t xcd) {
    lift_stream_kernel<true, PLO_PIPE_NI, 16, 16, 32>(ix, bt, wk, lo, mid, hi, t0, t1, xcd);
}

// Items of tiles (or of the lane kernel) whose intermediates overflowed the shared capacity: the tile code again, RETRY_PER
// items per wave with a larger LDS slice (retry_cap)
constexpr uint32_t RETRY_PER = 1;
template <bool SP>
PLO_DEV void lift_retry_kernel(const DevIndex &ix, const DevBatch &bt, const DevWork &wk, uint32_t stages, uint32_t n_retry, int big_thresh, int cap) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    TileMem m = carve_tile_mem(smem, cap);
    WaveCtx ctx;
    Coop<1> co;
    if (n_retry == 0xffffffffu) n_retry = (uint32_t)wk.counters[CNT_NRETRY];  // as the tile kernel left it (this kernel hands on to big_list)
    for (uint32_t r = blockIdx.x * RETRY_PER; r < n_retry; r += gridDim.x * RETRY_PER) {
        uint32_t left = n_retry - r;
        lift_tile<1, SP>(co, ix, bt, wk, stages, r, (int)(left < RETRY_PER ? left : RETRY_PER), m, wk.retry_list, LEVEL_RETRY, big_thresh, ctx);
        wv::sync();
    }
    wave_ctx_flush(wk, ctx, blockIdx.x);
}
__global__ __launch_bounds__(64) void k_lift_retry(DevIndex ix, DevBatch bt, DevWork wk, uint32_t stages, uint32_t n_retry,
                                                   int big_thresh, int cap) {
    lift_retry_kernel<false>(ix, bt, wk, stages, n_retry, big_thresh, cap);
}
__global__ __launch_bounds__(64) void k_lift_retry_sp(DevIndex ix, DevBatch bt, DevWork wk, uint32_t stages, uint32_t n_retry,
                                                      int big_thresh, int cap) {
    lift_retry_kernel<true>(ix, bt, wk, stages, n_retry, big_thresh, cap);
}

// One workgroup of NW waves per item (Coop<NW>): the item's op stream, temporaries and block-map window live in the
// workgroup's LDS (cap elements), every pass walks it NW x 64 elements at a time.  Persistent workgroups over `list`.
constexpr int MID_CAPK = 512;  // staged block-map entries of the one item (4 KB); longer windows are read from global memory
// 128 VGPRs at most: 16 waves per CU, as one workgroup of 16 or two of 8
template <int NW, bool SP>
PLO_DEV void lift_mid_kernel(const DevIndex &ix, const DevBatch &bt, const DevWork &wk, uint32_t stages, uint32_t n_list, int mid_thresh, int cap) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    TileMem m = carve_tile_mem(smem, cap, MID_CAPK);
    Coop<NW> co;
    co.w = wv::wave_id();
    co.xch = (int *)(smem + ((tile_mem_bytes(cap, MID_CAPK) + 15) & ~(size_t)15));
    if (threadIdx.x < (unsigned)Coop<NW>::XCH_INTS) co.xch[threadIdx.x] = 0;
    __syncthreads();
    WaveCtx ctx;
    for (uint32_t i = blockIdx.x; i < n_list; i += gridDim.x) {
        lift_tile<NW, SP>(co, ix, bt, wk, stages, i, 1, m, wk.big_list, LEVEL_MID, mid_thresh, ctx);
        co.sync();
    }
    wave_ctx_flush(wk, ctx, blockIdx.x * NW + (uint32_t)co.w);
}
template <int NW>
__global__ __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_lift_mid(DevIndex ix, DevBatch bt, DevWork wk, uint32_t stages, uint32_t n_list,
                                                      int mid_thresh, int cap) {
    lift_mid_kernel<NW, false>(ix, bt, wk, stages, n_list, mid_thresh, cap);
}
template <int NW>
__global__ __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_lift_mid_sp(DevIndex ix, DevBatch bt, DevWork wk, uint32_t stages, uint32_t n_list,
                                                         int mid_thresh, int cap) {
    lift_mid_kernel<NW, true>(ix, bt, wk, stages, n_list, mid_thresh, cap);
}
template <int NW>
static size_t mid_lds_bytes(int cap) {
    return ((tile_mem_bytes(cap, MID_CAPK) + 15) & ~(size_t)15) + (size_t)Coop<NW>::XCH_INTS * 4;
}

template <bool SP>
PLO_DEV void lift_big_kernel(const DevIndex &ix, const DevBatch &bt, const DevWork &wk, uint32_t stages, uint32_t n_big, const uint32_t *list,
                             unsigned char *scratch, int big_cap, unsigned long long bytes_per_wave) {
    TileMem m = carve_tile_mem(scratch + (unsigned long long)blockIdx.x * bytes_per_wave, big_cap);
    WaveCtx ctx;
    Coop<1> co;
    for (uint32_t i = blockIdx.x; i < n_big; i += gridDim.x) {
        lift_tile<1, SP>(co, ix, bt, wk, stages, i, 1, m, list, LEVEL_LAST, 0, ctx);
        wv::sync();
    }
    wave_ctx_flush(wk, ctx, blockIdx.x);
}
__global__ __launch_bounds__(64) void k_lift_big(DevIndex ix, DevBatch bt, DevWork wk, uint32_t stages, uint32_t n_big,
                                                 const uint32_t *list, unsigned char *scratch, int big_cap,
                                                 unsigned long long bytes_per_wave) {
    lift_big_kernel<false>(ix, bt, wk, stages, n_big, list, scratch, big_cap, bytes_per_wave);
}
__global__ __launch_bounds__(64) void k_lift_big_sp(DevIndex ix, DevBatch bt, DevWork wk, uint32_t stages, uint32_t n_big,
                                                    const uint32_t *list, unsigned char *scratch, int big_cap,
                                                    unsigned long long bytes_per_wave) {
    lift_big_kernel<true>(ix, bt, wk, stages, n_big, list, scratch, big_cap, bytes_per_wave);
}

// Sparse bases (PLO_SEQ_BAM4_SPARSE): the items whose probes reached absent granules are lifted again from the complete bases of
// their reads, which the host gathers from plo_batch_in::seq_full.  k_miss_info tells it which reads; k_miss_patch points the
// items' sequence offsets (a copy of the descriptor column) into the side buffer with those bases.
__global__ void k_miss_info(const uint32_t *list, uint32_t n, DevBatch bt, DevWork wk, uint32_t *out_read, uint32_t *out_len) {
    uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const uint32_t rd = bt.seg_read[wk.item_seg[list[k]]];
    out_read[k] = rd;
    out_len[k] = bt.read_seq_len[rd];
}
__global__ void k_miss_patch(const uint32_t *list, uint32_t n, const uint64_t *vals, uint64_t *seq_off) {
    uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n) seq_off[list[k]] = vals[k];
}

// sums (and clears) the per-wave statistic slots of the lift kernels that have run since the last call into the batch counters
__global__ __launch_bounds__(256) void k_sum_stats(unsigned long long *ws, uint32_t n_slots, unsigned long long *counters) {
    __shared__ unsigned long long acc[5];
    if (threadIdx.x < 5) acc[threadIdx.x] = 0;
    __syncthreads();
    unsigned long long a[5] = {0, 0, 0, 0, 0};
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_slots; i += gridDim.x * blockDim.x) {
        unsigned long long *w = ws + (size_t)i * STAT_WORDS;
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            a[k] += w[k];
            w[k] = 0;
        }
    }
#pragma unroll
    for (int k = 0; k < 5; ++k) {
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) a[k] += (unsigned long long)__shfl_xor((long long)a[k], d, 64);
        if ((threadIdx.x & 63u) == 0) atomicAdd(&acc[k], a[k]);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicAdd(&counters[CNT_ALGO_BYTES], acc[0]);
        atomicAdd(&counters[CNT_IN_OPS], acc[1]);
        atomicAdd(&counters[CNT_OUT_OPS], acc[2]);
        atomicAdd(&counters[CNT_LANE_ACT], acc[3]);
        atomicAdd(&counters[CNT_LANE_TRIPS], acc[4]);
    }
}

// The one-round-trip path's last launch: k_lift_retry and k_sum_stats in one (a command of any size costs ~5 us on the stream, and a
// reference-sized window's call is made of a dozen of them).  Every wave lifts its share of the retry list as k_lift_retry does, then adds
// what it counted straight into the batch counters (nothing to add for the common wave without a retry item); the first SUM_BLOCKS waves
// also sum (and clear) the statistic slots the light-item kernel's waves left -- complete by stream order, `n_lane_slots` of them from slot 0.
constexpr uint32_t SUM_BLOCKS = 32;
__global__ __launch_bounds__(64) void k_lift_retry_sum(DevIndex ix, DevBatch bt, DevWork wk, uint32_t stages, int big_thresh, int cap, uint32_t n_lane_slots) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    TileMem m = carve_tile_mem(smem, cap);
    WaveCtx ctx;
    Coop<1> co;
    const uint32_t n_retry = (uint32_t)wk.counters[CNT_NRETRY];
    for (uint32_t r = blockIdx.x * RETRY_PER; r < n_retry; r += gridDim.x * RETRY_PER) {
        uint32_t left = n_retry - r;
        lift_tile<1, false>(co, ix, bt, wk, stages, r, (int)(left < RETRY_PER ? left : RETRY_PER), m, wk.retry_list, LEVEL_RETRY, big_thresh, ctx);
        wv::sync();
    }
    // this wave's own counts, per lane as wave_ctx_flush reads them; the slots of the light-item kernel on top (lane l of wave b: slots
    // b + SUM_BLOCKS (l + 64 j))
    unsigned long long a[5] = {ctx.algo_bytes, ctx.in_ops, ctx.out_ops, 0ull, 0ull};
    const uint32_t lane = (uint32_t)wv::lane();
    if (lane == 0) {
        a[3] = ctx.u_act;
        a[4] = ctx.u_trips;
    }
    if (blockIdx.x < SUM_BLOCKS) {
        for (uint32_t i = blockIdx.x + SUM_BLOCKS * lane; i < n_lane_slots; i += SUM_BLOCKS * 64u) {
            unsigned long long *w = wk.wave_stats + (size_t)i * STAT_WORDS;
#pragma unroll
            for (int k = 0; k < 5; ++k) {
                a[k] += w[k];
                w[k] = 0;
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 5; ++k) {
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) a[k] += (unsigned long long)__shfl_xor((long long)a[k], d, 64);
    }
    if (lane == 0) {
        if (a[0]) atomicAdd(&wk.counters[CNT_ALGO_BYTES], a[0]);
        if (a[1]) atomicAdd(&wk.counters[CNT_IN_OPS], a[1]);
        if (a[2]) atomicAdd(&wk.counters[CNT_OUT_OPS], a[2]);
        if (a[3]) atomicAdd(&wk.counters[CNT_LANE_ACT], a[3]);
        if (a[4]) atomicAdd(&wk.counters[CNT_LANE_TRIPS], a[4]);
    }
}

// ---- record finishing (finish_core.hpp) ----------------------------------------------------------------------------------
__global__ void k_finish_items(DevBatch bt, DevWork wk, DevFinish f) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < wk.n_items) finish_item(bt, wk, f, i);
}
__global__ void k_finish_reads(DevBatch bt, DevWork wk, DevFinish f) {
    uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r < bt.n_reads) finish_read(bt, wk, f, r);
}
__global__ void k_finish_offsets(DevBatch bt, DevWork wk, DevFinish f) {
    uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t n = wk.n_items;
    if (e >= n + bt.n_reads) return;
    uint64_t so = f.su[e] ? (uint64_t)f.soff[e] * 16u : PLO_NO_FLIP;
    uint64_t qo = f.qu[e] ? (uint64_t)f.qoff[e] * 16u : PLO_NO_FLIP;
    if (f.su[e]) f.flist[f.frank[e]] = e;
    if (e < n) {
        f.item_seq_off[e] = so;
        f.item_qual_off[e] = qo;
    } else {
        f.read_seq_off[e - n] = so;
        f.read_qual_off_out[e - n] = qo;
    }
}
// dense re-packing of the slab-allocated output CIGARs: eight lanes per item copy 32 contiguous bytes per step
__global__ __launch_bounds__(256) void k_compact_cigar(const uint32_t *src, uint64_t *cig_off, const uint32_t *cig_len,
                                                       const uint32_t *dense_off, uint32_t n, uint32_t *dst) {
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t i = t >> 3, sub = t & 7u;
    if (i >= n) return;
    const uint64_t from = cig_off[i];  // read by all eight lanes before lane 0 overwrites it below
    const uint32_t len = cig_len[i], to = dense_off[i];
    for (uint32_t k = sub; k < len; k += 8) dst[to + k] = src[from + k];
    if (sub == 0) cig_off[i] = to;
}

// SA tag text: thread per item (only the few items of reads with several lifted records produce text)
__global__ void k_sa_len(DevWork wk, DevSa sa) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < wk.n_items) sa.len[i] = sa_item_len(wk, sa, i);
}
__global__ void k_sa_emit(DevWork wk, DevSa sa) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < wk.n_items) sa_item_emit(wk, sa, i);
}
// reverse_alignment_seq_and_qual: one workgroup per record that needs it, streaming 16-byte-aligned outputs.
// HBM-bound: per flipped read L/2 + L bytes in, the same out.
__global__ __launch_bounds__(256) void k_revcomp(DevBatch bt, DevWork wk, DevFinish f) {
    uint32_t n = wk.n_items, n_flip = f.frank[n + bt.n_reads];
    for (uint32_t k = blockIdx.x; k < n_flip; k += gridDim.x) {
        uint32_t e = f.flist[k];
        uint32_t read = e < n ? f.item_read[e] : e - n;
        revcomp_record(bt, f, read, f.rev_seq + (uint64_t)f.soff[e] * 16u, f.rev_qual + (uint64_t)f.qoff[e] * 16u, (int)threadIdx.x,
                       (int)blockDim.x);
    }
}

// ---- BGZF inflate (inflate.hpp): every block of a chunk of the BAM stream at once, one wave per block ----------------------
struct BgzfBlk {
    unsigned long long coff, uoff;  // offsets of the block's deflate data / inflated bytes inside the chunk buffers
    uint32_t clen, ulen;
};
struct InfWave {  // wave primitives of the decoder's I/O policy (InfWaveIO, inflate.hpp)
    PLO_DEV int lane() const { return wv::lane(); }
    PLO_DEV void sync() const {  // LDS windows / tables and the wave's own global stores (far match sources) are ordered for all its lanes
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_s_waitcnt(0);  // vmcnt(0) expcnt(0) lgkmcnt(0): the write-backs of the ring have reached L2
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    }
    PLO_DEV uint32_t uniform(uint32_t v) const { return wv::bcast_first(v); }
    PLO_DEV uint32_t scalar(uint32_t v) const { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
    PLO_DEV uint32_t read_lane(uint32_t v, uint32_t l) const {  // l wave-uniform
        return (uint32_t)__builtin_amdgcn_readlane((int)v, __builtin_amdgcn_readfirstlane((int)l));
    }
    PLO_DEV int popcount64(unsigned long long v) const { return __builtin_popcountll(v); }
    PLO_DEV uint32_t rank_below(unsigned long long mask) const {  // set bits of mask below this lane
        return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
    }
    PLO_DEV long long clock() const { return wv::clock(); }
    // a byte this wave wrote back earlier: read at device scope (L2), never from a line the CU's vector cache fetched while the
    // line was still being filled
    PLO_DEV uint8_t load_written(const uint8_t *p) const { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
};
struct InfLds {
    InfWork ws;
    InfWaveMem io;
};
// Four waves per workgroup, each on its own block with its own LDS: the workgroup's waves sit on the four SIMDs of the CU, which
// one-wave workgroups do not (the decoder is bound by instruction issue once a SIMD holds several of them).
constexpr int INF_WAVES = 4;
__global__ __launch_bounds__(INF_WAVES * 64) void k_bgzf_inflate(const uint8_t *comp, const BgzfBlk *blks, uint32_t n, uint8_t *out, int *status) {
    __shared__ InfLds lds_all[INF_WAVES];  // 4 x 15.9 KB: two workgroups, eight blocks in flight per CU
    InfLds &lds = lds_all[wv::wave_id()];
    InfWaveIO<InfWave> io;
    io.m = &lds.io;
    for (uint32_t b = blockIdx.x * INF_WAVES + (uint32_t)wv::wave_id(); b < n; b += gridDim.x * INF_WAVES) {
        const BgzfBlk k = blks[b];
        uint32_t w = 0;
#ifdef PLO_INF_TIMING
        const long long tb = wv::clock();
#endif
        int rc = k.ulen ? inflate_block(io, comp + k.coff, k.clen, out + k.uoff, k.ulen, lds.ws, &w) : 0;
#ifdef PLO_INF_TIMING
        if (b == 7 && wv::lane() == 0)
            printf("[inflate] block %u: %u -> %u bytes, %lld cycles: input chunks %d (%lld), write-backs %d (%lld), matches %d (%lld), far %d (%lld), tables %d (%lld), literals in runs %d (%lld)\n", b, k.clen, k.ulen,
                   wv::clock() - tb, io.n_load, io.t_load, io.n_flush, io.t_flush, io.n_match, io.t_match, io.n_far, io.t_far, io.n_table, io.t_table, io.n_run, io.t_run);
#endif
        if (rc == 0 && w != k.ulen) rc = -9;  // the stream ended before ISIZE bytes
        if (wv::lane() == 0) status[b] = rc;
        io.sync();
    }
}

// The CRC-32 of every inflated block against the one its BGZF trailer carries (crc32_wave, inflate.hpp: a wave per block), behind
// k_bgzf_inflate on the same stream: the host then never reads the inflated bytes to check them (round 4: the CRC pass and the copy of the
// compressed bytes out of the file mapping were 50 of the 63 ms a 470 MB refill took).  status: -10 where the CRC differs.
__global__ __launch_bounds__(INF_WAVES * 64) void k_bgzf_crc(const uint8_t *out, const BgzfBlk *blks, uint32_t n, const uint32_t *crcs, int *status) {
    __shared__ uint32_t tab[256];
    tab[threadIdx.x & 255u] = crc32_table_entry(threadIdx.x & 255u);
    __syncthreads();
    for (uint32_t b = blockIdx.x * INF_WAVES + (uint32_t)wv::wave_id(); b < n; b += gridDim.x * INF_WAVES) {
        const BgzfBlk k = blks[b];
        const uint32_t c = k.ulen ? crc32_wave(out + k.uoff, k.ulen, tab) : 0u;
        if (wv::lane() == 0 && k.ulen && status[b] == 0 && c != crcs[b]) status[b] = -10;
    }
}

// ---- self-test of the wave primitives (plo_selftest) -------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_selftest(const int *in, int *out) {
    int lane = wv::lane();
    int x = in[lane];
    out[0 * 64 + lane] = wv::scan_add(x);
    out[1 * 64 + lane] = wv::scan_max(x);
    out[2 * 64 + lane] = wv::shfl_up1(x, -7);
    out[3 * 64 + lane] = wv::shfl(x, (lane * 7 + 3) & 63);
    out[4 * 64 + lane] = wv::bcast_last(x);
    out[5 * 64 + lane] = wv::bcast_first(x);
    wv::MinPlus f;
    f.a = in[64 + lane];
    f.b = in[128 + lane];
    f.s = in[192 + lane];
    wv::MinPlus F = wv::scan_minplus(f);
    out[6 * 64 + lane] = F.a;
    out[7 * 64 + lane] = F.b;
    out[8 * 64 + lane] = F.s;
    unsigned long long bm = wv::ballot((x & 1) != 0);
    out[9 * 64 + lane] = (int)((bm >> lane) & 1ull);
    Coop<1> co;
    AddScan as(co);
    MaxScan ms(co, -1);
    int a0 = as.excl(x & 15), a1 = as.excl((x >> 4) & 15);
    int m0 = ms.incl((x & 3) == 0 ? lane : -1);
    int e0 = ms.excl_of(m0);
    int m1 = ms.incl((x & 3) == 1 ? 64 + lane : -1);
    int e1 = ms.excl_of(m1);
    out[10 * 64 + lane] = a0;
    out[11 * 64 + lane] = a1;
    out[12 * 64 + lane] = e0;
    out[13 * 64 + lane] = e1;
}

// ---------------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------------

namespace {

struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    hipError_t ensure(size_t bytes) {
        if (bytes <= cap) return hipSuccess;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        size_t want = bytes + bytes / 4 + 256;
        hipError_t e = hipMalloc(&p, want);
        if (e == hipSuccess) cap = want;
        return e;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
    template <class T>
    T *as() const {
        return (T *)p;
    }
};

struct HostBuf {  // pinned
    void *p = nullptr;
    size_t cap = 0;
    hipError_t ensure(size_t bytes) {
        if (bytes <= cap) return hipSuccess;
        if (p) (void)hipHostFree(p);
        p = nullptr;
        cap = 0;
        size_t want = bytes + bytes / 4 + 256;
        hipError_t e = hipHostMalloc(&p, want, hipHostMallocDefault);
        if (e == hipSuccess) cap = want;
        return e;
    }
    void release() {
        if (p) (void)hipHostFree(p);
        p = nullptr;
        cap = 0;
    }
    template <class T>
    T *as() const {
        return (T *)p;
    }
};

}  // namespace

struct plo_index {
    int device = 0;
    PackedIndex host;  // host copy of the packed arrays (plo_index_segment_map, validation)
    DevIndex d{};
    std::vector<void *> owned;  // device allocations owned by the index
    uint32_t max_segs_per_contig = 0;
};

struct plo_ctx {
    const plo_index *ix = nullptr;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    std::string err;
    // workspace
    DevBuf f_flag, f_bin, f_end, f_prim, f_isoff, f_iqoff, f_iread, f_nl, f_pitem, f_uflag, f_rsoff, f_rqoff, f_su, f_qu, f_soff,
        f_qoff, f_rseq, f_rqual, f_fflag, f_frank, f_flist, sa_len, sa_off, sa_text;
    DevWork last_wk{};
    DevBatch last_bt{};
    bool have_last = false, have_finish = false;
    hipEvent_t fev[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    DevBuf item_region, lane_groups;
    DevBuf misc, whist, cls_partial, lane_scratch, item_cls, retry_list, perm, nin_p, seg_reflen, seg_readlen, seg_nm, seg_cnt, seg_off, scan_partial, item_seg, item_cseg, item_nin, op_prefix, counters, big_list, huge_list, scratch, tile_lo, verr, miss_list, miss_info, miss_vals, miss_seq_off, miss_side;
    DevBuf d_n_m, d_in_off, d_n_in, d_pos1, d_w0, d_w1, d_kv0, d_kv1, d_flags, d_contig, d_seq_len, d_seq_off, d_shift_ref, d_shift_ref_len,
        d_chrom_ref, d_chrom_ref_len, d_read_len;
    // outputs (device)
    DevBuf o_status, o_flip, o_mapq, o_chrom, o_pos, o_coff, o_clen, o_cigar, o_dense_off, o_cigar_dense, wave_stats;
    DevBuf fast_blk;  // liftover_fast: class totals, validation flags and batch counters in one block (one fill, one copy back)
    bool fast_timing = false;  // the last batch took liftover_fast: events 0, 1, 4 only
    bool scan_chain = false;   // liftover_fast: the segments' item offsets by k_scan_chain (PLO_SCAN_CHAIN=1; measured slower than the three scan launches)
    bool fast_fuse = true;     // liftover_fast: the counters' sum inside the retry launch (k_lift_retry_sum; PLO_FAST_FUSE=0: k_lift_retry + k_sum_stats)
    bool phase_events = true;  // liftover_fast records events 0, 1, 4 around its phases (plo_ctx_set_phase_events; every record is a ~5 us bubble on the stream)
    bool fast_no_events = false;  // the last batch took liftover_fast without them: plo_ctx_timing reports no times for it
    // host staging for plo_liftover_batch
    DevBuf i_read_rev, i_read_len, i_read_off, i_seq, i_seg_read, i_seg_contig, i_seg_pos, i_seg_fwd, i_seg_coff, i_cigar,
        i_item_seg, i_item_cseg;
    HostBuf h_item_seg, h_item_cseg, h_status, h_flip, h_mapq, h_chrom, h_pos, h_coff, h_clen, h_cigar, h_counters, h_miss, h_side;
    hipEvent_t ev[7] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    // plo_liftover_batch uploads the read bases on a stream of their own: the enumerate pass does not look at them, so that copy
    // (most of the batch's bytes) runs under it; the first lift kernel waits for ev_seq
    hipStream_t copy_stream = nullptr;
    hipEvent_t ev_seq = nullptr;
    hipEvent_t ev_cls = nullptr;  // class totals copied to the host (k_permute2 runs behind it)
    bool seq_pending = false;
    bool ev_big = false, ev_mid = false;
    uint64_t dense_total = 0;
    plo_timing timing{};
    unsigned long long phase_cycles[12] = {0};
    // tuning
    // cap sets the LDS slice of a wave (34 B per element + 1.25 KB): 320 -> 12,160 B -> 3 four-wave blocks = 12 waves per CU,
    // which is also what the kernel's VGPR budget allows; measured on MI355X (wgs30x, 2M reads): cap 512 (8 waves/CU) 8.34 ms,
    // cap 384 (still 2 blocks) 8.33 ms, cap 320 6.26 ms, cap 256 6.27 ms + overflow items.
    int window = 256, big_thresh = 176, cap = 320;
    bool adaptive = true;  // geometry chosen per batch (off when any of PLO_WINDOW / PLO_BIG_THRESH / PLO_CAP is set)
    int n_cus = 256;
    int tile_waves = TILE_WAVES;
    bool small_window_tight = false;  // the 256-element slice re-ran too many items with the wide window: keep 64 elements of allowance
    // lane-per-item kernel (lane_core.hpp): items up to lane_max_w (item_weight) take it, 64 per wave, every wave with an LDS
    // slice of lane_capw dwords; heavier items take the wave-cooperative kernels below.  (Round 1's first attempt at one lane
    // per item -- 36 KB of LDS per wave, synchronous probes -- was a net loss and had been removed; this is a different kernel.)
    int lane_max_w = 192;
    int lane_capw = 3072;
    bool lane_sort = true;  // k_chunk_sort before the lane kernel
    // groups of the lane kernel dealt dynamically (lane_tiles_persistent): two ticket counters used in turn (a launch zeroes the next one's),
    // the rounds every wave takes by fixed slots first (PLO_LANE_STATIC; < 0: fixed slots only)
    DevBuf lane_ticket;
    uint32_t lane_epoch = 0;
    int lane_static_rounds = 0;  // 0: by the launch's rounds (lane_ticket_arm); > 0: that many; < 0: fixed slots only
    int lane_tail_rounds = 1;  // PLO_LANE_TAIL: rounds' worth of cheap groups (no shift stage) dealt last (0: classes one to one)
    int lane_kvs = LANE_KVS;  // block-map entries staged per wave (PLO_LANE_KVS: 64 .. LANE_KVS_MAX)
    bool lane_h16 = false;    // PLO_LANE_H16=1: 16-bit regions (k_lift_lanes16; stage sets with the liftover)
    bool lane_stats = false;  // PLO_LANE_STATS=1: the light-item kernel that counts algorithmic bytes and lane utilisation (k_lift_lanes_stats)
    int lane_sort_window = LANE_SORT_WINDOW;
    // groups cut by LDS budget inside larger sort windows (k_chunk_sort, lane_groups_cut): on for batches whose groups are of 64
    bool lane_budget = false;  // (measured, MI355X, wgs30x 2 M reads: 1.42 ms with windows of 512 against 1.29 ms with fixed groups in windows of 128 -- DESIGN.md section 6)
    int lane_budget_window = 512;
    bool lane_stream = true;       // heavy items of all-stage batches through the streaming kernel (lane_stream.hpp)
    // the one-round-trip path (liftover_fast): what the last batch of this context left -- the segments / items its arrays are sized for, its
    // class counts, whether it had heavy items (a context that sees windows of one shape lifts them without asking the device for counts)
    bool fast = true;
    uint32_t fast_ns_cap = 0, fast_item_cap = 0, fast_n0 = 0, fast_n1 = 0;
    uint32_t fast_ref_items = 0, fast_ref_ns = 0;  // items and segments of the last careful batch: the one-round-trip path's launch bound
    bool fast_launch_bound = true;                 // (PLO_FAST_LAUNCH_BOUND=0: grids by the arrays' capacity whatever the batch's size)
    uint32_t item_cap_now = 0;   // items the per-item arrays hold
    uint32_t fast_extra_syncs = 0;  // a fast attempt that fell back: its round trip counts in the call's host_syncs
    bool fast_light_only = false;
    uint32_t lane_groups_cap = 0;  // groups the list of the budget-cut groups has room for (set with the list, attempt 0 of a batch)
    // k_lift_lanes_g (heavy items through the lane-per-item code, regions in global scratch behind LDS windows): lane_heavy_min >= 0 = for
    // batches with at least that many heavy items; < 0 (default) = by lane_heavy_ratio, see the routing in liftover_core (0: never).
    // Stress workload, heavy items -> k_lift_mid / k_lift_lanes_g: 20 k 2.8 / 7.9 ms, 60 k 8.0 / 9.6 ms, 80 k 10.5 / 9.6 ms,
    // 100 k 13.1 / 9.9 ms, 250 k 32.5 / 16-20 ms.
    int lane_heavy_min = -1;
    int lane_heavy_ratio = 50000;
    int lane_stream_ratio = 30000;  // the same for the streaming kernel (PLO_LANE_STREAM_RATIO)
    // workgroup-per-item kernel for the items a shared tile cannot hold (k_lift_mid): waves per workgroup (8 or 16; 0 = off,
    // such items then run one wave each from global scratch) and the largest LDS capacity in elements
    int mid_waves = 16;
    int mid_cap_max = 4096;
};

#define HIP_TRY(ctx, call)                                                                      \
    do {                                                                                        \
        hipError_t _e = (call);                                                                 \
        if (_e != hipSuccess) {                                                                 \
            (ctx)->err = std::string(#call) + ": " + hipGetErrorString(_e);                     \
            return (_e == hipErrorOutOfMemory) ? PLO_ERR_OUT_OF_MEMORY : PLO_ERR_HIP;           \
        }                                                                                       \
    } while (0)

static thread_local std::string g_index_err;

template <class T>
static hipError_t upload(std::vector<void *> &owned, const std::vector<T> &v, const T **out) {
    void *p = nullptr;
    size_t bytes = std::max<size_t>(v.size() * sizeof(T), 16);
    hipError_t e = hipMalloc(&p, bytes);
    if (e != hipSuccess) return e;
    owned.push_back(p);
    if (!v.empty()) {
        e = hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice);
        if (e != hipSuccess) return e;
    }
    *out = (const T *)p;
    return hipSuccess;
}

extern "C" {

const char *plo_version(void) { return "portello-liftover-mi355x 0.1 (gfx950)"; }
uint32_t plo_api_version(void) { return PLO_API_VERSION; }

plo_status plo_index_create(const plo_index_desc *desc, int device, plo_index **out) {
    if (!desc || !out) return PLO_ERR_INVALID_ARG;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) return PLO_ERR_NO_DEVICE;
    if (hipSetDevice(device) != hipSuccess) return PLO_ERR_NO_DEVICE;
    plo_index *ix = new plo_index();
    ix->device = device;
    std::string err;
    plo_status st = pack_index(desc, ix->host, err, /*build_kv=*/false);
    if (st != PLO_OK) {
        g_index_err = err;
        delete ix;
        return st;
    }
    for (uint32_t c = 0; c < desc->n_contigs; ++c)
        ix->max_segs_per_contig = std::max(ix->max_segs_per_contig, ix->host.contig_seg_off[c + 1] - ix->host.contig_seg_off[c]);
    auto fail = [&](plo_status s) {
        for (void *p : ix->owned) (void)hipFree(p);
        delete ix;
        return s;
    };
    DevIndex &d = ix->d;
    const PackedIndex &h = ix->host;
    {   // block maps: built on the device from the contig->reference CIGARs
        uint32_t ns = desc->n_segments;
        std::vector<uint32_t> cg(desc->seg_cigar, desc->seg_cigar + (ns ? desc->seg_cigar_off[ns] : 0));
        std::vector<uint32_t> cgoff(desc->seg_cigar_off, desc->seg_cigar_off + (ns ? ns + 1 : 0));
        if (!ns) cgoff.assign(1, 0);
        std::vector<int64_t> spos(desc->seg_pos, desc->seg_pos + ns);
        std::vector<void *> tmp;
        const uint32_t *d_cg = nullptr, *d_cgoff = nullptr;
        const int64_t *d_pos = nullptr;
        const int *d_cnt = nullptr;
        std::vector<int> cnt(std::max(1u, ns), 0);
        auto drop_tmp = [&]() {
            for (void *p : tmp) (void)hipFree(p);
        };
        if (upload(tmp, cg, &d_cg) != hipSuccess || upload(tmp, cgoff, &d_cgoff) != hipSuccess || upload(tmp, spos, &d_pos) != hipSuccess ||
            upload(tmp, cnt, &d_cnt) != hipSuccess) {
            drop_tmp();
            return fail(PLO_ERR_HIP);
        }
        if (ns) hipLaunchKernelGGL(k_map_build, dim3(std::min<uint32_t>(ns, 65535u * 16u)), dim3(64), 0, 0, d_cg, d_cgoff, d_pos, ns, (const uint32_t *)nullptr,
                                   (KV *)nullptr, (int *)d_cnt);
        if (hipMemcpy(cnt.data(), d_cnt, cnt.size() * 4, hipMemcpyDeviceToHost) != hipSuccess) {
            drop_tmp();
            return fail(PLO_ERR_HIP);
        }
        ix->host.cs_kv_off.assign(ns + 1, 0);
        for (uint32_t g = 0; g < ns; ++g) {
            if (cnt[g] < 0) {
                g_index_err = "contig segment coordinate outside the 31-bit BAM range or invalid CIGAR op code";
                drop_tmp();
                return fail(PLO_ERR_RANGE);
            }
            ix->host.cs_kv_off[g + 1] = ix->host.cs_kv_off[g] + (uint32_t)cnt[g];
        }
        if (upload(ix->owned, h.cs_kv_off, &d.cs_kv_off) != hipSuccess) {
            drop_tmp();
            return fail(PLO_ERR_HIP);
        }
        void *kvp = nullptr;
        if (hipMalloc(&kvp, std::max<size_t>((size_t)h.cs_kv_off[ns] * sizeof(KV), 16)) != hipSuccess) {
            drop_tmp();
            return fail(PLO_ERR_OUT_OF_MEMORY);
        }
        ix->owned.push_back(kvp);
        d.kv = (const KV *)kvp;
        if (ns) hipLaunchKernelGGL(k_map_build, dim3(std::min<uint32_t>(ns, 65535u * 16u)), dim3(64), 0, 0, d_cg, d_cgoff, d_pos, ns, d.cs_kv_off, (KV *)kvp,
                                   (int *)nullptr);
        hipError_t e = hipDeviceSynchronize();
        drop_tmp();
        if (e != hipSuccess) return fail(PLO_ERR_HIP);
    }
    if (upload(ix->owned, h.cs_chrom, &d.cs_chrom) != hipSuccess) return fail(PLO_ERR_HIP);
    if (upload(ix->owned, h.cs_is_fwd, &d.cs_is_fwd) != hipSuccess) return fail(PLO_ERR_HIP);
    if (upload(ix->owned, h.cs_mapq, &d.cs_mapq) != hipSuccess) return fail(PLO_ERR_HIP);
    if (upload(ix->owned, h.cs_start, &d.cs_start) != hipSuccess) return fail(PLO_ERR_HIP);
    if (upload(ix->owned, h.cs_end, &d.cs_end) != hipSuccess) return fail(PLO_ERR_HIP);
    if (upload(ix->owned, h.contig_seg_off, &d.contig_seg_off) != hipSuccess) return fail(PLO_ERR_HIP);
    if (upload(ix->owned, h.contig_len, &d.contig_len) != hipSuccess) return fail(PLO_ERR_HIP);
    if (upload(ix->owned, h.chrom_len, &d.chrom_len) != hipSuccess) return fail(PLO_ERR_HIP);
    d.n_contigs = desc->n_contigs;
    d.n_segments = desc->n_segments;
    d.n_chroms = desc->n_chroms;
    // sequences: copied from the host or borrowed from the device
    std::vector<const uint8_t *> chrom_ptr(desc->n_chroms, nullptr), rev_ptr(desc->n_contigs, nullptr);
    for (uint32_t c = 0; c < desc->n_chroms; ++c) {
        const uint8_t *src = desc->chrom_seq[c];
        if (!src && desc->chrom_len[c] > 0) return fail(PLO_ERR_INVALID_ARG);
        if (desc->seq_mem == PLO_MEM_DEVICE) {
            chrom_ptr[c] = src;
        } else {
            void *p = nullptr;
            size_t bytes = std::max<size_t>((size_t)desc->chrom_len[c], 16);
            if (hipMalloc(&p, bytes) != hipSuccess) return fail(PLO_ERR_OUT_OF_MEMORY);
            ix->owned.push_back(p);
            if (desc->chrom_len[c] > 0 && hipMemcpy(p, src, (size_t)desc->chrom_len[c], hipMemcpyHostToDevice) != hipSuccess)
                return fail(PLO_ERR_HIP);
            chrom_ptr[c] = (const uint8_t *)p;
        }
    }
    for (uint32_t c = 0; c < desc->n_contigs; ++c) {
        const uint8_t *src = desc->rev_contig_seq ? desc->rev_contig_seq[c] : nullptr;
        if (!src) continue;
        if (desc->seq_mem == PLO_MEM_DEVICE) {
            rev_ptr[c] = src;
        } else {
            void *p = nullptr;
            size_t bytes = std::max<size_t>((size_t)desc->contig_len[c], 16);
            if (hipMalloc(&p, bytes) != hipSuccess) return fail(PLO_ERR_OUT_OF_MEMORY);
            ix->owned.push_back(p);
            if (desc->contig_len[c] > 0 && hipMemcpy(p, src, (size_t)desc->contig_len[c], hipMemcpyHostToDevice) != hipSuccess)
                return fail(PLO_ERR_HIP);
            rev_ptr[c] = (const uint8_t *)p;
        }
    }
    if (upload(ix->owned, chrom_ptr, &d.chrom_seq) != hipSuccess) return fail(PLO_ERR_HIP);
    if (upload(ix->owned, rev_ptr, &d.contig_revseq) != hipSuccess) return fail(PLO_ERR_HIP);
    *out = ix;
    return PLO_OK;
}

void plo_index_destroy(plo_index *ix) {
    if (!ix) return;
    (void)hipSetDevice(ix->device);
    for (void *p : ix->owned) (void)hipFree(p);
    delete ix;
}

plo_status plo_index_segment_map(const plo_index *ix, uint32_t g, uint32_t cap, int64_t *keys, int64_t *vals, uint32_t *n) {
    if (!ix || !n || g >= ix->d.n_segments) return PLO_ERR_INVALID_ARG;
    uint32_t k0 = ix->host.cs_kv_off[g], k1 = ix->host.cs_kv_off[g + 1];
    *n = k1 - k0;
    if (keys && vals && k1 > k0) {
        if (hipSetDevice(ix->device) != hipSuccess) return PLO_ERR_NO_DEVICE;
        std::vector<KV> tmp(k1 - k0);  // the maps live on the device only
        if (hipMemcpy(tmp.data(), ix->d.kv + k0, tmp.size() * sizeof(KV), hipMemcpyDeviceToHost) != hipSuccess) return PLO_ERR_HIP;
        for (uint32_t i = 0; i < std::min(cap, k1 - k0); ++i) {
            keys[i] = tmp[i].key;
            vals[i] = tmp[i].val == NONE32 ? INT64_MIN : (int64_t)tmp[i].val;
        }
    }
    return PLO_OK;
}

plo_status plo_ctx_create(const plo_index *ix, void *hip_stream, plo_ctx **out) {
    if (!ix || !out) return PLO_ERR_INVALID_ARG;
    *out = nullptr;
    if (hipSetDevice(ix->device) != hipSuccess) return PLO_ERR_NO_DEVICE;
    plo_ctx *c = new plo_ctx();
    c->ix = ix;
    if (hip_stream) {
        c->stream = (hipStream_t)hip_stream;
    } else {
        if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
            delete c;
            return PLO_ERR_HIP;
        }
        c->own_stream = true;
    }
    if (hipEventCreateWithFlags(&c->ev_cls, hipEventDisableTiming) != hipSuccess) {
        if (c->own_stream) (void)hipStreamDestroy(c->stream);
        delete c;
        return PLO_ERR_HIP;
    }
    for (int i = 0; i < 7; ++i)
        if (hipEventCreate(&c->ev[i]) != hipSuccess) {
            for (int k = 0; k < i; ++k) (void)hipEventDestroy(c->ev[k]);
            if (c->own_stream) (void)hipStreamDestroy(c->stream);
            delete c;
            return PLO_ERR_HIP;
        }
    {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, ix->device) == hipSuccess && prop.multiProcessorCount > 0) c->n_cus = prop.multiProcessorCount;
    }
    (void)hipFuncSetAttribute((const void *)k_lift_tiles, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void *)k_lift_tiles_sp, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void *)k_lift_tiles_c256, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void *)k_lift_retry_sp, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void *)k_lift_mid_sp<8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void *)k_lift_mid_sp<16>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (getenv("PLO_DEBUG")) {
        for (int cap : {256, 320, 384, 512}) {
            int nb = -1;
            size_t lds = ((tile_mem_bytes(cap) + 15) & ~(size_t)15) * TILE_WAVES;
            hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void *)k_lift_tiles, TILE_WAVES * 64, lds);
            fprintf(stderr, "[plo] cap %d: dynamic LDS %zu B/block -> %d blocks/CU (%s)\n", cap, lds, nb, hipGetErrorString(e));
        }
    }
    (void)hipFuncSetAttribute((const void *)k_lift_mid<8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void *)k_lift_mid<16>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (const char *e = getenv("PLO_MID_WAVES")) {
        int v = atoi(e);
        c->mid_waves = v >= 16 ? 16 : (v >= 8 ? 8 : 0);
    }
    if (const char *e = getenv("PLO_MID_CAP")) c->mid_cap_max = std::min(4096, std::max(256, atoi(e) & ~63));
    (void)hipFuncSetAttribute((const void *)k_lift_retry, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void *)k_lift_lanes, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void *)k_lift_lanes_sp, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void *)k_lift_lanes_stats, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void *)k_lift_lanes16, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void *)k_lift_lanes16_stats, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (const char *e = getenv("PLO_LANE_MAX_W")) c->lane_max_w = atoi(e);
    if (const char *e = getenv("PLO_LANE_SORT")) c->lane_sort = atoi(e) != 0;
    if (const char *e = getenv("PLO_LANE_STATIC")) c->lane_static_rounds = atoi(e);
    if (const char *e = getenv("PLO_LANE_STATS")) c->lane_stats = atoi(e) != 0;
    if (const char *e = getenv("PLO_LANE_H16")) c->lane_h16 = atoi(e) != 0;
    if (const char *e = getenv("PLO_LANE_TAIL")) c->lane_tail_rounds = std::max(0, atoi(e));
    if (const char *e = getenv("PLO_LANE_KVS")) c->lane_kvs = std::min(LANE_KVS_MAX, std::max(64, atoi(e) & ~63));
    if (!c->lane_h16) c->lane_kvs = std::min(c->lane_kvs, LANE_KVS);
    if (const char *e = getenv("PLO_LANE_SORT_WINDOW")) c->lane_sort_window = c->lane_budget_window = std::min(2048, std::max(64, atoi(e) & ~63));
    if (const char *e = getenv("PLO_LANE_BUDGET")) c->lane_budget = atoi(e) != 0;
    if (const char *e = getenv("PLO_LANE_HEAVY_MIN")) c->lane_heavy_min = atoi(e);
    if (const char *e = getenv("PLO_LANE_HEAVY_RATIO")) c->lane_heavy_ratio = std::max(0, atoi(e));
    if (const char *e = getenv("PLO_LANE_STREAM_RATIO")) c->lane_stream_ratio = std::max(1, atoi(e));
    if (const char *e = getenv("PLO_FAST_PATH")) c->fast = atoi(e) != 0;
    if (const char *e = getenv("PLO_SCAN_CHAIN")) c->scan_chain = atoi(e) != 0;
    if (const char *e = getenv("PLO_FAST_LAUNCH_BOUND")) c->fast_launch_bound = atoi(e) != 0;
    if (const char *e = getenv("PLO_FAST_FUSE")) c->fast_fuse = atoi(e) != 0;
    if (const char *e = getenv("PLO_PHASE_EVENTS")) c->phase_events = atoi(e) != 0;
    if (const char *e = getenv("PLO_LANE_CAPW")) c->lane_capw = std::min(40000, std::max(64, atoi(e))) & ~3;
    if (c->lane_max_w + LANE_SLACK > c->lane_capw) c->lane_max_w = c->lane_capw - LANE_SLACK;
    if (const char *e = getenv("PLO_TILE_WAVES")) c->tile_waves = std::min(TILE_WAVES, std::max(1, atoi(e)));
    if (const char *e = getenv("PLO_WINDOW")) c->window = std::max(16, atoi(e)), c->adaptive = false;
    if (const char *e = getenv("PLO_BIG_THRESH")) c->big_thresh = std::max(1, atoi(e)), c->adaptive = false;
    if (const char *e = getenv("PLO_CAP")) c->cap = std::max(64, atoi(e)), c->adaptive = false;
    *out = c;
    return PLO_OK;
}

void plo_ctx_destroy(plo_ctx *c) {
    if (!c) return;
    (void)hipSetDevice(c->ix->device);
    (void)hipStreamSynchronize(c->stream);
    DevBuf *bufs[] = {&c->f_flag, &c->f_bin, &c->f_end, &c->f_prim, &c->f_isoff, &c->f_iqoff, &c->f_iread, &c->f_nl, &c->f_pitem,
                      &c->f_uflag, &c->f_rsoff, &c->f_rqoff, &c->f_su, &c->f_qu, &c->f_soff, &c->f_qoff, &c->f_rseq, &c->f_rqual, &c->f_fflag, &c->f_frank, &c->f_flist, &c->sa_len, &c->sa_off, &c->sa_text,
                      &c->item_region, &c->lane_groups, &c->lane_ticket, &c->misc, &c->whist, &c->cls_partial, &c->lane_scratch, &c->item_cls, &c->retry_list, &c->perm, &c->nin_p, &c->seg_reflen, &c->seg_readlen, &c->seg_nm, &c->seg_cnt, &c->seg_off, &c->scan_partial, &c->item_seg, &c->item_cseg, &c->item_nin, &c->op_prefix,
                      &c->counters, &c->big_list, &c->huge_list, &c->verr, &c->scratch, &c->tile_lo, &c->d_n_m, &c->d_in_off, &c->d_n_in, &c->d_pos1,
                      &c->d_w0, &c->d_w1, &c->d_kv0, &c->d_kv1, &c->d_flags, &c->d_contig, &c->d_seq_len, &c->d_seq_off, &c->d_shift_ref,
                      &c->d_shift_ref_len, &c->d_chrom_ref, &c->d_chrom_ref_len, &c->d_read_len, &c->o_status, &c->o_flip, &c->o_mapq, &c->o_chrom, &c->o_pos,
                      &c->o_coff, &c->o_clen, &c->o_cigar, &c->o_dense_off, &c->o_cigar_dense, &c->wave_stats, &c->fast_blk, &c->i_read_rev, &c->i_read_len, &c->i_read_off, &c->i_seq,
                      &c->i_seg_read, &c->i_seg_contig, &c->i_seg_pos, &c->i_seg_fwd, &c->i_seg_coff, &c->i_cigar,
                      &c->i_item_seg, &c->i_item_cseg, &c->miss_list, &c->miss_info, &c->miss_vals, &c->miss_seq_off, &c->miss_side};
    for (DevBuf *b : bufs) b->release();
    HostBuf *hb[] = {&c->h_item_seg, &c->h_item_cseg, &c->h_status, &c->h_flip, &c->h_mapq, &c->h_chrom, &c->h_pos,
                     &c->h_coff, &c->h_clen, &c->h_cigar, &c->h_counters, &c->h_miss, &c->h_side};
    for (HostBuf *b : hb) b->release();
    for (int i = 0; i < 7; ++i)
        if (c->ev[i]) (void)hipEventDestroy(c->ev[i]);
    for (int i = 0; i < 5; ++i)
        if (c->fev[i]) (void)hipEventDestroy(c->fev[i]);
    if (c->ev_seq) (void)hipEventDestroy(c->ev_seq);
    if (c->ev_cls) (void)hipEventDestroy(c->ev_cls);
    if (c->copy_stream) (void)hipStreamDestroy(c->copy_stream);
    if (c->own_stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

const char *plo_last_error(const plo_ctx *c) { return c ? c->err.c_str() : g_index_err.c_str(); }

void *plo_ctx_stream(plo_ctx *c) { return c ? (void *)c->stream : nullptr; }
int plo_ctx_device(plo_ctx *c) { return c ? c->ix->device : -1; }

plo_status plo_ctx_set_stats(plo_ctx *c, int on) {
    if (!c) return PLO_ERR_INVALID_ARG;
    c->lane_stats = on != 0;
    return PLO_OK;
}

plo_status plo_ctx_set_phase_events(plo_ctx *c, int on) {
    if (!c) return PLO_ERR_INVALID_ARG;
    c->phase_events = on != 0;
    return PLO_OK;
}

plo_status plo_ctx_sync(plo_ctx *c) {
    if (!c) return PLO_ERR_INVALID_ARG;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return PLO_OK;
}

// debugging aid (timing builds, -DPLO_PHASE_TIMING): shader cycles spent per pipeline phase, summed over waves
void plo_ctx_phase_cycles(plo_ctx *c, unsigned long long *out12) {
    for (int k = 0; k < 12; ++k) out12[k] = c ? c->phase_cycles[k] : 0;
}
// debugging aid: the item indices the last batch's lane kernels handed to the retry list (first `max_items` of them); returns how many were copied
unsigned plo_ctx_debug_retry_list(plo_ctx *c, unsigned *out, unsigned max_items) {
    if (!c || !out || !c->retry_list.p) return 0;
    const unsigned n = (unsigned)std::min<size_t>(std::min<size_t>(max_items, c->timing.n_retry_items), c->retry_list.cap / 4);
    if (hipSetDevice(c->ix->device) != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess) return 0;
    if (n && hipMemcpy(out, c->retry_list.p, (size_t)n * 4, hipMemcpyDeviceToHost) != hipSuccess) return 0;
    return n;
}
// debugging aid (timing builds): the raw statistic slots of the last batch's lift waves, STAT_WORDS words each -- [5] / [6] the wave's first and
// last tick of the constant 100 MHz clock, [7] its HW_ID | XCC_ID << 16 (tools/wave_timeline.py); returns the slots copied
unsigned plo_ctx_wave_clocks(plo_ctx *c, unsigned long long *out, unsigned max_slots) {
    if (!c || !out || !c->wave_stats.p) return 0;
    const unsigned n = (unsigned)std::min<size_t>(max_slots, c->wave_stats.cap / (STAT_WORDS * 8));
    if (hipSetDevice(c->ix->device) != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess) return 0;
    if (hipMemcpy(out, c->wave_stats.p, (size_t)n * STAT_WORDS * 8, hipMemcpyDeviceToHost) != hipSuccess) return 0;
    return n;
}

plo_status plo_ctx_download(plo_ctx *c, void *host_dst, const void *dev_src, size_t bytes) {
    if (!c || (bytes && (!host_dst || !dev_src))) return PLO_ERR_INVALID_ARG;
    HIP_TRY(c, hipSetDevice(c->ix->device));
    if (bytes) HIP_TRY(c, hipMemcpyAsync(host_dst, dev_src, bytes, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return PLO_OK;
}

static plo_status scan_u32(plo_ctx *c, const uint32_t *in, uint32_t n, uint32_t *out /* n+1 */) {
    uint32_t nb = (n + SCAN_BLOCK - 1) / SCAN_BLOCK;
    if (nb == 0) nb = 1;
    if ((((uintptr_t)in) | ((uintptr_t)out)) & 15u) {  // (the kernels move eight values per thread as two 16-byte accesses)
        c->err = "scan_u32: unaligned buffer";
        return PLO_ERR_INVALID_ARG;
    }
    HIP_TRY(c, c->scan_partial.ensure((size_t)nb * 4));
    uint32_t *partial = c->scan_partial.as<uint32_t>();
    hipLaunchKernelGGL(k_scan_sums, dim3(nb), dim3(SCAN_THREADS), 0, c->stream, in, n, partial);
    hipLaunchKernelGGL(k_scan_partials, dim3(1), dim3(SCAN_THREADS), 0, c->stream, partial, nb, out, n);
    hipLaunchKernelGGL(k_scan_apply, dim3(nb), dim3(SCAN_THREADS), 0, c->stream, in, n, (const uint32_t *)partial, out);
    HIP_TRY(c, hipGetLastError());
    return PLO_OK;
}

// which instantiation of the light-item kernel a call takes
typedef void (*lane_kernel_t)(DevIndex, DevBatch, DevWork, uint32_t, uint32_t, uint32_t, uint32_t, int, const uint32_t *);
static lane_kernel_t lane_kernel_of(const plo_ctx *c, uint32_t stages, bool sp) {
    if (sp) return k_lift_lanes_sp;
    const bool h16 = c->lane_h16 && (stages & PLO_STAGE_LIFTOVER) != 0u;  // (16-bit regions need LOAD's merged op codes)
    if (h16) return c->lane_stats ? k_lift_lanes16_stats : k_lift_lanes16;
    return c->lane_stats ? k_lift_lanes_stats : k_lift_lanes;
}

// The lane kernel's ticket counters (lane_tiles_persistent): this launch's -- zero, because the context's previous launch cleared it or the
// allocation did -- and the next launch's, which this one clears.  Without them (PLO_LANE_STATIC < 0, or no memory) the waves keep to fixed slots.
// `groups`, `waves`: the launch's groups (an estimate will do) and waves -- the rounds of fixed slots in front of the tickets follow them unless
// PLO_LANE_STATIC names a number: all but the last two and a half rounds.  Measured on wgs30x (10.5 rounds; EXPERIMENTS 6.2b): 1 / 4 / 8 / 9
// fixed rounds 1.126-1.147 / 1.095-1.103 / 1.085-1.097 / 1.133-1.148 ms -- neighbouring groups on neighbouring waves of one XCD share their
// descriptor, CIGAR and block-map lines in its L2, which the tickets give up; the last rounds are what levels the waves.
static void lane_ticket_arm(plo_ctx *c, DevWork &wk, uint32_t groups, uint32_t waves) {
    wk.lane_kvs = (uint32_t)c->lane_kvs;
    wk.lane_tail_rounds = (uint32_t)c->lane_tail_rounds;
    wk.lane_ticket = nullptr;
    wk.lane_ticket_next = nullptr;
    wk.lane_static_rounds = 0;
    if (c->lane_static_rounds < 0) return;
    if (!c->lane_ticket.p) {
        if (c->lane_ticket.ensure(256) != hipSuccess) return;
        if (hipMemsetAsync(c->lane_ticket.p, 0, 256, c->stream) != hipSuccess) {
            c->lane_ticket.release();
            return;
        }
    }
    uint32_t *const t = c->lane_ticket.as<uint32_t>();  // (the two counters on lines of their own)
    wk.lane_ticket = t + 32 * (c->lane_epoch & 1u);
    wk.lane_ticket_next = t + 32 * ((c->lane_epoch + 1u) & 1u);
    const uint32_t rounds = waves ? groups / waves : 0u;  // whole rounds of the launch
    wk.lane_static_rounds = c->lane_static_rounds > 0 ? (uint32_t)c->lane_static_rounds : std::max(1u, rounds > 2u ? rounds - 2u : 1u);
    ++c->lane_epoch;
}

// the item work list and the per-item outputs of a batch of `n_items` items in the context's buffers
static void fill_work(plo_ctx *c, DevWork &wk, uint32_t n_items) {
    memset(&wk, 0, sizeof(wk));
    wk.n_items = n_items;
    wk.item_seg = c->item_seg.as<uint32_t>();
    wk.item_cseg = c->item_cseg.as<uint32_t>();
    wk.item_nin = c->item_nin.as<uint32_t>();
    wk.item_cls = c->item_cls.as<uint32_t>();
    wk.perm = c->perm.as<uint32_t>();
    wk.retry_list = c->retry_list.as<uint32_t>();
    wk.lane_max_w = c->lane_max_w;
    wk.item_op_prefix = c->op_prefix.as<uint32_t>();
    wk.d.in_off = c->d_in_off.as<uint32_t>();
    wk.d.n_in = c->d_n_in.as<uint32_t>();
    wk.d.n_m = c->d_n_m.as<uint32_t>();
    wk.d.pos1 = c->d_pos1.as<int>();
    wk.d.w0 = c->d_w0.as<uint32_t>();
    wk.d.w1 = c->d_w1.as<uint32_t>();
    wk.d.kv0 = c->d_kv0.as<uint32_t>();
    wk.d.kv1 = c->d_kv1.as<uint32_t>();
    wk.d.flags = c->d_flags.as<uint32_t>();
    wk.d.contig = c->d_contig.as<uint32_t>();
    wk.d.seq_len = c->d_seq_len.as<uint32_t>();
    wk.d.seq_off = c->d_seq_off.as<uint64_t>();
    wk.d.shift_ref = c->d_shift_ref.as<uint64_t>();
    wk.d.shift_ref_len = c->d_shift_ref_len.as<int>();
    wk.d.chrom_ref = c->d_chrom_ref.as<uint64_t>();
    wk.d.chrom_ref_len = c->d_chrom_ref_len.as<int>();
    wk.d.read_len = c->d_read_len.as<uint32_t>();
    wk.seg_readlen = c->seg_readlen.as<uint32_t>();
    wk.seg_nm = c->seg_nm.as<uint32_t>();
    wk.status = c->o_status.as<uint8_t>();
    wk.flip = c->o_flip.as<uint8_t>();
    wk.mapq = c->o_mapq.as<uint8_t>();
    wk.chrom = c->o_chrom.as<uint32_t>();
    wk.pos = c->o_pos.as<int64_t>();
    wk.cig_off = c->o_coff.as<uint64_t>();
    wk.cig_len = c->o_clen.as<uint32_t>();
    wk.counters = c->counters.as<unsigned long long>();
    wk.big_list = c->big_list.as<uint32_t>();
    wk.miss_list = c->miss_list.as<uint32_t>();
    wk.wave_stats = c->wave_stats.as<unsigned long long>();
}

static plo_status liftover_fast(plo_ctx *c, const plo_batch_in *in, uint32_t stages, plo_batch_out *out, const DevBatch &bt, bool &fallback);

plo_status plo_liftover_batch_dev(plo_ctx *c, const plo_batch_in *in, uint32_t stages, plo_batch_out *out) {
    if (!c || !in || !out) return PLO_ERR_INVALID_ARG;
    memset(out, 0, sizeof(*out));
    c->err.clear();
    HIP_TRY(c, hipSetDevice(c->ix->device));
    if (in->n_segs && (!in->seg_read || !in->seg_contig || !in->seg_pos || !in->seg_is_fwd_strand || !in->seg_cigar_off)) {
        c->err = "plo_batch_in: NULL segment array";
        return PLO_ERR_INVALID_ARG;
    }
    if (in->seq_fmt != PLO_SEQ_BAM4 && in->seq_fmt != PLO_SEQ_ASCII && in->seq_fmt != PLO_SEQ_BAM4_SPARSE) {
        c->err = "plo_batch_in: unknown seq_fmt";
        return PLO_ERR_INVALID_ARG;
    }
    DevBatch bt;
    bt.read_is_reverse = in->read_is_reverse;
    bt.read_seq_len = in->read_seq_len;
    bt.read_seq_off = in->read_seq_off;
    bt.seq = in->seq;
    bt.seq_fmt = in->seq_fmt;
    bt.seq_bytes = in->seq_bytes;
    bt.seg_read = in->seg_read;
    bt.seg_contig = in->seg_contig;
    bt.seg_pos = in->seg_pos;
    bt.seg_is_fwd = in->seg_is_fwd_strand;
    bt.seg_cigar_off = in->seg_cigar_off;
    bt.cigar = in->cigar;
    bt.n_reads = in->n_reads;
    bt.n_segs = in->n_segs;
    const DevIndex &ix = c->ix->d;
    const bool sp = in->seq_fmt == PLO_SEQ_BAM4_SPARSE;  // the `_sp` kernels: probes look granules up
    hipStream_t st = c->stream;
    memset(&c->timing, 0, sizeof(c->timing));
    c->fast_timing = false;
    c->fast_no_events = false;
    uint32_t n_syncs = 0;  // host round trips of the call
    c->ev_big = false;
    c->ev_mid = false;

    // A context that has just lifted a batch of light items only, with arrays that hold this one too: everything is launched with counts
    // read from device memory and the host looks ONCE, at the end (liftover_fast); a batch that turns out not to fit -- more items than
    // the arrays hold, heavy items, an overflow -- is run again on the path below, which asks for the counts first
    if (c->fast && c->fast_light_only && !in->item_seg && !sp && in->n_segs && in->n_segs <= c->fast_ns_cap && c->fast_item_cap && c->lane_max_w >= 0 && c->lane_sort &&
        !c->lane_budget && c->lane_sort_window <= 128 && c->o_cigar.cap) {
        bool fallback = false;
        plo_status fs = liftover_fast(c, in, stages, out, bt, fallback);
        if (!fallback) return fs;
        c->fast_light_only = false;
        c->fast_extra_syncs = 1;
        memset(out, 0, sizeof(*out));
        memset(&c->timing, 0, sizeof(c->timing));
    }
    HIP_TRY(c, hipEventRecord(c->ev[0], st));
    // ---- items: count -> scan -> resolve descriptors -> scan op counts -> tile bounds ----
    uint32_t n_items = 0;
    HIP_TRY(c, c->h_counters.ensure(64 * 8 + CNT_N * 8 + 64));
    uint32_t ns = in->n_segs;
    HIP_TRY(c, c->verr.ensure(16));
    HIP_TRY(c, hipMemsetAsync(c->verr.p, 0, 16, st));
    auto verr_status = [&](uint32_t flags) -> plo_status {
        if (flags & VERR_INDEX) {
            c->err = "plo_batch_in: an index points outside its array (seg_read / seg_contig / seg_cigar_off / read_seq_off / item_seg / item_cseg)";
            return PLO_ERR_INVALID_ARG;
        }
        c->err = "plo_batch_in: coordinate outside the 31-bit BAM range, CIGAR op code above 8, or a CIGAR spanning more than 2^30 bases";
        return PLO_ERR_RANGE;
    };
    {   // k_seg_count always runs: it is the boundary check of the batch (and counts the items unless the caller lists them)
        HIP_TRY(c, c->seg_cnt.ensure((size_t)std::max(1u, ns) * 4));
        HIP_TRY(c, c->seg_off.ensure((size_t)(ns + 1) * 4));
        HIP_TRY(c, c->seg_reflen.ensure((size_t)std::max(1u, ns) * 4));
        HIP_TRY(c, c->seg_readlen.ensure((size_t)std::max(1u, ns) * 4));
        HIP_TRY(c, c->seg_nm.ensure((size_t)std::max(1u, ns) * 4));
        if (ns)
            hipLaunchKernelGGL(k_seg_count, dim3((unsigned)(((unsigned long long)ns * SEG_LANES + 255) / 256)), dim3(256), 0, st, ix, bt,
                               c->seg_cnt.as<uint32_t>(), c->seg_reflen.as<int>(), c->seg_readlen.as<uint32_t>(), c->seg_nm.as<uint32_t>(), c->verr.as<uint32_t>());
        plo_status s = scan_u32(c, c->seg_cnt.as<uint32_t>(), ns, c->seg_off.as<uint32_t>());
        if (s != PLO_OK) return s;
        uint32_t *h = c->h_counters.as<uint32_t>();
        HIP_TRY(c, hipMemcpyAsync(h, c->seg_off.as<uint32_t>() + ns, 4, hipMemcpyDeviceToHost, st));
        HIP_TRY(c, hipMemcpyAsync(h + 1, c->verr.p, 4, hipMemcpyDeviceToHost, st));
        HIP_TRY(c, hipStreamSynchronize(st)); ++n_syncs;
        if (h[1]) return verr_status(h[1]);
        n_items = in->item_seg ? in->n_items : h[0];
    }
    size_t ni = std::max(1u, n_items);
    HIP_TRY(c, c->item_seg.ensure(ni * 4));
    HIP_TRY(c, c->item_cseg.ensure(ni * 4));
    HIP_TRY(c, c->item_nin.ensure(ni * 4));
    HIP_TRY(c, c->item_cls.ensure(ni * 4));
    HIP_TRY(c, c->retry_list.ensure(ni * 4));
    HIP_TRY(c, c->perm.ensure(ni * 4));
    HIP_TRY(c, c->nin_p.ensure(ni * 4));
    HIP_TRY(c, c->op_prefix.ensure((size_t)(n_items + 1) * 4));
    HIP_TRY(c, c->d_in_off.ensure(ni * 4));
    HIP_TRY(c, c->d_n_in.ensure(ni * 4));
    HIP_TRY(c, c->d_n_m.ensure(ni * 4));
    HIP_TRY(c, c->d_pos1.ensure(ni * 4));
    HIP_TRY(c, c->d_w0.ensure(ni * 4));
    HIP_TRY(c, c->d_w1.ensure(ni * 4));
    HIP_TRY(c, c->d_kv0.ensure(ni * 4));
    HIP_TRY(c, c->d_kv1.ensure(ni * 4));
    HIP_TRY(c, c->d_flags.ensure(ni * 4));
    HIP_TRY(c, c->d_contig.ensure(ni * 4));
    HIP_TRY(c, c->d_seq_len.ensure(ni * 4));
    HIP_TRY(c, c->d_seq_off.ensure(ni * 8));
    HIP_TRY(c, c->d_shift_ref.ensure(ni * 8));
    HIP_TRY(c, c->d_shift_ref_len.ensure(ni * 4));
    HIP_TRY(c, c->d_chrom_ref.ensure(ni * 8));
    HIP_TRY(c, c->d_chrom_ref_len.ensure(ni * 4));
    HIP_TRY(c, c->d_read_len.ensure(ni * 4));
    HIP_TRY(c, c->o_status.ensure(ni));
    HIP_TRY(c, c->o_flip.ensure(ni));
    HIP_TRY(c, c->o_mapq.ensure(ni));
    HIP_TRY(c, c->o_chrom.ensure(ni * 4));
    HIP_TRY(c, c->o_pos.ensure(ni * 8));
    HIP_TRY(c, c->o_coff.ensure(ni * 8));
    HIP_TRY(c, c->o_clen.ensure(ni * 4));
    HIP_TRY(c, c->big_list.ensure(ni * 4));
    if (in->seq_fmt == PLO_SEQ_BAM4_SPARSE) HIP_TRY(c, c->miss_list.ensure(ni * 4));
    HIP_TRY(c, c->counters.ensure(CNT_N * 8));
    {   // items the per-item arrays hold (they grow with head-room): the one-round-trip path's capacity (the last batch's item
        // count refused a window with one item more and cost it two passes)
        size_t cap_items = (size_t)0xffffffffu;
        const struct { const DevBuf *b; size_t el; } arrs[] = {
            {&c->item_seg, 4}, {&c->item_cseg, 4}, {&c->item_nin, 4}, {&c->item_cls, 4}, {&c->retry_list, 4}, {&c->perm, 4}, {&c->nin_p, 4},
            {&c->d_in_off, 4}, {&c->d_n_in, 4}, {&c->d_n_m, 4}, {&c->d_pos1, 4}, {&c->d_w0, 4}, {&c->d_w1, 4}, {&c->d_kv0, 4}, {&c->d_kv1, 4},
            {&c->d_flags, 4}, {&c->d_contig, 4}, {&c->d_seq_len, 4}, {&c->d_seq_off, 8}, {&c->d_shift_ref, 8}, {&c->d_shift_ref_len, 4},
            {&c->d_chrom_ref, 8}, {&c->d_chrom_ref_len, 4}, {&c->d_read_len, 4}, {&c->o_status, 1}, {&c->o_flip, 1}, {&c->o_mapq, 1},
            {&c->o_chrom, 4}, {&c->o_pos, 8}, {&c->o_coff, 8}, {&c->o_clen, 4}, {&c->big_list, 4}};
        for (const auto &a : arrs) cap_items = std::min(cap_items, a.b->cap / a.el);
        cap_items = std::min(cap_items, c->op_prefix.cap / 4 ? c->op_prefix.cap / 4 - 1 : 0);
        c->item_cap_now = (uint32_t)std::min<size_t>(cap_items, 0x7fffffffu);
    }
    DevWork wk;
    fill_work(c, wk, n_items);
    {   // statistic slots of the lift kernels' waves: zeroed once, cleared again by every k_sum_stats
        const size_t want = (size_t)STAT_SLOTS * STAT_WORDS * 8;
        if (c->wave_stats.cap < want) {
            HIP_TRY(c, c->wave_stats.ensure(want));
            HIP_TRY(c, hipMemsetAsync(c->wave_stats.p, 0, want, st));
        }
    }
    wk.wave_stats = c->wave_stats.as<unsigned long long>();
    // every lift launch of a batch leaves its waves' statistics in its own range of slots (PLO_STAT_RANGE before the launch); they
    // are added up once before the host reads the counters (PLO_READ_COUNTERS), not after every kernel
    uint32_t stat_used = 0;
#define PLO_STAT_RANGE(n_waves_)     \
    do {                             \
        wk.stat_base = stat_used;    \
        stat_used += (n_waves_);     \
    } while (0)
#define PLO_READ_COUNTERS()                                                                                                          \
    do {                                                                                                                             \
        if (stat_used)                                                                                                               \
            hipLaunchKernelGGL(k_sum_stats, dim3(std::min<uint32_t>((stat_used + 255) / 256, 16u)), dim3(256), 0, st,                 \
                               c->wave_stats.as<unsigned long long>(), stat_used, c->counters.as<unsigned long long>());            \
        stat_used = 0;                                                                                                               \
        HIP_TRY(c, hipMemcpyAsync(hc, c->counters.p, CNT_N * 8, hipMemcpyDeviceToHost, st));                                         \
        HIP_TRY(c, hipStreamSynchronize(st)); ++n_syncs;                                                                                        \
    } while (0)
    wk.big_list = c->big_list.as<uint32_t>();
    if (c->lane_sort && c->lane_budget && c->lane_max_w >= 0) {
        HIP_TRY(c, c->item_region.ensure(ni * 4));
        wk.item_region = c->item_region.as<uint32_t>();
    }
    wk.miss_list = c->miss_list.as<uint32_t>();
    if (n_items) {
        if (in->item_seg)
            hipLaunchKernelGGL(k_item_desc, dim3((n_items + 255) / 256), dim3(256), 0, st, ix, bt, wk, stages, in->item_seg,
                               in->item_cseg, c->verr.as<uint32_t>());
        else
            hipLaunchKernelGGL(k_item_emit, dim3((ns + 255) / 256), dim3(256), 0, st, ix, bt, wk, stages,
                               (const uint32_t *)c->seg_off.as<uint32_t>(), c->seg_reflen.as<int>(), 0xffffffffu, c->verr.as<uint32_t>());
        HIP_TRY(c, hipGetLastError());
    }
    // ---- class order: block counts -> scan -> permutation (three launches); the host learns the class counts and the weights'
    // sum / maximum from one copy, made while k_permute2 runs
    uint32_t total_ops = 0, max_nin = 0, n_small = 0, h_cls[3] = {0, 0, 0};
    unsigned long long all_ops = 0;
    const uint32_t cls_nb = (n_items + CLS_BLOCK - 1) / CLS_BLOCK;
    {
        HIP_TRY(c, c->misc.ensure(256));
        HIP_TRY(c, hipMemsetAsync(c->misc.p, 0, 256, st));
        HIP_TRY(c, c->cls_partial.ensure((size_t)std::max(1u, cls_nb) * 6 * 4));
        uint32_t *m_ = c->h_counters.as<uint32_t>() + 64;  // clear of h[0..47] below
        memset(m_, 0, 8 * 4);
        if (n_items) {
            hipLaunchKernelGGL(k_cls_hist, dim3(cls_nb), dim3(CLS_THREADS), 0, st, (const uint32_t *)c->item_cls.as<uint32_t>(),
                               (const uint32_t *)c->item_nin.as<uint32_t>(), n_items, cls_nb, c->cls_partial.as<uint32_t>(), (const uint32_t *)nullptr);
            hipLaunchKernelGGL(k_cls_scan, dim3(1), dim3(SCAN_THREADS), 0, st, c->cls_partial.as<uint32_t>(), cls_nb, c->misc.as<uint32_t>());
            HIP_TRY(c, hipMemcpyAsync(m_, c->misc.p, 6 * 4, hipMemcpyDeviceToHost, st));
            HIP_TRY(c, hipMemcpyAsync(m_ + 6, c->verr.p, 4, hipMemcpyDeviceToHost, st));
            HIP_TRY(c, hipEventRecord(c->ev_cls, st));
            hipLaunchKernelGGL(k_permute2, dim3(cls_nb), dim3(CLS_THREADS), 0, st, (const uint32_t *)c->item_cls.as<uint32_t>(),
                               (const uint32_t *)c->item_nin.as<uint32_t>(), (const uint32_t *)c->cls_partial.as<uint32_t>(),
                               (const uint32_t *)c->misc.as<uint32_t>(), n_items, cls_nb, c->perm.as<uint32_t>(), c->nin_p.as<uint32_t>(),
                               c->adaptive ? (uint32_t)((WHIST_BINS - 1) * WHIST_STEP) : 0xffffffffu, (const uint32_t *)nullptr);
            HIP_TRY(c, hipGetLastError());
            HIP_TRY(c, hipEventSynchronize(c->ev_cls)); ++n_syncs;  // (the copies are done; k_permute2 may still be running)
        }
        if (m_[6]) return verr_status(m_[6]);
        h_cls[0] = m_[0];
        h_cls[1] = m_[1];
        h_cls[2] = m_[2];
        n_small = m_[0] + m_[1];
        max_nin = m_[3];
        all_ops = (unsigned long long)m_[4] | ((unsigned long long)m_[5] << 32);
        if (all_ops > 0x7fffffffull) {  // op indices are int: the weights bound the ops of every stage, output included
            c->err = "batch too large: the item weights (CIGAR ops + 2 x block-map entries) sum to more than 2^31; split the batch";
            return PLO_ERR_RANGE;
        }
    }
    // ---- items too heavy for an LDS region.  Many of them: the lane-per-item code again, regions in global scratch (k_lift_lanes_g).
    // A few: tiles of the wave-cooperative kernel (weight prefix, per-batch geometry).
    // The lane kernel takes as long as its longest item takes a wave (about 3.5 us per op), however many items there are; the
    // workgroup-per-item kernel takes about 70 ps per op of the batch.  So: the lanes when the batch's ops (weights) outnumber the
    // longest item's by lane_heavy_ratio -- 50 000, the measured crossing on the stress profile (67 k reads of 2 000 ops, longest
    // 2 800 + 240 block-map allowance).  PLO_LANE_HEAVY_MIN: an item count instead (0: always).
    const uint32_t n_heavy_all = n_items - n_small;
    // The streaming kernel (a team of waves per 64 item slots, lane_stream.hpp; all stages only) has a third of the lane kernel's latency per
    // item -- its stages run side by side -- at about the same instructions per op: measured on the stress profile it is the fastest of the
    // three from ~40 k heavy items (50 k reads: 5.9 ms against 6.65 ms workgroup-per-item and 8.6 ms k_lift_lanes_g) to ~200 k (100 k reads:
    // 9.1 against 9.5 ms; 500 k: 30.6 against 25.9 ms k_lift_lanes_g_w3, whose waves each hold several groups by then).  So: the
    // batch's ops against its longest item's by lane_stream_ratio (30 000) for the lower bound, the three-waves-per-SIMD rule of
    // k_lift_lanes_g_w3 for the upper.  PLO_LANE_STREAM=0 / 1: never / whenever the lane path is taken.
    bool stream_ok = (stages & PLO_STAGES_ALL) == PLO_STAGES_ALL && c->lane_stream && n_items < (1u << 28);
    bool stream_forced = false;
    if (const char *e = getenv("PLO_LANE_STREAM")) {
        stream_ok = stream_ok && atoi(e) != 0;
        stream_forced = stream_ok;
    }
    const bool stream_size = n_heavy_all >= 8192u && n_heavy_all < (uint32_t)c->n_cus * 3u * LANE_G_WAVES * 64u &&
                             all_ops >= (unsigned long long)c->lane_stream_ratio * std::min<unsigned long long>(max_nin, 16384ull);
    const bool heavy_lanes =
        n_heavy_all > 0 && c->lane_max_w >= 0 &&
        (c->lane_heavy_min >= 0 ? n_heavy_all >= (uint32_t)c->lane_heavy_min
                                : ((stream_ok && stream_size) ||
                                   (c->lane_heavy_ratio > 0 && n_heavy_all >= 8192u &&
                                    all_ops >= (unsigned long long)c->lane_heavy_ratio * std::min<unsigned long long>(max_nin, 16384ull))));
    const bool heavy_stream = heavy_lanes && stream_ok && (stream_forced || stream_size || c->lane_heavy_min >= 0);
    if (n_items > n_small && !heavy_lanes) {
        plo_status s = scan_u32(c, c->nin_p.as<uint32_t>(), n_items, c->op_prefix.as<uint32_t>());
        if (s != PLO_OK) return s;
        HIP_TRY(c, c->whist.ensure(256));
        HIP_TRY(c, hipMemsetAsync(c->whist.p, 0, 256, st));
        hipLaunchKernelGGL(k_max_u32, dim3(std::min<uint32_t>((n_items + 255) / 256, 256u)), dim3(256), 0, st,
                           (const uint32_t *)c->item_nin.as<uint32_t>(), n_items, c->whist.as<uint32_t>(),
                           (const uint32_t *)c->op_prefix.as<uint32_t>() + n_items, (const uint32_t *)c->misc.as<uint32_t>(),
                           (const uint32_t *)c->misc.as<uint32_t>() + 1);
        // one copy: [0] max weight, [2..3] weight sum, [4] tiled weight, [8..] weight histogram
        uint32_t *m_ = c->h_counters.as<uint32_t>() + 64;
        HIP_TRY(c, hipMemcpyAsync(m_, c->whist.p, (WHIST_AT + WHIST_BINS) * 4, hipMemcpyDeviceToHost, st));
        HIP_TRY(c, hipStreamSynchronize(st)); ++n_syncs;
        uint32_t *h = c->h_counters.as<uint32_t>();
        for (int k = 0; k < WHIST_BINS; ++k) h[8 + k] = m_[WHIST_AT + k];
        total_ops = m_[4];  // weight of the tiled items
        if (c->adaptive) {
            // Tile geometry from the batch's weight distribution: the routing threshold covers all but 0.2 % of the items
            // (those take the large-item kernel), the LDS slice holds one window plus the overhang of its last item.  Dense
            // contig block maps (many blocks per read) thus get larger slices and fewer resident waves instead of a
            // large-item kernel that runs every read alone from global scratch.
            const uint32_t *hist = h + 8;
            // items in the open-ended last bin take the large-item kernel whatever the geometry: the threshold is chosen for the rest
            unsigned long long above = 0, allow = (n_items - std::min<uint32_t>(n_items, hist[WHIST_BINS - 1])) / 500;
            int b = WHIST_BINS - 2;
            for (; b > 0; --b) {  // lowest threshold (b * WHIST_STEP) with at most `allow` items above it
                if (above + hist[b] > allow) break;
                above += hist[b];
            }
            int thresh = std::min(std::max(176, (b + 1) * WHIST_STEP), (WHIST_BINS - 1) * WHIST_STEP);
            c->big_thresh = thresh;
            // up to a threshold of 256 the 320-element slice (12 waves per CU) stays: the few tiles it cannot hold are re-run
            // item by item (k_lift_retry); measured on MI355X, 1 M reads, contig indel rate 1e-3: 4.3 ms against 5.3-7.2 ms
            // with 384-element slices.  Workgroup width per slice size as measured (tools/tune.py --contig-indel).
            // Lowest threshold (no heavy tail at all): the 256-element slice, whose kernel has the capacity compiled in and runs 16
            // waves per CU (wgs30x, 2 M reads: 2.28 ms against 2.43 ms with 320-element slices at 12 waves per CU).
            c->cap = thresh <= 176 && !sp && !getenv("PLO_NO_SMALL_CAP") ? TILE_CAP_SMALL : (thresh <= 256 ? 320 : (thresh + 144 + 63) & ~63);
            // the slice holds one window plus the overhang of its last item; weights count two units per block-map entry, ops only one,
            // so the small slice gets by with half the allowance (measured: window 192 2.27 ms, 208 2.22 ms, 224 2.21 ms with 204 of
            // 2 M items re-run) -- until a batch re-runs more than 0.5 % of its items
            c->window = c->cap - (c->cap == TILE_CAP_SMALL && !c->small_window_tight ? 32 : 64);
            c->tile_waves = c->cap <= 320 ? TILE_WAVES : (c->cap <= 448 ? 2 : 1);
            if (getenv("PLO_DEBUG_GEOMETRY"))
                fprintf(stderr, "[plo] batch geometry: thresh %d cap %d window %d tile_waves %d (max weight %u, %llu items beyond the 0.2 %% cut)\n", thresh, c->cap,
                        c->window, c->tile_waves, max_nin, above);
        }
    }
    const uint32_t n_tiles = total_ops / (uint32_t)c->window + 1;
    if (n_items > n_small && !heavy_lanes) {
        HIP_TRY(c, c->tile_lo.ensure((size_t)(n_tiles + 1) * 4));
        hipLaunchKernelGGL(k_tile_bounds, dim3((n_tiles + 1 + 255) / 256), dim3(256), 0, st, (const uint32_t *)c->op_prefix.as<uint32_t>(),
                           n_items, n_tiles, c->window, n_small, c->tile_lo.as<uint32_t>());
    }
    wk.tile_lo = c->tile_lo.as<uint32_t>();
    wk.n_small = n_small;
    HIP_TRY(c, hipEventRecord(c->ev[1], st));

    // launch geometry of the lane-per-item kernel (light items), needed here already: its waves own one output slab each from the start
    uint32_t lane_gs = 64, lane_nblk = 0;
    const size_t lane_lds = (size_t)(c->lane_capw + 2 * c->lane_kvs) * 4 * LANE_WAVES;  // slices + staged block-map entries
    if (n_small) {
        int occ = 1;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, sp ? (const void *)k_lift_lanes_sp : (const void *)k_lift_lanes, LANE_WAVES * 64, lane_lds) != hipSuccess || occ < 1)
            occ = 1;
        // items per group: 64, fewer when that leaves resident waves without a group (small batches; the few light items of an
        // indel-dense batch, whose regions would also take a wave's LDS slice several rounds at 64 a group)
        const uint32_t n0 = h_cls[0], n1 = h_cls[1];
        const uint32_t slots = (uint32_t)(c->n_cus * occ) * LANE_WAVES;
        while (lane_gs > 8u && (n0 + lane_gs / 2 - 1) / (lane_gs / 2) + (n1 + lane_gs / 2 - 1) / (lane_gs / 2) <= slots) lane_gs >>= 1;
        if (const char *e = getenv("PLO_LANE_GROUP")) lane_gs = std::min(64u, std::max(1u, (uint32_t)atoi(e)));
        const uint32_t groups = (n0 + lane_gs - 1) / lane_gs + (n1 + lane_gs - 1) / lane_gs;
        lane_nblk = std::min<uint32_t>((groups + LANE_WAVES - 1) / LANE_WAVES, (uint32_t)(c->n_cus * occ));
        lane_nblk = (lane_nblk + 7u) & ~7u;
    }
    size_t want_cigar = (size_t)all_ops * 2 + (size_t)n_items * 8 + 4096 + (size_t)std::min<uint32_t>(n_tiles + 1024, (uint32_t)c->n_cus * 16) * SLAB_OPS +
                        (size_t)lane_nblk * LANE_WAVES * SLAB_OPS;  // (exactly the slabs the lane kernel's waves pre-own: slab_offset below)
    if (c->o_cigar.cap < want_cigar * 4) HIP_TRY(c, c->o_cigar.ensure(want_cigar * 4));

    if (c->seq_pending) {  // (plo_liftover_batch: the read bases arrive on the copy stream)
        HIP_TRY(c, hipStreamWaitEvent(st, c->ev_seq, 0));
        c->seq_pending = false;
    }
    unsigned long long *hc = c->h_counters.as<unsigned long long>();
    uint32_t n_big = 0, n_retry = 0, n_mid = 0, n_huge = 0, n_miss = 0, heavy_kernel = 0;
    float miss_ms = 0.f;
    for (int attempt = 0;; ++attempt) {
        wk.out_cigar = c->o_cigar.as<uint32_t>();
        wk.out_cap = c->o_cigar.cap / 4;
        HIP_TRY(c, hipMemsetAsync(c->counters.p, 0, CNT_N * 8, st));
        if (attempt == 0) HIP_TRY(c, hipEventRecord(c->ev[1], st));
        wk.slab_pre = 0;
        wk.slab_offset = 0;
        if (n_small) {
            const uint32_t n0 = h_cls[0], n1 = h_cls[1];
            // groups cut by LDS budget (wider sort windows without the extra rounds of over-full groups) when the groups are of 64
            const bool budget = c->lane_sort && wk.item_region && lane_gs == 64u;
            uint32_t lane_groups_cap_now = 0;
            if (attempt == 0 && c->lane_sort) {
                const uint32_t ch = (uint32_t)(budget ? c->lane_budget_window : c->lane_sort_window);
                const uint32_t chunks = (n0 + ch - 1) / ch + (n1 + ch - 1) / ch;
                uint32_t *n_groups_dev = c->misc.as<uint32_t>() + 32;  // (misc was cleared before the class kernels)
                if (budget) {
                    // most groups a window can have: every group holds at least cap / (largest region) items
                    // (k_chunk_sort cuts a group at 64 items as well as at the slice's capacity)
                    const uint32_t per_group = std::min<uint32_t>(64u, std::max<uint32_t>(1u, (uint32_t)c->lane_capw / (uint32_t)std::max(1, c->lane_max_w)));
                    const size_t max_groups = (size_t)chunks * (ch / per_group + 2);
                    lane_groups_cap_now = (uint32_t)std::min<size_t>(max_groups, 0xffffffffu);
                    HIP_TRY(c, c->lane_groups.ensure(max_groups * 8));
                    if (getenv("PLO_DEBUG_GEOMETRY"))
                        fprintf(stderr, "[plo] lane groups cut by LDS budget: windows of %u, slices of %d dwords, at most %zu groups\n", ch, c->lane_capw, max_groups);
                }
                if (!budget && ch <= 128u)
                    hipLaunchKernelGGL(k_chunk_sort_w, dim3((chunks + 3u) / 4u), dim3(256), 0, st, c->perm.as<uint32_t>(), (const uint32_t *)c->d_n_m.as<uint32_t>(),
                                       (const uint32_t *)c->d_w0.as<uint32_t>(), (const uint32_t *)c->d_w1.as<uint32_t>(), n0, n1, ch, (const uint32_t *)nullptr);
                else
                    hipLaunchKernelGGL(k_chunk_sort, dim3(chunks), dim3(LANE_SORT_THREADS), 0, st, c->perm.as<uint32_t>(), (const uint32_t *)c->d_n_m.as<uint32_t>(),
                                       (const uint32_t *)c->d_w0.as<uint32_t>(), (const uint32_t *)c->d_w1.as<uint32_t>(), n0, n1, ch,
                                       (const uint32_t *)c->item_region.as<uint32_t>(), (uint32_t)c->lane_capw,
                                       budget ? c->lane_groups.as<uint32_t>() : (uint32_t *)nullptr, n_groups_dev, lane_groups_cap_now);
            }
            wk.lane_groups = budget ? c->lane_groups.as<uint32_t>() : nullptr;
            wk.lane_n_groups = c->misc.as<uint32_t>() + 32;
            if (attempt == 0) c->lane_groups_cap = lane_groups_cap_now;
            wk.lane_groups_cap = c->lane_groups_cap;
            const size_t lds = lane_lds;
            const uint32_t gs = lane_gs, nblk = lane_nblk;
            wk.slab_pre = 1u;  // first slab by wave id
            wk.slab_offset = (unsigned long long)nblk * LANE_WAVES * SLAB_OPS;
            PLO_STAT_RANGE(nblk * LANE_WAVES);
            lane_ticket_arm(c, wk, (n0 + gs - 1) / gs + (n1 + gs - 1) / gs, nblk * LANE_WAVES);
            hipLaunchKernelGGL(lane_kernel_of(c, stages, sp), dim3(nblk), dim3(LANE_WAVES * 64), lds, st, ix, bt, wk, stages, n0, n1, gs, c->lane_capw, (const uint32_t *)nullptr);
            HIP_TRY(c, hipGetLastError());
        }
        HIP_TRY(c, hipEventRecord(c->ev[4], st));
        if (heavy_lanes) {
            // as many items per wave as it takes to give every resident wave a group (at least 8, at most 64 lanes at work)
            const uint32_t n_heavy = n_items - n_small, n2 = h_cls[2], n3 = n_heavy - n2;
            // three waves per SIMD once every one of their slots gets a full group (k_lift_lanes_g_w3 above); PLO_LANE_G_W3=0/1 forces
            bool w3 = !sp && n_heavy >= (uint32_t)c->n_cus * 3u * LANE_G_WAVES * 64u;
            if (const char *e = getenv("PLO_LANE_G_W3")) w3 = !sp && atoi(e) != 0;
            const int occ = w3 ? 3 : PLO_LANE_G_WPE;  // (4 waves a workgroup, one per SIMD)
            const uint32_t slots = (uint32_t)(c->n_cus * occ) * LANE_G_WAVES;
            uint32_t per = std::min(64u, std::max(8u, (n_heavy + slots - 1) / slots));
            if (const char *e = getenv("PLO_LANE_HEAVY_PER")) per = std::min(64u, std::max(1u, (uint32_t)atoi(e)));
            const uint32_t groups = (n2 + per - 1) / per + (n3 + per - 1) / per;
            uint32_t nblk = std::min<uint32_t>((groups + LANE_G_WAVES - 1) / LANE_G_WAVES, (uint32_t)(c->n_cus * occ));
            // all stages: the streaming kernel (lane_stream.hpp; PLO_LANE_STREAM=0: the kernel over global regions below, which also
            // takes the stage subsets) -- no scratch, every wave an equal contiguous share of either class
            const bool stream = heavy_stream;
            if (stream) {
                // Items per team: 64, one per lane -- more teams than the chip holds are started as others retire, which levels the load
                // better than lanes taking a second item does (measured, stress profile: 64 / 128 / 256 items per team 9.9 / 11.4 / 20.8 ms
                // at 100 k reads, 30.6 / 32.6 / 41 ms at 500 k: a team is as slow as its last lane, and with 80 items on 64 lanes the second
                // half of its run has a quarter of the lanes at work).  A batch too small to give every CU four teams of 64 is spread over
                // as many teams with fewer lanes each (at least 8).
                const uint32_t full = 64u * (uint32_t)c->n_cus * 4u;
                uint32_t per_team = n_heavy >= full ? 64u : std::min(64u, std::max(8u, (n_heavy + (uint32_t)c->n_cus * 4u - 1) / ((uint32_t)c->n_cus * 4u)));
                int tocc = 0;
                if (getenv("PLO_DEBUG_GEOMETRY") && hipOccupancyMaxActiveBlocksPerMultiprocessor(&tocc, sp ? (const void *)k_lift_stream_sp : (const void *)k_lift_stream, PIPE_WAVES * 64, 0) != hipSuccess) tocc = 0;
                if (const char *e = getenv("PLO_LANE_HEAVY_PER")) per_team = std::min(4096u, std::max(1u, (uint32_t)atoi(e)));
                // (every wave of the launch has a statistics slot)
                while (((unsigned long long)(n2 + per_team - 1) / per_team + (n3 + per_team - 1) / per_team) * PIPE_WAVES + stat_used > STAT_SLOTS) per_team *= 2;
                const uint32_t t0 = (n2 + per_team - 1) / per_team, t1 = (n3 + per_team - 1) / per_team;
                nblk = t0 + t1;
                if (getenv("PLO_DEBUG_GEOMETRY"))
                    fprintf(stderr, "[plo] heavy items through the streaming lane kernel: %u + %u items, %u + %u teams of %d waves (%d teams per CU), %u items per team\n", n2, n3,
                            t0, t1, PIPE_WAVES, tocc, per_team);
                wk.slab_pre = 0u;
                if (!n_small) wk.slab_offset = 0ull;
                PLO_STAT_RANGE(nblk * PIPE_WAVES);
                heavy_kernel = 3u;
                const uint32_t xcd = getenv("PLO_PIPE_XCD") ? (uint32_t)(atoi(getenv("PLO_PIPE_XCD")) != 0) : 0u;  // (measured: 12.05 against 9.21 ms with it, FETCH_SIZE +17 %: more teams than the chip holds are dispatched in order, and the remap sends the early ones to one XCD)
                if (sp) hipLaunchKernelGGL(k_lift_stream_sp, dim3(nblk), dim3(PIPE_WAVES * 64), 0, st, ix, bt, wk, n_small, n_small + n2, n_items, t0, t1, xcd);
                else hipLaunchKernelGGL(k_lift_stream, dim3(nblk), dim3(PIPE_WAVES * 64), 0, st, ix, bt, wk, n_small, n_small + n2, n_items, t0, t1, xcd);
            } else {
                // Regions start on 128-byte lines and hold the heaviest item of the batch (64: room for the liftover's gap, lane_region_gap)
                // -- unless that one is an outlier: the regions of all resident lanes together are kept within 8 GB, an item too long for
                // its region then is handed to the retry list (wave-cooperative code, whatever its size) like any other that outgrows it.
                int stride = (int)((max_nin + LANE_SLACK + 64u + LANE_REGION_PAD + 31u) & ~31u);
                {
                    const unsigned long long fit = (8ull << 30) / (4ull * per * LANE_G_WAVES * std::max(1u, nblk));
                    const int cap_stride = (int)std::max<unsigned long long>(1024ull, std::min<unsigned long long>(fit, 0x7fffffe0ull) & ~31ull);
                    stride = std::min(stride, cap_stride);
                    if (const char *e = getenv("PLO_LANE_HEAVY_STRIDE")) stride = std::max(LANE_REGION_PAD + 64, atoi(e)) & ~31;
                }
                const unsigned long long per_wave = (unsigned long long)per * (unsigned long long)stride * 4ull;
                nblk = (uint32_t)std::max<unsigned long long>(1ull, std::min<unsigned long long>(nblk, (16ull << 30) / (per_wave * LANE_G_WAVES)));
                HIP_TRY(c, c->lane_scratch.ensure((size_t)(per_wave * LANE_G_WAVES * nblk)));
                if (getenv("PLO_DEBUG_GEOMETRY"))
                    fprintf(stderr, "[plo] heavy items through the lane-per-item code: %u items, %u per wave, %u workgroups, regions of %d dwords, %.1f MB scratch\n", n_heavy, per,
                            nblk, stride, per_wave * LANE_G_WAVES * nblk / 1e6);
                wk.slab_pre = 0u;  // (a group's output is a slab of its own; the light items' waves own the first slabs, if any)
                if (!n_small) wk.slab_offset = 0ull;
                PLO_STAT_RANGE(nblk * LANE_G_WAVES);
                heavy_kernel = w3 ? 2u : 1u;
                if (sp) hipLaunchKernelGGL(k_lift_lanes_g_sp, dim3(nblk), dim3(LANE_G_WAVES * 64), 0, st, ix, bt, wk, stages, n_small, n_small + n2, n_items, per, c->lane_scratch.as<uint32_t>(), stride);
                else if (w3) hipLaunchKernelGGL(k_lift_lanes_g_w3, dim3(nblk), dim3(LANE_G_WAVES * 64), 0, st, ix, bt, wk, stages, n_small, n_small + n2, n_items, per, c->lane_scratch.as<uint32_t>(), stride);
                else hipLaunchKernelGGL(k_lift_lanes_g, dim3(nblk), dim3(LANE_G_WAVES * 64), 0, st, ix, bt, wk, stages, n_small, n_small + n2, n_items, per, c->lane_scratch.as<uint32_t>(), stride);
            }
            HIP_TRY(c, hipGetLastError());
            HIP_TRY(c, hipEventRecord(c->ev[2], st));
        } else if (n_items > n_small) {
            uint32_t lds_per_wave = (uint32_t)((tile_mem_bytes(c->cap) + 15) & ~(size_t)15);
            if (const char *e = getenv("PLO_LDS_PAD")) lds_per_wave += (uint32_t)atoi(e) & ~15u;  // occupancy experiments
            const uint32_t tw = (uint32_t)c->tile_waves;
            uint32_t nblk = (n_tiles + tw - 1) / tw;
            // persistent grid: what the chip keeps resident (CUs x blocks per CU), a multiple of 8 for the XCD mapping
            int occ = 1;
            const bool c256 = !sp && c->cap == TILE_CAP_SMALL && tw == (uint32_t)TILE_WAVES && lds_per_wave == (uint32_t)((tile_mem_bytes(TILE_CAP_SMALL) + 15) & ~(size_t)15);
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, sp ? (const void *)k_lift_tiles_sp : (c256 ? (const void *)k_lift_tiles_c256 : (const void *)k_lift_tiles), (int)tw * 64,
                                                             (size_t)lds_per_wave * tw) != hipSuccess || occ < 1)
                occ = 1;
            nblk = std::min<uint32_t>(nblk, (uint32_t)(c->n_cus * occ));
            nblk = (nblk + 7u) & ~7u;
            wk.slab_pre = n_small == 0 ? 1u : 0u;  // nothing has been reserved yet (the lane kernel did not run)
            if (wk.slab_pre) wk.slab_offset = (unsigned long long)nblk * tw * SLAB_OPS;  // (else: the lane kernel's waves own the first slabs)
            PLO_STAT_RANGE(nblk * tw);
            if (sp)
                hipLaunchKernelGGL(k_lift_tiles_sp, dim3(nblk), dim3(tw * 64), lds_per_wave * tw, st, ix, bt, wk, stages, n_tiles, c->window,
                                   c->big_thresh, c->cap, lds_per_wave);
            else if (c256)
                hipLaunchKernelGGL(k_lift_tiles_c256, dim3(nblk), dim3(tw * 64), lds_per_wave * tw, st, ix, bt, wk, stages, n_tiles, c->window,
                                   c->big_thresh, c->cap, lds_per_wave);
            else
                hipLaunchKernelGGL(k_lift_tiles, dim3(nblk), dim3(tw * 64), lds_per_wave * tw, st, ix, bt, wk, stages, n_tiles, c->window,
                                   c->big_thresh, c->cap, lds_per_wave);
            HIP_TRY(c, hipGetLastError());
            HIP_TRY(c, hipEventRecord(c->ev[2], st));
        } else {
            HIP_TRY(c, hipEventRecord(c->ev[2], st));
        }
        if (n_items) {
            // Items of tiles whose intermediates overflowed the slice (and items the lane kernel handed on), one per wave with a slice of twice the threshold (the shift /
            // simplify stages at most double an item's ops).  Launched without asking the host how many there are: the kernel reads
            // the count the tile kernel left (mostly zero -- a few microseconds -- and a host round trip less when it is not).
            // Batches without tiles (light items only, or heavy items through the lane code) have computed no tile geometry: the routing
            // threshold then comes from the batch's heaviest item, not from whatever an earlier batch left in the context.
            int r_thresh = c->big_thresh, r_cap = c->cap;
            if (c->adaptive && !(n_items > n_small && !heavy_lanes)) {
                r_thresh = heavy_lanes ? (int)std::min<uint32_t>(std::max<uint32_t>(256u, max_nin), 2016u) : 256;
                r_cap = 320;
            }
            const int retry_cap = std::min(4096, std::max(r_cap, (2 * r_thresh + 64 + 63) & ~63));
            uint32_t lds = (uint32_t)((tile_mem_bytes(retry_cap) + 15) & ~(size_t)15);
            uint32_t nw = (uint32_t)c->n_cus * 2u;
            PLO_STAT_RANGE(nw);
            if (sp) hipLaunchKernelGGL(k_lift_retry_sp, dim3(nw), dim3(64), lds, st, ix, bt, wk, stages, 0xffffffffu, r_thresh, retry_cap);
            else hipLaunchKernelGGL(k_lift_retry, dim3(nw), dim3(64), lds, st, ix, bt, wk, stages, 0xffffffffu, r_thresh, retry_cap);
            HIP_TRY(c, hipGetLastError());
        }
        HIP_TRY(c, hipEventRecord(c->ev[5], st));
        PLO_READ_COUNTERS();
        n_retry = (uint32_t)hc[CNT_NRETRY];
        n_big = (uint32_t)hc[CNT_NBIG];
        n_mid = 0;
        n_huge = n_big;
        const uint32_t *huge_src = c->big_list.as<uint32_t>();
        if (n_big && c->mid_waves) {
            // One workgroup per item.  Capacity from the heaviest item of the batch: the stages at most add a deletion per block
            // (already in the weight) and re-shape indel clusters, so a quarter on top of the weight holds every intermediate of
            // all but pathological items; those, and items beyond the largest slice, go on to the global-scratch kernel.
            int cap = std::min<unsigned long long>((unsigned long long)c->mid_cap_max, (((unsigned long long)max_nin * 5 / 4 + 64 + 63) & ~63ull));
            cap = std::max(cap, 512);
            const int mid_thresh = (cap - 64) * 4 / 5;
            const int nw = c->mid_waves;
            const size_t lds = nw == 16 ? mid_lds_bytes<16>(cap) : mid_lds_bytes<8>(cap);
            const void *fn = sp ? (nw == 16 ? (const void *)k_lift_mid_sp<16> : (const void *)k_lift_mid_sp<8>)
                                : (nw == 16 ? (const void *)k_lift_mid<16> : (const void *)k_lift_mid<8>);
            int occ = 1;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, fn, nw * 64, lds) != hipSuccess || occ < 1) occ = 1;
            uint32_t nblk = std::min<uint32_t>(n_big, (uint32_t)(c->n_cus * occ));
            HIP_TRY(c, c->huge_list.ensure((size_t)n_big * 4));
            wk.huge_list = c->huge_list.as<uint32_t>();
            if (getenv("PLO_DEBUG_GEOMETRY"))
                fprintf(stderr, "[plo] workgroup-per-item kernel: %u items, %d waves, cap %d (threshold %d), %zu B LDS, %d workgroups per CU\n", n_big, nw,
                        cap, mid_thresh, lds, occ);
            PLO_STAT_RANGE(nblk * (uint32_t)nw);
            if (sp && nw == 16)
                hipLaunchKernelGGL(k_lift_mid_sp<16>, dim3(nblk), dim3(16 * 64), lds, st, ix, bt, wk, stages, n_big, mid_thresh, cap);
            else if (sp)
                hipLaunchKernelGGL(k_lift_mid_sp<8>, dim3(nblk), dim3(8 * 64), lds, st, ix, bt, wk, stages, n_big, mid_thresh, cap);
            else if (nw == 16)
                hipLaunchKernelGGL(k_lift_mid<16>, dim3(nblk), dim3(16 * 64), lds, st, ix, bt, wk, stages, n_big, mid_thresh, cap);
            else
                hipLaunchKernelGGL(k_lift_mid<8>, dim3(nblk), dim3(8 * 64), lds, st, ix, bt, wk, stages, n_big, mid_thresh, cap);
            HIP_TRY(c, hipGetLastError());
            HIP_TRY(c, hipEventRecord(c->ev[6], st));
            c->ev_mid = true;
            PLO_READ_COUNTERS();
            n_huge = (uint32_t)hc[CNT_NHUGE];
            n_mid = n_big - n_huge;
            huge_src = c->huge_list.as<uint32_t>();
        }
        if (n_huge) {
            // size the wave-private scratch from the largest possible intermediate of a single item
            // (pieces <= ops + blocks, raw ops <= 2 x pieces); an item that still overflows is reported (CNT_ERROR)
            int big_cap = 4096;
            while (big_cap < (1 << 22) && (unsigned long long)big_cap < 6ull * max_nin + 2048ull) big_cap <<= 1;
            if (const char *e = getenv("PLO_BIG_CAP")) big_cap = std::max(1024, atoi(e));
            unsigned long long bpw = (tile_mem_bytes(big_cap) + 255) & ~(unsigned long long)255;
            // as many waves as the chip keeps resident (register-limited: 3 per SIMD), bounded by a 4 GiB scratch
            uint32_t nw = std::min<uint32_t>(n_huge, (uint32_t)c->n_cus * 12u);
            nw = (uint32_t)std::max<unsigned long long>(1ull, std::min<unsigned long long>(nw, (4ull << 30) / bpw));
            HIP_TRY(c, c->scratch.ensure((size_t)bpw * nw));
            PLO_STAT_RANGE(nw);
            if (sp)
                hipLaunchKernelGGL(k_lift_big_sp, dim3(nw), dim3(64), 0, st, ix, bt, wk, stages, n_huge, huge_src, c->scratch.as<unsigned char>(),
                                   big_cap, bpw);
            else
                hipLaunchKernelGGL(k_lift_big, dim3(nw), dim3(64), 0, st, ix, bt, wk, stages, n_huge, huge_src, c->scratch.as<unsigned char>(),
                                   big_cap, bpw);
            HIP_TRY(c, hipGetLastError());
            HIP_TRY(c, hipEventRecord(c->ev[3], st));
            c->ev_big = true;
            PLO_READ_COUNTERS();
        }
        n_miss = (uint32_t)hc[CNT_NMISS];
        if (n_miss && in->seq_full && in->read_seq_full_off) {
            // second look at the items that reached bases a sparse batch does not carry: their reads' complete bases go up now
            const auto t0 = std::chrono::steady_clock::now();
            HIP_TRY(c, c->miss_info.ensure((size_t)n_miss * 8));
            HIP_TRY(c, c->h_miss.ensure((size_t)n_miss * 16));
            uint32_t *d_read = c->miss_info.as<uint32_t>(), *d_len = d_read + n_miss;
            hipLaunchKernelGGL(k_miss_info, dim3((n_miss + 255) / 256), dim3(256), 0, st, (const uint32_t *)c->miss_list.as<uint32_t>(), n_miss,
                               bt, wk, d_read, d_len);
            uint32_t *h_read = c->h_miss.as<uint32_t>(), *h_len = h_read + n_miss;
            HIP_TRY(c, hipMemcpyAsync(h_read, d_read, (size_t)n_miss * 8, hipMemcpyDeviceToHost, st));
            HIP_TRY(c, hipStreamSynchronize(st)); ++n_syncs;
            std::vector<uint32_t> order(n_miss);
            for (uint32_t k = 0; k < n_miss; ++k) order[k] = k;
            std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return h_read[a] < h_read[b]; });
            uint64_t *h_vals = (uint64_t *)(h_read + 2 * (size_t)n_miss);
            uint64_t total = 0;
            for (uint32_t k = 0; k < n_miss; ++k) {  // one copy per read, 16-byte aligned, 16 spare bytes for the wide window loads
                const uint32_t a = order[k];
                if (k && h_read[order[k - 1]] == h_read[a]) {
                    h_vals[a] = h_vals[order[k - 1]];
                    continue;
                }
                h_vals[a] = total;
                total += ((((uint64_t)h_len[a] + 1) / 2 + 15) & ~15ull) + 16;
            }
            HIP_TRY(c, c->h_side.ensure((size_t)total));
            HIP_TRY(c, c->miss_side.ensure((size_t)total));
            HIP_TRY(c, c->miss_vals.ensure((size_t)n_miss * 8));
            HIP_TRY(c, c->miss_seq_off.ensure((size_t)std::max(1u, n_items) * 8));
            for (uint32_t k = 0; k < n_miss; ++k) {
                const uint32_t a = order[k];
                if (k && h_read[order[k - 1]] == h_read[a]) continue;
                const size_t nb = ((size_t)h_len[a] + 1) / 2;
                uint8_t *dst = c->h_side.as<uint8_t>() + h_vals[a];
                memcpy(dst, in->seq_full + in->read_seq_full_off[h_read[a]], nb);
                memset(dst + nb, 0, (((nb + 15) & ~(size_t)15) + 16) - nb);
            }
            HIP_TRY(c, hipMemcpyAsync(c->miss_side.p, c->h_side.p, (size_t)total, hipMemcpyHostToDevice, st));
            HIP_TRY(c, hipMemcpyAsync(c->miss_vals.p, h_vals, (size_t)n_miss * 8, hipMemcpyHostToDevice, st));
            hipLaunchKernelGGL(k_miss_patch, dim3((n_miss + 255) / 256), dim3(256), 0, st, (const uint32_t *)c->miss_list.as<uint32_t>(), n_miss,
                               (const uint64_t *)c->miss_vals.as<uint64_t>(), c->miss_seq_off.as<uint64_t>());
            DevBatch bt2 = bt;
            bt2.seq = c->miss_side.as<uint8_t>();
            bt2.seq_bytes = total;
            bt2.seq_fmt = PLO_SEQ_BAM4;
            DevWork wk2 = wk;
            wk2.d.seq_off = c->miss_seq_off.as<uint64_t>();
            int big_cap = 4096;
            while (big_cap < (1 << 22) && (unsigned long long)big_cap < 6ull * max_nin + 2048ull) big_cap <<= 1;
            if (const char *e = getenv("PLO_BIG_CAP")) big_cap = std::max(1024, atoi(e));
            unsigned long long bpw = (tile_mem_bytes(big_cap) + 255) & ~(unsigned long long)255;
            uint32_t nw = std::min<uint32_t>(n_miss, (uint32_t)c->n_cus * 12u);
            nw = (uint32_t)std::max<unsigned long long>(1ull, std::min<unsigned long long>(nw, (4ull << 30) / bpw));
            HIP_TRY(c, c->scratch.ensure((size_t)bpw * nw));
            wk2.stat_base = stat_used;
            stat_used += nw;
            hipLaunchKernelGGL(k_lift_big, dim3(nw), dim3(64), 0, st, ix, bt2, wk2, stages, n_miss, (const uint32_t *)c->miss_list.as<uint32_t>(),
                               c->scratch.as<unsigned char>(), big_cap, bpw);
            HIP_TRY(c, hipGetLastError());
            PLO_READ_COUNTERS();
            miss_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
        }
        if (hc[CNT_OVERFLOW] == 0) break;
        if (attempt >= 3) {
            c->err = "output CIGAR buffer kept overflowing";
            return PLO_ERR_INTERNAL;
        }
        HIP_TRY(c, c->o_cigar.ensure((size_t)(hc[CNT_CIGAR] + wk.slab_offset + 4096 + (size_t)std::min<uint32_t>(n_tiles + 1024, (uint32_t)c->n_cus * 16) * SLAB_OPS) * 4));
    }
    if (hc[CNT_ERROR]) {
        c->err = "an item exceeded the large-item scratch capacity (raise PLO_BIG_CAP), or the lane kernel's group list its capacity";
        return PLO_ERR_INTERNAL;
    }
    c->timing.n_items = n_items;
    c->timing.n_big_items = n_huge;
    c->timing.n_mid_items = n_mid;
    c->timing.n_miss_items = n_miss;
    c->timing.tile_cap = (uint32_t)c->cap;
    c->timing.tile_window = (uint32_t)c->window;
    c->timing.miss_ms = miss_ms;
    c->timing.n_lane_items = n_small;
    c->timing.n_heavy_lane_items = heavy_lanes ? n_items - n_small : 0u;
    c->timing.heavy_kernel = heavy_kernel;
    c->timing.host_syncs = n_syncs + c->fast_extra_syncs;
    c->fast_extra_syncs = 0;
    c->fast_ns_cap = ns;
    // (PLO_FAST_CAP_EXACT=1, tests: the last batch's item count, so that a window with a few items more exercises the VERR_CAP fallback)
    c->fast_item_cap = getenv("PLO_FAST_CAP_EXACT") ? n_items : std::max(n_items, c->item_cap_now);
    c->fast_ref_items = n_items;
    c->fast_ref_ns = ns;
    c->fast_n0 = h_cls[0];
    c->fast_n1 = h_cls[1];
    // (items the lane kernel handed to the retry kernel are fine: the fast path runs that kernel too; items IT hands on are not)
    c->fast_light_only = n_items > 0 && n_items == n_small && n_big == 0 && n_miss == 0;
    c->timing.n_retry_items = n_retry;
    if (c->adaptive && c->cap == TILE_CAP_SMALL && n_retry > n_items / 200) c->small_window_tight = true;
    c->timing.n_in_ops = hc[CNT_IN_OPS];
    c->timing.n_out_ops = hc[CNT_OUT_OPS];
    c->timing.algo_bytes = hc[CNT_ALGO_BYTES];
    c->timing.lane_utilisation = hc[CNT_LANE_TRIPS] ? (float)((double)hc[CNT_LANE_ACT] / (64.0 * (double)hc[CNT_LANE_TRIPS])) : 0.0f;
    for (int k = 0; k < 12; ++k) c->phase_cycles[k] = hc[CNT_PHASE0 + k];

    out->n_items = n_items;
    out->item_seg = c->item_seg.as<uint32_t>();
    out->item_cseg = c->item_cseg.as<uint32_t>();
    out->item_status = c->o_status.as<uint8_t>();
    out->item_need_flipped = c->o_flip.as<uint8_t>();
    out->item_mapq = c->o_mapq.as<uint8_t>();
    out->item_chrom_index = c->o_chrom.as<uint32_t>();
    out->item_ref_pos = c->o_pos.as<int64_t>();
    out->item_cigar_off = c->o_coff.as<uint64_t>();
    out->item_cigar_len = c->o_clen.as<uint32_t>();
    out->cigar = c->o_cigar.as<uint32_t>();
    out->n_cigar = hc[CNT_CIGAR] + wk.slab_offset;
    c->last_wk = wk;
    c->last_bt = bt;
    c->have_last = true;
    c->have_finish = false;
    return PLO_OK;
}

// One host round trip per batch.  The enumerate kernels, the class order, k_lift_lanes and k_lift_retry are launched
// back to back with the item and class counts read from DEVICE memory (k_cls_hist / k_permute2 / k_chunk_sort_w / k_lift_lanes take them by
// pointer; grids are sized by the arrays' capacity, surplus workgroups leave at once); the one copy at the end brings the item count, the
// validation flags, the class counts and the counters.  `fallback`: the batch does not fit what the context's last batch left (more items
// than the arrays hold, heavy items, items handed on by the retry kernel, an output overflow): nothing of it is used.
static plo_status liftover_fast(plo_ctx *c, const plo_batch_in *in, uint32_t stages, plo_batch_out *out, const DevBatch &bt, bool &fallback) {
    fallback = false;
    const DevIndex &ix = c->ix->d;
    hipStream_t st = c->stream;
    const uint32_t ns = in->n_segs;
    // The item capacity the kernels are launched with: what the arrays hold -- but no more than this batch can plausibly need (a quarter more
    // items per segment than the last careful batch had, + 1 024).  A context that has once lifted a 2 M-read batch would otherwise run every
    // 50 k-read window with grids for 2 M items: surplus workgroups leave at once, yet a thousand of them per class kernel and a lane grid of
    // every wave slot cost the call ~20 us (tools/ab_window.py).  A batch with more items raises VERR_CAP like one beyond the arrays.
    uint32_t cap = c->fast_item_cap;
    if (c->fast_launch_bound && c->fast_ref_ns) {
        const unsigned long long per = ((unsigned long long)ns * c->fast_ref_items + c->fast_ref_ns - 1) / c->fast_ref_ns;
        cap = (uint32_t)std::min<unsigned long long>(cap, per + per / 4 + 1024ull);
    }
    // Everything the kernels of this path count in, one block: cleared by ONE fill and read back by ONE copy (a launch of any size costs
    // ~5 us on the stream: three fills and four copies were a quarter of a 50 k-read call's kernels, tools/trace_window.sh)
    //   [0, 256) the class totals ([6] = the item count) and the lane kernel's group counter, [256, 272) the validation flags,
    //   [512, 512 + 8 CNT_N) the batch counters
    constexpr size_t FB_VERR = 256, FB_COUNTERS = 512, FB_BYTES = 1024, FB_BACK = FB_COUNTERS + CNT_N * 8;
    static_assert(FB_BACK <= FB_BYTES, "fast block");
    //   [288, 292) the scan's ticket, [FB_BYTES, ...) the scan's tile words
    constexpr size_t FB_TICKET = 288;
    const bool chain = c->scan_chain;
    const uint32_t scan_nb = std::max(1u, (ns + SCAN_BLOCK - 1) / SCAN_BLOCK);
    const size_t fb_bytes = FB_BYTES + (chain ? (size_t)scan_nb * 8 : 0);
    HIP_TRY(c, c->fast_blk.ensure(fb_bytes));
    HIP_TRY(c, c->h_counters.ensure(FB_BYTES));
    uint8_t *const fb = c->fast_blk.as<uint8_t>();
    uint32_t *const misc_d = (uint32_t *)fb, *const verr_d = (uint32_t *)(fb + FB_VERR);
    unsigned long long *const counters_d = (unsigned long long *)(fb + FB_COUNTERS);
    const bool ev_on = c->phase_events;  // (plo_ctx_set_phase_events(ctx, 0): no event records -- two bubbles of ~6 us fewer on the stream, no phase times)
    if (ev_on) HIP_TRY(c, hipEventRecord(c->ev[0], st));
    HIP_TRY(c, hipMemsetAsync(fb, 0, fb_bytes, st));
    hipLaunchKernelGGL(k_seg_count, dim3((unsigned)(((unsigned long long)ns * SEG_LANES + 255) / 256)), dim3(256), 0, st, ix, bt, c->seg_cnt.as<uint32_t>(),
                       c->seg_reflen.as<int>(), c->seg_readlen.as<uint32_t>(), c->seg_nm.as<uint32_t>(), verr_d);
    if (chain) {
        hipLaunchKernelGGL(k_scan_chain, dim3(scan_nb), dim3(SCAN_THREADS), 0, st, (const uint32_t *)c->seg_cnt.as<uint32_t>(), ns, c->seg_off.as<uint32_t>(),
                           (unsigned long long *)(fb + FB_BYTES), (uint32_t *)(fb + FB_TICKET));
    } else {
        plo_status s = scan_u32(c, c->seg_cnt.as<uint32_t>(), ns, c->seg_off.as<uint32_t>());
        if (s != PLO_OK) return s;
    }
    const uint32_t *n_dev = c->seg_off.as<uint32_t>() + ns, *totals_dev = misc_d;
    DevWork wk;
    fill_work(c, wk, cap);
    wk.counters = counters_d;
    hipLaunchKernelGGL(k_item_emit, dim3((ns + 255) / 256), dim3(256), 0, st, ix, bt, wk, stages, (const uint32_t *)c->seg_off.as<uint32_t>(), c->seg_reflen.as<int>(),
                       cap, verr_d);
    const uint32_t cls_nb = (cap + CLS_BLOCK - 1) / CLS_BLOCK;
    HIP_TRY(c, c->cls_partial.ensure((size_t)std::max(1u, cls_nb) * 6 * 4));
    hipLaunchKernelGGL(k_cls_hist, dim3(cls_nb), dim3(CLS_THREADS), 0, st, (const uint32_t *)c->item_cls.as<uint32_t>(), (const uint32_t *)c->item_nin.as<uint32_t>(), cap,
                       cls_nb, c->cls_partial.as<uint32_t>(), n_dev);
    hipLaunchKernelGGL(k_permute2_s, dim3(cls_nb), dim3(CLS_THREADS), 0, st, (const uint32_t *)c->item_cls.as<uint32_t>(), (const uint32_t *)c->item_nin.as<uint32_t>(),
                       (const uint32_t *)c->cls_partial.as<uint32_t>(), misc_d, cap, cls_nb, c->perm.as<uint32_t>(), c->nin_p.as<uint32_t>(),
                       c->adaptive ? (uint32_t)((WHIST_BINS - 1) * WHIST_STEP) : 0xffffffffu, n_dev);
    if (ev_on) HIP_TRY(c, hipEventRecord(c->ev[1], st));
    // launch geometry from the last batch's class counts (a window of the same shape has the same): group size, persistent grid
    uint32_t lane_gs = 64, lane_nblk = 0;
    const size_t lane_lds = (size_t)(c->lane_capw + 2 * c->lane_kvs) * 4 * LANE_WAVES;
    {
        int occ = 1;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void *)k_lift_lanes, LANE_WAVES * 64, lane_lds) != hipSuccess || occ < 1) occ = 1;
        const uint32_t n0 = c->fast_n0, n1 = c->fast_n1, slots = (uint32_t)(c->n_cus * occ) * LANE_WAVES;
        while (lane_gs > 8u && (n0 + lane_gs / 2 - 1) / (lane_gs / 2) + (n1 + lane_gs / 2 - 1) / (lane_gs / 2) <= slots) lane_gs >>= 1;
        if (const char *e = getenv("PLO_LANE_GROUP")) lane_gs = std::min(64u, std::max(1u, (uint32_t)atoi(e)));
        // (groups of this batch: at most cap / gs + 2 -- one partial group per class)
        const uint32_t groups = cap / lane_gs + 2;
        lane_nblk = std::min<uint32_t>((groups + LANE_WAVES - 1) / LANE_WAVES, (uint32_t)(c->n_cus * occ));
        lane_nblk = (lane_nblk + 7u) & ~7u;
    }
    const uint32_t ch = (uint32_t)c->lane_sort_window;
    hipLaunchKernelGGL(k_chunk_sort_w, dim3((cap / ch + 2u + 3u) / 4u), dim3(256), 0, st, c->perm.as<uint32_t>(), (const uint32_t *)c->d_n_m.as<uint32_t>(),
                       (const uint32_t *)c->d_w0.as<uint32_t>(), (const uint32_t *)c->d_w1.as<uint32_t>(), 0u, 0u, ch, totals_dev);
    wk.out_cigar = c->o_cigar.as<uint32_t>();
    wk.out_cap = c->o_cigar.cap / 4;
    wk.slab_pre = 1u;
    wk.slab_offset = (unsigned long long)lane_nblk * LANE_WAVES * SLAB_OPS;
    wk.n_small = cap;  // (unused by the lane kernel; the retry kernel's tiles take their items from the retry list)
    wk.lane_n_groups = misc_d + 32;
    wk.stat_base = 0;
    uint32_t stat_used = lane_nblk * LANE_WAVES;
    if (wk.slab_offset + SLAB_OPS > wk.out_cap) {  // the output buffer of the last batch does not even hold the waves' first slabs
        fallback = true;
        return PLO_OK;
    }
    if (c->seq_pending) {  // (plo_liftover_batch: the read bases arrive on the copy stream)
        HIP_TRY(c, hipStreamWaitEvent(st, c->ev_seq, 0));
        c->seq_pending = false;
    }
    lane_ticket_arm(c, wk, (c->fast_n0 + lane_gs - 1) / lane_gs + (c->fast_n1 + lane_gs - 1) / lane_gs, lane_nblk * LANE_WAVES);  // (the last batch's class counts)
    hipLaunchKernelGGL(lane_kernel_of(c, stages, false), dim3(lane_nblk), dim3(LANE_WAVES * 64), lane_lds, st, ix, bt, wk, stages, 0u, 0u, lane_gs, c->lane_capw, totals_dev);
    HIP_TRY(c, hipGetLastError());
    if (ev_on) HIP_TRY(c, hipEventRecord(c->ev[4], st));  // (the last event of this path: every record is a ~5 us bubble on the stream; retry and counters are not timed)
    {
        const int retry_cap = 320;
        const uint32_t lds = (uint32_t)((tile_mem_bytes(retry_cap) + 15) & ~(size_t)15), nw = (uint32_t)c->n_cus * 2u;
#ifdef PLO_PHASE_TIMING
        const bool fuse = false;  // (timing builds: the waves' clocks travel in their slots, wave_ctx_flush)
#else
        const bool fuse = c->fast_fuse;
#endif
        if (fuse) {  // the counters' sum rides in the retry launch: its waves add their own counts, the first of them the light-item kernel's slots
            hipLaunchKernelGGL(k_lift_retry_sum, dim3(std::max(nw, SUM_BLOCKS)), dim3(64), lds, st, ix, bt, wk, stages, 256, retry_cap, stat_used);
            HIP_TRY(c, hipGetLastError());
        } else {
            wk.stat_base = stat_used;
            stat_used += nw;
            hipLaunchKernelGGL(k_lift_retry, dim3(nw), dim3(64), lds, st, ix, bt, wk, stages, 0xffffffffu, 256, retry_cap);
            HIP_TRY(c, hipGetLastError());
            hipLaunchKernelGGL(k_sum_stats, dim3(std::min<uint32_t>((stat_used + 255) / 256, 16u)), dim3(256), 0, st, c->wave_stats.as<unsigned long long>(), stat_used,
                               counters_d);
        }
    }
    // the one look: class totals and item count, validation flags, counters
    uint8_t *const hb = c->h_counters.as<uint8_t>();
    HIP_TRY(c, hipMemcpyAsync(hb, fb, FB_BACK, hipMemcpyDeviceToHost, st));
    HIP_TRY(c, hipStreamSynchronize(st));
    const unsigned long long *hc = (const unsigned long long *)(hb + FB_COUNTERS);
    const uint32_t *const hm = (const uint32_t *)hb;
    const uint32_t hx[8] = {hm[6], *(const uint32_t *)(hb + FB_VERR), hm[0], hm[1], hm[2], hm[3], hm[4], hm[5]};
    const uint32_t n_items = hx[0], verr = hx[1], n0 = hx[2], n1 = hx[3];
    if (verr & (VERR_INDEX | VERR_RANGE)) {
        if (verr & VERR_INDEX) {
            c->err = "plo_batch_in: an index points outside its array (seg_read / seg_contig / seg_cigar_off / read_seq_off / item_seg / item_cseg)";
            return PLO_ERR_INVALID_ARG;
        }
        c->err = "plo_batch_in: coordinate outside the 31-bit BAM range, CIGAR op code above 8, or a CIGAR spanning more than 2^30 bases";
        return PLO_ERR_RANGE;
    }
    const unsigned long long all_ops = (unsigned long long)hx[6] | ((unsigned long long)hx[7] << 32);
    if ((verr & VERR_CAP) || n_items > cap || n0 + n1 != n_items || all_ops > 0x7fffffffull || hc[CNT_OVERFLOW] || hc[CNT_NBIG] || hc[CNT_NMISS] || hc[CNT_ERROR] ||
        (unsigned long long)(n0 / lane_gs + n1 / lane_gs + 2) > (unsigned long long)lane_nblk * LANE_WAVES * 0xffffull) {
        fallback = true;  // more items than the arrays hold / heavy items / handed-on items / output overflow: the careful path runs the batch
        return PLO_OK;
    }
    memset(&c->timing, 0, sizeof(c->timing));
    c->ev_big = false;
    c->ev_mid = false;
    c->timing.n_items = n_items;
    c->timing.tile_cap = (uint32_t)c->cap;
    c->timing.tile_window = (uint32_t)c->window;
    c->timing.n_lane_items = n_items;
    c->timing.n_retry_items = (uint32_t)hc[CNT_NRETRY];
    c->timing.n_in_ops = hc[CNT_IN_OPS];
    c->timing.n_out_ops = hc[CNT_OUT_OPS];
    c->timing.algo_bytes = hc[CNT_ALGO_BYTES];
    c->timing.lane_utilisation = hc[CNT_LANE_TRIPS] ? (float)((double)hc[CNT_LANE_ACT] / (64.0 * (double)hc[CNT_LANE_TRIPS])) : 0.0f;
    c->timing.host_syncs = 1;
    c->fast_timing = true;
    c->fast_no_events = !ev_on;
    for (int k = 0; k < 12; ++k) c->phase_cycles[k] = hc[CNT_PHASE0 + k];
    c->fast_n0 = n0;
    c->fast_n1 = n1;
    out->n_items = n_items;
    out->item_seg = c->item_seg.as<uint32_t>();
    out->item_cseg = c->item_cseg.as<uint32_t>();
    out->item_status = c->o_status.as<uint8_t>();
    out->item_need_flipped = c->o_flip.as<uint8_t>();
    out->item_mapq = c->o_mapq.as<uint8_t>();
    out->item_chrom_index = c->o_chrom.as<uint32_t>();
    out->item_ref_pos = c->o_pos.as<int64_t>();
    out->item_cigar_off = c->o_coff.as<uint64_t>();
    out->item_cigar_len = c->o_clen.as<uint32_t>();
    out->cigar = c->o_cigar.as<uint32_t>();
    out->n_cigar = hc[CNT_CIGAR] + wk.slab_offset;
    wk.n_items = n_items;
    wk.n_small = n_items;
    c->last_wk = wk;
    c->last_bt = bt;
    c->have_last = true;
    c->have_finish = false;
    return PLO_OK;
}

plo_status plo_finish_batch_dev(plo_ctx *c, const plo_batch_in *in, const plo_finish_in *fin, plo_finish_out *out) {
    if (!c || !in || !fin || !out) return PLO_ERR_INVALID_ARG;
    memset(out, 0, sizeof(*out));
    c->err.clear();
    if (!c->have_last || c->last_bt.n_segs != in->n_segs || c->last_bt.n_reads != in->n_reads) {
        c->err = "plo_finish_batch_dev: call plo_liftover_batch_dev on the same batch first";
        return PLO_ERR_INVALID_ARG;
    }
    if (c->last_bt.seq_fmt == PLO_SEQ_BAM4_SPARSE) {
        c->err = "plo_finish_batch_dev: the batch came with sparse bases (PLO_SEQ_BAM4_SPARSE); the flipped sequences are written from complete ones";
        return PLO_ERR_INVALID_ARG;
    }
    HIP_TRY(c, hipSetDevice(c->ix->device));
    hipStream_t st = c->stream;
    for (int i = 0; i < 3; ++i)
        if (!c->fev[i]) HIP_TRY(c, hipEventCreate(&c->fev[i]));
    const DevWork &wk = c->last_wk;
    DevBatch bt = c->last_bt;
    uint32_t n = wk.n_items, nr = bt.n_reads, ne = n + nr;
    size_t a = std::max(1u, n), b = std::max(1u, nr), e = std::max(1u, ne);
    HIP_TRY(c, c->f_flag.ensure(a * 2));
    HIP_TRY(c, c->f_bin.ensure(a * 2));
    HIP_TRY(c, c->f_end.ensure(a * 8));
    HIP_TRY(c, c->f_prim.ensure(a));
    HIP_TRY(c, c->f_isoff.ensure(a * 8));
    HIP_TRY(c, c->f_iqoff.ensure(a * 8));
    HIP_TRY(c, c->f_iread.ensure(a * 4));
    HIP_TRY(c, c->f_nl.ensure(b * 4));
    HIP_TRY(c, c->f_pitem.ensure(b * 4));
    HIP_TRY(c, c->f_uflag.ensure(b * 2));
    HIP_TRY(c, c->f_rsoff.ensure(b * 8));
    HIP_TRY(c, c->f_rqoff.ensure(b * 8));
    HIP_TRY(c, c->f_su.ensure(e * 4));
    HIP_TRY(c, c->f_qu.ensure(e * 4));
    HIP_TRY(c, c->f_soff.ensure((e + 1) * 4));
    HIP_TRY(c, c->f_qoff.ensure((e + 1) * 4));
    HIP_TRY(c, c->f_fflag.ensure(e * 4));
    HIP_TRY(c, c->f_frank.ensure((e + 1) * 4));
    HIP_TRY(c, c->f_flist.ensure(e * 4));
    DevFinish f;
    memset(&f, 0, sizeof(f));
    f.read_flags = fin->read_flags;
    f.qual = fin->qual;
    f.read_qual_off = fin->read_qual_off;
    f.qual_bytes = fin->qual_bytes;
    f.seq_bytes = in->seq_bytes;
    f.item_flag = c->f_flag.as<uint16_t>();
    f.item_bin = c->f_bin.as<uint16_t>();
    f.item_ref_end = c->f_end.as<int64_t>();
    f.item_is_primary = c->f_prim.as<uint8_t>();
    f.item_seq_off = c->f_isoff.as<uint64_t>();
    f.item_qual_off = c->f_iqoff.as<uint64_t>();
    f.item_read = c->f_iread.as<uint32_t>();
    f.read_n_lifted = c->f_nl.as<uint32_t>();
    f.read_primary_item = c->f_pitem.as<uint32_t>();
    f.read_unmapped_flag = c->f_uflag.as<uint16_t>();
    f.read_seq_off = c->f_rsoff.as<uint64_t>();
    f.read_qual_off_out = c->f_rqoff.as<uint64_t>();
    f.su = c->f_su.as<uint32_t>();
    f.qu = c->f_qu.as<uint32_t>();
    f.soff = c->f_soff.as<uint32_t>();
    f.qoff = c->f_qoff.as<uint32_t>();
    f.fflag = c->f_fflag.as<uint32_t>();
    f.frank = c->f_frank.as<uint32_t>();
    f.flist = c->f_flist.as<uint32_t>();
    HIP_TRY(c, c->verr.ensure(16));
    HIP_TRY(c, hipMemsetAsync(c->verr.p, 0, 16, st));
    f.n_fault = c->verr.as<unsigned>() + 2;
    HIP_TRY(c, hipEventRecord(c->fev[0], st));
    if (n) hipLaunchKernelGGL(k_finish_items, dim3((n + 255) / 256), dim3(256), 0, st, bt, wk, f);
    if (nr) hipLaunchKernelGGL(k_finish_reads, dim3((nr + 255) / 256), dim3(256), 0, st, bt, wk, f);
    HIP_TRY(c, hipGetLastError());
    plo_status s = scan_u32(c, f.su, ne, c->f_soff.as<uint32_t>());
    if (s != PLO_OK) return s;
    s = scan_u32(c, f.qu, ne, c->f_qoff.as<uint32_t>());
    if (s != PLO_OK) return s;
    s = scan_u32(c, f.fflag, ne, c->f_frank.as<uint32_t>());
    if (s != PLO_OK) return s;
    if (ne) hipLaunchKernelGGL(k_finish_offsets, dim3((ne + 255) / 256), dim3(256), 0, st, bt, wk, f);
    uint32_t *h = c->h_counters.as<uint32_t>();
    HIP_TRY(c, hipMemcpyAsync(h, c->f_soff.as<uint32_t>() + ne, 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(c, hipMemcpyAsync(h + 1, c->f_qoff.as<uint32_t>() + ne, 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(c, hipMemcpyAsync(h + 2, c->verr.as<unsigned>() + 2, 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(c, hipEventRecord(c->fev[1], st));
    HIP_TRY(c, hipStreamSynchronize(st));
    if (h[2]) {
        c->err = "plo_finish_batch_dev: " + std::to_string(h[2]) + " item(s) ended LEN_MISMATCH / PANIC; the reference aborts here (src/read_alignment_scanner.rs:207-229)";
        return PLO_ERR_DATA;
    }
    uint64_t sb = (uint64_t)h[0] * 16u, qb = (uint64_t)h[1] * 16u;
    HIP_TRY(c, c->f_rseq.ensure(std::max<uint64_t>(sb, 16)));
    HIP_TRY(c, c->f_rqual.ensure(std::max<uint64_t>(qb, 16)));
    f.rev_seq = c->f_rseq.as<uint8_t>();
    f.rev_qual = c->f_rqual.as<uint8_t>();
    if (sb) {
        uint32_t nblk = std::min<uint32_t>(ne, (uint32_t)c->n_cus * 16u);
        hipLaunchKernelGGL(k_revcomp, dim3(nblk), dim3(256), 0, st, bt, wk, f);
        HIP_TRY(c, hipGetLastError());
    }
    HIP_TRY(c, hipEventRecord(c->fev[2], st));
    HIP_TRY(c, hipStreamSynchronize(st));
    (void)hipEventElapsedTime(&out->finish_ms, c->fev[0], c->fev[1]);
    (void)hipEventElapsedTime(&out->revcomp_ms, c->fev[1], c->fev[2]);
    out->item_flag = f.item_flag;
    out->item_bin = f.item_bin;
    out->item_ref_end = f.item_ref_end;
    out->item_is_primary = f.item_is_primary;
    out->item_seq_off = f.item_seq_off;
    out->item_qual_off = f.item_qual_off;
    out->read_n_lifted = f.read_n_lifted;
    out->read_primary_item = f.read_primary_item;
    out->read_unmapped_flag = f.read_unmapped_flag;
    out->read_seq_off = f.read_seq_off;
    out->read_qual_off = f.read_qual_off_out;
    out->rev_seq = f.rev_seq;
    out->rev_qual = f.rev_qual;
    out->rev_seq_bytes = sb;
    out->rev_qual_bytes = qb;
    out->n_items = n;
    out->n_reads = nr;
    c->have_finish = true;
    return PLO_OK;
}

plo_status plo_sa_segments_dev(plo_ctx *c, const plo_sa_in *in, plo_sa_out *out) {
    if (!c || !in || !out) return PLO_ERR_INVALID_ARG;
    memset(out, 0, sizeof(*out));
    c->err.clear();
    if (!c->have_last || !c->have_finish) {
        c->err = "plo_sa_segments_dev: call plo_liftover_batch_dev and plo_finish_batch_dev on the batch first";
        return PLO_ERR_INVALID_ARG;
    }
    if (in->n_chroms < c->ix->d.n_chroms || !in->chrom_name_off || (!in->chrom_names && in->n_chroms)) {
        c->err = "plo_sa_in: one label per chromosome of the index is required";
        return PLO_ERR_INVALID_ARG;
    }
    HIP_TRY(c, hipSetDevice(c->ix->device));
    hipStream_t st = c->stream;
    for (int i = 3; i < 5; ++i)
        if (!c->fev[i]) HIP_TRY(c, hipEventCreate(&c->fev[i]));
    const DevWork &wk = c->last_wk;
    uint32_t n = wk.n_items;
    HIP_TRY(c, c->sa_len.ensure(((size_t)n + 1) * 4));
    HIP_TRY(c, c->sa_off.ensure(((size_t)n + 1) * 4));
    DevSa sa;
    sa.chrom_name_off = in->chrom_name_off;
    sa.chrom_names = in->chrom_names;
    sa.item_flag = c->f_flag.as<uint16_t>();
    sa.item_read = c->f_iread.as<uint32_t>();
    sa.read_n_lifted = c->f_nl.as<uint32_t>();
    sa.len = c->sa_len.as<uint32_t>();
    sa.off = c->sa_off.as<uint32_t>();
    sa.text = nullptr;
    HIP_TRY(c, hipEventRecord(c->fev[3], st));
    if (n) hipLaunchKernelGGL(k_sa_len, dim3((n + 255) / 256), dim3(256), 0, st, wk, sa);
    HIP_TRY(c, hipGetLastError());
    plo_status s = scan_u32(c, sa.len, n, c->sa_off.as<uint32_t>());
    if (s != PLO_OK) return s;
    uint32_t *h = c->h_counters.as<uint32_t>();
    HIP_TRY(c, hipMemcpyAsync(h, c->sa_off.as<uint32_t>() + n, 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(c, hipStreamSynchronize(st));
    uint64_t bytes = n ? h[0] : 0;
    HIP_TRY(c, c->sa_text.ensure(std::max<uint64_t>(bytes, 16)));
    sa.text = c->sa_text.as<uint8_t>();
    if (bytes) hipLaunchKernelGGL(k_sa_emit, dim3((n + 255) / 256), dim3(256), 0, st, wk, sa);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipEventRecord(c->fev[4], st));
    HIP_TRY(c, hipStreamSynchronize(st));
    (void)hipEventElapsedTime(&out->sa_ms, c->fev[3], c->fev[4]);
    out->n_items = n;
    out->item_sa_off = c->sa_off.as<uint32_t>();
    out->sa_text = sa.text;
    out->sa_bytes = bytes;
    return PLO_OK;
}

plo_status plo_compact_output_dev(plo_ctx *c, plo_batch_out *out) {
    if (!c || !out) return PLO_ERR_INVALID_ARG;
    c->err.clear();
    if (!c->have_last || out->n_items != c->last_wk.n_items) {
        c->err = "plo_compact_output_dev: `out` is not the result of the context's last plo_liftover_batch_dev";
        return PLO_ERR_INVALID_ARG;
    }
    if (c->last_wk.out_cigar == c->o_cigar_dense.as<uint32_t>()) {  // already dense
        out->cigar = c->o_cigar_dense.as<uint32_t>();
        out->n_cigar = c->dense_total;
        return PLO_OK;
    }
    HIP_TRY(c, hipSetDevice(c->ix->device));
    hipStream_t st = c->stream;
    const uint32_t n = out->n_items;
    HIP_TRY(c, c->o_dense_off.ensure(((size_t)n + 1) * 4));
    plo_status s = scan_u32(c, c->o_clen.as<uint32_t>(), n, c->o_dense_off.as<uint32_t>());
    if (s != PLO_OK) return s;
    uint32_t *h = c->h_counters.as<uint32_t>();
    h[0] = 0;
    if (n) HIP_TRY(c, hipMemcpyAsync(h, c->o_dense_off.as<uint32_t>() + n, 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(c, hipStreamSynchronize(st));
    const uint64_t total = h[0];
    HIP_TRY(c, c->o_cigar_dense.ensure(std::max<uint64_t>(total, 4) * 4));
    if (n)
        hipLaunchKernelGGL(k_compact_cigar, dim3((unsigned)(((unsigned long long)n * 8 + 255) / 256)), dim3(256), 0, st,
                           (const uint32_t *)c->o_cigar.as<uint32_t>(), c->o_coff.as<uint64_t>(), (const uint32_t *)c->o_clen.as<uint32_t>(),
                           (const uint32_t *)c->o_dense_off.as<uint32_t>(), n, c->o_cigar_dense.as<uint32_t>());
    HIP_TRY(c, hipGetLastError());
    c->last_wk.out_cigar = c->o_cigar_dense.as<uint32_t>();
    out->cigar = c->o_cigar_dense.as<uint32_t>();
    out->n_cigar = total;
    c->dense_total = total;
    return PLO_OK;
}

// Inflates the n BGZF blocks of one chunk on the device (bam_host.cpp's reader when PLO_BGZF_DEVICE is not 0): `comp` = the
// chunk's compressed bytes (host, pageable or pinned), `blks` = BgzfBlk[n] with offsets inside comp / out, `out` = host
// destination (page-locked for a direct DMA).  Returns 0, or a negative number when the device is unusable / a block is corrupt
// (the caller then inflates on the host, which also produces the diagnostics).  Not part of the public ABI.
static uint32_t bgzf_workgroups() {  // resident workgroups of k_bgzf_inflate on the current device
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    return (uint32_t)std::max(1, cus) * (uint32_t)std::max<size_t>(1, (160u << 10) / (sizeof(InfLds) * INF_WAVES));
}
uint32_t plo_internal_bgzf_slots(void) { return bgzf_workgroups() * INF_WAVES; }
// Two slots (stream + device buffers each), so that the caller stages the next group of blocks and checks the CRCs of the previous
// one while the device inflates: begin() enqueues upload, kernel and download of a group, wait() returns when its bytes are in
// `out`.  acquire() / release() bracket a sequence of begin / wait calls (one user at a time).
namespace {
struct InfSlot {
    DevBuf d_comp, d_out, d_blk, d_st;
    HostBuf h_blk, h_st;
    hipStream_t st = nullptr;
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    uint32_t n = 0;
    size_t comp_bytes = 0, out_bytes = 0;
    bool busy = false;
};
struct InfDev {  // one set of slots per device (a reader inflates on the device it was opened for)
    InfSlot slot[2];
    int ok = -1;
};
std::map<int, InfDev> g_inf_devs;
InfDev *g_inf_cur = nullptr;  // the set of the sequence in progress (under g_inf_mu)
int g_inf_prev_dev = -1;      // the calling thread's device before set_device(), restored by release()
std::mutex g_inf_mu;
}  // namespace
void plo_internal_bgzf_acquire(void) {
    g_inf_mu.lock();
    g_inf_cur = nullptr;
    g_inf_prev_dev = -1;
}
int plo_internal_bgzf_set_device(int dev) {  // under acquire(): the sequence runs on `dev` (the calling thread's device until release())
    if (dev < 0) dev = 0;
    int prev = -1;
    if (hipGetDevice(&prev) != hipSuccess) return -100;
    if (hipSetDevice(dev) != hipSuccess) return -100;
    g_inf_prev_dev = prev;
    g_inf_cur = &g_inf_devs[dev];
    return 0;
}
void plo_internal_bgzf_release(void) {
    if (g_inf_prev_dev >= 0) (void)hipSetDevice(g_inf_prev_dev);
    g_inf_cur = nullptr;
    g_inf_prev_dev = -1;
    g_inf_mu.unlock();
}
int plo_internal_bgzf_begin(int slot, const uint8_t *comp, size_t comp_bytes, const void *blks, uint32_t n, uint8_t *out, size_t out_bytes, const uint32_t *crcs) {
    if (slot < 0 || slot > 1) return -101;
    if (!g_inf_cur) {  // no set_device(): the calling thread's current device
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess) return -100;
        g_inf_cur = &g_inf_devs[dev];
    }
    InfDev &D = *g_inf_cur;
    if (D.ok < 0) {
        int nd = 0;
        D.ok = (hipGetDeviceCount(&nd) == hipSuccess && nd > 0 && hipStreamCreateWithFlags(&D.slot[0].st, hipStreamNonBlocking) == hipSuccess &&
                hipStreamCreateWithFlags(&D.slot[1].st, hipStreamNonBlocking) == hipSuccess)
                   ? 1
                   : 0;
    }
    if (!D.ok) return -100;
    InfSlot &q = D.slot[slot];
    q.n = n;
    q.busy = false;
    if (!n) return 0;
    // (the blocks' descriptors and, behind them, their expected CRCs travel as one array)
    const size_t blk_bytes = (size_t)n * sizeof(BgzfBlk), crc_bytes = crcs ? (size_t)n * 4 : 0;
    if (q.d_comp.ensure(comp_bytes + 16) != hipSuccess || q.d_out.ensure(out_bytes + 16) != hipSuccess || q.d_blk.ensure(blk_bytes + crc_bytes) != hipSuccess ||
        q.d_st.ensure((size_t)n * 4) != hipSuccess || q.h_blk.ensure(blk_bytes + crc_bytes) != hipSuccess || q.h_st.ensure((size_t)n * 4) != hipSuccess)
        return -101;
    const bool dbg = getenv("PLO_DEBUG_INFLATE") != nullptr;
    if (dbg && !q.ev[0])
        for (auto &e : q.ev) (void)hipEventCreate(&e);
    memcpy(q.h_blk.p, blks, blk_bytes);  // (the caller's arrays may go away before the copy runs)
    if (crcs) memcpy((uint8_t *)q.h_blk.p + blk_bytes, crcs, crc_bytes);
    hipStream_t st = q.st;
    if (dbg) (void)hipEventRecord(q.ev[0], st);
    if (hipMemcpyAsync(q.d_comp.p, comp, comp_bytes, hipMemcpyHostToDevice, st) != hipSuccess) return -102;
    if (hipMemcpyAsync(q.d_blk.p, q.h_blk.p, blk_bytes + crc_bytes, hipMemcpyHostToDevice, st) != hipSuccess) return -102;
    if (dbg) (void)hipEventRecord(q.ev[1], st);
    hipLaunchKernelGGL(k_bgzf_inflate, dim3(std::min<uint32_t>((n + INF_WAVES - 1) / INF_WAVES, bgzf_workgroups())), dim3(INF_WAVES * 64), 0, st,
                       (const uint8_t *)q.d_comp.p, (const BgzfBlk *)q.d_blk.p, n, (uint8_t *)q.d_out.p, (int *)q.d_st.p);
    if (crcs)
        hipLaunchKernelGGL(k_bgzf_crc, dim3(std::min<uint32_t>((n + INF_WAVES - 1) / INF_WAVES, (uint32_t)bgzf_workgroups() * 4u)), dim3(INF_WAVES * 64), 0, st,
                           (const uint8_t *)q.d_out.p, (const BgzfBlk *)q.d_blk.p, n, (const uint32_t *)((const uint8_t *)q.d_blk.p + blk_bytes), (int *)q.d_st.p);
    if (hipGetLastError() != hipSuccess) return -103;
    if (dbg) (void)hipEventRecord(q.ev[2], st);
    if (hipMemcpyAsync(out, q.d_out.p, out_bytes, hipMemcpyDeviceToHost, st) != hipSuccess) return -104;
    if (hipMemcpyAsync(q.h_st.p, q.d_st.p, (size_t)n * 4, hipMemcpyDeviceToHost, st) != hipSuccess) return -104;
    if (dbg) (void)hipEventRecord(q.ev[3], st);
    q.comp_bytes = comp_bytes;
    q.out_bytes = out_bytes;
    q.busy = true;
    return 0;
}
int plo_internal_bgzf_wait(int slot) {
    if (slot < 0 || slot > 1) return -101;
    if (!g_inf_cur) return 0;  // nothing was begun
    InfSlot &q = g_inf_cur->slot[slot];
    if (!q.busy) return 0;
    q.busy = false;
    if (hipStreamSynchronize(q.st) != hipSuccess) return -105;
    if (getenv("PLO_DEBUG_INFLATE") && q.ev[0]) {
        float a = 0, b = 0, c2 = 0;
        (void)hipEventElapsedTime(&a, q.ev[0], q.ev[1]);
        (void)hipEventElapsedTime(&b, q.ev[1], q.ev[2]);
        (void)hipEventElapsedTime(&c2, q.ev[2], q.ev[3]);
        fprintf(stderr, "[plo] device inflate: %u blocks, %.1f MB -> %.1f MB: H2D %.2f ms, kernel %.2f ms, D2H %.2f ms\n", q.n, q.comp_bytes / 1e6, q.out_bytes / 1e6, a, b, c2);
    }
    const int *stv = q.h_st.as<int>();
    for (uint32_t i = 0; i < q.n; ++i)
        if (stv[i] != 0) return -200;
    return 0;
}
int plo_internal_bgzf_inflate(const uint8_t *comp, size_t comp_bytes, const void *blks, uint32_t n, uint8_t *out, size_t out_bytes) {
    std::lock_guard<std::mutex> g(g_inf_mu);
    g_inf_cur = nullptr;
    int rc = plo_internal_bgzf_begin(0, comp, comp_bytes, blks, n, out, out_bytes, nullptr);
    rc = rc ? rc : plo_internal_bgzf_wait(0);
    g_inf_cur = nullptr;
    return rc;
}

plo_status plo_host_alloc(size_t bytes, void **out) {
    if (!out) return PLO_ERR_INVALID_ARG;
    *out = nullptr;
    void *p = nullptr;
    hipError_t e = hipHostMalloc(&p, bytes ? bytes : 16, hipHostMallocPortable);
    if (e == hipErrorNoDevice || e == hipErrorInsufficientDriver) return PLO_ERR_NO_DEVICE;
    if (e != hipSuccess) return e == hipErrorOutOfMemory ? PLO_ERR_OUT_OF_MEMORY : PLO_ERR_HIP;
    *out = p;
    return PLO_OK;
}
void plo_host_free(void *p) {
    if (p) (void)hipHostFree(p);
}

plo_status plo_ctx_timing(plo_ctx *c, plo_timing *t) {
    if (!c || !t) return PLO_ERR_INVALID_ARG;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    float a = 0, l = 0, b = 0, r = 0, g = 0, md = 0;
    if (!c->fast_no_events) {  // (a one-round-trip call on a context with plo_ctx_set_phase_events(ctx, 0) recorded nothing: counts only)
        (void)hipEventElapsedTime(&a, c->ev[0], c->ev[1]);
        (void)hipEventElapsedTime(&l, c->ev[1], c->ev[4]);
    }
    if (!c->fast_timing) {  // (liftover_fast records events 0, 1 and 4 only: its retry launch and the counters' sum are not timed)
        (void)hipEventElapsedTime(&b, c->ev[4], c->ev[2]);
        (void)hipEventElapsedTime(&r, c->ev[2], c->ev[5]);
    }
    if (c->ev_mid) (void)hipEventElapsedTime(&md, c->ev[5], c->ev[6]);
    if (c->ev_big) (void)hipEventElapsedTime(&g, c->ev_mid ? c->ev[6] : c->ev[5], c->ev[3]);
    c->timing.enumerate_ms = a;
    c->timing.lanes_ms = l;
    c->timing.lift_ms = c->timing.n_heavy_lane_items ? 0.0f : b;
    c->timing.heavy_lanes_ms = c->timing.n_heavy_lane_items ? b : 0.0f;
    c->timing.retry_ms = r;
    c->timing.mid_ms = md;
    c->timing.big_ms = g;
    c->timing.total_ms = a + l + b + r + md + g;
    // the caller says how much of the struct it knows (API version 4): never write past that
    const uint32_t have = t->struct_size;
    if (have < 8u || have > (1u << 16)) {
        c->err = "plo_ctx_timing: plo_timing::struct_size must be set to sizeof(plo_timing) by the caller";
        return PLO_ERR_INVALID_ARG;
    }
    const uint32_t n = std::min<uint32_t>(have, (uint32_t)sizeof(plo_timing));
    c->timing.struct_size = n;
    memcpy(t, &c->timing, n);
    return PLO_OK;
}

plo_status plo_liftover_batch(plo_ctx *c, const plo_batch_in *in, uint32_t stages, plo_batch_out *out) {
    if (!c || !in || !out) return PLO_ERR_INVALID_ARG;
    memset(out, 0, sizeof(*out));
    c->err.clear();
    HIP_TRY(c, hipSetDevice(c->ix->device));
    hipStream_t st = c->stream;
    uint32_t nr = in->n_reads, ns = in->n_segs;
    // NULL arrays are refused here; everything else about the batch (index ranges, 31-bit coordinates, op codes) is checked on
    // the device by the enumerate kernels of plo_liftover_batch_dev
    if (ns && (!in->seg_read || !in->seg_contig || !in->seg_pos || !in->seg_is_fwd_strand || !in->seg_cigar_off) ) {
        c->err = "plo_batch_in: NULL segment array";
        return PLO_ERR_INVALID_ARG;
    }
    if (nr && (!in->read_is_reverse || !in->read_seq_len || !in->read_seq_off)) {
        c->err = "plo_batch_in: NULL read array";
        return PLO_ERR_INVALID_ARG;
    }
    if (in->item_seg && !in->item_cseg) {
        c->err = "plo_batch_in: item_seg without item_cseg";
        return PLO_ERR_INVALID_ARG;
    }
    uint32_t n_cigar = ns ? in->seg_cigar_off[ns] : 0;
    if ((n_cigar && !in->cigar) || (in->seq_bytes && !in->seq)) {
        c->err = "plo_batch_in: NULL cigar / seq array";
        return PLO_ERR_INVALID_ARG;
    }
    if (n_cigar > 0x7fffffffu) {
        c->err = "plo_batch_in: more than 2^31 - 1 CIGAR ops in one batch; split the batch";
        return PLO_ERR_RANGE;
    }
#define UP(buf, src, bytes)                                                                          \
    do {                                                                                             \
        HIP_TRY(c, (buf).ensure(std::max<size_t>((bytes), 16)));                                     \
        if ((bytes) > 0) HIP_TRY(c, hipMemcpyAsync((buf).p, (src), (bytes), hipMemcpyHostToDevice, st)); \
    } while (0)
    UP(c->i_read_rev, in->read_is_reverse, (size_t)nr);
    UP(c->i_read_len, in->read_seq_len, (size_t)nr * 4);
    UP(c->i_read_off, in->read_seq_off, (size_t)nr * 8);
    UP(c->i_seg_read, in->seg_read, (size_t)ns * 4);
    UP(c->i_seg_contig, in->seg_contig, (size_t)ns * 4);
    UP(c->i_seg_pos, in->seg_pos, (size_t)ns * 8);
    UP(c->i_seg_fwd, in->seg_is_fwd_strand, (size_t)ns);
    HIP_TRY(c, c->i_seg_coff.ensure((size_t)(ns + 1) * 4));
    if (ns) {
        HIP_TRY(c, hipMemcpyAsync(c->i_seg_coff.p, in->seg_cigar_off, (size_t)(ns + 1) * 4, hipMemcpyHostToDevice, st));
    } else {
        HIP_TRY(c, hipMemsetAsync(c->i_seg_coff.p, 0, 4, st));
    }
    UP(c->i_cigar, in->cigar, (size_t)n_cigar * 4);
    {   // the bases last, on the copy stream (the buffers are idle: the previous call on this context has been waited for)
        if (!c->copy_stream) HIP_TRY(c, hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
        if (!c->ev_seq) HIP_TRY(c, hipEventCreateWithFlags(&c->ev_seq, hipEventDisableTiming));
        HIP_TRY(c, c->i_seq.ensure(std::max<size_t>((size_t)in->seq_bytes, 16)));
        if (in->seq_bytes) {
            HIP_TRY(c, hipMemcpyAsync(c->i_seq.p, in->seq, (size_t)in->seq_bytes, hipMemcpyHostToDevice, c->copy_stream));
            HIP_TRY(c, hipEventRecord(c->ev_seq, c->copy_stream));
            c->seq_pending = true;
        }
    }
    plo_batch_in din = *in;
    din.read_is_reverse = c->i_read_rev.as<uint8_t>();
    din.read_seq_len = c->i_read_len.as<uint32_t>();
    din.read_seq_off = c->i_read_off.as<uint64_t>();
    din.seq = c->i_seq.as<uint8_t>();
    din.seg_read = c->i_seg_read.as<uint32_t>();
    din.seg_contig = c->i_seg_contig.as<uint32_t>();
    din.seg_pos = c->i_seg_pos.as<int64_t>();
    din.seg_is_fwd_strand = c->i_seg_fwd.as<uint8_t>();
    din.seg_cigar_off = c->i_seg_coff.as<uint32_t>();
    din.cigar = c->i_cigar.as<uint32_t>();
    if (in->item_seg) {
        UP(c->i_item_seg, in->item_seg, (size_t)in->n_items * 4);
        UP(c->i_item_cseg, in->item_cseg, (size_t)in->n_items * 4);
        din.item_seg = c->i_item_seg.as<uint32_t>();
        din.item_cseg = c->i_item_cseg.as<uint32_t>();
    }
#undef UP
    plo_batch_out dout;
    plo_status s = plo_liftover_batch_dev(c, &din, stages, &dout);
    if (c->seq_pending) {  // no lift kernel waited for the bases (an error, or a batch without items): the caller's buffer is still being read
        (void)hipStreamSynchronize(c->copy_stream);
        c->seq_pending = false;
    }
    if (s != PLO_OK) return s;
    s = plo_compact_output_dev(c, &dout);  // no slab gaps over the bus
    if (s != PLO_OK) return s;
    size_t ni = dout.n_items, nc = (size_t)dout.n_cigar;
#define DOWN(hbuf, src, bytes)                                                                        \
    do {                                                                                              \
        HIP_TRY(c, (hbuf).ensure(std::max<size_t>((bytes), 16)));                                     \
        if ((bytes) > 0) HIP_TRY(c, hipMemcpyAsync((hbuf).p, (src), (bytes), hipMemcpyDeviceToHost, st)); \
    } while (0)
    DOWN(c->h_item_seg, dout.item_seg, ni * 4);
    DOWN(c->h_item_cseg, dout.item_cseg, ni * 4);
    DOWN(c->h_status, dout.item_status, ni);
    DOWN(c->h_flip, dout.item_need_flipped, ni);
    DOWN(c->h_mapq, dout.item_mapq, ni);
    DOWN(c->h_chrom, dout.item_chrom_index, ni * 4);
    DOWN(c->h_pos, dout.item_ref_pos, ni * 8);
    DOWN(c->h_coff, dout.item_cigar_off, ni * 8);
    DOWN(c->h_clen, dout.item_cigar_len, ni * 4);
    DOWN(c->h_cigar, dout.cigar, nc * 4);
#undef DOWN
    HIP_TRY(c, hipStreamSynchronize(st));
    out->n_items = dout.n_items;
    out->item_seg = c->h_item_seg.as<uint32_t>();
    out->item_cseg = c->h_item_cseg.as<uint32_t>();
    out->item_status = c->h_status.as<uint8_t>();
    out->item_need_flipped = c->h_flip.as<uint8_t>();
    out->item_mapq = c->h_mapq.as<uint8_t>();
    out->item_chrom_index = c->h_chrom.as<uint32_t>();
    out->item_ref_pos = c->h_pos.as<int64_t>();
    out->item_cigar_off = c->h_coff.as<uint64_t>();
    out->item_cigar_len = c->h_clen.as<uint32_t>();
    out->cigar = c->h_cigar.as<uint32_t>();
    out->n_cigar = dout.n_cigar;
    return PLO_OK;
}


// Runs the wave primitives on the device and checks them against host-computed expectations.
// 0 = ok, >0 = index of the first failing primitive + 1, <0 = HIP error.
int plo_selftest(int device) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return -1;
    if (hipSetDevice(device) != hipSuccess) return -1;
    std::vector<int> in(256), out(14 * 64, 0), exp(14 * 64, 0);
    unsigned rs = 12345u;
    auto rnd = [&]() {
        rs = rs * 1664525u + 1013904223u;
        return (int)(rs >> 8);
    };
    for (int l = 0; l < 64; ++l) {
        in[l] = rnd() % 1000 - 300;
        in[64 + l] = rnd() % 50;
        in[128 + l] = (rnd() % 4 == 0) ? 0x7fffffff : rnd() % 80;
        in[192 + l] = (rnd() % 9 == 0) ? 1 : 0;
    }
    int acc = 0, mx = (int)0x80000000;
    int fa = 0, fb = 0x7fffffff, fs = 0;
    int c0 = 0, c1 = 0, lm = -1;
    for (int l = 0; l < 64; ++l) c1 += in[l] & 15;
    for (int l = 0; l < 64; ++l) {
        int x = in[l];
        acc += x;
        mx = std::max(mx, x);
        exp[0 * 64 + l] = acc;
        exp[1 * 64 + l] = mx;
        exp[2 * 64 + l] = l ? in[l - 1] : -7;
        exp[3 * 64 + l] = in[(l * 7 + 3) & 63];
        exp[4 * 64 + l] = in[63];
        exp[5 * 64 + l] = in[0];
        int a = in[64 + l], b = in[128 + l], sflag = in[192 + l];
        if (sflag) {
            fa = a;
            fb = b;
            fs = 1;
        } else {
            long long na = (long long)fa + a, nb = (long long)fb + a;
            fa = na > 0x7fffffffLL ? 0x7fffffff : (int)na;
            int nbb = nb > 0x7fffffffLL ? 0x7fffffff : (int)nb;
            fb = std::min(nbb, b);
        }
        exp[6 * 64 + l] = fa;
        exp[7 * 64 + l] = fb;
        exp[8 * 64 + l] = fs;
        exp[9 * 64 + l] = x & 1;
        exp[10 * 64 + l] = c0;
        c0 += x & 15;
        exp[11 * 64 + l] = c1;
        c1 += (x >> 4) & 15;
        exp[12 * 64 + l] = lm;
        if ((x & 3) == 0) lm = l;
    }
    int lm2 = lm;
    for (int l = 0; l < 64; ++l) {
        exp[13 * 64 + l] = lm2;
        if ((in[l] & 3) == 1) lm2 = std::max(lm2, 64 + l);
    }
    int *din = nullptr, *dout = nullptr;
    if (hipMalloc(&din, in.size() * 4) != hipSuccess || hipMalloc(&dout, out.size() * 4) != hipSuccess) return -2;
    if (hipMemcpy(din, in.data(), in.size() * 4, hipMemcpyHostToDevice) != hipSuccess) return -3;
    hipLaunchKernelGGL(k_selftest, dim3(1), dim3(64), 0, 0, (const int *)din, dout);
    if (hipDeviceSynchronize() != hipSuccess) return -4;
    if (hipMemcpy(out.data(), dout, out.size() * 4, hipMemcpyDeviceToHost) != hipSuccess) return -5;
    (void)hipFree(din);
    (void)hipFree(dout);
    for (int k = 0; k < 14; ++k)
        for (int l = 0; l < 64; ++l)
            if (out[k * 64 + l] != exp[k * 64 + l]) {
                fprintf(stderr, "plo_selftest: primitive %d lane %d got %d expected %d\n", k, l, out[k * 64 + l], exp[k * 64 + l]);
                return k + 1;
            }
    return 0;
}

}  // extern "C"
