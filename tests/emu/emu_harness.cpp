// tests/emu/emu_harness.cpp -- runs the DEVICE algorithm (portello_amd/csrc/lift_core.hpp) on the CPU under the
// wave64 emulator (tests/emu/plo_wave.hpp).  TEST INFRASTRUCTURE ONLY: never linked into the product library.
#include <plo_wave.hpp>

#include <algorithm>
#include <string>
#include <vector>

#include "../../portello_amd/csrc/enumerate.hpp"
#include "../../portello_amd/csrc/finish_core.hpp"
#include "../../portello_amd/csrc/index_pack.hpp"
#include "../../portello_amd/csrc/inflate.hpp"
#include "../../portello_amd/csrc/lift_core.hpp"
#include "../../portello_amd/csrc/lane_core.hpp"
#include "../../portello_amd/csrc/lane_stream.hpp"

using namespace plo;

namespace {
struct Out {
    std::vector<uint32_t> item_seg, item_cseg, chrom, cig_len, cigar;
    std::vector<uint8_t> status, flip, mapq;
    std::vector<int64_t> pos;
    std::vector<uint64_t> cig_off;
};
}  // namespace

extern "C" int emu_liftover_batch(const plo_index_desc *ixd, const plo_batch_in *in, uint32_t stages, int cap, int window,
                                  int big_thresh, int big_cap, unsigned order_seed, int mid_waves, int mid_cap, int lane_max_w, int lane_capw,
                                  int lane_heavy_per, int lane_budget, int lane_stream, plo_batch_out *out, unsigned long long *counters_out) {
    // lane_stream (with lane_heavy_per > 0, all stages): the heavy classes through the streaming kernel (lane_stream.hpp) -- teams of three waves, 1: the GPU's
    // rings, 2: the smallest Q2 the code allows (unreleased tails outgrow it: retry list)
    // lane_max_w >= 0: items up to that weight run through the lane-per-item code (lane_core.hpp) with an LDS slice of lane_capw
    // dwords per wave; what it cannot hold goes to the retry list like on the GPU
    // mid_waves: 0 = items beyond big_thresh run one wave each (LEVEL_LAST); 2..16 = they first go through the workgroup-per-item
    // code (lift_tile<NW>, LEVEL_MID) with an LDS capacity of mid_cap elements, under the multi-wave emulator
    PackedIndex pk;
    std::string err;
    if (pack_index(ixd, pk, err) != PLO_OK) {
        fprintf(stderr, "emu: pack_index failed: %s\n", err.c_str());
        return 1;
    }
    DevIndex ix;
    ix.kv = pk.kv.data();
    ix.cs_kv_off = pk.cs_kv_off.data();
    ix.cs_chrom = pk.cs_chrom.data();
    ix.cs_is_fwd = pk.cs_is_fwd.data();
    ix.cs_mapq = pk.cs_mapq.data();
    ix.cs_start = pk.cs_start.data();
    ix.cs_end = pk.cs_end.data();
    ix.contig_seg_off = pk.contig_seg_off.data();
    ix.contig_len = pk.contig_len.data();
    ix.contig_revseq = ixd->rev_contig_seq;
    ix.chrom_seq = ixd->chrom_seq;
    ix.chrom_len = pk.chrom_len.data();
    ix.n_contigs = ixd->n_contigs;
    ix.n_segments = ixd->n_segments;
    ix.n_chroms = ixd->n_chroms;

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

    Out *o = new Out();
    // ---- item list: count, then resolve descriptors (what k_seg_count / k_item_emit / k_item_desc do on the GPU) ----
    std::vector<uint32_t> seg_cnt(in->n_segs, 0);
    uint32_t n_items = 0;
    if (in->item_seg) {
        n_items = in->n_items;
    } else {
        for (uint32_t s = 0; s < in->n_segs; ++s) {
            seg_cnt[s] = enumerate_segment(ix, bt, s, nullptr, 0, 0);
            n_items += seg_cnt[s];
        }
    }
    size_t a = n_items ? n_items : 1;
    o->item_seg.assign(a, 0);
    o->item_cseg.assign(a, 0);
    o->status.assign(a, 0xEE);
    o->flip.assign(a, 0);
    o->mapq.assign(a, 0);
    o->chrom.assign(a, 0);
    o->pos.assign(a, 0);
    o->cig_off.assign(a, 0);
    o->cig_len.assign(a, 0);
    std::vector<uint32_t> item_cls(a, 0), item_nin(a, 0), d_in_off(a), d_n_in(a), d_w0(a), d_w1(a), d_kv0(a), d_kv1(a), d_flags(a), d_contig(a),
        d_seq_len(a), big_list(a, 0), huge_list(a, 0), miss_list(a, 0);
    std::vector<int> d_pos1(a);
    std::vector<uint64_t> d_seq_off(a), d_shift_ref(a), d_chrom_ref(a);
    std::vector<int> d_shift_ref_len(a), d_chrom_ref_len(a);
    std::vector<uint32_t> d_read_len(a), d_n_m(a);
    unsigned long long counters[CNT_N];
    memset(counters, 0, sizeof(counters));

    DevWork wk;
    memset(&wk, 0, sizeof(wk));
    wk.n_items = n_items;
    wk.lane_max_w = lane_max_w;
    wk.item_seg = o->item_seg.data();
    wk.item_cseg = o->item_cseg.data();
    wk.item_nin = item_nin.data();
    wk.item_cls = item_cls.data();
    std::vector<uint32_t> item_region(a, 0);
    wk.item_region = item_region.data();
    wk.d.in_off = d_in_off.data();
    wk.d.n_in = d_n_in.data();
    wk.d.pos1 = d_pos1.data();
    wk.d.w0 = d_w0.data();
    wk.d.w1 = d_w1.data();
    wk.d.kv0 = d_kv0.data();
    wk.d.kv1 = d_kv1.data();
    wk.d.flags = d_flags.data();
    wk.d.contig = d_contig.data();
    wk.d.seq_len = d_seq_len.data();
    wk.d.seq_off = d_seq_off.data();
    wk.d.shift_ref = d_shift_ref.data();
    wk.d.shift_ref_len = d_shift_ref_len.data();
    wk.d.chrom_ref = d_chrom_ref.data();
    wk.d.chrom_ref_len = d_chrom_ref_len.data();
    wk.d.read_len = d_read_len.data();
    wk.d.n_m = d_n_m.data();
    wk.status = o->status.data();
    wk.flip = o->flip.data();
    wk.mapq = o->mapq.data();
    wk.chrom = o->chrom.data();
    wk.pos = o->pos.data();
    wk.cig_off = o->cig_off.data();
    wk.cig_len = o->cig_len.data();
    wk.counters = counters;
    unsigned long long wave_stats[STAT_WORDS] = {0, 0, 0, 0, 0, 0, 0, 0};  // the emulated waves run one after another: one slot, summed after each
    wk.wave_stats = wave_stats;
    auto sum_stats = [&]() {  // k_sum_stats
        counters[CNT_ALGO_BYTES] += wave_stats[0];
        counters[CNT_IN_OPS] += wave_stats[1];
        counters[CNT_OUT_OPS] += wave_stats[2];
        counters[CNT_LANE_ACT] += wave_stats[3];
        counters[CNT_LANE_TRIPS] += wave_stats[4];
        wave_stats[0] = wave_stats[1] = wave_stats[2] = wave_stats[3] = wave_stats[4] = 0;
    };
    wk.big_list = big_list.data();
    wk.huge_list = huge_list.data();
    wk.miss_list = miss_list.data();
    const bool sp = in->seq_fmt == PLO_SEQ_BAM4_SPARSE;  // (sparse bases: such items keep PLO_ITEM_NEED_BASES here, the second look is the engine's)
    if (in->item_seg) {
        for (uint32_t i = 0; i < n_items; ++i)
            build_item_desc(ix, bt, wk, stages, i, in->item_seg[i], in->item_cseg[i], segment_ref_len(bt, in->item_seg[i]));
    } else {
        uint32_t off = 0;
        for (uint32_t s = 0; s < in->n_segs; ++s) {
            if (seg_cnt[s]) enumerate_segment(ix, bt, s, &wk, stages, off);
            off += seg_cnt[s];
        }
    }
    // class order (k_class_flags + scans + k_permute on the GPU)
    std::vector<uint32_t> r0(n_items + 1, 0), r1(n_items + 1, 0), r2(n_items + 1, 0), perm(a, 0), nin_p(a, 0), retry_list(a, 0);
    for (uint32_t i = 0; i < n_items; ++i) {
        r0[i + 1] = r0[i] + (item_cls[i] == 0);
        r1[i + 1] = r1[i] + (item_cls[i] == 1);
        r2[i + 1] = r2[i] + (item_cls[i] == 2);
    }
    for (uint32_t i = 0; i < n_items; ++i) {
        uint32_t j = class_order_pos(i, item_cls[i], r0[i], r1[i], r2[i], r0[n_items], r1[n_items], r2[n_items]);
        perm[j] = i;
        nin_p[j] = item_cls[i] >= 2 ? item_nin[i] : 0;  // only the large items are tiled
    }
    const uint32_t n_small = r0[n_items] + r1[n_items], n_large = n_items - n_small;
    wk.perm = perm.data();
    wk.n_small = n_small;
    wk.retry_list = retry_list.data();
    std::vector<uint32_t> prefix(n_items + 1, 0);
    uint64_t all_ops = 0;
    for (uint32_t i = 0; i < n_items; ++i) all_ops += item_nin[i];
    for (uint32_t i = 0; i < n_items; ++i) prefix[i + 1] = prefix[i] + nin_p[i];
    uint64_t total_ops = prefix[n_items];
    wk.item_op_prefix = prefix.data();
    uint32_t n_tiles = (uint32_t)(total_ops / (uint64_t)window) + 1;
    std::vector<uint32_t> tile_lo(n_tiles + 1);
    for (uint32_t t = 0; t <= n_tiles; ++t)
        tile_lo[t] = std::max(n_small, prefix_lower_bound(prefix.data(), n_items, (unsigned long long)t * (unsigned)window));  // k_tile_bounds
    wk.tile_lo = tile_lo.data();
    total_ops = all_ops;

    uint64_t out_cap = 4 * total_ops + 64ull * n_items + 1024 + 8 * SLAB_OPS;
    for (int attempt = 0; attempt < 6; ++attempt) {
        o->cigar.assign(out_cap, 0);
        memset(counters, 0, sizeof(counters));
        wk.out_cigar = o->cigar.data();
        wk.out_cap = out_cap;
        std::vector<unsigned char> lds(tile_mem_bytes(cap) + 64);
        const uint32_t n_waves = 3;  // persistent waves striding over the tiles, like k_lift_tiles
        // lane_heavy_per > 0: the heavy classes through the lane-per-item code with fixed regions (k_lift_lanes_g) instead of the tiles
        const bool heavy_lanes = lane_max_w >= 0 && lane_heavy_per > 0 && n_large > 0;
        if (heavy_lanes) {
            uint32_t max_w = 0;
            for (uint32_t i = 0; i < n_items; ++i) max_w = std::max(max_w, item_nin[i]);
            int stride = (int)((max_w + LANE_SLACK + 64u + LANE_REGION_PAD + 31u) & ~31u);
            if (const char *e = getenv("PLO_EMU_HEAVY_STRIDE")) stride = std::max(LANE_REGION_PAD + 64, atoi(e)) & ~31;  // (regions too small for the longer items: retry list)
            std::vector<uint32_t> regions((size_t)lane_heavy_per * stride + 16, 0xdeadbeefu), windows((size_t)64 * LANE_WIN_DWORDS + LANE_KVS_DWORDS, 0xdeadbeefu);
            const bool stream = lane_stream > 0 && (stages & PLO_STAGES_ALL) == PLO_STAGES_ALL;
            std::vector<uint32_t> slds((size_t)stream_lds_dwords(32, 16, 16, 32) + 16, 0xdeadbeefu);
            if (stream) {
                // teams of three waves (lane_stream.hpp): `lane_heavy_per` items per team, the forward class first
                const uint32_t lo_ = n_small, mid_ = n_small + r2[n_items], hi_ = n_items;
                const uint32_t per_ = (uint32_t)lane_heavy_per;
                const uint32_t t0 = (mid_ - lo_ + per_ - 1) / per_, t1 = (hi_ - mid_ + per_ - 1) / per_;
                for (uint32_t team = 0; team < t0 + t1; ++team) {
                    wv::EmuWave w;
                    w.nw = PIPE_WAVES;
                    w.order_seed = order_seed ? order_seed + 61 + team : 0;
                    std::fill(slds.begin(), slds.end(), 0xdeadbeefu);
                    w.run([&]() {
                        WaveCtx ctx;
                        uint32_t b_, e_;
                        bool hs;
                        pipe_team_span(team, t0, t1, lo_, mid_, hi_, b_, e_, hs);
                        if (lane_stream == 1) {
                            if (sp) pipe_team<true, 32, 16, 16, 32>(ix, bt, wk, b_, e_, hs, slds.data(), ctx);
                            else pipe_team<false, 32, 16, 16, 32>(ix, bt, wk, b_, e_, hs, slds.data(), ctx);
                        } else {
                            if (sp) pipe_team<true, 32, 16, 8, 32>(ix, bt, wk, b_, e_, hs, slds.data(), ctx);
                            else pipe_team<false, 32, 16, 8, 32>(ix, bt, wk, b_, e_, hs, slds.data(), ctx);
                        }
                        // the waves of the emulated workgroup share the one statistics slot: one wave at a time
                        for (int ww = 0; ww < PIPE_WAVES; ++ww) {
                            if (ww == wv::wave_id()) {
                                wave_ctx_flush(wk, ctx, 0);
                                if (wv::lane() == 0) sum_stats();
                            }
                            wv::block_sync();
                        }
                    });
                }
            }
            for (uint32_t wv_id = 0; wv_id < n_waves && !stream; ++wv_id) {
                wv::EmuWave w;
                w.order_seed = order_seed ? order_seed + 61 + wv_id : 0;
                w.run([&]() {
                    WaveCtx ctx;
                    if (sp) lane_heavy_persistent<true>(ix, bt, wk, stages, wv_id, n_waves, n_small, n_small + r2[n_items], n_items, (uint32_t)lane_heavy_per, windows.data(), regions.data(), stride, ctx);
                    else lane_heavy_persistent<false>(ix, bt, wk, stages, wv_id, n_waves, n_small, n_small + r2[n_items], n_items, (uint32_t)lane_heavy_per, windows.data(), regions.data(), stride, ctx);
                    wave_ctx_flush(wk, ctx, 0);
                });
                sum_stats();
            }
        }
        // (items per group as the host picks them for small batches: 64 without an order seed, else 64 / 32 / 16 / 8 by the seed)
        const uint32_t lane_group = order_seed ? 64u >> (order_seed % 4u) : 64u;
        // lane_budget: the groups cut by LDS budget inside windows of 128 positions of either class, as k_chunk_sort does on the GPU
        // (lane_groups_cut, enumerate.hpp) -- listed groups of fewer than 64 items instead of fixed ones
        std::vector<uint32_t> glist;
        uint32_t n_glist = 0;
        if (n_small && lane_budget && lane_group == 64u) {
            for (int cls = 0; cls < 2; ++cls) {
                const uint32_t c_lo = cls ? r0[n_items] : 0u, c_hi = cls ? n_small : r0[n_items];
                for (uint32_t lo = c_lo; lo < c_hi; lo += 128u) {
                    const uint32_t n = std::min(128u, c_hi - lo);
                    std::vector<uint32_t> reg(n), start(n + 1);
                    for (uint32_t k = 0; k < n; ++k) reg[k] = item_region[perm[lo + k]];
                    const uint32_t ng = lane_groups_cut(reg.data(), n, (uint32_t)lane_capw, start.data(), n);
                    for (uint32_t k = 0; k < ng; ++k) {
                        glist.push_back(lo + start[k]);
                        glist.push_back(start[k + 1] - start[k]);
                    }
                    n_glist += ng;
                }
            }
            wk.lane_groups = glist.data();
            wk.lane_n_groups = &n_glist;
            wk.lane_groups_cap = n_glist;
        }
        if (n_small) {  // k_lift_lanes: persistent waves over the groups of the two lane classes
            std::vector<uint32_t> llds((size_t)lane_capw + LANE_KVS_DWORDS + 16, 0xdeadbeefu);
            // groups dealt by tickets behind 1 .. 3 rounds of fixed slots (runs with an order seed; the emulated waves run one after the
            // other, so the first wave to get there takes every ticket: each group must still be lifted exactly once)
            uint32_t ticket = 0, ticket_next = 0xffffffffu;
            if (order_seed) {
                wk.lane_ticket = &ticket;
                wk.lane_ticket_next = &ticket_next;
                wk.lane_static_rounds = 1u + (order_seed >> 2) % 3u;
                wk.lane_tail_rounds = (order_seed >> 4) % 3u;
            }
            for (uint32_t wv_id = 0; wv_id < n_waves; ++wv_id) {
                wv::EmuWave w;
                w.order_seed = order_seed ? order_seed + 31 + wv_id : 0;
                w.run([&]() {
                    WaveCtx ctx;
                    // (PLO_EMU_H16=1: 16-bit regions for the stage sets with the liftover, as the engine routes them under PLO_LANE_H16=1)
                    const bool h16 = (stages & PLO_STAGE_LIFTOVER) != 0u && getenv("PLO_EMU_H16") && atoi(getenv("PLO_EMU_H16")) != 0;
                    if (sp && h16) lane_tiles_persistent<true, false, true, true>(ix, bt, wk, stages, wv_id, n_waves, r0[n_items], r1[n_items], lane_group, llds.data(), lane_capw, ctx);
                    else if (sp) lane_tiles_persistent<true>(ix, bt, wk, stages, wv_id, n_waves, r0[n_items], r1[n_items], lane_group, llds.data(), lane_capw, ctx);
                    else if (h16) lane_tiles_persistent<false, false, true, true>(ix, bt, wk, stages, wv_id, n_waves, r0[n_items], r1[n_items], lane_group, llds.data(), lane_capw, ctx);
                    else lane_tiles_persistent<false>(ix, bt, wk, stages, wv_id, n_waves, r0[n_items], r1[n_items], lane_group, llds.data(), lane_capw, ctx);
                    wave_ctx_flush(wk, ctx, 0);
                });
                sum_stats();
            }
        }
        for (uint32_t wv_id = 0; wv_id < n_waves && n_large && !heavy_lanes; ++wv_id) {
            wv::EmuWave w;
            w.order_seed = order_seed ? order_seed + wv_id : 0;
            TileMem m = carve_tile_mem(lds.data(), cap);
            w.run([&]() {
                WaveCtx ctx;
                if (sp) lift_tiles_persistent<true>(ix, bt, wk, stages, wv_id, n_waves, n_tiles, big_thresh, m, ctx);
                else lift_tiles_persistent<false>(ix, bt, wk, stages, wv_id, n_waves, n_tiles, big_thresh, m, ctx);
                wave_ctx_flush(wk, ctx, 0);
            });
            sum_stats();
        }
        {
            const int retry_cap = std::max(cap, (2 * big_thresh + 64 + 63) & ~63);
            std::vector<unsigned char> rlds(tile_mem_bytes(retry_cap) + 64);
            // retry list -> tile code, a few items per wave
            uint32_t n_retry = (uint32_t)counters[CNT_NRETRY];
            const uint32_t per = 1;
            for (uint32_t r = 0; r < n_retry; r += per) {
                wv::EmuWave w;
                w.order_seed = order_seed ? order_seed + 555 + r : 0;
                TileMem m = carve_tile_mem(rlds.data(), retry_cap);
                w.run([&]() {
                    WaveCtx ctx;
                    Coop<1> co;
                    if (sp) lift_tile<std::remove_reference_t<decltype(co)>::NWAVES, true>(co, ix, bt, wk, stages, r, (int)std::min<uint32_t>(per, n_retry - r), m, wk.retry_list, LEVEL_RETRY, big_thresh, ctx);
 else lift_tile<std::remove_reference_t<decltype(co)>::NWAVES, false>(co, ix, bt, wk, stages, r, (int)std::min<uint32_t>(per, n_retry - r), m, wk.retry_list, LEVEL_RETRY, big_thresh, ctx);
                    wave_ctx_flush(wk, ctx, 0);
                });
                sum_stats();
            }
        }
        uint32_t n_big = (uint32_t)counters[CNT_NBIG];
        const uint32_t *last_list = wk.big_list;
        if (n_big && mid_waves > 1) {  // k_lift_mid: one emulated workgroup per item
            auto run_mid = [&](auto nw_c) {
                constexpr int NW = decltype(nw_c)::value;
                std::vector<unsigned char> mlds(((tile_mem_bytes(mid_cap) + 15) & ~(size_t)15) + Coop<NW>::XCH_INTS * 4 + 64);
                const int mid_thresh = (mid_cap - 64) * 4 / 5;
                const uint32_t n_blocks = 2;
                for (uint32_t blk = 0; blk < n_blocks; ++blk) {
                    wv::EmuWave w;
                    w.nw = NW;
                    w.order_seed = order_seed ? order_seed + 4242 + blk : 0;
                    TileMem m = carve_tile_mem(mlds.data(), mid_cap);
                    int *xch = (int *)(mlds.data() + ((tile_mem_bytes(mid_cap) + 15) & ~(size_t)15));
                    memset(xch, 0, Coop<NW>::XCH_INTS * 4);
                    w.run([&]() {
                        WaveCtx ctx;
                        Coop<NW> co;
                        co.w = wv::wave_id();
                        co.xch = xch;
                        for (uint32_t i = blk; i < n_big; i += n_blocks) {
                            if (sp) lift_tile<std::remove_reference_t<decltype(co)>::NWAVES, true>(co, ix, bt, wk, stages, i, 1, m, wk.big_list, LEVEL_MID, mid_thresh, ctx);
 else lift_tile<std::remove_reference_t<decltype(co)>::NWAVES, false>(co, ix, bt, wk, stages, i, 1, m, wk.big_list, LEVEL_MID, mid_thresh, ctx);
                            co.sync();
                        }
                        // the waves of the emulated workgroup share the one statistics slot: one wave at a time
                        for (int ww = 0; ww < NW; ++ww) {
                            if (ww == co.w) {
                                wave_ctx_flush(wk, ctx, 0);
                                if (wv::lane() == 0) sum_stats();
                            }
                            co.sync();
                        }
                    });
                }
            };
            if (mid_waves >= 16) run_mid(std::integral_constant<int, 16>{});
            else if (mid_waves >= 8) run_mid(std::integral_constant<int, 8>{});
            else if (mid_waves >= 4) run_mid(std::integral_constant<int, 4>{});
            else run_mid(std::integral_constant<int, 2>{});
            n_big = (uint32_t)counters[CNT_NHUGE];
            last_list = wk.huge_list;
        }
        if (n_big) {
            std::vector<unsigned char> scratch(tile_mem_bytes(big_cap) + 64);
            const uint32_t n_bw = 2;
            for (uint32_t wv_id = 0; wv_id < n_bw; ++wv_id) {
                wv::EmuWave w;
                w.order_seed = order_seed ? order_seed + 7777 + wv_id : 0;
                TileMem m = carve_tile_mem(scratch.data(), big_cap);
                w.run([&]() {
                    WaveCtx ctx;
                    Coop<1> co;
                    for (uint32_t i = wv_id; i < n_big; i += n_bw) {
                        if (sp) lift_tile<std::remove_reference_t<decltype(co)>::NWAVES, true>(co, ix, bt, wk, stages, i, 1, m, last_list, LEVEL_LAST, 0, ctx);
 else lift_tile<std::remove_reference_t<decltype(co)>::NWAVES, false>(co, ix, bt, wk, stages, i, 1, m, last_list, LEVEL_LAST, 0, ctx);
                        wv::sync();
                    }
                    wave_ctx_flush(wk, ctx, 0);
                });
                sum_stats();
            }
        }
        if (counters[CNT_OVERFLOW] == 0) break;
        out_cap = counters[CNT_CIGAR] + 8 * SLAB_OPS;
    }
    if (counters_out) {
        memcpy(counters_out, counters, sizeof(counters));
        counters_out[23] = n_small;  // items that took the lane-per-item path (test visibility)
    }

    out->n_items = n_items;
    out->item_seg = o->item_seg.data();
    out->item_cseg = o->item_cseg.data();
    out->item_status = o->status.data();
    out->item_need_flipped = o->flip.data();
    out->item_mapq = o->mapq.data();
    out->item_chrom_index = o->chrom.data();
    out->item_ref_pos = o->pos.data();
    out->item_cigar_off = o->cig_off.data();
    out->item_cigar_len = o->cig_len.data();
    out->cigar = o->cigar.data();
    out->n_cigar = counters[CNT_CIGAR];
    // the Out object stays alive until emu_free_last()
    extern void *g_last_emu_out;
    g_last_emu_out = o;
    return counters[CNT_ERROR] ? 2 : 0;
}

void *g_last_emu_out = nullptr;

extern "C" void emu_free_last() {
    delete (Out *)g_last_emu_out;
    g_last_emu_out = nullptr;
}

// ---- record finishing (finish_core.hpp) run on the host: what k_finish_items / k_finish_reads / k_finish_offsets /
// k_revcomp do on the GPU, with plain loops standing in for the threads ------------------------------------------------
namespace {
struct FinOut {
    std::vector<uint16_t> flag, bin, uflag;
    std::vector<int64_t> rend;
    std::vector<uint8_t> prim, rseq, rqual;
    std::vector<uint64_t> isoff, iqoff, rsoff, rqoff;
    std::vector<uint32_t> iread, nl, pitem, su, qu, soff, qoff, fflag, frank, flist;
};
FinOut *g_fin = nullptr;
}  // namespace

extern "C" int emu_finish_batch(const plo_batch_in *in, const plo_finish_in *fin, const plo_batch_out *lift, int nthreads,
                                plo_finish_out *out) {
    DevBatch bt;
    memset(&bt, 0, sizeof(bt));
    bt.read_seq_len = in->read_seq_len;
    bt.read_seq_off = in->read_seq_off;
    bt.seq = in->seq;
    bt.seq_fmt = in->seq_fmt;
    bt.seq_bytes = in->seq_bytes;
    bt.seg_read = in->seg_read;
    bt.n_reads = in->n_reads;
    bt.n_segs = in->n_segs;
    DevWork wk;
    memset(&wk, 0, sizeof(wk));
    uint32_t n = lift->n_items, nr = in->n_reads, ne = n + nr;
    wk.n_items = n;
    wk.item_seg = (uint32_t *)lift->item_seg;
    wk.status = (uint8_t *)lift->item_status;
    wk.flip = (uint8_t *)lift->item_need_flipped;
    wk.mapq = (uint8_t *)lift->item_mapq;
    wk.pos = (int64_t *)lift->item_ref_pos;
    wk.cig_off = (uint64_t *)lift->item_cigar_off;
    wk.cig_len = (uint32_t *)lift->item_cigar_len;
    wk.out_cigar = (uint32_t *)lift->cigar;
    delete g_fin;
    FinOut *o = g_fin = new FinOut();
    size_t a = n ? n : 1, b = nr ? nr : 1, e = ne ? ne : 1;
    o->flag.assign(a, 0); o->bin.assign(a, 0); o->rend.assign(a, 0); o->prim.assign(a, 0); o->isoff.assign(a, 0); o->iqoff.assign(a, 0);
    o->iread.assign(a, 0); o->nl.assign(b, 0); o->pitem.assign(b, 0); o->uflag.assign(b, 0); o->rsoff.assign(b, 0); o->rqoff.assign(b, 0);
    o->su.assign(e, 0); o->qu.assign(e, 0); o->soff.assign(e + 1, 0); o->qoff.assign(e + 1, 0); o->fflag.assign(e, 0);
    o->frank.assign(e + 1, 0); o->flist.assign(e, 0);
    DevFinish f;
    memset(&f, 0, sizeof(f));
    unsigned n_fault = 0;
    f.n_fault = &n_fault;
    f.read_flags = fin->read_flags; f.qual = fin->qual; f.read_qual_off = fin->read_qual_off; f.qual_bytes = fin->qual_bytes;
    f.seq_bytes = in->seq_bytes;
    f.item_flag = o->flag.data(); f.item_bin = o->bin.data(); f.item_ref_end = o->rend.data(); f.item_is_primary = o->prim.data();
    f.item_seq_off = o->isoff.data(); f.item_qual_off = o->iqoff.data(); f.item_read = o->iread.data();
    f.read_n_lifted = o->nl.data(); f.read_primary_item = o->pitem.data(); f.read_unmapped_flag = o->uflag.data();
    f.read_seq_off = o->rsoff.data(); f.read_qual_off_out = o->rqoff.data();
    f.su = o->su.data(); f.qu = o->qu.data(); f.soff = o->soff.data(); f.qoff = o->qoff.data();
    f.fflag = o->fflag.data(); f.frank = o->frank.data(); f.flist = o->flist.data();
    for (uint32_t i = 0; i < n; ++i) finish_item(bt, wk, f, i);
    for (uint32_t r = 0; r < nr; ++r) finish_read(bt, wk, f, r);
    for (uint32_t k = 0; k < ne; ++k) {
        o->soff[k + 1] = o->soff[k] + o->su[k];
        o->qoff[k + 1] = o->qoff[k] + o->qu[k];
        o->frank[k + 1] = o->frank[k] + o->fflag[k];
    }
    for (uint32_t k = 0; k < ne; ++k) {
        uint64_t so = o->su[k] ? (uint64_t)o->soff[k] * 16u : PLO_NO_FLIP, qo = o->qu[k] ? (uint64_t)o->qoff[k] * 16u : PLO_NO_FLIP;
        if (o->su[k]) o->flist[o->frank[k]] = k;
        if (k < n) { o->isoff[k] = so; o->iqoff[k] = qo; } else { o->rsoff[k - n] = so; o->rqoff[k - n] = qo; }
    }
    // 16-byte aligned destination buffers (vector of uint64 pairs)
    o->rseq.assign((size_t)o->soff[ne] * 16 + 32, 0xEE);
    o->rqual.assign((size_t)o->qoff[ne] * 16 + 32, 0xEE);
    uint8_t *rs = o->rseq.data() + ((16 - ((uintptr_t)o->rseq.data() & 15)) & 15);
    uint8_t *rq = o->rqual.data() + ((16 - ((uintptr_t)o->rqual.data() & 15)) & 15);
    f.rev_seq = rs;
    f.rev_qual = rq;
    for (uint32_t k = 0; k < o->frank[ne]; ++k) {
        uint32_t en = o->flist[k];
        uint32_t read = en < n ? o->iread[en] : en - n;
        for (int t = 0; t < nthreads; ++t)
            revcomp_record(bt, f, read, rs + (uint64_t)o->soff[en] * 16u, rq + (uint64_t)o->qoff[en] * 16u, t, nthreads);
    }
    out->item_flag = f.item_flag; out->item_bin = f.item_bin; out->item_ref_end = f.item_ref_end; out->item_is_primary = f.item_is_primary;
    out->item_seq_off = f.item_seq_off; out->item_qual_off = f.item_qual_off; out->read_n_lifted = f.read_n_lifted;
    out->read_primary_item = f.read_primary_item; out->read_unmapped_flag = f.read_unmapped_flag; out->read_seq_off = f.read_seq_off;
    out->read_qual_off = f.read_qual_off_out; out->rev_seq = rs; out->rev_qual = rq;
    out->rev_seq_bytes = (uint64_t)o->soff[ne] * 16u; out->rev_qual_bytes = (uint64_t)o->qoff[ne] * 16u;
    out->n_items = n; out->n_reads = ne - n;
    return 0;
}

// the device comp_base (lift_core.hpp), for an exhaustive comparison with the oracle
extern "C" int emu_comp_base(int b) { return plo::comp_base(b); }

// crc32_wave (k_bgzf_crc's wave per BGZF block, inflate.hpp) under the emulator
extern "C" uint32_t emu_crc32_wave(const uint8_t *p, uint32_t n, unsigned order_seed) {
    uint32_t tab[256];
    for (uint32_t e = 0; e < 256; ++e) tab[e] = plo::crc32_table_entry(e);
    uint32_t res = 0;
    wv::EmuWave w;
    w.order_seed = order_seed;
    w.run([&]() {
        uint32_t c = plo::crc32_wave(p, n, tab);
        if (wv::lane() == 17) res = c;
    });
    return res;
}

// build_segment_map_wave (k_map_build's wave per contig segment) under the emulator, and the sequential build_segment_map beside it
extern "C" int emu_map_build(const uint32_t *cigar, uint32_t n, long long ref_pos, int wave, plo::KV *out, unsigned order_seed) {
    if (!wave) return plo::build_segment_map(cigar, n, ref_pos, out);
    int cnt = 0;
    wv::EmuWave w;
    w.order_seed = order_seed;
    w.run([&]() {
        int c = plo::build_segment_map_wave(cigar, n, ref_pos, out);
        if (wv::lane() == 0) cnt = c;
    });
    return cnt;
}


// SA tag segments: finish_core.hpp's sa_item_len / sa_item_emit executed on the host.  `lift` and the finish arrays are host
// copies; text/off are malloc'ed and released by emu_sa_free.
extern "C" int emu_sa_segments(const plo_batch_out *lift, const uint16_t *item_flag, const uint32_t *item_read, const uint32_t *read_n_lifted,
                               const uint32_t *chrom_name_off, const uint8_t *chrom_names, uint32_t **off_out, uint8_t **text_out) {
    DevWork wk;
    memset(&wk, 0, sizeof(wk));
    wk.n_items = lift->n_items;
    wk.status = (uint8_t *)lift->item_status;
    wk.chrom = (uint32_t *)lift->item_chrom_index;
    wk.pos = (int64_t *)lift->item_ref_pos;
    wk.mapq = (uint8_t *)lift->item_mapq;
    wk.cig_off = (uint64_t *)lift->item_cigar_off;
    wk.cig_len = (uint32_t *)lift->item_cigar_len;
    wk.out_cigar = (uint32_t *)lift->cigar;
    uint32_t n = lift->n_items;
    std::vector<uint32_t> len(n + 1, 0);
    uint32_t *off = (uint32_t *)malloc(((size_t)n + 1) * 4);
    DevSa sa;
    sa.chrom_name_off = chrom_name_off;
    sa.chrom_names = chrom_names;
    sa.item_flag = item_flag;
    sa.item_read = item_read;
    sa.read_n_lifted = read_n_lifted;
    sa.len = len.data();
    sa.off = off;
    sa.text = nullptr;
    for (uint32_t i = 0; i < n; ++i) len[i] = sa_item_len(wk, sa, i);
    off[0] = 0;
    for (uint32_t i = 0; i < n; ++i) off[i + 1] = off[i] + len[i];
    uint8_t *text = (uint8_t *)malloc(off[n] ? off[n] : 1);
    sa.text = text;
    for (uint32_t i = 0; i < n; ++i) sa_item_emit(wk, sa, i);
    *off_out = off;
    *text_out = text;
    return 0;
}
extern "C" void emu_sa_free(uint32_t *off, uint8_t *text) {
    free(off);
    free(text);
}

// the block-map searches of the descriptor code (enumerate.hpp) on one sorted key array: kv_upper_bound, kv_lower_bound,
// kv_lower_bound_near -- for the unit test against numpy.searchsorted
extern "C" void emu_kv_search(const int *keys, int n, int lo, int hi, int x, int near_from, int *out /*[3]*/) {
    std::vector<plo::KV> kv((size_t)(n > 0 ? n : 1));
    for (int i = 0; i < n; ++i) kv[(size_t)i] = plo::KV{keys[i], 0};
    out[0] = plo::kv_upper_bound(kv.data(), lo, hi, x);
    out[1] = plo::kv_lower_bound(kv.data(), lo, hi, x);
    out[2] = plo::kv_lower_bound_near(kv.data(), near_from, hi, x);
}

// the device's DEFLATE decoder (inflate.hpp) executed on the host: serially, and by the 64 lanes of an emulated wave
extern "C" int emu_inflate(const uint8_t *in, uint32_t in_len, uint8_t *out, uint32_t out_len, uint32_t *written) {
    plo::InfWork ws;
    plo::InfSerial io;
    return plo::inflate_block(io, in, in_len, out, out_len, ws, written);
}
namespace {
struct InfEmuWave {
    int lane() const { return wv::lane(); }
    void sync() const { wv::sync(); }
    uint32_t uniform(uint32_t v) const { return (uint32_t)wv::bcast_first((int)v); }
    uint32_t scalar(uint32_t v) const { return v; }
    uint32_t read_lane(uint32_t v, uint32_t l) const { return (uint32_t)wv::shfl((int)v, (int)(l & 63)); }
    int popcount64(unsigned long long v) const { return __builtin_popcountll(v); }
    uint32_t rank_below(unsigned long long mask) const { return (uint32_t)__builtin_popcountll(mask & ((1ull << wv::lane()) - 1ull)); }
    uint8_t load_written(const uint8_t *p) const { return *p; }
};
}  // namespace
extern "C" int emu_inflate_wave(const uint8_t *in, uint32_t in_len, uint8_t *out, uint32_t out_len, uint32_t *written, unsigned order_seed) {
    plo::InfWork ws;
    plo::InfWaveMem mem;  // the wave's LDS windows
    int rc_all[64];
    uint32_t w_all[64];
    wv::EmuWave w;
    w.order_seed = order_seed;
    w.run([&]() {
        uint32_t wr = 0;
        plo::InfWaveIO<InfEmuWave> io;  // per lane, like the registers of the device code
        io.m = &mem;
        int rc = plo::inflate_block(io, in, in_len, out, out_len, ws, &wr);
        rc_all[wv::lane()] = rc;
        w_all[wv::lane()] = wr;
    });
    for (int l = 1; l < 64; ++l)
        if (rc_all[l] != rc_all[0] || (rc_all[0] == 0 && w_all[l] != w_all[0])) return -1000;  // the lanes must agree
    *written = w_all[0];
    return rc_all[0];
}
