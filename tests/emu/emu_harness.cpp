// tests/emu/emu_harness.cpp -- runs the DEVICE algorithm (portello_amd/csrc/lift_core.hpp) on the CPU under the
// wave64 emulator (tests/emu/plo_wave.hpp).  TEST INFRASTRUCTURE ONLY: never linked into the product library.
#include <plo_wave.hpp>

#include <string>
#include <vector>

#include "../../portello_amd/csrc/enumerate.hpp"
#include "../../portello_amd/csrc/index_pack.hpp"
#include "../../portello_amd/csrc/lift_core.hpp"

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
                                  int big_thresh, int big_cap, unsigned order_seed, plo_batch_out *out,
                                  unsigned long long *counters_out) {
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
    bt.seg_read = in->seg_read;
    bt.seg_contig = in->seg_contig;
    bt.seg_pos = in->seg_pos;
    bt.seg_is_fwd = in->seg_is_fwd_strand;
    bt.seg_cigar_off = in->seg_cigar_off;
    bt.cigar = in->cigar;
    bt.n_reads = in->n_reads;
    bt.n_segs = in->n_segs;

    Out *o = new Out();
    std::vector<uint32_t> item_nin;
    if (in->item_seg) {
        o->item_seg.assign(in->item_seg, in->item_seg + in->n_items);
        o->item_cseg.assign(in->item_cseg, in->item_cseg + in->n_items);
        for (uint32_t i = 0; i < in->n_items; ++i)
            item_nin.push_back(in->seg_cigar_off[in->item_seg[i] + 1] - in->seg_cigar_off[in->item_seg[i]]);
    } else {
        for (uint32_t s = 0; s < in->n_segs; ++s) {
            uint32_t n = enumerate_segment(ix, bt, s, nullptr, nullptr, nullptr, 0);
            size_t base = o->item_seg.size();
            o->item_seg.resize(base + n);
            o->item_cseg.resize(base + n);
            item_nin.resize(base + n);
            enumerate_segment(ix, bt, s, o->item_seg.data(), o->item_cseg.data(), item_nin.data(), (uint32_t)base);
        }
    }
    uint32_t n_items = (uint32_t)o->item_seg.size();
    std::vector<uint32_t> prefix(n_items + 1, 0);
    for (uint32_t i = 0; i < n_items; ++i) prefix[i + 1] = prefix[i] + item_nin[i];
    uint64_t total_ops = prefix[n_items];

    size_t a = n_items ? n_items : 1;
    o->status.assign(a, 0xEE);
    o->flip.assign(a, 0);
    o->mapq.assign(a, 0);
    o->chrom.assign(a, 0);
    o->pos.assign(a, 0);
    o->cig_off.assign(a, 0);
    o->cig_len.assign(a, 0);
    uint64_t out_cap = 4 * total_ops + 64ull * n_items + 1024;
    for (uint32_t g = 0; g < ixd->n_segments; ++g) out_cap += 0;  // (blocks add pieces; bounded below by retry)
    std::vector<uint32_t> big_list(a, 0);
    unsigned long long counters[CNT_N];

    for (int attempt = 0; attempt < 6; ++attempt) {
        o->cigar.assign(out_cap, 0);
        memset(counters, 0, sizeof(counters));
        DevWork wk;
        wk.n_items = n_items;
        wk.item_seg = o->item_seg.data();
        wk.item_cseg = o->item_cseg.data();
        wk.item_op_prefix = prefix.data();
        wk.status = o->status.data();
        wk.flip = o->flip.data();
        wk.mapq = o->mapq.data();
        wk.chrom = o->chrom.data();
        wk.pos = o->pos.data();
        wk.cig_off = o->cig_off.data();
        wk.cig_len = o->cig_len.data();
        wk.out_cigar = o->cigar.data();
        wk.out_cap = out_cap;
        wk.counters = counters;
        wk.big_list = big_list.data();

        uint32_t n_tiles = (uint32_t)(total_ops / (uint64_t)window) + 1;
        std::vector<unsigned char> lds(tile_mem_bytes(cap) + 64);
        for (uint32_t t = 0; t < n_tiles && n_items; ++t) {
            wv::EmuWave w;
            w.order_seed = order_seed ? order_seed + t : 0;
            TileMem m = carve_tile_mem(lds.data(), cap);
            w.run([&]() { lift_window(ix, bt, wk, stages, t, window, big_thresh, m); });
        }
        uint32_t n_big = (uint32_t)counters[CNT_NBIG];
        if (n_big) {
            std::vector<unsigned char> scratch(tile_mem_bytes(big_cap) + 64);
            for (uint32_t i = 0; i < n_big; ++i) {
                wv::EmuWave w;
                w.order_seed = order_seed ? order_seed + 7777 + i : 0;
                TileMem m = carve_tile_mem(scratch.data(), big_cap);
                w.run([&]() { lift_tile(ix, bt, wk, stages, i, 1, m, true, 0); });
            }
        }
        if (counters[CNT_OVERFLOW] == 0) break;
        out_cap = counters[CNT_CIGAR] + 1024;
    }
    if (counters_out) memcpy(counters_out, counters, sizeof(counters));

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
    // the Out object is leaked into the caller's hands; emu_free releases it
    static_assert(sizeof(void *) == 8, "");
    ((void **)&out->n_cigar)[0] = ((void **)&out->n_cigar)[0];
    extern void *g_last_emu_out;
    g_last_emu_out = o;
    return counters[CNT_ERROR] ? 2 : 0;
}

void *g_last_emu_out = nullptr;

extern "C" void emu_free_last() {
    delete (Out *)g_last_emu_out;
    g_last_emu_out = nullptr;
}
