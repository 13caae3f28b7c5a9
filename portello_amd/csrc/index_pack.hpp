// index_pack.hpp -- host-side packing of the contig->reference index into the flat arrays the kernels read.
//
// Builds, per contig split segment, the block map of get_read_segment_to_ref_pos_tree_map
// (/root/reference/lib/rust-vc-utils/src/bam_utils/read_to_ref_map.rs:101-137, ignore_hard_clip = false as at
// src/contig_alignment_scanner/mod.rs:98-102) as a sorted {key,val} array: the device form of ReadToRefTreeMap.
// One-time O(#contig CIGAR ops) set-up, shared by the engine and by the wave-emulator test harness.
#pragma once
#include <stdint.h>

#include <string>
#include <vector>

#include "../../include/portello_liftover.h"
#include <plo_wave.hpp>

#include "enumerate.hpp"
#include "lift_types.hpp"

namespace plo {

struct PackedIndex {
    std::vector<KV> kv;
    std::vector<uint32_t> cs_kv_off, cs_chrom, contig_seg_off;
    std::vector<uint8_t> cs_is_fwd, cs_mapq;
    std::vector<int> cs_start, cs_end, contig_len, chrom_len;
};

inline bool fits31(int64_t v) { return v >= 0 && v <= 0x7ffffff0LL; }

inline plo_status pack_index(const plo_index_desc *d, PackedIndex &out, std::string &err, bool build_kv = true) {
    if (!d || (d->n_contigs && (!d->contig_len || !d->contig_seg_off)) ||
        (d->n_segments && (!d->seg_chrom_index || !d->seg_pos || !d->seg_is_fwd_strand || !d->seg_mapq || !d->seg_seq_order_start ||
                           !d->seg_seq_order_end || !d->seg_cigar_off || (!d->seg_cigar && d->seg_cigar_off[d->n_segments]))) ||
        (d->n_chroms && (!d->chrom_len || !d->chrom_seq))) {
        err = "plo_index_desc: NULL array";
        return PLO_ERR_INVALID_ARG;
    }
    out = PackedIndex();
    out.contig_len.resize(d->n_contigs);
    out.contig_seg_off.assign(d->n_contigs + 1, 0);
    for (uint32_t c = 0; c < d->n_contigs; ++c) {
        if (!fits31(d->contig_len[c])) {
            err = "contig length outside the 31-bit BAM range";
            return PLO_ERR_RANGE;
        }
        out.contig_len[c] = (int)d->contig_len[c];
    }
    for (uint32_t c = 0; c <= d->n_contigs; ++c) {
        out.contig_seg_off[c] = d->n_contigs ? d->contig_seg_off[c] : 0;
        if (c && out.contig_seg_off[c] < out.contig_seg_off[c - 1]) {
            err = "contig_seg_off not monotone";
            return PLO_ERR_INVALID_ARG;
        }
    }
    if (out.contig_seg_off[d->n_contigs] != d->n_segments) {
        err = "contig_seg_off[n_contigs] != n_segments";
        return PLO_ERR_INVALID_ARG;
    }
    out.chrom_len.resize(d->n_chroms);
    for (uint32_t c = 0; c < d->n_chroms; ++c) {
        if (!fits31(d->chrom_len[c])) {
            err = "chromosome length outside the 31-bit BAM range";
            return PLO_ERR_RANGE;
        }
        out.chrom_len[c] = (int)d->chrom_len[c];
    }
    uint32_t ns = d->n_segments;
    out.cs_kv_off.assign(ns + 1, 0);
    out.cs_chrom.resize(ns);
    out.cs_is_fwd.resize(ns);
    out.cs_mapq.resize(ns);
    out.cs_start.resize(ns);
    out.cs_end.resize(ns);
    for (uint32_t g = 0; g < ns; ++g) {
        if (d->seg_chrom_index[g] >= d->n_chroms) {
            err = "seg_chrom_index out of range";
            return PLO_ERR_INVALID_ARG;
        }
        if (!fits31(d->seg_pos[g]) || !fits31(d->seg_seq_order_start[g]) || !fits31(d->seg_seq_order_end[g])) {
            err = "segment coordinate outside the 31-bit BAM range";
            return PLO_ERR_RANGE;
        }
        out.cs_chrom[g] = d->seg_chrom_index[g];
        out.cs_is_fwd[g] = d->seg_is_fwd_strand[g] ? 1 : 0;
        out.cs_mapq[g] = d->seg_mapq[g];
        out.cs_start[g] = (int)d->seg_seq_order_start[g];
        out.cs_end[g] = (int)d->seg_seq_order_end[g];

        // get_read_segment_to_ref_pos_tree_map: the same build_segment_map the device kernels run (enumerate.hpp)
        uint32_t c0 = d->seg_cigar_off[g], c1 = d->seg_cigar_off[g + 1];
        if (c1 < c0) {
            err = "seg_cigar_off not monotone";
            return PLO_ERR_INVALID_ARG;
        }
        out.cs_kv_off[g] = (uint32_t)out.kv.size();
        if (build_kv) {
            int cnt = build_segment_map(d->seg_cigar + c0, c1 - c0, d->seg_pos[g], nullptr);
            if (cnt < 0) {
                err = "contig segment coordinate outside the 31-bit BAM range or invalid CIGAR op code";
                return PLO_ERR_RANGE;
            }
            size_t begin = out.kv.size();
            out.kv.resize(begin + (size_t)cnt);
            build_segment_map(d->seg_cigar + c0, c1 - c0, d->seg_pos[g], out.kv.data() + begin);
        }
    }
    out.cs_kv_off[ns] = (uint32_t)out.kv.size();
    return PLO_OK;
}

}  // namespace plo
