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
#include "lift_types.hpp"

namespace plo {

struct PackedIndex {
    std::vector<KV> kv;
    std::vector<uint32_t> cs_kv_off, cs_chrom, contig_seg_off;
    std::vector<uint8_t> cs_is_fwd, cs_mapq;
    std::vector<int> cs_start, cs_end, contig_len, chrom_len;
};

inline bool fits31(int64_t v) { return v >= 0 && v <= 0x7ffffff0LL; }

// BTreeMap::insert on a sorted array: overwrite on equal key (read_to_ref_map.rs:105-112)
inline void kv_insert(std::vector<KV> &m, size_t begin, int key, int val) {
    size_t lo = begin, hi = m.size();
    while (lo < hi) {
        size_t mid = (lo + hi) / 2;
        if (m[mid].key < key)
            lo = mid + 1;
        else
            hi = mid;
    }
    if (lo < m.size() && m[lo].key == key) {
        m[lo].val = val;
        return;
    }
    KV e;
    e.key = key;
    e.val = val;
    m.insert(m.begin() + (long)lo, e);
}

inline plo_status pack_index(const plo_index_desc *d, PackedIndex &out, std::string &err) {
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

        // get_read_segment_to_ref_pos_tree_map (read_to_ref_map.rs:101-137)
        size_t begin = out.kv.size();
        out.cs_kv_off[g] = (uint32_t)begin;
        int64_t ref_pos = d->seg_pos[g];
        int64_t read_pos = 0, match_len = 0;
        uint32_t c0 = d->seg_cigar_off[g], c1 = d->seg_cigar_off[g + 1];
        if (c1 < c0) {
            err = "seg_cigar_off not monotone";
            return PLO_ERR_INVALID_ARG;
        }
        bool range_err = false;
        auto update_map = [&]() {
            if (match_len > 0) {
                if (!fits31(read_pos) || !fits31(ref_pos)) {
                    range_err = true;
                } else {
                    kv_insert(out.kv, begin, (int)(read_pos - match_len), (int)(ref_pos - match_len));
                    kv_insert(out.kv, begin, (int)read_pos, NONE32);
                }
                match_len = 0;
            }
        };
        for (uint32_t i = c0; i < c1; ++i) {
            uint32_t c = d->seg_cigar[i];
            int t = (int)(c & 15u);
            int64_t len = (int64_t)(c >> 4);
            if (t > 8) {
                err = "invalid CIGAR op code in contig segment";
                return PLO_ERR_RANGE;
            }
            bool is_m = (t == OP_M || t == OP_EQ || t == OP_X);
            if (is_m)
                match_len += len;
            else
                update_map();
            if ((0x1B3 >> t) & 1) read_pos += len;  // M I S H = X (ignore_hard_clip = false)
            if ((0x18D >> t) & 1) ref_pos += len;   // M D N = X
        }
        update_map();
        if (range_err) {
            err = "contig segment coordinate outside the 31-bit BAM range";
            return PLO_ERR_RANGE;
        }
    }
    out.cs_kv_off[ns] = (uint32_t)out.kv.size();
    return PLO_OK;
}

}  // namespace plo
