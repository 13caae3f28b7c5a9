// enumerate.hpp -- item enumeration: which contig split segments does a read split segment touch?
// Restates get_contig_split_segments_from_read_mapping (/root/reference/src/read_alignment_scanner.rs:80-103) with
// IntRange::intersect_range (lib/rust-vc-utils/src/int_range.rs:56-58; note the `>=`: left adjacency counts).
#pragma once
#include <plo_wave.hpp>

#include "lift_types.hpp"

namespace plo {

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

// Counts (and, when item_seg != nullptr, writes at out_off) the items of one read segment, in contig-segment order.
PLO_DEV uint32_t enumerate_segment(const DevIndex &ix, const DevBatch &bt, uint32_t seg, uint32_t *item_seg,
                                   uint32_t *item_cseg, uint32_t *item_nin, uint32_t out_off) {
    uint32_t contig = bt.seg_contig[seg];
    if (contig >= ix.n_contigs) return 0;
    uint32_t g0 = ix.contig_seg_off[contig], g1 = ix.contig_seg_off[contig + 1];
    if (g0 == g1) return 0;  // contig never seen in the asm->ref BAM (contig_alignment_scanner/mod.rs:364-367)
    long long r_start = (long long)bt.seg_pos[seg];
    long long r_end = r_start + segment_ref_len(bt, seg);
    uint32_t n = 0;
    for (uint32_t g = g0; g < g1; ++g) {
        // segment_range.intersect_range(&read_range): other.end >= self.start && other.start < self.end
        if (r_end >= (long long)ix.cs_start[g] && r_start < (long long)ix.cs_end[g]) {
            if (item_seg) {
                item_seg[out_off + n] = seg;
                item_cseg[out_off + n] = g - g0;
                item_nin[out_off + n] = bt.seg_cigar_off[seg + 1] - bt.seg_cigar_off[seg];
            }
            ++n;
        }
    }
    return n;
}

}  // namespace plo
