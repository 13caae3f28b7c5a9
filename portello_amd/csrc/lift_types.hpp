// lift_types.hpp -- plain data types shared by the host engine, the kernels and the test harness.
#pragma once
#include <stdint.h>

#include "../../include/portello_liftover.h"

namespace plo {

enum : int { OP_M = 0, OP_I = 1, OP_D = 2, OP_N = 3, OP_S = 4, OP_H = 5, OP_P = 6, OP_EQ = 7, OP_X = 8 };
constexpr int NONE32 = (int)0x80000000;  // Option::None for 32-bit positions / block values
constexpr int IMAX = 0x7fffffff;
constexpr int ITEM_NEED_BIG = 0xFF;  // internal: tile capacity exceeded, item re-queued for the large-item kernel
constexpr int MAXI = 64;             // items per tile = lanes per wave
// PLO_SEQ_BAM4_SPARSE: bytes of a read's granule header ({u32 mask, u32 rank} per 1024 bases, padded to 16 bytes)
constexpr uint32_t sparse_header_bytes(uint32_t seq_len) { return ((((seq_len + 1023u) >> 10) * 8u) + 15u) & ~15u; }

struct alignas(8) KV {
    int key;  // contig position starting a block          (ReadToRefTreeMap key, read_to_ref_map.rs:59-64)
    int val;  // reference position of the block or NONE32  (Option<i64>)
};

// Device-resident packed index (plo_index): everything phase 2 reads from AllContigMappingInfo + reference.
struct DevIndex {
    const KV *kv;                   // block maps of all contig segments, concatenated
    const uint32_t *cs_kv_off;      // [n_segments+1] CSR into kv
    const uint32_t *cs_chrom;       // [n_segments]
    const uint8_t *cs_is_fwd;       // [n_segments]
    const uint8_t *cs_mapq;         // [n_segments]
    const int *cs_start;            // [n_segments] seq_order_read_start
    const int *cs_end;              // [n_segments] seq_order_read_end
    const uint32_t *contig_seg_off; // [n_contigs+1]
    const int *contig_len;          // [n_contigs]
    const uint8_t *const *contig_revseq;  // [n_contigs] device pointers (NULL = none)
    const uint8_t *const *chrom_seq;      // [n_chroms] device pointers
    const int *chrom_len;           // [n_chroms]
    uint32_t n_contigs, n_segments, n_chroms;
};

struct DevBatch {
    const uint8_t *read_is_reverse;
    const uint32_t *read_seq_len;
    const uint64_t *read_seq_off;
    const uint8_t *seq;
    int seq_fmt;
    uint64_t seq_bytes;  // size of `seq`
    const uint32_t *seg_read;
    const uint32_t *seg_contig;
    const int64_t *seg_pos;
    const uint8_t *seg_is_fwd;
    const uint32_t *seg_cigar_off;
    const uint32_t *cigar;
    uint32_t n_reads, n_segs;
};

enum { CNT_CIGAR = 0, CNT_OVERFLOW = 1, CNT_NBIG = 2, CNT_ALGO_BYTES = 3, CNT_IN_OPS = 4, CNT_ERROR = 5, CNT_OUT_OPS = 6, CNT_NRETRY = 7, CNT_PHASE0 = 8, CNT_NHUGE = 20, CNT_NMISS = 21, CNT_LANE_ACT = 22, CNT_LANE_TRIPS = 23, CNT_N = 24 };

// Resolved per-item descriptors, written once per batch by the item kernels (thread per item, full occupancy) so that
// the tile kernel starts from ONE level of coalesced loads instead of chasing item -> segment -> contig -> block map.
struct ItemDesc {
    uint32_t *in_off;   // first input CIGAR op of the item's read segment
    uint32_t *n_in;     // number of input ops
    uint32_t *n_m;      // ... after merging neighbouring alignment-match ops (= X M): what the lane-per-item kernel walks (lane_core.hpp LOAD)
    int *pos1;          // start on the map strand: seg_pos, or rev_pos for reverse-mapped contig segments (STRAND)
    uint32_t *w0, *w1;  // window [w0, w1) of the contig segment's block map that can intersect the item
    uint32_t *kv0, *kv1;  // the contig segment's whole block map
    uint32_t *flags;    // bit0 reverse the CIGAR, bit1 need_flipped, bit2 contig segment maps forward
    uint32_t *contig;
    uint32_t *seq_len;
    uint64_t *seq_off;
    uint64_t *shift_ref;   // rev_contig_seq of the contig (device address, 0 = none): ref_seq of left_shift_indels
    int *shift_ref_len;    // contig length
    uint64_t *chrom_ref;   // reference[chrom_index]: ref_seq of simplify_alignment_indels
    int *chrom_ref_len;
    uint32_t *read_len;    // read bases the input CIGAR consumes (get_cigar_read_offset(cigar, false)), saturated at 2^32 - 1
};
enum { ITF_REV = 1, ITF_FLIP = 2, ITF_CONTIG_FWD = 4 };

// The item work list and the per-item outputs (all device memory)
struct DevWork {
    uint32_t n_items;
    uint32_t *item_seg;
    uint32_t *item_cseg;
    uint32_t *item_nin;              // input op count per item
    uint32_t *item_cls;              // bit0: the item goes through the left-shift stage (reverse-mapped contig segment),
                                     // bit1: too heavy for the lane-per-item path
    // Work is cut from the items in *class order* (class 0, 1, 2, 3, each in input order): the first n_small positions
    // (classes 0-1) go to the lane-per-item kernel in groups of 64, the rest to the tile kernel; groups and tiles are
    // strand-homogeneous so that forward ones skip the shift stage altogether.  Outputs keep the input order.
    const uint32_t *perm;            // [n_items] class order -> item index
    uint32_t n_small;                // positions [0, n_small) of perm: lane-per-item kernel
    int lane_max_w;                  // heaviest item (item_weight) routed to the lane-per-item kernel (lane_core.hpp); < 0: none
    const uint32_t *item_op_prefix;  // [n_items+1] exclusive prefix, in class order, of the op counts of the large items
    const uint32_t *tile_lo;         // [n_tiles+1] first class-order position (>= n_small) of every tile
    uint32_t *retry_list;            // items of the lane kernel whose intermediates overflowed: re-run by the tile code
    ItemDesc d;
    uint8_t *status;
    uint8_t *flip;
    uint8_t *mapq;
    uint32_t *chrom;
    int64_t *pos;
    uint64_t *cig_off;
    uint32_t *cig_len;
    uint32_t *out_cigar;
    uint64_t out_cap;
    unsigned long long *counters;  // [CNT_N]
    // output slabs: when slab_pre is set, wave w of the tile kernel owns ops [w*SLAB_OPS, (w+1)*SLAB_OPS) from the start (no
    // atomic for its first slab) and every later reservation is counters[CNT_CIGAR] + slab_offset
    unsigned long long slab_offset;
    uint32_t slab_pre;
    unsigned long long *wave_stats;  // [waves of the launch][4]: algorithmic bytes, input ops, output ops of every wave (summed by k_sum_stats)
    uint32_t *big_list;            // items too heavy for a shared tile: workgroup-per-item kernel (k_lift_mid)
    const uint32_t *seg_readlen;   // [n_segs] read bases consumed by every read segment's CIGAR (k_seg_count); NULL: computed per item
    const uint32_t *seg_nm;        // [n_segs] op count of every read segment's CIGAR with neighbouring match ops merged (k_seg_count); NULL: per item
    uint32_t *huge_list;           // items too heavy for that one too: one wave per item in global scratch (k_lift_big)
    uint32_t stat_base;            // first statistic slot of the launch (every lift launch of a batch has its own range)
    uint32_t *miss_list;           // PLO_SEQ_BAM4_SPARSE: items whose probes needed absent bases (PLO_ITEM_NEED_BASES)
    // lane-per-item kernel, groups cut by LDS budget (k_chunk_sort): the dwords of every light item's region (lane_region_dwords; written by
    // build_item_desc when non-NULL), the groups {first position of the class order, items} and their number (device memory; NULL: fixed groups)
    uint32_t *item_region;
    const uint32_t *lane_groups;    // [2 * n]: lo, count
    const uint32_t *lane_n_groups;  // [1]
    uint32_t lane_groups_cap;       // groups `lane_groups` has room for (more would be a sizing bug: reported through CNT_ERROR)
    // lane-per-item kernel, groups dealt dynamically (lane_tiles_persistent): the launch's ticket counter (zero at its start; NULL: fixed slots
    // only), the counter of the context's NEXT launch (zeroed by this one's first wave), the rounds every wave takes by fixed slots first
    uint32_t *lane_ticket;
    uint32_t *lane_ticket_next;
    uint32_t lane_static_rounds;
    uint32_t lane_tail_rounds;  // rounds' worth of groups without the shift stage kept for the end of the launch (0: the classes one to one, the longer one's rest last)
    uint32_t lane_kvs;  // block-map entries a wave of the light-item kernel stages in LDS (0: LANE_KVS; a multiple of 64 up to LANE_KVS_MAX)
};

}  // namespace plo
