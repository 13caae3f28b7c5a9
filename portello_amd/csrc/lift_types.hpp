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
    const uint32_t *seg_read;
    const uint32_t *seg_contig;
    const int64_t *seg_pos;
    const uint8_t *seg_is_fwd;
    const uint32_t *seg_cigar_off;
    const uint32_t *cigar;
    uint32_t n_reads, n_segs;
};

enum { CNT_CIGAR = 0, CNT_OVERFLOW = 1, CNT_NBIG = 2, CNT_ALGO_BYTES = 3, CNT_IN_OPS = 4, CNT_ERROR = 5, CNT_N = 8 };

// The item work list and the per-item outputs (all device memory)
struct DevWork {
    uint32_t n_items;
    const uint32_t *item_seg;
    const uint32_t *item_cseg;
    const uint32_t *item_op_prefix;  // [n_items+1] exclusive prefix of the items' input op counts
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
    uint32_t *big_list;            // items re-queued for the large-item kernel
};

}  // namespace plo
