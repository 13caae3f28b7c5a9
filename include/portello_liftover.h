/*
 * portello_liftover.h -- C ABI of the MI355X-native liftover engine.
 *
 * This is the drop-in boundary for portello's per-read CIGAR-composition hot path.  The reference has no FFI:
 * the path is three `pub fn`s called synchronously, one (read segment x contig segment) pair at a time, from
 *   get_liftover_alignment_for_read_and_contig_segment        src/read_alignment_scanner.rs:136-288
 * inside scan_chromosome_segment (src/read_alignment_scanner.rs:369-492).  A Rust caller binds the functions
 * below with `extern "C"`, keeps one `plo_ctx` per BamReaderWorkerThreadData
 * (src/worker_thread_data.rs:8-18) and turns the per-record loop into: collect window -> one batch call ->
 * finish records.
 *
 * Conventions
 *   - every function returns a plo_status (0 = ok) and never unwinds; plo_last_error() gives a message;
 *   - a plo_index is immutable after creation and may be shared by any number of contexts/threads;
 *   - a plo_ctx is single-threaded (one per worker thread), owns one HIP stream, its workspaces and its
 *     output buffers; output pointers stay valid until the next call on the same context;
 *   - one *item* = one call of get_liftover_alignment_for_read_and_contig_segment, i.e. one
 *     (read split segment x contig split segment) pair;
 *   - CIGAR ops use the BAM encoding `len << 4 | op`, op in M0 I1 D2 N3 S4 H5 P6 =7 X8 (rust-htslib `Cigar`);
 *   - all positions are 0-based; coordinates must fit the BAM 31-bit range (else PLO_ERR_RANGE);
 *   - there is NO CPU fallback: without a usable HIP device every entry point fails with PLO_ERR_NO_DEVICE.
 */
#ifndef PORTELLO_LIFTOVER_H
#define PORTELLO_LIFTOVER_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PLO_API_VERSION 6 /* 4: plo_timing starts with struct_size (the callee fills no more than the caller's struct holds); 5: plo_gather_*, plo_ctx_set_stats (plo_timing::algo_bytes / lane_utilisation of light items only on request), plo_ctx_stream / plo_ctx_device; 6: plo_ctx_set_phase_events */

typedef enum plo_status {
    PLO_OK = 0,
    PLO_ERR_INVALID_ARG = 1,
    PLO_ERR_NO_DEVICE = 2,
    PLO_ERR_HIP = 3,
    PLO_ERR_OUT_OF_MEMORY = 4,
    PLO_ERR_RANGE = 5,    /* coordinate outside the 31-bit BAM range or invalid CIGAR op code            */
    PLO_ERR_INTERNAL = 6, /* device-side capacity exceeded even in the large-item path                  */
    PLO_ERR_IO = 7,       /* portello_bam.h: file cannot be opened / read / written, truncated or corrupt BGZF   */
    PLO_ERR_DATA = 8      /* input the reference aborts on: an item ended LEN_MISMATCH / PANIC when records are
                             finished (src/read_alignment_scanner.rs:207-229), or a record whose SA tag it would
                             panic on (portello_bam.h)                                                  */
} plo_status;

/* Per-item result status.  The reference expresses these as Option::None / panic!:
 *   LIFTED        Some(record)                                   src/read_alignment_scanner.rs:185,284
 *   NO_LIFTOVER   liftover_read_alignment returned None           src/liftover_read_alignment.rs:218
 *   LEN_MISMATCH  read length of lifted CIGAR != seq_len; the reference aborts   src/read_alignment_scanner.rs:204-229
 *   PANIC         a sequence index would be out of bounds (Rust slice-index panic in
 *                 simplify_alignment_indels.rs:58-60,74-77 / indel_breakend_homology.rs:38-39)
 *   NEED_BASES    no counterpart in the reference: the batch came with PLO_SEQ_BAM4_SPARSE bases, a sequence comparison
 *                 of this item reached bases the batch does not carry, and no `seq_full` was given to look them up in.
 *                 The item has no result; lift it again from a batch that holds the read's complete bases.       */
enum {
    PLO_ITEM_LIFTED = 0,
    PLO_ITEM_NO_LIFTOVER = 1,
    PLO_ITEM_LEN_MISMATCH = 2,
    PLO_ITEM_PANIC = 3,
    PLO_ITEM_NEED_BASES = 4
};

/* Read-sequence encodings accepted at the boundary */
enum {
    PLO_SEQ_BAM4 = 0, /* BAM 4-bit packing, 2 bases per byte, high nibble first, code table "=ACMGRSVTWYHKDBN"
                         (what bam::Record::seq() holds; decoded by as_bytes() at read_alignment_scanner.rs:170,238) */
    PLO_SEQ_ASCII = 1, /* one byte per base */
    PLO_SEQ_BAM4_SPARSE = 2
    /* BAM 4-bit packing, but only the bases the kernels are likely to look at travel to the device (the sequence
       comparisons of left_shift_indels / simplify_alignment_indels touch a few dozen bases around each indel, the other
       ~97 % of a HiFi read never leave the host).  A read's bases are cut into granules of 32 bases (16 bytes).  At
       read_seq_off[r] (a multiple of 16): a header of ceil(seq_len / 1024) pairs {uint32 mask, uint32 rank} -- mask bit k of
       pair b set = granule 32 b + k is present, rank = number of present granules before granule 32 b -- padded to a
       multiple of 16 bytes, followed by the present granules in ascending order, each the 16 bytes of the dense BAM4
       encoding (the last one zero-padded).  plo_sparse_seq_pack (portello_bam.h) writes this form from records / dense
       bases and CIGARs.  A comparison that reaches an absent granule is detected on the device and the item is lifted
       again from the read's complete bases (plo_batch_in::seq_full), so results never depend on which granules were sent. */
};

/* Where the sequence / batch buffers of a descriptor live */
enum {
    PLO_MEM_HOST = 0,  /* host memory: the library copies to the device */
    PLO_MEM_DEVICE = 1 /* device memory on the index's device: borrowed, caller keeps it alive */
};

/* Pipeline stages (bit mask).  PLO_STAGES_ALL reproduces get_liftover_alignment_for_read_and_contig_segment;
 * subsets expose the individual reference functions so that they can be pinned against the reference's own
 * known-answer tests. */
enum {
    PLO_STAGE_STRAND = 1u << 0,   /* caller glue src/read_alignment_scanner.rs:149-176: need_flipped, and for
                                     reverse-mapped contig segments rev_pos + reversed CIGAR                     */
    PLO_STAGE_LSHIFT = 1u << 1,   /* left_shift_indels (lib/rust-vc-utils/.../shift_indels/left_shift_indels.rs:17-39);
                                     with STRAND: only items on reverse-mapped contig segments (as the reference),
                                     without STRAND: every item, CIGAR/pos used as given, ref_seq = the contig's
                                     rev_contig_seq                                                              */
    PLO_STAGE_LIFTOVER = 1u << 2, /* liftover_read_alignment  src/liftover_read_alignment.rs:137-223            */
    PLO_STAGE_LENCHECK = 1u << 3, /* seq_len == get_cigar_read_offset(cigar,false)  read_alignment_scanner.rs:204-229 */
    PLO_STAGE_SIMPLIFY = 1u << 4, /* simplify_alignment_indels src/simplify_alignment_indels.rs:119-156,
                                     ref_seq = reference[chrom_index of the contig segment]                      */
    PLO_STAGES_ALL = 0x1f
};

/* ------------------------------------------------------------------------------------------------------------
 * Index: the contig->reference mapping consumed by phase 2, i.e. AllContigMappingInfo
 * (src/contig_alignment_scanner/mod.rs:25-47,76) + the reference sequences (src/main.rs:24-62) + contig lengths
 * (ChromList of the read->contig BAM, src/read_alignment_scanner.rs:163-164).
 * plo_index_create builds, per contig segment, the block map of get_read_segment_to_ref_pos_tree_map
 * (lib/rust-vc-utils/src/bam_utils/read_to_ref_map.rs:101-137, ignore_hard_clip = false as at
 * contig_alignment_scanner/mod.rs:98-102) on the device and packs everything into HBM once.
 * ---------------------------------------------------------------------------------------------------------- */
typedef struct plo_index_desc {
    /* contigs (index = tid in the read->contig BAM) */
    uint32_t n_contigs;
    const int64_t *contig_len;      /* [n_contigs] */
    const uint32_t *contig_seg_off; /* [n_contigs+1] CSR into the segment arrays = ordered_contig_segment_info */

    /* contig split segments, sequencing order within each contig (SeqOrderSplitReadSegment,
       lib/rust-vc-utils/src/bam_utils/split_read.rs:15-32) */
    uint32_t n_segments;
    const uint32_t *seg_chrom_index;      /* reference chromosome the segment maps to            */
    const int64_t *seg_pos;               /* 0-based reference start of the segment alignment    */
    const uint8_t *seg_is_fwd_strand;     /* 1 = contig segment maps to the forward strand       */
    const uint8_t *seg_mapq;
    const int64_t *seg_seq_order_start;   /* seq_order_read_start (contig coordinates)           */
    const int64_t *seg_seq_order_end;     /* seq_order_read_end                                  */
    const uint32_t *seg_cigar_off;        /* [n_segments+1] CSR into seg_cigar                   */
    const uint32_t *seg_cigar;            /* contig->reference CIGARs, BAM-encoded ops           */

    /* reference chromosomes: `reference: &[Vec<u8>]` (upper-cased ASCII, src/main.rs:24-62) */
    uint32_t n_chroms;
    const int64_t *chrom_len;             /* [n_chroms] */
    const uint8_t *const *chrom_seq;      /* [n_chroms] pointers (see seq_mem) */

    /* ContigMappingInfo::rev_contig_seq (contig_alignment_scanner/mod.rs:113-125): ASCII, length contig_len,
       NULL where the contig has no reverse-mapped segment */
    const uint8_t *const *rev_contig_seq; /* [n_contigs] pointers or NULL array */

    int32_t seq_mem;                      /* PLO_MEM_HOST or PLO_MEM_DEVICE for chrom_seq / rev_contig_seq */
} plo_index_desc;

typedef struct plo_index plo_index;
typedef struct plo_ctx plo_ctx;

/* ------------------------------------------------------------------------------------------------------------
 * Batch input: a window of primary read records with their sequencing-order split segments
 * (get_seq_order_read_split_segments output, read_alignment_scanner.rs:421).
 * ---------------------------------------------------------------------------------------------------------- */
typedef struct plo_batch_in {
    /* reads = primary bam::Record's */
    uint32_t n_reads;
    const uint8_t *read_is_reverse;  /* [n_reads] record.is_reverse()                          */
    const uint32_t *read_seq_len;    /* [n_reads] record.seq_len()                             */
    const uint64_t *read_seq_off;    /* [n_reads] byte offset of the read's bases inside `seq` */
    const uint8_t *seq;              /* all read bases, encoding `seq_fmt`                      */
    uint64_t seq_bytes;              /* size of `seq` in bytes                                  */
    int32_t seq_fmt;                 /* PLO_SEQ_BAM4 / PLO_SEQ_ASCII / PLO_SEQ_BAM4_SPARSE      */

    /* read split segments (SeqOrderSplitReadSegment) */
    uint32_t n_segs;
    const uint32_t *seg_read;        /* [n_segs] index of the owning read                      */
    const uint32_t *seg_contig;      /* [n_segs] chrom_index = contig the segment is aligned to */
    const int64_t *seg_pos;          /* [n_segs] 0-based position on the forward contig         */
    const uint8_t *seg_is_fwd_strand;/* [n_segs]                                                */
    const uint32_t *seg_cigar_off;   /* [n_segs+1] CSR into `cigar`                             */
    const uint32_t *cigar;           /* read->contig CIGARs                                     */

    /* Optional explicit item list.  NULL: the engine enumerates items itself with the overlap rule of
       get_contig_split_segments_from_read_mapping (read_alignment_scanner.rs:80-103).  Non-NULL: item i pairs
       read segment item_seg[i] with contig segment item_cseg[i] (index *within the contig's segment list*). */
    uint32_t n_items;
    const uint32_t *item_seg;
    const uint32_t *item_cseg;

    /* PLO_SEQ_BAM4_SPARSE only, optional (NULL: items that reach absent bases end PLO_ITEM_NEED_BASES).  HOST memory in both
       entry points: the dense BAM4 bases of read r start at seq_full + read_seq_full_off[r] (e.g. inside the BAM record the
       read came from).  Read only for the reads of such items, after the first pass over the batch. */
    const uint8_t *seq_full;
    const uint64_t *read_seq_full_off; /* [n_reads] */
} plo_batch_in;

/* Batch output (SoA, one entry per item, ordered by (read segment, contig segment index)).
 * plo_liftover_batch: pointers are host (pinned) memory owned by the context.
 * plo_liftover_batch_dev: pointers are device memory owned by the context. */
typedef struct plo_batch_out {
    uint32_t n_items;
    const uint32_t *item_seg;          /* read segment index                                                  */
    const uint32_t *item_cseg;         /* contig_segment_index (as used in the PS tag, :257-262)              */
    const uint8_t *item_status;        /* PLO_ITEM_*                                                          */
    const uint8_t *item_need_flipped;  /* need_flipped_read_alignment (:153-157)                              */
    const uint8_t *item_mapq;          /* contig segment MAPQ adopted by the record (:250-252)                */
    const uint32_t *item_chrom_index;  /* tid of the lifted record (:232-247)                                 */
    const int64_t *item_ref_pos;       /* lifted 0-based position (valid when LIFTED/LEN_MISMATCH)            */
    const uint64_t *item_cigar_off;    /* offset of the item's CIGAR inside `cigar`                           */
    const uint32_t *item_cigar_len;    /* number of ops                                                       */
    const uint32_t *cigar;             /* lifted CIGARs                                                       */
    uint64_t n_cigar;                  /* extent of `cigar` in ops (items index it through item_cigar_off; the
                                          buffer is slab-allocated on the device and may contain unused gaps)  */
} plo_batch_out;

/* Per-call device timing measured with HIP events on the context's stream.
   `struct_size`: set by the CALLER to sizeof(plo_timing) of the header it was built against before plo_ctx_timing;
   the library writes at most that many bytes (fields are only ever appended), and stores the size it filled. */
typedef struct plo_timing {
    uint32_t struct_size;
    float total_ms;      /* first kernel start -> last kernel end                                   */
    float enumerate_ms;  /* item enumeration + scans                                                */
    float lift_ms;       /* the wave-cooperative tile kernels of the scan formulation (k_lift_tiles*: light items of batches the lane
                            kernel does not take)                                                      */
    float big_ms;        /* one-wave-per-item kernel in global scratch (0 if not launched)          */
    uint32_t n_items;
    uint32_t n_big_items; /* items of that kernel */
    uint64_t n_in_ops;   /* input CIGAR ops over all items                                          */
    uint64_t n_out_ops;  /* output CIGAR ops                                                        */
    uint64_t algo_bytes; /* algorithmic bytes of the call, SURVEY.md 8(d) formula, counted on device.  The light-item kernel
                            (k_lift_lanes) counts them only on a context with plo_ctx_set_stats(ctx, 1): its production
                            instantiation is compiled without the counters (0 from its items otherwise)              */
    float lanes_ms;      /* the lane-per-item kernel (k_lift_lanes: items whose working region fits an LDS share)  */
    float retry_ms;      /* items of tiles that overflowed their LDS slice, re-run one per wave       */
    uint32_t n_lane_items;
    uint32_t n_retry_items;
    float mid_ms;        /* workgroup-per-item kernel (items too heavy for a shared tile)             */
    uint32_t n_mid_items;
    uint32_t n_miss_items; /* PLO_SEQ_BAM4_SPARSE: items that reached absent bases (lifted again from seq_full) */
    float miss_ms;         /* that second pass: list download, host gather, upload, kernel (wall clock)         */
    uint32_t tile_cap;     /* geometry of the tile kernel for this batch: elements per LDS slice (256: the variant with the
                              capacity compiled in, k_lift_tiles_c256) and window of item weights per tile        */
    uint32_t tile_window;
    float heavy_lanes_ms;        /* the heavy items' lane-per-item kernel, `heavy_kernel` says which (k_lift_lanes_g / _w3: regions in
                                    global scratch behind LDS windows; k_lift_stream: teams of waves); lift_ms / mid_ms are 0 then */
    uint32_t n_heavy_lane_items;
    float lane_utilisation;      /* lane-per-item kernels: lanes at work / (64 x loop trips), summed over the liftover loop and the
                                    shift stage's event rounds of all waves (0 when no such kernel ran; the light-item kernel
                                    counts only under plo_ctx_set_stats, as for algo_bytes)                                   */
    uint32_t heavy_kernel;       /* which kernel `heavy_lanes_ms` is the time of: 0 none, 1 k_lift_lanes_g, 2 k_lift_lanes_g_w3,
                                    3 k_lift_stream (teams of waves, stages chained through LDS rings)                       */
    uint32_t host_syncs;         /* host round trips of the call (stream synchronisations that returned counts to the host)  */
} plo_timing;

plo_status plo_index_create(const plo_index_desc *desc, int device, plo_index **out);
void plo_index_destroy(plo_index *index);
/* number of block-map entries of contig segment `global_seg` (= contig_seg_off[contig] + i), and a copy of them
   (keys = contig positions, vals = reference positions, INT64_MIN for None), downloaded from the device, for
   inspection/tests */
plo_status plo_index_segment_map(const plo_index *index, uint32_t global_seg, uint32_t cap, int64_t *keys,
                                 int64_t *vals, uint32_t *n_entries);

/* `hip_stream`: a hipStream_t to run on (e.g. torch's current stream) or NULL to create a private one */
plo_status plo_ctx_create(const plo_index *index, void *hip_stream, plo_ctx **out);
void plo_ctx_destroy(plo_ctx *ctx);
/* on != 0: the light-item kernel of this context's later calls counts plo_timing::algo_bytes and lane_utilisation (an instantiation with
   the counters in its loops, a few per cent slower); default: off, or what the environment's PLO_LANE_STATS says at plo_ctx_create */
plo_status plo_ctx_set_stats(plo_ctx *ctx, int on);
/* on == 0: the one-round-trip calls of this context (plo_liftover_batch_dev on a context whose last batch had light items only) record no
   HIP events between their phases -- every record is a bubble of ~6 us on the stream, 5 % of a reference-sized window's call -- and
   plo_ctx_timing then reports the counts of such a call but no times (enumerate_ms, lanes_ms, total_ms = 0).  A production caller that
   never asks for the times switches them off; default: on (or what the environment's PLO_PHASE_EVENTS says at plo_ctx_create). */
plo_status plo_ctx_set_phase_events(plo_ctx *ctx, int on);

/* Host buffers in, host (pinned, context-owned) buffers out; synchronous. */
plo_status plo_liftover_batch(plo_ctx *ctx, const plo_batch_in *in, uint32_t stages, plo_batch_out *out);
/* Device buffers in, device (context-owned) buffers out; returns after the result sizes are known, output is
   complete on the context's stream (call plo_ctx_sync or synchronise the stream before reading it).
   The batch is checked on the device before any lift kernel runs: an index outside its array -> PLO_ERR_INVALID_ARG;
   a coordinate outside the 31-bit BAM range, an op code above 8, a CIGAR spanning more than 2^30 bases or more than
   2^31 - 1 ops / item weights in the batch -> PLO_ERR_RANGE. */
plo_status plo_liftover_batch_dev(plo_ctx *ctx, const plo_batch_in *in, uint32_t stages, plo_batch_out *out);

/* ------------------------------------------------------------------------------------------------------------
 * Record finishing (the part of src/read_alignment_scanner.rs:245-284 and :310-346 that is arithmetic on the
 * record, not htslib bookkeeping): for the result of the LAST plo_liftover_batch_dev call on this context
 *   - per lifted item: flags (BAM_FREVERSE toggled when need_flipped :274-276,:126; supplementary set :282, cleared
 *     on the primary :346), reference end (get_alignment_end, lib/rust-vc-utils/src/bam_utils/bam_record_utils.rs:21-27)
 *     and bin (bam_reg2bin, lib/rust-vc-utils/src/bam_utils/util.rs:10-35, :278-279);
 *   - per read: number of lifted records, the primary one (max MAPQ, first wins :338-346), and for reads without any
 *     lifted record the flags of the unmapped copy (:317-335);
 *   - reverse_alignment_seq_and_qual (:125-133) for every record that needs it (lifted items with need_flipped,
 *     unmapped copies of reverse-strand reads): 4-bit reverse complement (decode, comp_base, re-encode) + reversed
 *     qualities, written into two context-owned buffers.
 * Requires seg_read to be non-decreasing (segments grouped by read, as get_seq_order_read_split_segments yields them).
 * All pointers are device pointers.
 * ---------------------------------------------------------------------------------------------------------- */
typedef struct plo_finish_in {
    const uint16_t *read_flags;    /* [n_reads] record.flags() of the primary read->contig records   */
    const uint8_t *qual;           /* base qualities of all reads (1 byte per base)                  */
    const uint64_t *read_qual_off; /* [n_reads] byte offset of the read's qualities inside `qual`     */
    uint64_t qual_bytes;           /* size of `qual`                                                  */
} plo_finish_in;

#define PLO_NO_FLIP UINT64_MAX

typedef struct plo_finish_out {
    /* per item (same order as plo_batch_out) */
    const uint16_t *item_flag;     /* flags of the lifted record (valid for LIFTED items)             */
    const uint16_t *item_bin;
    const int64_t *item_ref_end;
    const uint8_t *item_is_primary;
    const uint64_t *item_seq_off;  /* byte offset of the flipped bases inside rev_seq, PLO_NO_FLIP if the record keeps
                                      the read's original seq/qual                                    */
    const uint64_t *item_qual_off; /* same for rev_qual                                                */
    /* per read */
    const uint32_t *read_n_lifted;
    const uint32_t *read_primary_item; /* item index of the primary record (UINT32_MAX if none)       */
    const uint16_t *read_unmapped_flag; /* flags of the unmapped copy (valid when read_n_lifted == 0)  */
    const uint64_t *read_seq_off;  /* flipped bases of the unmapped copy, PLO_NO_FLIP if not needed    */
    const uint64_t *read_qual_off;
    /* flipped sequences */
    const uint8_t *rev_seq;        /* same encoding as the batch's seq_fmt, every record 16-byte aligned */
    const uint8_t *rev_qual;
    uint64_t rev_seq_bytes, rev_qual_bytes;
    float finish_ms, revcomp_ms;   /* HIP-event times of the two kernels groups                        */
    uint32_t n_items, n_reads;     /* extents of the per-item / per-read arrays: the batch they were made for (API version 4;
                                      plo_records_build_finished refuses arrays of another batch before indexing them) */
} plo_finish_out;

plo_status plo_finish_batch_dev(plo_ctx *ctx, const plo_batch_in *in, const plo_finish_in *fin, plo_finish_out *out);

/* ---- SA tag text (device-resident) ----------------------------------------------------------------------------
 * Consumes the results of the preceding plo_liftover_batch_dev + plo_finish_batch_dev on the same context and writes,
 * for every LIFTED item of a read that has at least two lifted records, the text of get_sa_tag_segment
 * (src/read_alignment_scanner.rs:292-301): "{chrom},{pos+1},{strand},{cigar},{mapq},0;".  The SA:Z value of record j is
 * the concatenation of the segments of the read's other lifted items in item order (:352-364, done by the caller:
 * items of a read are consecutive).  Chromosome labels are ChromList labels (the reference BAM header's @SQ names).
 * All pointers are device pointers; outputs are owned by the context and valid until its next call. */
typedef struct plo_sa_in {
    uint32_t n_chroms;
    const uint32_t *chrom_name_off; /* [n_chroms + 1] byte offsets into chrom_names                    */
    const uint8_t *chrom_names;     /* concatenated labels, no terminators                             */
} plo_sa_in;

typedef struct plo_sa_out {
    uint32_t n_items;
    const uint32_t *item_sa_off;    /* [n_items + 1] byte offsets into sa_text; empty range = no segment */
    const uint8_t *sa_text;
    uint64_t sa_bytes;
    float sa_ms;
} plo_sa_out;

plo_status plo_sa_segments_dev(plo_ctx *ctx, const plo_sa_in *in, plo_sa_out *out);

/* (plo_finish_batch_dev returns PLO_ERR_DATA when an item of the batch ended LEN_MISMATCH or PANIC -- the reference aborts
   there, :207-229 -- and leaves is_target_region handling (:318-320: no unmapped copy) to the caller.)

   Packs the output CIGARs of the last plo_liftover_batch_dev result densely (items in order, no gaps): the kernels
   allocate them in per-wave slabs, so plo_batch_out.cigar spans up to 16 Ki unused ops per resident wave.  Rewrites
   item_cigar_off, cigar and n_cigar of `out` (and what plo_finish_batch_dev / plo_sa_segments_dev will read).  Worth it
   before the arrays leave the device (plo_liftover_batch does it itself before its device-to-host copy). */
/* Like the kernels of plo_liftover_batch_dev it completes on the context's stream, not at return: order consumers on that
   stream (or call plo_ctx_sync). */
plo_status plo_compact_output_dev(plo_ctx *ctx, plo_batch_out *out);

/* ------------------------------------------------------------------------------------------------------------
 * The record gather of several GPUs (one process per GPU; SURVEY.md 8(e), INTEGRATION.md section 6 route (b)): the reference's sink is
 * one locked writer behind all workers (src/read_alignment_scanner.rs:24, :483) -- here rank `root` receives every rank's result
 * arrays over RCCL: a 16-byte size all-gather, then ONE group of ncclSend (peers) / ncclRecv (root) per array, device memory to device
 * memory (xGMI is point to point: every peer's link runs into the root at once).  RCCL is bound by name at the first call.
 *   rank 0:      plo_gather_unique_id(id); hand `id` to the other ranks (a file, a socket, MPI, torch.distributed ...)
 *   every rank:  plo_gather_create(id, rank, world, device, &g);
 *   per batch:   plo_liftover_batch_dev(ctx, ..., &out); plo_compact_output_dev(ctx, &out);
 *                plo_gather_records(g, ctx, &out, root, gathered = an array of `world` structs on root, NULL elsewhere); ... plo_gather_wait(g);
 * plo_gather_records returns when the exchange is posted on the context's stream (behind the compaction); the sizes in `gathered` are
 * valid at once, the arrays after plo_gather_wait (or any synchronisation of that stream).  gathered[root] points at the rank's own
 * arrays, the others at buffers the gather object owns until its next call.  A context must not start its next batch before the
 * exchange that reads its buffers is through (two contexts taking turns hide the exchange under the next batch's kernels).
 * ---------------------------------------------------------------------------------------------------------- */
#define PLO_GATHER_ID_BYTES 128 /* = NCCL_UNIQUE_ID_BYTES */
typedef struct plo_gather plo_gather;
plo_status plo_gather_unique_id(uint8_t id[PLO_GATHER_ID_BYTES]);
plo_status plo_gather_create(const uint8_t id[PLO_GATHER_ID_BYTES], int rank, int world, int device, plo_gather **out);
void plo_gather_destroy(plo_gather *g);
plo_status plo_gather_records(plo_gather *g, plo_ctx *ctx, const plo_batch_out *out, int root, plo_batch_out *gathered);
plo_status plo_gather_wait(plo_gather *g);
const char *plo_gather_last_error(const plo_gather *g); /* g == NULL: the calling thread's last plo_gather_unique_id / _create failure */
/* the HIP stream and device ordinal a context runs on (what a caller orders its own work behind / selects before its own HIP calls) */
void *plo_ctx_stream(plo_ctx *ctx);
int plo_ctx_device(plo_ctx *ctx);

/* Page-locked host memory for the arrays handed to plo_liftover_batch: copies from such buffers are direct DMA transfers
   (measured on MI355X, chr20 batch of 50 k reads / 385 MB: pageable 14 GB/s, page-locked see DESIGN.md).  The caller fills
   them in place (e.g. one set per worker thread, reused from batch to batch) and releases them with plo_host_free. */
plo_status plo_host_alloc(size_t bytes, void **out);
void plo_host_free(void *p);

plo_status plo_ctx_sync(plo_ctx *ctx);
/* Copies `bytes` from device memory (e.g. a plo_liftover_batch_dev output array) to host memory on the context's
   stream and waits for it. */
plo_status plo_ctx_download(plo_ctx *ctx, void *host_dst, const void *dev_src, size_t bytes);
plo_status plo_ctx_timing(plo_ctx *ctx, plo_timing *out);
const char *plo_last_error(const plo_ctx *ctx);
const char *plo_version(void);
/* PLO_API_VERSION the library was built with */
uint32_t plo_api_version(void);
/* Device self-test of the wavefront primitives (DPP scans, cross-lane reads): 0 = ok. */
int plo_selftest(int device);

#ifdef __cplusplus
}
#endif
#endif /* PORTELLO_LIFTOVER_H */
