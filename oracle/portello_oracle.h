/*
 * portello_oracle.h -- CPU restatement of portello's liftover hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library; the product
 * (portello_amd/, libportello_liftover.so) never links, imports or calls it.
 *
 * Every function restates, line by line, a function of the reference (paths relative to /root/reference).
 * Parity pinning: the restatement is checked against every known-answer vector of the reference's own unit
 * tests for this path (tests/golden/reference_vectors.json, transcribed from the files cited there).  The Rust
 * reference itself cannot be built in this image (no cargo/rustc, no htslib), so there is no oracle/_ref.
 * The caller glue (orc_item: strand flip, rev_pos, segment selection) has no reference test: for that part
 * "parity unpinned" -- it is a literal restatement only.
 */
#ifndef PORTELLO_ORACLE_H
#define PORTELLO_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#include "../include/portello_liftover.h"

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_NONE INT64_MIN /* Option::None for map values */
#define ORC_PANIC (-2)     /* the reference would panic (slice index out of bounds) */

/* lib/rust-vc-utils/src/bam_utils/cigar/mod.rs */
int orc_is_alignment_match(uint32_t c);                                   /* :22-24  */
uint64_t orc_cigarseg_read_offset(uint32_t c, int ignore_hard_clip);      /* :26-39  */
int64_t orc_cigarseg_ref_offset(uint32_t c);                              /* :41-47  */
uint64_t orc_cigar_read_offset(const uint32_t *cig, size_t n, int ihc);   /* :164-170 */
int64_t orc_cigar_ref_offset(const uint32_t *cig, size_t n);              /* :174-180 */
void orc_read_clip_positions(const uint32_t *cig, size_t n, int ihc, uint64_t out3[3]); /* :85-118 */
size_t orc_compress_cigar(const uint32_t *in, size_t n, uint32_t *out);   /* :204-228 */
uint64_t orc_clean_up_cigar_edge_indels(uint32_t *cig, size_t n);         /* :265-291 */

/* lib/rust-vc-utils/src/seq_util.rs:1-40 */
uint8_t orc_comp_base(uint8_t b);
void orc_rev_comp_in_place(uint8_t *dna, size_t len);
/* bam::record::Seq::as_bytes(): 4-bit -> "=ACMGRSVTWYHKDBN" */
void orc_decode_bam4(const uint8_t *packed, size_t n_bases, uint8_t *out);

/* lib/rust-vc-utils/src/indel_breakend_homology.rs:24-73; returns 0 or ORC_PANIC */
int orc_indel_breakend_homology(const uint8_t *ref_seq, int64_t ref_len, int64_t ref_start, int64_t ref_end,
                                const uint8_t *read_seq, int64_t read_len, int64_t read_start, int64_t read_end,
                                int64_t *hom_start, int64_t *hom_end);

/* lib/rust-vc-utils/src/bam_utils/cigar/shift_indels/{cigar_indel_shifter,left_shift_indels,right_shift_indels}.rs
   dir: 0 = left, 1 = right.  out must hold 2*n+2 ops.  returns 0 or ORC_PANIC */
int orc_shift_indels(int dir, int64_t ref_pos, const uint32_t *cig, size_t n, const uint8_t *ref_seq,
                     int64_t ref_len, const uint8_t *read_seq, int64_t read_len, int64_t *out_pos, uint32_t *out,
                     size_t *n_out);

/* lib/rust-vc-utils/src/bam_utils/read_to_ref_map.rs:59-137.  keys/vals must hold 2*n+2 entries */
size_t orc_map_build(int64_t ref_pos, const uint32_t *cig, size_t n, int ignore_hard_clip, uint64_t *keys,
                     int64_t *vals);
int64_t orc_map_get_ref_pos(const uint64_t *keys, const int64_t *vals, size_t nk, uint64_t read_pos);     /* :66-72 */
void orc_map_get_ref_range(const uint64_t *keys, size_t nk, uint64_t a, uint64_t b, size_t *i0, size_t *i1); /* :74-85 */

/* src/liftover_read_alignment.rs:137-223.  returns 1 = Some, 0 = None.  out must hold 2*(n+nk)+2 ops */
int orc_liftover_read_alignment(const uint64_t *keys, const int64_t *vals, size_t nk, int64_t start,
                                const uint32_t *cig, size_t n, int64_t *out_pos, uint32_t *out, size_t *n_out);

/* src/simplify_alignment_indels.rs:119-156.  out must hold 2*n+2 ops; returns 0 or ORC_PANIC */
int orc_simplify_alignment_indels(int64_t ref_pos, const uint32_t *cig, size_t n, const uint8_t *ref_seq,
                                  int64_t ref_len, const uint8_t *read_seq, int64_t read_len, int64_t *out_pos,
                                  uint32_t *out, size_t *n_out);

/* Whole-batch restatement over the product's ABI structs: item enumeration
 * (src/read_alignment_scanner.rs:80-103) + get_liftover_alignment_for_read_and_contig_segment (:136-288, the
 * (pos, cigar, status) part).  Output arrays are malloc'ed into *out (free with orc_batch_free).  All buffers
 * are host memory.  n_threads > 1 splits the read segments over a pthread pool (cpu_baseline leg). */
int orc_liftover_batch(const plo_index_desc *index, const plo_batch_in *in, uint32_t stages, int n_threads,
                       plo_batch_out *out);
void orc_batch_free(plo_batch_out *out);

/* lib/rust-vc-utils/src/bam_utils/util.rs:10-35 */
uint16_t orc_bam_reg2bin(uint64_t begin, uint64_t end);
/* Record finishing restated from src/read_alignment_scanner.rs:125-133 (reverse_alignment_seq_and_qual), :245-284
 * (flags / end / bin of a lifted record), :310-346 (primary selection, unmapped copy).  Host pointers; `lift` is the
 * result of orc_liftover_batch (or a host copy of the engine's result) for the same batch.  Output arrays are
 * malloc'ed (orc_finish_free).  Layout of rev_seq/rev_qual: records in the order items, then reads; every record
 * starts on a 16-byte boundary. */
int orc_finish_batch(const plo_batch_in *in, const plo_finish_in *fin, const plo_batch_out *lift, plo_finish_out *out);
void orc_finish_free(plo_finish_out *out);

/* SA:Z values restated from src/read_alignment_scanner.rs:292-301 (get_sa_tag_segment) and :348-364 (every lifted record
 * of a read gets the segments of the read's other lifted records, in record order; nothing when there are none).
 * `item_flag` = flags of the lifted records (orc_finish_batch), `chrom_names` = ChromList labels.  values[i] is a malloc'ed
 * NUL-terminated string, or NULL when record i gets no SA tag (orc_sa_free).  The CIGAR text follows rust-htslib 0.50.0's
 * Display for CigarStringView ("{len}{char}" per op; third-party, not under /root/reference: parity unpinned for it). */
int orc_sa_values(const plo_batch_in *in, const plo_batch_out *lift, const uint16_t *item_flag, const char *const *chrom_names,
                  char **values);
void orc_sa_free(char **values, uint32_t n_items);

#ifdef __cplusplus
}
#endif
#endif
