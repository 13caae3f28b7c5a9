/*
 * portello_bam.h -- host side of the liftover path: BAM/BGZF input, batch construction, BAM record bytes, output.
 *
 * These entry points surround plo_liftover_batch (portello_liftover.h) with what the reference does through
 * rust-htslib (third party, not under /root/reference: rust-htslib 0.50.0 / htslib, Cargo.lock:740-742,1459-1461):
 *
 *   plo_bam_open / plo_bam_header / plo_bam_read_window
 *        bam::IndexedReader + fetch + read loop of scan_chromosome_segment      src/read_alignment_scanner.rs:382-406
 *        (records are taken in file order instead of per-window index fetches: the set of primary, mapped
 *        records -- and therefore the output set -- is the same; output order is unspecified in the reference,
 *        docs/user_guide.md:227-230) and of scan_unmapped_reads                  src/read_alignment_scanner.rs:537-559
 *   plo_bam_window_batch
 *        get_seq_order_read_split_segments for every primary record              lib/rust-vc-utils/src/bam_utils/split_read.rs:56-155
 *        (SA:Z parsing: aux/sa_tag_parser.rs:25-59; clip positions: cigar/mod.rs:85-118), laid out as the
 *        plo_batch_in of portello_liftover.h
 *   plo_records_build
 *        the record bytes of get_liftover_alignment_for_read_and_contig_segment  src/read_alignment_scanner.rs:245-284
 *        (clone_record :105-118, PS/ZM tags :254-268, pos/cigar :270-271, reverse_alignment_seq_and_qual :125-133,
 *        bin :278-279, supplementary :282) and finish_remapped_alignment_set      :310-366
 *        (unmapped copy :317-335, primary selection :338-346, SA tags :348-364), serialised as htslib's bam_write1
 *        does (BAM specification 4.2; CG:B,I for more than 65535 CIGAR ops)
 *   plo_bam_output_header / plo_bam_writer_*
 *        get_alignment_file_header :35-59, get_shared_bam_writer :61-78 (level 0 = uncompressed BGZF as for stdout)
 *
 * Parity: the aux / set() / bam_write1 byte semantics restate third-party code that is absent from the reference
 * tree -- "parity unpinned by the reference's tests"; the SA / split-segment parsing is pinned by split_read.rs:198-232
 * and sa_tag_parser.rs:66-77 (transcribed in tests/test_bam.py: test_sa_parser_reference_vector, test_split_segments_reference_vectors).
 *
 * Conventions as in portello_liftover.h: int status, no unwinding, plo_bam_last_error() for the message; objects are
 * single-threaded (internally they use `n_threads` worker threads for (de)compression and record assembly).
 */
#ifndef PORTELLO_BAM_H
#define PORTELLO_BAM_H

#include "portello_liftover.h"

#ifdef __cplusplus
extern "C" {
#endif

/* status codes PLO_ERR_IO / PLO_ERR_DATA (portello_liftover.h): PLO_ERR_DATA marks records the reference would panic on --
   SA segment without aligned bases, SA read length different from the primary's, unknown SA contig, empty split segment
   (split_read.rs:112-151), malformed SA text (sa_tag_parser.rs:27-31) */

typedef struct plo_bam_reader plo_bam_reader;
typedef struct plo_bam_writer plo_bam_writer;
typedef struct plo_bam_window plo_bam_window;

plo_status plo_bam_open(const char *path, int n_threads, plo_bam_reader **out);
/* the same with the choice of plo_bam_set_device_inflate made before the first block is read: device >= 0 that GPU, -1 the host,
   -2 as plo_bam_open (host unless the environment says PLO_BGZF_DEVICE=1) */
plo_status plo_bam_open_device(const char *path, int n_threads, int device, plo_bam_reader **out);
/* one PART of the file for one rank / worker (several GPUs: INTEGRATION.md section 6): the compressed file is cut at size x part /
   n_parts; the part owns the records whose first byte lies in a BGZF block that starts inside its stretch (it reads on past the
   stretch's end to finish the last one), found without an index: the first block by its header chain, the first record as the
   offset from which eight records in a row parse.  The parts' record sets are disjoint and their union is the file's.  `device` as
   in plo_bam_open_device.  Replaces the per-worker IndexedReader fetch of src/worker_thread_data.rs:21-30,
   src/read_alignment_scanner.rs:382. */
plo_status plo_bam_open_range(const char *path, int n_threads, int device, uint32_t part, uint32_t n_parts, plo_bam_reader **out);
void plo_bam_close(plo_bam_reader *r);
/* header text and the @SQ list as stored in the BAM header (ChromList::from_bam_header, chrom_list.rs:27-37).
   The pointers stay valid until plo_bam_close. */
plo_status plo_bam_header(const plo_bam_reader *r, const char **text, uint32_t *l_text, uint32_t *n_ref,
                          const char *const **ref_names, const uint32_t **ref_lens);

/* Next window of the file: at most max_records primary records.  A window also ends after 4 x max_records + 1024 unmapped
 * records or max(1 GB, 64 KB x max_records) of records, so a window with 0 primary records is NOT the end of the file (the unmapped tail of a sorted BAM
 * comes as several such windows).  The end of the file is a window with 0 primary AND 0 unmapped records -- or, without
 * counting, plo_bam_window_eof() != 0: the stream ended inside or right behind this window (its records, if any, are the last).
 * Records are classified as the reference does:
 *   unmapped flag set, no reference id    -> pass-through list (scan_unmapped_reads :551-555)
 *   unmapped flag set, reference id >= 0  -> PLO_ERR_DATA (the reference's window loop asserts !is_unmapped(), :396)
 *   supplementary flag set                -> skipped (:404)
 *   everything else                       -> a primary read of the batch
 * The window owns copies of the record bytes. */
/* Inflate the BGZF blocks on GPU `device` (one wavefront per block, groups of blocks pipelined with the host's staging and CRC passes)
 * instead of on the reader's threads, from the next refill on.  The kernel inflates about three times what 16
 * host cores do; the refill as a whole takes as long either way (the host still stages the compressed bytes and checks the CRCs),
 * but it leaves the cores to the other stages of a pipeline.  Falls back to the host when no device is usable; the environment
 * variable PLO_BGZF_DEVICE=0/1 overrides. */
void plo_bam_set_device_inflate(plo_bam_reader *r, int device); /* HIP device index, or -1: inflate on the host */
plo_status plo_bam_read_window(plo_bam_reader *r, uint32_t max_records, plo_bam_window **out);
void plo_bam_window_free(plo_bam_window *w);
uint32_t plo_bam_window_n_records(const plo_bam_window *w);
/* 1: the reader reached the end of the file while collecting this window (nothing follows it), 0: more windows follow */
int plo_bam_window_eof(const plo_bam_window *w);
/* primary record i of the window as it stands in the BAM stream: block_size word + block_size bytes (valid until plo_bam_window_free) */
plo_status plo_bam_window_record(const plo_bam_window *w, uint32_t i, const uint8_t **bytes, uint32_t *n_bytes);
/* unmapped records of the window's stretch of the file as BAM record bytes (block_size prefixed), ready for
   plo_bam_write to the "unassembled" output */
void plo_bam_window_unmapped(const plo_bam_window *w, const uint8_t **bytes, uint64_t *n_bytes, uint32_t *n_records);

/* The window as a plo_batch_in (host arrays owned by the window, page-locked when a HIP device is usable):
 * reads = the primary records, segments = get_seq_order_read_split_segments of each, read bases = the records' 4-bit
 * packed sequences.  Also returns the qualities and flags that plo_finish_batch_dev takes (optional, may be NULL). */
plo_status plo_bam_window_batch(plo_bam_window *w, plo_batch_in *batch, plo_finish_in *fin);

/* The same with the read bases as PLO_SEQ_BAM4_SPARSE (portello_liftover.h): only the granules within `margin` bases of an
 * insertion or deletion of a read->contig CIGAR are copied out of the records (about 3 % of a HiFi read at margin 32), and
 * seq_full / read_seq_full_off point at the complete bases inside the window's records, where plo_liftover_batch looks when a
 * comparison runs past what was sent.  For the path that finishes records on the host (plo_records_build); the result of
 * plo_liftover_batch is the same as with plo_bam_window_batch whatever the margin. */
plo_status plo_bam_window_batch_sparse(plo_bam_window *w, uint32_t margin, plo_batch_in *batch, plo_finish_in *fin);
/* The same, told where the reverse-mapped contig segments are (`index`: the description the plo_index was made of; read: contig_seg_off,
 * seg_seq_order_start / _end, seg_is_fwd_strand).  A read segment that touches no reverse-mapped contig segment (the overlap rule of
 * src/read_alignment_scanner.rs:80-103) never takes the left shift (:159-176 runs it for reverse-mapped contig segments only), so only
 * the bases of its CIGAR's insertions (+ 16 on either side) are sent -- what simplify_alignment_indels may compare -- and neither the
 * margins nor the deletions' flanks: about a third of the bytes for such reads.  As with every sparse batch the result does not depend on
 * what was sent (an item that reaches an absent granule is lifted again from the complete bases). */
plo_status plo_bam_window_batch_sparse_strand(plo_bam_window *w, uint32_t margin, const plo_index_desc *index, plo_batch_in *batch,
                                              plo_finish_in *fin);

/* The same transformation for a batch that already exists with dense PLO_SEQ_BAM4 bases in host memory (seg_read non-decreasing).
 * `out` needs plo_sparse_seq_bound(dense) bytes (page-locked memory from plo_host_alloc makes the upload faster), out_read_off
 * n_reads entries.  *sparse = *dense with seq / seq_bytes / seq_fmt / read_seq_off replaced and seq_full / read_seq_full_off
 * pointing at the dense bases. */
uint64_t plo_sparse_seq_bound(const plo_batch_in *dense);
plo_status plo_sparse_seq_pack(const plo_batch_in *dense, uint32_t margin, int n_threads, uint8_t *out, uint64_t out_cap,
                               uint64_t *out_read_off, plo_batch_in *sparse);

/* Output records of a window.  `lift` = host result of plo_liftover_batch on the window's batch (items ordered by read
 * segment, contig segment).  For every read, in order: its lifted records (item order) with flags, bin, PS/ZM/SA tags as
 * the reference writes them, or -- when nothing lifted and !is_target_region -- the unmapped copy.
 * LEN_MISMATCH / PANIC items are the reference's aborts: the call fails with PLO_ERR_DATA. */
typedef struct plo_records_params {
    const plo_index_desc *index;        /* contig segments: is_fwd_strand for the PS tag, ordering                  */
    const char *const *contig_names;    /* [n_contigs] labels of the read->contig BAM header (PS tag)               */
    const char *const *ref_names;       /* [n_chroms] labels of the reference ChromList (SA tag)                    */
    int32_t is_target_region;           /* src/read_alignment_scanner.rs:318-320                                    */
    int32_t n_threads;
} plo_records_params;

typedef struct plo_record_buf {
    const uint8_t *bytes;       /* BAM records, each prefixed by its block_size (what bam_write1 emits)             */
    uint64_t n_bytes;
    uint32_t n_records;
    const uint64_t *record_off; /* [n_records + 1] */
    uint32_t n_lifted, n_unmapped_copies;
} plo_record_buf;

plo_status plo_records_build(plo_bam_window *w, const plo_batch_out *lift, const plo_records_params *params,
                             plo_record_buf *out);

/* The same records from a batch that was finished on the device: `fin` and `sa` are HOST copies of what
 * plo_finish_batch_dev and plo_sa_segments_dev returned for this window's batch (every array; `sa` may be NULL, the SA
 * text is then written here).  Flags, bin, the primary record, reverse_alignment_seq_and_qual's bases and qualities
 * (src/read_alignment_scanner.rs:125-133) and get_sa_tag_segment's text (:292-301) are taken from there and copied into
 * place; this call lays out the records, cuts the aux fields and writes PS / ZM.  Byte-identical to plo_records_build.
 * The window's batch must have been the dense one (plo_bam_window_batch): the device needs every base to reverse them. */
plo_status plo_records_build_finished(plo_bam_window *w, const plo_batch_out *lift, const plo_finish_out *fin,
                                      const plo_sa_out *sa, const plo_records_params *params, plo_record_buf *out);

/* Header of the output files (get_alignment_file_header :35-59): @HD VN:1.6 SO:unsorted, one @SQ per chromosome,
   @PG PN/ID/VN/CL.  Returns a malloc'ed NUL-terminated text (free with plo_bam_free_text). */
char *plo_bam_output_header(uint32_t n_ref, const char *const *ref_names, const uint32_t *ref_lens, const char *program_name,
                            const char *program_version, const char *cmdline);
void plo_bam_free_text(char *text);

/* level 0: BGZF blocks with stored (uncompressed) deflate data, what the reference selects for stdout (:67-71);
   1..9: zlib deflate levels */
plo_status plo_bam_writer_open(const char *path, const char *header_text, uint32_t n_ref, const char *const *ref_names,
                               const uint32_t *ref_lens, int level, int n_threads, plo_bam_writer **out);
plo_status plo_bam_write(plo_bam_writer *w, const uint8_t *record_bytes, uint64_t n_bytes);
plo_status plo_bam_writer_close(plo_bam_writer *w);

/* ------------------------------------------------------------------------------------------------------------------
 * Phase 1: the contig->reference index from the assembly->reference BAM (scan_contig_bam,
 * src/contig_alignment_scanner/mod.rs:290-459): primary records -> sequencing-order split segments (:91-133), exact CIGARs
 * of the supplementary records matched by (chrom, pos, strand, leading clip, trailing clip) (:135-183, :371-416), optional
 * target-region filter (non_targeted_segment_filter.rs:7-39), clip_repeated_contig_matches (gap-compressed identity, then
 * MAPQ; contig_repeated_match_trimmer.rs:214-303), join_colinear_contig_segments (same chromosome / strand / MAPQ,
 * reference gap 0..1000; contig_colinear_segment_joiner.rs:124-186), and the reverse-complemented contig sequence of every
 * contig with a reverse-mapped segment (:113-125).  The result is handed to plo_index_create as a plo_index_desc (which
 * builds the block maps on the device); the caller adds the reference sequences (chrom_len / chrom_seq).
 * Pinned by the reference's vectors of contig_repeated_match_trimmer.rs:311-397 and clip_alignment.rs:188-248.
 * Inputs the reference panics on (missing supplementary record in a whole-genome run, ambiguous supplementary key, M ops in
 * an overlap that needs the identity, unknown contig name) return PLO_ERR_DATA.
 * ---------------------------------------------------------------------------------------------------------------- */
typedef struct plo_phase1 plo_phase1;
typedef struct plo_target_region {  /* GenomeSegment of --target-region (debug option, src/cli.rs:24-75) */
    uint32_t chrom_index;
    int64_t start, end;
} plo_target_region;

/* contig_names / contig_lens: the @SQ list of the read->contig BAM (ChromList of the assembly contigs) */
plo_status plo_phase1_scan(const char *asm_to_ref_bam, uint32_t n_contigs, const char *const *contig_names, const int64_t *contig_lens,
                           const plo_target_region *target_region /* NULL = whole genome */, int n_threads, plo_phase1 **out);
/* every array of the descriptor except chrom_len / chrom_seq (pointers owned by the plo_phase1 object) */
plo_status plo_phase1_index_desc(const plo_phase1 *ph, plo_index_desc *desc);
/* reference ChromList of the assembly->reference BAM header + the counters the reference logs (trimmer.rs:302, joiner.rs:185) */
plo_status plo_phase1_info(const plo_phase1 *ph, uint32_t *n_ref, const char *const **ref_names, const uint32_t **ref_lens,
                           uint32_t *segments_clipped, uint32_t *segments_joined, uint32_t *n_records);
void plo_phase1_free(plo_phase1 *ph);

const char *plo_bam_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* PORTELLO_BAM_H */
