/*
 * portello_oracle.c -- CPU restatement of portello's liftover hot path.  TEST INFRASTRUCTURE ONLY
 * (see portello_oracle.h).  Plain C11, one item at a time, same allocation pattern as the reference.
 * Citations are relative to /root/reference.
 */
#define _GNU_SOURCE
#include "portello_oracle.h"

#include <pthread.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

enum { OP_M = 0, OP_I = 1, OP_D = 2, OP_N = 3, OP_S = 4, OP_H = 5, OP_P = 6, OP_EQ = 7, OP_X = 8 };
#define CIG(op, len) ((((uint32_t)(len)) << 4) | (uint32_t)(op))
#define CIG_OP(c) ((c)&0xfu)
#define CIG_LEN(c) ((c) >> 4)

/* ---- growable Vec<Cigar> ---------------------------------------------------------------------------------- */
typedef struct {
    uint32_t *v;
    size_t n, cap;
} cvec;
static void cvec_push(cvec *c, uint32_t x) {
    if (c->n == c->cap) {
        c->cap = c->cap ? c->cap * 2 : 16;
        c->v = (uint32_t *)realloc(c->v, c->cap * sizeof(uint32_t));
    }
    c->v[c->n++] = x;
}
static void cvec_free(cvec *c) {
    free(c->v);
    c->v = NULL;
    c->n = c->cap = 0;
}

/* ---- lib/rust-vc-utils/src/bam_utils/cigar/mod.rs --------------------------------------------------------- */

/* :22-24 is_alignment_match */
int orc_is_alignment_match(uint32_t c) {
    uint32_t op = CIG_OP(c);
    return op == OP_M || op == OP_EQ || op == OP_X;
}

/* :26-39 get_cigarseg_read_offset */
uint64_t orc_cigarseg_read_offset(uint32_t c, int ignore_hard_clip) {
    switch (CIG_OP(c)) {
        case OP_I:
        case OP_S:
        case OP_X:
        case OP_EQ:
        case OP_M:
            return CIG_LEN(c);
        case OP_H:
            return ignore_hard_clip ? 0 : CIG_LEN(c);
        default:
            return 0;
    }
}

/* :41-47 get_cigarseg_ref_offset */
int64_t orc_cigarseg_ref_offset(uint32_t c) {
    switch (CIG_OP(c)) {
        case OP_D:
        case OP_N:
        case OP_X:
        case OP_EQ:
        case OP_M:
            return (int64_t)CIG_LEN(c);
        default:
            return 0;
    }
}

/* :164-170 get_cigar_read_offset */
uint64_t orc_cigar_read_offset(const uint32_t *cig, size_t n, int ihc) {
    uint64_t read_pos = 0;
    for (size_t i = 0; i < n; ++i) read_pos += orc_cigarseg_read_offset(cig[i], ihc);
    return read_pos;
}

/* :174-180 get_cigar_ref_offset */
int64_t orc_cigar_ref_offset(const uint32_t *cig, size_t n) {
    int64_t ref_pos = 0;
    for (size_t i = 0; i < n; ++i) ref_pos += orc_cigarseg_ref_offset(cig[i]);
    return ref_pos;
}

/* :85-118 get_read_clip_positions */
void orc_read_clip_positions(const uint32_t *cig, size_t n, int ihc, uint64_t out3[3]) {
    uint64_t read_pos = 0, left_clip_size = 0, right_clip_size = 0;
    int left_clip = 1;
    for (size_t i = 0; i < n; ++i) {
        uint32_t c = cig[i];
        switch (CIG_OP(c)) {
            case OP_S:
                if (left_clip)
                    left_clip_size += CIG_LEN(c);
                else
                    right_clip_size += CIG_LEN(c);
                break;
            case OP_H:
                if (!ihc) {
                    if (left_clip)
                        left_clip_size += CIG_LEN(c);
                    else
                        right_clip_size += CIG_LEN(c);
                }
                break;
            default:
                left_clip = 0;
        }
        read_pos += orc_cigarseg_read_offset(c, ihc);
    }
    out3[0] = left_clip_size;
    out3[1] = read_pos - right_clip_size;
    out3[2] = read_pos;
}

/* :204-228 compress_cigar.  `last_elem` starts as Match(0); zero-length ops are filtered first; the merge
 * pattern (:210-212) lists every variant except Pad, so a Pad following a Pad is dropped without summing. */
size_t orc_compress_cigar(const uint32_t *in, size_t n, uint32_t *out) {
    size_t n_out = 0;
    uint32_t last_elem = CIG(OP_M, 0);
    for (size_t i = 0; i < n; ++i) {
        uint32_t new_elem = in[i];
        if (CIG_LEN(new_elem) == 0) continue; /* .filter(|x| !x.is_empty()) */
        if (CIG_OP(new_elem) == CIG_OP(last_elem)) {
            if (CIG_OP(last_elem) != OP_P) last_elem = CIG(CIG_OP(last_elem), CIG_LEN(last_elem) + CIG_LEN(new_elem));
        } else {
            if (CIG_LEN(last_elem) != 0) out[n_out++] = last_elem;
            last_elem = new_elem;
        }
    }
    if (CIG_LEN(last_elem) != 0) out[n_out++] = last_elem;
    return n_out;
}

/* :265-291 clean_up_cigar_edge_indels */
static uint64_t edge_update_element(uint32_t *c) {
    uint64_t ret = 0;
    if (CIG_OP(*c) == OP_D) {
        ret = CIG_LEN(*c);
        *c = CIG(OP_S, 0);
    } else if (CIG_OP(*c) == OP_I) {
        *c = CIG(OP_S, CIG_LEN(*c));
    }
    return ret;
}
uint64_t orc_clean_up_cigar_edge_indels(uint32_t *cig, size_t n) {
    uint64_t del_shift = 0;
    for (size_t i = 0; i < n && !orc_is_alignment_match(cig[i]); ++i) del_shift += edge_update_element(&cig[i]);
    for (size_t i = n; i > 0 && !orc_is_alignment_match(cig[i - 1]); --i) edge_update_element(&cig[i - 1]);
    return del_shift;
}

/* ---- lib/rust-vc-utils/src/seq_util.rs -------------------------------------------------------------------- */

/* :1-15 comp_base */
uint8_t orc_comp_base(uint8_t b) {
    switch (b) {
        case 'A': return 'T';
        case 'T': return 'A';
        case 'C': return 'G';
        case 'G': return 'C';
        case 'N': return 'N';
        case 'a': return 't';
        case 't': return 'a';
        case 'c': return 'g';
        case 'g': return 'c';
        case 'n': return 'n';
        default: return 'N';
    }
}

/* :28-40 rev_comp_in_place */
void orc_rev_comp_in_place(uint8_t *dna, size_t len) {
    size_t halflen = len - len / 2;
    for (size_t i = 0; i < halflen; ++i) {
        dna[i] = orc_comp_base(dna[i]);
        size_t rev_i = len - 1 - i;
        if (i != rev_i) {
            dna[rev_i] = orc_comp_base(dna[rev_i]);
            uint8_t t = dna[i];
            dna[i] = dna[rev_i];
            dna[rev_i] = t;
        }
    }
}

/* rust-htslib 0.50.0 bam::record::Seq::as_bytes(): DECODE_BASE = b"=ACMGRSVTWYHKDBN", high nibble first
 * (third-party, not under /root/reference; the table is the BAM specification's) */
void orc_decode_bam4(const uint8_t *packed, size_t n_bases, uint8_t *out) {
    static const char tbl[] = "=ACMGRSVTWYHKDBN";
    for (size_t i = 0; i < n_bases; ++i) {
        uint8_t b = packed[i >> 1];
        out[i] = (uint8_t)tbl[(i & 1) ? (b & 0xf) : (b >> 4)];
    }
}

/* ---- lib/rust-vc-utils/src/indel_breakend_homology.rs:24-73 ------------------------------------------------ */
int orc_indel_breakend_homology(const uint8_t *ref_seq, int64_t ref_len, int64_t ref_start, int64_t ref_end,
                                const uint8_t *read_seq, int64_t read_len, int64_t read_start, int64_t read_end,
                                int64_t *hom_start, int64_t *hom_end) {
    /* :32-47 left */
    int64_t max_left_offset = ref_start < read_start ? ref_start : read_start;
    int64_t left_offset = 0;
    for (;;) {
        if (left_offset >= max_left_offset) break;
        int64_t ri = ref_end - left_offset - 1;
        int64_t qi = read_end - left_offset - 1;
        if (ri < 0 || ri >= ref_len || qi < 0 || qi >= read_len) return ORC_PANIC;
        if (ref_seq[ri] != read_seq[qi]) break;
        left_offset += 1;
    }
    /* :51-68 right */
    int64_t a = ref_len - ref_end, b = read_len - read_end;
    int64_t max_right_offset = a < b ? a : b;
    int64_t right_offset = 0;
    for (;;) {
        if (right_offset >= max_right_offset) break;
        int64_t ri = ref_start + right_offset;
        int64_t qi = read_start + right_offset;
        if (ri < 0 || ri >= ref_len || qi < 0 || qi >= read_len) return ORC_PANIC;
        if (ref_seq[ri] != read_seq[qi]) break;
        right_offset += 1;
    }
    *hom_start = -left_offset;
    *hom_end = right_offset;
    return 0;
}

/* ---- lib/rust-vc-utils/src/bam_utils/cigar/shift_indels/cigar_indel_shifter.rs:10-165 ---------------------- */
typedef struct {
    int dir; /* 0 left, 1 right */
    const uint8_t *ref_seq;
    int64_t ref_len;
    const uint8_t *read_seq;
    int64_t read_len;
    uint32_t match_block_size;
    int is_in_indel_block;
    int64_t indel_block_ref_start;
    uint64_t indel_block_read_start;
    uint32_t indel_block_del_size;
    uint32_t indel_block_ins_size;
    cvec shift_cigar;
    int panicked;
} shift_builder;

/* :63-71 add_indel */
static void sb_add_indel(shift_builder *s, int64_t ref_pos, uint64_t read_pos) {
    if (s->dir == 1 || !s->is_in_indel_block) {
        s->indel_block_ref_start = ref_pos;
        s->indel_block_read_start = read_pos;
        if (!s->is_in_indel_block) s->is_in_indel_block = 1;
    }
}
/* :87-99 push_del_segment / push_ins_segment */
static void sb_push_del_segment(shift_builder *s) {
    if (s->indel_block_del_size > 0) {
        cvec_push(&s->shift_cigar, CIG(OP_D, s->indel_block_del_size));
        s->indel_block_del_size = 0;
    }
}
static void sb_push_ins_segment(shift_builder *s) {
    if (s->indel_block_ins_size > 0) {
        cvec_push(&s->shift_cigar, CIG(OP_I, s->indel_block_ins_size));
        s->indel_block_ins_size = 0;
    }
}
/* :101-148 end_indel */
static void sb_end_indel(shift_builder *s) {
    if (!s->is_in_indel_block) return;
    s->is_in_indel_block = 0;

    int64_t hs = 0, he = 0;
    int rc = orc_indel_breakend_homology(s->ref_seq, s->ref_len, s->indel_block_ref_start,
                                         s->indel_block_ref_start + (int64_t)s->indel_block_del_size, s->read_seq,
                                         s->read_len, (int64_t)s->indel_block_read_start,
                                         (int64_t)s->indel_block_read_start + (int64_t)s->indel_block_ins_size, &hs,
                                         &he);
    if (rc != 0) {
        s->panicked = 1;
        return;
    }
    int64_t sl = (s->dir == 0) ? -hs : he;
    uint32_t shift_len = (uint32_t)(sl > 0 ? sl : 0);

    uint32_t actual_shift_len = s->match_block_size < shift_len ? s->match_block_size : shift_len;
    uint32_t shifted_match_block_size = s->match_block_size - actual_shift_len;
    if (shifted_match_block_size > 0) cvec_push(&s->shift_cigar, CIG(OP_M, shifted_match_block_size));
    s->match_block_size = actual_shift_len;

    if (s->dir == 0) sb_push_ins_segment(s);
    sb_push_del_segment(s);
    if (s->dir == 1) sb_push_ins_segment(s);
}
/* :155-165 add_other */
static void sb_add_other(shift_builder *s, const uint32_t *cigar_seg) {
    sb_end_indel(s);
    if (s->match_block_size > 0) {
        cvec_push(&s->shift_cigar, CIG(OP_M, s->match_block_size));
        s->match_block_size = 0;
    }
    if (cigar_seg) cvec_push(&s->shift_cigar, *cigar_seg);
}
/* :43-52 add_element */
static void sb_add_element(shift_builder *s, uint32_t c, int64_t ref_pos, uint64_t read_pos) {
    uint32_t len = CIG_LEN(c);
    switch (CIG_OP(c)) {
        case OP_D: /* :73-78 add_del */
            if (len > 0) {
                sb_add_indel(s, ref_pos, read_pos);
                s->indel_block_del_size += len;
            }
            break;
        case OP_I: /* :80-85 add_ins */
            if (len > 0) {
                sb_add_indel(s, ref_pos, read_pos);
                s->indel_block_ins_size += len;
            }
            break;
        case OP_M:
        case OP_EQ:
        case OP_X: /* :150-153 add_match */
            sb_end_indel(s);
            s->match_block_size += len;
            break;
        default:
            sb_add_other(s, &c);
    }
}

/* left_shift_indels.rs:17-39 and right_shift_indels.rs:20-50 */
int orc_shift_indels(int dir, int64_t ref_pos, const uint32_t *cig, size_t n, const uint8_t *ref_seq,
                     int64_t ref_len, const uint8_t *read_seq, int64_t read_len, int64_t *out_pos, uint32_t *out,
                     size_t *n_out) {
    shift_builder s;
    memset(&s, 0, sizeof(s));
    s.dir = dir;
    s.ref_seq = ref_seq;
    s.ref_len = ref_len;
    s.read_seq = read_seq;
    s.read_len = read_len;

    if (dir == 0) {
        int64_t ref_head_pos = ref_pos;
        uint64_t read_head_pos = 0;
        for (size_t i = 0; i < n; ++i) {
            sb_add_element(&s, cig[i], ref_head_pos, read_head_pos);
            read_head_pos += orc_cigarseg_read_offset(cig[i], 0);
            ref_head_pos += orc_cigarseg_ref_offset(cig[i]);
        }
    } else {
        int64_t *rp = (int64_t *)malloc((n + 1) * sizeof(int64_t));
        uint64_t *qp = (uint64_t *)malloc((n + 1) * sizeof(uint64_t));
        int64_t ref_head_pos = ref_pos;
        uint64_t read_head_pos = 0;
        for (size_t i = 0; i < n; ++i) {
            rp[i] = ref_head_pos;
            qp[i] = read_head_pos;
            read_head_pos += orc_cigarseg_read_offset(cig[i], 0);
            ref_head_pos += orc_cigarseg_ref_offset(cig[i]);
        }
        for (size_t i = n; i > 0; --i) sb_add_element(&s, cig[i - 1], rp[i - 1], qp[i - 1]);
        free(rp);
        free(qp);
    }
    /* :54-60 get_cigar */
    sb_add_other(&s, NULL);
    if (dir == 1) {
        for (size_t i = 0, j = s.shift_cigar.n; i + 1 < j; ++i) {
            --j;
            uint32_t t = s.shift_cigar.v[i];
            s.shift_cigar.v[i] = s.shift_cigar.v[j];
            s.shift_cigar.v[j] = t;
        }
    }
    if (s.panicked) {
        cvec_free(&s.shift_cigar);
        *n_out = 0;
        return ORC_PANIC;
    }
    uint64_t ref_pos_shift = orc_clean_up_cigar_edge_indels(s.shift_cigar.v, s.shift_cigar.n);
    *n_out = orc_compress_cigar(s.shift_cigar.v, s.shift_cigar.n, out);
    *out_pos = ref_pos + (int64_t)ref_pos_shift;
    cvec_free(&s.shift_cigar);
    return 0;
}

/* ---- lib/rust-vc-utils/src/bam_utils/read_to_ref_map.rs ---------------------------------------------------- */

/* BTreeMap<usize, Option<i64>> restated as a sorted array; insert = overwrite on equal key */
static size_t map_insert(uint64_t *keys, int64_t *vals, size_t nk, uint64_t k, int64_t v) {
    size_t lo = 0, hi = nk;
    while (lo < hi) {
        size_t mid = (lo + hi) / 2;
        if (keys[mid] < k)
            lo = mid + 1;
        else
            hi = mid;
    }
    if (lo < nk && keys[lo] == k) {
        vals[lo] = v;
        return nk;
    }
    memmove(keys + lo + 1, keys + lo, (nk - lo) * sizeof(uint64_t));
    memmove(vals + lo + 1, vals + lo, (nk - lo) * sizeof(int64_t));
    keys[lo] = k;
    vals[lo] = v;
    return nk + 1;
}

/* :101-137 get_read_segment_to_ref_pos_tree_map */
size_t orc_map_build(int64_t ref_pos, const uint32_t *cig, size_t n, int ignore_hard_clip, uint64_t *keys,
                     int64_t *vals) {
    size_t nk = 0;
    uint64_t read_pos = 0;
    uint64_t match_len = 0;
#define UPDATE_MAP()                                                                          \
    do {                                                                                      \
        if (match_len > 0) {                                                                  \
            nk = map_insert(keys, vals, nk, read_pos - match_len, ref_pos - (int64_t)match_len); \
            nk = map_insert(keys, vals, nk, read_pos, ORC_NONE);                              \
            match_len = 0;                                                                    \
        }                                                                                     \
    } while (0)
    for (size_t i = 0; i < n; ++i) {
        uint32_t c = cig[i];
        if (orc_is_alignment_match(c)) {
            match_len += CIG_LEN(c);
        } else {
            UPDATE_MAP();
        }
        read_pos += orc_cigarseg_read_offset(c, ignore_hard_clip);
        ref_pos += orc_cigarseg_ref_offset(c);
    }
    UPDATE_MAP();
#undef UPDATE_MAP
    return nk;
}

/* index of the greatest key <= x, or -1: map.range(..=x).next_back() */
static int64_t map_floor(const uint64_t *keys, size_t nk, uint64_t x) {
    size_t lo = 0, hi = nk;
    while (lo < hi) {
        size_t mid = (lo + hi) / 2;
        if (keys[mid] <= x)
            lo = mid + 1;
        else
            hi = mid;
    }
    return (int64_t)lo - 1;
}
/* index of the first key >= x */
static size_t map_lower_bound(const uint64_t *keys, size_t nk, uint64_t x) {
    size_t lo = 0, hi = nk;
    while (lo < hi) {
        size_t mid = (lo + hi) / 2;
        if (keys[mid] < x)
            lo = mid + 1;
        else
            hi = mid;
    }
    return lo;
}

/* :66-72 get_ref_pos */
int64_t orc_map_get_ref_pos(const uint64_t *keys, const int64_t *vals, size_t nk, uint64_t read_pos) {
    int64_t i = map_floor(keys, nk, read_pos);
    if (i < 0) return ORC_NONE;
    if (vals[i] == ORC_NONE) return ORC_NONE;
    return vals[i] + (int64_t)(read_pos - keys[i]);
}

/* :74-85 get_ref_range -> half-open index range [i0,i1) of the entries of map.range(start_block..read_end_pos) */
void orc_map_get_ref_range(const uint64_t *keys, size_t nk, uint64_t a, uint64_t b, size_t *i0, size_t *i1) {
    int64_t f = map_floor(keys, nk, a);
    uint64_t read_start_block_pos = (f >= 0) ? keys[f] : a;
    *i0 = map_lower_bound(keys, nk, read_start_block_pos);
    *i1 = map_lower_bound(keys, nk, b);
    if (*i1 < *i0) *i1 = *i0;
}

/* ---- src/liftover_read_alignment.rs ------------------------------------------------------------------------- */

typedef struct {
    int some;
    uint64_t key;
    int64_t val; /* ORC_NONE for (key, None) */
} opt_block;

typedef struct {
    int some;
    int64_t v;
} opt_i64;

/* :35-133 update_ref2_cigar_segment */
static void update_ref2_cigar_segment(opt_block this_blk, opt_block last_blk, int64_t ref1_cigar_segment_end_pos,
                                      uint32_t ref1_cigar_segment, int64_t *block_ref1_pos, opt_i64 *ref2_start_pos,
                                      opt_i64 *ref2_end_pos, cvec *ref2_cigar) {
    /* :62-67 */
    int64_t ref1_remapped_segment_end_pos;
    if (this_blk.some) {
        int64_t k = (int64_t)this_blk.key;
        ref1_remapped_segment_end_pos = k < ref1_cigar_segment_end_pos ? k : ref1_cigar_segment_end_pos;
    } else {
        ref1_remapped_segment_end_pos = ref1_cigar_segment_end_pos;
    }

    if (ref1_remapped_segment_end_pos > *block_ref1_pos) { /* :69 */
        uint32_t remapped_segment_len = (uint32_t)(ref1_remapped_segment_end_pos - *block_ref1_pos);
        int is_match_segment = orc_is_alignment_match(ref1_cigar_segment);

        if (last_blk.some) { /* :79 */
            if (last_blk.val != ORC_NONE) { /* :83 */
                int64_t last_block_ref1_start_pos = (int64_t)last_blk.key;
                int64_t last_block_ref2_start_pos = last_blk.val;
                if (is_match_segment && !ref2_start_pos->some) { /* :84-88 */
                    ref2_start_pos->some = 1;
                    ref2_start_pos->v = last_block_ref2_start_pos + (*block_ref1_pos - last_block_ref1_start_pos);
                }
                if (ref2_end_pos->some) { /* :91-96 */
                    int64_t deletion_len = last_block_ref2_start_pos - ref2_end_pos->v;
                    if (deletion_len > 0 && ref2_start_pos->some) cvec_push(ref2_cigar, CIG(OP_D, (uint32_t)deletion_len));
                }
                /* :98-100 */
                int64_t last_block_expected_ref2_len = ref1_remapped_segment_end_pos - last_block_ref1_start_pos;
                ref2_end_pos->some = 1;
                ref2_end_pos->v = last_block_ref2_start_pos + last_block_expected_ref2_len;

                if (is_match_segment || ref2_start_pos->some) { /* :102-109 */
                    uint32_t op = CIG_OP(ref1_cigar_segment);
                    uint32_t new_op = (op == OP_D) ? OP_D : (op == OP_N) ? OP_N : OP_M;
                    cvec_push(ref2_cigar, CIG(new_op, remapped_segment_len));
                }
            } else { /* :111-115 */
                if (is_match_segment) cvec_push(ref2_cigar, CIG(OP_I, remapped_segment_len));
            }
        } else { /* :117-123 */
            if (is_match_segment) cvec_push(ref2_cigar, CIG(OP_S, remapped_segment_len));
        }
        *block_ref1_pos = ref1_remapped_segment_end_pos; /* :124 */
    }
}

/* :137-223 liftover_read_alignment */
int orc_liftover_read_alignment(const uint64_t *keys, const int64_t *vals, size_t nk, int64_t start,
                                const uint32_t *cig, size_t n, int64_t *out_pos, uint32_t *out, size_t *n_out) {
    int64_t ref1_cigar_segment_start_pos = start;
    opt_i64 ref2_start_pos = {0, 0};
    opt_i64 ref2_end_pos = {0, 0};
    cvec ref2_cigar = {0, 0, 0};

    for (size_t ci = 0; ci < n; ++ci) {
        uint32_t seg = cig[ci];
        switch (CIG_OP(seg)) {
            case OP_I:
            case OP_S:
            case OP_H: /* :157-160 */
                cvec_push(&ref2_cigar, seg);
                break;
            case OP_X:
            case OP_EQ:
            case OP_M:
            case OP_D:
            case OP_N: { /* :161-212 */
                opt_block last_blk = {0, 0, 0};
                int64_t block_ref1_pos = ref1_cigar_segment_start_pos;
                int64_t ref1_cigar_segment_end_pos = ref1_cigar_segment_start_pos + (int64_t)CIG_LEN(seg);
                size_t i0, i1;
                orc_map_get_ref_range(keys, nk, (uint64_t)ref1_cigar_segment_start_pos,
                                      (uint64_t)ref1_cigar_segment_end_pos, &i0, &i1);
                for (size_t bi = i0; bi < i1; ++bi) {
                    opt_block this_blk = {1, keys[bi], vals[bi]};
                    update_ref2_cigar_segment(this_blk, last_blk, ref1_cigar_segment_end_pos, seg, &block_ref1_pos,
                                              &ref2_start_pos, &ref2_end_pos, &ref2_cigar);
                    last_blk = this_blk;
                }
                opt_block none_blk = {0, 0, 0};
                update_ref2_cigar_segment(none_blk, last_blk, ref1_cigar_segment_end_pos, seg, &block_ref1_pos,
                                          &ref2_start_pos, &ref2_end_pos, &ref2_cigar);
                break;
            }
            default: /* Pad :213 */
                break;
        }
        ref1_cigar_segment_start_pos += orc_cigarseg_ref_offset(seg); /* :215 */
    }

    int ret = 0;
    *n_out = 0;
    if (ref2_start_pos.some) { /* :218-222 */
        uint64_t shift = orc_clean_up_cigar_edge_indels(ref2_cigar.v, ref2_cigar.n);
        *n_out = orc_compress_cigar(ref2_cigar.v, ref2_cigar.n, out);
        *out_pos = ref2_start_pos.v + (int64_t)shift;
        ret = 1;
    }
    cvec_free(&ref2_cigar);
    return ret;
}

/* ---- src/simplify_alignment_indels.rs ----------------------------------------------------------------------- */
typedef struct {
    int is_in_indel_block;
    int64_t block_ref_start;
    uint64_t block_read_start;
    uint32_t block_del_size;
    uint32_t block_ins_size;
} cigar_block_info;

/* :35-111 end_indel; appends to `ret`; returns 0 or ORC_PANIC */
static int cbi_end_indel(cigar_block_info *b, const uint8_t *ref_seq, int64_t ref_len, const uint8_t *read_seq,
                         int64_t read_len, cvec *ret) {
    if (b->is_in_indel_block) {
        b->is_in_indel_block = 0;
        uint32_t del_len = b->block_del_size, ins_len = b->block_ins_size;
        if (del_len == 0 && ins_len == 0) {
        } else if (del_len == 0) {
            cvec_push(ret, CIG(OP_I, ins_len));
        } else if (ins_len == 0) {
            cvec_push(ret, CIG(OP_D, del_len));
        } else if (del_len == 1 && ins_len == 1) {
            cvec_push(ret, CIG(OP_M, 1));
        } else {
            uint32_t pre_match_len = 0, post_match_len = 0;
            while (del_len > 0 && ins_len > 0) { /* :55-68 */
                int64_t ri = b->block_ref_start + (int64_t)del_len - 1;
                int64_t qi = (int64_t)b->block_read_start + (int64_t)ins_len - 1;
                if (ri < 0 || ri >= ref_len || qi < 0 || qi >= read_len) return ORC_PANIC;
                if (ref_seq[ri] == read_seq[qi]) {
                    del_len -= 1;
                    ins_len -= 1;
                    post_match_len += 1;
                } else {
                    break;
                }
            }
            while (del_len > 0 && ins_len > 0) { /* :71-85 */
                int64_t ri = b->block_ref_start + (int64_t)pre_match_len;
                int64_t qi = (int64_t)b->block_read_start + (int64_t)pre_match_len;
                if (ri < 0 || ri >= ref_len || qi < 0 || qi >= read_len) return ORC_PANIC;
                if (ref_seq[ri] == read_seq[qi]) {
                    del_len -= 1;
                    ins_len -= 1;
                    pre_match_len += 1;
                } else {
                    break;
                }
            }
            if (del_len == 1 && ins_len == 1) { /* :88-92 */
                del_len -= 1;
                ins_len -= 1;
                post_match_len += 1;
            }
            if (pre_match_len) cvec_push(ret, CIG(OP_M, pre_match_len)); /* :101-104 */
            if (ins_len) cvec_push(ret, CIG(OP_I, ins_len));
            if (del_len) cvec_push(ret, CIG(OP_D, del_len));
            if (post_match_len) cvec_push(ret, CIG(OP_M, post_match_len));
        }
        b->block_ins_size = 0;
        b->block_del_size = 0;
    }
    return 0;
}

/* :119-156 simplify_alignment_indels */
int orc_simplify_alignment_indels(int64_t ref_pos, const uint32_t *cig, size_t n, const uint8_t *ref_seq,
                                  int64_t ref_len, const uint8_t *read_seq, int64_t read_len, int64_t *out_pos,
                                  uint32_t *out, size_t *n_out) {
    int64_t ref_head_pos = ref_pos;
    uint64_t read_head_pos = 0;
    cigar_block_info b;
    memset(&b, 0, sizeof(b));
    cvec simple = {0, 0, 0};
    *n_out = 0;

    for (size_t i = 0; i < n; ++i) {
        uint32_t c = cig[i];
        if (CIG_OP(c) == OP_D || CIG_OP(c) == OP_I) {
            if (!b.is_in_indel_block) { /* :16-22 _add_indel */
                b.is_in_indel_block = 1;
                b.block_ref_start = ref_head_pos;
                b.block_read_start = read_head_pos;
            }
            if (CIG_OP(c) == OP_D)
                b.block_del_size += CIG_LEN(c);
            else
                b.block_ins_size += CIG_LEN(c);
        } else {
            if (cbi_end_indel(&b, ref_seq, ref_len, read_seq, read_len, &simple) != 0) {
                cvec_free(&simple);
                return ORC_PANIC;
            }
            cvec_push(&simple, c);
        }
        read_head_pos += orc_cigarseg_read_offset(c, 0);
        ref_head_pos += orc_cigarseg_ref_offset(c);
    }
    if (cbi_end_indel(&b, ref_seq, ref_len, read_seq, read_len, &simple) != 0) {
        cvec_free(&simple);
        return ORC_PANIC;
    }
    uint64_t ref_pos_shift = orc_clean_up_cigar_edge_indels(simple.v, simple.n);
    *n_out = orc_compress_cigar(simple.v, simple.n, out);
    *out_pos = ref_pos + (int64_t)ref_pos_shift;
    cvec_free(&simple);
    return 0;
}

/* ---- caller glue: src/read_alignment_scanner.rs ------------------------------------------------------------- */

typedef struct {
    /* one contig segment's block map */
    uint64_t *keys;
    int64_t *vals;
    size_t nk;
} seg_map;

typedef struct {
    const plo_index_desc *index;
    const plo_batch_in *in;
    uint32_t stages;
    const seg_map *maps;
    uint32_t seg_begin, seg_end; /* read segments (or explicit items) handled by this worker */
    /* outputs (growable) */
    uint32_t *item_seg, *item_cseg, *item_chrom, *item_clen;
    uint8_t *item_status, *item_flip, *item_mapq;
    int64_t *item_pos;
    size_t n_items, cap_items;
    cvec cigar;
} worker;

static void worker_reserve(worker *w) {
    if (w->n_items < w->cap_items) return;
    size_t c = w->cap_items ? w->cap_items * 2 : 1024;
    w->item_seg = (uint32_t *)realloc(w->item_seg, c * 4);
    w->item_cseg = (uint32_t *)realloc(w->item_cseg, c * 4);
    w->item_chrom = (uint32_t *)realloc(w->item_chrom, c * 4);
    w->item_clen = (uint32_t *)realloc(w->item_clen, c * 4);
    w->item_status = (uint8_t *)realloc(w->item_status, c);
    w->item_flip = (uint8_t *)realloc(w->item_flip, c);
    w->item_mapq = (uint8_t *)realloc(w->item_mapq, c);
    w->item_pos = (int64_t *)realloc(w->item_pos, c * 8);
    w->cap_items = c;
}

static uint8_t *get_read_seq(const plo_batch_in *in, uint32_t read, int flip) {
    /* record.seq().as_bytes() (+ rev_comp_in_place)   read_alignment_scanner.rs:170-173, 238-241 */
    uint32_t len = in->read_seq_len[read];
    uint8_t *s = (uint8_t *)malloc(len ? len : 1);
    if (in->seq_fmt == PLO_SEQ_BAM4)
        orc_decode_bam4(in->seq + in->read_seq_off[read], len, s);
    else
        memcpy(s, in->seq + in->read_seq_off[read], len);
    if (flip) orc_rev_comp_in_place(s, len);
    return s;
}

/* get_liftover_alignment_for_read_and_contig_segment  :136-288 (the (pos, cigar, status) part) */
static void run_item(worker *w, uint32_t seg, uint32_t cseg) {
    const plo_index_desc *ix = w->index;
    const plo_batch_in *in = w->in;
    uint32_t stages = w->stages;
    uint32_t read = in->seg_read[seg];
    uint32_t contig = in->seg_contig[seg];
    uint32_t gseg = ix->contig_seg_off[contig] + cseg;
    const seg_map *map = &w->maps[gseg];
    int contig_is_fwd_strand = ix->seg_is_fwd_strand[gseg] != 0;
    const uint32_t *in_cig = in->cigar + in->seg_cigar_off[seg];
    size_t n_in = in->seg_cigar_off[seg + 1] - in->seg_cigar_off[seg];

    worker_reserve(w);
    size_t it = w->n_items++;
    w->item_seg[it] = seg;
    w->item_cseg[it] = cseg;
    w->item_chrom[it] = ix->seg_chrom_index[gseg];
    w->item_mapq[it] = ix->seg_mapq[gseg];
    w->item_pos[it] = -1;
    w->item_clen[it] = 0;
    w->item_status[it] = PLO_ITEM_LIFTED;

    /* :153-157 */
    int need_flipped = 0;
    if (stages & PLO_STAGE_STRAND) {
        int read_segment_changes_strand_from_primary =
            ((in->read_is_reverse[read] != 0) == (in->seg_is_fwd_strand[seg] != 0));
        need_flipped = (!contig_is_fwd_strand) ^ read_segment_changes_strand_from_primary;
    }
    w->item_flip[it] = (uint8_t)need_flipped;

    /* :159-176 */
    int64_t cur_pos = in->seg_pos[seg];
    size_t cur_n = n_in;
    uint32_t *cur = (uint32_t *)malloc((n_in + 1) * sizeof(uint32_t)); /* .to_vec() */
    memcpy(cur, in_cig, n_in * sizeof(uint32_t));
    int status = PLO_ITEM_LIFTED;

    if ((stages & PLO_STAGE_STRAND) && !contig_is_fwd_strand) {
        int64_t contig_length = ix->contig_len[contig];
        int64_t read_segment_end = in->seg_pos[seg] + orc_cigar_ref_offset(in_cig, n_in);
        cur_pos = contig_length - read_segment_end;
        for (size_t i = 0; i < n_in; ++i) cur[i] = in_cig[n_in - 1 - i];
    }
    int do_shift = (stages & PLO_STAGE_LSHIFT) && (!(stages & PLO_STAGE_STRAND) || !contig_is_fwd_strand);
    if (do_shift) {
        const uint8_t *rev_contig_seq = ix->rev_contig_seq ? ix->rev_contig_seq[contig] : NULL;
        if (!rev_contig_seq) { /* .unwrap() on None, :174 */
            status = PLO_ITEM_PANIC;
        } else {
            uint8_t *read_seq = get_read_seq(in, read, need_flipped);
            uint32_t *o = (uint32_t *)malloc((2 * cur_n + 2) * sizeof(uint32_t));
            size_t no = 0;
            int64_t np = 0;
            int rc = orc_shift_indels(0, cur_pos, cur, cur_n, rev_contig_seq, ix->contig_len[contig], read_seq,
                                      in->read_seq_len[read], &np, o, &no);
            free(read_seq);
            free(cur);
            cur = o;
            cur_n = no;
            cur_pos = np;
            if (rc != 0) status = PLO_ITEM_PANIC;
        }
    }

    /* :179-183 */
    if (status == PLO_ITEM_LIFTED && (stages & PLO_STAGE_LIFTOVER)) {
        uint32_t *o = (uint32_t *)malloc((2 * (cur_n + map->nk) + 2) * sizeof(uint32_t));
        size_t no = 0;
        int64_t np = 0;
        int some = orc_liftover_read_alignment(map->keys, map->vals, map->nk, cur_pos, cur, cur_n, &np, o, &no);
        free(cur);
        cur = o;
        cur_n = no;
        cur_pos = np;
        if (!some) status = PLO_ITEM_NO_LIFTOVER;
    }

    /* :204-229 */
    if (status == PLO_ITEM_LIFTED && (stages & PLO_STAGE_LENCHECK)) {
        if ((uint64_t)in->read_seq_len[read] != orc_cigar_read_offset(cur, cur_n, 0)) status = PLO_ITEM_LEN_MISMATCH;
    }

    /* :232-243 */
    if (status == PLO_ITEM_LIFTED && (stages & PLO_STAGE_SIMPLIFY)) {
        uint32_t chrom_index = ix->seg_chrom_index[gseg];
        uint8_t *read_seq = get_read_seq(in, read, need_flipped);
        uint32_t *o = (uint32_t *)malloc((2 * cur_n + 2) * sizeof(uint32_t));
        size_t no = 0;
        int64_t np = 0;
        int rc = orc_simplify_alignment_indels(cur_pos, cur, cur_n, ix->chrom_seq[chrom_index], ix->chrom_len[chrom_index],
                                               read_seq, in->read_seq_len[read], &np, o, &no);
        free(read_seq);
        free(cur);
        cur = o;
        cur_n = no;
        cur_pos = np;
        if (rc != 0) status = PLO_ITEM_PANIC;
    }

    w->item_status[it] = (uint8_t)status;
    if (status == PLO_ITEM_LIFTED || status == PLO_ITEM_LEN_MISMATCH) {
        w->item_pos[it] = cur_pos;
        w->item_clen[it] = (uint32_t)cur_n;
        for (size_t i = 0; i < cur_n; ++i) cvec_push(&w->cigar, cur[i]);
    }
    free(cur);
}

/* get_contig_split_segments_from_read_mapping :80-103 + the loop at :456-471 */
static void run_segment(worker *w, uint32_t seg) {
    const plo_index_desc *ix = w->index;
    const plo_batch_in *in = w->in;
    uint32_t contig = in->seg_contig[seg];
    const uint32_t *cig = in->cigar + in->seg_cigar_off[seg];
    size_t n = in->seg_cigar_off[seg + 1] - in->seg_cigar_off[seg];
    int64_t r_start = in->seg_pos[seg];
    int64_t r_end = in->seg_pos[seg] + orc_cigar_ref_offset(cig, n);
    uint32_t c0 = ix->contig_seg_off[contig], c1 = ix->contig_seg_off[contig + 1];
    for (uint32_t g = c0; g < c1; ++g) {
        int64_t s_start = ix->seg_seq_order_start[g], s_end = ix->seg_seq_order_end[g];
        /* IntRange::intersect_range (lib/rust-vc-utils/src/int_range.rs:56-58): other.end >= self.start &&
           other.start < self.end with self = contig segment range, other = read range */
        if (r_end >= s_start && r_start < s_end) run_item(w, seg, g - c0);
    }
}

static void *worker_main(void *p) {
    worker *w = (worker *)p;
    if (w->in->item_seg) {
        for (uint32_t i = w->seg_begin; i < w->seg_end; ++i) run_item(w, w->in->item_seg[i], w->in->item_cseg[i]);
    } else {
        for (uint32_t s = w->seg_begin; s < w->seg_end; ++s) run_segment(w, s);
    }
    return NULL;
}

int orc_liftover_batch(const plo_index_desc *index, const plo_batch_in *in, uint32_t stages, int n_threads,
                       plo_batch_out *out) {
    memset(out, 0, sizeof(*out));
    /* index build: get_read_segment_to_ref_pos_tree_map per contig segment (contig_alignment_scanner/mod.rs:98-102) */
    seg_map *maps = (seg_map *)calloc(index->n_segments ? index->n_segments : 1, sizeof(seg_map));
    for (uint32_t g = 0; g < index->n_segments; ++g) {
        size_t n = index->seg_cigar_off[g + 1] - index->seg_cigar_off[g];
        maps[g].keys = (uint64_t *)malloc((2 * n + 2) * sizeof(uint64_t));
        maps[g].vals = (int64_t *)malloc((2 * n + 2) * sizeof(int64_t));
        maps[g].nk = orc_map_build(index->seg_pos[g], index->seg_cigar + index->seg_cigar_off[g], n, 0, maps[g].keys,
                                   maps[g].vals);
    }
    uint32_t total = in->item_seg ? in->n_items : in->n_segs;
    if (n_threads < 1) n_threads = 1;
    if ((uint32_t)n_threads > total && total > 0) n_threads = (int)total;
    worker *ws = (worker *)calloc((size_t)n_threads, sizeof(worker));
    pthread_t *th = (pthread_t *)calloc((size_t)n_threads, sizeof(pthread_t));
    for (int t = 0; t < n_threads; ++t) {
        ws[t].index = index;
        ws[t].in = in;
        ws[t].stages = stages;
        ws[t].maps = maps;
        ws[t].seg_begin = (uint32_t)((uint64_t)total * (uint64_t)t / (uint64_t)n_threads);
        ws[t].seg_end = (uint32_t)((uint64_t)total * (uint64_t)(t + 1) / (uint64_t)n_threads);
    }
    if (n_threads == 1) {
        worker_main(&ws[0]);
    } else {
        for (int t = 0; t < n_threads; ++t) pthread_create(&th[t], NULL, worker_main, &ws[t]);
        for (int t = 0; t < n_threads; ++t) pthread_join(th[t], NULL);
    }
    size_t n_items = 0, n_cigar = 0;
    for (int t = 0; t < n_threads; ++t) {
        n_items += ws[t].n_items;
        n_cigar += ws[t].cigar.n;
    }
    size_t a = n_items ? n_items : 1;
    uint32_t *o_seg = (uint32_t *)malloc(a * 4), *o_cseg = (uint32_t *)malloc(a * 4), *o_chrom = (uint32_t *)malloc(a * 4),
             *o_clen = (uint32_t *)malloc(a * 4);
    uint8_t *o_status = (uint8_t *)malloc(a), *o_flip = (uint8_t *)malloc(a), *o_mapq = (uint8_t *)malloc(a);
    int64_t *o_pos = (int64_t *)malloc(a * 8);
    uint64_t *o_coff = (uint64_t *)malloc(a * 8);
    uint32_t *o_cigar = (uint32_t *)malloc((n_cigar ? n_cigar : 1) * 4);
    size_t ii = 0, cc = 0;
    for (int t = 0; t < n_threads; ++t) {
        size_t local = 0;
        for (size_t i = 0; i < ws[t].n_items; ++i, ++ii) {
            o_seg[ii] = ws[t].item_seg[i];
            o_cseg[ii] = ws[t].item_cseg[i];
            o_chrom[ii] = ws[t].item_chrom[i];
            o_clen[ii] = ws[t].item_clen[i];
            o_status[ii] = ws[t].item_status[i];
            o_flip[ii] = ws[t].item_flip[i];
            o_mapq[ii] = ws[t].item_mapq[i];
            o_pos[ii] = ws[t].item_pos[i];
            o_coff[ii] = cc + local;
            local += ws[t].item_clen[i];
        }
        memcpy(o_cigar + cc, ws[t].cigar.v, ws[t].cigar.n * 4);
        cc += ws[t].cigar.n;
        free(ws[t].item_seg);
        free(ws[t].item_cseg);
        free(ws[t].item_chrom);
        free(ws[t].item_clen);
        free(ws[t].item_status);
        free(ws[t].item_flip);
        free(ws[t].item_mapq);
        free(ws[t].item_pos);
        cvec_free(&ws[t].cigar);
    }
    for (uint32_t g = 0; g < index->n_segments; ++g) {
        free(maps[g].keys);
        free(maps[g].vals);
    }
    free(maps);
    free(ws);
    free(th);
    out->n_items = (uint32_t)n_items;
    out->item_seg = o_seg;
    out->item_cseg = o_cseg;
    out->item_status = o_status;
    out->item_need_flipped = o_flip;
    out->item_mapq = o_mapq;
    out->item_chrom_index = o_chrom;
    out->item_ref_pos = o_pos;
    out->item_cigar_off = o_coff;
    out->item_cigar_len = o_clen;
    out->cigar = o_cigar;
    out->n_cigar = n_cigar;
    return 0;
}

void orc_batch_free(plo_batch_out *out) {
    free((void *)out->item_seg);
    free((void *)out->item_cseg);
    free((void *)out->item_status);
    free((void *)out->item_need_flipped);
    free((void *)out->item_mapq);
    free((void *)out->item_chrom_index);
    free((void *)out->item_ref_pos);
    free((void *)out->item_cigar_off);
    free((void *)out->item_cigar_len);
    free((void *)out->cigar);
    memset(out, 0, sizeof(*out));
}

/* ---- record finishing --------------------------------------------------------------------------------------- */

/* lib/rust-vc-utils/src/bam_utils/util.rs:10-27 hts_reg2bin, :33-35 bam_reg2bin */
static uint64_t hts_reg2bin(uint64_t begin, uint64_t end, unsigned min_shift, unsigned depth) {
    end = end - 1;
    unsigned l = depth, s = min_shift;
    uint64_t t = ((1ull << (depth * 3)) - 1) / 7;
    while (l > 0) {
        if ((begin >> s) == (end >> s)) return t + (begin >> s);
        l -= 1;
        s += 3;
        t -= 1ull << (l * 3);
    }
    return 0;
}
uint16_t orc_bam_reg2bin(uint64_t begin, uint64_t end) { return (uint16_t)hts_reg2bin(begin, end, 14, 5); }

/* rust-htslib 0.50.0 bam::Record::set(): ENCODE_BASE (htslib seq_nt16_table) -- only the letters comp_base can
 * produce from decoded bases matter here: A C G T N (third-party table, BAM specification 4.2.3) */
static uint8_t encode_base(uint8_t c) {
    switch (c) {
        case '=': return 0;
        case 'A': case 'a': return 1;
        case 'C': case 'c': return 2;
        case 'M': case 'm': return 3;
        case 'G': case 'g': return 4;
        case 'R': case 'r': return 5;
        case 'S': case 's': return 6;
        case 'V': case 'v': return 7;
        case 'T': case 't': return 8;
        case 'W': case 'w': return 9;
        case 'Y': case 'y': return 10;
        case 'H': case 'h': return 11;
        case 'K': case 'k': return 12;
        case 'D': case 'd': return 13;
        case 'B': case 'b': return 14;
        default: return 15;
    }
}

/* reverse_alignment_seq_and_qual (src/read_alignment_scanner.rs:125-133): seq().as_bytes(), rev_comp_in_place,
 * qual reversed, record.set() re-encodes */
static void reverse_seq_and_qual(const plo_batch_in *in, const plo_finish_in *fin, uint32_t read, uint8_t *dst_seq, uint8_t *dst_qual) {
    uint32_t len = in->read_seq_len[read];
    uint8_t *s = get_read_seq(in, read, 1);
    if (in->seq_fmt == PLO_SEQ_BAM4) {
        for (uint32_t i = 0; i < len; i += 2)
            dst_seq[i >> 1] = (uint8_t)((encode_base(s[i]) << 4) | (i + 1 < len ? encode_base(s[i + 1]) : 0));
    } else {
        memcpy(dst_seq, s, len);
    }
    free(s);
    const uint8_t *q = fin->qual + fin->read_qual_off[read];
    for (uint32_t i = 0; i < len; ++i) dst_qual[i] = q[len - 1 - i];
}

static uint64_t pad16(uint64_t x) { return (x + 15) & ~15ull; }

int orc_finish_batch(const plo_batch_in *in, const plo_finish_in *fin, const plo_batch_out *lift, plo_finish_out *out) {
    memset(out, 0, sizeof(*out));
    uint32_t ni = lift->n_items, nr = in->n_reads;
    size_t a = ni ? ni : 1, b = nr ? nr : 1;
    uint16_t *flag = calloc(a, 2), *bin = calloc(a, 2), *uflag = calloc(b, 2);
    int64_t *rend = calloc(a, 8);
    uint8_t *prim = calloc(a, 1);
    uint64_t *isoff = malloc(a * 8), *iqoff = malloc(a * 8), *rsoff = malloc(b * 8), *rqoff = malloc(b * 8);
    uint32_t *nl = calloc(b, 4), *pitem = malloc(b * 4);
    for (uint32_t r = 0; r < nr; ++r) pitem[r] = UINT32_MAX;
    uint64_t so = 0, qo = 0;
    /* pass 1: sizes / offsets in entry order (items, then reads) and per-item scalars */
    for (uint32_t i = 0; i < ni; ++i) {
        uint32_t r = in->seg_read[lift->item_seg[i]];
        isoff[i] = iqoff[i] = PLO_NO_FLIP;
        if (lift->item_status[i] != PLO_ITEM_LIFTED) continue;
        uint16_t f = fin->read_flags[r];                       /* clone_record :245 */
        if (lift->item_need_flipped[i]) f ^= 0x10;             /* :274-276 -> :126  */
        const uint32_t *cg = lift->cigar + lift->item_cigar_off[i];
        int64_t e = lift->item_ref_pos[i] + orc_cigar_ref_offset(cg, lift->item_cigar_len[i]); /* get_alignment_end :278 */
        rend[i] = e;
        bin[i] = orc_bam_reg2bin((uint64_t)lift->item_ref_pos[i], (uint64_t)e); /* :279 */
        f |= 0x800;                                            /* set_supplementary :282 */
        flag[i] = f;
        nl[r] += 1;
        if (pitem[r] == UINT32_MAX || lift->item_mapq[pitem[r]] < lift->item_mapq[i]) pitem[r] = i; /* :338-345 first max wins */
        if (lift->item_need_flipped[i]) {
            uint32_t len = in->read_seq_len[r];
            isoff[i] = so;
            iqoff[i] = qo;
            so += pad16(in->seq_fmt == PLO_SEQ_BAM4 ? (len + 1) / 2 : len);
            qo += pad16(len);
        }
    }
    for (uint32_t r = 0; r < nr; ++r) {
        rsoff[r] = rqoff[r] = PLO_NO_FLIP;
        if (nl[r] > 0) {
            prim[pitem[r]] = 1;
            flag[pitem[r]] &= (uint16_t)~0x800;                /* unset_supplementary :346 */
        } else {                                               /* unmapped copy :321-334 */
            uint16_t f = fin->read_flags[r];
            f |= 0x4;
            f &= (uint16_t)~0x800;
            if (f & 0x10) {                                    /* :330-332 */
                f ^= 0x10;
                uint32_t len = in->read_seq_len[r];
                rsoff[r] = so;
                rqoff[r] = qo;
                so += pad16(in->seq_fmt == PLO_SEQ_BAM4 ? (len + 1) / 2 : len);
                qo += pad16(len);
            }
            uflag[r] = f;
        }
    }
    uint8_t *rs = calloc(so ? so : 1, 1), *rq = calloc(qo ? qo : 1, 1);
    for (uint32_t i = 0; i < ni; ++i)
        if (isoff[i] != PLO_NO_FLIP) reverse_seq_and_qual(in, fin, in->seg_read[lift->item_seg[i]], rs + isoff[i], rq + iqoff[i]);
    for (uint32_t r = 0; r < nr; ++r)
        if (rsoff[r] != PLO_NO_FLIP) reverse_seq_and_qual(in, fin, r, rs + rsoff[r], rq + rqoff[r]);
    out->item_flag = flag;
    out->item_bin = bin;
    out->item_ref_end = rend;
    out->item_is_primary = prim;
    out->item_seq_off = isoff;
    out->item_qual_off = iqoff;
    out->read_n_lifted = nl;
    out->read_primary_item = pitem;
    out->read_unmapped_flag = uflag;
    out->read_seq_off = rsoff;
    out->read_qual_off = rqoff;
    out->rev_seq = rs;
    out->rev_qual = rq;
    out->rev_seq_bytes = so;
    out->rev_qual_bytes = qo;
    return 0;
}

void orc_finish_free(plo_finish_out *out) {
    free((void *)out->item_flag);
    free((void *)out->item_bin);
    free((void *)out->item_ref_end);
    free((void *)out->item_is_primary);
    free((void *)out->item_seq_off);
    free((void *)out->item_qual_off);
    free((void *)out->read_n_lifted);
    free((void *)out->read_primary_item);
    free((void *)out->read_unmapped_flag);
    free((void *)out->read_seq_off);
    free((void *)out->read_qual_off);
    free((void *)out->rev_seq);
    free((void *)out->rev_qual);
    memset(out, 0, sizeof(*out));
}

/* ---- SA tag text: src/read_alignment_scanner.rs:292-301, :348-364 ---------------------------------------------- */

typedef struct {
    char *p;
    size_t n, cap;
} sbuf;
static void sb_putf(sbuf *b, const char *fmt, ...) {
    va_list ap;
    for (;;) {
        va_start(ap, fmt);
        int w = vsnprintf(b->p + b->n, b->cap - b->n, fmt, ap);
        va_end(ap);
        if (w >= 0 && (size_t)w < b->cap - b->n) {
            b->n += (size_t)w;
            return;
        }
        b->cap = b->cap ? b->cap * 2 : 256;
        b->p = (char *)realloc(b->p, b->cap);
    }
}
/* :292-301 get_sa_tag_segment: "{chrom},{pos+1},{schar},{cigar},{mapq},0;" */
static void sa_tag_segment(sbuf *b, const char *chrom, int64_t pos, int is_reverse, const uint32_t *cig, uint32_t n, unsigned mapq) {
    static const char opc[] = "MIDNSHP=X";
    sb_putf(b, "%s,%lld,%c,", chrom, (long long)(pos + 1), is_reverse ? '-' : '+');
    for (uint32_t k = 0; k < n; ++k) sb_putf(b, "%u%c", (unsigned)CIG_LEN(cig[k]), CIG_OP(cig[k]) < 9 ? opc[CIG_OP(cig[k])] : '?');
    sb_putf(b, ",%u,0;", mapq);
}
int orc_sa_values(const plo_batch_in *in, const plo_batch_out *lift, const uint16_t *item_flag, const char *const *chrom_names,
                  char **values) {
    uint32_t ni = lift->n_items;
    for (uint32_t i = 0; i < ni; ++i) values[i] = NULL;
    uint32_t i0 = 0;
    while (i0 < ni) { /* the records of one read: consecutive items */
        uint32_t r = in->seg_read[lift->item_seg[i0]], i1 = i0;
        while (i1 < ni && in->seg_read[lift->item_seg[i1]] == r) ++i1;
        for (uint32_t i = i0; i < i1; ++i) { /* :352-364 */
            if (lift->item_status[i] != PLO_ITEM_LIFTED) continue;
            sbuf b = {NULL, 0, 0};
            for (uint32_t j = i0; j < i1; ++j) {
                if (j == i || lift->item_status[j] != PLO_ITEM_LIFTED) continue;
                sa_tag_segment(&b, chrom_names[lift->item_chrom_index[j]], lift->item_ref_pos[j], (item_flag[j] & 0x10) != 0,
                               lift->cigar + lift->item_cigar_off[j], lift->item_cigar_len[j], lift->item_mapq[j]);
            }
            if (b.n) values[i] = b.p; /* :360 !aux_str.is_empty() */
            else free(b.p);
        }
        i0 = i1;
    }
    return 0;
}
void orc_sa_free(char **values, uint32_t n_items) {
    for (uint32_t i = 0; i < n_items; ++i) free(values[i]);
}
