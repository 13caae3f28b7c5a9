// phase1.cpp -- construction of the contig->reference index from the assembly->reference BAM:
// scan_contig_bam and its helpers, host C++ (one-time set-up, O(#contigs); the block maps themselves are then built on
// the device by plo_index_create from the segments this produces).  Citations are relative to /root/reference.
//
//   scan_contig_bam / add_primary_read / add_split_read_cigar_to_supp_cigar_set      src/contig_alignment_scanner/mod.rs:91-183, 290-459
//   filter_non_targeted_segments                                                     .../non_targeted_segment_filter.rs:7-39
//   clip_repeated_contig_matches (+ get_seg_clip_info, clip_seg_isec_range, ...)      .../contig_repeated_match_trimmer.rs:18-303
//   join_colinear_contig_segments (+ are_segments_joinable, join_segments)           .../contig_colinear_segment_joiner.rs:14-186
//   clip_alignment_read_edges / clip_alignment_read_start                            lib/rust-vc-utils/src/bam_utils/cigar/clip_alignment.rs:104-181
//   get_gap_compressed_identity_no_align_match                                       lib/rust-vc-utils/src/bam_utils/cigar/score_alignment.rs:68-74,138-165
//   compress_cigar, strip_leading_clip, strip_trailing_clip                          lib/rust-vc-utils/src/bam_utils/cigar/mod.rs:204-228, 300-327
#include <map>
#include <tuple>

#include "bam_internal.hpp"

namespace {

using Cigar = std::vector<uint32_t>;
enum : uint32_t { M = 0, I = 1, D = 2, N = 3, S = 4, H = 5, P = 6, EQ = 7, X = 8 };
inline uint32_t mk(uint32_t t, uint64_t len) { return (uint32_t)(len << 4) | t; }
inline bool is_clip(uint32_t c) { return (c & 15u) == S || (c & 15u) == H; }  // cigar/mod.rs:16-18

uint64_t cigar_read_offset(const Cigar &c) {  // get_cigar_read_offset(cigar, false) :164-170
    uint64_t r = 0;
    for (uint32_t x : c) r += op_read_len(x);
    return r;
}
int64_t cigar_ref_offset(const Cigar &c) {  // :174-180
    int64_t r = 0;
    for (uint32_t x : c) r += op_ref_len(x);
    return r;
}

// compress_cigar (:204-228): zero-length ops dropped, equal neighbours merged (Pad is absent from the merge pattern:
// a Pad following a Pad keeps the first one's length)
Cigar compress_cigar(const Cigar &in) {
    Cigar out;
    for (uint32_t c : in) {
        if ((c >> 4) == 0) continue;
        if (!out.empty() && (out.back() & 15u) == (c & 15u)) {
            if ((c & 15u) != P) out.back() = mk(c & 15u, (uint64_t)(out.back() >> 4) + (c >> 4));
        } else {
            out.push_back(c);
        }
    }
    return out;
}

// clip_alignment_read_start (clip_alignment.rs:113-163)
void clip_alignment_read_start(const Cigar &in, uint64_t min_left_clip, Cigar &out, int64_t &left_ref_clip_shift) {
    out.clear();
    left_ref_clip_shift = 0;
    uint64_t read_pos = 0;
    for (uint32_t c : in) {
        const uint32_t t = c & 15u;
        const int64_t len = c >> 4;
        if (t == D || t == N) {
            if (read_pos <= min_left_clip) left_ref_clip_shift += len;
            else out.push_back(c);
        } else if (t == I) {
            out.push_back(read_pos < min_left_clip ? mk(S, (uint64_t)len) : c);
        } else if (t == M || t == X || t == EQ) {
            if (read_pos < min_left_clip) {
                int64_t remaining_clip = (int64_t)(min_left_clip - read_pos);
                int64_t match_size = std::max<int64_t>(len - remaining_clip, 0);
                int64_t clip_size = len - match_size;
                out.push_back(mk(S, (uint64_t)clip_size));
                if (match_size > 0) out.push_back(mk(t, (uint64_t)match_size));
                left_ref_clip_shift += clip_size;
            } else {
                out.push_back(c);
            }
        } else {
            out.push_back(c);
        }
        read_pos += op_read_len(c);
    }
}
// clip_alignment_read_edges (:166-181)
void clip_alignment_read_edges(const Cigar &in, uint64_t min_left_clip, uint64_t min_right_clip, Cigar &out, int64_t &ref_shift) {
    Cigar rev(in.rbegin(), in.rend()), right;
    int64_t ignore;
    clip_alignment_read_start(rev, min_right_clip, right, ignore);
    std::reverse(right.begin(), right.end());
    Cigar clipped;
    clip_alignment_read_start(right, min_left_clip, clipped, ref_shift);
    out = compress_cigar(clipped);
}

// get_gap_compressed_identity_no_align_match (score_alignment.rs:138-165); false when the CIGAR has an M op
bool gap_compressed_identity(const Cigar &c, double &gci) {
    uint32_t mismatch_events = 0, match_bases = 0;
    for (uint32_t x : c) {
        const uint32_t t = x & 15u, len = x >> 4;
        if (t == I || t == D || t == N) mismatch_events += 1;
        else if (t == X) mismatch_events += len;
        else if (t == EQ) match_bases += len;
        else if (t == M) return false;
    }
    gci = (match_bases + mismatch_events) == 0 ? 1.0 : (double)match_bases / (double)(match_bases + mismatch_events);  // :68-74
    return true;
}

struct Seg {  // ContigMappingSegmentInfo's seq_order_segment (the block map is derived from pos + cigar on the device)
    uint64_t so_start = 0, so_end = 0;
    uint32_t chrom = 0;
    int64_t pos = 0;
    bool fwd = true;
    uint8_t mapq = 0;
    bool primary = false;
    // target-region runs: an SA-tag segment whose supplementary record was never seen keeps the tag's approximate CIGAR but gets NO
    // block map (contig_to_ref_map stays empty, mod.rs:396-414) -- until the trimmer or the joiner rebuild the map from the CIGAR
    // (contig_repeated_match_trimmer.rs:130-134, contig_colinear_segment_joiner.rs:62-121)
    bool no_map = false;
    Cigar cigar;
};
struct Range {
    int64_t start, end;
    Range reverse(int64_t size) const { return Range{size - end, size - start}; }  // int_range.rs:89-94
};
using SplitReadKey = std::tuple<uint32_t, int64_t, bool, uint32_t, uint32_t>;  // mod.rs:49-56 (derive(Ord): field order)
struct Contig {
    bool have_primary = false;
    std::string qname;
    std::vector<Seg> segs;
    std::vector<uint8_t> rev_seq;
    bool have_rev_seq = false;
    std::map<SplitReadKey, Cigar> supp;
};

SplitReadKey key_of(uint32_t chrom, int64_t pos, bool fwd, const Cigar &c) {
    uint64_t rs, re, size;
    read_clip_positions(c.data(), c.size(), rs, re, size);
    return SplitReadKey{chrom, pos, fwd, (uint32_t)rs, (uint32_t)(size - re)};
}

// get_seg_gap_compressed_identity (trimmer.rs:18-50)
plo_status seg_gci(const std::string &qname, const Seg &seg, const Range &isec_so, double &gci, std::string &err) {
    const uint64_t read_len = cigar_read_offset(seg.cigar);
    Range r = seg.fwd ? isec_so : isec_so.reverse((int64_t)read_len);
    Cigar clipped;
    int64_t shift;
    clip_alignment_read_edges(seg.cigar, (uint64_t)r.start, read_len - (uint64_t)r.end, clipped, shift);
    if (!gap_compressed_identity(clipped, gci)) {
        err = "Error generating gap-compressed identity for overlapping split read segment in assembly contig '" + qname +
              "': Method assumes alignment CIGAR strings use seq match/mismatch (=/X) instead of alignment match (M)";
        return PLO_ERR_DATA;
    }
    return PLO_OK;
}

// clip_seg_isec_range (:55-115): true when the split read is eliminated entirely
bool clip_seg_isec_range(Seg &seg, const Range &isec_so) {
    const bool clipping_so_prefix = isec_so.start == (int64_t)seg.so_start;
    const bool clipping_prefix = clipping_so_prefix ^ (!seg.fwd);
    const uint64_t read_len = cigar_read_offset(seg.cigar);
    Range r = seg.fwd ? isec_so : isec_so.reverse((int64_t)read_len);
    uint64_t min_left = 0, min_right = 0;
    if (clipping_prefix) min_left = (uint64_t)r.end;
    else min_right = read_len - (uint64_t)r.start;
    Cigar shifted;
    int64_t ref_pos_shift;
    clip_alignment_read_edges(seg.cigar, min_left, min_right, shifted, ref_pos_shift);
    seg.cigar = shifted;
    seg.pos += ref_pos_shift;
    uint64_t left_read_pos, right_read_pos, size;
    read_clip_positions(seg.cigar.data(), seg.cigar.size(), left_read_pos, right_read_pos, size);
    if (left_read_pos >= right_read_pos) return true;
    if (clipping_prefix) r.end = (int64_t)left_read_pos;
    else r.start = (int64_t)right_read_pos;
    Range so = seg.fwd ? r : r.reverse((int64_t)read_len);
    if (clipping_so_prefix) seg.so_start = (uint64_t)so.end;
    else seg.so_end = (uint64_t)so.start;
    return false;
}

}  // namespace

struct plo_phase1 {
    std::vector<Contig> contigs;
    uint32_t segments_clipped = 0, segments_joined = 0, n_records = 0;
    // flattened plo_index_desc arrays
    std::vector<int64_t> contig_len, seg_pos, seg_so_start, seg_so_end;
    std::vector<uint32_t> contig_seg_off, seg_chrom, seg_cigar_off, seg_cigar;
    std::vector<uint8_t> seg_fwd, seg_mapq;
    std::vector<const uint8_t *> rev_ptrs;
    std::vector<std::string> ref_names;
    std::vector<const char *> ref_name_ptrs;
    std::vector<uint32_t> ref_lens;
};

extern "C" {

plo_status plo_phase1_scan(const char *asm_to_ref_bam, uint32_t n_contigs, const char *const *contig_names, const int64_t *contig_lens,
                           const plo_target_region *target_region, int n_threads, plo_phase1 **out) {
    if (!asm_to_ref_bam || !out || (n_contigs && (!contig_names || !contig_lens))) return PLO_ERR_INVALID_ARG;
    *out = nullptr;
    BgzfIn in;
    plo_status st = in.open(asm_to_ref_bam, n_threads);
    if (st != PLO_OK) {
        in.close();
        return st;
    }
    plo_phase1 *ph = new plo_phase1();
    auto bail = [&](plo_status s) {
        in.close();
        delete ph;
        return s;
    };
    // header: the reference ChromList comes from this BAM (src/main.rs, ChromList::from_bam_filename)
    uint8_t hd[8], w4[4];
    if ((st = in.read(hd, 8)) != PLO_OK) return bail(st);
    if (memcmp(hd, "BAM\1", 4) != 0) return bail(fail(PLO_ERR_IO, "not a BAM file (bad magic)"));
    uint32_t l_text = rd32(hd + 4);
    std::string text(l_text, '\0');
    if (l_text && (st = in.read(&text[0], l_text)) != PLO_OK) return bail(st);
    if ((st = in.read(w4, 4)) != PLO_OK) return bail(st);
    const uint32_t n_ref = rd32(w4);
    std::unordered_map<std::string, uint32_t> ref_index, contig_index;
    for (uint32_t i = 0; i < n_ref; ++i) {
        if ((st = in.read(w4, 4)) != PLO_OK) return bail(st);
        uint32_t l_name = rd32(w4);
        std::string name(l_name, '\0');
        if (l_name && (st = in.read(&name[0], l_name)) != PLO_OK) return bail(st);
        while (!name.empty() && name.back() == '\0') name.pop_back();
        if ((st = in.read(w4, 4)) != PLO_OK) return bail(st);
        ref_index[name] = i;
        ph->ref_names.push_back(name);
        ph->ref_lens.push_back(rd32(w4));
    }
    for (uint32_t c = 0; c < n_contigs; ++c) contig_index[contig_names[c]] = c;
    ph->contigs.resize(n_contigs);

    // ---- the scan (mod.rs:185-243; one pass over the file instead of per-window index fetches) ----
    std::vector<uint32_t> pcig;
    std::vector<SaSeg> sas;
    for (;;) {
        if ((st = in.fill(4)) != PLO_OK) return bail(st);
        if (in.avail() == 0) break;
        if (in.avail() < 4) return bail(fail(PLO_ERR_IO, "truncated BAM record"));
        uint32_t bs = rd32(in.buf.data() + in.bpos);
        if (bs < 32) return bail(fail(PLO_ERR_IO, "BAM record shorter than its fixed fields"));
        if ((st = in.fill(4 + (size_t)bs)) != PLO_OK) return bail(st);
        if (in.avail() < 4 + (size_t)bs) return bail(fail(PLO_ERR_IO, "truncated BAM record"));
        Rec rec{in.buf.data() + in.bpos + 4, bs};
        in.bpos += 4 + (size_t)bs;
        if (!rec.layout_ok()) return bail(fail(PLO_ERR_IO, "BAM record fields exceed its block_size"));
        ++ph->n_records;
        const uint16_t flag = rec.flag();
        if ((flag & 0x4) || (flag & 0x100)) continue;  // unmapped / secondary (:213-215)
        std::string qname((const char *)rec.qname());
        auto it = contig_index.find(qname);
        if (it == contig_index.end())
            return bail(fail(PLO_ERR_DATA, "contig '" + qname + "' of the assembly->reference BAM is not in the read->contig BAM header"));  // :224-225 (HashMap index panics)
        Contig &ct = ph->contigs[it->second];
        if (!(flag & 0x800)) {  // add_primary_read (:91-133)
            std::string err;
            if ((st = split_segments(ref_index, rec, sas, pcig, err)) != PLO_OK) return bail(fail(st, err));
            ct.segs.clear();
            bool need_rev = false;
            for (SaSeg &g : sas) {
                Seg s;
                s.so_start = g.so_start;
                s.so_end = g.so_end;
                s.chrom = g.contig;  // (chrom index in the reference list)
                s.pos = g.pos;
                s.fwd = g.fwd;
                s.mapq = g.mapq;
                s.primary = g.primary;
                s.cigar = g.primary ? pcig : g.cigar;
                need_rev |= !s.fwd;
                ct.segs.push_back(std::move(s));
            }
            ct.have_rev_seq = need_rev;
            ct.rev_seq.clear();
            if (need_rev) {  // :113-125: record.seq().as_bytes(), reverse-complemented unless the record is on the reverse strand
                const uint32_t n = rec.l_seq();
                ct.rev_seq.resize(n);
                const uint8_t *sq = rec.seq();
                static const char dec[] = "=ACMGRSVTWYHKDBN";
                const bool rc = !(flag & 0x10);
                for (uint32_t j = 0; j < n; ++j) {
                    uint8_t code = (j & 1) ? (sq[j >> 1] & 15) : (sq[j >> 1] >> 4);
                    uint8_t b = (uint8_t)dec[code];
                    if (rc) {
                        uint8_t cb;  // comp_base, seq_util.rs:1-15 (decoded bases are upper case)
                        switch (b) {
                            case 'A': cb = 'T'; break;
                            case 'T': cb = 'A'; break;
                            case 'C': cb = 'G'; break;
                            case 'G': cb = 'C'; break;
                            default: cb = 'N';
                        }
                        ct.rev_seq[n - 1 - j] = cb;
                    } else {
                        ct.rev_seq[j] = b;
                    }
                }
            }
            ct.qname = qname;
            ct.have_primary = true;
        } else {  // add_split_read_cigar_to_supp_cigar_set (:135-183)
            Cigar cg;
            real_cigar(rec, cg);  // (a long CIGAR comes in the CG tag, as for the primary record)
            SplitReadKey k = key_of((uint32_t)rec.tid(), rec.pos(), !(flag & 0x10), cg);
            if (!ct.supp.emplace(k, std::move(cg)).second)
                return bail(fail(PLO_ERR_DATA, "Can't uniquely identify split read alignment info in contig '" + qname + "'"));
        }
    }
    in.close();

    // ---- supplementary CIGARs: the SA tag's are approximate, the records' own replace them (:371-416) ----
    for (uint32_t c = 0; c < n_contigs; ++c) {
        Contig &ct = ph->contigs[c];
        for (Seg &s : ct.segs) {
            if (s.primary) continue;
            auto it = ct.supp.find(key_of(s.chrom, s.pos, s.fwd, s.cigar));
            if (it != ct.supp.end()) {
                s.cigar = it->second;
            } else if (target_region) {
                s.no_map = true;
            } else {
                delete ph;
                return fail(PLO_ERR_DATA, std::string("Can't find supplementary alignment record corresponding to segment reported in SA tag for contig '") +
                                              contig_names[c] + "'");
            }
        }
        ct.supp.clear();
    }
    // ---- filter_non_targeted_segments ----
    if (target_region) {
        for (Contig &ct : ph->contigs) {
            std::vector<Seg> keep;
            for (Seg &s : ct.segs) {
                // GenomeSegment::intersect of the target with [pos, pos + 1): other.end >= self.start && other.start < self.end
                if (s.chrom == target_region->chrom_index && s.pos + 1 >= target_region->start && s.pos < target_region->end) keep.push_back(std::move(s));
            }
            ct.segs.swap(keep);
        }
    }
    // ---- clip_repeated_contig_matches (trimmer.rs:214-303) ----
    for (Contig &ct : ph->contigs) {
        const size_t n = ct.segs.size();
        if (!n) continue;
        std::vector<char> eliminated(n, 0);
        for (size_t i1 = 0; i1 < n; ++i1) {
            for (size_t i2 = i1 + 1; i2 < n; ++i2) {
                if (eliminated[i1] || eliminated[i2]) continue;
                Seg &s1 = ct.segs[i1], &s2 = ct.segs[i2];
                if (s1.so_end <= s2.so_start) break;  // get_seg_clip_info -> None -> break (:160-162, :241-245)
                Range isec{(int64_t)s2.so_start, (int64_t)s1.so_end};
                double g1, g2;
                std::string err;
                if ((st = seg_gci(ct.qname, s1, isec, g1, err)) != PLO_OK || (st = seg_gci(ct.qname, s2, isec, g2, err)) != PLO_OK) {
                    delete ph;
                    return fail(st, err);
                }
                // seg2_gci.partial_cmp(&seg1_gci).then(seg2.mapq.cmp(&seg1.mapq)) == Greater (:188-194)
                const bool clip_seg1 = g2 > g1 || (g2 == g1 && s2.mapq > s1.mapq);
                const size_t ci = clip_seg1 ? i1 : i2;
                if (clip_seg_isec_range(ct.segs[ci], isec)) eliminated[ci] = 1;
                else ct.segs[ci].no_map = false;  // clip_seg_info_isec_range rebuilds the map of a segment it has clipped (:130-134)
                ++ph->segments_clipped;
            }
        }
        std::vector<Seg> keep;
        for (size_t i = 0; i < n; ++i)
            if (!eliminated[i]) keep.push_back(std::move(ct.segs[i]));
        ct.segs.swap(keep);
    }
    // ---- join_colinear_contig_segments (joiner.rs:124-186) ----
    for (uint32_t c = 0; c < n_contigs; ++c) {
        Contig &ct = ph->contigs[c];
        if (ct.segs.empty()) continue;
        std::vector<Seg> old;
        old.swap(ct.segs);
        for (Seg &seg : old) {
            if (ct.segs.empty()) {
                ct.segs.push_back(std::move(seg));
                continue;
            }
            Seg &last = ct.segs.back();
            if (seg.so_start < last.so_end) {
                delete ph;
                return fail(PLO_ERR_DATA, "Incomplete repeat trimming on qname: " + ct.qname);  // assert :149-156
            }
            auto ref_gap = [](const Seg &a, const Seg &b) -> int64_t {  // get_seg_ref_gap :14-22
                if (a.fwd) return b.pos - (a.pos + cigar_ref_offset(a.cigar));
                return a.pos - (b.pos + cigar_ref_offset(b.cigar));
            };
            bool joinable = false;  // are_segments_joinable :26-50
            if (last.chrom == seg.chrom && last.fwd == seg.fwd) {
                int64_t gap = ref_gap(last, seg);
                joinable = gap >= 0 && gap <= 1000 && last.mapq == seg.mapq;
            }
            if (!joinable) {
                ct.segs.push_back(std::move(seg));
                continue;
            }
            // join_segments :59-122
            const int64_t del = ref_gap(last, seg);
            const uint64_t ins = seg.so_start - last.so_end;
            auto join_cigars = [&](Cigar &a, Cigar &b) {  // :79-95
                {   // strip_trailing_clip(a) (cigar/mod.rs:315-327): every clip after the first non-clip op goes
                    Cigar t;
                    bool non_clip = false;
                    for (uint32_t x : a) {
                        if (non_clip) {
                            if (!is_clip(x)) t.push_back(x);
                        } else {
                            if (!is_clip(x)) non_clip = true;
                            t.push_back(x);
                        }
                    }
                    a.swap(t);
                }
                if (ins > 0) a.push_back(mk(I, ins));
                if (del > 0) a.push_back(mk(D, (uint64_t)del));
                bool non_clip = false;  // strip_leading_clip(b) (:300-312)
                for (uint32_t x : b) {
                    if (!non_clip && is_clip(x)) continue;
                    non_clip = true;
                    a.push_back(x);
                }
            };
            if (last.fwd) {
                join_cigars(last.cigar, seg.cigar);
            } else {
                join_cigars(seg.cigar, last.cigar);
                last.cigar.swap(seg.cigar);
                last.pos = seg.pos;
            }
            last.so_end = seg.so_end;
            last.no_map = false;  // join_segments rebuilds the map from the joined CIGAR (:62-121)
            ++ph->segments_joined;
        }
    }
    // ---- flatten into the arrays of plo_index_desc ----
    ph->contig_seg_off.push_back(0);
    ph->seg_cigar_off.push_back(0);
    for (uint32_t c = 0; c < n_contigs; ++c) {
        const Contig &ct = ph->contigs[c];
        ph->contig_len.push_back(contig_lens[c]);
        for (const Seg &s : ct.segs) {
            ph->seg_chrom.push_back(s.chrom);
            ph->seg_pos.push_back(s.pos);
            ph->seg_fwd.push_back(s.fwd ? 1 : 0);
            ph->seg_mapq.push_back(s.mapq);
            ph->seg_so_start.push_back((int64_t)s.so_start);
            ph->seg_so_end.push_back((int64_t)s.so_end);
            if (!s.no_map) ph->seg_cigar.insert(ph->seg_cigar.end(), s.cigar.begin(), s.cigar.end());  // (no CIGAR: an empty block map)
            ph->seg_cigar_off.push_back((uint32_t)ph->seg_cigar.size());
        }
        ph->contig_seg_off.push_back((uint32_t)ph->seg_chrom.size());
        ph->rev_ptrs.push_back(ct.have_rev_seq ? ct.rev_seq.data() : nullptr);
    }
    for (auto &nme : ph->ref_names) ph->ref_name_ptrs.push_back(nme.c_str());
    *out = ph;
    return PLO_OK;
}

plo_status plo_phase1_index_desc(const plo_phase1 *ph, plo_index_desc *d) {
    if (!ph || !d) return PLO_ERR_INVALID_ARG;
    memset(d, 0, sizeof(*d));
    d->n_contigs = (uint32_t)ph->contig_len.size();
    d->contig_len = ph->contig_len.data();
    d->contig_seg_off = ph->contig_seg_off.data();
    d->n_segments = (uint32_t)ph->seg_chrom.size();
    d->seg_chrom_index = ph->seg_chrom.data();
    d->seg_pos = ph->seg_pos.data();
    d->seg_is_fwd_strand = ph->seg_fwd.data();
    d->seg_mapq = ph->seg_mapq.data();
    d->seg_seq_order_start = ph->seg_so_start.data();
    d->seg_seq_order_end = ph->seg_so_end.data();
    d->seg_cigar_off = ph->seg_cigar_off.data();
    d->seg_cigar = ph->seg_cigar.data();
    d->n_chroms = (uint32_t)ph->ref_names.size();
    d->rev_contig_seq = ph->rev_ptrs.data();
    d->seq_mem = PLO_MEM_HOST;
    return PLO_OK;
}

plo_status plo_phase1_info(const plo_phase1 *ph, uint32_t *n_ref, const char *const **ref_names, const uint32_t **ref_lens, uint32_t *segments_clipped,
                           uint32_t *segments_joined, uint32_t *n_records) {
    if (!ph) return PLO_ERR_INVALID_ARG;
    if (n_ref) *n_ref = (uint32_t)ph->ref_names.size();
    if (ref_names) *ref_names = ph->ref_name_ptrs.data();
    if (ref_lens) *ref_lens = ph->ref_lens.data();
    if (segments_clipped) *segments_clipped = ph->segments_clipped;
    if (segments_joined) *segments_joined = ph->segments_joined;
    if (n_records) *n_records = ph->n_records;
    return PLO_OK;
}

void plo_phase1_free(plo_phase1 *ph) { delete ph; }

}  // extern "C"
