// inflate.hpp -- DEFLATE (RFC 1951) decoder for one BGZF block, written for ONE WAVEFRONT per block (k_bgzf_inflate,
// engine.hip) and compiled for the host as well (tests, tests/emu).  A BGZF file is a series of independent deflate streams
// of at most 64 KiB each, so a BAM of N GB is N x 16 k independent decode jobs: the host cores of a GPU node inflate
// ~0.4 GB/s each, the device takes every block of a 256 MB chunk at once.
//
// Decoding a Huffman stream is sequential, so all lanes of the wave run the SAME decode on the same bits -- uniform control
// flow, table look-ups that are LDS broadcasts, the decode state pinned to scalar registers -- and the wave spreads the data
// movement and, for runs of literals, the look-ups (InfWaveIO):
//   * the compressed bytes pass through a 2 KiB LDS window filled 1 KiB at a time with coalesced loads (first version: every
//     refill of the bit buffer was a dependent global load);
//   * the output goes to an 8 KiB LDS ring that is written back 4 KiB at a time with coalesced stores; matches whose source is
//     still in the ring (distance <= 7 680) are LDS-to-LDS copies by up to 64 lanes at once; the far ones (read names and tags
//     of the previous record) read the bytes this wave wrote back earlier from global memory;
//   * runs of literals -- most symbols of a BAM block are base qualities -- are decoded 64 bit offsets at a time: lane i looks
//     up the code that would start i bits after the current position, and the scalar unit follows the chain of code lengths
//     through the lanes (literal_run).  A lone wave issues an instruction every four cycles at best and waits ~130 cycles for
//     a dependent LDS look-up: ~300 cycles per symbol for a serial loop, ~160 (at most ~40 instructions) this way;
//   * 16 KiB of LDS per block in all (tables 5.8 KiB): ten blocks in flight per CU.
//   * a match whose codes are in the fast tables is taken inside that loop (one scalar look-up for the distance, the copy by the
//     lanes); the symbol-by-symbol general path is left with long codes, block ends and write-back boundaries.
// Measured on MI355X on a level-1 BAM: 5.5 GB/s first version, 6.6 GB/s with the LDS windows and scalar state, 15 GB/s with the
// lane-parallel literal runs, 20 GB/s with matches in the run and one wave per SIMD (16 host cores with libdeflate: 7 GB/s).
// Codes of up to 10 bits (nearly all) cost one table look-up; longer ones fall back to canonical decoding by code length.
// Input and output are bounds-checked: the decoder returns 0 or a negative error code and never reads or writes outside
// [in, in + in_len) / [out, out + out_len).
#pragma once
#include <stdint.h>

#ifndef PLO_HD
#define PLO_HD inline
#endif

namespace plo {

enum { INF_OK = 0, INF_ERR_INPUT = -1, INF_ERR_OUTPUT = -2, INF_ERR_BTYPE = -3, INF_ERR_STORED = -4, INF_ERR_TABLE = -5, INF_ERR_SYMBOL = -6,
       INF_ERR_DISTANCE = -7, INF_ERR_LENGTH = -8 };
constexpr int INF_FAST_BITS = 10;

// per-block workspace: LDS of the wave on the device, a stack object on the host
struct InfWork {
    uint16_t lfast[1 << INF_FAST_BITS];  // (symbol << 4) | code length for codes of up to INF_FAST_BITS bits, 0 = longer code
    uint16_t dfast[1 << INF_FAST_BITS];
    uint16_t lcount[16], lsym[288];      // canonical tables (codes by length) for the slow path
    uint16_t dcount[16], dsym[32];
    uint16_t code[288];                  // canonical code of every symbol while a fast table is filled
    uint16_t offs[16], next[16];         // running offsets / next codes by length while tables are built (indexed arrays in LDS:
                                         // as locals they would be scratch, i.e. global memory)
    uint8_t lengths[320];
};

// decoder state of the bit reader (64-bit buffer refilled with aligned 32-bit words, see inf_refill)
struct InfBits {
    uint32_t in_len, ipos;  // ipos: byte position of the next word to fetch (a multiple of 4)
    unsigned long long buf;
    int cnt;  // valid bits in buf
};

// ---- I/O policies ---------------------------------------------------------------------------------------------------------------
// The decoder reads its input as aligned 32-bit words (in_word) and hands its output over as literals, matches and stored
// runs; where those bytes live is the policy's business.
//
// host execution: one "lane", input and output used in place
struct InfSerial {
    const uint8_t *in = nullptr;
    uint8_t *out = nullptr;
    uint32_t in_len = 0;
    PLO_HD int lane() const { return 0; }
    PLO_HD int width() const { return 1; }
    PLO_HD void sync() const {}
    PLO_HD uint32_t uniform(uint32_t v) const { return v; }
    PLO_HD uint32_t scalar(uint32_t v) const { return v; }
    PLO_HD void pin() {}
    PLO_HD void begin(const uint8_t *in_, uint32_t in_len_, uint8_t *out_, uint32_t) {
        in = in_;
        in_len = in_len_;
        out = out_;
    }
    PLO_HD uint32_t in_word(uint32_t p) {  // bytes p .. p+3 of the input, little endian, zero beyond its end
        uint32_t v = 0;
        for (uint32_t k = 0; k < 4; ++k)
            if (p + k < in_len) v |= (uint32_t)in[p + k] << (8 * k);
        return v;
    }
    PLO_HD void put_literal(uint32_t pos, uint32_t v) { out[pos] = (uint8_t)v; }
    PLO_HD void put_match(uint32_t pos, uint32_t dist, uint32_t len) {
        for (uint32_t k = 0; k < len; ++k) out[pos + k] = out[pos + k - dist];
    }
    PLO_HD void put_stored(uint32_t pos, uint32_t in_pos, uint32_t len) {
        for (uint32_t k = 0; k < len; ++k) out[pos + k] = in[in_pos + k];
    }
    PLO_HD void advance(uint32_t) {}
    PLO_HD void end(uint32_t) {}
    PLO_HD long long literal_run(unsigned long long, uint32_t &, uint32_t, const uint16_t *, const uint16_t *) { return -1; }
};

// LDS of one wave's I/O (besides InfWork)
#ifndef PLO_INF_RING
#define PLO_INF_RING 8192
#endif
constexpr uint32_t INF_RING = PLO_INF_RING, INF_CHUNK = INF_RING / 2, INF_NEAR = INF_RING - 512;  // output ring, write-back unit, largest ring match
constexpr uint32_t INF_IN_RING = 2048, INF_IN_CHUNK = 1024;
struct InfWaveMem {
    alignas(16) uint8_t iring[INF_IN_RING];
    alignas(16) uint8_t oring[INF_RING];
};

// One wave per block.  `Prim` supplies the wave primitives: lane(), sync() (LDS and the wave's own global stores ordered
// for all its lanes), uniform(v) (value of the first lane), load_written(p) (a byte this wave stored to global memory earlier
// and has waited for with sync(): must not come from a stale line of the CU's vector cache), and scalar(v): v is the same in
// every lane and the device is told so (v_readfirstlane).  Everything the decoder computes from the words of the input and
// the table entries is wave-uniform; with those two kinds of LDS reads marked scalar, the whole decode state lives in scalar
// registers and runs on the scalar unit -- a lone wave issues a vector instruction every four cycles at best, and the
// per-symbol path was ~150 of them (measured: 800 cycles per symbol before).
template <class Prim>
struct InfWaveIO {
    Prim prim;
    InfWaveMem *m = nullptr;
    const uint8_t *in = nullptr;
    uint8_t *out = nullptr;
    uint32_t in_len = 0, in_hi = 0;  // input bytes [in_hi - INF_IN_RING, in_hi) are in the window (as far as they exist)
    uint32_t flushed = 0;            // output bytes [0, flushed) are in global memory
    uint32_t ring_from = 0;          // output bytes [ring_from, pos) are in the ring (unless older than INF_RING)
    uint32_t skip_runs = 0;          // literal_run attempts to skip (adaptive, see there)
#ifdef PLO_INF_TIMING
    long long t_load = 0, t_flush = 0, t_match = 0, t_far = 0, t_table = 0, t_run = 0;
    int n_load = 0, n_flush = 0, n_match = 0, n_far = 0, n_table = 0, n_run = 0;
#define INF_T0 long long t0_ = prim.clock();
#define INF_T1(acc, cnt) acc += prim.clock() - t0_; ++cnt;
#else
#define INF_T0
#define INF_T1(acc, cnt)
#endif

    PLO_HD int lane() const { return prim.lane(); }
    PLO_HD int width() const { return 64; }
    PLO_HD void sync() const { prim.sync(); }
    PLO_HD uint32_t uniform(uint32_t v) const { return prim.uniform(v); }
    PLO_HD uint32_t scalar(uint32_t v) const { return prim.scalar(v); }
    PLO_HD void pin() {  // (see inf_pin)
        in_hi = prim.scalar(in_hi);
        flushed = prim.scalar(flushed);
        ring_from = prim.scalar(ring_from);
    }
    PLO_HD void begin(const uint8_t *in_, uint32_t in_len_, uint8_t *out_, uint32_t) {
        in = in_;
        in_len = in_len_;
        out = out_;
        in_hi = 0;
        flushed = 0;
        ring_from = 0;
        skip_runs = 0;
    }
    PLO_HD void load_chunk() {  // the next INF_IN_CHUNK input bytes into the window: 16 coalesced byte loads per lane
        INF_T0
        const uint32_t l = (uint32_t)prim.lane();
        prim.sync();  // (the lanes are done reading the half that is overwritten)
        for (uint32_t j = 0; j < INF_IN_CHUNK / 64; ++j) {
            const uint32_t p = in_hi + j * 64 + l;
            m->iring[p & (INF_IN_RING - 1)] = p < in_len ? in[p] : (uint8_t)0;
        }
        in_hi += INF_IN_CHUNK;
        prim.sync();
        INF_T1(t_load, n_load)
    }
    PLO_HD uint32_t in_word(uint32_t p) {  // p is a multiple of 4 and never goes back by more than a word
        if (p >= in_len) return 0;
        if (p >= in_hi + INF_IN_CHUNK) in_hi = p & ~(INF_IN_CHUNK - 1);  // (a stored run was skipped: restart the window there)
        while (in_hi < in_len && p + 512 > in_hi) load_chunk();
        const uint32_t *r = (const uint32_t *)(m->iring + (p & (INF_IN_RING - 1)));  // (the window is 4-byte aligned, p a multiple of 4)
        return prim.scalar(*r);
    }
    PLO_HD void put_literal(uint32_t pos, uint32_t v) {
        if (prim.lane() == 0) m->oring[pos & (INF_RING - 1)] = (uint8_t)v;
    }
    // Runs of literals (most symbols of a BAM block: base qualities), SIXTY-FOUR BIT OFFSETS AT A TIME.  A serial decoder pays
    // one dependent LDS round trip (~130 cycles, ~300 with the loop around it) per symbol.  Here lane i looks up the code that
    // would start i bits after the current position -- straight from the input window, no bit buffer: two LDS round trips for
    // all 64 offsets -- and the scalar unit then only follows the chain of code lengths through the lanes (v_readlane, no memory):
    // ~8 literals per pair of round trips.  The lanes at which symbols start store them in one instruction, each at its rank
    // (v_mbcnt over the start mask).  Returns the bit position reached (the first symbol that is not a fast-table literal, or a
    // boundary: end of the loaded input window, the last byte before a write-back boundary of the ring, the end of the output),
    // or -1 without having touched anything.
    PLO_HD unsigned long long peek64(unsigned long long bitpos) {  // 64 input bits from a bit position inside the window (scalar)
        const uint32_t *ring32 = (const uint32_t *)m->iring;
        const uint32_t w = (uint32_t)(bitpos >> 5), sh = (uint32_t)(bitpos & 31u);
        const uint32_t a = prim.scalar(ring32[w & (INF_IN_RING / 4 - 1)]), b = prim.scalar(ring32[(w + 1) & (INF_IN_RING / 4 - 1)]);
        const uint32_t c = prim.scalar(ring32[(w + 2) & (INF_IN_RING / 4 - 1)]);
        const unsigned long long lo = ((unsigned long long)b << 32) | a;
        return sh ? (lo >> sh) | ((unsigned long long)c << (64 - sh)) : lo;
    }
    PLO_HD long long literal_run(unsigned long long bit0, uint32_t &pos, uint32_t out_len, const uint16_t *lfast, const uint16_t *dfast) {
        if (skip_runs) {  // the last runs got nowhere (long codes): the general path is cheaper there
            --skip_runs;
            return -1;
        }
        INF_T0
        const uint32_t l = (uint32_t)prim.lane();
        unsigned long long bp = bit0;
        uint32_t p = prim.scalar(pos);
        const uint32_t p0 = p;
        const uint32_t ilen = prim.scalar(in_len);
        const uint32_t stop = (uint32_t)((((unsigned long long)p | (INF_CHUNK - 1)) < out_len ? (p | (INF_CHUNK - 1)) : out_len));  // p < stop
        while (p < stop) {
            // (bp and p are the same in every lane: said once per round, so that the loop around the chain is scalar code too)
            bp = (unsigned long long)prim.scalar((uint32_t)bp) | ((unsigned long long)prim.scalar((uint32_t)(bp >> 32)) << 32);
            p = prim.scalar(p);
            const uint32_t byte0 = (uint32_t)(bp >> 3);
            uint32_t hi = prim.scalar(in_hi);
            while (hi < ilen && byte0 + 32 + 512 > hi) {  // keep the window ahead of the round (literal chain + one match: < 32 bytes)
                load_chunk();
                hi = prim.scalar(in_hi);
            }
            if (byte0 + 32 > hi) break;  // the last bytes of the input: the general path pads with zeros
            // bits [bp + l, bp + l + 10) of the input: two aligned words of the window, funnel-shifted
            const uint32_t b = (uint32_t)(bp & 31u) + l, w = (uint32_t)(bp >> 5) + (b >> 5);
            const uint32_t *ring32 = (const uint32_t *)m->iring;
            const uint32_t lo = ring32[w & (INF_IN_RING / 4 - 1)], hi32 = ring32[(w + 1) & (INF_IN_RING / 4 - 1)];
            const uint32_t idx = (uint32_t)((((unsigned long long)hi32 << 32) | lo) >> (b & 31u)) & ((1u << INF_FAST_BITS) - 1u);
            const uint32_t E = lfast[idx];  // (symbol << 4) | length of the code starting at this lane's offset, 0 = long code
            // where the symbol after this lane's starts: lane + length (< 128) for a fast-table literal, 128 + lane for anything else
            const uint32_t NEXT = (E != 0 && (E >> 4) < 256u) ? l + (E & 15u) : 128u + l;
            // follow the chain: one v_readlane and a handful of scalar instructions per literal
            unsigned long long starts = 0;
            uint32_t cur = 0;
            const uint32_t room = stop - p;
            bool full = false;  // stopped because the output boundary was reached, not at a symbol of another kind
            if (room >= 64) {
                for (;;) {
                    const uint32_t nx = prim.read_lane(NEXT, cur);
                    if (nx >= 128u) break;  // the symbol at `cur` is not a literal
                    starts |= 1ull << cur;
                    cur = nx;
                    if (cur >= 64u) break;
                }
            } else {  // close to a boundary of the output: count as well
                uint32_t n_ = 0;
                for (;;) {
                    const uint32_t nx = prim.read_lane(NEXT, cur);
                    if (nx >= 128u) break;
                    if (n_ >= room) {
                        full = true;
                        break;
                    }
                    starts |= 1ull << cur;
                    cur = nx;
                    ++n_;
                    if (cur >= 64u) break;
                }
            }
            if (starts != 0) {
                if ((starts >> l) & 1ull) m->oring[(p + prim.rank_below(starts)) & (INF_RING - 1)] = (uint8_t)(E >> 4);
                p += (uint32_t)prim.popcount64(starts);
                bp += cur;
            }
            if (cur >= 64u) continue;  // literals all the way: next round
            if (full) break;
            // the chain stopped at a symbol that is not a literal.  A match whose codes are in the fast tables is taken right here
            // (one scalar look-up for the distance, the copy by the lanes); everything else is the general path's
            const uint32_t e0 = prim.read_lane(E, cur), sym = e0 >> 4;
            if (e0 == 0 || sym < 257u || sym > 285u) break;  // long code, end of block, invalid
            const unsigned long long bits = peek64(bp);
            const uint32_t c = sym - 257u, ll = e0 & 15u;
            const uint32_t le = c < 8u || c == 28u ? 0u : (c - 4u) >> 2;
            const uint32_t lb = c < 8u ? 3u + c : (c == 28u ? 258u : 3u + ((4u + (c & 3u)) << le));
            const uint32_t len = lb + ((uint32_t)(bits >> ll) & ((1u << le) - 1u));
            uint32_t used = ll + le;
            const uint32_t de = prim.scalar(dfast[(uint32_t)(bits >> used) & ((1u << INF_FAST_BITS) - 1u)]);
            const uint32_t ds = de >> 4, dl = de & 15u;
            if (de == 0 || ds >= 30u) break;
            const uint32_t dx = ds < 4u ? 0u : (ds >> 1) - 1u;
            const uint32_t db = ds < 4u ? 1u + ds : 1u + ((2u + (ds & 1u)) << dx);
            const uint32_t dist = db + ((uint32_t)(bits >> (used + dl)) & ((1u << dx) - 1u));
            used += dl + dx;  // at most 10 + 5 + 10 + 13 bits
            if (dist > p || p + len > stop) break;  // an error, or a copy across a write-back boundary: the general path's
            put_match(p, dist, len);
            p += len;
            bp += used;
        }
        INF_T1(t_run, n_run)
        if (p == p0) {
            skip_runs = 3;
            return -1;
        }
#ifdef PLO_INF_TIMING
        n_run += (int)(p - p0) - 1;
#endif
        pos = p;
        return (long long)bp;
    }
    // byte k of the match = byte (k mod dist) of the `dist` bytes before pos (an overlapping match repeats them): all sources
    // are older than pos, so the lanes copy independently
    PLO_HD void put_match(uint32_t pos, uint32_t dist, uint32_t len) {
        const uint32_t l = (uint32_t)prim.lane();
        const uint32_t src0 = pos - dist;
        INF_T0
        prim.sync();
        if (dist <= INF_NEAR && src0 >= ring_from) {
            for (uint32_t k = l; k < len; k += 64) m->oring[(pos + k) & (INF_RING - 1)] = m->oring[(src0 + (dist >= len ? k : k % dist)) & (INF_RING - 1)];
        } else {
            // (part of) the source has left the ring: it was written back at least INF_NEAR - INF_CHUNK - 258 bytes ago, or
            // was never in the ring (a stored run); prim.sync() above has waited for those stores
            for (uint32_t k = l; k < len; k += 64) {
                const uint32_t sp = src0 + (dist >= len ? k : k % dist);
                const bool near_ = sp >= ring_from && pos - sp <= INF_NEAR;
                m->oring[(pos + k) & (INF_RING - 1)] = near_ ? m->oring[sp & (INF_RING - 1)] : prim.load_written(out + sp);
            }
            INF_T1(t_far, n_far)
        }
        prim.sync();
        INF_T1(t_match, n_match)
    }
    PLO_HD void flush(uint32_t upto) {  // ring bytes [flushed, upto) -> global memory, 64 consecutive bytes per store instruction
        INF_T0
        const uint32_t l = (uint32_t)prim.lane();
        prim.sync();
        for (uint32_t k = flushed + l; k < upto; k += 64) out[k] = m->oring[k & (INF_RING - 1)];
        flushed = upto;
        INF_T1(t_flush, n_flush)
    }
    PLO_HD void put_stored(uint32_t pos, uint32_t in_pos, uint32_t len) {  // straight from the input to the output, past the ring
        flush(pos);
        const uint32_t l = (uint32_t)prim.lane();
        for (uint32_t k = l; k < len; k += 64) out[pos + k] = in[in_pos + k];
        flushed = pos + len;
        ring_from = pos + len;
    }
    PLO_HD void advance(uint32_t pos) {  // after every symbol: write back the halves of the ring that are complete
        const uint32_t b = pos & ~(INF_CHUNK - 1);
        if (b > flushed) flush(b);
    }
    PLO_HD void end(uint32_t pos) {
        if (pos > flushed) flush(pos);
        prim.sync();
    }
};

// ---- bit reader: 64-bit buffer refilled with aligned 32-bit words ---------------------------------------------------------------
template <class Par>
PLO_HD void inf_refill(Par &par, InfBits &s) {  // at least 33 bits afterwards (zeros beyond the end of the input)
    if (s.cnt <= 32) {
        s.buf |= (unsigned long long)par.in_word(s.ipos) << s.cnt;
        s.cnt += 32;
        s.ipos += 4;
    }
}
PLO_HD uint32_t inf_take(InfBits &s, int n) {  // n <= 16
    uint32_t v = (uint32_t)(s.buf & ((1ull << n) - 1ull));
    s.buf >>= n;
    s.cnt -= n;
    return v;
}
PLO_HD bool inf_overrun(const InfBits &s) {  // more bits taken than the input has
    return (long long)s.ipos * 8 - (long long)s.cnt > (long long)s.in_len * 8;
}
template <class Par>
PLO_HD void inf_seek(Par &par, InfBits &s, uint32_t byte_pos) {  // continue reading at a byte position
    s.ipos = byte_pos & ~3u;
    s.buf = 0;
    s.cnt = 0;
    inf_refill(par, s);
    inf_take(s, (int)(8 * (byte_pos & 3u)));
    inf_refill(par, s);
}

// The decode state is the same in every lane; saying so once per symbol keeps it in scalar registers whatever the compiler can
// prove about the paths in between (a no-op for values that are scalar already, and on the host)
template <class Par>
PLO_HD void inf_pin(const Par &par, InfBits &s, uint32_t &pos) {
    s.buf = (unsigned long long)par.scalar((uint32_t)s.buf) | ((unsigned long long)par.scalar((uint32_t)(s.buf >> 32)) << 32);
    s.cnt = (int)par.scalar((uint32_t)s.cnt);
    s.ipos = par.scalar(s.ipos);
    pos = par.scalar(pos);
}

// canonical tables from code lengths: returns 0 complete, > 0 incomplete, < 0 over-subscribed
PLO_HD int inf_canonical(uint16_t *count, uint16_t *symbol, const uint8_t *length, int n, uint16_t *offs /*[16] workspace*/) {
    for (int l = 0; l < 16; ++l) count[l] = 0;
    for (int s = 0; s < n; ++s) count[length[s]]++;
    if (count[0] == n) return 0;
    int left = 1;
    for (int l = 1; l < 16; ++l) {
        left <<= 1;
        left -= count[l];
        if (left < 0) return left;
    }
    offs[1] = 0;
    for (int l = 1; l < 15; ++l) offs[l + 1] = (uint16_t)(offs[l] + count[l]);
    for (int s = 0; s < n; ++s)
        if (length[s] != 0) symbol[offs[length[s]]++] = (uint16_t)s;
    return left;
}
// slow path: one bit per step (codes longer than the fast table, and the code-length code)
template <class Par>
PLO_HD int inf_decode_slow(const Par &par, InfBits &s, const uint16_t *count, const uint16_t *symbol) {
    int code = 0, first = 0, index = 0;
    for (int len = 1; len <= 15; ++len) {
        code |= (int)inf_take(s, 1);
        int c = (int)par.scalar(count[len]);
        if (code - c < first) return (int)par.scalar(symbol[index + (code - first)]);
        index += c;
        first += c;
        first <<= 1;
        code <<= 1;
    }
    return -1;
}
// fills fast[] for the code given by length[0..n): every lane computes the canonical codes (uniform work), the table entries
// are written lane-strided
template <class Par>
PLO_HD void inf_fill_fast(const Par &par, uint16_t *fast, uint16_t *code, const uint16_t *count, const uint8_t *length, int n, uint16_t *next /*[16] workspace*/) {
    const int lane = par.lane(), width = par.width();
    for (int i = lane; i < (1 << INF_FAST_BITS); i += width) fast[i] = 0;
    if (lane == 0) {  // first code of every length (RFC 1951, 3.2.2; count[0] counts the unused symbols and stays out)
        int c = 0;
        next[0] = 0;
        for (int l = 1; l < 16; ++l) {
            c = (c + (l > 1 ? count[l - 1] : 0)) << 1;
            next[l] = (uint16_t)c;
        }
        for (int s = 0; s < n; ++s) code[s] = length[s] ? next[length[s]]++ : 0;
    }
    par.sync();
    for (int s = lane; s < n; s += width) {
        const int l = length[s];
        if (l == 0 || l > INF_FAST_BITS) continue;
        unsigned cv = code[s], rev = 0;
        for (int b = 0; b < l; ++b) rev |= ((cv >> b) & 1u) << (l - 1 - b);  // codes enter the stream most significant bit first
        const uint16_t e = (uint16_t)((s << 4) | l);
        for (unsigned i = rev; i < (1u << INF_FAST_BITS); i += 1u << l) fast[i] = e;
    }
    par.sync();
}

template <class Par>
PLO_HD int inflate_block(Par &par, const uint8_t *in, uint32_t in_len, uint8_t *out, uint32_t out_len, InfWork &ws, uint32_t *out_written) {
    // Length / distance code bases and extra bits (RFC 1951, 3.2.5) and the order of the code-length code lengths (3.2.7) are
    // computed / unpacked from constants: as indexed arrays they would live in global memory, one dependent load (a microsecond)
    // per look-up, four look-ups per match.
    //   length symbol 257 + c: c < 8: 3 + c, no extra bits; else e = (c - 4) >> 2 extra bits, base 3 + ((4 + (c & 3)) << e); c = 28: 258
    //   distance symbol d:     d < 4: 1 + d, no extra bits; else e = (d >> 1) - 1 extra bits, base 1 + ((2 + (d & 1)) << e)
    // clorder = 16 17 18 0 8 7 9 6 10 5 11 4 | 12 3 13 2 14 1 15, five bits each
    constexpr unsigned long long CLO_LO = 16ull | (17ull << 5) | (18ull << 10) | (0ull << 15) | (8ull << 20) | (7ull << 25) | (9ull << 30) | (6ull << 35) |
                                          (10ull << 40) | (5ull << 45) | (11ull << 50) | (4ull << 55);
    constexpr unsigned long long CLO_HI = 12ull | (3ull << 5) | (13ull << 10) | (2ull << 15) | (14ull << 20) | (1ull << 25) | (15ull << 30);
    const int lane = par.lane(), width = par.width();

    par.begin(in, in_len, out, out_len);
    InfBits s;
    s.in_len = in_len;
    s.ipos = 0;
    s.buf = 0;
    s.cnt = 0;
    uint32_t pos = 0;
    int last;
    do {
        inf_refill(par, s);
        last = (int)inf_take(s, 1);
        const int type = (int)inf_take(s, 2);
        if (inf_overrun(s)) return INF_ERR_INPUT;
        if (type == 0) {  // stored: back to the byte boundary, LEN, NLEN, bytes
            uint32_t bp = (uint32_t)(((long long)s.ipos * 8 - s.cnt + 7) >> 3);  // first whole byte not yet consumed
            if ((unsigned long long)bp + 4 > in_len) return INF_ERR_INPUT;
            inf_seek(par, s, bp);
            const uint32_t len = inf_take(s, 16);
            inf_refill(par, s);
            const uint32_t nlen = inf_take(s, 16);
            bp += 4;
            if ((len ^ 0xffffu) != nlen) return INF_ERR_STORED;
            if ((unsigned long long)bp + len > in_len) return INF_ERR_INPUT;
            if ((unsigned long long)pos + len > out_len) return INF_ERR_OUTPUT;
            if (len) par.put_stored(pos, bp, len);
            pos += len;
            inf_seek(par, s, bp + len);
            continue;
        }
        if (type == 3) return INF_ERR_BTYPE;
#ifdef PLO_INF_TIMING
        long long tt0_ = par.prim.clock();
#endif
        par.sync();  // the previous block's tables are no longer read
        if (type == 1) {  // fixed codes
            for (int sym = lane; sym < 288; sym += width) ws.lengths[sym] = (uint8_t)(sym < 144 ? 8 : (sym < 256 ? 9 : (sym < 280 ? 7 : 8)));
            par.sync();
            if (lane == 0) inf_canonical(ws.lcount, ws.lsym, ws.lengths, 288, ws.offs);
            par.sync();
            inf_fill_fast(par, ws.lfast, ws.code, ws.lcount, ws.lengths, 288, ws.next);
            for (int sym = lane; sym < 30; sym += width) ws.lengths[sym] = 5;
            par.sync();
            if (lane == 0) inf_canonical(ws.dcount, ws.dsym, ws.lengths, 30, ws.offs);
            par.sync();
            inf_fill_fast(par, ws.dfast, ws.code, ws.dcount, ws.lengths, 30, ws.next);
        } else {  // dynamic codes
            const int nlen = (int)inf_take(s, 5) + 257;
            const int ndist = (int)inf_take(s, 5) + 1;
            const int ncode = (int)inf_take(s, 4) + 4;
            if (inf_overrun(s)) return INF_ERR_INPUT;
            if (nlen > 286 || ndist > 30) return INF_ERR_TABLE;
            // every lane decodes the code lengths (uniform); lane 0 writes them
            for (int i = lane; i < 19; i += width) ws.lengths[i] = 0;
            par.sync();
            for (int i = 0; i < ncode; ++i) {
                inf_refill(par, s);
                const uint32_t v = inf_take(s, 3);
                const int at = (int)(((i < 12 ? CLO_LO >> (5 * i) : CLO_HI >> (5 * (i - 12)))) & 31ull);
                if (lane == 0) ws.lengths[at] = (uint8_t)v;
            }
            if (inf_overrun(s)) return INF_ERR_INPUT;
            par.sync();
            int err = 0;
            if (lane == 0) err = inf_canonical(ws.lcount, ws.lsym, ws.lengths, 19, ws.offs);
            err = (int)par.uniform((uint32_t)err);
            par.sync();
            if (err != 0) return INF_ERR_TABLE;  // the code-length code must be complete
            // the 19-symbol code is decoded bit by bit from ws.lcount / ws.lsym, which stay untouched until all lengths are read
            // (ws.lengths is overwritten as they arrive)
            const uint16_t *ccount = ws.lcount, *csym = ws.lsym;
            int idx = 0;
            while (idx < nlen + ndist) {
                inf_refill(par, s);
                int sym = inf_decode_slow(par, s, ccount, csym);
                if (sym < 0) return inf_overrun(s) ? INF_ERR_INPUT : INF_ERR_SYMBOL;
                if (sym < 16) {
                    if (lane == 0) ws.lengths[idx] = (uint8_t)sym;
                    ++idx;
                } else {
                    int len = 0, rep;
                    if (sym == 16) {
                        if (idx == 0) return INF_ERR_TABLE;
                        par.sync();
                        len = (int)par.scalar(ws.lengths[idx - 1]);
                        rep = 3 + (int)inf_take(s, 2);
                    } else if (sym == 17) {
                        rep = 3 + (int)inf_take(s, 3);
                    } else {
                        rep = 11 + (int)inf_take(s, 7);
                    }
                    if (inf_overrun(s)) return INF_ERR_INPUT;
                    if (idx + rep > nlen + ndist) return INF_ERR_TABLE;
                    if (lane == 0)
                        for (int r = 0; r < rep; ++r) ws.lengths[idx + r] = (uint8_t)len;
                    idx += rep;
                }
            }
            par.sync();
            if (ws.lengths[256] == 0) return INF_ERR_TABLE;  // no end-of-block code
            int e1 = 0, e2 = 0;
            if (lane == 0) {
                e1 = inf_canonical(ws.lcount, ws.lsym, ws.lengths, nlen, ws.offs);
                if (e1 > 0 && nlen == ws.lcount[0] + ws.lcount[1]) e1 = 0;  // incomplete only as a single one-bit code
                e2 = inf_canonical(ws.dcount, ws.dsym, ws.lengths + nlen, ndist, ws.offs);
                if (e2 > 0 && ndist == ws.dcount[0] + ws.dcount[1]) e2 = 0;
            }
            e1 = (int)par.uniform((uint32_t)(e1 | e2));
            par.sync();
            if (e1 != 0) return INF_ERR_TABLE;
            inf_fill_fast(par, ws.lfast, ws.code, ws.lcount, ws.lengths, nlen, ws.next);
            inf_fill_fast(par, ws.dfast, ws.code, ws.dcount, ws.lengths + nlen, ndist, ws.next);
        }
#ifdef PLO_INF_TIMING
        par.t_table += par.prim.clock() - tt0_;
        ++par.n_table;
#endif
        // literal / length + distance symbols until the end-of-block code
        for (;;) {
            {   // a run of literals, if one starts here: the policy decodes it and says at which bit the general path goes on
                par.sync();
                const long long bp = par.literal_run((unsigned long long)((long long)s.ipos * 8 - s.cnt), pos, out_len, ws.lfast, ws.dfast);
                if (bp >= 0) {
                    inf_seek(par, s, (uint32_t)(bp >> 3));
                    inf_take(s, (int)(bp & 7));
                    par.advance(pos);
                }
            }
            inf_pin(par, s, pos);
            par.pin();
            inf_refill(par, s);
            int sym;
            uint32_t e = par.scalar(ws.lfast[s.buf & ((1u << INF_FAST_BITS) - 1u)]);
            if (e) {
                sym = (int)(e >> 4);
                inf_take(s, (int)(e & 15u));
            } else {
                sym = inf_decode_slow(par, s, ws.lcount, ws.lsym);
                if (sym < 0) return inf_overrun(s) ? INF_ERR_INPUT : INF_ERR_SYMBOL;
            }
            if (sym < 256) {
                if (pos >= out_len) return INF_ERR_OUTPUT;
                par.put_literal(pos, (uint32_t)sym);
                ++pos;
            } else if (sym == 256) {
                if (inf_overrun(s)) return INF_ERR_INPUT;
                break;
            } else {
                sym -= 257;
                if (sym >= 29) return INF_ERR_LENGTH;
                const int le = sym < 8 || sym == 28 ? 0 : (sym - 4) >> 2;
                const uint32_t lb = sym < 8 ? 3u + (uint32_t)sym : (sym == 28 ? 258u : 3u + ((4u + ((uint32_t)sym & 3u)) << le));
                const uint32_t len = lb + inf_take(s, le);  // (at least 33 - 15 - 5 bits were left: no refill needed)
                inf_refill(par, s);
                int ds;
                e = par.scalar(ws.dfast[s.buf & ((1u << INF_FAST_BITS) - 1u)]);
                if (e) {
                    ds = (int)(e >> 4);
                    inf_take(s, (int)(e & 15u));
                } else {
                    ds = inf_decode_slow(par, s, ws.dcount, ws.dsym);
                    if (ds < 0) return inf_overrun(s) ? INF_ERR_INPUT : INF_ERR_SYMBOL;
                }
                if (ds >= 30) return INF_ERR_DISTANCE;
                const int de = ds < 4 ? 0 : (ds >> 1) - 1;
                const uint32_t db = ds < 4 ? 1u + (uint32_t)ds : 1u + ((2u + ((uint32_t)ds & 1u)) << de);
                const uint32_t dist = db + inf_take(s, de);
                if (inf_overrun(s)) return INF_ERR_INPUT;
                if (dist > pos) return INF_ERR_DISTANCE;
                if (pos + len > out_len) return INF_ERR_OUTPUT;
                par.put_match(pos, dist, len);
                pos += len;
            }
            par.advance(pos);
        }
    } while (!last);
    if (inf_overrun(s)) return INF_ERR_INPUT;
    par.end(pos);
    if (out_written) *out_written = pos;
    return INF_OK;
}

// -------------------------------------------------------------------------------------------------------------------
// CRC-32 of an inflated BGZF block by ONE WAVE (RFC 1952 section 8: the IEEE 802.3 polynomial, reflected, 0xEDB88320), so that the
// host need not read the inflated bytes to check them (k_bgzf_crc behind k_bgzf_inflate).  The block is cut into 64 consecutive chunks,
// lane i runs the table-driven CRC over chunk i, and the chunks' CRCs are combined with the identity
//     crc(A || B) = crc(A) * x^(8 |B|)  xor  crc(B)        (polynomials over GF(2) modulo the generator)
// i.e. lane i multiplies its CRC by x^(8 * bytes behind its chunk) -- a square-and-multiply power, 32-step carry-less products -- and
// the wave XORs the products.  `tab`: the 256-entry byte table in LDS (crc32_table_entry), shared by the workgroup.
// -------------------------------------------------------------------------------------------------------------------
constexpr uint32_t CRC32_POLY = 0xEDB88320u;
PLO_HD uint32_t crc32_table_entry(uint32_t e) {
    uint32_t c = e;
    for (int k = 0; k < 8; ++k) c = (c >> 1) ^ ((c & 1u) ? CRC32_POLY : 0u);
    return c;
}
// a(x) * b(x) mod p(x) in the reflected representation (bit 31 = x^0)
PLO_HD uint32_t crc32_mulmod(uint32_t a, uint32_t b) {
    uint32_t p = 0;
    for (int i = 0; i < 32; ++i) {
        p ^= (a & 0x80000000u) ? b : 0u;
        a <<= 1;
        b = (b >> 1) ^ ((b & 1u) ? CRC32_POLY : 0u);  // b * x
    }
    return p;
}
PLO_HD uint32_t crc32_xpow8(uint32_t n_bytes) {  // x^(8 n) mod p
    uint32_t r = 0x80000000u, base = 0x00800000u;   // 1, x^8
    for (uint32_t n = n_bytes; n; n >>= 1) {
        if (n & 1u) r = crc32_mulmod(r, base);
        base = crc32_mulmod(base, base);
    }
    return r;
}
// wave-uniform call; every lane returns the block's CRC-32
PLO_DEV uint32_t crc32_wave(const uint8_t *p, uint32_t n, const uint32_t *tab) {
    const uint32_t lane = (uint32_t)wv::lane();
    const uint32_t per = (n + 63u) / 64u;
    const uint32_t lo = lane * per < n ? lane * per : n, hi = lo + per < n ? lo + per : n;
    uint32_t c = 0xffffffffu;
    uint32_t a = lo;
    const PLO_GLOBAL uint8_t *g = (const PLO_GLOBAL uint8_t *)p;
    while (a < hi && (((uintptr_t)(p + a)) & 3u)) {
        c = tab[(c ^ g[a]) & 0xffu] ^ (c >> 8);
        ++a;
    }
    while (a + 4u <= hi) {
        uint32_t w = *(const PLO_GLOBAL uint32_t *)(p + a);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            c = tab[(c ^ w) & 0xffu] ^ (c >> 8);
            w >>= 8;
        }
        a += 4u;
    }
    while (a < hi) {
        c = tab[(c ^ g[a]) & 0xffu] ^ (c >> 8);
        ++a;
    }
    c = hi > lo ? c ^ 0xffffffffu : 0u;  // (the CRC of no bytes is 0: an empty chunk adds nothing)
    uint32_t s = crc32_mulmod(crc32_xpow8(n - hi), c);
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) s ^= wv::shfl(s, (int)(lane ^ (uint32_t)d));
    return s;
}

}  // namespace plo
