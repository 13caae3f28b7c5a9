// inflate.hpp -- DEFLATE (RFC 1951) decoder for one BGZF block, written for ONE WAVEFRONT per block (k_bgzf_inflate,
// engine.hip) and compiled for the host as well (tests, tests/emu).  A BGZF file is a series of independent deflate streams
// of at most 64 KiB each, so a BAM of N GB is N x 16 k independent decode jobs: the host cores of a GPU node inflate
// ~0.25 GB/s each, the device takes every block of a 256 MB chunk at once.
//
// Decoding a Huffman stream is sequential, so all lanes of the wave run the SAME decode on the same bits -- uniform control
// flow, table look-ups that are LDS broadcasts -- and the wave spreads only the data movement: the primary look-up tables are
// filled lane-strided, and a match of `len` bytes is copied by `len` lanes at once (byte k from position (k mod dist) of
// the source period, which also covers overlapping matches).  Codes of up to 10 bits (nearly all) cost one table look-up;
// longer ones fall back to canonical decoding by code length.  Input and output are bounds-checked: the decoder returns 0 or
// a negative error code and never reads or writes outside [in, in + in_len) / [out, out + out_len).
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
    uint8_t lengths[320];
};

// host execution: one "lane"
struct InfSerial {
    PLO_HD int lane() const { return 0; }
    PLO_HD int width() const { return 1; }
    PLO_HD void sync() const {}
    PLO_HD uint32_t uniform(uint32_t v) const { return v; }
};

struct InfBits {
    const uint8_t *in;
    uint32_t in_len, in_pos;
    unsigned long long buf;
    int cnt;  // valid bits in buf; negative after reading past the end of the input
};
PLO_HD void inf_refill(InfBits &s) {
    while (s.cnt <= 56 && s.in_pos < s.in_len) {
        s.buf |= (unsigned long long)s.in[s.in_pos++] << s.cnt;
        s.cnt += 8;
    }
}
PLO_HD uint32_t inf_take(InfBits &s, int n) {  // n <= 16; bits beyond the input read as zero and drive cnt negative
    uint32_t v = (uint32_t)(s.buf & ((1ull << n) - 1ull));
    s.buf >>= n;
    s.cnt -= n;
    return v;
}

// canonical tables from code lengths: returns 0 complete, > 0 incomplete, < 0 over-subscribed
PLO_HD int inf_canonical(uint16_t *count, uint16_t *symbol, const uint8_t *length, int n) {
    for (int l = 0; l < 16; ++l) count[l] = 0;
    for (int s = 0; s < n; ++s) count[length[s]]++;
    if (count[0] == n) return 0;
    int left = 1;
    for (int l = 1; l < 16; ++l) {
        left <<= 1;
        left -= count[l];
        if (left < 0) return left;
    }
    uint16_t offs[16];
    offs[1] = 0;
    for (int l = 1; l < 15; ++l) offs[l + 1] = (uint16_t)(offs[l] + count[l]);
    for (int s = 0; s < n; ++s)
        if (length[s] != 0) symbol[offs[length[s]]++] = (uint16_t)s;
    return left;
}
// slow path: one bit per step (codes longer than the fast table, and the code-length code)
PLO_HD int inf_decode_slow(InfBits &s, const uint16_t *count, const uint16_t *symbol) {
    int code = 0, first = 0, index = 0;
    for (int len = 1; len <= 15; ++len) {
        code |= (int)inf_take(s, 1);
        int c = count[len];
        if (code - c < first) return symbol[index + (code - first)];
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
PLO_HD void inf_fill_fast(const Par &par, uint16_t *fast, uint16_t *code, const uint16_t *count, const uint8_t *length, int n) {
    const int lane = par.lane(), width = par.width();
    for (int i = lane; i < (1 << INF_FAST_BITS); i += width) fast[i] = 0;
    uint16_t next[16];  // first code of every length (RFC 1951, 3.2.2; count[0] counts the unused symbols and stays out)
    int c = 0;
    next[0] = 0;
    for (int l = 1; l < 16; ++l) {
        c = (c + (l > 1 ? count[l - 1] : 0)) << 1;
        next[l] = (uint16_t)c;
    }
    if (lane == 0)
        for (int s = 0; s < n; ++s) code[s] = length[s] ? next[length[s]]++ : 0;
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
PLO_HD int inflate_block(const Par &par, const uint8_t *in, uint32_t in_len, uint8_t *out, uint32_t out_len, InfWork &ws, uint32_t *out_written) {
    // length / distance code bases and extra bits (RFC 1951, 3.2.5)
    const uint16_t lbase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
    const uint8_t lext[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
    const uint16_t dbase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
    const uint8_t dext[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
    const uint8_t clorder[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    const int lane = par.lane(), width = par.width();

    InfBits s;
    s.in = in;
    s.in_len = in_len;
    s.in_pos = 0;
    s.buf = 0;
    s.cnt = 0;
    uint32_t pos = 0;
    int last;
    do {
        inf_refill(s);
        last = (int)inf_take(s, 1);
        const int type = (int)inf_take(s, 2);
        if (s.cnt < 0) return INF_ERR_INPUT;
        if (type == 0) {  // stored: back to the byte boundary, LEN, NLEN, bytes
            const uint32_t whole = (uint32_t)s.cnt >> 3;  // unread whole bytes sitting in the bit buffer
            s.in_pos -= whole;
            s.buf = 0;
            s.cnt = 0;
            if (s.in_pos + 4 > s.in_len) return INF_ERR_INPUT;
            const uint32_t len = (uint32_t)in[s.in_pos] | ((uint32_t)in[s.in_pos + 1] << 8);
            const uint32_t nlen = (uint32_t)in[s.in_pos + 2] | ((uint32_t)in[s.in_pos + 3] << 8);
            s.in_pos += 4;
            if ((len ^ 0xffffu) != nlen) return INF_ERR_STORED;
            if (s.in_pos + len > s.in_len) return INF_ERR_INPUT;
            if (pos + len > out_len) return INF_ERR_OUTPUT;
            for (uint32_t k = (uint32_t)lane; k < len; k += (uint32_t)width) out[pos + k] = in[s.in_pos + k];
            pos += len;
            s.in_pos += len;
            continue;
        }
        if (type == 3) return INF_ERR_BTYPE;
        par.sync();  // the previous block's tables are no longer read
        if (type == 1) {  // fixed codes
            for (int sym = lane; sym < 288; sym += width) ws.lengths[sym] = (uint8_t)(sym < 144 ? 8 : (sym < 256 ? 9 : (sym < 280 ? 7 : 8)));
            par.sync();
            if (lane == 0) inf_canonical(ws.lcount, ws.lsym, ws.lengths, 288);
            par.sync();
            inf_fill_fast(par, ws.lfast, ws.code, ws.lcount, ws.lengths, 288);
            for (int sym = lane; sym < 30; sym += width) ws.lengths[sym] = 5;
            par.sync();
            if (lane == 0) inf_canonical(ws.dcount, ws.dsym, ws.lengths, 30);
            par.sync();
            inf_fill_fast(par, ws.dfast, ws.code, ws.dcount, ws.lengths, 30);
        } else {  // dynamic codes
            const int nlen = (int)inf_take(s, 5) + 257;
            const int ndist = (int)inf_take(s, 5) + 1;
            const int ncode = (int)inf_take(s, 4) + 4;
            if (s.cnt < 0) return INF_ERR_INPUT;
            if (nlen > 286 || ndist > 30) return INF_ERR_TABLE;
            // every lane decodes the code lengths (uniform); lane 0 writes them
            uint8_t cl[19];
            for (int i = 0; i < 19; ++i) cl[i] = 0;
            for (int i = 0; i < ncode; ++i) {
                inf_refill(s);
                cl[clorder[i]] = (uint8_t)inf_take(s, 3);
            }
            if (s.cnt < 0) return INF_ERR_INPUT;
            if (lane == 0)
                for (int i = 0; i < 19; ++i) ws.lengths[i] = cl[i];
            par.sync();
            int err = 0;
            if (lane == 0) err = inf_canonical(ws.lcount, ws.lsym, ws.lengths, 19);
            err = (int)par.uniform((uint32_t)err);
            par.sync();
            if (err != 0) return INF_ERR_TABLE;  // the code-length code must be complete
            // the 19-symbol code is decoded bit by bit from a private copy of its tables (ws.lengths is about to be overwritten)
            uint16_t ccount[16], csym[19];
            for (int l = 0; l < 16; ++l) ccount[l] = ws.lcount[l];
            for (int i = 0; i < 19; ++i) csym[i] = ws.lsym[i];
            par.sync();
            int idx = 0;
            while (idx < nlen + ndist) {
                inf_refill(s);
                int sym = inf_decode_slow(s, ccount, csym);
                if (sym < 0) return s.cnt < 0 ? INF_ERR_INPUT : INF_ERR_SYMBOL;
                if (sym < 16) {
                    if (lane == 0) ws.lengths[idx] = (uint8_t)sym;
                    ++idx;
                } else {
                    int len = 0, rep;
                    if (sym == 16) {
                        if (idx == 0) return INF_ERR_TABLE;
                        par.sync();
                        len = ws.lengths[idx - 1];
                        rep = 3 + (int)inf_take(s, 2);
                    } else if (sym == 17) {
                        rep = 3 + (int)inf_take(s, 3);
                    } else {
                        rep = 11 + (int)inf_take(s, 7);
                    }
                    if (s.cnt < 0) return INF_ERR_INPUT;
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
                e1 = inf_canonical(ws.lcount, ws.lsym, ws.lengths, nlen);
                if (e1 > 0 && nlen == ws.lcount[0] + ws.lcount[1]) e1 = 0;  // incomplete only as a single one-bit code
                e2 = inf_canonical(ws.dcount, ws.dsym, ws.lengths + nlen, ndist);
                if (e2 > 0 && ndist == ws.dcount[0] + ws.dcount[1]) e2 = 0;
            }
            e1 = (int)par.uniform((uint32_t)(e1 | e2));
            par.sync();
            if (e1 != 0) return INF_ERR_TABLE;
            inf_fill_fast(par, ws.lfast, ws.code, ws.lcount, ws.lengths, nlen);
            inf_fill_fast(par, ws.dfast, ws.code, ws.dcount, ws.lengths + nlen, ndist);
        }
        // literal / length + distance symbols until the end-of-block code
        for (;;) {
            inf_refill(s);
            int sym;
            uint32_t e = ws.lfast[s.buf & ((1u << INF_FAST_BITS) - 1u)];
            if (e) {
                sym = (int)(e >> 4);
                inf_take(s, (int)(e & 15u));
            } else {
                sym = inf_decode_slow(s, ws.lcount, ws.lsym);
                if (sym < 0) return s.cnt < 0 ? INF_ERR_INPUT : INF_ERR_SYMBOL;
            }
            if (s.cnt < 0) return INF_ERR_INPUT;
            if (sym < 256) {
                if (pos >= out_len) return INF_ERR_OUTPUT;
                if (lane == 0) out[pos] = (uint8_t)sym;
                ++pos;
            } else if (sym == 256) {
                break;
            } else {
                sym -= 257;
                if (sym >= 29) return INF_ERR_LENGTH;
                const uint32_t len = lbase[sym] + inf_take(s, lext[sym]);
                int ds;
                e = ws.dfast[s.buf & ((1u << INF_FAST_BITS) - 1u)];
                if (e) {
                    ds = (int)(e >> 4);
                    inf_take(s, (int)(e & 15u));
                } else {
                    ds = inf_decode_slow(s, ws.dcount, ws.dsym);
                    if (ds < 0) return s.cnt < 0 ? INF_ERR_INPUT : INF_ERR_SYMBOL;
                }
                if (ds >= 30) return INF_ERR_DISTANCE;
                const uint32_t dist = dbase[ds] + inf_take(s, dext[ds]);
                if (s.cnt < 0) return INF_ERR_INPUT;
                if (dist > pos) return INF_ERR_DISTANCE;
                if (pos + len > out_len) return INF_ERR_OUTPUT;
                // byte k of the match = byte (k mod dist) of the `dist` bytes before pos (an overlapping match repeats them)
                par.sync();  // earlier stores of the wave are visible to its loads
                if (width == 1) {
                    for (uint32_t k = 0; k < len; ++k) out[pos + k] = out[pos + k - dist];
                } else {
                    for (uint32_t k = (uint32_t)lane; k < len; k += (uint32_t)width) out[pos + k] = out[pos - dist + (k % dist)];
                    par.sync();
                }
                pos += len;
            }
        }
    } while (!last);
    if (out_written) *out_written = pos;
    return INF_OK;
}

}  // namespace plo
