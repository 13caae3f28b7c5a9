// tests/emu/plo_wave.hpp -- HOST wave64 emulator with the same interface as portello_amd/csrc/plo_wave.hpp.
//
// TEST INFRASTRUCTURE ONLY.  It lets the device algorithm (portello_amd/csrc/lift_core.hpp) be compiled with g++
// and executed lane by lane on the CPU (one ucontext fiber per lane), so that the wave-level decomposition can be
// diffed against the oracle, run under ASan/UBSan, and checked for wave-uniform use of the primitives -- none of
// which is possible on the GPU pool.  Every primitive is a rendezvous of all 64 lanes; a lane that calls a
// different primitive (or none) than its peers aborts the run ("divergent primitive").
#pragma once
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <ucontext.h>

#include <functional>
#include <vector>

#define PLO_DEV inline
#define PLO_HD inline
#define PLO_WAVE 64
#define PLO_GLOBAL
#define PLO_EMULATOR 1

namespace wv {

struct EmuWave {
    // one emulated workgroup of `nw` waves (nw = 1: the single wave of the tile / retry / large-item kernels)
    static constexpr int WAVE = 64;
    static constexpr size_t STACK = 256 * 1024;
    int nw = 1;
    int N = 64;
    ucontext_t sched;
    std::vector<ucontext_t> fiber;
    std::vector<char> stacks;
    std::vector<char> done;
    int cur = 0;
    // rendezvous state of the wave-level primitives: double-buffered by primitive parity
    struct Slot {
        int64_t v[4];
    };
    std::vector<Slot> slot[2];
    std::vector<int> tag[2];
    std::vector<unsigned long long> seq;
    // workgroup barrier (wv::block_sync): fibers wait for the generation to advance; the scheduler releases a generation
    // at the start of a round once every live fiber has arrived, so that the lanes of a wave leave it in the same round
    unsigned long long bar_gen = 0;
    int bar_arrived = 0;
    std::function<void()> body;
    unsigned order_seed = 0;  // != 0: shuffle the lane execution order every round (catches missing wv::sync())

    static EmuWave *&current() {
        static thread_local EmuWave *w = nullptr;
        return w;
    }
    static void trampoline() {
        EmuWave *w = current();
        w->body();
        w->done[w->cur] = 1;
        swapcontext(&w->fiber[w->cur], &w->sched);
    }
    void run(std::function<void()> fn) {
        body = std::move(fn);
        N = WAVE * nw;
        fiber.resize(N);
        done.assign(N, 0);
        seq.assign(N, 0);
        for (int p = 0; p < 2; ++p) {
            slot[p].assign(N, Slot{{0, 0, 0, 0}});
            tag[p].assign(N, 0);
        }
        stacks.assign(STACK * N, 0);
        bar_gen = 0;
        bar_arrived = 0;
        current() = this;
        for (int l = 0; l < N; ++l) {
            getcontext(&fiber[l]);
            fiber[l].uc_stack.ss_sp = stacks.data() + STACK * l;
            fiber[l].uc_stack.ss_size = STACK;
            fiber[l].uc_link = &sched;
            makecontext(&fiber[l], (void (*)())trampoline, 0);
        }
        std::vector<int> order(N);
        for (int l = 0; l < N; ++l) order[l] = l;
        unsigned rs = order_seed;
        for (;;) {
            int live = 0;
            for (int l = 0; l < N; ++l) live += done[l] ? 0 : 1;
            if (!live) break;
            if (bar_arrived > 0) {
                if (bar_arrived == live) {
                    bar_arrived = 0;
                    ++bar_gen;
                } else if (stalled) {
                    fprintf(stderr, "wave emulator: workgroup barrier reached by %d of %d live lanes only (divergent block_sync)\n", bar_arrived, live);
                    abort();
                }
            }
            if (order_seed) {
                for (int i = N - 1; i > 0; --i) {
                    rs = rs * 1664525u + 1013904223u;
                    int j = (int)((rs >> 8) % (unsigned)(i + 1));
                    int t = order[i];
                    order[i] = order[j];
                    order[j] = t;
                }
            }
            // waves of a workgroup drift apart between two workgroup barriers on the hardware: in shuffle mode whole waves sit
            // out rounds at random, so that a wave can run several wave-level primitives ahead of another one
            unsigned skip = 0;
            if (order_seed && nw > 1) {
                rs = rs * 1664525u + 1013904223u;
                skip = (rs >> 9) & (rs >> 17) & ((1u << nw) - 1u);  // each wave sits out with probability 1/4
                if (skip == (1u << nw) - 1u) skip = 0;
            }
            progress = 0;
            for (int i = 0; i < N; ++i) {
                int l = order[i];
                if (done[l] || ((skip >> (l >> 6)) & 1u)) continue;
                cur = l;
                swapcontext(&sched, &fiber[l]);
            }
            stalled = progress == 0 && skip == 0;
        }
        current() = nullptr;
    }
    int progress = 0;
    bool stalled = false;
    int wave_base() const { return cur & ~(WAVE - 1); }
    // rendezvous: publish (tag, payload), wait for the wave, return parity buffer index to read from
    int rendezvous(int t, const int64_t *payload, int n) {
        int par = (int)(seq[cur] & 1);
        for (int i = 0; i < n; ++i) slot[par][cur].v[i] = payload[i];
        tag[par][cur] = t;
        seq[cur]++;
        ++progress;
        int me = cur;
        swapcontext(&fiber[me], &sched);
        cur = me;
        return par;
    }
    void check(int par, int t) {
        int b = wave_base();
        for (int l = b; l < b + WAVE; ++l) {
            if (seq[l] < seq[cur] || tag[par][l] != t) {
                fprintf(stderr, "wave emulator: divergent primitive (lane %d tag %d vs lane %d tag %d, done=%d)\n", cur, t, l,
                        tag[par][l], (int)done[l]);
                abort();
            }
        }
    }
    int64_t peer(int par, int lane_in_wave, int i = 0) const { return slot[par][wave_base() + (lane_in_wave & (WAVE - 1))].v[i]; }
    void block_barrier() {
        int me = cur;
        unsigned long long g = bar_gen;
        ++bar_arrived;
        ++progress;
        while (bar_gen == g) {
            swapcontext(&fiber[me], &sched);
            cur = me;
        }
    }
};

inline EmuWave &W() { return *EmuWave::current(); }

inline int lane() { return W().cur & 63; }
inline int wave_id() { return W().cur >> 6; }
inline void block_sync() { W().block_barrier(); }
inline long long clock() { return 0; }

enum { T_SYNC = 1, T_SHFL, T_SHFL_UP1, T_BALLOT, T_BCAST_LAST, T_BCAST_FIRST, T_SCAN_ADD, T_SCAN_MAX, T_SCAN_MP };

inline void sync() {
    int64_t z = 0;
    int par = W().rendezvous(T_SYNC, &z, 1);
    W().check(par, T_SYNC);
}

inline int64_t shfl64(int64_t v, int src) {
    int par = W().rendezvous(T_SHFL, &v, 1);
    W().check(par, T_SHFL);
    return W().peer(par, src);
}
inline int shfl(int v, int src) { return (int)shfl64(v, src); }
inline unsigned shfl(unsigned v, int src) { return (unsigned)shfl64((int64_t)v, src); }
inline long long shfl(long long v, int src) { return (long long)shfl64(v, src); }
inline unsigned long long shfl(unsigned long long v, int src) { return (unsigned long long)shfl64((int64_t)v, src); }

inline int read_lane(int v, int l) { return shfl(v, l & 63); }

inline int shfl_up1(int v, int first) {
    int64_t p = v;
    int par = W().rendezvous(T_SHFL_UP1, &p, 1);
    W().check(par, T_SHFL_UP1);
    int l = lane();
    return l == 0 ? first : (int)W().peer(par, l - 1);
}

inline unsigned long long ballot(bool p) {
    int64_t v = p ? 1 : 0;
    int par = W().rendezvous(T_BALLOT, &v, 1);
    W().check(par, T_BALLOT);
    unsigned long long m = 0;
    for (int l = 0; l < 64; ++l)
        if (W().peer(par, l)) m |= 1ull << l;
    return m;
}
inline int bcast_last(int v) {
    int64_t p = v;
    int par = W().rendezvous(T_BCAST_LAST, &p, 1);
    W().check(par, T_BCAST_LAST);
    return (int)W().peer(par, 63);
}
inline int bcast_first(int v) {
    int64_t p = v;
    int par = W().rendezvous(T_BCAST_FIRST, &p, 1);
    W().check(par, T_BCAST_FIRST);
    return (int)W().peer(par, 0);
}
inline unsigned bcast_first(unsigned v) { return (unsigned)bcast_first((int)v); }
inline unsigned long long bcast_first(unsigned long long v) {
    unsigned lo = bcast_first((unsigned)(v & 0xffffffffull)), hi = bcast_first((unsigned)(v >> 32));
    return ((unsigned long long)hi << 32) | lo;
}
inline int imax(int a, int b) { return a > b ? a : b; }
inline int imin(int a, int b) { return a < b ? a : b; }
// byte primitives: plain C restatements of v_perm_b32 (selector bytes 0..7 only) and v_alignbyte
inline unsigned perm_bytes(unsigned hi, unsigned lo, unsigned sel) {
    unsigned long long src = ((unsigned long long)hi << 32) | lo;
    unsigned r = 0;
    for (int i = 0; i < 4; ++i) {
        unsigned sb = (sel >> (8 * i)) & 0xffu;
        unsigned b = sb < 8 ? (unsigned)((src >> (8 * sb)) & 0xffu) : (sb == 12 ? 0u : 0xffu);
        r |= b << (8 * i);
    }
    return r;
}
inline unsigned align_bytes(unsigned hi, unsigned lo, unsigned shift) {
    return (unsigned)(((((unsigned long long)hi) << 32) | lo) >> (8 * (shift & 3u)));
}
inline int clz32(unsigned x) { return __builtin_clz(x); }
inline int ctz32(unsigned x) { return __builtin_ctz(x); }

inline int scan_add(int x) {
    int64_t p = x;
    int par = W().rendezvous(T_SCAN_ADD, &p, 1);
    W().check(par, T_SCAN_ADD);
    unsigned s = 0;  // wrap-around like the hardware
    for (int l = 0; l <= lane(); ++l) s += (unsigned)(int)W().peer(par, l);
    return (int)s;
}
inline int scan_max(int x) {
    int64_t p = x;
    int par = W().rendezvous(T_SCAN_MAX, &p, 1);
    W().check(par, T_SCAN_MAX);
    int m = (int)0x80000000;
    for (int l = 0; l <= lane(); ++l) m = imax(m, (int)W().peer(par, l));
    return m;
}
inline int reduce_add(int x) { return bcast_last(scan_add(x)); }
inline int reduce_max(int x) { return bcast_last(scan_max(x)); }

struct MinPlus {
    int a, b, s;
};
inline int sat_add(int x, int y) {
    long long t = (long long)x + (long long)y;
    return t > 0x7fffffffLL ? 0x7fffffff : (int)t;
}
inline MinPlus mp_compose(MinPlus p, MinPlus c) {
    MinPlus r;
    if (c.s) return c;
    r.a = sat_add(p.a, c.a);
    r.b = imin(sat_add(p.b, c.a), c.b);
    r.s = p.s;
    return r;
}
inline MinPlus scan_minplus(MinPlus x) {
    int64_t p[3] = {x.a, x.b, x.s};
    int par = W().rendezvous(T_SCAN_MP, p, 3);
    W().check(par, T_SCAN_MP);
    MinPlus acc = {0, 0x7fffffff, 0};
    for (int l = 0; l <= lane(); ++l) {
        MinPlus c = {(int)W().peer(par, l, 0), (int)W().peer(par, l, 1), (int)W().peer(par, l, 2)};
        acc = mp_compose(acc, c);
    }
    return acc;
}
inline MinPlus bcast_last(MinPlus v) {
    MinPlus r;
    r.a = bcast_last(v.a);
    r.b = bcast_last(v.b);
    r.s = bcast_last(v.s);
    return r;
}

inline void atomic_add(int *p, int v) { *p += v; }
inline void atomic_add(unsigned *p, unsigned v) { *p += v; }
inline void atomic_min(int *p, int v) { if (v < *p) *p = v; }
inline void atomic_max(int *p, int v) { if (v > *p) *p = v; }
inline void atomic_or(int *p, int v) { *p |= v; }
inline unsigned long long atomic_add_global(unsigned long long *p, unsigned long long v) {
    unsigned long long o = *p;
    *p += v;
    return o;
}
inline unsigned atomic_add_global(unsigned *p, unsigned v) {
    unsigned o = *p;
    *p += v;
    return o;
}

}  // namespace wv
