// plo_wave.hpp -- wavefront-level primitives for gfx950 (CDNA4, 64-wide waves).
//
// Every primitive must be called in wave-uniform control flow (all 64 lanes reach the same call); predicates and
// per-lane values are passed as data.  The scans use the DPP row_shr / row_bcast cross-lane modes of the GFX9
// family (row = 16 lanes): four row_shr steps give an inclusive scan inside each row, row_bcast:15 and
// row_bcast:31 carry row totals across rows -- no LDS traffic, 6 VALU ops per 64-lane scan.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define PLO_DEV __device__ __forceinline__
#define PLO_HD __host__ __device__ __forceinline__
#define PLO_WAVE 64
// pointers rebuilt from integer addresses (item descriptors) are generic; this marks what they point to as global memory
#define PLO_GLOBAL __attribute__((address_space(1)))

namespace wv {

PLO_DEV int lane() { return (int)__lane_id(); }
PLO_DEV long long clock() { return (long long)__builtin_amdgcn_s_memtime(); }
PLO_DEV long long realtime() { return (long long)__builtin_amdgcn_s_memrealtime(); }  // constant 100 MHz, the same on every XCD
PLO_DEV unsigned hw_id() {  // HW_ID (wave / SIMD / CU / SH / SE) in the low half, XCC_ID in bits 16 ..
    unsigned h, x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)\n s_getreg_b32 %1, hwreg(HW_REG_XCC_ID)" : "=s"(h), "=s"(x));
    return (h & 0xffffu) | (x << 16);
}

// LDS (or wave-private global scratch) hand-off between lanes of ONE wave: DS operations of a wave execute in
// order, so all that is needed is to stop the compiler from moving memory accesses across this point.
PLO_DEV void sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

PLO_DEV int shfl(int v, int src) { return __shfl(v, src, PLO_WAVE); }
PLO_DEV unsigned shfl(unsigned v, int src) { return (unsigned)__shfl((int)v, src, PLO_WAVE); }
PLO_DEV long long shfl(long long v, int src) { return __shfl(v, src, PLO_WAVE); }
PLO_DEV unsigned long long shfl(unsigned long long v, int src) { return (unsigned long long)__shfl((long long)v, src, PLO_WAVE); }

// value of lane (l-1), `first` for lane 0
PLO_DEV int shfl_up1(int v, int first) {
    int r = __builtin_amdgcn_update_dpp(first, v, 0x138 /*wave_shr:1*/, 0xf, 0xf, false);
    return r;
}

PLO_DEV unsigned long long ballot(bool p) { return __ballot(p); }
PLO_DEV int bcast_last(int v) { return __builtin_amdgcn_readlane(v, 63); }
PLO_DEV int bcast_first(int v) { return __builtin_amdgcn_readfirstlane(v); }
PLO_DEV unsigned bcast_first(unsigned v) { return (unsigned)__builtin_amdgcn_readfirstlane((int)v); }
PLO_DEV unsigned long long bcast_first(unsigned long long v) {
    unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v & 0xffffffffull));
    unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32));
    return ((unsigned long long)hi << 32) | lo;
}
// Several waves of one workgroup working on the same tile (Coop<NW>, lift_core.hpp): index of the wave inside the workgroup
// (in an SGPR) and the workgroup barrier (LDS writes of every wave before it are visible to every wave after it)
PLO_DEV int read_lane(int v, int l) { return __builtin_amdgcn_readlane(v, l & 63); }  // l wave-uniform
PLO_DEV int wave_id() { return __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); }
PLO_DEV void block_sync() { __syncthreads(); }

#define PLO_DPP(old, x, ctrl, rmask) __builtin_amdgcn_update_dpp((old), (x), (ctrl), (rmask), 0xf, false)

PLO_DEV int scan_add(int x) {  // inclusive
    x += PLO_DPP(0, x, 0x111, 0xf);  // row_shr:1
    x += PLO_DPP(0, x, 0x112, 0xf);  // row_shr:2
    x += PLO_DPP(0, x, 0x114, 0xf);  // row_shr:4
    x += PLO_DPP(0, x, 0x118, 0xf);  // row_shr:8
    x += PLO_DPP(0, x, 0x142, 0xa);  // row_bcast:15 -> rows 1,3
    x += PLO_DPP(0, x, 0x143, 0xc);  // row_bcast:31 -> rows 2,3
    return x;
}

PLO_DEV int imax(int a, int b) { return a > b ? a : b; }
PLO_DEV int imin(int a, int b) { return a < b ? a : b; }

PLO_DEV int scan_max(int x) {  // inclusive, identity INT_MIN
    const int I = (int)0x80000000;
    x = imax(x, PLO_DPP(I, x, 0x111, 0xf));
    x = imax(x, PLO_DPP(I, x, 0x112, 0xf));
    x = imax(x, PLO_DPP(I, x, 0x114, 0xf));
    x = imax(x, PLO_DPP(I, x, 0x118, 0xf));
    x = imax(x, PLO_DPP(I, x, 0x142, 0xa));
    x = imax(x, PLO_DPP(I, x, 0x143, 0xc));
    return x;
}

// byte primitives (per lane): v_perm_b32 as a 4-wide byte gather over the 8 bytes {hi:lo} (selector bytes 0..7), v_alignbyte
// as a funnel shift by whole bytes
PLO_DEV unsigned perm_bytes(unsigned hi, unsigned lo, unsigned sel) { return __builtin_amdgcn_perm(hi, lo, sel); }
PLO_DEV unsigned align_bytes(unsigned hi, unsigned lo, unsigned shift) { return __builtin_amdgcn_alignbyte(hi, lo, shift); }
PLO_DEV int clz32(unsigned x) { return __builtin_clz(x); }  // x != 0
PLO_DEV int ctz32(unsigned x) { return __builtin_ctz(x); }  // x != 0

PLO_DEV int reduce_add(int x) { return bcast_last(scan_add(x)); }
PLO_DEV int reduce_max(int x) { return bcast_last(scan_max(x)); }

// Left-shift carry: functions f(r) = reset ? b : min(r + a, b) are closed under composition:
// (a1,b1,s1) then (a2,b2,s2) = s2 ? (a2,b2,1) : (a1+a2, min(b1+a2, b2), s1).  Inclusive scan of the composition.
struct MinPlus {
    int a, b, s;
};
PLO_DEV int sat_add(int x, int y) {  // x, y >= 0 (match lengths, homologies, IMAX): the unsigned sum cannot wrap
    unsigned t = (unsigned)x + (unsigned)y;
    return (int)(t < 0x7fffffffu ? t : 0x7fffffffu);
}
PLO_DEV MinPlus mp_compose(MinPlus p, MinPlus c) {  // apply p first, then c (branch-free)
    MinPlus r;
    r.a = c.s ? c.a : sat_add(p.a, c.a);
    r.b = c.s ? c.b : imin(sat_add(p.b, c.a), c.b);
    r.s = c.s ? c.s : p.s;
    return r;
}
PLO_DEV MinPlus scan_minplus(MinPlus x) {
#define PLO_MP_STEP(ctrl, rmask)                                  \
    {                                                             \
        MinPlus p;                                                \
        p.a = PLO_DPP(0, x.a, ctrl, rmask);                       \
        p.b = PLO_DPP(0x7fffffff, x.b, ctrl, rmask);              \
        p.s = PLO_DPP(0, x.s, ctrl, rmask);                       \
        x = mp_compose(p, x);                                     \
    }
    PLO_MP_STEP(0x111, 0xf)
    PLO_MP_STEP(0x112, 0xf)
    PLO_MP_STEP(0x114, 0xf)
    PLO_MP_STEP(0x118, 0xf)
    PLO_MP_STEP(0x142, 0xa)
    PLO_MP_STEP(0x143, 0xc)
#undef PLO_MP_STEP
    return x;
}
PLO_DEV MinPlus bcast_last(MinPlus v) {
    MinPlus r;
    r.a = bcast_last(v.a);
    r.b = bcast_last(v.b);
    r.s = bcast_last(v.s);
    return r;
}

// atomics on wave-private LDS / scratch words (lanes of the same wave may collide)
PLO_DEV void atomic_add(int *p, int v) { atomicAdd(p, v); }
PLO_DEV void atomic_add(unsigned *p, unsigned v) { atomicAdd(p, v); }
PLO_DEV void atomic_min(int *p, int v) { atomicMin(p, v); }
PLO_DEV void atomic_max(int *p, int v) { atomicMax(p, v); }
PLO_DEV void atomic_or(int *p, int v) { atomicOr(p, v); }
// device-scope atomics on global counters shared by all waves
PLO_DEV unsigned long long atomic_add_global(unsigned long long *p, unsigned long long v) { return atomicAdd(p, v); }
PLO_DEV unsigned atomic_add_global(unsigned *p, unsigned v) { return atomicAdd(p, v); }

}  // namespace wv
