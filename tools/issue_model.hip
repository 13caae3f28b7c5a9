// issue_model.hip -- how does a gfx950 SIMD issue the instruction mix of the lane kernels?  ("with SALU and VALU
// co-issuing from different waves the hardware floor is the VALU-only 0.69 ms" against DESIGN section 6's "time follows the TOTAL
// instruction count")  One question, three loop bodies, 1 .. 8 waves per SIMD:
//   valu   64 dependent-free integer VALU instructions per trip (v_add_u32 / v_and_b32 / v_cndmask_b32 on eight chains)
//   salu   64 SALU instructions per trip (s_add_u32 / s_and_b64 / s_or_b64 on four chains)
//   mix    the two interleaved 1:1 (32 + 32), the way hipcc lays out the lane kernels' flag arithmetic
//   mixlds mix + 2 ds_read_b32 + 1 ds_write_b32 per trip (the lane kernels' LDS share)
// Every wave runs the same trip count; cycles per trip per SIMD = kernel time x clock / (trips x waves per SIMD).  If scalar
// instructions of one wave issue beside vector instructions of another, `mix` at >= 2 waves per SIMD costs what its VALU half costs
// alone; if the SIMD issues one instruction per turn whatever its class, it costs the sum.
// usage (GPU box): hipcc --offload-arch=gfx950 -O3 -o tools/issue_model tools/issue_model.hip && tools/issue_model
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x)                                                    \
    do {                                                            \
        hipError_t e_ = (x);                                        \
        if (e_ != hipSuccess) {                                     \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); \
            exit(1);                                                \
        }                                                           \
    } while (0)

#define V8(a, b, c, d, e, f, g, h, k)                 \
    asm volatile("v_add_u32 %0, %0, %8\n"             \
                 "v_and_b32 %1, %1, %8\n"             \
                 "v_add_u32 %2, %2, %8\n"             \
                 "v_cndmask_b32 %3, %3, %8, vcc\n"    \
                 "v_add_u32 %4, %4, %8\n"             \
                 "v_or_b32 %5, %5, %8\n"              \
                 "v_add_u32 %6, %6, %8\n"             \
                 "v_cndmask_b32 %7, %7, %8, vcc\n"    \
                 : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) : "v"(k) : "vcc")
#define S8(p, q, r, s, m0, m1)                        \
    asm volatile("s_add_u32 %0, %0, 1\n"              \
                 "s_and_b64 %4, %4, %5\n"             \
                 "s_add_u32 %1, %1, 3\n"              \
                 "s_or_b64 %5, %5, %4\n"              \
                 "s_add_u32 %2, %2, 5\n"              \
                 "s_xor_b64 %4, %4, %5\n"             \
                 "s_add_u32 %3, %3, 7\n"              \
                 "s_andn2_b64 %5, %5, %4\n"           \
                 : "+s"(p), "+s"(q), "+s"(r), "+s"(s), "+s"(m0), "+s"(m1) : : "scc")
#define M8(a, b, c, d, k, p, q, m0, m1)               \
    asm volatile("v_add_u32 %0, %0, %8\n"             \
                 "s_add_u32 %4, %4, 1\n"              \
                 "v_and_b32 %1, %1, %8\n"             \
                 "s_and_b64 %6, %6, %7\n"             \
                 "v_add_u32 %2, %2, %8\n"             \
                 "s_add_u32 %5, %5, 3\n"              \
                 "v_cndmask_b32 %3, %3, %8, vcc\n"    \
                 "s_or_b64 %7, %7, %6\n"              \
                 : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+s"(p), "+s"(q), "+s"(m0), "+s"(m1) : "v"(k) : "vcc", "scc")

template <int MODE>
__global__ __launch_bounds__(256) void k_issue(uint32_t trips, uint32_t *sink, uint32_t lds_pad) {
    extern __shared__ uint32_t lds[];
    uint32_t a = threadIdx.x, b = a * 3, c = a * 5, d = a * 7, e = a * 11, f = a * 13, g = a * 17, h = a * 19, k = blockIdx.x | 1u;
    uint32_t p = blockIdx.x, q = p + 1, r = p + 2, s = p + 3;
    uint64_t m0 = 0x5555555555555555ull ^ blockIdx.x, m1 = 0x3333333333333333ull;
    uint32_t *my = lds + threadIdx.x;
    my[0] = a;
    for (uint32_t t = 0; t < trips; ++t) {
        if (MODE == 0) {
#pragma unroll
            for (int u = 0; u < 8; ++u) V8(a, b, c, d, e, f, g, h, k);
        } else if (MODE == 1) {
#pragma unroll
            for (int u = 0; u < 8; ++u) S8(p, q, r, s, m0, m1);
        } else {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (u & 1) M8(a, b, c, d, k, p, q, m0, m1);
                else M8(e, f, g, h, k, r, s, m0, m1);
            }
            if (MODE == 3) {
                uint32_t x = my[0], y = my[256];
                my[512] = x + y + a;
            }
        }
    }
    uint32_t acc = a ^ b ^ c ^ d ^ e ^ f ^ g ^ h ^ p ^ q ^ r ^ s ^ (uint32_t)m0 ^ (uint32_t)m1;
    if (acc == 0x12345678u) *sink = acc + lds[lds_pad & 1023u];
}

int main() {
    hipDeviceProp_t pr;
    CHECK(hipGetDeviceProperties(&pr, 0));
    const int cus = pr.multiProcessorCount;
    const double ghz = pr.clockRate / 1e6;
    uint32_t *sink;
    CHECK(hipMalloc(&sink, 4));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const uint32_t trips = 20000;
    const char *names[4] = {"valu", "salu", "mix", "mixlds"};
    printf("{\"device\": \"%s\", \"cus\": %d, \"clock_ghz\": %.3f, \"instructions_per_trip\": 64, \"trips\": %u, \"rows\": [\n", pr.name, cus, ghz, trips);
    bool first = true;
    for (int mode = 0; mode < 4; ++mode)
        for (int w = 1; w <= 8; ++w) {  // workgroups of 4 waves (one per SIMD) per CU; LDS sized so that exactly w of them fit a CU
            size_t lds_bytes = (160 * 1024 / w) & ~(size_t)255;
            if (lds_bytes < 4096) lds_bytes = 4096;
            if (lds_bytes > 64 * 1024) {
                auto set = [&](auto kern) { CHECK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes)); };
                if (mode == 0) set(k_issue<0>);
                if (mode == 1) set(k_issue<1>);
                if (mode == 2) set(k_issue<2>);
                if (mode == 3) set(k_issue<3>);
            }
            float best = 1e30f;
            for (int rep = 0; rep < 3; ++rep) {
                CHECK(hipEventRecord(e0, 0));
                dim3 grid(cus * w), block(256);
                if (mode == 0) hipLaunchKernelGGL(k_issue<0>, grid, block, lds_bytes, 0, trips, sink, 0u);
                if (mode == 1) hipLaunchKernelGGL(k_issue<1>, grid, block, lds_bytes, 0, trips, sink, 0u);
                if (mode == 2) hipLaunchKernelGGL(k_issue<2>, grid, block, lds_bytes, 0, trips, sink, 0u);
                if (mode == 3) hipLaunchKernelGGL(k_issue<3>, grid, block, lds_bytes, 0, trips, sink, 0u);
                CHECK(hipEventRecord(e1, 0));
                CHECK(hipEventSynchronize(e1));
                float ms;
                CHECK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < best) best = ms;
            }
            const double cyc_per_trip_simd = best * 1e-3 * ghz * 1e9 / ((double)trips * w);
            printf("%s {\"body\": \"%s\", \"waves_per_simd\": %d, \"ms\": %.4f, \"cycles_per_trip_per_wave_slot\": %.1f, \"cycles_per_instruction_per_simd\": %.3f}", first ? " " : ",",
                   names[mode], w, best, cyc_per_trip_simd, cyc_per_trip_simd / 64.0);
            printf("\n");
            first = false;
        }
    printf("]}\n");
    return 0;
}
