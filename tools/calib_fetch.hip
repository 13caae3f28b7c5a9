// calib_fetch.hip -- what does rocprofv3's FETCH_SIZE count for the lift kernels' access patterns?  (missing #5 / next #4a)
// MI355X_MICROARCH.md: FETCH_SIZE = TCC_EA0_RDREQ x 64 B and reports exactly HALF the bytes of a wide coalesced streaming read (the
// L2 asks the fabric for 128 bytes and tallies 64); "other access widths are uncalibrated: calibrate on a known byte count in your
// own access pattern".  The lift kernels' traffic is dominated by SCATTERED 16-byte-per-lane loads (homology probes: 16 + 4 bytes of
// a contig and of a read at unrelated addresses).  Four kernels, each with a known number of touched 128-byte lines:
//   k_stream      every lane 16 bytes, consecutive lanes consecutive addresses (the guide's calibrated case: expect 64 B per line)
//   k_scatter16   every lane 16 bytes at the START of a distinct pseudo-random 128-byte line
//   k_scatter_2h  every lane 2 x 16 bytes, at offsets 0 and 64 of its distinct line (both 64-byte halves)
//   k_probe20     every lane 16 + 4 bytes at a pseudo-random 4-byte-aligned offset of its distinct line (the probes' shape; a quarter
//                 of them straddle the line's halves, one in 32 the next line)
// One request per line, tallied at 64 B, whatever part of the line is touched => the L2 fetches whole 128-byte lines and the factor
// is 2 for scattered accesses too (counter x 2 = bytes moved).  Two requests for k_scatter_2h => 64-byte requests, factor 1.
// usage: rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "k_(stream|scatter|probe)" ... -- tools/calib_fetch   (tools/calib_fetch.sh)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x)                                                                  \
    do {                                                                          \
        hipError_t e_ = (x);                                                      \
        if (e_ != hipSuccess) {                                                   \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));               \
            exit(1);                                                              \
        }                                                                         \
    } while (0)

typedef uint32_t u4 __attribute__((ext_vector_type(4)));

// a permutation of [0, n_lines) (n_lines a power of two): odd multiplier + offset
__device__ __forceinline__ uint64_t line_of(uint64_t i, uint64_t mask) { return (i * 0x9E3779B97F4A7C15ull + 0x7F4A7C15ull) & mask; }

__global__ __launch_bounds__(256) void k_stream(const u4 *p, uint64_t n16, uint32_t *sink) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t acc = 0;
    for (; i < n16; i += (uint64_t)gridDim.x * blockDim.x) {
        u4 v = p[i];
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) *sink = acc;
}
__global__ __launch_bounds__(256) void k_scatter16(const uint8_t *p, uint64_t n_loads, uint64_t mask, uint32_t *sink) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t acc = 0;
    for (; i < n_loads; i += (uint64_t)gridDim.x * blockDim.x) {
        u4 v = *(const u4 *)(p + line_of(i, mask) * 128);
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) *sink = acc;
}
__global__ __launch_bounds__(256) void k_scatter_2h(const uint8_t *p, uint64_t n_loads, uint64_t mask, uint32_t *sink) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t acc = 0;
    for (; i < n_loads; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint8_t *q = p + line_of(i, mask) * 128;
        u4 v = *(const u4 *)q, w = *(const u4 *)(q + 64);
        acc ^= v.x ^ v.y ^ v.z ^ v.w ^ w.x ^ w.y ^ w.z ^ w.w;
    }
    if (acc == 0x12345678u) *sink = acc;
}
__global__ __launch_bounds__(256) void k_probe20(const uint8_t *p, uint64_t n_loads, uint64_t mask, uint32_t *sink) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t acc = 0;
    for (; i < n_loads; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t l = line_of(i, mask);
        const uint8_t *q = p + l * 128 + ((l >> 7) & 31) * 4;  // 4-byte-aligned offset 0 .. 124 inside the line
        typedef uint32_t u4u __attribute__((ext_vector_type(4), aligned(4)));
        u4u v = *(const u4u *)q;
        uint32_t t = *(const uint32_t *)(q + 16);
        acc ^= v.x ^ v.y ^ v.z ^ v.w ^ t;
    }
    if (acc == 0x12345678u) *sink = acc;
}

int main(int argc, char **argv) {
    const uint64_t gib = argc > 1 ? strtoull(argv[1], nullptr, 10) : 8;  // buffer size: well past the 256 MiB Infinity Cache
    const uint64_t bytes = gib << 30, n_lines = bytes / 128, mask = n_lines - 1;
    const uint64_t n_loads = argc > 2 ? strtoull(argv[2], nullptr, 10) : (1ull << 24);  // distinct lines touched per scattered launch (2 GiB of lines)
    if (n_lines & (n_lines - 1)) {
        fprintf(stderr, "buffer size must be a power of two GiB\n");
        return 1;
    }
    uint8_t *buf;
    uint32_t *sink;
    CHECK(hipMalloc(&buf, bytes + 256));
    CHECK(hipMalloc(&sink, 4));
    CHECK(hipMemset(buf, 1, bytes + 256));
    CHECK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const int grid = 256 * 16;
    const uint64_t stream_bytes = 2ull << 30;  // k_stream reads the first 2 GiB
    auto timed = [&](const char *name, uint64_t lines, uint64_t useful, auto launch) {
        for (int rep = 0; rep < 3; ++rep) {  // (every repetition shows up as one dispatch in the counter output)
            CHECK(hipEventRecord(e0));
            launch();
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float ms = 0;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            printf("{\"kernel\": \"%s\", \"rep\": %d, \"lines_128B\": %llu, \"useful_bytes\": %llu, \"ms\": %.4f, \"lines_per_s\": %.4g, \"GBps_at_128B_per_line\": %.1f}\n", name,
                   rep, (unsigned long long)lines, (unsigned long long)useful, ms, lines / (ms * 1e-3), lines * 128.0 / (ms * 1e-3) / 1e9);
        }
    };
    timed("k_stream", stream_bytes / 128, stream_bytes, [&] { hipLaunchKernelGGL(k_stream, dim3(grid), dim3(256), 0, 0, (const u4 *)buf, stream_bytes / 16, sink); });
    timed("k_scatter16", n_loads, n_loads * 16, [&] { hipLaunchKernelGGL(k_scatter16, dim3(grid), dim3(256), 0, 0, buf, n_loads, mask, sink); });
    timed("k_scatter_2h", n_loads, n_loads * 32, [&] { hipLaunchKernelGGL(k_scatter_2h, dim3(grid), dim3(256), 0, 0, buf, n_loads, mask, sink); });
    timed("k_probe20", n_loads, n_loads * 20, [&] { hipLaunchKernelGGL(k_probe20, dim3(grid), dim3(256), 0, 0, buf, n_loads, mask, sink); });
    CHECK(hipFree(buf));
    CHECK(hipFree(sink));
    return 0;
}
