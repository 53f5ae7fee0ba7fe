// Calibration of the HBM byte counters for the access pattern of the per-lane walks: every lane
// reads 16 bytes at random places of a table far larger than the caches, so (almost) every load
// misses everywhere and the lines the kernel asks memory for are known: lanes x loads.  A second
// kernel streams the same table with wide coalesced loads, the pattern the counters are known for.
// Run under rocprofv3 --pmc FETCH_SIZE (and TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_MISS_sum
// TCC_REQ_sum in a second pass) and compare with the counts this program prints.
// Build: hipcc --offload-arch=gfx950 -O2 tools/fetch_calib.hip -o tools/fetch_calib
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>

typedef uint32_t u4 __attribute__((ext_vector_type(4)));

constexpr int LOADS = 256;                     // per lane

__global__ void __launch_bounds__(256) k_random16(const u4* __restrict__ table, size_t n_slots, uint32_t* out) {
    const uint32_t tid = blockIdx.x * 256 + threadIdx.x;
    uint64_t s = (uint64_t)tid * 0x9E3779B97F4A7C15ull + 12345;
    uint32_t acc = 0;
    for (int i = 0; i < LOADS; i += 4) {           // four independent loads in flight per lane
        u4 v[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            s = s * 6364136223846793005ull + 1442695040888963407ull;
            v[k] = table[(s >> 20) % n_slots];
        }
#pragma unroll
        for (int k = 0; k < 4; k++) acc ^= v[k].x ^ v[k].w;
    }
    out[tid] = acc;
}

__global__ void __launch_bounds__(256) k_stream16(const u4* __restrict__ table, size_t n_slots, uint32_t* out) {
    const size_t tid = (size_t)blockIdx.x * 256 + threadIdx.x, stride = (size_t)gridDim.x * 256;
    uint32_t acc = 0;
    for (size_t i = tid; i < n_slots; i += stride) { const u4 v = table[i]; acc ^= v.x ^ v.w; }
    out[tid] = acc;
}

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

// With table sizes (MiB) as arguments: the random-line rate per size, second of two runs -- what a
// table that fits the 256 MiB last-level cache (or the L2s) delivers against one that does not.
static int sweep(int argc, char** argv) {
    const int blocks = 256 * 8 * 4;
    uint32_t* out;
    CHECK(hipMalloc(&out, (size_t)blocks * 256 * 4));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int a = 1; a < argc; a++) {
        const size_t bytes = (size_t)atol(argv[a]) << 20, n_slots = bytes / 16;
        u4* table;
        CHECK(hipMalloc(&table, bytes));
        CHECK(hipMemset(table, 1, bytes));
        float ms = 0;
        for (int rep = 0; rep < 2; rep++) {
            CHECK(hipEventRecord(e0));
            hipLaunchKernelGGL(k_random16, blocks, 256, 0, 0, table, n_slots, out);
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            CHECK(hipEventElapsedTime(&ms, e0, e1));
        }
        const double loads = (double)blocks * 256 * LOADS;
        printf("table %6ld MiB: %.3f ms, %.0f M random 16-byte lane loads/s = %.2f TB/s of 128-byte lines\n", atol(argv[a]), ms,
               loads / ms / 1e3, loads * 128 / ms / 1e9);
        CHECK(hipFree(table));
    }
    return 0;
}

int main(int argc, char** argv) {
    if (argc > 1) return sweep(argc, argv);
    const size_t bytes = 8ull << 30;               // 8 GiB: 32 x the 256 MiB last-level cache
    const size_t n_slots = bytes / 16;
    u4* table; uint32_t* out;
    CHECK(hipMalloc(&table, bytes));
    CHECK(hipMemset(table, 1, bytes));
    const int blocks = 256 * 8 * 4;
    CHECK(hipMalloc(&out, (size_t)blocks * 256 * 4));
    CHECK(hipDeviceSynchronize());
    hipEvent_t e0, e1, e2;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1)); CHECK(hipEventCreate(&e2));
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_random16, blocks, 256, 0, 0, table, n_slots, out);
    CHECK(hipEventRecord(e1));
    hipLaunchKernelGGL(k_stream16, blocks, 256, 0, 0, table, n_slots, out);
    CHECK(hipEventRecord(e2));
    CHECK(hipEventSynchronize(e2));
    float ms_r = 0, ms_s = 0;
    CHECK(hipEventElapsedTime(&ms_r, e0, e1));
    CHECK(hipEventElapsedTime(&ms_s, e1, e2));
    const double loads = (double)blocks * 256 * LOADS;
    printf("k_random16: %.0f lane loads of 16 bytes at random = %.0f lines of 64 B = %.3f GB as 64-byte lines (%.3f GB as 128-byte pairs), %.3f ms: %.0f M lines/s, %.2f TB/s of 64-byte lines\n",
           loads, loads, loads * 64 / 1e9, loads * 128 / 1e9, ms_r, loads / ms_r / 1e3, loads * 64 / ms_r / 1e9);
    printf("k_stream16: %.3f GB streamed, %.3f ms: %.2f TB/s\n", bytes / 1e9, ms_s, bytes / ms_s / 1e9);
    return 0;
}
