// What a wave-wide global load costs the address unit of a CU on this GPU, by load width, by which
// lanes take part and by how the addresses fall: cycles of one CU per load instruction with up to 8
// resident waves per SIMD, every wave keeping 8 independent loads in flight.  The table the lanes
// read is 16 KB, so every access is an L1 hit and the figure is the address path's, not the
// memory's.  Build: hipcc --offload-arch=gfx950 -O2 tools/ta_rate.hip -o tools/ta_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>

constexpr int ITERS = 2048;          // x 8 loads per iteration
constexpr int TABLE_BYTES = 16384;

typedef uint32_t u4 __attribute__((ext_vector_type(4)));
typedef uint32_t u2 __attribute__((ext_vector_type(2)));

// WIDTH: 1, 2, 4 dwords.  LDS: read the same table out of LDS instead (ds_read_b128).
template <int WIDTH, bool LDS>
__global__ void __launch_bounds__(256) k_loads(const uint8_t* table, uint32_t* out, unsigned long long* clk,
                                               unsigned long long lane_mask, int addr_mode) {
    __shared__ __attribute__((aligned(16))) uint8_t s_table[TABLE_BYTES];
    if (LDS) {
        for (int i = threadIdx.x; i < TABLE_BYTES / 16; i += blockDim.x) ((u4*)s_table)[i] = ((const u4*)table)[i];
        __syncthreads();
    }
    const unsigned lane = threadIdx.x & 63;
    const bool on = (lane_mask >> lane) & 1;
    // addr_mode 0: consecutive lanes read consecutive 16-byte slots; 1: a different 64-byte line per lane;
    // 2: pseudo-random slots, re-drawn every iteration
    uint32_t slot = addr_mode == 0 ? lane : addr_mode == 1 ? lane * 4 : lane * 2654435761u >> 22;
    uint32_t acc = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    if (on) {
        for (int i = 0; i < ITERS; i++) {
            uint32_t offs[8];
#pragma unroll
            for (int k = 0; k < 8; k++) {
                if (addr_mode == 2) slot = slot * 1664525u + 1013904223u;
                offs[k] = ((addr_mode == 2 ? slot >> 10 : slot + k * 256 + i * 8) & (TABLE_BYTES / 16 - 1)) * 16;
            }
            // all eight loads and the wait are ONE asm statement with early-clobber results: the compiler
            // does not know that a load's result arrives later, and would otherwise reuse a result
            // register (as the next address, say) while the load is still in flight
#define LOADS8(INSTR, WAIT, BASE_CONSTRAINT, BASE)                                                             \
            asm volatile(INSTR " %0, %8" BASE_CONSTRAINT "\n" INSTR " %1, %9" BASE_CONSTRAINT "\n"              \
                         INSTR " %2, %10" BASE_CONSTRAINT "\n" INSTR " %3, %11" BASE_CONSTRAINT "\n"            \
                         INSTR " %4, %12" BASE_CONSTRAINT "\n" INSTR " %5, %13" BASE_CONSTRAINT "\n"            \
                         INSTR " %6, %14" BASE_CONSTRAINT "\n" INSTR " %7, %15" BASE_CONSTRAINT "\n" WAIT       \
                         : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]),        \
                           "=&v"(v[6]), "=&v"(v[7])                                                            \
                         : "v"(offs[0]), "v"(offs[1]), "v"(offs[2]), "v"(offs[3]), "v"(offs[4]), "v"(offs[5]), \
                           "v"(offs[6]), "v"(offs[7]), "s"(BASE) : "memory")
            if (LDS) {
                u4 v[8];
                const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)s_table;
#pragma unroll
                for (int k = 0; k < 8; k++) offs[k] += lds_base;
                LOADS8("ds_read_b128", "s_waitcnt lgkmcnt(0)", "", 0);
#pragma unroll
                for (int k = 0; k < 8; k++) acc ^= v[k].x ^ v[k].w;
            } else if (WIDTH == 4) {
                u4 v[8];
                LOADS8("global_load_dwordx4", "s_waitcnt vmcnt(0)", ", %16", table);
#pragma unroll
                for (int k = 0; k < 8; k++) acc ^= v[k].x ^ v[k].w;
            } else if (WIDTH == 2) {
                u2 v[8];
                LOADS8("global_load_dwordx2", "s_waitcnt vmcnt(0)", ", %16", table);
#pragma unroll
                for (int k = 0; k < 8; k++) acc ^= v[k].x ^ v[k].y;
            } else {
                uint32_t v[8];
                LOADS8("global_load_dword", "s_waitcnt vmcnt(0)", ", %16", table);
#pragma unroll
                for (int k = 0; k < 8; k++) acc ^= v[k];
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if (threadIdx.x == 0) { clk[blockIdx.x * 2] = t1 - t0; clk[blockIdx.x * 2 + 1] = r1 - r0; }
}

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const int blocks = cus * 8;                       // 8 workgroups of 4 waves per CU: 8 waves per SIMD
    uint8_t* table; uint32_t* out; unsigned long long* clk;
    CHECK(hipMalloc(&table, TABLE_BYTES));
    CHECK(hipMemset(table, 1, TABLE_BYTES));
    CHECK(hipMalloc(&out, (size_t)blocks * 256 * 4));
    CHECK(hipMalloc(&clk, (size_t)blocks * 16));
    struct Mask { const char* name; unsigned long long m; };
    const Mask masks[] = {
        {"64 lanes", ~0ull},
        {"32 lanes, low half", 0xffffffffull},
        {"32 lanes, every other", 0x5555555555555555ull},
        {"16 lanes, low quarter", 0xffffull},
        {"16 lanes, every fourth", 0x1111111111111111ull},
        {"16 lanes, 4 of every 16", 0x000f000f000f000full},
        {"8 lanes, low", 0xffull},
        {"8 lanes, every eighth", 0x0101010101010101ull},
    };
    const char* addr_names[] = {"consecutive 16-byte slots", "one 64-byte line per lane", "random slots"};
    printf("device: %s, %d CUs; %d workgroups of 256, %d x 8 loads per wave\n", prop.gcnArchName, cus, blocks, ITERS);
    printf("%-10s %-28s %-26s %12s %14s\n", "load", "addresses", "lanes", "ms", "CU cycles/load");
    std::vector<unsigned long long> h((size_t)blocks * 2);
    for (int kind = 0; kind < 4; kind++) {
        for (int am = 0; am < 3; am++) {
            for (const Mask& mk : masks) {
                if (kind != 2 && am != 2) continue;                         // the full grid for dwordx4 only
                if (kind != 2 && mk.m != ~0ull && mk.m != 0xffffull && mk.m != 0x1111111111111111ull) continue;
                hipEvent_t e0, e1;
                CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
                for (int rep = 0; rep < 2; rep++) {                         // first repetition warms up
                    CHECK(hipEventRecord(e0));
                    if (kind == 0) hipLaunchKernelGGL((k_loads<1, false>), blocks, 256, 0, 0, table, out, clk, mk.m, am);
                    if (kind == 1) hipLaunchKernelGGL((k_loads<2, false>), blocks, 256, 0, 0, table, out, clk, mk.m, am);
                    if (kind == 2) hipLaunchKernelGGL((k_loads<4, false>), blocks, 256, 0, 0, table, out, clk, mk.m, am);
                    if (kind == 3) hipLaunchKernelGGL((k_loads<4, true>), blocks, 256, 0, 0, table, out, clk, mk.m, am);
                    CHECK(hipEventRecord(e1));
                    CHECK(hipEventSynchronize(e1));
                }
                float ms = 0;
                CHECK(hipEventElapsedTime(&ms, e0, e1));
                CHECK(hipMemcpy(h.data(), clk, h.size() * 8, hipMemcpyDeviceToHost));
                // shader clock from the ratio of the two counters of workgroup 0 (s_memrealtime runs at 100 MHz)
                const double ghz = (double)h[0] / ((double)h[1] * 10.0);
                const double loads_per_cu = 32.0 * ITERS * 8;               // wave-wide load instructions
                // (by wall time: the workgroups of a launch are not all resident from its start to its end, so a
                // workgroup's own cycle count undercounts; rocprofv3's TA_TA_BUSY per load agrees with this figure)
                const double cyc = ms * 1e-3 * ghz * 1e9 / loads_per_cu;
                const char* names[] = {"dword", "dwordx2", "dwordx4", "ds_b128"};
                printf("%-10s %-28s %-26s %12.3f %14.2f   (%.2f GHz)\n", names[kind], addr_names[am], mk.name, ms, cyc, ghz);
                CHECK(hipEventDestroy(e0)); CHECK(hipEventDestroy(e1));
            }
        }
    }
    return 0;
}
