// Issue cost of the vector instructions the sweep kernel is made of, on this GPU: cycles of one
// SIMD per wave64 instruction with 1, 2, 4 and 8 resident waves per SIMD, each wave running a
// long stream of independent instructions of one kind.  Build: hipcc --offload-arch=gfx950 -O2
// tools/valu_rate.hip -o tools/valu_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

#define REP8(x) x x x x x x x x
constexpr int ITERS = 4096;    // x 64 instructions per iteration

template <int KIND>
__global__ void __launch_bounds__(64) k_rate(float* out, unsigned long long* clk, float seed, const float* tab) {
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float b0 = a0 * 0.5f, b1 = a1 * 0.5f, b2 = a2 * 0.5f, b3 = a3 * 0.5f, b4 = a4 * 0.5f, b5 = a5 * 0.5f, b6 = a6 * 0.5f, b7 = a7 * 0.5f;
    const float m = 1.0000001f;
    unsigned long long mask = 0x5555555555555555ull;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < ITERS; i++) {
        if (KIND == 0) {         // v_fma_f32
            REP8(asm volatile("v_fma_f32 %0, %0, %8, %8\n v_fma_f32 %1, %1, %8, %8\n v_fma_f32 %2, %2, %8, %8\n v_fma_f32 %3, %3, %8, %8\n"
                              "v_fma_f32 %4, %4, %8, %8\n v_fma_f32 %5, %5, %8, %8\n v_fma_f32 %6, %6, %8, %8\n v_fma_f32 %7, %7, %8, %8"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m));)
        } else if (KIND == 1) {  // v_pk_mul_f32 on register pairs
            typedef float f2 __attribute__((ext_vector_type(2)));
            f2 p0 = {a0, b0}, p1 = {a1, b1}, p2 = {a2, b2}, p3 = {a3, b3}, p4 = {a4, b4}, p5 = {a5, b5}, p6 = {a6, b6}, p7 = {a7, b7}, mm = {m, m};
            REP8(asm volatile("v_pk_mul_f32 %0, %0, %8\n v_pk_mul_f32 %1, %1, %8\n v_pk_mul_f32 %2, %2, %8\n v_pk_mul_f32 %3, %3, %8\n"
                              "v_pk_mul_f32 %4, %4, %8\n v_pk_mul_f32 %5, %5, %8\n v_pk_mul_f32 %6, %6, %8\n v_pk_mul_f32 %7, %7, %8"
                              : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(mm));)
            a0 = p0.x; a1 = p1.x; a2 = p2.x; a3 = p3.x; a4 = p4.x; a5 = p5.x; a6 = p6.x; a7 = p7.x;
            b0 = p0.y; b1 = p1.y; b2 = p2.y; b3 = p3.y; b4 = p4.y; b5 = p5.y; b6 = p6.y; b7 = p7.y;
        } else if (KIND == 2) {  // v_cndmask_b32 with the lane mask in SGPRs
            REP8(asm volatile("v_cndmask_b32_e64 %0, %0, %8, %9\n v_cndmask_b32_e64 %1, %1, %8, %9\n v_cndmask_b32_e64 %2, %2, %8, %9\n v_cndmask_b32_e64 %3, %3, %8, %9\n"
                              "v_cndmask_b32_e64 %4, %4, %8, %9\n v_cndmask_b32_e64 %5, %5, %8, %9\n v_cndmask_b32_e64 %6, %6, %8, %9\n v_cndmask_b32_e64 %7, %7, %8, %9"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "s"(mask));)
        } else if (KIND == 3) {  // v_cmp_lt_f32 writing a lane mask to SGPRs
            unsigned long long m0, m1, m2, m3;
            REP8(asm volatile("v_cmp_lt_f32_e64 %0, %4, %5\n v_cmp_lt_f32_e64 %1, %5, %6\n v_cmp_lt_f32_e64 %2, %6, %7\n v_cmp_lt_f32_e64 %3, %7, %4\n"
                              "v_cmp_lt_f32_e64 %0, %4, %6\n v_cmp_lt_f32_e64 %1, %5, %7\n v_cmp_lt_f32_e64 %2, %6, %4\n v_cmp_lt_f32_e64 %3, %7, %5"
                              : "=&s"(m0), "=&s"(m1), "=&s"(m2), "=&s"(m3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3));)
            mask ^= m0 ^ m1 ^ m2 ^ m3;
        } else if (KIND == 4) {  // v_min3_f32
            REP8(asm volatile("v_min3_f32 %0, %0, %8, %8\n v_min3_f32 %1, %1, %8, %8\n v_min3_f32 %2, %2, %8, %8\n v_min3_f32 %3, %3, %8, %8\n"
                              "v_min3_f32 %4, %4, %8, %8\n v_min3_f32 %5, %5, %8, %8\n v_min3_f32 %6, %6, %8, %8\n v_min3_f32 %7, %7, %8, %8"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m));)
        } else if (KIND == 6) {  // s_and_b64 (scalar ALU)
            unsigned long long s0 = mask, s1 = mask + 1, s2 = mask + 2, s3 = mask + 3, s4 = mask + 4, s5 = mask + 5, s6 = mask + 6, s7 = mask + 7;
            REP8(asm volatile("s_and_b64 %0, %0, %8\n s_and_b64 %1, %1, %8\n s_and_b64 %2, %2, %8\n s_and_b64 %3, %3, %8\n"
                              "s_and_b64 %4, %4, %8\n s_and_b64 %5, %5, %8\n s_and_b64 %6, %6, %8\n s_and_b64 %7, %7, %8"
                              : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3), "+s"(s4), "+s"(s5), "+s"(s6), "+s"(s7) : "s"(mask) : "scc");)
            mask ^= s0 ^ s1 ^ s2 ^ s3 ^ s4 ^ s5 ^ s6 ^ s7;
        } else if (KIND == 7) {  // alternating v_fma_f32 / s_and_b64 in one wave
            unsigned long long s0 = mask, s1 = mask + 1, s2 = mask + 2, s3 = mask + 3;
            REP8(asm volatile("v_fma_f32 %0, %0, %8, %8\n s_and_b64 %4, %4, %9\n v_fma_f32 %1, %1, %8, %8\n s_and_b64 %5, %5, %9\n"
                              "v_fma_f32 %2, %2, %8, %8\n s_and_b64 %6, %6, %9\n v_fma_f32 %3, %3, %8, %8\n s_and_b64 %7, %7, %9"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : "v"(m), "s"(mask) : "scc");)
            mask ^= s0 ^ s1 ^ s2 ^ s3;
        } else if (KIND == 8) {  // v_cmp -> SGPR mask -> v_cndmask that reads it (4 independent pairs)
            REP8(asm volatile("v_cmp_lt_f32_e64 s[20:21], %0, %4\n v_cmp_lt_f32_e64 s[22:23], %1, %4\n v_cmp_lt_f32_e64 s[24:25], %2, %4\n v_cmp_lt_f32_e64 s[26:27], %3, %4\n"
                              "v_cndmask_b32_e64 %0, %0, %4, s[20:21]\n v_cndmask_b32_e64 %1, %1, %4, s[22:23]\n v_cndmask_b32_e64 %2, %2, %4, s[24:25]\n v_cndmask_b32_e64 %3, %3, %4, s[26:27]"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m) : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");)
        } else if (KIND == 9) {  // 8 v_fma_f32, then a compare + uniform branch that is never taken
            REP8(asm volatile("v_fma_f32 %0, %0, %8, %8\n v_fma_f32 %1, %1, %8, %8\n v_fma_f32 %2, %2, %8, %8\n v_fma_f32 %3, %3, %8, %8\n"
                              "v_fma_f32 %4, %4, %8, %8\n v_fma_f32 %5, %5, %8, %8\n v_fma_f32 %6, %6, %8, %8\n v_fma_f32 %7, %7, %8, %8\n"
                              "s_cmp_eq_u32 %9, 0x12345\n s_cbranch_scc1 1f\n s_nop 0\n1:"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "s"(i) : "scc");)
        } else if (KIND == 10) { // v_cmp -> SGPR -> s_and (scalar ALU reads what the vector ALU wrote) -> v_cndmask
            REP8(asm volatile("v_cmp_lt_f32_e64 s[20:21], %0, %4\n v_cmp_lt_f32_e64 s[22:23], %1, %4\n v_cmp_lt_f32_e64 s[24:25], %2, %4\n v_cmp_lt_f32_e64 s[26:27], %3, %4\n"
                              "s_and_b64 s[20:21], s[20:21], s[22:23]\n s_and_b64 s[24:25], s[24:25], s[26:27]\n"
                              "v_cndmask_b32_e64 %0, %0, %4, s[20:21]\n v_cndmask_b32_e64 %2, %2, %4, s[24:25]"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m) : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "scc");)
        } else if (KIND == 11 || KIND == 13) { // s_load_dwordx16 (KIND 13: x8) of a 4 KB table, two loads in flight per wave
            typedef float f16v __attribute__((ext_vector_type(16)));
            typedef float f8v __attribute__((ext_vector_type(8)));
            for (int r = 0; r < 32; r++) {
                const unsigned off0 = ((i * 64 + r * 2) & 63) * 64, off1 = ((i * 64 + r * 2 + 1) & 63) * 64;
                if (KIND == 11) {
                    f16v e0, e1;
                    asm volatile("s_load_dwordx16 %0, %2, %3\n s_load_dwordx16 %1, %2, %4\n s_waitcnt lgkmcnt(0)"
                                 : "=&s"(e0), "=&s"(e1) : "s"(tab), "s"(off0), "s"(off1) : "memory");
                    a0 += e0[0] + e1[15];
                } else {
                    f8v e0, e1;
                    asm volatile("s_load_dwordx8 %0, %2, %3\n s_load_dwordx8 %1, %2, %4\n s_waitcnt lgkmcnt(0)"
                                 : "=&s"(e0), "=&s"(e1) : "s"(tab), "s"(off0), "s"(off1) : "memory");
                    a0 += e0[0] + e1[7];
                }
            }
        } else if (KIND == 12 || KIND == 14) { // the sweep's mix: one s_load_dwordx16 issued ahead of 64 v_fma_f32 (KIND 14: no load)
            typedef float f16v __attribute__((ext_vector_type(16)));
            f16v e0;
            const unsigned off0 = (i & 63) * 64;
            if (KIND == 12) asm volatile("s_load_dwordx16 %0, %1, %2" : "=&s"(e0) : "s"(tab), "s"(off0) : "memory");
            REP8(asm volatile("v_fma_f32 %0, %0, %8, %8\n v_fma_f32 %1, %1, %8, %8\n v_fma_f32 %2, %2, %8, %8\n v_fma_f32 %3, %3, %8, %8\n"
                              "v_fma_f32 %4, %4, %8, %8\n v_fma_f32 %5, %5, %8, %8\n v_fma_f32 %6, %6, %8, %8\n v_fma_f32 %7, %7, %8, %8"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m));)
            if (KIND == 12) { asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(e0)); a1 += e0[3]; }
        } else if (KIND == 5) {  // v_pk_mul_f32 with one scalar-register operand pair
            typedef float f2 __attribute__((ext_vector_type(2)));
            f2 p0 = {a0, b0}, p1 = {a1, b1}, p2 = {a2, b2}, p3 = {a3, b3}, p4 = {a4, b4}, p5 = {a5, b5}, p6 = {a6, b6}, p7 = {a7, b7};
            REP8(asm volatile("v_pk_mul_f32 %0, %0, %8 op_sel_hi:[1,0]\n v_pk_mul_f32 %1, %1, %8 op_sel_hi:[1,0]\n v_pk_mul_f32 %2, %2, %8 op_sel_hi:[1,0]\n v_pk_mul_f32 %3, %3, %8 op_sel_hi:[1,0]\n"
                              "v_pk_mul_f32 %4, %4, %8 op_sel_hi:[1,0]\n v_pk_mul_f32 %5, %5, %8 op_sel_hi:[1,0]\n v_pk_mul_f32 %6, %6, %8 op_sel_hi:[1,0]\n v_pk_mul_f32 %7, %7, %8 op_sel_hi:[1,0]"
                              : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "s"(mask));)
            a0 = p0.x; a1 = p1.x; a2 = p2.x; a3 = p3.x; a4 = p4.x; a5 = p5.x; a6 = p6.x; a7 = p7.x;
            b0 = p0.y; b1 = p1.y; b2 = p2.y; b3 = p3.y; b4 = p4.y; b5 = p5.y; b6 = p6.y; b7 = p7.y;
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + b0 + b1 + b2 + b3 + b4 + b5 + b6 + b7 + (float)(mask & 1);
    if (threadIdx.x == 0) { clk[blockIdx.x * 2] = t1 - t0; clk[blockIdx.x * 2 + 1] = r1 - r0; }
}

template <int KIND>
static void run(const char* name, int simds, float* out, unsigned long long* clk, const float* tab = nullptr) {
    for (int w : {1, 2, 4, 8}) {
        const int blocks = simds * w;
        hipLaunchKernelGGL(k_rate<KIND>, dim3(blocks), dim3(64), 0, 0, out, clk, 1.0f, tab);   // warm
        hipDeviceSynchronize();
        hipLaunchKernelGGL(k_rate<KIND>, dim3(blocks), dim3(64), 0, 0, out, clk, 1.0f, tab);
        hipDeviceSynchronize();
        static unsigned long long h[2 * 8192];
        hipMemcpy(h, clk, sizeof(unsigned long long) * 2 * blocks, hipMemcpyDeviceToHost);
        double cyc = 0, real = 0;
        for (int b = 0; b < blocks; b++) { cyc += (double)h[2 * b]; real += (double)h[2 * b + 1]; }
        cyc /= blocks; real /= blocks;
        const double per_rep = KIND == 9 ? 10.0 : 8.0;    // instructions in one of the 8 repeats (loads: 64 per iteration)
        const double n_inst = (double)ITERS * 8 * per_rep;
        if (KIND == 11 || KIND == 13) {
            const double bytes = (double)ITERS * 64 * (KIND == 11 ? 64 : 32);
            printf("%-28s waves/SIMD %d: %.1f cycles per load per wave, %.2f bytes per cycle per CU\n", name, w, cyc / (ITERS * 64.0),
                   bytes * 4 * w / cyc);
            continue;
        }
        printf("%-28s waves/SIMD %d: %.2f cycles per instruction per wave, %.2f SIMD cycles per instruction, clock %.2f GHz\n",
               name, w, cyc / n_inst, cyc / n_inst / w, cyc / real * 0.1);
    }
}

int main() {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int simds = p.multiProcessorCount * 4;
    float* out; unsigned long long* clk;
    hipMalloc(&out, sizeof(float) * 64 * simds * 8);
    hipMalloc(&clk, sizeof(unsigned long long) * 2 * simds * 8);
    run<0>("v_fma_f32", simds, out, clk);
    run<1>("v_pk_mul_f32", simds, out, clk);
    run<5>("v_pk_mul_f32 (sgpr operand)", simds, out, clk);
    run<2>("v_cndmask_b32 (sgpr mask)", simds, out, clk);
    run<3>("v_cmp_lt_f32 -> sgpr", simds, out, clk);
    run<4>("v_min3_f32", simds, out, clk);
    run<6>("s_and_b64", simds, out, clk);
    run<7>("v_fma_f32 + s_and_b64 mix", simds, out, clk);
    run<8>("v_cmp->sgpr->v_cndmask", simds, out, clk);
    run<9>("8 v_fma + cmp + branch", simds, out, clk);
    run<10>("v_cmp->s_and->v_cndmask", simds, out, clk);
    float* tab;
    hipMalloc(&tab, 4096 + 64);
    hipMemset(tab, 0, 4096 + 64);
    run<11>("s_load_dwordx16 (4 KB table)", simds, out, clk, tab);
    run<13>("s_load_dwordx8 (4 KB table)", simds, out, clk, tab);
    run<14>("64 v_fma per iteration", simds, out, clk, tab);
    run<12>("64 v_fma + 1 s_load_dwordx16", simds, out, clk, tab);
    return 0;
}
