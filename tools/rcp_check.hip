// Exhaustive check: is the short reciprocal sequence below bit-identical to the correctly
// rounded 1.0f / x for every f32 x in the range the triangle test can see?
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/rcp_check.hip -o /tmp/rcp_check
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__device__ __forceinline__ float rcp_short(float x) {
    float r = __builtin_amdgcn_rcpf(x);
    float e = __builtin_fmaf(-x, r, 1.0f);
    r = __builtin_fmaf(e, r, r);
    e = __builtin_fmaf(-x, r, 1.0f);
    r = __builtin_fmaf(e, r, r);
    return r;
}

__global__ void check(unsigned long long* bad_in, unsigned long long* bad_out, unsigned long long* n_in, uint32_t* first_bad) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    unsigned long long bi = 0, bo = 0, ni = 0;
    for (uint64_t u = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; u < (1ull << 32); u += stride) {
        const float x = __uint_as_float((uint32_t)u);
        const float want = 1.0f / x;
        const float got = rcp_short(x);
        const float ax = fabsf(x);
        const bool in_range = ax >= 1.1920929e-7f && ax <= 8.5e37f;   // |det| >= eps, 1/det still normal
        const bool same = __float_as_uint(want) == __float_as_uint(got);
        if (in_range) { ni++; if (!same) { bi++; atomicMin(first_bad, (uint32_t)u); } }
        else if (!same && x == x) bo++;
    }
    atomicAdd(bad_in, bi); atomicAdd(bad_out, bo); atomicAdd(n_in, ni);
}

int main() {
    unsigned long long *d, h[3] = {0, 0, 0};
    uint32_t* fb; uint32_t hfb = 0xFFFFFFFFu;
    hipMalloc(&d, 24); hipMemset(d, 0, 24);
    hipMalloc(&fb, 4); hipMemcpy(fb, &hfb, 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(check, dim3(4096), dim3(256), 0, 0, d, d + 1, d + 2, fb);
    hipDeviceSynchronize();
    hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
    hipMemcpy(&hfb, fb, 4, hipMemcpyDeviceToHost);
    printf("in-range inputs %llu, mismatches in range %llu (first 0x%08x), mismatches outside range %llu\n", h[2], h[0], hfb, h[1]);
    return h[0] == 0 ? 0 : 1;
}
