// The wide tree of the opt-in RT_TRAVERSAL_WIDE walk built ON THE GPU (no counterpart in the
// reference, which builds its binary trees on the host: Bvh::new shape.rs:94-138, Mesh::build_bvh
// shape.rs:264-290).  rt_wide.cpp collapses the host-built binary trees; this file ignores them
// and builds the eight-child nodes directly from the scene's primitives:
//
//   1. every primitive entry of the device's entry array gets its padded box and a 30-bit Morton
//      code of the box centre inside the scene's centre bounds (x in bit 0, y in bit 1, z in bit 2
//      of every digit), keyed together with its entry index (64-bit keys, so the order is total);
//   2. the keys are sorted (rocPRIM radix sort through hipCUB);
//   3. top down, one launch per level: a node is a run of sorted keys, its children are the runs
//      of equal next Morton digit -- the digit IS the child slot, because bit k of a slot means "the
//      positive side along axis k" in the walk's ordering rule -- digits on which the whole run
//      agrees are skipped, runs that agree on all thirty bits are cut into equal parts, a run of
//      one key is a primitive child.  The node children of a node take consecutive slots from a
//      counter, from its lowest to its highest occupied child slot (child slot s lives at
//      child_base + s, so only the empty slots between two occupied ones cost memory); this pass
//      writes a four-word head per run, not the nodes, so the node array can be allocated at its
//      exact size afterwards;
//   4. bottom up, one launch per level: a node's children's boxes (a primitive's padded box, a
//      node's union) are quantised to the node's grid exactly as rt_wide.cpp does, one step of
//      slack on every side, and the node is written.
//
// The result has the format of WideScene and is walked by the same kernels.  The trees are an
// octree in disguise: as good as a surface-area tree where primitives are small and evenly spread
// (the triangle soup), worse where sizes are mixed; the build takes milliseconds instead of seconds.
// Primitives the reference tests without a box test before them (the objects of a list at the
// root: the primitive entries that no box entry spans) are found with a prefix maximum over the
// box entries' skip links and marked in word 7 of every node above them, as rt_wide.cpp does.

#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <stdint.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "rt_internal.hpp"

namespace rtamd {
namespace {

constexpr uint32_t WB = 256;

struct Counters { uint32_t n_prims, n_next, slots, prim_records, overflow, pad[3]; };
constexpr uint32_t BOXLESS = 1u << 31;         // flag beside an entry index (entry indices stay below 2^30)
struct Run { uint32_t b, e, slot; int32_t shift; };     // keys [b, e) form the node at `slot`; next digit at bit 32 + shift

__device__ __forceinline__ uint32_t ordered(float f) {           // an unsigned whose order is the floats'
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float unordered(uint32_t u) {
    return __uint_as_float((u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u);
}

// the padded box of a primitive from the first three float4 of its entry (WideBuilder::prim_box)
__device__ __forceinline__ void prim_box(const float4& a, const float4& b, const float4& c, float lo[3], float hi[3]) {
    const float v0[3] = {a.x, a.y, a.z}, e1[3] = {b.x, b.y, b.z}, e2[3] = {c.x, c.y, c.z};
    float big = 0.0f;
    if ((__float_as_uint(a.w) >> TAG_SHIFT) == TAG_TRI) {
        for (int k = 0; k < 3; k++) {
            const float p0 = v0[k], p1 = v0[k] + e1[k], p2 = v0[k] + e2[k];
            lo[k] = fminf(p0, fminf(p1, p2));
            hi[k] = fmaxf(p0, fmaxf(p1, p2));
        }
    } else {
        const float r = fabsf(b.x);
        for (int k = 0; k < 3; k++) { lo[k] = v0[k] - r; hi[k] = v0[k] + r; }
    }
    for (int k = 0; k < 3; k++) big = fmaxf(big, fmaxf(fabsf(lo[k]), fabsf(hi[k])));
    const float pad = 0.0001f + big * 4.76837158e-7f;
    for (int k = 0; k < 3; k++) { lo[k] -= pad; hi[k] += pad; }
}

// reach[e] = the entry behind the subtree of box entry e (0 for other entries): after an exclusive
// prefix maximum, a primitive entry e with reach[e] <= e lies under no box
__global__ void __launch_bounds__(WB)
k_reach(const float4* __restrict__ ent, uint32_t n_entries, uint32_t* __restrict__ reach) {
    const uint32_t e = blockIdx.x * WB + threadIdx.x;
    if (e >= n_entries) return;
    const uint32_t w = __float_as_uint(ent[4 * (size_t)e].w);
    reach[e] = (w >> TAG_SHIFT) == TAG_BOX ? (w & SKIP_MASK) : 0u;
}

// primitive entries -> compact list (entry index | BOXLESS); bounds of the box centres (as ordered
// unsigneds: [0..2] min, [3..5] max).  A workgroup takes a contiguous stretch of the entry array and
// passes over it twice: it counts its primitives, reserves their places with ONE atomic, then writes
// them, every wave numbering its own by ballot and prefix.  (Atomics of all waves on one counter are
// served one after the other, about 57 ns apiece on this part: one per wave and trip -- 312 k for
// 10 M triangles, which is also what the compiler makes of one per lane -- took 18 of the build's
// 22 ms of kernel time.)  `cover` is null when no primitive can be box-less.
__global__ void __launch_bounds__(WB)
k_collect(const float4* __restrict__ ent, uint32_t n_entries, uint32_t per_block, const uint32_t* __restrict__ cover,
          uint32_t* __restrict__ prim_entry, Counters* cnt, uint32_t* __restrict__ bounds) {
    __shared__ uint32_t wave_count[WB / 64], block_base, red[6][WB / 64];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t begin = blockIdx.x * per_block, end = min(n_entries, begin + per_block);   // (per_block is a multiple of WB)
    uint32_t mine = 0;
    for (uint32_t base = begin; base < end; base += WB) {        // (wave-uniform trip count)
        const uint32_t e = base + threadIdx.x;
        uint32_t tag = TAG_END;
        if (e < end) tag = __float_as_uint(ent[4 * (size_t)e].w) >> TAG_SHIFT;
        mine += (uint32_t)__popcll(__ballot(tag == TAG_TRI || tag == TAG_SPH));
    }
    if (lane == 0) wave_count[wave] = mine;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t total = 0;
        for (uint32_t w = 0; w < WB / 64; w++) total += wave_count[w];
        block_base = total ? atomicAdd(&cnt->n_prims, total) : 0u;
    }
    __syncthreads();
    uint32_t at = block_base;
    for (uint32_t w = 0; w < wave; w++) at += wave_count[w];
    uint32_t mn[3] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}, mx[3] = {0u, 0u, 0u};
    for (uint32_t base = begin; base < end; base += WB) {
        const uint32_t e = base + threadIdx.x;
        float4 a = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (e < end) a = ent[4 * (size_t)e];
        const uint32_t tag = __float_as_uint(a.w) >> TAG_SHIFT;
        const bool is_prim = e < end && (tag == TAG_TRI || tag == TAG_SPH);
        const unsigned long long m = __ballot(is_prim);
        const uint32_t place = at + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
        at += (uint32_t)__popcll(m);
        if (!is_prim) continue;
        const bool boxless = cover != nullptr && cover[e] <= e;
        prim_entry[place] = e | (boxless ? BOXLESS : 0u);
        float lo[3], hi[3];
        prim_box(a, ent[4 * (size_t)e + 1], ent[4 * (size_t)e + 2], lo, hi);
        for (int k = 0; k < 3; k++) {
            const uint32_t c = ordered(0.5f * lo[k] + 0.5f * hi[k]);
            mn[k] = min(mn[k], c);
            mx[k] = max(mx[k], c);
        }
    }
    // (every lane is back together here: one pair of atomics per workgroup and axis)
    for (int k = 0; k < 3; k++) {
        for (int off = 32; off > 0; off >>= 1) {
            mn[k] = min(mn[k], (uint32_t)__shfl_xor((int)mn[k], off));
            mx[k] = max(mx[k], (uint32_t)__shfl_xor((int)mx[k], off));
        }
        if (lane == 0) { red[k][wave] = mn[k]; red[3 + k][wave] = mx[k]; }
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        const uint32_t k = threadIdx.x;
        uint32_t v = red[k][0];
        for (uint32_t w = 1; w < WB / 64; w++) v = k < 3 ? min(v, red[k][w]) : max(v, red[k][w]);
        if (k < 3) atomicMin(&bounds[k], v); else atomicMax(&bounds[k], v);
    }
}

__device__ __forceinline__ uint32_t spread3(uint32_t v) {        // 10 bits -> every third bit
    v = (v | (v << 16)) & 0x030000FFu;
    v = (v | (v << 8)) & 0x0300F00Fu;
    v = (v | (v << 4)) & 0x030C30C3u;
    v = (v | (v << 2)) & 0x09249249u;
    return v;
}

__global__ void __launch_bounds__(WB)
k_keys(const float4* __restrict__ ent, const uint32_t* __restrict__ prim_entry, uint32_t n, const uint32_t* __restrict__ bounds,
       unsigned long long* __restrict__ keys) {
    const uint32_t i = blockIdx.x * WB + threadIdx.x;
    if (i >= n) return;
    const uint32_t ef = prim_entry[i], e = ef & ~BOXLESS;
    float lo[3], hi[3];
    prim_box(ent[4 * (size_t)e], ent[4 * (size_t)e + 1], ent[4 * (size_t)e + 2], lo, hi);
    uint32_t q[3];
    for (int k = 0; k < 3; k++) {
        const float mn = unordered(bounds[k]), mx = unordered(bounds[3 + k]);
        const float c = 0.5f * lo[k] + 0.5f * hi[k];
        float t = mx > mn ? (c - mn) / (mx - mn) * 1024.0f : 0.0f;
        t = t == t ? t : 0.0f;
        q[k] = (uint32_t)fminf(fmaxf(t, 0.0f), 1023.0f);
    }
    const uint32_t code = spread3(q[0]) | spread3(q[1]) << 1 | spread3(q[2]) << 2;
    keys[i] = (unsigned long long)code << 32 | ef;             // (the flag travels in the key's low word)
}

__device__ __forceinline__ uint32_t digit_of(unsigned long long key, int shift) { return (uint32_t)(key >> (32 + shift)) & 7u; }

// one level, top down: every run becomes a node and names its children.  heads[run] = (mask of
// the child slots that hold nodes, node slot of child slot 0, first primitive record, mask of the
// child slots that hold primitives); bit 8 + s of the first word: primitive child s is box-less.
__global__ void __launch_bounds__(WB)
k_split(const unsigned long long* __restrict__ keys, const float4* __restrict__ ent, Run* __restrict__ runs, uint32_t off_in,
        uint32_t n_in, uint32_t off_out, Counters* cnt, uint4* __restrict__ heads, uint32_t slot_capacity,
        float4* __restrict__ prims) {
    // (places in the three lists are handed out per workgroup -- a scan over its runs and one atomic
    // per list -- because atomics of many waves on one counter are served one after the other)
    typedef hipcub::BlockScan<uint32_t, WB> Scan;
    __shared__ typename Scan::TempStorage scan_tmp;
    __shared__ uint32_t list_base[3];
    const uint32_t i = blockIdx.x * WB + threadIdx.x;
    const bool valid = i < n_in;
    Run r{0u, 0u, 0u, 0};
    int shift = -1;
    uint32_t cb[9] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
    uint32_t imask = 0, lmask = 0;
    if (valid) {
        r = runs[off_in + i];
        shift = r.shift;
        while (shift >= 0 && digit_of(keys[r.b], shift) == digit_of(keys[r.e - 1], shift)) shift -= 3;
        if (shift >= 0) {                          // children = runs of equal digit; the digit is the slot
            cb[0] = r.b;
            for (uint32_t d = 1; d < 8; d++) {     // first key whose digit is >= d (digits do not decrease along the run)
                uint32_t lo = r.b, hi = r.e;
                while (lo < hi) {
                    const uint32_t mid = lo + (hi - lo) / 2;
                    if (digit_of(keys[mid], shift) < d) lo = mid + 1; else hi = mid;
                }
                cb[d] = lo;
            }
            cb[8] = r.e;
        } else {                                   // the same thirty bits throughout: equal parts
            const uint32_t size = r.e - r.b, parts = size < 8u ? size : 8u;
            for (uint32_t k = 0; k <= 8; k++) cb[k] = k <= parts ? r.b + (uint32_t)((unsigned long long)size * k / parts) : r.e;
        }
        for (uint32_t k = 0; k < 8; k++) {
            const uint32_t size = cb[k + 1] - cb[k];
            if (size == 1) lmask |= 1u << k;
            else if (size > 1) imask |= 1u << k;
        }
    }
    // the node children's slots: from the lowest to the highest occupied child slot
    const uint32_t low = imask ? (uint32_t)__ffs((int)imask) - 1u : 0u;
    const uint32_t span = imask ? 32u - (uint32_t)__clz((int)imask) - low : 0u;
    uint32_t my_slot, my_prim, my_next, all_slots, all_prims, all_next;
    Scan(scan_tmp).ExclusiveSum(span, my_slot, all_slots);
    __syncthreads();
    Scan(scan_tmp).ExclusiveSum((uint32_t)__popc(lmask), my_prim, all_prims);
    __syncthreads();
    Scan(scan_tmp).ExclusiveSum((uint32_t)__popc(imask), my_next, all_next);
    if (threadIdx.x == 0) {
        list_base[0] = all_slots ? atomicAdd(&cnt->slots, all_slots) : 0u;
        list_base[1] = all_prims ? atomicAdd(&cnt->prim_records, all_prims) : 0u;
        list_base[2] = all_next ? atomicAdd(&cnt->n_next, all_next) : 0u;
    }
    __syncthreads();
    if (!valid) return;
    uint32_t cbase = 0;
    if (imask) {
        const uint32_t first = list_base[0] + my_slot;
        if (first + span > slot_capacity) { cnt->overflow = 1u; return; }
        cbase = first - low;                   // (first >= 8 > low)
    }
    const uint32_t pbase = list_base[1] + my_prim;
    uint32_t rank = 0, boxless = 0;
    for (uint32_t k = 0; k < 8; k++) {
        if (!(lmask >> k & 1u)) continue;
        const uint32_t ef = (uint32_t)keys[cb[k]], e = ef & ~BOXLESS;
        if (ef & BOXLESS) boxless |= 1u << k;
        float4* dst = prims + 4 * (size_t)(pbase + rank);
        dst[0] = ent[4 * (size_t)e]; dst[1] = ent[4 * (size_t)e + 1]; dst[2] = ent[4 * (size_t)e + 2];
        dst[3] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        rank++;
    }
    if (imask) {
        uint32_t at = off_out + list_base[2] + my_next;
        for (uint32_t k = 0; k < 8; k++)
            if (imask >> k & 1u) runs[at++] = Run{cb[k], cb[k + 1], cbase + k, shift - 3};
    }
    heads[off_in + i] = make_uint4(imask | boxless << 8, cbase, pbase, lmask);
}

// grid of one axis of a node (WideBuilder::grid_for): step 2^(e - 127) and an origin two steps below lo
__device__ __forceinline__ bool grid_for(float lo, float hi, float& origin, uint32_t& e) {
    if (!(lo <= hi) || isinf(lo) || isinf(hi)) return false;
    const double ext = (double)hi - (double)lo;
    const double big = fmax(fabs((double)lo), fabs((double)hi));
    int k = -120;
    if (ext > 0.0) k = max(k, (int)ceil(log2(ext / 250.0)));
    if (big > 0.0) k = max(k, ilogb(big) - 21);
    for (;; k++) {
        if (k > 126) return false;
        const double s = ldexp(1.0, k);
        const float o = (float)((double)lo - 2.0 * s);
        if (isinf(o)) return false;
        if (floor(((double)lo - o) / s) - 1.0 >= 0.0 && ceil(((double)hi - o) / s) + 1.0 <= 255.0) {
            origin = o;
            e = (uint32_t)(k + 127);
            return e >= 1u && e <= 254u;
        }
    }
}

// one level, bottom up: boxes of the children, the node's grid, the planes
__global__ void __launch_bounds__(WB)
k_refit(const Run* __restrict__ runs, const uint4* __restrict__ heads, uint32_t off, uint32_t n, uint32_t* __restrict__ nodes,
        uint32_t stride, const float4* __restrict__ prims, float* __restrict__ node_box, uint8_t* __restrict__ node_boxless,
        Counters* cnt) {
    const uint32_t i = blockIdx.x * WB + threadIdx.x;
    if (i >= n) return;
    const uint32_t slot = runs[off + i].slot;
    uint32_t* w = nodes + (size_t)slot * stride;
    const uint4 head = heads[off + i];
    const uint32_t imask = head.x & 0xFFu, lmask = head.w, cbase = head.y, pbase = head.z;
    // children the reference reaches without a box test: box-less primitives, and nodes that hold one
    uint32_t amask = (head.x >> 8) & lmask;
    for (uint32_t s = 0; s < 8; s++)
        if ((imask >> s & 1u) && node_boxless[cbase + s]) amask |= 1u << s;
    node_boxless[slot] = amask ? 1 : 0;
    float clo[8][3], chi[8][3];
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    uint32_t rank = 0;
    for (uint32_t s = 0; s < 8; s++) {
        if (lmask >> s & 1u) {
            const float4* p = prims + 4 * (size_t)(pbase + rank);
            prim_box(p[0], p[1], p[2], clo[s], chi[s]);
            rank++;
        } else if (imask >> s & 1u) {
            const float* b = node_box + 6 * (size_t)(cbase + s);
            for (int k = 0; k < 3; k++) { clo[s][k] = b[k]; chi[s][k] = b[3 + k]; }
        } else {
            continue;
        }
        for (int k = 0; k < 3; k++) { lo[k] = fminf(lo[k], clo[s][k]); hi[k] = fmaxf(hi[k], chi[s][k]); }
    }
    float* nb = node_box + 6 * (size_t)slot;
    for (int k = 0; k < 3; k++) { nb[k] = lo[k]; nb[3 + k] = hi[k]; }
    float origin[3];
    uint32_t e[3];
    for (int k = 0; k < 3; k++)
        if (!grid_for(lo[k], hi[k], origin[k], e[k])) { cnt->overflow = 2u; return; }
    uint32_t planes[12] = {};
    for (uint32_t s = 0; s < 8; s++) {
        if (!((imask | lmask) >> s & 1u)) continue;
        for (int k = 0; k < 3; k++) {
            const double st = ldexp(1.0, (int)e[k] - 127);
            const double ql = floor(((double)clo[s][k] - origin[k]) / st) - 1.0;
            const double qh = ceil(((double)chi[s][k] - origin[k]) / st) + 1.0;
            if (ql < 0.0 || qh > 255.0) { cnt->overflow = 3u; return; }
            planes[2 * k + (s >> 2)] |= (uint32_t)ql << (8 * (s & 3u));
            planes[2 * (3 + k) + (s >> 2)] |= (uint32_t)qh << (8 * (s & 3u));
        }
    }
    w[0] = __float_as_uint(origin[0]); w[1] = __float_as_uint(origin[1]); w[2] = __float_as_uint(origin[2]);
    w[3] = e[0] | e[1] << 8 | e[2] << 16 | imask << 24;
    w[4] = cbase; w[5] = pbase; w[6] = lmask; w[7] = amask;
    for (int p = 0; p < 12; p++) w[8 + p] = planes[p];
}

#define HIP_OK(call)                                                                                     \
    do {                                                                                                 \
        const hipError_t err_ = (call);                                                                  \
        if (err_ != hipSuccess) { error = std::string(#call) + ": " + hipGetErrorString(err_); return -1; } \
    } while (0)

struct Scratch {                               // freed whichever way the build ends
    std::vector<void*> p;
    ~Scratch() { for (void* q : p) (void)hipFree(q); }
    hipError_t get(void** out, size_t bytes) {
        const hipError_t e = hipMalloc(out, std::max<size_t>(bytes, 256));
        if (e == hipSuccess) p.push_back(*out);
        return e;
    }
};

}  // namespace

int build_wide_on_device(void* stream, const void* d_entries, uint32_t n_entries, WideDeviceTree& out, std::string& error) {
    hipStream_t st = (hipStream_t)stream;
    const float4* ent = (const float4*)d_entries;
    out = WideDeviceTree();
    Scratch tmp;
    Counters* cnt = nullptr;
    uint32_t* bounds = nullptr;
    uint32_t* prim_entry = nullptr;
    HIP_OK(tmp.get((void**)&cnt, sizeof(Counters)));
    HIP_OK(tmp.get((void**)&bounds, 6 * sizeof(uint32_t)));
    HIP_OK(tmp.get((void**)&prim_entry, (size_t)n_entries * 4));
    HIP_OK(hipMemsetAsync(cnt, 0, sizeof(Counters), st));
    uint32_t first_word = 0;
    {
        const uint32_t init[6] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0u, 0u, 0u};
        HIP_OK(hipMemcpyAsync(bounds, init, sizeof(init), hipMemcpyHostToDevice, st));
        if (n_entries) HIP_OK(hipMemcpyAsync(&first_word, (const char*)d_entries + 12, 4, hipMemcpyDeviceToHost, st));
        HIP_OK(hipStreamSynchronize(st));
    }
    // Box-less primitives exist unless one box spans every entry (a Bvh or a mesh at the root): only
    // then is the prefix maximum over the boxes' skip links needed.
    uint32_t* cover = nullptr;
    const bool one_box_spans_all = n_entries && (first_word >> TAG_SHIFT) == TAG_BOX && (first_word & SKIP_MASK) == n_entries;
    if (n_entries && !one_box_spans_all) {
        uint32_t* reach = nullptr;
        HIP_OK(tmp.get((void**)&reach, (size_t)n_entries * 4));
        HIP_OK(tmp.get((void**)&cover, (size_t)n_entries * 4));
        hipLaunchKernelGGL(k_reach, dim3((n_entries + WB - 1) / WB), dim3(WB), 0, st, ent, n_entries, reach);
        size_t scan_bytes = 0;
        HIP_OK(hipcub::DeviceScan::ExclusiveScan(nullptr, scan_bytes, reach, cover, hipcub::Max(), 0u, (int)n_entries, st));
        void* scan_tmp = nullptr;
        HIP_OK(tmp.get(&scan_tmp, scan_bytes));
        HIP_OK(hipcub::DeviceScan::ExclusiveScan(scan_tmp, scan_bytes, reach, cover, hipcub::Max(), 0u, (int)n_entries, st));
    }
    if (n_entries) {
        // contiguous stretches of the entry array, about eight workgroups per CU
        const uint32_t per_block = (uint32_t)std::max<size_t>(WB, (((size_t)n_entries + 2047) / 2048 + WB - 1) / WB * WB);
        const uint32_t blocks = (n_entries + per_block - 1) / per_block;
        hipLaunchKernelGGL(k_collect, dim3(blocks), dim3(WB), 0, st, ent, n_entries, per_block, (const uint32_t*)cover, prim_entry, cnt, bounds);
    }
    Counters h{};
    HIP_OK(hipMemcpyAsync(&h, cnt, sizeof(h), hipMemcpyDeviceToHost, st));
    HIP_OK(hipStreamSynchronize(st));
    const uint32_t n = h.n_prims;
    if (n >= (1u << 30)) { error = "too many primitives"; return -1; }

    // the tree's arrays: one record per primitive now, the nodes once their number is known
    const uint32_t stride = (size_t)n * 40 > WIDE_PACKED_MAX_BYTES ? WIDE_LINE_WORDS : WIDE_NODE_WORDS;
    // (the walk's stack words hold 24-bit slot numbers)
    const size_t slot_capacity = (size_t)1 << 24;
    void* d_nodes = nullptr;
    void* d_prims = nullptr;
    if (hipMalloc(&d_prims, std::max<size_t>((size_t)n * WIDE_PRIM_FLOATS * 4, 64)) != hipSuccess) {
        error = "out of device memory for the wide tree's primitive records";
        return -1;
    }
    struct Owned {                             // released unless the build succeeds
        void *a, *b;
        bool keep = false;
        ~Owned() { if (!keep) { if (a) (void)hipFree(a); if (b) (void)hipFree(b); } }
    } owned{nullptr, d_prims};

    uint32_t depth = 1, slots = 1;
    if (n > 0) {
        unsigned long long *keys_in = nullptr, *keys = nullptr;
        Run* runs = nullptr;
        uint4* heads = nullptr;
        HIP_OK(tmp.get((void**)&keys_in, (size_t)n * 8));
        HIP_OK(tmp.get((void**)&keys, (size_t)n * 8));
        HIP_OK(tmp.get((void**)&runs, ((size_t)n + 1) * sizeof(Run)));
        HIP_OK(tmp.get((void**)&heads, ((size_t)n + 1) * sizeof(uint4)));
        hipLaunchKernelGGL(k_keys, dim3((n + WB - 1) / WB), dim3(WB), 0, st, ent, prim_entry, n, bounds, keys_in);
        size_t sort_bytes = 0;
        HIP_OK(hipcub::DeviceRadixSort::SortKeys(nullptr, sort_bytes, keys_in, keys, (int)n, 0, 62, st));
        void* sort_tmp = nullptr;
        HIP_OK(tmp.get(&sort_tmp, sort_bytes));
        HIP_OK(hipcub::DeviceRadixSort::SortKeys(sort_tmp, sort_bytes, keys_in, keys, (int)n, 0, 62, st));

        // top down.  Level l's runs stand at runs[off[l] .. off[l + 1])
        std::vector<uint32_t> off{0u, 1u};
        const Run root{0u, n, 0u, 27};
        HIP_OK(hipMemcpyAsync(runs, &root, sizeof(root), hipMemcpyHostToDevice, st));
        // (slot 0 is the root; the counter starts at 8 so that child_base = first slot - lowest occupied
        // child slot never falls below zero)
        h.slots = 8u; h.n_next = 0u; h.prim_records = 0u; h.overflow = 0u;
        HIP_OK(hipMemcpyAsync(cnt, &h, sizeof(h), hipMemcpyHostToDevice, st));
        HIP_OK(hipStreamSynchronize(st));      // (root and h leave scope as values; keep it simple)
        for (;;) {
            const uint32_t level = (uint32_t)off.size() - 2, n_in = off[level + 1] - off[level];
            hipLaunchKernelGGL(k_split, dim3((n_in + WB - 1) / WB), dim3(WB), 0, st, keys, ent, runs, off[level], n_in,
                               off[level + 1], cnt, heads, (uint32_t)slot_capacity, (float4*)d_prims);
            HIP_OK(hipMemcpyAsync(&h, cnt, sizeof(h), hipMemcpyDeviceToHost, st));
            HIP_OK(hipStreamSynchronize(st));
            if (h.overflow) { error = "the device-built wide tree needs more than 2^24 node slots"; return -1; }
            if (h.n_next == 0) break;
            if ((size_t)off.back() + h.n_next > (size_t)n + 1) { error = "internal error: wide tree run list overflow"; return -1; }
            off.push_back(off.back() + h.n_next);
            if (off.size() - 1 > WIDE_MAX_DEPTH) { error = "the wide tree is deeper than its walk's stack"; return -1; }
            HIP_OK(hipMemsetAsync(&cnt->n_next, 0, 4, st));
        }
        depth = (uint32_t)off.size() - 1;
        slots = h.slots;
        if (h.prim_records != n) { error = "internal error: wide tree lost primitives"; return -1; }
        // the nodes, at their exact number
        if (hipMalloc(&d_nodes, (size_t)slots * stride * 4) != hipSuccess) { error = "out of device memory for the wide tree's nodes"; return -1; }
        owned.a = d_nodes;
        HIP_OK(hipMemsetAsync(d_nodes, 0, (size_t)slots * stride * 4, st));
        float* node_box = nullptr;
        uint8_t* node_boxless = nullptr;
        HIP_OK(tmp.get((void**)&node_box, (size_t)slots * 6 * sizeof(float)));
        HIP_OK(tmp.get((void**)&node_boxless, (size_t)slots));
        HIP_OK(hipMemsetAsync(node_boxless, 0, (size_t)slots, st));
        // bottom up
        for (uint32_t level = depth; level-- > 0;) {
            const uint32_t n_in = off[level + 1] - off[level];
            hipLaunchKernelGGL(k_refit, dim3((n_in + WB - 1) / WB), dim3(WB), 0, st, runs, heads, off[level], n_in,
                               (uint32_t*)d_nodes, stride, (const float4*)d_prims, node_box, node_boxless, cnt);
        }
        HIP_OK(hipMemcpyAsync(&h, cnt, sizeof(h), hipMemcpyDeviceToHost, st));
        HIP_OK(hipStreamSynchronize(st));
        if (h.overflow) { error = "wide tree: a box is not finite"; return -1; }
    } else {
        if (hipMalloc(&d_nodes, (size_t)stride * 4) != hipSuccess) { error = "out of device memory for the wide tree's nodes"; return -1; }
        owned.a = d_nodes;
        HIP_OK(hipMemsetAsync(d_nodes, 0, (size_t)stride * 4, st));
    }
    HIP_OK(hipStreamSynchronize(st));
    if (hipGetLastError() != hipSuccess) { error = "a kernel of the device-side wide build failed to launch"; return -1; }
    owned.keep = true;
    out.nodes = d_nodes;
    out.prims = d_prims;
    out.nodes_bytes = (size_t)slots * stride * 4;
    out.prims_bytes = std::max<size_t>((size_t)n * WIDE_PRIM_FLOATS * 4, 64);
    out.n_slots = slots;
    out.n_prims = n;
    out.depth = depth;
    out.node_words = stride;
    if (std::getenv("RT_BUILD_LOG"))          // the tree's size on stderr
        std::fprintf(stderr, "[rt wide build, device] %u primitives, %u levels, %u node slots of %u bytes: %.1f MB of nodes, %.1f MB of primitive records\n",
                     n, depth, slots, stride * 4u, out.nodes_bytes / 1e6, out.prims_bytes / 1e6);
    return 0;
}

}  // namespace rtamd
