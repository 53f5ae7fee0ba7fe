// HIP kernels of the wavefront path tracer (gfx950, wave64).
//
// Stages per batch of paths:  raygen -> [ trace -> shade ] x depth -> accumulate.
// trace   : closest hit over the flattened pre-order entry array; terminates paths that miss
//           (sky), hit an emitter or reach the depth cap; queues the rest by material kind.
// shade   : walks the lambert, metal and dielectric queues one after the other (material-sorted,
//           so a wave runs one scatter routine at a time) and compacts the surviving rays
//           into the next ray buffer.
// Every wave owns one segment of the batch's path slots from raygen to the last bounce, so
// queueing and compaction use only the wave's own ballot and lane prefix: no atomics, no
// barriers, deterministic ray order.  By default one kernel (k_path) carries a segment through
// all its stages; k_raygen / k_trace / k_shade are the same stages as separate launches.
//
// Intersection arithmetic must equal the reference's IEEE f32 results bit for bit, so this file
// is built with -ffp-contract=off (no FMA), and relies on hipcc's default correctly rounded
// f32 divide / sqrt and on f32 denormals being preserved.

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <mutex>
#ifdef RT_STAMPS
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#endif

#include "rt_internal.hpp"

// Two translation units are built from this file.  The kernels of the per-lane walks are compiled
// without SLP vectorisation (rt_device_walk.hip includes this file with RT_TU_WALK defined and is
// built with -fno-slp-vectorize): the vectoriser packs pieces of their scalar arithmetic into
// v_pk_* instructions, which occupy the pipe as long as the two they replace, and pays for it with
// register shuffles and a dozen more VGPRs (world1 +10 %, thing.obj +17 % without it).  The wave
// sweep, whose arithmetic is written on 2-vectors, is 10 % faster WITH the vectoriser, so it stays
// in the default unit together with the kernels that do not depend on the traversal.  Diagnostic
// builds (-DRT_STAMPS, -DRT_UTIL) and -DRT_SINGLE_TU keep everything in one unit.
#if defined(RT_STAMPS) || defined(RT_UTIL) || defined(RT_SINGLE_TU)
#ifndef RT_TU_WALK
#define RT_HAS_SWEEP 1
#define RT_HAS_WALK 1
#endif
#elif defined(RT_TU_WALK)
#define RT_HAS_WALK 1
#else
#define RT_HAS_SWEEP 1
#endif
#if defined(RT_HAS_SWEEP) || defined(RT_HAS_WALK)

namespace rtamd {

#define RT_BLOCK 256   // grid-stride kernels
#ifndef RT_WAVE
#define RT_WAVE 64     // segment kernels: workgroup size (every wave owns one segment; one wave per
#endif                 // workgroup lets the dispatcher refill wave slots one by one: +3 % over 256)
#define RT_WPB (RT_WAVE / 64)

struct Counts { uint32_t box, tri, sph; };

__device__ __forceinline__ uint32_t f2u(float f) { return __float_as_uint(f); }
__device__ __forceinline__ float u2f(uint32_t u) { return __uint_as_float(u); }

// glam dot: three products, (x + y) + z  (no contraction in this file)
__device__ __forceinline__ float dot3(float ax, float ay, float az, float bx, float by, float bz) {
    return (ax * bx + ay * by) + az * bz;
}

// ---------------------------------------------------------------------------------------------
// Closest hit.  Restates Bvh::hit (shape.rs:142-158), Mesh::bvh_hit (shape.rs:292-324),
// Aabb::hit (math.rs:149-165), ray_triangle_intersection (shape.rs:411-450) and Sphere::hit
// (shape.rs:19-45) over the pre-order entry array.  best_t starts at the caller's t_max and is
// the reference's narrowed t_max at every visit.
// ---------------------------------------------------------------------------------------------
// One ray of one lane during the per-lane walk.
struct LaneRay {
    float ox, oy, oz, dx, dy, dz;
    float ix, iy, iz;          // math.rs:152: inv_d = 1.0 / ray_dir[axis], the same at every box
    float best_t;              // the reference's narrowed t_max
    uint32_t best_prim;        // prim id | PRIM_BACKFACE, RT_MISS while nothing is hit
    uint32_t i;                // cursor into the entry array
};

// TIE: the walk resolves exact ties in t by primitive key (rt_internal.hpp, "Primitive keys"); a
// ray that has hit nothing then holds KEY_NONE, which key_result() turns into RT_MISS at the end.
template <bool TIE = false>
__device__ __forceinline__ void lane_ray_init(LaneRay& r, float ox, float oy, float oz, float dx, float dy,
                                              float dz, float t_max) {
    r.ox = ox; r.oy = oy; r.oz = oz; r.dx = dx; r.dy = dy; r.dz = dz;
    r.ix = 1.0f / dx; r.iy = 1.0f / dy; r.iz = 1.0f / dz;
    r.best_t = t_max;
    r.best_prim = TIE ? KEY_NONE : RT_MISS;
    r.i = 0;
}
template <bool TIE>
__device__ __forceinline__ uint32_t key_result(uint32_t best_prim) {
    return TIE && best_prim == KEY_NONE ? RT_MISS : best_prim;
}
// rank of the primitive a hit record names (index into prim_geo / prim_meta / prim_order)
__device__ __forceinline__ uint32_t key_rank(uint32_t prim) {
    const uint32_t low = prim & PRIM_ID_MASK;
    return (prim & KEY_SPHERE) ? low : KEY_NONE - 1u - low;
}

// math.rs:149-165 on the box held in (a, b); returns the ray's next cursor.
// `t0 > t_min ? t0 : t_min` keeps t_min when t0 is NaN and so does fmaxf (likewise fminf over the
// three t1); the sign of a zero cannot change `t_max <= t_min`; and the per-axis early return
// equals the one test after the third axis because t_min only grows and t_max only shrinks.
// The narrowed t_max itself can be a NaN (the reference accepts a sphere hit whose discriminant
// overflowed, shape.rs:25-36, and hands its NaN t on as t_max): `t1 < t_max ? t1 : t_max` then
// keeps the NaN through all three axes and `t_max <= t_min` is false, i.e. every later box is
// entered.  near_or() is that select form for the last step, so a NaN best_t stays a NaN.
__device__ __forceinline__ float near_or(float far3, float best_t) { return far3 < best_t ? far3 : best_t; }
__device__ __forceinline__ uint32_t box_step(const float4& a, const float4& b, const LaneRay& r, float t_min) {
    const bool nx = r.ix < 0.0f, ny = r.iy < 0.0f, nz = r.iz < 0.0f;       // math.rs:155
    const float x0 = (a.x - r.ox) * r.ix, x1 = (b.x - r.ox) * r.ix;
    const float y0 = (a.y - r.oy) * r.iy, y1 = (b.y - r.oy) * r.iy;
    const float z0 = (a.z - r.oz) * r.iz, z1 = (b.z - r.oz) * r.iz;
    const float tmn = fmaxf(fmaxf(fmaxf(nx ? x1 : x0, ny ? y1 : y0), nz ? z1 : z0), t_min);
    const float tmx = near_or(fminf(fminf(nx ? x0 : x1, ny ? y0 : y1), nz ? z0 : z1), r.best_t);
    return tmx <= tmn ? (f2u(a.w) & SKIP_MASK) : f2u(b.w);            // missed: the skip link; hit: the first child
}

// 1 / x.  v_rcp_f32 followed by two Newton steps with exact (fused) residuals is bit-identical to the
// correctly rounded quotient for every f32 x with 2^-23 <= |x| <= 8.5e37 -- checked exhaustively over
// all 2.5e9 such inputs on gfx950 by tools/rcp_check.hip -- and costs 5 instructions instead of 11.
// The caller takes the full division when any lane is outside that range.
constexpr float RCP_SHORT_MAX = 8.5e37f;
__device__ __forceinline__ float rcp_short(float x) {
    float r = __builtin_amdgcn_rcpf(x);
    float e = __builtin_fmaf(-x, r, 1.0f);
    r = __builtin_fmaf(e, r, r);
    e = __builtin_fmaf(-x, r, 1.0f);
    r = __builtin_fmaf(e, r, r);
    return r;
}

// shape.rs:411-450 on the triangle (a.xyz = v0, b.xyz = v0v1, c.xyz = v0v2, b.w = prim id)
template <bool TIE = false>
__device__ __forceinline__ void tri_step(const float4& a, const float4& b, const float4& c, LaneRay& r, float t_min) {
    const float EPS = 1.1920929e-7f;           // f32::EPSILON
    const float px = r.dy * c.z - r.dz * c.y;          // pvec = dir x v0v2
    const float py = r.dz * c.x - r.dx * c.z;
    const float pz = r.dx * c.y - r.dy * c.x;
    const float det = dot3(b.x, b.y, b.z, px, py, pz);
    if (!(fabsf(det) < EPS)) {
        // (lanes with |det| < eps have left, so only huge / non-finite determinants need the full
        // division of shape.rs:434: inv_det = 1.0 / det)
        float inv_det;
        if (__ballot(!(fabsf(det) <= RCP_SHORT_MAX)) != 0ull) inv_det = 1.0f / det;
        else inv_det = rcp_short(det);
        const float tx = r.ox - a.x, ty = r.oy - a.y, tz = r.oz - a.z;
        const float u = dot3(tx, ty, tz, px, py, pz) * inv_det;
        if (!(u < 0.0f || u > 1.0f)) {
            const float qx = ty * b.z - tz * b.y;  // qvec = tvec x v0v1
            const float qy = tz * b.x - tx * b.z;
            const float qz = tx * b.y - ty * b.x;
            const float v = dot3(r.dx, r.dy, r.dz, qx, qy, qz) * inv_det;
            if (!(v < 0.0f || u + v > 1.0f)) {
                const float t = dot3(c.x, c.y, c.z, qx, qy, qz) * inv_det;
                // shape.rs:297; out of the reference's order a tie in t goes to the larger key
                if (t >= t_min && (t < r.best_t || (TIE && t == r.best_t && f2u(b.w) > (r.best_prim & KEY_MASK)))) {
                    r.best_t = t;
                    r.best_prim = f2u(b.w) | (det < EPS ? PRIM_BACKFACE : 0u);
                }
            }
        }
    }
}

// shape.rs:19-45 on the sphere (a.xyz = centre, b.x = radius, b.w = prim id)
template <bool TIE = false>
__device__ __forceinline__ void sphere_step(const float4& a, const float4& b, LaneRay& r, float t_min) {
    const float ocx = r.ox - a.x, ocy = r.oy - a.y, ocz = r.oz - a.z;
    const float qa = dot3(r.dx, r.dy, r.dz, r.dx, r.dy, r.dz);
    const float half_b = dot3(ocx, ocy, ocz, r.dx, r.dy, r.dz);
    const float qc = dot3(ocx, ocy, ocz, ocx, ocy, ocz) - b.x * b.x;
    const float disc = half_b * half_b - qa * qc;
    if (!(disc < 0.0f)) {
        const float sqrtd = sqrtf(disc);
        float root = (-half_b - sqrtd) / qa;
        bool ok = true;
        // shape.rs:32-36, closed interval: in the reference's order a sphere at t == t_max replaces what
        // was found before it; out of that order it does so only with the larger key
        const bool loses_ties = TIE && !(f2u(b.w) > (r.best_prim & KEY_MASK));
        if (root < t_min || r.best_t < root || (loses_ties && root == r.best_t)) {
            root = (-half_b + sqrtd) / qa;
            if (root < t_min || r.best_t < root || (loses_ties && root == r.best_t)) ok = false;
        }
        if (ok) {
            r.best_t = root;
            r.best_prim = f2u(b.w);
        }
    }
}

// Diagnostic build only (-DRT_UTIL): the fused kernel counts visits too, and every lane adds one
// to the sphere counter per loop trip of its wave (use on scenes without spheres):
// (box + triangle visits) / that count is the share of lanes a trip of the walk works for.
#ifdef RT_UTIL
#define RT_UTIL_TRIP(cnt) (cnt).sph++
#else
#define RT_UTIL_TRIP(cnt)
#endif
// Per-lane walk of a whole wave (call converged; lanes without a ray pass valid = false).  Every
// lane visits its own entries in the reference's order.  `ent` is the linked form of the entry
// array: 16-byte units (one float4 each), a box or a sphere in two units, a triangle in three;
// cursors and links count units, so a cursor is the index of the entry's first float4, and every
// entry names its successors itself: a box its skip link (a.w) and its first child (b.w), a
// triangle (c.w) and a sphere (b.y) the entry behind them.  Memory order is therefore the host's
// choice (the children of a box side by side in one 64-byte fetch, the top levels first).  A box
// visit touches 32 bytes.  A lane that stands at a primitive waits
// there while boxes remain to be tested by other lanes, until prim_wait lanes stand at
// primitives (or no lane stands at a box): the long primitive routines then run with many lanes
// instead of once per loop trip for a few.
// OCT (opt-in, not the reference's visiting order): `ent` holds eight copies of the n_ent units,
// one per direction octant, each in the pre-order that puts the nearer child of every box first
// for that octant's rays (links count from the start of the whole array); the lane walks the copy
// of its ray's octant.
__device__ __forceinline__ uint32_t octant_of(float dx, float dy, float dz) {
    return (dx >= 0.0f ? 0u : 1u) | (dy >= 0.0f ? 0u : 2u) | (dz >= 0.0f ? 0u : 4u);
}
// The units at a lane's cursor.  RT_WALK_LOADS of them are fetched whenever the cursor moves: two
// hold a box or a sphere, the third completes a triangle (otherwise it is fetched when the triangle
// is tested, a second round trip to memory), and with four the trip below can test the box BEHIND
// the one at the cursor as well.
#ifndef RT_WALK_LOADS
#define RT_WALK_LOADS 2
#endif
// 1: a trip that finds "the next entry to visit is the box right behind this one" (a missed box
// whose sibling lies beside it; in a pre-order layout a hit box and its first child) tests that box
// in the same trip.  No primitive lies between the two tests, so the second sees the best_t the
// reference's recursion would hand it; the visits and their order do not change, a lane just
// makes fewer trips (and every trip of a walk from memory is a round trip to L2 or HBM).
#ifndef RT_WALK_PAIR
#define RT_WALK_PAIR 0
#endif
#ifndef RT_WALK_MERGE
#define RT_WALK_MERGE 0
#endif
static_assert(RT_WALK_PAIR == 0 || RT_WALK_LOADS == 4, "the second box of a trip is the third and fourth unit");
struct LaneEntry { float4 a, b, c, d; };
__device__ __forceinline__ void fetch_entry(const float4* __restrict__ ent, uint32_t i, LaneEntry& e) {
    e.a = ent[i]; e.b = ent[i + 1];            // (behind the last entry: the TAG_END units)
    if (RT_WALK_LOADS >= 3) e.c = ent[i + 2];
    if (RT_WALK_LOADS >= 4) e.d = ent[i + 3];
}
// glam's Mat4::transform_point3 (w = 1) / transform_vector3 (w = 0) on the four columns at m:
// ((x_axis * x + y_axis * y) + z_axis * z) + w_axis * w in rounded f32 steps (no contraction here).
__device__ __forceinline__ void xf_apply(const float4* __restrict__ m, float x, float y, float z, float w, float& ox,
                                         float& oy, float& oz) {
    const float4 c0 = m[0], c1 = m[1], c2 = m[2], c3 = m[3];
    ox = c3.x * w + (c2.x * z + (c1.x * y + c0.x * x));
    oy = c3.y * w + (c2.y * z + (c1.y * y + c0.y * x));
    oz = c3.z * w + (c2.z * z + (c1.z * y + c0.z * x));
}

// The caller's ray of a lane that walks a scene with transformed meshes (XF forms).  The linked units of
// such a scene carry an ENTER marker in front of a transformed mesh's tree and a LEAVE marker behind it
// (rt_internal.hpp: END_CODE_*).  At ENTER the lane's ray goes to the mesh's local space, as Mesh::hit
// does before it descends (shape.rs:454: ray.transform(&self.inverse_transform)); t is the same
// parameter in both spaces, so best_t carries over; at LEAVE the caller's ray is back.  Meshes are
// leaves, so one saved ray is enough.
struct WorldRay { float ox, oy, oz, dx, dy, dz; };
// One trip of the per-lane walk for the whole wave: lanes at a box test it and move on; lanes at a
// primitive run its routine once enough of them wait (or no lane stands at a box).  Returns false
// when no lane has an entry left.  A lane without a further entry holds TAG_END in the tag word of
// e.a, so "at a box" and "at a primitive" are single compares whose lane masks are the ballots.
template <bool COUNT, bool TIE = false, bool XF = false>
__device__ __forceinline__ bool walk_trip(const float4* __restrict__ ent, uint32_t prim_wait, LaneRay& r, LaneEntry& e,
                                          float t_min, Counts& cnt, const float4* __restrict__ xforms = nullptr,
                                          const WorldRay* world = nullptr) {
    const uint32_t tag = f2u(e.a.w) >> TAG_SHIFT;
    const bool at_box = tag == TAG_BOX;
    const bool at_prim = tag - TAG_TRI < 2u;           // TAG_TRI or TAG_SPH
    static_assert(TAG_TRI == 1 && TAG_SPH == 2 && TAG_END == 3, "tag order");
    // (XF: a marker shares the tag of the end units and tells itself apart by its first word)
    const bool at_marker = XF && tag == TAG_END && f2u(e.a.x) != END_CODE_END;
    const unsigned long long mb = __ballot(at_box), mp = __ballot(at_prim), mm = XF ? __ballot(at_marker) : 0ull;
    if ((mb | mp | mm) == 0ull) return false;
    RT_UTIL_TRIP(cnt);
    if (XF && at_marker) {
        if (f2u(e.a.x) == END_CODE_ENTER) {
            const float4* inv = xforms + 8u * f2u(e.a.y) + 4u;
            xf_apply(inv, world->ox, world->oy, world->oz, 1.0f, r.ox, r.oy, r.oz);     // camera.rs:137-142
            xf_apply(inv, world->dx, world->dy, world->dz, 0.0f, r.dx, r.dy, r.dz);
        } else {
            r.ox = world->ox; r.oy = world->oy; r.oz = world->oz; r.dx = world->dx; r.dy = world->dy; r.dz = world->dz;
        }
        r.ix = 1.0f / r.dx; r.iy = 1.0f / r.dy; r.iz = 1.0f / r.dz;                     // math.rs:152
        r.i = f2u(e.b.y);                      // the entry behind the marker
        fetch_entry(ent, r.i, e);
    }
    if (at_box) {
        if (COUNT) cnt.box++;
        uint32_t nxt = box_step(e.a, e.b, r, t_min);
        if (RT_WALK_PAIR) {
            const bool chain = nxt == r.i + 2u && (f2u(e.c.w) >> TAG_SHIFT) == TAG_BOX;
            const uint32_t nxt2 = box_step(e.c, e.d, r, t_min);
            if (COUNT) cnt.box += chain ? 1u : 0u;
            nxt = chain ? nxt2 : nxt;
        }
        r.i = nxt;
        if (!RT_WALK_MERGE) fetch_entry(ent, r.i, e);
    }
    const bool run_prims = mb == 0ull || (uint32_t)__popcll(mp) >= prim_wait;      // wave-uniform
    if (run_prims) {
        if (at_prim) {
            if (tag == TAG_TRI) {
                if (COUNT) cnt.tri++;
                const float4 c = RT_WALK_LOADS >= 3 ? e.c : ent[r.i + 2];
                tri_step<TIE>(e.a, e.b, c, r, t_min);
                r.i = f2u(c.w);                // the entry behind the triangle
            } else {
                if (COUNT) cnt.sph++;
                sphere_step<TIE>(e.a, e.b, r, t_min);
                r.i = f2u(e.b.y);              // the entry behind the sphere
            }
            if (!RT_WALK_MERGE) fetch_entry(ent, r.i, e);
        }
    }
    // one fetch for every lane whose cursor moved in this trip (the address unit is busy for the
    // same 16-20 cycles per load instruction however few lanes take part)
    static_assert(!(XF && RT_WALK_MERGE), "the merged fetch does not know the markers");
    if (RT_WALK_MERGE && (at_box || (run_prims && at_prim))) fetch_entry(ent, r.i, e);
    return true;
}

// The split walk: the first `top` units of the array (the top levels of the trees, which every ray
// visits) are read from a copy in LDS, the rest from memory -- and the fetches from memory are
// batched.  A wave-wide load occupies the CU's address unit for at least 16 cycles however few
// lanes take part, and that unit is what the walks from memory wait for; LDS has a path of its own.
// A lane whose cursor leaves the top therefore does not fetch at once: it holds PEND_WORD in the tag
// word of e.a and waits until `deep_wait` lanes do (or no lane has an entry in hand), and one pair
// of loads then serves them all, while the lanes in the top go on out of LDS.  Every ray still
// visits its entries in the reference's order; only when a lane gets its next entry changes.
constexpr uint32_t PEND_WORD = 0xFFFFFFFFu;            // tag bits = TAG_END, but not the end units' word
constexpr uint32_t ENDED_WORD = TAG_END << TAG_SHIFT;
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef const f32x4 __attribute__((address_space(3)))* lds_units_ptr;   // (typed as LDS: its reads can never become flat loads)
__device__ __forceinline__ float4 lds_unit(lds_units_ptr p, uint32_t i) {
    const f32x4 v = p[i];
    return make_float4(v.x, v.y, v.z, v.w);
}
struct SplitCfg { lds_units_ptr lds; uint32_t top, deep_wait; };
template <bool COUNT>
__device__ __forceinline__ bool walk_trip_split(const float4* __restrict__ ent, const SplitCfg& sp, uint32_t prim_wait,
                                                LaneRay& r, LaneEntry& e, float t_min, Counts& cnt) {
    const uint32_t w = f2u(e.a.w);
    const uint32_t tag = w >> TAG_SHIFT;
    const bool at_box = tag == TAG_BOX;
    const bool at_prim = tag - TAG_TRI < 2u;
    const bool pending = w == PEND_WORD;
    const unsigned long long mb = __ballot(at_box), mp = __ballot(at_prim), mq = __ballot(pending);
    if ((mb | mp | mq) == 0ull) return false;
    RT_UTIL_TRIP(cnt);
    bool moved = false;
    if (at_box) {
        if (COUNT) cnt.box++;
        r.i = box_step(e.a, e.b, r, t_min);
        moved = true;
    }
    const bool run_prims = mb == 0ull || (uint32_t)__popcll(mp) >= prim_wait;      // wave-uniform
    if (run_prims && at_prim) {
        if (tag == TAG_TRI) {
            if (COUNT) cnt.tri++;
            float4 c;
            if (r.i < sp.top) c = lds_unit(sp.lds, r.i + 2); else c = ent[r.i + 2];
            tri_step(e.a, e.b, c, r, t_min);
            r.i = f2u(c.w);
        } else {
            if (COUNT) cnt.sph++;
            sphere_step(e.a, e.b, r, t_min);
            r.i = f2u(e.b.y);
        }
        moved = true;
    }
    const bool in_top = r.i < sp.top;
    if (moved && in_top) { e.a = lds_unit(sp.lds, r.i); e.b = lds_unit(sp.lds, r.i + 1); }
    const bool want = pending || (moved && !in_top);
    if (moved && !in_top) e.a.w = u2f(PEND_WORD);
    // lanes that hold an entry after this trip: those that just read one from LDS and those that wait at a primitive
    const unsigned long long m_has = __ballot((moved && in_top) || (at_prim && !run_prims));
    const unsigned long long m_want = __ballot(want);
    if (m_want != 0ull && ((uint32_t)__popcll(m_want) >= sp.deep_wait || m_has == 0ull)) {
        if (want) { e.a = ent[r.i]; e.b = ent[r.i + 1]; }
    }
    return true;
}

template <bool COUNT, bool OCT = false, bool TIE = OCT, bool XF = false>
__device__ __forceinline__ void closest_hit(const float4* __restrict__ ent, uint32_t n_ent_, uint32_t prim_wait,
                                            bool valid, float ox,
                                            float oy, float oz, float dx, float dy, float dz,
                                            float t_min, float& best_t, uint32_t& best_prim, Counts& cnt,
                                            const float4* __restrict__ xforms = nullptr) {
    static_assert(TIE || !OCT, "the octant copies are not in the reference's order");
    static_assert(!(XF && OCT), "the octant copies do not carry the transform markers");
    const WorldRay world{ox, oy, oz, dx, dy, dz};
    LaneRay r;
    lane_ray_init<TIE>(r, ox, oy, oz, dx, dy, dz, best_t);
    if (OCT) r.i = (n_ent_ + UNITS_END) * octant_of(dx, dy, dz);   // the lane's copy
    // The unit array ends with TAG_END units (every copy of it does, in the octant form), which is
    // where the last entries' links lead: a lane fetches its next entry unconditionally, and a
    // lane that has walked off the end -- or never had a ray -- holds TAG_END in the tag word of e.a.
    const float4 ended = make_float4(0.0f, 0.0f, 0.0f, u2f(TAG_END << TAG_SHIFT));
    LaneEntry e{ended, ended, ended, ended};
    if (valid) fetch_entry(ent, r.i, e);
    while (walk_trip<COUNT, TIE, XF>(ent, prim_wait, r, e, t_min, cnt, xforms, &world)) {}
    best_t = r.best_t;
    best_prim = key_result<TIE>(r.best_prim);
}

// What shading sees of a hit on a triangle of a TRANSFORMED mesh (shape.rs:298-307): the point is
// transform * (local ray at t), the normal normalize(transform * (+-v1.n)) as a vector, and the
// facing test pits the LOCAL ray's direction against that WORLD normal (the reference's `ray`
// there is the shadowing local one).  g = (raw v1.n, transform index + 1).
__device__ __forceinline__ void xf_hit(const float4* __restrict__ xforms, const float4& g, bool backface, float ox,
                                       float oy, float oz, float dx, float dy, float dz, float t, float& px, float& py,
                                       float& pz, float& nx, float& ny, float& nz, bool& front) {
    const float4* m = xforms + 8u * (f2u(g.w) - 1u);
    float lox, loy, loz, ldx, ldy, ldz;
    xf_apply(m + 4, ox, oy, oz, 1.0f, lox, loy, loz);      // Ray::transform(inverse_transform), camera.rs:137-142
    xf_apply(m + 4, dx, dy, dz, 0.0f, ldx, ldy, ldz);
    xf_apply(m, lox + ldx * t, loy + ldy * t, loz + ldz * t, 1.0f, px, py, pz);
    float wx, wy, wz;
    xf_apply(m, backface ? -g.x : g.x, backface ? -g.y : g.y, backface ? -g.z : g.z, 0.0f, wx, wy, wz);
    const float len = sqrtf(dot3(wx, wy, wz, wx, wy, wz));
    nx = wx / len; ny = wy / len; nz = wz / len;
    front = dot3(ldx, ldy, ldz, nx, ny, nz) < 0.0f;
}

// math.rs:149-165 as a predicate (the same arithmetic as box_step)
__device__ __forceinline__ bool box_hit(const float4& a, const float4& b, const LaneRay& r, float t_min) {
    const bool nx = r.ix < 0.0f, ny = r.iy < 0.0f, nz = r.iz < 0.0f;
    const float x0 = (a.x - r.ox) * r.ix, x1 = (b.x - r.ox) * r.ix;
    const float y0 = (a.y - r.oy) * r.iy, y1 = (b.y - r.oy) * r.iy;
    const float z0 = (a.z - r.oz) * r.iz, z1 = (b.z - r.oz) * r.iz;
    const float tmn = fmaxf(fmaxf(fmaxf(nx ? x1 : x0, ny ? y1 : y0), nz ? z1 : z0), t_min);
    const float tmx = near_or(fminf(fminf(nx ? x0 : x1, ny ? y0 : y1), nz ? z0 : z1), r.best_t);
    return !(tmx <= tmn);
}

// Ordered walk (opt-in; not the reference's visiting order): the same entries and the same tests,
// but at a box with two children the child that lies first along the ray is visited first and
// the other one waits on the lane's stack, so best_t has usually shrunk by the time it is popped
// and its box test fails.  `stack` is this wave's LDS region, slot s of a lane at
// stack[s * 64 + lane] (conflict-free whatever the lanes' depths); the host guarantees
// stack_depth slots suffice (a slot per box nesting level).  Top-level entries (a list root has
// several) are taken in array order whenever the stack runs empty.
template <bool COUNT>
__device__ __forceinline__ void closest_hit_ordered(const float4* __restrict__ ent, uint32_t n_ent, uint32_t prim_wait,
                                                    uint32_t* stack, uint32_t lane, bool valid, float ox, float oy,
                                                    float oz, float dx, float dy, float dz, float t_min,
                                                    float& best_t, uint32_t& best_prim, Counts& cnt) {
    LaneRay r;
    lane_ray_init<true>(r, ox, oy, oz, dx, dy, dz, best_t);
    uint32_t sp = 0;                           // slots in use
    uint32_t top = 0;                          // next top-level entry
    bool live = false;
    float4 a = make_float4(0.0f, 0.0f, 0.0f, 0.0f), b = a;
    // next entry to visit: the youngest waiting child, else the next top-level entry
    auto pop = [&]() {
        const bool from_top = sp == 0;
        if (from_top) r.i = top;
        else { sp--; r.i = stack[sp * 64u + lane]; }
        live = r.i < n_ent;
        if (live) {
            a = ent[r.i];
            b = ent[r.i + 1];
            if (from_top) {
                const uint32_t tg = f2u(a.w) >> TAG_SHIFT;
                top = tg == TAG_BOX ? (f2u(a.w) & SKIP_MASK) : r.i + (tg == TAG_TRI ? f2u(ent[r.i + 2].w) : 2u);
            }
        }
    };
    if (valid) pop();
    for (;;) {
        const uint32_t tag = f2u(a.w) >> TAG_SHIFT;
        const bool at_box = live && tag == TAG_BOX;
        const bool at_prim = live && tag != TAG_BOX;
        const unsigned long long mb = __ballot(at_box), mp = __ballot(at_prim);
        if ((mb | mp) == 0ull) break;
        if (at_box) {
            if (COUNT) cnt.box++;
            if (box_hit(a, b, r, t_min)) {
                const uint32_t link = f2u(b.w);
                const uint32_t second = link & ORD_RIGHT_MASK;
                uint32_t next = r.i + 2;
                if (second != 0u) {
                    const uint32_t axis = (link >> ORD_AXIS_SHIFT) & 3u;
                    const float d = axis == 0u ? dx : (axis == 1u ? dy : dz);
                    const bool first_is_near = (d >= 0.0f) == ((link & ORD_FIRST_LOWER) != 0u);
                    stack[sp * 64u + lane] = first_is_near ? second : next;
                    sp++;
                    next = first_is_near ? next : second;
                }
                r.i = next;
                a = ent[next];
                b = ent[next + 1];
            } else {
                pop();
            }
        }
        if (mb == 0ull || (uint32_t)__popcll(mp) >= prim_wait) {
            if (at_prim) {
                if (tag == TAG_TRI) {
                    if (COUNT) cnt.tri++;
                    const float4 c = ent[r.i + 2];
                    tri_step<true>(a, b, c, r, t_min);
                } else {
                    if (COUNT) cnt.sph++;
                    sphere_step<true>(a, b, r, t_min);
                }
                pop();
            }
        }
    }
    best_t = r.best_t;
    best_prim = key_result<true>(r.best_prim);
}

// Wide walk (opt-in; not the reference's visiting order, and not its trees): the scene's entries
// collapsed into nodes of up to eight children whose boxes the node holds itself, 8 bits per plane on
// a grid local to the node (80 bytes; the layout is described beside WIDE_NODE_WORDS).  One visit
// fetches a node, tests its eight boxes, queues the primitives among the children that were hit and
// descends into the nearest node among them; the others wait on the lane's stack as ONE word per
// level (node slot of the group << 8 | children still to visit).  The primitive tests are the
// reference's own (tri_step / sphere_step), so t is theirs.
// Box test: a child's planes are origin + byte * 2^e, and t = (plane - o) * inv_d is taken in the
// fused form byte * (2^e * inv_d) + (origin - o) * inv_d, one FMA per plane.  The builder keeps a
// grid step of slack on every side of the true box, which covers the form's rounding.  NaNs (a
// zero direction component against a plane the origin lies on) drop out of max / min, which only
// ever widens the interval.  A NaN best_t (shape.rs:25-36 accepts a sphere hit whose discriminant
// overflowed) leaves every box open in the reference; here it counts as +inf, which opens every
// box the ray passes at all -- no primitive outside those can be hit.
// Order: the builder puts a child in the slot whose diagonal matches its offset from the node's
// centre, so for a ray whose direction is positive along the axes in rb the children lie roughly in
// descending (slot ^ rb); the mask of hit children is permuted to that order once and the highest
// bit is taken first.
__device__ __forceinline__ float plane_byte(uint32_t w, int j) { return (float)((w >> (8 * j)) & 0xFFu); }
// A lane's place in the wide tree.
struct WideLane {
    uint32_t G;                // the group in hand: node slot of its child slot 0 << 8 | children still to visit, child slot s at bit s ^ rb
    uint32_t sp;               // stack slots in use
    uint32_t pend;             // primitives of the last node still to test, by child slot
    uint32_t pbase, lmask;     // that node's first primitive record and its mask of primitive slots
    uint32_t rb;               // axes along which the ray's direction is positive
    bool wild;                 // |d|^2 is outside [1e-30, 1e18) or a NaN: the reference's sphere test overflows (and
                               // accepts) or works on denormals (and accepts points far outside the sphere)
};
// what the walk derives from a ray's direction (again when the ray enters or leaves a transformed mesh)
__device__ __forceinline__ void wide_lane_dir(WideLane& w, float dx, float dy, float dz) {
    w.rb = (dx >= 0.0f ? 1u : 0u) | (dy >= 0.0f ? 2u : 0u) | (dz >= 0.0f ? 4u : 0u);
    const float dd = dot3(dx, dy, dz, dx, dy, dz);
    w.wild = !(dd < 1.0e18f && dd > 1.0e-30f);
}
__device__ __forceinline__ void wide_lane_init(WideLane& w, bool valid, float dx, float dy, float dz) {
    wide_lane_dir(w, dx, dy, dz);
    w.G = valid ? 1u << w.rb : 0u;             // the root is "child rb ^ rb = 0 of the group at node slot 0"
    w.sp = 0;
    w.pend = 0; w.pbase = 0; w.lmask = 0;
}
__device__ __forceinline__ bool wide_live(const WideLane& w) { return w.pend != 0u || (w.G & 0xFFu) != 0u || w.sp != 0u; }
// One trip of the wide walk for the whole wave: lanes that need a node take the nearest child still
// to visit (from the group in hand, else from the stack), fetch it and test its eight boxes; lanes
// with primitives of their last node in hand test one of them, once prim_wait lanes wait (or no lane
// needs a node).  `stack` is this wave's LDS region, slot s of a lane at stack[s * 64 + lane].
// Returns false when no lane has anything left.
// XF (scenes with transformed meshes): the root node of a mesh's tree names the mesh's transform.  A lane
// that fetches such a root puts its ray away (six LDS slots behind its stack, from `xf_slot`), pushes a
// marker under the subtree and goes on with the ray in the mesh's space (Mesh::hit, shape.rs:454; t is
// the same parameter there, so best_t carries over); when the marker comes off the stack the subtree
// is exhausted and the caller's ray is back.  Meshes are leaves of the scene, so one saved ray does.
constexpr uint32_t WIDE_XF_MARK = 0xFFFFFF00u;     // (a group without children at a node slot no tree has)
template <bool COUNT, bool XF = false>
__device__ __forceinline__ bool wide_trip(const uint4* __restrict__ nodes, uint32_t stride, const float4* __restrict__ prims,
                                          uint32_t prim_wait, uint32_t* stack, uint32_t lane, LaneRay& r,
                                          WideLane& w, float t_min, Counts& cnt, const float4* __restrict__ xforms = nullptr,
                                          uint32_t xf_slot = 0) {
    const bool at_prim = w.pend != 0u;
    const bool at_node = !at_prim && ((w.G & 0xFFu) != 0u || w.sp != 0u);
    const unsigned long long mb = __ballot(at_node), mp = __ballot(at_prim);
    if ((mb | mp) == 0ull) return false;
    RT_UTIL_TRIP(cnt);
    bool take_node = at_node;
    uint32_t G = w.G;
    if (at_node && (G & 0xFFu) == 0u) {
        w.sp--;
        G = stack[w.sp * 64u + lane];
        if (XF && G == WIDE_XF_MARK) {          // the mesh's tree is exhausted: the caller's ray again
            r.ox = u2f(stack[(xf_slot + 0u) * 64u + lane]); r.oy = u2f(stack[(xf_slot + 1u) * 64u + lane]);
            r.oz = u2f(stack[(xf_slot + 2u) * 64u + lane]); r.dx = u2f(stack[(xf_slot + 3u) * 64u + lane]);
            r.dy = u2f(stack[(xf_slot + 4u) * 64u + lane]); r.dz = u2f(stack[(xf_slot + 5u) * 64u + lane]);
            r.ix = 1.0f / r.dx; r.iy = 1.0f / r.dy; r.iz = 1.0f / r.dz;
            wide_lane_dir(w, r.dx, r.dy, r.dz);
            G = 0u;
            if (w.sp != 0u) { w.sp--; G = stack[w.sp * 64u + lane]; }
            take_node = (G & 0xFFu) != 0u;
            w.G = G;
        }
    }
    if (take_node) {
        if (COUNT) cnt.box++;
        uint32_t rb = w.rb;
        const uint32_t p = 31u - (uint32_t)__clz((int)(G & 0xFFu));
        G ^= 1u << p;
        const uint4* np = nodes + (size_t)((G >> 8) + (p ^ rb)) * stride;
        const uint4 q0 = np[0], q1 = np[1], q2 = np[2], q3 = np[3], q4 = np[4];
        if (XF && (q1.w >> 8) != 0u) {           // the root of a transformed mesh's tree
            stack[(xf_slot + 0u) * 64u + lane] = f2u(r.ox); stack[(xf_slot + 1u) * 64u + lane] = f2u(r.oy);
            stack[(xf_slot + 2u) * 64u + lane] = f2u(r.oz); stack[(xf_slot + 3u) * 64u + lane] = f2u(r.dx);
            stack[(xf_slot + 4u) * 64u + lane] = f2u(r.dy); stack[(xf_slot + 5u) * 64u + lane] = f2u(r.dz);
            if ((G & 0xFFu) != 0u) { stack[w.sp * 64u + lane] = G; w.sp++; }
            stack[w.sp * 64u + lane] = WIDE_XF_MARK; w.sp++;
            G = 0u;
            const float4* inv = xforms + 8u * ((q1.w >> 8) - 1u) + 4u;
            const float wox = r.ox, woy = r.oy, woz = r.oz, wdx = r.dx, wdy = r.dy, wdz = r.dz;
            xf_apply(inv, wox, woy, woz, 1.0f, r.ox, r.oy, r.oz);       // camera.rs:137-142
            xf_apply(inv, wdx, wdy, wdz, 0.0f, r.dx, r.dy, r.dz);
            r.ix = 1.0f / r.dx; r.iy = 1.0f / r.dy; r.iz = 1.0f / r.dz;  // math.rs:152
            wide_lane_dir(w, r.dx, r.dy, r.dz);
            rb = w.rb;
        }
        const bool nx = r.ix < 0.0f, ny = r.iy < 0.0f, nz = r.iz < 0.0f;
        const float sx = u2f((q0.w & 0xFFu) << 23) * r.ix;
        const float sy = u2f((q0.w >> 8 & 0xFFu) << 23) * r.iy;
        const float sz = u2f((q0.w >> 16 & 0xFFu) << 23) * r.iz;
        const float bx = (u2f(q0.x) - r.ox) * r.ix, by = (u2f(q0.y) - r.oy) * r.iy, bz = (u2f(q0.z) - r.oz) * r.iz;
        // the planes the ray meets first and last along each axis, slots 0-3 and 4-7
        const uint32_t nx0 = nx ? q3.z : q2.x, nx1 = nx ? q3.w : q2.y, fx0 = nx ? q2.x : q3.z, fx1 = nx ? q2.y : q3.w;
        const uint32_t ny0 = ny ? q4.x : q2.z, ny1 = ny ? q4.y : q2.w, fy0 = ny ? q2.z : q4.x, fy1 = ny ? q2.w : q4.y;
        const uint32_t nz0 = nz ? q4.z : q3.x, nz1 = nz ? q4.w : q3.y, fz0 = nz ? q3.x : q4.z, fz1 = nz ? q3.y : q4.w;
        const float bt = r.best_t == r.best_t ? r.best_t : __builtin_inff();
        uint32_t hm = 0;
#pragma unroll
        for (int s = 0; s < 8; s++) {
            const int j = s & 3;
            const float tnx = __builtin_fmaf(plane_byte(s < 4 ? nx0 : nx1, j), sx, bx);
            const float tny = __builtin_fmaf(plane_byte(s < 4 ? ny0 : ny1, j), sy, by);
            const float tnz = __builtin_fmaf(plane_byte(s < 4 ? nz0 : nz1, j), sz, bz);
            const float tfx = __builtin_fmaf(plane_byte(s < 4 ? fx0 : fx1, j), sx, bx);
            const float tfy = __builtin_fmaf(plane_byte(s < 4 ? fy0 : fy1, j), sy, by);
            const float tfz = __builtin_fmaf(plane_byte(s < 4 ? fz0 : fz1, j), sz, bz);
            const float tn = fmaxf(fmaxf(fmaxf(tnx, tny), tnz), t_min);
            const float tf = fminf(fminf(fminf(tfx, tfy), tfz), bt);
            hm |= tn <= tf ? 1u << s : 0u;
        }
        // A ray whose |d|^2 overflows the sphere test (shape.rs:25-36 then accepts the NaN) or is a denormal
        // (the root then lies anywhere) gets "hits" that have nothing to do with the sphere's box, and the
        // reference tests the objects of a list at the root without any box: such a ray enters the
        // children that stand for those objects whatever their boxes say.
        hm |= w.wild ? (q1.w & 0xFFu) : 0u;
        const uint32_t imask = q0.w >> 24;
        w.lmask = q1.z;
        w.pbase = q1.y;
        w.pend = hm & w.lmask;
        uint32_t ih = hm & imask;              // to visiting order: bit s -> bit s ^ rb
        ih = (rb & 1u) ? ((ih & 0x55u) << 1 | (ih >> 1 & 0x55u)) : ih;
        ih = (rb & 2u) ? ((ih & 0x33u) << 2 | (ih >> 2 & 0x33u)) : ih;
        ih = (rb & 4u) ? ((ih & 0x0Fu) << 4 | (ih >> 4 & 0x0Fu)) : ih;
        if (ih != 0u) {
            if ((G & 0xFFu) != 0u) { stack[w.sp * 64u + lane] = G; w.sp++; }
            G = q1.x << 8 | ih;
        }
        w.G = G;
    }
    if (mb == 0ull || (uint32_t)__popcll(mp) >= prim_wait) {
        if (at_prim) {
            const uint32_t s = (uint32_t)__ffs((int)w.pend) - 1u;
            w.pend &= w.pend - 1u;
            const float4* pp = prims + (size_t)(w.pbase + (uint32_t)__popc(w.lmask & ((1u << s) - 1u))) * 4u;
            const float4 a = pp[0], b = pp[1], c = pp[2];
            if ((f2u(a.w) >> TAG_SHIFT) == TAG_TRI) {
                if (COUNT) cnt.tri++;
                tri_step<true>(a, b, c, r, t_min);
            } else {
                if (COUNT) cnt.sph++;
                sphere_step<true>(a, b, r, t_min);
            }
        }
    }
    return true;
}

template <bool COUNT, bool XF = false>
__device__ __forceinline__ void closest_hit_wide(const uint4* __restrict__ nodes, uint32_t stride, const float4* __restrict__ prims,
                                                 uint32_t prim_wait, uint32_t* stack, uint32_t lane, bool valid,
                                                 float ox, float oy, float oz, float dx, float dy, float dz,
                                                 float t_min, float& best_t, uint32_t& best_prim, Counts& cnt,
                                                 const float4* __restrict__ xforms = nullptr, uint32_t xf_slot = 0) {
    LaneRay r;
    lane_ray_init<true>(r, ox, oy, oz, dx, dy, dz, best_t);
    WideLane w;
    wide_lane_init(w, valid, dx, dy, dz);
    while (wide_trip<COUNT, XF>(nodes, stride, prims, prim_wait, stack, lane, r, w, t_min, cnt, xforms, xf_slot)) {}
    best_t = r.best_t;
    best_prim = key_result<true>(r.best_prim);
}

// ---------------------------------------------------------------------------------------------
// Closest hit, wave-synchronous form for small scenes.  Skip links only point forward, so a
// wave can sweep the entry array ONCE in order: entry e is fetched with scalar loads into
// SGPRs (one fetch per wave instead of one per lane, and a wave-uniform branch on its type) and
// the lanes whose own cursor stands at e test it.  Every lane still visits exactly the entries,
// in exactly the order and with exactly the best_t, of the per-lane walk above, so results are
// bit-identical; lanes whose cursor is elsewhere idle for that entry.  Must be called with the
// whole wave converged; lanes without a ray pass valid = false.
// ---------------------------------------------------------------------------------------------
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef const f32x16 __attribute__((address_space(4)))* const_entry_ptr;

// ---------------------------------------------------------------------------------------------
// The sweep carries TWO rays per lane.  Their arithmetic is written on 2-vectors so that it
// compiles to packed f32 instructions (v_pk_mul_f32 / v_pk_add_f32: two IEEE f32 operations
// per lane per instruction) and the per-entry scalar fetch and branch are shared by 128 rays.
// Component k of every vector belongs to ray k; nothing mixes the two rays.
// ---------------------------------------------------------------------------------------------
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f32x2 dot3v(f32x2 ax, f32x2 ay, f32x2 az, f32x2 bx, f32x2 by, f32x2 bz) {
    return (ax * bx + ay * by) + az * bz;
}
__device__ __forceinline__ f32x2 dot3s(float ax, float ay, float az, f32x2 bx, f32x2 by, f32x2 bz) {
    return (ax * bx + ay * by) + az * bz;
}

struct Hit2 { float t[2]; uint32_t prim[2]; };

// rcp_short for both components: 6 instructions for two rays instead of 22
__device__ __forceinline__ f32x2 rcp_short2(f32x2 x) {
    f32x2 r = {__builtin_amdgcn_rcpf(x.x), __builtin_amdgcn_rcpf(x.y)};
    const f32x2 one = {1.0f, 1.0f};
    f32x2 e = __builtin_elementwise_fma(-x, r, one);
    r = __builtin_elementwise_fma(e, r, r);
    e = __builtin_elementwise_fma(-x, r, one);
    r = __builtin_elementwise_fma(e, r, r);
    return r;
}

// dst = (lane in mask) ? src : dst.  One instruction with a tied operand and the lane mask
// taken straight from scalar registers: the sweep's state is updated in place, in one block
// after the branches, and so stays in the same registers around the loop (left to the register
// allocator it gets renamed by every update and copied back, 16 moves per entry).
__device__ __forceinline__ void select_into(float& dst, float src, unsigned long long mask) {
    asm("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(dst) : "v"(src), "s"(mask));
}
__device__ __forceinline__ void select_into(uint32_t& dst, uint32_t src, unsigned long long mask) {
    asm("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(dst) : "v"(src), "s"(mask));
}

// SPHERES = false drops the sphere branch for scenes that have none.  TIE: the entry array is not in
// the reference's order (a surface-area tree): exact ties in t go to the larger primitive key.
template <bool COUNT, bool SPHERES, bool TIE>
__device__ __forceinline__ void closest_hit_sweep2(const float4* __restrict__ ent_, uint32_t n_ent, bool valid0,
                                                   bool valid1, f32x2 ox, f32x2 oy, f32x2 oz, f32x2 dx, f32x2 dy,
                                                   f32x2 dz, float t_min, float t_max, Hit2& h, Counts& cnt) {
    const const_entry_ptr ent = (const_entry_ptr)(uintptr_t)ent_;
    const f32x2 one = {1.0f, 1.0f};
    const f32x2 ix = one / dx, iy = one / dy, iz = one / dz;
    const bool nx0 = ix.x < 0.0f, ny0 = iy.x < 0.0f, nz0 = iz.x < 0.0f;   // math.rs:155: swap when inv_d < 0
    const bool nx1 = ix.y < 0.0f, ny1 = iy.y < 0.0f, nz1 = iz.y < 0.0f;
    const float EPS = 1.1920929e-7f;
    float bt0 = t_max, bt1 = t_max;
    uint32_t bp0 = TIE ? KEY_NONE : RT_MISS, bp1 = bp0;
    uint32_t cur0 = valid0 ? 0u : n_ent, cur1 = valid1 ? 0u : n_ent;
    // Next entry at or after e that some lane stands on: a scalar scan that leaves the vector
    // state alone; on0 / on1 = lanes whose ray 0 / ray 1 is at that entry.  The device array ends
    // with a TAG_END entry at index n_ent, where finished (and absent) rays park their cursor, so
    // the scan needs no bound and the walk is over when it fetches that entry.
    unsigned long long on0, on1;
    uint32_t e = 0;
    auto scan = [&]() {
        for (;;) {
            on0 = __ballot(cur0 == e);
            on1 = __ballot(cur1 == e);
            if ((on0 | on1) != 0ull) break;
            e++;
        }
    };
    scan();
    f32x16 E = ent[e];                                 // one s_load_dwordx16
    // (the loop tests at the bottom, after the state update, so that the only values alive on the
    // back edge are the updated ones and they keep their registers)
    if ((f2u(E[3]) >> TAG_SHIFT) != TAG_END) do {
        const uint32_t w = f2u(E[3]);
        const uint32_t tag = w >> TAG_SHIFT;
        // What the entry does to a lane that is on it.  The state is written once, after the
        // branches (a box leaves hit0 / hit1 empty and the candidate values undefined): with a
        // single update per trip the state keeps its registers around the loop.
        uint32_t nxt0, nxt1;
        unsigned long long hit0 = 0ull, hit1 = 0ull;
        float nt0, nt1;
        uint32_t np0, np1;
        if (tag == TAG_BOX) {
            if (COUNT) cnt.box += (uint32_t)((on0 >> (threadIdx.x & 63u)) & 1ull) + (uint32_t)((on1 >> (threadIdx.x & 63u)) & 1ull);
            // math.rs:153-162 as in box_step (a NaN best_t stays a NaN: near_or)
            const f32x2 x0 = (E[0] - ox) * ix, x1 = (E[4] - ox) * ix;
            const f32x2 y0 = (E[1] - oy) * iy, y1 = (E[5] - oy) * iy;
            const f32x2 z0 = (E[2] - oz) * iz, z1 = (E[6] - oz) * iz;
            const float tmn0 = fmaxf(fmaxf(fmaxf(nx0 ? x1.x : x0.x, ny0 ? y1.x : y0.x), nz0 ? z1.x : z0.x), t_min);
            const float tmx0 = near_or(fminf(fminf(nx0 ? x0.x : x1.x, ny0 ? y0.x : y1.x), nz0 ? z0.x : z1.x), bt0);
            const float tmn1 = fmaxf(fmaxf(fmaxf(nx1 ? x1.y : x0.y, ny1 ? y1.y : y0.y), nz1 ? z1.y : z0.y), t_min);
            const float tmx1 = near_or(fminf(fminf(nx1 ? x0.y : x1.y, ny1 ? y0.y : y1.y), nz1 ? z0.y : z1.y), bt1);
            const uint32_t skip = w & SKIP_MASK;
            nxt0 = tmx0 <= tmn0 ? skip : e + 1u;
            nxt1 = tmx1 <= tmn1 ? skip : e + 1u;
            asm("" : "=v"(nt0), "=v"(nt1), "=v"(np0), "=v"(np1));   // no candidate: any register will do
        } else {
            nxt0 = nxt1 = e + 1u;
            if (!SPHERES || tag == TAG_TRI) {
                if (COUNT) cnt.tri += (uint32_t)((on0 >> (threadIdx.x & 63u)) & 1ull) + (uint32_t)((on1 >> (threadIdx.x & 63u)) & 1ull);
                // E[0..2] = v0, E[4..6] = v0v1, E[8..10] = v0v2
                const f32x2 px = dy * E[10] - dz * E[9];
                const f32x2 py = dz * E[8] - dx * E[10];
                const f32x2 pz = dx * E[9] - dy * E[8];
                const f32x2 det = dot3s(E[4], E[5], E[6], px, py, pz);
                // lanes with |det| < eps never hit, so only huge / non-finite determinants need
                // the full division (shape.rs:434: inv_det = 1.0 / det)
                const unsigned long long wide = (on0 & __ballot(!(fabsf(det.x) <= RCP_SHORT_MAX))) |
                                                (on1 & __ballot(!(fabsf(det.y) <= RCP_SHORT_MAX)));
                f32x2 inv_det;
                if (wide != 0ull) inv_det = one / det;
                else inv_det = rcp_short2(det);
                const f32x2 tx = ox - E[0], ty = oy - E[1], tz = oz - E[2];
                const f32x2 u = dot3v(tx, ty, tz, px, py, pz) * inv_det;
                const f32x2 qx = ty * E[6] - tz * E[5];
                const f32x2 qy = tz * E[4] - tx * E[6];
                const f32x2 qz = tx * E[5] - ty * E[4];
                const f32x2 v = dot3v(dx, dy, dz, qx, qy, qz) * inv_det;
                const f32x2 t = dot3s(E[8], E[9], E[10], qx, qy, qz) * inv_det;
                const f32x2 uv = u + v;
                // the reference's early returns (shape.rs:428-445) and acceptance test (shape.rs:297)
                // (one ballot per comparison, combined in scalar registers: each is a single
                // compare that writes its lane mask)
                const uint32_t id = f2u(E[7]);
                unsigned long long near0 = __ballot(t.x < bt0), near1 = __ballot(t.y < bt1);
                if (TIE) {
                    near0 |= __ballot(t.x == bt0) & __ballot(id > (bp0 & KEY_MASK));
                    near1 |= __ballot(t.y == bt1) & __ballot(id > (bp1 & KEY_MASK));
                }
                hit0 = on0 & __ballot(!(fabsf(det.x) < EPS)) & __ballot(!(u.x < 0.0f)) & __ballot(!(u.x > 1.0f)) &
                       __ballot(!(v.x < 0.0f)) & __ballot(!(uv.x > 1.0f)) & __ballot(t.x >= t_min) & near0;
                hit1 = on1 & __ballot(!(fabsf(det.y) < EPS)) & __ballot(!(u.y < 0.0f)) & __ballot(!(u.y > 1.0f)) &
                       __ballot(!(v.y < 0.0f)) & __ballot(!(uv.y > 1.0f)) & __ballot(t.y >= t_min) & near1;
                nt0 = t.x; nt1 = t.y;
                // shape.rs:424: backface = det < EPS.  An accepted hit has |det| >= EPS (and a NaN
                // det gives a NaN t, never accepted), so there the flag is det's sign bit.
                static_assert(PRIM_BACKFACE == 0x80000000u, "the backface flag is the sign bit");
                np0 = id | (f2u(det.x) & PRIM_BACKFACE);
                np1 = id | (f2u(det.y) & PRIM_BACKFACE);
            } else {
                if (COUNT) cnt.sph += (uint32_t)((on0 >> (threadIdx.x & 63u)) & 1ull) + (uint32_t)((on1 >> (threadIdx.x & 63u)) & 1ull);
                // E[0..2] = centre, E[4] = radius
                const f32x2 ocx = ox - E[0], ocy = oy - E[1], ocz = oz - E[2];
                const f32x2 qa = dot3v(dx, dy, dz, dx, dy, dz);
                const f32x2 half_b = dot3v(ocx, ocy, ocz, dx, dy, dz);
                const f32x2 qc = dot3v(ocx, ocy, ocz, ocx, ocy, ocz) - E[4] * E[4];
                const f32x2 disc = half_b * half_b - qa * qc;
                const f32x2 sqrtd = {sqrtf(disc.x), sqrtf(disc.y)};
                const f32x2 r0 = (-half_b - sqrtd) / qa;
                const f32x2 r1 = (-half_b + sqrtd) / qa;
                np0 = np1 = f2u(E[7]);
                // shape.rs:32-36, closed interval
                const bool lose0 = TIE && !(np0 > (bp0 & KEY_MASK)), lose1 = TIE && !(np1 > (bp1 & KEY_MASK));
                const bool b00 = (r0.x < t_min) | (bt0 < r0.x) | (lose0 & (r0.x == bt0));
                const bool b01 = (r1.x < t_min) | (bt0 < r1.x) | (lose0 & (r1.x == bt0));
                const bool b10 = (r0.y < t_min) | (bt1 < r0.y) | (lose1 & (r0.y == bt1));
                const bool b11 = (r1.y < t_min) | (bt1 < r1.y) | (lose1 & (r1.y == bt1));
                hit0 = on0 & __ballot(!(disc.x < 0.0f) & !(b00 & b01));
                hit1 = on1 & __ballot(!(disc.y < 0.0f) & !(b10 & b11));
                nt0 = b00 ? r1.x : r0.x;
                nt1 = b10 ? r1.y : r0.y;
            }
        }
        select_into(bt0, nt0, hit0);
        select_into(bt1, nt1, hit1);
        select_into(bp0, np0, hit0);
        select_into(bp1, np1, hit1);
        select_into(cur0, nxt0, on0);
        select_into(cur1, nxt1, on1);
        e++;
        scan();
        E = ent[e];
    } while ((f2u(E[3]) >> TAG_SHIFT) != TAG_END);
    h.t[0] = bt0; h.t[1] = bt1; h.prim[0] = key_result<TIE>(bp0); h.prim[1] = key_result<TIE>(bp1);
}

__device__ __forceinline__ void write_hit(const DevScene& sc, rt_hit* __restrict__ out, size_t r, float ox, float oy,
                                          float oz, float dx, float dy, float dz, float t, uint32_t prim) {
    const float4* geo = (const float4*)sc.prim_geo;
    rt_hit h;
    if (prim == RT_MISS) {
        h.t = 0.0f; h.prim_id = RT_MISS;
        h.p[0] = h.p[1] = h.p[2] = 0.0f; h.n[0] = h.n[1] = h.n[2] = 0.0f;
        h.front_face = 0; h.mat = 0;
    } else {
        const uint32_t rank = key_rank(prim);
        const float4 g = geo[rank];
        const uint32_t meta = sc.prim_meta[rank];
        const uint32_t id = sc.prim_order[rank];       // the id the caller knows the primitive by
        float nx, ny, nz;
        if (meta & META_SPHERE) {                      // shape.rs:41
            const float sx = ox + dx * t, sy = oy + dy * t, sz = oz + dz * t;
            nx = (sx - g.x) / g.w; ny = (sy - g.y) / g.w; nz = (sz - g.z) / g.w;
        } else {                                       // shape.rs:298-307
            const bool bf = (prim & PRIM_BACKFACE) != 0;
            nx = bf ? -g.x : g.x; ny = bf ? -g.y : g.y; nz = bf ? -g.z : g.z;
        }
        bool front = dot3(dx, dy, dz, nx, ny, nz) < 0.0f;   // camera.rs:112-119
        float px = ox + dx * t, py = oy + dy * t, pz = oz + dz * t;   // Ray::at
        if (sc.n_xforms != 0u && !(meta & META_SPHERE) && f2u(g.w) != 0u)
            xf_hit((const float4*)sc.xforms, g, (prim & PRIM_BACKFACE) != 0, ox, oy, oz, dx, dy, dz, t, px, py, pz, nx, ny,
                   nz, front);
        h.t = t; h.prim_id = id;
        h.p[0] = px; h.p[1] = py; h.p[2] = pz;
        h.n[0] = front ? nx : -nx; h.n[1] = front ? ny : -ny; h.n[2] = front ? nz : -nz;
        h.front_face = front ? 1u : 0u;
        h.mat = meta & MAT_MASK;
    }
    out[r] = h;
}

// dynamic LDS: the per-lane stacks of the ordered walk (stack_depth * 64 words per wave), or the
// workgroup's copy of the unit array when a small scene is walked out of LDS
extern __shared__ uint32_t lds_stack[];

// SWEEP: 0 = per-lane walk, 1 = wave sweep (scene without spheres), 2 = wave sweep (any scene),
//        3 = ordered per-lane walk, 4 = per-lane walk over a copy of the units in LDS (k_path only),
//        5 / 6 = 0 / 4 with idle lanes refilled (k_path only), 7 = per-lane walk of the ray's octant copy,
//        8 = 7 with idle lanes refilled (k_path only), 9 = per-lane walk of a scene with transformed meshes,
//        11 = stack walk of the wide tree (10 is the split walk, k_path only), 12 = 11 with idle lanes refilled (k_path only;
//        16 = 12 built for six waves per SIMD; 17 / 18 = 11 / 12 for scenes with transformed meshes),
//        13 / 14 / 15 = 9 out of LDS / with idle lanes refilled / both (k_path only)
template <bool COUNT, int SWEEP, bool TIE>
__global__ void __launch_bounds__(RT_BLOCK)
k_trace_api(DevScene sc, const float* __restrict__ rays, size_t n, float t_min, float t_max,
            rt_hit* __restrict__ out, Totals* partials) {
    const float4* ent = (const float4*)sc.entries;
    Counts cnt{0, 0, 0};
    if (SWEEP == 1 || SWEEP == 2) {
        // the same two-rays-per-lane sweep the renderer uses; every wave takes 128 consecutive rays
        const size_t wave = (size_t)blockIdx.x * (RT_BLOCK / 64) + (threadIdx.x >> 6);
        const size_t n_waves = (size_t)gridDim.x * (RT_BLOCK / 64);
        const uint32_t lane = threadIdx.x & 63u;
        for (size_t r0 = wave * 128; r0 < n; r0 += n_waves * 128) {
            const size_t ra = r0 + lane, rb = ra + 64;
            const bool va = ra < n, vb = rb < n;
            float o0[3] = {0, 0, 0}, d0[3] = {1, 1, 1}, o1[3] = {0, 0, 0}, d1[3] = {1, 1, 1};
            if (va) for (int k = 0; k < 3; k++) { o0[k] = rays[ra * 6 + k]; d0[k] = rays[ra * 6 + 3 + k]; }
            if (vb) for (int k = 0; k < 3; k++) { o1[k] = rays[rb * 6 + k]; d1[k] = rays[rb * 6 + 3 + k]; }
            Hit2 h;
            closest_hit_sweep2<COUNT, SWEEP == 2, TIE>(ent, sc.n_entries, va, vb, f32x2{o0[0], o1[0]}, f32x2{o0[1], o1[1]},
                                      f32x2{o0[2], o1[2]}, f32x2{d0[0], d1[0]}, f32x2{d0[1], d1[1]},
                                      f32x2{d0[2], d1[2]}, t_min, t_max, h, cnt);
            if (va) write_hit(sc, out, ra, o0[0], o0[1], o0[2], d0[0], d0[1], d0[2], h.t[0], h.prim[0]);
            if (vb) write_hit(sc, out, rb, o1[0], o1[1], o1[2], d1[0], d1[1], d1[2], h.t[1], h.prim[1]);
        }
    } else {
        // whole waves enter the walk together; the lanes past the end carry no ray
        const size_t stride = (size_t)gridDim.x * RT_BLOCK;
        for (size_t r0 = (size_t)blockIdx.x * RT_BLOCK; r0 < n; r0 += stride) {
            const size_t r = r0 + threadIdx.x;
            const bool valid = r < n;
            float ox = 0.0f, oy = 0.0f, oz = 0.0f, dx = 1.0f, dy = 1.0f, dz = 1.0f;
            if (valid) {
                ox = rays[r * 6 + 0]; oy = rays[r * 6 + 1]; oz = rays[r * 6 + 2];
                dx = rays[r * 6 + 3]; dy = rays[r * 6 + 4]; dz = rays[r * 6 + 5];
            }
            float t = t_max;
            uint32_t prim;
            if (SWEEP == 3)
                closest_hit_ordered<COUNT>((const float4*)sc.units, sc.n_units, sc.prim_wait,
                                           lds_stack + (threadIdx.x >> 6) * sc.stack_depth * 64u, threadIdx.x & 63u,
                                           valid, ox, oy, oz, dx, dy, dz, t_min, t, prim, cnt);
            else if (SWEEP == 7)
                closest_hit<COUNT, true>((const float4*)sc.units_oct, sc.n_units, sc.prim_wait, valid, ox, oy, oz, dx, dy, dz, t_min, t, prim, cnt);
            else if (SWEEP == 9)
                closest_hit<COUNT, false, TIE, true>((const float4*)sc.walk, sc.n_walk, sc.prim_wait, valid, ox, oy, oz, dx, dy, dz, t_min, t, prim, cnt,
                                                     (const float4*)sc.xforms);
            else if (SWEEP == 17)
                closest_hit_wide<COUNT, true>((const uint4*)sc.wide_nodes, sc.wide_stride, (const float4*)sc.wide_prims, sc.prim_wait,
                                              lds_stack + (threadIdx.x >> 6) * sc.wide_stack * 64u, threadIdx.x & 63u, valid, ox, oy, oz, dx,
                                              dy, dz, t_min, t, prim, cnt, (const float4*)sc.xforms, sc.wide_stack - WIDE_XF_RAY_SLOTS);
            else if (SWEEP == 11)
                closest_hit_wide<COUNT>((const uint4*)sc.wide_nodes, sc.wide_stride, (const float4*)sc.wide_prims, sc.prim_wait,
                                        lds_stack + (threadIdx.x >> 6) * sc.wide_stack * 64u, threadIdx.x & 63u, valid,
                                        ox, oy, oz, dx, dy, dz, t_min, t, prim, cnt);
            else
                closest_hit<COUNT, false, TIE>((const float4*)sc.walk, sc.n_walk, sc.prim_wait, valid, ox, oy, oz, dx, dy, dz, t_min, t, prim, cnt);
            if (valid) write_hit(sc, out, r, ox, oy, oz, dx, dy, dz, t, prim);
        }
    }
    if (COUNT) {
        // The workgroup's visit counts go to its own record of `partials` (every workgroup writes
        // its record, zero or not, so the array needs no clearing; k_sum_partials adds the records
        // up): wave reduction, then the workgroup's waves through LDS.  No atomics.
        __shared__ unsigned long long red[RT_BLOCK / 64][3];
        unsigned long long b = cnt.box, t = cnt.tri, sp = cnt.sph;
        for (int off = 32; off > 0; off >>= 1) {
            b += __shfl_down(b, off); t += __shfl_down(t, off); sp += __shfl_down(sp, off);
        }
        if ((threadIdx.x & 63u) == 0) {
            red[threadIdx.x >> 6][0] = b; red[threadIdx.x >> 6][1] = t; red[threadIdx.x >> 6][2] = sp;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            Totals tot{0ull, 0ull, 0ull, 0ull};
            for (uint32_t w = 0; w < RT_BLOCK / 64; w++) { tot.box += red[w][0]; tot.tri += red[w][1]; tot.sph += red[w][2]; }
            partials[blockIdx.x] = tot;
        }
    }
}

#ifdef RT_HAS_SWEEP
// box / triangle / sphere totals of a traced batch: the sum of the workgroups' records, written
// (not added) to out3[0..2]
__global__ void __launch_bounds__(RT_BLOCK)
k_sum_partials(const Totals* __restrict__ partials, uint32_t n_records, unsigned long long* __restrict__ out3) {
    __shared__ unsigned long long red[RT_BLOCK / 64][3];
    unsigned long long b = 0, t = 0, sp = 0;
    for (uint32_t i = threadIdx.x; i < n_records; i += RT_BLOCK) {
        const Totals p = partials[i];
        b += p.box; t += p.tri; sp += p.sph;
    }
    for (int off = 32; off > 0; off >>= 1) {
        b += __shfl_down(b, off); t += __shfl_down(t, off); sp += __shfl_down(sp, off);
    }
    if ((threadIdx.x & 63u) == 0) {
        red[threadIdx.x >> 6][0] = b; red[threadIdx.x >> 6][1] = t; red[threadIdx.x >> 6][2] = sp;
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        unsigned long long v = 0;
        for (uint32_t w = 0; w < RT_BLOCK / 64; w++) v += red[w][threadIdx.x];
        out3[threadIdx.x] = v;
    }
}
#endif

// ---------------------------------------------------------------------------------------------
// wave-level compaction: ballot + lane prefix.  `count` is wave-uniform.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t lane_prefix(unsigned long long m) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}
// position of this lane's element when the lanes with pred set append behind `count`
__device__ __forceinline__ uint32_t wave_append(bool pred, uint32_t& count) {
    const unsigned long long m = __ballot(pred);
    const uint32_t pos = count + lane_prefix(m);
    count += (uint32_t)__popcll(m);
    return pos;
}

// ---------------------------------------------------------------------------------------------
// counter-based RNG: a 64-bit key per (pixel, sample) made by raygen, then
// value = hash(key, counter) with counter = depth * 16 + dimension
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t pcg(uint32_t v) {
    const uint32_t s = v * 747796405u + 2891336453u;
    const uint32_t w = ((s >> ((s >> 28u) + 4u)) ^ s) * 277803737u;
    return (w >> 22u) ^ w;
}
__device__ __forceinline__ uint32_t rng_u32(uint32_t klo, uint32_t khi, uint32_t ctr) {
    return pcg(pcg(klo + ctr * 0x9E3779B9u) ^ khi);
}
// uniform in [0,1)
__device__ __forceinline__ float rng_f01(uint32_t klo, uint32_t khi, uint32_t ctr) {
    return (float)(rng_u32(klo, khi, ctr) >> 8) * 5.9604645e-8f;
}
constexpr uint32_t RNG_CTR_CAMERA = 0xFFFF0000u;

// uniform on the unit sphere: the distribution of RandomSeries::random_on_unit_sphere
// (math.rs:78-80) without its rejection loop
__device__ __forceinline__ void sample_sphere(float u1, float u2, float& x, float& y, float& z) {
    z = 1.0f - 2.0f * u1;
    const float r = sqrtf(fmaxf(0.0f, 1.0f - z * z));
    x = r * __builtin_amdgcn_cosf(u2);         // cos(2*pi*u2): the hardware takes revolutions
    y = r * __builtin_amdgcn_sinf(u2);
}
// uniform in the unit ball: the distribution of random_in_unit_sphere (math.rs:69-76)
__device__ __forceinline__ void sample_ball(float u1, float u2, float u3, float& x, float& y, float& z) {
    sample_sphere(u1, u2, x, y, z);
    const float s = cbrtf(u3);
    x *= s; y *= s; z *= s;
}

// decode a pixel slot of the batch (8x8 tiles, row-major inside a tile)
__device__ __forceinline__ bool slot_pixel(const BatchGeom& g, uint32_t q, uint32_t& x, uint32_t& y) {
    const uint32_t i = q >> 6, in = q & 63u, tile = g.tile_base + i + (i >> g.tile_run_shift) * g.tile_gap;
    const uint32_t tx = tile % g.tiles_x;
    uint32_t ty = tile / g.tiles_x;                        // index among the owned tile rows
    if (g.band_count > 1u)
        ty = ((ty / g.band_tile_rows) * g.band_count + g.band_index) * g.band_tile_rows + ty % g.band_tile_rows;
    x = g.x0 + tx * 8u + (in & 7u);
    y = g.y0 + ty * 8u + (in >> 3);
    return x < g.x1 && y < g.y1;
}

// ---------------------------------------------------------------------------------------------
// raygen: the per-sample head of the tile loop (main.rs:362-364) + Camera::get_ray
// (camera.rs:80-89).  Wave s fills segment s with the camera rays of its path slots.
// ---------------------------------------------------------------------------------------------
// camera rays of segment `seg` into (A, B, C); returns how many there are
__device__ __forceinline__ uint32_t raygen_segment(const rt_camera& cam, const BatchGeom& g, uint32_t seed_lo,
                                                   uint32_t seed_hi, float4* __restrict__ A, float4* __restrict__ B,
                                                   float4* __restrict__ C, uint32_t base, uint32_t lane) {
    const float inv_w = 1.0f / (float)(g.width - 1u);      // main.rs:330-331
    const float inv_h = 1.0f / (float)(g.height - 1u);
    uint32_t out = 0;
    for (uint32_t j0 = 0; j0 < g.seg_len; j0 += 64u) {
        const uint32_t s = base + j0 + lane;
        bool valid = s < g.n_slots;
        uint32_t x = 0, y = 0;
        const uint32_t k = s / g.n_pix_slots, q = s - k * g.n_pix_slots;
        valid = valid && slot_pixel(g, q, x, y);
        float4 a = make_float4(0, 0, 0, 0), b = a, c = a;
        if (valid) {
            const uint32_t pix = y * g.width + x, smp = g.sample_base + k;
            const uint32_t klo = pcg(pix ^ pcg(smp ^ pcg(seed_lo)));
            const uint32_t khi = pcg(smp + pcg(pix + pcg(seed_hi ^ 0x85EBCA6Bu)));
            const float u = ((float)x + rng_f01(klo, khi, RNG_CTR_CAMERA + 0u)) * inv_w;
            const float v = ((float)y + rng_f01(klo, khi, RNG_CTR_CAMERA + 1u)) * inv_h;
            const float sx = u, sy = 1.0f - v;
            float offx = 0.0f, offy = 0.0f, offz = 0.0f;
            if (cam.lens_radius != 0.0f) {                 // uniform disk (math.rs:60-67)
                const float r = cam.lens_radius * sqrtf(rng_f01(klo, khi, RNG_CTR_CAMERA + 2u));
                const float ph = rng_f01(klo, khi, RNG_CTR_CAMERA + 3u);
                const float rx = r * __builtin_amdgcn_cosf(ph), ry = r * __builtin_amdgcn_sinf(ph);
                offx = cam.u[0] * rx + cam.v[0] * ry;
                offy = cam.u[1] * rx + cam.v[1] * ry;
                offz = cam.u[2] * rx + cam.v[2] * ry;
            }
            a.x = cam.origin[0] + offx; a.y = cam.origin[1] + offy; a.z = cam.origin[2] + offz;
            a.w = u2f(s);
            b.x = cam.lower_left[0] + sx * cam.horizontal[0] + sy * cam.vertical[0] - cam.origin[0] - offx;
            b.y = cam.lower_left[1] + sx * cam.horizontal[1] + sy * cam.vertical[1] - cam.origin[1] - offy;
            b.z = cam.lower_left[2] + sx * cam.horizontal[2] + sy * cam.vertical[2] - cam.origin[2] - offz;
            b.w = u2f(klo);
            c = make_float4(1.0f, 1.0f, 1.0f, u2f(khi));
        }
        const uint32_t pos = base + wave_append(valid, out);
        if (valid) { A[pos] = a; B[pos] = b; C[pos] = c; }
    }
    return out;
}

#ifdef RT_HAS_SWEEP
__global__ void __launch_bounds__(RT_WAVE)
k_raygen(rt_camera cam, BatchGeom g, uint32_t seed_lo, uint32_t seed_hi, float4* __restrict__ A,
         float4* __restrict__ B, float4* __restrict__ C, uint32_t* __restrict__ segcnt) {
    const uint32_t seg = blockIdx.x * RT_WPB + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
    if (seg >= g.n_seg) return;
    const uint32_t out = raygen_segment(cam, g, seed_lo, seed_hi, A, B, C, seg * g.seg_len, lane);
    if (lane == 0) segcnt[seg] = out;
}

#endif  // RT_HAS_SWEEP

// ---------------------------------------------------------------------------------------------
// Material::scatter (material.rs) for one ray (a, b, c: origin + path slot, direction + rng key,
// throughput + rng key) that hit `prim` at t: the scattered ray in (na, nb, nc); false when the
// material absorbs it (scatter() == false: the path ends with what it has emitted).
// ---------------------------------------------------------------------------------------------
template <int KIND>
__device__ __forceinline__ bool shade_one(const DevScene& sc, const float4& a, const float4& b, const float4& c, float t,
                                          uint32_t prim, uint32_t ctr, float4& na, float4& nb, float4& nc) {
    const float4* geo = (const float4*)sc.prim_geo;
    const float4* mats = (const float4*)sc.materials;
    bool alive = false;
    {
        const uint32_t id = key_rank(prim);
        const uint32_t meta = sc.prim_meta[id];
        const float4 g = geo[id];
        const float4 m0 = mats[2 * (meta & MAT_MASK)];
        const uint32_t klo = f2u(b.w), khi = f2u(c.w);
        // hit point and face-forwarded normal
        float px = a.x + b.x * t, py = a.y + b.y * t, pz = a.z + b.z * t;
        float nx, ny, nz;
        if (meta & META_SPHERE) {
            nx = (px - g.x) / g.w; ny = (py - g.y) / g.w; nz = (pz - g.z) / g.w;
        } else {
            const bool bf = (prim & PRIM_BACKFACE) != 0;
            nx = bf ? -g.x : g.x; ny = bf ? -g.y : g.y; nz = bf ? -g.z : g.z;
        }
        bool front = dot3(b.x, b.y, b.z, nx, ny, nz) < 0.0f;
        if (sc.n_xforms != 0u && !(meta & META_SPHERE) && f2u(g.w) != 0u)   // a triangle of a transformed mesh
            xf_hit((const float4*)sc.xforms, g, (prim & PRIM_BACKFACE) != 0, a.x, a.y, a.z, b.x, b.y, b.z, t, px, py, pz,
                   nx, ny, nz, front);
        if (!front) { nx = -nx; ny = -ny; nz = -nz; }
        float ddx, ddy, ddz, ar = m0.x, ag = m0.y, ab = m0.z;
        if (KIND == RT_MAT_LAMBERT) {                  // material.rs:27-56
            float sx, sy, sz;
            sample_sphere(rng_f01(klo, khi, ctr), rng_f01(klo, khi, ctr + 1u), sx, sy, sz);
            float ex = nx + sx, ey = ny + sy, ez = nz + sz;
            if (fabsf(ex) + fabsf(ey) + fabsf(ez) < 1.1920929e-7f) { ex = nx; ey = ny; ez = nz; }
            const float len = sqrtf(dot3(ex, ey, ez, ex, ey, ez));
            ddx = ex / len; ddy = ey / len; ddz = ez / len;
            const float4 m1 = mats[2 * (meta & MAT_MASK) + 1];
            if (f2u(m1.y) & RT_MAT_FLAG_NORMAL_SHADING) {
                ar = (nx + 1.0f) * 0.5f; ag = (ny + 1.0f) * 0.5f; ab = (nz + 1.0f) * 0.5f;
            }
            alive = true;
        } else if (KIND == RT_MAT_METAL) {             // material.rs:63-80
            const float dn2 = 2.0f * dot3(b.x, b.y, b.z, nx, ny, nz);
            const float rx = b.x - dn2 * nx, ry = b.y - dn2 * ny, rz = b.z - dn2 * nz;
            float sx, sy, sz;
            sample_ball(rng_f01(klo, khi, ctr), rng_f01(klo, khi, ctr + 1u), rng_f01(klo, khi, ctr + 2u),
                        sx, sy, sz);
            ddx = rx + m0.w * sx; ddy = ry + m0.w * sy; ddz = rz + m0.w * sz;
            alive = dot3(rx, ry, rz, nx, ny, nz) > 0.0f;
        } else {                                       // material.rs:87-120
            const float ratio = front ? 1.0f / m0.w : m0.w;
            const float len = sqrtf(dot3(b.x, b.y, b.z, b.x, b.y, b.z));
            const float ux = b.x / len, uy = b.y / len, uz = b.z / len;
            const float cos_t = fminf(dot3(-ux, -uy, -uz, nx, ny, nz), 1.0f);
            const float sin_t = sqrtf(1.0f - cos_t * cos_t);
            bool refl = ratio * sin_t > 1.0f;
            if (!refl) {                               // reflectance (math.rs:105-110)
                float r0 = (1.0f - ratio) / (1.0f + ratio);
                r0 = r0 * r0;
                const float om = 1.0f - cos_t;
                const float om2 = om * om;
                refl = r0 + (1.0f - r0) * (om2 * om2 * om) > rng_f01(klo, khi, ctr);
            }
            if (refl) {                                // reflect (math.rs:94-96)
                const float dn2 = 2.0f * dot3(ux, uy, uz, nx, ny, nz);
                ddx = ux - dn2 * nx; ddy = uy - dn2 * ny; ddz = uz - dn2 * nz;
            } else {                                   // refract (math.rs:98-103)
                const float ppx = ratio * (ux + cos_t * nx), ppy = ratio * (uy + cos_t * ny),
                            ppz = ratio * (uz + cos_t * nz);
                const float par = -sqrtf(fabsf(1.0f - dot3(ppx, ppy, ppz, ppx, ppy, ppz)));
                ddx = ppx + par * nx; ddy = ppy + par * ny; ddz = ppz + par * nz;
            }
            alive = true;
        }
        na = make_float4(px, py, pz, a.w);
        nb = make_float4(ddx, ddy, ddz, b.w);
        nc = make_float4(c.x * ar, c.y * ag, c.z * ab, c.w);
    }
    return alive;
}

// what happens to a traced ray: terminate it (sky, emitter, depth cap -> its radiance goes to
// Lbuf) or record the hit and report the material kind it must be queued under (7 = none)
// RECORD = 0: the caller shades the ray itself right away and needs no hit record; 2: it does so for
// Lambertian hits only.
template <int RECORD = 1>
__device__ __forceinline__ uint32_t finish_ray(const float4* __restrict__ mats, const uint32_t* __restrict__ prim_meta,
                                               const float4* __restrict__ C, uint2* __restrict__ hits,
                                               float4* __restrict__ Lbuf, uint32_t i, uint32_t slot, float dx,
                                               float dy, float dz, float t, uint32_t prim, uint32_t last_depth,
                                               float sky_intensity) {
    if (prim == RT_MISS) {
        // sky_color (main.rs:31-38): gradient on the y of the normalised direction
        const float4 c = C[i];
        const float len = sqrtf(dot3(dx, dy, dz, dx, dy, dz));
        const float tt = 0.5f * (dy / len + 1.0f);
        const float r = ((1.0f - tt) * 1.0f + tt * 0.5f) * sky_intensity;
        const float gg = ((1.0f - tt) * 1.0f + tt * 0.7f) * sky_intensity;
        const float bb = ((1.0f - tt) * 1.0f + tt * 1.0f) * sky_intensity;
        Lbuf[slot] = make_float4(c.x * r, c.y * gg, c.z * bb, 0.0f);
        return 7u;
    }
    const uint32_t mk = prim_meta[key_rank(prim)];
    const uint32_t k = (mk >> KIND_SHIFT) & 3u;
    if (k == RT_MAT_EMISSIVE) {                            // emit, no scatter (material.rs:126-130)
        const float4 c = C[i];
        const float4 m = mats[2 * (mk & MAT_MASK)];
        Lbuf[slot] = make_float4(c.x * m.x, c.y * m.y, c.z * m.z, 0.0f);
        return 7u;
    }
    if (last_depth) {                                      // the next ray_color returns 0 (main.rs:41-43)
        Lbuf[slot] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        return 7u;
    }
    if (RECORD == 1 || (RECORD == 2 && k != RT_MAT_LAMBERT)) hits[i] = make_uint2(f2u(t), prim);
    return k;
}

// ---------------------------------------------------------------------------------------------
// trace: closest hit for every live ray of the wave's segment, terminate or queue by kind
// ---------------------------------------------------------------------------------------------
// Register budget of the trace kernel: 8 waves per SIMD need at most 64 VGPRs and, because a wave's
// SGPR allocation is rounded up to 16 and carries 16 more, at most 80 SGPRs (96 would admit 6).
// The traversal is latency-bound (a wave issues vector work during a fifth of its lifetime), so
// resident waves are what fills the SIMD.  The SGPR cap makes the compiler park the kernel
// arguments that are only used after the walk in lanes of a spare VGPR instead of holding them
// in SGPRs across it; nothing is spilled inside the walk.
#ifndef RT_TRACE_SGPRS
#define RT_TRACE_SGPRS 80
#endif
#ifdef RT_STAMPS
// Diagnostic build only (-DRT_STAMPS): start / end time and placement of every wave of the first
// few trace launches, dumped by the host to the file named by RT_STAMPS_FILE
// (tools/stamps_report.py reads it).
constexpr uint32_t STAMP_SLOTS = 32, STAMP_WAVES = 16384;
__device__ unsigned long long rt_stamps[STAMP_SLOTS * STAMP_WAVES * 4];
#endif
// The per-lane walk over a whole segment with idle lanes refilled.  Walks differ a lot in length
// (a ray that crosses a dense mesh visits ten times the entries of one that meets a wall), and a
// wave that takes 64 rays and runs until the longest walk ends issues most of its loads for a
// handful of lanes: every wave-wide 16-byte load costs the CU's address unit 16 cycles whatever
// the number of lanes in it, and that unit is what bounds the walk from memory.  Here a lane whose
// walk has ended waits only until `refill_min` lanes are idle; these then finish their rays
// (finish_ray + queueing) and take the next rays of the segment.  Every ray is still walked by one
// lane in the reference's order; only which lane and when changes, and nothing downstream depends
// on that (queues are compacted per kind, radiance is stored per path slot).
template <bool COUNT, bool TIE, bool OCT = false, bool SPLIT = false, bool XF = false>
__device__ __forceinline__ void walk_segment(const float4* __restrict__ ent, uint32_t n_ent_, uint32_t prim_wait,
                                             uint32_t refill_min, const SplitCfg& sp, const float4* __restrict__ xforms,
                                             const float4* __restrict__ mats,
                                             const uint32_t* __restrict__ prim_meta, const float4* __restrict__ A,
                                             const float4* __restrict__ B, const float4* __restrict__ C,
                                             uint2* __restrict__ hits, uint32_t* __restrict__ q0,
                                             uint32_t* __restrict__ q1, uint32_t* __restrict__ q2,
                                             float4* __restrict__ Lbuf, uint32_t base, uint32_t n,
                                             uint32_t last_depth, float t_min, float sky_intensity, uint32_t& n0,
                                             uint32_t& n1, uint32_t& n2, Counts& cnt) {
    static_assert(TIE || !OCT, "the octant copies are not in the reference's order");
    static_assert(!(TIE && SPLIT), "the split walk is a reference-order form");
    static_assert(!(XF && (OCT || SPLIT)), "only the plain linked units carry the transform markers");
    WorldRay world{0.0f, 0.0f, 0.0f, 1.0f, 1.0f, 1.0f};
    LaneRay r;
    lane_ray_init<TIE>(r, 0.0f, 0.0f, 0.0f, 1.0f, 1.0f, 1.0f, __builtin_inff());
    uint32_t ray_i = 0, slot = 0;              // the lane's ray and its path slot
    bool has_ray = false;
    uint32_t next = 0;                         // rays of the segment handed out so far (wave-uniform)
    // (a lane without an entry to visit holds TAG_END in the tag word of e.a, as in closest_hit)
    const float end_word = u2f(TAG_END << TAG_SHIFT);
    const float4 ended = make_float4(0.0f, 0.0f, 0.0f, end_word);
    LaneEntry e{ended, ended, ended, ended};
    for (;;) {
        // (in the split walk a lane that waits for a unit from memory holds PEND_WORD there: it is live)
        const bool live = SPLIT ? f2u(e.a.w) != ENDED_WORD
                          : XF  ? (f2u(e.a.w) >> TAG_SHIFT) != TAG_END || f2u(e.a.x) != END_CODE_END
                                : (f2u(e.a.w) >> TAG_SHIFT) != TAG_END;
        const unsigned long long m_live = __ballot(live);
        const uint32_t n_idle = 64u - (uint32_t)__popcll(m_live);
        if (m_live == 0ull || (n_idle >= refill_min && next < n)) {
            uint32_t kind = 7u;
            if (has_ray && !live) {
                // (XF: a ray that ends inside a mesh's tree -- there is none: every tree is followed by its LEAVE
                // marker -- would hold the local direction; the sky wants the caller's)
                kind = finish_ray(mats, prim_meta, C, hits, Lbuf, ray_i, slot, XF ? world.dx : r.dx, XF ? world.dy : r.dy,
                                  XF ? world.dz : r.dz, r.best_t, key_result<TIE>(r.best_prim), last_depth, sky_intensity);
                has_ray = false;
            }
            uint32_t pos;
            pos = wave_append(kind == 0u, n0);
            if (kind == 0u) q0[base + pos] = ray_i;
            pos = wave_append(kind == 1u, n1);
            if (kind == 1u) q1[base + pos] = ray_i;
            pos = wave_append(kind == 2u, n2);
            if (kind == 2u) q2[base + pos] = ray_i;
            if (next >= n) {
                if (m_live == 0ull) break;
            } else {
                const uint32_t j = next + lane_prefix(~m_live);
                if (!live && j < n) {
                    ray_i = base + j;
                    const float4 ra = A[ray_i], rb = B[ray_i];
                    lane_ray_init<TIE>(r, ra.x, ra.y, ra.z, rb.x, rb.y, rb.z, __builtin_inff());
                    if (XF) world = WorldRay{ra.x, ra.y, ra.z, rb.x, rb.y, rb.z};
                    slot = f2u(ra.w);
                    has_ray = true;
                    if (OCT) r.i = (n_ent_ + UNITS_END) * octant_of(rb.x, rb.y, rb.z);
                    if (SPLIT) { e.a = lds_unit(sp.lds, r.i); e.b = lds_unit(sp.lds, r.i + 1); }    // (the split walk's top holds unit 0)
                    else fetch_entry(ent, r.i, e);
                }
                next = min(n, next + n_idle);
                continue;
            }
        }
        if (SPLIT) walk_trip_split<COUNT>(ent, sp, prim_wait, r, e, t_min, cnt);
        else walk_trip<COUNT, TIE, XF>(ent, prim_wait, r, e, t_min, cnt, xforms, &world);
    }
}

// The wide walk over a whole segment with idle lanes refilled, as walk_segment does for the per-lane
// walk: the wide walk is bound by instruction issue, and a wave that runs until its longest ray ends
// issues most of its instructions for a few lanes.
template <bool COUNT, bool XF = false>
__device__ __forceinline__ void wide_segment(const uint4* __restrict__ nodes, uint32_t stride, const float4* __restrict__ prims,
                                             uint32_t prim_wait, uint32_t refill_min, uint32_t* stack, uint32_t lane,
                                             const float4* __restrict__ mats, const uint32_t* __restrict__ prim_meta,
                                             const float4* __restrict__ A, const float4* __restrict__ B,
                                             const float4* __restrict__ C, uint2* __restrict__ hits,
                                             uint32_t* __restrict__ q0, uint32_t* __restrict__ q1,
                                             uint32_t* __restrict__ q2, float4* __restrict__ Lbuf, uint32_t base,
                                             uint32_t n, uint32_t last_depth, float t_min, float sky_intensity,
                                             uint32_t& n0, uint32_t& n1, uint32_t& n2, Counts& cnt,
                                             const float4* __restrict__ xforms = nullptr, uint32_t xf_slot = 0) {
    LaneRay r;
    lane_ray_init<true>(r, 0.0f, 0.0f, 0.0f, 1.0f, 1.0f, 1.0f, __builtin_inff());
    WideLane w;
    wide_lane_init(w, false, 1.0f, 1.0f, 1.0f);
    uint32_t ray_i = 0, slot = 0;              // the lane's ray and its path slot
    bool has_ray = false;
    uint32_t next = 0;                         // rays of the segment handed out so far (wave-uniform)
    for (;;) {
        const bool live = wide_live(w);
        const unsigned long long m_live = __ballot(live);
        const uint32_t n_idle = 64u - (uint32_t)__popcll(m_live);
        if (m_live == 0ull || (n_idle >= refill_min && next < n)) {
            uint32_t kind = 7u;
            if (has_ray && !live) {
                kind = finish_ray(mats, prim_meta, C, hits, Lbuf, ray_i, slot, r.dx, r.dy, r.dz, r.best_t,
                                  key_result<true>(r.best_prim), last_depth, sky_intensity);
                has_ray = false;
            }
            uint32_t pos;
            pos = wave_append(kind == 0u, n0);
            if (kind == 0u) q0[base + pos] = ray_i;
            pos = wave_append(kind == 1u, n1);
            if (kind == 1u) q1[base + pos] = ray_i;
            pos = wave_append(kind == 2u, n2);
            if (kind == 2u) q2[base + pos] = ray_i;
            if (next >= n) {
                if (m_live == 0ull) break;
            } else {
                const uint32_t j = next + lane_prefix(~m_live);
                if (!live && j < n) {
                    ray_i = base + j;
                    const float4 ra = A[ray_i], rb = B[ray_i];
                    lane_ray_init<true>(r, ra.x, ra.y, ra.z, rb.x, rb.y, rb.z, __builtin_inff());
                    wide_lane_init(w, true, rb.x, rb.y, rb.z);
                    slot = f2u(ra.w);
                    has_ray = true;
                }
                next = min(n, next + n_idle);
                continue;
            }
        }
        wide_trip<COUNT, XF>(nodes, stride, prims, prim_wait, stack, lane, r, w, t_min, cnt, xforms, xf_slot);
    }
}

// Closest hit for the n live rays of the segment at `base`: terminates paths (sky, emitter, depth
// cap) or records the hit and queues the ray under its material kind; n0 / n1 / n2 return the
// queue lengths.  Whole wave, converged.
// FUSE >= 0 (the fused kernel): a ray that hits a scattering material is shaded in registers right
// after its closest hit -- no hit record, no queue, no second pass over the ray records -- and the
// scattered ray goes straight to the segment's place in the next ray buffer (fuse->out counts them).
// FUSE = RT_MAT_LAMBERT: scenes whose only scattering kind is Lambertian (the next depth then sees
// the segment the queue would have given it, bit for bit); FUSE_ANY: every kind, one after the
// other under the lanes' masks (the next depth's rays then stand in ray order instead of sorted by
// material, which changes no path: a path's random numbers are keyed by pixel, sample and depth);
// FUSE_LAMBERT_MIXED: scenes with several kinds -- Lambertian hits are shaded here, metal and
// dielectric hits are recorded and queued for the shade stage, which appends behind them.
constexpr int FUSE_ANY = 7, FUSE_LAMBERT_MIXED = 4;
constexpr int fuse_record(int FUSE) { return FUSE < 0 ? 1 : FUSE == FUSE_LAMBERT_MIXED ? 2 : 0; }
struct FuseOut {
    float4 *nA, *nB, *nC;      // the next depth's ray records
    uint32_t ctr;              // RNG counter base of this depth
    uint32_t out;              // scattered rays so far
};
template <bool COUNT, int SWEEP, bool TIE, int FUSE = -1>
__device__ __forceinline__ void trace_segment(const DevScene& sc, const float4* __restrict__ A,
                                              const float4* __restrict__ B, const float4* __restrict__ C,
                                              uint2* __restrict__ hits, uint32_t* __restrict__ q0,
                                              uint32_t* __restrict__ q1, uint32_t* __restrict__ q2,
                                              float4* __restrict__ Lbuf, uint32_t base, uint32_t n, uint32_t lane,
                                              uint32_t last_depth, float t_min, float sky_intensity, uint32_t& n0,
                                              uint32_t& n1, uint32_t& n2, Counts& cnt, FuseOut* fuse = nullptr) {
    static_assert(FUSE < 0 || SWEEP == 0 || SWEEP == 1 || SWEEP == 2 || SWEEP == 4, "forms with a fused shade");
    // shades ray i (records a, b; hit t, prim) when its material kind is FUSE; appends the scattered ray
    auto shade_now = [&](uint32_t kind, uint32_t i, const float4& a, const float4& b, float t, uint32_t prim) {
        bool alive = false;
        float4 na = make_float4(0, 0, 0, 0), nb = na, nc = na;
        const bool scatters = FUSE == FUSE_ANY ? kind < 3u : kind == (uint32_t)(FUSE == FUSE_LAMBERT_MIXED ? RT_MAT_LAMBERT : FUSE);
        if (scatters) {
            const float4 c = C[i];
            if (FUSE == FUSE_ANY) {
                if (kind == RT_MAT_LAMBERT) alive = shade_one<RT_MAT_LAMBERT>(sc, a, b, c, t, prim, fuse->ctr, na, nb, nc);
                else if (kind == RT_MAT_METAL) alive = shade_one<RT_MAT_METAL>(sc, a, b, c, t, prim, fuse->ctr, na, nb, nc);
                else alive = shade_one<RT_MAT_DIELECTRIC>(sc, a, b, c, t, prim, fuse->ctr, na, nb, nc);
            } else {
                alive = shade_one<(FUSE >= 0 && FUSE < 3 ? FUSE : 0)>(sc, a, b, c, t, prim, fuse->ctr, na, nb, nc);
            }
        }
        const uint32_t pos = base + wave_append(alive, fuse->out);
        if (alive) { fuse->nA[pos] = na; fuse->nB[pos] = nb; fuse->nC[pos] = nc; }
        else if (scatters) Lbuf[f2u(a.w)] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);   // scatter() == false
        if (FUSE == FUSE_LAMBERT_MIXED) {      // the other kinds wait in their queues for the shade stage
            uint32_t qpos = wave_append(kind == 1u, n1);
            if (kind == 1u) q1[base + qpos] = i;
            qpos = wave_append(kind == 2u, n2);
            if (kind == 2u) q2[base + qpos] = i;
        }
    };
    const float4* ent = (const float4*)sc.entries;
    const float4* mats = (const float4*)sc.materials;
    const float t_max = __builtin_inff();                  // main.rs:46: t_max = f32::INFINITY
    n0 = n1 = n2 = 0;
    auto enqueue = [&](uint32_t kind, uint32_t i) {
        uint32_t pos;
        pos = wave_append(kind == 0u, n0);
        if (kind == 0u) q0[base + pos] = i;
        pos = wave_append(kind == 1u, n1);
        if (kind == 1u) q1[base + pos] = i;
        pos = wave_append(kind == 2u, n2);
        if (kind == 2u) q2[base + pos] = i;
    };
    if (SWEEP == 1 || SWEEP == 2) {
        // two rays per lane: rays j0 + lane and j0 + 64 + lane
        for (uint32_t j0 = 0; j0 < n; j0 += 128u) {
            const uint32_t ia = base + j0 + lane, ib = ia + 64u;
            const bool va = j0 + lane < n, vb = j0 + 64u + lane < n;
            float4 a0 = make_float4(0.0f, 0.0f, 0.0f, 0.0f), b0 = make_float4(1.0f, 1.0f, 1.0f, 0.0f);
            float4 a1 = a0, b1 = b0;
            if (va) { a0 = A[ia]; b0 = B[ia]; }
            if (vb) { a1 = A[ib]; b1 = B[ib]; }
            Hit2 h;
            closest_hit_sweep2<COUNT, SWEEP == 2, TIE>(ent, sc.n_entries, va, vb, f32x2{a0.x, a1.x}, f32x2{a0.y, a1.y},
                                      f32x2{a0.z, a1.z}, f32x2{b0.x, b1.x}, f32x2{b0.y, b1.y}, f32x2{b0.z, b1.z},
                                      t_min, t_max, h, cnt);
            uint32_t ka = 7u, kb = 7u;
            if (va) ka = finish_ray<fuse_record(FUSE)>(mats, sc.prim_meta, C, hits, Lbuf, ia, f2u(a0.w), b0.x, b0.y, b0.z, h.t[0],
                                                h.prim[0], last_depth, sky_intensity);
            if (vb) kb = finish_ray<fuse_record(FUSE)>(mats, sc.prim_meta, C, hits, Lbuf, ib, f2u(a1.w), b1.x, b1.y, b1.z, h.t[1],
                                                h.prim[1], last_depth, sky_intensity);
            if (FUSE >= 0) {
                shade_now(ka, ia, a0, b0, h.t[0], h.prim[0]);
                shade_now(kb, ib, a1, b1, h.t[1], h.prim[1]);
            } else {
                enqueue(ka, ia);
                enqueue(kb, ib);
            }
        }
    } else if (SWEEP == 5 || SWEEP == 6) {
        walk_segment<COUNT, TIE>(SWEEP == 6 ? (const float4*)lds_stack : (const float4*)sc.walk, sc.n_walk, sc.prim_wait,
                            sc.refill, SplitCfg{nullptr, 0u, 0u}, nullptr, mats, sc.prim_meta, A, B, C, hits, q0, q1, q2, Lbuf, base,
                            n, last_depth, t_min, sky_intensity, n0, n1, n2, cnt);
    } else if (SWEEP == 14 || SWEEP == 15) {   // the refilled walk of a scene with transformed meshes
        walk_segment<COUNT, TIE, false, false, true>(SWEEP == 15 ? (const float4*)lds_stack : (const float4*)sc.walk, sc.n_walk,
                            sc.prim_wait, sc.refill, SplitCfg{nullptr, 0u, 0u}, (const float4*)sc.xforms, mats, sc.prim_meta, A, B, C,
                            hits, q0, q1, q2, Lbuf, base, n, last_depth, t_min, sky_intensity, n0, n1, n2, cnt);
    } else if (SWEEP == 10) {
        walk_segment<COUNT, false, false, true>((const float4*)sc.walk, sc.n_walk, sc.prim_wait, sc.refill,
                                         SplitCfg{(lds_units_ptr)lds_stack, sc.split_top, sc.deep_wait}, nullptr, mats, sc.prim_meta,
                                         A, B, C, hits, q0, q1, q2, Lbuf, base, n, last_depth, t_min, sky_intensity, n0,
                                         n1, n2, cnt);
    } else if (SWEEP == 12 || SWEEP == 16) {
        wide_segment<COUNT>((const uint4*)sc.wide_nodes, sc.wide_stride, (const float4*)sc.wide_prims, sc.prim_wait, sc.refill,
                            lds_stack + (threadIdx.x >> 6) * sc.wide_stack * 64u, lane, mats, sc.prim_meta, A, B, C, hits,
                            q0, q1, q2, Lbuf, base, n, last_depth, t_min, sky_intensity, n0, n1, n2, cnt);
    } else if (SWEEP == 18) {                  // the refilled wide walk of a scene with transformed meshes
        wide_segment<COUNT, true>((const uint4*)sc.wide_nodes, sc.wide_stride, (const float4*)sc.wide_prims, sc.prim_wait, sc.refill,
                                  lds_stack + (threadIdx.x >> 6) * sc.wide_stack * 64u, lane, mats, sc.prim_meta, A, B, C, hits,
                                  q0, q1, q2, Lbuf, base, n, last_depth, t_min, sky_intensity, n0, n1, n2, cnt,
                                  (const float4*)sc.xforms, sc.wide_stack - WIDE_XF_RAY_SLOTS);
    } else if (SWEEP == 8) {
        walk_segment<COUNT, true, true>((const float4*)sc.units_oct, sc.n_units, sc.prim_wait, sc.refill, SplitCfg{nullptr, 0u, 0u},
                                  nullptr, mats, sc.prim_meta, A, B, C, hits, q0, q1, q2, Lbuf, base, n, last_depth, t_min,
                                  sky_intensity, n0, n1, n2, cnt);
    } else {
        // (RT_PREFETCH_RAYS: the next group's ray records are requested before this group is walked, so
        // their latency runs under the walk instead of in front of the next one)
#ifndef RT_PREFETCH_RAYS
#define RT_PREFETCH_RAYS 1
#endif
        const float4 no_a = make_float4(0.0f, 0.0f, 0.0f, 0.0f), no_b = make_float4(1.0f, 1.0f, 1.0f, 0.0f);
        float4 a_next = no_a, b_next = no_b;
        if (RT_PREFETCH_RAYS && lane < n) { a_next = A[base + lane]; b_next = B[base + lane]; }
        for (uint32_t j0 = 0; j0 < n; j0 += 64u) {
            const uint32_t i = base + j0 + lane;
            const bool valid = j0 + lane < n;
            uint32_t kind = 7u;
            float4 a = no_a, b = no_b;
            if (RT_PREFETCH_RAYS) {
                a = a_next; b = b_next;
                a_next = no_a; b_next = no_b;
                if (j0 + 64u + lane < n) { a_next = A[i + 64u]; b_next = B[i + 64u]; }
            } else if (valid) {
                a = A[i]; b = B[i];
            }
            float t = t_max;
            uint32_t prim;
            if (SWEEP == 3)
                closest_hit_ordered<COUNT>((const float4*)sc.units, sc.n_units, sc.prim_wait,
                                           lds_stack + (threadIdx.x >> 6) * sc.stack_depth * 64u, lane, valid, a.x,
                                           a.y, a.z, b.x, b.y, b.z, t_min, t, prim, cnt);
            else if (SWEEP == 4)
                closest_hit<COUNT, false, TIE>((const float4*)lds_stack, sc.n_walk, sc.prim_wait, valid, a.x, a.y, a.z, b.x, b.y, b.z, t_min, t, prim, cnt);
            else if (SWEEP == 7)
                closest_hit<COUNT, true>((const float4*)sc.units_oct, sc.n_units, sc.prim_wait, valid, a.x, a.y, a.z, b.x, b.y, b.z, t_min, t, prim, cnt);
            else if (SWEEP == 9 || SWEEP == 13)    // a scene with transformed meshes, from memory or out of LDS
                closest_hit<COUNT, false, TIE, true>(SWEEP == 13 ? (const float4*)lds_stack : (const float4*)sc.walk, sc.n_walk, sc.prim_wait, valid,
                                                     a.x, a.y, a.z, b.x, b.y, b.z, t_min, t, prim, cnt, (const float4*)sc.xforms);
            else if (SWEEP == 11)
                closest_hit_wide<COUNT>((const uint4*)sc.wide_nodes, sc.wide_stride, (const float4*)sc.wide_prims, sc.prim_wait,
                                        lds_stack + (threadIdx.x >> 6) * sc.wide_stack * 64u, lane, valid, a.x, a.y, a.z,
                                        b.x, b.y, b.z, t_min, t, prim, cnt);
            else if (SWEEP == 17)              // the wide walk of a scene with transformed meshes
                closest_hit_wide<COUNT, true>((const uint4*)sc.wide_nodes, sc.wide_stride, (const float4*)sc.wide_prims, sc.prim_wait,
                                              lds_stack + (threadIdx.x >> 6) * sc.wide_stack * 64u, lane, valid, a.x, a.y, a.z,
                                              b.x, b.y, b.z, t_min, t, prim, cnt, (const float4*)sc.xforms,
                                              sc.wide_stack - WIDE_XF_RAY_SLOTS);
            else
                closest_hit<COUNT, false, TIE>((const float4*)sc.walk, sc.n_walk, sc.prim_wait, valid, a.x, a.y, a.z, b.x, b.y, b.z, t_min, t, prim, cnt);
            if (valid)
                kind = finish_ray<fuse_record(FUSE)>(mats, sc.prim_meta, C, hits, Lbuf, i, f2u(a.w), b.x, b.y, b.z, t, prim, last_depth,
                                              sky_intensity);
            if (FUSE >= 0) shade_now(kind, i, a, b, t, prim);
            else enqueue(kind, i);
        }
    }
}

#ifdef RT_STAMPS
__device__ __forceinline__ void write_stamp(uint32_t stamp_slot, uint32_t seg, unsigned long long t0, uint32_t n) {
    if (stamp_slot < STAMP_SLOTS && seg < STAMP_WAVES) {
        unsigned long long* o = rt_stamps + ((size_t)stamp_slot * STAMP_WAVES + seg) * 4;
        o[0] = t0;
        o[1] = __builtin_amdgcn_s_memrealtime();
        o[2] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));     // HW_REG_HW_ID, all 32 bits
        o[3] = ((unsigned long long)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)) << 32) | n;   // HW_REG_XCC_ID | rays
    }
}
#define RT_STAMP_PARAM , uint32_t stamp_slot
#define RT_STAMP_ARG , stamp_slot
#else
#define RT_STAMP_PARAM
#define RT_STAMP_ARG
#endif

template <bool COUNT, int SWEEP, bool TIE>
__global__ void __launch_bounds__(RT_WAVE) __attribute__((amdgpu_num_sgpr(RT_TRACE_SGPRS), amdgpu_waves_per_eu(8, 8)))
k_trace(DevScene sc, BatchGeom g, const float4* __restrict__ A, const float4* __restrict__ B,
        const float4* __restrict__ C, uint2* __restrict__ hits, uint32_t* __restrict__ q0,
        uint32_t* __restrict__ q1, uint32_t* __restrict__ q2, float4* __restrict__ Lbuf,
        const uint32_t* __restrict__ segcnt, uint32_t* __restrict__ qcnt, Totals* totals,
        uint32_t last_depth, float t_min, float sky_intensity RT_STAMP_PARAM) {
    const uint32_t seg = blockIdx.x * RT_WPB + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
    if (seg >= g.n_seg) return;
#ifdef RT_STAMPS
    const unsigned long long stamp_t0 = __builtin_amdgcn_s_memrealtime();
#endif
    const uint32_t n = segcnt[seg];
    uint32_t n0, n1, n2;
    Counts cnt{0, 0, 0};
    trace_segment<COUNT, SWEEP, TIE>(sc, A, B, C, hits, q0, q1, q2, Lbuf, seg * g.seg_len, n, lane, last_depth, t_min,
                                sky_intensity, n0, n1, n2, cnt);
    if (lane == 0) {
        qcnt[seg] = n0;
        qcnt[g.n_seg + seg] = n1;
        qcnt[2u * g.n_seg + seg] = n2;
    }
#ifdef RT_STAMPS
    if (lane == 0) write_stamp(stamp_slot, seg, stamp_t0, n);
#endif
    if (COUNT) {
        atomicAdd(&totals->box, (unsigned long long)cnt.box);
        atomicAdd(&totals->tri, (unsigned long long)cnt.tri);
        atomicAdd(&totals->sph, (unsigned long long)cnt.sph);
    }
}

// ---------------------------------------------------------------------------------------------
// shade: Material::scatter (material.rs) -- one material kind at a time over the wave's queues;
// survivors are compacted into the wave's segment of the next ray buffer
// ---------------------------------------------------------------------------------------------
template <int KIND>
__device__ __forceinline__ void shade_queue(const DevScene& sc, const float4* __restrict__ A,
                                            const float4* __restrict__ B, const float4* __restrict__ C,
                                            const uint2* __restrict__ hits, const uint32_t* __restrict__ queue,
                                            float4* __restrict__ nA, float4* __restrict__ nB,
                                            float4* __restrict__ nC, float4* __restrict__ Lbuf, uint32_t base,
                                            uint32_t n, uint32_t ctr, uint32_t lane, uint32_t& out) {
    for (uint32_t j0 = 0; j0 < n; j0 += 64u) {
        const bool valid = j0 + lane < n;
        bool alive = false;
        float4 na = make_float4(0, 0, 0, 0), nb = na, nc = na;
        uint32_t slot = 0;
        if (valid) {
            const uint32_t i = queue[base + j0 + lane];
            const float4 a = A[i], b = B[i], c = C[i];
            const uint2 h = hits[i];
            slot = f2u(a.w);
            alive = shade_one<KIND>(sc, a, b, c, u2f(h.x), h.y, ctr, na, nb, nc);
        }
        const uint32_t pos = base + wave_append(alive, out);
        if (alive) { nA[pos] = na; nB[pos] = nb; nC[pos] = nc; }
        else if (valid) Lbuf[slot] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);   // scatter() == false: emitted only
    }
}

// Material::scatter for the segment's three queues in turn; returns the number of surviving rays,
// compacted to the front of the segment in (nA, nB, nC)
__device__ __forceinline__ uint32_t shade_segment(const DevScene& sc, const float4* __restrict__ A,
                                                  const float4* __restrict__ B, const float4* __restrict__ C,
                                                  const uint2* __restrict__ hits, const uint32_t* __restrict__ q0,
                                                  const uint32_t* __restrict__ q1, const uint32_t* __restrict__ q2,
                                                  float4* __restrict__ nA, float4* __restrict__ nB,
                                                  float4* __restrict__ nC, float4* __restrict__ Lbuf, uint32_t base,
                                                  uint32_t n0, uint32_t n1, uint32_t n2, uint32_t depth, uint32_t lane) {
    const uint32_t ctr = depth * 16u;
    uint32_t out = 0;
    if (sc.kinds_present & 1u)
        shade_queue<RT_MAT_LAMBERT>(sc, A, B, C, hits, q0, nA, nB, nC, Lbuf, base, n0, ctr, lane, out);
    if (sc.kinds_present & 2u)
        shade_queue<RT_MAT_METAL>(sc, A, B, C, hits, q1, nA, nB, nC, Lbuf, base, n1, ctr, lane, out);
    if (sc.kinds_present & 4u)
        shade_queue<RT_MAT_DIELECTRIC>(sc, A, B, C, hits, q2, nA, nB, nC, Lbuf, base, n2, ctr, lane, out);
    return out;
}

#ifdef RT_HAS_SWEEP
__global__ void __launch_bounds__(RT_WAVE)
k_shade(DevScene sc, BatchGeom g, const float4* __restrict__ A, const float4* __restrict__ B,
        const float4* __restrict__ C, const uint2* __restrict__ hits, const uint32_t* __restrict__ q0,
        const uint32_t* __restrict__ q1, const uint32_t* __restrict__ q2, float4* __restrict__ nA,
        float4* __restrict__ nB, float4* __restrict__ nC, float4* __restrict__ Lbuf,
        const uint32_t* __restrict__ qcnt, uint32_t* __restrict__ segcnt_next, uint32_t depth) {
    const uint32_t seg = blockIdx.x * RT_WPB + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
    if (seg >= g.n_seg) return;
    const uint32_t out = shade_segment(sc, A, B, C, hits, q0, q1, q2, nA, nB, nC, Lbuf, seg * g.seg_len, qcnt[seg],
                                       qcnt[g.n_seg + seg], qcnt[2u * g.n_seg + seg], depth, lane);
    if (lane == 0) segcnt_next[seg] = out;
}

#endif  // RT_HAS_SWEEP

// ---------------------------------------------------------------------------------------------
// The whole life of a segment's paths in one kernel: raygen, then trace and shade per depth.  A
// segment is touched by its own wave only, so nothing orders waves against each other: every
// wave runs through the depths at its own pace and a wave slot is refilled the moment its wave
// ends, instead of the chip draining at the end of every per-depth launch.  Between the stages
// the wave's own stores must have been performed before it reads them back (other lanes wrote
// what a lane reads): a workgroup-scope release + acquire fence, i.e. a wait for the stores.  The
// CU's L1 is write-through and every access to a segment comes from the one CU its wave runs on,
// so no cache has to be written back or invalidated (an agent-scope fence writes back and
// invalidates L2 on this multi-die GPU: once per wave and stage that cost 2-4x the frame).
// ---------------------------------------------------------------------------------------------
struct PathBuffers {
    float4 *A0, *B0, *C0, *A1, *B1, *C1;
    uint2* hits;
    uint32_t *q0, *q1, *q2;
    float4* Lbuf;
    uint32_t* segcnt;          // [max_bounces * n_seg]: live rays entering each depth (ray totals)
    Totals* totals;            // visit counters (diagnostic builds)
    // Profiled renders only (else null): the wave of segment 0 stores the shader-cycle counter and the
    // constant-rate (100 MHz) counter when it starts and when it ends, four words per launch; the
    // ratio of the differences is the shader clock the chip held while this very kernel ran.
    unsigned long long* clock_stamps;
};

__device__ __forceinline__ void stage_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// SWEEP == 4: a scene of a few tens of KB is walked per lane out of LDS.  Eight waves form a
// workgroup (RT_LDS_BLOCK threads) only to share one copy of the unit array; after the copy they
// never synchronise again.  The per-lane walk spends 60 % of its wave time waiting for entries
// from L2; LDS answers in a fraction of that.
#ifndef RT_LDS_BLOCK
#define RT_LDS_BLOCK 512
#endif
// The refilled wide walk over a tree that is read from HBM (a node per 128-byte line) is built for six
// waves per SIMD (80 registers) instead of eight (64): measured on 10 M triangles +8 % (seven waves
// +6 %, five +3 %), while trees that sit in L2 lose 2-3 % by it and keep eight.
#ifndef RT_WIDE_HBM_WAVES
#define RT_WIDE_HBM_WAVES 6
#endif
// (the forms that keep a copy of the units, or of their top, in LDS: workgroups of eight waves)
constexpr bool lds_form(int sweep) { return sweep == 4 || sweep == 6 || sweep == 10 || sweep == 13 || sweep == 15; }
template <int SWEEP, bool TIE, int FUSE = -1>
__global__ void __launch_bounds__(lds_form(SWEEP) ? RT_LDS_BLOCK : RT_WAVE)
    __attribute__((amdgpu_num_sgpr(RT_TRACE_SGPRS), amdgpu_waves_per_eu(SWEEP == 16 ? RT_WIDE_HBM_WAVES : 8, 8)))
k_path(DevScene sc, BatchGeom g, rt_camera cam, uint32_t seed_lo, uint32_t seed_hi, PathBuffers pb,
       uint32_t max_bounces, float t_min, float sky_intensity RT_STAMP_PARAM) {
    const uint32_t wpb = lds_form(SWEEP) ? RT_LDS_BLOCK / 64 : RT_WPB;
    const uint32_t seg = blockIdx.x * wpb + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
    if (lds_form(SWEEP)) {
        const float4* src = (const float4*)sc.walk;
        float4* dst = (float4*)lds_stack;
        const uint32_t n_copy = SWEEP == 10 ? sc.split_top : sc.n_walk + UNITS_END;   // (split walk: the top only)
        for (uint32_t i = threadIdx.x; i < n_copy; i += RT_LDS_BLOCK) dst[i] = src[i];
        __syncthreads();                       // before any wave of the workgroup may leave
    }
    if (seg >= g.n_seg) return;
#ifdef RT_STAMPS
    const unsigned long long stamp_t0 = __builtin_amdgcn_s_memrealtime();
#endif
    const uint32_t base = seg * g.seg_len;
    unsigned long long clk0 = 0, ref0 = 0;
    if (seg == 0 && pb.clock_stamps) {         // (wave-uniform; one wave of the launch)
        clk0 = __builtin_amdgcn_s_memtime();
        ref0 = __builtin_amdgcn_s_memrealtime();
    }
    uint32_t n = raygen_segment(cam, g, seed_lo, seed_hi, pb.A0, pb.B0, pb.C0, base, lane);
    Counts cnt{0, 0, 0};
    for (uint32_t d = 0; d < max_bounces; d++) {
        if (lane == 0) pb.segcnt[d * g.n_seg + seg] = n;
        if (n == 0) continue;                              // the remaining depths only record zero rays
        const bool odd = (d & 1u) != 0;
        float4* A = odd ? pb.A1 : pb.A0; float4* B = odd ? pb.B1 : pb.B0; float4* C = odd ? pb.C1 : pb.C0;
        float4* nA = odd ? pb.A0 : pb.A1; float4* nB = odd ? pb.B0 : pb.B1; float4* nC = odd ? pb.C0 : pb.C1;
        const bool last = d + 1u == max_bounces;
        stage_fence();
        uint32_t n0, n1, n2;
        if (FUSE >= 0) {                                   // closest hit and shade in one pass over the segment
            FuseOut fuse{nA, nB, nC, d * 16u, 0u};
            trace_segment<false, SWEEP, TIE, FUSE>(sc, A, B, C, pb.hits, pb.q0, pb.q1, pb.q2, pb.Lbuf, base, n, lane, last ? 1u : 0u,
                                                   t_min, sky_intensity, n0, n1, n2, cnt, &fuse);
            n = fuse.out;
            if (FUSE == FUSE_LAMBERT_MIXED && !last && (n1 | n2) != 0u) {
                stage_fence();
                const uint32_t ctr = d * 16u;
                if (sc.kinds_present & 2u)
                    shade_queue<RT_MAT_METAL>(sc, A, B, C, pb.hits, pb.q1, nA, nB, nC, pb.Lbuf, base, n1, ctr, lane, n);
                if (sc.kinds_present & 4u)
                    shade_queue<RT_MAT_DIELECTRIC>(sc, A, B, C, pb.hits, pb.q2, nA, nB, nC, pb.Lbuf, base, n2, ctr, lane, n);
            }
            continue;
        }
#ifdef RT_UTIL
        trace_segment<true, SWEEP, TIE>(sc, A, B, C, pb.hits, pb.q0, pb.q1, pb.q2, pb.Lbuf, base, n, lane, last ? 1u : 0u,
                                   t_min, sky_intensity, n0, n1, n2, cnt);
#else
        trace_segment<false, SWEEP, TIE>(sc, A, B, C, pb.hits, pb.q0, pb.q1, pb.q2, pb.Lbuf, base, n, lane, last ? 1u : 0u,
                                    t_min, sky_intensity, n0, n1, n2, cnt);
#endif
        if (last) break;
        stage_fence();
        n = shade_segment(sc, A, B, C, pb.hits, pb.q0, pb.q1, pb.q2, nA, nB, nC, pb.Lbuf, base, n0, n1, n2, d, lane);
    }
#ifdef RT_STAMPS
    if (lane == 0) write_stamp(stamp_slot, seg, stamp_t0, n);
#endif
    if (seg == 0 && pb.clock_stamps) {
        const unsigned long long clk1 = __builtin_amdgcn_s_memtime(), ref1 = __builtin_amdgcn_s_memrealtime();
        if (lane == 0) {
            pb.clock_stamps[0] = clk0; pb.clock_stamps[1] = ref0;
            pb.clock_stamps[2] = clk1; pb.clock_stamps[3] = ref1;
        }
    }
#ifdef RT_UTIL
    atomicAdd(&pb.totals->box, (unsigned long long)cnt.box);
    atomicAdd(&pb.totals->tri, (unsigned long long)cnt.tri);
    atomicAdd(&pb.totals->sph, (unsigned long long)cnt.sph);
#endif
}

#ifdef RT_HAS_SWEEP
// ---------------------------------------------------------------------------------------------
// accumulate: per pixel, add the batch's samples in sample order (one owner per pixel: no
// atomics, fixed summation order).
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(RT_BLOCK)
k_accumulate(BatchGeom g, const float4* __restrict__ Lbuf, float* __restrict__ accum) {
    for (uint32_t q = blockIdx.x * RT_BLOCK + threadIdx.x; q < g.n_pix_slots; q += gridDim.x * RT_BLOCK) {
        uint32_t x, y;
        if (!slot_pixel(g, q, x, y)) continue;
        float r = 0.0f, gg = 0.0f, b = 0.0f;
        for (uint32_t k = 0; k < g.n_samples; k++) {
            const float4 l = Lbuf[(size_t)k * g.n_pix_slots + q];
            r += l.x; gg += l.y; b += l.z;
        }
        float* dst = accum + ((size_t)y * g.width + x) * 3;
        dst[0] += r; dst[1] += gg; dst[2] += b;
    }
}

// rays traced by the batch = sum of the live-ray counts of every segment at every depth
__global__ void __launch_bounds__(RT_BLOCK)
k_count_rays(const uint32_t* __restrict__ segcnt, uint32_t n, Totals* totals) {
    unsigned long long s = 0;
    for (uint32_t i = blockIdx.x * RT_BLOCK + threadIdx.x; i < n; i += gridDim.x * RT_BLOCK) s += segcnt[i];
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off);
    if ((threadIdx.x & 63u) == 0 && s != 0) atomicAdd(&totals->rays, s);   // integer: order-independent
}

// main.rs:369-373 + rgb_to_u32 (math.rs:112-116)
__device__ __forceinline__ uint32_t sat_u32(float f) {     // Rust `as u32`
    if (!(f == f) || f <= 0.0f) return 0u;
    if (f >= 4294967296.0f) return 0xFFFFFFFFu;
    return (uint32_t)f;
}
__global__ void __launch_bounds__(RT_BLOCK)
k_tonemap(const float* __restrict__ accum, float scale, uint32_t* __restrict__ out, size_t npix) {
    for (size_t p = (size_t)blockIdx.x * RT_BLOCK + threadIdx.x; p < npix; p += (size_t)gridDim.x * RT_BLOCK) {
        float c[3];
        for (int k = 0; k < 3; k++) {
            const float s = sqrtf(accum[p * 3 + k] * scale);
            c[k] = s < 1.0f ? s : 1.0f;                    // Vec3::min(sqrt, one): minps
        }
        out[p] = (sat_u32(255.999f * c[0]) << 16) | (sat_u32(255.999f * c[1]) << 8) | sat_u32(255.999f * c[2]);
    }
}

#endif  // RT_HAS_SWEEP

// ---------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------
static inline uint32_t grid_for(size_t n, uint32_t cap) {
    size_t b = (n + RT_BLOCK - 1) / RT_BLOCK;
    if (b < 1) b = 1;
    if (b > cap) b = cap;
    return (uint32_t)b;
}
static inline int check_launch() { return hipGetLastError() == hipSuccess ? 0 : -1; }

// Which unit holds the kernel a scene is traced with: the wave sweep lives in the default unit,
// every per-lane walk in the unit built without SLP vectorisation.
static inline bool uses_sweep_kernels(const DevScene& sc) { return sc.sweep && !sc.ordered && !sc.octant && !sc.wide && !sc.n_xforms; }
int launch_trace_api_walk(const LaunchCfg&, const DevScene&, const void* d_rays, size_t n, float t_min, float t_max,
                          void* d_hits, Totals* d_counts, bool count, uint32_t grid);
int launch_trace_walk(const LaunchCfg&, const DevScene&, const BatchGeom&, const RenderBuffers&, uint32_t depth,
                      bool last_depth, float t_min, float sky_intensity, bool count);
int launch_path_walk(const LaunchCfg&, const DevScene&, const rt_camera&, const BatchGeom&, uint64_t seed,
                     const RenderBuffers&, uint32_t max_bounces, float t_min, float sky_intensity);

// (the _T forms take the tie rule as an argument; the plain ones pick it by the scene's flag: the forms
// that exist in the reference's order and out of it -- walks and sweeps over either tree)
#define RT_TRACE_API_T(CNT, SWP, TIE)                                                                   \
    hipLaunchKernelGGL((k_trace_api<CNT, SWP, TIE>), dim3(grid), dim3(RT_BLOCK), lds, st, sc, (const float*)d_rays, n, \
                       t_min, t_max, (rt_hit*)d_hits, d_counts)
#define RT_TRACE_API(CNT, SWP) do { if (sc.tie) RT_TRACE_API_T(CNT, SWP, true); else RT_TRACE_API_T(CNT, SWP, false); } while (0)
#define RT_TRACE_T(CNT, SWP, TIE)                                                                       \
    hipLaunchKernelGGL((k_trace<CNT, SWP, TIE>), dim3((g.n_seg + RT_WPB - 1) / RT_WPB), dim3(RT_WAVE), lds, st, sc, g, (const float4*)rb.cur.A, \
                       (const float4*)rb.cur.B, (const float4*)rb.cur.C, (uint2*)rb.hits, rb.queue[0],  \
                       rb.queue[1], rb.queue[2], (float4*)rb.Lbuf, segcnt, rb.qcnt, rb.totals,          \
                       last_depth ? 1u : 0u, t_min, sky_intensity RT_STAMP_ARG)
#define RT_TRACE(CNT, SWP) do { if (sc.tie) RT_TRACE_T(CNT, SWP, true); else RT_TRACE_T(CNT, SWP, false); } while (0)
#define RT_PATH_T(SWP, TIE)                                                                             \
    hipLaunchKernelGGL((k_path<SWP, TIE>), dim3((g.n_seg + RT_WPB - 1) / RT_WPB), dim3(RT_WAVE), lds, st, sc, g, cam, \
                       (uint32_t)seed, (uint32_t)(seed >> 32), pb, max_bounces, t_min, sky_intensity RT_STAMP_ARG)
#define RT_PATH(SWP) do { if (sc.tie) RT_PATH_T(SWP, true); else RT_PATH_T(SWP, false); } while (0)
// (scenes whose only scattering kind is Lambertian: shaded right after the closest hit)
#define RT_PATH_F(SWP, TIE, FUSE)                                                                       \
    hipLaunchKernelGGL((k_path<SWP, TIE, FUSE>), dim3((g.n_seg + RT_WPB - 1) / RT_WPB), dim3(RT_WAVE), lds, st, sc, g, cam, \
                       (uint32_t)seed, (uint32_t)(seed >> 32), pb, max_bounces, t_min, sky_intensity RT_STAMP_ARG)
#define RT_PATH_FUSED(SWP) do { if (sc.fuse == 2u) { if (sc.tie) RT_PATH_F(SWP, true, FUSE_ANY); else RT_PATH_F(SWP, false, FUSE_ANY); } \
                                else if (sc.fuse == 3u) { if (sc.tie) RT_PATH_F(SWP, true, FUSE_LAMBERT_MIXED); else RT_PATH_F(SWP, false, FUSE_LAMBERT_MIXED); } \
                                else if (sc.tie) RT_PATH_F(SWP, true, RT_MAT_LAMBERT); else RT_PATH_F(SWP, false, RT_MAT_LAMBERT); } while (0)
#define RT_PATH_LDS_F(SWP, TIE, FUSE)                                                                   \
    hipLaunchKernelGGL((k_path<SWP, TIE, FUSE>), dim3((g.n_seg + RT_LDS_BLOCK / 64 - 1) / (RT_LDS_BLOCK / 64)), dim3(RT_LDS_BLOCK), \
                       bytes, st, sc, g, cam, (uint32_t)seed, (uint32_t)(seed >> 32), pb, max_bounces, t_min, \
                       sky_intensity RT_STAMP_ARG)
#define RT_PATH_LDS_FUSED(SWP) do { if (sc.fuse == 2u) { if (sc.tie) RT_PATH_LDS_F(SWP, true, FUSE_ANY); else RT_PATH_LDS_F(SWP, false, FUSE_ANY); } \
                                    else if (sc.fuse == 3u) { if (sc.tie) RT_PATH_LDS_F(SWP, true, FUSE_LAMBERT_MIXED); else RT_PATH_LDS_F(SWP, false, FUSE_LAMBERT_MIXED); } \
                                    else if (sc.tie) RT_PATH_LDS_F(SWP, true, RT_MAT_LAMBERT); else RT_PATH_LDS_F(SWP, false, RT_MAT_LAMBERT); } while (0)
#define RT_PATH_LDS_T(SWP, TIE)                                                                         \
    hipLaunchKernelGGL((k_path<SWP, TIE>), dim3((g.n_seg + RT_LDS_BLOCK / 64 - 1) / (RT_LDS_BLOCK / 64)), dim3(RT_LDS_BLOCK), \
                       bytes, st, sc, g, cam, (uint32_t)seed, (uint32_t)(seed >> 32), pb, max_bounces, t_min, \
                       sky_intensity RT_STAMP_ARG)
#define RT_PATH_LDS(SWP) do { if (sc.tie) RT_PATH_LDS_T(SWP, true); else RT_PATH_LDS_T(SWP, false); } while (0)
#ifdef RT_STAMPS
static uint32_t stamped_launches = 0;
#endif

static inline PathBuffers path_buffers(const RenderBuffers& rb) {
    PathBuffers pb;
    pb.A0 = (float4*)rb.cur.A; pb.B0 = (float4*)rb.cur.B; pb.C0 = (float4*)rb.cur.C;
    pb.A1 = (float4*)rb.next.A; pb.B1 = (float4*)rb.next.B; pb.C1 = (float4*)rb.next.C;
    pb.hits = (uint2*)rb.hits;
    pb.q0 = rb.queue[0]; pb.q1 = rb.queue[1]; pb.q2 = rb.queue[2];
    pb.Lbuf = (float4*)rb.Lbuf;
    pb.segcnt = rb.segcnt;
    pb.totals = rb.totals;
    pb.clock_stamps = rb.clock_stamps;
    return pb;
}

#ifdef RT_HAS_WALK
// LDS of the walks that keep a per-lane stack: one word per lane and slot
static inline size_t stack_bytes(const DevScene& sc, uint32_t waves) {
    return (size_t)waves * (sc.wide ? sc.wide_stack : sc.ordered ? sc.stack_depth : 0u) * 64u * sizeof(uint32_t);
}
int launch_trace_api_walk(const LaunchCfg& cfg, const DevScene& sc, const void* d_rays, size_t n, float t_min,
                          float t_max, void* d_hits, Totals* d_counts, bool count, uint32_t grid) {
    hipStream_t st = (hipStream_t)cfg.stream;
    const size_t lds = stack_bytes(sc, RT_BLOCK / 64);
    if (sc.wide && sc.n_xforms) { if (count) RT_TRACE_API_T(true, 17, true); else RT_TRACE_API_T(false, 17, true); }
    else if (sc.wide) { if (count) RT_TRACE_API_T(true, 11, true); else RT_TRACE_API_T(false, 11, true); }
    else if (sc.n_xforms) { if (count) RT_TRACE_API(true, 9); else RT_TRACE_API(false, 9); }
    else if (sc.octant) { if (count) RT_TRACE_API_T(true, 7, true); else RT_TRACE_API_T(false, 7, true); }
    else if (sc.ordered) { if (count) RT_TRACE_API_T(true, 3, true); else RT_TRACE_API_T(false, 3, true); }
    else { if (count) RT_TRACE_API(true, 0); else RT_TRACE_API(false, 0); }
    return check_launch();
}

int launch_trace_walk(const LaunchCfg& cfg, const DevScene& sc, const BatchGeom& g, const RenderBuffers& rb,
                      uint32_t depth, bool last_depth, float t_min, float sky_intensity, bool count) {
    hipStream_t st = (hipStream_t)cfg.stream;
    const uint32_t* segcnt = rb.segcnt + (size_t)depth * g.n_seg;
    const size_t lds = stack_bytes(sc, RT_WPB);
#ifdef RT_STAMPS
    const uint32_t stamp_slot = (!count && stamped_launches < STAMP_SLOTS) ? stamped_launches++ : 0xFFFFFFFFu;
#endif
    if (sc.wide && sc.n_xforms) { if (count) RT_TRACE_T(true, 17, true); else RT_TRACE_T(false, 17, true); }
    else if (sc.wide) { if (count) RT_TRACE_T(true, 11, true); else RT_TRACE_T(false, 11, true); }
    else if (sc.n_xforms) { if (count) RT_TRACE(true, 9); else RT_TRACE(false, 9); }
    else if (sc.octant) { if (count) RT_TRACE_T(true, 7, true); else RT_TRACE_T(false, 7, true); }
    else if (sc.ordered) { if (count) RT_TRACE_T(true, 3, true); else RT_TRACE_T(false, 3, true); }
    else { if (count) RT_TRACE(true, 0); else RT_TRACE(false, 0); }
    return check_launch();
}

int launch_path_walk(const LaunchCfg& cfg, const DevScene& sc, const rt_camera& cam, const BatchGeom& g, uint64_t seed,
                     const RenderBuffers& rb, uint32_t max_bounces, float t_min, float sky_intensity) {
    hipStream_t st = (hipStream_t)cfg.stream;
    const PathBuffers pb = path_buffers(rb);
    const size_t lds = stack_bytes(sc, RT_WPB);
#ifdef RT_STAMPS
    const uint32_t stamp_slot = stamped_launches < STAMP_SLOTS ? stamped_launches++ : 0xFFFFFFFFu;
#endif
    if (sc.wide && sc.n_xforms) {
        if (sc.refill) RT_PATH_T(18, true); else RT_PATH_T(17, true);
        return check_launch();
    }
    if (sc.wide) {
        if (!sc.refill) RT_PATH_T(11, true);
        else if (sc.wide_stride == WIDE_LINE_WORDS / 4u) RT_PATH_T(16, true);   // (the stride counts uint4)
        else RT_PATH_T(12, true);
        return check_launch();
    }
    if (sc.n_xforms) {                         // the per-lane walks that follow the transform markers
        if (sc.lds_walk) {
            const size_t bytes = (size_t)(sc.n_walk + UNITS_END) * 16u;
            if (sc.refill) RT_PATH_LDS(15); else RT_PATH_LDS(13);
        } else if (sc.refill) {
            RT_PATH(14);
        } else {
            RT_PATH(9);
        }
        return check_launch();
    }
    if (sc.lds_walk && !sc.ordered && !sc.octant) {
        const size_t bytes = (size_t)(sc.n_walk + UNITS_END) * 16u;
        if (sc.refill) RT_PATH_LDS(6);
        else if (sc.fuse) RT_PATH_LDS_FUSED(4);
        else RT_PATH_LDS(4);
        return check_launch();
    }
    if (sc.split_top && sc.refill && !sc.ordered && !sc.octant && !sc.tie) {
        const size_t bytes = (size_t)sc.split_top * 16u;
        RT_PATH_LDS_T(10, false);
        return check_launch();
    }
    if (sc.octant) { if (sc.refill) RT_PATH_T(8, true); else RT_PATH_T(7, true); }
    else if (sc.ordered) RT_PATH_T(3, true);
    else if (sc.refill) RT_PATH(5);
    else if (sc.fuse) RT_PATH_FUSED(0);
    else RT_PATH(0);
    return check_launch();
}
#endif  // RT_HAS_WALK

#ifdef RT_HAS_SWEEP
uint32_t trace_api_grid(const LaunchCfg& cfg, size_t n) { return grid_for(n, cfg.grid_cap); }

int launch_trace_api(const LaunchCfg& cfg, const DevScene& sc, const void* d_rays, size_t n, float t_min,
                     float t_max, void* d_hits, Totals* d_counts, bool count, uint32_t grid, void* d_out3) {
    hipStream_t st = (hipStream_t)cfg.stream;
    if (n == 0) grid = 0;
    if (n != 0) {
        if (!uses_sweep_kernels(sc)) {
            if (launch_trace_api_walk(cfg, sc, d_rays, n, t_min, t_max, d_hits, d_counts, count, grid)) return -1;
        } else {
            const size_t lds = 0;
            if (sc.has_spheres) { if (count) RT_TRACE_API(true, 2); else RT_TRACE_API(false, 2); }
            else { if (count) RT_TRACE_API(true, 1); else RT_TRACE_API(false, 1); }
            if (check_launch()) return -1;
        }
    }
    if (count)
        hipLaunchKernelGGL(k_sum_partials, dim3(1), dim3(RT_BLOCK), 0, st, (const Totals*)d_counts, grid,
                           (unsigned long long*)d_out3);
    return check_launch();
}

// ---------------------------------------------------------------------------------------------
// Scratch of rt_trace_device_static(): one 32-byte record of partial visit counts per workgroup of
// the closest-hit kernel, in a device array the library owns for the life of the process (every
// GPU holds its own copy at its own address).  In host code `scratch` is only a handle for
// hipGetSymbolAddress(); the kernels are handed the address that call returns.
// ---------------------------------------------------------------------------------------------
static __device__ __align__(256) unsigned long long scratch[(size_t)TRACE_STATIC_MAX_BLOCKS * 4];

int static_scratch_address(int device, void** out) {
    constexpr int MAX_DEVICES = 64;
    static void* known[MAX_DEVICES] = {};
    static std::mutex lock;
    if (device < 0 || device >= MAX_DEVICES || !out) return -1;
    std::lock_guard<std::mutex> guard(lock);
    if (!known[device]) {                      // first context on this device (which is current)
        void* p = nullptr;
        if (hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)) != hipSuccess || !p) return -1;
        known[device] = p;
    }
    *out = known[device];
    return 0;
}

int launch_raygen(const LaunchCfg& cfg, const rt_camera& cam, const BatchGeom& g, uint64_t seed,
                  const RenderBuffers& rb) {
    hipStream_t st = (hipStream_t)cfg.stream;
    hipLaunchKernelGGL(k_raygen, dim3((g.n_seg + RT_WPB - 1) / RT_WPB), dim3(RT_WAVE), 0, st, cam, g, (uint32_t)seed,
                       (uint32_t)(seed >> 32), (float4*)rb.cur.A, (float4*)rb.cur.B, (float4*)rb.cur.C, rb.segcnt);
    return check_launch();
}

int launch_trace(const LaunchCfg& cfg, const DevScene& sc, const BatchGeom& g, const RenderBuffers& rb,
                 uint32_t depth, bool last_depth, float t_min, float sky_intensity, bool count) {
    if (!uses_sweep_kernels(sc)) return launch_trace_walk(cfg, sc, g, rb, depth, last_depth, t_min, sky_intensity, count);
    hipStream_t st = (hipStream_t)cfg.stream;
    const uint32_t* segcnt = rb.segcnt + (size_t)depth * g.n_seg;
    const size_t lds = 0;
#ifdef RT_STAMPS
    const uint32_t stamp_slot = (!count && stamped_launches < STAMP_SLOTS) ? stamped_launches++ : 0xFFFFFFFFu;
#endif
    if (sc.has_spheres) { if (count) RT_TRACE(true, 2); else RT_TRACE(false, 2); }
    else { if (count) RT_TRACE(true, 1); else RT_TRACE(false, 1); }
    return check_launch();
}

int launch_path(const LaunchCfg& cfg, const DevScene& sc, const rt_camera& cam, const BatchGeom& g, uint64_t seed,
                const RenderBuffers& rb, uint32_t max_bounces, float t_min, float sky_intensity) {
    if (!uses_sweep_kernels(sc)) return launch_path_walk(cfg, sc, cam, g, seed, rb, max_bounces, t_min, sky_intensity);
    hipStream_t st = (hipStream_t)cfg.stream;
    const PathBuffers pb = path_buffers(rb);
    const size_t lds = 0;
#ifdef RT_STAMPS
    const uint32_t stamp_slot = stamped_launches < STAMP_SLOTS ? stamped_launches++ : 0xFFFFFFFFu;
#endif
    if (sc.has_spheres) { if (sc.fuse) RT_PATH_FUSED(2); else RT_PATH(2); }
    else if (sc.fuse) RT_PATH_FUSED(1);
    else RT_PATH(1);
    return check_launch();
}

#ifdef RT_STAMPS
void dump_stamps() {
    const char* path = getenv("RT_STAMPS_FILE");
    if (!path) return;
    std::vector<unsigned long long> h((size_t)STAMP_SLOTS * STAMP_WAVES * 4);
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(rt_stamps), h.size() * 8);
    if (FILE* f = fopen(path, "wb")) { fwrite(h.data(), 8, h.size(), f); fclose(f); }
}
#endif

int launch_shade(const LaunchCfg& cfg, const DevScene& sc, const BatchGeom& g, const RenderBuffers& rb,
                 uint32_t depth) {
    hipStream_t st = (hipStream_t)cfg.stream;
    hipLaunchKernelGGL(k_shade, dim3((g.n_seg + RT_WPB - 1) / RT_WPB), dim3(RT_WAVE), 0, st, sc, g, (const float4*)rb.cur.A,
                       (const float4*)rb.cur.B, (const float4*)rb.cur.C, (const uint2*)rb.hits,
                       (const uint32_t*)rb.queue[0], (const uint32_t*)rb.queue[1], (const uint32_t*)rb.queue[2],
                       (float4*)rb.next.A, (float4*)rb.next.B, (float4*)rb.next.C, (float4*)rb.Lbuf,
                       (const uint32_t*)rb.qcnt, rb.segcnt + (size_t)(depth + 1) * g.n_seg, depth);
    return check_launch();
}

int launch_accumulate(const LaunchCfg& cfg, const BatchGeom& g, const RenderBuffers& rb, void* d_accum,
                      uint32_t max_bounces) {
    hipStream_t st = (hipStream_t)cfg.stream;
    hipLaunchKernelGGL(k_accumulate, dim3(grid_for(g.n_pix_slots, cfg.grid_cap)), dim3(RT_BLOCK), 0, st, g,
                       (const float4*)rb.Lbuf, (float*)d_accum);
    hipLaunchKernelGGL(k_count_rays, dim3(grid_for((size_t)max_bounces * g.n_seg, 256)), dim3(RT_BLOCK), 0, st,
                       (const uint32_t*)rb.segcnt, max_bounces * g.n_seg, rb.totals);
    return check_launch();
}

// dst[i] += src[i]: sums the per-worker frames when several workers share one GPU (where RCCL,
// which refuses two ranks on a device, cannot be used) and adds a reduced frame to the caller's
__global__ void __launch_bounds__(RT_BLOCK)
k_add_frame(float* __restrict__ dst, const float* __restrict__ src, size_t n) {
    for (size_t i = (size_t)blockIdx.x * RT_BLOCK + threadIdx.x; i < n; i += (size_t)gridDim.x * RT_BLOCK)
        dst[i] += src[i];
}
int launch_add_frame(const LaunchCfg& cfg, void* d_dst, const void* d_src, size_t n) {
    if (n == 0) return 0;
    hipLaunchKernelGGL(k_add_frame, dim3(grid_for(n, cfg.grid_cap)), dim3(RT_BLOCK), 0, (hipStream_t)cfg.stream,
                       (float*)d_dst, (const float*)d_src, n);
    return check_launch();
}

int launch_tonemap(const LaunchCfg& cfg, const void* d_accum, uint32_t spp, void* d_out, size_t npix) {
    if (npix == 0) return 0;
    hipStream_t st = (hipStream_t)cfg.stream;
    hipLaunchKernelGGL(k_tonemap, dim3(grid_for(npix, cfg.grid_cap)), dim3(RT_BLOCK), 0, st,
                       (const float*)d_accum, 1.0f / (float)spp, (uint32_t*)d_out, npix);
    return check_launch();
}
#endif  // RT_HAS_SWEEP

}  // namespace rtamd

#endif  // this unit holds kernels
