// Host-side scene builder: reproduces the reference's tree TOPOLOGY and box VALUES
// (Bvh::new shape.rs:94-138, Mesh::build_bvh shape.rs:264-290, compute_aabb shape.rs:218-233)
// and writes them straight into the pre-order, skip-linked entry array the traversal kernel
// walks.  Keeping topology and boxes identical is what makes per-ray hits bit-exact: a better
// tree would test different boxes and turn some of the reference's rounding-induced misses
// into hits.
//
// Built with -ffp-contract=off: box and edge values must be the plain IEEE f32 results.

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <limits>
#include <numeric>
#include <thread>

#include "rt_internal.hpp"

namespace rtamd {
namespace {

struct Box { float lo[3], hi[3]; };

// glam's Vec3::min / max are minps / maxps: the second operand when the compare is false
inline float minps(float a, float b) { return a < b ? a : b; }
inline float maxps(float a, float b) { return a > b ? a : b; }

inline Box box_infinity() {                    // math.rs:141-146
    const float inf = std::numeric_limits<float>::infinity();
    return Box{{inf, inf, inf}, {-inf, -inf, -inf}};
}
inline void grow_point(Box& b, const float p[3]) {   // math.rs:174-179
    for (int k = 0; k < 3; k++) { b.lo[k] = minps(b.lo[k], p[k]); b.hi[k] = maxps(b.hi[k], p[k]); }
}
inline Box surrounding(const Box& a, const Box& b) { // math.rs:167-172 (self = a)
    Box r;
    for (int k = 0; k < 3; k++) { r.lo[k] = minps(a.lo[k], b.lo[k]); r.hi[k] = maxps(a.hi[k], b.hi[k]); }
    return r;
}
inline float u2f(uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; }

// glam 0.8.7 Mat4 (column-major; m[4 * column + row]) as Mesh uses it (shape.rs:241-251).
// transform_point3 / transform_vector3 = mul_vec4 of (v, 1) / (v, 0), truncated, and mul_vec4 =
// ((x_axis * x + y_axis * y) + z_axis * z) + w_axis * w in rounded f32 steps; inverse() is the
// cofactor expansion the crate ports from GLM.  (Restated from the crate's published source; the
// reference holds nothing that could confirm the operation order.)
inline void mat4_mul_vec(const float* m, const float v[3], float w, float out[3]) {
    for (int k = 0; k < 3; k++) {
        float acc = m[k] * v[0];
        acc = m[4 + k] * v[1] + acc;
        acc = m[8 + k] * v[2] + acc;
        acc = m[12 + k] * w + acc;
        out[k] = acc;
    }
}
void mat4_inverse(const float* m, float* out) {
    const float m00 = m[0], m01 = m[1], m02 = m[2], m03 = m[3];
    const float m10 = m[4], m11 = m[5], m12 = m[6], m13 = m[7];
    const float m20 = m[8], m21 = m[9], m22 = m[10], m23 = m[11];
    const float m30 = m[12], m31 = m[13], m32 = m[14], m33 = m[15];
    const float coef00 = m22 * m33 - m32 * m23, coef02 = m12 * m33 - m32 * m13, coef03 = m12 * m23 - m22 * m13;
    const float coef04 = m21 * m33 - m31 * m23, coef06 = m11 * m33 - m31 * m13, coef07 = m11 * m23 - m21 * m13;
    const float coef08 = m21 * m32 - m31 * m22, coef10 = m11 * m32 - m31 * m12, coef11 = m11 * m22 - m21 * m12;
    const float coef12 = m20 * m33 - m30 * m23, coef14 = m10 * m33 - m30 * m13, coef15 = m10 * m23 - m20 * m13;
    const float coef16 = m20 * m32 - m30 * m22, coef18 = m10 * m32 - m30 * m12, coef19 = m10 * m22 - m20 * m12;
    const float coef20 = m20 * m31 - m30 * m21, coef22 = m10 * m31 - m30 * m11, coef23 = m10 * m21 - m20 * m11;
    const float fac0[4] = {coef00, coef00, coef02, coef03}, fac1[4] = {coef04, coef04, coef06, coef07};
    const float fac2[4] = {coef08, coef08, coef10, coef11}, fac3[4] = {coef12, coef12, coef14, coef15};
    const float fac4[4] = {coef16, coef16, coef18, coef19}, fac5[4] = {coef20, coef20, coef22, coef23};
    const float vec0[4] = {m10, m00, m00, m00}, vec1[4] = {m11, m01, m01, m01};
    const float vec2[4] = {m12, m02, m02, m02}, vec3[4] = {m13, m03, m03, m03};
    const float sign_a[4] = {1.0f, -1.0f, 1.0f, -1.0f}, sign_b[4] = {-1.0f, 1.0f, -1.0f, 1.0f};
    float inv[4][4];
    for (int k = 0; k < 4; k++) {
        inv[0][k] = ((vec1[k] * fac0[k] - vec2[k] * fac1[k]) + vec3[k] * fac2[k]) * sign_a[k];
        inv[1][k] = ((vec0[k] * fac0[k] - vec2[k] * fac3[k]) + vec3[k] * fac4[k]) * sign_b[k];
        inv[2][k] = ((vec0[k] * fac1[k] - vec1[k] * fac3[k]) + vec3[k] * fac5[k]) * sign_a[k];
        inv[3][k] = ((vec0[k] * fac2[k] - vec1[k] * fac4[k]) + vec2[k] * fac5[k]) * sign_b[k];
    }
    const float d0 = m00 * inv[0][0], d1 = m01 * inv[1][0], d2 = m02 * inv[2][0], d3 = m03 * inv[3][0];
    const float det = (d0 + d1) + (d2 + d3);
    const float rcp = 1.0f / det;
    for (int c = 0; c < 4; c++)
        for (int k = 0; k < 4; k++) out[4 * c + k] = inv[c][k] * rcp;
}

struct Builder {
    const std::vector<rt_material>& mats;
    std::vector<SceneObject>& objects;
    FlatScene& out;
    std::vector<Box> obj_box;                  // Hittable::get_aabb() per object
    const std::vector<SceneGroup>* groups = nullptr;
    std::vector<Box> group_box;                // get_aabb() of every container
    std::vector<uint32_t> xform_of;            // per object: index of its transform + 1, 0 for none
    // scratch for the mesh currently being emitted
    const SceneObject* mesh = nullptr;
    std::vector<float> tri_min;                // Triangle::get_aabb().min per triangle (3 floats)
    std::vector<float> tri_cen;                // SAH build: 2 x centre of every triangle's box
    bool sah = false;                          // split meshes by the surface-area heuristic

    Builder(const std::vector<rt_material>& m, std::vector<SceneObject>& o, FlatScene& f)
        : mats(m), objects(o), out(f) {}

    uint32_t matkind(uint32_t mat) const { return mat | (mats[mat].kind << KIND_SHIFT); }

    size_t push_entry() {
        size_t i = out.n_entries();
        out.entries.resize(out.entries.size() + ENTRY_FLOATS, 0.0f);
        return i;
    }
    float* entry(size_t i) { return &out.entries[i * ENTRY_FLOATS]; }

    size_t push_box(const Box& b) {
        size_t i = push_entry();
        float* e = entry(i);
        e[0] = b.lo[0]; e[1] = b.lo[1]; e[2] = b.lo[2];
        e[4] = b.hi[0]; e[5] = b.hi[1]; e[6] = b.hi[2];
        return i;
    }
    void close_box(size_t i) {                 // the skip link: first entry after the subtree
        entry(i)[3] = u2f((TAG_BOX << TAG_SHIFT) | (uint32_t)out.n_entries());
    }

    void push_sphere(const SceneObject& o) {
        float* e = entry(push_entry());
        e[0] = o.sphere.c[0]; e[1] = o.sphere.c[1]; e[2] = o.sphere.c[2];
        e[3] = u2f(TAG_SPH << TAG_SHIFT);
        e[4] = o.sphere.r;
        e[7] = u2f(o.prim_base);
    }
    void write_triangle(size_t at, uint32_t local) {
        const rt_triangle& t = mesh->tris[local];
        float* e = entry(at);
        for (int k = 0; k < 3; k++) {
            e[k] = t.p[0][k];
            e[4 + k] = t.p[1][k] - t.p[0][k];  // v0v1, shape.rs:417
            e[8 + k] = t.p[2][k] - t.p[0][k];  // v0v2, shape.rs:418
        }
        e[3] = u2f(TAG_TRI << TAG_SHIFT);
        e[7] = u2f(mesh->prim_base + local);
    }
    void write_box(size_t at, const Box& b, size_t skip) {
        float* e = entry(at);
        e[0] = b.lo[0]; e[1] = b.lo[1]; e[2] = b.lo[2];
        e[4] = b.hi[0]; e[5] = b.hi[1]; e[6] = b.hi[2];
        e[3] = u2f((TAG_BOX << TAG_SHIFT) | (uint32_t)skip);
    }

    // compute_aabb (shape.rs:218-233): tight box of the triangles, then +-0.0001 on every face
    Box padded_box(const uint32_t* idx, size_t n) const {
        Box b = box_infinity();
        for (size_t i = 0; i < n; i++) {
            const rt_triangle& t = mesh->tris[idx[i]];
            grow_point(b, t.p[0]); grow_point(b, t.p[1]); grow_point(b, t.p[2]);
        }
        for (int k = 0; k < 3; k++) b.hi[k] += 0.0001f;
        for (int k = 0; k < 3; k++) b.lo[k] -= 0.0001f;
        return b;
    }

    // Mesh::build_bvh (shape.rs:264-290) on the triangles r[0..n), written in pre-order at entries
    // [at, at + 2n - 1): a subtree of n >= 1 triangles always takes 2n - 1 entries, so every
    // node's place is known before its children are built and the two halves of a big node can
    // be built by different threads (they sort disjoint ranges and write disjoint entries).
    // Every record carries its triangle's three sort keys, so a sort moves records and never looks
    // a key up elsewhere; big ranges are sorted by a stable radix sort (sort_keys).  Returns the
    // tight box of the range: a node's box is the union of its halves' (minima and maxima are exact,
    // so the union equals compute_aabb's scan over all the vertices, shape.rs:218-233, except for the
    // sign of a zero, which the padding removes: -0.0 - 0.0001 = +0.0 - 0.0001).
    static size_t mesh_entries(size_t n) { return n ? 2 * n - 1 : 0; }
    struct TriKey { float k[3]; uint32_t i; };     // Triangle::get_aabb().min (shape.rs:195-199) and the triangle
    // an unsigned whose order is the floats' order, -0.0 and +0.0 alike (they compare equal)
    static uint32_t order_bits(float f) {
        uint32_t u;
        std::memcpy(&u, &f, 4);
        if (u == 0x80000000u) u = 0u;
        return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
    }
    // r[0..n) in ascending k[axis], equal keys in their old order: slice::sort_by with partial_cmp
    // (shape.rs:268-272; a stable sort).  tmp: n records of scratch.
    static void sort_keys(TriKey* r, TriKey* tmp, size_t n, int axis) {
        // (RT_BUILD_NO_RADIX=1: the comparison sort at every size, for tests of the radix sort against it)
        if (n < 4096 || std::getenv("RT_BUILD_NO_RADIX")) {
            std::stable_sort(r, r + n, [axis](const TriKey& a, const TriKey& b) { return a.k[axis] < b.k[axis]; });
            return;
        }
        TriKey *src = r, *dst = tmp;               // least significant digit first: 11 + 11 + 10 bits
        std::vector<size_t> count(2048);
        for (int pass = 0; pass < 3; pass++) {
            const int shift = pass * 11;
            const uint32_t mask = pass == 2 ? 0x3FFu : 0x7FFu;
            std::fill(count.begin(), count.end(), 0);
            for (size_t i = 0; i < n; i++) count[(order_bits(src[i].k[axis]) >> shift) & mask]++;
            size_t sum = 0;
            for (size_t& c : count) { const size_t here = c; c = sum; sum += here; }
            for (size_t i = 0; i < n; i++) dst[count[(order_bits(src[i].k[axis]) >> shift) & mask]++] = src[i];
            std::swap(src, dst);
        }
        std::memcpy(r, tmp, n * sizeof(TriKey));   // (three passes: the result stands in tmp)
    }
    Box triangle_box(uint32_t local) const {
        const rt_triangle& t = mesh->tris[local];
        Box b{{t.p[0][0], t.p[0][1], t.p[0][2]}, {t.p[0][0], t.p[0][1], t.p[0][2]}};
        grow_point(b, t.p[1]); grow_point(b, t.p[2]);
        return b;
    }
    static Box padded(Box b) {                     // compute_aabb's last step: +-0.0001 on every face
        for (int k = 0; k < 3; k++) b.hi[k] += 0.0001f;
        for (int k = 0; k < 3; k++) b.lo[k] -= 0.0001f;
        return b;
    }
    Box fill_mesh_range(TriKey* r, TriKey* tmp, size_t n, int axis, size_t at, int fork_levels) {
        if (n == 0) return box_infinity();         // MeshBvhNode::None never hits
        if (n == 1) { write_triangle(at, r[0].i); return triangle_box(r[0].i); }
        if (n == 2) {
            const Box tight = surrounding(triangle_box(r[0].i), triangle_box(r[1].i));
            write_box(at, padded(tight), at + 3);
            write_triangle(at + 1, r[1].i);        // left = pop() = last
            write_triangle(at + 2, r[0].i);        // right = pop() = first
            return tight;
        }
        sort_keys(r, tmp, n, axis);
        const size_t half = n / 2;                 // split_off(len / 2)
        const size_t left_at = at + 1, right_at = left_at + mesh_entries(half);
        const int next = (axis + 1) % 3;
        Box left_box, right_box;
        if (fork_levels > 0 && n >= 65536) {
            // (an exception in either half -- bad_alloc from a sort's scratch -- must not reach a joinable
            // thread's destructor: the left half reports through `thrown`, the guard joins on every path)
            std::exception_ptr thrown;
            std::thread left([&, this]() {
                try { left_box = fill_mesh_range(r, tmp, half, next, left_at, fork_levels - 1); }
                catch (...) { thrown = std::current_exception(); }
            });
            struct Join { std::thread& t; ~Join() { if (t.joinable()) t.join(); } } join{left};
            right_box = fill_mesh_range(r + half, tmp + half, n - half, next, right_at, fork_levels - 1);
            left.join();
            if (thrown) std::rethrow_exception(thrown);
        } else {
            left_box = fill_mesh_range(r, tmp, half, next, left_at, 0);
            right_box = fill_mesh_range(r + half, tmp + half, n - half, next, right_at, 0);
        }
        const Box tight = surrounding(left_box, right_box);
        write_box(at, padded(tight), at + mesh_entries(n));
        return tight;
    }

    // The opt-in tree: same entry format, boxes and leaves, but every node is split where a binned
    // surface-area heuristic is cheapest instead of at the median of min[axis] (shape.rs:264-290).
    // Big and small triangles then part ways near the root, which the median split never
    // achieves (cornell_box.obj: every one of the upper 13 levels spans the room).
    static float half_area(const Box& b) {
        const float dx = b.hi[0] - b.lo[0], dy = b.hi[1] - b.lo[1], dz = b.hi[2] - b.lo[2];
        return dx * dy + dy * dz + dz * dx;
    }
    Box tight_box(const uint32_t* idx, size_t n) const {
        Box b = box_infinity();
        for (size_t i = 0; i < n; i++) {
            const rt_triangle& t = mesh->tris[idx[i]];
            grow_point(b, t.p[0]); grow_point(b, t.p[1]); grow_point(b, t.p[2]);
        }
        return b;
    }
    // Binned SAH can peel one triangle off per level on a skewed mesh; below SAH_MAX_DEPTH levels the
    // nodes are halved instead, so the recursion (and every later walk over the tree) stays shallow.
    static constexpr int SAH_MAX_DEPTH = 40;
    // Leaves of several triangles (sah_leaf > 1): a node of at most sah_leaf triangles is not split
    // further; its box is followed by the triangles themselves, in file order, as siblings (the
    // walks take a run of siblings one after the other, narrowing t_max as they go, exactly as they
    // take a pair).  The subtree then needs fewer entries than the 2n - 1 reserved for it: the
    // slots left over are marked in slot_unused and squeezed out once the mesh is complete.
    uint32_t sah_leaf = 1;
    std::vector<uint8_t> slot_unused;          // per entry slot of the mesh being built (offset slot_base)
    size_t slot_base = 0;
    void fill_mesh_range_sah(uint32_t* idx, size_t n, size_t at, int fork_levels, int depth = 0) {
        if (n == 0) return;
        if (n == 1) { write_triangle(at, idx[0]); return; }
        write_box(at, padded_box(idx, n), at + mesh_entries(n));
        if (n > 2 && n <= sah_leaf) {
            std::sort(idx, idx + n);
            for (size_t i = 0; i < n; i++) write_triangle(at + 1 + i, idx[i]);
            for (size_t sl = at + 1 + n; sl < at + mesh_entries(n); sl++) slot_unused[sl - slot_base] = 1;
            return;
        }
        size_t half = n / 2;
        if (n > 2 && depth < SAH_MAX_DEPTH) {
            constexpr int BINS = 16;
            const float* cen = tri_cen.data();
            float clo[3] = {INFINITY, INFINITY, INFINITY}, chi[3] = {-INFINITY, -INFINITY, -INFINITY};
            for (size_t i = 0; i < n; i++)
                for (int k = 0; k < 3; k++) {
                    clo[k] = std::min(clo[k], cen[idx[i] * 3 + k]);
                    chi[k] = std::max(chi[k], cen[idx[i] * 3 + k]);
                }
            float best = INFINITY;
            int best_axis = -1, best_bin = 0;
            float best_scale = 0.0f;
            for (int k = 0; k < 3; k++) {
                const float ext = chi[k] - clo[k];
                if (!(ext > 0.0f)) continue;
                const float scale = (float)BINS * (1.0f - 1e-6f) / ext;
                Box bb[BINS];
                size_t cnt[BINS] = {};
                for (int j = 0; j < BINS; j++) bb[j] = box_infinity();
                for (size_t i = 0; i < n; i++) {
                    const int j = std::min(BINS - 1, std::max(0, (int)((cen[idx[i] * 3 + k] - clo[k]) * scale)));
                    const rt_triangle& t = mesh->tris[idx[i]];
                    grow_point(bb[j], t.p[0]); grow_point(bb[j], t.p[1]); grow_point(bb[j], t.p[2]);
                    cnt[j]++;
                }
                float right_area[BINS];
                size_t right_cnt[BINS];
                Box acc = box_infinity();
                size_t c = 0;
                for (int j = BINS - 1; j > 0; j--) {
                    if (cnt[j]) acc = surrounding(acc, bb[j]);
                    c += cnt[j];
                    right_area[j] = c ? half_area(acc) : 0.0f;
                    right_cnt[j] = c;
                }
                acc = box_infinity();
                c = 0;
                for (int j = 0; j + 1 < BINS; j++) {       // split between bin j and j + 1
                    if (cnt[j]) acc = surrounding(acc, bb[j]);
                    c += cnt[j];
                    if (c == 0 || right_cnt[j + 1] == 0) continue;
                    const float cost = half_area(acc) * (float)c + right_area[j + 1] * (float)right_cnt[j + 1];
                    if (cost < best) { best = cost; best_axis = k; best_bin = j; best_scale = scale; }
                }
            }
            if (best_axis >= 0) {
                const int k = best_axis;
                const float lo = clo[k];
                uint32_t* mid = std::partition(idx, idx + n, [&](uint32_t t) {
                    const int j = std::min(BINS - 1, std::max(0, (int)((cen[t * 3 + k] - lo) * best_scale)));
                    return j <= best_bin;
                });
                half = (size_t)(mid - idx);
                if (half == 0 || half == n) half = n / 2;  // (cannot happen: both sides were non-empty)
            }
        }
        const size_t left_at = at + 1, right_at = left_at + mesh_entries(half);
        if (fork_levels > 0 && n >= 65536) {
            std::exception_ptr thrown;
            std::thread left([=, &thrown]() {
                try { fill_mesh_range_sah(idx, half, left_at, fork_levels - 1, depth + 1); }
                catch (...) { thrown = std::current_exception(); }
            });
            struct Join { std::thread& t; ~Join() { if (t.joinable()) t.join(); } } join{left};
            fill_mesh_range_sah(idx + half, n - half, right_at, fork_levels - 1, depth + 1);
            left.join();
            if (thrown) std::rethrow_exception(thrown);
        } else {
            fill_mesh_range_sah(idx, half, left_at, 0, depth + 1);
            fill_mesh_range_sah(idx + half, n - half, right_at, 0, depth + 1);
        }
    }

    // drops the unused slots of the mesh at entries [at, at + count) and maps the skip links
    void squeeze_mesh_slots(size_t at, size_t count) {
        std::vector<uint32_t> new_of(count + 1);
        uint32_t used = 0;
        for (size_t k = 0; k < count; k++) { new_of[k] = (uint32_t)at + used; used += slot_unused[k] ? 0u : 1u; }
        new_of[count] = (uint32_t)at + used;
        for (size_t k = 0; k < count; k++) {
            if (slot_unused[k]) continue;
            float* e = entry(at + k);
            uint32_t w;
            std::memcpy(&w, &e[3], 4);
            if ((w >> TAG_SHIFT) == TAG_BOX) {
                w = (TAG_BOX << TAG_SHIFT) | new_of[(w & SKIP_MASK) - at];
                std::memcpy(&e[3], &w, 4);
            }
            if (new_of[k] != at + k) std::memmove(entry(new_of[k]), e, ENTRY_FLOATS * sizeof(float));
        }
        out.entries.resize((size_t)new_of[count] * ENTRY_FLOATS);
    }

    // the markers around the tree of a transformed mesh (rt_internal.hpp: END_CODE_*)
    void push_marker(uint32_t code, uint32_t xform_index) {
        float* e = entry(push_entry());
        e[0] = u2f(code); e[1] = u2f(xform_index);
        e[3] = u2f(TAG_END << TAG_SHIFT);
    }
    void emit_object(uint32_t oi) {
        const SceneObject& o = objects[oi];
        if (!o.is_mesh) { push_sphere(o); return; }
        if (xform_of[oi]) {                    // Mesh::hit: ray.transform(&inverse_transform), shape.rs:454
            push_marker(END_CODE_ENTER, xform_of[oi] - 1);
            emit_mesh(o);
            push_marker(END_CODE_LEAVE, xform_of[oi] - 1);
            return;
        }
        emit_mesh(o);
    }
    void emit_mesh(const SceneObject& o) {
        mesh = &o;
        size_t n = o.tris.size();
        tri_min.resize(n * 3);
        for (size_t i = 0; i < n; i++) {       // Triangle::get_aabb().min, shape.rs:195-199
            Box b{{o.tris[i].p[0][0], o.tris[i].p[0][1], o.tris[i].p[0][2]},
                  {o.tris[i].p[0][0], o.tris[i].p[0][1], o.tris[i].p[0][2]}};
            grow_point(b, o.tris[i].p[1]); grow_point(b, o.tris[i].p[2]);
            tri_min[i * 3] = b.lo[0]; tri_min[i * 3 + 1] = b.lo[1]; tri_min[i * 3 + 2] = b.lo[2];
        }
        const size_t at = out.n_entries();
        out.entries.resize(out.entries.size() + mesh_entries(n) * ENTRY_FLOATS, 0.0f);
        // Mesh::hit enters at the root node (shape.rs:455); up to 2^5 threads on a big mesh
        if (sah) {
            std::vector<uint32_t> idx(n);
            std::iota(idx.begin(), idx.end(), 0u);
            tri_cen.resize(n * 3);
            for (size_t i = 0; i < n; i++) {
                const Box tb = tight_box(&idx[i], 1);
                for (int k = 0; k < 3; k++) tri_cen[i * 3 + k] = tb.lo[k] + tb.hi[k];
            }
            slot_base = at;
            slot_unused.assign(sah_leaf > 2 ? mesh_entries(n) : 0, 0);
            fill_mesh_range_sah(idx.data(), n, at, 5);
            if (sah_leaf > 2) squeeze_mesh_slots(at, mesh_entries(n));
        } else {
            std::vector<TriKey> recs(n), tmp(n);
            for (size_t i = 0; i < n; i++) recs[i] = TriKey{{tri_min[i * 3], tri_min[i * 3 + 1], tri_min[i * 3 + 2]}, (uint32_t)i};
            fill_mesh_range(recs.data(), tmp.data(), n, 0, at, 5);
        }
        mesh = nullptr;
    }

    // An item is an object (>= 0) or a container (< 0: groups[-1 - item]).
    const Box& item_box(long item) const { return item >= 0 ? obj_box[(size_t)item] : group_box[(size_t)(-1 - item)]; }
    void emit_item(long item) {
        if (item >= 0) { emit_object((uint32_t)item); return; }
        const SceneGroup& g = (*groups)[(size_t)(-1 - item)];
        if (g.kind == RT_ROOT_LIST) {          // HittableList::hit (shape.rs:60-74): no box of its own
            for (long c : g.items) emit_item(c);
        } else {
            std::vector<long> ids(g.items);
            emit_bvh(ids.data(), ids.size(), g.axis, false);
        }
    }
    // Bvh::new (shape.rs:94-138) on ids[0..n); returns the node's aabb.  dry: only the box (what
    // get_aabb() of a container inside another Bvh answers), nothing is emitted.
    Box emit_bvh(long* ids, size_t n, int axis, bool dry) {
        if (n == 0) {
            Box z{{0, 0, 0}, {0, 0, 0}};
            if (!dry) close_box(push_box(z));
            return z;
        }
        if (n == 1) {
            Box b = item_box(ids[0]);
            if (dry) return b;
            size_t bx = push_box(b);
            emit_item(ids[0]);
            close_box(bx);
            return b;
        }
        if (n == 2) {
            long left = ids[1], right = ids[0];        // pop(), pop()
            Box b = surrounding(item_box(left), item_box(right));
            if (dry) return b;
            size_t bx = push_box(b);
            emit_item(left);
            emit_item(right);
            close_box(bx);
            return b;
        }
        size_t half = n / 2;
        if (sah) {
            // The surface-area trees split the objects of a Bvh too: where a 16-bin surface-area heuristic
            // over the centres of their boxes is cheapest (the median of min[axis] where the centres give
            // no split).  One sphere as large as the scene then parts from the rest near the root instead
            // of widening the boxes of a whole branch.  Boxes stay the plain unions of Bvh::new.
            if (!sah_split_items(ids, n, half)) {
                std::stable_sort(ids, ids + n, [this, axis](long a, long b) { return item_box(a).lo[axis] < item_box(b).lo[axis]; });
                half = n / 2;
            }
        } else {
            std::stable_sort(ids, ids + n, [this, axis](long a, long b) {
                return item_box(a).lo[axis] < item_box(b).lo[axis];
            });
        }
        size_t bx = dry ? 0 : push_box(Box{});  // box is known only after both children exist
        Box l = emit_bvh(ids, half, (axis + 1) % 3, dry);
        Box r = emit_bvh(ids + half, n - half, (axis + 1) % 3, dry);
        Box b = surrounding(l, r);
        if (dry) return b;
        float* e = entry(bx);
        e[0] = b.lo[0]; e[1] = b.lo[1]; e[2] = b.lo[2];
        e[4] = b.hi[0]; e[5] = b.hi[1]; e[6] = b.hi[2];
        close_box(bx);
        return b;
    }
    // partitions ids[0..n) (n >= 3) at the cheapest of the 3 x 15 binned surface-area splits; false when
    // the centres of the boxes give none
    bool sah_split_items(long* ids, size_t n, size_t& half) const {
        constexpr int BINS = 16;
        auto centre2 = [this](long it, int k) { const Box& b = item_box(it); return b.lo[k] + b.hi[k]; };
        float clo[3] = {INFINITY, INFINITY, INFINITY}, chi[3] = {-INFINITY, -INFINITY, -INFINITY};
        for (size_t i = 0; i < n; i++)
            for (int k = 0; k < 3; k++) {
                const float c = centre2(ids[i], k);
                if (c == c) { clo[k] = std::min(clo[k], c); chi[k] = std::max(chi[k], c); }
            }
        float best = INFINITY, best_scale = 0.0f;
        int best_axis = -1, best_bin = 0;
        for (int k = 0; k < 3; k++) {
            const float ext = chi[k] - clo[k];
            if (!(ext > 0.0f) || !(ext < INFINITY)) continue;
            const float scale = (float)BINS * (1.0f - 1e-6f) / ext;
            Box bb[BINS];
            size_t cnt[BINS] = {};
            for (int j = 0; j < BINS; j++) bb[j] = box_infinity();
            for (size_t i = 0; i < n; i++) {
                const float c = centre2(ids[i], k);
                const int j = c == c ? std::min(BINS - 1, std::max(0, (int)((c - clo[k]) * scale))) : 0;
                bb[j] = surrounding(bb[j], item_box(ids[i]));
                cnt[j]++;
            }
            float right_area[BINS];
            size_t right_cnt[BINS];
            Box acc = box_infinity();
            size_t c = 0;
            for (int j = BINS - 1; j > 0; j--) {
                if (cnt[j]) acc = surrounding(acc, bb[j]);
                c += cnt[j];
                right_area[j] = c ? half_area(acc) : 0.0f;
                right_cnt[j] = c;
            }
            acc = box_infinity();
            c = 0;
            for (int j = 0; j + 1 < BINS; j++) {
                if (cnt[j]) acc = surrounding(acc, bb[j]);
                c += cnt[j];
                if (c == 0 || right_cnt[j + 1] == 0) continue;
                const float cost = half_area(acc) * (float)c + right_area[j + 1] * (float)right_cnt[j + 1];
                if (cost < best) { best = cost; best_axis = k; best_bin = j; best_scale = scale; }
            }
        }
        if (best_axis < 0) return false;
        const int k = best_axis;
        const float lo = clo[k];
        long* mid = std::stable_partition(ids, ids + n, [&](long it) {
            const float c = centre2(it, k);
            const int j = c == c ? std::min(BINS - 1, std::max(0, (int)((c - lo) * best_scale))) : 0;
            return j <= best_bin;
        });
        half = (size_t)(mid - ids);
        return half > 0 && half < n;
    }

    // get_aabb() of every container, innermost first (a container's index is above its parent's)
    void compute_group_boxes() {
        group_box.assign(groups->size(), box_infinity());
        for (size_t g = groups->size(); g-- > 1;) {
            const SceneGroup& gr = (*groups)[g];
            if (gr.kind == RT_ROOT_LIST) {     // shape.rs:76-85: fold of obj.get_aabb().surrounding(&acc)
                Box acc = box_infinity();
                for (long c : gr.items) acc = surrounding(item_box(c), acc);
                group_box[g] = acc;
            } else {
                std::vector<long> ids(gr.items);
                group_box[g] = emit_bvh(ids.data(), ids.size(), gr.axis, true);
            }
        }
    }
};

// Child links for the ordered walk (rt_internal.hpp: ORD_*), derived from the finished entry
// array: a box's first child is the next entry; it has a second child when the first child's
// subtree ends before the box's own skip link.  Also measures the nesting depth of the boxes.
void annotate_children(FlatScene& out) {
    const size_t n = out.n_entries();
    out.ordered_ok = n <= ORD_RIGHT_MASK;
    auto word = [&](size_t i, int k) { uint32_t u; std::memcpy(&u, &out.entries[i * ENTRY_FLOATS + k], 4); return u; };
    auto is_box = [&](size_t i) { return (word(i, 3) >> TAG_SHIFT) == TAG_BOX; };
    auto end_of = [&](size_t i) { return is_box(i) ? (size_t)(word(i, 3) & SKIP_MASK) : i + 1; };
    // 2 x centre of an entry's bounds (only differences along one axis are compared)
    auto centre2 = [&](size_t i, float c[3]) {
        const float* e = &out.entries[i * ENTRY_FLOATS];
        const uint32_t tag = word(i, 3) >> TAG_SHIFT;
        for (int k = 0; k < 3; k++) {
            if (tag == TAG_BOX) c[k] = e[k] + e[4 + k];
            else if (tag == TAG_SPH) c[k] = 2.0f * e[k];
            else if (tag == TAG_END) c[k] = 0.0f;          // a marker around a transformed mesh: no bounds
            else {
                const float p1 = e[k] + e[4 + k], p2 = e[k] + e[8 + k];
                c[k] = std::min(e[k], std::min(p1, p2)) + std::max(e[k], std::max(p1, p2));
            }
        }
    };
    std::vector<size_t> open_until;            // skip links of the boxes the cursor is inside
    uint32_t depth = 0;
    for (size_t i = 0; i < n; i++) {
        while (!open_until.empty() && open_until.back() <= i) open_until.pop_back();
        if (!is_box(i)) continue;
        const size_t end = word(i, 3) & SKIP_MASK;
        open_until.push_back(end);
        depth = std::max<uint32_t>(depth, (uint32_t)open_until.size());
        uint32_t link = 0;
        if (i + 1 < end) {
            const size_t second = end_of(i + 1);
            // (a box over a run of more than two siblings -- a leaf of several triangles, a
            // HittableList inside a Bvh -- keeps link 0: its children are taken in array order, which
            // the octant copies do and the stack of the ordered walk cannot)
            if (second < end && end_of(second) != end) out.multi_child = true;
            if (second < end && end_of(second) == end && out.ordered_ok) {
                float c1[3], c2[3];
                centre2(i + 1, c1);
                centre2(second, c2);
                int axis = 0;
                for (int k = 1; k < 3; k++)
                    if (std::fabs(c1[k] - c2[k]) > std::fabs(c1[axis] - c2[axis])) axis = k;
                link = (uint32_t)second | ((uint32_t)axis << ORD_AXIS_SHIFT) | (c1[axis] <= c2[axis] ? ORD_FIRST_LOWER : 0u);
            }
        }
        std::memcpy(&out.entries[i * ENTRY_FLOATS + 7], &link, 4);
    }
    out.depth = depth;
    if (depth > ORDERED_MAX_DEPTH) out.ordered_ok = false;
}

// The compact form for the per-lane walks (FlatScene::units): 16-byte units, a box or a sphere in
// two (the first two float4 of its entry), a triangle in three (its first three float4) or four:
// the last word of a triangle's third unit holds the distance to the entry behind it, and it is
// four where that entry is a box or a sphere that would otherwise start at an odd unit, so that
// every box and sphere (32 bytes) starts at a multiple of 32 bytes and never straddles a line.
bool pack_units(FlatScene& out) {       // false: the unit links do not fit 30 bits
    const size_t n = out.n_entries();
    auto word = [&](size_t i, int k) { uint32_t u; std::memcpy(&u, &out.entries[i * ENTRY_FLOATS + k], 4); return u; };
    auto tag_of = [&](size_t i) { return word(i, 3) >> TAG_SHIFT; };
    std::vector<uint32_t> unit_of(n + 1);
    uint64_t u = 0;
    for (size_t i = 0; i < n; i++) {
        // only a triangle makes the count odd, so an odd count here means the entry before is one
        if (tag_of(i) != TAG_TRI && (u & 1u)) u++;
        unit_of[i] = (uint32_t)u;
        u += tag_of(i) == TAG_TRI ? 3 : 2;
    }
    if (u + UNITS_END > SKIP_MASK) return false;
    unit_of[n] = (uint32_t)u;
    if (u > ORD_RIGHT_MASK) out.ordered_ok = false;
    out.units.assign((size_t)(u + UNITS_END) * 4, 0.0f);
    {
        const uint32_t end_word = TAG_END << TAG_SHIFT;     // the end units: only the tag word matters
        for (uint32_t k = 0; k < UNITS_END; k++) std::memcpy(&out.units[((size_t)u + k) * 4 + 3], &end_word, 4);
    }
    for (size_t i = 0; i < n; i++) {
        const float* e = &out.entries[i * ENTRY_FLOATS];
        float* d = &out.units[(size_t)unit_of[i] * 4];
        const uint32_t tag = tag_of(i);
        std::memcpy(d, e, tag == TAG_TRI ? 48 : 32);
        if (tag == TAG_BOX) {
            const uint32_t skip = (TAG_BOX << TAG_SHIFT) | unit_of[word(i, 3) & SKIP_MASK];
            std::memcpy(&d[3], &skip, 4);
            uint32_t link = word(i, 7);
            if (link & ORD_RIGHT_MASK) link = (link & ~ORD_RIGHT_MASK) | (unit_of[link & ORD_RIGHT_MASK] & ORD_RIGHT_MASK);
            std::memcpy(&d[7], &link, 4);
        } else if (tag == TAG_TRI) {
            const uint32_t stride = unit_of[i + 1] - unit_of[i];      // 3, or 4 with a padding unit
            std::memcpy(&d[11], &stride, 4);
        }
    }
    return true;
}

// Word offsets of the explicit links inside an entry's units (FlatScene::walk, octant copies).
constexpr int LINK_BOX_CHILD = 7, LINK_TRI_NEXT = 11, LINK_SPH_NEXT = 5;

// FlatScene::walk: see rt_internal.hpp.  A "group" is the run of siblings under one box (or the
// top-level entries); groups are laid out breadth first until `top_budget` units are used, the
// rest depth first (a group, then the groups below its members in order).  The successor of an
// entry in the reference's visiting order is the entry behind its subtree in pre-order, whatever
// the memory order, so with pos[] = unit of every entry the links are pos[skip], pos[i + 1].
bool pack_walk_units(FlatScene& out, uint32_t top_budget, uint32_t pair_align) {
    const size_t n = out.n_entries();
    auto word = [&](size_t i, int k) { uint32_t u; std::memcpy(&u, &out.entries[i * ENTRY_FLOATS + k], 4); return u; };
    auto tag_of = [&](size_t i) { return word(i, 3) >> TAG_SHIFT; };
    auto end_of = [&](size_t i) { return tag_of(i) == TAG_BOX ? (size_t)(word(i, 3) & SKIP_MASK) : i + 1; };
    std::vector<uint32_t> pos(n + 1);
    uint64_t u = 0;
    struct Group { size_t first, end; };
    auto place = [&](const Group& g) {         // the siblings side by side
        bool only_pairs = true;
        size_t members = 0;
        for (size_t c = g.first; c < g.end; c = end_of(c), members++) only_pairs = only_pairs && tag_of(c) != TAG_TRI;
        if (only_pairs && members >= 2 && pair_align > 1) u = (u + pair_align - 1) / pair_align * pair_align;
        for (size_t c = g.first; c < g.end; c = end_of(c)) {
            if (tag_of(c) != TAG_TRI && (u & 1u)) u++;     // boxes and spheres: 32-byte aligned
            pos[c] = (uint32_t)u;
            u += tag_of(c) == TAG_TRI ? 3 : 2;
            if (u > SKIP_MASK) return;
        }
    };
    std::vector<Group> queue{Group{0, n}};
    size_t head = 0;
    if (std::getenv("RT_LAYOUT_PREORDER")) {   // experiments: the entries in pre-order (a box, then its first child)
        for (size_t i = 0; i < n; i++) {
            if (tag_of(i) != TAG_TRI && (u & 1u)) u++;
            pos[i] = (uint32_t)u;
            u += tag_of(i) == TAG_TRI ? 3 : 2;
            if (u > SKIP_MASK) return false;
        }
        queue.clear();
    }
    while (head < queue.size() && u < top_budget) {        // breadth first: the top of the trees
        const Group g = queue[head++];
        place(g);
        for (size_t c = g.first; c < g.end; c = end_of(c))
            if (tag_of(c) == TAG_BOX && c + 1 < end_of(c)) queue.push_back(Group{c + 1, end_of(c)});
    }
    out.walk_top_units = (uint32_t)std::min<uint64_t>(u, top_budget ? UINT32_MAX : 0);
    std::vector<Group> stack;                  // depth first for the rest, in breadth-first order of their roots
    for (size_t q = queue.size(); q > head; q--) stack.push_back(queue[q - 1]);
    std::vector<Group> below;
    while (!stack.empty()) {
        const Group g = stack.back();
        stack.pop_back();
        place(g);
        if (u > SKIP_MASK) return false;
        below.clear();
        for (size_t c = g.first; c < g.end; c = end_of(c))
            if (tag_of(c) == TAG_BOX && c + 1 < end_of(c)) below.push_back(Group{c + 1, end_of(c)});
        for (size_t k = below.size(); k > 0; k--) stack.push_back(below[k - 1]);   // first member's group next
    }
    if (u & 1u) u++;
    if (u + UNITS_END > SKIP_MASK) return false;
    pos[n] = (uint32_t)u;                      // where the last entries' links lead: the end units
    out.walk.assign((size_t)(u + UNITS_END) * 4, 0.0f);
    const uint32_t end_word = TAG_END << TAG_SHIFT;
    for (uint32_t k = 0; k < UNITS_END; k++) std::memcpy(&out.walk[((size_t)u + k) * 4 + 3], &end_word, 4);
    for (size_t i = 0; i < n; i++) {
        const float* e = &out.entries[i * ENTRY_FLOATS];
        float* d = &out.walk[(size_t)pos[i] * 4];
        const uint32_t tag = tag_of(i);
        std::memcpy(d, e, tag == TAG_TRI ? 48 : 32);
        const uint32_t next = pos[end_of(i)];
        if (tag == TAG_BOX) {
            const uint32_t skip = (TAG_BOX << TAG_SHIFT) | next, child = pos[i + 1];   // (a childless box: child == skip)
            std::memcpy(&d[3], &skip, 4);
            std::memcpy(&d[LINK_BOX_CHILD], &child, 4);
        } else if (tag == TAG_TRI) {
            std::memcpy(&d[LINK_TRI_NEXT], &next, 4);
        } else {
            std::memcpy(&d[LINK_SPH_NEXT], &next, 4);
        }
    }
    return true;
}

}  // namespace

namespace {
// one octant's copy of the unit array (build_octant_units)
struct OctantCopy {
    const uint32_t* src;       // the reference-order units, as words
    uint32_t* dst;             // this octant's copy
    uint32_t base;             // first unit of the copy in the whole array
    uint32_t oct;

    uint32_t tag(uint32_t u) const { return src[(size_t)u * 4 + 3] >> TAG_SHIFT; }
    uint32_t end_of(uint32_t u) const {
        const uint32_t t = tag(u);
        return t == TAG_BOX ? (src[(size_t)u * 4 + 3] & SKIP_MASK) : u + (t == TAG_TRI ? src[(size_t)u * 4 + 11] : 2u);
    }
    // copies the subtree at unit `u` to position `at` of the copy; returns the position behind it
    uint32_t emit(uint32_t u, uint32_t at) {
        if (tag(u) != TAG_BOX) {
            const uint32_t n = end_of(u) - u;
            std::memcpy(dst + (size_t)at * 4, src + (size_t)u * 4, (size_t)n * 16);
            return at + n;
        }
        std::memcpy(dst + (size_t)at * 4, src + (size_t)u * 4, 32);
        const uint32_t end = end_of(u), link = src[(size_t)u * 4 + 7];
        uint32_t pos = at + 2;
        const uint32_t second = link & ORD_RIGHT_MASK;
        if (second != 0u) {
            const uint32_t axis = (link >> ORD_AXIS_SHIFT) & 3u;
            const bool dir_positive = ((oct >> axis) & 1u) == 0u;
            const bool first_is_near = dir_positive == ((link & ORD_FIRST_LOWER) != 0u);
            pos = emit(first_is_near ? u + 2 : second, pos);
            pos = emit(first_is_near ? second : u + 2, pos);
        } else {
            for (uint32_t c = u + 2; c < end; c = end_of(c)) pos = emit(c, pos);
        }
        dst[(size_t)at * 4 + 3] = (TAG_BOX << TAG_SHIFT) | (base + pos);
        dst[(size_t)at * 4 + 7] = 0u;
        return pos;
    }
};
}  // namespace

void build_octant_units(const uint32_t* units, size_t n_units, std::vector<uint32_t>& out) {
    const size_t stride = n_units + UNITS_END;               // units per copy
    out.assign(stride * 4 * 8, 0u);
    std::vector<std::thread> workers;
    for (uint32_t oct = 0; oct < 8; oct++)
        workers.emplace_back([=, &out]() {
            OctantCopy c{units, out.data() + (size_t)oct * stride * 4, (uint32_t)(oct * stride), oct};
            uint32_t pos = 0;
            for (uint32_t u = 0; u < n_units; u = c.end_of(u)) pos = c.emit(u, pos);   // top-level entries in order
            for (uint32_t k = 0; k < UNITS_END; k++)                                    // this copy's end units
                c.dst[((size_t)pos + k) * 4 + 3] = TAG_END << TAG_SHIFT;
            // the walk reads its successors from the entries: child / next links of this pre-order
            for (uint32_t u = 0; u < pos;) {
                uint32_t* d = c.dst + (size_t)u * 4;
                const uint32_t t = d[3] >> TAG_SHIFT;
                if (t == TAG_BOX) { d[LINK_BOX_CHILD] = c.base + u + 2; u += 2; }
                else if (t == TAG_TRI) { const uint32_t stride = d[LINK_TRI_NEXT]; d[LINK_TRI_NEXT] = c.base + u + stride; u += stride; }
                else { d[LINK_SPH_NEXT] = c.base + u + 2; u += 2; }
            }
        });
    for (std::thread& w : workers) w.join();
}

// ---------------------------------------------------------------------------------------------
// scene cache
// ---------------------------------------------------------------------------------------------
namespace {
struct LayoutParams { uint32_t top_budget, pair_align, preorder; };
LayoutParams layout_params() {                 // experiments: RT_LAYOUT_TOP / RT_LAYOUT_ALIGN (units), RT_LAYOUT_PREORDER
    LayoutParams p{2048, 4, 0};
    if (const char* e = std::getenv("RT_LAYOUT_TOP")) p.top_budget = (uint32_t)std::max(0l, std::atol(e));
    if (const char* e = std::getenv("RT_LAYOUT_ALIGN")) p.pair_align = (uint32_t)std::max(1l, std::atol(e));
    if (std::getenv("RT_LAYOUT_PREORDER")) p.preorder = 1;
    return p;
}
// 64-bit hash of a byte range, eight bytes at a time (multiply-xorshift mixing); not cryptographic:
// it only has to tell scenes apart
uint64_t hash_bytes(uint64_t h, const void* data, size_t n) {
    const unsigned char* p = (const unsigned char*)data;
    auto mix = [](uint64_t x) { x ^= x >> 32; x *= 0xD6E8FEB86659FD93ull; x ^= x >> 32; x *= 0xD6E8FEB86659FD93ull; return x ^ (x >> 32); };
    size_t i = 0;
    for (; i + 8 <= n; i += 8) {
        uint64_t w;
        std::memcpy(&w, p + i, 8);
        h = mix(h ^ w) + 0x9E3779B97F4A7C15ull;
    }
    uint64_t tail = 0;
    std::memcpy(&tail, p + i, n - i);
    return mix(h ^ tail ^ ((uint64_t)n << 56));
}
constexpr char CACHE_MAGIC[8] = {'R', 'T', 'A', 'M', 'D', 'S', 'C', '4'};
// Version of what the builder writes into the arrays (tree construction, entry and unit formats,
// primitive keys, layout rules).  It is part of the cache key: bump it with any change to the
// output of this file, and files of older builders are simply never found.
constexpr uint32_t BUILDER_VERSION = 6;
struct CacheHeader {
    char magic[8];
    uint64_t key;
    uint64_t n_entries_f, n_units_f, n_walk_f, n_geo_f, n_meta, n_mat_f, n_xform_f, n_order;   // element counts of the arrays
    uint32_t kinds_present, has_spheres, depth, ordered_ok, walk_top_units, multi_child;
};

// Cheap structural checks of arrays read from a file: a damaged file whose header and tail
// survived must not hand out-of-range links to the GPU walks.
bool cache_arrays_sane(const FlatScene& f) {
    auto wordv = [](const std::vector<float>& v, size_t i) { uint32_t u; std::memcpy(&u, &v[i], 4); return u; };
    if (f.entries.size() % ENTRY_FLOATS || f.units.size() % 4 || f.walk.size() % 4 || f.prim_geo.size() % 4 ||
        f.materials.size() % 8 || f.xforms.size() % 32)
        return false;
    const size_t n = f.n_entries(), n_prims = f.prim_meta.size(), n_mats = f.materials.size() / 8;
    if (f.prim_geo.size() != n_prims * 4 || f.prim_order.size() != n_prims || n_prims > MAX_PRIMS || n_mats == 0) return false;
    if (f.units.size() < UNITS_END * 4 || f.walk.size() < UNITS_END * 4) return false;
    const size_t n_units = f.n_units(), n_walk = f.n_walk();
    if (f.walk_top_units > n_walk || f.depth > n + 1) return false;
    std::vector<uint8_t> seen(n_prims, 0);
    for (uint32_t id : f.prim_order) {
        if (id >= n_prims || seen[id]) return false;
        seen[id] = 1;
    }
    for (uint32_t m : f.prim_meta)
        if ((m & MAT_MASK) >= n_mats) return false;
    size_t prims_seen = 0;
    for (size_t i = 0; i < n; i++) {
        const uint32_t w = wordv(f.entries, i * ENTRY_FLOATS + 3), tag = w >> TAG_SHIFT;
        if (tag == TAG_BOX) {
            const uint32_t skip = w & SKIP_MASK;
            if (skip <= i || skip > n) return false;
        } else if (tag == TAG_TRI || tag == TAG_SPH) {
            const uint32_t key = wordv(f.entries, i * ENTRY_FLOATS + 7);
            if (((key & KEY_SPHERE) != 0) != (tag == TAG_SPH) || (key & ~KEY_MASK)) return false;
            const uint32_t rank = (key & KEY_SPHERE) ? (key & PRIM_ID_MASK) : (KEY_NONE - 1u - key);
            if (rank >= n_prims) return false;
            prims_seen++;
        } else if (wordv(f.entries, i * ENTRY_FLOATS) == END_CODE_ENTER || wordv(f.entries, i * ENTRY_FLOATS) == END_CODE_LEAVE) {
            if (wordv(f.entries, i * ENTRY_FLOATS + 1) >= f.xforms.size() / 32) return false;
        } else {
            return false;
        }
    }
    if (prims_seen != n_prims) return false;
    // pre-order units: entries follow each other, a triangle names its own length
    for (size_t u = 0; u < n_units;) {
        const uint32_t w = wordv(f.units, u * 4 + 3), tag = w >> TAG_SHIFT;
        if (tag == TAG_BOX) {
            if ((w & SKIP_MASK) <= u || (w & SKIP_MASK) > n_units) return false;
            u += 2;
        } else if (tag == TAG_TRI) {
            const uint32_t stride = wordv(f.units, u * 4 + 11);
            if (stride != 3 && stride != 4) return false;
            u += stride;
        } else {
            u += 2;
        }
        if (u > n_units) return false;
    }
    // linked units: every link reachable from unit 0 stays inside the array (n_walk = "done")
    std::vector<uint8_t> visited(n_walk + 1, 0);
    std::vector<uint32_t> todo;
    if (n_walk) todo.push_back(0u);
    auto push = [&](uint32_t u) -> bool {
        if (u > n_walk) return false;
        if (u < n_walk && !visited[u]) { visited[u] = 1; todo.push_back(u); }
        return true;
    };
    if (n_walk) visited[0] = 1;
    while (!todo.empty()) {
        const uint32_t u = todo.back();
        todo.pop_back();
        if ((size_t)u + 2 > n_walk) return false;
        const uint32_t w = wordv(f.walk, (size_t)u * 4 + 3), tag = w >> TAG_SHIFT;
        if (tag == TAG_BOX) {
            if (!push(w & SKIP_MASK) || !push(wordv(f.walk, (size_t)u * 4 + LINK_BOX_CHILD))) return false;
        } else if (tag == TAG_TRI) {
            if ((size_t)u + 3 > n_walk || !push(wordv(f.walk, (size_t)u * 4 + LINK_TRI_NEXT))) return false;
        } else {                                   // a sphere, or a marker around a transformed mesh
            if (!push(wordv(f.walk, (size_t)u * 4 + LINK_SPH_NEXT))) return false;
        }
    }
    return true;
}
}  // namespace

uint64_t scene_cache_key(const std::vector<rt_material>& mats, const std::vector<SceneObject>& objects,
                         const std::vector<SceneGroup>& groups, int root_kind, int build_mode) {
    uint64_t h = 0x5254414D44ull;
    const LayoutParams lp = layout_params();
    const uint32_t leaf = (uint32_t)(std::getenv("RT_SAH_LEAF") ? std::atol(std::getenv("RT_SAH_LEAF")) : 1l);
    const uint32_t head[10] = {(uint32_t)root_kind, (uint32_t)build_mode, lp.top_budget, lp.pair_align, lp.preorder,
                               ENTRY_FLOATS, UNITS_END, (uint32_t)mats.size(), build_mode == RT_BUILD_SAH ? leaf : 0u,
                               BUILDER_VERSION};
    h = hash_bytes(h, head, sizeof(head));
    if (!mats.empty()) h = hash_bytes(h, mats.data(), mats.size() * sizeof(rt_material));
    for (const SceneObject& o : objects) {
        const uint32_t kind[2] = {o.is_mesh ? 1u : 0u, o.has_xform ? 1u : 0u};
        h = hash_bytes(h, kind, sizeof(kind));
        if (!o.is_mesh) h = hash_bytes(h, &o.sphere, sizeof(o.sphere));
        else if (!o.tris.empty()) h = hash_bytes(h, o.tris.data(), o.tris.size() * sizeof(rt_triangle));
        if (o.has_xform) h = hash_bytes(h, o.xform, sizeof(o.xform));
    }
    for (const SceneGroup& g : groups) {
        const int32_t gk[2] = {g.kind, g.axis};
        h = hash_bytes(h, gk, sizeof(gk));
        if (!g.items.empty()) h = hash_bytes(h, g.items.data(), g.items.size() * sizeof(long));
    }
    return h;
}

bool scene_cache_load(const std::string& path, uint64_t key, FlatScene& out) {
    FILE* f = std::fopen(path.c_str(), "rb");
    if (!f) return false;
    CacheHeader hd;
    bool ok = std::fread(&hd, sizeof(hd), 1, f) == 1 && std::memcmp(hd.magic, CACHE_MAGIC, 8) == 0 && hd.key == key;
    FlatScene fs;
    auto rd = [&](auto& vec, uint64_t n) {
        if (!ok) return;
        if (n > ((uint64_t)1 << 36)) { ok = false; return; }
        vec.resize((size_t)n);
        if (n && std::fread(vec.data(), sizeof(vec[0]), (size_t)n, f) != (size_t)n) ok = false;
    };
    rd(fs.entries, hd.n_entries_f); rd(fs.units, hd.n_units_f); rd(fs.walk, hd.n_walk_f); rd(fs.prim_geo, hd.n_geo_f);
    rd(fs.prim_meta, hd.n_meta); rd(fs.materials, hd.n_mat_f); rd(fs.xforms, hd.n_xform_f); rd(fs.prim_order, hd.n_order);
    if (ok) {
        uint64_t tail = 0;                     // the key once more behind the arrays: a complete file
        ok = std::fread(&tail, 8, 1, f) == 1 && tail == key;
    }
    std::fclose(f);
    if (!ok) return false;
    fs.kinds_present = hd.kinds_present; fs.has_spheres = hd.has_spheres != 0; fs.depth = hd.depth;
    fs.ordered_ok = hd.ordered_ok != 0; fs.walk_top_units = hd.walk_top_units; fs.multi_child = hd.multi_child != 0;
    if (!cache_arrays_sane(fs)) return false;  // the caller then builds the scene (and rewrites the file)
    out = std::move(fs);
    return true;
}

bool scene_cache_store(const std::string& path, uint64_t key, const FlatScene& fl) {
    const std::string tmp = path + ".tmp" + std::to_string((unsigned long long)(uintptr_t)&fl) + "_" + std::to_string((long)std::rand());
    FILE* f = std::fopen(tmp.c_str(), "wb");
    if (!f) return false;
    CacheHeader hd{};
    std::memcpy(hd.magic, CACHE_MAGIC, 8);
    hd.key = key;
    hd.n_entries_f = fl.entries.size(); hd.n_units_f = fl.units.size(); hd.n_walk_f = fl.walk.size();
    hd.n_geo_f = fl.prim_geo.size(); hd.n_meta = fl.prim_meta.size(); hd.n_mat_f = fl.materials.size();
    hd.n_xform_f = fl.xforms.size(); hd.n_order = fl.prim_order.size();
    hd.kinds_present = fl.kinds_present; hd.has_spheres = fl.has_spheres ? 1u : 0u; hd.depth = fl.depth;
    hd.ordered_ok = fl.ordered_ok ? 1u : 0u; hd.walk_top_units = fl.walk_top_units; hd.multi_child = fl.multi_child ? 1u : 0u;
    bool ok = std::fwrite(&hd, sizeof(hd), 1, f) == 1;
    auto wr = [&](const auto& vec) {
        if (ok && !vec.empty()) ok = std::fwrite(vec.data(), sizeof(vec[0]), vec.size(), f) == vec.size();
    };
    wr(fl.entries); wr(fl.units); wr(fl.walk); wr(fl.prim_geo); wr(fl.prim_meta); wr(fl.materials); wr(fl.xforms); wr(fl.prim_order);
    ok = ok && std::fwrite(&key, 8, 1, f) == 1;
    ok = std::fclose(f) == 0 && ok;
    if (ok) ok = std::rename(tmp.c_str(), path.c_str()) == 0;
    if (!ok) std::remove(tmp.c_str());
    return ok;
}

std::string build_flat_scene(const std::vector<rt_material>& mats, std::vector<SceneObject>& objects,
                             const std::vector<SceneGroup>& groups, int root_kind, int build_mode, FlatScene& out,
                             const FlatScene* reference) {
    out = FlatScene();
    if (mats.empty()) return "no materials set";
    if (mats.size() > MAT_MASK) return "too many materials";
    for (const rt_material& m : mats)
        if (m.kind > RT_MAT_EMISSIVE) return "unknown material kind";

    // primitive ids, per-primitive shading data, per-object boxes
    Builder b(mats, objects, out);
    b.sah = build_mode == RT_BUILD_SAH;
    // Triangles per leaf of the opt-in trees: RT_SAH_LEAF, default 1.  Measured on one MI355X, leaves of
    // 1 / 2 / 4 / 8: the 10 M-triangle soup 311 / 311 / 277 / 230 Mrays/s (octant walk 317 / 317 / 297 / 263:
    // 342 -> 324 -> 287 box tests per ray, but 22 -> 47 -> 121 triangle tests, each three units against
    // a box's two), cornell_box.obj 6.5 -> 5.5 Grays/s at 4, thing.obj 14.2 -> 12.5: where triangles are
    // small against their spacing a box test is the cheaper way to cull them one by one.
    if (b.sah)
        b.sah_leaf = (uint32_t)std::min(16l, std::max(1l, std::getenv("RT_SAH_LEAF") ? std::atol(std::getenv("RT_SAH_LEAF")) : 1l));
    uint64_t prim = 0;
    for (SceneObject& o : objects) {
        o.prim_base = (uint32_t)prim;
        prim += o.is_mesh ? o.tris.size() : 1;
    }
    if (prim > MAX_PRIMS) return "too many primitives";
    out.prim_geo.resize(prim * 4, 0.0f);
    out.prim_meta.resize(prim, 0);
    b.obj_box.resize(objects.size());
    b.xform_of.assign(objects.size(), 0u);
    b.groups = &groups;
    if (groups.empty()) return "internal error: no top-level group";
    for (size_t oi = 0; oi < objects.size(); oi++) {
        SceneObject& o = objects[oi];
        if (!o.is_mesh) {
            if (o.sphere.mat >= mats.size()) return "sphere material index out of range";
            float* g = &out.prim_geo[(size_t)o.prim_base * 4];
            g[0] = o.sphere.c[0]; g[1] = o.sphere.c[1]; g[2] = o.sphere.c[2]; g[3] = o.sphere.r;
            out.prim_meta[o.prim_base] = b.matkind(o.sphere.mat) | META_SPHERE;
            out.has_spheres = true;
            out.kinds_present |= 1u << mats[o.sphere.mat].kind;
            Box bx;                            // Sphere::get_aabb, shape.rs:47-52
            for (int k = 0; k < 3; k++) { bx.lo[k] = o.sphere.c[k] - o.sphere.r; bx.hi[k] = o.sphere.c[k] + o.sphere.r; }
            b.obj_box[oi] = bx;
        } else {
            if (o.has_xform) {                 // Mesh.transform / inverse_transform, shape.rs:212-213, 250
                b.xform_of[oi] = (uint32_t)(out.xforms.size() / 32) + 1u;
                out.xforms.resize(out.xforms.size() + 32);
                float* x = &out.xforms[out.xforms.size() - 32];
                std::memcpy(x, o.xform, 64);
                mat4_inverse(o.xform, x + 16);
            }
            Box bx = box_infinity();           // Mesh.aabb: unpadded, of the transformed vertices, shape.rs:241-245
            for (size_t i = 0; i < o.tris.size(); i++) {
                const rt_triangle& t = o.tris[i];
                if (t.mat >= mats.size()) return "triangle material index out of range";
                float* g = &out.prim_geo[((size_t)o.prim_base + i) * 4];
                if (o.has_xform) {
                    for (int k = 0; k < 3; k++) {
                        float p[3];
                        mat4_mul_vec(o.xform, t.p[k], 1.0f, p);
                        grow_point(bx, p);
                    }
                    // the normal goes through the transform at every hit (shape.rs:303-307): keep it raw
                    g[0] = t.n0[0]; g[1] = t.n0[1]; g[2] = t.n0[2]; g[3] = u2f(b.xform_of[oi]);
                } else {
                    grow_point(bx, t.p[0]); grow_point(bx, t.p[1]); grow_point(bx, t.p[2]);
                    // shape.rs:306: normalize(+-v1.n) = +-(v1.n / |v1.n|); glam: v / sqrt((x*x+y*y)+z*z)
                    float len = std::sqrt((t.n0[0] * t.n0[0] + t.n0[1] * t.n0[1]) + t.n0[2] * t.n0[2]);
                    g[0] = t.n0[0] / len; g[1] = t.n0[1] / len; g[2] = t.n0[2] / len; g[3] = 0.0f;
                }
                out.prim_meta[o.prim_base + i] = b.matkind(t.mat);
                out.kinds_present |= 1u << mats[t.mat].kind;
            }
            b.obj_box[oi] = bx;
        }
    }
    out.materials.resize(mats.size() * 8, 0.0f);
    for (size_t i = 0; i < mats.size(); i++) {
        float* m = &out.materials[i * 8];
        m[0] = mats[i].rgb[0]; m[1] = mats[i].rgb[1]; m[2] = mats[i].rgb[2]; m[3] = mats[i].param;
        m[4] = u2f(mats[i].kind); m[5] = u2f(mats[i].flags);
    }

    for (const SceneGroup& g : groups) {
        if (g.kind != RT_ROOT_BVH && g.kind != RT_ROOT_LIST) return "unknown container kind";
        if (g.axis < 0 || g.axis > 2) return "Bvh axis must be 0, 1 or 2";
        for (long it : g.items)
            if (it >= (long)objects.size() || -1 - it >= (long)groups.size()) return "internal error: bad container item";
    }
    // RT_BUILD_TIMING=1: seconds per phase of the flattening on stderr
    const bool timing = std::getenv("RT_BUILD_TIMING") != nullptr;
    auto t_last = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) {
        if (!timing) return;
        const auto now = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[rt build] %-28s %.3f s\n", what, std::chrono::duration<double>(now - t_last).count());
        t_last = now;
    };
    b.compute_group_boxes();
    lap("geometry tables, boxes");
    std::vector<long> ids(groups[0].items);    // the children of World.hittable
    if (root_kind == RT_ROOT_SINGLE) {
        if (ids.size() != 1) return "RT_ROOT_SINGLE needs exactly one object";
        b.emit_item(ids[0]);
    } else if (root_kind == RT_ROOT_LIST) {
        // HittableList::hit (shape.rs:60-74): every object in order, t_max = closest so far
        for (long it : ids) b.emit_item(it);
    } else if (root_kind == RT_ROOT_BVH) {
        b.emit_bvh(ids.data(), ids.size(), 0, false);
    } else {
        return "unknown root kind";
    }
    lap("trees");
    if (out.n_entries() > SKIP_MASK) return "scene too large for 30-bit entry links";
    {   // Primitive keys (rt_internal.hpp): the rank of a primitive is its place among the primitive
        // entries of the REFERENCE-topology trees in pre-order, i.e. in the reference's visiting order.
        auto word = [](const FlatScene& f, size_t i, int k) { uint32_t u; std::memcpy(&u, &f.entries[i * ENTRY_FLOATS + k], 4); return u; };
        auto is_prim = [&](const FlatScene& f, size_t i) { const uint32_t t = word(f, i, 3) >> TAG_SHIFT; return t == TAG_TRI || t == TAG_SPH; };
        std::vector<uint32_t> rank_of(prim, 0u);
        if (!b.sah) {
            uint32_t rank = 0;
            for (size_t i = 0; i < out.n_entries(); i++)
                if (is_prim(out, i)) rank_of[word(out, i, 7)] = rank++;
        } else if (reference && reference->prim_order.size() == prim) {
            for (uint32_t r = 0; r < (uint32_t)prim; r++) rank_of[reference->prim_order[r]] = r;
        } else {                                   // another tree: the reference's trees once more, for their order alone
            FlatScene ref;
            Builder rb(mats, objects, ref);
            rb.obj_box = b.obj_box; rb.xform_of = b.xform_of; rb.groups = &groups; rb.group_box = b.group_box;
            std::vector<long> rids(groups[0].items);
            if (root_kind == RT_ROOT_SINGLE) rb.emit_item(rids[0]);
            else if (root_kind == RT_ROOT_LIST) { for (long it : rids) rb.emit_item(it); }
            else rb.emit_bvh(rids.data(), rids.size(), 0, false);
            uint32_t rank = 0;
            for (size_t i = 0; i < ref.n_entries(); i++)
                if (is_prim(ref, i)) rank_of[word(ref, i, 7)] = rank++;
            lap("reference order");
        }
        out.prim_order.assign(prim, 0u);
        for (uint32_t id = 0; id < (uint32_t)prim; id++) out.prim_order[rank_of[id]] = id;
        std::vector<float> geo(out.prim_geo.size());
        std::vector<uint32_t> meta(out.prim_meta.size());
        for (uint32_t r = 0; r < (uint32_t)prim; r++) {
            const uint32_t id = out.prim_order[r];
            std::memcpy(&geo[(size_t)r * 4], &out.prim_geo[(size_t)id * 4], 16);
            meta[r] = out.prim_meta[id];
        }
        out.prim_geo.swap(geo);
        out.prim_meta.swap(meta);
        for (size_t i = 0; i < out.n_entries(); i++)
            if (is_prim(out, i)) {
                const uint32_t key = prim_key(rank_of[word(out, i, 7)], (word(out, i, 3) >> TAG_SHIFT) == TAG_SPH);
                std::memcpy(&out.entries[i * ENTRY_FLOATS + 7], &key, 4);
            }
        lap("primitive keys");
    }
    annotate_children(out);
    lap("child links");
    if (!pack_units(out)) return "scene too large for 30-bit unit links";
    lap("pre-order units");
    const LayoutParams lp = layout_params();
    if (!pack_walk_units(out, lp.top_budget, lp.pair_align)) return "scene too large for 30-bit unit links";
    lap("linked units");
    if (!out.xforms.empty()) out.ordered_ok = false;       // the near-first walks do not follow the transform markers
    return std::string();
}

double expected_cost_per_ray(const float* entries, size_t n_entries) {
    auto word = [&](size_t i, int w) { uint32_t u; std::memcpy(&u, entries + i * ENTRY_FLOATS + w, 4); return u; };
    auto tag = [&](size_t i) { return word(i, 3) >> TAG_SHIFT; };
    auto area = [&](size_t i) -> double {              // of a box entry
        const float* e = entries + i * ENTRY_FLOATS;
        const double dx = (double)e[4] - e[0], dy = (double)e[5] - e[1], dz = (double)e[6] - e[2];
        const double a = dx * dy + dy * dz + dz * dx;
        return a == a && a > 0.0 ? a : 0.0;
    };
    // the scene's bounds: the union of what stands at the top (boxes as stored, primitives by their extent)
    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
    for (size_t i = 0; i < n_entries;) {
        const float* e = entries + i * ENTRY_FLOATS;
        const uint32_t t = tag(i);
        if (t == TAG_BOX) {
            for (int k = 0; k < 3; k++) { lo[k] = std::min(lo[k], (double)e[k]); hi[k] = std::max(hi[k], (double)e[4 + k]); }
            i = word(i, 3) & SKIP_MASK;
        } else {
            if (t == TAG_TRI)
                for (int k = 0; k < 3; k++) {
                    const double p0 = e[k], p1 = (double)e[k] + e[4 + k], p2 = (double)e[k] + e[8 + k];
                    lo[k] = std::min(lo[k], std::min(p0, std::min(p1, p2))); hi[k] = std::max(hi[k], std::max(p0, std::max(p1, p2)));
                }
            else if (t == TAG_SPH)
                for (int k = 0; k < 3; k++) { lo[k] = std::min(lo[k], (double)e[k] - std::fabs(e[4])); hi[k] = std::max(hi[k], (double)e[k] + std::fabs(e[4])); }
            i++;                                       // (markers of transformed meshes: what follows is in the mesh's space)
        }
    }
    const double rx = hi[0] - lo[0], ry = hi[1] - lo[1], rz = hi[2] - lo[2];
    double root_area = rx * ry + ry * rz + rz * rx;
    if (!(root_area > 0.0) || root_area != root_area) return 0.0;
    struct Level { size_t end; double p; };            // p: the probability that an entry directly inside is tested
    std::vector<Level> levels{Level{n_entries, 1.0}};
    double cost = 0.0, p_enter = 1.0, norm = root_area;
    bool inside_mesh = false;
    for (size_t i = 0; i < n_entries; i++) {
        while (levels.size() > 1 && levels.back().end <= i) levels.pop_back();
        const double p = levels.back().p;
        const uint32_t t = tag(i);
        if (t == TAG_BOX) {
            cost += p;
            const double a = area(i);
            if (inside_mesh && norm == 0.0) norm = a;  // the root box of a transformed mesh: areas inside count against it
            const double hit = norm > 0.0 ? std::min(1.0, a / norm) : 1.0;
            levels.push_back(Level{word(i, 3) & SKIP_MASK, (inside_mesh ? p_enter : 1.0) * hit});
        } else if (t == TAG_TRI) {
            cost += 1.6 * p;
        } else if (t == TAG_SPH) {
            cost += 1.3 * p;
        } else if (word(i, 0) == END_CODE_ENTER) {
            inside_mesh = true; p_enter = p; norm = 0.0;
        } else if (word(i, 0) == END_CODE_LEAVE) {
            inside_mesh = false; norm = root_area;
        }
    }
    return cost;
}

}  // namespace rtamd
