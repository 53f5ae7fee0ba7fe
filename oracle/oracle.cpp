// CPU oracle -- TEST INFRASTRUCTURE ONLY (see oracle.h).  A restatement, in plain C++, of the
// Rust reference's radiance loop.  Every function cites the reference lines it follows.
// Build with -O2 -ffp-contract=off -fno-fast-math: rustc never contracts a*b+c into an FMA and
// the baseline x86-64 target has none, so neither may this file.
//
// "parity unpinned": the Vec3 helpers restate glam 0.8.7 (Cargo.lock:285) and the OBJ reader
// restates tobj 3.2.2 (Cargo.lock:826) from their published behaviour; neither crate's source
// is available here and the reference has no test that would pin them, so bit-level agreement
// with the Rust binary is unproven.  At image level the restatement IS pinned, by the one output
// the reference ships: its screenshot.png (cornell_box.obj, 400 x 400, 400 spp) against this
// renderer's frame of the same job (tests/test_oracle_kat.py).

#include "oracle.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <limits>
#include <map>
#include <memory>
#include <sstream>
#include <string>
#include <thread>
#include <vector>

namespace {

// ---------------------------------------------------------------------------------------------
// glam 0.8.7 Vec3 (SSE2 backed): lane-wise IEEE f32 ops, no FMA.
// ---------------------------------------------------------------------------------------------
struct V3 {
    float x, y, z;
    float operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
    float& at(int i) { return i == 0 ? x : (i == 1 ? y : z); }
};
inline V3 v3(float x, float y, float z) { return V3{x, y, z}; }
inline V3 splat(float s) { return V3{s, s, s}; }
inline V3 operator+(V3 a, V3 b) { return V3{a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 operator-(V3 a, V3 b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 operator*(V3 a, V3 b) { return V3{a.x * b.x, a.y * b.y, a.z * b.z}; }
inline V3 operator*(V3 a, float s) { return V3{a.x * s, a.y * s, a.z * s}; }
inline V3 operator*(float s, V3 a) { return V3{s * a.x, s * a.y, s * a.z}; }
inline V3 operator/(V3 a, float s) { return V3{a.x / s, a.y / s, a.z / s}; }
// glam 0.8 Neg for the SSE2 Vec3 is 0 - v
inline V3 neg(V3 a) { return V3{0.0f - a.x, 0.0f - a.y, 0.0f - a.z}; }
// three products, then (x+y)+z
inline float dot(V3 a, V3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
inline float length_squared(V3 a) { return dot(a, a); }
inline V3 cross(V3 a, V3 b) {
    return V3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
// v / sqrt(dot(v,v)), lane-wise divide
inline V3 normalize(V3 a) { float l = std::sqrt(dot(a, a)); return V3{a.x / l, a.y / l, a.z / l}; }
// minps / maxps: second operand when the compare is false (also on NaN)
inline float minps(float a, float b) { return a < b ? a : b; }
inline float maxps(float a, float b) { return a > b ? a : b; }
inline V3 vmin(V3 a, V3 b) { return V3{minps(a.x, b.x), minps(a.y, b.y), minps(a.z, b.z)}; }
inline V3 vmax(V3 a, V3 b) { return V3{maxps(a.x, b.x), maxps(a.y, b.y), maxps(a.z, b.z)}; }

constexpr float F32_EPSILON = 1.1920929e-7f;   // f32::EPSILON
const float F32_INF = std::numeric_limits<float>::infinity();

// ---------------------------------------------------------------------------------------------
// glam 0.8.7 Mat4 (column-major, four Vec4 columns), as far as Mesh uses it (shape.rs:236-251,
// 303-307, camera.rs:137-142).  PARITY UNPINNED: restated from the crate's published source --
// transform_point3 / transform_vector3 = mul_vec4 of (v, 1) / (v, 0) truncated, with mul_vec4 =
// ((x_axis * x + y_axis * y) + z_axis * z) + w_axis * w, every step a rounded f32 operation (the
// crate's mul_add is a multiply and an add without the `fma` feature); inverse() is the cofactor
// expansion the crate ports from GLM.  Nothing in the reference (no test, no output: every shipped
// scene uses Mat4::identity()) can confirm the operation order.
// ---------------------------------------------------------------------------------------------
struct Mat4 {
    float m[16];                               // m[4 * column + row]
    static Mat4 identity() {
        Mat4 r{};
        r.m[0] = r.m[5] = r.m[10] = r.m[15] = 1.0f;
        return r;
    }
    bool is_identity() const {
        const Mat4 i = identity();
        return std::memcmp(m, i.m, sizeof(m)) == 0;
    }
    V3 mul_vec(V3 v, float w) const {
        float r[3];
        for (int k = 0; k < 3; k++) {
            float acc = m[k] * v.x;
            acc = m[4 + k] * v.y + acc;
            acc = m[8 + k] * v.z + acc;
            acc = m[12 + k] * w + acc;
            r[k] = acc;
        }
        return V3{r[0], r[1], r[2]};
    }
    V3 transform_point3(V3 v) const { return mul_vec(v, 1.0f); }
    V3 transform_vector3(V3 v) const { return mul_vec(v, 0.0f); }
    Mat4 inverse() const {
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
        const float vec2[4] = {m12, m02, m02, m02}, vec3_[4] = {m13, m03, m03, m03};
        const float sign_a[4] = {1.0f, -1.0f, 1.0f, -1.0f}, sign_b[4] = {-1.0f, 1.0f, -1.0f, 1.0f};
        float inv[4][4];
        for (int k = 0; k < 4; k++) {
            inv[0][k] = ((vec1[k] * fac0[k] - vec2[k] * fac1[k]) + vec3_[k] * fac2[k]) * sign_a[k];
            inv[1][k] = ((vec0[k] * fac0[k] - vec2[k] * fac3[k]) + vec3_[k] * fac4[k]) * sign_b[k];
            inv[2][k] = ((vec0[k] * fac1[k] - vec1[k] * fac3[k]) + vec3_[k] * fac5[k]) * sign_a[k];
            inv[3][k] = ((vec0[k] * fac2[k] - vec1[k] * fac4[k]) + vec2[k] * fac5[k]) * sign_b[k];
        }
        const float d0 = m00 * inv[0][0], d1 = m01 * inv[1][0], d2 = m02 * inv[2][0], d3 = m03 * inv[3][0];
        const float det = (d0 + d1) + (d2 + d3);
        const float rcp = 1.0f / det;
        Mat4 r;
        for (int c = 0; c < 4; c++)
            for (int k = 0; k < 4; k++) r.m[4 * c + k] = inv[c][k] * rcp;
        return r;
    }
};

// ---------------------------------------------------------------------------------------------
// math.rs:4-80  RandomSeries
// ---------------------------------------------------------------------------------------------
struct RandomSeries {
    uint32_t state;
    explicit RandomSeries(uint32_t seed) : state(seed) {}
    uint32_t xorshift32() {                    // math.rs:13-20
        uint32_t x = state;
        x ^= x << 13;
        x ^= x >> 17;
        x ^= x << 5;
        state = x;
        return x;
    }
    static float to01(uint32_t x) {            // math.rs:23
        return (float)((double)x / (double)4294967295u);
    }
    float random01() { return to01(xorshift32()); }
    float random(float lo, float hi) { return random01() * (hi - lo) + lo; }   // math.rs:28-30
    int32_t range_i32(int32_t lo, int32_t hi) {                                  // math.rs:32-39
        int32_t diff = hi - lo;
        if (diff > 0) return (int32_t)(xorshift32() % (uint32_t)diff) + lo;
        return lo;
    }
    V3 random_vec3() {                         // math.rs:52-58 (arguments evaluate left to right)
        float a = random(-1.0f, 1.0f);
        float b = random(-1.0f, 1.0f);
        float c = random(-1.0f, 1.0f);
        return v3(a, b, c);
    }
    V3 random_in_disk() {                      // math.rs:60-67
        for (;;) {
            float a = random(-1.0f, 1.0f);
            float b = random(-1.0f, 1.0f);
            V3 p = v3(a, b, 0.0f);
            if (length_squared(p) < 1.0f) return p;
        }
    }
    V3 random_in_unit_sphere() {               // math.rs:69-76
        for (;;) {
            V3 p = random_vec3();
            if (length_squared(p) < 1.0f) return p;
        }
    }
    V3 random_on_unit_sphere() { return normalize(random_in_unit_sphere()); }   // math.rs:78-80
};

// math.rs:94-110
inline V3 reflect(V3 v, V3 n) { return v - (2.0f * dot(v, n)) * n; }
inline V3 refract(V3 uv, V3 n, float etai_over_etat) {
    float cos_theta = std::fmin(dot(neg(uv), n), 1.0f);
    V3 r_out_perp = etai_over_etat * (uv + cos_theta * n);
    V3 r_out_parallel = (-std::sqrt(std::fabs(1.0f - length_squared(r_out_perp)))) * n;
    return r_out_perp + r_out_parallel;
}
inline float reflectance(float cosine, float ref_idx) {
    float r0 = (1.0f - ref_idx) / (1.0f + ref_idx);
    r0 = r0 * r0;
    return r0 + (1.0f - r0) * std::pow(1.0f - cosine, 5.0f);
}
// Rust `as u32`: saturating, NaN -> 0
inline uint32_t f32_as_u32(float f) {
    if (!(f == f)) return 0;
    if (f <= 0.0f) return 0;
    if (f >= 4294967296.0f) return 0xFFFFFFFFu;
    return (uint32_t)f;
}
inline uint32_t rgb_to_u32(V3 c) {             // math.rs:112-116
    return (f32_as_u32(255.999f * c.x) << 16) | (f32_as_u32(255.999f * c.y) << 8) |
           f32_as_u32(255.999f * c.z);
}

// ---------------------------------------------------------------------------------------------
// instrumentation the reference lacks
// ---------------------------------------------------------------------------------------------
struct Counters {
    uint64_t rays = 0, paths = 0, box = 0, tri = 0, sph = 0, hits = 0;
    void add(const Counters& o) {
        rays += o.rays; paths += o.paths; box += o.box; tri += o.tri; sph += o.sph; hits += o.hits;
    }
};
thread_local Counters tl_cnt;
struct RayRecorder { oracle_ray* out; size_t cap, n, seen; uint32_t stride; };
thread_local RayRecorder* tl_rec = nullptr;

// math.rs:118-180  Aabb
struct Aabb {
    V3 min, max;
    static Aabb infinity() { return Aabb{splat(F32_INF), splat(-F32_INF)}; }
    bool hit(V3 from, V3 dir, float t_min, float t_max) const {     // math.rs:149-165
        tl_cnt.box++;
        for (int axis = 0; axis < 3; axis++) {
            float inv_d = 1.0f / dir[axis];
            float t0 = (min[axis] - from[axis]) * inv_d;
            float t1 = (max[axis] - from[axis]) * inv_d;
            if (inv_d < 0.0f) std::swap(t0, t1);
            t_min = t0 > t_min ? t0 : t_min;
            t_max = t1 < t_max ? t1 : t_max;
            if (t_max <= t_min) return false;
        }
        return true;
    }
    Aabb surrounding(const Aabb& o) const { return Aabb{vmin(min, o.min), vmax(max, o.max)}; }
    Aabb surrounding_point(V3 p) const { return Aabb{vmin(min, p), vmax(max, p)}; }
};

// camera.rs:126-143
struct Ray {
    V3 origin, dir;
    V3 at(float t) const { return origin + dir * t; }
    Ray transform(const Mat4& t) const { return Ray{t.transform_point3(origin), t.transform_vector3(dir)}; }
};

// camera.rs:92-124, plus the primitive id and material index the reference does not expose
struct Hit {
    V3 p{0, 0, 0}, normal{0, 0, 0};
    float t = 0.0f;
    bool front_face = true;
    int material = -1;
    uint32_t prim_id = 0xFFFFFFFFu;
    void set_face_normal(const Ray& ray, V3 outward) {     // camera.rs:112-119
        front_face = dot(ray.dir, outward) < 0.0f;
        normal = front_face ? outward : neg(outward);
    }
};

// camera.rs:6-90
struct Camera {
    V3 look_from{0, 0, 0}, look_at{0, 0, -1}, vup{0, 1, 0};
    float vfov = 90, aperture = 0, focus_dist = 1;
    V3 origin{}, horizontal{}, vertical{}, lower_left_corner{}, w{}, u{}, v{};
    float lens_radius = 0;
    void setup(float aspect_ratio) {           // camera.rs:55-78
        float theta = vfov * 3.14159265358979323846f / 180.0f;
        float h = std::tan(theta / 2.0f);
        float viewport_height = 2.0f * h;
        float viewport_width = aspect_ratio * viewport_height;
        w = normalize(look_from - look_at);
        u = normalize(cross(vup, w));
        v = cross(w, u);
        origin = look_from;
        horizontal = (focus_dist * viewport_width) * u;
        vertical = (focus_dist * viewport_height) * v;
        lower_left_corner = origin - horizontal / 2.0f - vertical / 2.0f - focus_dist * w;
        lens_radius = aperture / 2.0f;
    }
    Ray get_ray(float s, float t, RandomSeries& r) const {   // camera.rs:80-89
        V3 rd = lens_radius * r.random_in_disk();
        V3 offset = u * rd.x + v * rd.y;
        return Ray{origin + offset,
                   lower_left_corner + s * horizontal + t * vertical - origin - offset};
    }
};

// ---------------------------------------------------------------------------------------------
// material.rs
// ---------------------------------------------------------------------------------------------
struct Material {
    uint32_t kind;
    V3 rgb;        // albedo or emitted colour
    float param;   // fuzz or ior
    bool normal_shading;   // LambertianMaterial.normal
    bool scatter(const Ray& ray_in, const Hit& hit, V3& attenuation, Ray& ray_out,
                 RandomSeries& series) const {
        switch (kind) {
        case ORACLE_LAMBERT: {                 // material.rs:27-56
            V3 d = hit.normal + series.random_on_unit_sphere();
            if (std::fabs(d.x) + std::fabs(d.y) + std::fabs(d.z) < F32_EPSILON) d = hit.normal;
            ray_out = Ray{hit.p, normalize(d)};
            attenuation = normal_shading ? (hit.normal + v3(1.0f, 1.0f, 1.0f)) * 0.5f : rgb;
            return true;
        }
        case ORACLE_METAL: {                   // material.rs:63-80
            V3 d = reflect(ray_in.dir, hit.normal);
            ray_out = Ray{hit.p, d + param * series.random_in_unit_sphere()};
            attenuation = rgb;
            return dot(d, hit.normal) > 0.0f;
        }
        case ORACLE_DIELECTRIC: {              // material.rs:87-120
            attenuation = rgb;
            float refraction_ratio = hit.front_face ? 1.0f / param : param;
            V3 dir = normalize(ray_in.dir);
            float cos_theta = std::fmin(dot(neg(dir), hit.normal), 1.0f);
            float sin_theta = std::sqrt(1.0f - cos_theta * cos_theta);
            bool cannot_refract = (refraction_ratio * sin_theta) > 1.0f;
            V3 direction;
            if (cannot_refract || reflectance(cos_theta, refraction_ratio) > series.random01())
                direction = reflect(dir, hit.normal);
            else
                direction = refract(dir, hit.normal, refraction_ratio);
            ray_out = Ray{hit.p, direction};
            return true;
        }
        default:                               // material.rs:13-15 (Emissive keeps the default)
            return false;
        }
    }
    V3 emit() const { return kind == ORACLE_EMISSIVE ? rgb : v3(0, 0, 0); }   // material.rs:17-19,126-130
};

// ---------------------------------------------------------------------------------------------
// shape.rs
// ---------------------------------------------------------------------------------------------
struct Hittable {
    virtual ~Hittable() {}
    virtual bool hit(const Ray& ray, float t_min, float t_max, Hit& hit) const = 0;
    virtual Aabb get_aabb() const = 0;
};

struct Sphere : Hittable {                     // shape.rs:11-53
    float radius; V3 center; int material; uint32_t prim_id;
    bool hit(const Ray& ray, float t_min, float t_max, Hit& hit) const override {
        tl_cnt.sph++;
        V3 oc = ray.origin - center;
        float a = length_squared(ray.dir);
        float half_b = dot(oc, ray.dir);
        float c = length_squared(oc) - radius * radius;
        float discriminant = half_b * half_b - a * c;
        if (discriminant < 0.0f) return false;
        float sqrtd = std::sqrt(discriminant);
        float root = (-half_b - sqrtd) / a;
        if (root < t_min || t_max < root) {
            root = (-half_b + sqrtd) / a;
            if (root < t_min || t_max < root) return false;
        }
        hit.t = root;
        hit.p = ray.at(hit.t);
        V3 outward_normal = (hit.p - center) / radius;
        hit.set_face_normal(ray, outward_normal);
        hit.material = material;
        hit.prim_id = prim_id;
        return true;
    }
    Aabb get_aabb() const override { return Aabb{center - splat(radius), center + splat(radius)}; }
};

struct HittableList : Hittable {               // shape.rs:55-85
    std::vector<const Hittable*> objects;
    bool hit(const Ray& ray, float t_min, float t_max, Hit& hit) const override {
        Hit tmp;
        bool hit_anything = false;
        float closest_so_far = t_max;
        for (const Hittable* o : objects) {
            if (o->hit(ray, t_min, closest_so_far, tmp)) {
                hit_anything = true;
                closest_so_far = tmp.t;
                hit = tmp;
            }
        }
        return hit_anything;
    }
    Aabb get_aabb() const override {
        Aabb acc = Aabb::infinity();
        for (const Hittable* o : objects) acc = o->get_aabb().surrounding(acc);
        return acc;
    }
};

struct Bvh : Hittable {                        // shape.rs:87-163
    const Hittable* left = nullptr;
    const Hittable* right = nullptr;
    std::unique_ptr<Bvh> own_left, own_right;
    Aabb aabb{};
    Bvh(std::vector<const Hittable*> objects, int axis) {     // shape.rs:94-138
        size_t n = objects.size();
        if (n == 0) {
            aabb = Aabb{v3(0, 0, 0), v3(0, 0, 0)};
        } else if (n == 1) {
            left = objects.back();
            aabb = left->get_aabb();
        } else if (n == 2) {
            left = objects.back(); objects.pop_back();
            right = objects.back(); objects.pop_back();
            aabb = left->get_aabb().surrounding(right->get_aabb());
        } else {
            // slice::sort_by is stable
            std::stable_sort(objects.begin(), objects.end(),
                             [axis](const Hittable* a, const Hittable* b) {
                                 return a->get_aabb().min[axis] < b->get_aabb().min[axis];
                             });
            size_t half = n / 2;               // split_off(len/2): left keeps the first half
            std::vector<const Hittable*> l(objects.begin(), objects.begin() + half);
            std::vector<const Hittable*> r(objects.begin() + half, objects.end());
            own_left.reset(new Bvh(std::move(l), (axis + 1) % 3));
            own_right.reset(new Bvh(std::move(r), (axis + 1) % 3));
            left = own_left.get();
            right = own_right.get();
            aabb = left->get_aabb().surrounding(right->get_aabb());
        }
    }
    bool hit(const Ray& ray, float t_min, float t_max, Hit& hit) const override {   // shape.rs:142-158
        if (!aabb.hit(ray.origin, ray.dir, t_min, t_max)) return false;
        bool hit_left = left ? left->hit(ray, t_min, t_max, hit) : false;
        bool hit_right = right ? right->hit(ray, t_min, hit_left ? hit.t : t_max, hit) : false;
        return hit_left || hit_right;
    }
    Aabb get_aabb() const override { return aabb; }
};

struct Triangle {                              // shape.rs:165-200 (only v1.n is ever read, shape.rs:298)
    V3 p1, p2, p3, n1;
    int material_index;
    uint32_t prim_id;
    Aabb get_aabb() const { return Aabb{p1, p1}.surrounding_point(p2).surrounding_point(p3); }
};

// shape.rs:411-450
inline bool ray_triangle_intersection(const Ray& ray, V3 v0, V3 v1, V3 v2, bool& backface, float& t,
                                      float& u, float& v) {
    tl_cnt.tri++;
    V3 v0v1 = v1 - v0;
    V3 v0v2 = v2 - v0;
    V3 pvec = cross(ray.dir, v0v2);
    float det = dot(v0v1, pvec);
    if (std::fabs(det) < F32_EPSILON) return false;
    backface = det < F32_EPSILON;
    float inv_det = 1.0f / det;
    V3 tvec = ray.origin - v0;
    u = dot(tvec, pvec) * inv_det;
    if (u < 0.0f || u > 1.0f) return false;
    V3 qvec = cross(tvec, v0v1);
    v = dot(ray.dir, qvec) * inv_det;
    if (v < 0.0f || u + v > 1.0f) return false;
    t = dot(v0v2, qvec) * inv_det;
    return true;
}

struct Mesh : Hittable {                       // shape.rs:209-324, 452-494
    enum Kind { TRI, NODE, NONE };
    struct Node { Kind kind; Triangle tri; size_t left, right; Aabb aabb; };
    std::vector<Triangle> triangles;
    std::vector<Node> bvh_nodes;
    Aabb aabb;
    // A mesh whose transform is bit for bit Mat4::identity() (every shipped scene) skips the
    // transform arithmetic: for finite vectors it changes nothing except that a -0.0 component
    // may come out as +0.0, which can flip the sign of inv_d (math.rs:152-156) for a ray with a
    // negative-zero direction component; hit parity is not claimed for such rays.
    Mat4 transform = Mat4::identity(), inverse_transform = Mat4::identity();
    bool transformed = false;

    static Aabb compute_aabb(const std::vector<Triangle>& tris) {      // shape.rs:218-233
        Aabb a = Aabb::infinity();
        for (const Triangle& t : tris)
            a = a.surrounding_point(t.p1).surrounding_point(t.p2).surrounding_point(t.p3);
        a.max.x += 0.0001f; a.max.y += 0.0001f; a.max.z += 0.0001f;
        a.min.x -= 0.0001f; a.min.y -= 0.0001f; a.min.z -= 0.0001f;
        return a;
    }
    explicit Mesh(std::vector<Triangle> tris, const Mat4& xf = Mat4::identity()) : triangles(std::move(tris)) {   // shape.rs:236-257
        transformed = !xf.is_identity();
        if (transformed) { transform = xf; inverse_transform = xf.inverse(); }
        aabb = Aabb::infinity();
        for (const Triangle& t : triangles)
            aabb = transformed ? aabb.surrounding_point(transform.transform_point3(t.p1))
                                     .surrounding_point(transform.transform_point3(t.p2))
                                     .surrounding_point(transform.transform_point3(t.p3))
                               : aabb.surrounding_point(t.p1).surrounding_point(t.p2).surrounding_point(t.p3);
        bvh_nodes.reserve(triangles.size() * 2 + 1);
        build_bvh(triangles, 0);
    }
    size_t make_triangle_node(const Triangle& t) {
        bvh_nodes.push_back(Node{TRI, t, 0, 0, Aabb{}});
        return bvh_nodes.size() - 1;
    }
    size_t build_bvh(std::vector<Triangle> tris, int axis) {           // shape.rs:264-290
        if (tris.empty()) {
            bvh_nodes.push_back(Node{NONE, Triangle{}, 0, 0, Aabb{}});
            return bvh_nodes.size() - 1;
        } else if (tris.size() == 1) {
            return make_triangle_node(tris.back());
        } else if (tris.size() == 2) {
            Aabb box = compute_aabb(tris);
            size_t l = make_triangle_node(tris.back()); tris.pop_back();
            size_t r = make_triangle_node(tris.back()); tris.pop_back();
            bvh_nodes.push_back(Node{NODE, Triangle{}, l, r, box});
            return bvh_nodes.size() - 1;
        } else {
            std::stable_sort(tris.begin(), tris.end(), [axis](const Triangle& a, const Triangle& b) {
                return a.get_aabb().min[axis] < b.get_aabb().min[axis];
            });
            Aabb box = compute_aabb(tris);
            size_t half = tris.size() / 2;
            std::vector<Triangle> right_tris(tris.begin() + half, tris.end());
            tris.resize(half);
            size_t l = build_bvh(std::move(tris), (axis + 1) % 3);
            size_t r = build_bvh(std::move(right_tris), (axis + 1) % 3);
            bvh_nodes.push_back(Node{NODE, Triangle{}, l, r, box});
            return bvh_nodes.size() - 1;
        }
    }
    bool bvh_hit(size_t node_index, const Ray& ray, float t_min, float t_max, Hit& hit) const {   // shape.rs:292-324
        const Node& node = bvh_nodes[node_index];
        if (node.kind == TRI) {
            const Triangle& tr = node.tri;
            bool backface = false; float t = 0, u = 0, v = 0;
            bool is_hit = ray_triangle_intersection(ray, tr.p1, tr.p2, tr.p3, backface, t, u, v);
            if (is_hit && t >= t_min && t < t_max) {
                V3 hit_normal = tr.n1;
                if (backface) hit_normal = hit_normal * -1.0f;
                hit.t = t;
                // shape.rs:303-307: the point and the normal go through the transform; the facing test
                // uses the LOCAL ray (the parameter shadows the caller's) against the WORLD normal
                hit.p = transformed ? transform.transform_point3(ray.at(t)) : ray.at(t);
                hit.set_face_normal(ray, normalize(transformed ? transform.transform_vector3(hit_normal) : hit_normal));
                hit.material = tr.material_index;
                hit.prim_id = tr.prim_id;
                return true;
            }
            return false;
        } else if (node.kind == NODE) {
            if (!node.aabb.hit(ray.origin, ray.dir, t_min, t_max)) return false;
            bool hit_left = bvh_hit(node.left, ray, t_min, t_max, hit);
            bool hit_right = bvh_hit(node.right, ray, t_min, hit_left ? hit.t : t_max, hit);
            return hit_left || hit_right;
        }
        return false;
    }
    bool hit(const Ray& ray, float t_min, float t_max, Hit& hit) const override {   // shape.rs:453-455
        if (transformed) return bvh_hit(bvh_nodes.size() - 1, ray.transform(inverse_transform), t_min, t_max, hit);
        return bvh_hit(bvh_nodes.size() - 1, ray, t_min, t_max, hit);
    }
    Aabb get_aabb() const override { return aabb; }
};

// ---------------------------------------------------------------------------------------------
// OBJ ingest: tobj::load_obj(single_index, triangulate) as used at shape.rs:352-363, then the
// per-model expansion of shape.rs:372-403.  material_names maps MTL material names to indices
// of the caller's material table; a name that is not in the map gets index 0 (shape.rs:377-381).
// ---------------------------------------------------------------------------------------------
struct ObjModel { std::string name; int material_id; std::vector<int> vi, ni; /* 3 per triangle */ };

bool parse_face_vertex(const std::string& tok, int nv, int nn, int& vi, int& ni) {
    // v, v/vt, v//vn, v/vt/vn ; negative = relative to the end
    int idx[3] = {0, 0, 0};
    bool have[3] = {false, false, false};
    size_t pos = 0; int k = 0;
    while (k < 3 && pos <= tok.size()) {
        size_t slash = tok.find('/', pos);
        std::string part = tok.substr(pos, slash == std::string::npos ? std::string::npos : slash - pos);
        if (!part.empty()) { idx[k] = std::atoi(part.c_str()); have[k] = true; }
        k++;
        if (slash == std::string::npos) break;
        pos = slash + 1;
    }
    if (!have[0]) return false;
    vi = idx[0] > 0 ? idx[0] - 1 : nv + idx[0];
    ni = have[2] ? (idx[2] > 0 ? idx[2] - 1 : nn + idx[2]) : -1;
    return true;
}

bool load_obj(const std::string& path, std::vector<V3>& pos, std::vector<V3>& nrm,
              std::vector<ObjModel>& models, std::vector<std::string>& mtl_names, size_t* n_faces) {
    std::ifstream in(path);
    if (!in) return false;
    std::string dir;
    size_t slash = path.find_last_of('/');
    if (slash != std::string::npos) dir = path.substr(0, slash + 1);
    std::map<std::string, int> mat_map;
    ObjModel cur; cur.name = "unnamed_object"; cur.material_id = -1;
    size_t faces = 0;
    std::string line;
    auto flush = [&]() {
        if (!cur.vi.empty()) models.push_back(cur);
        cur.vi.clear(); cur.ni.clear();
    };
    while (std::getline(in, line)) {
        if (!line.empty() && line.back() == '\r') line.pop_back();
        std::istringstream ss(line);
        std::string key;
        if (!(ss >> key)) continue;
        if (key == "v") {
            float x, y, z; ss >> x >> y >> z; pos.push_back(v3(x, y, z));
        } else if (key == "vn") {
            float x, y, z; ss >> x >> y >> z; nrm.push_back(v3(x, y, z));
        } else if (key == "f") {
            std::vector<int> fv, fn;
            std::string tok;
            while (ss >> tok) {
                int vi, ni;
                if (!parse_face_vertex(tok, (int)pos.size(), (int)nrm.size(), vi, ni)) return false;
                fv.push_back(vi); fn.push_back(ni);
            }
            if (fv.size() < 3) continue;       // points and lines are ignored
            faces++;
            // fan from the first vertex: (0,1,2), (0,2,3), ...
            for (size_t i = 1; i + 1 < fv.size(); i++) {
                cur.vi.push_back(fv[0]); cur.vi.push_back(fv[i]); cur.vi.push_back(fv[i + 1]);
                cur.ni.push_back(fn[0]); cur.ni.push_back(fn[i]); cur.ni.push_back(fn[i + 1]);
            }
        } else if (key == "o" || key == "g") {
            flush();
            std::string name; std::getline(ss, name);
            size_t b = name.find_first_not_of(" \t");
            cur.name = b == std::string::npos ? "" : name.substr(b);
        } else if (key == "usemtl") {
            std::string name; ss >> name;
            auto it = mat_map.find(name);
            int new_mat = it == mat_map.end() ? -1 : it->second;
            if (new_mat != cur.material_id && !cur.vi.empty()) flush();
            cur.material_id = new_mat;
        } else if (key == "mtllib") {
            std::string name; ss >> name;
            std::ifstream mtl(dir + name);
            std::string ml;
            while (std::getline(mtl, ml)) {
                std::istringstream ms(ml);
                std::string mk;
                if ((ms >> mk) && mk == "newmtl") {
                    std::string mname; ms >> mname;
                    mat_map[mname] = (int)mtl_names.size();
                    mtl_names.push_back(mname);
                }
            }
        }
    }
    flush();
    if (n_faces) *n_faces = faces;
    return true;
}

}  // namespace

// ---------------------------------------------------------------------------------------------
// scene container
// ---------------------------------------------------------------------------------------------
struct oracle_scene {
    std::vector<Material> materials;
    std::vector<std::unique_ptr<Hittable>> objects;    // insertion order
    std::vector<int> object_kind;                      // 0 sphere, 1 mesh
    std::unique_ptr<Hittable> root_owner;
    const Hittable* root = nullptr;
    int root_kind = ORACLE_ROOT_BVH;
    // Containers inside the object list (a Bvh or HittableList whose children are further
    // objects, shape.rs:55-57, 87-91): items >= 0 name objects[], items < 0 name groups[-1 - item].
    // groups[0] is the top level (World.hittable's children).
    struct Group { int kind; int axis; std::vector<long> items; };
    std::vector<Group> groups{Group{ORACLE_ROOT_BVH, 0, {}}};
    std::vector<size_t> open_groups{0};
    std::vector<std::unique_ptr<Hittable>> containers;
    std::vector<Mat4> mesh_xforms;                     // per object (identity for spheres)
    Camera camera;
    float sky_intensity = 1.0f;
    uint32_t next_prim = 0;
    size_t n_spheres = 0, n_tris = 0;
};

namespace {

Material make_material(const oracle_material& m) {
    return Material{m.kind, v3(m.rgb[0], m.rgb[1], m.rgb[2]), m.param, (m.flags & 1u) != 0};
}

void add_sphere(oracle_scene* s, V3 c, float r, int mat) {
    auto sp = std::make_unique<Sphere>();
    sp->center = c; sp->radius = r; sp->material = mat; sp->prim_id = s->next_prim++;
    s->groups[s->open_groups.back()].items.push_back((long)s->objects.size());
    s->objects.push_back(std::move(sp));
    s->object_kind.push_back(0);
    s->mesh_xforms.push_back(Mat4::identity());
    s->n_spheres++;
}

void add_mesh(oracle_scene* s, std::vector<Triangle> tris, const Mat4& xf = Mat4::identity()) {
    for (Triangle& t : tris) t.prim_id = s->next_prim++;
    s->n_tris += tris.size();
    s->groups[s->open_groups.back()].items.push_back((long)s->objects.size());
    s->objects.push_back(std::make_unique<Mesh>(std::move(tris), xf));
    s->object_kind.push_back(1);
    s->mesh_xforms.push_back(xf);
}

// shape.rs:347-406
bool mesh_from_file(oracle_scene* s, const std::string& path, const std::map<std::string, int>& material_map) {
    std::vector<V3> pos, nrm;
    std::vector<ObjModel> models;
    std::vector<std::string> mtl_names;
    if (!load_obj(path, pos, nrm, models, mtl_names, nullptr)) return false;
    if (mtl_names.empty()) return false;       // "Failed to load materials"
    std::vector<Triangle> tris;
    for (const ObjModel& m : models) {
        const std::string& mname = mtl_names[m.material_id < 0 ? 0 : m.material_id];
        auto it = material_map.find(mname);
        int material_index = it == material_map.end() ? 0 : it->second;
        for (size_t k = 0; k + 2 < m.vi.size(); k += 3) {
            if (m.ni[k] < 0) return false;     // the reference indexes mesh.normals and would panic
            Triangle t;
            t.p1 = pos[m.vi[k]]; t.p2 = pos[m.vi[k + 1]]; t.p3 = pos[m.vi[k + 2]];
            t.n1 = nrm[m.ni[k]];
            t.material_index = material_index;
            t.prim_id = 0;
            tris.push_back(t);
        }
    }
    add_mesh(s, std::move(tris));
    return true;
}

// the Hittable a group's item stands for; inner groups are built first (Bvh::new / HittableList of
// their own children), exactly as a caller nests the reference's constructors
const Hittable* build_item(oracle_scene* s, long item) {
    if (item >= 0) return s->objects[(size_t)item].get();
    const oracle_scene::Group& g = s->groups[(size_t)(-1 - item)];
    std::vector<const Hittable*> kids;
    for (long it : g.items) kids.push_back(build_item(s, it));
    if (g.kind == ORACLE_ROOT_LIST) {
        auto l = std::make_unique<HittableList>();
        l->objects = kids;
        s->containers.push_back(std::move(l));
    } else {
        s->containers.push_back(std::make_unique<Bvh>(kids, g.axis));
    }
    return s->containers.back().get();
}

void commit(oracle_scene* s) {
    std::vector<const Hittable*> objs;
    s->containers.clear();
    for (long it : s->groups[0].items) objs.push_back(build_item(s, it));
    s->root_owner.reset();
    if (s->root_kind == ORACLE_ROOT_SINGLE && objs.size() == 1) {
        s->root = objs[0];
    } else if (s->root_kind == ORACLE_ROOT_LIST) {
        auto l = std::make_unique<HittableList>();
        l->objects = objs;
        s->root = l.get();
        s->root_owner = std::move(l);
    } else {
        auto b = std::make_unique<Bvh>(objs, 0);
        s->root = b.get();
        s->root_owner = std::move(b);
    }
}

V3 sky_color(const Ray& ray, float sky_intensity) {        // main.rs:31-38
    V3 dir = normalize(ray.dir);
    float t = 0.5f * (dir.y + 1.0f);
    V3 white = v3(1.0f, 1.0f, 1.0f);
    V3 blue = v3(0.5f, 0.7f, 1.0f);
    return ((1.0f - t) * white + t * blue) * sky_intensity;
}

V3 ray_color(const Ray& ray, const oracle_scene& world, RandomSeries& series, uint32_t depth,
             uint32_t max_bounces) {                        // main.rs:40-67
    if (depth >= max_bounces) return v3(0, 0, 0);
    tl_cnt.rays++;
    if (tl_rec) {
        RayRecorder* r = tl_rec;
        if (r->seen % r->stride == 0 && r->n < r->cap) {
            oracle_ray& o = r->out[r->n++];
            o.o[0] = ray.origin.x; o.o[1] = ray.origin.y; o.o[2] = ray.origin.z;
            o.d[0] = ray.dir.x; o.d[1] = ray.dir.y; o.d[2] = ray.dir.z;
        }
        r->seen++;
    }
    Hit hit;
    if (world.root->hit(ray, 0.001f, F32_INF, hit)) {
        tl_cnt.hits++;
        Ray bounce{v3(0, 0, 0), v3(0, 0, 0)};
        const Material& m = world.materials[hit.material];
        V3 emitted = m.emit();
        V3 attenuation = v3(0, 0, 0);
        if (m.scatter(ray, hit, attenuation, bounce, series))
            return emitted + attenuation * ray_color(bounce, world, series, depth + 1, max_bounces);
        return emitted;
    }
    return sky_color(ray, world.sky_intensity);
}

// one tile job, main.rs:336-413 (the red in-progress border is display only and not restated)
void render_tile(const oracle_scene& world, uint32_t width, uint32_t height, uint32_t spp,
                 uint32_t max_bounces, uint32_t block, uint32_t i, uint32_t j, uint32_t seed_offset,
                 float* accum_rgb, uint32_t* image) {
    uint32_t bw = (i + 1) * block > width ? block - ((i + 1) * block - width) : block;
    uint32_t bh = (j + 1) * block > height ? block - ((j + 1) * block - height) : block;
    float one_over_width = 1.0f / (float)(width - 1);
    float one_over_height = 1.0f / (float)(height - 1);
    RandomSeries series(j + 1 + seed_offset);
    std::vector<V3> acc((size_t)bw * bh, v3(0, 0, 0));
    std::vector<uint32_t> blockpx((size_t)bw * bh, 0);
    uint32_t ix = i * block, iy = j * block;
    uint32_t sample_count = 0;
    uint32_t next_sample_count = std::min<uint32_t>(10, spp);
    for (;;) {
        for (uint32_t y = 0; y < bh; y++) {
            for (uint32_t x = 0; x < bw; x++) {
                size_t pixel_index = (size_t)y * bw + x;
                for (uint32_t sidx = sample_count; sidx < next_sample_count; sidx++) {
                    float u = ((float)(ix + x) + series.random01()) * one_over_width;
                    float v = ((float)(iy + y) + series.random01()) * one_over_height;
                    Ray ray = world.camera.get_ray(u, 1.0f - v, series);
                    tl_cnt.paths++;
                    acc[pixel_index] = acc[pixel_index] + ray_color(ray, world, series, 0, max_bounces);
                }
                float scale = 1.0f / (float)next_sample_count;
                V3 c = acc[pixel_index] * scale;
                V3 pc = vmin(v3(std::sqrt(c.x), std::sqrt(c.y), std::sqrt(c.z)), v3(1, 1, 1));
                blockpx[pixel_index] = rgb_to_u32(pc);
            }
        }
        if (next_sample_count >= spp) break;
        sample_count = next_sample_count;
        next_sample_count = std::min(next_sample_count * 2, spp);
    }
    for (uint32_t y = 0; y < bh; y++)
        for (uint32_t x = 0; x < bw; x++) {
            size_t src = (size_t)y * bw + x;
            size_t dst = (size_t)(iy + y) * width + (ix + x);
            if (accum_rgb) {
                accum_rgb[dst * 3 + 0] = acc[src].x;
                accum_rgb[dst * 3 + 1] = acc[src].y;
                accum_rgb[dst * 3 + 2] = acc[src].z;
            }
            if (image) image[dst] = blockpx[src];
        }
}

void fill_stats(oracle_stats* st, const Counters& c, double seconds) {
    if (!st) return;
    st->rays = c.rays; st->paths = c.paths; st->box_tests = c.box; st->tri_tests = c.tri;
    st->sph_tests = c.sph; st->hits = c.hits; st->seconds = seconds;
}

double now_s() {
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

Camera make_camera(V3 from, V3 at, V3 up, float vfov, float aperture, float focus) {
    Camera c; c.look_from = from; c.look_at = at; c.vup = up; c.vfov = vfov; c.aperture = aperture;
    c.focus_dist = focus;
    return c;
}

uint32_t g_world1_rng_state = 0;

}  // namespace

// ---------------------------------------------------------------------------------------------
// C API
// ---------------------------------------------------------------------------------------------
extern "C" {

oracle_scene* oracle_scene_new(void) { return new oracle_scene(); }
void oracle_scene_free(oracle_scene* s) { delete s; }

void oracle_scene_set_materials(oracle_scene* s, const oracle_material* m, size_t n) {
    s->materials.clear();
    for (size_t i = 0; i < n; i++) s->materials.push_back(make_material(m[i]));
}
void oracle_scene_add_spheres(oracle_scene* s, const oracle_sphere* sp, size_t n) {
    for (size_t i = 0; i < n; i++)
        add_sphere(s, v3(sp[i].c[0], sp[i].c[1], sp[i].c[2]), sp[i].r, (int)sp[i].mat);
}
void oracle_scene_add_mesh(oracle_scene* s, const oracle_triangle* t, size_t n) {
    std::vector<Triangle> tris(n);
    for (size_t i = 0; i < n; i++) {
        tris[i].p1 = v3(t[i].p[0][0], t[i].p[0][1], t[i].p[0][2]);
        tris[i].p2 = v3(t[i].p[1][0], t[i].p[1][1], t[i].p[1][2]);
        tris[i].p3 = v3(t[i].p[2][0], t[i].p[2][1], t[i].p[2][2]);
        tris[i].n1 = v3(t[i].n0[0], t[i].n0[1], t[i].n0[2]);
        tris[i].material_index = (int)t[i].mat;
        tris[i].prim_id = 0;
    }
    add_mesh(s, std::move(tris));
}
void oracle_scene_add_mesh_xf(oracle_scene* s, const oracle_triangle* t, size_t n, const float xform[16]) {
    std::vector<Triangle> tris(n);
    for (size_t i = 0; i < n; i++) {
        tris[i].p1 = v3(t[i].p[0][0], t[i].p[0][1], t[i].p[0][2]);
        tris[i].p2 = v3(t[i].p[1][0], t[i].p[1][1], t[i].p[1][2]);
        tris[i].p3 = v3(t[i].p[2][0], t[i].p[2][1], t[i].p[2][2]);
        tris[i].n1 = v3(t[i].n0[0], t[i].n0[1], t[i].n0[2]);
        tris[i].material_index = (int)t[i].mat;
        tris[i].prim_id = 0;
    }
    Mat4 m = Mat4::identity();
    if (xform) std::memcpy(m.m, xform, sizeof(m.m));
    add_mesh(s, std::move(tris), m);
}
void oracle_scene_begin_group(oracle_scene* s, int kind, int axis) {
    s->groups.push_back(oracle_scene::Group{kind, axis, {}});
    s->groups[s->open_groups.back()].items.push_back(-1 - (long)(s->groups.size() - 1));
    s->open_groups.push_back(s->groups.size() - 1);
}
void oracle_scene_end_group(oracle_scene* s) {
    if (s->open_groups.size() > 1) s->open_groups.pop_back();
}
void oracle_mat4_inverse(const float m[16], float out[16]) {
    Mat4 a;
    std::memcpy(a.m, m, sizeof(a.m));
    const Mat4 r = a.inverse();
    std::memcpy(out, r.m, sizeof(r.m));
}
void oracle_mat4_transform(const float m[16], const float v[3], float w, float out[3]) {
    Mat4 a;
    std::memcpy(a.m, m, sizeof(a.m));
    const V3 r = a.mul_vec(v3(v[0], v[1], v[2]), w);
    out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
void oracle_scene_set_root(oracle_scene* s, int k) { s->root_kind = k; }
void oracle_scene_set_camera(oracle_scene* s, const float f[3], const float a[3], const float u[3],
                             float vfov, float aperture, float focus) {
    s->camera = make_camera(v3(f[0], f[1], f[2]), v3(a[0], a[1], a[2]), v3(u[0], u[1], u[2]), vfov,
                            aperture, focus);
}
void oracle_scene_set_sky(oracle_scene* s, float sky) { s->sky_intensity = sky; }
int oracle_scene_commit(oracle_scene* s) { commit(s); return 0; }

oracle_scene* oracle_create_world1(void) {     // main.rs:75-176
    oracle_scene* s = new oracle_scene();
    // indices: 0 center, 1 ground, 2 left, 3 right, 4 glass, 5 glow
    s->materials = {
        Material{ORACLE_LAMBERT, v3(1, 1, 1), 0.0f, true},
        Material{ORACLE_LAMBERT, v3(0.13f, 0.8f, 0.13f), 0.0f, false},
        Material{ORACLE_METAL, splat(0.8f), 0.3f, false},
        Material{ORACLE_METAL, splat(0.4f), 1.0f, false},
        Material{ORACLE_DIELECTRIC, splat(1.0f), 1.3f, false},
        Material{ORACLE_EMISSIVE, splat(10.0f), 0.0f, false},
    };
    add_sphere(s, v3(0.0f, -1.0f, -200.5f), 200.0f, 1);
    add_sphere(s, v3(-1.0f, -1.0f, 0.0f), 0.5f, 4);
    add_sphere(s, v3(1.0f, -1.0f, 0.0f), 0.5f, 3);
    add_sphere(s, v3(0.0f, -1.0f, 0.0f), 0.5f, 0);
    add_sphere(s, v3(-0.4f, -1.3f, -0.5f + 0.1f), 0.1f, 3);
    add_sphere(s, v3(0.4f, -1.3f, -0.5f + 0.1f), 0.1f, 2);
    RandomSeries r(10);
    const int pick[4] = {0, 2, 3, 4};          // [center, left, right, glass], main.rs:134-139
    for (int i = 0; i < 500; i++) {
        float x = r.random(-5.0f, 5.0f);
        float y = r.random(-5.0f, 5.0f);
        float z = r.random(-0.25f, 5.0f);
        int m = pick[r.range_i32(0, 4)];
        add_sphere(s, v3(x, y, z), 0.25f, m);
    }
    g_world1_rng_state = r.state;
    add_sphere(s, v3(0.0f, 0.0f, 50.0f), 12.0f, 5);
    s->camera = make_camera(v3(2.0f, -3.0f, 1.0f), v3(0.0f, 1.0f, 0.0f), v3(0, 0, 1), 60.0f, 0.0f, 10.0f);
    s->sky_intensity = 0.02f;
    s->root_kind = ORACLE_ROOT_BVH;
    commit(s);
    return s;
}

oracle_scene* oracle_create_world2(const char* thing_obj, int variant) {   // main.rs:178-241
    oracle_scene* s = new oracle_scene();
    std::map<std::string, int> mm;
    if (variant == 0 || variant == 2) {
        s->materials = {Material{ORACLE_LAMBERT, v3(1, 1, 1), 0.0f, false},
                        Material{ORACLE_EMISSIVE, splat(4.0f), 0.0f, false}};
        mm["default"] = 0;
        if (variant == 2) {                    // the two planes main.rs:195-204 keeps in a comment
            for (float z : {-0.5f, 1.0f}) {    // Mesh::plane(2.0, 2.0, material1, Mat4::from_translation(..)), shape.rs:326-345
                const float w = 2.0f, h = 2.0f;
                Triangle a, b;
                a.p1 = v3(-w, -h, 0); a.p2 = v3(-w, h, 0); a.p3 = v3(w, -h, 0);
                b.p1 = v3(w, -h, 0); b.p2 = v3(-w, h, 0); b.p3 = v3(w, h, 0);
                a.n1 = b.n1 = v3(0, 0, 1);
                a.material_index = b.material_index = 0;
                a.prim_id = b.prim_id = 0;
                Mat4 m = Mat4::identity();
                m.m[12] = 0.0f; m.m[13] = 0.0f; m.m[14] = z;
                add_mesh(s, {a, b}, m);
            }
        }
    } else {
        // the "glass + diffuse" material map: same geometry, MTL names mapped explicitly
        s->materials = {Material{ORACLE_LAMBERT, splat(0.73f), 0.0f, false},
                        Material{ORACLE_EMISSIVE, splat(4.0f), 0.0f, false},
                        Material{ORACLE_DIELECTRIC, v3(1, 1, 1), 1.5f, false}};
        mm["None"] = 0;
        mm["Material"] = 2;
    }
    if (!mesh_from_file(s, thing_obj, mm)) { delete s; return nullptr; }
    add_sphere(s, v3(0.0f, 0.0f, 1.5f), 0.65f, 1);
    s->camera = make_camera(v3(0.0f, -2.0f, 0.0f), v3(0.0f, 1.0f, 0.0f), v3(0, 0, 1), 88.0f, 0.001f, 1.0f);
    s->sky_intensity = 1.5f;
    s->root_kind = ORACLE_ROOT_BVH;
    commit(s);
    return s;
}

oracle_scene* oracle_create_cornell_box(const char* obj_path) {   // main.rs:243-317
    oracle_scene* s = new oracle_scene();
    s->materials = {
        Material{ORACLE_LAMBERT, v3(0.5f, 0.5f, 0.5f), 0.0f, false},       // default
        Material{ORACLE_LAMBERT, v3(0.65f, 0.05f, 0.05f), 0.0f, false},    // red
        Material{ORACLE_LAMBERT, v3(0.12f, 0.45f, 0.15f), 0.0f, false},    // green
        Material{ORACLE_LAMBERT, v3(0.73f, 0.73f, 0.73f), 0.0f, false},    // white
        Material{ORACLE_EMISSIVE, v3(10.0f, 10.0f, 10.0f), 0.0f, false},   // light
        Material{ORACLE_DIELECTRIC, v3(1.0f, 1.0f, 1.0f), 2.3f, false},    // glass
        Material{ORACLE_METAL, splat(0.8f), 3.6f, false},                  // metal
    };
    std::map<std::string, int> mm = {{"default", 0}, {"red", 1}, {"green", 2}, {"white", 3},
                                     {"light", 4}, {"glass", 5}, {"metal", 6}};
    if (!mesh_from_file(s, obj_path, mm)) { delete s; return nullptr; }
    s->camera = make_camera(v3(760.0f, -279.5f, 274.5f), v3(0.0f, -279.5f, 274.5f), v3(0, 0, 1), 40.0f,
                            0.0f, 200.0f);
    s->sky_intensity = 0.5f;
    s->root_kind = ORACLE_ROOT_SINGLE;
    commit(s);
    return s;
}

size_t oracle_scene_num_objects(const oracle_scene* s) { return s->objects.size(); }
size_t oracle_scene_num_prims(const oracle_scene* s) { return s->next_prim; }
size_t oracle_scene_num_materials(const oracle_scene* s) { return s->materials.size(); }
size_t oracle_scene_num_spheres(const oracle_scene* s) { return s->n_spheres; }
size_t oracle_scene_num_triangles(const oracle_scene* s) { return s->n_tris; }
void oracle_scene_get_materials(const oracle_scene* s, oracle_material* out) {
    for (size_t i = 0; i < s->materials.size(); i++) {
        const Material& m = s->materials[i];
        out[i].kind = m.kind; out[i].rgb[0] = m.rgb.x; out[i].rgb[1] = m.rgb.y; out[i].rgb[2] = m.rgb.z;
        out[i].param = m.param; out[i].flags = m.normal_shading ? 1u : 0u;
    }
}
int oracle_scene_object_info(const oracle_scene* s, size_t i, size_t* count) {
    if (s->object_kind[i] == 0) { *count = 1; return 0; }
    *count = static_cast<const Mesh*>(s->objects[i].get())->triangles.size();
    return 1;
}
void oracle_scene_get_sphere(const oracle_scene* s, size_t obj, oracle_sphere* out) {
    const Sphere* sp = static_cast<const Sphere*>(s->objects[obj].get());
    out->c[0] = sp->center.x; out->c[1] = sp->center.y; out->c[2] = sp->center.z;
    out->r = sp->radius; out->mat = (uint32_t)sp->material;
}
void oracle_scene_get_mesh(const oracle_scene* s, size_t obj, oracle_triangle* out) {
    const Mesh* m = static_cast<const Mesh*>(s->objects[obj].get());
    for (size_t i = 0; i < m->triangles.size(); i++) {
        const Triangle& t = m->triangles[i];
        const V3 p[3] = {t.p1, t.p2, t.p3};
        for (int k = 0; k < 3; k++) { out[i].p[k][0] = p[k].x; out[i].p[k][1] = p[k].y; out[i].p[k][2] = p[k].z; }
        out[i].n0[0] = t.n1.x; out[i].n0[1] = t.n1.y; out[i].n0[2] = t.n1.z;
        out[i].mat = (uint32_t)t.material_index;
    }
}
int oracle_scene_get_xform(const oracle_scene* s, size_t obj, float out[16]) {
    const Mat4& m = s->mesh_xforms[obj];
    std::memcpy(out, m.m, sizeof(m.m));
    return m.is_identity() ? 0 : 1;
}
float oracle_scene_sky(const oracle_scene* s) { return s->sky_intensity; }
int oracle_scene_root_kind(const oracle_scene* s) { return s->root_kind; }
void oracle_scene_camera(const oracle_scene* s, float aspect, oracle_camera* out) {
    Camera c = s->camera;
    c.setup(aspect);
    auto put = [](float* d, V3 v) { d[0] = v.x; d[1] = v.y; d[2] = v.z; };
    put(out->origin, c.origin); put(out->horizontal, c.horizontal); put(out->vertical, c.vertical);
    put(out->lower_left, c.lower_left_corner); put(out->u, c.u); put(out->v, c.v);
    out->lens_radius = c.lens_radius;
}
void oracle_scene_camera_params(const oracle_scene* s, float out[12]) {
    const Camera& c = s->camera;
    out[0] = c.look_from.x; out[1] = c.look_from.y; out[2] = c.look_from.z;
    out[3] = c.look_at.x; out[4] = c.look_at.y; out[5] = c.look_at.z;
    out[6] = c.vup.x; out[7] = c.vup.y; out[8] = c.vup.z;
    out[9] = c.vfov; out[10] = c.aperture; out[11] = c.focus_dist;
}

void oracle_trace(const oracle_scene* s, const oracle_ray* rays, size_t n, float t_min, float t_max,
                  oracle_hit* out, oracle_stats* stats, int nthreads) {
    double t0 = now_s();
    if (nthreads < 1) nthreads = 1;
    std::vector<Counters> per(nthreads);
    auto work = [&](int tid) {
        tl_cnt = Counters();
        size_t b = n * (size_t)tid / nthreads, e = n * (size_t)(tid + 1) / nthreads;
        for (size_t i = b; i < e; i++) {
            Ray r{v3(rays[i].o[0], rays[i].o[1], rays[i].o[2]), v3(rays[i].d[0], rays[i].d[1], rays[i].d[2])};
            Hit h;
            tl_cnt.rays++;
            bool ok = s->root->hit(r, t_min, t_max, h);
            oracle_hit& o = out[i];
            if (ok) {
                tl_cnt.hits++;
                o.t = h.t; o.prim_id = h.prim_id;
                o.p[0] = h.p.x; o.p[1] = h.p.y; o.p[2] = h.p.z;
                o.n[0] = h.normal.x; o.n[1] = h.normal.y; o.n[2] = h.normal.z;
                o.front_face = h.front_face ? 1u : 0u; o.mat = (uint32_t)h.material;
            } else {
                std::memset(&o, 0, sizeof(o));
                o.prim_id = 0xFFFFFFFFu;
            }
        }
        per[tid] = tl_cnt;
    };
    std::vector<std::thread> th;
    for (int t = 1; t < nthreads; t++) th.emplace_back(work, t);
    work(0);
    for (auto& t : th) t.join();
    Counters tot;
    for (auto& c : per) tot.add(c);
    fill_stats(stats, tot, now_s() - t0);
}

void oracle_render(const oracle_scene* s_in, uint32_t width, uint32_t height, uint32_t spp,
                   uint32_t max_bounces, uint32_t block, uint32_t seed_offset, int nthreads,
                   float* accum_rgb, uint32_t* image, oracle_stats* stats) {
    double t0 = now_s();
    oracle_scene& s = *const_cast<oracle_scene*>(s_in);
    Camera saved = s.camera;
    s.camera.setup((float)width / (float)height);          // main.rs:324-325
    uint32_t nvb = (height + block - 1) / block, nhb = (width + block - 1) / block;
    if (nthreads < 1) nthreads = 1;
    std::atomic<uint32_t> next{0};
    std::vector<Counters> per(nthreads);
    auto work = [&](int tid) {
        tl_cnt = Counters();
        for (;;) {
            uint32_t job = next.fetch_add(1);   // jobs are queued j-major, i-minor (main.rs:332-333)
            if (job >= nvb * nhb) break;
            render_tile(s, width, height, spp, max_bounces, block, job % nhb, job / nhb, seed_offset,
                        accum_rgb, image);
        }
        per[tid] = tl_cnt;
    };
    std::vector<std::thread> th;
    for (int t = 1; t < nthreads; t++) th.emplace_back(work, t);
    work(0);
    for (auto& t : th) t.join();
    s.camera = saved;
    Counters tot;
    for (auto& c : per) tot.add(c);
    fill_stats(stats, tot, now_s() - t0);
}

size_t oracle_record_rays(const oracle_scene* s_in, uint32_t width, uint32_t height, uint32_t spp,
                          uint32_t max_bounces, uint32_t seed_offset, oracle_ray* out, size_t cap,
                          uint32_t stride) {
    oracle_scene& s = *const_cast<oracle_scene*>(s_in);
    Camera saved = s.camera;
    s.camera.setup((float)width / (float)height);
    RayRecorder rec{out, cap, 0, 0, stride ? stride : 1};
    tl_rec = &rec;
    tl_cnt = Counters();
    const uint32_t block = 64;
    uint32_t nvb = (height + block - 1) / block, nhb = (width + block - 1) / block;
    for (uint32_t j = 0; j < nvb; j++)
        for (uint32_t i = 0; i < nhb; i++)
            render_tile(s, width, height, spp, max_bounces, block, i, j, seed_offset, nullptr, nullptr);
    tl_rec = nullptr;
    s.camera = saved;
    return rec.n;
}

void oracle_xorshift32(uint32_t seed, uint32_t* out, size_t n) {
    RandomSeries r(seed);
    for (size_t i = 0; i < n; i++) out[i] = r.xorshift32();
}
void oracle_random01(uint32_t seed, float* out, size_t n) {
    RandomSeries r(seed);
    for (size_t i = 0; i < n; i++) out[i] = r.random01();
}
float oracle_random01_of(uint32_t x) { return RandomSeries::to01(x); }
uint32_t oracle_rgb_to_u32(float r, float g, float b) { return rgb_to_u32(v3(r, g, b)); }
void oracle_tonemap(const float* a, uint32_t spp, uint32_t* out, size_t npix) {
    float scale = 1.0f / (float)spp;
    for (size_t i = 0; i < npix; i++) {
        V3 c = v3(a[3 * i], a[3 * i + 1], a[3 * i + 2]) * scale;
        V3 pc = vmin(v3(std::sqrt(c.x), std::sqrt(c.y), std::sqrt(c.z)), v3(1, 1, 1));
        out[i] = rgb_to_u32(pc);
    }
}
int oracle_aabb_hit(const float bmin[3], const float bmax[3], const float o[3], const float d[3],
                    float t_min, float t_max) {
    Aabb b{v3(bmin[0], bmin[1], bmin[2]), v3(bmax[0], bmax[1], bmax[2])};
    return b.hit(v3(o[0], o[1], o[2]), v3(d[0], d[1], d[2]), t_min, t_max) ? 1 : 0;
}
int oracle_ray_triangle(const float o[3], const float d[3], const float v0[3], const float v1[3],
                        const float v2[3], float out[4]) {
    Ray r{v3(o[0], o[1], o[2]), v3(d[0], d[1], d[2])};
    bool bf = false; float t = 0, u = 0, v = 0;
    bool h = ray_triangle_intersection(r, v3(v0[0], v0[1], v0[2]), v3(v1[0], v1[1], v1[2]),
                                       v3(v2[0], v2[1], v2[2]), bf, t, u, v);
    out[0] = bf ? 1.0f : 0.0f; out[1] = t; out[2] = u; out[3] = v;
    return h ? 1 : 0;
}
size_t oracle_obj_count(const char* path, size_t* n_models, size_t* n_faces) {
    std::vector<V3> pos, nrm;
    std::vector<ObjModel> models;
    std::vector<std::string> names;
    if (!load_obj(path, pos, nrm, models, names, n_faces)) return (size_t)-1;
    if (n_models) *n_models = models.size();
    size_t t = 0;
    for (auto& m : models) t += m.vi.size() / 3;
    return t;
}
int oracle_scatter(const oracle_material* m, const float ro[3], const float rd[3], const float p[3],
                   const float n[3], int front_face, uint32_t* rng_state, float out[9]) {
    Material mat = make_material(*m);
    Ray in{v3(ro[0], ro[1], ro[2]), v3(rd[0], rd[1], rd[2])};
    Hit h; h.p = v3(p[0], p[1], p[2]); h.normal = v3(n[0], n[1], n[2]); h.front_face = front_face != 0;
    RandomSeries rs(*rng_state);
    V3 att = v3(0, 0, 0); Ray o{v3(0, 0, 0), v3(0, 0, 0)};
    bool ok = mat.scatter(in, h, att, o, rs);
    *rng_state = rs.state;
    out[0] = att.x; out[1] = att.y; out[2] = att.z;
    out[3] = o.origin.x; out[4] = o.origin.y; out[5] = o.origin.z;
    out[6] = o.dir.x; out[7] = o.dir.y; out[8] = o.dir.z;
    return ok ? 1 : 0;
}
uint32_t oracle_rng_state_after_world1(void) {
    oracle_scene* s = oracle_create_world1();
    delete s;
    return g_world1_rng_state;
}

}  // extern "C"
