// Host scene API: the C++ mirror of the reference's Camera / Shape / Material / World types.
// A scene written against the reference API (Sphere{..}, Mesh::from_file(..), Bvh::new(objects),
// World{camera, hittable, sky_intensity}) is written the same way here; GpuRenderer then
// describes it to the GPU library through the C ABI (rt_abi.h) and stands where render()
// (main.rs:319-416) stood.
#pragma once

#include <cmath>
#include <cstdint>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "rt_abi.h"
#include "rt_host.h"

namespace rthost {

// ---- math.rs --------------------------------------------------------------------------------
struct Vec3 {
    float x = 0, y = 0, z = 0;
    Vec3() = default;
    Vec3(float x_, float y_, float z_) : x(x_), y(y_), z(z_) {}
    static Vec3 splat(float s) { return Vec3(s, s, s); }
    static Vec3 zero() { return Vec3(0, 0, 0); }
    static Vec3 one() { return Vec3(1, 1, 1); }
    static Vec3 unit_z() { return Vec3(0, 0, 1); }
};
inline Vec3 operator+(Vec3 a, Vec3 b) { return Vec3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline Vec3 operator-(Vec3 a, Vec3 b) { return Vec3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline Vec3 operator*(float s, Vec3 a) { return Vec3(s * a.x, s * a.y, s * a.z); }
inline Vec3 operator*(Vec3 a, float s) { return Vec3(a.x * s, a.y * s, a.z * s); }
inline Vec3 operator/(Vec3 a, float s) { return Vec3(a.x / s, a.y / s, a.z / s); }
inline float dot(Vec3 a, Vec3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
inline Vec3 cross(Vec3 a, Vec3 b) {
    return Vec3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
inline Vec3 normalize(Vec3 a) { float l = std::sqrt(dot(a, a)); return Vec3(a.x / l, a.y / l, a.z / l); }

// math.rs:4-39.  Scene constructors draw sphere positions from this stream (main.rs:133-152).
class RandomSeries {
public:
    explicit RandomSeries(uint32_t seed) : state_(seed) {}
    uint32_t xorshift32();
    float random01();
    float random(float lo, float hi);
    int32_t range_i32(int32_t lo, int32_t hi);
    uint32_t state() const { return state_; }
private:
    uint32_t state_;
};

// ---- material.rs ----------------------------------------------------------------------------
struct Material {
    virtual ~Material() {}
    virtual rt_material describe() const = 0;
};
using MaterialRef = std::shared_ptr<const Material>;       // Arc<dyn Material>

struct LambertianMaterial : Material {
    Vec3 albedo; bool normal;
    LambertianMaterial(Vec3 a, bool n) : albedo(a), normal(n) {}
    rt_material describe() const override;
};
struct MetalMaterial : Material {
    Vec3 albedo; float fuzz;
    MetalMaterial(Vec3 a, float f) : albedo(a), fuzz(f) {}
    rt_material describe() const override;
};
struct DielectricMaterial : Material {
    Vec3 albedo; float ior;
    DielectricMaterial(Vec3 a, float i) : albedo(a), ior(i) {}
    rt_material describe() const override;
};
struct EmissiveMaterial : Material {
    Vec3 color;
    explicit EmissiveMaterial(Vec3 c) : color(c) {}
    rt_material describe() const override;
};

// ---- camera.rs ------------------------------------------------------------------------------
class Camera {
public:
    Camera() = default;
    Camera(Vec3 look_from, Vec3 look_at, Vec3 vup, float vfov, float aperture, float focus_dist)
        : look_from_(look_from), look_at_(look_at), vup_(vup), vfov_(vfov), aperture_(aperture),
          focus_dist_(focus_dist) {}
    void setup(float aspect_ratio);            // camera.rs:55-78
    rt_camera describe() const;                // the post-setup fields
private:
    Vec3 look_from_, look_at_{0, 0, -1}, vup_{0, 1, 0};
    float vfov_ = 90, aperture_ = 0, focus_dist_ = 1;
    Vec3 origin_, horizontal_, vertical_, lower_left_corner_, w_, u_, v_;
    float lens_radius_ = 0;
};

// ---- shape.rs -------------------------------------------------------------------------------
struct Mat4 {
    float m[16];                               // column-major
    static Mat4 identity();
    static Mat4 from_translation(Vec3 t);
    static Mat4 from_scale(Vec3 s);
    bool is_identity() const;
};

struct Hittable {
    virtual ~Hittable() {}
};
using HittableBox = std::unique_ptr<Hittable>;             // Box<dyn Hittable>

struct Sphere : Hittable {
    float radius; Vec3 center; MaterialRef material;
    Sphere(Vec3 c, float r, MaterialRef m) : radius(r), center(c), material(std::move(m)) {}
};

struct Vertex { Vec3 p, n; };
struct Triangle { Vertex v1, v2, v3; size_t material_index; };

struct Mesh : Hittable {
    std::vector<Triangle> triangles;
    std::vector<MaterialRef> materials;
    Mat4 transform;
    Mesh(std::vector<Triangle> t, std::vector<MaterialRef> m, Mat4 x)
        : triangles(std::move(t)), materials(std::move(m)), transform(x) {}
    // shape.rs:326-345
    static std::unique_ptr<Mesh> plane(float w, float h, MaterialRef material, Mat4 transform);
    // shape.rs:347-406.  Throws std::runtime_error when the OBJ or its MTL cannot be read or a
    // face has no normals (the reference panics in those cases).
    static std::unique_ptr<Mesh> from_file(const std::string& filename,
                                           const std::vector<std::pair<std::string, MaterialRef>>& material_map,
                                           Mat4 transform);
};

struct HittableList : Hittable {               // shape.rs:55-57
    std::vector<HittableBox> objects;
    explicit HittableList(std::vector<HittableBox> o) : objects(std::move(o)) {}
};

struct Bvh : Hittable {                        // shape.rs:87-91; the tree itself is built by the GPU library
    std::vector<HittableBox> objects;
    int axis = 0;                              // Bvh::new(objects, axis), shape.rs:94
    explicit Bvh(std::vector<HittableBox> o, int axis_ = 0) : objects(std::move(o)), axis(axis_) {}
};

struct World {                                 // main.rs:69-73
    Camera camera;
    HittableBox hittable;
    float sky_intensity = 1.0f;
};

// main.rs:75-317
World create_world1();
// variant 0: as shipped; 1: the glass + diffuse material map; 2: as shipped plus the two planes the
// reference keeps in a comment (Mesh::plane with Mat4::from_translation, main.rs:195-204)
World create_world2(const std::string& thing_obj, int variant = 0);
World create_cornell_box(const std::string& filename);
World create_soup(uint64_t n_tris, uint32_t grid, uint32_t seed);

// What the C ABI needs to know about a World.
struct SceneDescription {
    std::vector<rt_material> materials;
    // n_tris stays the mesh's triangle count after `tris` has been released (big meshes, once committed)
    struct Object {
        bool is_mesh; rt_sphere sphere; std::vector<rt_triangle> tris; size_t n_tris = 0; bool released = false;
        bool has_xform = false; float xform[16] = {};       // Mesh.transform when it is not the identity
    };
    std::vector<Object> objects;               // the leaf objects (spheres, meshes) in the order they are met
    int root_kind = RT_ROOT_BVH;
    // how the leaf objects nest: the calls that rebuild the world's tree through the C ABI
    struct Step { enum Kind { OBJECT, BEGIN_GROUP, END_GROUP } kind; size_t object; int group_kind; int axis; };
    std::vector<Step> steps;
};
// Throws std::runtime_error for what the ABI cannot express (an object without a material, ...).
SceneDescription describe(const World& world);

// a failed rt_* call inside the host layer: carries the call's rt_status and message
struct StatusError : std::runtime_error {
    int status;
    StatusError(int s, const std::string& msg) : std::runtime_error(msg), status(s) {}
};

// The drop-in for render(): owns the committed GPU scene of one World.
class GpuRenderer {
public:
    GpuRenderer(rt_context* ctx, World world);
    ~GpuRenderer();
    GpuRenderer(const GpuRenderer&) = delete;
    GpuRenderer& operator=(const GpuRenderer&) = delete;
    rt_scene* scene() const { return scene_; }
    const World& world() const { return world_; }
    const SceneDescription& description() const { return desc_; }
    // all pixels, samples [0, spp); adds linear sums into accum_rgb (host memory)
    int render(uint32_t width, uint32_t height, uint32_t spp, uint32_t max_bounces, uint64_t seed,
               float* accum_rgb, rt_stats* stats);
    // the same frame in the reference's progressive steps, every 64 x 64 block delivered after each
    int render_progressive(uint32_t width, uint32_t height, uint32_t spp, uint32_t max_bounces, uint64_t seed,
                           rt_block_fn on_block, void* user, float* accum_rgb, rt_stats* stats);
private:
    rt_context* ctx_ = nullptr;
    World world_;
    SceneDescription desc_;
    rt_scene* scene_ = nullptr;
};

std::vector<uint32_t> progressive_schedule(uint32_t spp);

// OBJ ingest shared by Mesh::from_file and rt_obj_count
struct ObjModel { std::string name; int material_id = -1; std::vector<uint32_t> vi, ni; };
struct ObjData {
    std::vector<float> positions, normals;     // 3 floats each
    std::vector<ObjModel> models;
    std::vector<std::string> material_names;   // newmtl order of the MTL file
    size_t n_faces = 0;
};
// returns an empty string on success; threads = 0: one chunk per 4 MB of text, up to the cores (or RT_OBJ_THREADS)
std::string load_obj(const std::string& path, ObjData& out, int threads = 0);

}  // namespace rthost
