// Host scene API implementation: see scene_api.hpp.

#include "scene_api.hpp"

#include <cstring>
#include <stdexcept>

namespace rthost {

// ---- RandomSeries (math.rs:4-39) ------------------------------------------------------------
uint32_t RandomSeries::xorshift32() {
    uint32_t x = state_;
    x ^= x << 13;
    x ^= x >> 17;
    x ^= x << 5;
    state_ = x;
    return x;
}
float RandomSeries::random01() { return (float)((double)xorshift32() / (double)UINT32_MAX); }
float RandomSeries::random(float lo, float hi) { return random01() * (hi - lo) + lo; }
int32_t RandomSeries::range_i32(int32_t lo, int32_t hi) {
    const int32_t diff = hi - lo;
    return diff > 0 ? (int32_t)(xorshift32() % (uint32_t)diff) + lo : lo;
}

// ---- materials ------------------------------------------------------------------------------
static rt_material mk(uint32_t kind, Vec3 c, float param, uint32_t flags) {
    rt_material m;
    m.kind = kind; m.rgb[0] = c.x; m.rgb[1] = c.y; m.rgb[2] = c.z; m.param = param; m.flags = flags;
    return m;
}
rt_material LambertianMaterial::describe() const {
    return mk(RT_MAT_LAMBERT, albedo, 0.0f, normal ? RT_MAT_FLAG_NORMAL_SHADING : 0u);
}
rt_material MetalMaterial::describe() const { return mk(RT_MAT_METAL, albedo, fuzz, 0); }
rt_material DielectricMaterial::describe() const { return mk(RT_MAT_DIELECTRIC, albedo, ior, 0); }
rt_material EmissiveMaterial::describe() const { return mk(RT_MAT_EMISSIVE, color, 0.0f, 0); }

// ---- Camera (camera.rs:55-78) ---------------------------------------------------------------
void Camera::setup(float aspect_ratio) {
    const float theta = vfov_ * 3.14159265358979323846f / 180.0f;
    const float h = std::tan(theta / 2.0f);
    const float viewport_height = 2.0f * h;
    const float viewport_width = aspect_ratio * viewport_height;
    w_ = normalize(look_from_ - look_at_);
    u_ = normalize(cross(vup_, w_));
    v_ = cross(w_, u_);
    origin_ = look_from_;
    horizontal_ = (focus_dist_ * viewport_width) * u_;
    vertical_ = (focus_dist_ * viewport_height) * v_;
    lower_left_corner_ = origin_ - horizontal_ / 2.0f - vertical_ / 2.0f - focus_dist_ * w_;
    lens_radius_ = aperture_ / 2.0f;
}
rt_camera Camera::describe() const {
    rt_camera c;
    auto put = [](float* d, Vec3 v) { d[0] = v.x; d[1] = v.y; d[2] = v.z; };
    put(c.origin, origin_); put(c.horizontal, horizontal_); put(c.vertical, vertical_);
    put(c.lower_left, lower_left_corner_); put(c.u, u_); put(c.v, v_);
    c.lens_radius = lens_radius_;
    return c;
}

// ---- Mat4 -----------------------------------------------------------------------------------
Mat4 Mat4::identity() {
    Mat4 r;
    std::memset(r.m, 0, sizeof(r.m));
    r.m[0] = r.m[5] = r.m[10] = r.m[15] = 1.0f;
    return r;
}
Mat4 Mat4::from_translation(Vec3 t) {
    Mat4 r = identity();
    r.m[12] = t.x; r.m[13] = t.y; r.m[14] = t.z;
    return r;
}
Mat4 Mat4::from_scale(Vec3 s) {
    Mat4 r = identity();
    r.m[0] = s.x; r.m[5] = s.y; r.m[10] = s.z;
    return r;
}
bool Mat4::is_identity() const {
    const Mat4 i = identity();
    return std::memcmp(m, i.m, sizeof(m)) == 0;
}

// ---- Mesh -----------------------------------------------------------------------------------
std::unique_ptr<Mesh> Mesh::plane(float w, float h, MaterialRef material, Mat4 transform) {
    const Vec3 n(0.0f, 0.0f, 1.0f);
    std::vector<Triangle> t = {
        Triangle{Vertex{Vec3(-w, -h, 0), n}, Vertex{Vec3(-w, h, 0), n}, Vertex{Vec3(w, -h, 0), n}, 0},
        Triangle{Vertex{Vec3(w, -h, 0), n}, Vertex{Vec3(-w, h, 0), n}, Vertex{Vec3(w, h, 0), n}, 0},
    };
    return std::make_unique<Mesh>(std::move(t), std::vector<MaterialRef>{std::move(material)}, transform);
}

std::unique_ptr<Mesh> Mesh::from_file(const std::string& filename,
                                      const std::vector<std::pair<std::string, MaterialRef>>& material_map,
                                      Mat4 transform) {
    ObjData obj;
    const std::string err = load_obj(filename, obj);
    if (!err.empty()) throw std::runtime_error("Failed to load OBJ file: " + err);
    if (obj.material_names.empty()) throw std::runtime_error("Failed to load materials: " + filename);
    std::vector<MaterialRef> materials;
    std::map<std::string, size_t> index;       // shape.rs:365-370
    for (const auto& kv : material_map) {
        index[kv.first] = materials.size();
        materials.push_back(kv.second);
    }
    std::vector<Triangle> triangles;
    for (const ObjModel& m : obj.models) {     // shape.rs:373-403
        const std::string& mat_name = obj.material_names[m.material_id < 0 ? 0 : (size_t)m.material_id];
        auto it = index.find(mat_name);
        const size_t material_index = it == index.end() ? 0 : it->second;
        auto vertex = [&](size_t k) {
            const float* p = &obj.positions[3 * (size_t)m.vi[k]];
            const float* n = &obj.normals[3 * (size_t)m.ni[k]];
            return Vertex{Vec3(p[0], p[1], p[2]), Vec3(n[0], n[1], n[2])};
        };
        for (size_t k = 0; k + 2 < m.vi.size(); k += 3)
            triangles.push_back(Triangle{vertex(k), vertex(k + 1), vertex(k + 2), material_index});
    }
    return std::make_unique<Mesh>(std::move(triangles), std::move(materials), transform);
}

// ---- scene constructors (main.rs:75-317) ----------------------------------------------------
World create_world1() {
    auto material_center = std::make_shared<LambertianMaterial>(Vec3::one(), true);
    auto material_ground = std::make_shared<LambertianMaterial>(Vec3(0.13f, 0.8f, 0.13f), false);
    auto material_left = std::make_shared<MetalMaterial>(Vec3::splat(0.8f), 0.3f);
    auto material_right = std::make_shared<MetalMaterial>(Vec3::splat(0.4f), 1.0f);
    auto material_glass = std::make_shared<DielectricMaterial>(Vec3::splat(1.0f), 1.3f);
    auto material_glow = std::make_shared<EmissiveMaterial>(Vec3::splat(10.0f));

    std::vector<HittableBox> objects;
    objects.push_back(std::make_unique<Sphere>(Vec3(0.0f, -1.0f, -200.5f), 200.0f, material_ground));
    objects.push_back(std::make_unique<Sphere>(Vec3(-1.0f, -1.0f, 0.0f), 0.5f, material_glass));
    objects.push_back(std::make_unique<Sphere>(Vec3(1.0f, -1.0f, 0.0f), 0.5f, material_right));
    objects.push_back(std::make_unique<Sphere>(Vec3(0.0f, -1.0f, 0.0f), 0.5f, material_center));
    objects.push_back(std::make_unique<Sphere>(Vec3(-0.4f, -1.3f, -0.5f + 0.1f), 0.1f, material_right));
    objects.push_back(std::make_unique<Sphere>(Vec3(0.4f, -1.3f, -0.5f + 0.1f), 0.1f, material_left));

    RandomSeries r(10);
    const std::vector<MaterialRef> materials = {material_center, material_left, material_right, material_glass};
    for (int i = 0; i < 500; i++) {
        const float x = r.random(-5.0f, 5.0f);
        const float y = r.random(-5.0f, 5.0f);
        const float z = r.random(-0.25f, 5.0f);
        const MaterialRef& m = materials[(size_t)r.range_i32(0, (int32_t)materials.size())];
        objects.push_back(std::make_unique<Sphere>(Vec3(x, y, z), 0.25f, m));
    }
    objects.push_back(std::make_unique<Sphere>(Vec3(0.0f, 0.0f, 50.0f), 12.0f, material_glow));

    World w;
    w.camera = Camera(Vec3(2.0f, -3.0f, 1.0f), Vec3(0.0f, 1.0f, 0.0f), Vec3::unit_z(), 60.0f, 0.0f, 10.0f);
    w.hittable = std::make_unique<Bvh>(std::move(objects));
    w.sky_intensity = 0.02f;
    return w;
}

World create_world2(const std::string& thing_obj, int variant) {
    auto material_glow = std::make_shared<EmissiveMaterial>(Vec3::splat(4.0f));
    std::vector<std::pair<std::string, MaterialRef>> map;
    if (variant == 0) {
        map.push_back({"default", std::make_shared<LambertianMaterial>(Vec3::one(), false)});
    } else {
        map.push_back({"None", std::make_shared<LambertianMaterial>(Vec3::splat(0.73f), false)});
        map.push_back({"Material", std::make_shared<DielectricMaterial>(Vec3::one(), 1.5f)});
    }
    std::vector<HittableBox> objects;
    if (variant == 2) {                        // main.rs:195-204, commented out in the reference
        map.clear();
        auto material1 = std::make_shared<LambertianMaterial>(Vec3::one(), false);
        map.push_back({"default", material1});
        objects.push_back(Mesh::plane(2.0f, 2.0f, material1, Mat4::from_translation(Vec3(0.0f, 0.0f, -0.5f))));
        objects.push_back(Mesh::plane(2.0f, 2.0f, material1, Mat4::from_translation(Vec3(0.0f, 0.0f, 1.0f))));
    }
    objects.push_back(Mesh::from_file(thing_obj, map, Mat4::identity()));
    objects.push_back(std::make_unique<Sphere>(Vec3(0.0f, 0.0f, 1.5f), 0.65f, material_glow));
    World w;
    w.camera = Camera(Vec3(0.0f, -2.0f, 0.0f), Vec3(0.0f, 1.0f, 0.0f), Vec3::unit_z(), 88.0f, 0.001f, 1.0f);
    w.hittable = std::make_unique<Bvh>(std::move(objects));
    w.sky_intensity = 1.5f;
    return w;
}

World create_cornell_box(const std::string& filename) {
    std::vector<std::pair<std::string, MaterialRef>> map = {
        {"default", std::make_shared<LambertianMaterial>(Vec3(0.5f, 0.5f, 0.5f), false)},
        {"red", std::make_shared<LambertianMaterial>(Vec3(0.65f, 0.05f, 0.05f), false)},
        {"green", std::make_shared<LambertianMaterial>(Vec3(0.12f, 0.45f, 0.15f), false)},
        {"white", std::make_shared<LambertianMaterial>(Vec3(0.73f, 0.73f, 0.73f), false)},
        {"light", std::make_shared<EmissiveMaterial>(Vec3(10.0f, 10.0f, 10.0f))},
        {"glass", std::make_shared<DielectricMaterial>(Vec3(1.0f, 1.0f, 1.0f), 2.3f)},
        {"metal", std::make_shared<MetalMaterial>(Vec3::splat(0.8f), 3.6f)},
    };
    World w;
    w.camera = Camera(Vec3(760.0f, -279.5f, 274.5f), Vec3(0.0f, -279.5f, 274.5f), Vec3(0.0f, 0.0f, 1.0f), 40.0f,
                      0.0f, 200.0f);
    w.hittable = Mesh::from_file(filename, map, Mat4::identity());
    w.sky_intensity = 0.5f;
    return w;
}

World create_soup(uint64_t n_tris, uint32_t grid, uint32_t seed) {
    std::vector<MaterialRef> mats = {
        std::make_shared<LambertianMaterial>(Vec3::splat(0.73f), false),
        std::make_shared<MetalMaterial>(Vec3::splat(0.8f), 0.3f),
        std::make_shared<DielectricMaterial>(Vec3::one(), 1.5f),
        std::make_shared<LambertianMaterial>(Vec3(0.65f, 0.05f, 0.05f), false),
    };
    RandomSeries r(seed);
    std::vector<Triangle> tris;
    tris.reserve(n_tris);
    for (uint64_t i = 0; i < n_tris; i++) {
        const float cx = r.random(-100.0f, 100.0f);
        const float cy = r.random(-100.0f, 100.0f);
        const float cz = r.random(0.0f, 100.0f);
        Vec3 p[3];
        for (int k = 0; k < 3; k++) {
            const float ox = r.random(-0.15f, 0.15f);
            const float oy = r.random(-0.15f, 0.15f);
            const float oz = r.random(-0.15f, 0.15f);
            p[k] = Vec3(cx + ox, cy + oy, cz + oz);
        }
        Vec3 n = cross(p[1] - p[0], p[2] - p[0]);
        const float l2 = dot(n, n);
        n = l2 > 0.0f ? normalize(n) : Vec3(0.0f, 0.0f, 1.0f);
        tris.push_back(Triangle{Vertex{p[0], n}, Vertex{p[1], n}, Vertex{p[2], n}, (size_t)(i % 4)});
    }
    std::vector<HittableBox> objects;
    objects.push_back(std::make_unique<Mesh>(std::move(tris), mats, Mat4::identity()));
    const float half = 0.5f * 3.0f * (float)(grid ? grid - 1 : 0);
    for (uint32_t j = 0; j < grid; j++)
        for (uint32_t i = 0; i < grid; i++)
            objects.push_back(std::make_unique<Sphere>(Vec3(3.0f * (float)i - half, 3.0f * (float)j - half, -1.0f),
                                                       0.5f, mats[(i + j) % 4]));
    objects.push_back(std::make_unique<Sphere>(Vec3(0.0f, 0.0f, 300.0f), 30.0f,
                                               std::make_shared<EmissiveMaterial>(Vec3::splat(10.0f))));
    World w;
    w.camera = Camera(Vec3(260.0f, -260.0f, 160.0f), Vec3(0.0f, 0.0f, 40.0f), Vec3::unit_z(), 40.0f, 0.0f, 200.0f);
    w.hittable = std::make_unique<Bvh>(std::move(objects));
    w.sky_intensity = 0.5f;
    return w;
}

// ---- World -> ABI description ---------------------------------------------------------------
namespace {

struct Describer {
    SceneDescription& d;
    std::map<const Material*, uint32_t> mat_index;

    uint32_t material(const MaterialRef& m) {
        if (!m) throw std::runtime_error("object without a material");
        auto it = mat_index.find(m.get());
        if (it != mat_index.end()) return it->second;
        const uint32_t i = (uint32_t)d.materials.size();
        d.materials.push_back(m->describe());
        mat_index[m.get()] = i;
        return i;
    }
    void leaf(SceneDescription::Object&& o) {
        d.steps.push_back(SceneDescription::Step{SceneDescription::Step::OBJECT, d.objects.size(), 0, 0});
        d.objects.push_back(std::move(o));
    }
    void object(const Hittable& h) {
        if (const Sphere* s = dynamic_cast<const Sphere*>(&h)) {
            SceneDescription::Object o;
            o.is_mesh = false;
            o.sphere.c[0] = s->center.x; o.sphere.c[1] = s->center.y; o.sphere.c[2] = s->center.z;
            o.sphere.r = s->radius;
            o.sphere.mat = material(s->material);
            leaf(std::move(o));
        } else if (const Mesh* m = dynamic_cast<const Mesh*>(&h)) {
            SceneDescription::Object o;
            o.is_mesh = true;
            if (!m->transform.is_identity()) {
                o.has_xform = true;
                std::memcpy(o.xform, m->transform.m, sizeof(o.xform));
            }
            std::vector<uint32_t> local(m->materials.size());
            for (size_t i = 0; i < local.size(); i++) local[i] = material(m->materials[i]);
            o.tris.resize(m->triangles.size());
            o.n_tris = o.tris.size();
            for (size_t i = 0; i < o.tris.size(); i++) {
                const Triangle& t = m->triangles[i];
                rt_triangle& r = o.tris[i];
                const Vec3 p[3] = {t.v1.p, t.v2.p, t.v3.p};
                for (int k = 0; k < 3; k++) { r.p[k][0] = p[k].x; r.p[k][1] = p[k].y; r.p[k][2] = p[k].z; }
                r.n0[0] = t.v1.n.x; r.n0[1] = t.v1.n.y; r.n0[2] = t.v1.n.z;
                if (t.material_index >= local.size()) throw std::runtime_error("triangle material index out of range");
                r.mat = local[t.material_index];
            }
            leaf(std::move(o));
        } else if (const Bvh* b = dynamic_cast<const Bvh*>(&h)) {          // a container inside a container
            d.steps.push_back(SceneDescription::Step{SceneDescription::Step::BEGIN_GROUP, 0, RT_ROOT_BVH, b->axis});
            for (const HittableBox& c : b->objects) object(*c);
            d.steps.push_back(SceneDescription::Step{SceneDescription::Step::END_GROUP, 0, 0, 0});
        } else if (const HittableList* l = dynamic_cast<const HittableList*>(&h)) {
            d.steps.push_back(SceneDescription::Step{SceneDescription::Step::BEGIN_GROUP, 0, RT_ROOT_LIST, 0});
            for (const HittableBox& c : l->objects) object(*c);
            d.steps.push_back(SceneDescription::Step{SceneDescription::Step::END_GROUP, 0, 0, 0});
        } else {
            throw std::runtime_error("unknown Hittable");
        }
    }
};

}  // namespace

SceneDescription describe(const World& world) {
    SceneDescription d;
    Describer ds{d, {}};
    if (!world.hittable) throw std::runtime_error("World without a hittable");
    if (const Bvh* b = dynamic_cast<const Bvh*>(world.hittable.get())) {
        if (b->axis != 0) throw std::runtime_error("the root Bvh starts at axis 0 (Bvh::new(objects, 0), main.rs:170, 234)");
        d.root_kind = RT_ROOT_BVH;
        for (const HittableBox& o : b->objects) ds.object(*o);
    } else if (const HittableList* l = dynamic_cast<const HittableList*>(world.hittable.get())) {
        d.root_kind = RT_ROOT_LIST;
        for (const HittableBox& o : l->objects) ds.object(*o);
    } else {
        d.root_kind = RT_ROOT_SINGLE;
        ds.object(*world.hittable);
    }
    return d;
}

// ---- GpuRenderer ----------------------------------------------------------------------------
GpuRenderer::GpuRenderer(rt_context* ctx, World world) : ctx_(ctx), world_(std::move(world)) {
    desc_ = describe(world_);
    auto ok = [](int rc) { if (rc != RT_OK) throw StatusError(rc, rt_last_error()); };
    ok(rt_scene_create(ctx, &scene_));
    try {
        ok(rt_scene_set_materials(scene_, desc_.materials.data(), desc_.materials.size()));
        for (const SceneDescription::Step& st : desc_.steps) {
            if (st.kind == SceneDescription::Step::BEGIN_GROUP) { ok(rt_scene_begin_group(scene_, st.group_kind, st.axis)); continue; }
            if (st.kind == SceneDescription::Step::END_GROUP) { ok(rt_scene_end_group(scene_)); continue; }
            const SceneDescription::Object& o = desc_.objects[st.object];
            if (o.is_mesh) ok(rt_scene_add_mesh(scene_, o.tris.data(), o.tris.size(), o.has_xform ? o.xform : nullptr));
            else ok(rt_scene_add_spheres(scene_, &o.sphere, 1));
        }
        ok(rt_scene_set_root(scene_, desc_.root_kind));
        ok(rt_scene_commit(scene_));
    } catch (...) {
        rt_scene_destroy(scene_);
        scene_ = nullptr;
        throw;
    }
    // the reference keeps its triangles alive inside the Mesh; the GPU copy is what renders, so
    // the host copies of big meshes are dropped once committed (a host-only world keeps them: it
    // exists to be inspected)
    if (!ctx) return;
    if (Bvh* b = dynamic_cast<Bvh*>(world_.hittable.get()))
        for (HittableBox& o : b->objects)
            if (Mesh* m = dynamic_cast<Mesh*>(o.get()))
                if (m->triangles.size() > (1u << 20)) { m->triangles.clear(); m->triangles.shrink_to_fit(); }
    for (SceneDescription::Object& o : desc_.objects)
        if (o.tris.size() > (1u << 20)) { o.tris.clear(); o.tris.shrink_to_fit(); o.released = true; }
}

GpuRenderer::~GpuRenderer() {
    if (scene_) rt_scene_destroy(scene_);
}

int GpuRenderer::render(uint32_t width, uint32_t height, uint32_t spp, uint32_t max_bounces, uint64_t seed,
                        float* accum_rgb, rt_stats* stats) {
    Camera cam = world_.camera;
    cam.setup((float)width / (float)height);   // main.rs:324-325
    const rt_camera c = cam.describe();
    rt_render_params p;
    std::memset(&p, 0, sizeof(p));
    p.width = width; p.height = height;
    p.x0 = 0; p.y0 = 0; p.x1 = width; p.y1 = height;
    p.sample_begin = 0; p.sample_end = spp;
    p.max_bounces = max_bounces;
    p.t_min = 0.001f;                          // main.rs:46
    p.sky_intensity = world_.sky_intensity;
    p.seed = seed;
    // a context over several GPUs renders the frame on all of them (one reduce to its root)
    if (rt_context_num_devices(ctx_) > 1) return rt_render_multi(scene_, &c, &p, accum_rgb, stats);
    return rt_render(scene_, &c, &p, accum_rgb, stats);
}

// main.rs:355-356, 409-411
std::vector<uint32_t> progressive_schedule(uint32_t spp) {
    std::vector<uint32_t> counts;
    if (spp == 0) return counts;
    uint32_t next = std::min<uint32_t>(10u, spp);
    for (;;) {
        counts.push_back(next);
        if (next >= spp) break;
        next = (uint32_t)std::min<uint64_t>((uint64_t)next * 2u, spp);
    }
    return counts;
}

int GpuRenderer::render_progressive(uint32_t width, uint32_t height, uint32_t spp, uint32_t max_bounces,
                                    uint64_t seed, rt_block_fn on_block, void* user, float* accum_rgb,
                                    rt_stats* stats) {
    constexpr uint32_t BLOCK = 64;             // BLOCK_SIZE, main.rs:22
    Camera cam = world_.camera;
    cam.setup((float)width / (float)height);   // main.rs:324-325
    const rt_camera c = cam.describe();
    rt_render_params p;
    std::memset(&p, 0, sizeof(p));
    p.width = width; p.height = height;
    p.x0 = 0; p.y0 = 0; p.x1 = width; p.y1 = height;
    p.max_bounces = max_bounces;
    p.t_min = 0.001f;                          // main.rs:46
    p.sky_intensity = world_.sky_intensity;
    p.seed = seed;
    const size_t npix = (size_t)width * height;
    std::vector<float> accum(npix * 3, 0.0f);
    std::vector<uint32_t> frame(npix), block;
    rt_stats total;
    std::memset(&total, 0, sizeof(total));
    uint32_t done = 0;
    bool stop = false;
    for (uint32_t upto : progressive_schedule(spp)) {
        p.sample_begin = done; p.sample_end = upto;
        rt_stats st;
        int rc = rt_context_num_devices(ctx_) > 1 ? rt_render_multi(scene_, &c, &p, accum.data(), &st)
                                                  : rt_render(scene_, &c, &p, accum.data(), &st);
        if (rc) return rc;
        total.n_devices = st.n_devices; total.kernel_form = st.kernel_form; total.ms_reduce += st.ms_reduce;
        total.rays_traced += st.rays_traced; total.paths += st.paths;
        total.trace_launches += st.trace_launches; total.ms_render += st.ms_render;
        total.ms_trace += st.ms_trace; total.ms_shade += st.ms_shade;
        done = upto;
        rc = rt_tonemap(ctx_, accum.data(), done, frame.data(), npix);    // main.rs:369-373
        if (rc) return rc;
        if (on_block) {
            const bool unfinished = done < spp;
            for (uint32_t by = 0; by < height; by += BLOCK)
                for (uint32_t bx = 0; bx < width; bx += BLOCK) {
                    const uint32_t bw = std::min(BLOCK, width - bx), bh = std::min(BLOCK, height - by);
                    block.resize((size_t)bw * bh);
                    for (uint32_t y = 0; y < bh; y++)
                        std::memcpy(&block[(size_t)y * bw], &frame[(size_t)(by + y) * width + bx], bw * 4u);
                    if (unfinished) {          // main.rs:377-396: red marks in the corners
                        const uint32_t red = 0x00FF0000u;
                        auto set = [&](uint32_t x, uint32_t y) { block[(size_t)y * bw + x] = red; };
                        for (uint32_t x = 0; x < std::min(5u, bw); x++) {
                            set(x, 0); set(x, bh - 1); set(bw - x - 1, 0); set(bw - x - 1, bh - 1);
                        }
                        for (uint32_t y = 0; y < std::min(5u, bh); y++) {
                            set(0, y); set(bw - 1, y); set(0, bh - y - 1); set(bw - 1, bh - y - 1);
                        }
                    }
                    if (on_block(user, bx, by, bw, bh, block.data(), done, spp) != 0) stop = true;
                }
        }
        if (stop) break;
    }
    if (accum_rgb) std::memcpy(accum_rgb, accum.data(), accum.size() * sizeof(float));
    if (stats) *stats = total;
    return RT_OK;
}

}  // namespace rthost
