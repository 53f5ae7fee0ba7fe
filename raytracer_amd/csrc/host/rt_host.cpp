// C entry points of the host scene layer (include/rt_host.h).

#include "rt_host.h"

#include <cstdio>
#include <cstring>
#include <exception>
#include <new>
#include <string>
#include <vector>

#include "scene_api.hpp"

using namespace rthost;

void rt_internal_set_error(const char* msg);   // rt_api.cpp

struct rt_world {
    std::unique_ptr<GpuRenderer> renderer;
};

namespace {

// rt_last_error() reads the library's thread-local message; route host-layer failures through
// a call that sets it.
int host_fail(const std::string& msg) {
    rt_internal_set_error(msg.c_str());
    return RT_ERR_INVALID;
}

// no exception may cross the C ABI
#define RT_HOST_GUARD_BEGIN try {
#define RT_HOST_GUARD_END(name)                                                        \
    } catch (const std::exception& e) {                                                \
        return host_fail(std::string(name ": ") + e.what());                           \
    } catch (...) {                                                                    \
        return host_fail(name ": unknown exception");                                  \
    }

template <class F>
int make_world(rt_context* ctx, rt_world** out, F&& ctor) {
    if (!out) return host_fail("rt_world_create: out is NULL");
    *out = nullptr;
    try {
        std::unique_ptr<rt_world> w(new rt_world());
        w->renderer.reset(new GpuRenderer(ctx, ctor()));
        *out = w.release();
        return RT_OK;
    } catch (const StatusError& e) {           // a failed rt_scene_* call: its own status
        rt_internal_set_error(e.what());
        return e.status;
    } catch (const std::bad_alloc&) {
        rt_internal_set_error("rt_world_create: out of host memory");
        return RT_ERR_NOMEM;
    } catch (const std::exception& e) {
        return host_fail(e.what());
    } catch (...) {
        return host_fail("rt_world_create: unknown exception");
    }
}

}  // namespace

extern "C" {

int rt_world_create_world1(rt_context* ctx, rt_world** out) {
    return make_world(ctx, out, [] { return create_world1(); });
}
int rt_world_create_world2(rt_context* ctx, const char* thing_obj, int variant, rt_world** out) {
    if (!thing_obj) return host_fail("rt_world_create_world2: NULL path");
    const std::string p(thing_obj);
    return make_world(ctx, out, [&] { return create_world2(p, variant); });
}
int rt_world_create_cornell_box(rt_context* ctx, const char* obj_path, rt_world** out) {
    if (!obj_path) return host_fail("rt_world_create_cornell_box: NULL path");
    const std::string p(obj_path);
    return make_world(ctx, out, [&] { return create_cornell_box(p); });
}
int rt_world_create_soup(rt_context* ctx, uint64_t n_tris, uint32_t grid, uint32_t seed, rt_world** out) {
    return make_world(ctx, out, [&] { return create_soup(n_tris, grid, seed); });
}
void rt_world_destroy(rt_world* w) { delete w; }

rt_scene* rt_world_scene(rt_world* w) { return w ? w->renderer->scene() : nullptr; }
float rt_world_sky_intensity(const rt_world* w) { return w ? w->renderer->world().sky_intensity : 0.0f; }

int rt_world_camera(const rt_world* w, float aspect_ratio, rt_camera* out) {
    RT_HOST_GUARD_BEGIN
    if (!w || !out) return host_fail("rt_world_camera: NULL argument");
    Camera c = w->renderer->world().camera;
    c.setup(aspect_ratio);
    *out = c.describe();
    return RT_OK;
    RT_HOST_GUARD_END("rt_world_camera")
}

size_t rt_world_num_materials(const rt_world* w) { return w ? w->renderer->description().materials.size() : 0; }
int rt_world_get_materials(const rt_world* w, rt_material* out) {
    RT_HOST_GUARD_BEGIN
    if (!w || !out) return host_fail("rt_world_get_materials: NULL argument");
    const auto& m = w->renderer->description().materials;
    std::memcpy(out, m.data(), m.size() * sizeof(rt_material));
    return RT_OK;
    RT_HOST_GUARD_END("rt_world_get_materials")
}
size_t rt_world_num_objects(const rt_world* w) { return w ? w->renderer->description().objects.size() : 0; }
int rt_world_object_info(const rt_world* w, size_t i, size_t* count) {
    if (!w || !count || i >= w->renderer->description().objects.size()) return host_fail("rt_world_object_info: bad argument");
    const auto& o = w->renderer->description().objects[i];
    *count = o.is_mesh ? o.n_tris : 1;
    return o.is_mesh ? 1 : 0;
}
int rt_world_get_sphere(const rt_world* w, size_t i, rt_sphere* out) {
    if (!w || !out || i >= w->renderer->description().objects.size()) return host_fail("rt_world_get_sphere: bad argument");
    *out = w->renderer->description().objects[i].sphere;
    return RT_OK;
}
int rt_world_get_mesh(const rt_world* w, size_t i, rt_triangle* out) {
    RT_HOST_GUARD_BEGIN
    if (!w || !out || i >= w->renderer->description().objects.size()) return host_fail("rt_world_get_mesh: bad argument");
    const auto& o = w->renderer->description().objects[i];
    if (o.released) {
        rt_internal_set_error("rt_world_get_mesh: the host copy of this mesh was released after the commit");
        return RT_ERR_STATE;
    }
    const auto& t = o.tris;
    std::memcpy(out, t.data(), t.size() * sizeof(rt_triangle));
    return RT_OK;
    RT_HOST_GUARD_END("rt_world_get_mesh")
}
int rt_world_get_mesh_xform(const rt_world* w, size_t i, float out[16]) {
    if (!w || !out || i >= w->renderer->description().objects.size()) return host_fail("rt_world_get_mesh_xform: bad argument");
    const auto& o = w->renderer->description().objects[i];
    static const float ident[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    std::memcpy(out, o.has_xform ? o.xform : ident, sizeof(ident));
    return o.has_xform ? 1 : 0;
}
int rt_world_root_kind(const rt_world* w) { return w ? w->renderer->description().root_kind : -1; }

int rt_camera_setup(const float f[3], const float a[3], const float u[3], float vfov, float aperture,
                    float focus_dist, float aspect_ratio, rt_camera* out) {
    RT_HOST_GUARD_BEGIN
    if (!f || !a || !u || !out) return host_fail("rt_camera_setup: NULL argument");
    Camera c(Vec3(f[0], f[1], f[2]), Vec3(a[0], a[1], a[2]), Vec3(u[0], u[1], u[2]), vfov, aperture, focus_dist);
    c.setup(aspect_ratio);
    *out = c.describe();
    return RT_OK;
    RT_HOST_GUARD_END("rt_camera_setup")
}

int rt_world_render(rt_world* w, uint32_t width, uint32_t height, uint32_t spp, uint32_t max_bounces,
                    uint64_t seed, float* accum_rgb, rt_stats* stats) {
    RT_HOST_GUARD_BEGIN
    if (!w || !accum_rgb) return host_fail("rt_world_render: NULL argument");
    if (spp == 0) return host_fail("rt_world_render: spp must be positive");
    return w->renderer->render(width, height, spp, max_bounces, seed, accum_rgb, stats);
    RT_HOST_GUARD_END("rt_world_render")
}

size_t rt_progressive_schedule(uint32_t spp, uint32_t* counts, size_t cap) {
    const std::vector<uint32_t> s = progressive_schedule(spp);
    for (size_t i = 0; i < s.size() && i < cap && counts; i++) counts[i] = s[i];
    return s.size();
}

int rt_world_render_progressive(rt_world* w, uint32_t width, uint32_t height, uint32_t spp, uint32_t max_bounces,
                                uint64_t seed, rt_block_fn on_block, void* user, float* accum_rgb, rt_stats* stats) {
    RT_HOST_GUARD_BEGIN
    if (!w) return host_fail("rt_world_render_progressive: NULL world");
    if (spp == 0) return host_fail("rt_world_render_progressive: spp must be positive");
    if (width < 2 || height < 2) return host_fail("rt_world_render_progressive: width and height must be at least 2");
    return w->renderer->render_progressive(width, height, spp, max_bounces, seed, on_block, user, accum_rgb, stats);
    RT_HOST_GUARD_END("rt_world_render_progressive")
}

int rt_write_ppm(const char* path, const uint32_t* px, uint32_t width, uint32_t height) {
    RT_HOST_GUARD_BEGIN
    if (!path || !px) return host_fail("rt_write_ppm: NULL argument");
    FILE* f = std::fopen(path, "wb");
    if (!f) return host_fail(std::string("rt_write_ppm: cannot open ") + path);
    std::fprintf(f, "P6\n%u %u\n255\n", width, height);
    std::vector<unsigned char> row((size_t)width * 3);
    bool ok = true;
    for (uint32_t y = 0; y < height && ok; y++) {
        for (uint32_t x = 0; x < width; x++) {
            const uint32_t c = px[(size_t)y * width + x];
            row[x * 3] = (unsigned char)(c >> 16); row[x * 3 + 1] = (unsigned char)(c >> 8); row[x * 3 + 2] = (unsigned char)c;
        }
        ok = std::fwrite(row.data(), 1, row.size(), f) == row.size();
    }
    ok = std::fclose(f) == 0 && ok;
    return ok ? RT_OK : host_fail(std::string("rt_write_ppm: write failed: ") + path);
    RT_HOST_GUARD_END("rt_write_ppm")
}

int rt_obj_count(const char* path, size_t* n_tris, size_t* n_models, size_t* n_faces) {
    RT_HOST_GUARD_BEGIN
    if (!path) return host_fail("rt_obj_count: NULL path");
    ObjData d;
    const std::string err = load_obj(path, d);
    if (!err.empty()) return host_fail(err);
    size_t t = 0;
    for (const ObjModel& m : d.models) t += m.vi.size() / 3;
    if (n_tris) *n_tris = t;
    if (n_models) *n_models = d.models.size();
    if (n_faces) *n_faces = d.n_faces;
    return RT_OK;
    RT_HOST_GUARD_END("rt_obj_count")
}

}  // extern "C"
