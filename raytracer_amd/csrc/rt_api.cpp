// C-ABI glue (include/rt_abi.h): contexts, scene upload, the host side of the wavefront loop.
// There is no CPU fallback anywhere in this file: without a working HIP device every entry
// point that needs one fails with RT_ERR_DEVICE.

#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <new>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <string>
#include <thread>
#include <vector>

#include "rt_internal.hpp"

using namespace rtamd;

namespace {

thread_local std::string g_last_error;
std::atomic<int> g_default_build{RT_BUILD_AUTO};        // rt_set_default_build
std::mutex g_cache_dir_lock;
std::string g_default_cache_dir;                        // rt_set_default_cache_dir

int fail(int code, const std::string& msg) {
    g_last_error = msg;
    return code;
}
}  // namespace
// lets the host scene layer (host/rt_host.cpp) report through rt_last_error()
void rt_internal_set_error(const char* msg) { g_last_error = msg ? msg : ""; }
namespace {
#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess)                                                                      \
            return fail(RT_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_));         \
    } while (0)

// No exception may cross the C ABI: the entry points that allocate on the host (or start threads)
// run inside these brackets and turn what is thrown into a status and a message.
#define RT_GUARD_BEGIN try {
#define RT_GUARD_END(name)                                                                         \
    } catch (const std::bad_alloc&) {                                                              \
        return fail(RT_ERR_NOMEM, name ": out of host memory");                                    \
    } catch (const std::exception& e) {                                                            \
        return fail(RT_ERR_INVALID, std::string(name ": ") + e.what());                            \
    } catch (...) {                                                                                \
        return fail(RT_ERR_INVALID, name ": unknown exception");                                   \
    }

struct DeviceBuffer {
    void* p = nullptr;
    size_t bytes = 0;
    ~DeviceBuffer() { release(); }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
    }
    hipError_t ensure(size_t n) {
        if (n <= bytes) return hipSuccess;
        release();
        hipError_t e = hipMalloc(&p, n);
        if (e == hipSuccess) bytes = n;
        return e;
    }
};

}  // namespace

// One wavefront pipeline: a stream and the state of the batch in flight on it.  A render keeps
// two batches in flight (one per pipeline) so that the latency-bound stages of one batch
// (shade, raygen, accumulate) run under the issue-bound closest-hit kernel of the other.
struct Pipe {
    hipStream_t stream = nullptr;
    DeviceBuffer rayA[2], rayB[2], rayC[2], hits, queue[3], lbuf, segcnt, qcnt;
    size_t capacity = 0;
    std::vector<hipEvent_t> events;     // profiling pool
    hipEvent_t accum_done = nullptr;    // this pipe's last accumulate
    hipEvent_t joined = nullptr;        // end of this pipe's work in a render call
};

constexpr int MAX_PIPES = 4;

// A host thread that lives as long as its multi-GPU context and drives one peer GPU: rt_render_multi
// hands it one job per call instead of starting a thread per GPU and call.  Jobs must not throw.
struct HostWorker {
    std::mutex m;
    std::condition_variable cv;
    std::function<void()> job;
    bool has_job = false, done = true, quit = false;
    std::thread th;
    HostWorker() : th([this]() { loop(); }) {}
    ~HostWorker() {
        { std::lock_guard<std::mutex> g(m); quit = true; }
        cv.notify_all();
        if (th.joinable()) th.join();
    }
    void loop() {
        std::unique_lock<std::mutex> lk(m);
        for (;;) {
            cv.wait(lk, [&]() { return has_job || quit; });
            if (quit) return;
            std::function<void()> j = std::move(job);
            has_job = false;
            lk.unlock();
            j();
            lk.lock();
            done = true;
            cv.notify_all();
        }
    }
    void submit(std::function<void()> j) {
        { std::lock_guard<std::mutex> g(m); job = std::move(j); has_job = true; done = false; }
        cv.notify_all();
    }
    void wait() {
        std::unique_lock<std::mutex> lk(m);
        cv.wait(lk, [&]() { return done; });
    }
};

struct rt_context {
    int device = 0;
    hipStream_t stream = nullptr;       // = pipes[0].stream: the stream callers may order work on
    uint32_t grid_cap = 2048;
    Pipe pipes[MAX_PIPES];
    DeviceBuffer totals, scratch_in, scratch_out;
    DeviceBuffer trace_partials;        // rt_trace_device with counts: one record per workgroup
    DeviceBuffer clock_stamps;          // profiled renders: four words per launch of the per-batch kernel
    DeviceBuffer frame_in, frame_out;   // rt_render / rt_tonemap: device copies of the caller's host buffers
    void* static_scratch = nullptr;     // this device's copy of the library's static array (rt_trace_device_static)
    uint32_t wave_target = 8192, wave_target_sweep = 8192;
    int n_pipes = 2;
    hipEvent_t ev_fork = nullptr;
    hipEvent_t ev_begin = nullptr, ev_end = nullptr;   // bracket a render call (rt_stats.ms_render)
    // The owner holds one reference and every scene created on the context another, so a scene
    // destroyed after rt_context_destroy() still finds its device and streams; work on a
    // destroyed context is refused.
    std::atomic<int> refs{1};
    bool destroyed = false;
    // A multi-GPU context is its root (this object, GPU dev_ids[0]) plus one context per further GPU.
    std::vector<rt_context*> peers;     // GPUs 1 .. n-1 (owned)
    std::vector<ncclComm_t> comms;      // one communicator per GPU, [0] the root's; empty when the GPUs are not distinct
    bool multi = false;                 // made by rt_context_create_multi (even with one GPU)
    DeviceBuffer frame;                 // this GPU's full-frame accumulation buffer of a rt_render_multi call
    DeviceBuffer stage;                 // root: where a peer's frame lands before it is added (ordered reduce)
    int reduce_mode = RT_REDUCE_RCCL;   // rt_context_set_reduce
    std::vector<std::unique_ptr<HostWorker>> workers;   // root: one host thread per peer GPU

    LaunchCfg cfg(int pipe = 0) const { return LaunchCfg{(void*)pipes[pipe].stream, grid_cap, wave_target}; }
    ~rt_context() {
        for (Pipe& p : pipes) {
            for (hipEvent_t e : p.events) (void)hipEventDestroy(e);
            if (p.accum_done) (void)hipEventDestroy(p.accum_done);
            if (p.joined) (void)hipEventDestroy(p.joined);
            if (p.stream) (void)hipStreamDestroy(p.stream);
        }
        if (ev_fork) (void)hipEventDestroy(ev_fork);
        if (ev_begin) (void)hipEventDestroy(ev_begin);
        if (ev_end) (void)hipEventDestroy(ev_end);
    }
};

// One flattened form of a scene's trees on one GPU: the reference-topology trees, or the
// surface-area trees of RT_BUILD_SAH / RT_BUILD_AUTO.  The per-primitive tables are the scene's.
struct SceneTree {
    bool present = false;
    uint32_t n_entries = 0, n_units = 0, n_walk = 0, walk_top_units = 0, depth = 0;
    bool ordered_ok = false;
    bool multi_child = false;           // a box with more than two children: no ordered walk (the octant walk is fine)
    bool lds_walk = false;              // the linked units fit the LDS of a workgroup
    double expected_cost = 0.0;         // expected_cost_per_ray of its entries (0: not known)
    DeviceBuffer d_entries, d_units, d_walk, d_units_oct;
    DeviceBuffer d_wide_nodes, d_wide_prims;            // RT_TRAVERSAL_WIDE: made on first use
    WideScene wide;                                     // host-only scenes keep it for rt_scene_get_wide()
};
enum { TREE_REFERENCE = 0, TREE_FAST = 1 };

struct rt_scene {
    rt_context* ctx = nullptr;
    std::vector<rt_material> mats;
    std::vector<SceneObject> objects;
    std::vector<SceneGroup> groups{SceneGroup{}};      // [0]: the top level
    std::vector<size_t> open_groups{0};                // rt_scene_begin_group nesting
    int root_kind = RT_ROOT_BVH;
    int build_mode = RT_BUILD_AUTO;
    int traversal = RT_TRAVERSAL_AUTO;
    bool auto_tuned = false;            // the first render has timed the forms of the closest-hit loop that apply
    uint32_t env_prim_wait = 0;         // RT_PRIM_WAIT, 0 when unset
    uint32_t env_refill = 0;            // RT_REFILL, 0 when unset
    bool committed = false;
    std::string cache_dir;              // rt_scene_set_cache_dir; empty: the scene is always built
    int cache_state = RT_CACHE_OFF;
    FlatScene flat;                     // the tree host-only scenes show: the reference's (RT_BUILD_SAH: the surface-area one)
    SceneTree trees[2];                 // [TREE_REFERENCE], [TREE_FAST]; which exist follows from the build mode
    int cur = TREE_REFERENCE;           // the tree `dev` is bound to
    DeviceBuffer d_geo, d_meta, d_order, d_mats, d_xforms;
    int wide_build = RT_WIDE_BUILD_COLLAPSE;            // where the wide tree is made: rt_scene_set_wide_build
    DevScene dev{};
    std::vector<rt_scene*> replicas;    // the committed scene on GPUs 1 .. n-1 of a multi-GPU context (owned)

    SceneTree& tree() { return trees[cur]; }
    const SceneTree& tree() const { return trees[cur]; }
    // the tree explicit traversal modes run on: the reference's wherever the scene has it
    int base_tree() const { return trees[TREE_REFERENCE].present ? TREE_REFERENCE : TREE_FAST; }
};

#ifdef RT_STAMPS
namespace rtamd { void dump_stamps(); }
#endif

namespace {

int set_device(rt_context* ctx) {
    if (ctx->destroyed) return fail(RT_ERR_STATE, "the context of this scene has been destroyed");
    HIP_TRY(hipSetDevice(ctx->device));
    return RT_OK;
}

void release_context(rt_context* ctx) {
    if (ctx->refs.fetch_sub(1) == 1) {
        (void)hipSetDevice(ctx->device);
        delete ctx;
    }
}

int ensure_capacity(rt_context* ctx, Pipe& pp, size_t paths, size_t n_seg) {
    HIP_TRY(ctx->totals.ensure(sizeof(Totals)));
    HIP_TRY(pp.segcnt.ensure((size_t)(MAX_DEPTH + 1) * n_seg * sizeof(uint32_t)));
    HIP_TRY(pp.qcnt.ensure((size_t)3 * n_seg * sizeof(uint32_t)));
    if (paths <= pp.capacity) return RT_OK;
    // a grow that fails part-way leaves some buffers released: the pipeline then holds nothing
    // until every buffer has been resized
    pp.capacity = 0;
    for (int k = 0; k < 2; k++) {
        HIP_TRY(pp.rayA[k].ensure(paths * 16));
        HIP_TRY(pp.rayB[k].ensure(paths * 16));
        HIP_TRY(pp.rayC[k].ensure(paths * 16));
    }
    HIP_TRY(pp.hits.ensure(paths * 8));
    for (int k = 0; k < 3; k++) HIP_TRY(pp.queue[k].ensure(paths * 4));
    HIP_TRY(pp.lbuf.ensure(paths * 16));
    pp.capacity = paths;
    return RT_OK;
}

int get_events(Pipe& pp, size_t n) {
    while (pp.events.size() < n) {
        hipEvent_t e;
        HIP_TRY(hipEventCreate(&e));
        pp.events.push_back(e);
    }
    return RT_OK;
}

// paths in flight over both pipelines: ~132 B of wavefront state each (8.4 GB of the 288 GB);
// long launches amortise the tail in which the last waves of a kernel run alone
constexpr size_t DEFAULT_MAX_PATHS = 64u << 20;
#ifndef RT_SPLIT_MI
#define RT_SPLIT_MI 16
#endif

}  // namespace

extern "C" {

const char* rt_last_error(void) { return g_last_error.c_str(); }

int rt_context_create(int device_id, rt_context** out) {
    RT_GUARD_BEGIN
    if (!out) return fail(RT_ERR_INVALID, "rt_context_create: out is NULL");
    *out = nullptr;
    int n = 0;
    HIP_TRY(hipGetDeviceCount(&n));
    if (device_id < 0 || device_id >= n)
        return fail(RT_ERR_DEVICE, "rt_context_create: no such HIP device " + std::to_string(device_id) +
                                       " (" + std::to_string(n) + " visible)");
    std::unique_ptr<rt_context> ctx(new rt_context());
    ctx->device = device_id;
    HIP_TRY(hipSetDevice(device_id));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device_id));
    ctx->grid_cap = (uint32_t)std::max(1, prop.multiProcessorCount) * 8u;
    // Segments (= waves) per batch.  Per-lane walk: 256 per CU, eight per wave slot of the chip
    // (4 SIMDs x 8 waves per CU); segments differ a lot in how long their paths live, and the
    // surplus lets the dispatcher even that out (10 M-triangle soup at depth 16: +75 % over two per
    // slot; the sphere and OBJ scenes +0-10 %).  Wave sweep: one per wave slot (its cost per group
    // of rays hardly depends on the rays, and short segments waste partly filled groups).
    ctx->wave_target = (uint32_t)std::max(1, prop.multiProcessorCount) * 256u;
    ctx->wave_target_sweep = (uint32_t)std::max(1, prop.multiProcessorCount) * 32u;
    if (const char* e = std::getenv("RT_WAVE_TARGET")) {
        const long v = std::atol(e);
        if (v > 0) ctx->wave_target = ctx->wave_target_sweep = (uint32_t)v;
    }
    for (Pipe& pp : ctx->pipes) {
        HIP_TRY(hipStreamCreateWithFlags(&pp.stream, hipStreamNonBlocking));
        HIP_TRY(hipEventCreateWithFlags(&pp.accum_done, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&pp.joined, hipEventDisableTiming));
    }
    ctx->stream = ctx->pipes[0].stream;
    HIP_TRY(hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming));
    HIP_TRY(hipEventCreate(&ctx->ev_begin));
    HIP_TRY(hipEventCreate(&ctx->ev_end));
    if (const char* e = std::getenv("RT_PIPES")) ctx->n_pipes = std::min(MAX_PIPES, std::max(1, std::atoi(e)));
    if (static_scratch_address(device_id, &ctx->static_scratch))
        return fail(RT_ERR_DEVICE, "rt_context_create: hipGetSymbolAddress of the static scratch array failed");
    *out = ctx.release();
    return RT_OK;
    RT_GUARD_END("rt_context_create")
}

int rt_context_create_multi(int n_dev, const int* dev_ids, rt_context** out) {
    RT_GUARD_BEGIN
    if (!out) return fail(RT_ERR_INVALID, "rt_context_create_multi: out is NULL");
    *out = nullptr;
    if (n_dev < 1 || n_dev > 64 || !dev_ids) return fail(RT_ERR_INVALID, "rt_context_create_multi: needs 1 to 64 device ids");
    bool distinct = true;
    for (int i = 0; i < n_dev; i++)
        for (int j = 0; j < i; j++) distinct = distinct && dev_ids[i] != dev_ids[j];
    rt_context* root = nullptr;
    int rc = rt_context_create(dev_ids[0], &root);
    if (rc) return rc;
    root->multi = true;
    if (const char* e = std::getenv("RT_REDUCE_ORDERED"))
        if (std::atoi(e) != 0) root->reduce_mode = RT_REDUCE_ORDERED;
    for (int i = 1; i < n_dev; i++) {
        rt_context* peer = nullptr;
        rc = rt_context_create(dev_ids[i], &peer);
        if (rc) { rt_context_destroy(root); return rc; }
        root->peers.push_back(peer);
    }
    try {
        for (int i = 1; i < n_dev; i++) root->workers.emplace_back(new HostWorker());
    } catch (...) {
        rt_context_destroy(root);
        throw;
    }
    if (distinct) {                            // one RCCL rank per GPU, all driven by this process
        root->comms.assign((size_t)n_dev, nullptr);
        const ncclResult_t nr = ncclCommInitAll(root->comms.data(), n_dev, dev_ids);
        if (nr != ncclSuccess) {
            root->comms.clear();
            rt_context_destroy(root);
            return fail(RT_ERR_DEVICE, std::string("rt_context_create_multi: ncclCommInitAll: ") + ncclGetErrorString(nr));
        }
    }
    *out = root;
    return RT_OK;
    RT_GUARD_END("rt_context_create_multi")
}

int rt_context_num_devices(const rt_context* ctx) { return ctx ? 1 + (int)ctx->peers.size() : 0; }

int rt_context_set_reduce(rt_context* ctx, int mode) {
    if (!ctx) return fail(RT_ERR_INVALID, "rt_context_set_reduce: NULL context");
    if (mode != RT_REDUCE_RCCL && mode != RT_REDUCE_ORDERED) return fail(RT_ERR_INVALID, "rt_context_set_reduce: unknown mode");
    if (!ctx->multi) return fail(RT_ERR_STATE, "rt_context_set_reduce needs a context made by rt_context_create_multi");
    ctx->reduce_mode = mode;
    return RT_OK;
}

int rt_rccl_version(void) {
    int v = 0;
    return ncclGetVersion(&v) == ncclSuccess ? v : 0;
}

void rt_context_destroy(rt_context* ctx) {
    if (!ctx) return;
    if (ctx->destroyed) return;
    ctx->workers.clear();                      // joins the peers' host threads
    for (ncclComm_t c : ctx->comms)
        if (c) (void)ncclCommDestroy(c);
    ctx->comms.clear();
    for (rt_context* peer : ctx->peers) rt_context_destroy(peer);
    ctx->peers.clear();
    (void)hipSetDevice(ctx->device);
    for (int i = 0; i < ctx->n_pipes; i++)
        if (ctx->pipes[i].stream) (void)hipStreamSynchronize(ctx->pipes[i].stream);
    ctx->destroyed = true;
    release_context(ctx);
}

void* rt_context_stream(rt_context* ctx) { return ctx ? (void*)ctx->stream : nullptr; }

int rt_context_synchronize(rt_context* ctx) {
    if (!ctx) return fail(RT_ERR_INVALID, "rt_context_synchronize: NULL context");
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return RT_OK;
}

int rt_scene_create(rt_context* ctx, rt_scene** out) {
    RT_GUARD_BEGIN
    // ctx may be NULL: the scene is then host-only (it can be committed and its flattened
    // entries inspected, but nothing can be traced or rendered with it)
    if (!out) return fail(RT_ERR_INVALID, "rt_scene_create: NULL argument");
    if (ctx && ctx->destroyed) return fail(RT_ERR_STATE, "rt_scene_create: the context has been destroyed");
    *out = new rt_scene();
    (*out)->ctx = ctx;
    (*out)->build_mode = g_default_build.load();
    {
        std::lock_guard<std::mutex> guard(g_cache_dir_lock);
        (*out)->cache_dir = g_default_cache_dir;
    }
    if (ctx) ctx->refs.fetch_add(1);
    return RT_OK;
    RT_GUARD_END("rt_scene_create")
}

void rt_scene_destroy(rt_scene* scene) {
    if (!scene) return;
    for (rt_scene* r : scene->replicas) rt_scene_destroy(r);
    scene->replicas.clear();
    rt_context* ctx = scene->ctx;
    if (ctx) {
        (void)hipSetDevice(ctx->device);
        for (int i = 0; i < ctx->n_pipes; i++)
            if (ctx->pipes[i].stream) (void)hipStreamSynchronize(ctx->pipes[i].stream);
    }
    delete scene;                              // frees the scene's device arrays
    if (ctx) release_context(ctx);
}

int rt_scene_set_materials(rt_scene* s, const rt_material* mats, size_t n) {
    RT_GUARD_BEGIN
    if (!s || (!mats && n)) return fail(RT_ERR_INVALID, "rt_scene_set_materials: NULL argument");
    if (s->committed) return fail(RT_ERR_STATE, "scene already committed");
    s->mats.assign(mats, mats + n);
    return RT_OK;
    RT_GUARD_END("rt_scene_set_materials")
}

int rt_scene_add_spheres(rt_scene* s, const rt_sphere* sp, size_t n) {
    RT_GUARD_BEGIN
    if (!s || (!sp && n)) return fail(RT_ERR_INVALID, "rt_scene_add_spheres: NULL argument");
    if (s->committed) return fail(RT_ERR_STATE, "scene already committed");
    for (size_t i = 0; i < n; i++) {
        SceneObject o;
        o.is_mesh = false;
        o.sphere = sp[i];
        s->groups[s->open_groups.back()].items.push_back((long)s->objects.size());
        s->objects.push_back(std::move(o));
    }
    return RT_OK;
    RT_GUARD_END("rt_scene_add_spheres")
}

int rt_scene_add_mesh(rt_scene* s, const rt_triangle* tris, size_t n, const float* xform) {
    RT_GUARD_BEGIN
    if (!s || (!tris && n)) return fail(RT_ERR_INVALID, "rt_scene_add_mesh: NULL argument");
    if (s->committed) return fail(RT_ERR_STATE, "scene already committed");
    SceneObject o;
    o.is_mesh = true;
    if (xform) {
        static const float ident[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
        if (std::memcmp(xform, ident, sizeof(ident)) != 0) {
            for (int k = 0; k < 16; k++)
                if (!(xform[k] == xform[k]) || xform[k] - xform[k] != 0.0f)
                    return fail(RT_ERR_INVALID, "rt_scene_add_mesh: the transform holds a NaN or an infinity");
            o.has_xform = true;
            std::memcpy(o.xform, xform, sizeof(o.xform));
        }
    }
    o.tris.assign(tris, tris + n);
    s->groups[s->open_groups.back()].items.push_back((long)s->objects.size());
    s->objects.push_back(std::move(o));
    return RT_OK;
    RT_GUARD_END("rt_scene_add_mesh")
}

int rt_scene_begin_group(rt_scene* s, int kind, int axis) {
    RT_GUARD_BEGIN
    if (!s) return fail(RT_ERR_INVALID, "rt_scene_begin_group: NULL scene");
    if (s->committed) return fail(RT_ERR_STATE, "scene already committed");
    if (kind != RT_ROOT_BVH && kind != RT_ROOT_LIST) return fail(RT_ERR_INVALID, "rt_scene_begin_group: the container is RT_ROOT_BVH or RT_ROOT_LIST");
    if (axis < 0 || axis > 2) return fail(RT_ERR_INVALID, "rt_scene_begin_group: axis must be 0, 1 or 2");
    if (s->open_groups.size() >= 64) return fail(RT_ERR_UNSUPPORTED, "rt_scene_begin_group: containers nested deeper than 64");
    SceneGroup g;
    g.kind = kind;
    g.axis = axis;
    s->groups.push_back(std::move(g));
    s->groups[s->open_groups.back()].items.push_back(-1 - (long)(s->groups.size() - 1));
    s->open_groups.push_back(s->groups.size() - 1);
    return RT_OK;
    RT_GUARD_END("rt_scene_begin_group")
}

int rt_scene_end_group(rt_scene* s) {
    if (!s) return fail(RT_ERR_INVALID, "rt_scene_end_group: NULL scene");
    if (s->committed) return fail(RT_ERR_STATE, "scene already committed");
    if (s->open_groups.size() < 2) return fail(RT_ERR_STATE, "rt_scene_end_group without rt_scene_begin_group");
    s->open_groups.pop_back();
    return RT_OK;
}

int rt_scene_set_root(rt_scene* s, int root_kind) {
    if (!s) return fail(RT_ERR_INVALID, "rt_scene_set_root: NULL scene");
    if (s->committed) return fail(RT_ERR_STATE, "scene already committed");
    if (root_kind < RT_ROOT_BVH || root_kind > RT_ROOT_LIST) return fail(RT_ERR_INVALID, "unknown root kind");
    s->root_kind = root_kind;
    return RT_OK;
}

// binds the per-tree fields of s->dev to one of the scene's trees
static void bind_tree(rt_scene* s, int which) {
    const SceneTree& t = s->trees[which];
    s->cur = which;
    s->dev.entries = t.d_entries.p;
    s->dev.units = t.d_units.p;
    s->dev.walk = t.d_walk.p;
    s->dev.units_oct = t.d_units_oct.p;
    s->dev.n_entries = t.n_entries;
    s->dev.n_units = t.n_units;
    s->dev.n_walk = t.n_walk;
    s->dev.walk_top_units = t.walk_top_units;
    s->dev.stack_depth = std::max<uint32_t>(t.depth, 1u);
    s->dev.lds_walk = t.lds_walk ? 1u : 0u;
    s->dev.wide_nodes = t.d_wide_nodes.p;
    s->dev.wide_prims = t.d_wide_prims.p;
    // (a scene with transformed meshes: one slot more for the marker under a mesh's subtree, and six behind the
    // stack for the caller's ray while the lane is inside a mesh)
    s->dev.wide_stack = std::max(1u, t.wide.depth) + (s->dev.n_xforms ? 1u + WIDE_XF_RAY_SLOTS : 0u);
    s->dev.wide_stride = t.wide.node_words / 4u;
}

// the eight direction-ordered copies of the bound tree's units (octant walk), made on first use
static int ensure_octant_units(rt_scene* s) {
    SceneTree& t = s->tree();
    if (!t.ordered_ok)
        return fail(RT_ERR_UNSUPPORTED, "octant traversal needs fewer than 2^29 entries and boxes nested at most 64 deep");
    if (((uint64_t)t.n_units + UNITS_END) * 8u > SKIP_MASK)
        return fail(RT_ERR_UNSUPPORTED, "octant traversal: eight copies of the trees exceed the 30-bit links");
    if (!s->ctx) return fail(RT_ERR_DEVICE, "octant traversal needs a device context");
    if (!t.d_units_oct.p) {
        int rc = set_device(s->ctx);
        if (rc) return rc;
        std::vector<uint32_t> units((size_t)t.n_units * 4), oct;
        HIP_TRY(hipMemcpy(units.data(), t.d_units.p, units.size() * 4, hipMemcpyDeviceToHost));
        build_octant_units(units.data(), t.n_units, oct);
        HIP_TRY(t.d_units_oct.ensure(oct.size() * 4));
        HIP_TRY(hipMemcpy(t.d_units_oct.p, oct.data(), oct.size() * 4, hipMemcpyHostToDevice));
    }
    return RT_OK;
}

// the wide tree of the bound tree, made on first use: built on this GPU from the primitives, or
// collapsed on the host from the entry array (once for all GPUs of a context)
static int ensure_wide_tree(rt_scene* s) {
    SceneTree& t = s->tree();
    const int which = s->cur;
    if (!t.d_wide_nodes.p && s->wide_build == RT_WIDE_BUILD_DEVICE) {
        if (s->dev.n_xforms != 0)
            return fail(RT_ERR_UNSUPPORTED, "wide traversal: the device-side build does not take transformed meshes (RT_WIDE_BUILD_COLLAPSE does)");
        int rc = set_device(s->ctx);
        if (rc) return rc;
        WideDeviceTree w;
        std::string err;
        const auto t_begin = std::chrono::steady_clock::now();
        if (build_wide_on_device((void*)s->ctx->stream, t.d_entries.p, t.n_entries, w, err))
            return fail(RT_ERR_DEVICE, "wide traversal, device build: " + err);
        if (std::getenv("RT_BUILD_TIMING"))
            std::fprintf(stderr, "[rt build] wide tree on the GPU         %.3f s (%u node slots, %u primitives, %u levels)\n",
                         std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count(), w.n_slots,
                         w.n_prims, w.depth);
        t.d_wide_nodes.release(); t.d_wide_nodes.p = w.nodes; t.d_wide_nodes.bytes = w.nodes_bytes;
        t.d_wide_prims.release(); t.d_wide_prims.p = w.prims; t.d_wide_prims.bytes = w.prims_bytes;
        t.wide.depth = w.depth;
        t.wide.node_words = w.node_words;
    }
    if (!t.d_wide_nodes.p) {
        int rc = set_device(s->ctx);
        if (rc) return rc;
        std::vector<float> entries((size_t)t.n_entries * ENTRY_FLOATS);
        HIP_TRY(hipMemcpy(entries.data(), t.d_entries.p, entries.size() * 4, hipMemcpyDeviceToHost));
        std::vector<float> xforms((size_t)s->dev.n_xforms * 32);
        if (!xforms.empty()) HIP_TRY(hipMemcpy(xforms.data(), s->d_xforms.p, xforms.size() * 4, hipMemcpyDeviceToHost));
        WideScene w;
        const auto t_begin = std::chrono::steady_clock::now();
        const std::string err = build_wide_scene(entries.data(), t.n_entries, w, xforms.empty() ? nullptr : xforms.data(), s->dev.n_xforms);
        if (!err.empty()) return fail(RT_ERR_UNSUPPORTED, "wide traversal: " + err);
        if (std::getenv("RT_BUILD_TIMING"))
            std::fprintf(stderr, "[rt build] wide tree                    %.3f s (%zu node slots, %zu primitives, %u levels)\n",
                         std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count(), w.n_nodes(),
                         w.n_prims(), w.depth);
        entries.clear(); entries.shrink_to_fit();
        // the tree is built once: the scene's copies on the other GPUs of a multi-GPU context get the same arrays
        std::vector<rt_scene*> targets{s};
        for (rt_scene* r : s->replicas)
            if (!r->trees[which].d_wide_nodes.p) targets.push_back(r);
        for (rt_scene* g : targets) {
            SceneTree& gt = g->trees[which];
            rc = set_device(g->ctx);
            if (rc) return rc;
            gt.wide.depth = w.depth;
            gt.wide.node_words = w.node_words;
            HIP_TRY(gt.d_wide_prims.ensure(std::max<size_t>(w.prims.size() * 4, 64)));
            HIP_TRY(hipMemcpy(gt.d_wide_prims.p, w.prims.data(), w.prims.size() * 4, hipMemcpyHostToDevice));
            HIP_TRY(gt.d_wide_nodes.ensure(w.nodes.size() * 4));
            HIP_TRY(hipMemcpy(gt.d_wide_nodes.p, w.nodes.data(), w.nodes.size() * 4, hipMemcpyHostToDevice));
        }
        rc = set_device(s->ctx);
        if (rc) return rc;
    }
    return RT_OK;
}

// the kernel form a committed scene is traced with.  An explicit mode runs on the scene's base tree
// (the reference's trees unless the scene was built with RT_BUILD_SAH alone); AUTO starts there too,
// and the first render may move a scene of RT_BUILD_AUTO to its other tree.
static int apply_traversal(rt_scene* s, int mode) {
    bind_tree(s, s->base_tree());
    const SceneTree& t = s->tree();
    if (s->dev.n_xforms != 0 && mode != RT_TRAVERSAL_AUTO && mode != RT_TRAVERSAL_PER_LANE && mode != RT_TRAVERSAL_WIDE)
        return fail(RT_ERR_UNSUPPORTED, "a scene with transformed meshes is traced by the per-lane walk or the wide walk only");
    if (mode == RT_TRAVERSAL_ORDERED && t.multi_child)
        return fail(RT_ERR_UNSUPPORTED, "ordered traversal takes boxes of at most two children (no leaves of several triangles, "
                                        "no list inside a Bvh); the octant traversal visits the same entries without that limit");
    if (mode == RT_TRAVERSAL_ORDERED && !t.ordered_ok)
        return fail(RT_ERR_UNSUPPORTED, "ordered traversal needs fewer than 2^29 entries and boxes nested at most 64 deep");
    if (mode == RT_TRAVERSAL_OCTANT) {
        int rc = ensure_octant_units(s);
        if (rc) return rc;
    }
    if (mode == RT_TRAVERSAL_WIDE && s->ctx) {
        int rc = ensure_wide_tree(s);
        if (rc) return rc;
    }
    bind_tree(s, s->cur);                      // (the arrays made above)
    {   // experiment knobs, read whenever the kernel form is (re)chosen.  RT_PRIM_WAIT: lanes that must
        // wait at primitives before their routine runs (best of {1 .. 64} measured on the sphere and OBJ
        // scenes, from memory and out of LDS alike: 8; 2 .. 12 lie within 3 %).  RT_REFILL: idle lanes
        // that trigger a refill in the fused kernel's per-lane walk (pins that form; 0 pins the plain one).
        const char* e = std::getenv("RT_PRIM_WAIT");
        s->env_prim_wait = e ? (uint32_t)std::max(1, std::atoi(e)) : 0u;
        e = std::getenv("RT_REFILL");
        s->env_refill = e ? (uint32_t)std::min(64, std::max(0, std::atoi(e))) : 0u;
    }
    s->dev.octant = mode == RT_TRAVERSAL_OCTANT ? 1u : 0u;
    s->dev.ordered = mode == RT_TRAVERSAL_ORDERED ? 1u : 0u;
    s->dev.wide = mode == RT_TRAVERSAL_WIDE && s->ctx ? 1u : 0u;
    // out of the reference's visiting order (another tree, or nearer child first) exact ties in t are
    // resolved by primitive key
    s->dev.tie = (s->cur != TREE_REFERENCE || s->dev.octant || s->dev.ordered || s->dev.wide) ? 1u : 0u;
    s->auto_tuned = false;
    s->dev.refill = s->env_refill;
    s->dev.prim_wait = s->env_prim_wait ? s->env_prim_wait : 8u;
    s->dev.sweep = mode == RT_TRAVERSAL_SWEEP ||
                   (mode == RT_TRAVERSAL_AUTO && s->dev.n_entries <= SWEEP_MAX_ENTRIES) ? 1u : 0u;
    s->dev.split_top = 0;
    s->dev.deep_wait = 0;
    if (const char* e = std::getenv("RT_SPLIT")) {     // experiment: the split walk, RT_SPLIT = lanes a fetch from memory waits for
        const uint32_t wait = (uint32_t)std::min(64, std::max(0, std::atoi(e)));
        const uint64_t top_bytes = (uint64_t)s->dev.walk_top_units * 16u;
        if (wait > 0 && mode != RT_TRAVERSAL_ORDERED && mode != RT_TRAVERSAL_OCTANT && !s->dev.sweep && s->dev.n_xforms == 0 &&
            top_bytes > 0 && top_bytes <= SPLIT_TOP_MAX_BYTES && s->dev.walk_top_units < s->dev.n_walk && !s->dev.lds_walk &&
            !s->dev.tie) {
            s->dev.split_top = s->dev.walk_top_units;      // (whole sibling groups: no entry straddles the boundary)
            s->dev.deep_wait = wait;
            if (s->dev.refill == 0) s->dev.refill = 16;
            s->auto_tuned = true;
        }
    }
    if (s->dev.wide && !std::getenv("RT_REFILL")) s->dev.refill = 16;      // the wide walk refills idle lanes by default
    if (s->dev.n_xforms != 0) s->dev.sweep = 0;   // the per-lane walks (from memory, out of LDS, refilled) follow the transform markers
    return RT_OK;
}

static bool known_build(int m) { return m == RT_BUILD_REFERENCE || m == RT_BUILD_SAH || m == RT_BUILD_AUTO; }

int rt_set_default_build(int build_mode) {
    if (!known_build(build_mode)) return fail(RT_ERR_INVALID, "unknown build mode");
    g_default_build.store(build_mode);
    return RT_OK;
}

int rt_scene_set_build(rt_scene* s, int build_mode) {
    if (!s) return fail(RT_ERR_INVALID, "rt_scene_set_build: NULL scene");
    if (s->committed) return fail(RT_ERR_STATE, "scene already committed");
    if (!known_build(build_mode)) return fail(RT_ERR_INVALID, "unknown build mode");
    s->build_mode = build_mode;
    return RT_OK;
}

int rt_set_default_cache_dir(const char* dir) {
    RT_GUARD_BEGIN
    std::lock_guard<std::mutex> guard(g_cache_dir_lock);
    g_default_cache_dir = dir ? dir : "";
    return RT_OK;
    RT_GUARD_END("rt_set_default_cache_dir")
}

int rt_scene_set_cache_dir(rt_scene* s, const char* dir) {
    RT_GUARD_BEGIN
    if (!s) return fail(RT_ERR_INVALID, "rt_scene_set_cache_dir: NULL scene");
    if (s->committed) return fail(RT_ERR_STATE, "scene already committed");
    s->cache_dir = dir ? dir : "";
    return RT_OK;
    RT_GUARD_END("rt_scene_set_cache_dir")
}

int rt_scene_cache_state(const rt_scene* s) { return s ? s->cache_state : RT_CACHE_OFF; }

int rt_scene_set_wide_build(rt_scene* s, int where) {
    if (!s) return fail(RT_ERR_INVALID, "rt_scene_set_wide_build: NULL scene");
    if (where != RT_WIDE_BUILD_COLLAPSE && where != RT_WIDE_BUILD_DEVICE) return fail(RT_ERR_INVALID, "unknown wide build");
    if (s->trees[0].d_wide_nodes.p || s->trees[1].d_wide_nodes.p)
        return fail(RT_ERR_STATE, "rt_scene_set_wide_build: the scene's wide tree has been made already");
    s->wide_build = where;
    for (rt_scene* r : s->replicas) {
        int rc = rt_scene_set_wide_build(r, where);
        if (rc) return rc;
    }
    return RT_OK;
}

int rt_scene_set_traversal(rt_scene* s, int mode) {
    RT_GUARD_BEGIN
    if (!s) return fail(RT_ERR_INVALID, "rt_scene_set_traversal: NULL scene");
    if (mode < RT_TRAVERSAL_AUTO || mode > RT_TRAVERSAL_WIDE) return fail(RT_ERR_INVALID, "unknown traversal mode");
    if (s->committed) {
        int rc = apply_traversal(s, mode);
        if (rc) return rc;
    }
    s->traversal = mode;
    for (rt_scene* r : s->replicas) {          // the scene's copies on the other GPUs follow
        int rc = rt_scene_set_traversal(r, mode);
        if (rc) return rc;
    }
    return RT_OK;
    RT_GUARD_END("rt_scene_set_traversal")
}

// what a tree's SceneTree records of its flattened form (host-only scenes stop here)
static void describe_tree(SceneTree& t, const FlatScene& f) {
    t.present = true;
    t.n_entries = (uint32_t)f.n_entries();
    t.n_units = (uint32_t)f.n_units();
    t.n_walk = (uint32_t)f.n_walk();
    t.walk_top_units = f.walk_top_units;
    t.depth = f.depth;
    t.ordered_ok = f.ordered_ok;
    t.multi_child = f.multi_child;
    t.lds_walk = t.n_walk > 0 && ((size_t)t.n_walk + UNITS_END) * 16u <= LDS_WALK_MAX_BYTES && !std::getenv("RT_NO_LDS_WALK");
    t.expected_cost = expected_cost_per_ray(f.entries.data(), f.n_entries());
}

// uploads one flattened tree to the GPU of s->ctx
static int upload_tree(rt_scene* s, int which, const FlatScene& f) {
    rt_context* ctx = s->ctx;
    SceneTree& t = s->trees[which];
    describe_tree(t, f);
    int rc = set_device(ctx);
    if (rc) return rc;
    auto upload = [&](DeviceBuffer& b, const void* src, size_t bytes) -> hipError_t {
        hipError_t e = b.ensure(std::max<size_t>(bytes, 16));
        if (e != hipSuccess || bytes == 0) return e;
        return hipMemcpyAsync(b.p, src, bytes, hipMemcpyHostToDevice, ctx->stream);
    };
    {   // the entries, then the TAG_END entry the wave sweep stops at
        float end_entry[ENTRY_FLOATS] = {};
        const uint32_t w = TAG_END << TAG_SHIFT;
        std::memcpy(&end_entry[3], &w, 4);
        HIP_TRY(t.d_entries.ensure(f.entries.size() * 4 + sizeof(end_entry)));
        HIP_TRY(hipMemcpyAsync((char*)t.d_entries.p + f.entries.size() * 4, end_entry, sizeof(end_entry),
                               hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));    // end_entry leaves scope
    }
    HIP_TRY(upload(t.d_entries, f.entries.data(), f.entries.size() * 4));
    HIP_TRY(upload(t.d_units, f.units.data(), f.units.size() * 4));
    HIP_TRY(upload(t.d_walk, f.walk.data(), f.walk.size() * 4));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return RT_OK;
}

// uploads the per-primitive tables (the same for every tree of the scene) and chooses the kernel form
static int upload_tables(rt_scene* s, const FlatScene& f) {
    rt_context* ctx = s->ctx;
    int rc = set_device(ctx);
    if (rc) return rc;
    auto upload = [&](DeviceBuffer& b, const void* src, size_t bytes) -> hipError_t {
        hipError_t e = b.ensure(std::max<size_t>(bytes, 16));
        if (e != hipSuccess || bytes == 0) return e;
        return hipMemcpyAsync(b.p, src, bytes, hipMemcpyHostToDevice, ctx->stream);
    };
    HIP_TRY(upload(s->d_geo, f.prim_geo.data(), f.prim_geo.size() * 4));
    HIP_TRY(upload(s->d_meta, f.prim_meta.data(), f.prim_meta.size() * 4));
    HIP_TRY(upload(s->d_order, f.prim_order.data(), f.prim_order.size() * 4));
    HIP_TRY(upload(s->d_mats, f.materials.data(), f.materials.size() * 4));
    HIP_TRY(upload(s->d_xforms, f.xforms.data(), f.xforms.size() * 4));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    s->dev.n_prims = (uint32_t)f.n_prims();
    s->dev.prim_geo = s->d_geo.p;
    s->dev.prim_meta = (const uint32_t*)s->d_meta.p;
    s->dev.prim_order = (const uint32_t*)s->d_order.p;
    s->dev.materials = s->d_mats.p;
    s->dev.xforms = s->d_xforms.p;
    s->dev.n_xforms = (uint32_t)(f.xforms.size() / 32);
    s->dev.n_mats = (uint32_t)(f.materials.size() / 8);
    s->dev.kinds_present = f.kinds_present;
    s->dev.has_spheres = f.has_spheres ? 1u : 0u;
    // (kinds_present: bit k for material kind k; bits 0-2 are the kinds that scatter)
    // RT_FUSE (experiments): 0 never, 1 Lambert-only scenes, 2 every kind of every scene, 3 the Lambertian hits
    // of every scene (dev.fuse: 1 Lambert-only scene, 2 every kind, 3 Lambertian hits of a scene with several kinds)
    const int fuse_knob = std::getenv("RT_FUSE") ? std::atoi(std::getenv("RT_FUSE")) : 3;
    const bool lambert_only = (f.kinds_present & 7u) == (1u << RT_MAT_LAMBERT);
    const bool has_lambert = (f.kinds_present & (1u << RT_MAT_LAMBERT)) != 0;
    s->dev.fuse = s->dev.n_xforms != 0 || fuse_knob <= 0 ? 0u : lambert_only ? 1u : fuse_knob == 2 ? 2u : (fuse_knob >= 3 && has_lambert) ? 3u : 0u;
    return apply_traversal(s, s->traversal);
}

int rt_scene_commit(rt_scene* s) {
    RT_GUARD_BEGIN
    if (!s) return fail(RT_ERR_INVALID, "rt_scene_commit: NULL scene");
    if (s->committed) return fail(RT_ERR_STATE, "scene already committed");
    if (s->open_groups.size() != 1) return fail(RT_ERR_STATE, "rt_scene_commit: a container opened with rt_scene_begin_group is not closed");
    // Which trees the scene gets: the reference's (REFERENCE, AUTO) and / or the surface-area ones (SAH,
    // AUTO).  A scene without a GPU shows one tree to its inspectors and builds no second one.
    // (The two differ in how meshes of more than two triangles and containers of more than two objects
    // are split; a scene without either has one shape.)
    bool splits_differ = s->root_kind == RT_ROOT_BVH && s->groups[0].items.size() > 2;
    for (const SceneObject& o : s->objects) splits_differ = splits_differ || (o.is_mesh && o.tris.size() > 2);
    for (size_t g = 1; g < s->groups.size(); g++)
        splits_differ = splits_differ || (s->groups[g].kind == RT_ROOT_BVH && s->groups[g].items.size() > 2);
    const bool want_ref = s->build_mode != RT_BUILD_SAH;
    const bool want_fast = s->build_mode == RT_BUILD_SAH || (s->build_mode == RT_BUILD_AUTO && s->ctx != nullptr && splits_differ);
    // with a cache directory: the flattened arrays come from the file of this description's key when
    // there is one (and it is complete and sane); otherwise they are built and the file is written for next time
    s->cache_state = RT_CACHE_OFF;
    auto flatten = [&](int mode, FlatScene& out, const FlatScene* reference) -> int {
        std::string cache_path;
        uint64_t cache_key = 0;
        if (!s->cache_dir.empty()) {
            cache_key = scene_cache_key(s->mats, s->objects, s->groups, s->root_kind, mode);
            char name[40];
            std::snprintf(name, sizeof(name), "/%016llx.rtscene", (unsigned long long)cache_key);
            cache_path = s->cache_dir + name;
            if (scene_cache_load(cache_path, cache_key, out)) {
                if (s->cache_state == RT_CACHE_OFF) s->cache_state = RT_CACHE_LOADED;
                return RT_OK;
            }
        }
        std::string err = build_flat_scene(s->mats, s->objects, s->groups, s->root_kind, mode, out, reference);
        if (!err.empty()) return fail(RT_ERR_INVALID, "rt_scene_commit: " + err);
        if (!cache_path.empty()) {
            const bool stored = scene_cache_store(cache_path, cache_key, out);
            if (!stored) s->cache_state = RT_CACHE_STORE_FAILED;
            else if (s->cache_state != RT_CACHE_STORE_FAILED) s->cache_state = RT_CACHE_STORED;
        }
        return RT_OK;
    };
    FlatScene fast;
    if (want_ref) {
        int rc = flatten(RT_BUILD_REFERENCE, s->flat, nullptr);
        if (rc) return rc;
    }
    if (want_fast) {
        int rc = flatten(RT_BUILD_SAH, want_ref ? fast : s->flat, want_ref ? &s->flat : nullptr);
        if (rc) return rc;
    }
    const FlatScene& f_ref = s->flat;
    const FlatScene& f_fast = want_ref ? fast : s->flat;
    rt_context* ctx = s->ctx;
    if (want_ref) describe_tree(s->trees[TREE_REFERENCE], f_ref);
    if (want_fast) describe_tree(s->trees[TREE_FAST], f_fast);
    s->dev.n_prims = (uint32_t)s->flat.n_prims();
    bind_tree(s, s->base_tree());
    if (!ctx) {                                // host-only scene: keep the flattened arrays
        s->committed = true;
        return RT_OK;
    }
    auto upload_all = [&](rt_scene* dst) -> int {
        if (want_ref) { int rc = upload_tree(dst, TREE_REFERENCE, f_ref); if (rc) return rc; }
        if (want_fast) { int rc = upload_tree(dst, TREE_FAST, f_fast); if (rc) return rc; }
        return upload_tables(dst, s->flat);
    };
    int rc = upload_all(s);
    if (rc) return rc;
    // a multi-GPU context: the scene was flattened once; every further GPU gets the same arrays
    for (rt_context* peer : ctx->peers) {
        rt_scene* r = new rt_scene();
        s->replicas.push_back(r);
        r->ctx = peer;
        peer->refs.fetch_add(1);
        r->root_kind = s->root_kind;
        r->build_mode = s->build_mode;
        r->traversal = s->traversal;
        r->wide_build = s->wide_build;
        rc = upload_all(r);
        if (rc) {                              // leave no half-made copies behind: a later commit starts afresh
            const std::string why = g_last_error;
            for (rt_scene* made : s->replicas) rt_scene_destroy(made);
            s->replicas.clear();
            return fail(rc, why);
        }
        r->flat.prim_order = s->flat.prim_order;
        r->committed = true;
    }
    // the host copies of the geometry are no longer needed
    for (SceneObject& o : s->objects) { o.tris.clear(); o.tris.shrink_to_fit(); }
    s->flat.entries.clear(); s->flat.entries.shrink_to_fit();
    s->flat.units.clear(); s->flat.units.shrink_to_fit();
    s->flat.walk.clear(); s->flat.walk.shrink_to_fit();
    s->flat.prim_geo.clear(); s->flat.prim_geo.shrink_to_fit();
    s->committed = true;
    return RT_OK;
    RT_GUARD_END("rt_scene_commit")
}

double rt_scene_expected_cost(const rt_scene* s) { return s && s->committed ? s->trees[s->base_tree()].expected_cost : 0.0; }

size_t rt_scene_num_prims(const rt_scene* s) { return s ? s->dev.n_prims : 0; }
size_t rt_scene_num_entries(const rt_scene* s) { return s ? s->dev.n_entries : 0; }

int rt_scene_get_entries(const rt_scene* s, float* out, size_t n_floats) {
    if (!s || !out) return fail(RT_ERR_INVALID, "rt_scene_get_entries: NULL argument");
    if (!s->committed || s->ctx) return fail(RT_ERR_STATE, "rt_scene_get_entries needs a committed host-only scene");
    if (n_floats != s->flat.entries.size()) return fail(RT_ERR_INVALID, "rt_scene_get_entries: size mismatch");
    std::memcpy(out, s->flat.entries.data(), n_floats * sizeof(float));
    return RT_OK;
}

int rt_scene_get_prim_order(const rt_scene* s, uint32_t* out, size_t n) {
    if (!s || (!out && n)) return fail(RT_ERR_INVALID, "rt_scene_get_prim_order: NULL argument");
    if (!s->committed) return fail(RT_ERR_STATE, "scene not committed");
    if (n != s->flat.prim_order.size()) return fail(RT_ERR_INVALID, "rt_scene_get_prim_order: size mismatch");
    if (n) std::memcpy(out, s->flat.prim_order.data(), n * sizeof(uint32_t));
    return RT_OK;
}

size_t rt_scene_num_units(const rt_scene* s) { return s ? s->dev.n_units : 0; }

int rt_scene_get_units(const rt_scene* s, float* out, size_t n_floats) {
    if (!s || !out) return fail(RT_ERR_INVALID, "rt_scene_get_units: NULL argument");
    if (!s->committed || s->ctx) return fail(RT_ERR_STATE, "rt_scene_get_units needs a committed host-only scene");
    if (n_floats != s->flat.n_units() * 4) return fail(RT_ERR_INVALID, "rt_scene_get_units: size mismatch");
    std::memcpy(out, s->flat.units.data(), n_floats * sizeof(float));   // (without the end units)
    return RT_OK;
}

// the wide tree of a committed host-only scene, collapsed from its entries on the first call
static int host_wide(const rt_scene* cs) {
    rt_scene* s = const_cast<rt_scene*>(cs);
    if (!s->committed || s->ctx) return fail(RT_ERR_STATE, "the wide tree can be inspected on a committed host-only scene");
    if (s->tree().wide.nodes.empty()) {
        const std::string err = build_wide_scene(s->flat.entries.data(), s->flat.n_entries(), s->tree().wide,
                                                 s->flat.xforms.empty() ? nullptr : s->flat.xforms.data(), s->flat.xforms.size() / 32);
        if (!err.empty()) return fail(RT_ERR_UNSUPPORTED, "wide traversal: " + err);
    }
    return RT_OK;
}

int rt_scene_wide_info(const rt_scene* s, size_t* n_nodes, size_t* n_prims, uint32_t* depth, uint32_t* node_words) {
    RT_GUARD_BEGIN
    if (!s) return fail(RT_ERR_INVALID, "rt_scene_wide_info: NULL scene");
    int rc = host_wide(s);
    if (rc) return rc;
    if (n_nodes) *n_nodes = s->tree().wide.n_nodes();
    if (n_prims) *n_prims = s->tree().wide.n_prims();
    if (depth) *depth = s->tree().wide.depth;
    if (node_words) *node_words = s->tree().wide.node_words;
    return RT_OK;
    RT_GUARD_END("rt_scene_wide_info")
}

int rt_scene_get_wide(const rt_scene* s, uint32_t* nodes, size_t n_node_words, float* prims, size_t n_prim_floats) {
    RT_GUARD_BEGIN
    if (!s || !nodes || (!prims && n_prim_floats)) return fail(RT_ERR_INVALID, "rt_scene_get_wide: NULL argument");
    int rc = host_wide(s);
    if (rc) return rc;
    if (n_node_words != s->tree().wide.nodes.size() || n_prim_floats != s->tree().wide.prims.size())
        return fail(RT_ERR_INVALID, "rt_scene_get_wide: size mismatch");
    std::memcpy(nodes, s->tree().wide.nodes.data(), n_node_words * 4);
    if (n_prim_floats) std::memcpy(prims, s->tree().wide.prims.data(), n_prim_floats * 4);
    return RT_OK;
    RT_GUARD_END("rt_scene_get_wide")
}

size_t rt_scene_num_walk_units(const rt_scene* s) { return s ? s->dev.n_walk : 0; }

int rt_scene_get_walk_units(const rt_scene* s, float* out, size_t n_floats) {
    if (!s || !out) return fail(RT_ERR_INVALID, "rt_scene_get_walk_units: NULL argument");
    if (!s->committed || s->ctx) return fail(RT_ERR_STATE, "rt_scene_get_walk_units needs a committed host-only scene");
    if (n_floats != s->flat.n_walk() * 4) return fail(RT_ERR_INVALID, "rt_scene_get_walk_units: size mismatch");
    std::memcpy(out, s->flat.walk.data(), n_floats * sizeof(float));   // (without the end units)
    return RT_OK;
}

int rt_trace_device(rt_scene* s, const void* d_rays, size_t n, float t_min, float t_max, void* d_hits,
                    uint64_t* counts) {
    RT_GUARD_BEGIN
    if (!s || (n && (!d_rays || !d_hits))) return fail(RT_ERR_INVALID, "rt_trace_device: NULL argument");
    if (!s->committed) return fail(RT_ERR_STATE, "scene not committed");
    if (!s->ctx) return fail(RT_ERR_DEVICE, "host-only scene: no GPU context");
    rt_context* ctx = s->ctx;
    int rc = set_device(ctx);
    if (rc) return rc;
    const uint32_t grid = trace_api_grid(ctx->cfg(), n);
    if (counts) {
        HIP_TRY(ctx->totals.ensure(sizeof(Totals)));
        HIP_TRY(ctx->trace_partials.ensure((size_t)std::max(grid, 1u) * sizeof(Totals)));
    }
    if (launch_trace_api(ctx->cfg(), s->dev, d_rays, n, t_min, t_max, d_hits, (Totals*)ctx->trace_partials.p,
                         counts != nullptr, grid, ctx->totals.p))
        return fail(RT_ERR_DEVICE, "rt_trace_device: kernel launch failed");
    if (counts) {
        uint64_t t[3];
        HIP_TRY(hipMemcpyAsync(t, ctx->totals.p, sizeof(t), hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        counts[0] = t[0]; counts[1] = t[1]; counts[2] = t[2];
    } else {
        HIP_TRY(hipStreamSynchronize(ctx->stream));
    }
    return RT_OK;
    RT_GUARD_END("rt_trace_device")
}

// rt_trace_device with visit counts for callers that cannot allocate: the per-workgroup records live
// in the library's static device array and the totals go to device memory of the caller's.  Calls
// nothing that allocates and does not wait for the stream.
int rt_trace_device_static(rt_scene* s, const void* d_rays, size_t n, float t_min, float t_max, void* d_hits,
                           void* d_counts) {
    if (!s || !d_counts || (n && (!d_rays || !d_hits))) return fail(RT_ERR_INVALID, "rt_trace_device_static: NULL argument");
    if (!s->committed) return fail(RT_ERR_STATE, "scene not committed");
    if (!s->ctx) return fail(RT_ERR_DEVICE, "host-only scene: no GPU context");
    if (n > (size_t)RT_TRACE_STATIC_MAX_RAYS)
        return fail(RT_ERR_INVALID, "rt_trace_device_static: more than RT_TRACE_STATIC_MAX_RAYS (4194304) rays in one call; "
                                    "split the batch or use rt_trace_device");
    rt_context* ctx = s->ctx;
    int rc = set_device(ctx);
    if (rc) return rc;
    if (!ctx->static_scratch) return fail(RT_ERR_STATE, "rt_trace_device_static: the context has no scratch address");
    static_assert((uint64_t)TRACE_STATIC_MAX_BLOCKS * TRACE_STATIC_BLOCK_RAYS == RT_TRACE_STATIC_MAX_RAYS, "limit");
    const uint32_t grid = (uint32_t)((n + TRACE_STATIC_BLOCK_RAYS - 1) / TRACE_STATIC_BLOCK_RAYS);   // one record each
    if (launch_trace_api(ctx->cfg(), s->dev, d_rays, n, t_min, t_max, d_hits, (Totals*)ctx->static_scratch, true, grid,
                         d_counts))
        return fail(RT_ERR_DEVICE, "rt_trace_device_static: kernel launch failed");
    return RT_OK;
}

int rt_trace(rt_scene* s, const rt_ray* rays, size_t n, float t_min, float t_max, rt_hit* out) {
    RT_GUARD_BEGIN
    if (!s || (n && (!rays || !out))) return fail(RT_ERR_INVALID, "rt_trace: NULL argument");
    if (!s->committed) return fail(RT_ERR_STATE, "scene not committed");
    if (!s->ctx) return fail(RT_ERR_DEVICE, "host-only scene: no GPU context");
    if (n == 0) return RT_OK;
    rt_context* ctx = s->ctx;
    int rc = set_device(ctx);
    if (rc) return rc;
    HIP_TRY(ctx->scratch_in.ensure(n * sizeof(rt_ray)));
    HIP_TRY(ctx->scratch_out.ensure(n * sizeof(rt_hit)));
    HIP_TRY(hipMemcpyAsync(ctx->scratch_in.p, rays, n * sizeof(rt_ray), hipMemcpyHostToDevice, ctx->stream));
    rc = rt_trace_device(s, ctx->scratch_in.p, n, t_min, t_max, ctx->scratch_out.p, nullptr);
    if (rc) return rc;
    HIP_TRY(hipMemcpy(out, ctx->scratch_out.p, n * sizeof(rt_hit), hipMemcpyDeviceToHost));
    return RT_OK;
    RT_GUARD_END("rt_trace")
}

int rt_render_device(rt_scene* s, const rt_camera* cam, const rt_render_params* p, void* d_accum,
                     rt_stats* stats) {
    RT_GUARD_BEGIN
    if (!s || !cam || !p || !d_accum) return fail(RT_ERR_INVALID, "rt_render_device: NULL argument");
    if (!s->committed) return fail(RT_ERR_STATE, "scene not committed");
    if (!s->ctx) return fail(RT_ERR_DEVICE, "host-only scene: no GPU context");
    if (p->width < 2 || p->height < 2) return fail(RT_ERR_INVALID, "width and height must be at least 2");
    if (p->x0 >= p->x1 || p->y0 >= p->y1 || p->x1 > p->width || p->y1 > p->height)
        return fail(RT_ERR_INVALID, "bad pixel rectangle");
    if (p->sample_begin >= p->sample_end) return fail(RT_ERR_INVALID, "empty sample range");
    if (p->max_bounces < 1 || p->max_bounces > MAX_DEPTH)
        return fail(RT_ERR_INVALID, "max_bounces must be in [1, " + std::to_string(MAX_DEPTH) + "]");
    rt_context* ctx = s->ctx;
    int rc = set_device(ctx);
    if (rc) return rc;

    const bool profile = (p->flags & RT_RENDER_PROFILE) != 0;
    const bool count = (p->flags & RT_RENDER_COUNT_VISITS) != 0;
#ifdef RT_UTIL                                 // diagnostic build: the fused kernel counts visits itself
    const bool staged = (p->flags & RT_RENDER_STAGED) != 0 || std::getenv("RT_STAGED") != nullptr;
#else
    const bool staged = count || (p->flags & RT_RENDER_STAGED) != 0 || std::getenv("RT_STAGED") != nullptr;
#endif
    const uint32_t rw = p->x1 - p->x0, rh = p->y1 - p->y0;
    const uint32_t tiles_x = (rw + 7) / 8, tiles_y_all = (rh + 7) / 8;
    const uint32_t band_count = p->band_count > 1 ? p->band_count : 1;
    uint32_t band_tile_rows = 1, tiles_y = tiles_y_all;
    uint64_t n_pixels = (uint64_t)rw * rh;
    if (band_count > 1) {
        if (p->band_rows == 0 || p->band_rows % 8 != 0) return fail(RT_ERR_INVALID, "band_rows must be a positive multiple of 8");
        if (p->band_index >= band_count) return fail(RT_ERR_INVALID, "band_index out of range");
        band_tile_rows = p->band_rows / 8;
        tiles_y = 0;
        n_pixels = 0;
        for (uint32_t ty = 0; ty < tiles_y_all; ty++)
            if ((ty / band_tile_rows) % band_count == p->band_index) {
                tiles_y++;
                n_pixels += (uint64_t)rw * std::min(8u, rh - ty * 8u);
            }
        if (tiles_y == 0) {                    // this band group owns nothing of the rectangle
            if (stats) std::memset(stats, 0, sizeof(*stats));
            return RT_OK;
        }
    }
    const size_t n_tiles = (size_t)tiles_x * tiles_y;
    const uint32_t n_samples = p->sample_end - p->sample_begin;
    size_t max_paths = p->max_paths_in_flight ? p->max_paths_in_flight : DEFAULT_MAX_PATHS;
    if (max_paths < 64) max_paths = 64;
    // batch = (a run of whole tiles) x (a run of samples); with more than one batch, n_pipes are
    // kept in flight, so each gets its share of the path budget
    size_t tiles_per_batch = std::min(n_tiles, std::max<size_t>(1, max_paths / 64));
    uint32_t samples_per_batch =
        (uint32_t)std::min<size_t>(n_samples, std::max<size_t>(1, max_paths / (tiles_per_batch * 64)));
    size_t n_batches = ((n_tiles + tiles_per_batch - 1) / tiles_per_batch) *
                       ((n_samples + samples_per_batch - 1) / samples_per_batch);
    int n_pipes = (int)std::min<size_t>(n_batches, (size_t)ctx->n_pipes);
    // A job that fits one batch but is big enough for several chip-filling ones is cut up as well
    // when it is traced with the sweep or out of LDS: two kernels in flight hide each other's drain
    // (1024^2 x 64 spp Cornell box +24 %, world1 at 16 spp +13 %).  Not the walk from memory: its
    // refilled form lives on long segments (cornell_box.obj at 256 spp: -30 % when cut up).
    if (s->dev.sweep || (s->dev.lds_walk && !s->dev.octant && !s->dev.ordered && !s->dev.wide)) {
        const char* e = std::getenv("RT_SPLIT_MI");        // experiments: the threshold in Mi paths (0: never)
        const size_t split_mi = e ? (size_t)std::max(0, std::atoi(e)) : (size_t)RT_SPLIT_MI;
        if (!p->max_paths_in_flight && split_mi && n_tiles * 64 * (size_t)n_samples >= (split_mi << 20))
            n_pipes = ctx->n_pipes;
    }
    if (n_pipes > 1) {
        max_paths = std::max<size_t>(64, max_paths / (size_t)n_pipes);
        // A small job (one rank's share of a frame) is cut into at least 8 batches as long as a
        // batch still fills the chip (8 Mi paths): the pipelines only overlap between batches, and
        // the drain of the last batch is shorter when the batch is.
        const size_t total = n_tiles * 64 * (size_t)n_samples;
        if (!p->max_paths_in_flight)
            max_paths = std::min(max_paths, std::max<size_t>((size_t)8 << 20, total / 8));
        tiles_per_batch = std::min(n_tiles, std::max<size_t>(1, max_paths / 64));
        samples_per_batch =
            (uint32_t)std::min<size_t>(n_samples, std::max<size_t>(1, max_paths / (tiles_per_batch * 64)));
    }
    // Forms of the closest-hit loop that give the same frame bit for bit but whose speed depends on
    // the scene are timed against each other on the scene's first render, on a probe batch of the
    // frame itself (written to the pipeline's buffers only); the scene keeps the winner.
    //  - small scenes (AUTO only): wave sweep against the per-lane walk out of LDS.  How much of the
    //    scene a ray visits decides (a closed room: the sweep, by 1.5x; scattered objects under a
    //    sky: the walk, by 1.2-1.5x).
    //  - scenes walked per lane from memory: plain loop against the loop that refills idle lanes.
    //    Walks of very uneven length decide (a room with one dense mesh: refill, by 1.3-1.7x; a
    //    uniform triangle soup: plain, by 1.1-1.2x).
    struct Form { int tree; uint32_t sweep, refill, prim_wait, wide; };
    auto form_target = [&](const DevScene& d) -> uint32_t {
        if (d.sweep) return ctx->wave_target_sweep;
        if (std::getenv("RT_WAVE_TARGET") || d.ordered || staged) return ctx->wave_target;
        if (d.wide) return d.refill ? ctx->wave_target / 2 : ctx->wave_target;
        if (d.lds_walk && !d.octant) return ctx->wave_target / 4;   // 8-wave workgroups even out less finely
        return d.refill ? ctx->wave_target / 2 : ctx->wave_target;   // refill wants long segments
    };
    auto seg_len_for = [&](size_t n_slots, uint32_t target) -> uint32_t {
        return std::max<uint32_t>(64u, (uint32_t)(((n_slots + target - 1) / target + 63u) / 64u * 64u));
    };
    auto set_form = [&](const Form& f) {
        bind_tree(s, f.tree);
        s->dev.sweep = f.sweep; s->dev.refill = f.refill; s->dev.prim_wait = f.prim_wait; s->dev.wide = f.wide;
        s->dev.tie = (f.tree != TREE_REFERENCE || f.wide || s->dev.octant || s->dev.ordered) ? 1u : 0u;
    };
    if (!s->auto_tuned && !staged && !s->dev.ordered && !s->dev.wide && !std::getenv("RT_NO_AUTOTUNE")) {
        std::vector<Form> forms;
        const uint32_t wait_mem = s->env_prim_wait ? s->env_prim_wait : 8u, wait_refill = s->env_prim_wait ? s->env_prim_wait : 16u;
        const bool pinned_refill = std::getenv("RT_REFILL") != nullptr;
        // RT_AUTO_PROBE_ALL=1: time every form on every tree (what the rules below are derived from)
        const bool probe_all = std::getenv("RT_AUTO_PROBE_ALL") != nullptr;
        if (s->traversal == RT_TRAVERSAL_AUTO) {
            // The forms that report the reference's hits, on the trees the scene has (the base tree first: it
            // keeps a draw).  A scene with two trees runs on the one whose expected cost per ray (the
            // surface-area measure over its boxes) is lower: the surface-area tree on every benchmark workload,
            // where it beat the median-split tree in every form by 1.07-2x; the median-split tree where
            // triangles are as large as their mesh and overlap throughout (per-lane walk 0.23 against 0.19 Grays/s).
            const int base = s->base_tree();
            const bool two = s->build_mode == RT_BUILD_AUTO && s->trees[TREE_FAST].present && s->trees[TREE_REFERENCE].present;
            const double c_ref = s->trees[TREE_REFERENCE].expected_cost, c_fast = s->trees[TREE_FAST].expected_cost;
            // (a scene that will take the wide walk collapses the surface-area tree whatever the measure says: the
            // collapse makes its own nodes, and over the surface-area tree they were the better ones on every scene
            // measured -- 10 M triangles 1.34 against 1.26 Grays/s, cornell_box.obj 7.2 against 5.5, random
            // transformed meshes 0.59 against 0.51 where the measure preferred the median-split tree)
            const bool wide_tree = two && !s->trees[TREE_FAST].lds_walk && !std::getenv("RT_AUTO_NO_WIDE");
            const int pick = !two ? base : (!wide_tree && c_ref > 0.0 && c_fast > 0.0 && c_ref < c_fast && !std::getenv("RT_AUTO_FAST_TREE")) ? TREE_REFERENCE : TREE_FAST;
            const bool fast = two;
            if (std::getenv("RT_AUTOTUNE_LOG") && two)
                std::fprintf(stderr, "[rt auto] expected cost per ray: reference tree %.2f, surface-area tree %.2f -> the %s tree\n", c_ref, c_fast,
                             pick == TREE_FAST ? "surface-area" : "reference");
            for (int which : {base, 1 - base}) {
                const SceneTree& t = s->trees[which];
                if (!t.present) continue;
                if (two && which != pick && !probe_all) continue;
                if (t.n_entries <= SWEEP_MAX_ENTRIES && s->dev.n_xforms == 0) forms.push_back(Form{which, 1u, 0u, s->dev.prim_wait, 0u});
                if (t.lds_walk) forms.push_back(Form{which, 0u, pinned_refill ? s->env_refill : 0u, s->dev.prim_wait, 0u});
                else if (pinned_refill) forms.push_back(Form{which, 0u, s->env_refill, s->dev.prim_wait, 0u});
                else { forms.push_back(Form{which, 0u, 0u, wait_mem, 0u}); forms.push_back(Form{which, 0u, 16u, wait_refill, 0u}); }
            }
            // The wide walk over the surface-area trees, for scenes that are walked from memory.  Where it
            // applies it is taken without a timing: on frames of real size it beat both binary walks on
            // every such scene measured (cornell_box.obj 1.15-1.4x, 10 M triangles 3.6x), and a timing of
            // three whole batches costs a one-frame job several times its frame.
            if (fast && !s->trees[pick].lds_walk && !std::getenv("RT_AUTO_NO_WIDE")) {
                const int was = s->cur;
                bind_tree(s, pick);
                const int wrc = ensure_wide_tree(s);
                bind_tree(s, was);
                if (wrc == RT_OK) {
                    if (!probe_all) forms.clear();
                    forms.push_back(Form{pick, 0u, pinned_refill ? s->env_refill : 16u, s->dev.prim_wait, 1u});
                } else if (wrc != RT_ERR_UNSUPPORTED) {
                    return wrc;
                }
            }
        } else if (!s->dev.sweep && (!s->dev.lds_walk || s->dev.octant) && !pinned_refill) {
            forms = {Form{s->cur, 0u, 0u, wait_mem, 0u}, Form{s->cur, 0u, 16u, wait_refill, 0u}};
        }
        // A timing needs a batch of the size real frames have: below 8 Mi paths the segments are too short
        // for the refilled forms and too few for the sweep, and what wins there loses on a real frame (a
        // 192 x 160 x 8 spp frame of a 12 k-triangle scene: plain walk 5.8, wide 7.6, refilled 9.4 ms per Mi
        // paths; the reverse order from 16 Mi paths on).  Such a frame runs on the form that wins most
        // often and the scene stays untimed until a larger frame comes.
        bool timed = true;
        if (forms.size() > 1 && tiles_per_batch * 64 * (size_t)samples_per_batch < ((size_t)8 << 20) && !std::getenv("RT_PROBE_ANY_SIZE")) {
            size_t prior = 0;
            for (size_t f = 0; f < forms.size(); f++) {
                const bool lds = s->trees[forms[f].tree].lds_walk && !forms[f].sweep;
                if (forms[f].wide || (lds && !forms[prior].wide) || (!forms[f].sweep && !forms[f].refill && forms[prior].sweep)) prior = f;
            }
            set_form(forms[prior]);
            forms.clear();
            timed = false;
        }
        if (forms.size() == 1) { set_form(forms[0]); forms.clear(); }
        size_t best = 0;
        double best_ms = 0.0;              // per Mi paths
        // an error return from the probe leaves the scene on the form it had
        struct Restore {
            rt_scene* s;
            Form was;
            uint32_t tie;
            bool keep = false;
            ~Restore() {
                if (keep) return;
                bind_tree(s, was.tree);
                s->dev.sweep = was.sweep; s->dev.refill = was.refill; s->dev.prim_wait = was.prim_wait; s->dev.wide = was.wide;
                s->dev.tie = tie;
            }
        } restore{s, Form{s->cur, s->dev.sweep, s->dev.refill, s->dev.prim_wait, s->dev.wide}, s->dev.tie};
        const bool big_scene = !s->trees[s->base_tree()].lds_walk;
        const int probe_passes = std::getenv("RT_PROBE_PASSES") ? std::max(1, std::atoi(std::getenv("RT_PROBE_PASSES"))) : 1;
        for (int pass = 0; pass < probe_passes; pass++)
        for (size_t f = 0; f < forms.size(); f++) {
            set_form(forms[f]);
            BatchGeom g;
            g.width = p->width; g.height = p->height;
            g.x0 = p->x0; g.y0 = p->y0; g.x1 = p->x1; g.y1 = p->y1;
            g.tiles_x = tiles_x;
            g.tile_base = 0;
            // The probe is the frame's own first batch, cut into the segments the form really gets, on
            // each pipeline the frame will use (the second pipeline takes the next samples of the same
            // tiles, as in the frame).  Anything smaller misleads: the sweep fills its groups of 128
            // rays worse from short segments; the refilled forms live on long segments; and a launch
            // with fewer waves than the frame's has another balance between its steady state and its
            // drain (cornell_box.obj at 256 spp, one batch of 64 Mi paths in 32 Ki segments: wide walk
            // 36 ms, refilled binary walk 51 ms; half the segments: 27 ms against 28 ms).  When a batch
            // holds only a part of the frame's tiles (beyond 64 Mi pixels) it is probed on runs of
            // consecutive tiles spread evenly over the rectangle: the first tiles of a frame are
            // often sky.
            const size_t batch_paths = tiles_per_batch * 64 * (size_t)samples_per_batch;
            const size_t nt = tiles_per_batch;
            const int probe_pipes = std::max(1, std::min(n_pipes, 2));
            uint32_t pipe_tile_base[2] = {0, 0}, pipe_sample_base[2] = {p->sample_begin, p->sample_begin};
            if (n_samples >= 2 * samples_per_batch) pipe_sample_base[1] += samples_per_batch;
            g.n_pix_slots = (uint32_t)(nt * 64);
            g.n_samples = samples_per_batch;
            g.n_slots = g.n_pix_slots * g.n_samples;
            g.band_tile_rows = band_tile_rows; g.band_count = band_count; g.band_index = p->band_index;
            g.seg_len = seg_len_for(batch_paths, form_target(s->dev));
            g.n_seg = (g.n_slots + g.seg_len - 1) / g.seg_len;
            if (nt < n_tiles) {
                uint32_t shift = 6;                                     // runs of 64 tiles, or a segment's tiles if that is more
                while (((size_t)64 << shift) < g.seg_len) shift++;
                const size_t run = (size_t)1 << shift, n_runs = nt / run;
                if (n_runs >= 1 && nt % run == 0 && n_tiles / run >= 2 * n_runs) {
                    size_t stride = (n_tiles / run) / n_runs;
                    // (runs that all start in the same tile column would cover a stripe of the frame)
                    if (run % tiles_x != 0 && (stride * run) % tiles_x == 0) stride--;
                    if (stride > 1) {
                        g.tile_run_shift = shift;
                        g.tile_gap = (uint32_t)((stride - 1) * run);
                    }
                }
            }
            RenderBuffers rbs[2];
            for (int k = 0; k < probe_pipes; k++) {
                Pipe& pk = ctx->pipes[k];
                rc = ensure_capacity(ctx, pk, g.n_slots, g.n_seg);
                if (rc) return rc;
                RenderBuffers& rb = rbs[k];
                rb.hits = pk.hits.p;
                for (int q = 0; q < 3; q++) rb.queue[q] = (uint32_t*)pk.queue[q].p;
                rb.Lbuf = pk.lbuf.p;
                rb.totals = (Totals*)ctx->totals.p;
                rb.capacity = pk.capacity;
                rb.segcnt = (uint32_t*)pk.segcnt.p;
                rb.qcnt = (uint32_t*)pk.qcnt.p;
                rb.cur = RayBuffers{pk.rayA[0].p, pk.rayB[0].p, pk.rayC[0].p};
                rb.next = RayBuffers{pk.rayA[1].p, pk.rayB[1].p, pk.rayC[1].p};
            }
            Pipe& pp = ctx->pipes[0];
            for (int rep = 0; rep < 2; rep++) {            // the second run is the timed one
                BatchGeom gr = g;
                if (rep == 0 && big_scene) {               // (a big scene's first run only warms code, TLBs and the top of the tree)
                    gr.n_seg = std::max<uint32_t>(1, g.n_seg / 8);
                    gr.n_slots = std::min<uint32_t>(g.n_slots, gr.n_seg * g.seg_len);
                }
                HIP_TRY(hipEventRecord(ctx->ev_begin, pp.stream));
                for (int k = 1; k < probe_pipes; k++) {
                    HIP_TRY(hipEventRecord(ctx->ev_fork, pp.stream));
                    HIP_TRY(hipStreamWaitEvent(ctx->pipes[k].stream, ctx->ev_fork, 0));
                }
                for (int k = 0; k < probe_pipes; k++) {
                    gr.tile_base = pipe_tile_base[k];
                    gr.sample_base = pipe_sample_base[k];
                    if (launch_path(ctx->cfg(k), s->dev, *cam, gr, p->seed, rbs[k], p->max_bounces, p->t_min, p->sky_intensity))
                        return fail(RT_ERR_DEVICE, "path kernel launch failed");
                }
                for (int k = 1; k < probe_pipes; k++) {
                    HIP_TRY(hipEventRecord(ctx->pipes[k].joined, ctx->pipes[k].stream));
                    HIP_TRY(hipStreamWaitEvent(pp.stream, ctx->pipes[k].joined, 0));
                }
                HIP_TRY(hipEventRecord(ctx->ev_end, pp.stream));
            }
            HIP_TRY(hipStreamSynchronize(pp.stream));
            float ms_run = 0.0f;
            HIP_TRY(hipEventElapsedTime(&ms_run, ctx->ev_begin, ctx->ev_end));
            const double ms = (double)ms_run * 1048576.0 / ((double)g.n_slots * probe_pipes);
            if (std::getenv("RT_AUTOTUNE_LOG"))       // the probe's timings on stderr
                std::fprintf(stderr, "[rt probe] tree %d sweep %u wide %u refill %u prim_wait %u: %.3f ms per Mi paths (%.3f ms, %u slots in %u segments, runs of %u tiles %u apart, %d pipelines)\n",
                             forms[f].tree, forms[f].sweep, forms[f].wide, forms[f].refill, forms[f].prim_wait, ms, ms_run, g.n_slots, g.n_seg,
                             1u << g.tile_run_shift, g.tile_gap + (1u << g.tile_run_shift), probe_pipes);
            // The probe is one batch per pipeline, started together and drained to the end, and that flatters
            // the walk: in the frame itself batch follows batch and the sweep's long-lived waves (one per wave
            // slot) lose less to the drain than here.  Measured on the plane Cornell box, a 1/8 share of the
            // frame: probe 11.4 ms sweep against 10.8-11.4 ms walk, frame 37.6 ms against 40.8 ms.  A form that
            // is not a sweep therefore has to beat a sweep by a tenth to be taken (where the walks really win
            // -- objects scattered under a sky, the surface-area trees -- they win by 1.2-1.5x).
            const double margin = (f > 0 && forms[best].sweep && !forms[f].sweep) ? 0.9 : 1.0;
            if (f == 0 || ms < best_ms * margin) { best = f; best_ms = ms; }
        }
        if (!forms.empty()) set_form(forms[best]);
        restore.keep = true;
        s->auto_tuned = timed;
    }
    const uint32_t wave_target = form_target(s->dev);
    auto segments = [&](BatchGeom& g) {        // one wave per segment, a multiple of 64 slots long
        g.seg_len = seg_len_for(g.n_slots, wave_target);
        g.n_seg = (g.n_slots + g.seg_len - 1) / g.seg_len;
    };
    {
        BatchGeom g0;
        g0.n_slots = (uint32_t)(tiles_per_batch * 64 * samples_per_batch);
        segments(g0);
        // a shorter batch (a tapered chunk, the last tile group) is cut into shorter segments and may
        // have more of them than the longest batch: never more than the wave target or one per 64 slots
        const size_t seg_cap = std::max<size_t>(g0.n_seg, std::min<size_t>(wave_target, (g0.n_slots + 63u) / 64u));
        for (int k = 0; k < n_pipes; k++) {
            rc = ensure_capacity(ctx, ctx->pipes[k], g0.n_slots, seg_cap);
            if (rc) return rc;
        }
    }

    hipStream_t st0 = ctx->pipes[0].stream;
    const hipEvent_t ev_begin = ctx->ev_begin, ev_end = ctx->ev_end;
    struct Span { size_t e0, e1; int what; };
    std::vector<Span> spans[MAX_PIPES];
    size_t ev_used[MAX_PIPES] = {};
    double ms_by_kind[2] = {0.0, 0.0};
    // the closest-hit kernels of the pipelines overlap: besides their summed durations, the span from
    // the first start to the last end (both measured from the call's begin event)
    double first_start_ms = 1e300, last_end_ms = 0.0;
    auto fold_spans = [&]() -> int {           // needs every stream to be idle
        for (int k = 0; k < MAX_PIPES; k++) {
            for (const Span& sp : spans[k]) {
                float m = 0.0f;
                HIP_TRY(hipEventElapsedTime(&m, ctx->pipes[k].events[sp.e0], ctx->pipes[k].events[sp.e1]));
                ms_by_kind[sp.what] += m;
                if (sp.what == 0) {
                    float a = 0.0f, b = 0.0f;
                    HIP_TRY(hipEventElapsedTime(&a, ev_begin, ctx->pipes[k].events[sp.e0]));
                    HIP_TRY(hipEventElapsedTime(&b, ev_begin, ctx->pipes[k].events[sp.e1]));
                    first_start_ms = std::min(first_start_ms, (double)a);
                    last_end_ms = std::max(last_end_ms, (double)b);
                }
            }
            spans[k].clear();
            ev_used[k] = 0;
        }
        return RT_OK;
    };
    constexpr size_t CLOCK_SLOTS = 4096;       // launches whose clock stamps are kept (a ring)
    if (profile && !staged) {
        HIP_TRY(ctx->clock_stamps.ensure(CLOCK_SLOTS * 4 * sizeof(unsigned long long)));
        HIP_TRY(hipMemsetAsync(ctx->clock_stamps.p, 0, CLOCK_SLOTS * 4 * sizeof(unsigned long long), st0));
    }

    HIP_TRY(hipMemsetAsync(ctx->totals.p, 0, sizeof(Totals), st0));
    HIP_TRY(hipEventRecord(ev_begin, st0));
    if (n_pipes > 1) {                         // the other streams start after whatever precedes the call
        HIP_TRY(hipEventRecord(ctx->ev_fork, st0));
        for (int k = 1; k < n_pipes; k++) HIP_TRY(hipStreamWaitEvent(ctx->pipes[k].stream, ctx->ev_fork, 0));
    }

    uint64_t trace_launches = 0;
    size_t batch = 0;
    int last_pipe = -1;
    for (size_t tile0 = 0; tile0 < n_tiles; tile0 += tiles_per_batch) {
        const size_t nt = std::min(tiles_per_batch, n_tiles - tile0);
        // sample chunks of this tile group; the very last ones of the call taper off (1/2, 1/2, 1/4,
        // 1/4, 1/8 x 4 of a chunk, not below 1 Mi paths): the call ends when the slowest wave of its
        // last batch does, and that wave's life is proportional to the batch
        std::vector<uint32_t> chunks;
        for (uint32_t s0 = 0; s0 < n_samples; s0 += samples_per_batch)
            chunks.push_back(std::min(samples_per_batch, n_samples - s0));
        if (n_pipes > 1 && tile0 + tiles_per_batch >= n_tiles && chunks.size() >= 4 && !std::getenv("RT_NO_TAPER")) {
            const uint32_t min_chunk = (uint32_t)std::max<size_t>(1, ((size_t)1 << 20) / (nt * 64));
            uint32_t rest = chunks[chunks.size() - 1] + chunks[chunks.size() - 2];
            chunks.resize(chunks.size() - 2);
            const uint32_t full = samples_per_batch;
            for (uint32_t div : {2u, 2u, 4u, 4u, 8u, 8u, 8u, 8u}) {
                // never above a whole chunk: the pipelines' buffers are sized for that
                const uint32_t c = std::min(rest, std::min(full, std::max(min_chunk, full / div)));
                if (c == 0) break;
                chunks.push_back(c);
                rest -= c;
            }
            for (; rest; rest -= std::min(rest, full)) chunks.push_back(std::min(rest, full));
        }
        uint32_t s0 = 0;
        for (size_t ci = 0; ci < chunks.size(); s0 += chunks[ci], ci++, batch++) {
            const uint32_t chunk = chunks[ci];
            const int k = (int)(batch % (size_t)n_pipes);
            Pipe& pp = ctx->pipes[k];
            hipStream_t st = pp.stream;
            const LaunchCfg cfg = ctx->cfg(k);
            BatchGeom g;
            g.width = p->width; g.height = p->height;
            g.x0 = p->x0; g.y0 = p->y0; g.x1 = p->x1; g.y1 = p->y1;
            g.tiles_x = tiles_x;
            g.tile_base = (uint32_t)tile0;
            g.n_pix_slots = (uint32_t)(nt * 64);
            g.sample_base = p->sample_begin + s0;
            g.n_samples = chunk;
            g.n_slots = g.n_pix_slots * g.n_samples;
            g.band_tile_rows = band_tile_rows; g.band_count = band_count; g.band_index = p->band_index;
            segments(g);
            if (g.n_slots > pp.capacity || (size_t)(MAX_DEPTH + 1) * g.n_seg * sizeof(uint32_t) > pp.segcnt.bytes ||
                (size_t)3 * g.n_seg * sizeof(uint32_t) > pp.qcnt.bytes)
                return fail(RT_ERR_STATE, "internal error: a batch is larger than the pipeline's buffers");

            RenderBuffers rb;
            rb.hits = pp.hits.p;
            for (int q = 0; q < 3; q++) rb.queue[q] = (uint32_t*)pp.queue[q].p;
            rb.Lbuf = pp.lbuf.p;
            rb.totals = (Totals*)ctx->totals.p;
            rb.capacity = pp.capacity;
            rb.segcnt = (uint32_t*)pp.segcnt.p;
            rb.qcnt = (uint32_t*)pp.qcnt.p;
            int cur = 0;
            auto bind = [&]() {
                rb.cur = RayBuffers{pp.rayA[cur].p, pp.rayB[cur].p, pp.rayC[cur].p};
                rb.next = RayBuffers{pp.rayA[cur ^ 1].p, pp.rayB[cur ^ 1].p, pp.rayC[cur ^ 1].p};
            };
            auto mark = [&](int what, auto&& launch) -> int {
                if (!profile) return launch();
                int r = get_events(pp, ev_used[k] + 2);
                if (r) return r;
                if (hipEventRecord(pp.events[ev_used[k]], st) != hipSuccess) return fail(RT_ERR_DEVICE, "hipEventRecord");
                int lr = launch();
                if (hipEventRecord(pp.events[ev_used[k] + 1], st) != hipSuccess) return fail(RT_ERR_DEVICE, "hipEventRecord");
                spans[k].push_back(Span{ev_used[k], ev_used[k] + 1, what});
                ev_used[k] += 2;
                return lr;
            };
            bind();
            if (profile && !staged)
                rb.clock_stamps = (unsigned long long*)ctx->clock_stamps.p + (trace_launches % CLOCK_SLOTS) * 4;
            if (!staged) {
                int lr = mark(0, [&]() {
                    return launch_path(cfg, s->dev, *cam, g, p->seed, rb, p->max_bounces, p->t_min, p->sky_intensity);
                });
                if (lr) return fail(RT_ERR_DEVICE, "path kernel launch failed");
                trace_launches++;
            } else {
                if (launch_raygen(cfg, *cam, g, p->seed, rb)) return fail(RT_ERR_DEVICE, "raygen launch failed");
                for (uint32_t d = 0; d < p->max_bounces; d++) {
                    const bool last = d + 1 == p->max_bounces;
                    int lr = mark(0, [&]() {
                        return launch_trace(cfg, s->dev, g, rb, d, last, p->t_min, p->sky_intensity, count);
                    });
                    if (lr) return fail(RT_ERR_DEVICE, "trace launch failed");
                    trace_launches++;
                    if (last) break;
                    lr = mark(1, [&]() { return launch_shade(cfg, s->dev, g, rb, d); });
                    if (lr) return fail(RT_ERR_DEVICE, "shade launch failed");
                    cur ^= 1;
                    bind();
                }
            }
            // batches add to the same pixels: accumulate in batch order, across the streams
            if (last_pipe >= 0 && last_pipe != k)
                HIP_TRY(hipStreamWaitEvent(st, ctx->pipes[last_pipe].accum_done, 0));
            if (launch_accumulate(cfg, g, rb, d_accum, p->max_bounces))
                return fail(RT_ERR_DEVICE, "accumulate launch failed");
            HIP_TRY(hipEventRecord(pp.accum_done, st));
            last_pipe = k;
            if (profile && ev_used[k] > 4096) {
                // keep the event pools bounded on long renders: fold what has finished so far
                for (int q = 0; q < n_pipes; q++) HIP_TRY(hipStreamSynchronize(ctx->pipes[q].stream));
                rc = fold_spans();
                if (rc) return rc;
            }
        }
    }
    for (int k = 1; k < n_pipes; k++) {        // join: everything is complete when stream 0 is
        HIP_TRY(hipEventRecord(ctx->pipes[k].joined, ctx->pipes[k].stream));
        HIP_TRY(hipStreamWaitEvent(st0, ctx->pipes[k].joined, 0));
    }
    HIP_TRY(hipEventRecord(ev_end, st0));

    if (stats) {
        std::memset(stats, 0, sizeof(*stats));
        Totals t;
        HIP_TRY(hipMemcpyAsync(&t, ctx->totals.p, sizeof(t), hipMemcpyDeviceToHost, st0));
        HIP_TRY(hipStreamSynchronize(st0));
        float ms = 0.0f;
        HIP_TRY(hipEventElapsedTime(&ms, ev_begin, ev_end));
        stats->ms_render = ms;
        stats->rays_traced = t.rays;
        stats->paths = n_pixels * n_samples;
        stats->box_tests = t.box; stats->tri_tests = t.tri; stats->sph_tests = t.sph;
        stats->trace_launches = trace_launches;
        rc = fold_spans();
        if (rc) return rc;
        stats->ms_trace = ms_by_kind[0];
        stats->ms_shade = ms_by_kind[1];
        stats->ms_kernel_span = last_end_ms > first_start_ms ? last_end_ms - first_start_ms : 0.0;
        stats->ms_render_min = ms;
        if (profile && !staged && trace_launches) {    // the shader clock the per-batch kernels ran at
            const size_t n = (size_t)std::min<uint64_t>(trace_launches, CLOCK_SLOTS);
            std::vector<unsigned long long> w(n * 4);
            HIP_TRY(hipMemcpy(w.data(), ctx->clock_stamps.p, w.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
            double cyc = 0.0, ref = 0.0;
            for (size_t i = 0; i < n; i++)
                if (w[i * 4 + 3] > w[i * 4 + 1] && w[i * 4 + 2] > w[i * 4]) {
                    cyc += (double)(w[i * 4 + 2] - w[i * 4]);
                    ref += (double)(w[i * 4 + 3] - w[i * 4 + 1]);
                }
            if (ref > 0.0) stats->shader_clock_ghz = cyc / ref * 0.1;      // the reference counter runs at 100 MHz
        }
        stats->n_devices = 1;
        stats->tree = s->cur == TREE_REFERENCE ? RT_TREE_REFERENCE : RT_TREE_SAH;
        const DevScene& d = s->dev;            // the form the launches above took
        stats->kernel_form = d.wide ? (d.refill && !staged ? RT_FORM_WIDE_REFILL : RT_FORM_WIDE)
                             : d.ordered ? RT_FORM_ORDERED
                             : d.octant ? (d.refill && !staged ? RT_FORM_OCTANT_REFILL : RT_FORM_OCTANT)
                             : d.sweep ? RT_FORM_SWEEP
                             : staged ? RT_FORM_WALK
                             : d.lds_walk ? (d.refill ? RT_FORM_WALK_LDS_REFILL : RT_FORM_WALK_LDS)
                                          : (d.refill ? RT_FORM_WALK_REFILL : RT_FORM_WALK);
    } else {
        HIP_TRY(hipStreamSynchronize(st0));    // the call returns with the frame complete either way
    }
#ifdef RT_STAMPS
    { static bool dumped = false; if (!dumped && !count) { dumped = true; rtamd::dump_stamps(); } }
#endif
    return RT_OK;
    RT_GUARD_END("rt_render_device")
}

int rt_render(rt_scene* s, const rt_camera* cam, const rt_render_params* p, float* accum_rgb, rt_stats* stats) {
    RT_GUARD_BEGIN
    if (!s || !p || !accum_rgb) return fail(RT_ERR_INVALID, "rt_render: NULL argument");
    if (!s->ctx) return fail(RT_ERR_DEVICE, "host-only scene: no GPU context");
    rt_context* ctx = s->ctx;
    int rc = set_device(ctx);
    if (rc) return rc;
    const size_t bytes = (size_t)p->width * p->height * 3 * sizeof(float);
    DeviceBuffer& d = ctx->frame_in;           // kept by the context: progressive renders call this per step
    HIP_TRY(d.ensure(bytes));
    HIP_TRY(hipMemcpyAsync(d.p, accum_rgb, bytes, hipMemcpyHostToDevice, ctx->stream));
    rt_stats local;
    rc = rt_render_device(s, cam, p, d.p, stats ? stats : &local);
    if (rc) return rc;
    HIP_TRY(hipMemcpy(accum_rgb, d.p, bytes, hipMemcpyDeviceToHost));
    return RT_OK;
    RT_GUARD_END("rt_render")
}

int rt_partition(uint32_t n_ranks, uint32_t rank, rt_render_params* p) {
    if (!p) return fail(RT_ERR_INVALID, "rt_partition: NULL params");
    if (n_ranks < 1 || rank >= n_ranks) return fail(RT_ERR_INVALID, "rt_partition: rank out of range");
    if (p->band_count > 1) return fail(RT_ERR_INVALID, "rt_partition: params already describe a band group");
    if (p->sample_begin > p->sample_end) return fail(RT_ERR_INVALID, "rt_partition: bad sample range");
    const uint32_t spp = p->sample_end - p->sample_begin;
    uint32_t groups = (n_ranks >= 4 && n_ranks % 2 == 0) ? 2u : 1u;
    uint32_t parts = n_ranks / groups;
    if (parts > spp) { groups = n_ranks; parts = 1; }      // fewer samples than parts: split by bands only
    const uint32_t group = rank % groups, part = rank / groups;
    const uint32_t base = spp / parts, extra = spp % parts;
    const uint32_t begin = part * base + std::min(part, extra);
    const uint32_t count = base + (part < extra ? 1u : 0u);
    p->sample_begin += begin;
    p->sample_end = p->sample_begin + count;
    p->band_rows = 64;                         // BLOCK_SIZE, main.rs:22
    p->band_count = groups;
    p->band_index = group;
    return RT_OK;
}

int rt_render_multi_device(rt_scene* s, const rt_camera* cam, const rt_render_params* p, void* d_accum,
                           rt_stats* stats) {
    RT_GUARD_BEGIN
    if (!s || !cam || !p || !d_accum) return fail(RT_ERR_INVALID, "rt_render_multi: NULL argument");
    if (!s->committed) return fail(RT_ERR_STATE, "scene not committed");
    if (!s->ctx) return fail(RT_ERR_DEVICE, "host-only scene: no GPU context");
    rt_context* root = s->ctx;
    if (root->destroyed) return fail(RT_ERR_STATE, "the context of this scene has been destroyed");
    if (!root->multi) return fail(RT_ERR_STATE, "rt_render_multi needs a context made by rt_context_create_multi");
    if (p->band_count > 1) return fail(RT_ERR_INVALID, "rt_render_multi splits the frame itself: band_count must be unset");
    const int n = 1 + (int)root->peers.size();
    if ((int)s->replicas.size() != n - 1) return fail(RT_ERR_STATE, "rt_render_multi: the scene was not committed on this context");
    const size_t n_floats = (size_t)p->width * p->height * 3;
    const size_t frame_bytes = n_floats * sizeof(float);
    auto ctx_of = [&](int r) { return r == 0 ? root : root->peers[(size_t)r - 1]; };

    // Every GPU renders its share into a zeroed frame of its own -- the root too: the caller's buffer
    // is touched only after every GPU has succeeded, so a call that fails can be repeated -- on a host
    // thread of its own (the peers' threads belong to the context).
    struct Worker { int rc = RT_OK; std::string err; rt_stats st; };
    std::vector<Worker> work((size_t)n);
    auto job = [&](int r) {
        Worker& w = work[(size_t)r];
        std::memset(&w.st, 0, sizeof(w.st));
        try {
            rt_scene* sc = r == 0 ? s : s->replicas[(size_t)r - 1];
            rt_context* ctx = sc->ctx;
            rt_render_params pr = *p;
            w.rc = rt_partition((uint32_t)n, (uint32_t)r, &pr);
            if (w.rc == RT_OK) {
                hipError_t e = hipSetDevice(ctx->device);
                if (e == hipSuccess) e = ctx->frame.ensure(frame_bytes);
                if (e == hipSuccess) e = hipMemsetAsync(ctx->frame.p, 0, frame_bytes, ctx->stream);
                if (e != hipSuccess) w.rc = fail(RT_ERR_DEVICE, std::string("rt_render_multi: frame buffer: ") + hipGetErrorString(e));
            }
            if (w.rc == RT_OK && pr.sample_begin < pr.sample_end) w.rc = rt_render_device(sc, cam, &pr, ctx->frame.p, &w.st);
            else if (w.rc == RT_OK) w.rc = hipStreamSynchronize(ctx->stream) == hipSuccess ? RT_OK : fail(RT_ERR_DEVICE, "rt_render_multi: synchronize");
        } catch (const std::bad_alloc&) {
            w.rc = fail(RT_ERR_NOMEM, "rt_render_multi: out of host memory");
        } catch (...) {
            w.rc = fail(RT_ERR_INVALID, "rt_render_multi: exception in a GPU's host thread");
        }
        if (w.rc != RT_OK) w.err = g_last_error;   // the message lives in this thread
    };
    for (int r = 1; r < n; r++) root->workers[(size_t)r - 1]->submit([&job, r]() { job(r); });
    job(0);
    for (int r = 1; r < n; r++) root->workers[(size_t)r - 1]->wait();
    for (const Worker& w : work)
        if (w.rc != RT_OK) return fail(w.rc, w.err);

    // combine the frames on the root (every render above has drained its stream), then add the sum
    // to the caller's buffer
    const auto t0 = std::chrono::steady_clock::now();
    uint32_t rccl_ranks = 0;
    int reduce_mode = root->reduce_mode;
    HIP_TRY(hipSetDevice(root->device));
    if (n > 1 && !root->comms.empty() && reduce_mode == RT_REDUCE_RCCL) {
        // the one collective: ncclReduce(sum, f32) to the root, in place there
        ncclResult_t nr = ncclGroupStart();
        for (int r = 0; r < n && nr == ncclSuccess; r++) {
            rt_context* ctx = ctx_of(r);
            nr = ncclReduce(ctx->frame.p, root->frame.p, n_floats, ncclFloat, ncclSum, 0, root->comms[(size_t)r], ctx->stream);
        }
        const ncclResult_t ne = ncclGroupEnd();
        if (nr == ncclSuccess) nr = ne;
        if (nr != ncclSuccess) return fail(RT_ERR_DEVICE, std::string("rt_render_multi: ncclReduce: ") + ncclGetErrorString(nr));
        for (int r = n - 1; r >= 0; r--) {
            rt_context* ctx = ctx_of(r);
            HIP_TRY(hipSetDevice(ctx->device));
            HIP_TRY(hipStreamSynchronize(ctx->stream));
        }
        int count = 0;
        if (ncclCommCount(root->comms[0], &count) == ncclSuccess) rccl_ranks = (uint32_t)count;
    } else if (n == 1 && !root->comms.empty() && reduce_mode == RT_REDUCE_RCCL) {
        // one GPU: the same call, a reduce over one rank (in place)
        const ncclResult_t nr = ncclReduce(root->frame.p, root->frame.p, n_floats, ncclFloat, ncclSum, 0, root->comms[0], root->stream);
        if (nr != ncclSuccess) return fail(RT_ERR_DEVICE, std::string("rt_render_multi: ncclReduce: ") + ncclGetErrorString(nr));
        int count = 0;
        if (ncclCommCount(root->comms[0], &count) == ncclSuccess) rccl_ranks = (uint32_t)count;
    } else if (n > 1 && reduce_mode == RT_REDUCE_ORDERED) {
        // strict order: every peer's frame is copied to the root and added there, rank by rank, on the
        // root's stream (two staging halves, so the copy of rank r + 1 runs under the add of rank r)
        HIP_TRY(root->stage.ensure(2 * frame_bytes));
        hipStream_t copy_stream = root->pipes[1].stream;
        for (int r = 1; r < n; r++) {
            rt_context* peer = ctx_of(r);
            char* half = (char*)root->stage.p + (size_t)(r & 1) * frame_bytes;
            // the half is free once the add of rank r - 2 has run
            if (r >= 3) HIP_TRY(hipStreamWaitEvent(copy_stream, root->pipes[(r & 1)].accum_done, 0));
            HIP_TRY(hipMemcpyPeerAsync(half, root->device, peer->frame.p, peer->device, frame_bytes, copy_stream));
            HIP_TRY(hipEventRecord(root->pipes[2 + (r & 1)].joined, copy_stream));
            HIP_TRY(hipStreamWaitEvent(root->stream, root->pipes[2 + (r & 1)].joined, 0));
            if (launch_add_frame(root->cfg(), root->frame.p, half, n_floats))
                return fail(RT_ERR_DEVICE, "rt_render_multi: add launch failed");
            HIP_TRY(hipEventRecord(root->pipes[(r & 1)].accum_done, root->stream));
        }
    } else if (n > 1) {
        // the workers share one GPU (RCCL refuses two ranks on a device): add their frames in rank order
        reduce_mode = RT_REDUCE_ORDERED;
        for (int r = 1; r < n; r++)
            if (launch_add_frame(root->cfg(), root->frame.p, ctx_of(r)->frame.p, n_floats))
                return fail(RT_ERR_DEVICE, "rt_render_multi: add launch failed");
    }
    HIP_TRY(hipSetDevice(root->device));
    if (launch_add_frame(root->cfg(), d_accum, root->frame.p, n_floats)) return fail(RT_ERR_DEVICE, "rt_render_multi: add launch failed");
    HIP_TRY(hipStreamSynchronize(root->stream));
    const double ms_reduce = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (stats) {
        *stats = work[0].st;
        stats->ms_render_min = work[0].st.ms_render;
        stats->slowest_device = 0;
        for (int r = 0; r < n; r++) {
            const rt_stats& o = work[(size_t)r].st;
            if (r < RT_STATS_MAX_DEVICES) {
                stats->device_ms_render[r] = (float)o.ms_render;
                stats->device_form[r] = (uint8_t)o.kernel_form;
                stats->device_tree[r] = (uint8_t)o.tree;
            }
            if (r == 0) continue;
            stats->rays_traced += o.rays_traced; stats->paths += o.paths;
            stats->box_tests += o.box_tests; stats->tri_tests += o.tri_tests; stats->sph_tests += o.sph_tests;
            stats->trace_launches += o.trace_launches;
            if (o.ms_render > stats->ms_render) { stats->ms_render = o.ms_render; stats->slowest_device = (uint32_t)r; }
            if (o.paths) stats->ms_render_min = std::min(stats->ms_render_min, o.ms_render);
            stats->ms_trace = std::max(stats->ms_trace, o.ms_trace);
            stats->ms_shade = std::max(stats->ms_shade, o.ms_shade);
            stats->ms_kernel_span = std::max(stats->ms_kernel_span, o.ms_kernel_span);
        }
        stats->ms_reduce = ms_reduce;
        stats->n_devices = (uint32_t)n;
        stats->rccl_ranks = rccl_ranks;
        stats->reduce_mode = (uint32_t)reduce_mode;
    }
    return RT_OK;
    RT_GUARD_END("rt_render_multi")
}

int rt_render_multi(rt_scene* s, const rt_camera* cam, const rt_render_params* p, float* accum_rgb, rt_stats* stats) {
    RT_GUARD_BEGIN
    if (!s || !p || !accum_rgb) return fail(RT_ERR_INVALID, "rt_render_multi: NULL argument");
    if (!s->ctx) return fail(RT_ERR_DEVICE, "host-only scene: no GPU context");
    rt_context* ctx = s->ctx;
    int rc = set_device(ctx);
    if (rc) return rc;
    const size_t bytes = (size_t)p->width * p->height * 3 * sizeof(float);
    DeviceBuffer& d = ctx->frame_in;
    HIP_TRY(d.ensure(bytes));
    HIP_TRY(hipMemcpy(d.p, accum_rgb, bytes, hipMemcpyHostToDevice));
    rt_stats local;
    rt_stats* st = stats ? stats : &local;
    rc = rt_render_multi_device(s, cam, p, d.p, st);
    if (rc) return rc;
    const auto t0 = std::chrono::steady_clock::now();
    HIP_TRY(hipMemcpy(accum_rgb, d.p, bytes, hipMemcpyDeviceToHost));
    st->ms_reduce += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return RT_OK;
    RT_GUARD_END("rt_render_multi")
}

int rt_tonemap_device(rt_context* ctx, const void* d_accum, uint32_t spp, void* d_out, size_t npix) {
    RT_GUARD_BEGIN
    if (!ctx || (npix && (!d_accum || !d_out))) return fail(RT_ERR_INVALID, "rt_tonemap_device: NULL argument");
    if (spp == 0) return fail(RT_ERR_INVALID, "spp must be positive");
    int rc = set_device(ctx);
    if (rc) return rc;
    if (launch_tonemap(ctx->cfg(), d_accum, spp, d_out, npix)) return fail(RT_ERR_DEVICE, "tonemap launch failed");
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return RT_OK;
    RT_GUARD_END("rt_tonemap_device")
}

int rt_tonemap(rt_context* ctx, const float* accum_rgb, uint32_t spp, uint32_t* out, size_t npix) {
    RT_GUARD_BEGIN
    if (!ctx || (npix && (!accum_rgb || !out))) return fail(RT_ERR_INVALID, "rt_tonemap: NULL argument");
    if (npix == 0) return RT_OK;
    int rc = set_device(ctx);
    if (rc) return rc;
    DeviceBuffer &in = ctx->frame_in, &o = ctx->frame_out;
    HIP_TRY(in.ensure(npix * 12));
    HIP_TRY(o.ensure(npix * 4));
    HIP_TRY(hipMemcpyAsync(in.p, accum_rgb, npix * 12, hipMemcpyHostToDevice, ctx->stream));
    rc = rt_tonemap_device(ctx, in.p, spp, o.p, npix);
    if (rc) return rc;
    HIP_TRY(hipMemcpy(out, o.p, npix * 4, hipMemcpyDeviceToHost));
    return RT_OK;
    RT_GUARD_END("rt_tonemap")
}

}  // extern "C"
