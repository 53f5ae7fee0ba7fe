/*
 * rt_abi.h -- C ABI of the MI355X (gfx950) wavefront path tracer.
 *
 * The Rust reference has no FFI of its own; the seam this library replaces is the
 * tile-per-thread renderer `render()` (src/main.rs:319-416) together with the closest-hit
 * query `Hittable::hit` (src/shape.rs:6-9) it is built on.  Because `Box<dyn Hittable>` and
 * `Arc<dyn Material>` are opaque trait objects, a scene is *described* to the library with
 * plain structs that mirror the reference types field by field; each entry point below cites
 * the reference item it stands in for.
 *
 * Conventions: every call returns 0 on success or a negative rt_status; nothing throws or
 * aborts across the boundary; rt_last_error() gives the message of the last failure on the
 * calling thread.  Inputs are copied; the caller keeps ownership of everything it passes.
 * Calls on one context must be serialised by the caller.  A context drives one GPU
 * (rt_context_create) or several GPUs of one node (rt_context_create_multi): a scene committed on
 * a multi-GPU context is flattened once and uploaded to every GPU, and rt_render_multi() splits a
 * frame over them and combines the per-GPU accumulation buffers with one RCCL sum-reduce.  A job of
 * one process per GPU does the same with rt_partition() + rt_render_device() and its own reduce.
 */
#ifndef RT_ABI_H
#define RT_ABI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rt_context rt_context;
typedef struct rt_scene rt_scene;

typedef enum {
    RT_OK = 0,
    RT_ERR_INVALID = -1,      /* bad argument */
    RT_ERR_DEVICE = -2,       /* HIP runtime failure (no GPU, launch or copy failed) */
    RT_ERR_STATE = -3,        /* call order (scene not committed, already committed, ...) */
    RT_ERR_UNSUPPORTED = -4,  /* e.g. the wave sweep on a scene with transformed meshes */
    RT_ERR_NOMEM = -5
} rt_status;

/* material.rs:22-25 / 58-61 / 82-85 / 122-124 */
enum { RT_MAT_LAMBERT = 0, RT_MAT_METAL = 1, RT_MAT_DIELECTRIC = 2, RT_MAT_EMISSIVE = 3 };
#define RT_MAT_FLAG_NORMAL_SHADING 1u  /* LambertianMaterial.normal (material.rs:24,48-52) */
typedef struct {
    uint32_t kind;   /* RT_MAT_* */
    float rgb[3];    /* albedo, or emitted colour */
    float param;     /* fuzz (metal) or ior (dielectric) */
    uint32_t flags;
} rt_material;

/* shape.rs:11-15 */
typedef struct { float c[3]; float r; uint32_t mat; } rt_sphere;

/* shape.rs:165-183.  Only the first vertex normal is ever read (shape.rs:298). */
typedef struct { float p[3][3]; float n0[3]; uint32_t mat; } rt_triangle;

/* camera.rs:126-130 */
typedef struct { float o[3]; float d[3]; } rt_ray;

/* camera.rs:92-99 plus the primitive id the reference does not expose.  Primitive ids count
 * objects in insertion order: a sphere takes one id, a mesh one id per triangle in the order
 * the triangles were given. */
#define RT_MISS 0xFFFFFFFFu
typedef struct {
    float t;
    uint32_t prim_id;     /* RT_MISS when nothing was hit */
    float p[3];
    float n[3];           /* face-forwarded normal (camera.rs:112-119) */
    uint32_t front_face;
    uint32_t mat;
} rt_hit;

/* The fields Camera::setup() computes (camera.rs:17-24, 55-78). */
typedef struct {
    float origin[3], horizontal[3], vertical[3], lower_left[3], u[3], v[3];
    float lens_radius;
} rt_camera;

/* What World.hittable is (main.rs:69-73): a Bvh over the object list (main.rs:170,234), the
 * single object itself (main.rs:289) or a HittableList (shape.rs:55). */
enum { RT_ROOT_BVH = 0, RT_ROOT_SINGLE = 1, RT_ROOT_LIST = 2 };

/* main.rs:20-24 made run-time parameters, plus the partition this process renders. */
typedef struct {
    uint32_t width, height;            /* full frame */
    uint32_t x0, y0, x1, y1;           /* pixel rectangle to render, [x0,x1) x [y0,y1) */
    uint32_t sample_begin, sample_end; /* sample indices to render, [begin,end) */
    uint32_t max_bounces;              /* MAX_BOUNCES (main.rs:21) */
    float t_min;                       /* 0.001 in the reference (main.rs:46) */
    float sky_intensity;               /* World.sky_intensity (main.rs:72) */
    uint64_t seed;                     /* key of the counter-based RNG */
    uint32_t flags;                    /* RT_RENDER_* */
    uint32_t max_paths_in_flight;      /* 0 = library default */
    /* Interleaved row bands inside the rectangle, for splitting a frame by image tile over
     * several GPUs with even load: with band_count > 1 only the bands b (of band_rows rows each,
     * counted from y0) with b % band_count == band_index are rendered.  band_rows must be a
     * multiple of 8; band_count 0 or 1 renders the whole rectangle. */
    uint32_t band_rows, band_count, band_index;
    uint32_t reserved;
} rt_render_params;
#define RT_RENDER_PROFILE 1u       /* time the kernels that hold the closest hit with HIP events */
#define RT_RENDER_COUNT_VISITS 2u  /* count box / triangle / sphere tests (slower; implies STAGED) */
#define RT_RENDER_STAGED 4u        /* one launch per stage and depth (raygen, closest hit, shade) instead
                                      of one kernel per batch that carries every segment of paths from
                                      raygen to its last bounce; same image bit for bit, slower, but
                                      the stages can be timed apart */

/* which form of the closest-hit loop the render ran (rt_stats.kernel_form): for a scene whose
 * traversal is RT_TRAVERSAL_AUTO the first render times the forms that apply and keeps the faster,
 * so two runs may differ here (never in the frame) */
enum { RT_FORM_SWEEP = 0, RT_FORM_WALK = 1, RT_FORM_WALK_LDS = 2, RT_FORM_WALK_REFILL = 3,
       RT_FORM_WALK_LDS_REFILL = 4, RT_FORM_ORDERED = 5, RT_FORM_OCTANT = 6, RT_FORM_OCTANT_REFILL = 7,
       RT_FORM_WIDE = 8, RT_FORM_WIDE_REFILL = 9 };
/* which of the scene's trees the render walked (rt_stats.tree) */
enum { RT_TREE_REFERENCE = 0, RT_TREE_SAH = 1 };

typedef struct {
    uint64_t rays_traced;    /* closest-hit queries, i.e. executions of main.rs:46 */
    uint64_t paths;          /* camera samples */
    uint64_t box_tests, tri_tests, sph_tests;   /* with RT_RENDER_COUNT_VISITS */
    uint64_t trace_launches; /* launches of the kernel that holds the closest hit (one per batch, or
                                one per batch and depth with RT_RENDER_STAGED) */
    double ms_render;        /* wall time of the call on the GPU stream */
    double ms_trace;         /* with RT_RENDER_PROFILE: summed time of those kernels */
    double ms_shade;         /* with RT_RENDER_PROFILE and RT_RENDER_STAGED: summed shade kernel time */
    double ms_reduce;        /* rt_render_multi: host wall time of the sum-reduce (and the copy to the host) */
    uint32_t n_devices;      /* GPUs that rendered (1 for rt_render / rt_render_device) */
    uint32_t kernel_form;    /* RT_FORM_* the render ran (device 0's, for rt_render_multi) */
    uint32_t tree;           /* RT_TREE_* the render walked (device 0's, for rt_render_multi) */
    uint32_t rccl_ranks;     /* rt_render_multi: ranks of the RCCL communicator the reduce ran on (ncclCommCount);
                                0 when the frames were added without RCCL */
    double ms_kernel_span;   /* with RT_RENDER_PROFILE: from the start of the first to the end of the last of the
                                kernels that hold the closest hit, over all pipelines (they overlap, so this, not
                                their sum ms_trace, is what the call's wall time contains) */
    double ms_render_min;    /* rt_render_multi: the fastest GPU's render time (ms_render is the slowest's) */
    uint32_t slowest_device; /* rt_render_multi: index into dev_ids of the GPU whose render took longest */
    uint32_t reduce_mode;    /* rt_render_multi: RT_REDUCE_* the frames were combined with */
    double shader_clock_ghz; /* with RT_RENDER_PROFILE (not STAGED): the shader clock the per-batch kernels ran at,
                                from the cycle counter and the constant-rate 100 MHz counter read by one wave of
                                every launch at its start and at its end (0: not measured) */
    /* rt_render_multi: per GPU (the first RT_STATS_MAX_DEVICES of the context, in dev_ids order) the render
     * time, the RT_FORM_* and the RT_TREE_* of its share */
    float device_ms_render[16];
    uint8_t device_form[16];
    uint8_t device_tree[16];
} rt_stats;
#define RT_STATS_MAX_DEVICES 16

/* ---- context: stands where ThreadPool::new() does (main.rs:443) ---------------------- */
int rt_context_create(int device_id, rt_context** out);
/* One context over n_dev GPUs of this node (render() owns all its workers, main.rs:319-336, 443):
 * a stream set per GPU, one host thread per GPU during rt_render_multi(), and an RCCL communicator
 * over the GPUs (ncclCommInitAll).  dev_ids[0] is the root: it receives the reduced frame, and every
 * single-GPU entry point (rt_trace, rt_render, rt_tonemap, ...) runs on it.  If dev_ids names one GPU
 * more than once (several workers on one GPU: test rigs) RCCL, which refuses that, is left out
 * and the buffers, which then all live on that GPU, are added in rank order by a kernel. */
int rt_context_create_multi(int n_dev, const int* dev_ids, rt_context** out);
int rt_context_num_devices(const rt_context* ctx);
void rt_context_destroy(rt_context* ctx);
const char* rt_last_error(void);
/* the HIP stream the context launches on (a hipStream_t), for callers that order their own
 * work (e.g. the RCCL reduce) after a render */
void* rt_context_stream(rt_context* ctx);
/* waits for everything queued on the context's stream */
int rt_context_synchronize(rt_context* ctx);

/* ---- scene: Sphere / Mesh::new / Bvh::new / HittableList (shape.rs) ------------------ */
/* ctx may be NULL for a host-only scene: it can be filled, committed and inspected with
 * rt_scene_get_entries(), but tracing or rendering it fails with RT_ERR_DEVICE */
int rt_scene_create(rt_context* ctx, rt_scene** out);
void rt_scene_destroy(rt_scene* scene);
int rt_scene_set_materials(rt_scene* scene, const rt_material* mats, size_t n);
/* each sphere is one object, appended in order (Vec<Box<dyn Hittable>>, main.rs:100-158) */
int rt_scene_add_spheres(rt_scene* scene, const rt_sphere* spheres, size_t n);
/* one Mesh object (Mesh::new, shape.rs:236-257).  xform is Mesh.transform as a column-major 4x4
 * (glam's Mat4: x_axis, y_axis, z_axis, w_axis) or NULL for the identity.  With a transform the
 * mesh is traced as the reference traces it: the tree lives in the mesh's local space, a ray that
 * reaches the mesh is taken there with the inverse transform (camera.rs:137-142, shape.rs:454),
 * and the hit point and normal come back through the transform (shape.rs:303-307).  The Mat4
 * arithmetic is restated from glam's published source and cannot be checked against the
 * reference here (it ships no scene with a transform).  A scene with transformed meshes is traced
 * by the per-lane walk only. */
int rt_scene_add_mesh(rt_scene* scene, const rt_triangle* tris, size_t n, const float* xform);
/* Containers inside the object list (Bvh.left/right and HittableList.objects hold any Hittable,
 * shape.rs:55-57, 87-91): the objects added between begin and end are the children of ONE
 * container object -- Bvh::new(children, axis) for RT_ROOT_BVH, a HittableList for RT_ROOT_LIST --
 * which is itself a child of the enclosing container (or of the scene's root).  Groups nest.
 * Primitive ids count spheres and triangles in the order they were added, whatever their nesting. */
int rt_scene_begin_group(rt_scene* scene, int kind, int axis);
int rt_scene_end_group(rt_scene* scene);
int rt_scene_set_root(rt_scene* scene, int root_kind);
/* How the trees inside meshes are split (before rt_scene_commit).  REFERENCE: Mesh::build_bvh's
 * stable median split on min[axis] (shape.rs:264-290): the reference's trees, and in the PER_LANE
 * and SWEEP traversals the reference's visiting order, hits exact by construction.  SAH has no
 * counterpart in the reference: the same entry format, box padding and one triangle per leaf, but
 * nodes are split where a binned surface-area heuristic is cheapest, which cuts the box tests per
 * ray severalfold on meshes that mix large and small triangles.  The box and primitive tests are
 * the reference's and exact ties in t are resolved by primitive key (see rt_scene_get_prim_order),
 * so the closest hit is the reference's primitive at the reference's t on every ray the tests and
 * fuzz campaigns of this library have traced with |d|^2 in [1e-30, 1e18); what is not excluded by
 * construction is a box that rounding closes in front of a primitive whose own test accepts the
 * ray, which another tree's boxes need not repeat, and outside that range of |d|^2 the reference's
 * sphere test overflows or works on denormals and reports spheres whose boxes the ray misses, so
 * its result there depends on which of its own boxes the ray passes (the surface-area trees group
 * the objects of a Bvh their own way).  AUTO (the default) builds both: the explicit traversal modes run on the
 * reference's trees, and RT_TRAVERSAL_AUTO may move the scene to the surface-area trees on its
 * first render (see rt_scene_set_traversal).  A scene without a GPU context builds the
 * reference's trees only under AUTO. */
enum { RT_BUILD_REFERENCE = 0, RT_BUILD_SAH = 1, RT_BUILD_AUTO = 2 };
int rt_scene_set_build(rt_scene* scene, int build_mode);
/* the build mode of scenes created from now on (process-wide; the scene constructors of
 * rt_host.h create their scenes internally and take it from here) */
int rt_set_default_build(int build_mode);
/* Binary scene cache (no counterpart in the reference, whose ingest -- Mesh::from_file and
 * Mesh::new, shape.rs:236-257, 347-406 -- parses and builds on every start).  With a cache
 * directory, rt_scene_commit() looks for <dir>/<key>.rtscene, where the key is a 64-bit hash of
 * everything the flattened arrays depend on (materials, every sphere and triangle, transforms,
 * containers, root kind, build mode, layout parameters); a complete file of that key is loaded
 * instead of building the trees, otherwise they are built and the file is written (to a temporary
 * name first, then renamed, so concurrent processes never see half a file).  The directory must
 * exist.  rt_set_default_cache_dir() sets the directory of scenes created afterwards (NULL or ""
 * turns it off), the scene constructors of rt_host.h included. */
enum { RT_CACHE_OFF = 0, RT_CACHE_STORED = 1, RT_CACHE_LOADED = 2, RT_CACHE_STORE_FAILED = 3 };
int rt_set_default_cache_dir(const char* dir);
int rt_scene_set_cache_dir(rt_scene* scene, const char* dir);
int rt_scene_cache_state(const rt_scene* scene);   /* what the commit did: RT_CACHE_* */
/* How the closest-hit kernel walks the flattened trees.  PER_LANE and SWEEP visit the same
 * entries in the same order with the same arithmetic as the reference's recursion
 * (shape.rs:142-158, 292-324) and give bit-identical hits by construction (on the reference's
 * trees); they differ in speed only.  PER_LANE: every lane follows its own cursor (any scene
 * size).  SWEEP: the wave sweeps the entry array once with scalar fetches (small scenes).
 * AUTO takes the fastest form whose hits are the reference's (rt_stats.kernel_form and
 * rt_stats.tree say which).  Every form reports the reference's primitive and t on every ray (ties
 * in t by primitive key), so the frame does not depend on the choice.  On the reference's trees the
 * forms are the sweep (up to 192 entries) and the per-lane walk (out of LDS when the trees fit, else
 * from memory, plain or with idle lanes refilled); a scene of RT_BUILD_AUTO has the same forms on
 * its surface-area trees and, when those do not fit LDS, the wide walk over them.  What is decided
 * by rule (from measurements on every workload): of two trees the one with the lower
 * rt_scene_expected_cost() (the surface-area trees on every scene tried), and the wide walk where
 * it applies.  What is decided by timing, on the first render whose first
 * batch has 8 Mi paths or more (that batch, once per candidate, output discarded): sweep against
 * walk out of LDS for scenes of up to 192 entries, plain against refilled walk for scenes walked
 * from memory without a wide tree.  A smaller frame runs on the likelier form and leaves the
 * timing to a later one.
 * Environment: RT_AUTOTUNE_LOG=1 prints the timings on stderr, RT_NO_AUTOTUNE=1 keeps the scene's
 * first form, RT_AUTO_PROBE_ALL=1 times every form on either tree, RT_PROBE_ANY_SIZE=1 times on
 * frames of any size.
 * ORDERED has no counterpart in the reference: the same trees, the same box and primitive tests,
 * but at every box the child nearer along the ray is visited first (per-lane stack), so the
 * narrowed t_max culls far more boxes; ties in t are resolved by primitive key.  Fails with
 * RT_ERR_UNSUPPORTED for scenes of 2^29 entries or more, or boxes nested deeper than 64.
 * OCTANT visits what ORDERED visits, without a stack: the device holds eight copies of the
 * flattened trees, one per sign combination of a ray's direction, each in the pre-order in which
 * that octant's rays meet the nearer child of every box first; a ray walks the copy of its octant
 * with the same skip links as PER_LANE.  Eight times the tree memory (the 10 M-triangle scene: about
 * 6.5 GB of the 288 GB) buys the ordered walk's culling at the plain walk's cost per visit.  Same
 * limits as ORDERED; needs a device context.
 * WIDE walks another tree: on its first use the scene's flattened trees are collapsed into nodes of
 * up to eight children whose boxes the node stores itself, quantised to 8 bits per plane on a grid
 * of the node (80 bytes per node, always enclosing the true boxes), with every primitive a leaf of
 * its own.  A lane fetches a node, tests its eight boxes at once, and visits the children that were
 * hit nearest first, the rest waiting on a per-lane stack of one word per level.  The primitive
 * tests are the reference's, so t is theirs, and ties go by primitive key.  A transformed mesh
 * (Mesh::new with a transform, shape.rs:236-257) is a node of its own, collapsed in the mesh's
 * space: a lane that reaches it takes its ray there (Ray::transform, camera.rs:137-142, as
 * Mesh::hit does, shape.rs:454) and restores it when the mesh's subtree is exhausted.  Fails with
 * RT_ERR_UNSUPPORTED for trees deeper than 60 wide levels or of 2^24 node slots and more.
 * The explicit modes run on the reference's trees (on the surface-area trees for a scene built with
 * RT_BUILD_SAH).  May be called before or after rt_scene_commit(). */
enum { RT_TRAVERSAL_AUTO = 0, RT_TRAVERSAL_PER_LANE = 1, RT_TRAVERSAL_SWEEP = 2, RT_TRAVERSAL_ORDERED = 3,
       RT_TRAVERSAL_OCTANT = 4, RT_TRAVERSAL_WIDE = 5 };
int rt_scene_set_traversal(rt_scene* scene, int mode);
/* Where the tree of RT_TRAVERSAL_WIDE is made (before its first use; no counterpart in the
 * reference, which builds on the host: shape.rs:94-138, 264-290).  COLLAPSE (default): on the host,
 * from the scene's flattened binary trees, once for all GPUs of a context.  DEVICE: on each GPU, from
 * the scene's primitives alone -- padded boxes, 30-bit Morton codes of their centres, a radix sort,
 * then one launch per level down (a node's children are the runs of equal next Morton digit, which
 * is the child slot) and one per level up (boxes, 8-bit planes): milliseconds instead of seconds
 * for 10 M triangles.  Same node format, same walk, same guarantees; the trees are as good as the
 * collapsed ones where primitives are small and evenly spread, worse where sizes are mixed.  The
 * node children of a node take the slots from its lowest to its highest occupied child slot, not
 * first-fit compacted as on the host, so the limit of 2^24 node slots is reached earlier (the
 * 10 M-triangle soup: 6.1 M slots here, 3.7 M collapsed): roughly 27 M evenly spread triangles.
 * DEVICE does not take scenes with transformed meshes (RT_ERR_UNSUPPORTED when the tree is made). */
enum { RT_WIDE_BUILD_COLLAPSE = 0, RT_WIDE_BUILD_DEVICE = 1 };
int rt_scene_set_wide_build(rt_scene* scene, int where);
/* builds the trees of the scene's build mode (the reference topology: shape.rs:94-138, 264-290),
 * flattens them and uploads */
int rt_scene_commit(rt_scene* scene);
size_t rt_scene_num_prims(const rt_scene* scene);
/* What a ray that enters the scene's bounds is expected to cost on the scene's trees (of a scene of
 * RT_BUILD_AUTO: on its reference-topology trees), in box tests: the surface-area measure -- an entry
 * is tested when the ray passes the box above it, for uniformly distributed rays with probability
 * area(box) / area(scene bounds); a triangle test counts 1.6 box tests, a sphere test 1.3.  A scene
 * of RT_BUILD_AUTO renders on whichever of its two trees has the lower figure.  0: not committed or
 * not known (an empty or unbounded scene).  No counterpart in the reference. */
double rt_scene_expected_cost(const rt_scene* scene);
size_t rt_scene_num_entries(const rt_scene* scene);   /* flattened traversal entries */
/* host-only scenes: copies the flattened pre-order entry array (16 floats per entry).  Word 3 of an
 * entry holds its kind in bits 30-31 (0 box, 1 triangle, 2 sphere) and, for a box, the index of the
 * entry behind its subtree; word 7 of a triangle or sphere entry holds the primitive's KEY. */
int rt_scene_get_entries(const rt_scene* scene, float* out, size_t n_floats);
/* Primitive keys (no counterpart in the reference, which has no primitive ids at all).  Inside the
 * library a primitive is named by a key made from its RANK: its place in the order in which the
 * reference's recursion visits the primitives (Bvh::hit / Mesh::bvh_hit left child first,
 * shape.rs:142-158, 292-324; HittableList in order, shape.rs:60-74), i.e. the pre-order of the
 * reference-topology trees -- whatever tree and visiting order the scene is traced with:
 *     triangle: key = 0x3FFFFFFE - rank          sphere: key = 0x40000000 | rank
 * The keys carry the reference's resolution of exact ties in t into walks that visit in another
 * order.  The reference lets a triangle replace the closest hit only when t < t_max (shape.rs:297)
 * and a sphere when root <= t_max (shape.rs:32-36); so of the primitives that tie at the smallest
 * t it reports the sphere it visits last, or without a sphere the triangle it visits first: the
 * primitive of the largest key.  Every walk that is not in the reference's order accepts a
 * primitive when t < t_max, or when t == t_max and its key is the larger one.
 * rt_scene_get_prim_order() copies the table rank -> primitive id (n = rt_scene_num_prims()). */
int rt_scene_get_prim_order(const rt_scene* scene, uint32_t* out, size_t n);
/* host-only scenes: the same entries as the per-lane walks read them, in 16-byte units (4 floats):
 * a box or a sphere is the first 8 floats of its entry in two units, a triangle the first 11 in
 * three, followed by the distance to the next entry (3, or 4 where a padding unit keeps a box or
 * sphere 32-byte aligned); skip links count units */
size_t rt_scene_num_units(const rt_scene* scene);
int rt_scene_get_units(const rt_scene* scene, float* out, size_t n_floats);
/* host-only scenes: the units as the reference-order per-lane walks read them.  The same entries
 * (a box or sphere in two 16-byte units, a triangle in three), but every entry names its
 * successors itself -- a box: word 3 = tag | skip (where the walk goes when the box is missed),
 * word 7 = its first child; a triangle: word 11, a sphere: word 5 = the entry behind it; all
 * unit indices, the index one past the last unit meaning "done" -- so the order in memory is free
 * and is chosen for locality (the children of a box side by side, a pair of boxes inside one
 * 64-byte line, the top levels of the trees first).  Entry 0 of rt_scene_get_entries() is at unit
 * 0; following the links from there visits the entries in the reference's order. */
size_t rt_scene_num_walk_units(const rt_scene* scene);
int rt_scene_get_walk_units(const rt_scene* scene, float* out, size_t n_floats);
/* host-only scenes: the wide tree RT_TRAVERSAL_WIDE walks (no counterpart in the reference).  A
 * node is 20 words: 0-2 the origin of its grid (floats); 3 the exponent bytes of the grid step
 * along x, y, z (step = 2^(e - 127)) and, in bits 24-31, the mask of child slots that hold nodes;
 * 4 the node slot of child slot 0 (child slot s lives at node slot [4] + s); 5 the index of the
 * node's first primitive record; 6 the mask of child slots that hold primitives (their records
 * follow each other in slot order); 7 in its upper 24 bits the number (from 1) of the transform when the
 * node is the root of a transformed mesh's tree (whose nodes are in the mesh's own space), in its low
 * byte the mask of child slots whose objects the reference tests
 * without a box test before them (the objects of a HittableList at the root: entered whatever their
 * boxes say by rays whose |d|^2 is outside [1e-30, 1e18), where the reference's sphere test overflows
 * or works on denormals and accepts points that have nothing to do with the sphere's box); 8-19 the children's planes, one byte per slot, two words per
 * plane in the order min x, min y, min z, max x, max y, max z (plane = origin + byte * step).  A
 * primitive record is 16 floats: the first 12 floats of the primitive's entry in
 * rt_scene_get_entries().  Node slot 0 is the root; depth = levels of nodes.  node_words is the
 * stride of the node array in words: 20, or 32 (the last 12 unused) for trees of more than 32 MB,
 * which are laid out one node per 128-byte line. */
int rt_scene_wide_info(const rt_scene* scene, size_t* n_nodes, size_t* n_prims, uint32_t* depth, uint32_t* node_words);
int rt_scene_get_wide(const rt_scene* scene, uint32_t* nodes, size_t n_node_words, float* prims, size_t n_prim_floats);

/* ---- closest hit: Hittable::hit (shape.rs:7) on a batch -- the hit-parity surface ------ */
int rt_trace(rt_scene* scene, const rt_ray* rays, size_t n, float t_min, float t_max, rt_hit* out);
/* same with device pointers; counts[3] (may be NULL, host memory) receives the box / triangle /
 * sphere test totals of the batch */
int rt_trace_device(rt_scene* scene, const void* d_rays, size_t n, float t_min, float t_max,
                    void* d_hits, uint64_t* counts);

/* No counterpart in the reference.  rt_trace_device() with visit counts for callers that cannot
 * allocate device memory (code under HIP graph capture, tight serving loops, harnesses that forbid
 * hipMalloc): it calls no hipMalloc / hipFree or other allocating routine, and it does not wait for
 * the stream -- the work is queued on rt_context_stream() and the caller orders itself after it
 * (rt_context_synchronize(), or an event on that stream).  The per-workgroup partial counts that
 * rt_trace_device() keeps in a buffer it allocates live here in a static device array of the
 * library (512 KB on every GPU the library runs on, held until the process ends): one 32-byte
 * record per workgroup of 256 rays, hence at most RT_TRACE_STATIC_MAX_RAYS rays per call (more:
 * RT_ERR_INVALID).  d_counts is DEVICE memory for three uint64_t (box, triangle, sphere tests) and
 * is overwritten.  Every call on a GPU uses that GPU's one array: calls that overlap on one GPU
 * (two contexts, or two host threads) race; calls queued one after the other on the context's
 * stream do not.  Nothing in the array has to start at zero (a call overwrites the records it
 * reads).  Hits are those of rt_trace_device(). */
#define RT_TRACE_STATIC_MAX_RAYS 4194304u
int rt_trace_device_static(rt_scene* scene, const void* d_rays, size_t n, float t_min, float t_max,
                           void* d_hits, void* d_counts);

/* ---- render: render() and its tile closure (main.rs:319-416) --------------------------- */
/* Adds, for every pixel of the rectangle, the sum over the sample range of the path radiance
 * (ray_color, main.rs:40-67) to accum_rgb (width*height*3 floats, linear, row-major, host
 * memory).  The caller zeroes the buffer before the first call and divides by the sample
 * count (or calls rt_tonemap) afterwards. */
int rt_render(rt_scene* scene, const rt_camera* cam, const rt_render_params* params,
              float* accum_rgb, rt_stats* stats);
/* same, accumulating into a device buffer; returns when the frame part is complete in that
 * buffer.  This is the call a multi-GPU job makes on every rank before it sum-reduces the
 * buffers to rank 0. */
int rt_render_device(rt_scene* scene, const rt_camera* cam, const rt_render_params* params,
                     void* d_accum_rgb, rt_stats* stats);

/* The share of a job that rank `rank` of `n_ranks` renders (the tile closure's work split,
 * main.rs:332-336, over GPUs instead of threads): on entry *params describes the whole job
 * (rectangle, sample range, no bands); on return its sample range and band_* fields describe the
 * rank's share -- the frame is cut two ways at once, into interleaved bands of 64 rows (the
 * reference's block height, main.rs:22) and into sample ranges; every (pixel, sample) belongs to
 * exactly one rank.  With n >= 4 (even) ranks there are two band groups and n / 2 sample ranges,
 * otherwise one band group and n sample ranges; with fewer samples than ranges the split is by
 * bands alone, and a rank whose band group holds no band renders nothing (rt_render_device then
 * returns RT_OK with zero counts). */
int rt_partition(uint32_t n_ranks, uint32_t rank, rt_render_params* params);
/* render() on every GPU of a multi-GPU context: GPU r renders rt_partition(n, r) of the job into
 * its own full-frame buffer (the RNG is keyed by pixel and sample, so the split changes nothing
 * but f32 summation grouping), then ONE ncclReduce(sum, f32, root = GPU 0) over xGMI combines
 * them, and the result is added to accum_rgb (host memory, width*height*3; zeroed by the caller
 * before the first call, like rt_render).  params->band_* must be unset.  Deterministic for a
 * fixed seed and GPU count.  stats: rays and paths summed over the GPUs, ms_render = the slowest
 * GPU's, ms_reduce = the reduce.  Works on a one-GPU context too (the same code path). */
int rt_render_multi(rt_scene* scene, const rt_camera* cam, const rt_render_params* params,
                    float* accum_rgb, rt_stats* stats);
/* same, adding into a buffer on GPU 0 (the root) of the context */
int rt_render_multi_device(rt_scene* scene, const rt_camera* cam, const rt_render_params* params,
                           void* d_accum_rgb, rt_stats* stats);
/* How rt_render_multi combines the GPUs' frames (no counterpart in the reference, whose workers write
 * disjoint tiles of one buffer, main.rs:397-405).  RCCL (default): one ncclReduce(sum, f32, root 0);
 * the order in which RCCL adds the partial sums of a pixel is its own, fixed for a given node
 * topology and GPU count.  ORDERED: every peer's frame is copied to the root (hipMemcpyPeerAsync
 * over xGMI) and added there by a kernel in rank order, so the sum of a pixel is
 * ((f0 + f1) + f2) + ... whatever the topology: the strict-order form for callers who need frames
 * that are bit-identical across machines.  A context whose dev_ids name one GPU several times always
 * adds in rank order.  The environment variable RT_REDUCE_ORDERED=1 makes ORDERED the default of
 * contexts created afterwards. */
enum { RT_REDUCE_RCCL = 0, RT_REDUCE_ORDERED = 1 };
int rt_context_set_reduce(rt_context* ctx, int mode);
/* The version of the RCCL library this process resolved (ncclGetVersion: major * 10000 + minor * 100
 * + patch; 0 if the call fails).  The library is linked against librccl.so.1 by name: a process that
 * loaded PyTorch first runs on the RCCL bundled with it, any other on the system's. */
int rt_rccl_version(void);

/* ---- tonemap: main.rs:369-373 + rgb_to_u32 (math.rs:112-116) --------------------------- */
int rt_tonemap(rt_context* ctx, const float* accum_rgb, uint32_t spp, uint32_t* out_0rgb, size_t npix);
int rt_tonemap_device(rt_context* ctx, const void* d_accum_rgb, uint32_t spp, void* d_out_0rgb,
                      size_t npix);

#ifdef __cplusplus
}
#endif
#endif
