// Internal interfaces shared by the host scene builder, the C-ABI glue and the HIP kernels.
#pragma once

#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

#include "rt_abi.h"

namespace rtamd {

// ---------------------------------------------------------------------------------------------
// Flattened scene.
//
// The closest-hit query of the reference walks its trees left child first and narrows t_max to
// the closest hit found so far (shape.rs:142-158, 292-324).  That fixed visiting order is a
// pre-order walk, so the trees are stored as ONE array of entries in pre-order with a skip
// link per box: visit entry i; a box that is hit continues at i+1 (its left child), a box that
// is missed continues at its skip link (the entry after its subtree); a primitive continues
// at i+1.  No stack, and every box test sees exactly the t_max the reference's recursion would.
//
// One entry = four float4 (64 bytes, so that an entry is one aligned 64-byte fetch: a single
// s_load_dwordx16 for the wave sweep, one cache sector for the per-lane walk):
//   box      a = (min.xyz , TAG_BOX<<30 | skip)   b = (max.xyz , 0)
//   triangle a = (v0.xyz  , TAG_TRI<<30)          b = (e1.xyz  , prim key)  c = (e2.xyz, 0)
//   sphere   a = (c.xyz   , TAG_SPH<<30)          b = (r, 0, 0, prim key)
// The fourth float4 is padding.  e1 = v1 - v0 and e2 = v2 - v0 are the reference's v0v1 / v0v2
// (shape.rs:417-418) computed once instead of per test.
//
// Primitive keys.  Inside the library a primitive is not named by its id (its place in the order
// the caller added it) but by a key made from its RANK, its place in the order the reference's
// recursion visits the primitives (the pre-order of the reference-topology trees, whatever tree
// the scene is traced with):
//   triangle  key = KEY_NONE - 1 - rank        sphere  key = KEY_SPHERE | rank
// The per-primitive tables (prim_geo, prim_meta) are indexed by rank, and FlatScene::prim_order
// maps rank -> id for the one place that reports ids (rt_trace's hit records).  The keys make the
// reference's resolution of EXACT ties in t independent of the visiting order.  In the reference
// a triangle replaces the closest hit only when t < t_max (shape.rs:297) and a sphere when
// root <= t_max (shape.rs:32-36), so among the primitives that tie at the smallest t the winner
// is the sphere visited last if there is one, else the triangle visited first -- the primitive
// of the LARGEST key.  A walk that visits in another order (another tree, nearer child first)
// therefore accepts a primitive when t < best_t, or when t == best_t and its key is larger than
// the key in hand; KEY_NONE ("nothing hit yet") lies above every triangle and below every sphere,
// which is the reference's behaviour at t == t_max.  The reference-order walks need none of that
// and keep the plain comparisons.
// ---------------------------------------------------------------------------------------------
constexpr uint32_t ENTRY_FLOATS = 16;
constexpr uint32_t TAG_BOX = 0, TAG_TRI = 1, TAG_SPH = 2;
constexpr uint32_t TAG_END = 3;   // behind the last entry (where every walk stops) ...
// ... and, in scenes with transformed meshes only, the two markers around such a mesh's tree: the
// first word of the entry tells them apart (END_CODE_END = 0: the end).  ENTER carries the index of
// the mesh's transform in its second word: the ray goes to the mesh's local space
// (Ray::transform(inverse_transform), shape.rs:454); LEAVE restores the caller's ray.  Both are
// two-unit entries that name the entry behind them like a sphere does.
constexpr uint32_t END_CODE_END = 0, END_CODE_ENTER = 1, END_CODE_LEAVE = 2;
constexpr uint32_t TAG_SHIFT = 30;
constexpr uint32_t SKIP_MASK = (1u << TAG_SHIFT) - 1u;
// Box entries, fourth float of the second float4 (unused by the reference-order walks): links for
// the ordered walk.  Bits 0-28: entry index of the second child (0: the box has one child; the
// first child is always the next entry); bits 29-30: the axis along which the two children are
// furthest apart; bit 31: the first child is the lower one along that axis.
constexpr uint32_t ORD_RIGHT_MASK = (1u << 29) - 1u;
constexpr uint32_t ORD_AXIS_SHIFT = 29;
constexpr uint32_t ORD_FIRST_LOWER = 1u << 31;
constexpr uint32_t ORDERED_MAX_DEPTH = 64;   // 4 waves x 64 lanes x 64 slots x 4 B = 64 KB of LDS per workgroup
// the unit array (and each of its octant copies) is followed by this many TAG_END units, where the
// links of the last entries lead: the per-lane walks fetch their next entry unconditionally
// (four: a lane fetches up to four units at its cursor)
constexpr uint32_t UNITS_END = 4;
constexpr uint32_t KIND_SHIFT = 28;
constexpr uint32_t MAT_MASK = (1u << KIND_SHIFT) - 1u;
constexpr uint32_t PRIM_BACKFACE = 1u << 31;       // flag in the compact hit record
constexpr uint32_t PRIM_ID_MASK = 0x3FFFFFFFu;      // the rank bits of a key
constexpr uint32_t KEY_SPHERE = 1u << 30;
constexpr uint32_t KEY_NONE = 0x3FFFFFFFu;          // no hit yet (tie-aware walks); ranks stay below KEY_NONE - 1
constexpr uint32_t KEY_MASK = 0x7FFFFFFFu;          // a hit record's key (bit 31: PRIM_BACKFACE)
constexpr uint32_t MAX_PRIMS = KEY_NONE - 1u;
inline uint32_t prim_key(uint32_t rank, bool sphere) { return sphere ? (KEY_SPHERE | rank) : (KEY_NONE - 1u - rank); }
constexpr uint32_t META_SPHERE = 1u << 31;         // flag in prim_meta

struct FlatScene {
    std::vector<float> entries;        // ENTRY_FLOATS per entry
    std::vector<float> prim_geo;       // 4 floats per primitive, by rank: tri (unit n0, 0) / sphere (c, r)
    std::vector<uint32_t> prim_meta;   // by rank: material index | material kind << KIND_SHIFT | META_SPHERE
    std::vector<uint32_t> prim_order;  // rank -> primitive id (see "Primitive keys" above)
    std::vector<float> materials;      // 8 floats per material: rgb, param, kind, flags, 0, 0
    // 32 floats per transformed mesh: Mesh.transform then Mesh.inverse_transform (shape.rs:212-213),
    // column-major.  A triangle of such a mesh keeps its raw first-vertex normal in prim_geo and the
    // transform's index + 1 in the fourth word (0: no transform).
    std::vector<float> xforms;
    uint32_t kinds_present = 0;        // bit k set when a material of kind k is referenced
    bool has_spheres = false;
    // The same entries packed for the per-lane walks in 16-byte units (one float4 each): a box or a
    // sphere is the first two float4 of its entry, a triangle the first three; skip links and child
    // links count units.
    std::vector<float> units;
    size_t n_units() const { return units.size() / 4 - UNITS_END; }   // without the end units
    // The units once more in the order and with the links the reference-order per-lane walks use.
    // Every entry names its successors itself -- a box `skip` (a.w, where to go when it is missed)
    // and `child` (b.w, its first child), a triangle `next` (c.w) and a sphere `next` (b.y), all as
    // unit indices -- so memory order is free, and it is chosen for locality: the children of a box
    // lie side by side (a box that is hit gets BOTH children tested, so the 64 bytes of a pair of
    // boxes are one fetch that is used in full), a pair of boxes never straddles a 64-byte
    // boundary, the first child's own children follow the pair, and the top levels of the trees
    // come first, breadth first (walk_top_units of them: the part a kernel may keep in LDS).
    // Entry 0 is at unit 0.  Visiting order and arithmetic are those of `units`.
    std::vector<float> walk;
    size_t n_walk() const { return walk.size() / 4 - UNITS_END; }
    uint32_t walk_top_units = 0;       // units at the front of `walk` laid out breadth first
    uint32_t depth = 0;                // nesting depth of the boxes (slots the ordered walk's stack needs)
    bool ordered_ok = false;           // box entries carry the child links of the ordered walk
    bool multi_child = false;          // some box has more than two children (the ordered walk's stack cannot take it)
    size_t n_entries() const { return entries.size() / ENTRY_FLOATS; }
    size_t n_prims() const { return prim_meta.size(); }
};

struct SceneObject {
    bool is_mesh = false;
    rt_sphere sphere{};
    std::vector<rt_triangle> tris;
    uint32_t prim_base = 0;
    bool has_xform = false;            // Mesh.transform is not Mat4::identity() bit for bit
    float xform[16] = {};              // column-major
};

// A container object (Bvh or HittableList, shape.rs:55-57, 87-91) and its children in order:
// item >= 0 is objects[item], item < 0 is groups[-1 - item].  groups[0] is the top level, whose
// kind is the scene's root kind.
struct SceneGroup {
    int kind = RT_ROOT_BVH;
    int axis = 0;                      // Bvh::new(objects, axis)
    std::vector<long> items;
};

// Builds the reference-topology trees and flattens them.  Returns an empty string on success,
// otherwise the error message.
// build_mode RT_BUILD_SAH replaces the reference's median split inside meshes by a binned
// surface-area-heuristic split (same entry format, same box padding, one triangle per leaf).
// `reference`: the same scene flattened with RT_BUILD_REFERENCE, when the caller has it: a build
// of another mode takes the primitives' ranks from it instead of building the reference's trees
// once more for their order.
std::string build_flat_scene(const std::vector<rt_material>& mats, std::vector<SceneObject>& objects,
                             const std::vector<SceneGroup>& groups, int root_kind, int build_mode, FlatScene& out,
                             const FlatScene* reference = nullptr);

// Binary scene cache: the flattened arrays of a scene on disk, keyed by what they were built from.
// scene_cache_key() hashes the description (materials, every sphere and triangle, transforms,
// containers, root kind) together with the build mode and the layout parameters; the file name is
// that key, and the file repeats it in its header with the array sizes.  load returns false (and
// leaves `out` untouched) when the file is absent, truncated or of another key or format version.
uint64_t scene_cache_key(const std::vector<rt_material>& mats, const std::vector<SceneObject>& objects,
                         const std::vector<SceneGroup>& groups, int root_kind, int build_mode);
bool scene_cache_load(const std::string& path, uint64_t key, FlatScene& out);
bool scene_cache_store(const std::string& path, uint64_t key, const FlatScene& flat);   // temp file + rename

// The unit array once per direction octant (bit k of the octant: the ray's direction is negative
// along axis k), copy o at units [o * (n_units + UNITS_END), ...) with its own end units and links
// that count from the start of the whole array: in copy o the two children of every box are stored nearer child first
// for the rays of that octant (by the axis and order annotate_children recorded), so the plain
// skip-link walk meets them front to back.  `units` must carry the child links (ordered_ok).
void build_octant_units(const uint32_t* units, size_t n_units, std::vector<uint32_t>& out);   // units as 32-bit words
// (the copies carry the explicit links of FlatScene::walk -- child, next -- each in its own pre-order)

// The wide tree of the opt-in RT_TRAVERSAL_WIDE walk (rt_wide.cpp): nodes of up to eight children
// whose boxes the node stores itself, quantised to 8 bits per plane.  One node = 20 words (80 bytes):
//   0-2   origin of the node's grid (float)
//   3     exponent bytes of the grid step along x, y, z (step = 2^(e - 127)) | mask of the slots that hold nodes << 24
//   4     node slot of child slot 0 (the child in slot s lives at node slot [4] + s)
//   5     index of the node's first primitive record (those of its primitive slots follow in slot order)
//   6     mask of the slots that hold primitives
//   7     mask of the slots whose children the reference reaches without a box test (objects of a list at the root)
//         | (index of the transform + 1) << 8 when the node is the root of a transformed mesh's tree
//   8-19  planes, one byte per slot: min x, min y, min z, max x, max y, max z (two words each);
//         plane = origin + byte * step
// One primitive record = 16 floats: the first three float4 of the primitive's entry.
constexpr uint32_t WIDE_NODE_WORDS = 20;
// A node array of more than WIDE_PACKED_MAX_BYTES is stored with one node per 128-byte line (stride
// WIDE_LINE_WORDS, the last 12 words unused): such a tree is read from HBM, where a node that
// straddles two lines costs two fetches (10 M-triangle soup: +15 %); smaller trees live in the caches,
// where the packed array's smaller footprint is worth more.
constexpr uint32_t WIDE_LINE_WORDS = 32;
constexpr size_t WIDE_PACKED_MAX_BYTES = (size_t)32 << 20;
constexpr uint32_t WIDE_PRIM_FLOATS = 16;
constexpr uint32_t WIDE_MAX_DEPTH = 60;        // stack slots per lane the walk may need (one per level)
constexpr uint32_t WIDE_XF_RAY_SLOTS = 6;      // behind a lane's stack: the caller's ray while the lane walks a transformed mesh
struct WideScene {
    std::vector<uint32_t> nodes;
    std::vector<float> prims;
    uint32_t depth = 0;                        // levels of nodes
    uint32_t node_words = WIDE_NODE_WORDS;     // stride of the node array: WIDE_NODE_WORDS or WIDE_LINE_WORDS
    size_t n_nodes() const { return nodes.size() / node_words; }
    size_t n_prims() const { return prims.size() / WIDE_PRIM_FLOATS; }
};
// What a ray that enters the scene's bounds is expected to cost on a flattened tree, in box tests
// (the surface-area measure: an entry is tested when the ray passes the box above it, which for
// uniformly distributed rays happens with probability area(box) / area(root); a triangle test counts
// 1.6 box tests, a sphere test 1.3).  What RT_BUILD_AUTO compares its two trees by.  0: not known.
double expected_cost_per_ray(const float* entries, size_t n_entries);

// collapses the pre-order entry array into a wide tree; returns an empty string on success.
// `xforms`: FlatScene::xforms (32 floats per transformed mesh) when the array holds ENTER / LEAVE markers
std::string build_wide_scene(const float* entries, size_t n_entries, WideScene& out, const float* xforms = nullptr,
                             size_t n_xforms = 0);

// The same tree built on the GPU from the device's entry array (rt_wide_device.hip): Morton-sorted
// primitives, children = runs of equal Morton digit.  The arrays are device memory the caller owns
// (hipFree) on success.  Runs on `stream` (a hipStream_t) of the current device and waits for it.
struct WideDeviceTree {
    void* nodes = nullptr;
    void* prims = nullptr;
    size_t nodes_bytes = 0, prims_bytes = 0;
    uint32_t n_slots = 0, n_prims = 0, depth = 0, node_words = WIDE_NODE_WORDS;
};
int build_wide_on_device(void* stream, const void* d_entries, uint32_t n_entries, WideDeviceTree& out, std::string& error);

// ---------------------------------------------------------------------------------------------
// Device side
// ---------------------------------------------------------------------------------------------
struct DevScene {
    const void* entries;      // float4[4 * n_entries] (+ the TAG_END entry): the wave sweep's form
    const void* units;        // float4[n_units + UNITS_END]: pre-order units (ordered walk; source of the octant copies)
    const void* walk;         // float4[n_walk + UNITS_END]: linked units in locality order (reference-order per-lane walks)
    uint32_t n_walk;
    uint32_t walk_top_units;  // the first units of `walk` hold the top levels of the trees, breadth first
    const void* prim_geo;     // float4[n_prims], by rank
    const uint32_t* prim_meta; // by rank
    const uint32_t* prim_order;   // rank -> primitive id
    uint32_t fuse;            // the fused kernel shades a ray right after its closest hit instead of queueing it for a
                              // shade pass: 1 (Lambertian is the scene's only scattering kind), 2 (every kind), 0: off
    uint32_t tie;             // 1: the walk does not visit in the reference's order (another tree, or nearer child
                              // first): exact ties in t are resolved by primitive key
    const void* materials;    // float4[2 * n_mats]
    uint32_t n_entries;
    uint32_t n_units;
    uint32_t n_prims;
    uint32_t n_mats;
    uint32_t kinds_present;
    uint32_t sweep;           // 1: wave-synchronous sweep of the entry array, 0: per-lane walk
    uint32_t has_spheres;     // the scene holds at least one sphere
    uint32_t prim_wait;       // per-lane walk: lanes that must stand at primitives before their routine runs
    uint32_t ordered;         // 1: near-child-first walk with a per-lane stack (not the reference's visiting order)
    uint32_t stack_depth;     // ordered walk: stack slots per lane (the tree's depth)
    uint32_t lds_walk;        // 1: the unit array is small enough to be walked out of LDS
    uint32_t refill;          // per-lane walk in the fused kernel: idle lanes that trigger a refill from the segment (0: off)
    const void* xforms;       // float4[8 * n_xforms]: transform and inverse of every transformed mesh
    uint32_t n_xforms;        // > 0: the scene is walked by the form that follows ENTER / LEAVE markers
    const void* units_oct;    // float4[8 * (n_units + UNITS_END)]: the unit array once per direction octant (octant walk)
    uint32_t octant;          // 1: every ray walks the copy of its direction octant (near child first, no stack)
    uint32_t split_top;       // > 0: the fused kernel keeps the first split_top units of `walk` in LDS (split walk)
    uint32_t deep_wait;       // split walk: lanes that must wait for a unit from memory before the wave fetches for them
    const void* wide_nodes;   // uint4[5 * node slots]: the wide tree (RT_TRAVERSAL_WIDE)
    const void* wide_prims;   // float4[4 * primitives of the wide tree]
    uint32_t wide;            // 1: stack walk of the wide tree, nearer children first (not the reference's visiting order)
    uint32_t wide_stack;      // stack slots per lane (the wide tree's depth)
    uint32_t wide_stride;     // uint4 per node slot: 5 (packed) or 8 (one node per 128-byte line)
};

// scenes with at most this many entries are traced with the wave-synchronous sweep by default
constexpr uint32_t SWEEP_MAX_ENTRIES = 192;
// unit arrays up to this size are copied to LDS by every 8-wave workgroup of the per-lane walk
// (three such workgroups fit the 160 KB of a CU: 6 waves per SIMD)
constexpr uint32_t LDS_WALK_MAX_BYTES = 52u * 1024u;
// the split walk keeps at most this much of the top of the trees in LDS (four 8-wave workgroups per CU)
constexpr uint32_t SPLIT_TOP_MAX_BYTES = 36u * 1024u;

constexpr uint32_t MAX_DEPTH = 64;

// Totals in device memory, read back once per render call.
struct Totals {
    unsigned long long rays, box, tri, sph;
};

// One wavefront batch: pixels [pix_base, pix_base + n_pix_slots) of the rectangle in 8x8-tile
// order, samples [sample_base, sample_base + n_samples).  Path slot = sample * n_pix_slots + pixel.
//
// The batch's path slots are cut into n_seg segments of seg_len slots (a multiple of 64) and
// every segment is owned by ONE wave for the whole batch: that wave generates the segment's
// camera rays, traces them, shades them and compacts the survivors back into the same
// segment of the other ray buffer.  Compaction therefore needs only the wave's own ballot and
// lane prefix -- no atomics, no barriers, no cross-wave traffic -- and the order of rays in a
// segment is deterministic.  segcnt[d * n_seg + s] = live rays of segment s entering depth d.
struct BatchGeom {
    uint32_t width, height;       // full frame
    uint32_t x0, y0, x1, y1;      // rectangle
    uint32_t tiles_x;             // 8x8 tiles per row of the rectangle
    uint32_t tile_base;           // first tile of this batch
    // tile of pixel slot q, with i = q / 64: tile_base + i + (i >> tile_run_shift) * tile_gap -- runs of
    // consecutive tiles with gaps between them (a frame's batches: no gaps)
    uint32_t tile_run_shift = 0, tile_gap = 0;
    uint32_t n_pix_slots;         // 64 * tiles in this batch
    uint32_t sample_base, n_samples;
    uint32_t n_slots;             // n_pix_slots * n_samples
    uint32_t n_seg, seg_len;
    // interleaved row bands: the batch's tile rows are the owned ones, in order (band_count 1: all)
    uint32_t band_tile_rows, band_count, band_index;
};

struct RayBuffers {
    void* A;   // float4: origin.xyz, path slot
    void* B;   // float4: dir.xyz,    rng key low
    void* C;   // float4: throughput, rng key high
};

struct RenderBuffers {
    RayBuffers cur, next;
    void* hits;            // uint2 per ray: t bits, prim id | PRIM_BACKFACE
    uint32_t* queue[3];    // ray indices per scattering material kind, segment-local
    void* Lbuf;            // float4 per path slot: the path's radiance
    uint32_t* segcnt;      // [(MAX_DEPTH + 1) * n_seg]
    uint32_t* qcnt;        // [3 * n_seg] queue lengths of the current depth
    Totals* totals;
    size_t capacity;       // paths
    unsigned long long* clock_stamps = nullptr;   // profiled renders: four words for this launch's clock stamps (else null)
};

struct LaunchCfg {
    void* stream;          // hipStream_t
    uint32_t grid_cap;     // upper bound on blocks per launch (grid-stride kernels)
    uint32_t wave_target;  // segments (waves) a full-size batch is cut into
};

// kernels' host-side launchers (rt_device.hip)
// closest hit of a ray batch on `grid` workgroups; with `count` every workgroup writes its visit
// counts to its record of d_partials (>= grid records) and their sum goes to d_out3[0..2]
// (box, triangle, sphere tests; 64-bit)
uint32_t trace_api_grid(const LaunchCfg&, size_t n);
int launch_trace_api(const LaunchCfg&, const DevScene&, const void* d_rays, size_t n, float t_min,
                     float t_max, void* d_hits, Totals* d_partials, bool count, uint32_t grid, void* d_out3);
// rt_trace_device_static(): workgroup records of the library's static device array, and that
// array's address on `device` (which must be the current device)
constexpr uint32_t TRACE_STATIC_MAX_BLOCKS = 16384;
constexpr uint32_t TRACE_STATIC_BLOCK_RAYS = 256;
int static_scratch_address(int device, void** out);
int launch_raygen(const LaunchCfg&, const rt_camera&, const BatchGeom&, uint64_t seed, const RenderBuffers&);
int launch_trace(const LaunchCfg&, const DevScene&, const BatchGeom&, const RenderBuffers&, uint32_t depth,
                 bool last_depth, float t_min, float sky_intensity, bool count);
int launch_shade(const LaunchCfg&, const DevScene&, const BatchGeom&, const RenderBuffers&, uint32_t depth);
// raygen + (trace + shade) x max_bounces for every segment in one kernel
int launch_path(const LaunchCfg&, const DevScene&, const rt_camera&, const BatchGeom&, uint64_t seed,
                const RenderBuffers&, uint32_t max_bounces, float t_min, float sky_intensity);
int launch_accumulate(const LaunchCfg&, const BatchGeom&, const RenderBuffers&, void* d_accum,
                      uint32_t max_bounces);
int launch_tonemap(const LaunchCfg&, const void* d_accum, uint32_t spp, void* d_out, size_t npix);
int launch_add_frame(const LaunchCfg&, void* d_dst, const void* d_src, size_t n_floats);   // dst += src

}  // namespace rtamd
