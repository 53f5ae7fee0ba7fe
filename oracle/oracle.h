/*
 * CPU oracle for the path tracer's per-pixel radiance loop -- TEST INFRASTRUCTURE ONLY.
 *
 * This is a plain C++ restatement of the algorithm in the Rust reference
 * (src/main.rs, src/shape.rs, src/material.rs, src/camera.rs, src/math.rs).  It exists
 * so that the GPU path can be checked against it (tests/), so that smoke() has a
 * checker, and so that bench.py has a host-CPU baseline to time.  Nothing in the product
 * library links, loads or calls it.
 *
 * Pinning status: the reference ships no tests, golden vectors or fixtures, and it cannot
 * be compiled here (no Rust toolchain), so everything that depends on the third-party
 * crates glam 0.8.7 (Vec3 arithmetic: dot association, normalize, min/max) and tobj 3.2.2
 * (OBJ triangulation) is "parity unpinned".  What IS pinned: the xorshift32 stream, the
 * random01 bit patterns, the create_world1 sphere table and the OBJ triangle counts, all
 * derived from the reference's definitions and checked in tests/test_oracle_kat.py; and, at
 * image level, the whole pipeline (OBJ ingest, camera, trees, materials, tonemap) by the one
 * rendered frame the reference ships, screenshot.png of its default job, which this oracle
 * reproduces block for block within Monte-Carlo noise (same test file).
 *
 * The C API below is what tests bind with ctypes.
 */
#ifndef RT_ORACLE_H
#define RT_ORACLE_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct oracle_scene oracle_scene;

/* material kinds, same numbering as the product ABI */
enum { ORACLE_LAMBERT = 0, ORACLE_METAL = 1, ORACLE_DIELECTRIC = 2, ORACLE_EMISSIVE = 3 };
/* what World.hittable is: a Bvh over the objects (main.rs:170,234), the single object itself
 * (main.rs:289: the Mesh is the root) or a HittableList (main.rs:169, commented out there) */
enum { ORACLE_ROOT_BVH = 0, ORACLE_ROOT_SINGLE = 1, ORACLE_ROOT_LIST = 2 };

typedef struct { uint32_t kind; float rgb[3]; float param; uint32_t flags; } oracle_material;
typedef struct { float c[3]; float r; uint32_t mat; } oracle_sphere;
typedef struct { float p[3][3]; float n0[3]; uint32_t mat; } oracle_triangle;
typedef struct { float o[3]; float d[3]; } oracle_ray;
typedef struct {
    float t; uint32_t prim_id; float p[3]; float n[3]; uint32_t front_face; uint32_t mat;
} oracle_hit;
typedef struct {
    float origin[3], horizontal[3], vertical[3], lower_left[3], u[3], v[3];
    float lens_radius;
} oracle_camera;
typedef struct {
    uint64_t rays;        /* closest-hit queries (main.rs:46) */
    uint64_t paths;       /* camera samples */
    uint64_t box_tests;   /* Aabb::hit calls (math.rs:149) */
    uint64_t tri_tests;   /* ray_triangle_intersection calls (shape.rs:411) */
    uint64_t sph_tests;   /* Sphere::hit calls (shape.rs:19) */
    uint64_t hits;        /* queries that returned true */
    double seconds;       /* wall time of the call */
} oracle_stats;

/* ---- generic scene builder (objects keep insertion order, as Vec<Box<dyn Hittable>>) ---- */
oracle_scene* oracle_scene_new(void);
void oracle_scene_free(oracle_scene*);
void oracle_scene_set_materials(oracle_scene*, const oracle_material*, size_t n);
void oracle_scene_add_spheres(oracle_scene*, const oracle_sphere*, size_t n);
/* one Mesh with the identity transform */
void oracle_scene_add_mesh(oracle_scene*, const oracle_triangle*, size_t n);
/* one Mesh with a transform (column-major 4x4, glam Mat4 order; NULL = identity): shape.rs:236-257 */
void oracle_scene_add_mesh_xf(oracle_scene*, const oracle_triangle*, size_t n, const float xform[16]);
/* Objects added between begin and end are the children of ONE container object -- a Bvh
 * (kind ORACLE_ROOT_BVH, built with Bvh::new(children, axis), shape.rs:94) or a HittableList
 * (ORACLE_ROOT_LIST) -- which itself is a child of the enclosing container; groups nest.
 * Primitive ids count leaf objects in the order they were added, whatever their nesting. */
void oracle_scene_begin_group(oracle_scene*, int kind, int axis);
void oracle_scene_end_group(oracle_scene*);
/* glam Mat4::inverse / transform_point3 (w = 1) / transform_vector3 (w = 0) as restated here */
void oracle_mat4_inverse(const float m[16], float out[16]);
void oracle_mat4_transform(const float m[16], const float v[3], float w, float out[3]);
void oracle_scene_set_root(oracle_scene*, int root_kind);
void oracle_scene_set_camera(oracle_scene*, const float look_from[3], const float look_at[3],
                             const float vup[3], float vfov, float aperture, float focus_dist);
void oracle_scene_set_sky(oracle_scene*, float sky_intensity);
/* builds Bvh / Mesh BVHs; returns 0 on success */
int oracle_scene_commit(oracle_scene*);

/* ---- the reference's scene constructors ---- */
oracle_scene* oracle_create_world1(void);                          /* main.rs:75-176  */
/* variant 0: as shipped (all "default" lambert); variant 1: Material->glass 1.5, None->lambert .73;
 * variant 2: as shipped plus the two translated planes main.rs:195-204 keeps in a comment */
oracle_scene* oracle_create_world2(const char* thing_obj, int variant);   /* main.rs:178-241 */
oracle_scene* oracle_create_cornell_box(const char* obj_path);    /* main.rs:243-317 */

/* introspection (used by KAT tests and to mirror a scene into the product ABI) */
size_t oracle_scene_num_objects(const oracle_scene*);
size_t oracle_scene_num_prims(const oracle_scene*);
size_t oracle_scene_num_materials(const oracle_scene*);
size_t oracle_scene_num_spheres(const oracle_scene*);
size_t oracle_scene_num_triangles(const oracle_scene*);
void oracle_scene_get_materials(const oracle_scene*, oracle_material* out);
/* object i: returns 0 for sphere, 1 for mesh; *count = 1 or triangle count */
int oracle_scene_object_info(const oracle_scene*, size_t i, size_t* count);
void oracle_scene_get_sphere(const oracle_scene*, size_t obj, oracle_sphere* out);
void oracle_scene_get_mesh(const oracle_scene*, size_t obj, oracle_triangle* out);
/* the transform of object obj (identity for spheres); returns 1 when it is not the identity */
int oracle_scene_get_xform(const oracle_scene*, size_t obj, float out[16]);
float oracle_scene_sky(const oracle_scene*);
int oracle_scene_root_kind(const oracle_scene*);
/* Camera::setup(aspect) (camera.rs:55-78) then the post-setup fields */
void oracle_scene_camera(const oracle_scene*, float aspect, oracle_camera* out);
/* raw camera parameters: from[3], at[3], up[3], vfov, aperture, focus */
void oracle_scene_camera_params(const oracle_scene*, float out[12]);

/* ---- hot path ---- */
/* world.hittable.hit(ray, t_min, t_max) for a batch (shape.rs:7); prim_id = 0xFFFFFFFF on miss */
void oracle_trace(const oracle_scene*, const oracle_ray* rays, size_t n, float t_min, float t_max,
                  oracle_hit* out, oracle_stats* stats, int nthreads);
/* The tile-per-task renderer of main.rs:319-416 (64x64 tiles, RandomSeries(j+1+seed_offset),
 * progressive batches 10,20,40,...).  accum_rgb (W*H*3, may be NULL) gets the linear per-pixel
 * SUM of ray_color over the samples; image (W*H, may be NULL) gets the 0x00RRGGBB tonemap. */
void oracle_render(const oracle_scene*, uint32_t width, uint32_t height, uint32_t spp,
                   uint32_t max_bounces, uint32_t block_size, uint32_t seed_offset, int nthreads,
                   float* accum_rgb, uint32_t* image, oracle_stats* stats);
/* record up to cap rays the renderer would trace (primary and bounce) for parity batches */
size_t oracle_record_rays(const oracle_scene*, uint32_t width, uint32_t height, uint32_t spp,
                          uint32_t max_bounces, uint32_t seed_offset, oracle_ray* out, size_t cap,
                          uint32_t stride);

/* ---- small pieces exposed for known-answer tests ---- */
void oracle_xorshift32(uint32_t seed, uint32_t* out, size_t n);          /* math.rs:13-20 */
void oracle_random01(uint32_t seed, float* out, size_t n);               /* math.rs:22-23 */
float oracle_random01_of(uint32_t x);                                    /* the conversion alone */
uint32_t oracle_rgb_to_u32(float r, float g, float b);                   /* math.rs:112-116 */
void oracle_tonemap(const float* accum_rgb, uint32_t spp, uint32_t* out, size_t npix); /* main.rs:369-373 */
int oracle_aabb_hit(const float bmin[3], const float bmax[3], const float o[3], const float d[3],
                    float t_min, float t_max);                           /* math.rs:149-165 */
/* returns hit flag; out = {backface, t, u, v} */
int oracle_ray_triangle(const float o[3], const float d[3], const float v0[3], const float v1[3],
                        const float v2[3], float out[4]);               /* shape.rs:411-450 */
/* OBJ ingest (shape.rs:347-406 + tobj semantics): returns triangle count, fills per-model counts */
size_t oracle_obj_count(const char* path, size_t* n_models, size_t* n_faces);
/* scatter one ray with the given material; returns scatter flag. out = atten[3], o[3], d[3] */
int oracle_scatter(const oracle_material* m, const float ray_o[3], const float ray_d[3],
                   const float p[3], const float n[3], int front_face, uint32_t* rng_state,
                   float out[9]);
uint32_t oracle_rng_state_after_world1(void);

#ifdef __cplusplus
}
#endif
#endif
