/*
 * rt_host.h -- C entry points of the host scene layer that sits above rt_abi.h.
 *
 * The host layer (raytracer_amd/csrc/host/) is the C++ mirror of the reference's scene API:
 * Camera (src/camera.rs), Sphere / Mesh / Bvh / HittableList (src/shape.rs), the four
 * materials (src/material.rs), World and the scene constructors (src/main.rs:69-317).  A
 * rt_world is the reference's `World { camera, hittable, sky_intensity }` (main.rs:69-73) with
 * the hittable already committed as a rt_scene on the GPU.
 */
#ifndef RT_HOST_H
#define RT_HOST_H

#include "rt_abi.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rt_world rt_world;

/* create_world1() (main.rs:75-176): 6 fixed + 500 random spheres + emitter, Bvh root */
int rt_world_create_world1(rt_context* ctx, rt_world** out);
/* create_world2() (main.rs:178-241): thing.obj + emissive sphere, Bvh root.
 * variant 0: the shipped material map ({"default"}); variant 1: the "glass + diffuse" map
 * {"Material": dielectric(1,1,1; ior 1.5), "None": lambert(.73)}; variant 2: variant 0 plus the two
 * planes the reference keeps in a comment (Mesh::plane(2, 2, material1,
 * Mat4::from_translation((0, 0, -0.5) and (0, 0, 1))), main.rs:195-204) */
int rt_world_create_world2(rt_context* ctx, const char* thing_obj, int variant, rt_world** out);
/* create_cornell_box(filename) (main.rs:243-317): the Mesh is the root */
int rt_world_create_cornell_box(rt_context* ctx, const char* obj_path, rt_world** out);
/* Synthetic triangle soup + sphere field (the bandwidth stress scene): n_tris triangles with
 * centres uniform in [-100,100]^2 x [0,100] and vertices = centre + U(-0.15,0.15)^3, material
 * = index mod 4 over {lambert .73, metal(.8, .3), dielectric 1.5, lambert(.65,.05,.05)}; a
 * grid x grid field of r=0.5 spheres at z=-1 (pitch 3) and one emissive r=30 sphere at
 * (0,0,300); all randoms from xorshift32 (math.rs:13-20) seeded with `seed`; Bvh root. */
int rt_world_create_soup(rt_context* ctx, uint64_t n_tris, uint32_t grid, uint32_t seed, rt_world** out);
void rt_world_destroy(rt_world* world);

rt_scene* rt_world_scene(rt_world* world);
float rt_world_sky_intensity(const rt_world* world);
/* Camera::setup(aspect_ratio) (camera.rs:55-78) */
int rt_world_camera(const rt_world* world, float aspect_ratio, rt_camera* out);
/* the description the world was built from, for mirroring it elsewhere (tests do) */
size_t rt_world_num_materials(const rt_world* world);
int rt_world_get_materials(const rt_world* world, rt_material* out);
size_t rt_world_num_objects(const rt_world* world);
/* returns 0 for a sphere, 1 for a mesh; *count = 1 or the triangle count */
int rt_world_object_info(const rt_world* world, size_t i, size_t* count);
int rt_world_get_sphere(const rt_world* world, size_t i, rt_sphere* out);
int rt_world_get_mesh(const rt_world* world, size_t i, rt_triangle* out);
/* Mesh.transform of object i, column-major; returns 1 when it is not the identity, 0 when it is
 * (or the object is a sphere), negative on error */
int rt_world_get_mesh_xform(const rt_world* world, size_t i, float out[16]);
int rt_world_root_kind(const rt_world* world);

/* Camera::new + Camera::setup without a world */
int rt_camera_setup(const float look_from[3], const float look_at[3], const float vup[3], float vfov,
                    float aperture, float focus_dist, float aspect_ratio, rt_camera* out);

/* render() (main.rs:319-416) as one call: camera.setup(width/height), all pixels, samples
 * [0, spp), accumulating linear sums into accum_rgb (host, width*height*3, zeroed by the caller) */
int rt_world_render(rt_world* world, uint32_t width, uint32_t height, uint32_t spp, uint32_t max_bounces,
                    uint64_t seed, float* accum_rgb, rt_stats* stats);

/* The progressive schedule of the tile job (main.rs:355-356, 409-411): cumulative sample counts
 * min(10, spp), then doubling, capped at spp.  Writes up to `cap` of them and returns how many
 * steps there are. */
size_t rt_progressive_schedule(uint32_t spp, uint32_t* counts, size_t cap);

/* One finished or in-progress block, as RenderEvent::BlockComplete(x, y, w, h, Vec<u32>) delivers
 * it (main.rs:26-29, 399-407): 0x00RRGGBB pixels, row-major in the block.  Return non-zero to stop
 * the render after the current step. */
typedef int (*rt_block_fn)(void* user, uint32_t x, uint32_t y, uint32_t w, uint32_t h, const uint32_t* block_0rgb,
                           uint32_t samples_done, uint32_t samples_total);

/* render() with its live updates: the frame is rendered in the schedule's steps; after each, every
 * 64 x 64 block (main.rs:22; edge blocks clipped, main.rs:337-346) is tonemapped with the samples
 * so far (main.rs:369-373), gets the red corner marks of an unfinished block while more samples
 * are to come (main.rs:377-396), and is handed to `on_block` in the reference's j-then-i order.
 * accum_rgb (host, width*height*3, may be NULL) receives the final linear sums. */
int rt_world_render_progressive(rt_world* world, uint32_t width, uint32_t height, uint32_t spp,
                                uint32_t max_bounces, uint64_t seed, rt_block_fn on_block, void* user,
                                float* accum_rgb, rt_stats* stats);

/* binary PPM (P6) of a 0x00RRGGBB frame */
int rt_write_ppm(const char* path, const uint32_t* pixels_0rgb, uint32_t width, uint32_t height);

/* OBJ ingest (Mesh::from_file, shape.rs:347-406): number of triangles / models / faces */
int rt_obj_count(const char* path, size_t* n_tris, size_t* n_models, size_t* n_faces);

#ifdef __cplusplus
}
#endif
#endif
