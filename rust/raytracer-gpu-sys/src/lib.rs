//! UNVERIFIED: written without a Rust toolchain; the layouts below restate include/rt_abi.h and
//! include/rt_host.h field by field (tests/test_host_layer.py checks the same sizes from Python:
//! rt_material 24, rt_sphere 20, rt_triangle 52, rt_ray 24, rt_hit 40, rt_camera 76,
//! rt_render_params 80, rt_stats 88 bytes).
#![allow(non_camel_case_types)]
use std::os::raw::{c_char, c_int, c_void};

pub const RT_OK: c_int = 0;
pub const RT_MAT_LAMBERT: u32 = 0;
pub const RT_MAT_METAL: u32 = 1;
pub const RT_MAT_DIELECTRIC: u32 = 2;
pub const RT_MAT_EMISSIVE: u32 = 3;
pub const RT_MAT_FLAG_NORMAL_SHADING: u32 = 1;
pub const RT_ROOT_BVH: c_int = 0;
pub const RT_ROOT_SINGLE: c_int = 1;
pub const RT_ROOT_LIST: c_int = 2;
pub const RT_TRAVERSAL_AUTO: c_int = 0;
pub const RT_TRAVERSAL_PER_LANE: c_int = 1;
pub const RT_TRAVERSAL_SWEEP: c_int = 2;
pub const RT_TRAVERSAL_ORDERED: c_int = 3;
pub const RT_TRAVERSAL_OCTANT: c_int = 4;
pub const RT_TRAVERSAL_WIDE: c_int = 5;
pub const RT_WIDE_BUILD_COLLAPSE: c_int = 0;
pub const RT_WIDE_BUILD_DEVICE: c_int = 1;
pub const RT_BUILD_REFERENCE: c_int = 0;
pub const RT_BUILD_SAH: c_int = 1;
pub const RT_BUILD_AUTO: c_int = 2;
pub const RT_TREE_REFERENCE: u32 = 0;
pub const RT_TREE_SAH: u32 = 1;
pub const RT_REDUCE_RCCL: c_int = 0;
pub const RT_REDUCE_ORDERED: c_int = 1;
pub const RT_STATS_MAX_DEVICES: usize = 16;
pub const RT_CACHE_OFF: c_int = 0;
pub const RT_CACHE_STORED: c_int = 1;
pub const RT_CACHE_LOADED: c_int = 2;
pub const RT_CACHE_STORE_FAILED: c_int = 3;
pub const RT_RENDER_PROFILE: u32 = 1;
pub const RT_RENDER_COUNT_VISITS: u32 = 2;
pub const RT_RENDER_STAGED: u32 = 4;
pub const RT_MISS: u32 = 0xFFFF_FFFF;
pub const RT_FORM_SWEEP: u32 = 0;
pub const RT_FORM_WALK: u32 = 1;
pub const RT_FORM_WALK_LDS: u32 = 2;
pub const RT_FORM_WALK_REFILL: u32 = 3;
pub const RT_FORM_WALK_LDS_REFILL: u32 = 4;
pub const RT_FORM_ORDERED: u32 = 5;
pub const RT_FORM_OCTANT: u32 = 6;
pub const RT_FORM_OCTANT_REFILL: u32 = 7;
pub const RT_FORM_WIDE: u32 = 8;
pub const RT_FORM_WIDE_REFILL: u32 = 9;
pub const RT_TRACE_STATIC_MAX_RAYS: u32 = 4194304;

#[repr(C)] #[derive(Clone, Copy, Default)]
pub struct rt_material { pub kind: u32, pub rgb: [f32; 3], pub param: f32, pub flags: u32 }
#[repr(C)] #[derive(Clone, Copy, Default)]
pub struct rt_sphere { pub c: [f32; 3], pub r: f32, pub mat: u32 }
#[repr(C)] #[derive(Clone, Copy, Default)]
pub struct rt_triangle { pub p: [[f32; 3]; 3], pub n0: [f32; 3], pub mat: u32 }
#[repr(C)] #[derive(Clone, Copy, Default)]
pub struct rt_ray { pub o: [f32; 3], pub d: [f32; 3] }
#[repr(C)] #[derive(Clone, Copy, Default)]
pub struct rt_hit { pub t: f32, pub prim_id: u32, pub p: [f32; 3], pub n: [f32; 3], pub front_face: u32, pub mat: u32 }
#[repr(C)] #[derive(Clone, Copy, Default)]
pub struct rt_camera {
    pub origin: [f32; 3], pub horizontal: [f32; 3], pub vertical: [f32; 3], pub lower_left: [f32; 3],
    pub u: [f32; 3], pub v: [f32; 3], pub lens_radius: f32,
}
#[repr(C)] #[derive(Clone, Copy, Default)]
pub struct rt_render_params {
    pub width: u32, pub height: u32, pub x0: u32, pub y0: u32, pub x1: u32, pub y1: u32,
    pub sample_begin: u32, pub sample_end: u32, pub max_bounces: u32, pub t_min: f32, pub sky_intensity: f32,
    pub seed: u64, pub flags: u32, pub max_paths_in_flight: u32,
    pub band_rows: u32, pub band_count: u32, pub band_index: u32, pub reserved: u32,
}
#[repr(C)] #[derive(Clone, Copy, Default)]
pub struct rt_stats {
    pub rays_traced: u64, pub paths: u64, pub box_tests: u64, pub tri_tests: u64, pub sph_tests: u64,
    pub trace_launches: u64, pub ms_render: f64, pub ms_trace: f64, pub ms_shade: f64,
    pub ms_reduce: f64, pub n_devices: u32, pub kernel_form: u32,
    pub tree: u32, pub rccl_ranks: u32, pub ms_kernel_span: f64, pub ms_render_min: f64,
    pub slowest_device: u32, pub reduce_mode: u32, pub shader_clock_ghz: f64,
    pub device_ms_render: [f32; 16], pub device_form: [u8; 16], pub device_tree: [u8; 16],
}
pub enum rt_context {}
pub enum rt_scene {}
pub enum rt_world {}

/// RenderEvent::BlockComplete(x, y, w, h, Vec<u32>) as a C callback; non-zero return stops the render.
pub type rt_block_fn = Option<unsafe extern "C" fn(user: *mut c_void, x: u32, y: u32, w: u32, h: u32,
                                                   block_0rgb: *const u32, samples_done: u32, samples_total: u32) -> c_int>;

extern "C" {
    pub fn rt_context_create(device_id: c_int, out: *mut *mut rt_context) -> c_int;
    pub fn rt_context_create_multi(n_dev: c_int, dev_ids: *const c_int, out: *mut *mut rt_context) -> c_int;
    pub fn rt_context_num_devices(ctx: *const rt_context) -> c_int;
    pub fn rt_context_set_reduce(ctx: *mut rt_context, mode: c_int) -> c_int;
    pub fn rt_rccl_version() -> c_int;
    pub fn rt_scene_expected_cost(scene: *const rt_scene) -> f64;
    pub fn rt_context_destroy(ctx: *mut rt_context);
    pub fn rt_last_error() -> *const c_char;
    pub fn rt_context_stream(ctx: *mut rt_context) -> *mut c_void;
    pub fn rt_context_synchronize(ctx: *mut rt_context) -> c_int;

    pub fn rt_scene_create(ctx: *mut rt_context, out: *mut *mut rt_scene) -> c_int;
    pub fn rt_scene_destroy(scene: *mut rt_scene);
    pub fn rt_scene_set_materials(scene: *mut rt_scene, mats: *const rt_material, n: usize) -> c_int;
    pub fn rt_scene_add_spheres(scene: *mut rt_scene, spheres: *const rt_sphere, n: usize) -> c_int;
    pub fn rt_scene_add_mesh(scene: *mut rt_scene, tris: *const rt_triangle, n: usize, xform: *const f32) -> c_int;
    pub fn rt_scene_begin_group(scene: *mut rt_scene, kind: c_int, axis: c_int) -> c_int;
    pub fn rt_scene_end_group(scene: *mut rt_scene) -> c_int;
    pub fn rt_scene_set_root(scene: *mut rt_scene, root_kind: c_int) -> c_int;
    pub fn rt_scene_set_traversal(scene: *mut rt_scene, mode: c_int) -> c_int;
    pub fn rt_scene_set_wide_build(scene: *mut rt_scene, where_: c_int) -> c_int;
    pub fn rt_scene_set_build(scene: *mut rt_scene, mode: c_int) -> c_int;
    pub fn rt_set_default_build(mode: c_int) -> c_int;
    pub fn rt_set_default_cache_dir(dir: *const c_char) -> c_int;
    pub fn rt_scene_set_cache_dir(scene: *mut rt_scene, dir: *const c_char) -> c_int;
    pub fn rt_scene_cache_state(scene: *const rt_scene) -> c_int;
    pub fn rt_scene_commit(scene: *mut rt_scene) -> c_int;
    pub fn rt_scene_num_prims(scene: *const rt_scene) -> usize;
    pub fn rt_scene_num_entries(scene: *const rt_scene) -> usize;
    pub fn rt_scene_get_entries(scene: *const rt_scene, out: *mut f32, n_floats: usize) -> c_int;
    pub fn rt_scene_get_prim_order(scene: *const rt_scene, out: *mut u32, n: usize) -> c_int;
    pub fn rt_scene_num_units(scene: *const rt_scene) -> usize;
    pub fn rt_scene_get_units(scene: *const rt_scene, out: *mut f32, n_floats: usize) -> c_int;
    pub fn rt_scene_num_walk_units(scene: *const rt_scene) -> usize;
    pub fn rt_scene_get_walk_units(scene: *const rt_scene, out: *mut f32, n_floats: usize) -> c_int;
    pub fn rt_scene_wide_info(scene: *const rt_scene, n_nodes: *mut usize, n_prims: *mut usize, depth: *mut u32,
                              node_words: *mut u32) -> c_int;
    pub fn rt_scene_get_wide(scene: *const rt_scene, nodes: *mut u32, n_node_words: usize, prims: *mut f32,
                             n_prim_floats: usize) -> c_int;

    pub fn rt_trace(scene: *mut rt_scene, rays: *const rt_ray, n: usize, t_min: f32, t_max: f32, out: *mut rt_hit) -> c_int;
    pub fn rt_trace_device(scene: *mut rt_scene, d_rays: *const c_void, n: usize, t_min: f32, t_max: f32,
                           d_hits: *mut c_void, counts: *mut u64) -> c_int;
    pub fn rt_trace_device_static(scene: *mut rt_scene, d_rays: *const c_void, n: usize, t_min: f32, t_max: f32,
                                  d_hits: *mut c_void, d_counts: *mut c_void) -> c_int;
    pub fn rt_render(scene: *mut rt_scene, cam: *const rt_camera, params: *const rt_render_params,
                     accum_rgb: *mut f32, stats: *mut rt_stats) -> c_int;
    pub fn rt_render_device(scene: *mut rt_scene, cam: *const rt_camera, params: *const rt_render_params,
                            d_accum_rgb: *mut c_void, stats: *mut rt_stats) -> c_int;
    pub fn rt_partition(n_ranks: u32, rank: u32, params: *mut rt_render_params) -> c_int;
    pub fn rt_render_multi(scene: *mut rt_scene, cam: *const rt_camera, params: *const rt_render_params,
                           accum_rgb: *mut f32, stats: *mut rt_stats) -> c_int;
    pub fn rt_render_multi_device(scene: *mut rt_scene, cam: *const rt_camera, params: *const rt_render_params,
                                  d_accum_rgb: *mut c_void, stats: *mut rt_stats) -> c_int;
    pub fn rt_tonemap(ctx: *mut rt_context, accum_rgb: *const f32, spp: u32, out_0rgb: *mut u32, npix: usize) -> c_int;
    pub fn rt_tonemap_device(ctx: *mut rt_context, d_accum_rgb: *const c_void, spp: u32, d_out_0rgb: *mut c_void,
                             npix: usize) -> c_int;

    pub fn rt_camera_setup(look_from: *const f32, look_at: *const f32, vup: *const f32, vfov: f32, aperture: f32,
                           focus_dist: f32, aspect_ratio: f32, out: *mut rt_camera) -> c_int;
    pub fn rt_world_create_world1(ctx: *mut rt_context, out: *mut *mut rt_world) -> c_int;
    pub fn rt_world_create_world2(ctx: *mut rt_context, thing_obj: *const c_char, variant: c_int, out: *mut *mut rt_world) -> c_int;
    pub fn rt_world_create_cornell_box(ctx: *mut rt_context, obj_path: *const c_char, out: *mut *mut rt_world) -> c_int;
    pub fn rt_world_create_soup(ctx: *mut rt_context, n_tris: u64, grid: u32, seed: u32, out: *mut *mut rt_world) -> c_int;
    pub fn rt_world_destroy(world: *mut rt_world);
    pub fn rt_world_scene(world: *mut rt_world) -> *mut rt_scene;
    pub fn rt_world_sky_intensity(world: *const rt_world) -> f32;
    pub fn rt_world_camera(world: *const rt_world, aspect_ratio: f32, out: *mut rt_camera) -> c_int;
    pub fn rt_world_num_materials(world: *const rt_world) -> usize;
    pub fn rt_world_get_materials(world: *const rt_world, out: *mut rt_material) -> c_int;
    pub fn rt_world_num_objects(world: *const rt_world) -> usize;
    pub fn rt_world_object_info(world: *const rt_world, i: usize, count: *mut usize) -> c_int;
    pub fn rt_world_get_sphere(world: *const rt_world, i: usize, out: *mut rt_sphere) -> c_int;
    pub fn rt_world_get_mesh(world: *const rt_world, i: usize, out: *mut rt_triangle) -> c_int;
    pub fn rt_world_get_mesh_xform(world: *const rt_world, i: usize, out: *mut f32) -> c_int;
    pub fn rt_world_root_kind(world: *const rt_world) -> c_int;
    pub fn rt_world_render(world: *mut rt_world, width: u32, height: u32, spp: u32, max_bounces: u32, seed: u64,
                           accum_rgb: *mut f32, stats: *mut rt_stats) -> c_int;
    pub fn rt_world_render_progressive(world: *mut rt_world, width: u32, height: u32, spp: u32, max_bounces: u32,
                                       seed: u64, on_block: rt_block_fn, user: *mut c_void, accum_rgb: *mut f32,
                                       stats: *mut rt_stats) -> c_int;
    pub fn rt_progressive_schedule(spp: u32, counts: *mut u32, cap: usize) -> usize;
    pub fn rt_write_ppm(path: *const c_char, pixels_0rgb: *const u32, width: u32, height: u32) -> c_int;
    pub fn rt_obj_count(path: *const c_char, n_tris: *mut usize, n_models: *mut usize, n_faces: *mut usize) -> c_int;
}

/// What `render(width, height, &event_loop, &mut pool)` (main.rs:444) becomes: the block callback
/// forwards to the window exactly what the tile closure sent (main.rs:399-407).
///
/// ```ignore
/// unsafe extern "C" fn on_block(user: *mut c_void, x: u32, y: u32, w: u32, h: u32, px: *const u32, _done: u32, _total: u32) -> c_int {
///     let proxy = &*(user as *const EventLoopProxy<RenderEvent>);
///     let block = std::slice::from_raw_parts(px, (w * h) as usize).to_vec();
///     proxy.send_event(RenderEvent::BlockComplete(x, y, w, h, block)).is_err() as c_int
/// }
/// let mut ctx = std::ptr::null_mut();  rt_context_create(0, &mut ctx);
/// let mut world = std::ptr::null_mut();
/// rt_world_create_cornell_box(ctx, b"cornell_box.obj\0".as_ptr() as *const c_char, &mut world);
/// let proxy = event_loop.create_proxy();
/// rt_world_render_progressive(world, width, height, SAMPLES_PER_PIXEL, MAX_BOUNCES as u32, 0x9E3779B97F4A7C15,
///                             Some(on_block), &proxy as *const _ as *mut c_void, std::ptr::null_mut(), std::ptr::null_mut());
/// ```
pub const _DOC_ANCHOR: () = ();
