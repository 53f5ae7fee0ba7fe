"""raytracer_amd -- ctypes binding of the MI355X wavefront path tracer's C ABI (include/rt_abi.h).

The shared library librt_amd.so (HIP kernels for gfx950 + host scene builder) is the product;
this module only marshals numpy arrays and device pointers across the C ABI.  There is no CPU
fallback: if the library is missing the import fails, and every call that needs a GPU raises
RtError when the HIP runtime reports none.
"""
import atexit
import ctypes as C
import os
import weakref

import numpy as np

# PyTorch (device memory, streams, torch.distributed) ships its own copy of the HIP runtime
# under the same soname.  Two HIP/HSA runtimes in one process do not work ("No HIP GPUs are
# available"), so torch's copy must be the one that is resident before librt_amd.so is loaded.
try:
    import torch  # noqa: F401
except ImportError:  # a torch-free process uses the system runtime the library was linked to
    torch = None

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RT_AMD_LIB") or os.path.join(_HERE, "librt_amd.so")   # override: A/B runs of kernel builds
ASSET_DIR = os.path.join(_HERE, "assets")

MAT_LAMBERT, MAT_METAL, MAT_DIELECTRIC, MAT_EMISSIVE = 0, 1, 2, 3
MAT_FLAG_NORMAL_SHADING = 1
ROOT_BVH, ROOT_SINGLE, ROOT_LIST = 0, 1, 2
TRAVERSAL_AUTO, TRAVERSAL_PER_LANE, TRAVERSAL_SWEEP, TRAVERSAL_ORDERED, TRAVERSAL_OCTANT, TRAVERSAL_WIDE = 0, 1, 2, 3, 4, 5
BUILD_REFERENCE, BUILD_SAH, BUILD_AUTO = 0, 1, 2
TREE_NAMES = ("reference", "sah")
REDUCE_RCCL, REDUCE_ORDERED = 0, 1
WIDE_BUILD_COLLAPSE, WIDE_BUILD_DEVICE = 0, 1
CACHE_OFF, CACHE_STORED, CACHE_LOADED, CACHE_STORE_FAILED = 0, 1, 2, 3
MISS = 0xFFFFFFFF
RENDER_PROFILE, RENDER_COUNT_VISITS, RENDER_STAGED = 1, 2, 4
FORM_NAMES = ("sweep", "walk", "walk_lds", "walk_refill", "walk_lds_refill", "ordered", "octant", "octant_refill", "wide", "wide_refill")
TRACE_STATIC_MAX_RAYS = 4194304
DEFAULT_SEED = 0x9E3779B97F4A7C15

material_dt = np.dtype([("kind", "<u4"), ("rgb", "<f4", 3), ("param", "<f4"), ("flags", "<u4")])
sphere_dt = np.dtype([("c", "<f4", 3), ("r", "<f4"), ("mat", "<u4")])
triangle_dt = np.dtype([("p", "<f4", (3, 3)), ("n0", "<f4", 3), ("mat", "<u4")])
ray_dt = np.dtype([("o", "<f4", 3), ("d", "<f4", 3)])
hit_dt = np.dtype([("t", "<f4"), ("prim_id", "<u4"), ("p", "<f4", 3), ("n", "<f4", 3),
                   ("front_face", "<u4"), ("mat", "<u4")])
camera_dt = np.dtype([("origin", "<f4", 3), ("horizontal", "<f4", 3), ("vertical", "<f4", 3),
                      ("lower_left", "<f4", 3), ("u", "<f4", 3), ("v", "<f4", 3),
                      ("lens_radius", "<f4")])


class RenderParams(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32),
                ("x0", C.c_uint32), ("y0", C.c_uint32), ("x1", C.c_uint32), ("y1", C.c_uint32),
                ("sample_begin", C.c_uint32), ("sample_end", C.c_uint32),
                ("max_bounces", C.c_uint32), ("t_min", C.c_float), ("sky_intensity", C.c_float),
                ("seed", C.c_uint64), ("flags", C.c_uint32), ("max_paths_in_flight", C.c_uint32),
                ("band_rows", C.c_uint32), ("band_count", C.c_uint32), ("band_index", C.c_uint32),
                ("reserved", C.c_uint32)]


class Stats(C.Structure):
    _fields_ = [("rays_traced", C.c_uint64), ("paths", C.c_uint64), ("box_tests", C.c_uint64),
                ("tri_tests", C.c_uint64), ("sph_tests", C.c_uint64), ("trace_launches", C.c_uint64),
                ("ms_render", C.c_double), ("ms_trace", C.c_double), ("ms_shade", C.c_double),
                ("ms_reduce", C.c_double), ("n_devices", C.c_uint32), ("kernel_form", C.c_uint32),
                ("tree", C.c_uint32), ("rccl_ranks", C.c_uint32), ("ms_kernel_span", C.c_double),
                ("ms_render_min", C.c_double), ("slowest_device", C.c_uint32), ("reduce_mode", C.c_uint32),
                ("shader_clock_ghz", C.c_double), ("device_ms_render", C.c_float * 16),
                ("device_form", C.c_uint8 * 16), ("device_tree", C.c_uint8 * 16)]

    def as_dict(self):
        return {k: (list(getattr(self, k)) if k.startswith("device_") else getattr(self, k)) for k, _ in self._fields_}


class RtError(RuntimeError):
    pass


# rt_block_fn: (user, x, y, w, h, block_0rgb, samples_done, samples_total) -> int
BLOCK_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                       C.POINTER(C.c_uint32), C.c_uint32, C.c_uint32)


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C raytracer_amd/csrc`.  There is no fallback path.")
    L = C.CDLL(LIB_PATH)
    vp, sz, u32, f32, i32 = C.c_void_p, C.c_size_t, C.c_uint32, C.c_float, C.c_int
    pp = C.POINTER(C.c_void_p)
    sig = {
        "rt_context_create": (i32, [i32, pp]),
        "rt_context_create_multi": (i32, [i32, C.POINTER(i32), pp]),
        "rt_context_num_devices": (i32, [vp]),
        "rt_context_set_reduce": (i32, [vp, i32]),
        "rt_rccl_version": (i32, []),
        "rt_scene_expected_cost": (C.c_double, [vp]),
        "rt_context_destroy": (None, [vp]),
        "rt_last_error": (C.c_char_p, []),
        "rt_context_stream": (vp, [vp]),
        "rt_context_synchronize": (i32, [vp]),
        "rt_scene_create": (i32, [vp, pp]),
        "rt_scene_destroy": (None, [vp]),
        "rt_scene_set_materials": (i32, [vp, vp, sz]),
        "rt_scene_add_spheres": (i32, [vp, vp, sz]),
        "rt_scene_add_mesh": (i32, [vp, vp, sz, vp]),
        "rt_scene_begin_group": (i32, [vp, i32, i32]),
        "rt_scene_end_group": (i32, [vp]),
        "rt_scene_set_root": (i32, [vp, i32]),
        "rt_scene_set_traversal": (i32, [vp, i32]),
        "rt_scene_set_wide_build": (i32, [vp, i32]),
        "rt_scene_set_build": (i32, [vp, i32]),
        "rt_set_default_build": (i32, [i32]),
        "rt_set_default_cache_dir": (i32, [C.c_char_p]),
        "rt_scene_set_cache_dir": (i32, [vp, C.c_char_p]),
        "rt_scene_cache_state": (i32, [vp]),
        "rt_scene_commit": (i32, [vp]),
        "rt_scene_num_prims": (sz, [vp]),
        "rt_scene_num_entries": (sz, [vp]),
        "rt_scene_get_entries": (i32, [vp, vp, sz]),
        "rt_scene_get_prim_order": (i32, [vp, vp, sz]),
        "rt_scene_num_units": (sz, [vp]),
        "rt_scene_get_units": (i32, [vp, vp, sz]),
        "rt_scene_num_walk_units": (sz, [vp]),
        "rt_scene_get_walk_units": (i32, [vp, vp, sz]),
        "rt_scene_wide_info": (i32, [vp, C.POINTER(sz), C.POINTER(sz), C.POINTER(u32), C.POINTER(u32)]),
        "rt_scene_get_wide": (i32, [vp, vp, sz, vp, sz]),
        "rt_trace": (i32, [vp, vp, sz, f32, f32, vp]),
        "rt_trace_device": (i32, [vp, vp, sz, f32, f32, vp, C.POINTER(C.c_uint64)]),
        "rt_trace_device_static": (i32, [vp, vp, sz, f32, f32, vp, vp]),
        "rt_render": (i32, [vp, vp, C.POINTER(RenderParams), vp, C.POINTER(Stats)]),
        "rt_render_device": (i32, [vp, vp, C.POINTER(RenderParams), vp, C.POINTER(Stats)]),
        "rt_partition": (i32, [u32, u32, C.POINTER(RenderParams)]),
        "rt_render_multi": (i32, [vp, vp, C.POINTER(RenderParams), vp, C.POINTER(Stats)]),
        "rt_render_multi_device": (i32, [vp, vp, C.POINTER(RenderParams), vp, C.POINTER(Stats)]),
        "rt_tonemap": (i32, [vp, vp, u32, vp, sz]),
        "rt_tonemap_device": (i32, [vp, vp, u32, vp, sz]),
        # host scene layer (include/rt_host.h)
        "rt_world_create_world1": (i32, [vp, pp]),
        "rt_world_create_world2": (i32, [vp, C.c_char_p, i32, pp]),
        "rt_world_create_cornell_box": (i32, [vp, C.c_char_p, pp]),
        "rt_world_create_soup": (i32, [vp, C.c_uint64, u32, u32, pp]),
        "rt_world_destroy": (None, [vp]),
        "rt_world_scene": (vp, [vp]),
        "rt_world_sky_intensity": (f32, [vp]),
        "rt_world_camera": (i32, [vp, f32, vp]),
        "rt_world_num_materials": (sz, [vp]),
        "rt_world_get_materials": (i32, [vp, vp]),
        "rt_world_num_objects": (sz, [vp]),
        "rt_world_object_info": (i32, [vp, sz, C.POINTER(sz)]),
        "rt_world_get_sphere": (i32, [vp, sz, vp]),
        "rt_world_get_mesh": (i32, [vp, sz, vp]),
        "rt_world_get_mesh_xform": (i32, [vp, sz, vp]),
        "rt_world_root_kind": (i32, [vp]),
        "rt_camera_setup": (i32, [vp, vp, vp, f32, f32, f32, f32, vp]),
        "rt_world_render": (i32, [vp, u32, u32, u32, u32, C.c_uint64, vp, C.POINTER(Stats)]),
        "rt_obj_count": (i32, [C.c_char_p, C.POINTER(sz), C.POINTER(sz), C.POINTER(sz)]),
        "rt_progressive_schedule": (sz, [u32, vp, sz]),
        "rt_world_render_progressive": (i32, [vp, u32, u32, u32, u32, C.c_uint64, BLOCK_FN, vp, vp, C.POINTER(Stats)]),
        "rt_write_ppm": (i32, [C.c_char_p, vp, u32, u32]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    return L


lib = _load()
ABI_SYMBOLS = [
    "rt_context_create", "rt_context_create_multi", "rt_context_num_devices", "rt_context_set_reduce", "rt_rccl_version", "rt_scene_expected_cost", "rt_context_destroy", "rt_last_error",
    "rt_context_stream", "rt_partition", "rt_render_multi", "rt_render_multi_device",
    "rt_context_synchronize", "rt_scene_create", "rt_scene_destroy", "rt_scene_set_materials",
    "rt_scene_add_spheres", "rt_scene_add_mesh", "rt_scene_begin_group", "rt_scene_end_group", "rt_scene_set_root", "rt_scene_set_build", "rt_set_default_build",
    "rt_scene_set_traversal", "rt_scene_set_wide_build", "rt_set_default_cache_dir", "rt_scene_set_cache_dir", "rt_scene_cache_state", "rt_scene_commit",
    "rt_scene_num_prims", "rt_scene_num_entries", "rt_scene_get_entries", "rt_scene_get_prim_order", "rt_scene_num_units",
    "rt_scene_get_units", "rt_scene_num_walk_units", "rt_scene_get_walk_units", "rt_scene_wide_info", "rt_scene_get_wide", "rt_trace",
    "rt_trace_device", "rt_trace_device_static", "rt_render", "rt_render_device", "rt_tonemap", "rt_tonemap_device",
]
HOST_SYMBOLS = [
    "rt_world_create_world1", "rt_world_create_world2", "rt_world_create_cornell_box",
    "rt_world_create_soup", "rt_world_destroy", "rt_world_scene", "rt_world_sky_intensity",
    "rt_world_camera", "rt_world_num_materials", "rt_world_get_materials", "rt_world_num_objects",
    "rt_world_object_info", "rt_world_get_sphere", "rt_world_get_mesh", "rt_world_get_mesh_xform", "rt_world_root_kind",
    "rt_camera_setup", "rt_world_render", "rt_obj_count", "rt_progressive_schedule",
    "rt_world_render_progressive", "rt_write_ppm",
]


# Handles still open when the interpreter exits are closed here, scenes and worlds before their
# contexts, while the HIP runtime is still up (object finalisation order at exit is arbitrary).
_live = weakref.WeakSet()


def _close_all():
    objs = list(_live)
    for rank in (0, 1):
        for o in objs:
            if isinstance(o, Context) == bool(rank):
                try:
                    o.close()
                except Exception:
                    pass


atexit.register(_close_all)


def _check(rc):
    if rc != 0:
        raise RtError(f"rt error {rc}: {lib.rt_last_error().decode(errors='replace')}")


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


class Context:
    """One GPU (rt_context), or with devices=[...] the GPUs of one node behind one context
    (rt_context_create_multi; devices[0] is the root that single-GPU calls run on)."""

    def __init__(self, device_id=0, devices=None):
        h = C.c_void_p()
        if devices is None:
            _check(lib.rt_context_create(device_id, C.byref(h)))
        else:
            ids = (C.c_int * len(devices))(*devices)
            _check(lib.rt_context_create_multi(len(devices), ids, C.byref(h)))
            device_id = devices[0]
        self.h = h
        self.device_id = device_id
        _live.add(self)

    @property
    def num_devices(self):
        return lib.rt_context_num_devices(self.h)

    def set_reduce(self, mode):
        """REDUCE_RCCL (one ncclReduce) or REDUCE_ORDERED (peer frames copied to the root, added in rank order)."""
        _check(lib.rt_context_set_reduce(self.h, mode))

    def close(self):
        if getattr(self, "h", None):
            lib.rt_context_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def synchronize(self):
        _check(lib.rt_context_synchronize(self.h))

    @property
    def stream(self):
        return lib.rt_context_stream(self.h)

    def tonemap(self, accum, spp):
        a = np.ascontiguousarray(accum, dtype=np.float32).reshape(-1, 3)
        out = np.zeros(len(a), dtype=np.uint32)
        _check(lib.rt_tonemap(self.h, _ptr(a), spp, _ptr(out), len(a)))
        return out

    def tonemap_device(self, d_accum, spp, d_out, npix):
        _check(lib.rt_tonemap_device(self.h, d_accum, spp, d_out, npix))


class Scene:
    """rt_scene.  ctx=None gives a host-only scene (flattening can be inspected, nothing runs)."""

    @classmethod
    def _borrow(cls, ctx, handle):
        """A view of a scene owned by a World."""
        s = cls.__new__(cls)
        s.ctx = ctx
        s.h = C.c_void_p(handle)
        s._borrowed = True
        return s

    def __init__(self, ctx, materials, objects, root_kind=ROOT_BVH, build=None, _leave_open=False, cache_dir=None):
        self.ctx = ctx
        self._borrowed = False
        h = C.c_void_p()
        _check(lib.rt_scene_create(ctx.h if ctx is not None else None, C.byref(h)))
        self.h = h
        _live.add(self)
        m = np.ascontiguousarray(materials, dtype=material_dt)
        _check(lib.rt_scene_set_materials(self.h, _ptr(m), len(m)))
        def add(items):
            """('spheres', arr) | ('mesh', arr[, xform 4x4 column-major]) | ('bvh', [items][, axis]) |
            ('list', [items]): the last two are container objects whose children are further items."""
            for item in items:
                kind, arr = item[0], item[1]
                if kind == "spheres":
                    a = np.ascontiguousarray(arr, dtype=sphere_dt)
                    _check(lib.rt_scene_add_spheres(self.h, _ptr(a), len(a)))
                elif kind == "mesh":
                    a = np.ascontiguousarray(arr, dtype=triangle_dt)
                    x = None
                    if len(item) > 2 and item[2] is not None:
                        x = np.ascontiguousarray(item[2], dtype=np.float32).reshape(16)
                    _check(lib.rt_scene_add_mesh(self.h, _ptr(a), len(a), _ptr(x) if x is not None else None))
                elif kind in ("bvh", "list"):
                    _check(lib.rt_scene_begin_group(self.h, ROOT_BVH if kind == "bvh" else ROOT_LIST,
                                                    item[2] if kind == "bvh" and len(item) > 2 else 0))
                    add(arr)
                    _check(lib.rt_scene_end_group(self.h))
                else:
                    raise ValueError(kind)
        add(objects)
        if _leave_open:                        # (tests: a container that is never closed)
            _check(lib.rt_scene_begin_group(self.h, ROOT_BVH, 0))
        _check(lib.rt_scene_set_root(self.h, root_kind))
        if build is not None:
            _check(lib.rt_scene_set_build(self.h, build))
        if cache_dir is not None:
            _check(lib.rt_scene_set_cache_dir(self.h, str(cache_dir).encode()))
        _check(lib.rt_scene_commit(self.h))

    def close(self):
        if getattr(self, "h", None):
            if not self._borrowed:
                lib.rt_scene_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def cache_state(self):
        """What the commit did with the scene cache: CACHE_OFF / STORED / LOADED / STORE_FAILED."""
        return lib.rt_scene_cache_state(self.h)

    def set_traversal(self, mode):
        _check(lib.rt_scene_set_traversal(self.h, mode))

    def set_wide_build(self, where):
        """WIDE_BUILD_COLLAPSE (host, from the binary trees) or WIDE_BUILD_DEVICE (on the GPU, from the primitives)."""
        _check(lib.rt_scene_set_wide_build(self.h, where))

    @property
    def num_prims(self):
        return lib.rt_scene_num_prims(self.h)

    @property
    def num_entries(self):
        return lib.rt_scene_num_entries(self.h)

    @property
    def expected_cost(self):
        """Expected cost of a ray on the scene's (reference-topology, under BUILD_AUTO) trees, in box tests."""
        return lib.rt_scene_expected_cost(self.h)

    # The flattened arrays name a primitive by its key (rt_abi.h, "Primitive keys").  The accessors
    # below return the arrays as the device holds them with raw=True; by default the key words are
    # replaced by primitive ids, which is what a reader of the arrays wants to compare hits by.
    def _keys_to_ids(self, words, where):
        if len(where):
            words[where] = self.prim_of_key(words[where])

    def entries(self, raw=False):
        n = self.num_entries
        out = np.zeros((n, 16), dtype=np.float32)
        _check(lib.rt_scene_get_entries(self.h, _ptr(out), out.size))
        if not raw:
            u = out.view(np.uint32)
            tags = u[:, 3] >> 30
            rows = np.nonzero((tags == 1) | (tags == 2))[0]
            col = u[:, 7].copy()
            self._keys_to_ids(col, rows)
            u[:, 7] = col
        return out

    def prim_order(self):
        """rank -> primitive id: the order in which the reference's recursion visits the primitives."""
        out = np.zeros(self.num_prims, dtype=np.uint32)
        _check(lib.rt_scene_get_prim_order(self.h, _ptr(out), len(out)))
        return out

    def prim_of_key(self, key):
        """Primitive id(s) of the key(s) the flattened arrays name primitives by (MISS stays MISS)."""
        k = np.asarray(key, dtype=np.uint32)
        low = (k & np.uint32(0x3FFFFFFF)).astype(np.int64)
        rank = np.where((k & np.uint32(0x40000000)) != 0, low, 0x3FFFFFFE - low)
        order = self.prim_order()
        miss = k == np.uint32(MISS)
        if len(order) == 0:
            out = np.full(k.shape, MISS, dtype=np.uint32)
        else:
            out = np.where(miss, np.uint32(MISS), order[np.where(miss, 0, rank) % len(order)])
        return out if out.ndim else int(out)

    def units(self, raw=False):
        """The entries as the per-lane walks read them: (n_units, 4) floats, 16-byte units."""
        out = np.zeros((lib.rt_scene_num_units(self.h), 4), dtype=np.float32)
        _check(lib.rt_scene_get_units(self.h, _ptr(out), out.size))
        if not raw and len(out):
            u = out.view(np.uint32)
            starts, i, n = [], 0, len(out)
            while i < n:                           # entries follow each other; a triangle names its own length
                tag = int(u[i, 3]) >> 30
                if tag in (1, 2):
                    starts.append(i)
                i += int(u[i + 2, 3]) if tag == 1 else 2
            rows = np.asarray(starts, dtype=np.int64) + 1
            col = u[:, 3].copy()
            self._keys_to_ids(col, rows)
            u[:, 3] = col
        return out

    def walk_units(self, raw=False):
        """The units in the linked, locality-ordered form of the reference-order per-lane walks."""
        out = np.zeros((lib.rt_scene_num_walk_units(self.h), 4), dtype=np.float32)
        _check(lib.rt_scene_get_walk_units(self.h, _ptr(out), out.size))
        if not raw and len(out):
            u = out.view(np.uint32)
            n = len(out)
            seen, todo, starts = set([0]), [0], []
            while todo:                            # every entry is reached through the links, from unit 0
                i = todo.pop()
                tag = int(u[i, 3]) >> 30
                if tag == 0:
                    nxt = [int(u[i, 3]) & 0x3FFFFFFF, int(u[i + 1, 3])]
                elif tag == 1:
                    starts.append(i)
                    nxt = [int(u[i + 2, 3])]
                else:
                    if tag == 2:
                        starts.append(i)
                    nxt = [int(u[i + 1, 1])]
                for j in nxt:
                    if j < n and j not in seen:
                        seen.add(j)
                        todo.append(j)
            rows = np.asarray(starts, dtype=np.int64) + 1
            col = u[:, 3].copy()
            self._keys_to_ids(col, rows)
            u[:, 3] = col
        return out

    def wide(self, raw=False):
        """The wide tree of a host-only scene: (nodes (n, 20) uint32, primitive records (m, 16) float32, depth)."""
        n_nodes, n_prims, depth, words = C.c_size_t(), C.c_size_t(), C.c_uint32(), C.c_uint32()
        _check(lib.rt_scene_wide_info(self.h, C.byref(n_nodes), C.byref(n_prims), C.byref(depth), C.byref(words)))
        nodes = np.zeros((n_nodes.value, words.value), dtype=np.uint32)
        prims = np.zeros((n_prims.value, 16), dtype=np.float32)
        _check(lib.rt_scene_get_wide(self.h, _ptr(nodes), nodes.size, _ptr(prims) if prims.size else None, prims.size))
        if not raw and len(prims):
            u = prims.view(np.uint32)
            col = u[:, 7].copy()
            self._keys_to_ids(col, np.arange(len(prims)))
            u[:, 7] = col
        return nodes, prims, depth.value

    def trace(self, rays, t_min=0.001, t_max=np.inf):
        rays = np.ascontiguousarray(rays, dtype=ray_dt)
        out = np.zeros(len(rays), dtype=hit_dt)
        _check(lib.rt_trace(self.h, _ptr(rays), len(rays), t_min, t_max, _ptr(out)))
        return out

    def trace_device(self, d_rays, n, d_hits, t_min=0.001, t_max=np.inf, count=False):
        counts = (C.c_uint64 * 3)() if count else None
        _check(lib.rt_trace_device(self.h, d_rays, n, t_min, t_max, d_hits, counts))
        return tuple(counts) if count else None

    def trace_device_static(self, d_rays, n, d_hits, d_counts, t_min=0.001, t_max=np.inf):
        """rt_trace_device_static: allocates nothing, queues on the context's stream and returns;
        d_counts is device memory for three uint64 (box, triangle, sphere tests)."""
        _check(lib.rt_trace_device_static(self.h, d_rays, n, t_min, t_max, d_hits, d_counts))

    @staticmethod
    def params(width, height, spp, max_bounces, sky, rect=None, samples=None, seed=DEFAULT_SEED,
               flags=0, t_min=0.001, max_paths=0, bands=None):
        """bands = (band_rows, band_count, band_index) renders only that interleaved band group."""
        x0, y0, x1, y1 = rect if rect is not None else (0, 0, width, height)
        s0, s1 = samples if samples is not None else (0, spp)
        br, bc, bi = bands if bands is not None else (0, 0, 0)
        return RenderParams(width, height, x0, y0, x1, y1, s0, s1, max_bounces, t_min, sky, seed,
                            flags, max_paths, br, bc, bi, 0)

    def render(self, camera, params, accum=None):
        """Host-buffer render; returns (accum[H,W,3] sums, stats dict)."""
        cam = np.ascontiguousarray(camera, dtype=camera_dt).reshape(1)
        if accum is None:
            accum = np.zeros((params.height, params.width, 3), dtype=np.float32)
        st = Stats()
        _check(lib.rt_render(self.h, _ptr(cam), C.byref(params), _ptr(accum), C.byref(st)))
        return accum, st.as_dict()

    def render_multi(self, camera, params, accum=None):
        """rt_render_multi: the frame split over every GPU of the (multi-GPU) context and sum-reduced
        to the root; returns (accum[H,W,3] sums, stats dict)."""
        cam = np.ascontiguousarray(camera, dtype=camera_dt).reshape(1)
        if accum is None:
            accum = np.zeros((params.height, params.width, 3), dtype=np.float32)
        st = Stats()
        _check(lib.rt_render_multi(self.h, _ptr(cam), C.byref(params), _ptr(accum), C.byref(st)))
        return accum, st.as_dict()

    def render_multi_device(self, camera, params, d_accum):
        """same, adding into a device buffer on the context's root GPU"""
        cam = np.ascontiguousarray(camera, dtype=camera_dt).reshape(1)
        st = Stats()
        _check(lib.rt_render_multi_device(self.h, _ptr(cam), C.byref(params), d_accum, C.byref(st)))
        return st.as_dict()

    def render_device(self, camera, params, d_accum, want_stats=True):
        cam = np.ascontiguousarray(camera, dtype=camera_dt).reshape(1)
        st = Stats()
        _check(lib.rt_render_device(self.h, _ptr(cam), C.byref(params), d_accum,
                                    C.byref(st) if want_stats else None))
        return st.as_dict() if want_stats else None


class World:
    """rt_world: the reference's World {camera, hittable, sky_intensity} (main.rs:69-73) built by
    the C++ host layer and committed to the GPU.  ctx=None builds a host-only world."""

    def __init__(self, ctx, handle):
        self.ctx = ctx
        self.h = handle
        self.scene = Scene._borrow(ctx, lib.rt_world_scene(handle))
        _live.add(self)

    @staticmethod
    def _make(ctx, fn, *args):
        h = C.c_void_p()
        _check(fn(ctx.h if ctx is not None else None, *args, C.byref(h)))
        return World(ctx, h)

    @staticmethod
    def world1(ctx):
        return World._make(ctx, lib.rt_world_create_world1)

    @staticmethod
    def world2(ctx, thing_obj=None, variant=0):
        path = thing_obj or os.path.join(ASSET_DIR, "thing.obj")
        return World._make(ctx, lib.rt_world_create_world2, path.encode(), variant)

    @staticmethod
    def cornell_box(ctx, obj_path=None):
        path = obj_path or os.path.join(ASSET_DIR, "cornell_box.obj")
        return World._make(ctx, lib.rt_world_create_cornell_box, path.encode())

    @staticmethod
    def soup(ctx, n_tris, grid=64, seed=12345):
        return World._make(ctx, lib.rt_world_create_soup, n_tris, grid, seed)

    def close(self):
        if getattr(self, "h", None):
            self.scene.close()
            lib.rt_world_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def sky(self):
        return lib.rt_world_sky_intensity(self.h)

    @property
    def root_kind(self):
        return lib.rt_world_root_kind(self.h)

    def camera(self, width, height):
        """Camera::setup(width / height) -> rt_camera record."""
        out = np.zeros(1, dtype=camera_dt)
        _check(lib.rt_world_camera(self.h, np.float32(width) / np.float32(height), _ptr(out)))
        return out[0]

    def materials(self):
        out = np.zeros(lib.rt_world_num_materials(self.h), dtype=material_dt)
        _check(lib.rt_world_get_materials(self.h, _ptr(out)))
        return out

    def objects(self):
        """[('spheres', sphere_dt[]) | ('mesh', triangle_dt[])] in insertion order."""
        out, run = [], []
        for i in range(lib.rt_world_num_objects(self.h)):
            cnt = C.c_size_t(0)
            kind = lib.rt_world_object_info(self.h, i, C.byref(cnt))
            if kind == 0:
                sp = np.zeros(1, dtype=sphere_dt)
                _check(lib.rt_world_get_sphere(self.h, i, _ptr(sp)))
                run.append(sp)
            else:
                if run:
                    out.append(("spheres", np.concatenate(run)))
                    run = []
                tr = np.zeros(cnt.value, dtype=triangle_dt)
                _check(lib.rt_world_get_mesh(self.h, i, _ptr(tr)))
                xf = np.zeros(16, dtype=np.float32)
                has = lib.rt_world_get_mesh_xform(self.h, i, _ptr(xf))
                if has < 0:
                    _check(has)
                out.append(("mesh", tr, xf) if has else ("mesh", tr))
        if run:
            out.append(("spheres", np.concatenate(run)))
        return out

    def params(self, width, height, spp, max_bounces, **kw):
        return Scene.params(width, height, spp, max_bounces, self.sky, **kw)

    def render_progressive(self, width, height, spp, max_bounces, on_block=None, seed=DEFAULT_SEED):
        """render() with its live updates: on_block(x, y, w, h, block[h,w] uint32, done, total) after
        every step of the progressive schedule for every 64x64 block (return True to stop).
        Returns (sums[H,W,3], stats)."""
        acc = np.zeros((height, width, 3), dtype=np.float32)
        st = Stats()

        def thunk(_user, x, y, w, h, ptr, done, total):
            if on_block is None:
                return 0
            blk = np.ctypeslib.as_array(ptr, shape=(h, w)).copy()
            return 1 if on_block(x, y, w, h, blk, done, total) else 0

        cb = BLOCK_FN(thunk)
        _check(lib.rt_world_render_progressive(self.h, width, height, spp, max_bounces, seed, cb, None, _ptr(acc),
                                               C.byref(st)))
        return acc, st.as_dict()

    def render(self, width, height, spp, max_bounces, seed=DEFAULT_SEED):
        """render() drop-in: full frame, samples [0, spp); returns (sums[H,W,3], stats)."""
        acc = np.zeros((height, width, 3), dtype=np.float32)
        st = Stats()
        _check(lib.rt_world_render(self.h, width, height, spp, max_bounces, seed, _ptr(acc), C.byref(st)))
        return acc, st.as_dict()


def partition_params(params, n_ranks, rank):
    """rt_partition: a copy of `params` (the whole job) narrowed to the share of `rank` of `n_ranks`."""
    p = RenderParams.from_buffer_copy(params)
    _check(lib.rt_partition(n_ranks, rank, C.byref(p)))
    return p


def set_default_cache_dir(path):
    """Directory of the binary scene cache for scenes created from now on (None turns it off)."""
    _check(lib.rt_set_default_cache_dir(str(path).encode() if path else None))


def set_default_build(mode):
    """Build mode (BUILD_REFERENCE / BUILD_SAH) of scenes created from now on, World constructors included."""
    _check(lib.rt_set_default_build(mode))


def progressive_schedule(spp):
    """Cumulative sample counts of the reference's progressive tile job (main.rs:355-356, 409-411)."""
    n = lib.rt_progressive_schedule(spp, None, 0)
    out = np.zeros(n, dtype=np.uint32)
    lib.rt_progressive_schedule(spp, _ptr(out), n)
    return [int(v) for v in out]


def write_ppm(path, pixels_0rgb):
    px = np.ascontiguousarray(pixels_0rgb, dtype=np.uint32)
    _check(lib.rt_write_ppm(str(path).encode(), _ptr(px), px.shape[1], px.shape[0]))


def camera_setup(look_from, look_at, vup, vfov, aperture, focus_dist, aspect_ratio):
    f = np.ascontiguousarray(look_from, dtype=np.float32)
    a = np.ascontiguousarray(look_at, dtype=np.float32)
    u = np.ascontiguousarray(vup, dtype=np.float32)
    out = np.zeros(1, dtype=camera_dt)
    _check(lib.rt_camera_setup(_ptr(f), _ptr(a), _ptr(u), vfov, aperture, focus_dist, aspect_ratio, _ptr(out)))
    return out[0]


def obj_count(path):
    t, m, f = C.c_size_t(0), C.c_size_t(0), C.c_size_t(0)
    _check(lib.rt_obj_count(path.encode(), C.byref(t), C.byref(m), C.byref(f)))
    return t.value, m.value, f.value
