"""ctypes binding of the CPU oracle (oracle/liboracle.so) -- TEST INFRASTRUCTURE ONLY.

Imported by tests/, by __graft_entry__.smoke() and by bench.py's cpu_baseline leg.  The product
package (raytracer_amd) never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")

LAMBERT, METAL, DIELECTRIC, EMISSIVE = 0, 1, 2, 3
ROOT_BVH, ROOT_SINGLE, ROOT_LIST = 0, 1, 2
MISS = 0xFFFFFFFF

# numpy mirrors of the C structs (all fields 4-byte, no padding)
material_dt = np.dtype([("kind", "<u4"), ("rgb", "<f4", 3), ("param", "<f4"), ("flags", "<u4")])
sphere_dt = np.dtype([("c", "<f4", 3), ("r", "<f4"), ("mat", "<u4")])
triangle_dt = np.dtype([("p", "<f4", (3, 3)), ("n0", "<f4", 3), ("mat", "<u4")])
ray_dt = np.dtype([("o", "<f4", 3), ("d", "<f4", 3)])
hit_dt = np.dtype([("t", "<f4"), ("prim_id", "<u4"), ("p", "<f4", 3), ("n", "<f4", 3),
                   ("front_face", "<u4"), ("mat", "<u4")])
camera_dt = np.dtype([("origin", "<f4", 3), ("horizontal", "<f4", 3), ("vertical", "<f4", 3),
                      ("lower_left", "<f4", 3), ("u", "<f4", 3), ("v", "<f4", 3),
                      ("lens_radius", "<f4")])


class Stats(C.Structure):
    _fields_ = [("rays", C.c_uint64), ("paths", C.c_uint64), ("box_tests", C.c_uint64),
                ("tri_tests", C.c_uint64), ("sph_tests", C.c_uint64), ("hits", C.c_uint64),
                ("seconds", C.c_double)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


def build():
    """Compile liboracle.so with the Makefile next to this file."""
    subprocess.check_call(["make", "-s", "-C", _HERE, "liboracle.so"])


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        build()
    L = C.CDLL(_LIB_PATH)
    vp, sz, u32, f32, i32 = C.c_void_p, C.c_size_t, C.c_uint32, C.c_float, C.c_int
    sig = {
        "oracle_scene_new": (vp, []),
        "oracle_scene_free": (None, [vp]),
        "oracle_scene_set_materials": (None, [vp, vp, sz]),
        "oracle_scene_add_spheres": (None, [vp, vp, sz]),
        "oracle_scene_add_mesh": (None, [vp, vp, sz]),
        "oracle_scene_add_mesh_xf": (None, [vp, vp, sz, vp]),
        "oracle_scene_begin_group": (None, [vp, i32, i32]),
        "oracle_scene_end_group": (None, [vp]),
        "oracle_mat4_inverse": (None, [vp, vp]),
        "oracle_mat4_transform": (None, [vp, vp, f32, vp]),
        "oracle_scene_set_root": (None, [vp, i32]),
        "oracle_scene_set_camera": (None, [vp, vp, vp, vp, f32, f32, f32]),
        "oracle_scene_set_sky": (None, [vp, f32]),
        "oracle_scene_commit": (i32, [vp]),
        "oracle_create_world1": (vp, []),
        "oracle_create_world2": (vp, [C.c_char_p, i32]),
        "oracle_create_cornell_box": (vp, [C.c_char_p]),
        "oracle_scene_num_objects": (sz, [vp]),
        "oracle_scene_num_prims": (sz, [vp]),
        "oracle_scene_num_materials": (sz, [vp]),
        "oracle_scene_num_spheres": (sz, [vp]),
        "oracle_scene_num_triangles": (sz, [vp]),
        "oracle_scene_get_materials": (None, [vp, vp]),
        "oracle_scene_object_info": (i32, [vp, sz, C.POINTER(sz)]),
        "oracle_scene_get_sphere": (None, [vp, sz, vp]),
        "oracle_scene_get_mesh": (None, [vp, sz, vp]),
        "oracle_scene_get_xform": (i32, [vp, sz, vp]),
        "oracle_scene_sky": (f32, [vp]),
        "oracle_scene_root_kind": (i32, [vp]),
        "oracle_scene_camera": (None, [vp, f32, vp]),
        "oracle_scene_camera_params": (None, [vp, vp]),
        "oracle_trace": (None, [vp, vp, sz, f32, f32, vp, C.POINTER(Stats), i32]),
        "oracle_render": (None, [vp, u32, u32, u32, u32, u32, u32, i32, vp, vp, C.POINTER(Stats)]),
        "oracle_record_rays": (sz, [vp, u32, u32, u32, u32, u32, vp, sz, u32]),
        "oracle_xorshift32": (None, [u32, vp, sz]),
        "oracle_random01": (None, [u32, vp, sz]),
        "oracle_random01_of": (f32, [u32]),
        "oracle_rgb_to_u32": (u32, [f32, f32, f32]),
        "oracle_tonemap": (None, [vp, u32, vp, sz]),
        "oracle_aabb_hit": (i32, [vp, vp, vp, vp, f32, f32]),
        "oracle_ray_triangle": (i32, [vp, vp, vp, vp, vp, vp]),
        "oracle_obj_count": (sz, [C.c_char_p, C.POINTER(sz), C.POINTER(sz)]),
        "oracle_scatter": (i32, [vp, vp, vp, vp, vp, i32, C.POINTER(u32), vp]),
        "oracle_rng_state_after_world1": (u32, []),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def _f3(v):
    return np.ascontiguousarray(v, dtype=np.float32)


class Scene:
    """Owns an oracle_scene*."""

    def __init__(self, handle):
        if not handle:
            raise RuntimeError("oracle scene construction failed")
        self.h = C.c_void_p(handle)

    def __del__(self):
        try:
            if self.h:
                lib().oracle_scene_free(self.h)
                self.h = None
        except Exception:
            pass

    # ---- constructors -------------------------------------------------------------------
    @staticmethod
    def world1():
        return Scene(lib().oracle_create_world1())

    @staticmethod
    def world2(thing_obj, variant=0):
        return Scene(lib().oracle_create_world2(thing_obj.encode(), variant))

    @staticmethod
    def cornell_box(obj_path):
        return Scene(lib().oracle_create_cornell_box(obj_path.encode()))

    @staticmethod
    def from_description(materials, objects, root_kind=ROOT_BVH, camera=None, sky=1.0):
        """objects: list of ('spheres', sphere_dt array) / ('mesh', triangle_dt array)."""
        L = lib()
        s = Scene(L.oracle_scene_new())
        m = np.ascontiguousarray(materials, dtype=material_dt)
        L.oracle_scene_set_materials(s.h, _ptr(m), len(m))

        def add(items):
            """('spheres', arr) | ('mesh', arr[, xform 4x4 column-major]) | ('bvh', [items][, axis]) |
            ('list', [items]): the last two are container objects whose children are further items."""
            for item in items:
                kind, arr = item[0], item[1]
                if kind == "spheres":
                    a = np.ascontiguousarray(arr, dtype=sphere_dt)
                    L.oracle_scene_add_spheres(s.h, _ptr(a), len(a))
                elif kind == "mesh":
                    a = np.ascontiguousarray(arr, dtype=triangle_dt)
                    if len(item) > 2 and item[2] is not None:
                        x = np.ascontiguousarray(item[2], dtype=np.float32).reshape(16)
                        L.oracle_scene_add_mesh_xf(s.h, _ptr(a), len(a), _ptr(x))
                    else:
                        L.oracle_scene_add_mesh(s.h, _ptr(a), len(a))
                elif kind in ("bvh", "list"):
                    L.oracle_scene_begin_group(s.h, ROOT_BVH if kind == "bvh" else ROOT_LIST,
                                               item[2] if kind == "bvh" and len(item) > 2 else 0)
                    add(arr)
                    L.oracle_scene_end_group(s.h)
                else:
                    raise ValueError(kind)
        add(objects)
        L.oracle_scene_set_root(s.h, root_kind)
        if camera is not None:
            f, a, u, vfov, ap, focus = camera
            L.oracle_scene_set_camera(s.h, _ptr(_f3(f)), _ptr(_f3(a)), _ptr(_f3(u)), vfov, ap, focus)
        L.oracle_scene_set_sky(s.h, sky)
        L.oracle_scene_commit(s.h)
        return s

    # ---- introspection ------------------------------------------------------------------
    def materials(self):
        n = lib().oracle_scene_num_materials(self.h)
        out = np.zeros(n, dtype=material_dt)
        lib().oracle_scene_get_materials(self.h, _ptr(out))
        return out

    def objects(self):
        """The leaf objects in insertion order, in from_description() form (spheres one by one are
        merged into runs; a transformed mesh carries its matrix; containers are not reproduced)."""
        L = lib()
        out = []
        run = []
        for i in range(L.oracle_scene_num_objects(self.h)):
            cnt = C.c_size_t(0)
            kind = L.oracle_scene_object_info(self.h, i, C.byref(cnt))
            if kind == 0:
                sp = np.zeros(1, dtype=sphere_dt)
                L.oracle_scene_get_sphere(self.h, i, _ptr(sp))
                run.append(sp)
            else:
                if run:
                    out.append(("spheres", np.concatenate(run)))
                    run = []
                tr = np.zeros(cnt.value, dtype=triangle_dt)
                L.oracle_scene_get_mesh(self.h, i, _ptr(tr))
                xf = np.zeros(16, dtype=np.float32)
                if L.oracle_scene_get_xform(self.h, i, _ptr(xf)):
                    out.append(("mesh", tr, xf))
                else:
                    out.append(("mesh", tr))
        if run:
            out.append(("spheres", np.concatenate(run)))
        return out

    @property
    def sky(self):
        return lib().oracle_scene_sky(self.h)

    @property
    def root_kind(self):
        return lib().oracle_scene_root_kind(self.h)

    @property
    def num_prims(self):
        return lib().oracle_scene_num_prims(self.h)

    @property
    def num_triangles(self):
        return lib().oracle_scene_num_triangles(self.h)

    @property
    def num_spheres(self):
        return lib().oracle_scene_num_spheres(self.h)

    def camera(self, aspect):
        out = np.zeros(1, dtype=camera_dt)
        lib().oracle_scene_camera(self.h, aspect, _ptr(out))
        return out[0]

    def camera_params(self):
        out = np.zeros(12, dtype=np.float32)
        lib().oracle_scene_camera_params(self.h, _ptr(out))
        return out

    # ---- hot path -----------------------------------------------------------------------
    def trace(self, rays, t_min=0.001, t_max=np.inf, nthreads=8):
        rays = np.ascontiguousarray(rays, dtype=ray_dt)
        out = np.zeros(len(rays), dtype=hit_dt)
        st = Stats()
        lib().oracle_trace(self.h, _ptr(rays), len(rays), t_min, t_max, _ptr(out), C.byref(st), nthreads)
        return out, st.as_dict()

    def render(self, width, height, spp, max_bounces, block=64, seed_offset=0, nthreads=8,
               want_image=True):
        acc = np.zeros((height, width, 3), dtype=np.float32)
        img = np.zeros((height, width), dtype=np.uint32) if want_image else None
        st = Stats()
        lib().oracle_render(self.h, width, height, spp, max_bounces, block, seed_offset, nthreads,
                            _ptr(acc), _ptr(img) if want_image else None, C.byref(st))
        return acc, img, st.as_dict()

    def record_rays(self, width, height, spp, max_bounces, cap, stride=1, seed_offset=0):
        out = np.zeros(cap, dtype=ray_dt)
        n = lib().oracle_record_rays(self.h, width, height, spp, max_bounces, seed_offset, _ptr(out),
                                     cap, stride)
        return out[:n].copy()


# ---- small pieces ---------------------------------------------------------------------------
def xorshift32(seed, n):
    out = np.zeros(n, dtype=np.uint32)
    lib().oracle_xorshift32(seed, _ptr(out), n)
    return out


def random01(seed, n):
    out = np.zeros(n, dtype=np.float32)
    lib().oracle_random01(seed, _ptr(out), n)
    return out


def random01_of(x):
    return np.float32(lib().oracle_random01_of(x))


def rgb_to_u32(r, g, b):
    return lib().oracle_rgb_to_u32(r, g, b)


def tonemap(accum, spp):
    a = np.ascontiguousarray(accum, dtype=np.float32).reshape(-1, 3)
    out = np.zeros(len(a), dtype=np.uint32)
    lib().oracle_tonemap(_ptr(a), spp, _ptr(out), len(a))
    return out


def aabb_hit(bmin, bmax, o, d, t_min, t_max):
    return bool(lib().oracle_aabb_hit(_ptr(_f3(bmin)), _ptr(_f3(bmax)), _ptr(_f3(o)), _ptr(_f3(d)),
                                      t_min, t_max))


def ray_triangle(o, d, v0, v1, v2):
    out = np.zeros(4, dtype=np.float32)
    h = lib().oracle_ray_triangle(_ptr(_f3(o)), _ptr(_f3(d)), _ptr(_f3(v0)), _ptr(_f3(v1)),
                                  _ptr(_f3(v2)), _ptr(out))
    return bool(h), bool(out[0]), float(out[1]), float(out[2]), float(out[3])


def obj_count(path):
    nm, nf = C.c_size_t(0), C.c_size_t(0)
    n = lib().oracle_obj_count(path.encode(), C.byref(nm), C.byref(nf))
    return n, nm.value, nf.value


def scatter(material, ray_o, ray_d, p, n, front_face, rng_state):
    m = np.ascontiguousarray(material, dtype=material_dt).reshape(1)
    st = C.c_uint32(rng_state)
    out = np.zeros(9, dtype=np.float32)
    ok = lib().oracle_scatter(_ptr(m), _ptr(_f3(ray_o)), _ptr(_f3(ray_d)), _ptr(_f3(p)), _ptr(_f3(n)),
                              int(front_face), C.byref(st), _ptr(out))
    return bool(ok), out[0:3].copy(), out[3:6].copy(), out[6:9].copy(), st.value


def mat4_inverse(m):
    a = np.ascontiguousarray(m, dtype=np.float32).reshape(16)
    out = np.zeros(16, dtype=np.float32)
    lib().oracle_mat4_inverse(_ptr(a), _ptr(out))
    return out


def mat4_transform(m, v, w):
    a = np.ascontiguousarray(m, dtype=np.float32).reshape(16)
    out = np.zeros(3, dtype=np.float32)
    lib().oracle_mat4_transform(_ptr(a), _ptr(_f3(v)), w, _ptr(out))
    return out


def rng_state_after_world1():
    return lib().oracle_rng_state_after_world1()
