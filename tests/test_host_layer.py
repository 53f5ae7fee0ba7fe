"""CPU-side checks of the product's host code (no GPU needed): the C-ABI library loads and
exports every declared symbol, the C++ scene layer reproduces the oracle's scene descriptions,
and the flattened pre-order entry array encodes the reference's traversal (checked by walking it
in Python with f32 arithmetic and comparing hits with the oracle)."""
import ctypes
import os
import re

import numpy as np
import pytest

from helpers import ROOT, asset, random_rays, random_scene


def _declared(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rt_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(rt):
    names = _declared("rt_abi.h") + _declared("rt_host.h")
    assert len(names) >= 35
    raw = ctypes.CDLL(rt.LIB_PATH)
    missing = [n for n in names if not hasattr(raw, n)]
    assert not missing, missing
    assert sorted(rt.ABI_SYMBOLS) == _declared("rt_abi.h")
    assert sorted(rt.HOST_SYMBOLS) == _declared("rt_host.h")


def test_struct_layouts_match_header(rt):
    assert rt.material_dt.itemsize == 24 and rt.sphere_dt.itemsize == 20
    assert rt.triangle_dt.itemsize == 52 and rt.ray_dt.itemsize == 24
    assert rt.hit_dt.itemsize == 40 and rt.camera_dt.itemsize == 76
    assert ctypes.sizeof(rt.RenderParams) == 80 and ctypes.sizeof(rt.Stats) == 224


def test_no_gpu_fails_loudly(rt):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(rt.RtError):
        rt.Context(0)
    w = rt.World.cornell_box(None, asset("cornell_box_plane.obj"))      # host-only world
    with pytest.raises(rt.RtError, match="host-only"):
        w.scene.trace(np.zeros(4, dtype=rt.ray_dt))
    with pytest.raises(rt.RtError, match="host-only"):
        w.render(16, 16, 1, 2)


def test_argument_errors(rt):
    mats = np.zeros(1, dtype=rt.material_dt)
    sp = np.zeros(1, dtype=rt.sphere_dt)
    sp["mat"] = 1
    with pytest.raises(rt.RtError, match="material index"):
        rt.Scene(None, mats, [("spheres", sp)])
    with pytest.raises(rt.RtError, match="exactly one object"):
        rt.Scene(None, mats, [], rt.ROOT_SINGLE)
    with pytest.raises(rt.RtError, match="no materials"):
        rt.Scene(None, mats[:0], [])
    bad = mats.copy()
    bad["kind"] = 9
    with pytest.raises(rt.RtError, match="material kind"):
        rt.Scene(None, bad, [])
    with pytest.raises(rt.RtError):
        rt.World.cornell_box(None, "/nonexistent/file.obj")
    with pytest.raises(rt.RtError):
        rt.obj_count("/nonexistent/file.obj")


def _resolved(materials, objects):
    """Object list with material indices replaced by the material records they point at."""
    out = []
    for item in objects:
        kind, arr = item[0], item[1]
        geo = b"".join(np.ascontiguousarray(arr[n]).tobytes() for n in arr.dtype.names if n != "mat")
        xf = np.asarray(item[2], dtype=np.float32).tobytes() if len(item) > 2 else b""
        out.append((kind, geo, materials[arr["mat"]].tobytes(), xf))
    return out


@pytest.mark.parametrize("which", ["world1", "world2", "world2_glass", "world2_planes", "cornell", "cornell_plane"])
def test_scene_constructors_match_oracle(rt, ob, which):
    if which == "world1":
        w, o = rt.World.world1(None), ob.Scene.world1()
    elif which == "world2":
        w, o = rt.World.world2(None, asset("thing.obj"), 0), ob.Scene.world2(asset("thing.obj"), 0)
    elif which == "world2_glass":
        w, o = rt.World.world2(None, asset("thing.obj"), 1), ob.Scene.world2(asset("thing.obj"), 1)
    elif which == "world2_planes":             # Mesh::plane + Mat4::from_translation (main.rs:195-204)
        w, o = rt.World.world2(None, asset("thing.obj"), 2), ob.Scene.world2(asset("thing.obj"), 2)
        assert [len(it) for it in w.objects()[:3]] == [3, 3, 2] and w.objects()[0][2][14] == np.float32(-0.5)
    elif which == "cornell":
        w, o = rt.World.cornell_box(None, asset("cornell_box.obj")), ob.Scene.cornell_box(asset("cornell_box.obj"))
    else:
        w = rt.World.cornell_box(None, asset("cornell_box_plane.obj"))
        o = ob.Scene.cornell_box(asset("cornell_box_plane.obj"))
    assert _resolved(w.materials(), w.objects()) == _resolved(o.materials(), o.objects())
    assert w.root_kind == o.root_kind and w.sky == o.sky
    for wd, ht in [(1920, 1080), (512, 512), (100, 77)]:
        assert w.camera(wd, ht).tobytes() == o.camera(np.float32(wd) / np.float32(ht)).tobytes()
    assert w.scene.num_prims == o.num_prims


def test_obj_loader_counts(rt, ob):
    for name in ["cornell_box.obj", "cornell_box_plane.obj", "thing.obj"]:
        assert rt.obj_count(asset(name)) == ob.obj_count(asset(name))


def test_obj_loader_syntax_variants(rt, ob, tmp_path):
    (tmp_path / "m.mtl").write_text("newmtl a\nKd 1 1 1\nnewmtl  b \n")
    text = """# comment
mtllib m.mtl
v 0 0 0
v 1 0 0
v 1 1 0
v 0 1 0
v 0.5 0.5 1
vt 0 0
vn 0 0 1
vn 0 1 0
g first
usemtl a
f 1/1/1 2/1/1 3/1/1 4/1/1
f -5//1 -4//1 -1//2
usemtl b
f 1//2 2//2 5//2
f 1//1 2//1 3//1 4//1 5//1
l 1 2
p 1
o second
usemtl nosuch
f 3//1 4//1 5//1
"""
    (tmp_path / "t.obj").write_text(text.replace("\n", "\r\n"))
    path = str(tmp_path / "t.obj")
    want = ob.obj_count(path)
    assert rt.obj_count(path) == want == (8, 3, 5)      # quad 2 + tri 1 | tri 1 + pentagon 3 | tri 1


def test_obj_loader_chunks_give_the_single_pass_result(rt, ob, tmp_path, monkeypatch):
    """The loader parses a file in chunks cut at line ends (one per thread on big files) and replays
    the state-changing lines in file order.  Whatever the number of chunks -- forced here through
    RT_OBJ_THREADS on files of a few lines to a few MB, so that cuts fall between every kind of
    line -- models, faces and triangles must be those of one chunk: relative indices, material
    changes and object names straddling the cuts included."""
    lines = ["mtllib m.mtl"]
    rng = np.random.default_rng(5)
    n_v = 0
    for obj in range(7):
        lines.append(f"o part{obj}")
        for f in range(9):
            if f % 3 == 0:
                lines.append("usemtl " + ["white", "red", "green", "nosuch"][(obj + f) % 4])
            pts = rng.uniform(-1, 1, (4, 3))
            for q in pts:
                lines.append("v %.6f %.6f %.6f" % tuple(q))
            lines.append("vn %.6f %.6f %.6f" % tuple(rng.uniform(-1, 1, 3)))
            n_v += 4
            if f % 2:
                lines.append(f"f -4//-1 -3//-1 -2//-1 -1//-1")
            else:
                lines.append(f"f {n_v - 3}//{obj * 9 + f + 1} {n_v - 2}//{obj * 9 + f + 1} {n_v}//{obj * 9 + f + 1}")
    (tmp_path / "m.mtl").write_text("newmtl white\nnewmtl red\nnewmtl green\n")
    (tmp_path / "gen.obj").write_text("\n".join(lines))            # no newline at the end
    files = [str(tmp_path / "gen.obj"), asset("cornell_box.obj"), asset("cornell_box_plane.obj"), asset("thing.obj")]
    for path in files:
        monkeypatch.setenv("RT_OBJ_THREADS", "1")
        want_count = rt.obj_count(path)
        assert want_count == ob.obj_count(path)
        w = rt.World.cornell_box(None, path)
        want = w.objects()
        w.close()
        for n in (2, 3, 5, 16, 61):
            monkeypatch.setenv("RT_OBJ_THREADS", str(n))
            assert rt.obj_count(path) == want_count
            w = rt.World.cornell_box(None, path)
            got = w.objects()
            w.close()
            assert len(got) == len(want)
            for (ka, a), (kb, b) in zip(got, want):
                assert ka == kb and np.array_equal(a, b)
    monkeypatch.delenv("RT_OBJ_THREADS")
    n_tris, _, n_faces = rt.obj_count(files[0])
    assert n_faces == 63 and n_tris == 63 + 7 * 4      # 28 of the faces are quads


def test_camera_setup_matches_oracle(rt, ob):
    rng = np.random.default_rng(0)
    for _ in range(20):
        f, a = rng.uniform(-10, 10, 3), rng.uniform(-10, 10, 3)
        vfov, ap, focus, aspect = rng.uniform(10, 120), rng.uniform(0, 0.5), rng.uniform(0.5, 300), rng.uniform(0.5, 2)
        s = ob.Scene.from_description(np.zeros(1, dtype=ob.material_dt), [], ob.ROOT_BVH,
                                      camera=(f, a, [0, 0, 1], vfov, ap, focus))
        got = rt.camera_setup(f, a, [0, 0, 1], vfov, ap, focus, aspect)
        assert got.tobytes() == s.camera(aspect).tobytes()


# ---- the flattened entry array --------------------------------------------------------------
F = np.float32


def _dot(a, b):
    return F(F(a[0] * b[0]) + F(a[1] * b[1])) + F(a[2] * b[2])


def _box_miss(e, o, inv, t_min, best_t):
    tmn, tmx = F(t_min), best_t
    for k in range(3):
        t0 = F(F(e[k] - o[k]) * inv[k])
        t1 = F(F(e[4 + k] - o[k]) * inv[k])
        if inv[k] < 0:
            t0, t1 = t1, t0
        tmn = t0 if t0 > tmn else tmn
        tmx = t1 if t1 < tmx else tmx
        if tmx <= tmn:
            return True
    return False


def _prim_test(e, tag, prim, o, d, t_min, best_t, best):
    eps = F(1.1920929e-7)
    if tag == 1:
        v0, e1, e2 = e[0:3], e[4:7], e[8:11]
        p = [F(F(d[1] * e2[2]) - F(d[2] * e2[1])), F(F(d[2] * e2[0]) - F(d[0] * e2[2])),
             F(F(d[0] * e2[1]) - F(d[1] * e2[0]))]
        det = _dot(e1, p)
        if not (abs(det) < eps):
            invd = F(1.0) / det
            tv = [F(o[k] - v0[k]) for k in range(3)]
            u = F(_dot(tv, p) * invd)
            if not (u < 0 or u > 1):
                q = [F(F(tv[1] * e1[2]) - F(tv[2] * e1[1])), F(F(tv[2] * e1[0]) - F(tv[0] * e1[2])),
                     F(F(tv[0] * e1[1]) - F(tv[1] * e1[0]))]
                v = F(_dot(d, q) * invd)
                if not (v < 0 or F(u + v) > 1):
                    t = F(_dot(e2, q) * invd)
                    if t >= t_min and t < best_t:
                        return t, prim
        return best_t, best
    c, r = e[0:3], e[4]
    oc = [F(o[k] - c[k]) for k in range(3)]
    a = _dot(d, d)
    hb = _dot(oc, d)
    cc = F(_dot(oc, oc) - F(r * r))
    disc = F(F(hb * hb) - F(a * cc))
    if not (disc < 0):
        sq = np.sqrt(disc)
        root = F(F(-hb - sq) / a)
        if root < t_min or best_t < root:
            root = F(F(-hb + sq) / a)
            if root < t_min or best_t < root:
                return best_t, best
        return root, prim
    return best_t, best


def _mat_vec(m, v, w):
    """glam Mat4 * (v, w), truncated: ((x_axis * x + y_axis * y) + z_axis * z) + w_axis * w in f32 steps"""
    out = []
    for k in range(3):
        acc = F(m[k] * v[0])
        acc = F(F(m[4 + k] * v[1]) + acc)
        acc = F(F(m[8 + k] * v[2]) + acc)
        acc = F(F(m[12 + k] * F(w)) + acc)
        out.append(acc)
    return out


def _walk(entries, o, d, t_min, t_max, inverses=None, count=False):
    """Pre-order walk with skip links, f32 arithmetic; mirrors the device loop.  Tag 3 entries are the
    markers around a transformed mesh: word 0 = 1 takes the ray to the mesh's space with the inverse
    transform word 1 names, word 0 = 2 restores it."""
    ent_u = entries.view(np.uint32)
    n = len(entries)
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        wo, wd = [F(x) for x in o], [F(x) for x in d]
        o, d = wo, wd
        inv = [F(1.0) / d[k] for k in range(3)]
        best_t, best = F(t_max), 0xFFFFFFFF
        i = 0
        n_box = n_prim = 0
        while i < n:
            e = entries[i]
            w = int(ent_u[i, 3])
            tag = w >> 30
            if tag == 0:
                n_box += 1
                i = (w & 0x3FFFFFFF) if _box_miss(e, o, inv, t_min, best_t) else i + 1
            elif tag == 3:
                if int(ent_u[i, 0]) == 1:
                    m = inverses[int(ent_u[i, 1])]
                    o, d = _mat_vec(m, wo, 1.0), _mat_vec(m, wd, 0.0)
                else:
                    assert int(ent_u[i, 0]) == 2
                    o, d = wo, wd
                inv = [F(1.0) / d[k] for k in range(3)]
                i += 1
            else:
                n_prim += 1
                best_t, best = _prim_test(e, tag, int(ent_u[i, 7]), o, d, t_min, best_t, best)
                i += 1
    return (best, best_t, n_box, n_prim) if count else (best, best_t)


def _walk_ordered(entries, o, d, t_min, t_max):
    """The opt-in ordered walk: same tests, nearer child first, the other child on a stack."""
    ent_u = entries.view(np.uint32)
    n = len(entries)
    visits = 0
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        inv = [F(1.0) / d[k] for k in range(3)]
        best_t, best = F(t_max), 0xFFFFFFFF
        top, stack = 0, []
        while True:
            if stack:
                i = stack.pop()
            elif top < n:
                i = top
                w = int(ent_u[i, 3])
                top = (w & 0x3FFFFFFF) if (w >> 30) == 0 else i + 1
            else:
                break
            while True:
                visits += 1
                e = entries[i]
                w = int(ent_u[i, 3])
                tag = w >> 30
                if tag != 0:
                    best_t, best = _prim_test(e, tag, int(ent_u[i, 7]), o, d, t_min, best_t, best)
                    break
                if _box_miss(e, o, inv, t_min, best_t):
                    break
                link = int(ent_u[i, 7])
                second = link & 0x1FFFFFFF
                nxt = i + 1
                if second:
                    axis = (link >> 29) & 3
                    first_is_near = (d[axis] >= 0) == bool(link >> 31)
                    stack.append(second if first_is_near else nxt)
                    nxt = nxt if first_is_near else second
                i = nxt
    return best, best_t, visits


@pytest.mark.parametrize("root", [0, 2])
def test_flattened_entries_encode_reference_traversal(rt, ob, root):
    mats, objects, _ = random_scene(ob, 11 + root, n_spheres=9, n_meshes=2, tris_per_mesh=13, extent=4.0)
    osc = ob.Scene.from_description(mats, objects, root)
    sc = rt.Scene(None, mats, objects, root)
    ent = sc.entries()
    assert len(ent) == sc.num_entries and sc.num_prims == osc.num_prims
    rays = random_rays(ob, 5, 400, -5.0, 5.0)
    want, _ = osc.trace(rays, 0.001, np.inf, nthreads=1)
    for k in range(len(rays)):
        prim, t = _walk(ent, rays["o"][k], rays["d"][k], F(0.001), F(np.inf))
        assert prim == want["prim_id"][k]
        if prim != 0xFFFFFFFF:
            assert F(t).view(np.uint32) == want["t"][k].view(np.uint32)
    assert (want["prim_id"] != 0xFFFFFFFF).sum() > 50


def test_radix_sorted_mesh_tree_equals_the_comparison_sorted_one(rt, ob, monkeypatch):
    """Mesh::build_bvh sorts by Triangle::get_aabb().min[axis] with a stable comparison sort
    (shape.rs:268-272).  Ranges of 4096 triangles and more are sorted by a stable radix sort on the key
    bits here: on a mesh whose keys are full of duplicates, zeros of both signs, denormals, negative
    and huge values, the flattened tree must equal the comparison-sorted one bit for bit, and a walk
    over it must give the oracle's hits and the oracle's visit counts (the oracle sorts by comparison)."""
    rng = np.random.default_rng(77)
    n = 9000
    pool = np.array([0.0, -0.0, 1e-42, -1e-42, 1.0, -1.0, 1.0000001, 3e30, -3e30, 0.5, -0.5, 2.0 ** -126, 7.25],
                    dtype=np.float32)
    tr = np.zeros(n, dtype=ob.triangle_dt)
    base = pool[rng.integers(0, len(pool), (n, 1, 3))]
    fine = rng.uniform(-2.0, 2.0, (n, 1, 3)).astype(np.float32)
    corner = np.where(rng.random((n, 1, 3)) < 0.6, base, fine)               # the minimum corner: the sort key
    tr["p"] = corner + np.abs(rng.uniform(0.0, 0.7, (n, 3, 3))).astype(np.float32) * (rng.random((n, 3, 1)) < 0.7)
    tr["p"][:, 0] = corner[:, 0]
    tr["n0"] = [0.0, 0.0, 1.0]
    tr["mat"] = 0
    mats, _, _ = random_scene(ob, 3)
    objects = [("mesh", tr)]
    sc = rt.Scene(None, mats, objects, 1)
    monkeypatch.setenv("RT_BUILD_NO_RADIX", "1")
    plain = rt.Scene(None, mats, objects, 1)
    monkeypatch.delenv("RT_BUILD_NO_RADIX")
    ent = sc.entries()
    assert np.array_equal(ent.view(np.uint32), plain.entries().view(np.uint32))
    assert np.array_equal(sc.walk_units().view(np.uint32), plain.walk_units().view(np.uint32))
    osc = ob.Scene.from_description(mats, objects, 1)
    rays = random_rays(ob, 9, 60, -2.5, 2.5)
    want, st = osc.trace(rays, 0.001, np.inf, nthreads=1)
    boxes = tris = 0
    for k in range(len(rays)):
        prim, t, nb, nt = _walk(ent, rays["o"][k], rays["d"][k], F(0.001), F(np.inf), count=True)
        assert prim == want["prim_id"][k]
        boxes += nb
        tris += nt
    assert (boxes, tris) == (st["box_tests"], st["tri_tests"])


def _nested_description(ob, seed, with_xforms):
    """Containers inside containers (a Bvh of [spheres, a HittableList of [mesh, Bvh(axis 2) of spheres],
    mesh]) and, optionally, transformed meshes."""
    from helpers import random_mesh
    rng = np.random.default_rng(seed)
    mats, _, _ = random_scene(ob, seed)

    def spheres(n):
        sp = np.zeros(n, dtype=ob.sphere_dt)
        sp["c"] = rng.uniform(-4, 4, (n, 3)).astype(np.float32)
        sp["r"] = rng.uniform(0.2, 0.9, n).astype(np.float32)
        sp["mat"] = rng.integers(0, 6, n)
        return sp

    def xf(kind):
        if not with_xforms:
            return None
        m = np.eye(4, dtype=np.float32)
        if kind == 0:
            m[:3, 3] = [1.5, -0.75, 0.25]                              # Mat4::from_translation
        elif kind == 1:
            m[0, 0], m[1, 1], m[2, 2] = 1.5, 0.5, 2.0                  # a scale
            m[:3, 3] = [-1.0, 0.5, 0.0]
        else:
            a = np.float32(0.7)                                        # rotation about z, shear and translation
            m[:2, :2] = [[np.cos(a), -np.sin(a)], [np.sin(a), np.cos(a)]]
            m[0, 2] = 0.3
            m[:3, 3] = [0.2, 0.1, -0.6]
        return m.T.reshape(16).copy()                                  # column-major

    inner = ("bvh", [("spheres", spheres(5))], 2)
    lst = ("list", [("mesh", random_mesh(ob, rng, 9, extent=3.0), xf(0)), inner, ("spheres", spheres(1))])
    objects = [("spheres", spheres(4)), lst, ("mesh", random_mesh(ob, rng, 14, extent=3.0), xf(1)),
               ("bvh", [("mesh", random_mesh(ob, rng, 1, extent=2.0), xf(2)), ("bvh", [], 1)]),
               ("mesh", random_mesh(ob, rng, 6, extent=3.0))]
    return mats, objects


@pytest.mark.parametrize("root", [0, 2])
@pytest.mark.parametrize("with_xforms", [False, True])
def test_nested_containers_and_transformed_meshes(rt, ob, root, with_xforms):
    """Bvh / HittableList objects inside one another flatten into the same pre-order array (a nested
    Bvh is a subtree built with its own starting axis, a nested list a run of siblings without a box),
    and a transformed mesh is its tree between an ENTER and a LEAVE marker.  Walking the entries (and
    the linked units) must give the oracle's hits, which builds the containers as the reference's
    constructors nest (shape.rs:55-57, 87-138) and transforms rays as Mesh::hit does (shape.rs:454)."""
    mats, objects = _nested_description(ob, 61 + root, with_xforms)
    osc = ob.Scene.from_description(mats, objects, root)
    sc = rt.Scene(None, mats, objects, root)
    assert sc.num_prims == osc.num_prims == 4 + 9 + 5 + 1 + 14 + 1 + 6
    ent = sc.entries()
    u = ent.view(np.uint32)
    markers = u[(u[:, 3] >> 30) == 3]
    assert len(markers) == (6 if with_xforms else 0)
    inverses = None
    if with_xforms:
        assert markers[:, 0].tolist() == [1, 2, 1, 2, 1, 2] and sorted(set(markers[:, 1].tolist())) == [0, 1, 2]
        flat = [it for it in _flatten(objects) if it[0] == "mesh" and len(it) > 2 and it[2] is not None]
        inverses = [ob.mat4_inverse(it[2]) for it in flat]             # transforms are numbered in the order added
    rays = random_rays(ob, 12, 400, -4.0, 4.0)
    want, _ = osc.trace(rays, 0.001, np.inf, nthreads=1)
    assert (want["prim_id"] != 0xFFFFFFFF).sum() > 40
    assert len(set(want["prim_id"].tolist())) > 15                     # spheres and triangles of several objects
    for k in range(len(rays)):
        prim, t = _walk(ent, rays["o"][k], rays["d"][k], F(0.001), F(np.inf), inverses)
        assert prim == want["prim_id"][k]
        if prim != 0xFFFFFFFF:
            assert F(t).view(np.uint32) == want["t"][k].view(np.uint32)
    if not with_xforms:
        units = sc.walk_units()
        for k in range(0, len(rays), 3):
            prim, t, _ = _walk_linked(units, rays["o"][k], rays["d"][k], F(0.001), F(np.inf))
            assert prim == want["prim_id"][k]
    with pytest.raises(rt.RtError, match="not closed"):
        s2 = rt.Scene.__new__(rt.Scene)
        rt.Scene.__init__(s2, None, mats, [("bvh", [])], 0, _leave_open=True)


def _flatten(objects):
    for it in objects:
        if it[0] in ("bvh", "list"):
            yield from _flatten(it[1])
        else:
            yield it


def test_child_links_of_the_ordered_walk(rt, ob):
    """Box entries carry (second child, axis, first-child-is-lower); the ordered walk over them
    finds the reference's hits on a scene without coincident primitives, with fewer visits."""
    mats, objects, _ = random_scene(ob, 21, n_spheres=40, n_meshes=2, tris_per_mesh=60, extent=6.0)
    osc = ob.Scene.from_description(mats, objects, 0)
    ent = rt.Scene(None, mats, objects, 0).entries()
    u = ent.view(np.uint32)
    n = len(ent)
    end_of = lambda i: int(u[i, 3] & 0x3FFFFFFF) if (u[i, 3] >> 30) == 0 else i + 1
    two = 0
    for i in range(n):
        if (u[i, 3] >> 30) != 0:
            continue
        end, link = end_of(i), int(u[i, 7])
        second = link & 0x1FFFFFFF
        if i + 1 < end and end_of(i + 1) < end:
            assert second == end_of(i + 1) and end_of(second) == end         # exactly two children
            assert ((link >> 29) & 3) < 3
            two += 1
        else:
            assert link == 0
    assert two > 50
    rays = random_rays(ob, 8, 300, -7.0, 7.0)
    want, st = osc.trace(rays, 0.001, np.inf, nthreads=1)
    visits = 0
    for k in range(len(rays)):
        prim, t, v = _walk_ordered(ent, rays["o"][k], rays["d"][k], F(0.001), F(np.inf))
        visits += v
        assert prim == want["prim_id"][k]
        if prim != 0xFFFFFFFF:
            assert F(t).view(np.uint32) == want["t"][k].view(np.uint32)
    assert visits < st["box_tests"] + st["tri_tests"] + st["sph_tests"]


@pytest.mark.parametrize("leaf", [1, 4])
def test_sah_build_keeps_format_and_hits(rt, ob, leaf, monkeypatch):
    """RT_BUILD_SAH moves only the split positions of mesh trees.  With one triangle per leaf
    (the default) every box still has exactly two children and the entry count is unchanged; with
    RT_SAH_LEAF=4 a node of at most four triangles is a leaf: its box is followed by the triangles in
    file order.  Either way every box is the padded union of what is below it, every primitive appears
    once, and the walks over the entries find the oracle's closest hits."""
    monkeypatch.setenv("RT_SAH_LEAF", str(leaf))
    mats, objects, _ = random_scene(ob, 41, n_spheres=12, n_meshes=2, tris_per_mesh=57, extent=5.0)
    osc = ob.Scene.from_description(mats, objects, 0)
    ref = rt.Scene(None, mats, objects, 0)
    sah = rt.Scene(None, mats, objects, 0, build=rt.BUILD_SAH)
    ent, u = sah.entries(), sah.entries().view(np.uint32)
    assert sah.num_prims == ref.num_prims
    assert (len(ent) == ref.num_entries) if leaf == 1 else (len(ent) < ref.num_entries - 20)
    tags = u[:, 3] >> 30
    assert sorted(u[tags != 0, 7].tolist()) == list(range(sah.num_prims))       # each primitive once
    end_of = lambda i: int(u[i, 3] & 0x3FFFFFFF) if tags[i] == 0 else i + 1
    pad = np.float32(0.0001)
    mesh_of_prim = np.concatenate([np.full(len(o[1]), k if o[0] == "mesh" else -1 - k) for k, o in enumerate(objects)])
    big_leaves = 0
    for i in np.nonzero(tags == 0)[0]:
        end = end_of(i)
        first = end_of(i + 1)                              # one child (an object-level leaf), two, or a leaf's run
        n_kids, c = 0, i + 1
        while c < end:
            n_kids, c = n_kids + 1, end_of(c)
        if n_kids > 2:
            assert leaf > 1 and n_kids <= leaf and all(tags[i + 1:end] == 1)
            assert u[i + 1:end, 7].tolist() == sorted(u[i + 1:end, 7].tolist())     # file order
            big_leaves += 1
        else:
            assert first == end or end_of(first) == end
        kids = np.arange(i + 1, end)
        tri = kids[tags[kids] == 1]
        # a node of a mesh tree: n leaves of ONE mesh, n - 1 boxes (the surface-area split also re-groups the
        # objects of a Bvh, whose boxes are plain unions: two small meshes under one such box do not count)
        if len(kids) + 1 == 2 * len(tri) - 1 and len(set(mesh_of_prim[u[tri, 7]].tolist())) == 1:
            p0 = ent[tri, 0:3]
            pts = np.concatenate([p0, p0 + ent[tri, 4:7], p0 + ent[tri, 8:11]])
            # vertices are stored as p0 and two edges; the sums round, so compare loosely
            np.testing.assert_allclose(ent[i, 0:3], pts.min(axis=0) - pad, atol=1e-5)
            np.testing.assert_allclose(ent[i, 4:7], pts.max(axis=0) + pad, atol=1e-5)
    assert (big_leaves > 5) == (leaf > 1)
    rays = random_rays(ob, 9, 300, -6.0, 6.0)
    want, _ = osc.trace(rays, 0.001, np.inf, nthreads=1)
    for k in range(len(rays)):
        for walk in ((_walk, _walk_ordered) if leaf == 1 else (_walk,)):
            prim, t = walk(ent, rays["o"][k], rays["d"][k], F(0.001), F(np.inf))[:2]
            assert prim == want["prim_id"][k]
            if prim != 0xFFFFFFFF:
                assert F(t).view(np.uint32) == want["t"][k].view(np.uint32)
    assert (want["prim_id"] != 0xFFFFFFFF).sum() > 50
    with pytest.raises(rt.RtError):
        rt.Scene(None, mats, objects, 0, build=7)


def test_entry_layout_small_cases(rt, ob):
    mats = np.zeros(2, dtype=rt.material_dt)
    mats["kind"] = [rt.MAT_LAMBERT, rt.MAT_METAL]
    tr = np.zeros(3, dtype=rt.triangle_dt)
    tr["p"][:, 1, 0] = 1.0
    tr["p"][:, 2, 1] = 1.0
    tr["p"][:, :, 2] = [[0.0], [1.0], [2.0]]
    tr["n0"][:, 2] = 1.0
    tr["mat"] = [0, 1, 0]
    ent = rt.Scene(None, mats, [("mesh", tr)], rt.ROOT_SINGLE).entries()
    u = ent.view(np.uint32)
    # 3 triangles sorted on x (all equal => stable): left = build([t0]) = leaf; right = build([t1,t2])
    # = box{ t2 (pop = last), t1 }.  Pre-order: box, t0, box, t2, t1.
    assert len(ent) == 5
    assert [int(x >> 30) for x in u[:, 3]] == [0, 1, 0, 1, 1]
    assert [int(x) for x in u[[1, 3, 4], 7]] == [0, 2, 1]
    assert int(u[0, 3] & 0x3FFFFFFF) == 5 and int(u[2, 3] & 0x3FFFFFFF) == 5
    np.testing.assert_array_equal(ent[0, 0:3], np.float32([0, 0, 0]) - np.float32(0.0001))
    np.testing.assert_array_equal(ent[0, 4:7], np.float32([1, 1, 2]) + np.float32(0.0001))
    np.testing.assert_array_equal(ent[2, 0:3], np.float32([0, 0, 1]) - np.float32(0.0001))
    assert ent.shape[1] == 16 and not ent[:, 12:].any()               # 64-byte entries, last float4 is padding
    # child links of the ordered walk: box 0 = {t0 | box 2}, box 2 = {t2 | t1}; z separates both pairs
    assert [int(x & 0x1FFFFFFF) for x in u[[0, 2], 7]] == [2, 4]
    assert [int((x >> 29) & 3) for x in u[[0, 2], 7]] == [2, 2]
    assert [int(x >> 31) for x in u[[0, 2], 7]] == [1, 0]
    np.testing.assert_array_equal(ent[1, 4:7], [1, 0, 0])              # v0v1
    np.testing.assert_array_equal(ent[1, 8:11], [0, 1, 0])             # v0v2


def test_units_are_the_entries_repacked(rt, ob):
    """The per-lane walks read 16-byte units: a box or a sphere is the first two float4 of its entry,
    a triangle the first three (plus the distance to the next entry, which is 4 where a padding unit
    follows), links count units.  Walking the units must give the entries back, with every link
    mapped."""
    mats, objects, _ = random_scene(ob, 31, n_spheres=17, n_meshes=2, tris_per_mesh=23, extent=5.0)
    sc = rt.Scene(None, mats, objects, 0)
    ent, units = sc.entries(), sc.units()
    assert units.shape[1] == 4
    eu, uu = ent.view(np.uint32), units.view(np.uint32)
    unit_of, u = [], 0
    for i in range(len(ent)):
        if (eu[i, 3] >> 30) != 1 and u % 2:       # boxes and spheres start at even units (32-byte aligned):
            u += 1                                # a padding unit behind the triangle before them
        unit_of.append(u)
        u += 3 if (eu[i, 3] >> 30) == 1 else 2
    unit_of.append(u)
    assert u == len(units)
    assert any(b - a == 4 for a, b in zip(unit_of, unit_of[1:]))
    for i in range(len(ent)):
        tag, k = int(eu[i, 3] >> 30), unit_of[i]
        assert int(uu[k, 3] >> 30) == tag
        np.testing.assert_array_equal(units[k, 0:3], ent[i, 0:3])
        np.testing.assert_array_equal(units[k + 1, 0:3], ent[i, 4:7])
        if tag == 0:
            assert int(uu[k, 3] & 0x3FFFFFFF) == unit_of[int(eu[i, 3] & 0x3FFFFFFF)]
            second = int(eu[i, 7] & 0x1FFFFFFF)
            assert int(uu[k + 1, 3] & 0x1FFFFFFF) == (unit_of[second] if second else 0)
            assert int(uu[k + 1, 3] >> 29) == int(eu[i, 7] >> 29)
        else:
            assert uu[k + 1, 3] == eu[i, 7]                               # primitive id
            if tag == 1:
                np.testing.assert_array_equal(units[k + 2, 0:3], ent[i, 8:11])
                assert uu[k + 2, 3] == unit_of[i + 1] - k          # distance to the next entry: 3 or 4


def _walk_linked(units, o, d, t_min, t_max):
    """The device's reference-order per-lane loop over the linked units: a box continues at its
    child (word 7) or its skip link (word 3), a triangle at word 11, a sphere at word 5."""
    uu = units.view(np.uint32)
    n = len(units)
    visited = []
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        inv = [F(1.0) / d[k] for k in range(3)]
        best_t, best = F(t_max), 0xFFFFFFFF
        i = 0
        while i < n:
            visited.append(i)
            tag = int(uu[i, 3] >> 30)
            e = np.concatenate([units[i], units[i + 1], units[i + 2] if tag == 1 else np.zeros(4, F)])
            if tag == 0:
                i = int(uu[i, 3] & 0x3FFFFFFF) if _box_miss(e, o, inv, t_min, best_t) else int(uu[i + 1, 3])
            else:
                best_t, best = _prim_test(e, tag, int(uu[i + 1, 3]), o, d, t_min, best_t, best)
                i = int(uu[i + 2, 3]) if tag == 1 else int(uu[i + 1, 1])
        assert i == n                                  # every walk ends on the end units, nowhere else
    return best, best_t, visited


@pytest.mark.parametrize("root", [0, 2])
@pytest.mark.parametrize("top", ["0", "64", "100000"])
@pytest.mark.parametrize("build", [0, 1])
def test_linked_units_keep_the_reference_order(rt, ob, root, top, build, monkeypatch):
    """The locality-ordered, explicitly linked units of the per-lane walks: every entry appears once,
    boxes and spheres are 32-byte aligned, the two children of a box lie side by side and a pair of
    boxes inside one 64-byte line, and following the links visits exactly the entries, in exactly the
    order, of the pre-order walk -- whatever part of the tree is laid out breadth first."""
    monkeypatch.setenv("RT_LAYOUT_TOP", top)
    mats, objects, _ = random_scene(ob, 51 + root, n_spheres=21, n_meshes=2, tris_per_mesh=37, extent=5.0)
    osc = ob.Scene.from_description(mats, objects, root)
    sc = rt.Scene(None, mats, objects, root, build=build)     # (the surface-area tree's units are laid out the same way)
    ent, units = sc.entries(), sc.walk_units()
    eu, uu = ent.view(np.uint32), units.view(np.uint32)
    n = len(ent)
    # recover pos[] by following child / next links from unit 0 in pre-order
    pos = {}

    def follow(i, k):                                  # entry i lives at unit k
        assert i not in pos
        pos[i] = k
        tag = int(eu[i, 3] >> 30)
        assert int(uu[k, 3] >> 30) == tag
        np.testing.assert_array_equal(units[k, 0:3], ent[i, 0:3])
        np.testing.assert_array_equal(units[k + 1, 0:(1 if tag == 2 else 3)], ent[i, 4:(5 if tag == 2 else 7)])
        if tag == 0:
            assert k % 2 == 0
            end = int(eu[i, 3] & 0x3FFFFFFF)
            c, ck = i + 1, int(uu[k + 1, 3])
            kids = []
            while c < end:                             # the children sit side by side
                follow(c, ck)
                kids.append((c, ck))
                ctag = int(eu[c, 3] >> 30)
                nxt = int(uu[ck, 3] & 0x3FFFFFFF) if ctag == 0 else (int(uu[ck + 2, 3]) if ctag == 1 else int(uu[ck + 1, 1]))
                c = int(eu[c, 3] & 0x3FFFFFFF) if ctag == 0 else c + 1
                if c < end:
                    assert nxt - ck in (2, 3, 4)       # the sibling is adjacent (one padding unit at most)
                    ck = nxt
                else:
                    assert nxt == int(uu[k, 3] & 0x3FFFFFFF)     # the last child leaves where its parent does
            if len(kids) == 2 and all(int(eu[c, 3] >> 30) != 1 for c, _ in kids):
                assert kids[0][1] // 4 == (kids[1][1] + 1) // 4  # both boxes inside one 64-byte line
        else:
            assert uu[k + 1, 3] == eu[i, 7]
            if tag == 1:
                np.testing.assert_array_equal(units[k + 2, 0:3], ent[i, 8:11])
            else:
                assert k % 2 == 0

    i, k = 0, 0
    while i < n:                                       # top-level entries
        follow(i, k)
        tag = int(eu[i, 3] >> 30)
        k = int(uu[k, 3] & 0x3FFFFFFF) if tag == 0 else (int(uu[k + 2, 3]) if tag == 1 else int(uu[k + 1, 1]))
        i = int(eu[i, 3] & 0x3FFFFFFF) if tag == 0 else i + 1
    assert k == len(units) and sorted(pos) == list(range(n))
    rays = random_rays(ob, 6, 150, -6.0, 6.0)
    want, _ = osc.trace(rays, 0.001, np.inf, nthreads=1)
    unit_to_entry = {v: key for key, v in pos.items()}
    for r in range(len(rays)):
        prim, t, visited = _walk_linked(units, rays["o"][r], rays["d"][r], F(0.001), F(np.inf))
        assert prim == want["prim_id"][r]
        if prim != 0xFFFFFFFF:
            assert F(t).view(np.uint32) == want["t"][r].view(np.uint32)
        order = [unit_to_entry[v] for v in visited]
        assert order == sorted(order)                  # pre-order = the reference's visiting order
    assert (want["prim_id"] != 0xFFFFFFFF).sum() > 30


def test_scene_cache_round_trip(rt, ob, tmp_path, monkeypatch):
    """The binary scene cache: a 500 k-triangle soup committed with a cache directory writes
    <key>.rtscene; committing the same description again loads it (no build) and gives identical
    flattened arrays; another build mode or layout parameter is another key; a damaged file is
    ignored and rebuilt."""
    import time
    rt.set_default_cache_dir(tmp_path)
    try:
        t0 = time.time()
        a = rt.World.soup(None, 500_000, grid=16, seed=77)
        t_build = time.time() - t0
        assert a.scene.cache_state == rt.CACHE_STORED
        files = sorted(tmp_path.glob("*.rtscene"))
        assert len(files) == 1 and files[0].stat().st_size > 500_000 * 2 * 64
        t0 = time.time()
        b = rt.World.soup(None, 500_000, grid=16, seed=77)
        t_load = time.time() - t0
        assert b.scene.cache_state == rt.CACHE_LOADED
        print(f"500 k triangles: generate + build + store {t_build:.2f} s, generate + load {t_load:.2f} s")
        for get in (lambda s: s.entries(), lambda s: s.units(), lambda s: s.walk_units()):
            x, y = get(a.scene), get(b.scene)
            assert x.shape == y.shape and np.array_equal(x.view(np.uint32), y.view(np.uint32))
        assert a.scene.num_prims == b.scene.num_prims == 500_000 + 16 * 16 + 1
        # another key: other triangles, another build mode, another layout
        c = rt.World.soup(None, 500_000, grid=16, seed=78)
        assert c.scene.cache_state == rt.CACHE_STORED and len(list(tmp_path.glob("*.rtscene"))) == 2
        rt.set_default_build(rt.BUILD_SAH)
        d = rt.World.soup(None, 500_000, grid=16, seed=77)
        rt.set_default_build(rt.BUILD_AUTO)
        assert d.scene.cache_state == rt.CACHE_STORED
        monkeypatch.setenv("RT_LAYOUT_TOP", "0")
        e = rt.World.soup(None, 500_000, grid=16, seed=77)
        monkeypatch.delenv("RT_LAYOUT_TOP")
        assert e.scene.cache_state == rt.CACHE_STORED and len(list(tmp_path.glob("*.rtscene"))) == 4
        # a truncated file is not trusted
        data = files[0].read_bytes()
        files[0].write_bytes(data[:len(data) // 2])
        f = rt.World.soup(None, 500_000, grid=16, seed=77)
        assert f.scene.cache_state == rt.CACHE_STORED
        assert np.array_equal(f.scene.walk_units().view(np.uint32), a.scene.walk_units().view(np.uint32))
        assert files[0].stat().st_size == len(data)
        # the hits of a cached scene are the oracle's (small scene through the core ABI)
        mats, objects, _ = random_scene(ob, 5, n_spheres=12, n_meshes=2, tris_per_mesh=40, extent=4.0)
        s1 = rt.Scene(None, mats, objects, 0, cache_dir=tmp_path)
        s2 = rt.Scene(None, mats, objects, 0, cache_dir=tmp_path)
        assert (s1.cache_state, s2.cache_state) == (rt.CACHE_STORED, rt.CACHE_LOADED)
        osc = ob.Scene.from_description(mats, objects, 0)
        rays = random_rays(ob, 3, 100, -5.0, 5.0)
        want, _ = osc.trace(rays, 0.001, np.inf, nthreads=1)
        ent = s2.entries()
        for k in range(len(rays)):
            assert _walk(ent, rays["o"][k], rays["d"][k], F(0.001), F(np.inf))[0] == want["prim_id"][k]
        # a directory that does not exist: the scene is built, the store fails, nothing else does
        s3 = rt.Scene(None, mats, objects, 0, cache_dir=tmp_path / "missing")
        assert s3.cache_state == rt.CACHE_STORE_FAILED and s3.num_prims == s1.num_prims
    finally:
        rt.set_default_cache_dir(None)
        rt.set_default_build(rt.BUILD_AUTO)


def test_progressive_schedule_and_ppm(rt, tmp_path):
    """main.rs:355-356, 409-411: min(10, spp), then doubling, capped at spp."""
    assert rt.progressive_schedule(1) == [1]
    assert rt.progressive_schedule(10) == [10]
    assert rt.progressive_schedule(11) == [10, 11]
    assert rt.progressive_schedule(50) == [10, 20, 40, 50]
    assert rt.progressive_schedule(1024) == [10, 20, 40, 80, 160, 320, 640, 1024]
    px = np.array([[0x00FF0000, 0x0000FF00, 0x000000FF], [0x00102030, 0, 0x00FFFFFF]], dtype=np.uint32)
    path = tmp_path / "f.ppm"
    rt.write_ppm(path, px)
    raw = path.read_bytes()
    assert raw.startswith(b"P6\n3 2\n255\n")
    assert raw[len(b"P6\n3 2\n255\n"):] == bytes([255, 0, 0, 0, 255, 0, 0, 0, 255, 0x10, 0x20, 0x30, 0, 0, 0, 255, 255, 255])
    with pytest.raises(rt.RtError):
        rt.write_ppm(tmp_path / "no_such_dir" / "f.ppm", px)


def test_headless_main_fails_loudly_without_a_gpu(rt):
    """examples/rt_render_main.cpp (the stand-in for the reference's main()) links the library only;
    on a box without a GPU it must stop at rt_context_create with the library's message."""
    import subprocess
    import torch
    from helpers import ROOT
    exe = os.path.join(ROOT, "raytracer_amd", "rt_render")
    assert os.path.exists(exe), "build() compiles raytracer_amd/rt_render"
    bad = subprocess.run([exe, "--bogus"], capture_output=True, text=True, timeout=60)
    assert bad.returncode == 2 and "unknown argument" in bad.stderr
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: the render itself is covered by the GPU tests")
    out = subprocess.run([exe, "--scene", "cornell_plane", "--obj", asset("cornell_box_plane.obj"), "--size", "32x32",
                          "--spp", "2", "--out", os.devnull], capture_output=True, text=True, timeout=60)
    assert out.returncode == 1 and "rt_context_create" in out.stderr


def test_rust_bindings_cover_the_headers():
    """rust/raytracer-gpu-sys/src/lib.rs cannot be compiled here (no Rust toolchain), so its text is
    held against the headers: every C function is declared once with the same number of
    arguments, and every enumerator / flag it names has the header's value."""
    def strip(text):
        return re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    c_funcs, c_consts = {}, {}
    for header in ("rt_abi.h", "rt_host.h"):
        text = strip(open(os.path.join(ROOT, "include", header)).read())
        for m in re.finditer(r"^[\w \*]+?\b(rt_\w+)\s*\(([^;{]*?)\)\s*;", text, flags=re.M):
            args = m.group(2).strip()
            c_funcs[m.group(1)] = 0 if args in ("", "void") else args.count(",") + 1
        for body in re.findall(r"enum\s*\{(.*?)\}", text, flags=re.S):
            for name, val in re.findall(r"(RT_\w+)\s*=\s*(-?\w+)", body):
                c_consts[name] = int(val, 0)
        for name, val in re.findall(r"#define\s+(RT_\w+)\s+(0x[0-9A-Fa-f]+|\d+)u?\b", text):
            c_consts[name] = int(val, 0)
    rust = open(os.path.join(ROOT, "rust", "raytracer-gpu-sys", "src", "lib.rs")).read()
    rust = re.sub(r"//[^\n]*", "", rust)
    r_funcs = {}
    for m in re.finditer(r"pub fn (rt_\w+)\s*\((.*?)\)\s*(?:->[^;]*)?;", rust, flags=re.S):
        assert m.group(1) not in r_funcs, m.group(1)
        args = m.group(2).strip()
        r_funcs[m.group(1)] = 0 if not args else len(re.findall(r"\w+\s*:", args))
    assert r_funcs == c_funcs
    r_consts = {n: int(v.replace("_", ""), 0) for n, v in re.findall(r"pub const (RT_\w+): \w+ = (-?[0-9A-Fa-fx_]+);", rust)}
    assert len(r_consts) >= 20
    for name, val in r_consts.items():
        assert name in c_consts, name
        assert c_consts[name] == val or (name == "RT_MISS" and val == 0xFFFFFFFF), name


def test_headers_are_plain_c_and_link_from_c(tmp_path):
    """The boundary is a C ABI: both headers must compile as strict C99 (no C++ in them), and a C
    program must link against the library with nothing but the headers."""
    import subprocess
    src = tmp_path / "c_check.c"
    src.write_text('''#include "rt_abi.h"
#include "rt_host.h"
#include <stdio.h>
int main(void) {
    rt_context* ctx = 0;
    rt_material m; rt_sphere s; rt_triangle t; rt_ray r; rt_hit h; rt_camera c; rt_render_params p; rt_stats st;
    (void)m; (void)s; (void)t; (void)r; (void)h; (void)c; (void)p; (void)st;
    printf("%d %d %d %d %d %d %d %d\\n", (int)sizeof m, (int)sizeof s, (int)sizeof t, (int)sizeof r, (int)sizeof h,
           (int)sizeof c, (int)sizeof p, (int)sizeof st);
    if (rt_context_create(-1, &ctx) == RT_OK) return 1;      /* must fail: no such device */
    printf("%s\\n", rt_last_error());
    return 0;
}
''')
    inc = os.path.join(ROOT, "include")
    lib = os.path.join(ROOT, "raytracer_amd")
    exe = tmp_path / "c_check"
    subprocess.run(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", f"-I{inc}", str(src), "-o", str(exe),
                    f"-L{lib}", "-l:librt_amd.so", f"-Wl,-rpath,{lib}"], check=True, capture_output=True, text=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=60)
    assert out.returncode == 0, out.stdout + out.stderr
    sizes = [int(v) for v in out.stdout.splitlines()[0].split()]
    assert sizes == [24, 20, 52, 24, 40, 76, 80, 224]           # the layouts the Python and Rust stubs assume
    assert out.stdout.splitlines()[1].strip()                  # a message, not an empty string
