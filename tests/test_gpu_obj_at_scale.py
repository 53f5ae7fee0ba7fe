"""OBJ ingest at scale under the GPU tests: a million-triangle OBJ + MTL written here, read by the
library's mirror of Mesh::from_file (shape.rs:347-406; chunked, threaded) with a scene-cache
directory, read once more from the cache file, and traced against the oracle's own reader of the
same file -- primitive id exact, t bit-exact, both loads, in the reference's visiting order and in
the form the library picks by itself."""
import os

import numpy as np
import pytest

from helpers import assert_hits_equal

pytestmark = pytest.mark.gpu

N_TRIS = int(os.environ.get("RT_OBJ_TRIS", "1000000"))
NAMES = ["white", "red", "green", "glass", "metal"]       # names create_cornell_box maps to materials (main.rs:244-288)


def write_obj(path, n, seed=5):
    """n triangles spread over the Cornell room (the camera create_cornell_box sets up looks into it),
    in runs of 1/10 of the triangles per `usemtl`, a normal per vertex as the reference demands
    (shape.rs:393-395); coordinates with nine significant digits, so every reader parses the same f32."""
    rng = np.random.default_rng(seed)
    mtl = os.path.splitext(path)[0] + ".mtl"
    with open(mtl, "w") as f:
        for name in NAMES + ["light", "default"]:
            f.write(f"newmtl {name}\nKd 0.5 0.5 0.5\n")
    with open(path, "w") as f:
        f.write(f"mtllib {os.path.basename(mtl)}\no soup\n")
        run = max(1, n // 10)
        for k, base in enumerate(range(0, n, run)):
            m = min(run, n - base)
            c = np.stack([rng.uniform(-555.0, -5.0, m), rng.uniform(-550.0, -5.0, m), rng.uniform(5.0, 540.0, m)], axis=1)[:, None, :]
            p = (c + rng.uniform(-4.0, 4.0, (m, 3, 3))).astype(np.float32)
            nn = np.cross(p[:, 1] - p[:, 0], p[:, 2] - p[:, 0])
            nn = (nn / np.maximum(np.linalg.norm(nn, axis=1, keepdims=True), 1e-20)).astype(np.float32)
            f.write("".join(f"v {x:.9g} {y:.9g} {z:.9g}\n" for x, y, z in p.reshape(-1, 3).tolist()))
            f.write("".join(f"vn {x:.9g} {y:.9g} {z:.9g}\n" * 3 for x, y, z in nn.tolist()))
            f.write(f"usemtl {NAMES[k % len(NAMES)]}\n")
            idx = np.arange(base * 3 + 1, (base + m) * 3 + 1).reshape(-1, 3)
            f.write("".join(f"f {a}//{a} {b}//{b} {c_}//{c_}\n" for a, b, c_ in idx.tolist()))
    return os.path.getsize(path) + os.path.getsize(mtl)


def test_million_triangle_obj_through_the_cache(rt, ob, ctx, tmp_path):
    import time
    path = str(tmp_path / "soup.obj")
    cache = tmp_path / "cache"
    cache.mkdir()
    t0 = time.time()
    size = write_obj(path, N_TRIS)
    t_write = time.time() - t0
    tris, models, faces = rt.obj_count(path)
    assert tris == N_TRIS == faces and models == 10

    t0 = time.time()
    osc = ob.Scene.cornell_box(path)                           # the oracle's reader and the reference's trees
    t_oracle = time.time() - t0
    assert osc.num_prims == N_TRIS
    rays = np.concatenate([osc.record_rays(256, 256, 1, 6, cap=150000), _room_rays(ob, 3, 150000)])
    assert len(rays) >= 200000
    want, _ = osc.trace(rays)
    assert (want["prim_id"] != ob.MISS).mean() > 0.5 and len(np.unique(want["prim_id"])) > 50000

    rt.set_default_cache_dir(str(cache))
    try:
        loads = []
        for expect in (rt.CACHE_STORED, rt.CACHE_LOADED):
            t0 = time.time()
            world = rt.World.cornell_box(ctx, path)            # Mesh::from_file + commit (RT_BUILD_AUTO: both trees)
            loads.append(time.time() - t0)
            assert world.scene.cache_state == expect and world.scene.num_prims == N_TRIS
            world.scene.set_traversal(rt.TRAVERSAL_PER_LANE)   # the reference's order on the reference's trees
            assert_hits_equal(world.scene.trace(rays), want)
            world.scene.set_traversal(rt.TRAVERSAL_AUTO)
            img, st = world.render(128, 128, 2, 6)             # the first render picks the form ...
            assert_hits_equal(world.scene.trace(rays), want)   # ... whose hits are the same hits
            form = (rt.TREE_NAMES[st["tree"]], rt.FORM_NAMES[st["kernel_form"]])
            world.close()
    finally:
        rt.set_default_cache_dir(None)
    files = sorted(os.listdir(cache))
    assert len(files) == 2 and all(f.endswith(".rtscene") for f in files)      # one file per tree
    print(f"{N_TRIS} triangles: {size / 1e6:.0f} MB of OBJ written in {t_write:.1f} s; oracle read + build {t_oracle:.1f} s; "
          f"library parse + build + store {loads[0]:.1f} s, parse + load from the cache {loads[1]:.1f} s; "
          f"{len(rays)} rays exact on both loads; picked {form}")


def _room_rays(ob, seed, n):
    rng = np.random.default_rng(seed)
    rays = np.zeros(n, dtype=ob.ray_dt)
    rays["o"] = np.stack([rng.uniform(-555, -5, n), rng.uniform(-550, -5, n), rng.uniform(5, 540, n)], axis=1).astype(np.float32)
    d = rng.normal(size=(n, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    d = (d * rng.uniform(0.5, 20.0, (n, 1))).astype(np.float32)
    d[d == 0] = np.float32(1e-3)
    rays["d"] = d
    return rays
