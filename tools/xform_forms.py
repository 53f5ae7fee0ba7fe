"""Rates of the forms that take transformed meshes on one scene (run on the GPU box): four meshes of
`tris` triangles, three of them rotated, scaled and translated, and a few spheres under a sky.
    python tools/xform_forms.py [tris per mesh] [plain]      # plain: the same meshes without their transforms"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
import torch

import raytracer_amd as rt
from helpers import product_camera, random_mesh, random_scene
from oracle import binding as ob

tris = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
plain = len(sys.argv) > 2 and sys.argv[2] == "plain"
rng = np.random.default_rng(11)
mats, _, _ = random_scene(ob, 3)
objects = []
for k in range(4):
    a = np.float32(0.4 * k + 0.2)
    m = np.eye(4, dtype=np.float32)
    m[:2, :2] = [[np.cos(a), -np.sin(a)], [np.sin(a), np.cos(a)]]
    m[2, 2] = 1.0 + 0.25 * k
    m[:3, 3] = [3.0 * (k % 2) - 1.5, 3.0 * (k // 2) - 1.5, 0.5 * k]
    mesh = random_mesh(ob, rng, tris, 2.5)
    objects.append(("mesh", mesh, m.T.reshape(16).copy()) if k != 3 and not plain else ("mesh", mesh))
sp = np.zeros(30, dtype=ob.sphere_dt)
sp["c"] = rng.uniform(-6, 6, (30, 3)).astype(np.float32)
sp["r"] = rng.uniform(0.2, 0.8, 30).astype(np.float32)
sp["mat"] = rng.integers(0, 6, 30)
objects.append(("spheres", sp))
cam = ((9.0, -10.0, 6.0), (0.0, 0.0, 1.0), (0.0, 0.0, 1.0), 45.0, 0.0, 12.0)
osc = ob.Scene.from_description(mats, objects, ob.ROOT_BVH, camera=cam, sky=0.9)
ctx = rt.Context(0)
w = h = 1024
spp = 32
pc = product_camera(rt, osc, w, h)
p = rt.Scene.params(w, h, spp, 6, osc.sky)
accum = torch.zeros(h * w * 3, dtype=torch.float32, device="cuda")
sums = set()
for label, build, mode, env in (("reference tree, per-lane", rt.BUILD_REFERENCE, rt.TRAVERSAL_PER_LANE, {}),
                                ("surface-area tree, per-lane", rt.BUILD_SAH, rt.TRAVERSAL_PER_LANE, {}),
                                ("surface-area tree, wide", rt.BUILD_SAH, rt.TRAVERSAL_WIDE, {}),
                                ("defaults", rt.BUILD_AUTO, rt.TRAVERSAL_AUTO, {})):
    sc = rt.Scene(ctx, mats, objects, ob.ROOT_BVH, build=build)
    sc.set_traversal(mode)
    best = 1e9
    for i in range(4):
        accum.zero_(); torch.cuda.synchronize()
        st = sc.render_device(pc, p, accum.data_ptr())
        if i:
            best = min(best, st["ms_render"])
    sums.add(float(accum.double().sum()))
    accum.zero_(); torch.cuda.synchronize()
    ct = sc.render_device(pc, rt.Scene.params(w, h, 2, 6, osc.sky, flags=rt.RENDER_COUNT_VISITS), accum.data_ptr())
    tests = f"{ct['box_tests'] / ct['rays_traced']:.0f} box + {ct['tri_tests'] / ct['rays_traced']:.0f} triangle tests per ray"
    print(f"{4 * tris} triangles, {label}: {st['rays_traced'] / best / 1e6:.2f} Grays/s ({best:.1f} ms), "
          f"{rt.FORM_NAMES[st['kernel_form']]} on the {rt.TREE_NAMES[st['tree']]} tree, {tests}", flush=True)
    sc.close()
print("frames identical" if len(sums) == 1 else f"FRAMES DIFFER: {sums}")
