"""The plane Cornell box without any tree: its 36 triangles as a list of one-triangle meshes (no box
entries at all), swept and walked, against the forms the library picks (run on the GPU box)."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
import torch
import raytracer_amd as rt
from oracle import binding as ob
from helpers import asset, product_camera

osc = ob.Scene.cornell_box(asset("cornell_box_plane.obj"))
mats, objs = osc.materials(), osc.objects()
tris = objs[0][1]
flat = [("mesh", tris[i:i + 1].copy()) for i in range(len(tris))]
host = rt.Scene(None, mats, flat, ob.ROOT_LIST, build=rt.BUILD_REFERENCE)
e = host.entries()
print("flat list:", len(e), "entries", flush=True)
host.close()
if not torch.cuda.is_available():
    sys.exit(0)
ctx = rt.Context(0)
w = h = 1024
spp = 256
cam = product_camera(rt, osc, w, h)
p = rt.Scene.params(w, h, spp, 6, osc.sky)
accum = torch.zeros(h * w * 3, dtype=torch.float32, device="cuda")
def run(label, sc):
    best = 1e9
    for i in range(4):
        accum.zero_(); torch.cuda.synchronize()
        st = sc.render_device(cam, p, accum.data_ptr())
        if i: best = min(best, st["ms_render"])
    print(f"{label}: {st['rays_traced'] / best / 1e6:.1f} Grays/s, {best:.2f} ms, form {rt.FORM_NAMES[st['kernel_form']]} tree {rt.TREE_NAMES[st['tree']]}, sum {float(accum.double().sum()):.6e}", flush=True)
sc = rt.Scene(ctx, mats, objs, osc.root_kind); run("default (auto)", sc); sc.close()
sc = rt.Scene(ctx, mats, objs, osc.root_kind, build=rt.BUILD_REFERENCE); sc.set_traversal(rt.TRAVERSAL_SWEEP); run("reference sweep", sc); sc.close()
sc = rt.Scene(ctx, mats, flat, ob.ROOT_LIST, build=rt.BUILD_REFERENCE); sc.set_traversal(rt.TRAVERSAL_SWEEP); run("flat list, sweep", sc); sc.close()
sc = rt.Scene(ctx, mats, flat, ob.ROOT_LIST, build=rt.BUILD_REFERENCE); sc.set_traversal(rt.TRAVERSAL_PER_LANE); run("flat list, per-lane", sc); sc.close()
