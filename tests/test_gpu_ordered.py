"""The opt-in near-child-first walks (RT_TRAVERSAL_ORDERED: per-lane stack; RT_TRAVERSAL_OCTANT: one
pre-ordered copy of the trees per direction octant, no stack; RT_TRAVERSAL_WIDE: the trees collapsed
into eight-child nodes with 8-bit boxes, per-lane stack) against the oracle.

The first two run the reference's box and primitive tests on the reference's trees but visit the nearer
child first, so theirs is not the reference's visiting order; the wide walk runs the reference's
primitive tests behind boxes of its own that enclose the reference's.  The closest hit must still be the
oracle's on every ray: same hit/miss, bit-identical t and hit point, and the SAME PRIMITIVE -- also
where two primitives tie in t, which real scenes do produce (a ray through the shared diagonal of a
wall's two triangles; cornell_box.obj has overlapping coplanar faces): the walks resolve such ties by
primitive key, which is the reference's visiting-order rule made independent of the order."""
import numpy as np
import pytest

from helpers import asset, mirror, random_rays, random_scene

pytestmark = pytest.mark.gpu


@pytest.fixture(params=["ordered", "octant", "wide"])
def walk(request, rt):
    return {"ordered": rt.TRAVERSAL_ORDERED, "octant": rt.TRAVERSAL_OCTANT, "wide": rt.TRAVERSAL_WIDE}[request.param]


def _ordered_vs_oracle(rt, sc, walk, rays, want, t_min=0.001, t_max=np.inf):
    sc.set_traversal(walk)
    got = sc.trace(rays, t_min, t_max)
    sc.set_traversal(rt.TRAVERSAL_AUTO)
    miss_g, miss_w = got["prim_id"] == rt.MISS, want["prim_id"] == rt.MISS
    hit = ~miss_w
    n_t = int((got["t"][hit].view(np.uint32) != want["t"][hit].view(np.uint32)).sum()) + int((miss_g != miss_w).sum())
    n_p = int((got["prim_id"] != want["prim_id"]).sum())
    print(f"near-first walk {walk}: {len(rays)} rays, {n_t} t/miss mismatches, {n_p} primitive mismatches")
    assert n_t == 0 and n_p == 0
    assert np.array_equal(got["p"][hit], want["p"][hit])
    assert np.array_equal(got["front_face"][hit], want["front_face"][hit]) and np.array_equal(got["mat"][hit], want["mat"][hit])
    return got


@pytest.mark.parametrize("name", ["cornell_box_plane.obj", "cornell_box.obj"])
def test_cornell(rt, ob, ctx, walk, name):
    osc = ob.Scene.cornell_box(asset(name))
    sc = mirror(rt, ctx, osc)
    rays = osc.record_rays(96, 96, 4, 8, cap=200000, stride=1)
    want, _ = osc.trace(rays)
    _ordered_vs_oracle(rt, sc, walk, rays, want)


def test_world1_and_world2(rt, ob, ctx, walk):
    for osc in (ob.Scene.world1(), ob.Scene.world2(asset("thing.obj"), 1)):
        sc = mirror(rt, ctx, osc)
        rays = osc.record_rays(128, 96, 4, 6, cap=200000)
        want, _ = osc.trace(rays)
        _ordered_vs_oracle(rt, sc, walk, rays, want)


@pytest.mark.parametrize("root", [0, 2])
def test_random_scenes_and_ranges(rt, ob, ctx, walk, root):
    mats, objects, _ = random_scene(ob, 3, n_spheres=51, n_meshes=3, tris_per_mesh=150)
    osc = ob.Scene.from_description(mats, objects, root)
    sc = rt.Scene(ctx, mats, objects, root)
    rays = random_rays(ob, 13, 150000, -12.0, 12.0)
    for t_min, t_max in [(0.001, np.inf), (0.05, 1.0)]:
        want, _ = osc.trace(rays, t_min, t_max)
        _ordered_vs_oracle(rt, sc, walk, rays, want, t_min, t_max)


def test_degenerate_scenes(rt, ob, ctx, walk):
    from helpers import random_mesh
    mats, _, _ = random_scene(ob, 5)
    rays = random_rays(ob, 1, 5000, -3.0, 3.0)
    rng = np.random.default_rng(0)
    one = np.zeros(1, dtype=ob.sphere_dt)
    one["r"] = 1.0
    cases = [([("spheres", one)], ob.ROOT_BVH), ([("spheres", one)], ob.ROOT_SINGLE), ([], ob.ROOT_BVH),
             ([("mesh", random_mesh(ob, rng, 0))], ob.ROOT_SINGLE)]
    for n in (1, 2, 3, 5):
        cases.append(([("mesh", random_mesh(ob, rng, n, extent=2.0))], ob.ROOT_SINGLE))
    for objects, root in cases:
        osc = ob.Scene.from_description(mats, objects, root)
        sc = rt.Scene(ctx, mats, objects, root)
        want, _ = osc.trace(rays)
        _ordered_vs_oracle(rt, sc, walk, rays, want)


def test_coincident_primitives_keep_the_reference_primitive(rt, ob, ctx, walk):
    from helpers import random_mesh
    mats, _, _ = random_scene(ob, 5)
    rng = np.random.default_rng(4)
    base = random_mesh(ob, rng, 20, extent=2.0)
    dup = np.concatenate([base, base, base[::-1]])
    dup["mat"] = np.arange(len(dup)) % 6
    osc = ob.Scene.from_description(mats, [("mesh", dup)], ob.ROOT_SINGLE)
    sc = rt.Scene(ctx, mats, [("mesh", dup)], ob.ROOT_SINGLE)
    rays = random_rays(ob, 2, 100000, -4.0, 4.0)
    want, _ = osc.trace(rays)
    _ordered_vs_oracle(rt, sc, walk, rays, want)


def test_fewer_visits_and_same_image(rt, ob, ctx, monkeypatch):
    """Never more box tests per ray than the reference order (how many fewer depends on the
    scene: the upper boxes of cornell_box.obj all span the room and cannot be culled in any
    order); and a rendered frame that matches the reference-order frame (same RNG keys, so
    identical unless a hit differs)."""
    import torch
    osc = ob.Scene.cornell_box(asset("cornell_box.obj"))
    sc = mirror(rt, ctx, osc)
    rays = osc.record_rays(64, 64, 2, 8, cap=60000)
    d_rays = torch.from_numpy(rays.view(np.float32).reshape(-1, 6)).cuda()
    d_hits = torch.zeros(len(rays) * 10, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    counts = {}
    hits = {}
    for mode in (rt.TRAVERSAL_PER_LANE, rt.TRAVERSAL_ORDERED, rt.TRAVERSAL_OCTANT):
        sc.set_traversal(mode)
        counts[mode] = sc.trace_device(d_rays.data_ptr(), len(rays), d_hits.data_ptr(), count=True)
        hits[mode] = d_hits.cpu().numpy().copy()
    # the octant copies encode the very choice the ordered walk makes at every box: the same visits,
    # the same hits down to the primitive picked on a tie
    assert counts[rt.TRAVERSAL_OCTANT] == counts[rt.TRAVERSAL_ORDERED]
    assert np.array_equal(hits[rt.TRAVERSAL_OCTANT], hits[rt.TRAVERSAL_ORDERED])
    print("visits per ray, reference order vs ordered:",
          [round(c / len(rays), 2) for c in counts[rt.TRAVERSAL_PER_LANE]],
          [round(c / len(rays), 2) for c in counts[rt.TRAVERSAL_ORDERED]])
    assert counts[rt.TRAVERSAL_ORDERED][0] <= counts[rt.TRAVERSAL_PER_LANE][0]
    from helpers import product_camera
    cam = product_camera(rt, osc, 96, 96)
    p = rt.Scene.params(96, 96, 16, 8, osc.sky)
    sc.set_traversal(rt.TRAVERSAL_PER_LANE)
    ref, st_ref = sc.render(cam, p)
    sc.set_traversal(rt.TRAVERSAL_ORDERED)
    img, st = sc.render(cam, p)
    # the same hits on every ray, ties included: the same paths, the same frame
    assert st["rays_traced"] == st_ref["rays_traced"]
    differing = (img.view(np.uint32) != ref.view(np.uint32)).any(axis=-1).mean()
    print(f"pixels that differ between the two visiting orders: {differing:.2e}")
    assert differing == 0
    # the octant walk, plain and with idle lanes refilled (the scene's first render times both):
    # the ordered walk's frame bit for bit
    sc.set_traversal(rt.TRAVERSAL_AUTO)
    for refill in ("0", "16"):                 # (read when a scene is committed and when it renders)
        monkeypatch.setenv("RT_REFILL", refill)
        sc8 = mirror(rt, ctx, osc)
        sc8.set_traversal(rt.TRAVERSAL_OCTANT)
        img8, st8 = sc8.render(cam, p)
        assert st8["rays_traced"] == st["rays_traced"]
        assert np.array_equal(img8.view(np.uint32), img.view(np.uint32))
    monkeypatch.delenv("RT_REFILL", raising=False)


def test_wide_walk_adversarial_rays_and_frames(rt, ob, ctx, monkeypatch):
    """The wide walk on rays that stress the box test's comparison forms (zero and negative-zero
    direction components, axis-aligned rays, origins on integer planes, tiny and huge directions:
    its fused box test meets 0 * inf and inf - inf there and must stay conservative), on both build
    modes; its visit counts; and its frames against the reference-order walk's (same RNG keys and the
    same hits, so the same frame bit for bit)."""
    from helpers import adversarial_rays, product_camera
    mats, objects, _ = random_scene(ob, 9, n_spheres=37, n_meshes=3, tris_per_mesh=200)
    osc = ob.Scene.from_description(mats, objects, ob.ROOT_BVH)
    rays = np.concatenate([adversarial_rays(rt, 3, 120000, 10.0), random_rays(ob, 4, 60000, -12.0, 12.0)])
    want, _ = osc.trace(rays)
    for build in (rt.BUILD_REFERENCE, rt.BUILD_SAH):
        sc = rt.Scene(ctx, mats, objects, ob.ROOT_BVH, build=build)
        _ordered_vs_oracle(rt, sc, rt.TRAVERSAL_WIDE, rays, want)
    import torch
    for name in ("cornell_box.obj", "cornell_box_plane.obj"):
        osc = ob.Scene.cornell_box(asset(name))
        sc = mirror(rt, ctx, osc)
        rays = osc.record_rays(64, 64, 2, 8, cap=60000)
        d_rays = torch.from_numpy(rays.view(np.float32).reshape(-1, 6)).cuda()
        d_hits = torch.zeros(len(rays) * 10, dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        sc.set_traversal(rt.TRAVERSAL_PER_LANE)
        ref_counts = sc.trace_device(d_rays.data_ptr(), len(rays), d_hits.data_ptr(), count=True)
        sc.set_traversal(rt.TRAVERSAL_WIDE)
        counts = sc.trace_device(d_rays.data_ptr(), len(rays), d_hits.data_ptr(), count=True)
        print(f"{name}: per ray {ref_counts[0] / len(rays):.1f} box + {ref_counts[1] / len(rays):.1f} triangle tests in "
              f"reference order, {counts[0] / len(rays):.1f} wide nodes + {counts[1] / len(rays):.1f} triangle tests")
        assert counts[0] * 2 < ref_counts[0]
        cam = product_camera(rt, osc, 96, 96)
        p = rt.Scene.params(96, 96, 16, 8, osc.sky)
        sc.set_traversal(rt.TRAVERSAL_PER_LANE)
        ref, st_ref = sc.render(cam, p)
        sc.set_traversal(rt.TRAVERSAL_WIDE)
        img, st = sc.render(cam, p)
        assert rt.FORM_NAMES[st["kernel_form"]] == "wide_refill"
        staged, st_s = sc.render(cam, rt.Scene.params(96, 96, 16, 8, osc.sky, flags=rt.RENDER_STAGED))
        assert st_s["rays_traced"] == st["rays_traced"] and np.array_equal(staged.view(np.uint32), img.view(np.uint32))
        monkeypatch.setenv("RT_REFILL", "0")   # (read when the traversal is set): the fused kernel without refilling
        sc.set_traversal(rt.TRAVERSAL_WIDE)
        plain, st_p = sc.render(cam, p)
        monkeypatch.delenv("RT_REFILL")
        assert rt.FORM_NAMES[st_p["kernel_form"]] == "wide"
        assert st_p["rays_traced"] == st["rays_traced"] and np.array_equal(plain.view(np.uint32), img.view(np.uint32))
        assert st["rays_traced"] == st_ref["rays_traced"]
        differing = (img.view(np.uint32) != ref.view(np.uint32)).any(axis=-1).mean()
        print(f"{name}: pixels that differ between the wide walk and the reference order: {differing:.2e}")
        assert differing == 0


def test_wide_tree_built_on_the_gpu(rt, ob, ctx):
    """RT_WIDE_BUILD_DEVICE: the eight-child nodes made on the GPU from the primitives alone (Morton
    sort, runs of equal digit) instead of collapsed from the host's binary trees.  Same walk, same
    bar: hit / miss, t and the primitive as the oracle's on every ray -- twenty copies of one triangle
    included, whose Morton keys are all equal and whose ties only the primitive keys tell apart."""
    from helpers import adversarial_rays, random_mesh
    cases = []
    for name in ("cornell_box_plane.obj", "cornell_box.obj"):
        osc = ob.Scene.cornell_box(asset(name))
        cases.append((osc, osc.record_rays(96, 96, 4, 8, cap=150000, stride=1), True))
    for osc in (ob.Scene.world1(), ob.Scene.world2(asset("thing.obj"), 1)):
        cases.append((osc, osc.record_rays(128, 96, 4, 6, cap=150000), True))
    mats, objects, _ = random_scene(ob, 3, n_spheres=51, n_meshes=3, tris_per_mesh=150)
    rays = np.concatenate([random_rays(ob, 13, 120000, -12.0, 12.0), adversarial_rays(rt, 8, 60000, 10.0)])
    rays = rays[(np.abs(rays["d"]).max(axis=1) < 1e9) & (np.abs(rays["d"]).max(axis=1) > 1e-15)]   # (see the wide walk's rule for other rays)
    cases.append((ob.Scene.from_description(mats, objects, ob.ROOT_BVH), rays, True))
    rng = np.random.default_rng(0)
    one = np.zeros(1, dtype=ob.sphere_dt)
    one["r"] = 1.0
    same = np.repeat(random_mesh(ob, rng, 1, extent=2.0), 20)          # twenty copies of one triangle: equal keys throughout
    few = random_rays(ob, 1, 5000, -3.0, 3.0)
    for objs, root in (([("spheres", one)], ob.ROOT_SINGLE), ([], ob.ROOT_BVH),
                       ([("mesh", random_mesh(ob, rng, 3, extent=2.0))], ob.ROOT_SINGLE), ([("mesh", same)], ob.ROOT_SINGLE)):
        cases.append((ob.Scene.from_description(mats, objs, root), few, True))
    for osc, rays, _ in cases:
        sc = mirror(rt, ctx, osc)
        sc.set_wide_build(rt.WIDE_BUILD_DEVICE)
        want, _ = osc.trace(rays)
        _ordered_vs_oracle(rt, sc, rt.TRAVERSAL_WIDE, rays, want)
        with pytest.raises(rt.RtError):            # the tree exists: where it is made can no longer change
            sc.set_wide_build(rt.WIDE_BUILD_COLLAPSE)
