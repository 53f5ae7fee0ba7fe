"""Transformed meshes (Mesh.transform, shape.rs:236-257, 303-307, 452-456; Ray::transform,
camera.rs:137-142) and containers inside containers (shape.rs:55-57, 87-138) on the GPU against the
oracle: primitive id exact, t bit-exact, hit point / normal / facing equal, images within the
Monte-Carlo bound.  Parity here is GPU <-> oracle: glam's Mat4 arithmetic is restated on both
sides from the crate's published source, and the reference ships nothing that could pin it."""
import numpy as np
import pytest

from helpers import adversarial_rays, asset, assert_hits_equal, mirror, product_camera, random_rays, random_scene
from test_gpu_render import _compare
from test_host_layer import _nested_description

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("root", [0, 2])
@pytest.mark.parametrize("with_xforms", [False, True])
def test_nested_and_transformed_hits(rt, ob, ctx, root, with_xforms):
    mats, objects = _nested_description(ob, 71 + root, with_xforms)
    osc = ob.Scene.from_description(mats, objects, root)
    sc = rt.Scene(ctx, mats, objects, root)
    rays = random_rays(ob, 31, 200_000, -5.0, 5.0)
    for t_min, t_max in [(0.001, np.inf), (0.05, 2.0)]:
        want, st = osc.trace(rays, t_min, t_max)
        assert (want["prim_id"] != ob.MISS).sum() > (15_000 if t_max == np.inf else 1_000)
        assert_hits_equal(sc.trace(rays, t_min, t_max), want)
    if with_xforms:
        for mode in (rt.TRAVERSAL_SWEEP, rt.TRAVERSAL_ORDERED, rt.TRAVERSAL_OCTANT):
            with pytest.raises(rt.RtError, match="per-lane walk or the wide walk only"):
                sc.set_traversal(mode)
        # the wide walk: a transformed mesh is a node of its own whose root names the transform
        for build in (rt.BUILD_REFERENCE, rt.BUILD_SAH):
            wide = rt.Scene(ctx, mats, objects, root, build=build)
            wide.set_traversal(rt.TRAVERSAL_WIDE)
            for t_min, t_max in [(0.001, np.inf), (0.05, 2.0)]:
                assert_hits_equal(wide.trace(rays, t_min, t_max), osc.trace(rays, t_min, t_max)[0])
            wide.close()
        dev = rt.Scene(ctx, mats, objects, root)
        dev.set_wide_build(rt.WIDE_BUILD_DEVICE)
        with pytest.raises(rt.RtError, match="device-side build does not take transformed meshes"):
            dev.set_traversal(rt.TRAVERSAL_WIDE)
        dev.close()
        sc.set_traversal(rt.TRAVERSAL_PER_LANE)
        assert_hits_equal(sc.trace(rays[:50_000]), osc.trace(rays[:50_000])[0])
        # counts: the markers are not tests
        import torch
        d_rays = torch.from_numpy(rays[:50_000].view(np.float32).reshape(-1, 6)).cuda()
        d_hits = torch.zeros(50_000 * 10, dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        _, st = osc.trace(rays[:50_000])
        assert sc.trace_device(d_rays.data_ptr(), 50_000, d_hits.data_ptr(), count=True) == \
            (st["box_tests"], st["tri_tests"], st["sph_tests"])
    else:
        from test_gpu_trace_parity import _check_all_modes
        want, _ = osc.trace(rays)
        _check_all_modes(rt, sc, rays, want)
        adv = adversarial_rays(rt, 3, 50_000, 5.0)
        _check_all_modes(rt, sc, adv, osc.trace(adv)[0])


def test_world2_with_translated_planes(rt, ob, ctx):
    """create_world2 with the two planes the reference keeps in a comment (Mesh::plane with
    Mat4::from_translation, main.rs:195-204): host constructor == oracle constructor, hits and image."""
    osc = ob.Scene.world2(asset("thing.obj"), 2)
    w = rt.World.world2(ctx, variant=2)
    rays = np.concatenate([osc.record_rays(128, 96, 4, 6, cap=200_000), random_rays(ob, 5, 50_000, -2.0, 2.0)])
    want, _ = osc.trace(rays)
    on_planes = np.isin(want["prim_id"], [0, 1, 2, 3])
    assert on_planes.sum() > 10_000                                    # the planes are the first four primitives
    assert_hits_equal(w.scene.trace(rays), want)
    assert_hits_equal(mirror(rt, ctx, osc).trace(rays), want)
    w.close()
    _compare(rt, ob, ctx, osc, 96, 96, 48, 6, clip=6.0)


@pytest.mark.parametrize("scene", ["nested", "world2_planes"])
def test_transformed_meshes_in_every_per_lane_form(rt, ob, ctx, scene, monkeypatch, capfd):
    """A scene with transformed meshes is traced by every per-lane form -- from memory or out of LDS,
    plain or with idle lanes refilled, on the reference's trees or the surface-area trees -- and by the
    wide walk, and the frame is the staged launches' frame bit for bit whichever form runs.  world2 with its two translated planes fits LDS and renders from there."""
    if scene == "nested":
        mats, objects = _nested_description(ob, 83, True)
        cam = ((6.0, -7.0, 4.0), (0.0, 0.0, 0.0), (0.0, 0.0, 1.0), 50.0, 0.05, 9.0)
        osc = ob.Scene.from_description(mats, objects, 0, camera=cam, sky=0.8)
    else:
        osc = ob.Scene.world2(asset("thing.obj"), 2)
    w, h, spp = 128, 96, 16
    pc = product_camera(rt, osc, w, h)
    p = rt.Scene.params(w, h, spp, 6, osc.sky)
    rays = osc.record_rays(96, 72, 2, 6, cap=60_000)
    want, _ = osc.trace(rays)
    strict = rt.Scene(ctx, osc.materials(), osc.objects(), osc.root_kind, build=rt.BUILD_REFERENCE)
    staged, st0 = strict.render(pc, rt.Scene.params(w, h, spp, 6, osc.sky, flags=rt.RENDER_STAGED))
    seen = set()
    for build in (rt.BUILD_REFERENCE, rt.BUILD_SAH):
        for lds in (True, False):
            for refill in ("0", "16"):
                monkeypatch.setenv("RT_REFILL", refill)
                if not lds:
                    monkeypatch.setenv("RT_NO_LDS_WALK", "1")
                sc = rt.Scene(ctx, osc.materials(), osc.objects(), osc.root_kind, build=build)
                monkeypatch.delenv("RT_NO_LDS_WALK", raising=False)
                sc.set_traversal(rt.TRAVERSAL_PER_LANE)
                img, st = sc.render(pc, p)
                seen.add((rt.TREE_NAMES[st["tree"]], rt.FORM_NAMES[st["kernel_form"]]))
                assert st["rays_traced"] == st0["rays_traced"] and np.array_equal(img.view(np.uint32), staged.view(np.uint32))
                assert_hits_equal(sc.trace(rays), want)
                sc.close()
        for refill in ("0", "16"):             # the wide walk, plain and refilled
            monkeypatch.setenv("RT_REFILL", refill)
            sc = rt.Scene(ctx, osc.materials(), osc.objects(), osc.root_kind, build=build)
            sc.set_traversal(rt.TRAVERSAL_WIDE)
            img, st = sc.render(pc, p)
            seen.add((rt.TREE_NAMES[st["tree"]], rt.FORM_NAMES[st["kernel_form"]]))
            assert st["rays_traced"] == st0["rays_traced"] and np.array_equal(img.view(np.uint32), staged.view(np.uint32))
            assert_hits_equal(sc.trace(rays), want)
            sc.close()
    monkeypatch.delenv("RT_REFILL")
    assert seen == {(t, f) for t in ("reference", "sah")
                    for f in ("walk", "walk_refill", "walk_lds", "walk_lds_refill", "wide", "wide_refill")}
    monkeypatch.setenv("RT_AUTOTUNE_LOG", "1")
    sc = mirror(rt, ctx, osc)                                  # the defaults
    capfd.readouterr()
    img, st = sc.render(pc, p)
    probes = [l for l in capfd.readouterr().err.splitlines() if l.startswith("[rt probe]")]
    with capfd.disabled():
        print(f"\n{scene}: {len(probes)} forms timed, kept {rt.FORM_NAMES[st['kernel_form']]} on the {rt.TREE_NAMES[st['tree']]} tree")
    # both scenes fit LDS and have transformed meshes: one candidate, the walk out of LDS over the surface-area tree
    assert not probes and rt.FORM_NAMES[st["kernel_form"]] == "walk_lds" and rt.TREE_NAMES[st["tree"]] == "sah"
    assert np.array_equal(img.view(np.uint32), staged.view(np.uint32))
    assert_hits_equal(sc.trace(rays), want)
    sc.close()
    strict.close()


def test_transformed_scene_image_and_staged_launches(rt, ob, ctx):
    mats, objects = _nested_description(ob, 83, True)
    cam = ((6.0, -7.0, 4.0), (0.0, 0.0, 0.0), (0.0, 0.0, 1.0), 50.0, 0.05, 9.0)
    osc = ob.Scene.from_description(mats, objects, 0, camera=cam, sky=0.8)
    _compare(rt, ob, ctx, osc, 96, 72, 48, 6, clip=6.0)
    sc = rt.Scene(ctx, mats, objects, 0)
    pc = product_camera(rt, osc, 96, 72)
    fused, st = sc.render(pc, rt.Scene.params(96, 72, 16, 6, osc.sky, max_paths=96 * 72 * 5))
    staged, st2 = sc.render(pc, rt.Scene.params(96, 72, 16, 6, osc.sky, max_paths=96 * 72 * 5, flags=rt.RENDER_STAGED))
    assert st["rays_traced"] == st2["rays_traced"] and np.array_equal(fused.view(np.uint32), staged.view(np.uint32))


def test_large_transformed_scene_takes_the_wide_walk_by_default(rt, ob, ctx):
    """Transformed meshes too large for LDS: the defaults (both trees, RT_TRAVERSAL_AUTO) render on the
    wide walk over the surface-area trees -- every transformed mesh a node of its own -- and the frame
    is the strict reference-order frame bit for bit, the hits the oracle's."""
    from helpers import random_mesh
    rng = np.random.default_rng(11)
    mats, _, _ = random_scene(ob, 3)
    objects = []
    for k in range(4):
        a = np.float32(0.4 * k + 0.2)
        m = np.eye(4, dtype=np.float32)
        m[:2, :2] = [[np.cos(a), -np.sin(a)], [np.sin(a), np.cos(a)]]
        m[2, 2] = 1.0 + 0.25 * k
        m[:3, 3] = [3.0 * (k % 2) - 1.5, 3.0 * (k // 2) - 1.5, 0.5 * k]
        objects.append(("mesh", random_mesh(ob, rng, 3000, 2.5), m.T.reshape(16).copy()) if k != 3 else
                       ("mesh", random_mesh(ob, rng, 3000, 2.5)))
    sp = np.zeros(30, dtype=ob.sphere_dt)
    sp["c"] = rng.uniform(-6, 6, (30, 3)).astype(np.float32)
    sp["r"] = rng.uniform(0.2, 0.8, 30).astype(np.float32)
    sp["mat"] = rng.integers(0, 6, 30)
    objects.append(("spheres", sp))
    cam = ((9.0, -10.0, 6.0), (0.0, 0.0, 1.0), (0.0, 0.0, 1.0), 45.0, 0.0, 12.0)
    osc = ob.Scene.from_description(mats, objects, ob.ROOT_BVH, camera=cam, sky=0.9)
    w, h, spp = 192, 144, 8
    pc = product_camera(rt, osc, w, h)
    p = rt.Scene.params(w, h, spp, 5, osc.sky)
    strict = rt.Scene(ctx, mats, objects, ob.ROOT_BVH, build=rt.BUILD_REFERENCE)
    strict.set_traversal(rt.TRAVERSAL_PER_LANE)
    want_img, st0 = strict.render(pc, p)
    strict.close()
    sc = rt.Scene(ctx, mats, objects, ob.ROOT_BVH)             # the defaults
    img, st = sc.render(pc, p)
    assert (rt.TREE_NAMES[st["tree"]], rt.FORM_NAMES[st["kernel_form"]]) == ("sah", "wide_refill")
    assert st["rays_traced"] == st0["rays_traced"] and np.array_equal(img.view(np.uint32), want_img.view(np.uint32))
    rays = np.concatenate([osc.record_rays(96, 72, 2, 5, cap=100_000), random_rays(ob, 9, 100_000, -8.0, 8.0)])
    want, _ = osc.trace(rays)
    assert (want["prim_id"] != ob.MISS).sum() > 30_000
    assert_hits_equal(sc.trace(rays), want)
    sc.close()
