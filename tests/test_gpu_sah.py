"""The opt-in SAH build of mesh trees (RT_BUILD_SAH) against the oracle.

Same entry format, same box padding, same box and primitive tests, one triangle per leaf; only
where nodes are split differs from the reference.  The closest hit must still be the oracle's on
every ray: same hit/miss, bit-identical t and hit point, and the same primitive, exact ties in t
included (they are resolved by primitive key, the reference's visiting-order rule made independent
of the order)."""
import numpy as np
import pytest

from helpers import adversarial_rays, asset, mirror, product_camera, random_rays, random_scene

pytestmark = pytest.mark.gpu


def _sah(rt, ctx, osc):
    return rt.Scene(ctx, osc.materials(), osc.objects(), osc.root_kind, build=rt.BUILD_SAH)


def _check(rt, sc, rays, want, label, ordered=False):
    """(leaves of several triangles rule out the ordered walk's stack; the octant walk visits the same
    entries in the same order without one)"""
    for mode in (rt.TRAVERSAL_PER_LANE, rt.TRAVERSAL_SWEEP if sc.num_entries <= 4096 else rt.TRAVERSAL_PER_LANE,
                 rt.TRAVERSAL_ORDERED if ordered else rt.TRAVERSAL_OCTANT):
        sc.set_traversal(mode)
        got = sc.trace(rays)
        nan = np.isnan(want["t"])
        miss_g, miss_w = got["prim_id"] == rt.MISS, want["prim_id"] == rt.MISS
        hit = ~miss_w & ~nan
        assert np.array_equal(miss_g, miss_w)
        assert np.array_equal(np.isnan(got["t"]), nan)
        n_t = int((got["t"][hit].view(np.uint32) != want["t"][hit].view(np.uint32)).sum())
        n_p = int((got["prim_id"] != want["prim_id"]).sum())
        shading = ((got["n"] != want["n"]).any(axis=-1) | (got["front_face"] != want["front_face"])
                   | (got["mat"] != want["mat"]))[hit]
        print(f"SAH tree, {label}, mode {mode}: {len(rays)} rays, {n_t} t mismatches, {n_p} primitive mismatches, "
              f"{int(shading.sum())} with a different normal / facing / material")
        assert n_t == 0 and n_p == 0
        fin = hit & np.isfinite(want["p"]).all(axis=-1)
        assert np.array_equal(got["p"][fin], want["p"][fin])
        assert shading.sum() == 0
    sc.set_traversal(rt.TRAVERSAL_AUTO)


@pytest.mark.parametrize("name", ["cornell_box_plane.obj", "cornell_box.obj"])
def test_cornell(rt, ob, ctx, name):
    osc = ob.Scene.cornell_box(asset(name))
    sc = _sah(rt, ctx, osc)
    rays = osc.record_rays(96, 96, 4, 8, cap=200000, stride=1)
    want, _ = osc.trace(rays)
    _check(rt, sc, rays, want, name)


def test_world2_and_random_scenes(rt, ob, ctx, monkeypatch):
    osc = ob.Scene.world2(asset("thing.obj"), 1)
    rays = osc.record_rays(128, 96, 4, 6, cap=200000)
    want, _ = osc.trace(rays)
    _check(rt, _sah(rt, ctx, osc), rays, want, "thing.obj", ordered=True)
    for leaf in ("2", "4", "8"):               # leaves of several triangles (opt-in: RT_SAH_LEAF)
        monkeypatch.setenv("RT_SAH_LEAF", leaf)
        sc = _sah(rt, ctx, osc)
        _check(rt, sc, rays, want, f"thing.obj, leaves of {leaf}", ordered=leaf == "2")
        if leaf != "2":
            with pytest.raises(rt.RtError, match="at most two children"):
                sc.set_traversal(rt.TRAVERSAL_ORDERED)
    monkeypatch.delenv("RT_SAH_LEAF")
    for root in (0, 2):
        mats, objects, _ = random_scene(ob, 17 + root, n_spheres=40, n_meshes=3, tris_per_mesh=300)
        osc = ob.Scene.from_description(mats, objects, root)
        sc = rt.Scene(ctx, mats, objects, root, build=rt.BUILD_SAH)
        for rays in (random_rays(ob, 23, 150000, -12.0, 12.0), adversarial_rays(rt, 29, 60000, 12.0)):
            want, _ = osc.trace(rays)
            _check(rt, sc, rays, want, f"random scene, root {root}")


def test_fewer_box_tests_and_same_image(rt, ob, ctx):
    """cornell_box.obj mixes room-sized and millimetre triangles: the SAH tree must need well under
    half of the reference tree's box tests per ray, and the rendered frame must be the
    reference-tree frame bit for bit."""
    import torch
    osc = ob.Scene.cornell_box(asset("cornell_box.obj"))
    ref, sah = mirror(rt, ctx, osc), _sah(rt, ctx, osc)
    rays = osc.record_rays(64, 64, 2, 8, cap=60000)
    d_rays = torch.from_numpy(rays.view(np.float32).reshape(-1, 6)).cuda()
    d_hits = torch.zeros(len(rays) * 10, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    counts = [s.trace_device(d_rays.data_ptr(), len(rays), d_hits.data_ptr(), count=True) for s in (ref, sah)]
    print("box / triangle tests per ray, reference tree vs SAH tree:",
          [round(c / len(rays), 2) for c in counts[0][:2]], [round(c / len(rays), 2) for c in counts[1][:2]])
    assert counts[1][0] < 0.5 * counts[0][0]
    cam = product_camera(rt, osc, 96, 96)
    p = rt.Scene.params(96, 96, 16, 8, osc.sky)
    a, st_a = ref.render(cam, p)
    b, st_b = sah.render(cam, p)
    assert st_a["rays_traced"] == st_b["rays_traced"]
    differing = (a.view(np.uint32) != b.view(np.uint32)).any(axis=-1).mean()
    print(f"pixels that differ between the two trees: {differing:.2e}")
    assert differing == 0
