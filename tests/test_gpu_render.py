"""Image-level parity of the wavefront renderer against the oracle's tile renderer.

The GPU uses a counter-based RNG, so images are compared statistically: the RMSE between the
GPU image and an oracle image at matched spp must not exceed the RMSE between two oracle images
rendered with different seeds (the Monte-Carlo noise floor) by more than 10 %, on the linear
per-pixel mean radiance and on the tonemapped [0,1] values alike, and the image means must agree
within 2 %.  Bound: RMSE(gpu, oracle) <= 1.10 * RMSE(oracle, oracle')."""
import numpy as np
import pytest

from helpers import asset, mirror, product_camera, random_scene

pytestmark = pytest.mark.gpu


def _rmse(a, b):
    return float(np.sqrt(np.mean((a.astype(np.float64) - b.astype(np.float64)) ** 2)))


def _compare(rt, ob, ctx, osc, w, h, spp, depth, clip=None):
    sc = mirror(rt, ctx, osc)
    cam = product_camera(rt, osc, w, h)
    a, _, st_a = osc.render(w, h, spp, depth, seed_offset=0)
    b, _, _ = osc.render(w, h, spp, depth, seed_offset=7919)
    p = rt.Scene.params(w, h, spp, depth, osc.sky)
    g, st = sc.render(cam, p)
    a, b, g = a / spp, b / spp, g / spp
    if clip is not None:                        # tame fireflies so the RMSE is a stable statistic
        a, b, g = (np.minimum(x, clip) for x in (a, b, g))
    floor = _rmse(a, b)
    err = max(_rmse(g, a), _rmse(g, b))
    mean_o = 0.5 * (a.mean() + b.mean())
    assert np.isfinite(g).all()
    print(f"rmse {err:.5f} floor {floor:.5f} ratio {err / floor:.4f}")
    assert err <= 1.10 * floor, f"rmse {err:.5f} vs noise floor {floor:.5f}"
    # the same on the displayed values: sqrt tonemap, clamp at 1 (main.rs:369-373)
    ta, tb, tg = (np.minimum(np.sqrt(np.maximum(x, 0.0)), 1.0) for x in (a, b, g))
    tfloor, terr = _rmse(ta, tb), max(_rmse(tg, ta), _rmse(tg, tb))
    print(f"tonemapped rmse {terr:.5f} floor {tfloor:.5f} ratio {terr / tfloor:.4f}")
    assert terr <= 1.10 * tfloor, f"tonemapped rmse {terr:.5f} vs noise floor {tfloor:.5f}"
    assert abs(g.mean() - mean_o) <= 0.02 * mean_o, f"mean {g.mean():.5f} vs {mean_o:.5f}"
    # same estimator => same expected ray count per path
    assert abs(st["rays_traced"] / st["paths"] - st_a["rays"] / st_a["paths"]) < 0.03
    assert st["paths"] == w * h * spp
    return g, a


def test_cornell_plane_image(rt, ob, ctx):
    osc = ob.Scene.cornell_box(asset("cornell_box_plane.obj"))
    _compare(rt, ob, ctx, osc, 128, 128, 64, 6, clip=4.0)


def test_cornell_full_image_depth8(rt, ob, ctx):
    osc = ob.Scene.cornell_box(asset("cornell_box.obj"))
    _compare(rt, ob, ctx, osc, 96, 96, 32, 8, clip=4.0)


def test_world1_image(rt, ob, ctx):
    _compare(rt, ob, ctx, ob.Scene.world1(), 160, 90, 48, 6, clip=4.0)


@pytest.mark.parametrize("variant", [0, 1])
def test_world2_image(rt, ob, ctx, variant):
    _compare(rt, ob, ctx, ob.Scene.world2(asset("thing.obj"), variant), 96, 96, 48, 6, clip=6.0)


def test_deterministic_and_partition_invariant(rt, ob, ctx):
    osc = ob.Scene.world2(asset("thing.obj"), 1)
    sc = mirror(rt, ctx, osc)
    w, h, spp, depth = 100, 77, 12, 6          # not multiples of 8: padded tiles
    cam = product_camera(rt, osc, w, h)
    full, st = sc.render(cam, rt.Scene.params(w, h, spp, depth, osc.sky))
    again, st2 = sc.render(cam, rt.Scene.params(w, h, spp, depth, osc.sky))
    assert np.array_equal(full.view(np.uint32), again.view(np.uint32))
    assert st["rays_traced"] == st2["rays_traced"]
    # pixel rectangles (8-aligned cuts keep the same tiles): bit-identical
    acc = np.zeros_like(full)
    total_rays = 0
    for rect in [(0, 0, 48, 40), (48, 0, 100, 40), (0, 40, 100, 77)]:
        acc, s = sc.render(cam, rt.Scene.params(w, h, spp, depth, osc.sky, rect=rect), accum=acc)
        total_rays += s["rays_traced"]
    assert np.array_equal(full.view(np.uint32), acc.view(np.uint32))
    assert total_rays == st["rays_traced"]
    # sample ranges: same samples, different summation grouping
    acc = np.zeros_like(full)
    for rng_ in [(0, 5), (5, 12)]:
        acc, _ = sc.render(cam, rt.Scene.params(w, h, spp, depth, osc.sky, samples=rng_), accum=acc)
    np.testing.assert_allclose(acc, full, rtol=2e-5, atol=1e-6)
    # interleaved row bands (how ranks split a frame by tile): two band groups tile the frame
    acc = np.zeros_like(full)
    rays = 0
    for idx in (0, 1):
        acc, s = sc.render(cam, rt.Scene.params(w, h, spp, depth, osc.sky, bands=(16, 2, idx)), accum=acc)
        rays += s["rays_traced"]
        assert s["paths"] == sum(min(16, h - y0) for y0 in range(16 * idx, h, 32)) * w * spp
    np.testing.assert_allclose(acc, full, rtol=2e-5, atol=1e-6)
    assert rays == st["rays_traced"]
    only0, _ = sc.render(cam, rt.Scene.params(w, h, spp, depth, osc.sky, bands=(16, 2, 0)))
    assert (only0[16:32] == 0).all() and (only0[0:16].sum() > 0) and (only0[32:48].sum() > 0)
    with pytest.raises(rt.RtError):
        sc.render(cam, rt.Scene.params(w, h, spp, depth, osc.sky, bands=(12, 2, 0)))
    with pytest.raises(rt.RtError):
        sc.render(cam, rt.Scene.params(w, h, spp, depth, osc.sky, bands=(16, 2, 2)))
    # small batches (many wavefront batches) give the same image as one big batch
    small, _ = sc.render(cam, rt.Scene.params(w, h, spp, depth, osc.sky, max_paths=64 * 40))
    np.testing.assert_allclose(small, full, rtol=2e-5, atol=1e-6)
    # a different seed gives a different image
    other, _ = sc.render(cam, rt.Scene.params(w, h, spp, depth, osc.sky, seed=12345))
    assert not np.array_equal(other, full)


def test_traversal_modes_render_identical_images(rt, ob, ctx):
    """Per-lane walk and wave sweep give bit-identical hits, hence bit-identical images."""
    osc = ob.Scene.world2(asset("thing.obj"), 1)
    sc = mirror(rt, ctx, osc)
    cam = product_camera(rt, osc, 96, 64)
    imgs = []
    for mode in (rt.TRAVERSAL_PER_LANE, rt.TRAVERSAL_SWEEP):
        sc.set_traversal(mode)
        img, st = sc.render(cam, rt.Scene.params(96, 64, 8, 6, osc.sky, flags=rt.RENDER_COUNT_VISITS))
        imgs.append((img, st))
    assert np.array_equal(imgs[0][0].view(np.uint32), imgs[1][0].view(np.uint32))
    for k in ("rays_traced", "box_tests", "tri_tests", "sph_tests"):
        assert imgs[0][1][k] == imgs[1][1][k]


def test_depth_one_and_counts(rt, ob, ctx):
    osc = ob.Scene.cornell_box(asset("cornell_box_plane.obj"))
    sc = mirror(rt, ctx, osc)
    cam = product_camera(rt, osc, 64, 64)
    p = rt.Scene.params(64, 64, 4, 1, osc.sky, flags=rt.RENDER_COUNT_VISITS | rt.RENDER_PROFILE)
    img, st = sc.render(cam, p)
    assert st["rays_traced"] == st["paths"] == 64 * 64 * 4
    assert st["trace_launches"] == 1 and st["box_tests"] > 0 and st["tri_tests"] > 0
    assert st["ms_trace"] > 0
    # depth 1: only emitters and sky are visible; the light is 10 => pixel sums are 0 or 4*10
    assert img.max() <= 40.0 + 1e-3


def test_tonemap_matches_oracle(rt, ob, ctx):
    rng = np.random.default_rng(0)
    acc = (rng.uniform(0, 3, (5000, 3)) ** 3).astype(np.float32)
    acc[0] = [0, np.nan, np.inf]
    acc[1] = [-1.0, 1e-30, 16.0]
    want = ob.tonemap(acc, 16)
    got = ctx.tonemap(acc, 16)
    assert np.array_equal(got, want)


def test_world_render_entry_point(rt, ob, ctx):
    """rt_world_render (the render() drop-in of the host layer) equals the same render issued
    through the core ABI on a mirrored scene."""
    w = rt.World.world2(ctx, variant=1)
    img, st = w.render(80, 64, 6, 6)
    cam = w.camera(80, 64)
    img2, st2 = w.scene.render(cam, w.params(80, 64, 6, 6))
    assert np.array_equal(img.view(np.uint32), img2.view(np.uint32))
    assert st["rays_traced"] == st2["rays_traced"] and st["paths"] == 80 * 64 * 6
    osc = ob.Scene.world2(asset("thing.obj"), 1)
    sc = mirror(rt, ctx, osc)
    img3, _ = sc.render(product_camera(rt, osc, 80, 64), rt.Scene.params(80, 64, 6, 6, osc.sky))
    assert np.array_equal(img.view(np.uint32), img3.view(np.uint32))
    w.close()


def test_soup_scene_hits_match_oracle(rt, ob, ctx):
    """The synthetic triangle soup + sphere field (the bandwidth stress scene) at a small size:
    host generator -> GPU scene, same description -> oracle; hits must agree."""
    from helpers import assert_hits_equal, random_rays
    w = rt.World.soup(ctx, 20000, grid=8, seed=12345)
    osc = ob.Scene.from_description(w.materials(), w.objects(), w.root_kind)
    assert w.scene.num_prims == 20000 + 64 + 1
    rays = random_rays(ob, 9, 200000, -100.0, 100.0)
    rays["o"][:, 2] = np.abs(rays["o"][:, 2])
    want, _ = osc.trace(rays)
    assert (want["prim_id"] != ob.MISS).sum() > 1000
    assert_hits_equal(w.scene.trace(rays), want)
    w.close()


def test_short_reciprocal_is_exact_on_this_gpu():
    """tools/rcp_check compares the triangle test's short reciprocal with the correctly rounded
    1/x over every f32 bit pattern; it exits non-zero on any in-range mismatch."""
    import os
    import subprocess
    from helpers import ROOT
    exe = os.path.join(ROOT, "tools", "rcp_check")
    assert os.path.exists(exe), "build() compiles tools/rcp_check"
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "mismatches in range 0" in out.stdout


def test_one_kernel_per_batch_equals_staged_launches(rt, ob, ctx):
    """The default render runs one kernel per batch that carries every segment from raygen to the
    last bounce; RENDER_STAGED launches the stages one by one.  Same rays, same order, same sums:
    the frames must be bit-identical, for the sweep and for the per-lane walk, with several
    batches in flight."""
    for osc, size in ((ob.Scene.cornell_box(asset("cornell_box_plane.obj")), (72, 56)),
                      (ob.Scene.world2(asset("thing.obj"), 1), (64, 48)), (ob.Scene.world1(), (64, 40))):
        sc = mirror(rt, ctx, osc)
        w, h = size
        cam = product_camera(rt, osc, w, h)
        fused, st = sc.render(cam, rt.Scene.params(w, h, 24, 6, osc.sky, max_paths=w * h * 5))
        staged, st2 = sc.render(cam, rt.Scene.params(w, h, 24, 6, osc.sky, max_paths=w * h * 5, flags=rt.RENDER_STAGED))
        assert st["rays_traced"] == st2["rays_traced"] and st["paths"] == st2["paths"]
        assert st["trace_launches"] < st2["trace_launches"]
        assert np.array_equal(fused.view(np.uint32), staged.view(np.uint32))


def test_refilled_walk_renders_the_same_frame(rt, ob, ctx, monkeypatch):
    """The per-lane walk that hands idle lanes the next rays of the segment (chosen by the first
    render's timing on big scenes; forced here through RT_REFILL, which a scene reads when it is
    committed) changes which lane walks a ray and when, nothing else: the frame must equal the
    staged frame bit for bit, from memory (cornell_box.obj) and out of LDS (thing.obj), with long
    segments (one batch) and short ones (several batches), on the reference and the SAH tree."""
    for osc, size, spp in ((ob.Scene.cornell_box(asset("cornell_box.obj")), (96, 80), 48),
                           (ob.Scene.world2(asset("thing.obj"), 1), (64, 48), 24)):
        w, h = size
        cam = product_camera(rt, osc, w, h)
        for build in (rt.BUILD_REFERENCE, rt.BUILD_SAH):
            monkeypatch.delenv("RT_REFILL", raising=False)
            plain = rt.Scene(ctx, osc.materials(), osc.objects(), osc.root_kind, build=build)
            plain.set_traversal(rt.TRAVERSAL_PER_LANE)
            scenes = []
            for refill in ("8", "40"):
                monkeypatch.setenv("RT_REFILL", refill)
                sc = rt.Scene(ctx, osc.materials(), osc.objects(), osc.root_kind, build=build)
                sc.set_traversal(rt.TRAVERSAL_PER_LANE)
                scenes.append(sc)
            # (a pixel's samples are summed batch by batch, so frames are compared at equal batching)
            for max_paths in (0, w * h * 7):
                staged, st0 = plain.render(cam, rt.Scene.params(w, h, spp, 8, osc.sky, max_paths=max_paths,
                                                                flags=rt.RENDER_STAGED))
                for sc in scenes:
                    img, st = sc.render(cam, rt.Scene.params(w, h, spp, 8, osc.sky, max_paths=max_paths))
                    assert st["rays_traced"] == st0["rays_traced"]
                    assert np.array_equal(img.view(np.uint32), staged.view(np.uint32))
    monkeypatch.delenv("RT_REFILL", raising=False)


@pytest.mark.parametrize("top", ["2048", "200"])
def test_split_walk_renders_the_same_frame(rt, ob, ctx, monkeypatch, top):
    """The split walk (an experiment, RT_SPLIT = lanes a fetch from memory waits for): the top of the
    trees is read from a copy in LDS and the fetches from memory are batched.  Which lane gets its next
    entry when changes, nothing else: the frame equals the staged frame bit for bit, with a large and
    a small top, on the reference and the SAH tree."""
    monkeypatch.setenv("RT_LAYOUT_TOP", top)
    osc = ob.Scene.cornell_box(asset("cornell_box.obj"))
    w, h, spp = 96, 80, 48
    cam = product_camera(rt, osc, w, h)
    for build in (rt.BUILD_REFERENCE, rt.BUILD_SAH):
        monkeypatch.delenv("RT_REFILL", raising=False)
        monkeypatch.delenv("RT_SPLIT", raising=False)
        plain = rt.Scene(ctx, osc.materials(), osc.objects(), osc.root_kind, build=build)
        plain.set_traversal(rt.TRAVERSAL_PER_LANE)
        scenes = []
        for split, refill in (("1", "16"), ("8", "16"), ("24", "8"), ("64", "40")):
            monkeypatch.setenv("RT_REFILL", refill)
            monkeypatch.setenv("RT_SPLIT", split)
            sc = rt.Scene(ctx, osc.materials(), osc.objects(), osc.root_kind, build=build)
            sc.set_traversal(rt.TRAVERSAL_PER_LANE)
            scenes.append(sc)
        for max_paths in (0, w * h * 7):
            staged, st0 = plain.render(cam, rt.Scene.params(w, h, spp, 8, osc.sky, max_paths=max_paths, flags=rt.RENDER_STAGED))
            for sc in scenes:
                img, st = sc.render(cam, rt.Scene.params(w, h, spp, 8, osc.sky, max_paths=max_paths))
                assert st["rays_traced"] == st0["rays_traced"]
                assert np.array_equal(img.view(np.uint32), staged.view(np.uint32))
    monkeypatch.delenv("RT_REFILL", raising=False)
    monkeypatch.delenv("RT_SPLIT", raising=False)


def test_progressive_render_delivers_the_reference_block_events(rt, ob, ctx):
    """rt_world_render_progressive: steps 10, 20, 40, 50 of a 50-spp frame; after each, every 64x64
    block (edge blocks clipped) in j-then-i order, tonemapped with the samples so far, with red
    corner marks while unfinished (main.rs:355-411).  The final sums equal a one-call render up to
    f32 summation grouping."""
    wld = rt.World.cornell_box(ctx, asset("cornell_box_plane.obj"))
    w, h, spp = 150, 100, 50                               # 3 x 2 blocks, clipped on both edges
    events = []
    acc, st = wld.render_progressive(w, h, spp, 6, lambda *e: events.append(e) and False)
    assert st["paths"] == w * h * spp
    steps = rt.progressive_schedule(spp)
    blocks = [(x, y, min(64, w - x), min(64, h - y)) for y in range(0, h, 64) for x in range(0, w, 64)]
    assert [(e[0], e[1], e[2], e[3], e[5]) for e in events] == [b + (s,) for s in steps for b in blocks]
    assert all(e[6] == spp for e in events)
    final = ctx.tonemap(acc, spp).reshape(h, w)
    red = 0x00FF0000
    for x, y, bw, bh, blk, done, _ in events:
        assert blk.shape == (bh, bw)
        corners = blk[0, 0] == red and blk[bh - 1, bw - 1] == red and blk[0, bw - 1] == red and blk[bh - 1, 0] == red
        if done < spp:
            assert corners and (blk[0, :5] == red).all() and (blk[:5, 0] == red).all()
            assert (blk[1:-1, 1:-1] != red).any()
        else:
            assert np.array_equal(blk, final[y:y + bh, x:x + bw])
    one, st1 = wld.render(w, h, spp, 6)
    assert st1["rays_traced"] == st["rays_traced"]
    np.testing.assert_allclose(acc, one, rtol=2e-5, atol=1e-6)
    # a callback that returns True stops after the step it is in
    seen = []
    wld.render_progressive(w, h, spp, 6, lambda *e: seen.append(e[5]) or True)
    assert set(seen) == {10} and len(seen) == len(blocks)


def test_depth_extremes(rt, ob, ctx):
    """max_bounces = 1: one closest-hit query per path and only emitted / sky radiance survives
    (main.rs:41-43 returns black for the next bounce); max_bounces = 64 (the limit): paths run
    until they leave or are absorbed, ray counts agree with the oracle's."""
    osc = ob.Scene.world1()
    sc = mirror(rt, ctx, osc)
    cam = product_camera(rt, osc, 64, 48)
    g, st = sc.render(cam, rt.Scene.params(64, 48, 16, 1, osc.sky))
    assert st["rays_traced"] == st["paths"] == 64 * 48 * 16
    a, _, st_a = osc.render(64, 48, 16, 1)
    assert st_a["rays"] == st_a["paths"]
    assert abs(g.mean() - a.mean()) <= 0.05 * a.mean() + 1e-6
    _compare(rt, ob, ctx, ob.Scene.cornell_box(asset("cornell_box_plane.obj")), 96, 96, 64, 64, clip=4.0)
    with pytest.raises(rt.RtError):
        sc.render(cam, rt.Scene.params(64, 48, 1, 65, osc.sky))
    with pytest.raises(rt.RtError):
        sc.render(cam, rt.Scene.params(64, 48, 1, 0, osc.sky))


@pytest.mark.parametrize("root", [0, 2])
def test_random_scene_image(rt, ob, ctx, root):
    """Every material kind at once (lambert with and without normal shading, metal with fuzz 0.3 and
    1.0, glass, emitters), spheres and meshes mixed, thin-lens camera with a real aperture, Bvh and
    list roots: image statistics against the oracle."""
    mats, objects, _ = random_scene(ob, 40 + root, n_spheres=60, n_meshes=3, tris_per_mesh=80, extent=6.0)
    cam = ((0.0, -16.0, 3.0), (0.0, 0.0, 0.0), (0.0, 0.0, 1.0), 50.0, 0.4, 16.0)
    osc = ob.Scene.from_description(mats, objects, root, camera=cam, sky=0.8)
    _compare(rt, ob, ctx, osc, 112, 80, 48, 8, clip=6.0)


def test_headless_main_writes_the_library_frame(rt, ob, ctx, tmp_path):
    """raytracer_amd/rt_render (examples/rt_render_main.cpp, plain C++ on the C ABI) renders the
    reference's default job at a small size; its PPM must hold the pixels the same calls give
    through the Python stub, and the progressive run (blocks after 10, 20, 40 ... samples, copied
    into a framebuffer as the reference's event loop does) the same frame up to the last bit of
    8-bit rounding (its samples are summed step by step)."""
    import os
    import subprocess
    from helpers import ROOT
    exe = os.path.join(ROOT, "raytracer_amd", "rt_render")
    w, h, spp, depth = 96, 80, 50, 6
    world = rt.World.cornell_box(ctx, asset("cornell_box.obj"))
    img, st = world.render(w, h, spp, depth)
    want = ctx.tonemap(img, spp).reshape(h, w)
    want_rgb = np.stack([(want >> 16) & 255, (want >> 8) & 255, want & 255], axis=-1).astype(np.uint8)
    world.close()

    def run(*extra, same_rays=True):
        out = tmp_path / "frame.ppm"
        p = subprocess.run([exe, "--scene", "cornell", "--obj", asset("cornell_box.obj"), "--size", f"{w}x{h}",
                            "--spp", str(spp), "--depth", str(depth), "--out", str(out), *extra],
                           capture_output=True, text=True, timeout=120)
        assert p.returncode == 0, p.stdout + p.stderr
        assert "Finished in" in p.stdout
        rays = int(p.stdout.split(" rays,")[0].split()[-1])
        assert rays == st["rays_traced"] if same_rays else abs(rays - st["rays_traced"]) <= 1e-3 * st["rays_traced"]
        raw = out.read_bytes()
        header = f"P6\n{w} {h}\n255\n".encode()
        assert raw.startswith(header) and len(raw) == len(header) + w * h * 3
        return np.frombuffer(raw[len(header):], dtype=np.uint8).reshape(h, w, 3), p.stdout

    got, _ = run()
    assert np.array_equal(got, want_rgb)
    # the wide walk through the C ABI (its tree collapsed on the host, or built on the GPU): the same
    # rays and the same frame
    for flag in ("--wide", "--wide-device"):
        wide, _ = run(flag)
        assert np.array_equal(wide, want_rgb)
    prog, text = run("--progressive")
    assert "block events" in text
    assert np.abs(prog.astype(int) - want_rgb.astype(int)).max() <= 1


def test_frame_against_the_reference_screenshot(rt, ob, ctx):
    """The reference ships one rendered frame: screenshot.png, its window after the default job has
    finished (cornell_box.obj, 400 x 400, 400 samples per pixel, 6 bounces: main.rs:20-24, 322).  The
    same job on the GPU, tonemapped and packed as the reference does (main.rs:369-373), must be that
    picture up to Monte-Carlo noise at matched spp:
      per-pixel RMSE(gpu, screenshot) <= 1.10 x RMSE(gpu, gpu with another seed)   (stated bound),
      means of 20 x 20-pixel blocks within 0.005 RMS and 0.02 at worst, image means within 0.002
    (values in [0, 1]; measured: 0.0456 vs a floor of 0.0456, 0.0023, 0.0099, 0.0002)."""
    from helpers import block_means, reference_screenshot, unpack_0rgb
    shot = reference_screenshot()
    world = rt.World.cornell_box(ctx, asset("cornell_box.obj"))
    cam = world.camera(400, 400)
    frames = []
    for seed in (rt.DEFAULT_SEED, 2024):
        acc, st = world.scene.render(cam, world.params(400, 400, 400, 6, seed=seed))
        frames.append(unpack_0rgb(ctx.tonemap(acc, 400), 400, 400))
    world.close()
    rmse = lambda a, b: float(np.sqrt(((a - b) ** 2).mean()))
    floor = rmse(frames[0], frames[1])
    err = max(rmse(frames[0], shot), rmse(frames[1], shot))
    d = block_means(shot, 20) - block_means(frames[0], 20)
    print(f"gpu vs the reference's screenshot at 400 spp: per-pixel rmse {err:.4f} (gpu vs gpu {floor:.4f}), "
          f"block rmse {np.sqrt((d ** 2).mean()):.4f}, worst block {np.abs(d).max():.4f}, mean difference {d.mean():+.5f}")
    assert err <= 1.10 * floor
    assert np.sqrt((d ** 2).mean()) < 0.005 and np.abs(d).max() < 0.02 and abs(d.mean()) < 0.002
