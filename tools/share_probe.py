import sys, time, os
sys.path.insert(0, os.getcwd())
import torch, numpy as np
import raytracer_amd as rt
ctx = rt.Context(devices=[0]) if os.environ.get("PROBE_MULTI") else rt.Context(0)
w = rt.World.cornell_box(ctx, os.path.join(rt.ASSET_DIR, "cornell_box_plane.obj"))
W = H = 1024
cam = w.camera(W, H)
accum = torch.zeros(H * W * 3, dtype=torch.float32, device="cuda")
for n in (1, 2, 4, 8):
    for flags in (0, rt.RENDER_PROFILE):
        p = rt.partition_params(w.params(W, H, 1024, 6, flags=flags), n, 0)
        for it in range(4):
            accum.zero_(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            st = w.scene.render_device(cam, p, accum.data_ptr())
            dt = (time.perf_counter() - t0) * 1e3
        print(f"share 0/{n} flags {flags}: wall {dt:.2f} ms, ms_render {st['ms_render']:.2f}, launches {st['trace_launches']}, "
              f"rays {st['rays_traced']}, ideal {290.0 / n:.1f}", flush=True)
# the same share timed the way bench.py's step does: the clear of the frame and the wait for it inside the clock
p = rt.partition_params(w.params(W, H, 1024, 6, flags=rt.RENDER_PROFILE), 8, 0)
for label in ("zero + stream sync inside", "zero + device sync inside", "render only"):
    torch.cuda.synchronize()
    ts = []
    for it in range(6):
        if label == "render only":
            accum.zero_(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        if label == "zero + stream sync inside":
            accum.zero_(); torch.cuda.current_stream().synchronize()
        elif label == "zero + device sync inside":
            accum.zero_(); torch.cuda.synchronize()
        st = w.scene.render_device(cam, p, accum.data_ptr())
        ts.append((time.perf_counter() - t0) * 1e3)
    print(label, [round(t, 2) for t in ts], flush=True)
