"""What the pass is worth to a wave (CPU only): recorded rays of the plane Cornell box walked through the
surface-area tree as built and with the wide-open boxes taken out (the prototype in collapse_eval.py),
groups of 64 walks replayed under the kernel's trip policy (tools/lane_utilisation.py) and priced in issued
vector instructions per ray.
    python profiles/r03_experiments/sah_flatten/wave_replay.py [rays]"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.join(HERE, "..", "..", "..")
for p in (os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools"), ROOT, HERE):
    sys.path.insert(0, p)
import numpy as np

import raytracer_amd as rt
from helpers import asset
from oracle import binding as ob
from lane_utilisation import replay, visit_sequence

sys.argv, argv = sys.argv[:1] + ["none"], sys.argv
from collapse_eval import simple                      # noqa: E402  (its command line is not used here)


def main():
    n_rays = int(argv[1]) if len(argv) > 1 else 3200
    osc = ob.Scene.cornell_box(asset("cornell_box_plane.obj"))
    sc = rt.Scene(None, osc.materials(), osc.objects(), osc.root_kind, build=rt.BUILD_SAH)
    built = sc.entries(raw=True)
    flat, dropped = simple(built, 1.0)
    for kind, rays in (("camera rays of neighbouring pixels", osc.record_rays(256, 256, 1, 1, cap=n_rays)),
                       ("path rays of all depths, in path order", osc.record_rays(64, 64, 4, 6, cap=n_rays))):
        out = []
        for label, ent in (("as built", built), (f"{dropped} boxes taken out", flat)):
            eu = ent.view(np.uint32)
            seqs = [visit_sequence(ent, eu, r["o"], r["d"]) for r in rays]
            issued = useful = 0
            for g in range(0, len(seqs) - 63, 64):
                i, u = replay(seqs[g:g + 64], 8)
                issued += i
                useful += u
            n = (len(seqs) // 64) * 64
            out.append(issued / n)
            print(f"{kind}, {label}: {issued / n:.1f} issued vector instructions per ray, {64 * useful / (64 * issued):.1f} of 64 lanes working",
                  flush=True)
        print(f"{kind}: x{out[0] / out[1]:.3f} fewer issued instructions")
    sc.close()


if __name__ == "__main__":
    main()
