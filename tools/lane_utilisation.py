"""Where the lanes of the per-lane walk go idle (CPU only): walks recorded path rays of the plane Cornell box
through the flattened surface-area tree in Python, then replays groups of 64 walks under the kernel's
trip policy (lanes at a box step together; lanes at a primitive wait until `prim_wait` of them do, or
no lane stands at a box) and prices a box trip and a primitive routine at their vector instruction counts.
    python tools/lane_utilisation.py [rays] [prim_wait]
Prints the share of issued lane-slots that did work, and how much of the loss is walk length (a wave runs
until its longest walk ends) and how much is lanes standing at the other kind of entry."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np

import raytracer_amd as rt
from helpers import asset
from oracle import binding as ob
from test_host_layer import F, _box_miss, _prim_test

BOX_COST, PRIM_COST = 45, 75           # vector instructions of a box trip / a triangle routine (hipcc -S)


def visit_sequence(entries, ent_u, o, d, t_min=F(0.001)):
    """'b' / 'p' per visited entry, in order (the reference's narrowing of t_max included)."""
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        o, d = [F(x) for x in o], [F(x) for x in d]
        inv = [F(1.0) / d[k] for k in range(3)]
        best_t, best = F(np.inf), 0xFFFFFFFF
        i, n, seq = 0, len(entries), []
        while i < n:
            w = int(ent_u[i, 3])
            if w >> 30 == 0:
                seq.append("b")
                i = (w & 0x3FFFFFFF) if _box_miss(entries[i], o, inv, t_min, best_t) else i + 1
            else:
                seq.append("p")
                best_t, best = _prim_test(entries[i], w >> 30, int(ent_u[i, 7]), o, d, t_min, best_t, best)
                i += 1
    return "".join(seq)


def replay(seqs, prim_wait):
    """(issued instruction slots, useful lane-instructions) of one wave under the trip policy"""
    pos = [0] * len(seqs)
    issued = useful = 0
    while True:
        at_b = [l for l in range(len(seqs)) if pos[l] < len(seqs[l]) and seqs[l][pos[l]] == "b"]
        at_p = [l for l in range(len(seqs)) if pos[l] < len(seqs[l]) and seqs[l][pos[l]] == "p"]
        if not at_b and not at_p:
            return issued, useful
        if at_b:
            issued += BOX_COST
            useful += BOX_COST * len(at_b)
            for l in at_b:
                pos[l] += 1
        if not at_b or len(at_p) >= prim_wait:
            if at_p:
                issued += PRIM_COST
                useful += PRIM_COST * len(at_p)
                for l in at_p:
                    pos[l] += 1


def main():
    n_rays = int(sys.argv[1]) if len(sys.argv) > 1 else 6400
    prim_wait = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    osc = ob.Scene.cornell_box(asset("cornell_box_plane.obj"))
    for label, build in (("surface-area tree", rt.BUILD_SAH), ("reference tree", rt.BUILD_REFERENCE)):
        sc = rt.Scene(None, osc.materials(), osc.objects(), osc.root_kind, build=build)
        entries = sc.entries(raw=True)
        ent_u = entries.view(np.uint32)
        for kind, rays in (("camera rays of neighbouring pixels", osc.record_rays(256, 256, 1, 1, cap=n_rays)),
                           ("path rays of all depths, in path order", osc.record_rays(64, 64, 4, 6, cap=n_rays))):
            seqs = [visit_sequence(entries, ent_u, r["o"], r["d"]) for r in rays]
            issued = useful = longest = 0
            own = 0
            for g in range(0, len(seqs) - 63, 64):
                grp = seqs[g:g + 64]
                i, u = replay(grp, prim_wait)
                issued += i
                useful += u
                costs = [BOX_COST * s.count("b") + PRIM_COST * s.count("p") for s in grp]
                longest += max(costs)
                own += sum(costs)
            n_groups = len(seqs) // 64
            print(f"{label}, {kind}: {np.mean([s.count('b') for s in seqs]):.1f} box + {np.mean([s.count('p') for s in seqs]):.1f} "
                  f"primitive visits per ray; lanes doing work per issued instruction {64 * useful / (64 * issued):.1f} of 64; "
                  f"if every lane ran alone and the wave waited for its longest: {own / longest:.1f} of 64 "
                  f"(so walk length costs {1 - own / (64 * longest):.2f} of the slots, standing at the other kind of entry "
                  f"{own / (64 * longest) - useful / (64 * issued):.2f} more)", flush=True)
        sc.close()


if __name__ == "__main__":
    main()
