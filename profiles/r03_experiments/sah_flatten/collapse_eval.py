import os, sys
sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo/tools"); sys.path.insert(0, "/root/repo")
import numpy as np
import raytracer_amd as rt
from helpers import asset
from oracle import binding as ob
from lane_utilisation import visit_sequence

COST = {0: 1.0, 1: 1.6, 2: 1.3, 3: 0.0}
def area(e):
    d = np.maximum(e[4:7].astype(np.float64) - e[0:3].astype(np.float64), 0)
    return d[0]*d[1] + d[1]*d[2] + d[2]*d[0]

def structure(entries):
    eu = entries.view(np.uint32); n = len(entries)
    tag = (eu[:, 3] >> 30).astype(int); skip = (eu[:, 3] & 0x3FFFFFFF).astype(np.int64)
    kids = {}
    parent = np.full(n, -1)
    for i in range(n):
        if tag[i] == 0:
            out = []; c = i + 1
            while c < skip[i]:
                out.append(c); c = skip[c] if tag[c] == 0 else c + 1
            kids[i] = out
            for c in out: parent[c] = i
    return tag, skip, kids, parent

def apply_drop(entries, drop, skip, tag):
    n = len(entries)
    new_index = np.append(np.cumsum(~drop) - (~drop), (~drop).sum())
    out = entries[~drop].copy(); ou = out.view(np.uint32)
    k = 0
    for i in range(n):
        if drop[i]: continue
        if tag[i] == 0: ou[k, 3] = int(new_index[skip[i]])
        k += 1
    return out

def simple(entries, scale, max_kids=1000):
    tag, skip, kids, parent = structure(entries); n = len(entries)
    A = np.array([area(entries[i]) if tag[i] == 0 else 0.0 for i in range(n)])
    drop = np.zeros(n, bool)
    for i in range(n):
        if tag[i] != 0 or parent[i] < 0: continue
        if any(tag[c] == 3 for c in kids[i]) or (i > 0 and tag[i-1] == 3): continue
        a = parent[i]
        while drop[a]: a = parent[a]
        p = A[i] / max(A[a], 1e-30)
        S = sum(COST[tag[c]] for c in kids[i])
        if COST[0] < (1 - p) * S * scale: continue
        drop[i] = True
    return apply_drop(entries, drop, skip, tag), int(drop.sum())

def iterative(entries, max_kids):
    tag, skip, kids, parent = structure(entries); n = len(entries)
    A = np.array([area(entries[i]) if tag[i] == 0 else 0.0 for i in range(n)])
    drop = np.zeros(n, bool)
    def eff_kids(i):
        out = []
        for c in kids[i]:
            if tag[c] == 0 and drop[c]: out += eff_kids(c)
            else: out.append(c)
        return out
    changed = True; rounds = 0
    while changed and rounds < 20:
        changed = False; rounds += 1
        for i in range(n):
            if tag[i] != 0 or parent[i] < 0: continue
            if any(tag[c] == 3 for c in kids[i]) or (i > 0 and tag[i-1] == 3): continue
            a = parent[i]
            while drop[a]: a = parent[a]
            ek = eff_kids(i)
            S = sum(COST[tag[c]] for c in ek)
            # cost with i kept: A[a]*1 + A[i]*S ; dropped: A[a]*S
            kept = A[a] * COST[0] + A[i] * S
            dropped = A[a] * S
            # sibling-run cap: the surviving ancestor's effective children if i is dropped
            want = dropped < kept
            if want and not drop[i]:
                was = drop[i]; drop[i] = True
                if len(eff_kids(a)) > max_kids: drop[i] = was; continue
                changed = True
            elif not want and drop[i]:
                drop[i] = False; changed = True
    return apply_drop(entries, drop, skip, tag), int(drop.sum())

def evaluate(name, e, rays, label):
    eu = e.view(np.uint32)
    seqs = [visit_sequence(e, eu, r["o"], r["d"]) for r in rays]
    b = np.mean([s.count('b') for s in seqs]); p = np.mean([s.count('p') for s in seqs])
    print(f"{name} {label}: {len(e)} entries, {b:.2f} box + {p:.2f} prim visits per path ray, weighted {45*b+75*p:.0f}", flush=True)

def run(name, osc, n_rays):
    rays = osc.record_rays(64, 64, 4, 6, cap=n_rays)
    sc = rt.Scene(None, osc.materials(), osc.objects(), osc.root_kind, build=rt.BUILD_SAH)
    entries = sc.entries(raw=True)
    evaluate(name, entries, rays, "as built")
    for scale in (1.0, 0.85):
        e, nd = simple(entries, scale); evaluate(name, e, rays, f"simple top-down scale {scale} ({nd} dropped)")
    for mk in (4, 8, 1000):
        e, nd = iterative(entries, mk); evaluate(name, e, rays, f"iterative, at most {mk} children ({nd} dropped)")
    sc.close()
which = sys.argv[1]
if which == "plane": run("cornell_plane", ob.Scene.cornell_box(asset("cornell_box_plane.obj")), 3000)
elif which == "world1": run("world1", ob.Scene.world1(), 1200)
elif which == "thing": run("thing_glass", ob.Scene.world2(asset("thing.obj"), 1), 800)
elif which == "cornell": run("cornell_full", ob.Scene.cornell_box(asset("cornell_box.obj")), 300)
