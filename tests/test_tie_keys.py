"""Primitive keys and the order-independent tie rule (include/rt_abi.h, "Primitive keys").

The reference resolves exact ties in t by its visiting order (shape.rs:32-36: a sphere replaces the
hit at root <= t_max; shape.rs:297: a triangle only at t < t_max).  The flattened arrays name every
primitive by a key made from its rank in that order, and a walk in any other order accepts a
primitive at t == best_t only with the larger key.  These CPU tests hold the keys against the
reference order and the rule -- a Python mirror of the device comparison, walking the trees in
orders the reference never uses -- against the oracle, on a scene made of exact ties."""
import numpy as np
import pytest

from helpers import random_scene, tie_scene
from test_host_layer import F, _box_miss, _dot

KEY_SPHERE, KEY_NONE, KEY_MASK, MISS = 0x40000000, 0x3FFFFFFF, 0x7FFFFFFF, 0xFFFFFFFF


def _keys(sc):
    u = sc.entries(raw=True).view(np.uint32)
    tags = u[:, 3] >> 30
    rows = np.nonzero((tags == 1) | (tags == 2))[0]
    return tags[rows], u[rows, 7]


@pytest.mark.parametrize("root", [0, 2])
def test_keys_number_the_primitives_in_the_reference_order(rt, ob, root):
    mats, objects, _ = random_scene(ob, 31 + root, n_spheres=14, n_meshes=3, tris_per_mesh=33, extent=5.0)
    ref = rt.Scene(None, mats, objects, root)
    sah = rt.Scene(None, mats, objects, root, build=rt.BUILD_SAH)
    order = ref.prim_order()
    assert sorted(order.tolist()) == list(range(ref.num_prims))          # a permutation of the ids
    assert np.array_equal(order, sah.prim_order())                       # the rank does not depend on the tree traced
    tags, keys = _keys(ref)
    ranks = np.where(tags == 2, keys & 0x3FFFFFFF, 0x3FFFFFFE - (keys & 0x3FFFFFFF))
    assert ranks.tolist() == list(range(ref.num_prims))                  # pre-order of the reference trees
    assert np.array_equal((keys & KEY_SPHERE) != 0, tags == 2)
    assert (keys[tags == 1] < KEY_NONE).all() and (keys[tags == 2] > KEY_NONE).all()
    # the ids behind the keys: spheres and triangles in the order they were added
    ids = ref.prim_of_key(keys)
    n_sph = sum(len(o[1]) for o in objects if o[0] == "spheres")
    assert (tags == 2).sum() == n_sph and sorted(ids.tolist()) == list(range(ref.num_prims))
    assert ref.prim_of_key(np.uint32(MISS)) == MISS
    # the other tree holds the same primitives under the same keys, in another order
    tags_s, keys_s = _keys(sah)
    assert sorted(keys_s.tolist()) == sorted(keys.tolist()) and keys_s.tolist() != keys.tolist()
    # and the decoded view shows ids where the raw arrays show keys
    dec = ref.entries().view(np.uint32)
    assert sorted(dec[(dec[:, 3] >> 30 == 1) | (dec[:, 3] >> 30 == 2), 7].tolist()) == list(range(ref.num_prims))


def _prim_test_keys(e, tag, key, o, d, t_min, best_t, best):
    """tri_step<TIE> / sphere_step<TIE> of the kernels: the reference's tests, and at t == best_t the larger key."""
    eps = F(1.1920929e-7)
    wins = key > (best & KEY_MASK)
    if tag == 1:
        v0, e1, e2 = e[0:3], e[4:7], e[8:11]
        p = [F(F(d[1] * e2[2]) - F(d[2] * e2[1])), F(F(d[2] * e2[0]) - F(d[0] * e2[2])),
             F(F(d[0] * e2[1]) - F(d[1] * e2[0]))]
        det = _dot(e1, p)
        if not (abs(det) < eps):
            invd = F(1.0) / det
            tv = [F(o[k] - v0[k]) for k in range(3)]
            u = F(_dot(tv, p) * invd)
            if not (u < 0 or u > 1):
                q = [F(F(tv[1] * e1[2]) - F(tv[2] * e1[1])), F(F(tv[2] * e1[0]) - F(tv[0] * e1[2])),
                     F(F(tv[0] * e1[1]) - F(tv[1] * e1[0]))]
                v = F(_dot(d, q) * invd)
                if not (v < 0 or F(u + v) > 1):
                    t = F(_dot(e2, q) * invd)
                    if t >= t_min and (t < best_t or (t == best_t and wins)):
                        return t, key
        return best_t, best
    c, r = e[0:3], e[4]
    oc = [F(o[k] - c[k]) for k in range(3)]
    a = _dot(d, d)
    hb = _dot(oc, d)
    cc = F(_dot(oc, oc) - F(r * r))
    disc = F(F(hb * hb) - F(a * cc))
    if not (disc < 0):
        sq = np.sqrt(disc)
        out = lambda root: root < t_min or best_t < root or (root == best_t and not wins)
        root = F(F(-hb - sq) / a)
        if out(root):
            root = F(F(-hb + sq) / a)
            if out(root):
                return best_t, best
        return root, key
    return best_t, best


def _walk_any_order(entries, o, d, t_min, t_max, choose):
    """Stack walk of the entry array that visits the children of every box in the order `choose`
    gives them (0: as stored, 1: reversed, 2: by a hash of the box index), with the tie rule."""
    u = entries.view(np.uint32)
    n = len(entries)
    end_of = lambda i: int(u[i, 3] & 0x3FFFFFFF) if (u[i, 3] >> 30) == 0 else i + 1
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        o, d = [F(x) for x in o], [F(x) for x in d]
        inv = [F(1.0) / d[k] for k in range(3)]
        best_t, best = F(t_max), KEY_NONE
        tops, c = [], 0
        while c < n:
            tops.append(c)
            c = end_of(c)
        if choose:
            tops.reverse()
        stack = list(reversed(tops))
        while stack:
            i = stack.pop()
            tag = int(u[i, 3]) >> 30
            if tag != 0:
                best_t, best = _prim_test_keys(entries[i], tag, int(u[i, 7]), o, d, t_min, best_t, best)
                continue
            if _box_miss(entries[i], o, inv, t_min, best_t):
                continue
            kids, c = [], i + 1
            while c < end_of(i):
                kids.append(c)
                c = end_of(c)
            if choose == 1 or (choose == 2 and (i * 2654435761) & 0x10000):
                kids.reverse()
            stack.extend(reversed(kids))
    return (MISS if best == KEY_NONE else best), best_t


@pytest.mark.parametrize("root", [0, 2])
def test_the_tie_rule_gives_the_reference_primitive_in_any_order(rt, ob, root):
    mats, objects, rays, dyadic = tie_scene(ob, n_random_rays=260)
    osc = ob.Scene.from_description(mats, objects, root)
    scenes = [rt.Scene(None, mats, objects, root), rt.Scene(None, mats, objects, root, build=rt.BUILD_SAH)]
    ents = [s.entries(raw=True) for s in scenes]
    # (Ranges whose ends coincide with a tied t are left out: Aabb::hit rejects a box at
    # t_max <= t_min, math.rs:160-162, so with t_min or t_max -- or a box's entry plane -- exactly at
    # the tied t the reference reaches the later primitive or not depending on what it has found
    # before, which no rule that is independent of the order can repeat.)
    for t_min, t_max in ((0.001, np.inf), (0.001, 5.0), (0.5, np.inf)):
        want, _ = osc.trace(rays, t_min, t_max, nthreads=1)
        hits = want["prim_id"] != MISS
        if t_max == np.inf and t_min == 0.001:
            # the scene does what it was made for: the rays at the poles see the sphere and the plane's
            # triangles at one t, and the reference's answer depends on which it visits first
            # (a sphere wins such a tie wherever it stands: visited last it replaces, visited first it stays)
            up = np.arange(dyadic, len(rays), 2)
            assert hits[dyadic:].all() and np.array_equal(want["t"][up], F(4.0) / rays["d"][up, 2])
            sph_ids = [sum(len(o[1]) for o in objects[:k]) for k, o in enumerate(objects) if o[0] == "spheres"][-2:]
            assert set(want["prim_id"][up].tolist()) == set(sph_ids)
        picks = range(len(rays)) if t_max == np.inf and t_min == 0.001 else range(dyadic - 40, len(rays))
        for sc, ent in zip(scenes, ents):
            for choose in (0, 1, 2):
                for k in picks:
                    key, t = _walk_any_order(ent, rays["o"][k], rays["d"][k], F(t_min), F(t_max), choose)
                    assert sc.prim_of_key(np.uint32(key)) == want["prim_id"][k], (choose, k)
                    if key != MISS:
                        assert F(t).view(np.uint32) == want["t"][k].view(np.uint32)
