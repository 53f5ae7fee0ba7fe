"""The wide tree of the opt-in RT_TRAVERSAL_WIDE walk, on host-only scenes (no GPU).

The tree has no counterpart in the reference (its trees are binary, shape.rs:87-139, 185-192): it
is the scene's flattened entries collapsed into nodes of up to eight children with 8-bit boxes.
Checked here: the structure (every primitive of the entry array is a leaf exactly once, every
child's quantised box encloses what lies beneath it with a grid step to spare, the recorded depth),
and a Python mirror of the device loop (same stack discipline, same fused box test, the
reference's primitive tests in f32) against the oracle's closest hits: hit / miss and t bit for
bit, the primitive unless two tie in t."""
import numpy as np
import pytest

from helpers import asset, random_mesh, random_rays, random_scene
from test_host_layer import F, _prim_test


def _planes(node):
    """(origin[3], step[3], lo[8, 3], hi[8, 3]) of a node in float64 (grid points are exact there)."""
    origin = node[0:3].view(np.float32).astype(np.float64)
    e = np.array([node[3] & 0xFF, node[3] >> 8 & 0xFF, node[3] >> 16 & 0xFF], dtype=np.int64)
    step = np.ldexp(1.0, e - 127)
    q = node[8:20].view(np.uint8).reshape(6, 8).astype(np.float64)     # plane p, slot s
    lo = origin[None, :] + q[0:3].T * step[None, :]
    hi = origin[None, :] + q[3:6].T * step[None, :]
    return origin, step, lo, hi


def _prim_box(rec):
    tag = int(rec.view(np.uint32)[3]) >> 30
    if tag == 1:
        p = np.stack([rec[0:3], rec[0:3] + rec[4:7], rec[0:3] + rec[8:11]]).astype(np.float64)
        return p.min(axis=0), p.max(axis=0)
    c, r = rec[0:3].astype(np.float64), abs(float(rec[4]))
    return c - r, c + r


def _check_structure(nodes, prims, depth, entries):
    ent_u = entries.view(np.uint32)
    tags = ent_u[:, 3] >> 30
    want_ids = sorted(int(x) for x in ent_u[tags != 0, 7])
    got_ids = sorted(int(x) for x in prims.view(np.uint32)[:, 7])
    assert got_ids == want_ids                       # every primitive once
    seen_prims = np.zeros(len(prims), dtype=bool)
    seen_nodes = set()                               # (the blocks of child slots overlap: no slot may serve two nodes)
    max_level = 0

    def visit(slot, level):
        """returns the (lo, hi) of everything beneath the node"""
        nonlocal max_level
        max_level = max(max_level, level)
        assert slot not in seen_nodes
        seen_nodes.add(slot)
        node = nodes[slot]
        imask, lmask = int(node[3]) >> 24, int(node[6])
        assert imask & lmask == 0
        _, step, qlo, qhi = _planes(node)
        lo_all, hi_all = np.full(3, np.inf), np.full(3, -np.inf)
        rank = 0
        for s in range(8):
            if lmask >> s & 1:
                idx = int(node[5]) + rank
                rank += 1
                assert not seen_prims[idx]
                seen_prims[idx] = True
                lo, hi = _prim_box(prims[idx])
            elif imask >> s & 1:
                lo, hi = visit(int(node[4]) + s, level + 1)
            else:
                continue
            assert (qlo[s] <= lo - 0.5 * step).all() and (qhi[s] >= hi + 0.5 * step).all()
            lo_all, hi_all = np.minimum(lo_all, lo), np.maximum(hi_all, hi)
        return lo_all, hi_all

    if len(prims):
        visit(0, 1)
    assert seen_prims.all()
    assert len(seen_nodes) >= 0.6 * len(nodes) or len(nodes) < 16      # the array is mostly nodes, not holes
    assert max_level == depth or (len(prims) == 0 and depth == 1)


_MARK = 0xFFFFFF00


def _xf_apply(m, v, w):
    """glam's transform_point3 (w = 1) / transform_vector3 (w = 0) on the 16 column-major floats m, in f32 steps."""
    c = [[F(m[4 * j + i]) for i in range(3)] for j in range(4)]
    return [F(F(c[3][i] * F(w)) + F(F(c[2][i] * v[2]) + F(F(c[1][i] * v[1]) + F(c[0][i] * v[0])))) for i in range(3)]


def _walk_wide(nodes, prims, o, d, t_min, t_max, inverses=None, high_water=None):
    """Mirror of closest_hit_wide: returns (prim, t, node visits, primitive tests).  `inverses`: the inverse
    transforms (16 column-major floats each) of a scene with transformed meshes, by transform number.
    `high_water`: a one-element list that is raised to the largest number of words the stack ever held."""
    pu = prims.view(np.uint32)
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        o, d = [F(x) for x in o], [F(x) for x in d]
        inv = [F(1.0) / d[k] for k in range(3)]
        rb = sum(1 << k for k in range(3) if d[k] >= 0)
        best_t, best = F(t_max), 0xFFFFFFFF
        G, stack = 1 << rb, []
        world = None
        n_nodes = n_prims = 0
        while (G & 0xFF) or stack:
            if not (G & 0xFF):
                G = stack.pop()
                if G == _MARK:                       # the mesh's subtree is exhausted: the caller's ray again
                    o, d = world
                    inv = [F(1.0) / d[k] for k in range(3)]
                    rb = sum(1 << k for k in range(3) if d[k] >= 0)
                    G = stack.pop() if stack else 0
                    if not (G & 0xFF):
                        continue
            p = (G & 0xFF).bit_length() - 1
            G ^= 1 << p
            node = nodes[(G >> 8) + (p ^ rb)]
            n_nodes += 1
            if int(node[7]) >> 8:                    # the root of a transformed mesh's tree
                world = (o, d)
                if G & 0xFF:
                    stack.append(G)
                stack.append(_MARK)
                if high_water is not None:
                    high_water[0] = max(high_water[0], len(stack))
                G = 0
                m = inverses[(int(node[7]) >> 8) - 1]
                o, d = _xf_apply(m, o, 1.0), _xf_apply(m, d, 0.0)
                inv = [F(1.0) / d[k] for k in range(3)]
                rb = sum(1 << k for k in range(3) if d[k] >= 0)
            origin = node[0:3].view(np.float32)
            e = [int(node[3]) >> (8 * k) & 0xFF for k in range(3)]
            imask, lmask = int(node[3]) >> 24, int(node[6])
            q = node[8:20].view(np.uint8).reshape(6, 8)
            sx = [F(np.ldexp(F(1.0), e[k] - 127) * inv[k]) for k in range(3)]
            bx = [F(F(origin[k] - o[k]) * inv[k]) for k in range(3)]
            bt = best_t if best_t == best_t else F(np.inf)
            dd = F(F(F(d[0] * d[0]) + F(d[1] * d[1])) + F(d[2] * d[2]))
            wild = not (dd < F(1e18) and dd > F(1e-30))
            hm = int(node[7]) & 0xFF if wild else 0
            for s in range(8):
                tn, tf = F(t_min), bt
                for k in range(3):
                    near, far = (q[3 + k, s], q[k, s]) if inv[k] < 0 else (q[k, s], q[3 + k, s])
                    a = F(np.float64(near) * np.float64(sx[k]) + np.float64(bx[k]))
                    b = F(np.float64(far) * np.float64(sx[k]) + np.float64(bx[k]))
                    tn = a if a > tn else tn         # (a NaN drops out, as in fmaxf / fminf)
                    tf = b if b < tf else tf
                if tn <= tf:
                    hm |= 1 << s
            pend = hm & lmask
            while pend:
                s = (pend & -pend).bit_length() - 1
                pend &= pend - 1
                idx = int(node[5]) + bin(lmask & ((1 << s) - 1)).count("1")
                n_prims += 1
                best_t, best = _prim_test(prims[idx], int(pu[idx, 3]) >> 30, int(pu[idx, 7]), o, d, t_min, best_t, best)
            ih = 0
            for s in range(8):
                if (hm & imask) >> s & 1:
                    ih |= 1 << (s ^ rb)
            if ih:
                if G & 0xFF:
                    stack.append(G)
                    if high_water is not None:
                        high_water[0] = max(high_water[0], len(stack))
                G = int(node[4]) << 8 | ih
    return best, best_t, n_nodes, n_prims


def _compare(nodes, prims, rays, want, t_min=0.001, t_max=np.inf):
    ties = 0
    for k in range(len(rays)):
        prim, t, _, _ = _walk_wide(nodes, prims, rays["o"][k], rays["d"][k], F(t_min), F(t_max))
        wp = int(want["prim_id"][k])
        assert (prim == 0xFFFFFFFF) == (wp == 0xFFFFFFFF), k
        if wp != 0xFFFFFFFF and not np.isnan(want["t"][k]):
            assert F(t).view(np.uint32) == want["t"][k].view(np.uint32), k
            ties += prim != wp
    return ties


@pytest.mark.parametrize("root,build", [(0, 0), (2, 0), (0, 1)])
def test_wide_tree_structure_and_hits(rt, ob, root, build):
    mats, objects, _ = random_scene(ob, 21 + root, n_spheres=23, n_meshes=3, tris_per_mesh=70, extent=6.0)
    osc = ob.Scene.from_description(mats, objects, root)
    sc = rt.Scene(None, mats, objects, root, build=build)
    nodes, prims, depth = sc.wide()
    _check_structure(nodes, prims, depth, sc.entries())
    rays = random_rays(ob, 7, 300, -7.0, 7.0)
    want, _ = osc.trace(rays, 0.001, np.inf, nthreads=1)
    assert (want["prim_id"] != 0xFFFFFFFF).sum() > 50
    assert _compare(nodes, prims, rays, want) <= 1
    want, _ = osc.trace(rays, 0.05, 2.0, nthreads=1)
    assert _compare(nodes, prims, rays, want, 0.05, 2.0) <= 1


def test_wide_tree_degenerate_scenes(rt, ob):
    mats, _, _ = random_scene(ob, 5)
    rng = np.random.default_rng(0)
    one = np.zeros(1, dtype=ob.sphere_dt)
    one["r"] = 1.0
    many = np.zeros(30, dtype=ob.sphere_dt)            # a list of thirty objects: more than eight children
    many["c"] = rng.uniform(-3, 3, (30, 3)).astype(np.float32)
    many["r"] = 0.4
    cases = [([("spheres", one)], ob.ROOT_BVH), ([("spheres", one)], ob.ROOT_SINGLE), ([], ob.ROOT_BVH),
             ([("mesh", random_mesh(ob, rng, 0))], ob.ROOT_SINGLE), ([("spheres", many)], ob.ROOT_LIST)]
    for n in (1, 2, 3, 9):
        cases.append(([("mesh", random_mesh(ob, rng, n, extent=2.0))], ob.ROOT_SINGLE))
    flat = np.zeros(12, dtype=ob.triangle_dt)          # coplanar, axis-aligned: boxes without thickness
    flat["p"][:, :, :2] = rng.uniform(-2, 2, (12, 3, 2)).astype(np.float32)
    flat["n0"][:, 2] = 1.0
    cases.append(([("mesh", flat)], ob.ROOT_SINGLE))
    rays = random_rays(ob, 1, 120, -3.0, 3.0)
    for objects, root in cases:
        osc = ob.Scene.from_description(mats, objects, root)
        sc = rt.Scene(None, mats, objects, root)
        nodes, prims, depth = sc.wide()
        _check_structure(nodes, prims, depth, sc.entries())
        want, _ = osc.trace(rays, 0.001, np.inf, nthreads=1)
        assert _compare(nodes, prims, rays, want) == 0


def test_wide_walk_keeps_the_overflowed_sphere_hits_of_a_list_root(rt, ob):
    """A ray whose |d| is about 1e18 overflows the sphere test's discriminant to NaN, and the reference
    accepts such a hit (shape.rs:25-36) on every sphere it tests -- the objects of a HittableList at
    the root are tested without any box before them (shape.rs:60-74).  The wide tree marks those
    objects (word 7 of a node) and such rays enter them whatever the boxes say: hit / miss as the
    oracle's.  Under a Bvh root every box test of such a ray fails in the reference too."""
    from helpers import adversarial_rays
    for root in (ob.ROOT_LIST, ob.ROOT_BVH):
        mats, objects, _ = random_scene(ob, 31, n_spheres=19, n_meshes=1, tris_per_mesh=9, extent=6.0)
        osc = ob.Scene.from_description(mats, objects, root)
        sc = rt.Scene(None, mats, objects, root)
        nodes, prims, depth = sc.wide()
        marked = sum(bin(int(n[7])).count("1") for n in nodes)
        assert (marked > 0) == (root == ob.ROOT_LIST)
        rays = adversarial_rays(rt, 5, 1500, 8.0)
        rays = rays[np.abs(rays["d"]).max(axis=1) > 1e15][:120]
        assert len(rays) > 50
        want, _ = osc.trace(rays, 0.001, np.inf, nthreads=1)
        assert (np.isnan(want["t"]) & (want["prim_id"] != 0xFFFFFFFF)).any() == (root == ob.ROOT_LIST)
        _compare(nodes, prims, rays, want)


def test_wide_tree_of_the_obj_scenes(rt, ob):
    """thing.obj and the plane Cornell box: structure, depth, and fewer node visits than the binary
    walk makes box tests."""
    from test_host_layer import _walk
    for osc in (ob.Scene.world2(asset("thing.obj"), 1), ob.Scene.cornell_box(asset("cornell_box_plane.obj"))):
        sc = rt.Scene(None, osc.materials(), osc.objects(), osc.root_kind)
        nodes, prims, depth = sc.wide()
        ent = sc.entries()
        _check_structure(nodes, prims, depth, ent)
        rays = osc.record_rays(24, 24, 1, 4, cap=150)
        want, _ = osc.trace(rays, 0.001, np.inf, nthreads=1)
        assert _compare(nodes, prims, rays, want) <= 2
        wide_nodes = sum(_walk_wide(nodes, prims, r["o"], r["d"], F(0.001), F(np.inf))[2] for r in rays[:60])
        boxes = sum(_walk(ent, r["o"], r["d"], F(0.001), F(np.inf), count=True)[2] for r in rays[:60])
        print(f"node visits {wide_nodes} against {boxes} box tests on 60 rays, depth {depth}, {len(nodes)} node slots")
        assert wide_nodes * 2 < boxes


def test_wide_tree_holds_a_transformed_mesh_as_a_node_of_its_own(rt, ob):
    """Mesh::hit transforms the ray before it descends (shape.rs:454): in the wide tree the mesh's tree is
    collapsed in the mesh's own space, its root node names the transform (upper bits of word 7), and
    the box it has in the node above encloses the mesh as the caller sees it."""
    mats, _, _ = random_scene(ob, 5)
    rng = np.random.default_rng(1)
    m = np.eye(4, dtype=np.float32)
    m[3, 0:3] = (2.0, -1.0, 0.5)                                       # (column-major: the translation)
    m[0, 0], m[1, 1] = 1.5, 0.75
    tris = random_mesh(ob, rng, 40)
    sc = rt.Scene(None, mats, [("mesh", tris, m.reshape(-1)), ("mesh", random_mesh(ob, rng, 30))], ob.ROOT_BVH)
    nodes, prims, depth = sc.wide()
    assert len(prims) == 70
    named = np.nonzero(nodes[:, 7] >> 8)[0]
    assert len(named) == 1 and nodes[named[0], 7] >> 8 == 1            # one mesh's root, transform number 1
    # find its slot in the node above and compare that box with the transformed vertices
    world = tris["p"].reshape(-1, 3).astype(np.float64) * np.array([1.5, 0.75, 1.0]) + np.array([2.0, -1.0, 0.5])
    found = False
    for n in nodes:
        imask, base = int(n[3]) >> 24, int(n[4])
        for slot in range(8):
            if (imask >> slot & 1) and base + slot == named[0]:
                _, _, lo, hi = _planes(n)
                assert (lo[slot] <= world.min(axis=0)).all() and (hi[slot] >= world.max(axis=0)).all()
                assert ((hi[slot] - lo[slot]) < (world.max(axis=0) - world.min(axis=0)) * 1.2 + 0.1).all()
                found = True
    assert found
    # the mesh's own nodes hold it untransformed
    _, _, lo, hi = _planes(nodes[named[0]])
    local = tris["p"].reshape(-1, 3).astype(np.float64)
    used = (int(nodes[named[0], 3]) >> 24) | int(nodes[named[0], 6])
    box_lo = np.min([lo[s] for s in range(8) if used >> s & 1], axis=0)
    box_hi = np.max([hi[s] for s in range(8) if used >> s & 1], axis=0)
    assert (box_lo <= local.min(axis=0)).all() and (box_hi >= local.max(axis=0)).all()
    sc.close()


def test_threaded_wide_build_is_deterministic(rt, ob):
    """Big subtrees near the root are collapsed on threads of their own, each into its own arena, and
    appended in slot order: the arrays do not depend on timing.  200 k triangles (400 k entries: the
    root's children are past the size that gets a thread), twice, bit for bit; and every node slot
    the walk can reach holds a node (first-fit compaction left no dangling base)."""
    arrays = []
    for _ in range(2):
        w = rt.World.soup(None, 200000, grid=8, seed=7)
        nodes, prims, depth = w.scene.wide()
        arrays.append((nodes.copy(), prims.copy(), depth))
        w.close()
    assert arrays[0][2] == arrays[1][2] and 6 <= arrays[0][2] <= 16
    assert np.array_equal(arrays[0][0], arrays[1][0]) and np.array_equal(arrays[0][1].view(np.uint32), arrays[1][1].view(np.uint32))
    nodes, prims, _ = arrays[0]
    assert len(prims) == 200000 + 64 + 1
    imask, lmask = nodes[:, 3] >> 24, nodes[:, 6]
    reach = np.zeros(len(nodes), dtype=bool)
    reach[0] = True
    for s in range(8):                               # children named by some node
        has = (imask >> s & 1).astype(bool)
        reach[nodes[has, 4] + s] = True
    used = (imask | lmask) != 0
    assert np.array_equal(reach, used)               # named <=> holds a node
    assert used.mean() > 0.85                        # mostly nodes, few holes
    assert int(np.array([bin(int(m)).count("1") for m in lmask[used]]).sum()) == len(prims)


@pytest.mark.parametrize("root", [0, 2])
def test_wide_tree_of_nested_containers(rt, ob, root):
    """Containers inside containers (a Bvh of spheres, a HittableList holding a mesh and another Bvh,
    an empty Bvh): the collapse follows the flattened entries, so nesting costs nothing; under a list
    root only the root's own objects are marked as reached without a box."""
    from helpers import adversarial_rays
    from test_host_layer import _nested_description
    mats, objects = _nested_description(ob, 17, with_xforms=False)
    osc = ob.Scene.from_description(mats, objects, root)
    sc = rt.Scene(None, mats, objects, root)
    nodes, prims, depth = sc.wide()
    _check_structure(nodes, prims, depth, sc.entries())
    marked = sum(bin(int(n[7])).count("1") for n in nodes)
    assert (marked > 0) == (root == 2)
    rays = np.concatenate([random_rays(ob, 3, 250, -5.0, 5.0), adversarial_rays(rt, 4, 150, 5.0)])
    want, _ = osc.trace(rays, 0.001, np.inf, nthreads=1)
    assert (want["prim_id"] != 0xFFFFFFFF).sum() > 25
    assert _compare(nodes, prims, rays, want) <= 1


@pytest.mark.parametrize("root,build", [(0, 0), (2, 0), (0, 1)])
def test_wide_walk_through_transformed_meshes_on_the_cpu(rt, ob, root, build):
    """The walk of a wide tree with transformed meshes, mirrored on the CPU (ray to the mesh's space at the
    root of its tree, back when the subtree's marker comes off the stack), against the oracle.  The
    inverses are computed here in double precision and rounded, not by the library's restatement of
    glam's inverse, so t is compared to a few ulps of the transform's rounding instead of bit for bit
    (the GPU tests do that: test_gpu_xform.py)."""
    rng = np.random.default_rng(41 + root)
    mats, _, _ = random_scene(ob, 9)
    mats4 = []
    objects = []
    for k in range(3):
        a = 0.5 * k + 0.3
        m = np.eye(4)
        m[:2, :2] = [[np.cos(a), -np.sin(a)], [np.sin(a), np.cos(a)]]
        m[2, 2] = 1.0 + 0.5 * k
        m[0, 2] = 0.2 * k
        m[:3, 3] = [2.5 * k - 2.5, 1.0 - k, 0.3 * k]
        mats4.append(m)
        objects.append(("mesh", random_mesh(ob, rng, 40, 2.0), m.astype(np.float32).T.reshape(16).copy()))
    objects.append(("mesh", random_mesh(ob, rng, 30, 2.0)))
    sp = np.zeros(9, dtype=ob.sphere_dt)
    sp["c"] = rng.uniform(-5, 5, (9, 3)).astype(np.float32)
    sp["r"] = rng.uniform(0.3, 0.9, 9).astype(np.float32)
    sp["mat"] = rng.integers(0, 6, 9)
    objects.append(("spheres", sp))
    osc = ob.Scene.from_description(mats, objects, root)
    sc = rt.Scene(None, mats, objects, root, build=build)
    nodes, prims, depth = sc.wide()
    assert len(prims) == 150 + 9 and len(np.nonzero(nodes[:, 7] >> 8)[0]) == 3
    inverses = [np.linalg.inv(m.astype(np.float32).astype(np.float64)).astype(np.float32).T.reshape(16) for m in mats4]
    rays = random_rays(ob, 17, 400, -6.0, 6.0)
    want, _ = osc.trace(rays, 0.001, np.inf, nthreads=1)
    hits = want["prim_id"] != 0xFFFFFFFF
    assert hits.sum() > 80
    wrong = 0
    for k in range(len(rays)):
        prim, t, _, _ = _walk_wide(nodes, prims, rays["o"][k], rays["d"][k], F(0.001), F(np.inf), inverses)
        if prim != int(want["prim_id"][k]):
            wrong += 1
        elif prim != 0xFFFFFFFF:
            assert abs(float(t) - float(want["t"][k])) <= 2e-5 * max(1.0, abs(float(want["t"][k]))), k
    assert wrong <= 2, wrong                      # (an inverse that differs in its last bit may move a hit across a tie)
    sc.close()


@pytest.mark.parametrize("seed", [3, 6, 9, 12, 15, 18])
def test_wide_walk_stack_stays_within_the_recorded_depth(rt, ob, seed):
    """The kernels give every lane one stack word per level of the wide tree (and one more for the marker
    under a transformed mesh's subtree): the mirrored walk never holds more, on scenes of containers inside
    containers with half of the meshes transformed, for random and adversarial rays, on both trees."""
    from fuzz_parity import nested_objects
    from helpers import adversarial_rays
    rng = np.random.default_rng(seed)
    mats, _, _ = random_scene(ob, seed)
    objects = nested_objects(rng)
    for build in (0, 1):
        sc = rt.Scene(None, mats, objects, [0, 2][seed % 2], build=build)
        nodes, prims, depth = sc.wide()
        n_xf = len(np.nonzero(nodes[:, 7] >> 8)[0])
        inverses = []                                  # numbered in the order the meshes were described

        def collect(items):
            for it in items:
                if it[0] == "mesh" and len(it) > 2:
                    m = np.asarray(it[2], dtype=np.float32).reshape(4, 4).T.astype(np.float64)
                    inverses.append(np.linalg.inv(m).astype(np.float32).T.reshape(16))
                elif it[0] in ("bvh", "list"):
                    collect(it[1])
        collect(objects)
        assert len(inverses) == n_xf
        rays = np.concatenate([random_rays(ob, seed, 120, -7.0, 7.0), adversarial_rays(ob, seed, 40, 7.0)])
        mark = [0]
        for k in range(len(rays)):
            _walk_wide(nodes, prims, rays["o"][k], rays["d"][k], F(0.001), F(np.inf), inverses, mark)
        assert mark[0] <= depth + (1 if n_xf else 0), (mark[0], depth, n_xf)
        print(f"seed {seed} build {build}: depth {depth}, {n_xf} transformed meshes, stack high-water {mark[0]}")
        sc.close()
