"""Exact ties in t, every form of the closest-hit loop, both trees.

The reference reports, of the primitives that tie at the smallest t, the one its visiting order
leaves standing (shape.rs:32-36, 297).  The forms that visit in another order -- another tree, the
nearer child first, eight children at a time -- resolve ties by primitive key and must report the
same primitive on every ray of a scene made of ties."""
import numpy as np
import pytest

from helpers import assert_hits_equal, tie_scene

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("root", [0, 2])
def test_every_form_reports_the_reference_primitive_on_ties(rt, ob, ctx, root):
    mats, objects, rays, dyadic = tie_scene(ob, n_random_rays=120000)
    osc = ob.Scene.from_description(mats, objects, root)
    forms = [(build, mode, wide) for build in (rt.BUILD_REFERENCE, rt.BUILD_SAH)
             for mode, wide in ((rt.TRAVERSAL_PER_LANE, None), (rt.TRAVERSAL_SWEEP, None), (rt.TRAVERSAL_ORDERED, None),
                                (rt.TRAVERSAL_OCTANT, None), (rt.TRAVERSAL_WIDE, rt.WIDE_BUILD_COLLAPSE),
                                (rt.TRAVERSAL_WIDE, rt.WIDE_BUILD_DEVICE))]
    wants = {}
    for t_min, t_max in ((0.001, np.inf), (0.001, 5.0), (0.5, np.inf)):
        wants[(t_min, t_max)] = osc.trace(rays, t_min, t_max)[0]
    want = wants[(0.001, np.inf)]
    # how many rays really see a tie: the closest triangle of the doubled mesh is there twice
    n_mesh = len(objects[1][1])
    first_mesh = 3
    on_doubled = (want["prim_id"] >= first_mesh) & (want["prim_id"] < first_mesh + n_mesh)
    assert on_doubled.sum() > 10000
    for build, mode, wide in forms:
        sc = rt.Scene(ctx, mats, objects, root, build=build)
        if wide is not None:
            sc.set_wide_build(wide)
        sc.set_traversal(mode)
        for (t_min, t_max), w in wants.items():
            assert_hits_equal(sc.trace(rays, t_min, t_max), w)
        sc.close()
