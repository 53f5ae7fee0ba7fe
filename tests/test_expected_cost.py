"""The surface-area measure RT_BUILD_AUTO compares its two trees by (rt_scene_expected_cost)."""




def test_expected_cost_prefers_the_surface_area_trees(rt, ob):
    """rt_scene_expected_cost: the surface-area measure of a flattened tree (what a scene of
    RT_BUILD_AUTO compares its two trees by).  Lower on the surface-area trees of every scene of the
    reference, and in the order of the measured box + triangle tests per ray."""
    from helpers import asset
    measured = {}
    for name, make in (("cornell_box_plane.obj", lambda: ob.Scene.cornell_box(asset("cornell_box_plane.obj"))),
                       ("cornell_box.obj", lambda: ob.Scene.cornell_box(asset("cornell_box.obj"))),
                       ("world1", ob.Scene.world1), ("thing.obj", lambda: ob.Scene.world2(asset("thing.obj"), 1))):
        osc = make()
        cost = []
        for build in (rt.BUILD_REFERENCE, rt.BUILD_SAH):
            sc = rt.Scene(None, osc.materials(), osc.objects(), osc.root_kind, build=build)
            cost.append(sc.expected_cost)
            sc.close()
        measured[name] = cost
        assert 0.0 < cost[1] < cost[0], (name, cost)
    # (the plane box: 21.7 box + 12.1 triangle tests per path ray on the reference tree, 17.1 + 2.9 on the other)
    assert 30.0 < measured["cornell_box_plane.obj"][0] < 55.0 and 15.0 < measured["cornell_box_plane.obj"][1] < 35.0
    empty = rt.Scene(None, ob.Scene.world1().materials(), [], ob.ROOT_LIST)
    assert empty.expected_cost == 0.0
    empty.close()
