"""Round 6 (GPU): the opt-in nearest-first child order of the 4-wide walks (cl2_set_traversal_order), trees built with another leaf
size (create_scene(max_members=...)), and the refusal of the cross-check resolve kernel with the reproducible switch (in
test_gpu_parity.py).  The default order -- the reference's, trace.metal:157-160 -- is what every other test of the suite runs."""
import importlib.util
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIGHT, CAMERA = 0, 1


def _order_tool():
    spec = importlib.util.spec_from_file_location("exp_order_ab", os.path.join(ROOT, "tools", "exp_order_ab.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _glass(sub, w, h, **kw):
    import clive2_amd as c2
    from clive2_amd.load import get_materials
    from clive2_amd.meshes import icosphere
    mats = get_materials()
    mats["alpha"][5] = 0.1
    v, f = icosphere(sub, radius=2.0, center=(0.0, 1.0, 0.0))
    return c2.create_scene(w, h, np.array([0, 1.5, 6]), np.array([0, 0, -1]), file_specs=[dict(mesh=(v, f), material=5)], materials=mats, **kw)


def test_traversal_order_is_opt_in(cornell_small, glass_scene, oracle_mod):
    """Default 0 = the reference's order; 1 must be asked for; other values are refused and leave the setting alone.  A tree that
    lives in LDS (the Cornell box) has no 4-wide walk: it renders the same bytes either way.  With order 0 set explicitly the glass
    scene still equals the oracle bit for bit."""
    from clive2_amd.renderer import Renderer, RendererError, make_seeds
    seeds = make_seeds(cornell_small.pixel_width * cornell_small.pixel_height)
    a, b = Renderer(cornell_small, seeds=seeds), Renderer(cornell_small, seeds=seeds)
    assert a.traversal_order() == 0
    b.set_traversal_order(1)
    assert b.traversal_order() == 1
    with pytest.raises(RendererError):
        b.set_traversal_order(2)
    assert b.traversal_order() == 1
    a.run_samples(3); b.run_samples(3)
    assert a.export_paths(CAMERA).tobytes() == b.export_paths(CAMERA).tobytes()
    assert a.export_aggregators()["total_contribution"].tobytes() == b.export_aggregators()["total_contribution"].tobytes()
    a.close(); b.close()
    seeds = make_seeds(glass_scene.pixel_width * glass_scene.pixel_height)
    r, o = Renderer(glass_scene, seeds=seeds), oracle_mod.OracleRenderer(glass_scene, seeds=seeds)
    r.set_traversal_order(1); r.set_traversal_order(0)
    r.set_traversal_mode(5)
    r.run_samples(2); o.run_sample(); o.run_sample()
    assert r.export_paths(LIGHT).tobytes() == o.out_light_paths.tobytes() and r.export_paths(CAMERA).tobytes() == o.out_camera_paths.tobytes()
    assert r.export_aggregators()["total_contribution"].tobytes() == o.weight_aggregators["total_contribution"].tobytes()
    r.close()


def test_nearest_first_order_finds_the_same_hits_and_visits_less():
    """Config-3 geometry at 320 x 180, two samples of the pipeline: every subpath ray and every connection ray (rebuilt from the
    exact render's Path[]) through the 4-wide walk in both orders.  Exact-t ties between two triangles go to the one the reference
    meets first (the rank table, csrc/bvh_traverse.hpp tri_test_tie_rule): NO ray may differ by a tie; a ray may differ only where
    the nearer of the two hits lies in front of its own leaf box's entry distance (at 1080p: 2 of 3.4e8,
    profiles/r06_nearest_first_order_ab.log).  The nearest-first walk visits fewer nodes."""
    from clive2_amd.renderer import Renderer, make_seeds
    tool = _order_tool()
    scene = _glass(4, 320, 180)
    res = tool.compare_orders(scene, tool.pipeline_ray_chunks(scene, 2))
    assert res["rays"] > 3_000_000 and res["by_kind"]["connection"]["rays"] > res["by_kind"]["subpath"]["rays"] > 0
    assert res["identical_fraction"] >= 0.99999, res
    assert res["missed_by_order1"] == 0 and res["missed_by_order0"] == 0, res          # a difference is another triangle, never a miss
    assert res["in_front_of_own_leaf"] == res["differ"], res
    seeds = make_seeds(320 * 180)
    tallies = []
    for order in (0, 1):
        r = Renderer(scene, seeds=seeds)
        r.set_traversal_mode(5); r.set_traversal_order(order); r.set_counting(2)
        r.run_samples(1)
        t = r.walk_tallies()
        tallies.append((t["connection"]["wide_visits"] / t["connection"]["rays"], t["connection"]["tri_records"] / t["connection"]["rays"]))
        assert t["connection"]["rays"] > 0 and t["subpath"]["rays"] > 0
        r.close()
    assert tallies[1][0] < tallies[0][0] and tallies[1][1] <= tallies[0][1], tallies


def test_nearest_first_order_settles_ties_on_shared_edges_as_the_reference_does():
    """Rays aimed AT the vertices and edge midpoints of a 5,120-triangle sphere from four points of the room: where two to six triangles
    meet, several of them are hit at exactly the same t, and the reference keeps the one it meets first (trace.metal:170).  The
    nearest-first walk meets them in another order and must still report the reference's triangle (tri_rank[], csrc/bvh_traverse.hpp
    tri_test_tie_rule): (triangle, t bits) of both orders are compared over all rays; a difference must be explained by a hit in
    front of its own leaf box.  The test has rays to bite on: thousands of them hit on an edge of their triangle (u = 0, v = 0 or u + v = 1)."""
    tool = _order_tool()
    scene = _glass(4, 64, 36)
    t = scene.triangles
    mesh = t[(t["material"] == 5)]
    v0, v1, v2 = (mesh[k][:, :3].astype(np.float32) for k in ("v0", "v1", "v2"))
    targets = np.unique(np.concatenate([v0, v1, v2, (v0 + v1) / 2, (v1 + v2) / 2, (v2 + v0) / 2]).astype(np.float32), axis=0)
    chunks = []
    for origin in ([0.0, 1.5, 6.0], [3.5, 4.0, 3.0], [-3.0, 0.5, -3.5], [0.25, 8.5, 0.5]):
        o = np.broadcast_to(np.asarray(origin, np.float32), targets.shape).copy()
        d = (targets - o).astype(np.float32)
        d = (d / np.sqrt((d * d).sum(axis=1, dtype=np.float32))[:, None]).astype(np.float32)
        chunks.append(("aimed", o, d))
    res = tool.compare_orders(scene, chunks)
    assert res["rays"] == 4 * len(targets) > 40_000
    assert res["missed_by_order1"] == 0 and res["missed_by_order0"] == 0, res
    assert res["in_front_of_own_leaf"] == res["differ"], res
    # (these rays are the worst case for the OTHER order dependence: a vertex lies on the faces of its leaf's box, so a hit there is in
    # front of the box's entry distance about as often as behind it -- measured: 109 of 40,968 rays, all of them such hits)
    assert res["differ"] <= res["rays"] // 100, res
    # the rays do land on edges: u, v of the exact walk
    from clive2_amd import struct_types as st
    from clive2_amd.renderer import Renderer
    r = Renderer(scene)
    r.set_traversal_mode(5)
    rays = np.zeros(len(targets), dtype=st.Ray)
    rays["origin"][:, :3] = chunks[0][1]; rays["direction"][:, :3] = chunks[0][2]
    i, tt, u, v = r.probe_traverse(rays)
    r.close()
    on_edge = (i >= 0) & ((u == 0) | (v == 0) | (u + v == 1))
    assert on_edge.sum() > 100, int(on_edge.sum())


@pytest.mark.parametrize("max_members", [4, 2, 1])
def test_trees_with_another_leaf_size_render_bit_exactly(max_members, oracle_mod):
    """the builder's leaf size as an input (create_scene(max_members=...); the reference's constant is 8).
    The renderer and the oracle walk the SAME Box[]: subpaths, RNG state, aggregators and ray count agree bit for bit, with the
    binary walk (mode 2), the 4-wide walk (mode 5) and the automatic organisation."""
    from clive2_amd.renderer import Renderer, make_seeds
    scene = _glass(3, 96, 54, max_members=max_members, bvh_builder="native")
    leaves = scene.boxes[scene.boxes["right"] != 0]
    assert (leaves["right"] - leaves["left"]).max() <= max_members
    seeds = make_seeds(96 * 54)
    o = oracle_mod.OracleRenderer(scene, seeds=seeds)
    o.run_sample(); o.run_sample()
    for mode in (0, 2, 5):
        r = Renderer(scene, seeds=seeds)
        r.set_traversal_mode(mode)
        r.run_samples(2)
        assert np.array_equal(r.get_random_buffer(), o.rand_buffer), mode
        assert r.export_paths(LIGHT).tobytes() == o.out_light_paths.tobytes() and r.export_paths(CAMERA).tobytes() == o.out_camera_paths.tobytes(), mode
        agg = r.export_aggregators()
        for f in ("weights", "total_contribution", "contrib_weight_sum"):
            assert agg[f].tobytes() == o.weight_aggregators[f].tobytes(), (mode, f)
        assert r.counters()["rays"] == o.rays_traced
        r.close()
