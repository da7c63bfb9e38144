"""Round 5: the reproducible light image (cl2_set_reproducible) and the 4-wide walk's own tallies (cl2_set_counting(2)) --
through the C ABI, against the oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

LIGHT, CAMERA = 0, 1


def _pair(scene, orc, seeds=None, **kw):
    from clive2_amd.renderer import Renderer, make_seeds
    B = scene.pixel_width * scene.pixel_height
    seeds = make_seeds(B) if seeds is None else seeds
    return Renderer(scene, seeds=seeds, **kw), orc.OracleRenderer(scene, seeds=seeds)


@pytest.mark.parametrize("mode", [0, 2, 5])
@pytest.mark.parametrize("scene_name", ["cornell_small", "glass_scene"])
def test_reproducible_light_image_is_bitwise_the_oracles_stable_chain(scene_name, mode, request, oracle_mod):
    """With the switch on, NOTHING in the pipeline needs a tolerance: the light image is the reference's K8 (trace.metal:937-964)
    over the reference's records in slot order (a stable sort by target pixel instead of the bitonic network), and that is what
    the oracle computes with `stable=True` -- per-sample light image, sample weights and all four accumulators byte for byte,
    over three samples.  Two renders of the product agree byte for byte as well."""
    scene = request.getfixturevalue(scene_name)
    packed = []
    for attempt in range(2):
        r, o = _pair(scene, oracle_mod)
        r.set_traversal_mode(mode)
        r.set_reproducible(True)
        for x in (r, o):
            x.make_light_rays(); x.make_camera_rays(); x.trace_light_rays(); x.trace_camera_rays()
            x.join_paths(); x.finalize_samples()
        r.gather_light_image(); o.gather_light_image(stable=True)
        imgs = r.export_sample_images()
        assert (o.out_light_image[:, :3] > 0).any()
        assert imgs["light"][:, :3].tobytes() == o.out_light_image[:, :3].tobytes()
        r.process_images(); o.process_images()
        r.run_samples(2)
        o.run_sample(stable_light_sort=True); o.run_sample(stable_light_sort=True)
        img, wts, cnt, uni = r.read_accumulators()
        assert img.tobytes() == o.summed_image.tobytes()
        assert wts.tobytes() == o.summed_sample_weights.tobytes()
        assert np.array_equal(cnt, o.summed_sample_counts)
        assert uni.tobytes() == o.unidirectional_image_buffer.tobytes()
        packed.append(r.packed_accumulators().tobytes())
        r.close()
    assert packed[0] == packed[1]


def test_reproducible_light_image_with_sample_streams_and_ragged_frame(oracle_mod):
    """Two sample streams on a frame that is not a multiple of 64 pixels: every stream's records go to its own light image and
    the accumulators receive the streams in order -- equal, byte for byte, to two oracle renderers added in stream order; the
    default (atomic) path agrees with it to the tolerance the atomics need."""
    import clive2_amd as c2
    from clive2_amd.renderer import Renderer, stream_seeds
    scene = c2.create_scene_from_preset("empty", 37, 23)
    B = 37 * 23
    seeds = stream_seeds(B, 2)
    r = Renderer(scene, seeds=seeds, streams=2)
    r.set_reproducible(True)
    r.run_samples(2)
    # the oracle renderers add their samples to ONE pair of accumulators in the product's order: pass by pass, stream by stream
    os_ = [oracle_mod.OracleRenderer(scene, seeds=seeds[k]) for k in range(2)]
    img = np.zeros((23, 37, 3), np.float32); wts = np.zeros((23, 37, 1), np.float32)
    for _ in range(2):
        for o in os_:
            o.summed_image[:] = img; o.summed_sample_weights[:] = wts
            o.run_sample(stable_light_sort=True)
            img, wts = o.summed_image.copy(), o.summed_sample_weights.copy()
    gi, gw, gc, _ = r.read_accumulators()
    assert gi.tobytes() == img.tobytes() and gw.tobytes() == wts.tobytes() and (gc == 4).all()
    r2 = Renderer(scene, seeds=seeds, streams=2)
    r2.run_samples(2)
    ai, aw, _, _ = r2.read_accumulators()
    np.testing.assert_allclose(ai, gi, rtol=5e-5, atol=1e-8)
    np.testing.assert_allclose(aw, gw, rtol=5e-5, atol=1e-8)


def test_walk_tallies_of_the_wide_walk(glass_scene, oracle_mod):
    """cl2_set_counting(2): the 4-wide walk counts what it fetches.  Every ray of the pass is tallied (connection and subpath
    launches apart), a ray visits fewer wide nodes than the reference's walk tests boxes, reads at least the triangle records the
    reference tests (a pair re-reads nothing, an odd leaf's last record is counted once), and the render is unchanged."""
    r, o = _pair(glass_scene, oracle_mod)
    r.set_traversal_mode(5)
    r.set_counting(True)
    r.run_samples(1)
    c1 = r.counters()
    r.reset_counters()
    r.set_counting(2)
    r.run_samples(1)
    c2 = r.counters()
    t = r.walk_tallies()
    assert t["connection"]["rays"] == c2["rays_traverse_conn"] > 0
    assert t["subpath"]["rays"] == c2["rays_traverse_paths"] > 0
    rays = t["connection"]["rays"] + t["subpath"]["rays"]
    visits = t["connection"]["wide_visits"] + t["subpath"]["wide_visits"]
    tris = t["connection"]["tri_records"] + t["subpath"]["tri_records"]
    n_node, n_tri = c1["box_tests"] / c1["counted_rays"], c1["tri_tests"] / c1["counted_rays"]
    assert 0 < visits / rays < n_node
    assert 0.9 * n_tri < tris / rays < 1.1 * n_tri          # two different samples: same distribution, not the same rays
    r.set_counting(False)
    o.run_sample(); o.run_sample()
    r2, _ = _pair(glass_scene, oracle_mod)
    r2.set_traversal_mode(5)
    r2.run_samples(2)
    assert r2.export_aggregators()["total_contribution"].tobytes() == r.export_aggregators()["total_contribution"].tobytes() \
        == o.weight_aggregators["total_contribution"].tobytes()
    with pytest.raises(Exception):
        r.set_counting(3)


@pytest.mark.parametrize("mode", [4, 5])
def test_speculative_stack_top_expansion_is_a_pure_performance_knob(mode, glass_scene, oracle_mod):
    """Round 5: a lane of the 4-wide walk that is testing triangles expands the wide node on top of its stack in the same pass
    (csrc/bvh_wide.hpp).  With and without it (debug bit 13) -- and in the whole-subpath launch, which always has it -- subpaths,
    RNG state, aggregators and ray count equal the oracle's, over one serial and two pipelined samples."""
    outs = []
    for flags in (0, 1 << 13):
        r, o = _pair(glass_scene, oracle_mod)
        r.set_traversal_mode(mode)
        r.set_debug_flags(flags)
        r.run_samples(1); o.run_sample()
        assert r.export_paths(LIGHT).tobytes() == o.out_light_paths.tobytes()
        assert r.export_paths(CAMERA).tobytes() == o.out_camera_paths.tobytes()
        r.run_samples(2); o.run_sample(); o.run_sample()
        assert np.array_equal(r.get_random_buffer(), o.rand_buffer)
        agg = r.export_aggregators()
        for f in ("weights", "total_contribution", "contrib_weight_sum"):
            assert agg[f].tobytes() == o.weight_aggregators[f].tobytes(), (flags, f)
        assert r.counters()["rays"] == o.rays_traced
        outs.append(r.export_paths(CAMERA).tobytes())
        r.close()
    assert outs[0] == outs[1]
