"""Sample streams (cl2_set_sample_streams): a handle that carries K independent samples of the frame per pass.

The reference's Renderer owns ONE seed buffer (src/renderer.py:54, :86-87); K renderers -- the ranks of the sample split,
SURVEY 8e -- own K.  A K-stream handle must be exactly those K renderers side by side: stream k's RNG state, Path[] records
and filter aggregators equal, bit for bit, what the oracle renders from seed buffer k alone, and the accumulators hold the
sum over the streams (float sums: the tolerance of the light-image splat).  Launch sizes are K x W x H, so every launch
organisation is run again with K > 1, ragged frames included (a stream boundary inside a workgroup)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

LIGHT, CAMERA = 0, 1


def _oracles(scene, orc, K, samples):
    B = scene.pixel_width * scene.pixel_height
    out = []
    for k in range(K):
        o = orc.OracleRenderer(scene, seeds=orc.make_seeds(B, rank=k))
        for _ in range(samples):
            o.run_sample()
        out.append(o)
    return out


def _check_streams(r, oracles, samples, paths=True):
    K = len(oracles)
    seeds = r.get_random_buffer().reshape(K, -1, 2)
    for k, o in enumerate(oracles):
        assert np.array_equal(seeds[k], o.rand_buffer), f"stream {k}: RNG state"
        r.set_export_stream(k)
        if paths:
            assert r.export_paths(LIGHT).tobytes() == o.out_light_paths.tobytes(), f"stream {k}: light subpaths"
            assert r.export_paths(CAMERA).tobytes() == o.out_camera_paths.tobytes(), f"stream {k}: camera subpaths"
        agg = r.export_aggregators()
        for f in ("total_contribution", "weights", "contrib_weight_sum"):
            assert agg[f].tobytes() == o.weight_aggregators[f].tobytes(), f"stream {k}: aggregator {f}"
    img, wts, cnt, uni = r.read_accumulators()
    assert (cnt == K * samples).all() and r.samples == K * samples
    np.testing.assert_allclose(img, sum(o.summed_image for o in oracles), rtol=5e-5, atol=1e-8)
    np.testing.assert_allclose(wts, sum(o.summed_sample_weights for o in oracles), rtol=5e-5, atol=1e-8)
    np.testing.assert_allclose(uni, sum(o.unidirectional_image_buffer for o in oracles), rtol=2e-6, atol=0)
    assert r.counters()["rays"] == sum(o.rays_traced for o in oracles)


@pytest.mark.parametrize("scene_name", ["cornell_small", "glass_scene"])
def test_streams_equal_independent_renderers(scene_name, request, oracle_mod):
    """K = 3 on the LDS-resident scenes: three passes through run_samples, then one through the eight stage calls."""
    from clive2_amd.renderer import Renderer
    scene = request.getfixturevalue(scene_name)
    K = 3
    r = Renderer(scene, streams=K)
    assert r.organisation()["sample_streams"] == K
    r.run_samples(3)
    oracles = _oracles(scene, oracle_mod, K, 3)
    _check_streams(r, oracles, 3)
    r.make_light_rays(); r.make_camera_rays(); r.trace_light_rays(); r.trace_camera_rays()
    r.join_paths(); r.finalize_samples(); r.gather_light_image(); r.process_images()
    r.samples += K                                      # the stage calls do not count on the Python side (as in the reference)
    for o in oracles:
        o.run_sample()
    _check_streams(r, oracles, 4)
    r.close()


def _mesh_scene(w, h, subdiv=3):
    import clive2_amd as c2
    from clive2_amd.load import get_materials
    from clive2_amd.meshes import icosphere
    mats = get_materials()
    mats["alpha"][5] = 0.1
    v, f = icosphere(subdiv, radius=2.0, center=(0.0, 1.0, 0.0))
    return c2.create_scene(w, h, np.array([0, 1.5, 6]), np.array([0, 0, -1]),
                           file_specs=[dict(mesh=(v, f), material=5)], materials=mats)


@pytest.mark.parametrize("mode,stages,K", [(0, -1, 2), (2, 1, 3), (4, 0, 2), (5, 2, 4), (1, 1, 2), (3, 1, 2)])
def test_streams_in_every_launch_organisation(mode, stages, K, oracle_mod):
    """A tree that is read through the caches (1,296 triangles): persistent per-level launches, whole subpaths, the exact
    4-wide walk, fused kernels -- serial and pipelined -- each with K streams in every launch."""
    from clive2_amd.renderer import Renderer
    scene = _mesh_scene(80, 45)
    r = Renderer(scene, streams=K)
    r.set_traversal_mode(mode); r.set_pipelining(stages)
    assert not r.organisation()["tree_in_lds"]
    r.run_samples(3)
    _check_streams(r, _oracles(scene, oracle_mod, K, 3), 3)
    r.close()


@pytest.mark.parametrize("size,K", [((91, 60), 2), ((257, 1), 3), ((1, 1), 5), ((3, 2), 4)])
def test_streams_on_ragged_frames(size, K, oracle_mod):
    """Frames that are not a multiple of the 256-thread workgroup: a stream boundary then lies INSIDE a workgroup and a
    wave (entry k*W*H + p), and the last workgroup of the launch is partial."""
    import clive2_amd as c2
    from clive2_amd.renderer import Renderer
    w, h = size
    scene = c2.create_scene_from_preset("empty", w, h)
    r = Renderer(scene, streams=K)
    r.run_samples(3)
    _check_streams(r, _oracles(scene, oracle_mod, K, 3), 3)
    assert r.tone_mapped("image").shape == (h, w, 3)
    r.close()
    if size == (91, 60):
        scene = _mesh_scene(w, h, subdiv=2)                # 336 triangles: LDS-resident with a deep table; then the cached tree
        for mode in (0, 2):
            r = Renderer(scene, streams=K)
            r.set_traversal_mode(mode)
            r.run_samples(2)
            _check_streams(r, _oracles(scene, oracle_mod, K, 2), 2)
            r.close()


def test_stream_count_changes_and_error_behaviour(cornell_small, oracle_mod):
    """Switching the stream count re-allocates the per-pixel state and keeps the accumulators: 2 samples with one stream
    + 2 passes with two streams = the sum of the three oracle runs.  Refusals: too many entries for the 26-bit tag field,
    a seed buffer of the wrong size, an export stream out of range."""
    from clive2_amd.renderer import Renderer, RendererError, make_seeds
    B = cornell_small.pixel_width * cornell_small.pixel_height
    r = Renderer(cornell_small, seeds=make_seeds(B, rank=7))
    r.run_samples(2)
    o7 = oracle_mod.OracleRenderer(cornell_small, seeds=oracle_mod.make_seeds(B, rank=7))
    o7.run_sample(); o7.run_sample()
    r.set_sample_streams(2)
    assert (r.get_random_buffer() == 1).all()              # fresh state: the seeds must be set again
    with pytest.raises(RendererError):
        r.set_seeds(make_seeds(B))                         # one buffer for two streams
    r.set_seeds(np.stack([make_seeds(B, rank=0), make_seeds(B, rank=1)]))
    r.run_samples(2)
    os_ = _oracles(cornell_small, oracle_mod, 2, 2)
    img, wts, cnt, uni = r.read_accumulators()
    assert (cnt == 6).all() and r.samples == 6
    np.testing.assert_allclose(img, o7.summed_image + os_[0].summed_image + os_[1].summed_image, rtol=5e-5, atol=1e-8)
    with pytest.raises(RendererError):
        r.set_export_stream(2)
    with pytest.raises(RendererError):
        r.set_sample_streams(0)
    with pytest.raises(RendererError):
        r.set_sample_streams((1 << 26) // B + 1)
    r.set_export_stream(1)                                  # still usable after the refusals (not poisoned)
    assert r.export_paths(CAMERA).tobytes() == os_[1].out_camera_paths.tobytes()
    r.set_sample_streams(1)
    r.set_seeds(make_seeds(B, rank=7))
    r.reset_accumulators()
    r.run_samples(2)
    assert np.array_equal(r.get_random_buffer(), o7.rand_buffer)
    r.close()


def test_auto_streams_and_the_render_cli(tmp_path, cornell_small):
    """`streams="auto"`: 1 for an LDS-resident scene, else what brings a launch to about 2^24 entries (at most 8); the CLI takes
    `--sample-streams` (a number or auto), rounds the samples up to a multiple of it and writes the picture."""
    from clive2_amd.renderer import Renderer
    from clive2_amd import render
    from PIL import Image
    r = Renderer(cornell_small, streams="auto")
    assert r.streams == 1 and r.organisation()["sample_streams"] == 1
    r.close()
    r = Renderer(_mesh_scene(80, 45), streams="auto")
    assert r.streams == 8 and r.get_random_buffer().shape == (8, 80 * 45, 2)
    r.run_samples(1)
    assert (r.read_accumulators()[2] == 8).all()
    r.close()
    out = tmp_path / "c.png"
    assert render.main(["--scene", "empty", "--width", "64", "--height", "36", "--samples", "10", "--sample-streams", "4", "--out", str(out)]) == 0
    assert np.asarray(Image.open(out)).shape == (36, 64, 3)
    assert render.main(["--scene", "empty", "--width", "64", "--height", "36", "--samples", "3", "--sample-streams", "auto", "--out", str(out)]) == 0
