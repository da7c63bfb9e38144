"""Differential fuzzing in the GPU suite: a dozen random scenes (camera pose, meshes, all material types,
odd frame sizes, both BVH builders, every launch organisation) -- HIP path vs oracle, bit for bit.
The full run (`python tools/fuzz_parity.py 150`) is logged in profiles/r01_fuzz_parity.log."""
import importlib.util
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load_tool():
    spec = importlib.util.spec_from_file_location("fuzz_parity", os.path.join(ROOT, "tools", "fuzz_parity.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.mark.parametrize("seed", list(range(500, 512)))
def test_random_scene_matches_oracle(seed, oracle_mod):
    from clive2_amd.renderer import Renderer, make_seeds
    fz = _load_tool()
    rng = np.random.RandomState(1000 + seed)
    scene, desc = fz.random_scene(rng)
    B = scene.pixel_width * scene.pixel_height
    seeds = make_seeds(B, seed=seed)
    r, o = Renderer(scene, seeds=seeds), oracle_mod.OracleRenderer(scene, seeds=seeds)
    r.set_traversal_mode(int(rng.randint(0, 3)))
    r.set_levels_per_launch(int(rng.randint(1, 7)))
    for x in (r, o):
        x.make_light_rays(); x.make_camera_rays(); x.trace_light_rays(); x.trace_camera_rays()
    assert r.export_paths(0).tobytes() == o.out_light_paths.tobytes(), desc
    assert r.export_paths(1).tobytes() == o.out_camera_paths.tobytes(), desc
    for x in (r, o):
        x.join_paths(); x.finalize_samples(); x.gather_light_image(); x.process_images()
    agg = r.export_aggregators()
    assert agg["total_contribution"].tobytes() == o.weight_aggregators["total_contribution"].tobytes(), desc
    assert agg["weights"].tobytes() == o.weight_aggregators["weights"].tobytes(), desc
    np.testing.assert_allclose(r.read_accumulators()[0], o.summed_image, rtol=5e-5, atol=1e-8)
    assert r.counters()["rays"] == o.rays_traced
