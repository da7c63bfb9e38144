"""GPU <-> oracle parity, stage by stage, through the C ABI (ctypes -> libclive2_amd.so).

Bar: BIT-EXACT for everything that is a deterministic function of the inputs (rays, closest
hits, subpaths, RNG state, filter aggregators, finalized samples, unidirectional estimate);
rtol 1e-5 where float atomics reorder a sum (the t=1 light-image splat and what it feeds).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

LIGHT, CAMERA = 0, 1


def _pair(scene, orc, seeds=None):
    from clive2_amd.renderer import Renderer, make_seeds
    B = scene.pixel_width * scene.pixel_height
    seeds = make_seeds(B) if seeds is None else seeds
    return Renderer(scene, seeds=seeds), orc.OracleRenderer(scene, seeds=seeds)


def _diff_fields(got, ref, dt):
    bad = []
    for name in dt.names:
        if got[name].tobytes() != ref[name].tobytes():
            bad.append(name)
    return bad


def _run_to_paths(r, o):
    for x in (r, o):
        x.make_light_rays(); x.make_camera_rays(); x.trace_light_rays(); x.trace_camera_rays()


def _run_rest(r, o):
    for x in (r, o):
        x.join_paths(); x.finalize_samples(); x.gather_light_image(); x.process_images()


@pytest.mark.parametrize("scene_name", ["cornell_small", "glass_scene"])
def test_generated_rays_bit_exact(scene_name, request, oracle_mod):
    from clive2_amd import struct_types as st
    scene = request.getfixturevalue(scene_name)
    r, o = _pair(scene, oracle_mod)
    r.make_light_rays(); o.make_light_rays()
    assert _diff_fields(r.export_rays(LIGHT), o.light_ray_buffer, st.Ray) == []
    r.make_camera_rays(); o.make_camera_rays()
    assert _diff_fields(r.export_rays(CAMERA), o.camera_ray_buffer, st.Ray) == []
    assert np.array_equal(r.get_random_buffer(), o.rand_buffer)


@pytest.mark.parametrize("scene_name", ["cornell_small", "glass_scene"])
@pytest.mark.parametrize("mode", [0, 5])
def test_closest_hit_bit_exact(scene_name, mode, request, oracle_mod):
    """traverse_bvh alone: camera rays, light rays and axis-parallel rays (inv_direction = +-inf) -- through the
    one-ray-per-lane kernel (mode 0) and through the exact 4-wide walk (5: the lanes of the axis-parallel rays run the
    binary walk inside that launch, the others the wide walk)."""
    from clive2_amd import struct_types as st
    scene = request.getfixturevalue(scene_name)
    r, o = _pair(scene, oracle_mod)
    r.set_traversal_mode(mode)
    if mode == 5:
        assert r.organisation()["wide_nodes"] > 0
    o.make_light_rays(); o.make_camera_rays()
    axis = np.zeros(6 * 64, dtype=st.Ray)
    rng = np.random.RandomState(5)
    axis["origin"][:, :3] = rng.uniform(-9, 9, size=(len(axis), 3)).astype(np.float32)
    axis["origin"][::4, 0] = -10.0        # start exactly on a slab plane: 0 * inf = NaN in the box test
    for k in range(6):
        d = np.zeros(3, np.float32); d[k % 3] = 1.0 if k < 3 else -1.0
        axis["direction"][k::6, :3] = d
    with np.errstate(divide="ignore"):
        axis["inv_direction"][:, :3] = np.float32(1.0) / axis["direction"][:, :3]
    rays = np.concatenate([o.camera_ray_buffer, o.light_ray_buffer, axis])
    bi, bt, u, v = r.probe_traverse(rays)
    rbi, rbt, ru, rv, _ = oracle_mod.traverse(rays, scene.boxes, scene.triangles)
    assert np.array_equal(bi, rbi)
    hit = rbi >= 0
    assert hit.sum() > 0.9 * len(o.camera_ray_buffer)
    assert bt[hit].tobytes() == rbt[hit].tobytes()
    assert u[hit].tobytes() == ru[hit].tobytes() and v[hit].tobytes() == rv[hit].tobytes()


@pytest.mark.parametrize("scene_name", ["cornell_small", "glass_scene"])
def test_subpaths_bit_exact(scene_name, request, oracle_mod):
    from clive2_amd import struct_types as st
    scene = request.getfixturevalue(scene_name)
    r, o = _pair(scene, oracle_mod)
    _run_to_paths(r, o)
    for which, ref in ((LIGHT, o.out_light_paths), (CAMERA, o.out_camera_paths)):
        got = r.export_paths(which)
        assert np.array_equal(got["length"], ref["length"])
        assert _diff_fields(got["rays"], ref["rays"], st.Ray) == []
        assert got.tobytes() == ref.tobytes()
    assert np.array_equal(r.get_random_buffer(), o.rand_buffer)
    c = r.counters()
    assert c["rays"] == o.rays_traced
    if scene_name == "glass_scene":    # glass must actually be exercised: some paths end early
        assert (o.out_camera_paths["length"] < 6).any()


@pytest.mark.parametrize("scene_name", ["cornell_small", "glass_scene"])
def test_connection_stage(scene_name, request, oracle_mod):
    scene = request.getfixturevalue(scene_name)
    r, o = _pair(scene, oracle_mod)
    _run_to_paths(r, o)
    r.join_paths(); o.join_paths()
    agg = r.export_aggregators()
    for f in ("weights", "total_contribution", "contrib_weight_sum"):
        assert agg[f].tobytes() == o.weight_aggregators[f].tobytes(), f
    assert r.counters()["rays"] == o.rays_traced          # same number of closest-hit queries
    r.finalize_samples(); o.finalize_samples()
    imgs = r.export_sample_images()
    assert imgs["finalized"][:, :3].tobytes() == o.finalized_samples[:, :3].tobytes()
    assert imgs["sample_weights"].tobytes() == o.sample_weights.tobytes()
    assert imgs["unidirectional"].tobytes() == o.out_camera_image.tobytes()
    r.gather_light_image(); o.gather_light_image()
    # the light image is a sum in a different order (atomics vs sorted segments)
    np.testing.assert_allclose(imgs["light"][:, :3], o.out_light_image[:, :3], rtol=2e-5, atol=1e-9)
    assert (o.out_light_image[:, :3] > 0).any()
    r.process_images(); o.process_images()
    img, wts, cnt, uni = r.read_accumulators()
    np.testing.assert_allclose(img, o.summed_image, rtol=2e-5, atol=1e-9)
    np.testing.assert_allclose(wts, o.summed_sample_weights, rtol=2e-5, atol=1e-9)
    assert np.array_equal(cnt, o.summed_sample_counts)
    assert uni.tobytes() == o.unidirectional_image_buffer.tobytes()


@pytest.mark.parametrize("scene_name", ["cornell_small", "glass_scene"])
def test_multi_sample_accumulation(scene_name, request, oracle_mod):
    """run_samples(n) == n x run_sample of the oracle: RNG state carried across samples, accumulators."""
    scene = request.getfixturevalue(scene_name)
    r, o = _pair(scene, oracle_mod)
    r.run_samples(4)
    for _ in range(4):
        o.run_sample()
    assert np.array_equal(r.get_random_buffer(), o.rand_buffer)
    img, wts, cnt, uni = r.read_accumulators()
    np.testing.assert_allclose(img, o.summed_image, rtol=5e-5, atol=1e-8)
    np.testing.assert_allclose(wts, o.summed_sample_weights, rtol=5e-5, atol=1e-8)
    assert (cnt == 4).all()
    np.testing.assert_allclose(uni, o.unidirectional_image_buffer, rtol=1e-6, atol=0)
    # north-star metric: per-pixel L2 of the radiance image < 1e-3
    l2 = np.sqrt(((r.radiance - o.radiance) ** 2).sum(axis=2))
    assert l2.max() < 1e-3
    assert r.counters()["rays"] == o.rays_traced


@pytest.mark.parametrize("levels", [1, 2, 3, 4])
def test_levels_per_launch_is_a_pure_performance_knob(levels, glass_scene, oracle_mod):
    """Subpaths traced 1/2/3/4 bounces per launch (queue compaction in between) == one launch == oracle."""
    from clive2_amd import struct_types as st
    r, o = _pair(glass_scene, oracle_mod)
    r.set_levels_per_launch(levels)
    _run_to_paths(r, o)
    for which, ref in ((LIGHT, o.out_light_paths), (CAMERA, o.out_camera_paths)):
        assert r.export_paths(which).tobytes() == ref.tobytes()
    assert np.array_equal(r.get_random_buffer(), o.rand_buffer)
    assert r.counters()["rays"] == o.rays_traced
    with pytest.raises(Exception):
        r.set_levels_per_launch(7)


@pytest.mark.parametrize("mode", [2, 4, 5])
@pytest.mark.parametrize("flags", [(0, 0, 0), (1 << 12, 0, 0), (0, 1, 1), (1 << 12, 64, 100)])
@pytest.mark.parametrize("scene_name", ["cornell_small", "glass_scene"])
def test_persistent_traversal_mode_is_equivalent(scene_name, flags, mode, request, oracle_mod):
    """traversal_mode 2 (persistent launches with lane-level ray replacement + one bounce launch per
    level) and 4 (whole subpaths, light then camera, in ONE persistent launch with the bounces batched per
    wave: the large-scene organisation) reproduce the oracle exactly, like the fused mode -- in
    both forms of the step (two triangles per step for cache-resident trees, bit 12 selects the
    one-triangle form used for trees that stream from memory), and for any bounce batching (cl2_set_subpath_gather: lanes gathered, steps waited) of the whole-subpath launch."""
    scene = request.getfixturevalue(scene_name)
    r, o = _pair(scene, oracle_mod)
    r.set_traversal_mode(mode)
    r.set_debug_flags(flags[0])
    r.set_subpath_gather(flags[1], flags[2])
    _run_to_paths(r, o)
    for which, ref in ((LIGHT, o.out_light_paths), (CAMERA, o.out_camera_paths)):
        assert r.export_paths(which).tobytes() == ref.tobytes()
    assert np.array_equal(r.get_random_buffer(), o.rand_buffer)
    r.join_paths(); o.join_paths()
    agg = r.export_aggregators()
    assert agg["total_contribution"].tobytes() == o.weight_aggregators["total_contribution"].tobytes()
    assert r.counters()["rays"] == o.rays_traced
    r.set_counting(True)
    r.run_samples(1)
    c = r.counters()
    assert c["box_tests"] > 0 and c["tri_tests"] > 0


def test_exact_reciprocal_and_div_pi_proof(cornell_small):
    """rcp_exact / div_pi (csrc/vecmath.hpp) return the IEEE-correct bits for all 2^32 inputs."""
    from clive2_amd.renderer import Renderer
    assert Renderer(cornell_small).selftest_exact_math() == (0, 0)


@pytest.mark.parametrize("mode", [1, 2])
def test_oversized_leaves_and_big_material_table(mode, oracle_mod):
    """Hand-made tree: root with two leaves of 48 triangles each (a reference leaf may exceed 8 when
    the builder's depth limit trips, bvh.py:294) -> each leaf spans several 16-triangle device
    records; plus a 40-entry material table (beyond the LDS-staged 32)."""
    import clive2_amd as c2
    from clive2_amd import struct_types as st
    from clive2_amd.load import get_materials
    from clive2_amd.meshes import icosphere
    from clive2_amd.renderer import Renderer, make_seeds
    v, f = icosphere(1, radius=2.0, center=(0.0, 1.0, 0.0))
    mats = np.zeros(40, dtype=st.Material)
    mats[:8] = get_materials()
    mats[8:] = mats[4]
    mats["alpha"][5] = 0.2
    scene = c2.create_scene(48, 32, np.array([0, 1.5, 6]), np.array([0, 0, -1]),
                            file_specs=[dict(mesh=(v, f), material=5)], materials=mats)
    n = len(scene.triangles)
    assert n == 96
    tri = scene.triangles
    tri["material"][tri["material"] == 4] = 39                      # use a material beyond the LDS cap
    boxes = np.zeros(3, dtype=st.Box)
    pts = np.stack([tri["v0"], tri["v1"], tri["v2"]])[..., :3]
    for i, (a, b) in enumerate(((0, n), (0, 48), (48, n))):
        boxes["min"][i, :3] = pts[:, a:b].min(axis=(0, 1))
        boxes["max"][i, :3] = pts[:, a:b].max(axis=(0, 1))
    boxes["left"][0], boxes["right"][0] = 1, 0
    boxes["left"][1], boxes["right"][1] = 0, 48
    boxes["left"][2], boxes["right"][2] = 48, n
    scene.boxes = boxes
    assert scene.validate()
    seeds = make_seeds(48 * 32)
    r, o = Renderer(scene, seeds=seeds), oracle_mod.OracleRenderer(scene, seeds=seeds)
    r.set_traversal_mode(mode)
    _run_to_paths(r, o)
    for which, ref in ((LIGHT, o.out_light_paths), (CAMERA, o.out_camera_paths)):
        assert r.export_paths(which).tobytes() == ref.tobytes()
    r.join_paths(); o.join_paths()
    assert r.export_aggregators()["total_contribution"].tobytes() == o.weight_aggregators["total_contribution"].tobytes()
    assert r.counters()["rays"] == o.rays_traced
    r.set_counting(True); r.run_samples(1)
    c = r.counters()
    # 3 reference boxes -> 1 + 3 + 3 device records; box tests count records, triangle tests match the reference
    assert c["tri_tests"] > 0 and c["box_tests"] >= c["counted_rays"]


@pytest.mark.parametrize("size", [(1, 1), (1, 7), (257, 1), (3, 2)])
def test_tiny_and_ragged_frames(size, oracle_mod):
    """Frames of one pixel, one column, one row of 257 pixels (one full workgroup + 1): a single partial workgroup, wave-level
    batches with most lanes behind the end of the frame (the out-of-bounds read the round-3 fuzz found lived there).  Three
    samples through the pipeline: seeds, last sample's subpaths, image and ray tally as the oracle's; the device tone map runs."""
    import clive2_amd as c2
    from clive2_amd.renderer import Renderer, make_seeds
    w, h = size
    scene = c2.create_scene_from_preset("empty", w, h)
    seeds = make_seeds(w * h)
    r, o = Renderer(scene, seeds=seeds), oracle_mod.OracleRenderer(scene, seeds=seeds)
    r.run_samples(3)
    for _ in range(3):
        o.run_sample()
    assert np.array_equal(r.get_random_buffer(), o.rand_buffer)
    assert r.export_paths(LIGHT).tobytes() == o.out_light_paths.tobytes()
    assert r.export_paths(CAMERA).tobytes() == o.out_camera_paths.tobytes()
    assert np.allclose(r.read_accumulators()[0], o.summed_image, rtol=5e-5, atol=1e-8)
    assert r.counters()["rays"] == o.rays_traced
    assert r.tone_mapped("image").shape == (h, w, 3)
    r.close()


@pytest.mark.parametrize("size", [(91, 60), (257, 1)])
def test_ragged_frames_with_short_subpaths(size, oracle_mod):
    """The same frames over an OPEN scene with a rough-glass ball (emitter, floor, back wall only: most subpaths end
    after a bounce or two, so the wave-level `some lane has vertex slot v` ballots of k_connect_setup / k_connect_resolve are
    sparse and differ from wave to wave) at 91x60 = 21 x 256 + 84 pixels -- the frame of the round-3 fault -- and at one row
    of 257.  The contract those kernels keep for the lanes behind the frame's end is clamped_pid() (csrc/kernels.hpp)."""
    import clive2_amd as c2
    from clive2_amd.load import get_materials
    from clive2_amd.meshes import icosphere
    from clive2_amd.renderer import Renderer, make_seeds
    w, h = size
    mats = get_materials()
    mats["alpha"][5] = 0.1
    from clive2_amd.load import triangles_for_box
    keep = [t for t in triangles_for_box() if t.emitter or t.n[1] > 0.5 or t.n[2] > 0.5]          # emitter, floor, back wall
    scene = c2.create_scene(w, h, np.array([0, 1.5, 6]), np.array([0, 0, -1]), room=keep, materials=mats,
                            file_specs=[dict(mesh=icosphere(3, radius=1.5), material=5, offset=np.array([0.5, 0.0, -1.0]))])
    assert len(scene.triangles) > 512                                      # not LDS-resident: mode 0 = the persistent organisation
    seeds = make_seeds(w * h)
    for mode in (0, 1):                                     # persistent organisation (the tree is not LDS-resident) and one ray per lane
        r, o = Renderer(scene, seeds=seeds), oracle_mod.OracleRenderer(scene, seeds=seeds)
        r.set_traversal_mode(mode)
        r.run_samples(3)
        for _ in range(3):
            o.run_sample()
        lens = o.out_camera_paths["length"]
        assert (lens <= 2).mean() > 0.2 and (lens >= 4).any()          # short subpaths dominate, long ones exist
        assert np.array_equal(r.get_random_buffer(), o.rand_buffer)
        assert r.export_paths(LIGHT).tobytes() == o.out_light_paths.tobytes()
        assert r.export_paths(CAMERA).tobytes() == o.out_camera_paths.tobytes()
        agg = r.export_aggregators()
        assert agg["total_contribution"].tobytes() == o.weight_aggregators["total_contribution"].tobytes()
        assert np.allclose(r.read_accumulators()[0], o.summed_image, rtol=5e-5, atol=1e-8)
        assert r.counters()["rays"] == o.rays_traced
        r.close()


def test_more_camera_triangles_than_kernel_arguments(oracle_mod):
    """The resolve kernel takes the is_camera triangles as kernel arguments (up to 4: the reference's scenes have the 2 of
    the film quad) and falls back to the look-up in the shading records beyond.  Six flagged triangles (the film quad and
    two walls -- a scene no preset makes, but a legal Triangle[]): subpaths, aggregators, light splats and ray tally as the
    oracle's; and a scene with exactly 4."""
    import copy
    import clive2_amd as c2
    from clive2_amd.renderer import Renderer, make_seeds
    for extra in (4, 2):
        scene = copy.deepcopy(c2.create_scene_from_preset("empty", 48, 32))
        tri = scene.triangles
        walls = [i for i in range(len(tri)) if not tri["is_camera"][i] and not tri["is_light"][i]][:extra]
        tri["is_camera"][walls] = 1
        assert int(tri["is_camera"].sum()) == 2 + extra
        seeds = make_seeds(48 * 32)
        r, o = Renderer(scene, seeds=seeds), oracle_mod.OracleRenderer(scene, seeds=seeds)
        _run_to_paths(r, o)
        for which, ref in ((LIGHT, o.out_light_paths), (CAMERA, o.out_camera_paths)):
            assert r.export_paths(which).tobytes() == ref.tobytes()
        _run_rest(r, o)
        agg = r.export_aggregators()
        assert agg["total_contribution"].tobytes() == o.weight_aggregators["total_contribution"].tobytes()
        assert agg["weights"].tobytes() == o.weight_aggregators["weights"].tobytes()
        img, wts, _, _ = r.read_accumulators()
        assert np.allclose(img, o.summed_image, rtol=5e-5, atol=1e-8)              # the t = 1 splats (float atomics)
        assert np.allclose(wts, o.summed_sample_weights, rtol=5e-5, atol=1e-8)
        assert r.counters()["rays"] == o.rays_traced
        r.close()


def test_all_material_types_bit_exact(oracle_mod):
    """Material types the shipped table never reaches: type 1 with alpha 0 (smooth
    dielectric through the general GGX route), type 2 (Fresnel-weighted reflect / diffuse), type >= 3
    (always reflect), each on its own sphere, alpha 0 / 0.3 / 0.05 (trace.metal:474-487)."""
    import clive2_amd as c2
    from clive2_amd import struct_types as st
    from clive2_amd.load import get_materials
    from clive2_amd.meshes import icosphere
    from clive2_amd.renderer import Renderer, make_seeds
    mats = np.zeros(10, dtype=st.Material)
    mats[:8] = get_materials()
    mats[8], mats[9] = mats[5], mats[5]
    mats["type"][8], mats["alpha"][8] = 2, 0.3
    mats["type"][9], mats["alpha"][9] = 3, 0.05
    mats["color"][9, :3] = (0.9, 0.9, 0.9)
    specs = [dict(mesh=icosphere(2, radius=1.6, center=(x, 0.0, z)), material=m)
             for x, z, m in ((-3.5, 0.0, 0), (0.0, -1.0, 8), (3.5, 0.0, 9))]
    scene = c2.create_scene(96, 64, np.array([0, 1.5, 6]), np.array([0, 0, -1]), file_specs=specs, materials=mats)
    seeds = make_seeds(96 * 64)
    r, o = Renderer(scene, seeds=seeds), oracle_mod.OracleRenderer(scene, seeds=seeds)
    _run_to_paths(r, o)
    hit_mats = o.out_camera_paths["rays"]["material"][:, 1][o.out_camera_paths["length"] > 1]
    for m in (0, 8, 9):
        assert (hit_mats == m).sum() > 50, m                          # every sphere is seen
    for which, ref in ((LIGHT, o.out_light_paths), (CAMERA, o.out_camera_paths)):
        assert r.export_paths(which).tobytes() == ref.tobytes()
    _run_rest(r, o)
    assert r.export_aggregators()["total_contribution"].tobytes() == o.weight_aggregators["total_contribution"].tobytes()
    r.run_samples(2); o.run_sample(); o.run_sample()
    assert np.array_equal(r.get_random_buffer(), o.rand_buffer)
    np.testing.assert_allclose(r.read_accumulators()[0], o.summed_image, rtol=5e-5, atol=1e-8)


def test_oblique_turntable_camera_and_odd_frame(oracle_mod):
    """Camera not axis-aligned (turntable frame 1 of 8, scene.py:223-245) and a frame whose pixel
    count is not a multiple of the workgroup size (101 x 57)."""
    import clive2_amd as c2
    scene = c2.create_scene_from_preset_with_params("empty", 101, 57, frame_idx=1, total_frames=8)
    r, o = _pair(scene, oracle_mod)
    r.make_light_rays(); o.make_light_rays(); r.make_camera_rays(); o.make_camera_rays()
    assert r.export_rays(CAMERA).tobytes() == o.camera_ray_buffer.tobytes()
    r.trace_light_rays(); o.trace_light_rays(); r.trace_camera_rays(); o.trace_camera_rays()
    for which, ref in ((LIGHT, o.out_light_paths), (CAMERA, o.out_camera_paths)):
        assert r.export_paths(which).tobytes() == ref.tobytes()
    _run_rest(r, o)
    agg = r.export_aggregators()
    for f in ("weights", "total_contribution", "contrib_weight_sum"):
        assert agg[f].tobytes() == o.weight_aggregators[f].tobytes(), f
    img = r.read_accumulators()[0]
    np.testing.assert_allclose(img, o.summed_image, rtol=5e-5, atol=1e-8)
    assert (o.out_light_image[:, :3] > 0).any()                  # light paths do project onto the oblique film


def test_device_detmath_equals_oracle_detmath(cornell_small, oracle_mod):
    """csrc/detmath.hpp vs oracle/detmath.h on ~1e7 inputs per function: every 97th binary32 value of
    the ranges the tracer uses, plus specials.  Bitwise equal (NaNs compared as NaNs)."""
    from clive2_amd.renderer import Renderer
    r = Renderer(cornell_small)

    def floats_between(lo, hi, stride):
        a, b = np.float32(lo).view(np.uint32), np.float32(hi).view(np.uint32)
        return np.arange(int(a), int(b) + 1, stride, dtype=np.uint64).astype(np.uint32).view(np.float32)

    special = np.array([0.0, -0.0, 1.0, -1.0, np.inf, -np.inf, np.nan, 1e-45, 1e-38, 3.4e38, 0.5, 2.414213562373095, 88.0, -87.0],
                       dtype=np.float32)
    cases = {
        "sin": np.concatenate([floats_between(0.0, 6.2831855, 97), -floats_between(1e-3, 6.2831855, 977), special]),
        "cos": np.concatenate([floats_between(0.0, 6.2831855, 97), -floats_between(1e-3, 6.2831855, 977), special]),
        "acos": np.concatenate([floats_between(0.0, 1.0, 97), -floats_between(1e-6, 1.0, 977), special]),
        "asin": np.concatenate([floats_between(0.0, 1.0, 97), special]),
        "atan": np.concatenate([floats_between(0.0, 1e6, 197), -floats_between(1e-6, 100.0, 977), special]),
        "exp": np.concatenate([-floats_between(1e-8, 90.0, 97), floats_between(0.0, 10.0, 977), special]),
    }
    for name, x in cases.items():
        got, ref = r.probe_math(name, x), oracle_mod.det_math(name, x)
        same = (got.view(np.uint32) == ref.view(np.uint32)) | (np.isnan(got) & np.isnan(ref))
        assert same.all(), (name, x[~same][:5], got[~same][:5], ref[~same][:5])
        assert len(x) > 1_000_000
    # the exact reciprocal / x-over-pi against numpy's IEEE division
    x = np.concatenate([floats_between(1e-30, 1e30, 389), -floats_between(1e-30, 1e30, 3889), special])
    with np.errstate(divide="ignore", invalid="ignore", over="ignore", under="ignore"):
        for name, ref in (("rcp", np.float32(1.0) / x), ("div_pi", x / np.float32(3.14159265359))):
            got = r.probe_math(name, x)
            same = (got.view(np.uint32) == ref.view(np.uint32)) | (np.isnan(got) & np.isnan(ref))
            assert same.all(), name


@pytest.mark.parametrize("from_camera", [0, 1])
def test_device_bounce_routines_equal_oracle(from_camera, cornell_small, oracle_mod):
    """GGX_sample, degreve_fresnel, diffuse / reflect / transmit bounce (trace.metal:226-379) on 4e5 random
    configurations -- random frames, grazing and back-facing incidence, both sides of an ior-1.5
    interface (incl. total internal reflection), alpha from 0 to 1 -- bit for bit."""
    import ctypes as C
    from clive2_amd.renderer import Renderer
    rng = np.random.RandomState(11 + from_camera)
    n = 400_000
    def unit(v):
        return v / np.linalg.norm(v, axis=1, keepdims=True)
    nn = unit(rng.normal(size=(n, 3)))
    wi = unit(nn * rng.uniform(-0.2, 1.0, size=(n, 1)) + 0.7 * rng.normal(size=(n, 3)))
    items = np.zeros((n, 12), np.float32)
    items[:, 0:3], items[:, 3:6] = wi, nn
    items[:, 6:8] = rng.rand(n, 2)
    inside = rng.rand(n) < 0.5
    items[:, 8] = np.where(inside, 1.5, 1.0)
    items[:, 9] = np.where(inside, 1.0, 1.5)
    items[:, 10] = rng.choice([0.0, 0.0, 0.05, 0.1, 0.3, 1.0], size=n)
    items[:, 11] = rng.randint(0, 4, size=n)
    items[:64, 6:8] = rng.choice([0.0, 1.0], size=(64, 2))          # the closed ends of the RNG interval
    got = Renderer(cornell_small).probe_bounce(items, from_camera=from_camera)
    ref = np.empty_like(got)
    oracle_mod.lib().orc_bounce_batch(C.c_int(n), C.c_int(from_camera), items.ctypes.data_as(C.c_void_p),
                                      ref.ctypes.data_as(C.c_void_p))
    same = (got.view(np.uint32) == ref.view(np.uint32)) | (np.isnan(got) & np.isnan(ref))
    bad = np.flatnonzero(~same.all(axis=1))
    assert len(bad) == 0, (len(bad), items[bad[:3]], got[bad[:3]], ref[bad[:3]])
    assert np.isfinite(ref[:, 3]).mean() > 0.9 and (ref[items[:, 11] == 2, 3] != 0).mean() > 0.3


def test_render_cli_writes_an_image(tmp_path):
    """The reference's command-line surface (render.py:13-19) on top of the HIP path."""
    from clive2_amd import render
    out = tmp_path / "cornell.png"
    assert render.main(["--scene", "empty", "--width", "64", "--height", "36", "--samples", "3", "--out", str(out)]) == 0
    from PIL import Image
    img = np.asarray(Image.open(out))
    assert img.shape == (36, 64, 3) and img.std() > 1          # a picture, not a constant
    # --reproducible: two runs write the same file, byte for byte
    a, b = tmp_path / "a.png", tmp_path / "b.png"
    for f in (a, b):
        assert render.main(["--scene", "empty", "--width", "64", "--height", "36", "--samples", "3", "--reproducible", "--out", str(f)]) == 0
    assert a.read_bytes() == b.read_bytes()
    with pytest.raises(ValueError):
        render.main(["--scene", "no-such-preset", "--width", "8", "--height", "8", "--samples", "1"])


def test_rccl_reduce_path_on_one_rank(tmp_path):
    """The N>1 reduction as bench.py runs it -- cl2_comm_init_rank, the in-place RCCL all-reduce of the
    accumulators inside the library, cl2_comm_allreduce_f64, teardown -- on a one-rank communicator: the
    sum over one rank must return the same bytes, the renderer must keep working, and the process must
    EXIT.  Runs in a child process (tests/rccl_one_rank_child.py) under a time limit; a hang anywhere --
    creation, collective or teardown -- is a FAILURE, and the child's STEP trace says where."""
    import os, subprocess, sys
    child = os.path.join(os.path.dirname(__file__), "rccl_one_rank_child.py")
    env = dict(os.environ, CLIVE2_RENDEZVOUS_FILE=str(tmp_path / "rccl_id"), HSA_ENABLE_IPC_MODE_LEGACY="0")
    try:
        p = subprocess.run([sys.executable, child], capture_output=True, text=True, timeout=300, env=env)
        out, err, rc = p.stdout, p.stderr, p.returncode
    except subprocess.TimeoutExpired as e:
        dec = lambda b: b.decode(errors="replace") if isinstance(b, bytes) else (b or "")
        out, err, rc = dec(e.stdout), dec(e.stderr), None
    steps = [l.split()[1] for l in out.splitlines() if l.startswith("STEP ")]
    trace = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(trace):                                   # kept with the run's artefacts (profiles/ gets a copy)
        with open(os.path.join(trace, "rccl_one_rank_trace.log"), "w") as f:
            f.write(f"rc={rc}\n--- stdout ---\n{out}\n--- stderr (tail) ---\n{err[-6000:]}\n")
    assert rc is not None, f"the RCCL child hung after steps {steps}"
    assert rc == 0, (rc, steps, out[-2000:], err[-3000:])
    assert steps == ["rendered", "comm-up", "handover-ok", "host-allreduce-ok", "render-after-ok", "second-reduce-ok",
                     "comm-down", "closed"], steps
    assert "torch" not in out.split("MODULES")[-1]             # the product path never imported torch


def test_bench_line_proves_what_the_communicator_saw(tmp_path):
    """the job line must show that RCCL saw N ranks on N distinct GPUs -- from the communicator
    (ncclCommCount / ncclCommUserRank via cl2_comm_info, PCI addresses gathered through the library's all-reduce), not from
    the launcher's environment -- and what the all-reduce cost.  The N > 1 code path of bench.py on a one-rank communicator
    (CLIVE2_BENCH_FORCE_COMM=1; RCCL refuses two ranks on one device): `comm` = {nranks 1, one PCI address, allreduce_ms}."""
    import json, os, re, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, CLIVE2_BENCH_FORCE_COMM="1", CLIVE2_RENDEZVOUS_FILE=str(tmp_path / "id"), HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--width", "320", "--height", "180", "--steps", "6", "--warmup", "1",
                        "--no-mesh", "--no-cpu-baseline"], capture_output=True, text=True, timeout=300, env=env)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1                                     # stdout carries exactly the one JSON line
    out = json.loads(lines[0])
    comm = out["comm"]
    assert comm["nranks"] == 1 and comm["distinct_devices"] == 1 and comm["launcher_world_size"] == 1
    assert len(comm["devices"]) == 1 and re.fullmatch(r"[0-9a-f]{4}:[0-9a-f]{2}:[0-9a-f]{2}\.[0-7]", comm["devices"][0])
    assert comm["devices"][0] == comm["rank0_device"].lower()
    assert comm["allreduce_ms"] >= 0 and comm["allreduce_bytes"] == 8 * 320 * 180 * 4
    assert out["config"]["sample_streams"] == 1 and "sample-split x1" in out["config"]["parallelism"]


def test_written_png_is_upright(tmp_path):
    """The film sits behind the pinhole: row 0 of Renderer.image looks UP at the ceiling light, and the
    reference hands that array to cv2.imwrite unflipped (render.py:47-50).  The PNG written by the CLI
    must therefore have the emitter's bright band at the TOP (it used to be flipped)."""
    from clive2_amd import render, movie
    from clive2_amd.renderer import Renderer
    import clive2_amd as c2
    from PIL import Image
    out = tmp_path / "c.png"
    assert render.main(["--scene", "empty", "--width", "64", "--height", "48", "--samples", "24", "--out", str(out)]) == 0
    img = np.asarray(Image.open(out)).astype(np.float64)
    lum = img.mean(axis=2).mean(axis=1)                        # per-row luminance
    # ceiling + emitter: the first rows are the brightest of the picture (the floor at the bottom comes second)
    assert np.argmax(lum) < 3 and lum[:2].min() > lum[3:].max(), lum
    centre = img[:2, 24:40].mean()                               # the emitter itself: brighter than the ceiling beside it
    assert centre > img[:2, :8].mean() and centre > img[-2:, 24:40].mean()
    # the PNG is exactly the renderer's picture with the channels swapped (BGR -> RGB), nothing else
    r = Renderer(c2.create_scene_from_preset("empty", 64, 48))
    r.run_samples(24)
    assert np.array_equal(np.asarray(Image.open(out)), r.image[:, :, ::-1])
    movie.save_frame(str(tmp_path / "f.png"), r.image)
    assert np.array_equal(np.asarray(Image.open(tmp_path / "f.png")), r.image[:, :, ::-1])
    # channel order: the side walls are BGR (.541,.807,0) and (.8,.3,.3) (load.py:192-198): in an RGB file both
    # have more blue than red; saved with the channels unswapped they would come out red-heavy
    rgb = np.asarray(Image.open(out)).astype(np.float64)
    left, right = rgb[20:40, 2:10].mean(axis=(0, 1)), rgb[20:40, -10:-2].mean(axis=(0, 1))
    assert left[2] > left[0] and right[2] > right[0], (left, right)


def test_process_images_and_image_properties_match_reference_fixture():
    """tests/golden/renderer_glue.npz was produced by the REFERENCE's own numpy code (renderer.py:253-316 run
    through a fake metalcompute, tests/golden/make_fixtures.py).  Here its per-sample inputs go into the device
    buffers, `process_images` runs on the GPU (k_accumulate), and accumulators + tone-mapped images must equal
    the reference's outputs byte for byte."""
    import os
    import clive2_amd as c2
    from clive2_amd.renderer import Renderer
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "renderer_glue.npz"))
    W, H = int(g["width"]), int(g["height"])
    r = Renderer(c2.create_scene_from_preset("empty", W, H))
    for k in range(len(g["in_finalized"])):
        light = g["in_light"][k].copy()
        # the reference reads only the rgb of out_light_image (renderer.py:258-260) and K8 adds the t=1 weight
        # mass straight into sample_weights (trace.metal:963); here that mass travels in the 4th channel
        light[:, 3] = 0.0
        assert (g["in_counts"][k] == 1).all()
        r.import_sample_images(finalized=g["in_finalized"][k], light=light, sample_weights=g["in_weights"][k],
                               unidirectional=g["in_unidirectional"][k])
        r.process_images()
    img, wts, cnt, uni = r.read_accumulators()
    assert img.tobytes() == g["summed_image"].tobytes()
    assert wts.tobytes() == g["summed_sample_weights"].tobytes()
    assert cnt.tobytes() == g["summed_sample_counts"].tobytes()
    assert uni.tobytes() == g["unidirectional_image_buffer"].tobytes()
    with np.errstate(all="ignore"):
        assert np.array_equal(r.image, g["image"])
        assert np.array_equal(r.unweighted_image, g["unweighted_image"])
        assert np.array_equal(r.unidirectional_image, g["unidirectional_image"])
    assert r.samples == 0 and r.counters()["samples"] == 3


def test_device_tone_map_matches_reference_fixture_and_host_path(cornell_small):
    """cl2_tone_log_sum + cl2_tone_map (csrc/tonemap.hpp) against (a) the three pictures the REFERENCE's own
    `tone_map` made of the fixture's accumulators (tests/golden/renderer_glue.npz: NaN / +-inf / zero-weight entries
    included) and (b) the host path `Renderer.image` on a real render.  Same arithmetic and numpy dtypes; the float64
    log-luminance sum is added in another order than numpy's pairwise sum, so Lw may differ in its last bits and a byte
    may move where 255*x/(x+w) lies within ~1e-13 of an integer: at most one count, in at most 2 bytes per picture."""
    import os
    import clive2_amd as c2
    from clive2_amd.renderer import Renderer, make_seeds

    def close_enough(got, want):
        assert got.shape == want.shape and got.dtype == want.dtype == np.uint8
        d = np.abs(got.astype(np.int16) - want.astype(np.int16))
        assert d.max() <= 1 and int((d > 0).sum()) <= 2, (int(d.max()), int((d > 0).sum()))

    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "renderer_glue.npz"))
    W, H = int(g["width"]), int(g["height"])
    r = Renderer(c2.create_scene_from_preset("empty", W, H))
    for k in range(len(g["in_finalized"])):
        light = g["in_light"][k].copy()
        light[:, 3] = 0.0
        r.import_sample_images(finalized=g["in_finalized"][k], light=light, sample_weights=g["in_weights"][k],
                               unidirectional=g["in_unidirectional"][k])
        r.process_images()
    for which in ("image", "unweighted_image", "unidirectional_image"):
        close_enough(r.tone_mapped(which), g[which])
    # other exposure / white point than the properties use: against the host function on the same accumulators
    from clive2_amd.camera import tone_map
    close_enough(r.tone_mapped("image", exposure=2.0, white_point=1.5), tone_map(r.radiance, exposure=2.0, white_point=1.5))
    r.close()

    r = Renderer(cornell_small, seeds=make_seeds(cornell_small.pixel_width * cornell_small.pixel_height))
    r.run_samples(6)
    for which in ("image", "unweighted_image", "unidirectional_image"):
        got = r.tone_mapped(which)
        close_enough(got, getattr(r, which))
        assert got.std() > 5                              # a picture, not a constant
    assert r.tone_mapped("image").tobytes() == r.tone_mapped("image").tobytes()        # deterministic
    r.close()


def test_movie_cli_writes_turntable_frames(tmp_path):
    """The reference's turntable loop (movie.py:29-55): one scene + renderer per frame, one PNG each."""
    from clive2_amd import movie
    from PIL import Image
    rc = movie.main(["--scene", "empty", "--width", "48", "--height", "32", "--samples", "2", "--movie-frames", "3",
                     "--movie-name", "tt", "--out-root", str(tmp_path)])
    assert rc == 0
    frames = [np.asarray(Image.open(tmp_path / "tt" / f"frame_{f:04d}.png")) for f in range(3)]
    assert all(fr.shape == (32, 48, 3) for fr in frames)
    assert not np.array_equal(frames[0], frames[1])        # the camera moved


@pytest.mark.parametrize("mode", [1, 2, 3, 4, 5])
def test_sample_pipeline_is_a_pure_performance_knob(mode, glass_scene, oracle_mod):
    """run_samples as a pipeline over samples (later stages of sample i on their own streams beside
    the subpath stage of the next samples; 2 and 3 stages) == serial order == oracle: seeds, last
    subpaths, accumulators."""
    n = 5
    r1, o = _pair(glass_scene, oracle_mod)
    r2, _ = _pair(glass_scene, oracle_mod)
    r3, _ = _pair(glass_scene, oracle_mod)
    for r, stages in ((r1, 1), (r2, 0), (r3, 2)):
        r.set_traversal_mode(mode)
        r.set_pipelining(stages)
        r.run_samples(n)
    for _ in range(n):
        o.run_sample()
    for r in (r1, r2, r3):
        assert np.array_equal(r.get_random_buffer(), o.rand_buffer)
        for which, ref in ((LIGHT, o.out_light_paths), (CAMERA, o.out_camera_paths)):
            assert r.export_paths(which).tobytes() == ref.tobytes()       # the last sample's subpaths
        img, wts, cnt, uni = r.read_accumulators()
        assert (cnt == n).all()
        np.testing.assert_allclose(uni, o.unidirectional_image_buffer, rtol=1e-6, atol=0)
        np.testing.assert_allclose(img, o.summed_image, rtol=5e-5, atol=1e-8)
        np.testing.assert_allclose(wts, o.summed_sample_weights, rtol=5e-5, atol=1e-8)
        assert r.counters()["rays"] == o.rays_traced
    assert r1.read_accumulators()[3].tobytes() == r2.read_accumulators()[3].tobytes() == r3.read_accumulators()[3].tobytes()
    with pytest.raises(Exception):
        r1.set_pipelining(3)
    r1.set_pipelining(-1)                      # by frame size (the default)


@pytest.mark.parametrize("mode,levels,stages", [(1, 6, 0), (1, 2, 1), (2, 1, 2), (1, 0, 2), (3, 3, 1)])
def test_open_scene_paths_of_every_length(mode, levels, stages, oracle_mod):
    """An OPEN scene (floor, back wall, emitter and a glass ball; no other walls): most subpaths leave
    the scene after one to three bounces, so every subpath length occurs, queues shrink from level to
    level (compaction between launches, persistent launches that run dry early) and most (t,s) pairs
    do not exist.  Subpaths, seeds, aggregators and the accumulated image against the oracle."""
    import clive2_amd as c2
    from clive2_amd.load import triangles_for_box
    from clive2_amd.meshes import icosphere
    box = triangles_for_box()
    keep = [t for t in box if t.emitter or t.n[1] > 0.5 or t.n[2] > 0.5]          # emitter, floor, back wall
    assert len(keep) == 6
    scene = c2.create_scene(72, 40, np.array([0, 1.5, 6.0]), np.array([0, 0, -1.0]), room=keep,
                            file_specs=[dict(mesh=icosphere(2, radius=1.5), material=5, offset=np.array([0.5, 0.0, -1.0]))])
    r, o = _pair(scene, oracle_mod)
    r.set_traversal_mode(mode); r.set_levels_per_launch(levels); r.set_pipelining(stages)
    _run_to_paths(r, o)
    for which, ref in ((LIGHT, o.out_light_paths), (CAMERA, o.out_camera_paths)):
        assert r.export_paths(which).tobytes() == ref.tobytes()
    lens = np.bincount(o.out_camera_paths["length"], minlength=7)
    assert (lens[1:] > 0).all(), lens                         # every camera subpath length 1..6 occurs
    assert np.bincount(o.out_light_paths["length"], minlength=7)[:4].sum() > 0.3 * len(o.out_light_paths)
    _run_rest(r, o)
    agg = r.export_aggregators()
    assert agg["total_contribution"].tobytes() == o.weight_aggregators["total_contribution"].tobytes()
    assert agg["weights"].tobytes() == o.weight_aggregators["weights"].tobytes()
    r.run_samples(3)
    for _ in range(3):
        o.run_sample()
    assert np.array_equal(r.get_random_buffer(), o.rand_buffer)
    img, wts, cnt, uni = r.read_accumulators()
    np.testing.assert_allclose(img, o.summed_image, rtol=5e-5, atol=1e-8)
    np.testing.assert_allclose(wts, o.summed_sample_weights, rtol=5e-5, atol=1e-8)
    assert r.counters()["rays"] == o.rays_traced


@pytest.mark.parametrize("mode", [0, 1, 2])
@pytest.mark.parametrize("scene_name", ["cornell_small", "glass_scene"])
def test_pruned_table_is_equivalent(scene_name, mode, request, oracle_mod):
    """LDS-resident trees are walked through a table without the inner boxes whose test is expected to cost more
    than it saves (cl2_upload_scene: exact for rays with finite 1/d, the others walk the full table).  The Cornell
    box keeps its 3 leaves of 5 boxes.  Same subpaths, seeds and image as the full table (debug bit 7) and the oracle."""
    scene = request.getfixturevalue(scene_name)
    r, o = _pair(scene, oracle_mod)
    full, _ = _pair(scene, oracle_mod)
    full.set_debug_flags(1 << 7)
    org = r.organisation()
    assert org["tree_in_lds"] == 1 and 0 < org["pruned_records"] < org["n_records"], org
    if scene_name == "cornell_small":
        assert (org["n_records"], org["pruned_records"]) == (5, 3)
    assert full.organisation()["pruned_records"] == 0
    r.set_traversal_mode(mode); full.set_traversal_mode(mode)
    _run_to_paths(r, o)
    full.make_light_rays(); full.make_camera_rays(); full.trace_light_rays(); full.trace_camera_rays()
    for which, ref in ((LIGHT, o.out_light_paths), (CAMERA, o.out_camera_paths)):
        assert r.export_paths(which).tobytes() == ref.tobytes() == full.export_paths(which).tobytes()
    _run_rest(r, o)
    full.join_paths(); full.finalize_samples(); full.gather_light_image(); full.process_images()
    r.run_samples(3); full.run_samples(3)
    assert np.array_equal(r.get_random_buffer(), full.get_random_buffer())
    assert r.read_accumulators()[3].tobytes() == full.read_accumulators()[3].tobytes()       # unidirectional image: no atomics
    # the counting mode walks the full table: its tallies are those of the reference's walk
    r.reset_counters(); r.set_counting(True); full.reset_counters(); full.set_counting(True)
    r.run_samples(1); full.run_samples(1)
    assert r.counters() == full.counters()
    r.close(); full.close()


@pytest.mark.parametrize("mode", [0, 1])
def test_flat_pruned_table_walk_is_equivalent(mode, cornell_small, oracle_mod):
    """The Cornell box's pruned table is a plain list of its 3 leaves, so every ray visits the same records in the same
    order and the walk runs with wave-uniform control flow (scalar loops, a leaf's misses masked, the next triangle's
    record fetched while the current one is tested: closest_hit_flat).  Same subpaths, seeds, aggregators and image as the
    per-lane walk of the same table (debug bit 11) and as the oracle."""
    r, o = _pair(cornell_small, oracle_mod)
    lane, _ = _pair(cornell_small, oracle_mod)
    lane.set_debug_flags(1 << 11)
    assert r.organisation()["pruned_records"] == lane.organisation()["pruned_records"] == 3
    r.set_traversal_mode(mode); lane.set_traversal_mode(mode)
    _run_to_paths(r, o)
    lane.make_light_rays(); lane.make_camera_rays(); lane.trace_light_rays(); lane.trace_camera_rays()
    for which, ref in ((LIGHT, o.out_light_paths), (CAMERA, o.out_camera_paths)):
        assert r.export_paths(which).tobytes() == ref.tobytes() == lane.export_paths(which).tobytes()
    _run_rest(r, o)
    lane.join_paths(); lane.finalize_samples(); lane.gather_light_image(); lane.process_images()
    assert r.export_aggregators().tobytes() == lane.export_aggregators().tobytes()
    r.run_samples(3); lane.run_samples(3)
    assert np.array_equal(r.get_random_buffer(), lane.get_random_buffer())
    assert r.read_accumulators()[3].tobytes() == lane.read_accumulators()[3].tobytes()       # unidirectional image: no atomics
    assert r.counters()["rays"] == lane.counters()["rays"]
    r.close(); lane.close()


def test_scene_near_the_lds_caps_keeps_its_residency(oracle_mod):
    """the pruned table shares the workgroup's LDS with the full record table, the staged triangles and the
    subpath kernel's 9.7 KB of shading tables.  Near the 512-record / 512-triangle caps it is left out when it would
    take the kernels below three workgroups per CU (160 KB / 3); the launch succeeds either way and stays bit-exact."""
    import clive2_amd as c2
    from clive2_amd.load import get_materials
    from clive2_amd.meshes import icosphere
    mats = get_materials()
    mats["alpha"][5] = 0.1
    specs = [dict(mesh=icosphere(2, radius=1.6, center=(0.0, 0.5, 0.0)), material=5),
             dict(mesh=icosphere(1, radius=1.0, center=(-4.0, 0.0, -3.0)), material=3),
             dict(mesh=icosphere(1, radius=1.0, center=(4.0, 0.0, -3.0)), material=1)]
    scene = c2.create_scene(64, 48, np.array([0, 1.5, 6]), np.array([0, 0, -1]), file_specs=specs, materials=mats)
    assert 480 <= len(scene.triangles) <= 512
    r, o = _pair(scene, oracle_mod)
    org = r.organisation()
    assert org["tree_in_lds"] == 1
    lds = (2 * org["n_records"] + 3 * len(scene.triangles) + 2 * org["pruned_records"]) * 16 + 9728
    assert lds <= 160 * 1024 // 3 or org["pruned_records"] == 0, (org, lds)
    _run_to_paths(r, o)
    for which, ref in ((LIGHT, o.out_light_paths), (CAMERA, o.out_camera_paths)):
        assert r.export_paths(which).tobytes() == ref.tobytes()
    _run_rest(r, o)
    assert r.export_aggregators()["total_contribution"].tobytes() == o.weight_aggregators["total_contribution"].tobytes()
    r.run_samples(4)
    assert np.isfinite(r.read_accumulators()[0]).all()
    r.close()


@pytest.mark.parametrize("scene_name", ["cornell_small", "glass_scene"])
def test_wide_resolve_kernel_agrees(scene_name, request, oracle_mod):
    """The second implementation of the resolve stage (one wave per camera vertex, running total relayed
    between the waves; tests/connect_resolve_wide.hpp) reproduces the oracle's aggregators, unidirectional
    estimate and image exactly like the default one-thread-per-pixel kernel."""
    from clive2_amd.renderer import Renderer, RendererError, make_seeds
    scene = request.getfixturevalue(scene_name)
    seeds = make_seeds(scene.pixel_width * scene.pixel_height)
    # the cross-check kernel is built into the TEST variant of the library only; the product library refuses it
    plain = Renderer(scene, seeds=seeds)
    plain.set_debug_flags(7 << 4)
    plain.make_light_rays(); plain.make_camera_rays(); plain.trace_light_rays(); plain.trace_camera_rays()
    with pytest.raises(RendererError):
        plain.join_paths()
    plain.close()
    r, o = Renderer(scene, seeds=seeds, variant="test"), oracle_mod.OracleRenderer(scene, seeds=seeds)
    r.set_debug_flags(7 << 4)
    _run_to_paths(r, o)
    _run_rest(r, o)
    agg = r.export_aggregators()
    assert agg["total_contribution"].tobytes() == o.weight_aggregators["total_contribution"].tobytes()
    assert agg["weights"].tobytes() == o.weight_aggregators["weights"].tobytes()
    assert agg["contrib_weight_sum"].tobytes() == o.weight_aggregators["contrib_weight_sum"].tobytes()
    r.run_samples(3)
    for _ in range(3):
        o.run_sample()
    img, wts, cnt, uni = r.read_accumulators()
    np.testing.assert_allclose(uni, o.unidirectional_image_buffer, rtol=1e-6, atol=0)
    np.testing.assert_allclose(img, o.summed_image, rtol=5e-5, atol=1e-8)
    np.testing.assert_allclose(wts, o.summed_sample_weights, rtol=5e-5, atol=1e-8)
    # the cross-check kernel splats with atomics and writes no records, so with the reproducible switch on the sort +
    # gather would read buffers nobody filled (never allocated, on first use).  The combination is refused, the handle stays
    # usable, and with the bits cleared the reproducible render goes through.
    r.set_reproducible(True)
    r.make_light_rays(); r.make_camera_rays(); r.trace_light_rays(); r.trace_camera_rays()
    with pytest.raises(RendererError, match="reproducible"):
        r.join_paths()
    r.set_debug_flags(0)
    r.join_paths()
    r.finalize_samples(); r.gather_light_image(); r.process_images()
    assert np.isfinite(r.read_accumulators()[0]).all()
    r.close()


def test_c_abi_error_behaviour(cornell_small):
    """The boundary's error contract (include/clive2_amd.h): every entry point returns a negative code,
    leaves a message for cl2_last_error, never crashes, and the handle stays usable afterwards."""
    import ctypes as C
    from clive2_amd import _native
    from clive2_amd.renderer import Renderer, make_seeds
    L = _native.lib()
    B = cornell_small.pixel_width * cornell_small.pixel_height
    # creation: bad frame sizes, bad device
    h = C.c_void_p()
    assert L.cl2_create(0, 0, 16, C.byref(h)) < 0 and not h.value
    assert L.cl2_create(0, 1 << 14, 1 << 13, C.byref(h)) < 0 and not h.value           # >= 2^26 pixels
    assert L.cl2_create(9999, 16, 16, C.byref(h)) < 0 and not h.value
    assert b"device" in L.cl2_last_error(None) or b"range" in L.cl2_last_error(None)
    # a fresh handle without a scene refuses to render
    assert L.cl2_create(0, 16, 16, C.byref(h)) == 0 and h.value
    assert L.cl2_run_samples(h, 1) < 0 and b"scene" in L.cl2_last_error(h)
    assert L.cl2_make_light_rays(h) < 0
    L.cl2_destroy(h)
    # NULL handles
    assert L.cl2_run_samples(None, 1) < 0 and L.cl2_set_pipelining(None, 1) < 0
    L.cl2_destroy(None)                                                                # no-op

    r = Renderer(cornell_small, seeds=make_seeds(B))
    hh = r._h
    # wrong sizes / ranges
    buf = np.zeros(8 * B + 1, dtype=np.float32)
    assert L.cl2_read_accumulators_packed(hh, buf.ctypes.data_as(C.c_void_p), C.c_size_t(8 * B + 1)) < 0
    assert L.cl2_set_seeds(hh, buf.ctypes.data_as(C.c_void_p), C.c_size_t(3)) < 0
    assert L.cl2_run_samples(hh, -1) < 0
    assert L.cl2_set_traversal_mode(hh, 9) < 0 and L.cl2_set_pipelining(hh, 5) < 0 and L.cl2_set_levels_per_launch(hh, -2) < 0
    assert L.cl2_export_paths(hh, 0, buf.ctypes.data_as(C.c_void_p), C.c_size_t(B + 1)) < 0
    # a scene that fails validation is rejected and the old one stays in place
    boxes = np.array(cornell_small.boxes, copy=True)
    boxes["left"][0] = 9999                                                            # child index out of range
    import copy
    bad = copy.copy(cornell_small)
    bad.boxes = boxes
    bad.validate = lambda: None                # skip the host-side check: the library must catch it
    with pytest.raises(_native.RendererError):
        r.upload_scene(bad)
    # ... and the renderer still works
    r.upload_scene(cornell_small)
    r.run_samples(2)
    assert np.isfinite(r.packed_accumulators()).all() and r.counters()["rays"] > 0


def test_hip_path_equals_the_python_restatement_directly(oracle_mod):
    """The HIP kernels against oracle/py_kernels.py -- the second restatement of generate_paths / connect_paths, written from the
    Metal text separately from the C oracle -- with no C oracle in between except for the first vertices (K1 / K2, which
    oracle/np_kernels.py restates): both Path[] buffers, the seeds and the filter aggregators of a 16x16 rough-glass frame."""
    import clive2_amd as c2
    from clive2_amd.load import get_materials
    from clive2_amd.meshes import icosphere
    from clive2_amd.renderer import Renderer, make_seeds
    from oracle import py_kernels as pk
    mats = get_materials()
    mats["alpha"][5] = 0.1
    scene = c2.create_scene(16, 16, np.array([0, 1.5, 6]), np.array([0, 0, -1]), materials=mats,
                            file_specs=[dict(mesh=icosphere(1, radius=2.0, center=(0.0, 1.0, 0.0)), material=5)])
    seeds = make_seeds(256, seed=99)
    r = Renderer(scene, seeds=seeds)
    r.make_light_rays(); r.make_camera_rays()
    first_l, first_c = r.export_rays(LIGHT), r.export_rays(CAMERA)          # the generators' output in the reference's Ray layout
    sd = r.get_random_buffer().copy()
    r.trace_light_rays(); r.trace_camera_rays(); r.join_paths()
    o = oracle_mod                                                         # dtypes only
    tri, mat, box = (np.ascontiguousarray(a) for a in (scene.triangles, scene.materials, scene.boxes))
    _, paths_l = pk.generate_paths(first_l.view(o.Ray), box, tri, mat, sd, o.Ray, o.Path)
    _, paths_c = pk.generate_paths(first_c.view(o.Ray), box, tri, mat, sd, o.Ray, o.Path)
    assert r.export_paths(LIGHT).tobytes() == paths_l.tobytes()
    assert r.export_paths(CAMERA).tobytes() == paths_c.tobytes()
    assert np.array_equal(r.get_random_buffer(), sd)
    res = pk.connect_paths(paths_c, paths_l, tri, mat, box, np.ascontiguousarray(scene.camera), o.Ray, o.WeightAggregator, 4096)
    agg = r.export_aggregators()
    for f in ("weights", "total_contribution", "contrib_weight_sum"):
        assert agg[f].tobytes() == res["aggregators"][f].tobytes(), f
    assert (paths_c["length"] >= 4).sum() > 20 and (res["light_pixel_indices"] >= 0).sum() > 30
    r.close()
