"""Pins for the C oracle (oracle/bdpt_oracle.c).  The reference ships no tests or vectors
(SURVEY.md F8), so the oracle is pinned by: RNG known answers computed from trace.metal:87-93,
an independent numpy restatement (oracle/np_kernels.py), analytic cases, and the reference's own
built-in cross-estimator (BDPT vs unidirectional)."""
import ctypes as C
import numpy as np
import pytest

import clive2_amd as c2
from clive2_amd import struct_types as st


def test_xorshift_known_answers(oracle_mod):
    # SURVEY.md §8a row a4
    got = oracle_mod.xorshift_floats(1, 4)
    assert [s for s, _ in got] == [270369, 67634689, 2647435461, 307599695]
    np.testing.assert_allclose([f for _, f in got], [6.2950188e-05, 0.015747428, 0.61640412, 0.071618631], rtol=1e-7)
    got = oracle_mod.xorshift_floats(0xDEADBEEF, 2)
    assert [s for s, _ in got] == [0x477D20B7, 0x8E1D9142]
    np.testing.assert_allclose([f for _, f in got], [0.27925304, 0.55513865], rtol=1e-7)
    # 1.0 is reachable: (float)0xFFFFFFFF == 2^32
    from oracle import np_kernels as npk
    s, f = npk.xorshift(np.array([1, 0xDEADBEEF], np.uint32))
    assert list(s) == [270369, 0x477D20B7]
    assert np.float32(0xFFFFFFFF) / np.float32(4294967296.0) == 1.0


def _max_ulp(got, ref64):
    ref32 = ref64.astype(np.float32)
    ulp = np.spacing(np.maximum(np.abs(ref32), np.float32(1e-30))).astype(np.float64)
    return float(np.max(np.abs(got.astype(np.float64) - ref64) / ulp))


def test_detmath_accuracy_and_special_values(oracle_mod):
    x = np.linspace(0, 2 * np.pi, 400001).astype(np.float32)
    assert np.max(np.abs(oracle_mod.det_math("sin", x) - np.sin(x.astype(np.float64)))) < 1.5e-7
    assert np.max(np.abs(oracle_mod.det_math("cos", x) - np.cos(x.astype(np.float64)))) < 1.5e-7
    y = np.linspace(0, 1, 200001).astype(np.float32)
    assert _max_ulp(oracle_mod.det_math("acos", y)[:-1], np.arccos(y.astype(np.float64))[:-1]) <= 3.0
    z = np.concatenate([np.linspace(0, 4, 200001), np.logspace(0.6, 6, 2001)]).astype(np.float32)
    assert _max_ulp(oracle_mod.det_math("atan", z)[1:], np.arctan(z.astype(np.float64))[1:]) <= 3.0
    e = np.linspace(-80, 0, 200001).astype(np.float32)
    assert _max_ulp(oracle_mod.det_math("exp", e), np.exp(e.astype(np.float64))) <= 2.0
    one = lambda name, v: float(oracle_mod.det_math(name, np.array([v], np.float32))[0])
    assert one("sin", 0.0) == 0.0 and one("cos", 0.0) == 1.0 and one("atan", 0.0) == 0.0
    assert one("acos", 1.0) == 0.0 and one("exp", 0.0) == 1.0 and one("exp", -100.0) == 0.0
    assert one("atan", np.inf) == np.float32(np.pi / 2) and np.isnan(one("atan", np.nan))
    # libm build agrees to a few ulp (the statistical cross-check variant)
    s_libm = oracle_mod.det_math("sin", x, libm=True)
    assert np.max(np.abs(s_libm - oracle_mod.det_math("sin", x))) < 3e-7


def test_camera_rays_match_numpy_restatement(oracle_mod):
    from oracle import np_kernels as npk
    scene = c2.create_scene_from_preset("empty", 96, 64)
    o = oracle_mod.OracleRenderer(scene)
    seeds0 = o.rand_buffer.copy()
    o.make_camera_rays()
    org, d, c_imp, seeds1 = npk.generate_camera_rays(scene.camera, seeds0)
    assert o.camera_ray_buffer["origin"][:, :3].tobytes() == org.tobytes()
    assert o.camera_ray_buffer["direction"][:, :3].tobytes() == d.tobytes()
    assert (o.camera_ray_buffer["c_importance"] == c_imp).all()
    assert np.array_equal(o.rand_buffer, seeds1)
    assert (o.camera_ray_buffer["pixel_idx"] == np.arange(96 * 64)).all()
    # every camera ray passes through the focal point
    cam = scene.camera.reshape(-1)[0]
    to_f = cam["focal_point"][:3] - org
    assert np.allclose(np.cross(to_f, d), 0, atol=1e-5)


def test_traverse_matches_numpy_restatement(oracle_mod, glass_scene):
    from oracle import np_kernels as npk
    for scene in (c2.create_scene_from_preset("empty", 48, 32), glass_scene):
        o = oracle_mod.OracleRenderer(scene)
        o.make_light_rays(); o.make_camera_rays()
        rays = np.concatenate([o.camera_ray_buffer, o.light_ray_buffer])
        bi, bt, u, v, cnt = oracle_mod.traverse(rays, scene.boxes, scene.triangles)
        nbi, nbt, nu, nv = npk.traverse(rays["origin"][:, :3], rays["direction"][:, :3], scene.boxes, scene.triangles)
        assert np.array_equal(bi, nbi)
        hit = bi >= 0
        assert hit.mean() > 0.95
        assert bt[hit].tobytes() == nbt[hit].tobytes()
        assert u[hit].tobytes() == nu[hit].tobytes() and v[hit].tobytes() == nv[hit].tobytes()
        assert cnt["rays"] == len(rays)


def _ray(origin, direction):
    r = np.zeros(1, st.Ray)
    r["origin"][0, :3] = origin
    d = np.asarray(direction, np.float32)
    r["direction"][0, :3] = d
    with np.errstate(divide="ignore"):
        r["inv_direction"][0, :3] = np.float32(1.0) / d
    return r


def test_intersection_analytic_cases(oracle_mod):
    scene = c2.create_scene_from_preset("empty", 16, 16)
    T = lambda o, d: oracle_mod.traverse(_ray(o, d), scene.boxes, scene.triangles)
    # straight up from the box centre: hits the light quad (y = 9.5) at t = 8.5, tri 12 or 13
    bi, bt, *_ = T([0.3, 1.0, 0.1], [0, 1, 0])
    assert bi[0] in (12, 13) and abs(bt[0] - 8.5) < 1e-5
    # axis-parallel rays have inv_direction = +-inf in two components
    bi, bt, *_ = T([0, 1, 0], [1, 0, 0])
    assert scene.triangles["material"][bi[0]] == 2 and abs(bt[0] - 10.0) < 1e-5        # right wall
    bi, bt, *_ = T([0, 1, 0], [-1, 0, 0])
    assert scene.triangles["material"][bi[0]] == 1 and abs(bt[0] - 10.0) < 1e-5        # left wall
    bi, bt, *_ = T([0, 1, 0], [0, -1, 0])
    assert abs(bt[0] - 3.0) < 1e-5                                                     # floor y = -2
    # hits closer than DELTA = 1e-4 are rejected (trace.metal:136): start 5e-5 below the ceiling light
    bi, bt, *_ = T([0, 9.5 - 5e-5, 0], [0, 1, 0])
    assert bi[0] in (14, 15) and abs(bt[0] - 0.5) < 1e-4                                # passes through to the ceiling
    bi, bt, *_ = T([0, 9.5 - 2e-4, 0], [0, 1, 0])
    assert bi[0] in (12, 13)
    # a ray leaving through the camera quad's plane from in front of it hits an is_camera triangle
    bi, bt, *_ = T([0, 1.5, 0], [0, 0, 1])
    assert scene.triangles["is_camera"][bi[0]] == 1 and abs(bt[0] - 6.0) < 1e-5
    # shared diagonal of a wall quad: inclusive u,v bounds -> no crack
    bi, bt, *_ = T([0, 4, 0], [0, 0, -1])
    assert bi[0] >= 0 and abs(bt[0] - 10.0) < 1e-5


def test_fresnel_and_ggx_terms(oracle_mod):
    L = oracle_mod.lib()
    f3 = lambda *a: (C.c_float * 3)(*a)
    n = f3(0, 0, 1)
    # normal incidence on ior 1.5: ((1.5-1)/(1.5+1))^2 = 0.04
    assert abs(L.orc_fresnel(f3(0, 0, 1), n, C.c_float(1.0), C.c_float(1.5)) - 0.04) < 1e-7
    # total internal reflection beyond asin(1/1.5) = 41.81 deg when leaving the glass
    for deg, tir in ((41.0, False), (42.5, True), (80.0, True)):
        a = np.deg2rad(deg)
        F = L.orc_fresnel(f3(np.sin(a), 0, np.cos(a)), n, C.c_float(1.5), C.c_float(1.0))
        assert (F == 1.0) == tir
    # GGX D: alpha == 0 -> 1 (trace.metal:280); alpha > 0 -> integral D(m)(m.n) dw = 1
    assert L.orc_ggx_d(f3(0, 0, 1), n, C.c_float(0.0)) == 1.0
    theta = (np.arange(20000) + 0.5) / 20000 * (np.pi / 2)
    for alpha in (0.1, 0.3, 0.7):
        D = np.array([L.orc_ggx_d(f3(np.sin(t), 0, np.cos(t)), n, C.c_float(alpha)) for t in theta[::20]])
        tt = theta[::20]
        integral = np.sum(D * np.cos(tt) * np.sin(tt)) * 2 * np.pi * (tt[1] - tt[0])
        assert abs(integral - 1.0) < 2e-3, (alpha, integral)


def test_ggx_sample_follows_its_pdf(oracle_mod):
    """GGX_sample (trace.metal:226-233): cos(theta_m) must have CDF 1 - ... of D(m)cos; check the
    closed form tan^2(theta) = alpha^2 r/(1-r) and that samples stay in the upper hemisphere."""
    L = oracle_mod.lib()
    out = (C.c_float * 3)()
    n = (C.c_float * 3)(0, 1, 0)
    rng = np.random.RandomState(3)
    alpha = 0.3
    for rx, ry in rng.rand(200, 2):
        L.orc_ggx_sample(n, C.c_float(rx), C.c_float(ry), C.c_float(alpha), out)
        m = np.array(out[:])
        assert abs(np.linalg.norm(m) - 1) < 1e-6 and m[1] > 0
        tan2 = (1 - m[1] ** 2) / m[1] ** 2
        assert abs(tan2 - alpha ** 2 * ry / (1 - ry)) < 1e-3 * max(1.0, tan2)
    L.orc_ggx_sample(n, C.c_float(0.3), C.c_float(0.7), C.c_float(0.0), out)      # alpha 0 -> m == n
    assert np.allclose(out[:], [0, 1, 0], atol=1e-7)


def test_bounce_pdfs(oracle_mod):
    L = oracle_mod.lib()
    f3 = lambda *a: (C.c_float * 3)(*a)
    out = (C.c_float * 6)()
    n = f3(0, 0, 1)
    wi = np.array([0.3, 0.2, 0.9]); wi /= np.linalg.norm(wi)
    for from_camera in (0, 1):
        L.orc_bounce(0, f3(*wi), n, n, C.c_float(1), C.c_float(1.5), C.c_float(0), from_camera, C.c_float(0.37), C.c_float(0.81), out)
        wo, f, c_p, l_p = np.array(out[:3]), out[3], out[4], out[5]
        assert abs(np.linalg.norm(wo) - 1) < 1e-6 and wo[2] > 0
        fwd, rev = (c_p, l_p) if from_camera else (l_p, c_p)
        assert abs(f - wo[2] / np.pi) < 1e-6 and abs(fwd - wo[2] / np.pi) < 1e-6 and abs(rev - wi[2] / np.pi) < 1e-6
    # mirror reflection about m = n
    L.orc_bounce(1, f3(*wi), n, n, C.c_float(1), C.c_float(1.5), C.c_float(0), 1, C.c_float(0), C.c_float(0), out)
    assert np.allclose(out[:3], [-wi[0], -wi[1], wi[2]], atol=1e-6)
    # refraction: the reference's GGX_transmit (trace.metal:243-248) has `1 + eta*(c*c - 1)` under
    # the root (eta, not eta^2 -- eq. 40 of Walter et al. 2007 as printed), so the refracted direction
    # is NOT Snell's; the oracle reproduces the statement as written (quirk Q15 in DESIGN.md).
    L.orc_bounce(2, f3(*wi), n, n, C.c_float(1), C.c_float(1.5), C.c_float(0), 1, C.c_float(0), C.c_float(0), out)
    wo = np.array(out[:3])
    eta, c = 1 / 1.5, wi[2]
    ref = (eta * c - np.sqrt(1 + eta * (c * c - 1))) * np.array([0, 0, 1.0]) - eta * wi
    ref /= np.linalg.norm(ref)
    assert wo[2] < 0 and np.allclose(wo, ref, atol=1e-6)
    snell = np.hypot(wi[0], wi[1]) / 1.5
    assert abs(np.hypot(wo[0], wo[1]) - snell) > 1e-3


def test_bdpt_agrees_with_unidirectional_estimator(oracle_mod):
    """The reference's own cross-check (renderer.py:275-278, 309-316): sum over strategies of
    w*C (the un-normalised BDPT sum) and the unidirectional estimate have the same expectation up to
    the longer paths only BDPT reaches; means agree within 10 % on the diffuse Cornell box."""
    scene = c2.create_scene_from_preset("empty", 48, 48)
    o = oracle_mod.OracleRenderer(scene)
    n = 48
    for _ in range(n):
        o.run_sample()
    bdpt = (o.summed_image / n).mean(axis=(0, 1))
    uni = (o.unidirectional_image_buffer / n).mean(axis=(0, 1))
    assert np.all(np.abs(bdpt - uni) / uni < 0.10), (bdpt, uni)
    assert np.isfinite(o.radiance).all() and (o.summed_sample_counts == n).all()
    # mass of MIS weights per pixel-sample is O(number of path lengths), renderer.py:269-273
    assert 5 < (o.summed_sample_weights / n).mean() < 12


def test_libm_and_detmath_oracles_agree_statistically(oracle_mod):
    scene = c2.create_scene_from_preset("empty", 32, 32)
    a, b = oracle_mod.OracleRenderer(scene), oracle_mod.OracleRenderer(scene, libm=True)
    for _ in range(32):
        a.run_sample(); b.run_sample()
    ma, mb = a.radiance.mean(axis=(0, 1)), b.radiance.mean(axis=(0, 1))
    assert np.all(np.abs(ma - mb) / ma < 0.05)


def test_light_sort_gather_equals_direct_splat(oracle_mod):
    """The reference's K4/K7/bincount/K8 chain (renderer.py:212-250) is a per-pixel sum; the product
    replaces it by atomics.  Check the chain against a direct scatter-add of the same entries."""
    scene = c2.create_scene_from_preset("empty", 40, 24)
    o = oracle_mod.OracleRenderer(scene)
    o.make_light_rays(); o.make_camera_rays(); o.trace_light_rays(); o.trace_camera_rays()
    o.join_paths(); o.finalize_samples()
    B = o.batch_size
    pix = o.out_light_indices.copy()
    keep = np.flatnonzero((pix >= 0) & (pix < B))
    assert len(keep) > B                              # ~5 splats per pixel
    paths, rid = o.out_light_path_indices[keep], o.out_light_ray_indices[keep]
    w, sh = o.out_light_weights[keep], o.out_light_shade[keep]
    rays = o.out_light_paths["rays"]
    prior = rays[paths, np.maximum(rid - 1, 0)]["color"][:, :3]
    mat = o.materials["color"][rays[paths, rid]["material"]][:, :3]
    direct = np.zeros((B, 3), np.float64)
    np.add.at(direct, pix[keep], ((w * sh)[:, None] * prior) * mat)
    wsum = np.zeros(B, np.float64)
    np.add.at(wsum, pix[keep], w)
    fin_w = o.sample_weights.copy()
    o.gather_light_image()
    np.testing.assert_allclose(o.out_light_image[:, :3], direct, rtol=1e-5, atol=1e-12)
    np.testing.assert_allclose(o.sample_weights - fin_w, wsum, rtol=1e-4, atol=1e-6)
    assert (np.diff(o.out_light_indices) >= 0).all()      # bitonic network sorted the keys


def test_stable_light_order_is_the_same_chain_in_slot_order(oracle_mod):
    """`gather_light_image(stable=True)` (the order of the product's reproducible light image: a pixel's records by slot
    `id + s * total_pixels`) against the same records summed per pixel in float32, in exactly that order, in numpy -- byte for
    byte -- and against the bitonic network's order to the tolerance a re-ordered float sum needs."""
    scene = c2.create_scene_from_preset("empty", 40, 24)
    outs = []
    for stable in (False, True):
        o = oracle_mod.OracleRenderer(scene)
        o.make_light_rays(); o.make_camera_rays(); o.trace_light_rays(); o.trace_camera_rays()
        o.join_paths(); o.finalize_samples()
        if stable:
            B = o.batch_size
            pix = o.out_light_indices.copy()
            rays = o.out_light_paths["rays"]
            expect = np.zeros((B, 3), np.float32)
            wsum = np.zeros(B, np.float32)
            for slot in np.flatnonzero((pix >= 0) & (pix < B)):          # ascending slot = (s, source pixel)
                p, ri = o.out_light_path_indices[slot], o.out_light_ray_indices[slot]
                prior = rays[p, max(ri - 1, 0)]["color"][:3]
                mat = o.materials["color"][rays[p, ri]["material"]][:3]
                expect[pix[slot]] += ((o.out_light_weights[slot] * o.out_light_shade[slot]) * prior) * mat
                wsum[pix[slot]] += o.out_light_weights[slot]
            fin_w = o.sample_weights.copy()
        o.gather_light_image(stable=stable)
        if stable:
            assert o.out_light_image[:, :3].tobytes() == expect.tobytes()
            assert o.sample_weights.tobytes() == (fin_w + wsum).tobytes()
        outs.append(o.out_light_image[:, :3].copy())
    np.testing.assert_allclose(outs[1], outs[0], rtol=2e-5, atol=1e-9)
    assert (outs[0] > 0).any()


def test_zero_length_paths_are_defined(oracle_mod):
    """a camera path of length 0 contributes nothing and gets zero filter weights."""
    scene = c2.create_scene_from_preset("empty", 16, 16)
    o = oracle_mod.OracleRenderer(scene)
    o.make_light_rays(); o.make_camera_rays(); o.trace_light_rays(); o.trace_camera_rays()
    o.out_camera_paths[5] = np.zeros(1, oracle_mod.Path)[0]
    o.join_paths()
    assert (o.weight_aggregators[5]["weights"] == 0).all()
    assert (o.weight_aggregators[5]["total_contribution"] == 0).all()


def test_detmath_and_light_rays_match_numpy_restatement(oracle_mod):
    """Second restatement of detmath sin/cos/acos and of K1 (orthonormal frame, uniform hemisphere)
    in numpy float32 agrees with the C oracle bit for bit."""
    from oracle import np_kernels as npk
    x = np.linspace(0, 2 * np.pi, 200001).astype(np.float32)
    s, c = npk.det_sincos(x)
    assert s.tobytes() == oracle_mod.det_math("sin", x).tobytes()
    assert c.tobytes() == oracle_mod.det_math("cos", x).tobytes()
    y = np.linspace(0, 1, 100001).astype(np.float32)
    assert npk.det_acos(y).tobytes() == oracle_mod.det_math("acos", y).tobytes()
    scene = c2.create_scene_from_preset("empty", 80, 50)
    o = oracle_mod.OracleRenderer(scene)
    seeds0 = o.rand_buffer.copy()
    o.make_light_rays()
    org, d, l_imp, li, seeds1 = npk.generate_light_rays(scene.light_triangles, scene.light_surface_areas, seeds0)
    assert o.light_ray_buffer["origin"][:, :3].tobytes() == org.tobytes()
    assert o.light_ray_buffer["direction"][:, :3].tobytes() == d.tobytes()
    assert o.light_ray_buffer["l_importance"].tobytes() == l_imp.tobytes()
    assert np.array_equal(o.light_ray_buffer["triangle"], scene.light_triangle_indices[li])
    assert np.array_equal(o.rand_buffer, seeds1)
    # light rays leave the emitter downwards, from just below it (y = 9.5 - DELTA)
    assert (d[:, 1] <= 0).all() and np.allclose(org[:, 1], 9.5 - 1e-4, atol=1e-6)
    # cosine-hemisphere sampler of the diffuse bounce: unit vectors in the upper hemisphere of n
    n = np.tile(np.array([[0.0, 0.0, 1.0]], np.float32), (1000, 1))
    xa, ya = npk.orthonormal(n)
    rng = np.random.RandomState(1)
    wo = npk.random_hemisphere_cosine(xa, ya, n, rng.rand(1000).astype(np.float32), rng.rand(1000).astype(np.float32))
    assert np.allclose(np.linalg.norm(wo, axis=1), 1, atol=1e-6) and (wo[:, 2] >= 0).all()
    # ... and it is the one the oracle's diffuse bounce uses
    import ctypes as C
    out = (C.c_float * 6)()
    f3 = lambda a: (C.c_float * 3)(*a)
    rr = rng.rand(50, 2).astype(np.float32)
    ref = npk.random_hemisphere_cosine(xa[:50], ya[:50], n[:50], rr[:, 0].copy(), rr[:, 1].copy())
    for k in range(50):
        oracle_mod.lib().orc_bounce(0, f3([0.1, 0.2, 0.97]), f3([0, 0, 1]), f3([0, 0, 1]), C.c_float(1), C.c_float(1.5),
                                    C.c_float(0), 1, C.c_float(float(rr[k, 0])), C.c_float(float(rr[k, 1])), out)
        assert np.array(out[:3], np.float32).tobytes() == ref[k].tobytes()


# ---------------------------------------------------------------------------------------------------------
# Round 4: a SECOND restatement of the two megakernels' loop bookkeeping -- generate_paths
# (trace.metal:381-532) and connect_paths (:620-869) -- and of K6, the sort network and K8, written from the Metal text as scalar Python over the reference's own
# Ray / Path records (oracle/py_kernels.py), compared with the C oracle bit for bit on 16x16 frames of three scenes: the
# Cornell box, the rough-glass scene and the scene with every material type.
def _small_scenes():
    import clive2_amd as c2
    from clive2_amd import struct_types as st
    from clive2_amd.load import get_materials
    from clive2_amd.meshes import icosphere
    mats = get_materials()
    mats["alpha"][5] = 0.1
    glass = c2.create_scene(16, 16, np.array([0, 1.5, 6]), np.array([0, 0, -1]),
                            file_specs=[dict(mesh=icosphere(1, radius=2.0, center=(0.0, 1.0, 0.0)), material=5)], materials=mats)
    many = np.zeros(10, dtype=st.Material)
    many[:8] = get_materials()
    many[8], many[9] = many[5], many[5]
    many["type"][8], many["alpha"][8] = 2, 0.3
    many["type"][9], many["alpha"][9] = 3, 0.05
    many["color"][9, :3] = (0.9, 0.9, 0.9)
    many["alpha"][5] = 0.0                                  # type 1 with alpha 0: the smooth dielectric through the GGX route
    specs = [dict(mesh=icosphere(1, radius=1.6, center=(x, 0.0, z)), material=m)
             for x, z, m in ((-3.5, 0.0, 5), (0.0, -1.0, 8), (3.5, 0.0, 9))]
    all_types = c2.create_scene(16, 16, np.array([0, 1.5, 6]), np.array([0, 0, -1]), file_specs=specs, materials=many)
    return {"cornell": c2.create_scene_from_preset("empty", 16, 16), "glass": glass, "all_material_types": all_types}


@pytest.mark.parametrize("name", ["cornell", "glass", "all_material_types"])
def test_python_restatement_of_k3_and_k5_equals_the_c_oracle(name, oracle_mod):
    from oracle import py_kernels as pk
    scene = _small_scenes()[name]
    B = 16 * 16
    o = oracle_mod.OracleRenderer(scene, seeds=oracle_mod.make_seeds(B, seed=4242))
    for sample in range(2):                                 # the second sample starts from the seeds the first one left
        o.make_light_rays(); o.make_camera_rays()
        seeds = o.rand_buffer.copy()
        o.trace_light_rays()
        out_l, paths_l = pk.generate_paths(o.light_ray_buffer, o.boxes, o.triangles, o.materials, seeds, oracle_mod.Ray, oracle_mod.Path)
        assert paths_l.tobytes() == o.out_light_paths.tobytes(), "light Path[] (generate_paths, trace.metal:381-532)"
        assert out_l.tobytes() == o.out_light_image.tobytes()
        o.trace_camera_rays()
        out_c, paths_c = pk.generate_paths(o.camera_ray_buffer, o.boxes, o.triangles, o.materials, seeds, oracle_mod.Ray, oracle_mod.Path)
        assert paths_c.tobytes() == o.out_camera_paths.tobytes(), "camera Path[]"
        assert out_c.tobytes() == o.out_camera_image.tobytes(), "unidirectional estimate (trace.metal:523-528)"
        assert np.array_equal(seeds, o.rand_buffer)
        o.join_paths()
        res = pk.connect_paths(o.out_camera_paths, o.out_light_paths, o.triangles, o.materials, o.boxes, o.camera,
                               oracle_mod.Ray, oracle_mod.WeightAggregator, o.n_light)
        agg = o.weight_aggregators
        for f in ("weights", "total_contribution", "contrib_weight_sum"):
            assert res["aggregators"][f].tobytes() == agg[f].tobytes(), f"aggregator field {f} (connect_paths, trace.metal:620-869)"
        assert res["out"].tobytes() == o.out_samples.tobytes()
        assert np.array_equal(res["light_pixel_indices"], o.out_light_indices)
        assert np.array_equal(res["light_path_indices"], o.out_light_path_indices)
        assert np.array_equal(res["light_ray_indices"], o.out_light_ray_indices)
        assert res["light_weights"].tobytes() == o.out_light_weights.tobytes()
        assert res["light_shade"].tobytes() == o.out_light_shade.tobytes()
        # the comparison must have had something to compare: long subpaths, t = 1 splats, t >= 2 joins, every material type
        assert (o.out_camera_paths["length"] >= 4).sum() > 20 and (o.out_light_paths["length"] >= 4).sum() > 20
        assert (o.out_light_indices >= 0).sum() > 50 and (agg["contrib_weight_sum"] > 0).sum() > 100
        if name == "all_material_types":
            seen = set(np.unique(o.out_camera_paths["rays"]["material"][o.out_camera_paths["length"] > 1, 1]).tolist())
            assert {5, 8, 9} <= seen, seen
        # K6, the sort network with its driver loop, K8 (trace.metal:981-1018, :872-934 + renderer.py:212-231, :937-964)
        sorted_in = [a.copy() for a in (o.out_light_indices, o.out_light_path_indices, o.out_light_ray_indices, o.out_light_weights, o.out_light_shade)]
        o.finalize_samples()
        fin, counts, sw = pk.adaptive_finalize_samples(o.weight_aggregators, o.camera, o.summed_bins_buffer)
        assert fin.tobytes() == o.finalized_samples.tobytes() and np.array_equal(counts, o.sample_counts)
        assert sw.tobytes() == o.sample_weights.tobytes()
        o.gather_light_image()
        pk.light_sort_all(*sorted_in)
        for mine, theirs in zip(sorted_in, (o.out_light_indices, o.out_light_path_indices, o.out_light_ray_indices, o.out_light_weights, o.out_light_shade)):
            assert mine.tobytes() == theirs.tobytes(), "bitonic network (not a stable sort: the order inside a pixel's run decides the float sums)"
        bins, offset = o.light_bins()
        img = pk.light_image_gather(o.out_light_paths, o.materials, sorted_in[1], sorted_in[2], bins.astype(np.int32), offset, sorted_in[3], sorted_in[4], sw)
        assert img.tobytes() == o.out_light_image.tobytes() and sw.tobytes() == o.sample_weights.tobytes()
        o.process_images()
