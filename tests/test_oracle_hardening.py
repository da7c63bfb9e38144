"""Pins of the oracle that do NOT come from reading the reference a second time:

* closed forms that follow from physics / geometry alone: the differential form factor of the 5 x 5 emitter seen
  from the floor centre (cosine sampling + traversal), mean cosines of the two hemisphere samplers, the white-furnace
  building block "Lambert BRDF x cos / pdf = albedo" and the furnace series it sums to;
* the structure of the MIS stage (trace.metal:745-776): per connected pair the balance weights over all strategies
  of the same path sum to 1, and the pdf chain agrees with a float64 recomputation from the stored Path records;
* committed SHA-256 hashes of the oracle's stage outputs (tests/golden/oracle_stage_hashes.json), which the oracle (CPU)
  AND the HIP path (GPU) must reproduce -- so that an edit applied identically to both sides' arithmetic
  (oracle/detmath.h and csrc/detmath.hpp are sibling files) cannot pass unnoticed.  A mutation test shows that it would not.
"""
import ctypes as C
import json
import os
import shutil
import subprocess

import numpy as np
import pytest

import clive2_amd as c2

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden", "oracle_stage_hashes.json")


def _golden_scenes():
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_stage_hashes", os.path.join(ROOT, "tests", "golden", "make_stage_hashes.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


# ------------------------------------------------------------------ committed stage hashes (drift pin)
def test_oracle_reproduces_the_committed_stage_hashes(oracle_mod):
    gold = json.load(open(GOLDEN))
    m = _golden_scenes()
    for name, scene in m.scenes().items():
        assert m.oracle_hashes(scene, seed=gold["seed"]) == gold["scenes"][name], name


@pytest.mark.gpu
def test_hip_path_reproduces_the_committed_stage_hashes(oracle_mod):
    """The HIP path against the COMMITTED hashes (not against an oracle built from today's sources)."""
    import hashlib
    from clive2_amd.renderer import Renderer, make_seeds, LIGHT, CAMERA
    gold = json.load(open(GOLDEN))
    for name, scene in _golden_scenes().scenes().items():
        r = Renderer(scene, seeds=make_seeds(scene.pixel_width * scene.pixel_height, seed=gold["seed"]))
        r.make_light_rays(); r.make_camera_rays(); r.trace_light_rays(); r.trace_camera_rays(); r.join_paths()
        agg = r.export_aggregators()
        h = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()
        got = {"light_paths": h(r.export_paths(LIGHT)), "camera_paths": h(r.export_paths(CAMERA)),
               "aggregator_total": h(agg["total_contribution"]), "aggregator_weights": h(agg["weights"]),
               "aggregator_weight_sum": h(agg["contrib_weight_sum"]), "seeds": h(r.get_random_buffer())}
        r.close()
        assert got == gold["scenes"][name], name


def test_an_identical_edit_to_both_detmath_files_would_be_caught(tmp_path, oracle_mod):
    """Mutation test: one polynomial coefficient of det_sinf nudged by one decimal digit (the kind of edit that, applied to
    oracle/detmath.h and csrc/detmath.hpp alike, keeps GPU == oracle green).  The mutated oracle no longer reproduces the
    committed hashes of the glass scene."""
    src = tmp_path / "oracle"
    shutil.copytree(os.path.join(ROOT, "oracle"), src, ignore=shutil.ignore_patterns("*.so", "__pycache__", "_ref"))
    text = (src / "detmath.h").read_text()
    assert "8.3321608736E-3f" in text                      # a sine coefficient (Cephes sinf)
    (src / "detmath.h").write_text(text.replace("8.3321608736E-3f", "8.3321618736E-3f", 1))
    subprocess.run(["make", "-C", str(src), "-B", "liboracle.so"], check=True, stdout=subprocess.DEVNULL)
    code = f"""
import sys, json, importlib.util
sys.path.insert(0, {str(tmp_path)!r}); sys.path.insert(1, {ROOT!r})
spec = importlib.util.spec_from_file_location('m', {os.path.join(ROOT, 'tests', 'golden', 'make_stage_hashes.py')!r})
m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
sys.path.remove({ROOT!r}); sys.path.insert(0, {str(tmp_path)!r})
import oracle.oracle as o
assert o._HERE.startswith({str(tmp_path)!r}), o._HERE
print(json.dumps({{k: m.oracle_hashes(s) for k, s in m.scenes().items()}}))
"""
    env = dict(os.environ, PYTHONPATH=f"{tmp_path}:{ROOT}")
    out = subprocess.run(["python", "-c", code], stdout=subprocess.PIPE, text=True, check=True, env=env).stdout
    got = json.loads(out.strip().splitlines()[-1])
    gold = json.load(open(GOLDEN))["scenes"]
    assert got["glass_64x48"]["camera_paths"] != gold["glass_64x48"]["camera_paths"]
    assert got["glass_64x48"]["light_paths"] != gold["glass_64x48"]["light_paths"]


# ------------------------------------------------------------------ closed forms
def _quad_form_factor(a, b, h):
    """Differential area -> parallel rectangle [-a,a] x [-b,b] centred above it at height h (Howell catalogue B-3 x 4)."""
    def corner(a, b):
        A, B = np.hypot(a, h), np.hypot(b, h)
        return (a / A * np.arctan(b / A) + b / B * np.arctan(a / B)) / (2 * np.pi)
    return 4 * corner(a, b)


def test_direct_irradiance_matches_the_analytic_quad_form_factor(oracle_mod):
    """Irradiance at the floor centre under the 5 x 5 emitter (constants.py:33-36: x,z in [-2.5,2.5] at y = 9.5, floor at
    y = -2) is pi L F with F the differential form factor of the quad.  The path tracer's estimate of F is the share of
    cosine-distributed bounce directions (diffuse_bounce, trace.metal:334-346) whose closest hit (traverse_bvh, :144-176) is
    an emitter triangle."""
    scene = c2.create_scene_from_preset("empty", 16, 16)
    n = 1_000_000
    rng = np.random.RandomState(11)
    items = np.zeros((n, 12), np.float32)
    items[:, 0:3] = (0.0, 1.0, 0.0)          # wi: arriving from straight above
    items[:, 4] = 1.0                          # n = +y
    items[:, 6] = rng.random_sample(n).astype(np.float32)
    items[:, 7] = rng.random_sample(n).astype(np.float32)
    items[:, 8], items[:, 9] = 1.0, 1.5
    out = np.zeros((n, 8), np.float32)
    L = oracle_mod.lib()
    L.orc_bounce_batch(C.c_int(n), C.c_int(1), items.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p))
    wo = out[:, :3]
    assert np.allclose(np.linalg.norm(wo, axis=1), 1.0, atol=1e-5)
    # cosine-weighted: E[cos] = 2/3, f == pdf == cos/pi
    cos = wo[:, 1].astype(np.float64)
    assert abs(cos.mean() - 2.0 / 3.0) < 4 * np.sqrt(1.0 / 18.0 / n)
    assert np.allclose(out[:, 3], cos / np.pi, rtol=2e-6, atol=1e-9) and np.array_equal(out[:, 3], out[:, 4])
    rays = np.zeros(n, oracle_mod.Ray)
    rays["origin"][:, :3] = (0.0, -2.0, 0.0)
    rays["direction"][:, :3] = wo
    with np.errstate(divide="ignore"):
        rays["inv_direction"][:, :3] = np.float32(1.0) / wo
    bi, bt, _, _, _ = oracle_mod.traverse(rays, scene.boxes, scene.triangles)
    assert (bi >= 0).all()                                            # a closed room: every ray hits something
    hits_light = scene.triangles["is_light"][bi] != 0
    F = _quad_form_factor(2.5, 2.5, 11.5)
    assert 0.05 < F < 0.06
    sigma = np.sqrt(F * (1 - F) / n)
    assert abs(hits_light.mean() - F) < 4 * sigma, (hits_light.mean(), F, sigma)
    assert np.allclose(bt[hits_light] * wo[hits_light, 1], 11.5, rtol=1e-5)      # they end on the plane y = 9.5


def test_light_rays_leave_the_emitter_uniformly(oracle_mod):
    """K1 (trace.metal:1070-1124): origins uniform over the 25-unit^2 emitter, directions uniform over the hemisphere
    below it (mean cosine 1/2), l_importance = 1 / (count x area)."""
    scene = c2.create_scene_from_preset("empty", 512, 256)
    o = oracle_mod.OracleRenderer(scene, seeds=oracle_mod.make_seeds(512 * 256, seed=3))
    o.make_light_rays()
    r = o.light_ray_buffer
    n = len(r)
    org, d = r["origin"][:, :3].astype(np.float64), r["direction"][:, :3].astype(np.float64)
    assert np.allclose(org[:, 1], 9.5 - 1e-4, atol=1e-5)              # DELTA below the emitter plane
    assert np.abs(org[:, [0, 2]]).max() <= 2.5 + 1e-6
    cells = np.histogram2d(org[:, 0], org[:, 2], bins=5, range=[[-2.5, 2.5], [-2.5, 2.5]])[0]
    chi2 = ((cells - n / 25.0) ** 2 / (n / 25.0)).sum()
    assert chi2 < 60.0                                                # 24 degrees of freedom: p(chi2 > 60) ~ 6e-5
    cos = -d[:, 1]
    assert (cos >= 0).all() and abs(cos.mean() - 0.5) < 4 * np.sqrt(1.0 / 12.0 / n)
    assert np.allclose(r["l_importance"], 1.0 / (2 * 12.5), rtol=1e-6)


def _grey_room(albedo, w=48, h=32):
    from clive2_amd.load import get_materials
    mats = get_materials()
    mats["color"][:, :3] = albedo
    mats["type"][:] = 0
    return c2.create_scene(w, h, np.array([0, 1.5, 6]), np.array([0, 0, -1]), file_specs=[], materials=mats)


def test_white_furnace_series(oracle_mod):
    """Closed room, every surface Lambertian with albedo rho.  A path tracer that samples the cosine lobe carries the
    throughput  prod_k (f_k cos_k / pdf_k) rho = rho^k  -- the factor by which an emission Le seen after k bounces
    counts, so that the furnace sums to Le (1 - rho^n) / (1 - rho).  In the stored subpaths that product is
    color_k / (tot_importance_{k+1} / tot_importance_1): the BRDF factors sit in `color`, the pdfs in `tot_importance`
    (trace.metal:489-507)."""
    rho = 0.7
    scene = _grey_room(rho)
    o = oracle_mod.OracleRenderer(scene, seeds=oracle_mod.make_seeds(48 * 32, seed=5))
    o.make_light_rays(); o.make_camera_rays(); o.trace_light_rays(); o.trace_camera_rays()
    for paths, first_color in ((o.out_camera_paths, 1.0), (o.out_light_paths, 1.0)):
        full = paths[paths["length"] == 6]                 # a closed room: (nearly) every path reaches the bounce limit
        assert len(full) > 0.95 * len(paths)
        # the emitter quad and the camera quad hang INSIDE the room and can be hit from behind; the reference applies the
        # albedo on front-side reflection only (trace.metal:489-494), so those paths are left out of the identity
        inner = np.isin(full["rays"]["material"][:, 1:6], (6, 7)).any(axis=1)
        full = full[~inner]
        assert len(full) > 0.8 * len(paths)
        col = full["rays"]["color"][:, :, :3].astype(np.float64)         # [path, vertex, channel]
        tot = full["rays"]["tot_importance"].astype(np.float64)
        series = np.zeros(len(full))
        for k in range(1, 5):
            thr = col[:, k, 0] / col[:, 0, 0] / (tot[:, k + 1] / tot[:, 1])
            assert np.allclose(thr, rho ** k, rtol=2e-5), k
            series += thr
        assert np.allclose(1.0 + series, (1 - rho ** 5) / (1 - rho), rtol=2e-5)


# ------------------------------------------------------------------ MIS structure
def _unified(cam, lig, t, s, i):
    return lig["rays"][i] if i < s else cam["rays"][t + s - i - 1]      # get_ray, trace.metal:546-549


def _G(a, b):
    ca = abs(float(np.dot(a["direction"][:3].astype(np.float64), a["normal"][:3].astype(np.float64))))
    cb = abs(float(np.dot(b["direction"][:3].astype(np.float64), b["normal"][:3].astype(np.float64))))
    d = b["origin"][:3].astype(np.float64) - a["origin"][:3].astype(np.float64)
    return ca * cb / float(np.dot(d, d))


@pytest.mark.parametrize("which", ["cornell", "glass"])
def test_balance_weights_sum_to_one_and_follow_the_pdf_chain(which, oracle_mod, cornell_small, glass_scene):
    """For every connected (t,s): w_i = p_i / sum_j p_j over ALL strategies i of that path sums to 1 and w_s is the
    weight used (trace.metal:745-776); the excluded strategy s+t carries 0; specular vertices zero their two
    neighbours.  For t >= 2 the whole chain p_0..p_{s+t-1} is recomputed in float64 from the stored vertices."""
    scene = cornell_small if which == "cornell" else glass_scene
    o = oracle_mod.OracleRenderer(scene, seeds=oracle_mod.make_seeds(64 * 48, seed=9))
    o.make_light_rays(); o.make_camera_rays(); o.trace_light_rays(); o.trace_camera_rays()
    log = oracle_mod.strategy_log(o)
    mats = scene.materials
    connected = log[..., 0] == 1.0
    assert connected.sum() > 20000
    p = log[..., 9:22].astype(np.float64)
    k = log[..., 8].astype(int)
    w, p_s, total = log[..., 1].astype(np.float64), log[..., 2].astype(np.float64), log[..., 3].astype(np.float64)
    ids, ts, ss = np.nonzero(connected)
    # (a) weights of one path sum to one; the used weight is the s-th of them; p[s+t] = 0
    psum = p[connected].sum(axis=1)
    assert np.allclose(psum, total[connected], rtol=1e-5)
    assert np.allclose((p[connected] / psum[:, None]).sum(axis=1), 1.0, rtol=1e-12)
    assert np.allclose(w[connected], p[ids, ts, ss, ss] / psum, rtol=1e-5)
    assert (p[ids, ts, ss, k[connected]] == 0).all()
    assert ((w[connected] > 0) & (w[connected] <= 1.0 + 1e-6)).all()
    # a diffuse-only scene: nothing is zeroed below s+t, and p[s] is p_s itself
    if which == "cornell":
        assert np.array_equal(p[ids, ts, ss, ss].astype(np.float32), p_s[connected].astype(np.float32))
    # (b) float64 recomputation of the chain for a sample of t >= 2 pairs
    rng = np.random.RandomState(1)
    pick = rng.choice(np.nonzero(ts >= 2)[0], size=600, replace=False)
    checked_spec = 0
    for j in pick:
        pid, t, s = int(ids[j]), int(ts[j]), int(ss[j])
        cam, lig = o.out_camera_paths[pid], o.out_light_paths[pid]
        n = s + t
        x = [_unified(cam, lig, t, s, i) for i in range(n)]
        r = np.zeros(n)
        for i in range(n):
            if i == 0:
                r[i] = x[0]["l_importance"] / (x[0]["c_importance"] * _G(x[0], x[1]))
            elif i == n - 1:
                r[i] = x[i]["l_importance"] * _G(x[i], x[i - 1]) / x[i]["c_importance"]
            else:
                r[i] = x[i]["l_importance"] * _G(x[i - 1], x[i]) / (x[i]["c_importance"] * _G(x[i], x[i + 1]))
        q = np.zeros(n + 2)
        q[s] = float(cam["rays"][t - 1]["tot_importance"]) * (float(lig["rays"][s - 1]["tot_importance"]) if s > 0 else 1.0)
        for i in range(s, n):
            q[i + 1] = r[i] * q[i]
        for i in range(s - 1, -1, -1):
            q[i] = q[i + 1] / r[i]
        for i in range(n):
            if mats["type"][x[i]["material"]] > 0:
                q[i] = 0.0; q[i + 1] = 0.0
                checked_spec += 1
        q[n] = 0.0
        assert np.allclose(p[pid, t, s, :n + 1], q[:n + 1], rtol=2e-4, atol=0), (pid, t, s)
    if which == "glass":
        assert checked_spec > 0
