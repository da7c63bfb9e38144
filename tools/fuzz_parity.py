"""Differential fuzzing: random scenes (camera pose, meshes, materials, frame size, launch organisation)
rendered by the HIP path and by the oracle; subpaths, RNG state and filter aggregators must match bit for
bit, the accumulated image to 5e-5.   python tools/fuzz_parity.py [n_scenes] [first_seed]"""
import os
import sys
import time

os.environ.setdefault("OMP_NUM_THREADS", str(min(os.cpu_count() or 1, 16)))   # the oracle is OpenMP code
os.environ.setdefault("OMP_WAIT_POLICY", "passive")

sys.path.insert(0, ".")
import numpy as np

import clive2_amd as c2
from clive2_amd import struct_types as st
from clive2_amd.load import get_materials, triangles_for_box
from clive2_amd.meshes import icosphere, noisy_blob
from clive2_amd.renderer import Renderer, make_seeds
from oracle import oracle as orc


_FLOAT_FIELDS = ("origin", "direction", "inv_direction", "color", "normal", "c_importance", "l_importance", "tot_importance")


def canonical_nans(paths):
    """A copy of a Path[] array with every NaN of its float fields in ONE bit pattern.  A NaN that both sides compute in the same
    place (a reverse pdf of 0 / 0 at a grazing vertex: 3 scenes in 9,000, round 6) is the same result; its sign and payload are the
    hardware's default for an invalid operation -- 0x7fc00000 on the GPU, 0xffc00000 ("real indefinite") on the x86 host of the oracle,
    whatever Metal's is on the reference's GPU -- and say nothing about the computation."""
    out = paths.copy()
    rays = out["rays"]
    for f in _FLOAT_FIELDS:
        a = rays[f]
        a[np.isnan(a)] = np.float32(np.nan)
    return out


def random_scene(rng, max_members=None):
    w, h = int(rng.randint(17, 120)), int(rng.randint(11, 80))
    mats = np.zeros(12, dtype=st.Material)
    mats[:8] = get_materials()
    for m in range(8, 12):
        mats[m] = mats[rng.randint(0, 6)]
        mats["type"][m] = rng.randint(0, 4)
        mats["alpha"][m] = rng.choice([0.0, 0.02, 0.2, 0.8])
        mats["ior"][m] = rng.choice([1.1, 1.5, 2.4])
    mats["alpha"][5] = rng.choice([0.0, 0.1])
    specs = []
    for _ in range(rng.randint(0, 4)):
        sub = int(rng.randint(0, 4))
        mesh = icosphere(sub, radius=float(rng.uniform(0.5, 2.5))) if rng.rand() < 0.7 else noisy_blob(sub, radius=float(rng.uniform(0.8, 2.0)), center=(0, 0, 0), seed=int(rng.randint(1 << 30)))
        specs.append(dict(mesh=mesh, material=int(rng.choice([0, 1, 4, 5, 8, 9, 10, 11])),
                          offset=np.array([rng.uniform(-6, 6), rng.uniform(-1, 6), rng.uniform(-7, 3)])))
    theta = rng.uniform(0, 2 * np.pi) if rng.rand() < 0.5 else 0.0
    center = np.array([np.sin(theta) * 7.0, rng.uniform(0.0, 5.0), np.cos(theta) * 7.0])
    direction = np.array([-np.sin(theta), rng.choice([0.0, 0.0, -0.2, 0.15]), -np.cos(theta)])
    direction = direction / np.linalg.norm(direction)
    builder = rng.choice(["numpy", "native"])
    room = None
    if rng.rand() < 0.3:                    # an open scene: the emitter and a random subset of the walls
        box = triangles_for_box()
        room = [t for t in box if t.emitter or rng.rand() < 0.5]
    return (c2.create_scene(w, h, center, direction, file_specs=specs, materials=mats, bvh_builder=str(builder), room=room, max_members=max_members),
            (w, h, len(specs), str(builder), 'open' if room is not None else 'closed') + ((f"mm{max_members}",) if max_members else ()))


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    orc.build()
    bad = 0
    for k in range(first, first + n):
        rng = np.random.RandomState(1000 + k)
        # round 6: every fifth scene is built with another leaf size (create_scene(max_members=...): 4, 2 or 1 instead of the reference's 8;
        # taken from the scene number, not from the generator, so that the scenes themselves stay what earlier rounds rendered)
        scene, desc = random_scene(rng, max_members=(4, 2, 1)[(k // 5) % 3] if k % 5 == 4 else None)
        B = scene.pixel_width * scene.pixel_height
        mode, levels, stages = int(rng.randint(0, 6)), int(rng.randint(0, 7)), int(rng.randint(-1, 3))
        flags = 0
        gather = None
        if rng.rand() < 0.5:                # whole-subpath launch: random bounce batching, either step form
            budget, lanes, wait, step = int(rng.choice([0, 4, 5, 6, 7])), int(rng.randint(0, 65)), int(rng.randint(0, 100)), int(rng.randint(0, 2))
            flags |= step << 12                    # (the register-budget variants of round 2 are gone; the draw stays for the seeds' sake)
            gather = (lanes, wait)
        # round 4: K sample streams (stream j = the renderer seeded with buffer j alone) and the wide walk's LDS window (none / 32 / 64 / 96 nodes)
        K = int(rng.choice([1, 1, 2, 3]))
        flags |= int(rng.choice([0, 15, 1, 2, 3])) << 20
        if rng.rand() < 0.3:
            flags |= 8                             # the 4-wide walk for the per-level subpath launches in the serial order too
        # round 5: the 4-wide walk without its speculative stack-top expansion (bit 13); the reproducible light image (records +
        # stable sort + ordered gather instead of float atomics: the oracle then sums a pixel's records in slot order, and with one
        # sample stream image and weights must agree byte for byte)
        if rng.rand() < 0.3:
            flags |= 1 << 13
        repro = bool(rng.rand() < 0.4)
        # round 6: the 4-wide walks read 36-byte triangle records, a pair as one run of 72 bytes; bit 14 = the 48-byte records (drawn
        # after everything else, so that earlier rounds' scenes keep their settings)
        if rng.rand() < 0.3:
            flags |= 1 << 14
        seeds = [make_seeds(B, seed=k, rank=j) for j in range(K)]
        r = Renderer(scene, seeds=seeds[0] if K == 1 else np.stack(seeds), streams=K)
        os_ = [orc.OracleRenderer(scene, seeds=sd) for sd in seeds]
        o = os_[0]
        r.set_traversal_mode(mode)
        r.set_debug_flags(flags)
        if gather:
            r.set_subpath_gather(*gather)
        r.set_levels_per_launch(levels)
        r.set_pipelining(stages)
        r.set_reproducible(repro)
        t0 = time.time()
        ok = True
        failed = []                                  # names of the checks that failed (printed with a MISMATCH)

        def chk(name, cond):
            if not cond:
                failed.append(name)
            return bool(cond)

        def same_paths():
            good = True
            for j, oj in enumerate(os_):
                r.set_export_stream(j)
                for which, ref in ((0, oj.out_light_paths), (1, oj.out_camera_paths)):
                    got = r.export_paths(which)
                    same = got.tobytes() == ref.tobytes() or canonical_nans(got).tobytes() == canonical_nans(ref).tobytes()
                    good &= same
                    if got.tobytes() != ref.tobytes() and os.environ.get("FUZZ_EXPLAIN") == "1":
                        # which fields of which vertices differ (FUZZ_EXPLAIN=1 python tools/fuzz_parity.py 1 <scene>)
                        for name in got.dtype.names:
                            if name == "rays":
                                for f in got["rays"].dtype.names:
                                    a, b = got["rays"][f], ref["rays"][f]
                                    bad_ = np.argwhere(a.view(np.uint32 if a.dtype.itemsize == 4 else a.dtype) != b.view(np.uint32 if b.dtype.itemsize == 4 else b.dtype))
                                    if len(bad_):
                                        i = tuple(bad_[0])
                                        print(f"   stream {j} kind {which}: rays.{f} differs at {len(bad_)} places, first {i}: hip {a[i]!r} (bits {np.asarray(a[i]).view(np.uint32).ravel()[0]:#010x}) oracle {b[i]!r} (bits {np.asarray(b[i]).view(np.uint32).ravel()[0]:#010x}); path length there hip {got['length'][i[0]]} oracle {ref['length'][i[0]]}", flush=True)
                            elif not np.array_equal(got[name], ref[name]):
                                bad_ = np.argwhere(got[name] != ref[name])
                                print(f"   stream {j} kind {which}: {name} differs at {len(bad_)} places, first {tuple(bad_[0])}: hip {got[name][tuple(bad_[0])]} oracle {ref[name][tuple(bad_[0])]}", flush=True)
            return good

        r.make_light_rays(); r.make_camera_rays(); r.trace_light_rays(); r.trace_camera_rays()
        for x in os_:
            x.make_light_rays(); x.make_camera_rays(); x.trace_light_rays(); x.trace_camera_rays()
        ok &= chk('paths of the first sample', same_paths())
        r.join_paths(); r.finalize_samples(); r.gather_light_image(); r.process_images()
        for x in os_:
            x.join_paths(); x.finalize_samples(); x.gather_light_image(stable=repro); x.process_images()
        for j, oj in enumerate(os_):
            r.set_export_stream(j)
            agg = r.export_aggregators()
            ok &= chk('aggregators', agg["total_contribution"].tobytes() == oj.weight_aggregators["total_contribution"].tobytes())
            ok &= chk('aggregator weights', agg["weights"].tobytes() == oj.weight_aggregators["weights"].tobytes())
        r.run_samples(3)
        for x in os_:
            x.run_sample(repro); x.run_sample(repro); x.run_sample(repro)
        ok &= chk('RNG state', bool(np.array_equal(r.get_random_buffer().reshape(K, B, 2), np.stack([x.rand_buffer for x in os_]))))
        ok &= chk('paths of the last sample', same_paths())
        ok &= chk('unidirectional image (rtol 1e-6 / 2e-6)', bool(np.allclose(r.read_accumulators()[3], sum(x.unidirectional_image_buffer for x in os_), rtol=2e-6 if K > 1 else 1e-6, atol=0)))
        img = r.read_accumulators()[0]
        ok &= chk('image (rtol 5e-5)', bool(np.allclose(img, sum(x.summed_image for x in os_), rtol=5e-5, atol=1e-8)))
        if repro and K == 1:
            ok &= chk('reproducible image bytes', img.tobytes() == o.summed_image.tobytes() and r.read_accumulators()[1].tobytes() == o.summed_sample_weights.tobytes())
        ok &= chk('ray count', r.counters()["rays"] == sum(x.rays_traced for x in os_))
        # the device tone map against the host path on the same accumulators (a byte may move by one where 255*x/(x+w) sits on an integer)
        with np.errstate(all="ignore"):
            for which in ("image", "unidirectional_image"):
                dv, hv = r.tone_mapped(which), getattr(r, which)
                dd = np.abs(dv.astype(np.int16) - hv.astype(np.int16))
                ok &= chk(f'device tone map of {which} (max diff {int(dd.max())}, {int((dd > 0).sum())} bytes differ)', bool(dd.max() <= 1 and int((dd > 0).sum()) <= 2))
        print(f"scene {k}: {desc} tris={len(scene.triangles)} mode={mode} levels={levels} stages={stages} K={K} flags={flags:#x} repro={int(repro)} "
              f"len_c={o.out_camera_paths['length'].mean():.2f} {'OK' if ok else 'MISMATCH: ' + '; '.join(failed)} ({time.time() - t0:.1f}s)", flush=True)
        bad += not ok
        r.close()
    print(f"RESULT: {n - bad}/{n} scenes match")
    return 1 if bad else 0


if __name__ == "__main__":
    raise SystemExit(main())
