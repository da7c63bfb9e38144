#!/usr/bin/env python3
"""bench.py -- BASELINE.json metric: Mrays/s (whole job) on the 1080p Cornell box, N x MI355X.

One "step" = one sample of the full BDPT pipeline (Renderer.run_sample: light + camera subpaths,
all (t,s) connections, light splat, filter, accumulate) over the whole 1920x1080 frame on every
rank.  Ranks are weak-scaled (each integrates its own samples of the replicated scene, own seed
buffer); the accumulators are summed once with an RCCL all-reduce inside the timed region.
One "ray" = one closest-hit BVH query; rays are counted on the device.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--width 1920 --height 1080]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line.  `roofline` prices the dominant traversal kernel with the
algorithmic bytes of SURVEY.md §8(d): B_ray = 48 + 32*N_node + 36*N_tri, N_node/N_tri measured by
device counters in a separate (untimed) counting pass, kernel time from HIP events recorded on
the renderer's stream during the timed region.  `cpu_baseline` times the C oracle (the CPU
restatement of the reference's kernels, OpenMP over host cores) on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md


def cpu_baseline(width, height, samples):
    """Oracle (kind 'port'): the CPU restatement of trace.metal's kernels + renderer.py's host glue,
    timed end to end on this host's cores.  Checker code used as the reported baseline only."""
    # a GPU box exposes all host threads but grants a 16-CPU share: size the OpenMP team to it
    os.environ.setdefault("OMP_NUM_THREADS", str(min(os.cpu_count() or 1, 16)))
    os.environ.setdefault("OMP_WAIT_POLICY", "passive")
    import numpy as np
    import clive2_amd as c2
    from oracle import oracle as orc
    orc.build()
    scene = c2.create_scene_from_preset("empty", width, height)
    o = orc.OracleRenderer(scene, seeds=orc.make_seeds(width * height))
    t0 = time.perf_counter()
    for _ in range(samples):
        o.run_sample()
    dt = time.perf_counter() - t0
    threads = int(os.environ["OMP_NUM_THREADS"])
    return {"value": round(o.rays_traced / dt / 1e6, 3), "unit": "Mrays/s", "cores": threads, "kind": "port",
            "sample": f"Cornell box {width}x{height}, {samples} sample(s) of the full BDPT pipeline "
                      f"({o.rays_traced} rays, {dt:.1f} s; C oracle, OpenMP)"}


def measured_traffic(kernel):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes of this same command
    (FETCH_SIZE and WRITE_SIZE in separate passes, FETCH_SIZE doubled per the gfx950 note of
    MI355X_MICROARCH.md); None when no summary is committed."""
    path = os.path.join(ROOT, "profiles", "r01_final_pmc_hbm.json")
    try:
        table = json.load(open(path))["hbm_bytes_per_launch"]
    except (OSError, KeyError, ValueError):
        return None, None
    for name, val in table.items():
        if kernel in name and "true" not in name.split("<")[-1]:
            return val, "profiles/r01_final_pmc_hbm.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, 1080p Cornell)"
    return None, None


def measured_valu(kernel):
    """VALU wave-instructions per launch of `kernel` from the committed rocprofv3 SQ pass
    (profiles/r01_final_pmc_sq.txt, SQ_INSTS_VALU); None when no summary is committed."""
    path = os.path.join(ROOT, "profiles", "r01_final_pmc_sq.txt")
    try:
        for line in open(path):
            if line.startswith(kernel + "<false>") and "INSTS_VALU=" in line:
                return float(line.split("INSTS_VALU=")[1].split()[0])
    except (OSError, ValueError, IndexError):
        pass
    return None


def build_scene(name, W, H):
    import numpy as np
    import clive2_amd as c2
    if name == "cornell":
        return c2.create_scene_from_preset("empty", W, H), "Cornell box (scene preset 'empty', 16 tris / 5 boxes)"
    from clive2_amd.load import get_materials
    from clive2_amd import meshes
    mats = get_materials()
    mats["alpha"][5] = 0.1                       # rough glass (the shipped table has alpha 0)
    room = None
    if name == "open":
        from clive2_amd.load import triangles_for_box
        room = [t for t in triangles_for_box() if t.emitter or t.n[1] > 0.5 or t.n[2] > 0.5]
        specs = [dict(mesh=meshes.icosphere(2, radius=1.5), material=5, offset=np.array([0.5, 0.0, -1.0]))]
        desc = "open scene: emitter, floor, back wall and a 320-tri glass ball (most subpaths leave after 1-3 bounces)"
    elif name == "glass":
        specs = [dict(mesh=meshes.icosphere(4, radius=2.0, center=(0.0, 1.0, 0.0)), material=5)]
        desc = "Cornell box + 5,120-tri rough-glass icosphere (config 3 stand-in)"
    elif name == "blob":
        specs = [dict(mesh=meshes.noisy_blob(subdiv=6), material=5)]
        desc = "Cornell box + 81,920-tri noisy blob (config 4 stand-in)"
    else:
        specs = [dict(mesh=(v, f), material=m) for v, f, m in meshes.interior_grid()]
        desc = "Cornell box + 49 x 20,480-tri icospheres (config 5 stand-in)"
    s = c2.create_scene(W, H, np.array([0, 1.5, 6]), np.array([0, 0, -1]), file_specs=specs, materials=mats, room=room)
    return s, desc + f", {len(s.triangles)} tris / {len(s.boxes)} boxes"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=128)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--debug-flags", type=int, default=0, help="performance experiments only (invalid renders)")
    ap.add_argument("--scene", default="cornell", choices=["cornell", "glass", "blob", "interior", "open"],
                    help="cornell = the BASELINE metric's workload (default); the others are the mesh configs "
                         "(SURVEY 8d C3-C5 stand-ins), for profiling only")
    ap.add_argument("--levels-per-launch", type=int, default=0, help="subpath bounces per launch (1..6; 0 = by survival, the default)")
    ap.add_argument("--pipelining", type=int, default=-1, help="sample pipeline: 0 serial, 1 subpaths of sample i+1 beside the connection phase of sample i, 2 three stages, -1 by frame size (default)")
    ap.add_argument("--traversal-mode", type=int, default=0, help="0 auto, 1 fused, 2 persistent traversal with ray replacement, 3 fused subpaths + persistent connection rays")
    ap.add_argument("--cpu-width", type=int, default=1920)
    ap.add_argument("--cpu-height", type=int, default=1080)
    ap.add_argument("--cpu-samples", type=int, default=4)
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
        args.gpus = world

    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    # Rehearsal knobs (NOT used by the driver): CLIVE2_BENCH_BACKEND=gloo reduces through the host and
    # CLIVE2_BENCH_SHARE_GPU=1 puts every rank on cuda:0, so the N>1 code path can be exercised on a
    # one-GPU box.  The measured configuration is nccl (= RCCL over xGMI), one rank per GPU.
    backend = os.environ.get("CLIVE2_BENCH_BACKEND", "nccl")
    if os.environ.get("CLIVE2_BENCH_SHARE_GPU") == "1":
        local_rank = 0
    # a launcher may expose one GPU per rank (HIP_VISIBLE_DEVICES=<rank>): then the rank's GPU is device 0
    local_rank %= max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    import numpy as np
    import clive2_amd as c2
    from clive2_amd.renderer import Renderer, make_seeds

    W, H = args.width, args.height
    scene, scene_desc = build_scene(args.scene, W, H)
    r = Renderer(scene, seeds=make_seeds(W * H, rank=rank), device=local_rank)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    r.set_levels_per_launch(args.levels_per_launch)
    r.set_traversal_mode(args.traversal_mode)
    r.set_pipelining(args.pipelining)

    # untimed: warm-up (also the counting pass that measures N_node / N_tri per ray)
    r.set_counting(True)
    r.run_samples(max(args.warmup, 1))
    cw = r.counters()
    n_node = cw["box_tests"] / max(cw["counted_rays"], 1)
    n_tri = cw["tri_tests"] / max(cw["counted_rays"], 1)
    r.set_counting(False)
    r.reduce_accumulators()          # untimed: creates the RCCL communicator and its buffers before the clock starts
    r.reset_counters()
    r.reset_accumulators()
    r.set_profiling(1)               # timed region: HIP events around the connection-ray traversal launch only
    if args.debug_flags:
        r.set_debug_flags(args.debug_flags)

    barrier()
    t0 = time.perf_counter()
    r.run_samples(args.steps)
    r.reduce_accumulators()
    barrier()
    dt = time.perf_counter() - t0

    c = r.counters()
    rays_local = c["rays"]
    # untimed: per-stage breakdown (HIP events around every launch) over a few more samples, in serial
    # order on one stream -- with the sample pipeline on, spans of the two streams overlap and a
    # stage's span includes whatever ran beside it
    n_break = min(args.steps, 8)
    r.reset_counters()
    r.set_profiling(2)
    r.set_pipelining(False)
    r.run_samples(n_break)
    cb = r.counters()
    r.set_profiling(0)
    r.set_pipelining(args.pipelining)
    if world > 1:
        t = torch.tensor([float(rays_local), dt], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        tmax = t.clone()
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        rays_total, dt = t[0].item(), tmax[1].item()
    else:
        rays_total = float(rays_local)

    if rank == 0:
        img, wts, cnt, _ = r.read_accumulators()
        assert np.isfinite(img).all() and (cnt >= args.steps * world).all(), "accumulators corrupt"
        b_ray = 48.0 + 32.0 * n_node + 36.0 * n_tri
        stages = {k[3:]: round(cb[k] / n_break, 4) for k in cb if k.startswith("ms_")}
        # the roofline kernel: the connection-ray traversal launch (70 % of all rays, one launch per
        # sample), timed with HIP events on the renderer's stream inside the timed region
        k_ms, k_rays, k_launches = c["ms_traverse_conn"], c["rays_traverse_conn"], c["launches_traverse_conn"]
        k_name = "k_traverse_conn" if args.traversal_mode == 1 or (args.traversal_mode == 0 and args.scene == "cornell") \
            else "k_traverse_persistent<ConnRaySource>"
        achieved = (k_rays * b_ray) / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0
        traffic, traffic_src = measured_traffic(k_name) if args.scene == "cornell" and (W, H) == (1920, 1080) else (None, None)
        # what actually bounds the kernel on this scene (the tree is LDS-resident): VALU issue.  Wave-instructions
        # per launch from the committed SQ counters x the measured issue cost (tools/valu_rate.hip: 2.3 cycles
        # per fp32 wave-instruction; divides dearer, so this is a floor) over 1024 SIMDs at the 2.4 GHz clock,
        # against the duration of the launch running alone
        valu = None
        n_valu = measured_valu(k_name) if args.scene == "cornell" and (W, H) == (1920, 1080) else None
        if n_valu and cb["ms_traverse_conn"] > 0:
            issue_ms = n_valu * 2.3 / (1024 * 2.4e9) * 1e3
            alone_ms = cb["ms_traverse_conn"] / max(cb["launches_traverse_conn"], 1)
            valu = {"kernel": k_name, "wave_insts_per_launch": n_valu, "issue_floor_ms": round(issue_ms, 4),
                    "launch_alone_ms": round(alone_ms, 4), "frac": round(issue_ms / alone_ms, 4),
                    "source": "profiles/r01_final_pmc_sq.txt (SQ_INSTS_VALU) x 2.3 cycles (tools/valu_rate.hip)"}
        out = {
            "metric": "Mrays/sec (whole node) + HBM GB/s, 1080p Cornell box, 1/2/4/8 MI355X",
            "value": round(rays_total / dt / 1e6, 2),
            "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{scene_desc} {W}x{H}, BDPT{' diffuse-only' if args.scene == 'cornell' else ''}, "
                                   f"{args.steps} spp per GPU", "width": W, "height": H,
                       "rays_per_pixel_sample": round(rays_local / (args.steps * W * H), 3),
                       "parallelism": f"sample-split x{world}, one {'RCCL' if backend == 'nccl' else backend} all-reduce of the accumulators"},
            "hbm_gbs": {"kernel": k_name, "algorithmic": round(achieved, 1),
                        "measured_pmc": round(traffic / (k_ms / max(k_launches, 1) * 1e-3) / 1e9, 1) if traffic else None},
            "roofline": {"bound": "hbm", "kernel": k_name, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_src,
                         "bytes_per_ray": round(b_ray, 1), "n_node": round(n_node, 3), "n_tri": round(n_tri, 3),
                         "rays_per_launch": round(k_rays / max(k_launches, 1)),
                         "avg_launch_ms": round(k_ms / max(k_launches, 1), 4),
                         # the same launch with nothing beside it (serial breakdown pass below)
                         "avg_launch_ms_serial": round(cb["ms_traverse_conn"] / max(cb["launches_traverse_conn"], 1), 4),
                         "frac_serial": round((cb["rays_traverse_conn"] * b_ray) / max(cb["ms_traverse_conn"] * 1e-3, 1e-12) / 1e9 / HBM_PEAK_GBS, 4),
                         "sample_pipeline_stages": args.pipelining},
            "stage_ms_per_step_serial": stages,
            "valu": valu,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.cpu_width, args.cpu_height, args.cpu_samples)
        print(json.dumps(out), flush=True)

    r.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
