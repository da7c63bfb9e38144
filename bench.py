#!/usr/bin/env python3
"""bench.py -- BASELINE.json metric: Mrays/s (whole job) on the 1080p Cornell box, N x MI355X.

One "step" = one sample of the full BDPT pipeline (Renderer.run_sample: light + camera subpaths,
all (t,s) connections, light splat, filter, accumulate) over the whole 1920x1080 frame on every
rank.  Ranks are weak-scaled (each integrates its own samples of the replicated scene, own seed
buffer); the accumulators are summed once with an in-place RCCL all-reduce (inside the library,
cl2_reduce_accumulators) inside the timed region.  One "ray" = one closest-hit BVH query
(SURVEY.md §8d); rays are counted on the device.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--width 1920 --height 1080] [--total-spp S]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

`python bench.py --gpus N` with N > 1 and no launcher environment starts ITSELF as N rank processes (one per
GPU: RANK / LOCAL_RANK / WORLD_SIZE and one shared CLIVE2_RENDEZVOUS_FILE in their environment) before anything
touches the GPU -- the parent never loads the library -- waits for them, relays rank 0's JSON line and exits
non-zero if any rank did.  Under an external launcher only RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* are read
from its environment.  Default is weak scaling (every rank renders K samples of the frame); `--total-spp S`
splits a FIXED S samples over the ranks (strong scaling: configs 4 / 5 of BASELINE.json).  torch is
not imported (the library runs on the ROCm runtime it was built for; barrier, max-over-ranks clock
and ray tally go through cl2_comm_allreduce_f64).  Rank 0 prints ONE JSON line.

`roofline` describes the dominant kernel (the connection-ray traversal launch, 70 % of all rays):
its duration is measured live with HIP events on the stream it is launched on.  On the Cornell box
the whole tree lives in LDS, so the bound is VALU issue (wave-instructions x 2 cycles, SIMD-32
wave64, MI355X_MICROARCH.md) -- the algorithmic-bytes formula of SURVEY.md §8(d),
B_ray = 48 + 32*N_node + 36*N_tri, is reported beside it but prices bytes that never leave LDS.
Instruction counts and HBM bytes per launch come from committed rocprofv3 PMC passes and are
marked "static" (used only while the kernel sources still hash to what was profiled).
The mesh legs -- `roofline_mesh` (config 3), `roofline_blob` (config 4), `roofline_hbm` (config 5 at 1080p) and
`roofline_hbm_4k` (config 5 at its own 3840 x 2160) -- read their tree through the caches, and say WHICH resource binds the
launch: every one of them carries `fractions` = {valu_issue, l1_lookups, l2, beyond_l2, fabric} (mesh_roofline below), `bound` names the
largest and `frac` is that one (<= 1).  Round 6: stdout carries ONE line of at most 4,000 characters (compact_line: the contract's keys,
`roofline`, `cpu_baseline`, one compact `legs` object); the full result with every fraction goes to bench_detail.json beside this script.  SURVEY 8(d)'s formula figure (the reference's binary walk, 32 B per node test + 36 B
per triangle test) is kept beside them as `contract_algorithmic_gbs` with `served_from`: it prices bytes the 4-wide walk does
not read, against a level they are not served from, and is not a fraction of anything.  The stage-share tuner of the mesh
scenes (cl2_tune) runs in the warm-up, never inside the clock.  `serial_run_sample_ms` times the reference's own loop --
`run_sample()` then a read of the tone-mapped picture, every iteration (src/render.py:31-37).  `cpu_baseline` times the C
oracle (CPU restatement of the reference's kernels, OpenMP over host cores) on a bounded sample.
"""
import argparse
import glob
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # RCCL across processes needs dmabuf IPC on this pool

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
L2_PEAK_GBS = 34500.0      # aggregate L2 bandwidth (8 XCDs x 4 MiB), same guide, "L2 (per XCD)"
L2_BYTES = 32 << 20
IC_GATHER_PEAK_GBS = 8600.0   # scattered lines out of the Infinity Cache (38 MB table, uniformly random rows), same guide, "Indexed rows"
IC_BYTES = 256 << 20
HBM_ACHIEVABLE_GBS = 6300.0   # what a streaming read gets of the 8 TB/s, same guide, "HBM"
LINE_BYTES = 128              # one L2 request = one line (tools/l1_gather_rate.hip: a scattered 16-byte load that misses L1 costs a whole line of L2 bandwidth)
# Vector L1 (TCP) look-up rate: one cache-line look-up per clock and CU; a lane's 16-byte load of its own line is one look-up (lanes of
# one instruction that share a line are served together).  tools/l1_gather_rate.hip measures 0.92 per clock and CU for 64 scattered lanes.
N_CU, CLOCK_GHZ = 256, 2.4
L1_LOOKUP_PEAK_G = N_CU * CLOCK_GHZ
L1_LOOKUP_MEASURED_PER_CLK = 0.92
# VALU issue peak: 256 CUs x 4 SIMD-32, a wave64 instruction occupies its SIMD for 2 cycles at 2.4 GHz
VALU_PEAK_GINST = 1024 * 2.4e9 / 2.0 / 1e9


LINE_CAP = 4000            # characters of the ONE stdout line (the driver keeps 8,000 characters of stdout: BENCH_r05 lost its line to that)
DETAIL_FILE = "bench_detail.json"


def _pick(d, keys):
    return {k: d[k] for k in keys if isinstance(d, dict) and k in d and d[k] is not None}


def compact_leg(leg):
    """One mesh workload on the stdout line: numbers only, <= 200 characters (the full leg goes to bench_detail.json)."""
    if not isinstance(leg, dict):
        return None
    if "error" in leg and "mrays_per_s" not in leg:
        return {"error": str(leg["error"])[:80]}
    one = leg.get("one_stream") or {}
    ser = leg.get("serial_run_sample_ms") or {}
    sp = leg.get("subpath_walk") or {}
    c = {"mrays_per_s": leg.get("mrays_per_s"), "ms_per_step": leg.get("ms_per_step"), "streams": leg.get("sample_streams"),
         "bound": leg.get("bound"), "frac": leg.get("frac"), "frac_alone": leg.get("frac_launch_alone"),
         "sub_frac_alone": sp.get("frac_launch_alone"), "pass_fabric": leg.get("pass_fabric_frac"),
         "k1_mrays_per_s": one.get("mrays_per_s"), "serial_ms": ser.get("ms_per_iteration")}
    return {k: v for k, v in c.items() if v is not None}


def compact_line(out):
    """The ONE stdout line of the bench contract from the full result: the contract's keys, `roofline` (numbers only) and
    `cpu_baseline` of the headline workload, and one compact `legs` object for the other workloads.  Everything else -- the four
    fractions of every leg, own bytes, the contract figure, the subpath walk, prose -- is in bench_detail.json / on stderr.
    Never longer than LINE_CAP: `legs`, then the optional objects, are dropped before that could happen."""
    line = _pick(out, ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling"))
    line["vs_baseline"] = out.get("vs_baseline")
    line.update(_pick(out, ("dtype", "data")))
    line["config"] = _pick(out.get("config", {}), ("workload", "width", "height", "rays_per_pixel_sample", "samples_rendered_all_ranks",
                                                     "sample_streams", "parallelism"))
    roof = out.get("roofline") or {}
    r = _pick(roof, ("bound", "achieved", "peak", "unit", "frac", "frac_launch_alone", "lane_weighted_frac", "kernel", "rays_per_launch",
                     "avg_launch_ms", "avg_launch_ms_alone", "wave_insts_per_launch", "active_lanes_per_valu_inst"))
    r["traffic"] = roof.get("traffic")
    if isinstance(roof.get("hbm"), dict):
        r["hbm"] = _pick(roof["hbm"], ("algorithmic_gbs", "bytes_per_ray", "n_node", "n_tri", "measured_gbs", "measured_frac_of_peak"))
    if isinstance(roof.get("static"), dict):
        r["static"] = _pick(roof["static"], ("source", "sources_sha"))
    line["roofline"] = r
    if isinstance(out.get("stage_ms_per_step_serial"), dict):
        line["stage_ms"] = {k: round(v, 3) for k, v in out["stage_ms_per_step_serial"].items() if v}
    if isinstance(out.get("serial_run_sample_ms"), dict) and "ms_per_iteration" in out["serial_run_sample_ms"]:
        line["serial_ms"] = out["serial_run_sample_ms"]["ms_per_iteration"]
    names = (("roofline_mesh", "c3_glass_5k"), ("roofline_blob", "c4_blob_82k"), ("roofline_hbm", "c5_interior_1m"),
             ("roofline_hbm_4k", "c5_interior_1m_4k"))
    legs = {short: compact_leg(out[key]) for key, short in names if key in out}
    if legs:
        line["legs"] = legs
    if isinstance(out.get("comm"), dict):
        line["comm"] = _pick(out["comm"], ("nranks", "distinct_devices", "launcher_world_size", "allreduce_ms", "allreduce_bytes", "rank0_device"))
        if isinstance(out["comm"].get("devices"), list):
            line["comm"]["devices"] = [str(d)[:16] for d in out["comm"]["devices"][:16]]      # one PCI address per rank (<= 16 ranks on the line)
    if isinstance(out.get("strong_scaling"), dict):
        ss = out["strong_scaling"]
        line["strong_scaling"] = dict(_pick(ss, ("scaling", "value", "unit", "seconds", "samples_rendered_all_ranks", "ms_per_sample_whole_job")),
                                      **({"error": str(ss["error"])[:80]} if "error" in ss else {}))
    if isinstance(out.get("cpu_baseline"), dict):
        cb = _pick(out["cpu_baseline"], ("value", "unit", "cores", "kind", "runs"))
        cb["sample"] = str(out["cpu_baseline"].get("sample", ""))[:160]
        line["cpu_baseline"] = cb
    line["detail"] = DETAIL_FILE
    for obj in (line, line["config"], line["roofline"]):
        for k, v in obj.items():
            if isinstance(v, str) and len(v) > 300:
                obj[k] = v[:300]
    # the cap is a promise, not a hope: shed the extras (never the contract's keys, `roofline`, `cpu_baseline`) until it holds
    for victim in ("strong_scaling", "comm", "stage_ms", "legs"):
        if len(json.dumps(line)) <= LINE_CAP:
            break
        if victim == "legs" and "legs" in line:
            line["legs"] = {k: _pick(v, ("mrays_per_s", "ms_per_step", "frac")) for k, v in line["legs"].items() if v}
            if len(json.dumps(line)) <= LINE_CAP:
                break
        line.pop(victim, None)
    if len(json.dumps(line)) > LINE_CAP:
        line["roofline"].pop("static", None)
        for obj in (line, line["config"], line["roofline"], line.get("cpu_baseline", {})):
            for k, v in obj.items():
                if isinstance(v, str) and len(v) > 100:
                    obj[k] = v[:100]
    text = json.dumps(line)
    assert len(text) <= LINE_CAP and "\n" not in text, len(text)
    return text


def emit(out, stdout_fd=None):
    """Full result -> bench_detail.json beside this script (and under gpurun_out/ when that exists, which travels back from a GPU
    box); the compact line -> stdout.  stderr gets one short note, not the 20 KB: a driver that keeps a bounded tail of both
    streams must still find the line (CLIVE2_BENCH_DETAIL_STDERR=1 prints the full result there too)."""
    full = json.dumps(out)
    for path in (os.path.join(ROOT, DETAIL_FILE), os.path.join(ROOT, "gpurun_out", DETAIL_FILE)):
        try:
            if os.path.isdir(os.path.dirname(path)):
                with open(path, "w") as f:
                    f.write(full + "\n")
        except OSError as exc:
            print(f"bench.py: could not write {path}: {exc}", file=sys.stderr)
    if os.environ.get("CLIVE2_BENCH_DETAIL_STDERR") == "1":
        print("bench.py detail: " + full, file=sys.stderr, flush=True)
    else:
        print(f"bench.py: full result ({len(full)} bytes) in {DETAIL_FILE}", file=sys.stderr, flush=True)
    text = compact_line(out)
    sys.stdout.flush()
    if stdout_fd is not None:
        os.dup2(stdout_fd, 1)
    print(text, flush=True)


def cpu_baseline(width, height, samples, repeats=2):
    """Oracle (kind 'port'): the CPU restatement of trace.metal's kernels + renderer.py's host glue,
    timed end to end on this host's cores.  Checker code used as the reported baseline only.  The OpenMP team is
    sized to the box's CPU share (16 on a one-GPU box) and PINNED (one thread per core, spread): unpinned, the same
    binary measured 26 and 43 Mrays/s on two boxes.  Best of `repeats` runs; every run is reported."""
    try:
        n_aff = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        n_aff = os.cpu_count() or 1
    threads = max(1, min(n_aff, 16))
    os.environ.setdefault("OMP_NUM_THREADS", str(threads))
    os.environ.setdefault("OMP_WAIT_POLICY", "passive")
    os.environ.setdefault("OMP_PROC_BIND", "spread")
    os.environ.setdefault("OMP_PLACES", "cores")
    import clive2_amd as c2
    from oracle import oracle as orc
    orc.build()
    scene = c2.create_scene_from_preset("empty", width, height)
    runs, rays = [], 0
    for _ in range(max(1, repeats)):
        o = orc.OracleRenderer(scene, seeds=orc.make_seeds(width * height))
        t0 = time.perf_counter()
        for _ in range(samples):
            o.run_sample()
        dt = time.perf_counter() - t0
        rays = o.rays_traced
        runs.append(round(rays / dt / 1e6, 3))
    threads = int(os.environ["OMP_NUM_THREADS"])
    return {"value": max(runs), "unit": "Mrays/s", "cores": threads, "kind": "port", "runs": runs,
            "threads_pinned": f"OMP_PROC_BIND={os.environ['OMP_PROC_BIND']} OMP_PLACES={os.environ['OMP_PLACES']}, {n_aff} CPUs in the affinity mask",
            "sample": f"Cornell box {width}x{height}, {samples} sample(s) of the full BDPT pipeline per run "
                      f"({rays} rays; C oracle, OpenMP, best of {len(runs)} runs)"}


def kernel_sources_sha():
    """Hash of everything that is compiled into the library: a committed PMC summary is only quoted
    while the kernels it profiled are the kernels that run."""
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "clive2_amd", "csrc", "*"))):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def static_pmc(scene, W, H, streams=1):
    """Per-launch PMC figures of the connection-ray traversal kernel from profiles/r<NN>_pmc_<scene>.json
    (tools/profile_round.sh + tools/profile_summaries.py: rocprofv3 --pmc passes of this same command), or
    None when there is no summary for this scene / frame size, or the kernel sources have changed since
    it was taken."""
    sha = kernel_sources_sha()
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_pmc_{scene}*.json")), reverse=True):
        try:
            d = json.load(open(path))
        except (OSError, ValueError):
            continue
        if d.get("scene") != scene or d.get("sources_sha") != sha or (d.get("width"), d.get("height")) != (W, H) or d.get("sample_streams", 1) != streams:
            continue
        # the timed launch, not the tallying variant of the warm-up's counting pass (k_traverse_wide<REPS, Source, TALLY, SPEC>)
        wide = [k for k in d.get("kernels", {}) if k.startswith("k_traverse_wide<") and ",ConnRaySource,false" in k]
        name = wide[0] if wide else d.get("conn_traversal_kernel")
        row = d.get("kernels", {}).get(name)
        if row:
            sub = [k for k in d["kernels"] if k.startswith("k_traverse_wide<") and ",PathRaySource,false" in k]
            return dict(row, name=name, source=os.path.relpath(path, ROOT), sources_sha=d["sources_sha"],
                        subpath_kernel=dict(d["kernels"][sub[0]], name=sub[0]) if sub else None, all_kernels=d["kernels"])
    return None


def fabric_bytes(row):
    """Bytes one launch moves across the fabric, from its committed PMC row: reads by request size (TCC_EA0_RDREQ_32B / _64B / _128B)
    where that pass was taken, else 2 x FETCH_SIZE; plus WRITE_SIZE."""
    if not row or row.get("WRITE_SIZE") is None:
        return None
    reads = row.get("fabric_read_bytes_by_request_size")
    if reads is None:
        if row.get("FETCH_SIZE") is None:
            return None
        reads = 2 * row["FETCH_SIZE"] * 1024.0
    return reads + row["WRITE_SIZE"] * 1024.0


def pass_fabric(kernels, conn_name, sub_name, sub_launches_per_pass, pass_ms):
    """What a whole pass (K samples: generators, per-level subpath walks and bounces, connection set-up / walk / resolve, K6) moves
    across the fabric, against the 8 TB/s HBM peak: sum over kernels of committed fabric bytes per launch x launches per pass,
    divided by the pass time measured live.  Launches per pass: 1 for the connection walk, the live counter for the per-level
    subpath walks (their merged level-0 launch, `DualPathRaySource`, is one of them), and for every other kernel its launch count in
    the PMC run relative to k_connect_setup's (one per pass).  Tallying / counting variants of the walks run in the warm-up only."""
    if not kernels or pass_ms <= 0:
        return None
    setup = kernels.get("k_connect_setup")
    n_passes = (setup or {}).get("launches_fetch")
    flags = conn_name.split("ConnRaySource", 1)[1] if conn_name and "ConnRaySource" in conn_name else None      # ",false,true,true>"
    dual = [k for k in kernels if flags and k.startswith("k_traverse_wide<") and k.endswith("DualPathRaySource" + flags)]
    total, parts = 0.0, {}
    for name, row in kernels.items():
        b = fabric_bytes(row)
        if b is None:
            continue
        if name == conn_name:
            n = 1.0
        elif name == sub_name:
            n = max(sub_launches_per_pass - (1.0 if dual else 0.0), 0.0)
        elif name in dual:
            n = 1.0
        elif name.startswith(("k_traverse_wide<", "k_traverse_persistent<", "k_subpaths_persistent<", "k_traverse_paths", "k_traverse_conn")):
            continue
        elif n_passes:
            n = row.get("launches_fetch", 0) / n_passes
        else:
            continue
        total += b * n
        parts[name] = round(b * n / 1e9, 2)
    gbs = total / (pass_ms * 1e-3) / 1e9
    return {"bytes_per_pass": round(total), "gbs": round(gbs, 1), "frac": round(gbs / HBM_PEAK_GBS, 4), "peak_gbs": HBM_PEAK_GBS,
            "pass_ms": round(pass_ms, 3), "gb_per_pass_by_kernel": dict(sorted(parts.items(), key=lambda kv: -kv[1])[:8])}


def mesh_roofline(row, launch_ms, own_bytes, tree_bytes):
    """What binds a traversal launch whose tree is read through the caches: five fractions (round 6: the L1), each
    `achieved / peak` of one resource, from the committed PMC summary `row` of that kernel (per launch, hash-guarded) and the
    launch time measured live; `bound` = the largest, `frac` = that one.
      valu_issue   SQ_INSTS_VALU wave-instructions / t against 1024 SIMDs x 2.4 GHz / 2 cycles (active lanes beside it)
      l1_lookups   TCP_TOTAL_CACHE_ACCESSES (cache-line look-ups of the vector L1s: one per 16-byte load of a lane that no other lane of
                   the instruction shares a line with) / t against one look-up per clock and CU (256 x 2.4 GHz; measured for scattered
                   loads: 0.92, tools/l1_gather_rate.hip).  Round 6's finding: every mesh walk sits at 0.85-0.89 of it, whatever its
                   tree size -- a lane reads its 112-byte node with seven loads and a triangle pair with five or six
      l2           what L1 asks of L2: (TCC_HIT + TCC_MISS) requests x 128 B / t against the L2's 34.5 TB/s
      beyond_l2    L2 misses x 128 B / t against the level below: scattered lines out of the Infinity Cache (8.6 TB/s) while
                   the tree fits its 256 MiB, else HBM (6.3 TB/s achievable of 8)
      fabric       FETCH_SIZE + WRITE_SIZE as reported (raw) and with the guide's x2 on FETCH_SIZE (calibrated on coalesced
                   16 B / lane streams; these are scattered 112-byte node and 48-byte triangle reads: both shown) against the
                   8 TB/s HBM peak; the fraction that decides is the x2 one (the larger)
    `own_bytes`: what the walk ITSELF asks of L1 per launch (device tallies: 112 B per wide node + 36 B per triangle record +
    48 B per ray), reported with the share of it that L1 passes on."""
    if not row or launch_ms <= 0:
        return None
    t = launch_ms * 1e-3
    fr = {}
    n_valu = row.get("SQ_INSTS_VALU")
    if n_valu:
        fr["valu_issue"] = {"achieved": round(n_valu / t / 1e9, 1), "peak": round(VALU_PEAK_GINST, 1), "unit": "G wave-instructions/s",
                            "frac": round(n_valu / t / 1e9 / VALU_PEAK_GINST, 4), "active_lanes_per_inst": row.get("thread_cycles_per_valu_inst"),
                            "wave_insts_per_launch": n_valu, "wait_share_of_wave_cycles": row.get("wait_share")}
    n_l1 = row.get("TCP_TOTAL_CACHE_ACCESSES_sum")
    if n_l1:
        per_clk = n_l1 / t / 1e9 / L1_LOOKUP_PEAK_G
        fr["l1_lookups"] = {"achieved": round(n_l1 / t / 1e9, 1), "peak": round(L1_LOOKUP_PEAK_G, 1), "unit": "G look-ups/s", "frac": round(per_clk, 4),
                            "per_clock_per_cu": round(per_clk, 4), "measured_peak_per_clock_per_cu": L1_LOOKUP_MEASURED_PER_CLK,
                            "frac_of_measured_peak": round(per_clk / L1_LOOKUP_MEASURED_PER_CLK, 4), "lookups_per_launch": n_l1}
    hit, miss = row.get("TCC_HIT_sum"), row.get("TCC_MISS_sum")
    if hit is not None and miss is not None:
        l2_bytes, below_bytes = (hit + miss) * LINE_BYTES, miss * LINE_BYTES
        fr["l2"] = {"achieved": round(l2_bytes / t / 1e9, 1), "peak": L2_PEAK_GBS, "unit": "GB/s", "frac": round(l2_bytes / t / 1e9 / L2_PEAK_GBS, 4),
                    "requests_per_launch": hit + miss, "hit_rate": row.get("l2_hit_rate")}
        in_ic = tree_bytes <= IC_BYTES
        peak = IC_GATHER_PEAK_GBS if in_ic else HBM_ACHIEVABLE_GBS
        fr["beyond_l2"] = {"achieved": round(below_bytes / t / 1e9, 1), "peak": peak, "unit": "GB/s", "frac": round(below_bytes / t / 1e9 / peak, 4),
                           "served_from": "Infinity Cache (scattered-line rate)" if in_ic else "HBM (achievable streaming rate)", "misses_per_launch": miss}
    fetch, write = row.get("FETCH_SIZE"), row.get("WRITE_SIZE")
    if fetch is not None and write is not None:
        raw, x2 = (fetch + write) * 1024.0, (2 * fetch + write) * 1024.0
        # where the request-size mix was profiled (TCC_EA0_RDREQ_{32B,64B,128B}) the read bytes are known exactly and decide;
        # else the x2 figure does (the larger of the two readings of FETCH_SIZE)
        sized = row.get("fabric_read_bytes_by_request_size")
        best = (sized + write * 1024.0) if sized else x2
        fr["fabric"] = {"achieved": round(best / t / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(best / t / 1e9 / HBM_PEAK_GBS, 4),
                        "decided_by": "request-size mix (TCC_EA0_RDREQ_32B / _64B / _128B) + WRITE_SIZE" if sized else "2 x FETCH_SIZE + WRITE_SIZE",
                        "raw_gbs": round(raw / t / 1e9, 1), "raw_frac": round(raw / t / 1e9 / HBM_PEAK_GBS, 4),
                        "x2_gbs": round(x2 / t / 1e9, 1), "x2_frac": round(x2 / t / 1e9 / HBM_PEAK_GBS, 4),
                        "bytes_per_launch_raw": round(raw), "bytes_per_launch_x2": round(x2),
                        "bytes_per_launch_by_request_size": round(best) if sized else None}
    if not fr:
        return None
    bound = max(fr, key=lambda k: fr[k]["frac"])
    out = {"bound": {"valu_issue": "valu", "l1_lookups": "l1", "l2": "l2", "beyond_l2": "infinity_cache" if tree_bytes <= IC_BYTES else "hbm", "fabric": "hbm"}[bound],
           "bound_fraction": bound, "achieved": fr[bound]["achieved"], "peak": fr[bound]["peak"], "unit": fr[bound]["unit"],
           "frac": fr[bound]["frac"], "fractions": fr}
    if own_bytes:
        out["own_bytes"] = dict(own_bytes, gbs=round(own_bytes["bytes_per_launch"] / t / 1e9, 1))
        if "l2" in fr:
            # (above 1 where lines are used in part: a 48-byte triangle record that straddles two lines costs two requests)
            out["own_bytes"]["l2_line_bytes_per_own_byte"] = round(fr["l2"]["requests_per_launch"] * LINE_BYTES / max(own_bytes["bytes_per_launch"], 1), 3)
    return out


_SCENES = {}


def build_scene(name, W, H, builder=None, max_members=None):
    """(scene, description) of a bench workload; built once per process (the K-stream and the one-stream leg share it).  `max_members`: the
    builder's leaf size (None = the reference's 8; anything else is a labelled experiment, never a bench leg)."""
    key = (name, W, H, builder) if max_members is None else (name, W, H, builder, max_members)
    if key not in _SCENES:
        _SCENES[key] = _build_scene(name, W, H, builder, max_members)
    return _SCENES[key]


def _build_scene(name, W, H, builder=None, max_members=None):
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
    s = c2.create_scene(W, H, np.array([0, 1.5, 6]), np.array([0, 0, -1]), file_specs=specs, materials=mats, room=room,
                        bvh_builder=builder or os.environ.get("CLIVE2_BENCH_BVH_BUILDER", "auto"), max_members=max_members)
    return s, desc + f", {len(s.triangles)} tris / {len(s.boxes)} boxes" + (f" (max_members {max_members})" if max_members is not None else "")


def run_workload(args, scene_name, W, H, steps, warmup, rank, local_rank, world, with_comm, streams=1, order=0):
    """Warm-up (tuner + counting pass), the timed region, a serial per-stage breakdown.  `steps` = samples THIS
    rank renders inside the clock; with `streams` = K sample streams (cl2_set_sample_streams: K independent samples of the
    frame per pass, one seed buffer each) that is ceil(steps / K) passes.  Returns the pieces of the JSON line that depend
    on the workload."""
    import clive2_amd as c2  # noqa: F401
    from clive2_amd import _native
    from clive2_amd.renderer import Renderer, stream_seeds
    from clive2_amd.distributed import join_communicator

    scene, scene_desc = build_scene(scene_name, W, H)
    n_dev = max(_native.lib().cl2_device_count(), 1)
    passes = max(1, -(-steps // streams))
    steps = passes * streams                     # samples this rank renders inside the clock
    # a launcher may expose one GPU per rank (HIP_VISIBLE_DEVICES=<rank>): then the rank's GPU is device 0.
    # Seed buffers: stream k of rank r is buffer r * K + k of the job -- no two streams of the job share one
    r = Renderer(scene, seeds=stream_seeds(W * H, streams, first_rank=rank * streams), device=local_rank % n_dev, streams=streams)
    comm = None
    if with_comm:
        join_communicator(r, rank, world)        # untimed: communicator and its buffers exist before the clock starts
        # what the COMMUNICATOR says (not the launcher's environment): rank count, and the PCI address of every rank's GPU,
        # gathered as a sum of one-hot vectors through the library's own small all-reduce
        info = r.comm_info()
        addr = [0.0] * min(max(info["nranks"], 1), 16)
        if info["rank"] < len(addr):
            addr[info["rank"]] = float(info["pci_address"])
        addr = r.allreduce_host(addr, op="sum")
        comm = {"nranks": info["nranks"], "rank0_device": info["pci_bus_id"],
                "devices": ["%04x:%02x:%02x.%x" % (int(a) >> 16, (int(a) >> 8) & 0xFF, (int(a) >> 3) & 0x1F, int(a) & 7) for a in addr],
                "distinct_devices": len({int(a) for a in addr}), "launcher_world_size": world}

    def barrier():
        r.synchronize()
        if with_comm:
            r.allreduce_host([0.0], op="max")
        r.synchronize()

    r.set_levels_per_launch(args.levels_per_launch)
    if order:
        r.set_traversal_order(order)             # opt-in nearest-first child order: labelled extras of the detail file only, never a leg's figure
    r.set_traversal_mode(args.traversal_mode)
    r.set_pipelining(args.pipelining)
    if args.debug_flags:
        r.set_debug_flags(args.debug_flags)      # launch-organisation switches only: the library refuses the invalid-render bits

    # untimed: the measured launch-organisation choices (bounces per launch of a small scene, stage shares of a large one:
    # cl2_tune, 1 / 54 real samples), then the counting pass that measures N_node / N_tri per ray
    # (CLIVE2_BENCH_SKIP_TUNE=1: the rocprofv3 counter passes of tools/profile_round.sh -- per-launch counters do not depend on how the
    # stages share the machine, and 54 tuner samples under a serialising profiler take minutes on the 1M-triangle scene)
    tuned = 0 if os.environ.get("CLIVE2_BENCH_SKIP_TUNE") == "1" else r.tune()
    n_count = max(1, -(-max(warmup, 1) // streams))
    r.set_counting(True)
    r.run_samples(n_count)
    cw = r.counters()
    n_node = cw["box_tests"] / max(cw["counted_rays"], 1)
    n_tri = cw["tri_tests"] / max(cw["counted_rays"], 1)
    # what the walk that RUNS fetches (the exact 4-wide walk of the scenes whose tree is read through the caches; counting mode 2)
    own = None
    if not r.organisation()["tree_in_lds"] and r.organisation()["wide_nodes"] > 0:
        r.reset_counters()
        r.set_counting(2)
        r.run_samples(n_count)
        own = r.walk_tallies()
    r.set_counting(False)
    if with_comm:
        r.reduce_accumulators()      # untimed: first use of the collective (channel set-up) before the clock starts
    r.reset_counters()
    r.reset_accumulators()
    r.set_profiling(1)               # timed region: HIP events around the connection-ray traversal launch only
    org_before = r.organisation()

    barrier()
    t0 = time.perf_counter()
    r.run_samples(passes)
    t_reduce = time.perf_counter()
    if with_comm:
        r.reduce_accumulators()              # returns when the collective is complete on this rank
    t_reduce = time.perf_counter() - t_reduce
    barrier()
    dt = time.perf_counter() - t0

    c = r.counters()
    r_org = r.organisation()
    assert r_org["paths_share"] == org_before["paths_share"], "a tuning experiment ran inside the timed region"
    rays_local = c["rays"]
    # untimed: per-stage breakdown (HIP events around every launch) over a few more samples, in serial
    # order on one stream -- with the sample pipeline on, spans of the two streams overlap and a
    # stage's span includes whatever ran beside it
    n_break_passes = max(1, min(passes, max(1, 8 // streams)))
    n_break = n_break_passes * streams
    r.reset_counters()
    r.set_profiling(2)
    r.set_pipelining(0)
    r.run_samples(n_break_passes)
    cb = r.counters()
    r.set_profiling(0)
    r.set_pipelining(args.pipelining)
    if with_comm:
        rays_total, steps_total = r.allreduce_host([float(rays_local), float(steps)], op="sum")
        dt, t_reduce_max = r.allreduce_host([dt, t_reduce], op="max")
        comm["allreduce_ms"] = round(t_reduce_max * 1e3, 3)          # slowest rank's wait for the 8*W*H-float all-reduce (includes waiting for the last rank to arrive)
        comm["allreduce_bytes"] = 8 * W * H * 4
    else:
        rays_total, steps_total = float(rays_local), float(steps)

    out = None
    if rank == 0:
        import numpy as np
        img, wts, cnt, _ = r.read_accumulators()
        # after the reduce every rank holds the sums of all ranks (and the breakdown pass added n_break samples)
        assert np.isfinite(img).all() and (cnt >= int(steps_total)).all(), "accumulators corrupt"
        # cl2_upload_scene stages the whole tree in LDS up to 512 records and 512 triangles
        in_lds = bool(r_org["tree_in_lds"])
        org = r_org
        persistent = bool(org["persistent_connections"])
        k_name = ("k_traverse_wide<ConnRaySource>" if org["wide_connections"] else "k_traverse_persistent<ConnRaySource>") if persistent \
            else "k_traverse_conn"
        b_ray = 48.0 + 32.0 * n_node + 36.0 * n_tri
        k_ms, k_rays, k_launches = c["ms_traverse_conn"], c["rays_traverse_conn"], max(c["launches_traverse_conn"], 1)
        # (a launch of a handle with K sample streams carries the connection rays of K samples)
        avg_ms = k_ms / k_launches
        alone_ms = cb["ms_traverse_conn"] / max(cb["launches_traverse_conn"], 1)
        rays_per_launch = k_rays / k_launches
        alg_gbs = rays_per_launch * b_ray / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        alg_gbs_alone = (cb["rays_traverse_conn"] / max(cb["launches_traverse_conn"], 1)) * b_ray / max(alone_ms * 1e-3, 1e-12) / 1e9
        pmc = static_pmc(scene_name, W, H, streams)
        traffic = pmc.get("hbm_bytes") if pmc else None
        lanes = pmc.get("thread_cycles_per_valu_inst") if pmc else None     # average active lanes per VALU wave-instruction
        hbm = {"algorithmic_gbs": round(alg_gbs, 1), "algorithmic_gbs_launch_alone": round(alg_gbs_alone, 1),
               "bytes_per_ray": round(b_ray, 1), "n_node": round(n_node, 3), "n_tri": round(n_tri, 3),
               "measured_gbs": round(traffic / (avg_ms * 1e-3) / 1e9, 1) if traffic and avg_ms > 0 else None,
               "measured_frac_of_peak": round(traffic / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if traffic and avg_ms > 0 else None,
               "peak_gbs": HBM_PEAK_GBS}
        common = {"kernel": k_name, "rays_per_launch": round(rays_per_launch), "avg_launch_ms": round(avg_ms, 4),
                  "avg_launch_ms_alone": round(alone_ms, 4), "traffic": traffic, "active_lanes_per_valu_inst": lanes,
                  "static": {"what": "PMC counters per launch (rocprofv3 --pmc passes of this command, committed; quoted only while the kernel sources hash to what was profiled)",
                             "source": pmc["source"], "sources_sha": pmc["sources_sha"]} if pmc else None}
        if in_lds:
            # tree and triangles staged in LDS: the launch is bound by VALU issue, not by HBM
            n_valu = pmc.get("SQ_INSTS_VALU") if pmc else None
            ach = n_valu / (avg_ms * 1e-3) / 1e9 if n_valu and avg_ms > 0 else None
            frac = ach / VALU_PEAK_GINST if ach else None
            roof = {"bound": "valu", "achieved": round(ach, 1) if ach else None, "peak": round(VALU_PEAK_GINST, 1),
                    "unit": "G wave-instructions/s", "frac": round(frac, 4) if frac else None,
                    "frac_launch_alone": round(n_valu / (alone_ms * 1e-3) / 1e9 / VALU_PEAK_GINST, 4) if n_valu and alone_ms > 0 else None,
                    # the issue fraction counts instructions, not lanes: weighted by the lanes that were active in them
                    "lane_weighted_frac": round(frac * lanes / 64.0, 4) if frac and lanes else None,
                    "wave_insts_per_launch": n_valu, "cycles_per_wave_inst": 2.0, "hbm": hbm}
        else:
            # The tree is read through the caches.  Which resource binds the launch is decided by five fractions (mesh_roofline):
            # vector issue, the L1 look-up rate, L2 request bandwidth, the level below L2, the fabric.  `own` = what the 4-wide walk itself asks for.
            def own_bytes(tally, rays_launch):
                if not tally or not tally["rays"]:
                    return None
                per_ray = (112.0 * tally["wide_visits"] + 36.0 * tally["tri_records"] + 32.0 * tally["binary_records"]
                           + 16.0 * tally["stack_spills"]) / tally["rays"] + 48.0
                return {"bytes_per_ray": round(per_ray, 1), "bytes_per_launch": round(per_ray * rays_launch),
                        "wide_visits_per_ray": round(tally["wide_visits"] / tally["rays"], 3),
                        "tri_records_per_ray": round(tally["tri_records"] / tally["rays"], 3),
                        "stack_spills_per_ray": round(tally["stack_spills"] / tally["rays"], 4),
                        "binary_records_per_ray": round(tally["binary_records"] / tally["rays"], 4),
                        "formula": "112 B x wide-node visits + 36 B x triangle records (round 6: the walk's own 36-byte records) + 32 B x binary records + 16 B x stack spills (write + read) + 48 B per ray (device tallies, cl2_set_counting(2))"}
            tree = org["tree_bytes"]
            roof = mesh_roofline(pmc, avg_ms, own_bytes(own and own["connection"], rays_per_launch), tree)
            alone = mesh_roofline(pmc, alone_ms, None, tree)
            if roof is None:
                # no PMC summary of these sources yet: nothing can be said about the bound (the contract figure below does not say it either)
                roof = {"bound": None, "achieved": None, "peak": None, "unit": None, "frac": None, "fractions": None,
                        "own_bytes": own_bytes(own and own["connection"], rays_per_launch)}
            else:
                roof["frac_launch_alone"] = alone["frac"] if alone else None
                roof["fractions_launch_alone"] = {k: v["frac"] for k, v in alone["fractions"].items()} if alone else None
            # SURVEY 8(d)'s formula, as the contract defines it: every node and triangle test of the REFERENCE's binary walk at
            # 32 / 36 bytes.  A figure, not a fraction: the walk that runs is the 4-wide one, and a tree below 256 MiB is served
            # by L2 and the Infinity Cache, so these bytes may exceed what HBM could deliver without any work being skipped (the
            # device's ray / node-test / triangle-test tallies equal the oracle's: tests/test_gpu_fullsize.py)
            roof["contract_algorithmic_gbs"] = round(alg_gbs, 1)
            roof["contract"] = {"bytes_per_ray": round(b_ray, 1), "n_node": round(n_node, 3), "n_tri": round(n_tri, 3),
                                "algorithmic_gbs_launch_alone": round(alg_gbs_alone, 1),
                                "served_from": ("L2 (32 MiB): the tree is %.1f MB" % (tree / 1e6)) if tree <= L2_BYTES else
                                               ("Infinity Cache (256 MiB) + L2: the tree is %.0f MB" % (tree / 1e6)) if tree <= IC_BYTES else "HBM",
                                "algorithmic_over_fabric_traffic": round(rays_per_launch * b_ray / traffic, 3) if traffic else None}
            roof["tree_bytes"] = tree
            # the per-level subpath traversal launches (k_traverse_wide<., PathRaySource>): the other big block of a mesh sample
            sp_launches = max(c["launches_traverse_paths"], 1)
            sp_ms, sp_rays = c["ms_traverse_paths"] / sp_launches, c["rays_traverse_paths"] / sp_launches
            sp_row = pmc.get("subpath_kernel") if pmc else None
            sp = mesh_roofline(sp_row, sp_ms, own_bytes(own and own["subpath"], sp_rays), tree) or {"bound": None, "frac": None}
            # the same launches ALONE on the machine (serial breakdown pass): inside the pipeline they share the fabric with the
            # connection launch, and a fraction taken there says how the fabric is shared, not how far the walk is from its roof
            sp_alone_ms = cb["ms_traverse_paths"] / max(cb["launches_traverse_paths"], 1)
            sp_alone = mesh_roofline(sp_row, sp_alone_ms, None, tree)
            sp["frac_launch_alone"] = sp_alone["frac"] if sp_alone else None
            sp["fractions_launch_alone"] = {k: v["frac"] for k, v in sp_alone["fractions"].items()} if sp_alone else None
            sp["avg_launch_ms_alone"] = round(sp_alone_ms, 4)
            sp.update({"kernel": sp_row["name"] if sp_row else "k_traverse_wide<., PathRaySource>", "launches_per_pass": round(sp_launches / max(passes, 1), 2),
                       "avg_launch_ms": round(sp_ms, 4), "rays_per_launch": round(sp_rays),
                       "note": "avg_launch_ms is over ALL subpath traversal launches of the timed region (the merged level-0 launch included); "
                               "the PMC row is the single-kind launch's"})
            roof["subpath_walk"] = sp
            pf = pass_fabric(pmc.get("all_kernels") if pmc else None, pmc["name"] if pmc else None, sp_row["name"] if sp_row else None,
                             sp_launches / max(passes, 1), dt / max(passes, 1) * 1e3)
            roof["pass_fabric"] = pf
            roof["pass_fabric_frac"] = pf["frac"] if pf else None
        roof.update(common)
        roof["sample_streams"] = streams
        out = {"scene_desc": scene_desc, "rays_total": rays_total, "rays_local": rays_local, "dt": dt, "roofline": roof,
               "steps_total": int(steps_total), "steps_rank": steps, "tuner_samples_in_warmup": tuned * streams, "paths_share": org["paths_share"],
               "comm": comm, "sample_streams": streams,
               "stages": {k[3:]: round(cb[k] / n_break, 4) for k in cb if k.startswith("ms_")}}
    r.close()
    return out


def mesh_leg(args, key, name, W, H, n_steps, streams, local_rank):
    """One extra workload of the N = 1 line: the timed run with `streams` sample streams, the same with the reference's single
    seed buffer beside it, the reference's serial loop.  Whatever goes wrong in a leg is reported in its place and never costs
    the job its headline line."""
    t_setup = time.perf_counter()
    leg = {}
    try:
        m = run_workload(args, name, W, H, n_steps, 2, 0, local_rank, 1, with_comm=False, streams=streams)
        leg = dict(m["roofline"], workload=f"{m['scene_desc']} {W}x{H}, {m['steps_rank']} spp in {streams} sample streams",
                   mrays_per_s=round(m["rays_total"] / m["dt"] / 1e6, 2), ms_per_step=round(m["dt"] / m["steps_rank"] * 1e3, 3),
                   tuner_samples_in_warmup=m["tuner_samples_in_warmup"], paths_share=m["paths_share"],
                   stage_ms_per_step_serial=m["stages"])
        if streams != 1:
            m1 = run_workload(args, name, W, H, max(8, n_steps // 2), 2, 0, local_rank, 1, with_comm=False, streams=1)
            leg["one_stream"] = {"sample_streams": 1, "mrays_per_s": round(m1["rays_total"] / m1["dt"] / 1e6, 2),
                                 "ms_per_step": round(m1["dt"] / m1["steps_rank"] * 1e3, 3), "steps": m1["steps_rank"],
                                 "bound": m1["roofline"]["bound"], "frac": m1["roofline"]["frac"],
                                 "frac_launch_alone": m1["roofline"].get("frac_launch_alone"),
                                 "fractions": {k: v["frac"] for k, v in (m1["roofline"].get("fractions") or {}).items()} or None,
                                 "active_lanes_per_valu_inst": m1["roofline"].get("active_lanes_per_valu_inst"),
                                 "avg_launch_ms": m1["roofline"]["avg_launch_ms"], "paths_share": m1["paths_share"]}
        leg["serial_run_sample_ms"] = serial_loop(name, W, H, local_rank)
        if not args.no_order_extra:
            # a labelled extra, never the leg's figure: the same workload with the opt-in nearest-first child order of the 4-wide walks
            # (cl2_set_traversal_order(1): NOT bit-exact by construction -- tests/test_gpu_fullsize.py counts the rays whose hit differs)
            mo = run_workload(args, name, W, H, max(streams, n_steps // 2), 2, 0, local_rank, 1, with_comm=False, streams=streams, order=1)
            leg["nearest_first_extra"] = {"label": "opt-in cl2_set_traversal_order(1): nearest child first, NOT bit-exact (a hit that rounding puts in front of its own leaf box; exact-t ties go the reference's way); not the leg's figure",
                                          "mrays_per_s": round(mo["rays_total"] / mo["dt"] / 1e6, 2), "ms_per_step": round(mo["dt"] / mo["steps_rank"] * 1e3, 3),
                                          "sample_streams": streams, "steps": mo["steps_rank"]}
    except Exception as exc:                                   # noqa: BLE001
        print(f"bench.py: leg {key} failed: {exc!r}", file=sys.stderr)
        leg["error"] = repr(exc)[:300]
    leg["leg_wall_s"] = round(time.perf_counter() - t_setup, 1)
    _SCENES.pop((name, W, H, None), None)                      # a 1M-triangle scene is ~0.5 GB of host arrays: one at a time
    return leg


def serial_loop(scene_name, W, H, local_rank, n=8):
    """The reference's own way of driving a render (src/render.py:31-37): ONE `run_sample()` per iteration, then the picture
    is read (`Renderer.image` there; here its device-side form `tone_mapped()`, 3 bytes per pixel over PCIe) before the next
    sample starts -- no sample pipeline, no sample streams, the launch organisation of single calls (whole subpaths in one
    persistent launch on the mesh scenes).  ms per iteration over `n` iterations after 2 warm-up iterations."""
    from clive2_amd import _native
    from clive2_amd.renderer import Renderer, make_seeds
    scene, _ = build_scene(scene_name, W, H)
    r = Renderer(scene, seeds=make_seeds(W * H), device=local_rank % max(_native.lib().cl2_device_count(), 1))
    for _ in range(2):
        r.run_sample()
        r.tone_mapped()
    r.synchronize()
    t_run = t_map = 0.0
    for _ in range(n):
        t0 = time.perf_counter()
        r.run_sample()
        t1 = time.perf_counter()
        pic = r.tone_mapped()
        t2 = time.perf_counter()
        t_run += t1 - t0
        t_map += t2 - t1
    assert pic.shape == (H, W, 3)
    rays = r.counters()["rays"] / (n + 2)
    r.close()
    return {"ms_per_iteration": round((t_run + t_map) / n * 1e3, 3), "run_sample_ms": round(t_run / n * 1e3, 3),
            "tone_mapped_read_ms": round(t_map / n * 1e3, 3), "iterations": n,
            "mrays_per_s": round(rays / ((t_run + t_map) / n) / 1e6, 1),
            "what": "n x { run_sample(); tone_mapped() } on one handle with one seed buffer: the reference's loop, src/render.py:31-37"}


# ---------------------------------------------------------------- N > 1 without an external launcher
def dry_child(rank, local_rank, world):
    """`--dry-spawn`: what a rank process does up to the communicator bootstrap, without a GPU (CPU tests of the
    self-spawn path): the id hand-over through the shared rendezvous file with 128 random bytes standing in for
    ncclUniqueId.  Rank 0 prints one JSON line describing what every rank saw."""
    import hashlib as hl
    from clive2_amd.distributed import exchange_unique_id, finish_exchange, rendezvous_path
    if rank == int(os.environ.get("CLIVE2_BENCH_DRY_FAIL_RANK", "-1")):
        print(f"[rank {rank}] dry-spawn: failing on request", file=sys.stderr)
        sys.exit(3)
    hang_dir = os.environ.get("CLIVE2_BENCH_DRY_HANG_DIR")
    if hang_dir:
        # a rank that never finishes (stands for one blocked in a collective): the clean-up tests end the PARENT and look for us
        with open(os.path.join(hang_dir, f"rank{rank}.pid.tmp"), "w") as f:
            f.write(str(os.getpid()))
        os.replace(os.path.join(hang_dir, f"rank{rank}.pid.tmp"), os.path.join(hang_dir, f"rank{rank}.pid"))
        time.sleep(300.0)
        sys.exit(4)
    path = rendezvous_path()
    uid = exchange_unique_id(rank, world, lambda: os.urandom(128), 128, timeout=60.0)
    me = {"rank": rank, "local_rank": local_rank, "world": world, "pid": os.getpid(), "ppid": os.getppid(),
          "id_sha": hl.sha256(uid).hexdigest()[:16], "rendezvous": path,
          "explicit_file": os.environ.get("CLIVE2_RENDEZVOUS_FILE")}
    tmp = f"{path}.seen{rank}.tmp"
    with open(tmp, "w") as f:
        json.dump(me, f)
    os.replace(tmp, f"{path}.seen{rank}")
    if rank != 0:
        return
    seen, deadline = {}, time.monotonic() + 60.0
    while len(seen) < world and time.monotonic() < deadline:
        for k in range(world):
            if k not in seen and os.path.exists(f"{path}.seen{k}"):
                seen[k] = json.load(open(f"{path}.seen{k}"))
        time.sleep(0.01)
    for k in seen:
        os.unlink(f"{path}.seen{k}")
    finish_exchange(0)
    print(json.dumps({"dry_spawn": True, "n_gpus": world, "ranks": [seen[k] for k in sorted(seen)],
                      "ids_equal": len({v["id_sha"] for v in seen.values()}) == 1 and len(seen) == world}), flush=True)


def _die_with_parent():
    """preexec_fn of the rank processes: the kernel sends SIGTERM to the child when its parent dies -- however it dies,
    SIGKILL included, which no handler in the parent can see (prctl PR_SET_PDEATHSIG; Linux).  A rank that holds a GPU must
    not outlive the job that started it."""
    import ctypes
    import signal
    try:
        ctypes.CDLL(None, use_errno=True).prctl(1, int(signal.SIGTERM), 0, 0, 0)      # PR_SET_PDEATHSIG = 1
    except (OSError, AttributeError):
        pass


def spawn_ranks(n, argv, timeout):
    """The parent of `python bench.py --gpus N` (N > 1, no launcher environment).  It makes NO GPU / HIP call and never
    loads the library: it starts N fresh processes of this script (one per GPU) with RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_* and ONE shared CLIVE2_RENDEZVOUS_FILE (a fresh name: no stale file can exist), waits, relays rank 0's
    stdout (the JSON line) and returns the first non-zero exit code (after ending the other ranks), else 0.

    No exit path leaves a rank behind: the wait loop sits in try/finally, so an exception, Ctrl-C or a SIGTERM
    from the driver's time limit (turned into rc 128 + signal by the handlers below) ends every live child -- terminate,
    wait, then kill -- and removes the rendezvous file; and every child asks the kernel for a SIGTERM of its own when this
    process dies without running any of that (SIGKILL, out-of-memory kill): _die_with_parent."""
    import secrets
    import signal
    import socket
    import subprocess
    import tempfile
    import threading
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    rdv = os.path.join(tempfile.gettempdir(), f"clive2_bench_id_{os.getuid()}_{os.getpid()}_{secrets.token_hex(8)}")
    procs, chunks, reader = [], [], None
    stop = {"signal": 0}

    def on_signal(signum, _frame):
        stop["signal"] = signum

    previous = {sg: signal.signal(sg, on_signal) for sg in (signal.SIGTERM, signal.SIGINT, signal.SIGHUP)}
    rc = 0
    try:
        for rank in range(n):
            env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                       MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), CLIVE2_RENDEZVOUS_FILE=rdv,
                       HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
            # rank 0's stdout is the job's stdout (one JSON line); the other ranks' goes to stderr
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env, preexec_fn=_die_with_parent,
                                          stdout=subprocess.PIPE if rank == 0 else sys.stderr, stdin=subprocess.DEVNULL))
        reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
        reader.start()
        deadline = time.monotonic() + timeout
        live = set(range(n))
        while live and rc == 0:
            if stop["signal"]:
                print(f"bench.py: signal {stop['signal']}; ending the ranks", file=sys.stderr)
                rc = 128 + stop["signal"]
                break
            for k in sorted(live):
                code = procs[k].poll()
                if code is None:
                    continue
                live.discard(k)
                if code != 0:
                    print(f"bench.py: rank {k} exited with {code}; ending the other ranks", file=sys.stderr)
                    rc = code if code > 0 else 1
                    break
            if rc == 0 and live and time.monotonic() > deadline:
                print(f"bench.py: ranks {sorted(live)} still running after {timeout:.0f} s; ending them", file=sys.stderr)
                rc = 124
            if rc == 0 and live:
                time.sleep(0.05)
    except BaseException:
        rc = rc or 1
        raise
    finally:
        # exactly the processes started above, by pid: terminate, give them ten seconds, kill what is left
        alive = [p for p in procs if p.poll() is None]
        for p in alive:
            try:
                p.terminate()
            except OSError:
                pass
        t_kill = time.monotonic() + 10.0
        for p in alive:
            try:
                p.wait(timeout=max(0.1, t_kill - time.monotonic()))
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()
        if reader is not None:
            reader.join(timeout=10.0)
        for leftover in glob.glob(rdv + "*"):
            try:
                os.unlink(leftover)
            except OSError:
                pass
        for sg, h in previous.items():
            signal.signal(sg, h)
    if rc == 0:
        sys.stdout.write(b"".join(chunks).decode(errors="replace"))
        sys.stdout.flush()
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=128)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--total-spp", type=int, default=0,
                    help="strong scaling: a FIXED number of samples split over the ranks (samples_for_rank) instead of --steps per rank")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-order-extra", action="store_true", help="skip the labelled nearest-first extras of the mesh legs (detail file only)")
    ap.add_argument("--no-mesh", action="store_true", help="skip the mesh workloads (config 3 / 4 / 5 stand-ins) and the serial-loop figures of the N=1 line")
    ap.add_argument("--sample-streams", type=int, default=1,
                    help="sample streams of the headline workload (K independent samples of the frame per pass, cl2_set_sample_streams); "
                         "1 = the reference's single seed buffer per renderer")
    ap.add_argument("--mesh-streams", type=int, default=16, help="sample streams of the 1080p mesh legs (their K = 1 figure is reported beside it); round 6: 16 -- "
                         "per-level subpath launches of 33 M rays: configs 3 / 4 / 5 6.05 -> 5.94 / 7.18 -> 6.84 / 16.7 -> 16.4 ms per sample against 8, "
                         "24 and 32 no better (profiles/r06_sample_streams_16*.log); 3.4 KB of device memory per pixel entry: 113 GB at 16 x 1920 x 1080")
    ap.add_argument("--strong-spp", type=int, default=1024, help="N > 1: total samples of the strong-scaling leg (config 4 stand-in, split over the ranks); 0 = skip")
    ap.add_argument("--mesh-steps", type=int, default=64)
    ap.add_argument("--hbm-steps", type=int, default=32)
    ap.add_argument("--hbm4k-steps", type=int, default=8, help="samples of the config-5 leg at its own 3840 x 2160 (0 = skip)")
    ap.add_argument("--hbm4k-streams", type=int, default=2, help="sample streams of that leg (at 8.3 M pixels a launch is already large)")
    ap.add_argument("--debug-flags", type=int, default=0, help="launch-organisation switches (include/clive2_amd.h); results unchanged")
    ap.add_argument("--scene", default="cornell", choices=["cornell", "glass", "blob", "interior", "open"],
                    help="cornell = the BASELINE metric's workload (default); the others are the mesh configs "
                         "(SURVEY 8d C3-C5 stand-ins), for profiling only")
    ap.add_argument("--levels-per-launch", type=int, default=0, help="subpath bounces per launch (1..6; 0 = by survival, the default)")
    ap.add_argument("--pipelining", type=int, default=-1, help="sample pipeline: 0 serial, 1 subpaths of sample i+1 beside the connection phase of sample i, 2 three stages, -1 by frame size (default)")
    ap.add_argument("--traversal-mode", type=int, default=0, help="0 auto, 1 fused, 2 persistent traversal with ray replacement, 3 fused subpaths + persistent connection rays")
    ap.add_argument("--cpu-width", type=int, default=1920)
    ap.add_argument("--cpu-height", type=int, default=1080)
    ap.add_argument("--cpu-samples", type=int, default=4)
    ap.add_argument("--spawn-timeout", type=float, default=1500.0, help="self-spawned ranks are ended after this many seconds")
    ap.add_argument("--dry-spawn", action="store_true", help="rank processes stop after the id hand-over and need no GPU (CPU test of the self-spawn path)")
    args = ap.parse_args()
    if args.debug_flags & 7:
        sys.exit("bench.py: debug bits 0-2 skip parts of the resolve stage (invalid renders); they exist only in the test "
                 "variant of the library and are refused here")

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # no launcher: become the launcher, before anything touches the GPU
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:], args.spawn_timeout))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    args.gpus = world
    if args.dry_spawn:
        return dry_child(rank, local_rank, world)
    if not os.path.exists("/dev/kfd"):
        sys.exit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    # RCCL refuses two ranks on one device, so a one-GPU box cannot rehearse N > 1; CLIVE2_BENCH_FORCE_COMM=1 (rehearsal
    # only, NOT used by the driver) runs the N > 1 code path -- id exchange, communicator, barrier and clock through
    # cl2_comm_allreduce_f64, the all-reduce inside the timed region -- on a one-rank communicator.
    W, H = args.width, args.height
    with_comm = world > 1 or os.environ.get("CLIVE2_BENCH_FORCE_COMM") == "1"
    # stdout carries ONE JSON line.  Libraries write there too (RCCL prints a version banner on stdout when a communicator comes
    # up), so for the length of the run the file descriptor points at stderr; it is put back for the line itself.
    sys.stdout.flush()
    saved_stdout = os.dup(1)
    os.dup2(2, 1)
    strong = args.total_spp > 0
    if strong:
        from clive2_amd.distributed import samples_for_rank
        my_steps = samples_for_rank(args.total_spp, rank, world)
    else:
        my_steps = args.steps
    res = run_workload(args, args.scene, W, H, my_steps, args.warmup, rank, local_rank, world, with_comm=with_comm,
                       streams=args.sample_streams)
    strong_leg = None
    # (CLIVE2_BENCH_FORCE_STRONG=1: rehearsal of this leg on a one-rank communicator -- a second communicator in the same
    # process after the first one was destroyed; not used by the driver)
    want_strong = world > 1 or (with_comm and os.environ.get("CLIVE2_BENCH_FORCE_STRONG") == "1")
    if want_strong and not strong and args.strong_spp > 0 and not args.no_mesh and args.scene == "cornell":
        # the strong-scaling form on the same job: BASELINE config 4 -- a FIXED number of samples of the
        # 82k-triangle scene split over the ranks, one all-reduce at the end
        from clive2_amd.distributed import samples_for_rank
        t_leg = time.perf_counter()
        # this leg is an extra: whatever goes wrong in it must not cost the job its headline line.  A collective that a failed
        # rank never joins ends at the library's deadline, shortened here from its default 300 s
        os.environ["CLIVE2_COMM_TIMEOUT_S"] = os.environ.get("CLIVE2_BENCH_STRONG_TIMEOUT_S", "90")
        try:
            m = run_workload(args, "blob", W, H, samples_for_rank(args.strong_spp, rank, world), 2, rank, local_rank, world,
                             with_comm=True, streams=args.mesh_streams)
        except Exception as exc:                                   # noqa: BLE001 -- reported on the line, never fatal
            m = None
            print(f"[rank {rank}] strong-scaling leg failed: {exc!r}", file=sys.stderr)
            if rank == 0:
                strong_leg = {"error": repr(exc)[:300], "leg_wall_s": round(time.perf_counter() - t_leg, 1)}
        if rank == 0 and m is not None:
            strong_leg = {"workload": f"{m['scene_desc']} {W}x{H}, {args.strong_spp} spp in all, split over {world} GPUs "
                                      f"({m['steps_rank']} on rank 0, {args.mesh_streams} sample streams per GPU)",
                          "scaling": "strong", "samples_rendered_all_ranks": m["steps_total"], "seconds": round(m["dt"], 4),
                          "value": round(m["rays_total"] / m["dt"] / 1e6, 2), "unit": "Mrays/s",
                          "ms_per_sample_whole_job": round(m["dt"] / max(m["steps_total"], 1) * 1e3, 4),
                          "comm": m["comm"], "leg_wall_s": round(time.perf_counter() - t_leg, 1)}

    if rank == 0:
        steps_out = args.total_spp if strong else args.steps
        out = {
            "metric": "Mrays/sec (whole node) + HBM GB/s, 1080p Cornell box, 1/2/4/8 MI355X",
            "value": round(res["rays_total"] / res["dt"] / 1e6, 2),
            "unit": "Mrays/s",
            "n_gpus": world, "steps": steps_out, "warmup": args.warmup,
            # weak: every rank renders `steps` samples side by side; strong: `steps` samples in all, split over the ranks
            "ms_per_step": round(res["dt"] / (args.total_spp if strong else res["steps_rank"]) * 1e3, 3),
            "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{res['scene_desc']} {W}x{H}, BDPT{' diffuse-only' if args.scene == 'cornell' else ''}, "
                                   + (f"{args.total_spp} spp in all, split over {world} GPU(s)" if strong else f"{args.steps} spp per GPU"),
                       "width": W, "height": H,
                       "rays_per_pixel_sample": round(res["rays_total"] / (res["steps_total"] * W * H), 3),
                       "samples_rendered_all_ranks": res["steps_total"],
                       "sample_streams": res["sample_streams"],
                       "parallelism": f"sample-split x{world}, one in-place RCCL all-reduce of the accumulators"
                                      if with_comm else "one GPU (no collective)"},
            "roofline": res["roofline"],
            "stage_ms_per_step_serial": res["stages"],
        }
        if res["comm"]:
            # what RCCL itself reported: `nranks` ranks on `distinct_devices` different GPUs (PCI addresses of all ranks)
            out["comm"] = res["comm"]
        if strong_leg:
            out["strong_scaling"] = strong_leg
        if world == 1 and not args.no_mesh:
            try:
                out["serial_run_sample_ms"] = serial_loop(args.scene, W, H, local_rank)
            except Exception as exc:                               # noqa: BLE001 -- an extra never costs the job its headline
                out["serial_run_sample_ms"] = {"error": repr(exc)[:300]}
        if world == 1 and not args.no_mesh and args.scene == "cornell":
            # The other BASELINE.json configurations a single GPU holds, on the same line -- their tree is read through the caches,
            # and each leg says which resource binds its dominant launch (mesh_roofline): config 3 (5,136 triangles), config 4
            # (81,936), config 5 (1,003,536) at 1080p and at its own 3840 x 2160.  Their stage-share tuner runs in the warm-up
            # (cl2_tune).  Each leg runs with sample streams (every launch then carries K samples' rays: a per-level subpath
            # launch of ONE 1080p sample is 2 M rays on 524 k resident lanes and all tail) and once more with the reference's
            # single seed buffer (K = 1), reported beside it, and times the reference's serial loop (serial_run_sample_ms).
            K = args.mesh_streams
            legs = (("roofline_mesh", "glass", W, H, args.mesh_steps, K),
                    ("roofline_blob", "blob", W, H, args.mesh_steps, K),
                    ("roofline_hbm", "interior", W, H, args.hbm_steps, K),
                    ("roofline_hbm_4k", "interior", 3840, 2160, args.hbm4k_steps, min(K, args.hbm4k_streams)))
            for key, name, lw, lh, n_steps, k_leg in legs:
                if n_steps > 0:
                    out[key] = mesh_leg(args, key, name, lw, lh, n_steps, k_leg, local_rank)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.cpu_width, args.cpu_height, args.cpu_samples)
        emit(out, saved_stdout)
    os.close(saved_stdout)


if __name__ == "__main__":
    main()
