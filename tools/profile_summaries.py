"""Turn the rocprofv3 output of tools/profile_round.sh (gpurun_out/pf_<scene>_*) into the files committed under
profiles/:   python tools/profile_summaries.py r02 cornell [width height [suffix]]

Writes profiles/<tag>_kernel_stats_<scene>.csv, profiles/<tag>_bench_<scene>.log and profiles/<tag>_pmc_<scene>.json.
The JSON holds, per kernel and per launch: FETCH_SIZE / WRITE_SIZE (KiB as reported, separate passes), the HBM
bytes after the gfx950 correction of MI355X_MICROARCH.md (FETCH_SIZE counts 64 B per 128-B request on wide
coalesced reads: hbm_bytes = (2*FETCH + WRITE) * 1024), the SQ counters, TCC hits / misses, and the hash of the
kernel sources that were profiled -- bench.py quotes these figures (marked "static") only while that hash still
matches the sources it runs."""
import collections
import csv
import glob
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def short(name):
    """cl2::k_traverse_persistent<false, true, cl2::ConnRaySource>(...) -> k_traverse_persistent<false,true,ConnRaySource>"""
    n = name.split("(")[0].replace("void ", "").replace("cl2::", "").replace(" ", "")
    return n


def per_launch(pattern):
    files = glob.glob(pattern, recursive=True)
    if not files:
        return {}
    agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
    for r in csv.DictReader(open(files[0])):
        a = agg[short(r["Kernel_Name"])][r["Counter_Name"]]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
    return {k: {c: (v / n, n) for c, (n, v) in cs.items()} for k, cs in agg.items()}


def main():
    tag, scene = sys.argv[1], sys.argv[2]
    W, H = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (1920, 1080)
    suffix = sys.argv[5] if len(sys.argv) > 5 else ""
    import bench
    base = f"gpurun_out/pf_{scene}"
    streams = int(open(f"{base}_streams.txt").read()) if os.path.exists(f"{base}_streams.txt") else 1
    stats = glob.glob(f"{base}_stats/**/*_kernel_stats.csv", recursive=True)
    if stats:
        shutil.copy(stats[0], f"profiles/{tag}_kernel_stats_{scene}{suffix}.csv")
    if os.path.exists(f"{base}_bench.log"):
        shutil.copy(f"{base}_bench.log", f"profiles/{tag}_bench_{scene}{suffix}.log")
    kernels = collections.defaultdict(dict)
    for kind in ("fetch", "write", "sq", "tcc", "tcp", "ea", "ea2", "lvl"):
        for k, cs in per_launch(f"{base}_{kind}/**/*_counter_collection.csv").items():
            if "rocclr" in k or "export" in k:
                continue
            for c, (v, n) in cs.items():
                kernels[k][c] = round(v, 1)
                kernels[k]["launches_" + kind] = n
    for k, row in kernels.items():
        if "FETCH_SIZE" in row and "WRITE_SIZE" in row:
            row["hbm_bytes"] = round((2 * row["FETCH_SIZE"] + row["WRITE_SIZE"]) * 1024)
        if "TCC_HIT_sum" in row and "TCC_MISS_sum" in row and row["TCC_HIT_sum"] + row["TCC_MISS_sum"] > 0:
            row["l2_hit_rate"] = round(row["TCC_HIT_sum"] / (row["TCC_HIT_sum"] + row["TCC_MISS_sum"]), 4)
        if row.get("SQ_WAVE_CYCLES"):
            row["wait_share"] = round(row.get("SQ_WAIT_ANY", 0.0) / row["SQ_WAVE_CYCLES"], 4)
        # average active lanes per VALU wave-instruction: thread-cycles / (4 cycles per quad-cycle-counted instruction)
        # round 5: the fabric read bytes from the request-size mix (TCC_EA0_RDREQ = 32-byte + 64-byte + 128-byte requests): what
        # FETCH_SIZE (= RDREQ x 64 B) under- or over-counts for THIS access pattern
        if all(k in row for k in ("TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_32B_sum", "TCC_EA0_RDREQ_64B_sum", "TCC_EA0_RDREQ_128B_sum")):
            n32, n64, n128 = row["TCC_EA0_RDREQ_32B_sum"], row["TCC_EA0_RDREQ_64B_sum"], row["TCC_EA0_RDREQ_128B_sum"]
            other = row["TCC_EA0_RDREQ_sum"] - n32 - n64 - n128
            row["fabric_read_bytes_by_request_size"] = round(32 * n32 + 64 * n64 + 128 * n128 + 64 * max(other, 0))
            row["fabric_read_requests_unsized"] = round(other)
        # round 5: average latency of an L1 -> L2 read request (cycles) and of a vector memory read as the wave sees it
        if row.get("TCP_TCC_READ_REQ_sum") and row.get("TCP_TCC_READ_REQ_LATENCY_sum"):
            row["l1_miss_latency_cycles"] = round(row["TCP_TCC_READ_REQ_LATENCY_sum"] / row["TCP_TCC_READ_REQ_sum"], 1)
        if row.get("SQ_INST_LEVEL_VMEM") and row.get("SQ_INSTS_VMEM_RD"):
            row["vmem_read_latency_cycles"] = round(row["SQ_INST_LEVEL_VMEM"] / row["SQ_INSTS_VMEM_RD"], 1)
        if row.get("SQ_THREAD_CYCLES_VALU") and row.get("SQ_INSTS_VALU"):
            row["thread_cycles_per_valu_inst"] = round(row["SQ_THREAD_CYCLES_VALU"] / row["SQ_INSTS_VALU"], 2)
    # the connection-ray traversal launch: the 4-wide walk where the scene uses it (its left-over launch of the binary
    # kernel carries a handful of rays), else the binary persistent walk, else the LDS kernel of the small scenes
    # (k_traverse_wide<REPS, Source, TALLY, SPEC>: the tallying variant runs in the warm-up's counting pass only and is not the launch)
    conn = ([k for k in kernels if k.startswith("k_traverse_wide<") and "ConnRaySource,false" in k] or
            [k for k in kernels if k.startswith("k_traverse_persistent<false") and "ConnRaySource" in k] or
            [k for k in kernels if k.startswith("k_traverse_conn<false")])
    # The --stats average of a kernel covers EVERY launch of the process: the share tuner's 54 passes (other grid shares: slower launches),
    # the timed region, the serial-order breakdown pass.  What bench.py's live `avg_launch_ms` must agree with is the timed region's
    # launches alone: the last `passes + breakdown passes` launches of the connection kernel in the kernel trace, the breakdown's cut off.
    timed = None
    trace = glob.glob(f"{base}_stats/**/*_kernel_trace.csv", recursive=True)
    if trace and conn and os.path.exists(f"{base}_steps.txt"):
        steps = int(open(f"{base}_steps.txt").read())
        passes = max(1, -(-steps // streams))
        n_break = max(1, min(passes, max(1, 8 // streams)))
        spans = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in csv.DictReader(open(trace[0])) if short(r["Kernel_Name"]) == conn[0])
        if len(spans) >= passes + n_break:
            t = spans[-(passes + n_break):-n_break]
            a = spans[-n_break:]
            timed = {"kernel": conn[0], "launches_in_timed_region": passes, "avg_ms": round(sum(e - b for b, e in t) / len(t) / 1e6, 4),
                     "serial_breakdown_launches": n_break, "alone_avg_ms": round(sum(e - b for b, e in a) / len(a) / 1e6, 4),
                     "all_launches": len(spans), "all_launches_avg_ms": round(sum(e - b for b, e in spans) / len(spans) / 1e6, 4),
                     "what": "rocprofv3 --kernel-trace of `bench.py --scene ... --steps %d`: the launches of the timed region (pipelined), of the serial "
                             "breakdown pass (alone) and of the whole process (tuner included: what the --stats average covers)" % steps}
    out = {"note": "rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE, SQ_*, TCC_* each in its own run, --kernel-trace only) of "
                   f"`python3 bench.py --scene {scene} --no-cpu-baseline --no-mesh --sample-streams {streams} --width {W} --height {H} --steps 4 --warmup 1`; values are averages per launch "
                   f"(a launch of a handle with {streams} sample stream(s) carries the rays of {streams} sample(s)); "
                   "FETCH_SIZE/WRITE_SIZE in KiB as reported; hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950 FETCH_SIZE half-count, "
                   "MI355X_MICROARCH.md HBM section); SQ_WAVE_CYCLES/SQ_WAIT_*/SQ_ACTIVE_INST_* are quad-cycles",
           "scene": scene, "width": W, "height": H, "sample_streams": streams, "sources_sha": bench.kernel_sources_sha(),
           "conn_traversal_kernel": conn[0] if conn else None, "kernel_trace_timed_region": timed, "kernels": kernels}
    json.dump(out, open(f"profiles/{tag}_pmc_{scene}{suffix}.json", "w"), indent=1, sort_keys=True)
    if stats:
        for r in csv.DictReader(open(f"profiles/{tag}_kernel_stats_{scene}{suffix}.csv")):
            if float(r["Percentage"]) > 1:
                print(f'{r["Name"][:70]:70s} calls={r["Calls"]:>5s} avg_us={float(r["AverageNs"]) / 1e3:10.1f} pct={r["Percentage"]}')
    if timed:
        print("kernel trace:", timed)
    for k, row in sorted(kernels.items(), key=lambda kv: -kv[1].get("hbm_bytes", 0))[:8]:
        print(f"{k[:60]:60s} hbm {row.get('hbm_bytes', 0) / 1e6:9.1f} MB  VALU {row.get('SQ_INSTS_VALU', 0):.3g}  wait {row.get('wait_share')}  L2 hit {row.get('l2_hit_rate')}")


if __name__ == "__main__":
    main()
