"""Turn the rocprofv3 output of tools/profile_round.sh (gpurun_out/pf_*) into the files committed under
profiles/:   python tools/profile_summaries.py r01_final"""
import collections
import csv
import glob
import json
import shutil
import subprocess
import sys


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r01_final"
    shutil.copy(glob.glob("gpurun_out/pf_stats/**/*_kernel_stats.csv", recursive=True)[0], f"profiles/{tag}_kernel_stats.csv")
    shutil.copy("gpurun_out/pf_bench.log", f"profiles/{tag}_bench.log")
    out = {"note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) of `python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline`; "
                   "units KiB per launch as reported; gfx950: FETCH_SIZE counts 64 B per 128-B request for wide coalesced reads "
                   "(MI355X_MICROARCH.md HBM section) -> hbm_bytes_per_launch = (2*FETCH + WRITE)*1024"}
    per = {}
    for kind in ("fetch", "write"):
        f = glob.glob(f"gpurun_out/pf_{kind}/**/*_counter_collection.csv", recursive=True)[0]
        agg = collections.defaultdict(lambda: [0, 0.0])
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            agg[k][0] += 1
            agg[k][1] += float(r["Counter_Value"])
        out[kind.upper() + "_SIZE_KiB"] = {k: {"launches": n, "per_launch": round(v / n, 1)} for k, (n, v) in agg.items()}
        for k, (n, v) in agg.items():
            per.setdefault(k, {})[kind] = v / n
    out["hbm_bytes_per_launch"] = {k: round((2 * v.get("fetch", 0) + v.get("write", 0)) * 1024) for k, v in per.items() if "rocclr" not in k}
    json.dump(out, open(f"profiles/{tag}_pmc_hbm.json", "w"), indent=1)
    sq = subprocess.run([sys.executable, "tools/pmc_summary.py", "gpurun_out/pf_sq"], capture_output=True, text=True).stdout
    open(f"profiles/{tag}_pmc_sq.txt", "w").write(sq)
    for r in csv.DictReader(open(f"profiles/{tag}_kernel_stats.csv")):
        if float(r["Percentage"]) > 1:
            print(f'{r["Name"][:60]:60s} calls={r["Calls"]:>5s} avg_us={float(r["AverageNs"]) / 1e3:10.1f} pct={r["Percentage"]}')
    for k, v in sorted(out["hbm_bytes_per_launch"].items(), key=lambda kv: -kv[1])[:8]:
        print(f"{k:56s} {v / 1e6:10.1f} MB per launch")


if __name__ == "__main__":
    main()
