set -e
export TMPDIR=/tmp
mkdir -p gpurun_out
CLIVE2_BENCH_FORCE_COMM=1 CLIVE2_BENCH_FORCE_STRONG=1 timeout -k 10 400 python bench.py --gpus 1 --steps 32 --strong-spp 128 --no-cpu-baseline > gpurun_out/r5_bench_forced_comm.json 2> gpurun_out/r5_bench_forced_comm.err || { tail -20 gpurun_out/r5_bench_forced_comm.err; exit 1; }
python - <<'P'
import json
d=json.loads(open("gpurun_out/r5_bench_forced_comm.json").read().strip().splitlines()[-1])
print("value", d["value"], "comm", d.get("comm"), "strong", {k: d["strong_scaling"].get(k) for k in ("value","seconds","samples_rendered_all_ranks","error")} if d.get("strong_scaling") else None, "keys", [k for k in d if k.startswith("roofline") or k.startswith("serial")])
P
timeout -k 10 900 python tools/soak.py > gpurun_out/r5_soak.log 2>&1 || { tail -20 gpurun_out/r5_soak.log; exit 1; }
tail -12 gpurun_out/r5_soak.log
