set -e
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r5_pytest_gpu_a.log 2>&1 || { tail -30 gpurun_out/r5_pytest_gpu_a.log; exit 1; }
tail -3 gpurun_out/r5_pytest_gpu_a.log
( time timeout -k 10 600 python bench.py > gpurun_out/r5_bench_default_a.json 2> gpurun_out/r5_bench_default_a.err ) 2>&1 | tail -3
python - <<'P'
import json
d = json.loads(open("gpurun_out/r5_bench_default_a.json").read().strip().splitlines()[-1])
print("headline", d["value"], d["ms_per_step"], "serial", d.get("serial_run_sample_ms"))
for k in ("roofline_mesh", "roofline_blob", "roofline_hbm", "roofline_hbm_4k"):
    l = d.get(k, {})
    print(k, l.get("mrays_per_s"), l.get("ms_per_step"), "K1", (l.get("one_stream") or {}).get("mrays_per_s"), "bound", l.get("bound"), l.get("frac"), "own", (l.get("own_bytes") or {}).get("bytes_per_ray"), "serial", (l.get("serial_run_sample_ms") or {}).get("ms_per_iteration"), "wall", l.get("leg_wall_s"), l.get("error"))
print("cpu", d.get("cpu_baseline", {}).get("value"))
P
timeout -k 10 300 python tools/exp_mode4_streams.py > gpurun_out/r5_mode4_streams.log 2>&1
cat gpurun_out/r5_mode4_streams.log
