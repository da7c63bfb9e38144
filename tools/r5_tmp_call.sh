set -e
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 300 python tools/exp_reproducible_cost.py cornell 1 > gpurun_out/r5_repro_cost_cornell.log 2>&1; cat gpurun_out/r5_repro_cost_cornell.log
timeout -k 10 300 python tools/exp_reproducible_cost.py glass 8 > gpurun_out/r5_repro_cost_glass8.log 2>&1; cat gpurun_out/r5_repro_cost_glass8.log
