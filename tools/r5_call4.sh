set -e
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 300 python tools/exp_mesh_flags_ab.py glass 1 0 0x400 0x300 0x500 0x800 > gpurun_out/r5_share_glass1.log 2>&1
cat gpurun_out/r5_share_glass1.log
timeout -k 10 300 python tools/exp_reproducible_cost.py cornell 1 > gpurun_out/r5_repro_cost_cornell.log 2>&1
cat gpurun_out/r5_repro_cost_cornell.log
timeout -k 10 300 python tools/exp_reproducible_cost.py glass 8 > gpurun_out/r5_repro_cost_glass8.log 2>&1
cat gpurun_out/r5_repro_cost_glass8.log
timeout -k 10 600 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "reproducible" > gpurun_out/r5_pytest_repro_full.log 2>&1 || { tail -30 gpurun_out/r5_pytest_repro_full.log; exit 1; }
tail -3 gpurun_out/r5_pytest_repro_full.log
