set -e
export TMPDIR=/tmp
mkdir -p gpurun_out
rocprofv3 -L > gpurun_out/r5_counters_list.txt 2>&1 || true
timeout -k 10 900 python -m pytest tests/test_gpu_round5.py -x -q -m gpu > gpurun_out/r5_pytest_round5.log 2>&1
tail -3 gpurun_out/r5_pytest_round5.log
timeout -k 10 300 python tools/exp_mesh_flags_ab.py glass 8 0 0x2000 > gpurun_out/r5_dirq_glass8.log 2>&1
cat gpurun_out/r5_dirq_glass8.log
timeout -k 10 300 python tools/exp_mesh_flags_ab.py interior 8 0 0x2000 > gpurun_out/r5_dirq_interior8.log 2>&1
cat gpurun_out/r5_dirq_interior8.log
timeout -k 10 300 python tools/exp_mesh_flags_ab.py glass 1 0 0x2000 > gpurun_out/r5_dirq_glass1.log 2>&1
cat gpurun_out/r5_dirq_glass1.log
