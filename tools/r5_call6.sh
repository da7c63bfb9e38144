set -e
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_native_abi.py tests/test_gpu_round5.py -x -q -m gpu > gpurun_out/r5_pytest_gpu_b.log 2>&1 || { tail -30 gpurun_out/r5_pytest_gpu_b.log; exit 1; }
tail -3 gpurun_out/r5_pytest_gpu_b.log
for s in glass blob interior; do timeout -k 10 120 python tools/exp_serial_single.py $s 8 > gpurun_out/r5_serial_single_spec_$s.log 2>&1; cat gpurun_out/r5_serial_single_spec_$s.log; done
for v in base reps1 reps3 refill8 refill24; do
  echo "== variant $v"
  CL2_LIB=build/lib_$v.so timeout -k 10 200 python tools/exp_mesh_flags_ab.py glass 8 0 > gpurun_out/r5_var_${v}_glass8.log 2>&1; grep flags gpurun_out/r5_var_${v}_glass8.log
  CL2_LIB=build/lib_$v.so timeout -k 10 200 python tools/exp_mesh_flags_ab.py blob 8 0 > gpurun_out/r5_var_${v}_blob8.log 2>&1; grep flags gpurun_out/r5_var_${v}_blob8.log
done
for v in base refill8 refill24; do
  echo "== variant $v interior"
  CL2_LIB=build/lib_$v.so timeout -k 10 200 python tools/exp_mesh_flags_ab.py interior 8 0 > gpurun_out/r5_var_${v}_interior8.log 2>&1; grep flags gpurun_out/r5_var_${v}_interior8.log
done
