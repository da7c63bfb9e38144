set -e
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 600 python tools/exp_spec_parity.py 0x6000 > gpurun_out/r5_spec_parity.log 2>&1 || { cat gpurun_out/r5_spec_parity.log; exit 1; }
cat gpurun_out/r5_spec_parity.log
timeout -k 10 300 python tools/exp_mesh_flags_ab.py glass 8 0 0x2000 0x4000 0x6000 > gpurun_out/r5_spec_glass8.log 2>&1
cat gpurun_out/r5_spec_glass8.log
timeout -k 10 300 python tools/exp_mesh_flags_ab.py blob 8 0 0x6000 > gpurun_out/r5_spec_blob8.log 2>&1
cat gpurun_out/r5_spec_blob8.log
timeout -k 10 300 python tools/exp_mesh_flags_ab.py interior 8 0 0x2000 0x4000 0x6000 > gpurun_out/r5_spec_interior8.log 2>&1
cat gpurun_out/r5_spec_interior8.log
