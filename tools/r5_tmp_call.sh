set -e
export TMPDIR=/tmp
mkdir -p gpurun_out
for s in glass blob interior; do timeout -k 10 300 python tools/exp_mesh_flags_ab.py $s 1 0 > gpurun_out/r5_tuner_fix_${s}1.log 2>&1; grep flags gpurun_out/r5_tuner_fix_${s}1.log | cut -c1-60; done
timeout -k 10 300 python tools/exp_mesh_flags_ab.py glass 8 0 > gpurun_out/r5_tuner_fix_glass8.log 2>&1; grep flags gpurun_out/r5_tuner_fix_glass8.log | cut -c1-60
