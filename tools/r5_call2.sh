set -e
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 200 python tools/exp_mesh_flags_ab.py interior 8 0 0x2000 0x4000 0x6000 > gpurun_out/r5_pf_interior8.log 2>&1
cat gpurun_out/r5_pf_interior8.log
timeout -k 10 200 python tools/exp_mesh_flags_ab.py glass 8 0 0x2000 0x4000 0x6000 > gpurun_out/r5_pf_glass8.log 2>&1
cat gpurun_out/r5_pf_glass8.log
for s in glass blob interior; do timeout -k 10 120 python tools/exp_serial_single.py $s 8 > gpurun_out/r5_serial_single_$s.log 2>&1; cat gpurun_out/r5_serial_single_$s.log; done
timeout -k 10 600 bash tools/pmc_streams.sh interior 8 3 r5int8 > gpurun_out/r5_pmc_int8.log 2>&1
for k in sq tcc fetch write tcp ea lvl; do python3 tools/pmc_summary.py gpurun_out/pmc_r5int8_$k > gpurun_out/r5_pmc_int8_$k.txt 2>&1; done
cat gpurun_out/r5_pmc_int8_*.txt | grep "traverse_wide" | cut -c1-400
