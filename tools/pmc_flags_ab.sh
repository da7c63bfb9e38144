# Run ON THE GPU BOX: one rocprofv3 --pmc pass (eight SQ counters, --kernel-trace only) per debug-flag setting of the Cornell box,
# summarised by tools/pmc_summary.py -- how the instruction mix of a kernel changes with a launch-organisation switch.
set -e
export TMPDIR=/tmp
for f in 0 0x800; do
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_WAIT_ANY --kernel-trace --output-format csv -d gpurun_out/x_pmc_$f -o run -- python3 tools/one_flags.py $f cornell 3 > gpurun_out/x_pmc_$f.log 2>&1
  echo "== flags $f"; python3 tools/pmc_summary.py gpurun_out/x_pmc_$f | grep "traverse_conn\|trace_subpath"
done
