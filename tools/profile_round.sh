#!/bin/bash
# Run ON THE GPU BOX from the repo root: the rocprofv3 passes and the default bench whose summaries
# go into profiles/ (tools/profile_summaries.py turns gpurun_out/pf_* into the committed files).
#   bash tools/profile_round.sh
set -e
export TMPDIR=/tmp
OUT=gpurun_out
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/pf_stats -o runc -- python3 bench.py --no-cpu-baseline > $OUT/pf_stats.log 2>&1
echo "stats pass done"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pf_fetch -o runc -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline > $OUT/pf_fetch.log 2>&1
echo "fetch pass done"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pf_write -o runc -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline > $OUT/pf_write.log 2>&1
echo "write pass done"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $OUT/pf_sq -o runc -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline > $OUT/pf_sq.log 2>&1
echo "sq pass done"
python bench.py > $OUT/pf_bench.log 2>&1
tail -1 $OUT/pf_bench.log | cut -c1-200
