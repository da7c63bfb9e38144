#!/bin/bash
# Run ON THE GPU BOX from the repo root: the rocprofv3 passes of one bench.py workload whose summaries go into
# profiles/ (tools/profile_summaries.py turns gpurun_out/pf_<scene>_* into profiles/<tag>_*_<scene>.*).
#   bash tools/profile_round.sh [scene=cornell] [steps=32] [pmc_steps=4] [sample_streams=1] [width=1920] [height=1080]
# Counters are collected in their own passes, with --kernel-trace only (no other trace domain), FETCH_SIZE and
# WRITE_SIZE apart (they do not fit one pass, MI355X_MICROARCH.md "rocprofv3 PMC slots").
set -e
export TMPDIR=/tmp
SCENE=${1:-cornell}
STEPS=${2:-32}
PSTEPS=${3:-4}
K=${4:-1}
W=${5:-1920}
H=${6:-1080}
OUT=gpurun_out
mkdir -p $OUT
B="bench.py --scene $SCENE --no-cpu-baseline --no-mesh --sample-streams $K --width $W --height $H"
echo $K > $OUT/pf_${SCENE}_streams.txt
echo $STEPS > $OUT/pf_${SCENE}_steps.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/pf_${SCENE}_stats -o runc -- python3 $B --steps $STEPS --warmup 4 > $OUT/pf_${SCENE}_stats.log 2>&1
echo "stats pass done"
pass() {  # name, counters...
    local name=$1; shift
    CLIVE2_BENCH_SKIP_TUNE=1 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/pf_${SCENE}_$name -o runc -- python3 $B --steps $PSTEPS --warmup 1 > $OUT/pf_${SCENE}_$name.log 2>&1
    echo "$name pass done"
}
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass sq SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY
pass tcc TCC_HIT_sum TCC_MISS_sum SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU
# round 5: where a load is served and how long it takes (names from `rocprofv3 -L` on gfx950)
pass tcp TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum
pass ea TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum
pass ea2 TCC_REQ_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum
pass lvl SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_LEVEL_WAVES SQ_BUSY_CYCLES
python $B --steps $STEPS --warmup 4 > $OUT/pf_${SCENE}_bench.log 2>&1
tail -1 $OUT/pf_${SCENE}_bench.log | cut -c1-300
