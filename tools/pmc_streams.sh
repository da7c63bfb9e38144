#!/bin/bash
# Run ON THE GPU BOX from the repo root: rocprofv3 PMC passes (counters only + kernel trace, each set in its own pass) of
# tools/one_streams.py:   bash tools/pmc_streams.sh <scene> <K> <passes> <tag>
set -e
export TMPDIR=/tmp
SCENE=$1; K=$2; N=$3; TAG=$4
OUT=gpurun_out
export ONE_TUNE=0
pass() {
    local name=$1; shift
    rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/pmc_${TAG}_$name -o run -- python3 tools/one_streams.py $SCENE $K $N > $OUT/pmc_${TAG}_$name.log 2>&1
    echo "$name pass done"
}
pass sq SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY
pass tcc TCC_HIT_sum TCC_MISS_sum SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU
pass fetch FETCH_SIZE
pass write WRITE_SIZE
# round 5: where a load is served, how long it takes (names from `rocprofv3 -L` on gfx950)
pass tcp TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum
pass ea TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum
pass ea2 TCC_REQ_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum
pass lvl SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_LEVEL_WAVES SQ_BUSY_CYCLES
