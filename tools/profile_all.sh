#!/bin/bash
# Run ON THE GPU BOX from the repo root: the committed profiles of a round for every bench workload --
#   bash tools/profile_all.sh <tag> [small|interior|interior4k|k1|all]          (e.g. r05)
# tools/profile_round.sh per workload (rocprofv3 --kernel-trace --stats, then the PMC passes, each in its own run), summarised by
# tools/profile_summaries.py into gpurun_out/profiles_<tag>/ (copy those into profiles/).
set -e
TAG=${1:-r05}
OUT=gpurun_out/profiles_$TAG
mkdir -p $OUT profiles
run() {  # scene steps pmc_steps K W H suffix
    bash tools/profile_round.sh $1 $2 $3 $4 $5 $6 > gpurun_out/profile_round_$1$7.log 2>&1
    python tools/profile_summaries.py $TAG $1 $5 $6 $7 > gpurun_out/profile_summary_$1$7.log 2>&1
    cp profiles/${TAG}_kernel_stats_$1$7.csv profiles/${TAG}_bench_$1$7.log profiles/${TAG}_pmc_$1$7.json $OUT/
    cp gpurun_out/profile_summary_$1$7.log $OUT/${TAG}_profile_summary_$1$7.txt
    rm -rf gpurun_out/pf_$1_*      # the raw rocprofv3 output of a 1M-triangle workload is hundreds of MB
    echo "profiled $1$7"
}
case "${2:-all}" in
  small) run cornell 32 4 1 1920 1080 ""; run glass 64 16 16 1920 1080 ""; run blob 64 16 16 1920 1080 "" ;;
  interior) run interior 32 16 16 1920 1080 "" ;;
  interior4k) run interior 8 2 2 3840 2160 "_4k" ;;
  k1) run glass 32 4 1 1920 1080 "_k1"; run blob 32 4 1 1920 1080 "_k1"; run interior 16 4 1 1920 1080 "_k1"; run interior 8 2 1 3840 2160 "_4k_k1" ;;   # the legs' one_stream figures
  all) run cornell 32 4 1 1920 1080 ""; run glass 64 16 16 1920 1080 ""; run blob 64 16 16 1920 1080 ""; run interior 32 16 16 1920 1080 ""; run interior 8 2 2 3840 2160 "_4k" ;;
esac
