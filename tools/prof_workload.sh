#!/bin/bash
# kernel-trace stats of one bench workload (run on the GPU box through gpurun).   usage: tools/prof_workload.sh <workload> <tag> [env...]
W=$1; TAG=$2
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/raw -- python $ROOT/bench.py --workload $W --no-cpu-baseline > $OUT/bench.json 2> $OUT/err.log
cd $ROOT
python tools/rocpd_summary.py $(ls $OUT/raw/*/*.db | head -1) 250 > $OUT/${W}_kernel_stats.txt 2>&1
rm -rf $OUT/raw
head -30 $OUT/${W}_kernel_stats.txt | cut -c1-100,112-160
