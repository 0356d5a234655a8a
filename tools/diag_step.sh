#!/bin/bash
# dev tool (gpurun): kernel trace of the bench step -> full per-kernel table, idle-gap summary, convolution time by shape
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/diag
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/step -- python $ROOT/tools/steps_only.py 6 > $OUT/steps.log 2>&1
cd $ROOT
DB=$(ls $OUT/step/*/*.db | head -1)
python tools/rocpd_summary.py $DB 400 > $OUT/kernel_stats_full.txt 2>&1
python tools/gpu_gaps.py $DB 330 > $OUT/gaps.txt 2>&1
python tools/gap_context.py $DB 330 10 > $OUT/gap_context.txt 2>&1
python tools/step_convstats.py > $OUT/convstats.txt 2>&1
rm -rf $OUT/step
