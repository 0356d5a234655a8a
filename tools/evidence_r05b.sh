#!/bin/bash
# end-of-round evidence, second half of round 5 (gpurun): bench line, kernel-trace stats of the plain step, the stem / 3x3-stride-2 kernel
# tables.  (The bf16x6 PMC passes of tools/evidence_r05.sh are unchanged by this half of the round and are not repeated.)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/ev_r05b
rm -rf $OUT; mkdir -p $OUT
cd $ROOT
python bench.py > $OUT/bench_line.json 2> $OUT/bench.err
python tools/kbench_stem.py > $OUT/stem_vs_miopen.txt 2>/dev/null
python tools/kbench_c3s2.py 2>/dev/null | grep -v Warn > $OUT/conv3x3s2_vs_miopen.txt
python tools/fwd_host_vs_gpu.py 2>/dev/null | grep "host issue\|adjacent" > $OUT/fwd_host_vs_gpu.txt
cd /tmp && export TMPDIR=/tmp
DBEV_BENCH_PLAIN=1 rocprofv3 --kernel-trace --stats -d $OUT/step -- python $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/step_bench.json 2> $OUT/step.err
cd $ROOT
python tools/rocpd_summary.py $(ls $OUT/step/*/*.db | head -1) 150 > $OUT/step_kernel_stats.txt 2>&1
rm -rf $OUT/step
ls -la $OUT
