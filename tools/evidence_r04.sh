#!/bin/bash
# end-of-round evidence (gpurun): bench line, kernel-trace stats of the same command, bf16x6 table, PMC traffic of the bf16x6 kernels
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/ev_r04
rm -rf $OUT; mkdir -p $OUT
cd $ROOT
python bench.py > $OUT/bench_line.json 2> $OUT/bench.err
python tools/kbench_bf6.py > $OUT/gemm_bf6_vs_miopen.txt 2>/dev/null
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/step -- python $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/step_bench.json 2> $OUT/step.err
rocprofv3 --pmc FETCH_SIZE -d $OUT/pf -- python $ROOT/tools/kbench_bf6_one.py 48 256 1024 16 44 > $OUT/pf.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $OUT/pw -- python $ROOT/tools/kbench_bf6_one.py 48 256 1024 16 44 > $OUT/pw.log 2>&1
cd $ROOT
python tools/rocpd_summary.py $(ls $OUT/step/*/*.db | head -1) 60 > $OUT/step_kernel_stats.txt 2>&1
python tools/pmc_summary.py $(ls $OUT/pf/*/*.db | head -1) b6_ > $OUT/pmc_bf6_FETCH_SIZE.txt 2>&1
python tools/pmc_summary.py $(ls $OUT/pw/*/*.db | head -1) b6_ > $OUT/pmc_bf6_WRITE_SIZE.txt 2>&1
rm -rf $OUT/step $OUT/pf $OUT/pw
ls -la $OUT
