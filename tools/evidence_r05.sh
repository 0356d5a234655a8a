#!/bin/bash
# end-of-round evidence (gpurun): bench line, kernel-trace stats of the plain step, bf16x6 tables (round-4 vs round-5 kernels), SQ counters
# and HBM traffic of the bf16x6 forward kernel (separate --pmc passes, never combined with trace domains)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/ev_r05
rm -rf $OUT; mkdir -p $OUT
cd $ROOT
python bench.py > $OUT/bench_line.json 2> $OUT/bench.err
(echo "# round-4 kernels (DBEV_BF6_V=1 DBEV_BF6_WV=1: b6_fwd / b6_wgrad), same box, same run"; DBEV_BF6_V=1 DBEV_BF6_WV=1 DBEV_BF6_WGRAD_LIB_ROWS=1000000000 python tools/kbench_bf6.py 2>/dev/null;
 echo; echo "# round-5 kernels (default: b6_fwd2 / b6_wgrad2)"; DBEV_BF6_WGRAD_LIB_ROWS=1000000000 python tools/kbench_bf6.py 2>/dev/null) > $OUT/gemm_bf6_vs_miopen.txt
cd /tmp && export TMPDIR=/tmp
DBEV_BENCH_PLAIN=1 rocprofv3 --kernel-trace --stats -d $OUT/step -- python $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/step_bench.json 2> $OUT/step.err
rocprofv3 --pmc FETCH_SIZE -d $OUT/pf -- python $ROOT/tools/kbench_bf6_one.py 48 256 1024 16 44 > $OUT/pf.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $OUT/pw -- python $ROOT/tools/kbench_bf6_one.py 48 256 1024 16 44 > $OUT/pw.log 2>&1
cd $ROOT
python tools/rocpd_summary.py $(ls $OUT/step/*/*.db | head -1) 150 > $OUT/step_kernel_stats.txt 2>&1
python tools/pmc_summary.py $(ls $OUT/pf/*/*.db | head -1) b6_ > $OUT/pmc_bf6_FETCH_SIZE.txt 2>&1
python tools/pmc_summary.py $(ls $OUT/pw/*/*.db | head -1) b6_ > $OUT/pmc_bf6_WRITE_SIZE.txt 2>&1
rm -rf $OUT/step $OUT/pf $OUT/pw
(echo "# round 5: tools/pmc_bf6.sh 48 256 1024 16 44 -- SQ counters of the bf16x6 forward kernel, one counter group per pass";
 echo "## round-4 kernel (DBEV_BF6_V=1)"; DBEV_BF6_V=1 bash tools/pmc_bf6.sh 48 256 1024 16 44 2>&1 | grep -v "^$";
 echo "## round-5 kernel (b6_fwd2)"; bash tools/pmc_bf6.sh 48 256 1024 16 44 2>&1 | grep -v "^$") > $OUT/pmc_bf6.txt
rm -rf $ROOT/gpurun_out/pmc_bf6
ls -la $OUT
