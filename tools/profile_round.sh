#!/bin/bash
# Round profile (run on the GPU box through gpurun): kernel-trace stats of the two bench workloads + separate PMC passes.
# usage: tools/profile_round.sh r02
R=${1:-rXX}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_$R
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/step -- python $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/step_bench.json 2> $OUT/step.err
rocprofv3 --kernel-trace --stats -d $OUT/bevpool -- python $ROOT/bench.py --workload bev_pool --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bevpool_bench.json 2> $OUT/bevpool.err
rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch -- python $ROOT/tools/pmc_target.py > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_write -- python $ROOT/tools/pmc_target.py > $OUT/pmc_write.log 2>&1
cd $ROOT
for d in step bevpool; do python tools/rocpd_summary.py $(ls $OUT/$d/*/*.db | head -1) 120 > $OUT/${d}_kernel_stats.txt 2>&1; done
python tools/pmc_summary.py $(ls $OUT/pmc_fetch/*/*.db | head -1) > $OUT/pmc_FETCH_SIZE.txt 2>&1
python tools/pmc_summary.py $(ls $OUT/pmc_write/*/*.db | head -1) > $OUT/pmc_WRITE_SIZE.txt 2>&1
rm -rf $OUT/step $OUT/bevpool $OUT/pmc_fetch $OUT/pmc_write
ls -la $OUT
