#!/bin/bash
# round-6 evidence (gpurun): the bench line of every workload, kernel-trace stats of each, the clock probe.
# usage: tools/evidence_r06.sh [workload ...]   (default: all)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/ev_r06
mkdir -p $OUT
cd $ROOT
WL=${@:-"distill_step bevformer_distill bev_pool msda voxel_teacher"}
for w in $WL; do
  if [ "$w" = "distill_step" ]; then
    python bench.py > $OUT/bench_line.json 2> $OUT/bench.err
  else
    python bench.py --workload $w > $OUT/${w}_line.json 2> $OUT/${w}.err
  fi
  cd /tmp && export TMPDIR=/tmp
  DBEV_BENCH_PLAIN=1 rocprofv3 --kernel-trace --stats -d $OUT/raw_$w -- python $ROOT/bench.py --workload $w --steps 5 --warmup 2 --no-cpu-baseline > $OUT/${w}_traced.json 2> $OUT/${w}_traced.err
  cd $ROOT
  python tools/rocpd_summary.py $(ls $OUT/raw_$w/*/*.db | head -1) 150 > $OUT/${w}_kernel_stats.txt 2>&1
  rm -rf $OUT/raw_$w
done
ls -la $OUT
