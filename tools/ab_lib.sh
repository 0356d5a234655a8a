#!/bin/bash
# Alternating A/B runs of the distillation step under two builds of the kernel library (gpurun): tools/ab_lib.sh <variant name> [rounds]
#   variant = distill_bev_amd/libdbev_hip_<name>.so (tools/build_variant.sh) against the default library
V=$1; N=${2:-2}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
for i in $(seq 1 $N); do
  for lib in "" "_$V"; do
    DBEV_HIP_LIB=$ROOT/distill_bev_amd/libdbev_hip$lib.so DBEV_BENCH_PLAIN=1 python bench.py --no-cpu-baseline 2>/dev/null | LIBN="libdbev_hip$lib.so" python -c '
import json, os, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d["roofline"]; o = r["other_hot_kernels"]
rows = {k: round(o[k]["ms_per_step"], 3) for k in sorted(o) if k.startswith(("bn_apply", "b6_", "wino_fwd", "bn_stats"))}
print("%-24s ms_per_step %.2f  median %.2f  bn family %.2f  %s" % (os.environ["LIBN"], d["ms_per_step"], d["ms_per_step_median"], r["bn_family"]["ms_per_step"], rows))
'
  done
done
