#!/bin/bash
# Alternating A/B runs of the distillation step under one environment switch (gpurun): tools/ab_env.sh VAR A B [rounds]
#   e.g. tools/ab_env.sh DBEV_BN_TICKET 0 1 2   -> prints ms_per_step / median and the bn_* rows of each run
VAR=$1; A=$2; B=$3; N=${4:-2}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
for i in $(seq 1 $N); do
  for v in $A $B; do
    env $VAR=$v DBEV_BENCH_PLAIN=1 python bench.py --no-cpu-baseline 2>/dev/null | VAR=$VAR VAL=$v python -c '
import json, os, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d["roofline"]; o = r["other_hot_kernels"]
rows = {k: (o[k]["launches_per_step"], round(o[k]["ms_per_step"], 3)) for k in sorted(o) if k.startswith(("bn_", "b6_", "wino_", "stem_"))}
print("%s=%s  ms_per_step %.2f  median %.2f  bn family %.2f  %s" % (os.environ["VAR"], os.environ["VAL"], d["ms_per_step"], d["ms_per_step_median"], r["bn_family"]["ms_per_step"], rows))
'
  done
done
