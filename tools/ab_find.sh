#!/bin/bash
# A/B of MIOpen's find mode (torch.backends.cudnn.benchmark, DBEV_MIOPEN_FIND=1) vs the default immediate mode, both workloads,
# with the wall time of each run (the first-touch solver search is part of it).   tools/ab_find.sh > gpurun_out/ab_find.txt
for wl in bevformer_distill distill_step; do for f in 1 0; do
  t0=$(date +%s)
  line=$(DBEV_MIOPEN_FIND=$f python bench.py --workload $wl --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1)
  t1=$(date +%s)
  echo "$wl FIND=$f wall $((t1 - t0)) s  $(echo "$line" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print("ms_per_step", round(d["ms_per_step"], 2))')"
done; done
