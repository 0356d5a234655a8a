#!/bin/bash
# Regenerate distill_bev_amd/miopen_db/ on an MI355X: one exhaustive MIOpen search per convolution-bearing bench workload
# (torch.backends.cudnn.benchmark via DBEV_MIOPEN_FIND=1), written to a scratch user-db directory.  ~17 minutes.
#   tools/tune_miopen.sh && cp gpurun_out/miopen_db/*db.txt distill_bev_amd/miopen_db/
R=${GRAFT_REPO_ROOT:-/root/repo}
export MIOPEN_USER_DB_PATH=$R/gpurun_out/miopen_db
mkdir -p $MIOPEN_USER_DB_PATH
for wl in distill_step bevformer_distill voxel_teacher; do
  t0=$(date +%s)
  DBEV_MIOPEN_FIND=1 python $R/bench.py --workload $wl --steps 3 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-200
  echo "$wl wall $(( $(date +%s) - t0 )) s"
done
ls -la $MIOPEN_USER_DB_PATH
