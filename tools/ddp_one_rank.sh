#!/bin/bash
# One-rank probe of the data-parallel step (gpurun, 1 GPU): the training step with the GradReducer active on a world-size-1 RCCL group
# (DBEV_FORCE_DDP=1: bucket pack + all-reduce + unpack all run, the collective itself is a device copy) against the plain step.
# What it measures: the EXPOSED cost of the reducer's machinery on one GPU, with the overlap hooks off (default) and on.  What it
# cannot measure: the xGMI ring time at N > 1.   usage: tools/ddp_one_rank.sh [steps]   -> gpurun_out/ddp_one_rank.json
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/ddp_probe
mkdir -p $OUT
S=${1:-15}
cd $ROOT
export DBEV_BENCH_PLAIN=1 HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python bench.py --no-cpu-baseline --steps $S --warmup 4 > $OUT/plain.json 2> $OUT/plain.err
DBEV_FORCE_DDP=1 DBEV_DDP_OVERLAP=0 timeout 600 python bench.py --no-cpu-baseline --steps $S --warmup 4 > $OUT/ddp.json 2> $OUT/ddp.err
DBEV_FORCE_DDP=1 DBEV_DDP_OVERLAP=1 timeout 600 python bench.py --no-cpu-baseline --steps $S --warmup 4 > $OUT/ddp_overlap.json 2> $OUT/ddp_overlap.err
timeout 600 python bench.py --no-cpu-baseline --steps $S --warmup 4 > $OUT/plain2.json 2> $OUT/plain2.err
python - <<PY
import json
def ms(f):
    try:
        d = json.loads(open("$OUT/" + f).read().strip().splitlines()[-1])
        return {"ms_per_step": d["ms_per_step"], "ms_per_step_median": d["ms_per_step_median"], "backend": d["config"].get("collective_backend")}
    except Exception as e:
        return {"error": repr(e)}
out = {"what": "one MI355X, world size 1: plain step vs DBEV_FORCE_DDP=1 (RCCL group of one rank, GradReducer packing 217 MB of gradients in 32 MB "
               "buckets, all-reduce, unpack) with the overlap hooks off / on; plain measured before and after",
       "steps": $S, "plain": ms("plain.json"), "ddp_overlap_off": ms("ddp.json"), "ddp_overlap_on": ms("ddp_overlap.json"), "plain_again": ms("plain2.json")}
json.dump(out, open("$ROOT/gpurun_out/ddp_one_rank.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
