# A/B of the fused 1x1-conv + statistics GEMM inside the step, alternating on one box
for i in 1 2; do
for cfg in "DBEV_CONV1X1=0" "DBEV_CONV1X1_MIN_ROWS=300000" "DBEV_CONV1X1_MIN_ROWS=100000"; do
  env $cfg python bench.py --steps 15 --warmup 4 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); o=d['roofline']['other_hot_kernels']
print('$cfg', round(d['ms_per_step'],2), {k: round(v['ms_per_step'],2) for k,v in o.items() if k in ('bn_stats','c1x1_fwd','bn_apply<true,*>','bn_apply<false,*>')})"
done; done
