#!/bin/bash
# in-step A/B of the BN reduction-pass tunables: whole-step time + bn family time per variant (dev tool)
for cfg in "256 4 2048" "512 4 256" "512 4 1024" "1024 4 512" "256 4 512" "512 8 512"; do
  set -- $cfg
  DBEV_BN_RTPB=$1 DBEV_BN_RUNR=$2 DBEV_BN_RNBX=$3 python bench.py --steps 12 --warmup 4 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; o=r['other_hot_kernels']
print('$cfg', 'ms/step %.2f' % d['ms_per_step'], 'bn %.2f' % r['bn_family']['ms_per_step'], ' '.join('%s %.2f' % (k.split('<')[0]+k[-3:-1], v['ms_per_step']) for k,v in o.items() if k.startswith('bn_')))"
done
