#!/usr/bin/env python3
"""A/B of one environment switch on the default bench workload, alternating runs on the same box:
    python tools/ab_env.py DBEV_BN_STRIPE 0 1 [repeats] [steps]"""
import json, os, subprocess, sys
var, a, b = sys.argv[1:4]
rep = int(sys.argv[4]) if len(sys.argv) > 4 else 2
steps = sys.argv[5] if len(sys.argv) > 5 else "20"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for i in range(rep):
    for val in (a, b):
        env = dict(os.environ, **{var: val})
        out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", steps, "--warmup", "4", "--no-cpu-baseline"],
                             env=env, capture_output=True, text=True).stdout.strip().splitlines()[-1]
        d = json.loads(out)
        r = d["roofline"]
        print(f"{var}={val} ms_per_step {d['ms_per_step']:.2f} roofline.frac {r['frac']:.3f} bn_family {json.dumps(r.get('bn_family'))[:400]}", flush=True)
