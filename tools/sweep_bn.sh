#!/bin/bash
# A/B sweep of the BN reduction-pass tunables (dev tool): tools/sweep_bn.sh > gpurun_out/bn_sweep.txt
for tpb in 256 512 1024; do for unr in 4 8; do for nbx in 256 512 2048; do
  DBEV_BN_RTPB=$tpb DBEV_BN_RUNR=$unr DBEV_BN_RNBX=$nbx python tools/kbench_bn2.py 2>/dev/null
done; done; done
