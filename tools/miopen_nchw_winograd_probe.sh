#!/bin/bash
# Round 4 probe: what does MIOpen pick for the fp32 3x3 stride-1 layers of the step in NCHW (where its
# Winograd solvers are applicable) vs NHWC (implicit GEMM)?  Output: gpurun_out/r4/miopen_probe.txt
out=gpurun_out/r4/miopen_probe.txt
mkdir -p gpurun_out/r4
: > $out
export MIOPEN_USER_DB_PATH=/tmp/miopen_probe_db
mkdir -p $MIOPEN_USER_DB_PATH
run() {  # n c h w k layout
  for F in 1 2 4; do
    echo "### n=$1 c=$2 H=$3 W=$4 k=$5 layout=$6 F=$F $7" >> $out
    env $7 timeout 300 /opt/rocm/bin/MIOpenDriver conv -n $1 -c $2 -H $3 -W $4 -k $5 -y 3 -x 3 -p 1 -q 1 -u 1 -v 1 -l 1 -j 1 -m conv -g 1 \
       -F $F -t 1 -s 1 -V 0 -i 10 -I $6 -O $6 -f $6 2>&1 | grep -E "Algorithm|Elapsed|Solution|rror" >> $out
  done
}
run 48 256 16 44 256 NCHW ""
run 48 256 16 44 256 NHWC ""
run 48 256 16 44 256 NCHW "MIOPEN_DEBUG_AMD_MP_BD_WINOGRAD_F2X3=1 MIOPEN_DEBUG_AMD_MP_BD_WINOGRAD_F3X3=1 MIOPEN_DEBUG_AMD_MP_BD_XDLOPS_WINOGRAD_F2X3=1 MIOPEN_DEBUG_AMD_MP_BD_XDLOPS_WINOGRAD_F3X3=1"
run 8 64 128 128 64 NCHW ""
run 8 512 128 128 256 NCHW ""
run 48 64 64 176 64 NCHW ""
cat $out
