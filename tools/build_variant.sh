#!/bin/bash
# dev tool: a second library with wino.hip compiled under extra flags (A/B runs through DBEV_HIP_LIB).
# usage: [SRC=gemm_bf6.hip] tools/build_variant.sh <name> [extra hipcc flags...]   ->  distill_bev_amd/libdbev_hip_<name>.so
set -e
NAME=$1; shift
cd "$(dirname "$0")/../distill_bev_amd/csrc"
make -s -j8
mkdir -p /tmp/dbev_variant_$NAME
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -Wall -Wno-unused-result "$@" -c ${SRC:-wino.hip} -o /tmp/dbev_variant_$NAME/wino.o
OBJS=$(ls *.o | grep -v "^$(basename ${SRC:-wino.hip} .hip).o$")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS /tmp/dbev_variant_$NAME/wino.o -o ../libdbev_hip_$NAME.so
ls -la ../libdbev_hip_$NAME.so
