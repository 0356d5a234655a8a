#!/bin/bash
# dev tool (gpurun): SQ counters of the bf16x6 GEMM on one shape, one counter group per pass.  usage: tools/pmc_bf6.sh N Cin Cout H W
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/pmc_bf6
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA" "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS" "SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS" "TCC_HIT TCC_MISS TCC_EA0_RDREQ"; do
  i=$((i+1))
  rocprofv3 --pmc $grp -d $OUT/p$i -- python $ROOT/tools/kbench_bf6_one.py "$@" > $OUT/p$i.log 2>&1
  python $ROOT/tools/pmc_summary.py $(ls $OUT/p$i/*/*.db | head -1) b6_fwd 2>&1 | sed 's/avg value(KB)/avg value    /'
done
