#!/bin/bash
# dev tool (gpurun): counters of the stem kernels (csrc/stem.hip) at the bench shape, one counter group per rocprofv3 pass (never
# combined with trace domains).  usage: tools/pmc_stem.sh
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/pmc_stem
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_ANY" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $grp -d $OUT/p$i -- python $ROOT/tools/kbench_stem.py > $OUT/p$i.log 2>&1
  python $ROOT/tools/pmc_summary.py $(ls $OUT/p$i/*/*.db | head -1) stem_fwd stem_wgradE 2>&1 | sed 's/avg value(KB)/avg value    /' | cut -c1-60,91-140
done
