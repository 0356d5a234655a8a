#!/bin/bash
# dev tool (gpurun): SQ / GRBM / TCC counters of the Winograd forward on the image backbone's four 48-image shapes, one counter group per
# pass, launches spaced by a streaming pass as in the step.  usage: tools/pmc_wino.sh [outdir]
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=${1:-$ROOT/gpurun_out/pmc_wino}
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for shape in "48 256 256 16 44" "48 128 128 32 88" "48 64 64 64 176" "48 512 512 8 22"; do
  tag=$(echo $shape | tr ' ' 'x')
  echo "## shape (N C Co H W) = $shape" >> $OUT/summary.txt
  python $ROOT/tools/kbench_wino_one.py $shape 10 spaced nowgrad 2>/dev/null | grep shape >> $OUT/summary.txt
  python $ROOT/tools/kbench_wino_one.py $shape 10 nowgrad 2>/dev/null | grep shape >> $OUT/summary.txt
  i=0
  for grp in "SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS" "GRBM_GUI_ACTIVE SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT TCC_MISS"; do
    i=$((i+1))
    rocprofv3 --pmc $grp -d $OUT/$tag.p$i -- python $ROOT/tools/kbench_wino_one.py $shape 5 spaced nowgrad > $OUT/$tag.p$i.log 2>&1
    python $ROOT/tools/pmc_summary.py $(ls $OUT/$tag.p$i/*/*.db | head -1) wino_fwd 2>&1 | sed 's/avg value(KB)/avg value    /' >> $OUT/summary.txt
    rm -rf $OUT/$tag.p$i
  done
done
cat $OUT/summary.txt
