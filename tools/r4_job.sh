mkdir -p gpurun_out/r4
L=gpurun_out/r4/var.log; : > $L
for d in 0 1 2 3; do DBEV_WINO_DBG=$d python tools/kbench_wino_var.py 2>&1 | tail -1 >> $L; done
cd /tmp && export TMPDIR=/tmp
R=/root/repo
sh="8 512 512 64 64"; tag=v2
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU -d $R/gpurun_out/r4/pmc_a_$tag -- python $R/tools/kbench_wino_one.py $sh > $R/gpurun_out/r4/pmc_a_$tag.log 2>&1
rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SALU GRBM_GUI_ACTIVE -d $R/gpurun_out/r4/pmc_b_$tag -- python $R/tools/kbench_wino_one.py $sh > $R/gpurun_out/r4/pmc_b_$tag.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY -d $R/gpurun_out/r4/pmc_c_$tag -- python $R/tools/kbench_wino_one.py $sh > $R/gpurun_out/r4/pmc_c_$tag.log 2>&1
cd $R
for k in a b c; do python tools/pmc_summary.py $(ls gpurun_out/r4/pmc_${k}_$tag/*/*.db | head -1) wino_fwd > gpurun_out/r4/pmc_${k}_$tag.txt 2>&1; rm -rf gpurun_out/r4/pmc_${k}_$tag; done
cat $L; cat gpurun_out/r4/pmc_?_v2.txt | awk '{printf "%-12s %-28s %8s %16s %10s\n", substr($1,1,12), $(NF-3), $(NF-2), $(NF-1), $NF}'
