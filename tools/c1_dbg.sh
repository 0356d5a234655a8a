# ablation of dbev_conv1x1_forward: DBEV_C1_DBG bits: 1 no output stores, 2 no global loads, 4 no MFMAs, 8 no LDS staging writes, 16 no emit (stores + statistics)
for d in ${C1_DBG_LIST:-0 1 2 4 3 5 6 7}; do echo "== DBEV_C1_DBG=$d"; DBEV_C1_DBG=$d python tools/kbench_c1x1.py 3 2>&1 | grep "M=" | sed 's/MIOpen.*| ours/ours/' | cut -c1-150; done
