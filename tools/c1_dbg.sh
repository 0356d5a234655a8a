for d in 0 1 2 4 3 5 6 7; do echo "== DBEV_C1_DBG=$d"; DBEV_C1_DBG=$d python tools/kbench_c1x1.py 3 2>&1 | grep "M=" | sed 's/MIOpen.*| ours/ours/' | cut -c1-150; done
