for d in 0 4 24 512 1024 1536 1564 0; do DBEV_WINO_FWD_V=2 DBEV_WINO_DBG=$d python tools/kbench_wino_var.py 2>&1 | tail -1; done
