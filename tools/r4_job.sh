B=$PWD/tools/libdbev_hip_base.so
for i in 1 2 3; do
  DBEV_HIP_LIB=$B DBEV_WINO_FWD_V=2 python tools/kbench_wino_var.py 2>&1 | tail -1
  DBEV_WINO_FWD_V=2 python tools/kbench_wino_var.py 2>&1 | tail -1
done
