python -m pytest tests/test_gpu_wino.py -x -q -m gpu 2>&1 | tail -2
B=$PWD/tools/libdbev_hip_base.so
for i in 1 2 3; do
  DBEV_HIP_LIB=$B DBEV_WINO_FWD_V=2 python tools/kbench_wino_var.py 2>&1 | tail -1
  DBEV_WINO_FWD_V=2 python tools/kbench_wino_var.py 2>&1 | tail -1
done
for shp in "8 512 512 64 64" "48 256 256 16 44"; do
  DBEV_HIP_LIB=$B python tools/kbench_wgrad_dbg.py $shp 2>&1 | tail -1
  python tools/kbench_wgrad_dbg.py $shp 2>&1 | tail -1
done
