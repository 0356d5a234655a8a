mkdir -p gpurun_out/r4
(timeout 600 python -m pytest tests/test_gpu_wino.py -x -q 2>&1 | tail -3) > gpurun_out/r4/wino_test.log
python tools/kbench_wino.py > gpurun_out/r4/kbench_wino7.log 2>&1
DBEV_WINO_DBG=32 python tools/kbench_wino.py > gpurun_out/r4/kbench_wino7_norot.log 2>&1
cat gpurun_out/r4/wino_test.log; tail -15 gpurun_out/r4/kbench_wino7.log | cut -c1-60; tail -15 gpurun_out/r4/kbench_wino7_norot.log | cut -c1-60
