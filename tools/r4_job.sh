mkdir -p gpurun_out/r4
(timeout 600 python -m pytest tests/test_gpu_wino.py -x -q 2>&1 | tail -12) > gpurun_out/r4/wino_test.log
cat gpurun_out/r4/wino_test.log
