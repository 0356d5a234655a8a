mkdir -p gpurun_out/r4
(timeout 600 python -m pytest tests/test_gpu_wino.py -x -q 2>&1 | tail -4) > gpurun_out/r4/wino_test.log
(python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1) > gpurun_out/r4/bench_line_c.json
(DBEV_WINO_FWD_V=2 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-300) > gpurun_out/r4/bench_line_c_v2.json
cat gpurun_out/r4/wino_test.log; cut -c1-330 gpurun_out/r4/bench_line_c.json; echo; cut -c1-330 gpurun_out/r4/bench_line_c_v2.json
