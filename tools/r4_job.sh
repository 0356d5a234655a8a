mkdir -p gpurun_out/r4
(python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1) > gpurun_out/r4/bench_line_b.json
(timeout 900 python -m pytest tests/test_gpu_full_size.py tests/test_gpu_head_batch.py -x -q 2>&1 | tail -5) > gpurun_out/r4/fs_test.log
cut -c1-1900 gpurun_out/r4/bench_line_b.json; cat gpurun_out/r4/fs_test.log
