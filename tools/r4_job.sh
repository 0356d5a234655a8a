mkdir -p gpurun_out/r4
(python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-330) > gpurun_out/r4/bench_line_d.json
(timeout 1700 python -m pytest tests -m gpu -q -x 2>&1 | tail -4) > gpurun_out/r4/gputests4.log
cat gpurun_out/r4/bench_line_d.json; echo; cat gpurun_out/r4/gputests4.log
