mkdir -p gpurun_out/r4
(python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-400) > gpurun_out/r4/bench_wino2.log
(DBEV_WINO_WGRAD=0 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-400) > gpurun_out/r4/bench_wino2_nowg.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/r4/prof_step -- python /root/repo/bench.py --steps 5 --warmup 2 --no-cpu-baseline > /root/repo/gpurun_out/r4/prof_step_bench.json 2> /root/repo/gpurun_out/r4/prof_step.err
cd /root/repo
python tools/rocpd_summary.py $(ls gpurun_out/r4/prof_step/*/*.db | head -1) 120 > gpurun_out/r4/step_kernel_stats_wino2.txt 2>&1
rm -rf gpurun_out/r4/prof_step
cat gpurun_out/r4/bench_wino2.log gpurun_out/r4/bench_wino2_nowg.log
