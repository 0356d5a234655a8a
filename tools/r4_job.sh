mkdir -p gpurun_out/r4
R=/root/repo
for wl in voxel_teacher bevformer_distill bev_pool msda; do (python bench.py --workload $wl --no-cpu-baseline 2>&1 | tail -1 | cut -c1-700) > gpurun_out/r4/bench_$wl.json; done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r4/prof_step -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $R/gpurun_out/r4/prof_step_bench.json 2> $R/gpurun_out/r4/prof_step.err
cd $R
python tools/rocpd_summary.py $(ls gpurun_out/r4/prof_step/*/*.db | head -1) 120 > gpurun_out/r4/step_kernel_stats_wino3.txt 2>&1
rm -rf gpurun_out/r4/prof_step
(timeout 1700 python -m pytest tests -m gpu -q 2>&1 | tail -5) > gpurun_out/r4/gputests3.log
for wl in voxel_teacher bevformer_distill bev_pool msda; do cut -c1-420 gpurun_out/r4/bench_$wl.json; echo; done; cat gpurun_out/r4/gputests3.log
