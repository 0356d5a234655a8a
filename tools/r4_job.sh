mkdir -p gpurun_out/r4
R=/root/repo
cd $R
python bench.py > gpurun_out/r4/bench_line.json 2> gpurun_out/r4/bench_err.txt
tail -c 600 gpurun_out/r4/bench_line.json
python tools/small_kernel_sites.py > gpurun_out/r4/small_sites.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r4/ks -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $R/gpurun_out/r4/ks_line.json 2>/dev/null
cd $R; python tools/rocpd_summary.py $(ls gpurun_out/r4/ks/*/*.db | head -1) > gpurun_out/r4/ks_summary.txt 2>&1; rm -rf gpurun_out/r4/ks
head -40 gpurun_out/r4/ks_summary.txt | cut -c1-60,100-160
