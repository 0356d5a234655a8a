mkdir -p gpurun_out/r4
R=/root/repo
python bench.py > gpurun_out/r4/bench_line.json 2> gpurun_out/r4/bench_err.txt
python tools/kbench_wino.py > gpurun_out/r4/wino_vs_miopen.txt 2>&1
python tools/kbench_wino_v23.py > gpurun_out/r4/wino_v23.txt 2>&1
python tools/kbench_wino_bwd_streams.py > gpurun_out/r4/bwd_streams.txt 2>&1
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c -d $R/gpurun_out/r4/pmc_$c -- python $R/tools/kbench_wino_one.py 48 256 256 16 44 3 > /dev/null 2>&1
  (cd $R; python tools/pmc_summary.py $(ls gpurun_out/r4/pmc_$c/*/*.db | head -1) wino > gpurun_out/r4/pmc_wino_$c.txt 2>&1; rm -rf gpurun_out/r4/pmc_$c)
  rocprofv3 --pmc $c -d $R/gpurun_out/r4/pmc2_$c -- python $R/tools/kbench_wino_one.py 8 512 256 128 128 3 > /dev/null 2>&1
  (cd $R; python tools/pmc_summary.py $(ls gpurun_out/r4/pmc2_$c/*/*.db | head -1) wino > gpurun_out/r4/pmc_wino2_$c.txt 2>&1; rm -rf gpurun_out/r4/pmc2_$c)
done
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r4/ks -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $R/gpurun_out/r4/ks_line.json 2>/dev/null
cd $R; python tools/rocpd_summary.py $(ls gpurun_out/r4/ks/*/*.db | head -1) 40 > gpurun_out/r4/ks_summary.txt 2>&1; rm -rf gpurun_out/r4/ks
tail -c 300 gpurun_out/r4/bench_line.json
cat gpurun_out/r4/pmc_wino*_*.txt | grep "wino_fwd\|wino_wgrad2" | cut -c1-40,88-140
