mkdir -p gpurun_out/r4
cd /tmp && export TMPDIR=/tmp
R=/root/repo
: > $R/gpurun_out/r4/wg_trace.txt
for sh in "48 256 256 16 44" "8 512 512 64 64" "8 640 512 64 64" "8 64 64 128 128"; do
  tag=$(echo $sh | tr ' ' '_')
  rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r4/kt_$tag -- python $R/tools/kbench_wino_one.py $sh 10 > /dev/null 2>&1
  echo "### $sh" >> $R/gpurun_out/r4/wg_trace.txt
  (cd $R; python tools/rocpd_summary.py $(ls gpurun_out/r4/kt_$tag/*/*.db | head -1) 12 | cut -c1-60,110-160 >> gpurun_out/r4/wg_trace.txt; rm -rf gpurun_out/r4/kt_$tag)
done
cat $R/gpurun_out/r4/wg_trace.txt
