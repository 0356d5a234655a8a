for i in 1 2; do
DBEV_HIP_LIB=$PWD/tools/libdbev_hip_base.so python bench.py --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('base', d['ms_per_step'], d['roofline']['ms_per_step'], d['roofline']['other_hot_kernels']['c1x1_fwd']['ms_per_step'])"
python bench.py --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('new ', d['ms_per_step'], d['roofline']['ms_per_step'], d['roofline']['other_hot_kernels']['c1x1_fwd']['ms_per_step'])"
done
