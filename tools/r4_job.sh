python -m pytest tests/test_gpu_wino.py tests/test_gpu_full_size.py -x -q -m gpu 2>&1 | tail -2
for i in 1 2; do
python bench.py --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('new ', d['ms_per_step'], d['value'], d['roofline']['ms_per_step'])"
DBEV_WINO_HYBRID=0 python bench.py --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('off ', d['ms_per_step'], d['value'], d['roofline']['ms_per_step'])"
done
