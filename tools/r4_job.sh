for i in 1 2; do
(cd build_ab/head && python bench.py --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('head', d['ms_per_step'], d['roofline']['ms_per_step'])")
python bench.py --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('new ', d['ms_per_step'], d['roofline']['ms_per_step'])"
done
