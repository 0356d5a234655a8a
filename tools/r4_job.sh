python -m pytest tests/test_gpu_wino.py tests/test_gpu_bn_act.py -x -q -m gpu 2>&1 | tail -3
