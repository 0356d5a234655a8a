for e in "DBEV_WINO_FWD_V=3" "DBEV_WINO_FWD_V=2" "DBEV_WINO_HYBRID=0"; do
  echo "== $e"; env $e python -m pytest tests/test_gpu_wino.py tests/test_gpu_head_batch.py -q -m gpu 2>&1 | tail -3
done
