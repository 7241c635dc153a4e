set -e
timeout -k 10 300 python -m pytest tests/test_gpu_regressions.py -x -q -m gpu -k "grid_that_hits" 2>&1 | tail -3
