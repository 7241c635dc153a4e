set -e
python -m pytest tests/test_gpu_regressions.py -x -q -m gpu 2>&1 | tail -5
