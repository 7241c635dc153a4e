set -e
python -m pytest tests/test_gpu_regressions.py tests/test_gpu_api.py tests/test_gpu_properties.py tests/test_gpu_golden.py -x -q -m gpu 2>&1 | tail -30
