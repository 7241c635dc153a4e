set -e
python -m pytest tests -x -q -m gpu 2>&1 | tail -3
python scripts/split_probe.py 2>&1 | grep -E "GB|split=None|split=0"
python scripts/skew_probe.py 2>&1 | tail -12
