set -e
python -m pytest tests -x -q -m gpu 2>&1 | tail -3
python scripts/cfg3_probe.py
