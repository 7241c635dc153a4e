set -e
python -m pytest tests -x -q -m gpu 2>&1 | tail -3
python scripts/bench_configs.py 2>&1 | grep -E "ptr|idx|mask|buckets|cfg|pack\(\)|p.cat|roll|left"
