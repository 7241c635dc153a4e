set -e
python scripts/reduce_ab.py
