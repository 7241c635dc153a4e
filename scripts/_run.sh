set -e
python -c "import __graft_entry__ as g; g.smoke()"
python -m pytest tests -x -q -m gpu 2>&1 | tail -2
