set -e
python scripts/tile2d_ab.py > gpurun_out/tile2d.txt 2>&1 || { tail -20 gpurun_out/tile2d.txt; exit 1; }
cat gpurun_out/tile2d.txt
