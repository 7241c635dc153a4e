set -e
timeout -k 10 560 python scripts/soak.py 500 777 > gpurun_out/soak_r2_b.txt 2>&1 || { tail -30 gpurun_out/soak_r2_b.txt; exit 1; }
tail -2 gpurun_out/soak_r2_b.txt
