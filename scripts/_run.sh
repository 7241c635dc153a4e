set -e
python -m pytest tests -x -q -m gpu 2>&1 | tail -4
python scripts/width_sweep.py > gpurun_out/width_sweep.txt 2>&1
cat gpurun_out/width_sweep.txt
python scripts/bench_configs.py > gpurun_out/configs.txt 2>&1
cat gpurun_out/configs.txt
