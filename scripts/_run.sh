set -e
python bench.py --steps 24 --warmup 3 --trace-host --no-cpu-baseline > gpurun_out/bench_b.json 2> gpurun_out/bench_b.err
grep -E "kernel ms|host enqueue" gpurun_out/bench_b.err
python scripts/bench_configs.py 2>&1 | grep -E "c.pack\(\)|reduce_sum\(p\)" | tail -2
