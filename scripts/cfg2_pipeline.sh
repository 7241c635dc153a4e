#!/bin/bash
# Developer probe: steady-state pack->reduce at the cfg2 shape (host-bound): ms/step and the host enqueue median.
for i in 1 2 3; do
  python bench.py --no-cpu-baseline --batch 4096 --hidden 256 --steps 400 --warmup 30 --trace-host 2> /tmp/cfg2.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('ms/step', d['ms_per_step'], 'kernels', d['pipeline']['kernel_ms'])"
  python -c "
xs=sorted(float(x) for x in open('/tmp/cfg2.err').read().split(':')[-1].split())
print('  host enqueue median %.3f ms' % xs[len(xs)//2])"
done
