#!/bin/bash
# One GPU call that produces everything scripts/summarize_prof.py condenses into profiles/<tag>_*:
#   kernel trace + stats of the bench command, FETCH_SIZE and WRITE_SIZE in their own passes (the pool refuses --pmc
#   together with the trace domains), plus kernel stats over every BASELINE config and the integer kernels.
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
PRE=${PROF_PREFIX:-prof2_}
rm -rf gpurun_out/${PRE}*
python3 bench.py --steps 20 --warmup 3 > gpurun_out/${PRE}bench.json 2> gpurun_out/${PRE}bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${PRE}stats -o runc -- python3 bench.py --no-cpu-baseline --steps 20 --warmup 3 > gpurun_out/${PRE}stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/${PRE}fetch -o runc -- python3 bench.py --no-cpu-baseline --steps 4 --warmup 2 > gpurun_out/${PRE}fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/${PRE}write -o runc -- python3 bench.py --no-cpu-baseline --steps 4 --warmup 2 > gpurun_out/${PRE}write.log 2>&1
python3 scripts/bench_configs.py > gpurun_out/${PRE}configs.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${PRE}cfgstats -o runc -- python3 scripts/bench_configs.py > gpurun_out/${PRE}cfgstats.log 2>&1
# HBM traffic per OPERATOR (index_buckets, pads, narrow rows ...): scripts/pmc_ops.py, one pass per counter; summarise
# with `python3 scripts/pmc_ops.py summarize <tag>` -> profiles/<tag>_pmc_ops.json
python3 scripts/pmc_ops.py run --time > gpurun_out/pmcops_time.json 2> gpurun_out/${PRE}pmcops_time.err
rm -rf gpurun_out/pmcops_fetch gpurun_out/pmcops_write
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmcops_fetch -o run -- python3 scripts/pmc_ops.py run > gpurun_out/${PRE}pmcops_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmcops_write -o run -- python3 scripts/pmc_ops.py run > gpurun_out/${PRE}pmcops_write.log 2>&1
# the per-dispatch CSVs are large: keep the stats, drop the traces
find gpurun_out/${PRE}stats gpurun_out/${PRE}cfgstats -name '*kernel_trace.csv' -delete
du -sh gpurun_out/${PRE}* | tail -20
cat gpurun_out/${PRE}bench.json
