#!/bin/bash
# The round's closing GPU call, part 2 (after scripts/profile_round.sh): the sweeps behind profiles/<tag>_*_sweep.txt /
# _odd_widths.txt / _shape_cliffs.txt and the HBM counters of the operator families profile_round.sh does not cover.
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
PRE=${PROF_PREFIX:-fin_}
python3 scripts/backward_sweep.py > gpurun_out/${PRE}backward_sweep.txt 2>&1
python3 scripts/odd_width_probe.py > gpurun_out/${PRE}odd_widths.txt 2>&1
python3 scripts/width_sweep.py > gpurun_out/${PRE}width_sweep.txt 2>&1
python3 scripts/exp/devlens_cliffs.py > gpurun_out/${PRE}shape_cliffs.txt 2>&1
FAM=${PMC_FAMILIES:-n16,odd,bwd,w8}
python3 scripts/pmc_ops.py run --time --only=$FAM > gpurun_out/pmcx_time.json 2> gpurun_out/${PRE}pmcx_time.err
rm -rf gpurun_out/pmcx_fetch gpurun_out/pmcx_write
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmcx_fetch -o run -- python3 scripts/pmc_ops.py run --only=$FAM > gpurun_out/${PRE}pmcx_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmcx_write -o run -- python3 scripts/pmc_ops.py run --only=$FAM > gpurun_out/${PRE}pmcx_write.log 2>&1
tail -n +1 gpurun_out/${PRE}backward_sweep.txt gpurun_out/${PRE}odd_widths.txt
