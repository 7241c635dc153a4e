set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python3 scripts/aten_free_trace.py
rm -rf gpurun_out/prof2_aten
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof2_aten -o runc -- python3 scripts/aten_free_trace.py > gpurun_out/prof2_aten.log 2>&1
find gpurun_out/prof2_aten -name '*kernel_trace.csv' -delete
python3 scripts/aten_free_trace.py --summarize
