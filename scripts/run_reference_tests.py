"""Run the REFERENCE's own test files, unmodified, against an implementation of the torchrua API.

    python scripts/run_reference_tests.py --against reference [pytest args]   # the reference itself (CPU ok)
    python scripts/run_reference_tests.py --against amd [pytest args]         # torchrua_amd (needs an MI355X)

Needs the reference checkout (default /root/reference; it is read, never copied) and supplies the missing
third-party helper `torchnyan` from tests/shim/.  `--against amd` makes `import torchrua` resolve to
torchrua_amd before the reference's tests are imported.  The reference never travels to the GPU box used in
this project, so there only `--against reference` has been exercised (it validates the shim); the property
tests in tests/test_gpu_properties.py mirror these files for the GPU."""
import argparse
import os
import sys

ap = argparse.ArgumentParser()
ap.add_argument('--against', choices=['reference', 'amd'], required=True)
ap.add_argument('--reference', default='/root/reference')
args, rest = ap.parse_known_args()

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault('PYTHONDONTWRITEBYTECODE', '1')
sys.dont_write_bytecode = True
sys.path.insert(0, os.path.join(ROOT, 'tests', 'shim'))
if args.against == 'amd':
    sys.path.insert(0, ROOT)
    import torchrua_amd
    torchrua_amd.install_as_torchrua()
    torchrua_amd.patch_tensor_indexing()      # the reference applies this patch on import (core/get.py:18)
sys.path.insert(0, args.reference)           # `tests.expected` and (for --against reference) `torchrua`

import pytest  # noqa: E402

sys.exit(pytest.main([os.path.join(args.reference, 'tests'), '-q', '-p', 'no:cacheprovider', '--rootdir', '/tmp',
                      '-o', 'cache_dir=/tmp/.pytest_cache_ref'] + rest))
