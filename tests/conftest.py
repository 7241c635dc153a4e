import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    _ensure_built()


def _ensure_built():
    """The in-tree native pieces are build products (git-ignored): make them if a fresh checkout lacks them."""
    import shutil
    import subprocess
    lib = os.path.join(ROOT, 'torchrua_amd', 'librua_hip.so')
    if not os.path.exists(lib) and (shutil.which('hipcc') or os.path.exists('/opt/rocm/bin/hipcc')):
        subprocess.check_call(['make', '-C', os.path.join(ROOT, 'torchrua_amd', 'csrc'), '-j8'], stdout=subprocess.DEVNULL)
    if not os.path.exists(os.path.join(ROOT, 'oracle', 'librua_oracle.so')):
        subprocess.check_call(['make', '-C', os.path.join(ROOT, 'oracle')], stdout=subprocess.DEVNULL)


def pytest_collection_modifyitems(config, items):
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason='no GPU in this container')
    for item in items:
        if 'gpu' in item.keywords:
            item.add_marker(skip)
