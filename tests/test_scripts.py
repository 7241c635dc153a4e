"""scripts/ holds the probes behind the numbers in DESIGN.md / profiles/ (several had outlived the
kernels and flags they measured).  On the CPU: every script still byte-compiles and every `torchrua_amd` attribute it
names still exists; the GPU runs of the round (profiles/README.md) execute the ones that produce committed files."""
import ast
import os
import py_compile

import pytest

import torchrua_amd as ta
from torchrua_amd import _lib, _meta, _ops

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCRIPTS = sorted(f for f in os.listdir(os.path.join(ROOT, 'scripts')) if f.endswith('.py'))
ALIASES = {'ta': ta, 'torchrua_amd': ta, 'O': _ops, '_ops': _ops, 'M': _meta, '_meta': _meta, 'K': _lib, '_lib': _lib}


@pytest.mark.parametrize('name', SCRIPTS)
def test_script_compiles_and_names_live_symbols(name):
    path = os.path.join(ROOT, 'scripts', name)
    py_compile.compile(path, doraise=True)
    tree = ast.parse(open(path).read())
    for node in ast.walk(tree):
        if isinstance(node, ast.Attribute) and isinstance(node.value, ast.Name) and node.value.id in ALIASES:
            # only names the script imported under the conventional aliases are checked
            imported = any(isinstance(n, (ast.Import, ast.ImportFrom)) and any((a.asname or a.name.split('.')[-1]) == node.value.id
                                                                              for a in n.names) for n in ast.walk(tree))
            if imported:
                assert hasattr(ALIASES[node.value.id], node.attr), f'{name}: {node.value.id}.{node.attr} no longer exists'
