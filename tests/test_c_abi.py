"""include/rua.h is a C header, and librua_hip.so is usable from plain C with no Python or torch in the process."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_header_is_valid_c99():
    src = '#include "rua.h"\nint main(void) { rua_layout l; (void)l; return RUA_ABI_VERSION - 2; }\n'
    subprocess.run(['gcc', '-std=c99', '-Wall', '-Werror', '-pedantic', '-fsyntax-only', '-I',
                    os.path.join(ROOT, 'include'), '-x', 'c', '-'], input=src.encode(), check=True)


@pytest.mark.gpu
def test_plain_c_program_through_the_abi(tmp_path):
    rocm = '/opt/rocm'
    if not (shutil.which('gcc') and os.path.exists(os.path.join(rocm, 'include', 'hip', 'hip_runtime_api.h'))):
        pytest.skip('no gcc / ROCm headers on this box')
    exe = str(tmp_path / 'abi_smoke')
    libdir = os.path.join(ROOT, 'torchrua_amd')
    subprocess.run(['gcc', '-std=c99', '-O1', os.path.join(ROOT, 'tests', 'c', 'abi_smoke.c'), '-o', exe,
                    '-I', os.path.join(ROOT, 'include'), '-I', os.path.join(rocm, 'include'),
                    '-D__HIP_PLATFORM_AMD__', '-L', libdir, '-L', os.path.join(rocm, 'lib'),
                    '-l:librua_hip.so', '-lamdhip64', f'-Wl,-rpath,{libdir}', f'-Wl,-rpath,{rocm}/lib'], check=True)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert 'gfx950' in out.stdout and ' 0 mismatches' in out.stdout
