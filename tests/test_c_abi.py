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


ROCM = '/opt/rocm'


def _build_c(name: str, out_dir) -> str:
    exe = str(out_dir / name)
    libdir = os.path.join(ROOT, 'torchrua_amd')
    subprocess.run(['gcc', '-std=c99', '-O1', '-Wall', os.path.join(ROOT, 'tests', 'c', name + '.c'), '-o', exe,
                    '-I', os.path.join(ROOT, 'include'), '-I', os.path.join(ROCM, 'include'),
                    '-D__HIP_PLATFORM_AMD__', '-L', libdir, '-L', os.path.join(ROCM, 'lib'),
                    '-l:librua_hip.so', '-lamdhip64', '-lm', f'-Wl,-rpath,{libdir}', f'-Wl,-rpath,{ROCM}/lib'], check=True)
    return exe


def _have_toolchain() -> bool:
    return bool(shutil.which('gcc')) and os.path.exists(os.path.join(ROCM, 'include', 'hip', 'hip_runtime_api.h'))


@pytest.mark.skipif(not _have_toolchain() or not os.path.exists(os.path.join(ROOT, 'torchrua_amd', 'librua_hip.so')),
                    reason='no gcc / ROCm headers / built library here')
@pytest.mark.parametrize('name', ['abi_smoke', 'abi_pipeline'])
def test_plain_c_programs_compile_and_link(name, tmp_path):
    """No GPU needed: every symbol the C programs use resolves against librua_hip.so."""
    _build_c(name, tmp_path)


@pytest.mark.gpu
@pytest.mark.parametrize('name', ['abi_smoke', 'abi_pipeline'])
def test_plain_c_program_through_the_abi(name, tmp_path):
    """abi_smoke: the hand-checkable batch of SURVEY.md §8c.  abi_pipeline: 3 000 ragged sequences, some empty, through
    the host sort, pack, reduce, max with the reference's `initial`, its backward, and the scatter form."""
    if not _have_toolchain():
        pytest.skip('no gcc / ROCm headers on this box')
    exe = _build_c(name, tmp_path)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert 'gfx950' in out.stdout and ' 0 mismatches' in out.stdout
