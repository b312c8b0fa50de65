"""The C-ABI used from plain C (no Python, no torch in the loop): tests/c_abi/abi_smoke.c is compiled
against include/subpixal_hip.h, linked with libsubpixal_hip.so and run on the GPU."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c_program_against_the_shared_library(tmp_path):
    exe = str(tmp_path / 'abi_smoke')
    csrc = os.path.join(ROOT, 'subpixal_amd', 'csrc')
    # gcc, not hipcc: the caller's side of the boundary is plain C99 + the HIP runtime's C API
    cmd = ['gcc', '-std=c99', '-Wall', '-D__HIP_PLATFORM_AMD__', '-O1',
           os.path.join(ROOT, 'tests', 'c_abi', 'abi_smoke.c'), '-I', os.path.join(ROOT, 'include'),
           '-I', '/opt/rocm/include', '-L', csrc, '-L', '/opt/rocm/lib', '-lsubpixal_hip', '-lamdhip64', '-lm',
           '-Wl,-rpath,' + csrc, '-Wl,-rpath,/opt/rocm/lib', '-o', exe]
    subprocess.check_call(cmd)
    res = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    sys.stdout.write(res.stdout)
    sys.stderr.write(res.stderr)
    assert res.returncode == 0, res.stdout + res.stderr
    assert 'C ABI smoke OK' in res.stdout
