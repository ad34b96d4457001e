"""The C ABI from plain C: tests/c/abi_smoke.c is compiled with gcc against include/brainevent_amd.h (the header must be valid
C11 and every entry point it uses must link) and, on a GPU box, run — the INTEGRATION.md call sequence with no Python and no
torch in the process."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, 'tests', 'c', 'abi_smoke.c')
EXE = os.path.join(ROOT, 'tests', 'c', 'abi_smoke')
LIBDIR = os.path.join(ROOT, 'brainevent_amd', 'lib')


def build_program():
    gcc = shutil.which('gcc')
    assert gcc, 'gcc not found'
    cmd = [gcc, '-std=c11', '-O1', '-Wall', '-D__HIP_PLATFORM_AMD__', '-I', os.path.join(ROOT, 'include'), '-I', '/opt/rocm/include',
           SRC, '-L', LIBDIR, '-lbrainevent_amd', '-L', '/opt/rocm/lib', '-lamdhip64', '-lm',
           f'-Wl,-rpath,{LIBDIR}', '-Wl,-rpath,/opt/rocm/lib', '-o', EXE]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout
    return EXE


def test_header_compiles_as_c_and_the_program_links():
    exe = build_program()
    assert os.path.exists(exe)


@pytest.mark.gpu
def test_c_program_runs_the_integration_sequence():
    exe = build_program()
    r = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    print(r.stdout)
    assert r.returncode == 0 and 'abi_smoke: all ok' in r.stdout, r.stdout
