"""The boundary is a C ABI: include/flowz_hip.h must be valid C99 and usable from a plain C program."""
import os
import shutil
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
LIBDIR = os.path.join(ROOT, "zignal_amd", "lib")


def build(tmp_path):
    exe = str(tmp_path / "abi_smoke")
    subprocess.check_call(["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"),
                           os.path.join(HERE, "c", "abi_smoke.c"), "-L", LIBDIR, "-lflowz_hip",
                           f"-Wl,-rpath,{LIBDIR}", "-Wl,-rpath,/opt/rocm/lib", "-o", exe])
    return exe


@pytest.mark.skipif(shutil.which("gcc") is None, reason="needs gcc")
def test_header_is_c99_and_library_links_from_c(tmp_path):
    out = subprocess.run([build(tmp_path)], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "ok" in out.stdout


@pytest.mark.gpu
def test_c_program_runs_the_readme_integrator(tmp_path):
    out = subprocess.run([build(tmp_path), "run"], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "integrator: 1 3 6 10" in out.stdout
