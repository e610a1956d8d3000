"""Build and run the C++ front-end tests (g++ -std=gnu++14 for hex-float literals; the header itself is plain C++14; links libflowz_hip.so)."""
import os
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
BUILD = os.path.join(HERE, "cpp", "_build")


def build(name, extra=()):
    os.makedirs(BUILD, exist_ok=True)
    exe = os.path.join(BUILD, name)
    libdir = os.path.join(ROOT, "zignal_amd", "lib")
    cmd = ["g++", "-std=gnu++14", "-O1", "-Wall", "-Wextra", "-I", os.path.join(ROOT, "include"),
           os.path.join(HERE, "cpp", name + ".cpp"), "-o", exe, "-L", libdir, "-lflowz_hip",
           f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib", *extra]
    subprocess.check_call(cmd)
    return exe


def test_cpp_edsl_host_checks():
    out = subprocess.run([build("test_edsl_host")], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "all host EDSL checks passed" in out.stdout


@pytest.mark.gpu
def test_cpp_edsl_reference_tests_on_gpu():
    out = subprocess.run([build("test_edsl_gpu")], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "all GPU EDSL checks passed" in out.stdout


def test_cpp_block_api_compiles():
    build("test_block_api_gpu", ("-L/opt/rocm/lib", "-lamdhip64"))


@pytest.mark.gpu
def test_cpp_block_api_routes_agree_on_gpu():
    out = subprocess.run([build("test_block_api_gpu", ("-L/opt/rocm/lib", "-lamdhip64"))], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "all block API checks passed" in out.stdout


# (include/flowz/shard.hpp -- the C++ host's statistics reduction -- is the one header of the front end that needs the HIP runtime and RCCL)
TWO_DEV_FLAGS = ("-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-L/opt/rocm/lib", "-lrccl", "-lamdhip64", "-pthread")


def test_cpp_two_devices_compiles():
    build("test_two_devices_gpu", TWO_DEV_FLAGS)


@pytest.mark.gpu
def test_cpp_shards_on_several_devices_from_one_process():
    """SURVEY 8e from ONE host process: contiguous stream shards, a host thread + hipSetDevice + bank + stream per shard, the
    compiled program shared; every shard equals the single-device result.  On a one-GPU box both threads share device 0."""
    out = subprocess.run([build("test_two_devices_gpu", TWO_DEV_FLAGS)], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "all multi-device checks passed" in out.stdout and "statistics reduced over RCCL" in out.stdout
