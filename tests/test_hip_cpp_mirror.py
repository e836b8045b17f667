"""The C++ host mirror (generalsreinforcementlearning_amd/host/vec_engine.hpp) exercised with the reference's
engine tests: tests/cpp/engine_kat.cpp is compiled against the header + libgvec_hip.so and run on the GPU."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "generalsreinforcementlearning_amd")


def _build(tmp_path):
    exe = str(tmp_path / "engine_kat")
    cmd = ["g++", "-std=c++17", "-O1", "-Wall", "-I", ROOT, os.path.join(ROOT, "tests", "cpp", "engine_kat.cpp"),
           "-L", PKG, "-lgvec_hip", f"-Wl,-rpath,{PKG}", "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib", "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    return exe


def test_cpp_mirror_program_builds(tmp_path):
    """CPU: the program compiles and links against the header and the in-tree library (no GPU call is made)."""
    if not os.path.exists(os.path.join(PKG, "libgvec_hip.so")):
        pytest.skip("libgvec_hip.so not built")
    _build(tmp_path)


@pytest.mark.gpu
def test_cpp_mirror_runs_reference_engine_tests(tmp_path):
    exe = _build(tmp_path)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "all expectations hold" in r.stdout
