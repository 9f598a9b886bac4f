"""GPU: the C++ template shims (multiexp / evaluation_domain call shapes) run end to end through the C ABI."""
import os
import subprocess

import pytest

import bls12_381 as o
from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_cpp_shims_run(tmp_path):
    exe = str(tmp_path / "shim_check")
    libdir = os.path.join(ROOT, "vote_saver_protocol_amd")
    subprocess.check_call(["g++", "-std=c++17", "-O1", os.path.join(ROOT, "tests", "cpu_build", "shim_check.cpp"), "-o", exe,
                           "-L", libdir, "-lvsp_hip", "-Wl,-rpath," + libdir])
    p = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stdout + p.stderr
    two_g = o.G1.mul(o.G1.gen, 2)
    assert ("2G.x[0] = %016x" % (two_g[0] & 0xFFFFFFFFFFFFFFFF)) in p.stdout
    assert "kc_multiexp ok" in p.stdout
    assert "roundtrip ok" in p.stdout
