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


def test_upstream_shaped_prover_unit_runs_through_the_overlay_headers(tmp_path):
    """tests/cpu_build/prover_loop.cpp names only upstream's include paths and namespaces (nil::crypto3::algebra::multiexp,
    nil::crypto3::math::make_evaluation_domain, zk::commitments::kc_multiexp_with_mixed_addition); built with the overlay directory ahead
    on the include path and crypto3-shaped stand-in value types, its multiexp call reaches the GPU"""
    exe = str(tmp_path / "prover_loop")
    libdir = os.path.join(ROOT, "vote_saver_protocol_amd")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "include", "overlay"), "-I", os.path.join(ROOT, "tests", "cpu_build", "standin"),
                           os.path.join(ROOT, "tests", "cpu_build", "prover_loop.cpp"), "-o", exe, "-L", libdir, "-lvsp_hip", "-Wl,-rpath," + libdir])
    p = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stdout + p.stderr
    two_g = o.G1.mul(o.G1.gen, 2)
    assert ("2G.x[0] = %016x" % (two_g[0] & 0xFFFFFFFFFFFFFFFF)) in p.stdout
