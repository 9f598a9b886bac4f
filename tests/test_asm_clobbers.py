"""The hand-ABI routines are entered by s_swappc_b64 from asm statements whose operand and clobber lists are the ONLY thing the compiler
knows about them.  tools/check_asm_clobbers.py parses the generated routine texts and the statements that
enter them: every register a routine writes must be declared there, and no operand declared input-only may be written."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_every_register_a_routine_writes_is_declared_where_it_is_entered():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_asm_clobbers.py")], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [x for x in r.stdout.splitlines() if "entered at" in x]
    assert len(lines) >= 8 and all("undeclared writes: none" in x and "input-only operands written: none" in x for x in lines)
    for label in ("vsp_mm_12", "vsp_mm_8", "vsp_mm28", "vsp_sq28", "vsp_mm28x2", "vsp_mm29", "vsp_mm29q", "vsp_acc28"):
        assert any(("entered at " + label + ":") in x for x in lines), label


import pytest


@pytest.mark.parametrize("unit,min_functions", [("msm_g1.hip", 20), ("ntt.hip", 3)])
def test_no_read_of_a_clobbered_register_after_a_routine_entry_in_the_generated_isa(tmp_path, unit, min_functions):
    """the other side of the interface (round 4, tools/check_call_sites.py): in the ISA the compiler generates for the 28-bit MSM kernels
    (msm_g1.hip, shipped flags) and for the transforms (ntt.hip: vsp_mm29 and vsp_mm29q, two register maps), no instruction reads a register
    that a routine entry before it clobbered and nothing has rewritten -- linear scan between labels"""
    hipcc = "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not installed")
    out = str(tmp_path / (unit + ".s"))
    csrc = os.path.join(ROOT, "vote_saver_protocol_amd", "csrc")
    subprocess.check_call([hipcc, "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wno-unused-function", "-Wno-pass-failed", "-Wno-unused-value",
                           "-Wno-unused-result", "-mllvm", "-enable-misched=0", "--cuda-device-only", "-S", unit, "-o", out], cwd=csrc, stderr=subprocess.DEVNULL)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_call_sites.py"), out], capture_output=True, text=True)
    assert r.returncode == 0 and "0 suspicious read(s)" in r.stdout, r.stdout[-2000:]
    assert int(r.stdout.split("check_call_sites: ")[1].split(" ")[0]) >= min_functions
