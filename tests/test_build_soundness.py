"""CPU: the build's soundness flags (vote_saver_protocol_amd/csrc/Makefile, SOUND).

Round 4 found the cause of the "source right, result wrong" events of rounds 1-3: this toolchain's pre-RA machine scheduler reorders the
copies out of an inline-asm statement's fixed output registers and leaves the live intervals inconsistent; LLVM's own machine verifier
says so ("Bad machine code ... After Machine Instruction Scheduler").  The library is therefore built with the scheduler off and with the
verifier ON in every build.  This file keeps both facts from rotting:
  * the Makefile carries the flags, and the compile commands `make` would run really contain them;
  * a translation unit that enters the routines verifies clean with the shipped flags (device pass; the full build does this for every
    translation unit -- here one small one, to keep the CPU suite short);
  * the same translation unit with the DEFAULT scheduler does NOT verify on the pinned toolchain -- the canary: if this ever starts to
    pass, the toolchain has changed and the flags (and the pin in the Makefile) are due for a review, not a silent drop."""
import os
import re
import subprocess

import pytest

from conftest import ROOT

CSRC = os.path.join(ROOT, "vote_saver_protocol_amd", "csrc")
HIPCC = "/opt/rocm/bin/hipcc"
BASE = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wno-unused-function", "-Wno-pass-failed", "-Wno-unused-value", "-Wno-unused-result",
        "--cuda-device-only", "-c", "capi.hip", "-o", "/dev/null"]


def test_makefile_carries_the_soundness_flags():
    mk = open(os.path.join(CSRC, "Makefile")).read()
    m = re.search(r"^SOUND \?= (.*)$", mk, re.M)
    assert m and "-enable-misched=0" in m.group(1) and "-verify-machineinstrs" in m.group(1)
    assert re.search(r"^CXXFLAGS \?= .*\$\(SOUND\)", mk, re.M)
    dry = subprocess.run(["make", "-n", "-B", "-C", CSRC, "BUILD=/tmp/vsp_dry_build", "OUT=/tmp/vsp_dry.so"], capture_output=True, text=True)
    cmds = [l for l in dry.stdout.split("\n") if "hipcc" in l and " -c " in l]
    assert len(cmds) >= 11 and all("-enable-misched=0" in c and "-verify-machineinstrs" in c for c in cmds), dry.stdout[-2000:]


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_shipped_flags_verify_clean_and_default_scheduling_does_not():
    shipped = subprocess.run([HIPCC] + BASE + ["-mllvm", "-enable-misched=0", "-mllvm", "-verify-machineinstrs"], cwd=CSRC, capture_output=True, text=True)
    assert shipped.returncode == 0 and "Bad machine code" not in shipped.stderr, shipped.stderr[-3000:]
    default = subprocess.run([HIPCC] + BASE + ["-mllvm", "-verify-machineinstrs"], cwd=CSRC, capture_output=True, text=True)
    bad = "Bad machine code" in default.stderr and "After Machine Instruction Scheduler" in default.stderr
    if not bad:
        pytest.fail("the default machine scheduler now verifies clean on capi.hip: the toolchain changed -- review csrc/Makefile SOUND and the pin "
                    "(run the verifier over every translation unit before touching the flags)")
