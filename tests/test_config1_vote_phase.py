"""BASELINE config 1 counterpart.  The reference's live `cli` binary times one vote phase with
std::chrono and prints `Vote Phase Time_execution: <n>ms` (bin/cli/src/main.cpp:446-456; same shape in bin/cli/test/cli.cpp:69-84).
Circuit synthesis (blueprint) is out of scope, so the stand-in is the same flow around the part that is in scope: a SAVER-shaped
synthetic R1CS, one proof, the same line -- for the CPU restatement of the reference's prover everywhere, and on a GPU box for
vsp_groth16_prove on the same instance with identical proofs (tools/vote_phase_time.py)."""
import os
import re
import subprocess
import sys

import pytest

from conftest import ROOT

LINE = re.compile(r"Vote Phase Time_execution: (\d+)ms")


def run_tool(log_constraints):
    env = dict(os.environ)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "vote_phase_time.py"), "--log-constraints", str(log_constraints)],
                       capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 0, p.stdout + p.stderr
    return p.stdout


def test_vote_phase_timing_line_cpu_restatement(cref):
    out = run_tool(9)
    lines = [l for l in out.splitlines() if LINE.search(l)]
    assert lines and lines[0].startswith("[CPU restatement, 1 thread, 480 constraints]")
    assert int(LINE.search(lines[0]).group(1)) >= 0


@pytest.mark.gpu
def test_vote_phase_timing_line_gpu_same_proof(cref):
    out = run_tool(13)
    lines = [l for l in out.splitlines() if LINE.search(l)]
    assert len(lines) == 2 and lines[1].startswith("[MI355X, vsp_groth16_prove, same instance]"), out
    assert "proofs identical: True" in out
    assert int(LINE.search(lines[1]).group(1)) <= int(LINE.search(lines[0]).group(1))      # the CPU line is the slower one
