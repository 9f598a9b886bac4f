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
    assert len(lines) >= 7 and all("undeclared writes: none" in x and "input-only operands written: none" in x for x in lines)
    for label in ("vsp_mm_12", "vsp_mm_8", "vsp_mm28", "vsp_sq28", "vsp_mm28x2", "vsp_mm29", "vsp_acc28"):
        assert any(("entered at " + label + ":") in x for x in lines), label
