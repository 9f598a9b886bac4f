#!/usr/bin/env python3
"""Per-kernel register / scratch / spill table of one HIP translation unit, from the compiler's own resource remarks.

    python3 tools/kernel_resources.py msm_g1.hip [-DX=1 ...] [--check]

Compiles the device pass only (`hipcc --cuda-device-only -S -Rpass-analysis=kernel-resource-usage` with the library's soundness flags, no
GPU needed) and prints one line per kernel: VGPRs, scratch bytes per lane, VGPR / SGPR spills, occupancy.

With --check the exit status is 1 when a G1 kernel of the 28-bit bucket reduction or merge (k_dimsum, k_dimbits, k_dimweight, k_merge*
over Fp28) uses scratch: since round 4 their exceptional case (equal x) is finished in the 28-bit form by the same register-only routine
entries as every other step, no function call is left in them and they need no stack.  (k_dimsum_mixed is exempt: it keeps the NEXT
bucket record in flight beside the current one, 42 spilled registers, and is faster for it: 250 against 297 us.)
Spills as such are NOT unsound -- the round-3 wrong results came from the machine scheduler, not from spilling; the lane-pair
G2 kernels spill a few dozen registers and are exact.  The table is a performance instrument: scratch traffic in a hot loop is worth knowing about."""
import os
import re
import subprocess
import sys
import tempfile

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "vote_saver_protocol_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wno-pass-failed", "-Wno-unused-value", "-Wno-unused-function",
         "-Wno-unused-result", "--cuda-device-only", "-S", "-Rpass-analysis=kernel-resource-usage", "-mllvm", "-enable-misched=0"]
# the G1 kernels of the 28-bit bucket reduction and merges: no call, no stack
GUARDED = re.compile(r"(k_dimsum|k_dimbits|k_dimweight|k_merge_a|k_merge2)<vsp::Fp28[,>]")
EXEMPT = re.compile(r"k_dimsum_mixed")
VGPR_CAP = int(os.environ.get("VSP_GUARD_VGPR_CAP", "256"))


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
    return dict(zip(names, out))


def resources(src, extra):
    with tempfile.TemporaryDirectory() as td:
        p = subprocess.run([HIPCC] + FLAGS + extra + [src, "-o", os.path.join(td, "out.s")], cwd=CSRC, capture_output=True, text=True)
        if p.returncode:
            sys.stderr.write(p.stderr[-4000:])
            raise SystemExit("compile failed")
    rows, cur = [], None
    for line in p.stderr.split("\n"):
        m = re.search(r"remark: (?:\s*)Function Name: (\S+)", line)
        if m:
            cur = {"name": m.group(1)}
            rows.append(cur)
            continue
        m = re.search(r"remark:\s+([A-Za-z /\[\]]+): (\S+) \[-Rpass", line)
        if m and cur is not None:
            cur[m.group(1).strip()] = m.group(2)
    return rows


def main():
    args = sys.argv[1:]
    check = "--check" in args
    args = [a for a in args if a != "--check"]
    src, extra = args[0], args[1:]
    rows = resources(src, extra)
    dm = demangle([r["name"] for r in rows])
    bad = []
    print("%5s %7s %6s %6s %4s  kernel" % ("VGPR", "scratch", "vspill", "sspill", "occ"))
    for r in rows:
        name = re.sub(r"vsp::\(anonymous namespace\)::", "", dm[r["name"]])
        name = re.sub(r"\(.*", "", name).replace("void ", "")
        v, sc = int(r.get("VGPRs", 0)), int(r.get("ScratchSize [bytes/lane]", 0))
        vs, ss = int(r.get("VGPRs Spill", 0)), int(r.get("SGPRs Spill", 0))
        mark = ""
        if GUARDED.search(name) and not EXEMPT.search(name) and (sc or v > VGPR_CAP):
            mark = "   <-- scratch in a G1 28-bit reduction kernel"
            bad.append(name)
        print("%5d %7d %6d %6d %4s  %s%s" % (v, sc, vs, ss, r.get("Occupancy [waves/SIMD]", "?"), name, mark))
    if check and bad:
        sys.stderr.write("kernel_resources: %d kernel(s) use scratch: %s\n" % (len(bad), ", ".join(bad)))
        return 1
    return 0


if __name__ == "__main__":
    sys.exit(main())
