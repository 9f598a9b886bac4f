#!/usr/bin/env python3
"""Static check of the compiler's code AROUND the hand-laid-out routines, on the generated ISA of one kernel or a whole translation unit:

    hipcc --cuda-device-only -S ... msm_g1.hip -o msm_g1.s;  python3 tools/check_call_sites.py msm_g1.s [kernel-name-substring]

For every entry `s_swappc_b64` into a routine of mont_asm_gfx950.h the registers the routine's call-site statement declares as clobbered
(not its outputs, not its preserved inputs) become POISON; a later read of a poisoned register before it is written again is reported.
The scan follows the text linearly and restarts the poison set at every label (a join point: what is live there is not known to a linear
scan), so it proves nothing about values carried around a loop -- it catches the local mistake: a value left in a clobbered register across
one call.  (tools/check_asm_clobbers.py checks the other side: that the declarations cover what the routines write.)"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def call_sites():
    txt = open(os.path.join(ROOT, "vote_saver_protocol_amd", "csrc", "mont_asm_gfx950.h")).read()
    out = {}
    for sm in re.finditer(r"asm(?: volatile)?\(\s*\"(.*?)\);", txt, re.S):
        block = sm.group(0)
        lab = re.search(r"(vsp_\w+)@rel32@lo", block)
        if not lab:
            continue
        outs = {m.group(1) for m in re.finditer(r'"[=+]\{([vs]\d+)\}"', block)}
        tail = block.rsplit(":", 1)[-1]
        clob = {m.group(1) for m in re.finditer(r'"([vs]\d+|vcc|scc)"', tail)}
        out[lab.group(1)] = (outs, clob - outs)
    return out


def regs(tok):
    out = set()
    for m in re.finditer(r"\b([vs])\[(\d+):(\d+)\]", tok):
        out |= {"%s%d" % (m.group(1), i) for i in range(int(m.group(2)), int(m.group(3)) + 1)}
    for m in re.finditer(r"\b([vs])(\d+)\b", tok):
        out.add(m.group(1) + m.group(2))
    if re.search(r"\bvcc\b", tok):
        out.add("vcc")
    return out


NODST = ("s_cmp", "s_bitcmp", "s_cbranch", "s_branch", "s_waitcnt", "s_barrier", "s_nop", "s_endpgm", "global_store", "ds_write", "s_setpc",
         "scratch_store", "buffer_store", "s_sleep", "s_trap", "s_sendmsg", "s_setprio", "s_icache", "s_dcache")


def scan(lines, sites, name):
    poison, pending, hits = set(), None, []
    for n, l in enumerate(lines, 1):
        c = l.split(";")[0].strip()
        if not c or c.startswith("."):
            if c.endswith(":"):
                poison = set()
            continue
        if c.endswith(":"):
            poison = set()
            continue
        op, _, rest = c.partition(" ")
        ops = [x.strip() for x in rest.split(",")]
        m = re.search(r"(vsp_\w+)@rel32@lo", c)
        if m:
            pending = m.group(1)
        if op == "s_swappc_b64":
            if pending in sites:
                poison = set(sites[pending][1]) - {"scc"}
            pending = None
            continue
        if op == "s_getpc_b64":
            poison -= regs(ops[0])
            continue
        if op.startswith(NODST):
            rd, wr = regs(rest), set()
            if op.startswith("s_cbranch_vcc"):
                rd.add("vcc")
        elif op.startswith("v_cmp") and not op.startswith("v_cmpx"):
            wr = regs(ops[0]); rd = regs(",".join(ops[1:]))
        elif op.startswith("v_cmpx"):
            wr, rd = set(), regs(rest)
        else:
            wr = regs(ops[0]); rd = regs(",".join(ops[1:]))
            if op.startswith(("v_mad_u64_u32", "v_mad_i64_i32", "v_add_co", "v_addc_co", "v_sub_co", "v_subb_co", "v_subrev_co", "v_subbrev_co")) and len(ops) > 1:
                wr |= regs(ops[1]); rd = regs(",".join(ops[2:]))
            if op.startswith(("v_addc_co", "v_subb_co", "v_subbrev_co", "v_cndmask")) and op.endswith("_e32"):
                rd.add("vcc")
            if op.startswith(("s_and_saveexec", "s_or_saveexec", "s_andn2_saveexec")):
                rd = regs(ops[1])
        hit = rd & poison
        if hit:
            hits.append((name, n, sorted(hit), c))
        poison -= wr
    return hits


def main():
    path, want = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
    sites = call_sites()
    txt = open(path).read()
    total, kernels = 0, 0
    for m in re.finditer(r"^(_Z\w+):.*?^\.Lfunc_end\d+:", txt, re.S | re.M):
        name = m.group(1)
        if want and want not in name:
            continue
        body = m.group(0)
        if "s_swappc_b64" not in body:
            continue
        kernels += 1
        for h in scan(body.split("\n"), sites, name):
            total += 1
            print("%s line %d: reads %s after a routine clobbered it: %s" % h)
    print("check_call_sites: %d function(s) with routine entries scanned, %d suspicious read(s)" % (kernels, total))
    return 1 if total else 0


if __name__ == "__main__":
    sys.exit(main())
