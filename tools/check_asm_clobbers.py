"""Static check of the hand-ABI routines (mont_asm_gfx950.h, accum28_asm_gfx950.h): every VGPR / SGPR that a routine's text WRITES must be
declared at the place the routine is entered (an output, a read-write operand, or a clobber of the call-site asm statement); registers it
only reads must be inputs or set by the call sequence.  Destination = the first operand of an instruction (plus the carry-out pair of
v_mad_u64_u32 / v_add_co etc.); s_swappc / s_setpc / s_branch / s_cbranch / s_waitcnt / s_nop / s_barrier / stores have none."""
import re, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

def regs_of(tok):
    out = set()
    for m in re.finditer(r"\b([vs])\[(\d+):(\d+)\]", tok):
        out |= {f"{m.group(1)}{i}" for i in range(int(m.group(2)), int(m.group(3)) + 1)}
    for m in re.finditer(r"\b([vs])(\d+)\b", tok):
        out.add(f"{m.group(1)}{m.group(2)}")
    if "vcc" in tok: out.add("vcc")
    return out

NO_DST = ("s_swappc", "s_setpc", "s_branch", "s_cbranch", "s_waitcnt", "s_nop", "s_barrier", "global_store", "ds_write", "buffer_store", "s_endpgm", "s_cmp", "v_cmp_", "s_bitcmp", ".p2align", "s_sleep")
def written(line):
    line = line.split("//")[0].strip()
    if not line or line.endswith(":") or line.startswith("."): return set()
    op, _, rest = line.partition(" ")
    if op.startswith(NO_DST):
        return {"vcc"} if op.startswith("v_cmp_") and not op.startswith("v_cmpx") and not rest.strip().startswith("s") else (regs_of(rest.split(",")[0]) if op.startswith("v_cmp_") else set())
    ops = [x.strip() for x in rest.split(",")]
    w = regs_of(ops[0]) if ops else set()
    if op.startswith(("v_mad_u64_u32", "v_mad_i64_i32", "v_add_co", "v_addc_co", "v_sub_co", "v_subb_co", "v_subrev_co", "v_subbrev_co")) and len(ops) > 1:
        w |= regs_of(ops[1])
    return w

def parse(path):
    txt = open(path).read()
    # holders: label -> set of written registers; call sites: label -> declared registers
    results = []
    for hm in re.finditer(r"void (\w*holder\w*)\(\)\s*\{(?:\s*static_assert\([^;]*;)?\s*asm volatile\((.*?)\n\s*:", txt, re.S):
        body = [re.sub(r'\\n\\t|\\n', '', s) for s in re.findall(r'"((?:[^"\\]|\\.)*)"', hm.group(2))]
        labels = [b[:-1] for b in body if re.match(r"^[A-Za-z_]\w*:$", b)]
        wr = set()
        for b in body: wr |= written(b)
        results.append((hm.group(1), labels, wr))
    sites = {}
    for sm in re.finditer(r"asm(?: volatile)?\(\s*\"(.*?)\);", txt, re.S):
        block = sm.group(0)
        lab = re.search(r"(vsp_\w+)@rel32@lo", block)
        if not lab: continue
        decl = set(); in_only = set()
        for m in re.finditer(r'"([=+]?)\{([vs]\d+)\}"', block):
            decl.add(m.group(2))
            if not m.group(1): in_only.add(m.group(2))
        tail = block.rsplit(":", 1)[-1]
        for m in re.finditer(r'"([vs]\d+|vcc|scc)"', tail): decl.add(m.group(1))
        sites.setdefault(lab.group(1), []).append((decl, in_only))
    return results, sites

bad = 0
for f in ("mont_asm_gfx950.h", "accum28_asm_gfx950.h"):
    res, sites = parse(os.path.join(ROOT, "vote_saver_protocol_amd", "csrc", f))
    for holder, labels, wr in res:
        entry = [l for l in labels if l in sites or any(l.startswith(k) or k.startswith(l) for k in sites)]
        for lab, decls in sites.items():
            if lab not in labels: continue
            for decl, in_only in decls:
                missing = sorted(r for r in wr if r not in decl and r not in ("vcc", "scc", "exec", "m0"))
                spoiled = sorted(r for r in wr if r in in_only)        # an operand the compiler believes unchanged
                print(f"{f}: {holder} entered at {lab}: writes {len(wr)} registers, declared {len(decl)}; undeclared writes: {missing if missing else 'none'}; "
                      f"input-only operands written: {spoiled if spoiled else 'none'}")
                bad += bool(missing) + bool(spoiled)
sys.exit(1 if bad else 0)
