"""CPU: the generated accumulation routine of the G1 multi-exponentiation (tools/gen_accum28_asm.py ->
vote_saver_protocol_amd/csrc/accum28_asm_gfx950.h, entered by k_accum28) run INSTRUCTION BY INSTRUCTION in an interpreter -- 64 lanes, EXEC, VCC, SCC,
64-bit addresses, global memory -- against the group law in big integers.

The routine is ~4 800 hand-allocated instructions with its own loop, masks and look-ahead loads; a mistake in it would be a silent
soundness bug of a prover (or a wave that never leaves its loop).  So, before it ever runs on a GPU:
  * the header in the tree is what the generator writes (no hand edits);
  * every lane's result equals sum +-P_i over its range of the sorted index, for ranges of length 0 .. 9, with signs, points at infinity
    in the table, lanes outside EXEC (untouched), and the loop ends for every mix of lengths;
  * equal-x pairs (a doubled point, a point and its negative) raise the lane's flag -- the caller then redoes the part generically;
  * the limb bounds fp28.h states for the accumulator hold on exit, and no register outside the declared map is ever written.
The GPU side is the MSM parity suite (tests/test_gpu_msm.py), which runs through this routine."""
import importlib.util
import os
import random
import re

import numpy as np

import bls12_381 as o
from conftest import ROOT

spec = importlib.util.spec_from_file_location("gen_accum28_asm", os.path.join(ROOT, "tools", "gen_accum28_asm.py"))
gen = importlib.util.module_from_spec(spec)
spec.loader.exec_module(gen)

P = o.P
W, N = 28, 14
MASK = (1 << W) - 1
RP = 1 << (W * N)
LANES = 64
U32, U64 = np.uint32, np.uint64


def tight(v):
    return [(v >> (W * i)) & MASK for i in range(N - 1)] + [v >> (W * (N - 1))]


def val(limbs):
    return sum(int(x) << (W * i) for i, x in enumerate(limbs))


class Machine:
    """the subset of gfx950 the routine uses, 64 lanes as numpy vectors"""

    def __init__(self, ins):
        self.ins = ins
        self.labels = {x[:-1]: i for i, x in enumerate(ins) if x.endswith(":")}
        self.v = np.zeros((256, LANES), U32)
        self.s = np.zeros(104, U32)
        self.exec = (1 << LANES) - 1
        self.vcc = 0
        self.scc = 0
        self.mem = {}                      # base address -> numpy uint32 array
        self.written = set()
        self.steps = 0

    # ---- operands
    def lane_mask(self, m):
        return np.array([(m >> i) & 1 for i in range(LANES)], dtype=bool)

    def active(self):
        return self.lane_mask(self.exec)

    def sget64(self, name):
        if name == "exec":
            return self.exec
        if name == "vcc":
            return self.vcc
        lo = int(re.match(r"s\[(\d+):", name).group(1))
        return int(self.s[lo]) | (int(self.s[lo + 1]) << 32)

    def sset64(self, name, val):
        val &= (1 << 64) - 1
        if name == "exec":
            self.exec = val
        elif name == "vcc":
            self.vcc = val
        else:
            lo = int(re.match(r"s\[(\d+):", name).group(1))
            self.s[lo], self.s[lo + 1] = val & 0xFFFFFFFF, val >> 32

    def src32(self, tok):
        """a 32-bit source operand as a vector of 64 lanes"""
        if tok.startswith("v"):
            return self.v[int(tok[1:])].copy()
        if re.fullmatch(r"s\d+", tok):
            return np.full(LANES, self.s[int(tok[1:])], U32)
        return np.full(LANES, int(tok, 0) & 0xFFFFFFFF, U32)

    def src64(self, tok):
        if tok.startswith("v["):
            lo = int(re.match(r"v\[(\d+):", tok).group(1))
            return self.v[lo].astype(U64) | (self.v[lo + 1].astype(U64) << U64(32))
        return np.full(LANES, self.sget64(tok), U64)

    def vset(self, tok, vals):
        act = self.active()
        if tok.startswith("v["):
            lo = int(re.match(r"v\[(\d+):", tok).group(1))
            vals = vals.astype(U64)
            self.v[lo][act] = (vals & U64(0xFFFFFFFF)).astype(U32)[act]
            self.v[lo + 1][act] = (vals >> U64(32)).astype(U32)[act]
            self.written.update((lo, lo + 1))
        else:
            r = int(tok[1:])
            self.v[r][act] = vals.astype(U32)[act]
            self.written.add(r)

    def mask_of(self, cond):
        act = self.active()
        return sum(1 << i for i in range(LANES) if act[i] and cond[i])

    def load(self, addr, ndw):
        for base, arr in self.mem.items():
            if base <= addr and addr + 4 * ndw <= base + 4 * len(arr):
                off = (addr - base) // 4
                assert (addr - base) % 4 == 0
                return arr[off:off + ndw]
        raise AssertionError("load outside every buffer: 0x%x" % addr)

    # ---- execution
    def run(self, max_steps=2_000_000):
        pc = 0
        ins = self.ins
        while pc < len(ins):
            self.steps += 1
            assert self.steps < max_steps, "the routine does not terminate"
            line = ins[pc]
            pc += 1
            if line.endswith(":"):
                continue
            op, _, rest = line.partition(" ")
            a = [x.strip() for x in rest.split(",")] if rest else []
            if op == "s_mov_b32":
                self.s[int(a[0][1:])] = int(a[1], 0) & 0xFFFFFFFF
            elif op == "s_mov_b64":
                self.sset64(a[0], self.sget64(a[1]))
            elif op in ("s_and_b64", "s_andn2_b64", "s_or_b64"):
                x, y = self.sget64(a[1]), self.sget64(a[2])
                r = x & y if op == "s_and_b64" else (x & ~y if op == "s_andn2_b64" else x | y)
                r &= (1 << 64) - 1
                self.sset64(a[0], r); self.scc = int(r != 0)
            elif op in ("s_cbranch_scc0", "s_cbranch_scc1"):
                if self.scc == int(op.endswith("1")):
                    pc = self.labels[a[0]]
            elif op in ("s_waitcnt", "s_nop"):
                pass
            elif op == "v_mov_b32":
                self.vset(a[0], self.src32(a[1]))
            elif op.startswith("v_cmp_"):
                x, y = self.src32(a[1]), self.src32(a[2])
                kind = op[6:]
                if kind == "lt_u32":
                    c = x < y
                elif kind == "ne_u32":
                    c = x != y
                elif kind == "eq_u32":
                    c = x == y
                elif kind == "gt_i32":
                    c = x.astype(np.int32) > y.astype(np.int32)
                else:
                    raise AssertionError(op)
                self.sset64(a[0], self.mask_of(c))
            elif op == "v_mad_u64_u32":
                assert a[1] == "vcc"
                r = self.src32(a[2]).astype(U64) * self.src32(a[3]).astype(U64) + self.src64(a[4])      # wraps mod 2^64 like the hardware
                self.vset(a[0], r)
                self.vcc = 0                                   # carry-out: never consumed by the routine (a consumer would see this)
            elif op == "v_mul_lo_u32":
                self.vset(a[0], (self.src32(a[1]).astype(U64) * self.src32(a[2]).astype(U64)) & U64(0xFFFFFFFF))
            elif op in ("v_and_b32", "v_or_b32", "v_xor_b32", "v_add_u32", "v_sub_u32", "v_min_u32"):
                x, y = self.src32(a[1]), self.src32(a[2])
                r = {"v_and_b32": lambda: x & y, "v_or_b32": lambda: x | y, "v_xor_b32": lambda: x ^ y, "v_add_u32": lambda: x + y,
                     "v_sub_u32": lambda: x - y, "v_min_u32": lambda: np.minimum(x, y)}[op]()
                self.vset(a[0], r)
            elif op == "v_or3_b32":
                self.vset(a[0], self.src32(a[1]) | self.src32(a[2]) | self.src32(a[3]))
            elif op == "v_lshl_add_u32":
                self.vset(a[0], (self.src32(a[1]) << U32(int(a[2]))) + self.src32(a[3]))
            elif op == "v_lshlrev_b32":
                self.vset(a[0], self.src32(a[2]) << U32(int(a[1])))
            elif op == "v_lshrrev_b32":
                self.vset(a[0], self.src32(a[2]) >> U32(int(a[1])))
            elif op == "v_lshrrev_b64":
                self.vset(a[0], self.src64(a[2]) >> U64(int(a[1])))
            elif op == "v_cndmask_b32":
                m = self.lane_mask(self.sget64(a[3]))
                self.vset(a[0], np.where(m, self.src32(a[2]), self.src32(a[1])))
            elif op in ("global_load_dword", "global_load_dwordx4"):
                ndw = 4 if op.endswith("x4") else 1
                off = 0
                last = a[2].split()
                assert last[0] == "off"
                if len(last) > 1:
                    off = int(last[1].split(":")[1])
                addr = self.src64(a[1])
                act = self.active()
                dst = int(re.match(r"v\[?(\d+)", a[0]).group(1))
                for lane in range(LANES):
                    if act[lane]:
                        self.v[dst:dst + ndw, lane] = self.load(int(addr[lane]) + off, ndw)
                self.written.update(range(dst, dst + ndw))
            else:
                raise AssertionError("unknown instruction: " + line)


def mont_point(pt):
    """affine point -> the table row: x, y in Montgomery form (R' = 2^392) on 28-bit limbs, 32 words"""
    if pt is None:
        return [0] * 32
    return tight(pt[0] * RP % P) + tight(pt[1] * RP % P) + [0, 0, 0, 0]


TABLE_BASE, SORTED_BASE = 0x7f12_3400_0000, 0x7f55_0000_1000


def run_routine(table_pts, ranges, entries, exec_mask=(1 << LANES) - 1):
    ins = gen.gen_body()
    m = Machine(ins)
    m.mem[TABLE_BASE] = np.array([w for pt in table_pts for w in mont_point(pt)], dtype=U32)
    m.mem[SORTED_BASE] = np.array(entries, dtype=U32)
    m.s[36], m.s[37] = TABLE_BASE & 0xFFFFFFFF, TABLE_BASE >> 32
    m.s[38], m.s[39] = SORTED_BASE & 0xFFFFFFFF, SORTED_BASE >> 32
    rng = np.random.default_rng(5)
    m.v[:] = rng.integers(0, 1 << 32, size=m.v.shape, dtype=np.uint64).astype(U32)          # garbage everywhere: nothing may be assumed zero
    before = m.v.copy()
    for lane, (a, b) in enumerate(ranges):
        m.v[157, lane], m.v[158, lane] = a, b
    before[157], before[158] = m.v[157], m.v[158]
    m.exec = exec_mask
    m.run()
    assert m.exec == exec_mask, "EXEC must be restored"
    assert max(m.written) < gen.NVGPR, "a register outside the declared map was written"
    return m, before


def expected_sum(table_pts, entries, a, b):
    acc = None
    for e in entries[a:b]:
        pt = table_pts[e & 0x7FFFFFFF]
        if pt is None:
            continue
        if e >> 31:
            pt = o.G1.neg(pt)
        if acc is not None and acc[0] == pt[0]:
            return "equal-x"
        acc = pt if acc is None else o.G1.add(acc, pt)
    return acc


def lane_result(m, lane):
    blk = lambda k: [int(m.v[14 * k + i, lane]) for i in range(N)]
    X, Y, ZZ, ZZZ = blk(0), blk(1), blk(2), blk(3)
    return X, Y, ZZ, ZZZ


def test_header_in_tree_is_the_generators_output():
    text = gen.header_text(gen.gen_body())
    assert open(gen.OUT).read() == text, "accum28_asm_gfx950.h is stale: run python tools/gen_accum28_asm.py"
    assert gen.NVGPR <= 168, "the routine must leave the kernel at three waves per SIMD"


def test_every_lane_sums_its_range_and_the_loop_ends():
    rnd = random.Random(11)
    pts = [o.G1.mul(o.G1.gen, rnd.randrange(1, o.R)) for _ in range(40)]
    pts[7] = None; pts[23] = None                                   # points at infinity in the table
    entries, ranges = [], []
    for lane in range(LANES):
        length = [0, 1, 2, 3, 5, 9, 1, 4][lane % 8] if lane < 56 else rnd.randrange(0, 7)
        a = len(entries)
        idx = rnd.sample(range(40), length) if length else []      # distinct points: no equal-x pair inside a range
        for k in idx:
            entries.append(k | (rnd.randrange(2) << 31))
        ranges.append((a, a + length))
    ranges[3] = (ranges[3][1], ranges[3][0])                        # start > end: an empty range, like start == end
    entries += [0] * 4
    exec_mask = ((1 << LANES) - 1) & ~(1 << 5) & ~(1 << 40)         # two lanes outside EXEC
    m, before = run_routine(pts, ranges, entries, exec_mask)
    for lane in range(LANES):
        if not (exec_mask >> lane) & 1:
            assert np.array_equal(m.v[:, lane], before[:, lane]), "a lane outside EXEC was touched"
            continue
        a, b = ranges[lane]
        want = expected_sum(pts, entries, a, b) if a < b else None
        X, Y, ZZ, ZZZ = lane_result(m, lane)
        assert int(m.v[161, lane]) == 0
        if want is None:
            assert not any(ZZ) and not any(X) and not any(Y) and not any(ZZZ) or not any(ZZ)
            continue
        zz, zzz = val(ZZ) % P, val(ZZZ) % P
        rinv = pow(RP, -1, P)                                          # out of the Montgomery form: ZZ^3 = ZZZ^2 holds for the plain values
        assert zz and pow(zz * rinv, 3, P) == pow(zzz * rinv, 2, P)
        assert (val(X) * pow(zz, -1, P) % P, val(Y) * pow(zzz, -1, P) % P) == want, lane
        # the invariants of fp28.h between mixed additions: tight limbs, X < 9.5p, Y < 8p, ZZ, ZZZ < 1.1p
        for limbs, bound in ((X, 9.5), (Y, 8.0), (ZZ, 1.1), (ZZZ, 1.1)):
            assert all(x <= MASK for x in limbs[:-1]) and val(limbs) < bound * P


def test_equal_x_pairs_raise_the_flag():
    rnd = random.Random(12)
    pts = [o.G1.mul(o.G1.gen, rnd.randrange(1, o.R)) for _ in range(10)]
    entries = [2, 2, 3,                        # lane 0: the running sum is P, P comes again: a doubling
               4, 5 | (1 << 31), 5,            # lane 1: 4 - 5 + 5 never meets equal x: an ordinary lane
               6, 6 | (1 << 31),               # lane 2: P then -P: a cancellation
               7, 8, 9,                        # lane 3: ordinary
               1, 2, 2]                        # lane 4: (P1 + P2) + P2: not equal x either
    ranges = [(0, 3), (3, 6), (6, 8), (8, 11), (11, 14)] + [(0, 0)] * (LANES - 5)
    m, _ = run_routine(pts, ranges, entries + [0] * 4)
    assert [int(m.v[161, k]) for k in range(5)] == [1, 0, 1, 0, 0]
    assert [expected_sum(pts, entries, *ranges[k]) == "equal-x" for k in range(5)] == [True, False, True, False, False]
    for lane in (1, 3, 4):
        a, b = ranges[lane]
        X, Y, ZZ, ZZZ = lane_result(m, lane)
        zz, zzz = val(ZZ) % P, val(ZZZ) % P
        assert (val(X) * pow(zz, -1, P) % P, val(Y) * pow(zzz, -1, P) % P) == expected_sum(pts, entries, a, b)


def test_worst_case_accumulator_values_stay_inside_the_bounds():
    """long ranges (the accumulator settles into its steady-state bounds) with every sign pattern, checked limb by limb on exit"""
    rnd = random.Random(13)
    pts = [o.G1.mul(o.G1.gen, rnd.randrange(1, o.R)) for _ in range(64)]
    entries, ranges = [], []
    for lane in range(LANES):
        a = len(entries)
        for k in rnd.sample(range(64), 12):
            entries.append(k | ((lane >> (k % 6)) & 1) << 31)
        ranges.append((a, a + 12))
    m, _ = run_routine(pts, ranges, entries + [0] * 4)
    for lane in range(LANES):
        a, b = ranges[lane]
        X, Y, ZZ, ZZZ = lane_result(m, lane)
        zz, zzz = val(ZZ) % P, val(ZZZ) % P
        assert (val(X) * pow(zz, -1, P) % P, val(Y) * pow(zzz, -1, P) % P) == expected_sum(pts, entries, a, b)
        for limbs, bound in ((X, 9.5), (Y, 1.5), (ZZ, 1.1), (ZZZ, 1.1)):
            assert all(x <= MASK for x in limbs[:-1]) and val(limbs) < bound * P
