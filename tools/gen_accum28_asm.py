#!/usr/bin/env python3
"""Generates vote_saver_protocol_amd/csrc/accum28_asm_gfx950.h: the WHOLE bucket-part accumulation of the G1 multi-exponentiation
(kernel k_accum28, msm_impl.inc) as one hand-allocated routine, so that the kernel fits 168 VGPRs and runs THREE waves per SIMD.

Why.  tools/ubench_madd28.hip (profiles/r3_ubench_madd28.txt) times the real 28-bit routines in shader cycles at 1..4 waves per SIMD:
a wave that shares its SIMD with ONE other wave takes 1.8 x as long per product as a wave alone, and with TWO others still 1.8 x --
2052 cycles of SIMD time per product at two waves per SIMD, 1370 at three.  The compiler's register allocation around the fixed-register
product routines needs 227-238 VGPRs for a mixed addition (two waves per SIMD), and 92-139 spilled registers when held to 168, which
costs more than the third wave brings.  The mixed addition needs 11 field elements of 14 limbs live at its widest point plus a 64-bit
column accumulator: 157 registers, if every product reads its operands where they are and writes its result where it will be needed.
That is what this generator does: every product is emitted INLINE over the register blocks given to it (no marshalling moves, no
fixed operand registers), the lazy subtractions, carry passes and zero tests between them are emitted limb by limb, and the loop
around them (gather of the next point while the current addition runs, sign of the digit, points at infinity, the first point of a
part, the equal-x exit) is written in the same routine with EXEC masks.  The column schedule of a product is that of
tools/gen_mont_asm.py body28 (bit-identical results), the formulas and bounds those of fp28.h madd28.

Register map.  Field elements live in BLOCKS of 14 consecutive VGPRs:
    X Y ZZ ZZZ (the accumulator, the routine's result)   x y (the gathered point: 28 consecutive registers, 7 dwordx4 loads)
    T1 .. T5 (temporaries)                              v[154:155] column accumulator, v156 m_k scratch
    v157 i, v158 end (this lane's range in the sorted index), v159 entry of the current point, v160 entry of the next one,
    v161 flag (1: an equal-x pair was met, the part must be redone by the generic kernel), v[162:163] address, v164..v166 scratch
    s0..s13 p, s14 -1/p mod 2^28, s15 2^28 - 1, s16..s29 the redundant 32p (FP28_K32_L1), s[30:31] return address,
    s[36:37] table, s[38:39] sorted index, s[40:41] EXEC at entry, s[42:43] live lanes, s[44:49] masks   (s32..s35 reserved by the ABI)
The routine is entered by s_swappc_b64 from the trampoline at the end of the header (private convention, exact clobber list), exactly as
the product routines of mont_asm_gfx950.h are.

tests/test_accum28_asm.py runs the generated instruction stream in an interpreter (64 lanes, EXEC, VCC, SCC, memory) against the group
law in big integers -- on the CPU, before the kernel ever runs on a GPU.

Run:  python tools/gen_accum28_asm.py
"""
import os

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "vote_saver_protocol_amd", "csrc", "accum28_asm_gfx950.h")

P381 = 0x1A0111EA397FE69A4B1BA7B6434BACD764774B84F38512BF6730D2A0F6B0F6241EABFFFEB153FFFFB9FEFFFFFFFFAAAB
N, W = 14, 28
MASK = (1 << W) - 1


def limbs28(v):
    return [(v >> (W * i)) & MASK for i in range(N - 1)] + [v >> (W * (N - 1))]


def redundant(c, lend):
    k = limbs28(c * P381)
    out = [k[0] + lend * (1 << W)] + [k[i] + lend * (1 << W) - lend for i in range(1, N - 1)] + [k[N - 1] - lend]
    assert sum(x << (W * i) for i, x in enumerate(out)) == c * P381 and all(0 <= x < (1 << 32) for x in out)
    return out


P28 = limbs28(P381)
INV28 = (-pow(P381, -1, 1 << W)) % (1 << W)
ONE28 = limbs28((1 << (W * N)) % P381)
K8_L1, K8_L4, K32_L1 = redundant(8, 1), redundant(8, 4), redundant(32, 1)

# ---- register map
BLK = {name: i for i, name in enumerate(["X", "Y", "ZZ", "ZZZ", "x", "y", "T1", "T2", "T3", "T4", "T5"])}


def R(block, i):
    return f"v{14 * BLK[block] + i}"


ACC_LO, ACC_HI, ACC, MK = "v154", "v155", "v[154:155]", "v156"
V_I, V_END, V_E, V_EN, V_FLAG = "v157", "v158", "v159", "v160", "v161"
ADDR_LO, ADDR_HI, ADDR = "v162", "v163", "v[162:163]"
AUX0, AUX1, AUX2 = "v164", "v165", "v166"
NVGPR = 167
SP = lambda i: f"s{i}"                   # p limbs
S_INV, S_MASK = "s14", "s15"
SK32 = lambda i: f"s{16 + i}"            # FP28_K32_L1 limbs
# (s32..s35 are the stack, frame and base pointers of the function ABI: reserved, never touched)
S_TABLE, S_SORTED, S_EXEC0, S_LIVE, S_M0, S_M1, S_M2 = "s[36:37]", "s[38:39]", "s[40:41]", "s[42:43]", "s[44:45]", "s[46:47]", "s[48:49]"
NSGPR = 50
S_RESERVED = (30, 31, 32, 33, 34, 35)


class Emit:
    def __init__(self):
        self.ins = []

    def __call__(self, text):
        self.ins.append(text)

    # ---- products (column schedule of gen_mont_asm.body28; operands preserved; dst distinct from every operand)
    def product(self, dst, pairs, sqr_of=None):
        """dst = sum of a*b over `pairs` (one or two (a, b) block pairs), times 2^-392 mod p, lazily reduced.
        sqr_of = (a, a2): the square of block a given a2 = 2a (off-diagonal products once against the doubled operand)."""
        e = self
        assert all(dst not in pr for pr in pairs) and (sqr_of is None or dst not in sqr_of)
        e(f"v_mov_b32 {ACC_LO}, 0"); e(f"v_mov_b32 {ACC_HI}, 0")
        for k in range(2 * N - 1):
            for i in range(max(0, k - N + 1), min(k, N - 1) + 1):
                if sqr_of is not None:
                    a, a2 = sqr_of
                    if i < k - i:
                        e(f"v_mad_u64_u32 {ACC}, vcc, {R(a, i)}, {R(a2, k - i)}, {ACC}")
                    elif i == k - i:
                        e(f"v_mad_u64_u32 {ACC}, vcc, {R(a, i)}, {R(a, i)}, {ACC}")
                    continue
                for a, b in pairs:
                    e(f"v_mad_u64_u32 {ACC}, vcc, {R(a, i)}, {R(b, k - i)}, {ACC}")
            for i in (range(0, k) if k < N else range(k - N + 1, N)):
                e(f"v_mad_u64_u32 {ACC}, vcc, {R(dst, i)}, {SP(k - i)}, {ACC}")
            if k < N:
                e(f"v_mul_lo_u32 {MK}, {ACC_LO}, {S_INV}")
                e(f"v_and_b32 {R(dst, k)}, {S_MASK}, {MK}")
                e(f"v_mad_u64_u32 {ACC}, vcc, {R(dst, k)}, {SP(0)}, {ACC}")
            else:
                e(f"v_and_b32 {R(dst, k - N)}, {S_MASK}, {ACC_LO}")
            e(f"v_lshrrev_b64 {ACC}, {W}, {ACC}")
        e(f"v_mov_b32 {R(dst, N - 1)}, {ACC_LO}")

    def mul(self, dst, a, b):
        self.product(dst, [(a, b)])

    def mul2(self, dst, a, b, c, d):
        self.product(dst, [(a, b), (c, d)])

    def sqr(self, dst, a, a2):
        for i in range(N):
            self(f"v_lshlrev_b32 {R(a2, i)}, 1, {R(a, i)}")
        self.product(dst, [], sqr_of=(a, a2))

    # ---- lazy linear operations, limb by limb (32-bit wrapping arithmetic: the final value is in range, see fp28.h)
    def sub(self, dst, a, k, b):
        """dst = a + K - b; K: 's' = the SGPR copy of FP28_K32_L1, or a list of 14 literals"""
        for i in range(N):
            self(f"v_sub_u32 {R(dst, i)}, {R(a, i)}, {R(b, i)}")
            self(f"v_add_u32 {R(dst, i)}, {SK32(i) if k == 's' else hex(k[i])}, {R(dst, i)}")

    def neg(self, dst, k, b):
        for i in range(N):
            self(f"v_sub_u32 {R(dst, i)}, {SK32(i) if k == 's' else hex(k[i])}, {R(b, i)}")

    def norm(self, blk):
        """carry pass in place: loose -> tight, same value"""
        c = AUX0
        self(f"v_lshrrev_b32 {c}, {W}, {R(blk, 0)}"); self(f"v_and_b32 {R(blk, 0)}, {S_MASK}, {R(blk, 0)}")
        for i in range(1, N - 1):
            self(f"v_add_u32 {R(blk, i)}, {R(blk, i)}, {c}")
            self(f"v_lshrrev_b32 {c}, {W}, {R(blk, i)}")
            self(f"v_and_b32 {R(blk, i)}, {S_MASK}, {R(blk, i)}")
        self(f"v_add_u32 {R(blk, N - 1)}, {R(blk, N - 1)}, {c}")

    def copy(self, dst, src):
        for i in range(N):
            self(f"v_mov_b32 {R(dst, i)}, {R(src, i)}")

    def or_all(self, out, regs):
        """out = OR of the registers (out may not be among them)"""
        regs = list(regs)
        self(f"v_or_b32 {out}, {regs[0]}, {regs[1]}")
        rest = regs[2:]
        while rest:
            if len(rest) >= 2:
                self(f"v_or3_b32 {out}, {out}, {rest[0]}, {rest[1]}"); rest = rest[2:]
            else:
                self(f"v_or_b32 {out}, {out}, {rest[0]}"); rest = rest[1:]


def blk_regs(b):
    return [R(b, i) for i in range(N)]


def gen_body():
    """the routine, as a list of instruction / label lines"""
    e = Emit()
    # ---------------------------------------------------------------- constants
    for i in range(N):
        e(f"s_mov_b32 {SP(i)}, 0x{P28[i]:x}")
    e(f"s_mov_b32 {S_INV}, 0x{INV28:x}"); e(f"s_mov_b32 {S_MASK}, 0x{MASK:x}")
    for i in range(N):
        e(f"s_mov_b32 {SK32(i)}, 0x{K32_L1[i]:x}")
    e(f"s_mov_b64 {S_EXEC0}, exec")
    for b in ("X", "Y", "ZZ", "ZZZ"):
        for i in range(N):
            e(f"v_mov_b32 {R(b, i)}, 0")
    e(f"v_mov_b32 {V_FLAG}, 0")
    # live lanes: i < end
    e(f"v_cmp_lt_u32 vcc, {V_I}, {V_END}")
    e(f"s_and_b64 {S_LIVE}, {S_EXEC0}, vcc")
    e("s_cbranch_scc0 .Lvsp_acc28_exit")
    e(f"s_mov_b64 exec, {S_LIVE}")
    # entry of point i, entry of point min(i + 1, end - 1), row of point i
    e(f"v_mad_u64_u32 {ADDR}, vcc, {V_I}, 4, {S_SORTED}")
    e(f"global_load_dword {V_E}, {ADDR}, off")
    emit_next_entry_load(e)
    e("s_waitcnt vmcnt(1)")                                   # the first of the two loads: the current entry
    emit_row_load(e)
    e(".Lvsp_acc28_loop:")
    # ---------------------------------------------------------------- top of the loop: EXEC = live lanes
    e("s_waitcnt vmcnt(0)")
    # the point at infinity is the all-zero row; the accumulator at infinity has ZZ = 0
    e.or_all(AUX1, blk_regs("x") + blk_regs("y"))
    e(f"v_cmp_ne_u32 {S_M0}, 0, {AUX1}")                      # M0: a finite point
    e.or_all(AUX2, blk_regs("ZZ"))
    e(f"v_cmp_eq_u32 {S_M1}, 0, {AUX2}")                      # M1: accumulator at infinity
    # the digit's sign (bit 31 of the entry): y <- 8p - y
    e(f"v_cmp_gt_i32 vcc, 0, {V_E}")
    for i in range(N):
        e(f"v_sub_u32 {AUX0}, 0x{K8_L1[i]:x}, {R('y', i)}")
        e(f"v_cndmask_b32 {R('y', i)}, {R('y', i)}, {AUX0}, vcc")
    # ---- lanes whose accumulator is still at infinity take the point: X = x, Y = +-y (tight), ZZ = ZZZ = 1
    e(f"s_and_b64 {S_M2}, {S_M0}, {S_M1}")
    e(f"s_and_b64 exec, {S_M2}, {S_LIVE}")
    e("s_cbranch_scc0 .Lvsp_acc28_noinit")
    e.copy("X", "x"); e.copy("Y", "y"); e.norm("Y")
    for b in ("ZZ", "ZZZ"):
        for i in range(N):
            e(f"v_mov_b32 {R(b, i)}, 0x{ONE28[i]:x}")
    e(".Lvsp_acc28_noinit:")
    # ---- the others add it: body = live & finite point & finite accumulator
    e(f"s_andn2_b64 {S_M2}, {S_M0}, {S_M1}")
    e(f"s_and_b64 {S_M0}, {S_M2}, {S_LIVE}")                  # M0 := body lanes (kept until the end of the iteration)
    e(f"s_mov_b64 exec, {S_M0}")
    e("s_cbranch_scc0 .Lvsp_acc28_advance")
    e.mul("T1", "x", "ZZ"); e.sub("T1", "T1", "s", "X"); e.norm("T1")          # P = x ZZ + 32p - X
    e.mul("T2", "y", "ZZZ"); e.sub("T2", "T2", "s", "Y"); e.norm("T2")         # R = y ZZZ + 32p - Y
    e(".Lvsp_acc28_advance:")
    # ---- x and y are dead in every live lane: step to the next point and gather it while the products below run
    e(f"s_mov_b64 exec, {S_LIVE}")
    e(f"v_add_u32 {V_I}, 1, {V_I}")
    e(f"v_mov_b32 {V_E}, {V_EN}")
    emit_next_entry_load(e)
    emit_row_load(e)
    e(f"s_mov_b64 exec, {S_M0}")
    e(f"s_and_b64 {S_M2}, {S_M0}, {S_M0}")                    # SCC = any body lane
    e("s_cbranch_scc0 .Lvsp_acc28_next")
    e.sqr("T4", "T1", "T3")                                                    # PP = P^2
    # equal x (PP = 0 mod p: PP is 0 or p): flag the lane and take it out of the live set; its accumulator is void from here on
    e.or_all(AUX1, blk_regs("T4"))
    e(f"v_xor_b32 {AUX2}, {SP(0)}, {R('T4', 0)}")
    for i in range(1, N):
        e(f"v_xor_b32 {AUX0}, {SP(i)}, {R('T4', i)}")
        e(f"v_or_b32 {AUX2}, {AUX2}, {AUX0}")
    e(f"v_cmp_eq_u32 {S_M1}, 0, {AUX1}")
    e(f"v_cmp_eq_u32 {S_M2}, 0, {AUX2}")
    e(f"s_or_b64 {S_M1}, {S_M1}, {S_M2}")
    e(f"v_cndmask_b32 {V_FLAG}, {V_FLAG}, 1, {S_M1}")
    e(f"v_cndmask_b32 {V_END}, {V_END}, 0, {S_M1}")
    e.mul("T3", "T1", "T4")                                                    # PPP = P PP
    e.mul("T1", "X", "T4")                                                     # Q = X PP
    e.mul("T5", "ZZ", "T4")                                                    # ZZ' = ZZ PP
    e.mul("T4", "ZZZ", "T3")                                                   # ZZZ' = ZZZ PPP
    e.sqr("ZZZ", "T2", "ZZ")                                                   # R^2 (2R in the dead ZZ block)
    for i in range(N):
        e(f"v_lshl_add_u32 {R('ZZ', i)}, {R('T1', i)}, 1, {R('T3', i)}")        # PPP + 2Q
    e.sub("X", "ZZZ", K8_L4, "ZZ"); e.norm("X")                                # X3 = R^2 + 8p - (PPP + 2Q)
    e.sub("ZZ", "T1", "s", "X")                                                # Q + 32p - X3
    e.neg("ZZZ", "s", "Y")                                                     # 32p - Y
    e.mul2("Y", "T2", "ZZ", "ZZZ", "T3")                                       # Y3 = R (Q - X3) + (32p - Y) PPP
    e.copy("ZZ", "T5"); e.copy("ZZZ", "T4")
    e(".Lvsp_acc28_next:")
    # ---- next iteration while any lane has points left
    e(f"s_mov_b64 exec, {S_LIVE}")
    e(f"v_cmp_lt_u32 vcc, {V_I}, {V_END}")
    e(f"s_and_b64 {S_LIVE}, {S_LIVE}, vcc")
    e(f"s_mov_b64 exec, {S_LIVE}")
    e("s_cbranch_scc1 .Lvsp_acc28_loop")
    e(".Lvsp_acc28_exit:")
    e("s_waitcnt vmcnt(0)")                                   # the last iteration's look-ahead loads land in dead registers: let them, before the caller reuses them
    e(f"s_mov_b64 exec, {S_EXEC0}")
    return e.ins


def emit_next_entry_load(e):
    """V_EN = sorted[min(i + 1, end - 1)]  (i = V_I; clamped so that the look-ahead never leaves the lane's range)"""
    e(f"v_add_u32 {AUX0}, 1, {V_I}")
    e(f"v_add_u32 {AUX1}, -1, {V_END}")
    e(f"v_min_u32 {AUX0}, {AUX0}, {AUX1}")
    e(f"v_mad_u64_u32 {ADDR}, vcc, {AUX0}, 4, {S_SORTED}")
    e(f"global_load_dword {V_EN}, {ADDR}, off")


def emit_row_load(e):
    """x, y <- table[V_E & 0x7fffffff]: one 128-byte row, 112 bytes of payload in 7 loads"""
    e(f"v_and_b32 {AUX0}, 0x7fffffff, {V_E}")
    e(f"v_lshlrev_b32 {AUX0}, 1, {AUX0}")                      # index < 2^31: twice it fits 32 bits
    e(f"v_mad_u64_u32 {ADDR}, vcc, {AUX0}, 64, {S_TABLE}")
    base = 14 * BLK["x"]
    for j in range(7):
        e(f"global_load_dwordx4 v[{base + 4 * j}:{base + 4 * j + 3}], {ADDR}, off offset:{16 * j}")


def header_text(ins):
    n_mad = sum(1 for x in ins if x.startswith("v_mad_u64_u32"))
    lines = ["s_branch .Lvsp_acc28_end", ".p2align 8", "vsp_acc28:"] + ins + ["s_nop 4", "s_setpc_b64 s[30:31]", ".Lvsp_acc28_end:"]
    body_txt = "\n".join(f'        "{x}\\n\\t"' for x in lines)
    vclob = ", ".join(f'"v{i}"' for i in range(NVGPR))
    sclob = ", ".join(f'"s{i}"' for i in range(NSGPR) if i not in S_RESERVED)
    acc_out = ", ".join(f'"={{v{i}}}"(acc[{i}])' for i in range(56))
    return f"""// GENERATED by tools/gen_accum28_asm.py -- do not edit.  See that file for the design notes and the register map.
// The accumulation of one bucket part (G1, 14 x 28-bit limbs) as ONE routine: {len(ins)} instructions, {n_mad} of them v_mad_u64_u32 in the loop body,
// {NVGPR} VGPRs -- three waves per SIMD.
#pragma once
#include <stdint.h>

namespace vsp {{

// The holder only places the routine's text in the code object: entered as a function it branches over the body, so it clobbers
// nothing.  (The registers the ROUTINE uses are declared where it is entered, accum28_asm below -- in k_accum28's register budget; listed
// here they made the compiler call v128 .. v166 "reserved" against this function's own default budget of 128.)
template <int Instance> __device__ __attribute__((noinline, used)) void accum28_holder() {{
    asm volatile(
{body_txt}
        :
        :
        : "memory");
}}
// acc[56] = X | Y | ZZ | ZZZ of the sum of table[sorted[i] & 0x7fffffff] (negated where bit 31 is set), i in [start, end) -- all zero for an
// empty range; flag = 1 when an equal-x pair (a doubling or a cancellation) was met: acc is then void and the caller hands the part to
// the generic kernel.  Lanes outside EXEC are untouched.
__device__ __forceinline__ void accum28_asm(uint32_t *acc, uint32_t &flag, const void *table, const uint32_t *sorted, uint32_t start, uint32_t end) {{
    asm volatile("s_mov_b64 s[36:37], %[tab]\\n\\t"
        "s_mov_b64 s[38:39], %[srt]\\n\\t"
        "s_getpc_b64 s[30:31]\\n\\t"
        "s_add_u32 s30, s30, vsp_acc28@rel32@lo+4\\n\\t"
        "s_addc_u32 s31, s31, vsp_acc28@rel32@hi+12\\n\\t"
        "s_swappc_b64 s[30:31], s[30:31]"
        : {acc_out}, "={{{V_FLAG}}}"(flag), "+{{{V_I}}}"(start), "+{{{V_END}}}"(end)
        : [tab] "s"(table), [srt] "s"(sorted)
        : "memory", "vcc", "scc", "s30", "s31", {sclob}, {", ".join(f'"v{i}"' for i in range(56, NVGPR) if f"v{i}" not in (V_I, V_END, V_FLAG))});
    (void)&accum28_holder<0>;
}}

}}  // namespace vsp
"""


def main():
    ins = gen_body()
    open(OUT, "w").write(header_text(ins))
    print("wrote", os.path.normpath(OUT), len(ins), "instructions")


if __name__ == "__main__":
    main()
