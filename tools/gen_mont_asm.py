#!/usr/bin/env python3
"""Generates vote_saver_protocol_amd/csrc/mont_asm_gfx950.h: the Montgomery product of the device field types
(Fp: 12 x 32-bit limbs, Fr: 8) for gfx950 as a hand-laid-out routine with a PRIVATE calling convention.

Algorithm: finely integrated product scanning (Comba).  Per limb product exactly one v_mad_u64_u32 (64-bit
multiply-accumulate, carry-out in VCC) and one v_addc_co_u32 (carry into the third accumulator word); per column one
v_mov_b32 and one v_mul_lo_u32 (m_k); the final conditional subtraction runs in the dead operand registers.
Two 64-bit accumulator pairs alternate between columns: while a column sums into one pair its carries collect in the
HIGH register of the other pair -- exactly where the next column needs them.

Why a private convention: under the standard AMDGPU function ABI only every other block of 8 VGPRs above v40 survives a
call, so a kernel with ~100 live limbs around each product needed 243 VGPRs (2 waves/SIMD), and aggregate operands
beyond 16 dwords travel through scratch.  Here the routine body lives behind a label inside a never-called holder
function, uses ONLY v0..v(3N+3), s0..s12, s[30:31], VCC, and is entered with s_swappc_b64 from an inline-asm
trampoline whose operand constraints pin a -> v0.., b -> vN.., r <- v2N.. and whose clobber list is exact.  The
compiler keeps every other live value wherever it likes, with no save/restore and no memory traffic.

Register map (N limbs):  a: v0..vN-1   b: vN..v2N-1   m / r: v2N..v3N-1   accumulators: v[3N:3N+1], v[3N+2:3N+3]
                         s0..sN-1 = modulus limbs, s12 = -p^-1 mod 2^32, s[30:31] = return address

Run:  python tools/gen_mont_asm.py
"""
import os

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "vote_saver_protocol_amd", "csrc", "mont_asm_gfx950.h")


def body(N):
    A = lambda i: f"v{i}"
    B = lambda i: f"v{N + i}"
    M = lambda i: f"v{2 * N + i}"
    Pm = lambda i: f"s{i}"
    INV = "s12"
    base = 3 * N
    pairs = [(f"v{base}", f"v{base + 1}", f"v[{base}:{base + 1}]"), (f"v{base + 2}", f"v{base + 3}", f"v[{base + 2}:{base + 3}]")]
    ins = []
    cur, oth = 0, 1
    ins.append(f"v_mov_b32 {pairs[0][0]}, 0")
    ins.append(f"v_mov_b32 {pairs[0][1]}, 0")
    for k in range(2 * N - 1):
        lo_c, hi_c, pr_c = pairs[cur]
        lo_o, hi_o, _ = pairs[oth]
        terms = []
        if k < N:
            for i in range(k):
                terms.append((A(i), B(k - i)))
                terms.append((M(i), Pm(k - i)))
            terms.append((A(k), B(0)))
            terms.append(("MK", None))
        else:
            for i in range(k - N + 1, N):
                terms.append((A(i), B(k - i)))
                terms.append((M(i), Pm(k - i)))
        first = True
        for x, y in terms:
            if x == "MK":
                ins.append(f"v_mul_lo_u32 {M(k)}, {lo_c}, {INV}")
                x, y = M(k), Pm(0)
            ins.append(f"v_mad_u64_u32 {pr_c}, vcc, {x}, {y}, {pr_c}")
            if first:
                ins.append(f"v_addc_co_u32_e64 {hi_o}, vcc, 0, 0, vcc")
                first = False
            else:
                ins.append(f"v_addc_co_u32 {hi_o}, vcc, 0, {hi_o}, vcc")
        if k >= N:
            ins.append(f"v_mov_b32 {M(k - N)}, {lo_c}")      # m_j is dead from column N+j on: its register becomes r_j
        ins.append(f"v_mov_b32 {lo_o}, {hi_c}")
        cur, oth = oth, cur
    ins.append(f"v_mov_b32 {M(N - 1)}, {pairs[cur][0]}")
    # final conditional subtraction r - p in the dead a / b registers (SGPR + VCC carry-in would be two constant-bus reads)
    ins.append(f"v_subrev_co_u32 {A(0)}, vcc, {Pm(0)}, {M(0)}")
    for i in range(1, N):
        ins.append(f"v_mov_b32 {B(i)}, {Pm(i)}")
        ins.append(f"v_subb_co_u32 {A(i)}, vcc, {M(i)}, {B(i)}, vcc")
    for i in range(N):
        ins.append(f"v_cndmask_b32 {M(i)}, {A(i)}, {M(i)}, vcc")   # borrow -> keep r, else take r - p
    return ins


def gen(N):
    ins = body(N)
    n_mad = sum(1 for x in ins if x.startswith("v_mad"))
    label = f"vsp_mm_{N}"
    lines = [f's_branch .Lvsp_mm_{N}_end', '.p2align 8', f'{label}:']
    lines += [f's_mov_b32 s{i}, %[p{i}]' for i in range(N)]      # literal constants via "i" operands
    lines += ['s_mov_b32 s12, %[inv]']
    # the compiler's hazard recogniser does not see inside inline asm: a DPP (or readlane) read of a result register needs >= 2 wait
    # states after the v_cndmask that wrote it, and the lane-pair Fp2 code does follow products with DPP moves -- pad before returning
    lines += ins + ['s_nop 4', 's_setpc_b64 s[30:31]', f'.Lvsp_mm_{N}_end:']
    body_txt = "\n".join(f'        "{x}\\n\\t"' for x in lines)
    consts = ", ".join([f'[p{i}] "i"(P::MOD[{i}])' for i in range(N)] + ['[inv] "i"(P::INV)'])
    vclob = ", ".join(f'"v{i}"' for i in range(3 * N + 4))
    sclob = ", ".join(f'"s{i}"' for i in range(13))
    holder = f'''// ---- N = {N}: {n_mad} v_mad_u64_u32, {len(ins)} instructions, VGPRs v0..v{3 * N + 3} ----
template <class P> __device__ __attribute__((noinline, used)) void mont_mul_holder_{N}() {{
    static_assert(P::N == {N}, "limb count");
    asm volatile(
{body_txt}
        :
        : {consts}
        : "vcc", "scc", "s30", "s31", {sclob}, {vclob});
}}
'''
    outs = ", ".join([f'"={{v{2 * N + i}}}"(r[{i}])' for i in range(N)] + [f'"+{{v{i}}}"(a[{i}])' for i in range(N)] +
                     [f'"+{{v{N + i}}}"(b[{i}])' for i in range(N)])
    acc = ", ".join(f'"v{3 * N + i}"' for i in range(4))
    call = f'''// r = a*b*2^(-{32 * N}) mod p, fully reduced.  a[] and b[] are clobbered.
template <class P> __device__ __forceinline__ void mont_mul_asm_{N}(uint32_t *r, uint32_t *a, uint32_t *b) {{
    asm("s_getpc_b64 s[30:31]\\n\\t"
        "s_add_u32 s30, s30, {label}@rel32@lo+4\\n\\t"
        "s_addc_u32 s31, s31, {label}@rel32@hi+12\\n\\t"
        "s_swappc_b64 s[30:31], s[30:31]"
        : {outs}
        :
        : "vcc", "scc", "s30", "s31", {sclob}, {acc});
}}
'''
    return holder + "\n" + call



# ---------------------------------------------------------------- carry-free 14 x 28-bit product for Fp (R' = 2^392)
P381 = 0x1A0111EA397FE69A4B1BA7B6434BACD764774B84F38512BF6730D2A0F6B0F6241EABFFFEB153FFFFB9FEFFFFFFFFAAAB
N28, W28 = 14, 28
MASK28 = (1 << W28) - 1


def body28(dual=False, sqr=False):
    """Column sums of 28 products below 2^58 fit a 64-bit accumulator, so no v_addc follows the mads; per column one 64-bit shift
    and one mask.  Operand limbs may be loose: limb bounds 2^Ea, 2^Eb with Ea + Eb <= 59.  Output: limbs below 2^28 (the top limb
    holds the rest), value below a*b / 2^392 + p -- no final subtraction.  a, b are preserved.
    dual: a*b + c*d into the same column accumulators, one reduction (the sum of the two groups of 14 products must stay
    below 2^64 - 2^60: e.g. limb bounds 28+30 and 31+28).
    sqr: a*a with the caller passing b = 2a limb by limb: the off-diagonal products a_i a_j, i < j, are taken once against the doubled
    operand (a_i * 2a_j), the diagonal ones as a_i * a_i -- 105 limb products for the square instead of 196, the same column totals,
    hence the same m_k and bit-identical output.  Limb bound of a: 2^28 (then 2a < 2^29 and a column stays below 2^61)."""
    nin = 4 if dual else 2
    A = lambda i: f"v{i}"
    B = lambda i: f"v{N28 + i}"
    C = lambda i: f"v{2 * N28 + i}"
    D = lambda i: f"v{3 * N28 + i}"
    M = lambda i: f"v{nin * N28 + i}"
    Pm = lambda i: f"s{i}"
    INV, MSK = "s14", "s15"
    base = (nin + 1) * N28
    lo, hi, pr, tmp = f"v{base}", f"v{base + 1}", f"v[{base}:{base + 1}]", f"v{base + 2}"
    ins = [f"v_mov_b32 {lo}, 0", f"v_mov_b32 {hi}, 0"]
    for k in range(2 * N28 - 1):
        for i in range(max(0, k - N28 + 1), min(k, N28 - 1) + 1):
            if sqr:
                if i < k - i:
                    ins.append(f"v_mad_u64_u32 {pr}, vcc, {A(i)}, {B(k - i)}, {pr}")       # a_i * (2 a_j)
                elif i == k - i:
                    ins.append(f"v_mad_u64_u32 {pr}, vcc, {A(i)}, {A(i)}, {pr}")
                continue
            ins.append(f"v_mad_u64_u32 {pr}, vcc, {A(i)}, {B(k - i)}, {pr}")
            if dual:
                ins.append(f"v_mad_u64_u32 {pr}, vcc, {C(i)}, {D(k - i)}, {pr}")
        for i in (range(0, k) if k < N28 else range(k - N28 + 1, N28)):
            ins.append(f"v_mad_u64_u32 {pr}, vcc, {M(i)}, {Pm(k - i)}, {pr}")
        if k < N28:
            ins.append(f"v_mul_lo_u32 {tmp}, {lo}, {INV}")
            ins.append(f"v_and_b32 {M(k)}, {MSK}, {tmp}")
            ins.append(f"v_mad_u64_u32 {pr}, vcc, {M(k)}, {Pm(0)}, {pr}")
        else:
            ins.append(f"v_and_b32 {M(k - N28)}, {MSK}, {lo}")
        ins.append(f"v_lshrrev_b64 {pr}, {W28}, {pr}")
    ins.append(f"v_mov_b32 {M(N28 - 1)}, {lo}")
    return ins


def limbs28(v):
    return [(v >> (W28 * i)) & MASK28 for i in range(N28 - 1)] + [v >> (W28 * (N28 - 1))]


def redundant(c, lend):
    """c*p with every limb but the top raised by lend * 2^28 (borrowed from the limb above): subtracting a value whose limbs are
    below lend * 2^28 (top limb below the top limb here) never goes negative in any limb."""
    k = limbs28(c * P381)
    out = [k[0] + lend * (1 << W28)] + [k[i] + lend * (1 << W28) - lend for i in range(1, N28 - 1)] + [k[N28 - 1] - lend]
    assert sum(x << (W28 * i) for i, x in enumerate(out)) == c * P381 and all(0 <= x < (1 << 32) for x in out)
    return out


def gen28():
    p28 = limbs28(P381)
    inv28 = (-pow(P381, -1, 1 << W28)) % (1 << W28)
    ins = body28()
    lines = ['s_branch .Lvsp_mm28_end', '.p2align 8', 'vsp_mm28:']
    lines += [f's_mov_b32 s{i}, 0x{p28[i]:x}' for i in range(N28)] + [f's_mov_b32 s14, 0x{inv28:x}', f's_mov_b32 s15, 0x{MASK28:x}']
    lines += ins + ['s_nop 4', 's_setpc_b64 s[30:31]', '.Lvsp_mm28_end:']
    body_txt = "\n".join(f'        "{x}\\n\\t"' for x in lines)
    vclob = ", ".join(f'"v{i}"' for i in range(45))
    sclob = ", ".join(f'"s{i}"' for i in range(16))
    n_mad = sum(1 for x in ins if x.startswith("v_mad"))
    arr = lambda name, v: f"static constexpr uint32_t {name}[14] = {{" + ", ".join(f"0x{x:x}u" for x in v) + "};"
    R1 = (1 << 392) % P381
    consts = "\n".join([arr("FP28_P", p28), arr("FP28_ONE", limbs28(R1)), arr("FP28_R2", limbs28(R1 * R1 % P381)),
                        arr("FP28_K8_L1", redundant(8, 1)), arr("FP28_K8_L4", redundant(8, 4)),
                        arr("FP28_K32_L1", redundant(32, 1)), arr("FP28_K32_L4", redundant(32, 4)),
                        arr("FP28_K64_L1", redundant(64, 1)), arr("FP28_K64_L4", redundant(64, 4)),
                        arr("FP28_2P", limbs28(2 * P381)), arr("FP28_3P", limbs28(3 * P381))])
    outs = ", ".join(f'"={{v{2 * N28 + i}}}"(r[{i}])' for i in range(N28))
    inps = ", ".join([f'"{{v{i}}}"(a[{i}])' for i in range(N28)] + [f'"{{v{N28 + i}}}"(b[{i}])' for i in range(N28)])
    # dual variant
    ins2 = body28(dual=True)
    lines2 = ['s_branch .Lvsp_mm28x2_end', '.p2align 8', 'vsp_mm28x2:']
    lines2 += [f's_mov_b32 s{i}, 0x{p28[i]:x}' for i in range(N28)] + [f's_mov_b32 s14, 0x{inv28:x}', f's_mov_b32 s15, 0x{MASK28:x}']
    lines2 += ins2 + ['s_nop 4', 's_setpc_b64 s[30:31]', '.Lvsp_mm28x2_end:']
    body2_txt = "\n".join(f'        "{x}\\n\\t"' for x in lines2)
    vclob2 = ", ".join(f'"v{i}"' for i in range(5 * N28 + 3))
    outs2 = ", ".join(f'"={{v{4 * N28 + i}}}"(r[{i}])' for i in range(N28))
    inps2 = ", ".join([f'"{{v{i}}}"(a[{i}])' for i in range(N28)] + [f'"{{v{N28 + i}}}"(b[{i}])' for i in range(N28)] +
                      [f'"{{v{2 * N28 + i}}}"(c[{i}])' for i in range(N28)] + [f'"{{v{3 * N28 + i}}}"(d[{i}])' for i in range(N28)])
    n_mad2 = sum(1 for x in ins2 if x.startswith("v_mad"))
    # squaring variant: same register map as the single product (a in v0..13, 2a in v14..27, result v28..41)
    ins3 = body28(sqr=True)
    lines3 = ['s_branch .Lvsp_sq28_end', '.p2align 8', 'vsp_sq28:']
    lines3 += [f's_mov_b32 s{i}, 0x{p28[i]:x}' for i in range(N28)] + [f's_mov_b32 s14, 0x{inv28:x}', f's_mov_b32 s15, 0x{MASK28:x}']
    lines3 += ins3 + ['s_nop 4', 's_setpc_b64 s[30:31]', '.Lvsp_sq28_end:']
    body3_txt = "\n".join(f'        "{x}\\n\\t"' for x in lines3)
    n_mad3 = sum(1 for x in ins3 if x.startswith("v_mad"))
    return f"""// ---- Fp on 14 x 28-bit limbs, R' = 2^392: {n_mad} v_mad_u64_u32, {len(ins)} instructions, no carries, no final subtraction; VGPRs v0..v44 ----
// constants: p; R' mod p (the Montgomery one); R'^2 mod p; and multiples of p in the redundant form used by the lazy subtractions
// (FP28_Kc_Ll = c*p with every limb but the top raised by l * 2^28)
{consts}
template <int Instance> __device__ __attribute__((noinline, used)) void mont_mul28_holder() {{
    asm volatile(
{body_txt}
        :
        :
        : "vcc", "scc", "s30", "s31", {sclob}, {vclob});
}}
// r = a*b*2^(-392) mod p as described at body28(); a and b are preserved
__device__ __forceinline__ void mont_mul28_asm(uint32_t *r, const uint32_t *a, const uint32_t *b) {{
    asm("s_getpc_b64 s[30:31]\\n\\t"
        "s_add_u32 s30, s30, vsp_mm28@rel32@lo+4\\n\\t"
        "s_addc_u32 s31, s31, vsp_mm28@rel32@hi+12\\n\\t"
        "s_swappc_b64 s[30:31], s[30:31]"
        : {outs}
        : {inps}
        : "vcc", "scc", "s30", "s31", {sclob}, "v42", "v43", "v44");
}}
// ---- the square: r = a*a*2^(-392) with a2 = 2a passed by the caller; {n_mad3} v_mad_u64_u32, {len(ins3)} instructions; bit-identical to mont_mul28_asm(r, a, a)
template <int Instance> __device__ __attribute__((noinline, used)) void mont_sqr28_holder() {{
    asm volatile(
{body3_txt}
        :
        :
        : "vcc", "scc", "s30", "s31", {sclob}, {vclob});
}}
__device__ __forceinline__ void mont_sqr28_asm(uint32_t *r, const uint32_t *a, const uint32_t *a2) {{
    asm("s_getpc_b64 s[30:31]\\n\\t"
        "s_add_u32 s30, s30, vsp_sq28@rel32@lo+4\\n\\t"
        "s_addc_u32 s31, s31, vsp_sq28@rel32@hi+12\\n\\t"
        "s_swappc_b64 s[30:31], s[30:31]"
        : {outs}
        : {inps.replace("(b[", "(a2[")}
        : "vcc", "scc", "s30", "s31", {sclob}, "v42", "v43", "v44");
}}
// ---- the same for a*b + c*d with one reduction: {n_mad2} v_mad_u64_u32, {len(ins2)} instructions; operands v0..v55 (preserved), result v56..v69 ----
template <int Instance> __device__ __attribute__((noinline, used)) void mont_mul28x2_holder() {{
    asm volatile(
{body2_txt}
        :
        :
        : "vcc", "scc", "s30", "s31", {sclob}, {vclob2});
}}
__device__ __forceinline__ void mont_mul28x2_asm(uint32_t *r, const uint32_t *a, const uint32_t *b, const uint32_t *c, const uint32_t *d) {{
    asm("s_getpc_b64 s[30:31]\\n\\t"
        "s_add_u32 s30, s30, vsp_mm28x2@rel32@lo+4\\n\\t"
        "s_addc_u32 s31, s31, vsp_mm28x2@rel32@hi+12\\n\\t"
        "s_swappc_b64 s[30:31], s[30:31]"
        : {outs2}
        : {inps2}
        : "vcc", "scc", "s30", "s31", {sclob}, "v70", "v71", "v72");
}}
"""

# ---------------------------------------------------------------- carry-free 9 x 29-bit product for Fr (R' = 2^261): the NTT butterflies
R255 = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
N29, W29 = 9, 29
# vsp_mm29:  a v64..72, b v74..82, result v84..92, temp v73, accumulator v[94:95]
# vsp_mm29q: a v96..104, the SAME b v74..82, result v108..116, temp v105, accumulator v[106:107]
# HIGH registers on purpose: the compiler hands out registers from v0 upwards for its own short-lived values (addresses, loop counters), and
# fixed operand registers down there collide with them -- every collision is a copy before the call.  Up here the LDS and table loads of a
# radix-4 group land straight in the operand registers (k_ntt29_pass: 126 VGPRs in all, four waves per SIMD).
A29, B29, M29, T29, ACC29 = 64, 74, 84, 73, 94
QA29, QM29, QT29, QACC29 = 96, 108, 105, 106
MASK29 = (1 << W29) - 1


def limbs29(v):
    return [(v >> (W29 * i)) & MASK29 for i in range(N29 - 1)] + [v >> (W29 * (N29 - 1))]


def body29(a0=A29, b0=B29, m0=M29, tmpreg=T29, acc=ACC29):
    """The 28-bit scheme for Fr: a column is 9 + 9 products below 2^60 (operand limb bounds 2^Ea, 2^Eb with Ea + Eb <= 60: a twiddle is
    tight, 29 bits, so a data operand may have limbs up to 2^31), one 64-bit accumulator, no carries.  r = 1 mod 2^32, so -1/r = -1
    mod 2^29 and m_k is a negation and a mask instead of a multiplication.  Output: limbs below 2^29 (top limb the rest), value below
    a*b / 2^261 + r; a and b are preserved.  Register map: the arguments (a0, b0, m0: first register of a, b and the result; a temporary; the
    64-bit accumulator pair); s0..s8 = r's limbs, s9 = mask.  Every operand starts at an even register whose 4-register groups are even too:
    the three planes of a value (two 128-bit words and a dword) are loaded from LDS / memory straight into the operand registers and stored
    straight from the result registers -- register tuples of 64 bits and more start at even registers on gfx950, so round 3's b at v9..17
    forced nine copies per twiddle."""
    A = lambda i: f"v{a0 + i}"
    B = lambda i: f"v{b0 + i}"
    M = lambda i: f"v{m0 + i}"
    Pm = lambda i: f"s{i}"
    MSK = "s9"
    base = acc                            # 64-bit register pairs must start at an even register on gfx950
    assert base % 2 == 0 and a0 % 4 == 0 and m0 % 4 == 0 and b0 % 2 == 0
    lo, hi, pr, tmp = f"v{base}", f"v{base + 1}", f"v[{base}:{base + 1}]", f"v{tmpreg}"
    ins = [f"v_mov_b32 {lo}, 0", f"v_mov_b32 {hi}, 0"]
    for k in range(2 * N29 - 1):
        for i in range(max(0, k - N29 + 1), min(k, N29 - 1) + 1):
            ins.append(f"v_mad_u64_u32 {pr}, vcc, {A(i)}, {B(k - i)}, {pr}")
        for i in (range(0, k) if k < N29 else range(k - N29 + 1, N29)):
            ins.append(f"v_mad_u64_u32 {pr}, vcc, {M(i)}, {Pm(k - i)}, {pr}")
        if k < N29:
            ins.append(f"v_sub_u32 {tmp}, 0, {lo}")                      # m_k = -lo mod 2^29
            ins.append(f"v_and_b32 {M(k)}, {MSK}, {tmp}")
            ins.append(f"v_mad_u64_u32 {pr}, vcc, {M(k)}, {Pm(0)}, {pr}")
        else:
            ins.append(f"v_and_b32 {M(k - N29)}, {MSK}, {lo}")
        ins.append(f"v_lshrrev_b64 {pr}, {W29}, {pr}")
    ins.append(f"v_mov_b32 {M(N29 - 1)}, {lo}")
    return ins


def redundant29(c, lend):
    k = limbs29(c * R255)
    out = [k[0] + lend * (1 << W29)] + [k[i] + lend * (1 << W29) - lend for i in range(1, N29 - 1)] + [k[N29 - 1] - lend]
    assert sum(x << (W29 * i) for i, x in enumerate(out)) == c * R255 and all(0 <= x < (1 << 32) for x in out)
    return out


def gen29():
    r29 = limbs29(R255)
    assert (-pow(R255, -1, 1 << W29)) % (1 << W29) == MASK29
    ins = body29()
    insq = body29(QA29, B29, QM29, QT29, QACC29)
    consts_s = [f's_mov_b32 s{i}, 0x{r29[i]:x}' for i in range(N29)] + [f's_mov_b32 s9, 0x{MASK29:x}']
    lines = ['s_branch .Lvsp_mm29_end', '.p2align 8', 'vsp_mm29:']
    lines += consts_s + ins + ['s_nop 4', 's_setpc_b64 s[30:31]', '.Lvsp_mm29_end:']
    linesq = ['s_branch .Lvsp_mm29q_end', '.p2align 8', 'vsp_mm29q:'] + consts_s + insq + ['s_nop 4', 's_setpc_b64 s[30:31]', '.Lvsp_mm29q_end:']
    body_txt = "\n".join(f'        "{x}\\n\\t"' for x in lines)
    bodyq_txt = "\n".join(f'        "{x}\\n\\t"' for x in linesq)
    vclob = ", ".join(f'"v{i}"' for i in range(A29, ACC29 + 2))
    vclobq = ", ".join([f'"v{i}"' for i in range(B29, B29 + N29)] + [f'"v{i}"' for i in range(QA29, QM29 + N29)])
    sclob = ", ".join(f'"s{i}"' for i in range(10))
    n_mad = sum(1 for x in ins if x.startswith("v_mad"))
    arr = lambda name, v: f"static constexpr uint32_t {name}[9] = {{" + ", ".join(f"0x{x:x}u" for x in v) + "};"
    R1 = (1 << 261) % R255
    consts = "\n".join([arr("FR29_R", r29), arr("FR29_ONE", limbs29(R1)), arr("FR29_R2", limbs29(R1 * R1 % R255)),
                        arr("FR29_K2_L1", redundant29(2, 1)), arr("FR29_K4_L1", redundant29(4, 1))])
    outs = ", ".join(f'"={{v{M29 + i}}}"(r[{i}])' for i in range(N29))
    inps = ", ".join([f'"{{v{A29 + i}}}"(a[{i}])' for i in range(N29)] + [f'"{{v{B29 + i}}}"(b[{i}])' for i in range(N29)])
    outsq = ", ".join(f'"={{v{QM29 + i}}}"(r[{i}])' for i in range(N29))
    inpsq = ", ".join([f'"{{v{QA29 + i}}}"(a[{i}])' for i in range(N29)] + [f'"{{v{B29 + i}}}"(b[{i}])' for i in range(N29)])
    return f"""// ---- Fr on 9 x 29-bit limbs, R' = 2^261: {n_mad} v_mad_u64_u32, {len(ins)} instructions, no carries, no final subtraction; VGPRs v0..v31 ----
// constants: r; R' mod r (the Montgomery one); R'^2 mod r; 2r and 4r in the redundant form of the lazy subtractions
{consts}
template <int Instance> __device__ __attribute__((noinline, used)) void mont_mul29_holder() {{
    asm volatile(
{body_txt}
        :
        :
        : "vcc", "scc", "s30", "s31", {sclob}, {vclob});
}}
// r = a*b*2^(-261) mod r as described at body29(); a and b are preserved
__device__ __forceinline__ void mont_mul29_asm(uint32_t *r, const uint32_t *a, const uint32_t *b) {{
    asm("s_getpc_b64 s[30:31]\\n\\t"
        "s_add_u32 s30, s30, vsp_mm29@rel32@lo+4\\n\\t"
        "s_addc_u32 s31, s31, vsp_mm29@rel32@hi+12\\n\\t"
        "s_swappc_b64 s[30:31], s[30:31]"
        : {outs}
        : {inps}
        : "vcc", "scc", "s30", "s31", {sclob}, "v{T29}", "v{ACC29}", "v{ACC29 + 1}");
}}
template <int Instance> __device__ __attribute__((noinline, used)) void mont_mul29q_holder() {{
    asm volatile(
{bodyq_txt}
        :
        :
        : "vcc", "scc", "s30", "s31", {sclob}, {vclobq});
}}
// the same product with ANOTHER register map (a v{QA29}..{QA29 + 8}, the SAME b v{B29}..{B29 + 8}, result v{QM29}..{QM29 + 8}): the second product of a pair keeps its
// operand and its result in registers of its own, so the pair x1 w, x3 w of a radix-4 step needs no copies in or out
__device__ __forceinline__ void mont_mul29q_asm(uint32_t *r, const uint32_t *a, const uint32_t *b) {{
    asm("s_getpc_b64 s[30:31]\\n\\t"
        "s_add_u32 s30, s30, vsp_mm29q@rel32@lo+4\\n\\t"
        "s_addc_u32 s31, s31, vsp_mm29q@rel32@hi+12\\n\\t"
        "s_swappc_b64 s[30:31], s[30:31]"
        : {outsq}
        : {inpsq}
        : "vcc", "scc", "s30", "s31", {sclob}, "v{QT29}", "v{QACC29}", "v{QACC29 + 1}");
}}
"""


def main():
    text = '''// GENERATED by tools/gen_mont_asm.py -- do not edit.  See that file for the design notes.
#pragma once
#include <stdint.h>

namespace vsp {

''' + gen(12) + "\n" + gen(8) + "\n" + gen28() + "\n" + gen29() + "\n}  // namespace vsp\n"
    open(OUT, "w").write(text)
    print("wrote", os.path.normpath(OUT))


if __name__ == "__main__":
    main()
