// Fp on 14 x 28-bit limbs with lazy reduction, for the G1 bucket accumulation over precomputed window multiples.
//
// Why: with 28-bit limbs a column of the Montgomery product (28 partial products below 2^58) fits a 64-bit accumulator, so the
// v_addc_co_u32 that follows every v_mad_u64_u32 of the 12 x 32-bit routine disappears, and so does the final subtraction:
// 2284 cycles per wave-product against 3149 (profiles/r1_ubench_mont.txt).  The price is a discipline of bounds, kept here:
//
//   value form     Montgomery with R' = 2^392:  x stands for x * 2^392 mod p
//   "tight"        limbs below 2^28 (the top limb holds the rest) -- every product output, every normalised value
//   "loose"        limbs below 2^30 (results of the lazy subtractions a + K - b)
//   products       need limb bounds 2^Ea, 2^Eb with Ea + Eb <= 59 and values below 2^388; output tight, value < a*b/2^392 + p
//   a - b          = a + K - b limb by limb, K a multiple of p in a redundant form whose limbs dominate those of b
//                  (FP28_K8_L1: b tight, b < 4p ... FP28_K32_L4: b loose, b < 16p); no borrow ever crosses a limb
//
// Invariants of the accumulator between mixed additions (checked mechanically by tests/test_fp28_bounds.py: an exact limb model
// driven to these worst cases, and a worst-case propagation showing they are inductive):
//   X tight, value < 9.5 p;   Y tight, value < 8 p (1.5 p after the first addition);   ZZ, ZZZ tight, value < 1.1 p   (inf: all limbs zero)
// Derivation (p / 2^392 = 1 / 2521):  P = U2 + 32p - X < 33.1p,  PP = P^2 < (33.1^2 / 2521 + 1) p = 1.44p,  PPP, Q < 1.02p,
//   R = S2 + 32p - Y < 33.5p,  R^2 < 1.45p,  s = PPP + 2Q < 3.1p (limbs < 3 * 2^28),  X3 = R^2 + 8p - s < 9.5p,
//   Q - X3 + 32p < 33.1p (limbs < 2^30),  32p - Y1 < 32p (limbs < 2^29),  Y3 = [R (Q - X3) + (32p - Y1) PPP] / 2^392 + p < 1.5p.
// G1: the bucket sums stay in this form through the merges and the bucket reduction (XYZZ<Fp28> below); conversion happens once per
// table entry (at precomputation) and once per window result.  G2 converts at the store of each bucket part.
#pragma once
#include "curve.h"

namespace vsp {

struct Fp28 {                                          // 56 bytes (no over-alignment: that would pad it to 64)
    uint32_t l[14];
    VSP_HD static Fp28 zero() { Fp28 r; for (int i = 0; i < 14; i++) r.l[i] = 0; return r; }      // XYZZ<Fp28>::inf()
};
VSP_HD bool is_zero(const Fp28 &a) { uint32_t o = 0; for (int i = 0; i < 14; i++) o |= a.l[i]; return o == 0; }
// Table rows are padded to whole 128-byte cache lines: a gathered G1 point is ONE line (112 bytes of payload), a G2 point two.
// Unpadded 112 / 224-byte rows straddled two / three lines in 6 of 8 alignments: 273 bytes fetched per 112-byte gather (round 1 PMC).
struct alignas(128) Affine28 { Fp28 x, y; };
static_assert(sizeof(Fp28) == 56 && sizeof(Affine28) == 128, "G1 table rows are one 128-byte line");
struct XYZZ28 { Fp28 X, Y, ZZ, ZZZ; };
// G2 bucket sums in the 28-bit form: memory holds Fp2x28 values (c0, c1: 112 bytes; XYZZ<Fp2x28> = 448 bytes), a lane of a pair holds
// one component (Fp28L), exactly as Fp2 / Fp2L in field.h
struct Fp2x28 { Fp28 c0, c1; VSP_HD static Fp2x28 zero() { Fp2x28 r; r.c0 = Fp28::zero(); r.c1 = Fp28::zero(); return r; } };
struct Fp28L { Fp28 v; VSP_HD static Fp28L zero() { Fp28L r; r.v = Fp28::zero(); return r; } };
VSP_HD bool is_zero(const Fp2x28 &a) { return is_zero(a.c0) && is_zero(a.c1); }
struct alignas(256) Affine28x2 { Fp28 xc0, xc1, yc0, yc1; };     // one G2 point of the 28-bit table: 224 bytes of payload in two lines
static_assert(sizeof(Affine28x2) == 256, "G2 table rows are two 128-byte lines");

#if defined(__HIP_DEVICE_COMPILE__)        // the product routine exists in the device pass only; kernels guard their bodies alike
__device__ __forceinline__ Fp28 fp28_zero() { Fp28 r; for (int i = 0; i < 14; i++) r.l[i] = 0; return r; }
__device__ __forceinline__ Fp28 fp28_const(const uint32_t (&c)[14]) { Fp28 r; for (int i = 0; i < 14; i++) r.l[i] = c[i]; return r; }
__device__ __forceinline__ bool fp28_all_zero(const Fp28 &a) { uint32_t o = 0; for (int i = 0; i < 14; i++) o |= a.l[i]; return o == 0; }
__device__ __forceinline__ bool fp28_equals(const Fp28 &a, const uint32_t (&c)[14]) { uint32_t o = 0; for (int i = 0; i < 14; i++) o |= a.l[i] ^ c[i]; return o == 0; }

#if defined(VSP_PORTABLE_MUL)
// DIAGNOSTIC BUILD (`make portable`): the same column schedule as the generated routines (tools/gen_mont_asm.py body28) in plain C++ -- one
// 64-bit accumulator per column, m_k = (acc * (-1/p mod 2^28)) mod 2^28, no final subtraction -- as real functions (inlined they make the
// translation unit take tens of minutes to compile).  Bit-identical outputs by construction: the same column totals.
__device__ __noinline__ void mont28_portable(uint32_t *r, const uint32_t *a, const uint32_t *b, const uint32_t *c, const uint32_t *d) {
    uint32_t m[14];
    uint64_t acc = 0;
    for (int k = 0; k < 27; k++) {
        for (int i = (k > 13 ? k - 13 : 0); i <= (k < 13 ? k : 13); i++) {
            acc += (uint64_t)a[i] * b[k - i];
            if (c) acc += (uint64_t)c[i] * d[k - i];
        }
        if (k < 14) {
            for (int i = 0; i < k; i++) acc += (uint64_t)m[i] * FP28_P[k - i];
            m[k] = ((uint32_t)acc * 0xffcfffdu) & 0x0FFFFFFFu;
            acc += (uint64_t)m[k] * FP28_P[0];
        } else {
            for (int i = k - 13; i < 14; i++) acc += (uint64_t)m[i] * FP28_P[k - i];
            r[k - 14] = (uint32_t)acc & 0x0FFFFFFFu;
        }
        acc >>= 28;
    }
    r[13] = (uint32_t)acc;
}
__device__ __forceinline__ Fp28 mul28(const Fp28 &a, const Fp28 &b) { Fp28 r; mont28_portable(r.l, a.l, b.l, nullptr, nullptr); return r; }
__device__ __forceinline__ Fp28 sqr28(const Fp28 &a) { Fp28 r; mont28_portable(r.l, a.l, a.l, nullptr, nullptr); return r; }
__device__ __forceinline__ Fp28 mul28x2(const Fp28 &a, const Fp28 &b, const Fp28 &c, const Fp28 &d) { Fp28 r; mont28_portable(r.l, a.l, b.l, c.l, d.l); return r; }
#else
__device__ __forceinline__ Fp28 mul28(const Fp28 &a, const Fp28 &b) {
    Fp28 r;
    __builtin_amdgcn_sched_barrier(0);          // the machine scheduler of this toolchain crashes when it moves code across the call
    mont_mul28_asm(r.l, a.l, b.l);
    __builtin_amdgcn_sched_barrier(0);
    (void)&mont_mul28_holder<0>;
    return r;
}
// a*a for a TIGHT a (limbs below 2^28): the off-diagonal limb products are taken once against 2a -- 105 products for the square
// instead of 196 (301 v_mad_u64_u32 against 392); same column totals, so the result is bit-identical to mul28(a, a)
__device__ __forceinline__ Fp28 sqr28(const Fp28 &a) {
    Fp28 r, a2;
#pragma unroll
    for (int i = 0; i < 14; i++) a2.l[i] = 2u * a.l[i];
    __builtin_amdgcn_sched_barrier(0);
    mont_sqr28_asm(r.l, a.l, a2.l);
    __builtin_amdgcn_sched_barrier(0);
    (void)&mont_sqr28_holder<0>;
    return r;
}
// a*b + c*d with one reduction (both products into the same column accumulators): the limb bounds of the two groups must keep
// 14 * (2^(Ea+Eb) + 2^(Ec+Ed) + 2^56) below 2^64, e.g. 28+30 and 29+28; output tight, value < (a*b + c*d) / 2^392 + p
__device__ __forceinline__ Fp28 mul28x2(const Fp28 &a, const Fp28 &b, const Fp28 &c, const Fp28 &d) {
    Fp28 r;
    __builtin_amdgcn_sched_barrier(0);
    mont_mul28x2_asm(r.l, a.l, b.l, c.l, d.l);
    __builtin_amdgcn_sched_barrier(0);
    (void)&mont_mul28x2_holder<0>;
    return r;
}
#endif
// a + K - b, limb by limb (see the header for which K goes with which b)
__device__ __forceinline__ Fp28 sub28(const Fp28 &a, const uint32_t (&K)[14], const Fp28 &b) {
    Fp28 r;
#pragma unroll
    for (int i = 0; i < 14; i++) r.l[i] = a.l[i] + K[i] - b.l[i];
    return r;
}
__device__ __forceinline__ Fp28 neg28(const uint32_t (&K)[14], const Fp28 &b) {
    Fp28 r;
#pragma unroll
    for (int i = 0; i < 14; i++) r.l[i] = K[i] - b.l[i];
    return r;
}
// carry pass: loose -> tight (same value)
__device__ __forceinline__ Fp28 norm28(const Fp28 &a) {
    Fp28 r;
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < 13; i++) { uint32_t t = a.l[i] + c; r.l[i] = t & 0x0FFFFFFFu; c = t >> 28; }
    r.l[13] = a.l[13] + c;
    return r;
}
// a product output (tight, value < 2p) is congruent to zero exactly when it is 0 or p
__device__ __forceinline__ bool fp28_product_is_zero(const Fp28 &a) { return fp28_all_zero(a) || fp28_equals(a, FP28_P); }

// ---- conversion to and from the library's 12 x 32-bit Montgomery form (R = 2^384)
__device__ __forceinline__ Fp28 fp_to_fp28(const Fp &m) {
    __builtin_amdgcn_sched_barrier(0);
    Fp c = from_mont(m);                       // canonical residue, 12 x 32 bits
    __builtin_amdgcn_sched_barrier(0);
    Fp28 s;
#pragma unroll
    for (int i = 0; i < 14; i++) {
        const int bit = 28 * i, w = bit >> 5, sh = bit & 31;
        uint64_t v = c.l[w];
        if (w + 1 < 12) v |= (uint64_t)c.l[w + 1] << 32;
        s.l[i] = (uint32_t)(v >> sh) & (i < 13 ? 0x0FFFFFFFu : 0xFFFFFFFFu);
    }
    return mul28(s, fp28_const(FP28_R2));       // x * R'^2 / R' = x * R'
}
// any value within the invariants above (loose limbs allowed)
__device__ __forceinline__ Fp fp28_to_fp(const Fp28 &x) {
    Fp28 one = fp28_zero(); one.l[0] = 1;
    Fp28 v = mul28(x, one);                     // plain residue, tight, in [0, p + small)
    // conditional subtraction of p (exact: v is tight, so the limbs compare like digits)
    Fp28 d; uint32_t borrow = 0;
#pragma unroll
    for (int i = 0; i < 14; i++) {
        uint32_t t = v.l[i] - FP28_P[i] - borrow;
        borrow = (i < 13) ? ((t >> 28) & 1u) : (t >> 31);      // limbs < 2^28: a wrapped difference has bit 28 (top limb: bit 31) set
        d.l[i] = (i < 13) ? (t & 0x0FFFFFFFu) : t;
    }
    if (!borrow) v = d;
    Fp c;
#pragma unroll
    for (int w = 0; w < 12; w++) {
        uint32_t acc = 0;
#pragma unroll
        for (int i = 0; i < 14; i++) {
            const int lo = 28 * i - 32 * w;     // position of limb i's bit 0 inside word w
            if (lo > -28 && lo < 32) acc |= lo >= 0 ? (v.l[i] << lo) : (v.l[i] >> (-lo));
        }
        c.l[w] = acc;
    }
    __builtin_amdgcn_sched_barrier(0);
    Fp out = to_mont(c);
    __builtin_amdgcn_sched_barrier(0);
    return out;
}

__device__ __forceinline__ bool is_inf28(const Affine28 &p) { return fp28_all_zero(p.x) && fp28_all_zero(p.y); }
__device__ __forceinline__ bool is_inf28(const XYZZ28 &p) { return fp28_all_zero(p.ZZ); }
__device__ __forceinline__ XYZZ28 xyzz28_inf() { XYZZ28 r; r.X = r.Y = r.ZZ = r.ZZZ = fp28_zero(); return r; }
__device__ __forceinline__ XYZZ<Fp> xyzz28_to_fp(const XYZZ28 &a) {
    XYZZ<Fp> r;
    if (is_inf28(a)) return XYZZ<Fp>::inf();
    r.X = fp28_to_fp(a.X); r.Y = fp28_to_fp(a.Y); r.ZZ = fp28_to_fp(a.ZZ); r.ZZZ = fp28_to_fp(a.ZZZ);
    return r;
}

// acc += q (q affine, optionally negated): the mixed addition of curve.h (madd-2008-s) under the bounds of the header.
//   U2 = x2 ZZ1, S2 = y2 ZZZ1, P = U2 - X1, R = S2 - Y1, PP = P^2, PPP = P PP, Q = X1 PP,
//   X3 = R^2 - (PPP + 2Q), Y3 = R (Q - X3) - Y1 PPP, ZZ3 = ZZ1 PP, ZZZ3 = ZZZ1 PPP
// Returns false -- leaving acc untouched -- in the exceptional case of equal x (doubling or cancellation): the caller hands the
// whole bucket part to the generic 12 x 32-bit kernel instead (a call or the generic formulas inline would poison the hot loop's
// register allocation, and this toolchain's scheduler crashes on them).
__device__ __forceinline__ bool madd28(XYZZ28 &acc, const Affine28 &q, bool negate) {
    if (is_inf28(q)) return true;
    Fp28 qy = negate ? neg28(FP28_K8_L1, q.y) : q.y;                     // loose, < 8p
    if (is_inf28(acc)) { acc.X = q.x; acc.Y = negate ? norm28(qy) : qy; acc.ZZ = fp28_const(FP28_ONE); acc.ZZZ = acc.ZZ; return true; }
    Fp28 U2 = mul28(q.x, acc.ZZ);
    Fp28 S2 = mul28(qy, acc.ZZZ);
    Fp28 P = norm28(sub28(U2, FP28_K32_L1, acc.X));                      // X1 tight, < 16p  ->  P tight, < 33.1p
    Fp28 PP = sqr28(P);                                                  // P tight  ->  < 1.44p
    if (fp28_product_is_zero(PP)) return false;
    Fp28 R = norm28(sub28(S2, FP28_K32_L1, acc.Y));                      // Y1 tight, < 16p  ->  R tight, < 33.5p
    Fp28 PPP = mul28(P, PP);
    Fp28 Q = mul28(acc.X, PP);
    Fp28 s;                                                              // PPP + 2Q: limbs < 3 * 2^28, value < 4p
#pragma unroll
    for (int i = 0; i < 14; i++) s.l[i] = PPP.l[i] + 2u * Q.l[i];
    Fp28 X3 = norm28(sub28(sqr28(R), FP28_K8_L4, s));                    // R tight;  X3 tight, < 9.5p
    // Y3 = R (Q - X3) - Y1 PPP = R (Q - X3) + (32p - Y1) PPP: one dual product, one reduction; limb bounds 28+30 and 29+28
    acc.Y = mul28x2(R, sub28(Q, FP28_K32_L1, X3), neg28(FP28_K32_L1, acc.Y), PPP);      // tight, < 1.5p
    acc.X = X3;
    acc.ZZ = mul28(acc.ZZ, PP);
    acc.ZZZ = mul28(acc.ZZZ, PPP);
    return true;
}
// ---- the bucket sums of the G1 path stay in the 28-bit form through the merges and the bucket reduction: XYZZ<Fp28>, 224 bytes, the
// layout of XYZZ28.  acc += q, both XYZZ (add-2008-s), under the invariants of the header -- which the result satisfies again:
//   U1 = X1 ZZ2, U2 = X2 ZZ1, S1 = Y1 ZZZ2, S2 = Y2 ZZZ1            product outputs: tight, < 1.01p  (X < 9.5p, Y < 8p, ZZ, ZZZ < 1.1p)
//   P = U2 + 8p - U1, R = S2 + 8p - S1                               tight after the carry pass, < 9.01p
//   PP = P^2 < 1.04p, PPP = P PP, Q = U1 PP < 1.01p,  X3 = R^2 + 8p - (PPP + 2Q) < 9.1p
//   Y3 = [R (Q + 32p - X3) + (8p - S1) PPP] / 2^392 + p < 1.13p      limb bounds 28+30 and 29+28 as in madd28
//   ZZ3 = (ZZ1 ZZ2) PP, ZZZ3 = (ZZZ1 ZZZ2) PPP < 1.01p
// 12 products + 2 squares.  Equal x (P = 0: a doubling or a cancellation -- coinciding partial sums, e.g. duplicated bases) is finished
// in the same form (xyzz_dbl28 below).  tests/test_fp28_bounds.py models this function too.
__device__ __forceinline__ XYZZ<Fp28> xyzz28_from_fp(const XYZZ<Fp> &a) {
    XYZZ<Fp28> r;
    if (is_inf(a)) return XYZZ<Fp28>::inf();
    r.X = fp_to_fp28(a.X); r.Y = fp_to_fp28(a.Y); r.ZZ = fp_to_fp28(a.ZZ); r.ZZZ = fp_to_fp28(a.ZZZ);
    return r;
}
__device__ __forceinline__ XYZZ<Fp> xyzz28_to_fp(const XYZZ<Fp28> &a) {
    XYZZ<Fp> r;
    if (fp28_all_zero(a.ZZ)) return XYZZ<Fp>::inf();
    r.X = fp28_to_fp(a.X); r.Y = fp28_to_fp(a.Y); r.ZZ = fp28_to_fp(a.ZZ); r.ZZZ = fp28_to_fp(a.ZZZ);
    return r;
}
// Doubling in the 28-bit form (dbl-2008-s-1, a = 0), for the equal-x case of the full addition below: the SAME product routines as every
// other step, so the cold branch is a straight run of register-only routine entries -- no function call, no scratch, no conversion to the
// 12 x 32-bit form (round 3 took a real call here: DESIGN.md 3.7).  Input under the invariants of the header, output inside them again:
//   U = 2 Y1 (limbs < 2^29, < 16p),  V = U^2 < 1.11p (29+29),  W = U V < 1.01p,  S = X1 V < 1.005p,
//   M = 3 X1^2 carried to tight limbs, < 3.2p,  X3 = M^2 + 8p - 2S < 9.01p (tight),
//   Y3 = [M (S + 32p - X3) + (32p - Y1) W] / 2^392 + p < 1.06p      limb bounds 28+30 and 29+28 as in madd28
//   ZZ3 = V ZZ1, ZZZ3 = W ZZZ1 < 1.01p.          tests/test_fp28_bounds.py: dbl28 (exact limb model + worst-case states)
__device__ __forceinline__ XYZZ<Fp28> xyzz_dbl28(const XYZZ<Fp28> &a) {
    XYZZ<Fp28> r;
    Fp28 U;
#pragma unroll
    for (int i = 0; i < 14; i++) U.l[i] = 2u * a.Y.l[i];
    Fp28 V = mul28(U, U);
    Fp28 W = mul28(U, V);
    Fp28 S = mul28(a.X, V);
    Fp28 M = sqr28(a.X);
#pragma unroll
    for (int i = 0; i < 14; i++) M.l[i] *= 3u;
    M = norm28(M);
    Fp28 S2;
#pragma unroll
    for (int i = 0; i < 14; i++) S2.l[i] = 2u * S.l[i];
    r.X = norm28(sub28(sqr28(M), FP28_K8_L4, S2));
    r.Y = mul28x2(M, sub28(S, FP28_K32_L1, r.X), neg28(FP28_K32_L1, a.Y), W);
    r.ZZ = mul28(V, a.ZZ);
    r.ZZZ = mul28(W, a.ZZZ);
    return r;
}
// acc += q, both XYZZ.  Equal x (P = 0: coinciding partial sums -- duplicated bases, a witness of few distinct values) is decided exactly
// (PP and R^2 are product outputs: zero iff 0 or p) and finished in place: equal y doubles q, opposite y leaves infinity.
__device__ __forceinline__ void xyzz_add(XYZZ<Fp28> &acc, const XYZZ<Fp28> &q) {
    if (fp28_all_zero(q.ZZ)) return;
    if (fp28_all_zero(acc.ZZ)) { acc = q; return; }
    Fp28 U1 = mul28(acc.X, q.ZZ), U2 = mul28(q.X, acc.ZZ);
    Fp28 S1 = mul28(acc.Y, q.ZZZ), S2 = mul28(q.Y, acc.ZZZ);
    Fp28 P = norm28(sub28(U2, FP28_K8_L1, U1));
    Fp28 PP = sqr28(P);
    Fp28 R = norm28(sub28(S2, FP28_K8_L1, S1));
    if (fp28_product_is_zero(PP)) {
        if (fp28_product_is_zero(sqr28(R))) acc = xyzz_dbl28(q); else acc = XYZZ<Fp28>::inf();
        return;
    }
    Fp28 PPP = mul28(P, PP);
    Fp28 Q = mul28(U1, PP);
    Fp28 s;
#pragma unroll
    for (int i = 0; i < 14; i++) s.l[i] = PPP.l[i] + 2u * Q.l[i];
    Fp28 X3 = norm28(sub28(sqr28(R), FP28_K8_L4, s));
    acc.Y = mul28x2(R, sub28(Q, FP28_K32_L1, X3), neg28(FP28_K8_L1, S1), PPP);
    acc.X = X3;
    acc.ZZ = mul28(mul28(acc.ZZ, q.ZZ), PP);
    acc.ZZZ = mul28(mul28(acc.ZZZ, q.ZZZ), PPP);
}

// ================================================================ G2: Fp2 over the 28-bit form, one Fp2 value per lane pair
// (even lane c0, odd lane c1, as Fp2L in field.h).  A product is ONE dual product per lane (schoolbook, one reduction):
//   even lane  a0 b0 + a1 (K - b1),   odd lane  a0 b1 + a1 b0;     a square is one single product per lane:
//   even lane  (a0 + a1)(a0 + K - a1),   odd lane  a0 * 2 a1.
// Every component obeys the bounds of the G1 path; the places where they differ are annotated.

__device__ __forceinline__ Fp28 partner28(const Fp28 &a) {
    Fp28 r;
#pragma unroll
    for (int i = 0; i < 14; i++) r.l[i] = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)a.l[i], 0xB1, 0xF, 0xF, false);
    return r;
}
__device__ __forceinline__ Fp28 select28(bool c, const Fp28 &a, const Fp28 &b) {
    Fp28 r;
#pragma unroll
    for (int i = 0; i < 14; i++) r.l[i] = c ? a.l[i] : b.l[i];
    return r;
}
// own components a, b of two Fp2 values -> own component of their product.  KB negates the partner's b (it must dominate it);
// limb bounds: E(a) + E(b) <= 59 and E(a) + E(KB - b) <= 59 over both lanes
__device__ __forceinline__ Fp28 mulF2(const Fp28 &a, const Fp28 &b, const uint32_t (&KB)[14]) {
    const bool hi = (threadIdx.x & 1) != 0;
    Fp28 ap = partner28(a), bp = partner28(b);
    // own a times (own b | partner's b), partner's a times (K - partner's b | own b): the a side needs no selection at all (round 3 selected
    // both sides: 14 more v_cndmask per product)
    Fp28 s1 = select28(hi, bp, b);
    Fp28 s2 = select28(hi, b, neg28(KB, bp));
    return mul28x2(a, s1, ap, s2);               // even: a0 b0 + a1 (K - b1)     odd: a1 b0 + a0 b1
}
// KA dominates the partner component of a
__device__ __forceinline__ Fp28 sqrF2(const Fp28 &a, const uint32_t (&KA)[14]) {
    const bool hi = (threadIdx.x & 1) != 0;
    Fp28 ap = partner28(a), sum, dbl;
#pragma unroll
    for (int i = 0; i < 14; i++) { sum.l[i] = a.l[i] + ap.l[i]; dbl.l[i] = 2u * a.l[i]; }
    Fp28 x = select28(hi, ap, sum);              // odd: a0            even: a0 + a1
    Fp28 y = select28(hi, dbl, sub28(a, KA, ap));// odd: 2 a1          even: a0 + K - a1
    return mul28(x, y);
}
// a tight product output below 4p is congruent to zero exactly when it is 0, p, 2p or 3p; the Fp2 value is zero when both are
__device__ __forceinline__ bool fp28_small_is_zero(const Fp28 &a) {
    return fp28_all_zero(a) || fp28_equals(a, FP28_P) || fp28_equals(a, FP28_2P) || fp28_equals(a, FP28_3P);
}
__device__ __forceinline__ bool pair_all28(bool c) {
    int f = c ? 1 : 0;
    return (f & __builtin_amdgcn_update_dpp(0, f, 0xB1, 0xF, 0xF, false)) != 0;
}
struct AffineHalf28 { Fp28 x, y; };              // this lane's components of an affine G2 point
struct XYZZHalf28 { Fp28 X, Y, ZZ, ZZZ; };
__device__ __forceinline__ XYZZHalf28 xyzz_half28_inf() { XYZZHalf28 r; r.X = r.Y = r.ZZ = r.ZZZ = fp28_zero(); return r; }

// acc += q over Fp2 (madd-2008-s as madd28 above).  Component bounds where they differ from the G1 path:
//   PP, RR = squares: even lane (a0 + a1)(a0 + 64p - a1) with a < 33.5p -> < (67 * 97.5 / 2521 + 1) p = 3.6p  (zero test: 0, p, 2p, 3p)
//   X3 = RR + 8p - (PPP + 2Q) < 11.7p (tight);   Y3 = t1 + 8p - t2 < 10.4p (t1 = R D < (33.5 * 33.1 + 33.5 * 64) / 2521 p + p = 2.3p),
//   limbs < 2^30 (R subtracts it with FP28_K32_L4).  tests/test_fp28_bounds.py propagates these worst cases mechanically.
// Returns false in the equal-x case (both lanes alike).
__device__ __forceinline__ bool madd28_g2(XYZZHalf28 &acc, const AffineHalf28 &q, bool negate) {
    const bool hi = (threadIdx.x & 1) != 0;
    if (pair_all28(fp28_all_zero(q.x) && fp28_all_zero(q.y))) return true;                    // infinity
    Fp28 qy = negate ? neg28(FP28_K8_L1, q.y) : q.y;                                           // limbs < 2^29, < 8p
    if (pair_all28(fp28_all_zero(acc.ZZ))) {
        acc.X = q.x; acc.Y = qy;
        acc.ZZ = hi ? fp28_zero() : fp28_const(FP28_ONE); acc.ZZZ = acc.ZZ;
        return true;
    }
    Fp28 U2 = mulF2(q.x, acc.ZZ, FP28_K8_L1);                                                  // ZZ tight < 4p
    Fp28 S2 = mulF2(qy, acc.ZZZ, FP28_K8_L1);                                                  // 29+28, 29+29
    Fp28 P = norm28(sub28(U2, FP28_K32_L1, acc.X));                                            // X tight < 16p -> P tight < 33.1p
    Fp28 PP = sqrF2(P, FP28_K64_L1);                                                           // 29 + 30; < 3.6p
    if (pair_all28(fp28_small_is_zero(PP))) return false;
    Fp28 R = norm28(sub28(S2, FP28_K32_L4, acc.Y));                                            // Y limbs < 2^30, < 16p -> R tight < 33.5p
    Fp28 PPP = mulF2(P, PP, FP28_K8_L1);                                                       // PP tight < 4p
    Fp28 Q = mulF2(acc.X, PP, FP28_K8_L1);
    Fp28 s;
#pragma unroll
    for (int i = 0; i < 14; i++) s.l[i] = PPP.l[i] + 2u * Q.l[i];                             // limbs < 3 * 2^28, < 4p
    Fp28 X3 = norm28(sub28(sqrF2(R, FP28_K64_L1), FP28_K8_L4, s));                             // tight < 11.7p
    Fp28 D = sub28(Q, FP28_K32_L1, X3);                                                        // limbs < 2^30, < 33.1p
    Fp28 t1 = mulF2(R, D, FP28_K64_L4);                                                        // 28+30, 28+31 (64p - D: limbs < 5 * 2^28)
    Fp28 t2 = mulF2(acc.Y, PPP, FP28_K8_L1);                                                   // 30+28, 30+29
    acc.Y = sub28(t1, FP28_K8_L1, t2);                                                         // limbs < 2^30, < 10.4p
    acc.X = X3;
    acc.ZZ = mulF2(acc.ZZ, PP, FP28_K8_L1);
    acc.ZZZ = mulF2(acc.ZZZ, PPP, FP28_K8_L1);
    return true;
}

// acc += q, both XYZZ over Fp2 on lane pairs (add-2008-s), for the G2 merges and bucket reduction.  Invariants as madd28_g2 keeps them
// (X tight < 11.7p, Y limbs < 2^30 and < 10.4p, ZZ, ZZZ tight < 3.9p), and the result satisfies them again:
//   U1 = X1 ZZ2, U2 = X2 ZZ1 (28+28, 28+29) < 1.06p;  S1 = Y1 ZZZ2, S2 = Y2 ZZZ1 (30+28, 30+29) < 1.05p
//   P = U2 + 8p - U1, R = S2 + 8p - S1: tight after the carry pass, < 9.1p;  PP, RR = squares with K = 32p: (29, 29.6), < 1.3p
//   PPP = P PP, Q = U1 PP < 1.04p;  X3 = RR + 8p - (PPP + 2Q) < 9.3p;  D = Q + 32p - X3 (limbs < 2^30)
//   Y3 = R D + 8p - S1 PPP: t1 < 1.36p (28+30, 28+30.3), limbs of Y3 < 2^30, value < 9.4p;  ZZ3, ZZZ3 = two products each, < 1.01p
// Equal x is finished in place as on G1: equal y doubles q (xyzz_dbl28_g2), opposite y leaves infinity.
__device__ __forceinline__ bool is_zero(const Fp28L &a) { return pair_all28(fp28_all_zero(a.v)); }
// Doubling over Fp2 on lane pairs (dbl-2008-s-1, a = 0) under the G2 invariants (X tight < 11.7p, Y limbs < 2^30 and < 10.4p, ZZ, ZZZ tight < 3.9p):
//   Yn = Y carried to tight limbs;  U = 2 Yn carried again (tight, < 20.8p);  V = U^2 with K = 32p (29 + 30) < 1.9p;
//   W = U V < 1.1p,  S = X1 V < 1.06p;  M = 3 X1^2 (square with K = 32p, < 1.41p each) carried to tight limbs, < 4.3p;
//   X3 = M^2 + 8p - 2S < 9.1p (tight);  D = S + 32p - X3 (limbs < 2^30);  Y3 = M D + 8p - Yn W: limbs < 2^30, < 9.2p;
//   ZZ3 = V ZZ1, ZZZ3 = W ZZZ1 < 1.02p.          tests/test_fp28_bounds.py: dbl28_g2
__device__ __forceinline__ XYZZ<Fp28L> xyzz_dbl28_g2(const XYZZ<Fp28L> &a) {
    XYZZ<Fp28L> r;
    Fp28 Yn = norm28(a.Y.v), U;
#pragma unroll
    for (int i = 0; i < 14; i++) U.l[i] = 2u * Yn.l[i];
    U = norm28(U);
    Fp28 V = sqrF2(U, FP28_K32_L1);
    Fp28 W = mulF2(U, V, FP28_K8_L1);
    Fp28 S = mulF2(a.X.v, V, FP28_K8_L1);
    Fp28 M = sqrF2(a.X.v, FP28_K32_L1);
#pragma unroll
    for (int i = 0; i < 14; i++) M.l[i] *= 3u;
    M = norm28(M);
    Fp28 S2;
#pragma unroll
    for (int i = 0; i < 14; i++) S2.l[i] = 2u * S.l[i];
    Fp28 X3 = norm28(sub28(sqrF2(M, FP28_K8_L1), FP28_K8_L4, S2));
    Fp28 D = sub28(S, FP28_K32_L1, X3);
    Fp28 t1 = mulF2(M, D, FP28_K64_L4);
    Fp28 t2 = mulF2(Yn, W, FP28_K8_L1);
    r.X.v = X3;
    r.Y.v = sub28(t1, FP28_K8_L1, t2);
    r.ZZ.v = mulF2(V, a.ZZ.v, FP28_K8_L1);
    r.ZZZ.v = mulF2(W, a.ZZZ.v, FP28_K8_L1);
    return r;
}
__device__ __forceinline__ void xyzz_add(XYZZ<Fp28L> &acc, const XYZZ<Fp28L> &q) {
    if (pair_all28(fp28_all_zero(q.ZZ.v))) return;
    if (pair_all28(fp28_all_zero(acc.ZZ.v))) { acc = q; return; }
    Fp28 U1 = mulF2(acc.X.v, q.ZZ.v, FP28_K8_L1), U2 = mulF2(q.X.v, acc.ZZ.v, FP28_K8_L1);
    Fp28 S1 = mulF2(acc.Y.v, q.ZZZ.v, FP28_K8_L1), S2 = mulF2(q.Y.v, acc.ZZZ.v, FP28_K8_L1);
    Fp28 P = norm28(sub28(U2, FP28_K8_L1, U1));
    Fp28 PP = sqrF2(P, FP28_K32_L1);
    Fp28 R = norm28(sub28(S2, FP28_K8_L1, S1));
    if (pair_all28(fp28_product_is_zero(PP))) {
        if (pair_all28(fp28_product_is_zero(sqrF2(R, FP28_K32_L1)))) acc = xyzz_dbl28_g2(q); else acc = XYZZ<Fp28L>::inf();
        return;
    }
    Fp28 PPP = mulF2(P, PP, FP28_K8_L1);
    Fp28 Q = mulF2(U1, PP, FP28_K8_L1);
    Fp28 s;
#pragma unroll
    for (int i = 0; i < 14; i++) s.l[i] = PPP.l[i] + 2u * Q.l[i];
    Fp28 X3 = norm28(sub28(sqrF2(R, FP28_K32_L1), FP28_K8_L4, s));
    Fp28 D = sub28(Q, FP28_K32_L1, X3);
    Fp28 t1 = mulF2(R, D, FP28_K64_L4);
    Fp28 t2 = mulF2(S1, PPP, FP28_K8_L1);
    acc.Y.v = sub28(t1, FP28_K8_L1, t2);
    acc.X.v = X3;
    acc.ZZ.v = mulF2(mulF2(acc.ZZ.v, q.ZZ.v, FP28_K8_L1), PP, FP28_K8_L1);
    acc.ZZZ.v = mulF2(mulF2(acc.ZZZ.v, q.ZZZ.v, FP28_K8_L1), PPP, FP28_K8_L1);
}
#else
// host pass: kernels instantiated over XYZZ<Fp28> / XYZZ<Fp28L> are only type-checked here (the product routines exist in the device
// pass alone)
__device__ __forceinline__ void xyzz_add(XYZZ<Fp28> &, const XYZZ<Fp28> &) {}
__device__ __forceinline__ void xyzz_add(XYZZ<Fp28L> &, const XYZZ<Fp28L> &) {}
#endif

}  // namespace vsp
