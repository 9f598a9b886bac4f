// Fr on 9 x 29-bit limbs with lazy reduction, for the NTT butterflies (the scheme fp28.h uses for Fp in the bucket accumulation).
//
// Why: the 8 x 32-bit Montgomery product is 128 v_mad_u64_u32 + 128 v_addc_co_u32 + a conditional subtraction (~320 instructions),
// and every butterfly addition / subtraction is a full modular one (~28 instructions).  With 29-bit limbs a column of the product
// (9 + 9 partial products below 2^60) fits one 64-bit accumulator: 162 mads, no carries, no final subtraction (208 instructions);
// r = 1 mod 2^32 makes m_k a negation.  Additions are limb-wise, a subtraction is a + K - b with K = 2r in a redundant form whose
// limbs dominate those of b; one carry pass per butterfly output brings limbs back to 29 bits.  A radix-4 butterfly is ~850
// instructions instead of ~1180.
//
//   value form   data values are plain residues (NOT Montgomery): canonical on entry, "lazy" inside a transform -- congruent mod r,
//                limbs below 2^29 ("tight", top limb the rest), value below 48 r < 2^261 = 70.4 r
//   twiddles     Montgomery with R' = 2^261, canonical (< r), tight: mul29(x, w R') = x w + (multiple of r), below x r / 2^261 + r
//   products     operand limb bounds 2^Ea, 2^Eb with Ea + Eb <= 60 (a twiddle is tight, so a data operand may carry limbs up to 2^31);
//                output tight, value < 1.65 r for a data operand below 46 r
//   a - b        = a + K2 - b limb by limb, K2 = 2r with every limb but the top raised by 2^29: b must be a product output (tight,
//                < 1.65 r < 2r), so no limb goes negative
//   growth       a radix-4 step takes the bound V of its inputs to V + 4r; 11 steps from 1.02 r stay below 46 r
// tests/test_fr29_bounds.py checks these bounds with an exact model of the routine and a worst-case propagation.
#pragma once
#include "field.h"

namespace vsp {

struct Fr29 { uint32_t l[9]; };

#if defined(__HIP_DEVICE_COMPILE__)
static constexpr uint32_t FR29_MASK = 0x1FFFFFFFu;

__device__ __forceinline__ Fr29 fr29_const(const uint32_t (&c)[9]) { Fr29 r; for (int i = 0; i < 9; i++) r.l[i] = c[i]; return r; }

// canonical 8 x 32-bit words -> 9 x 29-bit limbs (a bit slice; value unchanged)
__device__ __forceinline__ Fr29 fr29_from_words(const Fr &c) {
    Fr29 s;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        const int bit = 29 * i, w = bit >> 5, sh = bit & 31;
        uint64_t v = c.l[w];
        if (w + 1 < 8) v |= (uint64_t)c.l[w + 1] << 32;
        s.l[i] = (uint32_t)(v >> sh) & (i < 8 ? FR29_MASK : 0xFFFFFFFFu);
    }
    return s;
}
// tight limbs with value below 2^256 -> 8 x 32-bit words
__device__ __forceinline__ Fr fr29_to_words(const Fr29 &v) {
    Fr c;
#pragma unroll
    for (int w = 0; w < 8; w++) {
        uint32_t acc = 0;
#pragma unroll
        for (int i = 0; i < 9; i++) {
            const int lo = 29 * i - 32 * w;
            if (lo > -29 && lo < 32) acc |= lo >= 0 ? (v.l[i] << lo) : (v.l[i] >> (-lo));
        }
        c.l[w] = acc;
    }
    return c;
}
#if defined(VSP_PORTABLE_MUL)
// DIAGNOSTIC BUILD (`make portable`): vsp_mm29's column schedule in plain C++ (r = 1 mod 2^32 makes m_k the negated column)
__device__ __noinline__ void mont29_portable(uint32_t *r, const uint32_t *a, const uint32_t *b) {
    uint32_t m[9];
    uint64_t acc = 0;
    for (int k = 0; k < 17; k++) {
        for (int i = (k > 8 ? k - 8 : 0); i <= (k < 8 ? k : 8); i++) acc += (uint64_t)a[i] * b[k - i];
        if (k < 9) {
            for (int i = 0; i < k; i++) acc += (uint64_t)m[i] * FR29_R[k - i];
            m[k] = (0u - (uint32_t)acc) & FR29_MASK;
            acc += (uint64_t)m[k] * FR29_R[0];
        } else {
            for (int i = k - 8; i < 9; i++) acc += (uint64_t)m[i] * FR29_R[k - i];
            r[k - 9] = (uint32_t)acc & FR29_MASK;
        }
        acc >>= 29;
    }
    r[8] = (uint32_t)acc;
}
__device__ __forceinline__ Fr29 mul29(const Fr29 &a, const Fr29 &b) { Fr29 r; mont29_portable(r.l, a.l, b.l); return r; }
__device__ __forceinline__ Fr29 mul29q(const Fr29 &a, const Fr29 &b) { return mul29(a, b); }
#else
__device__ __forceinline__ Fr29 mul29(const Fr29 &a, const Fr29 &b) {
    Fr29 r;
    __builtin_amdgcn_sched_barrier(0);          // as fp28.h: this toolchain's machine scheduler must not move code across the call
    mont_mul29_asm(r.l, a.l, b.l);
    __builtin_amdgcn_sched_barrier(0);
    (void)&mont_mul29_holder<0>;
    return r;
}
// the same product through the routine's second register map (vsp_mm29q: operand and result in registers of their own, b shared): the
// second product of a pair, so that neither its operand nor the first product's result has to be copied around the call
__device__ __forceinline__ Fr29 mul29q(const Fr29 &a, const Fr29 &b) {
    Fr29 r;
    __builtin_amdgcn_sched_barrier(0);
    mont_mul29q_asm(r.l, a.l, b.l);
    __builtin_amdgcn_sched_barrier(0);
    (void)&mont_mul29q_holder<0>;
    return r;
}
#endif
__device__ __forceinline__ Fr29 add29(const Fr29 &a, const Fr29 &b) {
    Fr29 r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.l[i] = a.l[i] + b.l[i];
    return r;
}
// a + 2r - b, b a product output
__device__ __forceinline__ Fr29 sub29(const Fr29 &a, const Fr29 &b) {
    Fr29 r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.l[i] = a.l[i] + FR29_K2_L1[i] - b.l[i];
    return r;
}
// carry pass: limbs below 2^32 - 8 -> tight, same value
__device__ __forceinline__ Fr29 norm29(const Fr29 &a) {
    Fr29 r;
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) { uint32_t t = a.l[i] + c; r.l[i] = t & FR29_MASK; c = t >> 29; }
    r.l[8] = a.l[8] + c;
    return r;
}
// v tight, value < 2r  ->  canonical (< r)
__device__ __forceinline__ Fr29 csub29(const Fr29 &v) {
    Fr29 d; uint32_t borrow = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        uint32_t t = v.l[i] - FR29_R[i] - borrow;
        borrow = t >> 31;                          // limbs are below 2^29 (top: 2^25), so a wrapped difference has bit 31 set
        d.l[i] = i < 8 ? (t & FR29_MASK) : t;
    }
    Fr29 r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.l[i] = borrow ? v.l[i] : d.l[i];
    return r;
}
// Montgomery form (R = 2^256, 8 x 32-bit words, as every other table of the library) -> canonical Montgomery form for R' = 2^261
__device__ __forceinline__ Fr29 fr29_from_mont256(const Fr &m) {
    Fr c = from_mont(m);
    return csub29(mul29(fr29_from_words(c), fr29_const(FR29_R2)));      // x R'^2 / R' = x R', below 2r -> below r
}
#endif

}  // namespace vsp
