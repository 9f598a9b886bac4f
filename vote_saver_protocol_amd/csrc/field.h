// Montgomery prime-field arithmetic for BLS12-381 Fp (381 bit) and Fr (255 bit), shared by the
// gfx950 kernels (32-bit limbs: the product of two limbs is one v_mad_u64_u32) and by the host-side
// finishing code of the same library (64-bit limbs).  No floating point anywhere; all results are
// canonical residues.
//
// Replaces, for the hot path only, the field layer the reference reaches through
// crypto3-algebra fields/detail/element/fp{,2}.hpp over crypto3-multiprecision's modular_adaptor
// (absent submodules, /root/reference/.gitmodules:5-9; `.data` accesses at
// bin/cli/include/nil/vote_saver/common.hpp:92,101).
#pragma once
#include <stdint.h>
#include <stddef.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define VSP_HD __host__ __device__ __forceinline__
// The Montgomery product is ~1.2k VALU instructions once unrolled; it is kept as ONE out-of-line
// function per field so that a curve addition (14-42 products) is a few hundred instructions plus
// calls, and the product's body stays resident in the instruction cache.
#define VSP_HD_CALL __host__ __device__ __attribute__((noinline))
#else
#define VSP_HD inline
#define VSP_HD_CALL inline
#endif

namespace vsp {

template <class L> struct WideOf;
template <> struct WideOf<uint32_t> { using T = uint64_t; };
template <> struct WideOf<uint64_t> { using T = unsigned __int128; };

// ---------------------------------------------------------------- parameter packs
struct FpP32 {
    using limb_t = uint32_t;
    static constexpr int N = 12;
    static constexpr limb_t MOD[N] = {0xffffaaabu, 0xb9feffffu, 0xb153ffffu, 0x1eabfffeu, 0xf6b0f624u, 0x6730d2a0u,
                                      0xf38512bfu, 0x64774b84u, 0x434bacd7u, 0x4b1ba7b6u, 0x397fe69au, 0x1a0111eau};
    static constexpr limb_t ONE[N] = {0x0002fffdu, 0x76090000u, 0xc40c0002u, 0xebf4000bu, 0x53c758bau, 0x5f489857u,
                                      0x70525745u, 0x77ce5853u, 0xa256ec6du, 0x5c071a97u, 0xfa80e493u, 0x15f65ec3u};
    static constexpr limb_t R2[N] = {0x1c341746u, 0xf4df1f34u, 0x09d104f1u, 0x0a76e6a6u, 0x4c95b6d5u, 0x8de5476cu,
                                     0x939d83c0u, 0x67eb88a9u, 0xb519952du, 0x9a793e85u, 0x92cae3aau, 0x11988fe5u};
    static constexpr limb_t INV = 0xfffcfffdu;
};
struct FrP32 {
    using limb_t = uint32_t;
    static constexpr int N = 8;
    static constexpr limb_t MOD[N] = {0x00000001u, 0xffffffffu, 0xfffe5bfeu, 0x53bda402u, 0x09a1d805u, 0x3339d808u, 0x299d7d48u, 0x73eda753u};
    static constexpr limb_t ONE[N] = {0xfffffffeu, 0x00000001u, 0x00034802u, 0x5884b7fau, 0xecbc4ff5u, 0x998c4fefu, 0xacc5056fu, 0x1824b159u};
    static constexpr limb_t R2[N] = {0xf3f29c6du, 0xc999e990u, 0x87925c23u, 0x2b6cedcbu, 0x7254398fu, 0x05d31496u, 0x9f59ff11u, 0x0748d9d9u};
    static constexpr limb_t INV = 0xffffffffu;
};
struct FpP64 {
    using limb_t = uint64_t;
    static constexpr int N = 6;
    static constexpr limb_t MOD[N] = {0xb9feffffffffaaabULL, 0x1eabfffeb153ffffULL, 0x6730d2a0f6b0f624ULL, 0x64774b84f38512bfULL, 0x4b1ba7b6434bacd7ULL, 0x1a0111ea397fe69aULL};
    static constexpr limb_t ONE[N] = {0x760900000002fffdULL, 0xebf4000bc40c0002ULL, 0x5f48985753c758baULL, 0x77ce585370525745ULL, 0x5c071a97a256ec6dULL, 0x15f65ec3fa80e493ULL};
    static constexpr limb_t R2[N] = {0xf4df1f341c341746ULL, 0x0a76e6a609d104f1ULL, 0x8de5476c4c95b6d5ULL, 0x67eb88a9939d83c0ULL, 0x9a793e85b519952dULL, 0x11988fe592cae3aaULL};
    static constexpr limb_t INV = 0x89f3fffcfffcfffdULL;
};
struct FrP64 {
    using limb_t = uint64_t;
    static constexpr int N = 4;
    static constexpr limb_t MOD[N] = {0xffffffff00000001ULL, 0x53bda402fffe5bfeULL, 0x3339d80809a1d805ULL, 0x73eda753299d7d48ULL};
    static constexpr limb_t ONE[N] = {0x00000001fffffffeULL, 0x5884b7fa00034802ULL, 0x998c4fefecbc4ff5ULL, 0x1824b159acc5056fULL};
    static constexpr limb_t R2[N] = {0xc999e990f3f29c6dULL, 0x2b6cedcb87925c23ULL, 0x05d314967254398fULL, 0x0748d9d99f59ff11ULL};
    static constexpr limb_t INV = 0xfffffffeffffffffULL;
};

// ---------------------------------------------------------------- element type
// Value is stored in Montgomery form (x * 2^(bits*N) mod p) unless a function says "canonical".
template <class P>
struct alignas(16) Mont {
    using Params = P;
    using L = typename P::limb_t;
    using W = typename WideOf<L>::T;
    static constexpr int N = P::N;
    static constexpr int LB = sizeof(L) * 8;
    L l[N];

    VSP_HD static Mont zero() { Mont r; for (int i = 0; i < N; i++) r.l[i] = 0; return r; }
    VSP_HD static Mont one() { Mont r; for (int i = 0; i < N; i++) r.l[i] = P::ONE[i]; return r; }
    VSP_HD static Mont r2() { Mont r; for (int i = 0; i < N; i++) r.l[i] = P::R2[i]; return r; }
    VSP_HD static Mont raw_one() { Mont r = zero(); r.l[0] = 1; return r; }
};

template <class P> VSP_HD bool is_zero(const Mont<P> &a) {
    typename P::limb_t o = 0;
#pragma unroll
    for (int i = 0; i < P::N; i++) o |= a.l[i];
    return o == 0;
}
template <class P> VSP_HD bool eq(const Mont<P> &a, const Mont<P> &b) {
    typename P::limb_t o = 0;
#pragma unroll
    for (int i = 0; i < P::N; i++) o |= a.l[i] ^ b.l[i];
    return o == 0;
}

// r = t - MOD if t >= MOD else t   (t < 2*MOD)
template <class P> VSP_HD Mont<P> reduce_once(const Mont<P> &t) {
    using L = typename P::limb_t; using W = typename WideOf<L>::T;
    constexpr int N = P::N, LB = sizeof(L) * 8;
    Mont<P> u; L borrow = 0;
#pragma unroll
    for (int i = 0; i < N; i++) {
        W d = (W)t.l[i] - P::MOD[i] - borrow;
        u.l[i] = (L)d; borrow = (L)(d >> LB) & 1;
    }
    Mont<P> r;
#pragma unroll
    for (int i = 0; i < N; i++) r.l[i] = borrow ? t.l[i] : u.l[i];
    return r;
}

template <class P> VSP_HD Mont<P> add(const Mont<P> &a, const Mont<P> &b) {
    using L = typename P::limb_t; using W = typename WideOf<L>::T;
    constexpr int N = P::N, LB = sizeof(L) * 8;
    Mont<P> t; W c = 0;
#pragma unroll
    for (int i = 0; i < N; i++) { c += (W)a.l[i] + b.l[i]; t.l[i] = (L)c; c >>= LB; }
    return reduce_once(t);      // a + b < 2^(bits): both moduli leave a spare top bit
}
template <class P> VSP_HD Mont<P> dbl(const Mont<P> &a) { return add(a, a); }

template <class P> VSP_HD Mont<P> sub(const Mont<P> &a, const Mont<P> &b) {
    using L = typename P::limb_t; using W = typename WideOf<L>::T;
    constexpr int N = P::N, LB = sizeof(L) * 8;
    Mont<P> t; L borrow = 0;
#pragma unroll
    for (int i = 0; i < N; i++) {
        W d = (W)a.l[i] - b.l[i] - borrow;
        t.l[i] = (L)d; borrow = (L)(d >> LB) & 1;
    }
    Mont<P> r; W c = 0;
#pragma unroll
    for (int i = 0; i < N; i++) { c += (W)t.l[i] + (borrow ? P::MOD[i] : (L)0); r.l[i] = (L)c; c >>= LB; }
    return r;
}
template <class P> VSP_HD Mont<P> neg(const Mont<P> &a) { return sub(Mont<P>::zero(), a); }

// ---- Montgomery product a*b*R^-1 mod p -------------------------------------------------------------------
// Host (and the CPU test build): CIOS in portable C.  Both moduli have a zero top bit in their top limb, so the
// running value never needs limb N+1 ("no-carry" form).
template <class P> VSP_HD Mont<P> mul_cios(const Mont<P> &a, const Mont<P> &b) {
    using L = typename P::limb_t; using W = typename WideOf<L>::T;
    constexpr int N = P::N, LB = sizeof(L) * 8;
    L t[N + 1];
#pragma unroll
    for (int i = 0; i <= N; i++) t[i] = 0;
#pragma unroll
    for (int i = 0; i < N; i++) {
        L c = 0;
#pragma unroll
        for (int j = 0; j < N; j++) {
            W x = (W)a.l[j] * b.l[i] + t[j] + c;
            t[j] = (L)x; c = (L)(x >> LB);
        }
        L tn = t[N] + c;
        L m = t[0] * P::INV;
        W x = (W)m * P::MOD[0] + t[0];
        c = (L)(x >> LB);
#pragma unroll
        for (int j = 1; j < N; j++) {
            x = (W)m * P::MOD[j] + t[j] + c;
            t[j - 1] = (L)x; c = (L)(x >> LB);
        }
        x = (W)tn + c;
        t[N - 1] = (L)x; t[N] = (L)(x >> LB);
    }
    Mont<P> r;
#pragma unroll
    for (int i = 0; i < N; i++) r.l[i] = t[i];
    return reduce_once(r);
}

#if defined(__HIP_DEVICE_COMPILE__)
}  // namespace vsp
#include "mont_asm_gfx950.h"
namespace vsp {
// gfx950: finely integrated product scanning (Comba) as a generated asm routine with a private calling
// convention (tools/gen_mont_asm.py): exactly one v_mad_u64_u32 + one v_addc_co_u32 per limb product -- measured
// issue cost 6.6 + 4.6 cycles per wave against ~21 cycles per product for the mov/add sequence the compiler derives
// from the portable CIOS form (profiles/r1_ubench_valu.txt) -- entered with s_swappc_b64, clobbering only v0..v39.
template <class P> __device__ __forceinline__ Mont<P> mul_comba(const Mont<P> &a, const Mont<P> &b) {
    constexpr int N = P::N;
    uint32_t x[N], y[N];
    Mont<P> r;
#pragma unroll
    for (int i = 0; i < N; i++) { x[i] = a.l[i]; y[i] = b.l[i]; }
    if constexpr (N == 12) { mont_mul_asm_12<P>(r.l, x, y); (void)&mont_mul_holder_12<P>; }
    else { mont_mul_asm_8<P>(r.l, x, y); (void)&mont_mul_holder_8<P>; }
    return r;
}
#endif

// host: one out-of-line copy per field (the host-side group formulas would otherwise inline dozens of 72-multiply bodies
// each and take minutes to compile)
template <class P> __attribute__((noinline)) Mont<P> mul_host(const Mont<P> &a, const Mont<P> &b) { return mul_cios<P>(a, b); }

template <class P> VSP_HD Mont<P> mul(const Mont<P> &a, const Mont<P> &b) {
#if defined(__HIP_DEVICE_COMPILE__) && defined(VSP_PORTABLE_MUL)
    // DIAGNOSTIC BUILD (`make portable`): no hand-laid-out routine anywhere -- the compiler's own code for every field product, so that a
    // result can be cross-checked against a library in which the routine / compiler interface does not exist (missing 3)
    return mul_cios<P>(a, b);
#elif defined(__HIP_DEVICE_COMPILE__)
    if constexpr (sizeof(typename P::limb_t) == 4) return mul_comba<P>(a, b);
    else return mul_cios<P>(a, b);          // 64-bit-limb (host) types are never run on the device
#else
    return mul_host<P>(a, b);
#endif
}
template <class P> VSP_HD Mont<P> sqr(const Mont<P> &a) { return mul(a, a); }
// a*b + c*d.  (A dual-product routine -- both products into one set of column accumulators, one reduction, 432 limb products
// instead of 576 -- was built and measured in round 1: +4 % on the G1 accumulation.  It passed its field tests, but kernels using it
// computed wrong bucket sums in some code arrangements and not in others (same source, same inputs; neither tied operands, spills,
// scheduling nor volatility of the asm explained it), so it is not used: a prover must not depend on which arrangement compiles right.)
template <class P> VSP_HD Mont<P> mul_add2(const Mont<P> &a, const Mont<P> &b, const Mont<P> &c, const Mont<P> &d) { return add(mul(a, b), mul(c, d)); }

// canonical (plain residue, same limb layout) <-> Montgomery
template <class P> VSP_HD Mont<P> to_mont(const Mont<P> &canon) { return mul(canon, Mont<P>::r2()); }
template <class P> VSP_HD Mont<P> from_mont(const Mont<P> &m) { return mul(m, Mont<P>::raw_one()); }

// a^e, e given as little-endian limbs of the same limb type (host-side use: inversion)
template <class P> VSP_HD Mont<P> pow_limbs(const Mont<P> &a, const typename P::limb_t *e, int nlimbs) {
    constexpr int LB = sizeof(typename P::limb_t) * 8;
    Mont<P> acc = Mont<P>::one(), base = a;
    for (int i = 0; i < nlimbs * LB; i++) {
        if ((e[i / LB] >> (i % LB)) & 1) acc = mul(acc, base);
        base = sqr(base);
    }
    return acc;
}
template <class P> VSP_HD Mont<P> inv(const Mont<P> &a) {   // a^(p-2); 0 -> 0
    using L = typename P::limb_t;
    L e[P::N];
    for (int i = 0; i < P::N; i++) e[i] = P::MOD[i];
    L borrow = 2;                                           // e = p - 2 with borrow propagation
    for (int i = 0; i < P::N && borrow; i++) { L old = e[i]; e[i] = old - borrow; borrow = old < borrow ? 1 : 0; }
    return pow_limbs(a, e, P::N);
}

// ---------------------------------------------------------------- Fp2 = Fp[u]/(u^2+1)
template <class F>
struct alignas(16) Fp2T {
    F c0, c1;
    VSP_HD static Fp2T zero() { Fp2T r; r.c0 = F::zero(); r.c1 = F::zero(); return r; }
    VSP_HD static Fp2T one() { Fp2T r; r.c0 = F::one(); r.c1 = F::zero(); return r; }
};
template <class F> VSP_HD bool is_zero(const Fp2T<F> &a) { return is_zero(a.c0) && is_zero(a.c1); }
template <class F> VSP_HD bool eq(const Fp2T<F> &a, const Fp2T<F> &b) { return eq(a.c0, b.c0) && eq(a.c1, b.c1); }
template <class F> VSP_HD Fp2T<F> add(const Fp2T<F> &a, const Fp2T<F> &b) { Fp2T<F> r; r.c0 = add(a.c0, b.c0); r.c1 = add(a.c1, b.c1); return r; }
template <class F> VSP_HD Fp2T<F> sub(const Fp2T<F> &a, const Fp2T<F> &b) { Fp2T<F> r; r.c0 = sub(a.c0, b.c0); r.c1 = sub(a.c1, b.c1); return r; }
template <class F> VSP_HD Fp2T<F> neg(const Fp2T<F> &a) { Fp2T<F> r; r.c0 = neg(a.c0); r.c1 = neg(a.c1); return r; }
template <class F> VSP_HD Fp2T<F> dbl(const Fp2T<F> &a) { Fp2T<F> r; r.c0 = dbl(a.c0); r.c1 = dbl(a.c1); return r; }
template <class F> VSP_HD Fp2T<F> mul(const Fp2T<F> &a, const Fp2T<F> &b) {   // Karatsuba, 3 base multiplications
    F t0 = mul(a.c0, b.c0), t1 = mul(a.c1, b.c1);
    F t2 = mul(add(a.c0, a.c1), add(b.c0, b.c1));
    Fp2T<F> r; r.c0 = sub(t0, t1); r.c1 = sub(sub(t2, t0), t1); return r;
}
template <class F> VSP_HD Fp2T<F> sqr(const Fp2T<F> &a) {                      // 2 base multiplications
    F t = mul(a.c0, a.c1);
    Fp2T<F> r; r.c0 = mul(add(a.c0, a.c1), sub(a.c0, a.c1)); r.c1 = dbl(t); return r;
}
template <class F> VSP_HD Fp2T<F> mul_add2(const Fp2T<F> &a, const Fp2T<F> &b, const Fp2T<F> &c, const Fp2T<F> &d) { return add(mul(a, b), mul(c, d)); }
// The group formulas ask for their products through two helpers so that each field can batch them its own way:
//   mul_pair(a, b, c, d, p, q): p = a*b, q = c*d (independent products)        prod_diff(a, b, c, d) = a*b - c*d
template <class P> VSP_HD void mul_pair(const Mont<P> &a, const Mont<P> &b, const Mont<P> &c, const Mont<P> &d, Mont<P> &p, Mont<P> &q) { p = mul(a, b); q = mul(c, d); }
template <class P> VSP_HD Mont<P> prod_diff(const Mont<P> &a, const Mont<P> &b, const Mont<P> &c, const Mont<P> &d) { return sub(mul(a, b), mul(c, d)); }
template <class F> VSP_HD void mul_pair(const Fp2T<F> &a, const Fp2T<F> &b, const Fp2T<F> &c, const Fp2T<F> &d, Fp2T<F> &p, Fp2T<F> &q) { p = mul(a, b); q = mul(c, d); }
template <class F> VSP_HD Fp2T<F> prod_diff(const Fp2T<F> &a, const Fp2T<F> &b, const Fp2T<F> &c, const Fp2T<F> &d) { return sub(mul(a, b), mul(c, d)); }
template <class F> VSP_HD Fp2T<F> inv(const Fp2T<F> &a) {
    F n = inv(add(sqr(a.c0), sqr(a.c1)));
    Fp2T<F> r; r.c0 = mul(a.c0, n); r.c1 = neg(mul(a.c1, n)); return r;
}
template <class F> VSP_HD Fp2T<F> to_mont(const Fp2T<F> &a) { Fp2T<F> r; r.c0 = to_mont(a.c0); r.c1 = to_mont(a.c1); return r; }
template <class F> VSP_HD Fp2T<F> from_mont(const Fp2T<F> &a) { Fp2T<F> r; r.c0 = from_mont(a.c0); r.c1 = from_mont(a.c1); return r; }

// device-side types
using Fp = Mont<FpP32>;
using Fr = Mont<FrP32>;
using Fp2 = Fp2T<Fp>;
// host-side types (same bytes in memory: 2 x u32 little-endian == 1 x u64, same Montgomery radix)
using HFp = Mont<FpP64>;
using HFr = Mont<FrP64>;
using HFp2 = Fp2T<HFp>;

// ---------------------------------------------------------------- Fp2 split over a lane pair (device only)
// One Fp2 value lives in two adjacent lanes: the even lane holds c0, the odd lane c1.  Per-lane state is that of an Fp
// computation, so a kernel over Fp2L runs at the register budget (and occupancy) of the G1 kernels, where one lane holding
// whole Fp2 values needs > 256 VGPRs and is confined to one wave per SIMD.  Products exchange operands with the partner lane
// by DPP quad_perm [1,0,3,2]; every lane of a pair must execute the same calls (conditions on Fp2L values are combined over
// the pair, so data-dependent branches stay pair-uniform).
struct Fp2L {
    Fp v;
    VSP_HD static Fp2L zero() { Fp2L r; r.v = Fp::zero(); return r; }
#if defined(__HIPCC__)
    __device__ __forceinline__ static Fp2L one() { Fp2L r; r.v = (threadIdx.x & 1) ? Fp::zero() : Fp::one(); return r; }
#endif
};
#if defined(__HIPCC__)
// The machine scheduler of this toolchain (clang 22, ROCm 7.2) crashes in LiveIntervals::handleMove when it moves the DPP / select
// code of the lane-pair products across the fixed-register asm call of the base product; a scheduling fence on either side of
// each call keeps it from trying (the surrounding code is a few dozen moves, nothing is lost).
#define VSP_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
__device__ __forceinline__ Fp lane_partner(const Fp &a) {
    Fp r;
#pragma unroll
    for (int i = 0; i < Fp::N; i++) r.l[i] = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)a.l[i], 0xB1, 0xF, 0xF, false);
    return r;
}
__device__ __forceinline__ bool pair_all(bool c) {
    int f = c ? 1 : 0;
    return (f & __builtin_amdgcn_update_dpp(0, f, 0xB1, 0xF, 0xF, false)) != 0;
}
__device__ __forceinline__ Fp lane_select(bool c, const Fp &a, const Fp &b) {
    Fp r;
#pragma unroll
    for (int i = 0; i < Fp::N; i++) r.l[i] = c ? a.l[i] : b.l[i];
    return r;
}
__device__ __forceinline__ bool is_zero(const Fp2L &a) { return pair_all(is_zero(a.v)); }
__device__ __forceinline__ Fp2L add(const Fp2L &a, const Fp2L &b) { Fp2L r; r.v = add(a.v, b.v); return r; }
__device__ __forceinline__ Fp2L sub(const Fp2L &a, const Fp2L &b) { Fp2L r; r.v = sub(a.v, b.v); return r; }
__device__ __forceinline__ Fp2L neg(const Fp2L &a) { Fp2L r; r.v = neg(a.v); return r; }
__device__ __forceinline__ Fp2L dbl(const Fp2L &a) { Fp2L r; r.v = dbl(a.v); return r; }
// (a0 + a1 u)(b0 + b1 u):  even lane a0 b0 - a1 b1,  odd lane a0 b1 + a1 b0  -- two base products per lane
__device__ __forceinline__ Fp2L mul(const Fp2L &a, const Fp2L &b) {
    const bool hi = (threadIdx.x & 1) != 0;
    Fp ap = lane_partner(a.v), bp = lane_partner(b.v);
    Fp u = lane_select(hi, ap, a.v), w = lane_select(hi, a.v, ap);
    Fp nb = lane_select(hi, bp, neg(bp));
    VSP_SCHED_FENCE();                               // see VSP_SCHED_FENCE
    Fp p1 = mul(u, b.v);                             // even: a0 b0      odd: a0 b1
    VSP_SCHED_FENCE();
    Fp p2 = mul(w, nb);                              // even: a1 (-b1)   odd: a1 b0
    VSP_SCHED_FENCE();
    Fp2L r; r.v = add(p1, p2); return r;
}
// (a0 + a1 u)^2:  even lane (a0 + a1)(a0 - a1),  odd lane a0 * 2 a1  -- one base product per lane
// Two independent products p = a*b, q = c*d over the lane pair by Karatsuba: SIX base products for the two (three per lane) instead
// of eight.  Even lane: a0 b0, c0 d0, (a0 + a1)(b0 + b1); odd lane: a1 b1, c1 d1, (c0 + c1)(d0 + d1); then one exchange round:
//   p0 = a0 b0 - a1 b1      p1 = [(a0 + a1)(b0 + b1) - a0 b0] - a1 b1      q0 = c0 d0 - c1 d1      q1 = [(c0 + c1)(d0 + d1) - c1 d1] - c0 d0
__device__ __forceinline__ void mul_pair(const Fp2L &a, const Fp2L &b, const Fp2L &c, const Fp2L &d, Fp2L &p, Fp2L &q) {
    const bool hi = (threadIdx.x & 1) != 0;
    // the operand pair whose cross sum this lane forms: even (a, b), odd (c, d)
    Fp x = lane_select(hi, c.v, a.v), y = lane_select(hi, d.v, b.v);
    Fp xs = lane_select(hi, a.v, c.v), ys = lane_select(hi, b.v, d.v);      // what the partner needs from this lane
    Fp s1 = add(x, lane_partner(xs)), s2 = add(y, lane_partner(ys));         // even: a0 + a1, b0 + b1      odd: c1 + c0, d1 + d0
    VSP_SCHED_FENCE();
    Fp m1 = mul(a.v, b.v);                                                   // even: a0 b0    odd: a1 b1
    VSP_SCHED_FENCE();
    Fp m2 = mul(c.v, d.v);                                                   // even: c0 d0    odd: c1 d1
    VSP_SCHED_FENCE();
    Fp m3 = mul(s1, s2);
    VSP_SCHED_FENCE();
    Fp t = sub(m3, lane_select(hi, m2, m1));                                 // even: (a0+a1)(b0+b1) - a0 b0     odd: (c0+c1)(d0+d1) - c1 d1
    Fp m2r = lane_partner(m2);                                               // even: c1 d1    odd: c0 d0
    Fp yr = lane_partner(lane_select(hi, m1, t));                            // even receives a1 b1, odd receives the even lane's t
    p.v = sub(lane_select(hi, yr, m1), lane_select(hi, m1, yr));             // even: a0 b0 - a1 b1     odd: t_even - a1 b1
    q.v = sub(lane_select(hi, t, m2), m2r);                                  // even: c0 d0 - c1 d1     odd: t_odd - c0 d0
}
__device__ __forceinline__ Fp2L prod_diff(const Fp2L &a, const Fp2L &b, const Fp2L &c, const Fp2L &d) { Fp2L p, q; mul_pair(a, b, c, d, p, q); return sub(p, q); }
__device__ __forceinline__ Fp2L mul_add2(const Fp2L &a, const Fp2L &b, const Fp2L &c, const Fp2L &d) { Fp2L p, q; mul_pair(a, b, c, d, p, q); return add(p, q); }
__device__ __forceinline__ Fp2L sqr(const Fp2L &a) {
    const bool hi = (threadIdx.x & 1) != 0;
    Fp ap = lane_partner(a.v);
    Fp x = lane_select(hi, ap, add(a.v, ap));        // odd: a0           even: a0 + a1
    Fp y = lane_select(hi, dbl(a.v), sub(a.v, ap));  // odd: 2 a1         even: a0 - a1
    VSP_SCHED_FENCE();
    Fp2L r; r.v = mul(x, y);
    VSP_SCHED_FENCE();
    return r;
}
#endif

}  // namespace vsp
