// Short-Weierstrass a=0 group law (BLS12-381 G1 over Fp, G2 over Fp2) in XYZZ coordinates:
//   x = X/ZZ, y = Y/ZZZ, ZZ^3 = ZZZ^2, infinity <=> ZZ = 0.
// Mixed addition costs 8M+2S, full addition 12M+2S, doubling 6M+3S (EFD madd-2008-s, add-2008-s,
// dbl-2008-s-1) -- cheaper than the Jacobian form the reference's CPU code uses
// (crypto3-algebra curves/detail/forms/short_weierstrass/jacobian_with_a4_0, absent submodule;
// coordinate names at bin/cli/include/nil/vote_saver/common.hpp:117-121).  Results leave this
// library only as affine points (or as Jacobian X,Y,Z records for the multi-GPU exchange), which
// are unique, so the internal coordinate system does not affect parity.
//
// Generic over the field type F (device: Fp / Fp2 with 32-bit limbs; host: HFp / HFp2).
#pragma once
#include "field.h"
#include <vector>

namespace vsp {

// affine point; infinity is encoded as x = y = 0 (not on either curve since b != 0)
template <class F> struct alignas(16) Affine {
    F x, y;
};
template <class F> VSP_HD bool is_inf(const Affine<F> &p) { return is_zero(p.x) && is_zero(p.y); }

template <class F> struct alignas(16) XYZZ {
    F X, Y, ZZ, ZZZ;
    VSP_HD static XYZZ inf() { XYZZ r; r.X = F::zero(); r.Y = F::zero(); r.ZZ = F::zero(); r.ZZZ = F::zero(); return r; }
};
template <class F> VSP_HD bool is_inf(const XYZZ<F> &p) { return is_zero(p.ZZ); }

template <class F> VSP_HD XYZZ<F> xyzz_from_affine(const Affine<F> &p) {
    XYZZ<F> r;
    if (is_inf(p)) return XYZZ<F>::inf();
    r.X = p.x; r.Y = p.y; r.ZZ = F::one(); r.ZZZ = F::one();
    return r;
}

template <class F> VSP_HD XYZZ<F> xyzz_dbl(const XYZZ<F> &p) {
    if (is_inf(p)) return p;
    F U = dbl(p.Y);
    F V = sqr(U);
    F W, S;
    mul_pair(U, V, p.X, V, W, S);
    F XX = sqr(p.X);
    F M = add(dbl(XX), XX);
    XYZZ<F> r;
    r.X = sub(sqr(M), dbl(S));
    r.Y = prod_diff(M, sub(S, r.X), W, p.Y);
    mul_pair(V, p.ZZ, W, p.ZZZ, r.ZZ, r.ZZZ);
    return r;       // Y = 0 cannot occur: the groups have odd order
}

// doubling of an affine point (ZZ = ZZZ = 1)
template <class F> VSP_HD XYZZ<F> xyzz_dbl_affine(const Affine<F> &p) {
    F U = dbl(p.y);
    F V = sqr(U);
    F W, S;
    mul_pair(U, V, p.x, V, W, S);
    F XX = sqr(p.x);
    F M = add(dbl(XX), XX);
    XYZZ<F> r;
    r.X = sub(sqr(M), dbl(S));
    r.Y = prod_diff(M, sub(S, r.X), W, p.y);
    r.ZZ = V;
    r.ZZZ = W;
    return r;
}

// acc += q (q affine, optionally negated).  All exceptional cases handled.
template <class F> VSP_HD void xyzz_madd(XYZZ<F> &acc, const Affine<F> &q_in, bool negate = false) {
    if (is_inf(q_in)) return;
    Affine<F> q = q_in;
    if (negate) q.y = neg(q.y);
    if (is_inf(acc)) { acc.X = q.x; acc.Y = q.y; acc.ZZ = F::one(); acc.ZZZ = F::one(); return; }
    F U2, S2;
    mul_pair(q.x, acc.ZZ, q.y, acc.ZZZ, U2, S2);
    F P = sub(U2, acc.X);
    F R = sub(S2, acc.Y);
    if (is_zero(P)) {
        if (is_zero(R)) acc = xyzz_dbl_affine(q);
        else acc = XYZZ<F>::inf();
        return;
    }
    F PP = sqr(P);
    F PPP, Q;
    mul_pair(P, PP, acc.X, PP, PPP, Q);
    F X3 = sub(sub(sqr(R), PPP), dbl(Q));
    acc.Y = prod_diff(R, sub(Q, X3), acc.Y, PPP);            // R (Q - X3) - Y1 PPP
    acc.X = X3;
    mul_pair(acc.ZZ, PP, acc.ZZZ, PPP, acc.ZZ, acc.ZZZ);
}

// acc += q (both XYZZ)
template <class F> VSP_HD void xyzz_add(XYZZ<F> &acc, const XYZZ<F> &q) {
    if (is_inf(q)) return;
    if (is_inf(acc)) { acc = q; return; }
    F U1, U2, S1, S2;
    mul_pair(acc.X, q.ZZ, q.X, acc.ZZ, U1, U2);
    mul_pair(acc.Y, q.ZZZ, q.Y, acc.ZZZ, S1, S2);
    F P = sub(U2, U1);
    F R = sub(S2, S1);
    if (is_zero(P)) {
        if (is_zero(R)) acc = xyzz_dbl(acc);
        else acc = XYZZ<F>::inf();
        return;
    }
    F PP = sqr(P);
    F PPP, Q;
    mul_pair(P, PP, U1, PP, PPP, Q);
    F X3 = sub(sub(sqr(R), PPP), dbl(Q));
    acc.Y = prod_diff(R, sub(Q, X3), S1, PPP);
    acc.X = X3;
    F zz, zzz;
    mul_pair(acc.ZZ, q.ZZ, acc.ZZZ, q.ZZZ, zz, zzz);
    mul_pair(zz, PP, zzz, PPP, acc.ZZ, acc.ZZZ);
}

template <class F> VSP_HD XYZZ<F> xyzz_neg(const XYZZ<F> &p) { XYZZ<F> r = p; r.Y = neg(p.Y); return r; }

// ---- host-side helpers (used with HFp / HFp2; cheap, a handful of calls per MSM / proof) ----
template <class F> VSP_HD Affine<F> xyzz_to_affine(const XYZZ<F> &p) {
    Affine<F> r;
    if (is_inf(p)) { r.x = F::zero(); r.y = F::zero(); return r; }
    F zi = inv(p.ZZZ);                 // 1/Z^3
    F z = mul(zi, p.ZZ);               // Z^2/Z^3 = 1/Z
    F zi2 = sqr(z);                    // 1/Z^2
    r.x = mul(p.X, zi2);
    r.y = mul(p.Y, zi);
    return r;
}

// XYZZ -> Jacobian (X', Y', Z') with x = X'/Z'^2, y = Y'/Z'^3: take Z' = ZZZ, X' = X*ZZ^2, Y' = Y*ZZZ^2.
// This is the 144-byte (G1) / 288-byte (G2) partial-sum record exchanged between GPUs.
template <class F> struct alignas(16) Jacobian { F X, Y, Z; };
template <class F> VSP_HD Jacobian<F> xyzz_to_jacobian(const XYZZ<F> &p) {
    Jacobian<F> r;
    if (is_inf(p)) { r.X = F::one(); r.Y = F::one(); r.Z = F::zero(); return r; }
    r.X = mul(p.X, sqr(p.ZZ));
    r.Y = mul(p.Y, sqr(p.ZZZ));
    r.Z = p.ZZZ;
    return r;
}
template <class F> VSP_HD XYZZ<F> jacobian_to_xyzz(const Jacobian<F> &p) {
    XYZZ<F> r;
    if (is_zero(p.Z)) return XYZZ<F>::inf();
    r.X = p.X; r.Y = p.Y; r.ZZ = sqr(p.Z); r.ZZZ = mul(r.ZZ, p.Z);
    return r;
}

// k * p for a canonical scalar given as nbits little-endian bits in 64-bit words (host side)
template <class F> VSP_HD XYZZ<F> xyzz_mul_scalar(const XYZZ<F> &p, const uint64_t *k, int nbits) {
    XYZZ<F> acc = XYZZ<F>::inf();
    for (int i = nbits - 1; i >= 0; i--) {
        acc = xyzz_dbl(acc);
        if ((k[i >> 6] >> (i & 63)) & 1) xyzz_add(acc, p);
    }
    return acc;
}

// the same with 4-bit windows (host side: the prover's s * A and r * B1): 14 additions for the table, then 4 doublings and at most one
// addition per window -- 255 + 78 group operations instead of 255 + ~127
template <class F> VSP_HD XYZZ<F> xyzz_mul_scalar_w4(const XYZZ<F> &p, const uint64_t *k) {
    XYZZ<F> tab[16];
    tab[0] = XYZZ<F>::inf(); tab[1] = p;
    for (int d = 2; d < 16; d++) { tab[d] = tab[d - 1]; if (d & 1) xyzz_add(tab[d], p); else tab[d] = xyzz_dbl(tab[d >> 1]); }
    XYZZ<F> acc = XYZZ<F>::inf();
    for (int w = 63; w >= 0; w--) {
        if (w != 63) { acc = xyzz_dbl(acc); acc = xyzz_dbl(acc); acc = xyzz_dbl(acc); acc = xyzz_dbl(acc); }
        const unsigned d = (unsigned)(k[w >> 4] >> ((w & 15) * 4)) & 15u;
        if (d) xyzz_add(acc, tab[d]);
    }
    return acc;
}
// fixed base: tab[255 w + d - 1] = d 2^(8 w) P for d = 1..255, w = 0..31 (build: 255 additions per window, 8 doublings between windows);
// k * P = the sum of one entry per non-zero byte of k: at most 32 additions
template <class F> inline void xyzz_fixed_base_table(const XYZZ<F> &p, std::vector<XYZZ<F>> &tab) {
    tab.resize(32 * 255);
    XYZZ<F> base = p;
    for (int w = 0; w < 32; w++) {
        XYZZ<F> cur = base;
        for (int d = 1; d <= 255; d++) { tab[(size_t)255 * w + d - 1] = cur; if (d < 255) xyzz_add(cur, base); }
        for (int i = 0; i < 8; i++) base = xyzz_dbl(base);
    }
}
template <class F> inline XYZZ<F> xyzz_mul_fixed(const std::vector<XYZZ<F>> &tab, const uint64_t *k) {
    XYZZ<F> acc = XYZZ<F>::inf();
    for (int w = 0; w < 32; w++) {
        const unsigned d = (unsigned)(k[w >> 3] >> ((w & 7) * 8)) & 255u;
        if (d) xyzz_add(acc, tab[(size_t)255 * w + d - 1]);
    }
    return acc;
}

using G1Affine = Affine<Fp>;
using G2Affine = Affine<Fp2>;
using G1XYZZ = XYZZ<Fp>;
using G2XYZZ = XYZZ<Fp2>;

}  // namespace vsp
