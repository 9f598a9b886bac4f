// CPU test build of the device math headers (field.h / curve.h compiled by g++ with the 32-bit-limb
// types the kernels use).  Test infrastructure only: lets `-m "not gpu"` tests check the kernel
// arithmetic against the oracle without a GPU.  All arguments canonical little-endian u64 limbs.
#include <string.h>
#include "../../vote_saver_protocol_amd/csrc/curve.h"
using namespace vsp;

template <class T> static T load(const uint64_t *p) { T t; memcpy(&t, p, sizeof(T)); return t; }
template <class T> static void store(uint64_t *p, const T &t) { memcpy(p, &t, sizeof(T)); }

extern "C" {
void chk_fp_mul(const uint64_t *a, const uint64_t *b, uint64_t *out) { store(out, from_mont(mul(to_mont(load<Fp>(a)), to_mont(load<Fp>(b))))); }
void chk_fp_add(const uint64_t *a, const uint64_t *b, uint64_t *out) { store(out, add(load<Fp>(a), load<Fp>(b))); }
void chk_fp_sub(const uint64_t *a, const uint64_t *b, uint64_t *out) { store(out, sub(load<Fp>(a), load<Fp>(b))); }
void chk_fp_inv(const uint64_t *a, uint64_t *out) { store(out, from_mont(inv(to_mont(load<Fp>(a))))); }
void chk_hfp_mul(const uint64_t *a, const uint64_t *b, uint64_t *out) { store(out, from_mont(mul(to_mont(load<HFp>(a)), to_mont(load<HFp>(b))))); }
void chk_hfp_inv(const uint64_t *a, uint64_t *out) { store(out, from_mont(inv(to_mont(load<HFp>(a))))); }
void chk_fr_mul(const uint64_t *a, const uint64_t *b, uint64_t *out) { store(out, from_mont(mul(to_mont(load<Fr>(a)), to_mont(load<Fr>(b))))); }
void chk_fr_add(const uint64_t *a, const uint64_t *b, uint64_t *out) { store(out, add(load<Fr>(a), load<Fr>(b))); }
void chk_fr_sub(const uint64_t *a, const uint64_t *b, uint64_t *out) { store(out, sub(load<Fr>(a), load<Fr>(b))); }
void chk_fr_inv(const uint64_t *a, uint64_t *out) { store(out, from_mont(inv(to_mont(load<Fr>(a))))); }
void chk_hfr_inv(const uint64_t *a, uint64_t *out) { store(out, from_mont(inv(to_mont(load<HFr>(a))))); }
// the NTT butterfly trick: montmul(canonical x, Montgomery w) = canonical x*w
void chk_fr_mul_mixed(const uint64_t *x, const uint64_t *w, uint64_t *out) { store(out, mul(load<Fr>(x), to_mont(load<Fr>(w)))); }
void chk_fp2_mul(const uint64_t *a, const uint64_t *b, uint64_t *out) { store(out, from_mont(mul(to_mont(load<Fp2>(a)), to_mont(load<Fp2>(b))))); }
void chk_fp2_sqr(const uint64_t *a, uint64_t *out) { store(out, from_mont(sqr(to_mont(load<Fp2>(a))))); }
void chk_fp2_inv(const uint64_t *a, uint64_t *out) { store(out, from_mont(inv(to_mont(load<Fp2>(a))))); }

}  // extern C
// op: 0 = madd(p, q), 1 = add(p, q) both XYZZ (q rescaled by a non-trivial Z first), 2 = dbl(p), 3 = madd(p, -q)
template <class F> static void curve_op(int op, const uint64_t *p, const uint64_t *q, uint64_t *out) {
    Affine<F> a = load<Affine<F>>(p), b = load<Affine<F>>(q);
    a.x = to_mont(a.x); a.y = to_mont(a.y); b.x = to_mont(b.x); b.y = to_mont(b.y);
    XYZZ<F> acc = xyzz_from_affine(a);
    if (op == 0) xyzz_madd(acc, b);
    else if (op == 3) xyzz_madd(acc, b, true);
    else if (op == 2) acc = xyzz_dbl(acc);
    else {
        XYZZ<F> qb = xyzz_from_affine(b);
        if (!is_inf(qb)) {                      // give q a non-trivial Z: (X z^2, Y z^3, z^2, z^3), z = 3
            F z = add(add(F::one(), F::one()), F::one()), z2 = sqr(z), z3 = mul(z2, z);
            qb.X = mul(qb.X, z2); qb.Y = mul(qb.Y, z3); qb.ZZ = z2; qb.ZZZ = z3;
        }
        acc = xyzz_dbl(acc);                    // and p a non-trivial Z as well: acc = 2p ...
        xyzz_madd(acc, a, true);                // ... - p = p
        xyzz_add(acc, qb);
    }
    Jacobian<F> j = xyzz_to_jacobian(acc);
    XYZZ<F> back = jacobian_to_xyzz(j);
    Affine<F> r = xyzz_to_affine(back);
    r.x = from_mont(r.x); r.y = from_mont(r.y);
    store(out, r);
}
// k * p three ways (host types): mode 0 = bit by bit (xyzz_mul_scalar), 1 = 4-bit windows (xyzz_mul_scalar_w4), 2 = fixed-base table of 32 x 255
// multiples (xyzz_fixed_base_table / xyzz_mul_fixed: the prover's multiples of delta)
template <class F> static void scalar_mul(int mode, const uint64_t *p, const uint64_t *k, uint64_t *out) {
    Affine<F> a = load<Affine<F>>(p);
    a.x = to_mont(a.x); a.y = to_mont(a.y);
    XYZZ<F> base = xyzz_from_affine(a), r;
    if (mode == 0) r = xyzz_mul_scalar(base, k, 256);
    else if (mode == 1) r = xyzz_mul_scalar_w4(base, k);
    else { std::vector<XYZZ<F>> tab; xyzz_fixed_base_table(base, tab); r = xyzz_mul_fixed(tab, k); }
    Affine<F> o = xyzz_to_affine(r);
    o.x = from_mont(o.x); o.y = from_mont(o.y);
    store(out, o);
}
extern "C" {
void chk_hg1_mul(int mode, const uint64_t *p, const uint64_t *k, uint64_t *out) { scalar_mul<HFp>(mode, p, k, out); }
void chk_hg2_mul(int mode, const uint64_t *p, const uint64_t *k, uint64_t *out) { scalar_mul<HFp2>(mode, p, k, out); }
void chk_g1_op(int op, const uint64_t *p, const uint64_t *q, uint64_t *out) { curve_op<Fp>(op, p, q, out); }
void chk_g2_op(int op, const uint64_t *p, const uint64_t *q, uint64_t *out) { curve_op<Fp2>(op, p, q, out); }
void chk_hg1_op(int op, const uint64_t *p, const uint64_t *q, uint64_t *out) { curve_op<HFp>(op, p, q, out); }
void chk_hg2_op(int op, const uint64_t *p, const uint64_t *q, uint64_t *out) { curve_op<HFp2>(op, p, q, out); }
}
