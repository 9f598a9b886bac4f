/* TEST INFRASTRUCTURE ONLY -- CPU restatement ("oracle") of the Groth16 prover hot path.
 *
 * What it restates: the algorithms the reference (NilFoundation/vote-saver-protocol) reaches below
 *   bin/cli/include/nil/vote_saver/common.hpp:1132-1135  (encrypt<> -> r1cs_gg_ppzksnark prover)
 *   bin/cli/include/nil/vote_saver/common.hpp:916-917    (zk::generate -> Groth16 generator)
 * i.e. crypto3-algebra multiexp (BDLO12), crypto3-math evaluation_domain (basic radix-2),
 * crypto3-zk r1cs_to_qap::witness_map and r1cs_gg_ppzksnark generator/prover.  Those live in
 * git submodules that are EMPTY in /root/reference (.gitmodules:5-12,47-48) with no recoverable
 * pinned version, so this file follows the published libff / libfqfft / libsnark algorithms they
 * descend from (multiexp.tcc, basic_radix2_domain{,_aux}.tcc, r1cs_to_qap.tcc, r1cs_gg_ppzksnark.tcc).
 *
 * PARITY UNPINNED: the reference's own tests hold no golden vector for this path
 * (bin/cli/test/cli.cpp has no value assertion).  This oracle is pinned only by (i) the pure-Python
 * big-int oracle oracle/bls12_381.py on every primitive, (ii) public BLS12-381 constants,
 * (iii) reference bin/cli/src/data.bin[0:192] for the proof wire format, (iv) the Groth16 pairing
 * equation checked in Python.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 * The product (vote_saver_protocol_amd / libvsp_hip.so) never links or calls it.
 *
 * Representation: 64-bit limbs, Montgomery form internally; every exported function takes and
 * returns canonical little-endian u64 limbs (Fr 4, Fp 6, G1 affine 12, G2 affine 24; infinity = 0).
 * Single thread, like the reference build (bin/cli/CMakeLists.txt:114-116: -O3, no OpenMP).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>

typedef unsigned __int128 u128;

/* ------------------------------------------------------------------ generic n-limb Montgomery */
static inline uint64_t adc(uint64_t a, uint64_t b, uint64_t *c) { u128 t = (u128)a + b + *c; *c = (uint64_t)(t >> 64); return (uint64_t)t; }
static inline uint64_t sbb(uint64_t a, uint64_t b, uint64_t *br) { u128 t = (u128)a - b - *br; *br = (uint64_t)(t >> 64) & 1; return (uint64_t)t; }

static inline int big_geq(const uint64_t *a, const uint64_t *b, int n) {
    for (int i = n - 1; i >= 0; i--) { if (a[i] > b[i]) return 1; if (a[i] < b[i]) return 0; }
    return 1;
}
static inline void big_sub(uint64_t *r, const uint64_t *a, const uint64_t *b, int n) {
    uint64_t br = 0; for (int i = 0; i < n; i++) r[i] = sbb(a[i], b[i], &br);
}
static inline void mod_add(uint64_t *r, const uint64_t *a, const uint64_t *b, const uint64_t *m, int n) {
    uint64_t c = 0, t[8];
    for (int i = 0; i < n; i++) t[i] = adc(a[i], b[i], &c);
    if (c || big_geq(t, m, n)) big_sub(t, t, m, n);
    memcpy(r, t, n * 8);
}
static inline void mod_sub(uint64_t *r, const uint64_t *a, const uint64_t *b, const uint64_t *m, int n) {
    uint64_t br = 0, t[8];
    for (int i = 0; i < n; i++) t[i] = sbb(a[i], b[i], &br);
    if (br) { uint64_t c = 0; for (int i = 0; i < n; i++) t[i] = adc(t[i], m[i], &c); }
    memcpy(r, t, n * 8);
}
/* CIOS Montgomery product r = a*b*2^(-64n) mod m */
static inline void mont_mul(uint64_t *r, const uint64_t *a, const uint64_t *b, const uint64_t *m, uint64_t inv, int n) {
    uint64_t t[10] = {0};
    for (int i = 0; i < n; i++) {
        uint64_t c = 0;
        for (int j = 0; j < n; j++) { u128 x = (u128)a[j] * b[i] + t[j] + c; t[j] = (uint64_t)x; c = (uint64_t)(x >> 64); }
        u128 x = (u128)t[n] + c; t[n] = (uint64_t)x; t[n + 1] = (uint64_t)(x >> 64);
        uint64_t q = t[0] * inv;
        x = (u128)q * m[0] + t[0]; c = (uint64_t)(x >> 64);
        for (int j = 1; j < n; j++) { x = (u128)q * m[j] + t[j] + c; t[j - 1] = (uint64_t)x; c = (uint64_t)(x >> 64); }
        x = (u128)t[n] + c; t[n - 1] = (uint64_t)x; t[n] = t[n + 1] + (uint64_t)(x >> 64);
    }
    if (t[n] || big_geq(t, m, n)) big_sub(t, t, m, n);
    memcpy(r, t, n * 8);
}

/* ------------------------------------------------------------------ Fp (381 bit) and Fr (255 bit) */
typedef struct { uint64_t l[6]; } fp_t;
typedef struct { uint64_t l[4]; } fr_t;

static const uint64_t FP_MOD[6] = {0xb9feffffffffaaabULL, 0x1eabfffeb153ffffULL, 0x6730d2a0f6b0f624ULL,
                                   0x64774b84f38512bfULL, 0x4b1ba7b6434bacd7ULL, 0x1a0111ea397fe69aULL};
static const uint64_t FR_MOD[4] = {0xffffffff00000001ULL, 0x53bda402fffe5bfeULL, 0x3339d80809a1d805ULL, 0x73eda753299d7d48ULL};
static uint64_t FP_INV, FR_INV;
static fp_t FP_R, FP_R2;     /* R mod p, R^2 mod p */
static fr_t FR_R, FR_R2;
static int g_init = 0;

static uint64_t neg_inv64(uint64_t m0) { uint64_t x = 1; for (int i = 0; i < 7; i++) x *= 2 - m0 * x; return (uint64_t)0 - x; }
static void pow2_mod(uint64_t *r, int bits, const uint64_t *m, int n) {
    uint64_t t[8] = {1};
    for (int i = 0; i < bits; i++) mod_add(t, t, t, m, n);
    memcpy(r, t, n * 8);
}

#define DEF_FIELD(P, T, N, MOD, INV, RR, R2)                                                                   \
    static inline void P##_add(T *r, const T *a, const T *b) { mod_add(r->l, a->l, b->l, MOD, N); }            \
    static inline void P##_sub(T *r, const T *a, const T *b) { mod_sub(r->l, a->l, b->l, MOD, N); }            \
    static inline void P##_mul(T *r, const T *a, const T *b) { mont_mul(r->l, a->l, b->l, MOD, INV, N); }      \
    static inline void P##_sqr(T *r, const T *a) { mont_mul(r->l, a->l, a->l, MOD, INV, N); }                  \
    static inline void P##_set_zero(T *r) { memset(r, 0, sizeof(T)); }                                         \
    static inline void P##_set_one(T *r) { *r = RR; }                                                          \
    static inline int P##_is_zero(const T *a) { uint64_t o = 0; for (int i = 0; i < N; i++) o |= a->l[i]; return o == 0; } \
    static inline int P##_eq(const T *a, const T *b) { return memcmp(a, b, sizeof(T)) == 0; }                  \
    static inline void P##_neg(T *r, const T *a) { T z; memset(&z, 0, sizeof z); mod_sub(r->l, z.l, a->l, MOD, N); } \
    static inline void P##_from_canon(T *r, const uint64_t *c) { T t; memcpy(t.l, c, N * 8); mont_mul(r->l, t.l, R2.l, MOD, INV, N); } \
    static inline void P##_to_canon(uint64_t *c, const T *a) { T one; memset(&one, 0, sizeof one); one.l[0] = 1; T t; mont_mul(t.l, a->l, one.l, MOD, INV, N); memcpy(c, t.l, N * 8); } \
    static void P##_pow(T *r, const T *a, const uint64_t *e, int nlimbs) {                                     \
        T acc = RR, base = *a;                                                                                 \
        for (int i = 0; i < nlimbs * 64; i++) { if ((e[i >> 6] >> (i & 63)) & 1) P##_mul(&acc, &acc, &base); P##_sqr(&base, &base); } \
        *r = acc; }                                                                                            \
    static void P##_inv(T *r, const T *a) { uint64_t e[N]; uint64_t two[N] = {2}; big_sub(e, MOD, two, N); P##_pow(r, a, e, N); }

DEF_FIELD(fp, fp_t, 6, FP_MOD, FP_INV, FP_R, FP_R2)
DEF_FIELD(fr, fr_t, 4, FR_MOD, FR_INV, FR_R, FR_R2)

/* ------------------------------------------------------------------ Fp2 = Fp[u]/(u^2+1) */
typedef struct { fp_t c0, c1; } fp2_t;
static inline void fp2_add(fp2_t *r, const fp2_t *a, const fp2_t *b) { fp_add(&r->c0, &a->c0, &b->c0); fp_add(&r->c1, &a->c1, &b->c1); }
static inline void fp2_sub(fp2_t *r, const fp2_t *a, const fp2_t *b) { fp_sub(&r->c0, &a->c0, &b->c0); fp_sub(&r->c1, &a->c1, &b->c1); }
static inline void fp2_neg(fp2_t *r, const fp2_t *a) { fp_neg(&r->c0, &a->c0); fp_neg(&r->c1, &a->c1); }
static inline void fp2_mul(fp2_t *r, const fp2_t *a, const fp2_t *b) {
    fp_t t0, t1, t2, t3;
    fp_mul(&t0, &a->c0, &b->c0); fp_mul(&t1, &a->c1, &b->c1);
    fp_mul(&t2, &a->c0, &b->c1); fp_mul(&t3, &a->c1, &b->c0);
    fp_sub(&r->c0, &t0, &t1); fp_add(&r->c1, &t2, &t3);
}
static inline void fp2_sqr(fp2_t *r, const fp2_t *a) { fp2_t t = *a; fp2_mul(r, &t, &t); }
static inline void fp2_set_zero(fp2_t *r) { memset(r, 0, sizeof *r); }
static inline void fp2_set_one(fp2_t *r) { fp_set_one(&r->c0); fp_set_zero(&r->c1); }
static inline int fp2_is_zero(const fp2_t *a) { return fp_is_zero(&a->c0) && fp_is_zero(&a->c1); }
static inline int fp2_eq(const fp2_t *a, const fp2_t *b) { return fp_eq(&a->c0, &b->c0) && fp_eq(&a->c1, &b->c1); }
static void fp2_inv(fp2_t *r, const fp2_t *a) {
    fp_t n, t; fp_sqr(&n, &a->c0); fp_sqr(&t, &a->c1); fp_add(&n, &n, &t); fp_inv(&n, &n);
    fp_mul(&r->c0, &a->c0, &n); fp_mul(&t, &a->c1, &n); fp_neg(&r->c1, &t);
}

/* ------------------------------------------------------------------ curves */
#define F(x) fp_##x
#define FT fp_t
#define C(x) g1_##x
#include "curve_tmpl.h"
#undef F
#undef FT
#undef C
#define F(x) fp2_##x
#define FT fp2_t
#define C(x) g2_##x
#include "curve_tmpl.h"
#undef F
#undef FT
#undef C

static const uint64_t G1X[6] = {0xfb3af00adb22c6bbULL, 0x6c55e83ff97a1aefULL, 0xa14e3a3f171bac58ULL, 0xc3688c4f9774b905ULL, 0x2695638c4fa9ac0fULL, 0x17f1d3a73197d794ULL};
static const uint64_t G1Y[6] = {0x0caa232946c5e7e1ULL, 0xd03cc744a2888ae4ULL, 0x00db18cb2c04b3edULL, 0xfcf5e095d5d00af6ULL, 0xa09e30ed741d8ae4ULL, 0x08b3f481e3aaa0f1ULL};
static const uint64_t G2X0[6] = {0xd48056c8c121bdb8ULL, 0x0bac0326a805bbefULL, 0xb4510b647ae3d177ULL, 0xc6e47ad4fa403b02ULL, 0x260805272dc51051ULL, 0x024aa2b2f08f0a91ULL};
static const uint64_t G2X1[6] = {0xe5ac7d055d042b7eULL, 0x334cf11213945d57ULL, 0xb5da61bbdc7f5049ULL, 0x596bd0d09920b61aULL, 0x7dacd3a088274f65ULL, 0x13e02b6052719f60ULL};
static const uint64_t G2Y0[6] = {0xe193548608b82801ULL, 0x923ac9cc3baca289ULL, 0x6d429a695160d12cULL, 0xadfd9baa8cbdd3a7ULL, 0x8cc9cdc6da2e351aULL, 0x0ce5d527727d6e11ULL};
static const uint64_t G2Y1[6] = {0xaaa9075ff05f79beULL, 0x3f370d275cec1da1ULL, 0x267492ab572e99abULL, 0xcb3e287e85a763afULL, 0x32acd2b02bc28b99ULL, 0x0606c4a02ea734ccULL};
static g1_aff_t G1_GEN;
static g2_aff_t G2_GEN;

void ref_init(void) {
    if (g_init) return;
    FP_INV = neg_inv64(FP_MOD[0]); FR_INV = neg_inv64(FR_MOD[0]);
    pow2_mod(FP_R.l, 384, FP_MOD, 6); pow2_mod(FP_R2.l, 768, FP_MOD, 6);
    pow2_mod(FR_R.l, 256, FR_MOD, 4); pow2_mod(FR_R2.l, 512, FR_MOD, 4);
    fp_from_canon(&G1_GEN.x, G1X); fp_from_canon(&G1_GEN.y, G1Y);
    fp_from_canon(&G2_GEN.x.c0, G2X0); fp_from_canon(&G2_GEN.x.c1, G2X1);
    fp_from_canon(&G2_GEN.y.c0, G2Y0); fp_from_canon(&G2_GEN.y.c1, G2Y1);
    g_init = 1;
}

/* ---- limb <-> struct helpers */
static void g1_aff_load(g1_aff_t *p, const uint64_t *l) { fp_from_canon(&p->x, l); fp_from_canon(&p->y, l + 6); }
static void g1_aff_store(uint64_t *l, const g1_aff_t *p) { fp_to_canon(l, &p->x); fp_to_canon(l + 6, &p->y); }
static void g2_aff_load(g2_aff_t *p, const uint64_t *l) {
    fp_from_canon(&p->x.c0, l); fp_from_canon(&p->x.c1, l + 6); fp_from_canon(&p->y.c0, l + 12); fp_from_canon(&p->y.c1, l + 18);
}
static void g2_aff_store(uint64_t *l, const g2_aff_t *p) {
    fp_to_canon(l, &p->x.c0); fp_to_canon(l + 6, &p->x.c1); fp_to_canon(l + 12, &p->y.c0); fp_to_canon(l + 18, &p->y.c1);
}
static g1_aff_t *g1_load_array(const uint64_t *l, size_t n) {
    g1_aff_t *a = (g1_aff_t *)malloc((n ? n : 1) * sizeof *a);
    for (size_t i = 0; i < n; i++) g1_aff_load(&a[i], l + 12 * i);
    return a;
}
static g2_aff_t *g2_load_array(const uint64_t *l, size_t n) {
    g2_aff_t *a = (g2_aff_t *)malloc((n ? n : 1) * sizeof *a);
    for (size_t i = 0; i < n; i++) g2_aff_load(&a[i], l + 24 * i);
    return a;
}

/* ------------------------------------------------------------------ exported primitives (canonical limbs) */
void ref_fp_mul(const uint64_t *a, const uint64_t *b, uint64_t *out) { ref_init(); fp_t x, y; fp_from_canon(&x, a); fp_from_canon(&y, b); fp_mul(&x, &x, &y); fp_to_canon(out, &x); }
void ref_fp_inv(const uint64_t *a, uint64_t *out) { ref_init(); fp_t x; fp_from_canon(&x, a); fp_inv(&x, &x); fp_to_canon(out, &x); }
void ref_fr_mul(const uint64_t *a, const uint64_t *b, uint64_t *out) { ref_init(); fr_t x, y; fr_from_canon(&x, a); fr_from_canon(&y, b); fr_mul(&x, &x, &y); fr_to_canon(out, &x); }
void ref_fr_inv(const uint64_t *a, uint64_t *out) { ref_init(); fr_t x; fr_from_canon(&x, a); fr_inv(&x, &x); fr_to_canon(out, &x); }

void ref_g1_add(const uint64_t *p, const uint64_t *q, uint64_t *out) {
    ref_init(); g1_aff_t a, b, r; g1_jac_t ja, jr; g1_aff_load(&a, p); g1_aff_load(&b, q);
    g1_jac_from_aff(&ja, &a); g1_jac_madd(&jr, &ja, &b); g1_jac_to_aff(&r, &jr); g1_aff_store(out, &r);
}
void ref_g1_mul(const uint64_t *p, const uint64_t *k, uint64_t *out) {
    ref_init(); g1_aff_t a, r; g1_jac_t ja, jr; g1_aff_load(&a, p);
    g1_jac_from_aff(&ja, &a); g1_jac_mul(&jr, &ja, k); g1_jac_to_aff(&r, &jr); g1_aff_store(out, &r);
}
void ref_g2_add(const uint64_t *p, const uint64_t *q, uint64_t *out) {
    ref_init(); g2_aff_t a, b, r; g2_jac_t ja, jr; g2_aff_load(&a, p); g2_aff_load(&b, q);
    g2_jac_from_aff(&ja, &a); g2_jac_madd(&jr, &ja, &b); g2_jac_to_aff(&r, &jr); g2_aff_store(out, &r);
}
void ref_g2_mul(const uint64_t *p, const uint64_t *k, uint64_t *out) {
    ref_init(); g2_aff_t a, r; g2_jac_t ja, jr; g2_aff_load(&a, p);
    g2_jac_from_aff(&ja, &a); g2_jac_mul(&jr, &ja, k); g2_jac_to_aff(&r, &jr); g2_aff_store(out, &r);
}

/* multiexp (a1), multiexp_with_mixed_addition (a2) */
void ref_msm_g1(const uint64_t *bases, const uint64_t *scalars, size_t n, uint64_t *out, int mixed) {
    ref_init(); g1_aff_t *b = g1_load_array(bases, n); g1_jac_t j; g1_aff_t r;
    if (mixed) g1_multiexp_mixed(&j, b, scalars, n); else g1_multiexp_bdlo12(&j, b, scalars, n);
    g1_jac_to_aff(&r, &j); g1_aff_store(out, &r); free(b);
}
void ref_msm_g2(const uint64_t *bases, const uint64_t *scalars, size_t n, uint64_t *out, int mixed) {
    ref_init(); g2_aff_t *b = g2_load_array(bases, n); g2_jac_t j; g2_aff_t r;
    if (mixed) g2_multiexp_mixed(&j, b, scalars, n); else g2_multiexp_bdlo12(&j, b, scalars, n);
    g2_jac_to_aff(&r, &j); g2_aff_store(out, &r); free(b);
}
/* out[i] = scalars[i] * generator (synthetic bases k_i*G of SURVEY.md 8(d), and the generator's batch_exp) */
void ref_g1_batch_mul_gen(const uint64_t *scalars, size_t n, uint64_t *out) {
    ref_init(); g1_aff_t *r = (g1_aff_t *)malloc((n ? n : 1) * sizeof *r);
    g1_batch_mul_fixed(r, &G1_GEN, scalars, n);
    for (size_t i = 0; i < n; i++) g1_aff_store(out + 12 * i, &r[i]);
    free(r);
}
void ref_g2_batch_mul_gen(const uint64_t *scalars, size_t n, uint64_t *out) {
    ref_init(); g2_aff_t *r = (g2_aff_t *)malloc((n ? n : 1) * sizeof *r);
    g2_batch_mul_fixed(r, &G2_GEN, scalars, n);
    for (size_t i = 0; i < n; i++) g2_aff_store(out + 24 * i, &r[i]);
    free(r);
}

/* ------------------------------------------------------------------ evaluation_domain (a6): basic radix-2 */
static fr_t fr_from_u64(uint64_t v) { uint64_t c[4] = {v, 0, 0, 0}; fr_t r; fr_from_canon(&r, c); return r; }
static fr_t fr_omega(unsigned log_m) {
    /* root_of_unity = 7^((r-1)/2^32); omega_m = root^(2^(32-log_m)) */
    uint64_t e[4]; uint64_t one[4] = {1, 0, 0, 0}; big_sub(e, FR_MOD, one, 4);
    /* e >>= 32 */
    for (int i = 0; i < 4; i++) e[i] = (e[i] >> 32) | (i < 3 ? e[i + 1] << 32 : 0);
    fr_t g = fr_from_u64(7), w; fr_pow(&w, &g, e, 4);
    for (unsigned i = log_m; i < 32; i++) fr_sqr(&w, &w);
    return w;
}
static void fr_bitrev_permute(fr_t *a, unsigned log_m) {
    size_t n = (size_t)1 << log_m;
    for (size_t i = 0; i < n; i++) {
        size_t j = 0; for (unsigned b = 0; b < log_m; b++) j |= ((i >> b) & 1) << (log_m - 1 - b);
        if (i < j) { fr_t t = a[i]; a[i] = a[j]; a[j] = t; }
    }
}
/* _basic_serial_radix2_FFT */
static void fr_fft_serial(fr_t *a, unsigned log_m, const fr_t *omega) {
    size_t n = (size_t)1 << log_m;
    fr_bitrev_permute(a, log_m);
    size_t m = 1;
    for (unsigned s = 1; s <= log_m; s++) {
        uint64_t e[1] = {n / (2 * m)};
        fr_t wm; fr_pow(&wm, omega, e, 1);
        for (size_t k = 0; k < n; k += 2 * m) {
            fr_t w = FR_R;
            for (size_t j = 0; j < m; j++) {
                fr_t t; fr_mul(&t, &w, &a[k + j + m]);
                fr_sub(&a[k + j + m], &a[k + j], &t);
                fr_add(&a[k + j], &a[k + j], &t);
                fr_mul(&w, &w, &wm);
            }
        }
        m *= 2;
    }
}
static void fr_mul_by_coset(fr_t *a, size_t n, const fr_t *g) {
    fr_t u = *g;
    for (size_t i = 1; i < n; i++) { fr_mul(&a[i], &a[i], &u); fr_mul(&u, &u, g); }
}
static void r2_fft(fr_t *a, unsigned log_m) { fr_t w = fr_omega(log_m); fr_fft_serial(a, log_m, &w); }
static void r2_ifft(fr_t *a, unsigned log_m) {
    fr_t w = fr_omega(log_m), wi; fr_inv(&wi, &w); fr_fft_serial(a, log_m, &wi);
    fr_t m = fr_from_u64((uint64_t)1 << log_m), mi; fr_inv(&mi, &m);
    for (size_t i = 0; i < ((size_t)1 << log_m); i++) fr_mul(&a[i], &a[i], &mi);
}
static void r2_coset_fft(fr_t *a, unsigned log_m, const fr_t *g) { fr_mul_by_coset(a, (size_t)1 << log_m, g); r2_fft(a, log_m); }
static void r2_icoset_fft(fr_t *a, unsigned log_m, const fr_t *g) { r2_ifft(a, log_m); fr_t gi; fr_inv(&gi, g); fr_mul_by_coset(a, (size_t)1 << log_m, &gi); }

void ref_ntt_fr(uint64_t *a, unsigned log_m, int inverse, const uint64_t *coset_g) {
    ref_init();
    size_t n = (size_t)1 << log_m;
    fr_t *v = (fr_t *)malloc(n * sizeof *v);
    for (size_t i = 0; i < n; i++) fr_from_canon(&v[i], a + 4 * i);
    fr_t g; if (coset_g) fr_from_canon(&g, coset_g);
    if (!inverse) { if (coset_g) r2_coset_fft(v, log_m, &g); else r2_fft(v, log_m); }
    else { if (coset_g) r2_icoset_fft(v, log_m, &g); else r2_ifft(v, log_m); }
    for (size_t i = 0; i < n; i++) fr_to_canon(a + 4 * i, &v[i]);
    free(v);
}

/* evaluate_all_lagrange_polynomials(t) for the radix-2 domain */
static void r2_lagrange(fr_t *u, unsigned log_m, const fr_t *t) {
    size_t m = (size_t)1 << log_m;
    fr_t tm = *t; for (unsigned i = 0; i < log_m; i++) fr_sqr(&tm, &tm);
    fr_t omega = fr_omega(log_m);
    if (fr_eq(&tm, &FR_R)) {
        fr_t w = FR_R;
        for (size_t i = 0; i < m; i++) { if (fr_eq(&w, t)) u[i] = FR_R; else fr_set_zero(&u[i]); fr_mul(&w, &w, &omega); }
        return;
    }
    fr_t Z; fr_sub(&Z, &tm, &FR_R);
    fr_t mm = fr_from_u64(m), mi; fr_inv(&mi, &mm);
    fr_t l; fr_mul(&l, &Z, &mi);
    fr_t r = FR_R;
    /* u[i] = l / (t - r); batch-invert the denominators */
    fr_t *den = (fr_t *)malloc(m * sizeof *den), *pre = (fr_t *)malloc(m * sizeof *pre);
    fr_t acc = FR_R;
    for (size_t i = 0; i < m; i++) { fr_sub(&den[i], t, &r); pre[i] = acc; fr_mul(&acc, &acc, &den[i]); fr_mul(&r, &r, &omega); }
    fr_t inv; fr_inv(&inv, &acc);
    for (size_t i = m; i-- > 0;) { fr_t di; fr_mul(&di, &inv, &pre[i]); fr_mul(&inv, &inv, &den[i]); den[i] = di; }
    for (size_t i = 0; i < m; i++) { fr_mul(&u[i], &l, &den[i]); fr_mul(&l, &l, &omega); }
    free(den); free(pre);
}

/* ---- the domain make_evaluation_domain(min_size) selects (libfqfft get_evaluation_domain.tcc order: basic_radix2(min_size),
 * extended_radix2, step_radix2(min_size), then the same three at big + rounded_small; for Fr (two-adicity 32) and sizes < 2^32
 * the extended domain (m = 2^33) never constructs and one of the others always does) ---- */
typedef struct { size_t m, big_m, small_m; unsigned log_big, log_small; int step; } dom_t;
static unsigned clog2(size_t n) { unsigned l = 0; while (((size_t)1 << l) < n) l++; return l; }
static int dom_try_basic(dom_t *d, size_t m) {
    if (m <= 1 || m != ((size_t)1 << clog2(m)) || clog2(m) > 32) return 0;
    d->m = d->big_m = m; d->small_m = 0; d->log_big = clog2(m); d->log_small = 0; d->step = 0; return 1;
}
static int dom_try_step(dom_t *d, size_t m) {
    if (m <= 1 || clog2(m) > 32) return 0;
    size_t big = (size_t)1 << (clog2(m) - 1), small = m - big;
    if (small != ((size_t)1 << clog2(small))) return 0;
    d->m = m; d->big_m = big; d->small_m = small; d->log_big = clog2(big); d->log_small = clog2(small); d->step = 1; return 1;
}
static int dom_make(dom_t *d, size_t min_size) {
    if (min_size <= 1) return 0;
    size_t big = (size_t)1 << (clog2(min_size) - 1), small = min_size - big, rounded = (size_t)1 << clog2(small);
    return dom_try_basic(d, min_size) || dom_try_step(d, min_size) || dom_try_basic(d, big + rounded) || dom_try_step(d, big + rounded);
}
static fr_t fr_pow_u64(const fr_t *b, uint64_t e) { uint64_t ee[1] = {e}; fr_t r; fr_pow(&r, b, ee, 1); return r; }

/* step_radix2_domain.tcc FFT / iFFT */
static void step_fft(const dom_t *D, fr_t *a) {
    size_t big = D->big_m, small = D->small_m;
    fr_t omega = fr_omega(D->log_big + 1);
    fr_t *c = (fr_t *)malloc(big * sizeof *c), *d = (fr_t *)malloc(big * sizeof *d), *e = (fr_t *)calloc(small, sizeof *e);
    fr_t w = FR_R;
    for (size_t i = 0; i < big; i++) {
        if (i < small) { fr_add(&c[i], &a[i], &a[i + big]); fr_t t; fr_sub(&t, &a[i], &a[i + big]); fr_mul(&d[i], &w, &t); }
        else { c[i] = a[i]; fr_mul(&d[i], &w, &a[i]); }
        fr_mul(&w, &w, &omega);
    }
    size_t compr = big / small;
    for (size_t i = 0; i < small; i++) for (size_t j = 0; j < compr; j++) fr_add(&e[i], &e[i], &d[i + j * small]);
    if (big > 1) r2_fft(c, D->log_big);
    if (small > 1) r2_fft(e, D->log_small);
    memcpy(a, c, big * sizeof *c); memcpy(a + big, e, small * sizeof *e);
    free(c); free(d); free(e);
}
static void step_ifft(const dom_t *D, fr_t *a) {
    size_t big = D->big_m, small = D->small_m;
    fr_t omega = fr_omega(D->log_big + 1);
    fr_t *U0 = (fr_t *)malloc(big * sizeof *U0), *U1 = (fr_t *)malloc(small * sizeof *U1), *tmp = (fr_t *)malloc(big * sizeof *tmp);
    memcpy(U0, a, big * sizeof *U0); memcpy(U1, a + big, small * sizeof *U1);
    if (big > 1) r2_ifft(U0, D->log_big);
    if (small > 1) r2_ifft(U1, D->log_small);
    fr_t w = FR_R;
    for (size_t i = 0; i < big; i++) { fr_mul(&tmp[i], &U0[i], &w); fr_mul(&w, &w, &omega); }
    for (size_t i = small; i < big; i++) a[i] = U0[i];
    size_t compr = big / small;
    for (size_t i = 0; i < small; i++) for (size_t j = 1; j < compr; j++) fr_sub(&U1[i], &U1[i], &tmp[i + j * small]);
    fr_t oi; fr_inv(&oi, &omega); w = FR_R;
    for (size_t i = 0; i < small; i++) { fr_mul(&U1[i], &U1[i], &w); fr_mul(&w, &w, &oi); }
    fr_t two = fr_from_u64(2), half; fr_inv(&half, &two);
    for (size_t i = 0; i < small; i++) {
        fr_t s, t; fr_add(&s, &U0[i], &U1[i]); fr_sub(&t, &U0[i], &U1[i]);
        fr_mul(&a[i], &s, &half); fr_mul(&a[big + i], &t, &half);
    }
    free(U0); free(U1); free(tmp);
}
static void dom_fft(const dom_t *D, fr_t *a) { if (D->step) step_fft(D, a); else r2_fft(a, D->log_big); }
static void dom_ifft(const dom_t *D, fr_t *a) { if (D->step) step_ifft(D, a); else r2_ifft(a, D->log_big); }
static void dom_coset_fft(const dom_t *D, fr_t *a, const fr_t *g) { fr_mul_by_coset(a, D->m, g); dom_fft(D, a); }
static void dom_icoset_fft(const dom_t *D, fr_t *a, const fr_t *g) { dom_ifft(D, a); fr_t gi; fr_inv(&gi, g); fr_mul_by_coset(a, D->m, &gi); }
static fr_t dom_element(const dom_t *D, size_t idx) {
    if (!D->step) { fr_t w = fr_omega(D->log_big); return fr_pow_u64(&w, idx); }
    fr_t omega = fr_omega(D->log_big + 1);
    if (idx < D->big_m) { fr_t bw; fr_sqr(&bw, &omega); return fr_pow_u64(&bw, idx); }
    fr_t sw = fr_omega(D->log_small), r = fr_pow_u64(&sw, idx - D->big_m); fr_mul(&r, &r, &omega); return r;
}
static fr_t dom_vanishing(const dom_t *D, const fr_t *t) {
    fr_t a = fr_pow_u64(t, D->big_m); fr_sub(&a, &a, &FR_R);
    if (!D->step) return a;
    fr_t omega = fr_omega(D->log_big + 1), b = fr_pow_u64(t, D->small_m), ws = fr_pow_u64(&omega, D->small_m);
    fr_sub(&b, &b, &ws); fr_mul(&a, &a, &b); return a;
}
static void fr_batch_inv(fr_t *v, size_t n) {
    fr_t *pre = (fr_t *)malloc((n ? n : 1) * sizeof *pre); fr_t acc = FR_R;
    for (size_t i = 0; i < n; i++) { pre[i] = acc; fr_mul(&acc, &acc, &v[i]); }
    fr_t inv; fr_inv(&inv, &acc);
    for (size_t i = n; i-- > 0;) { fr_t di; fr_mul(&di, &inv, &pre[i]); fr_mul(&inv, &inv, &v[i]); v[i] = di; }
    free(pre);
}
/* evaluate_all_lagrange_polynomials (step_radix2_domain.tcc; t outside the domain in the step case) */
static void dom_lagrange(const dom_t *D, fr_t *u, const fr_t *t) {
    if (!D->step) { r2_lagrange(u, D->log_big, t); return; }
    size_t big = D->big_m, small = D->small_m;
    fr_t omega = fr_omega(D->log_big + 1), oi; fr_inv(&oi, &omega);
    fr_t ts; fr_mul(&ts, t, &oi);
    if (big > 1) r2_lagrange(u, D->log_big, t); else u[0] = FR_R;
    if (small > 1) r2_lagrange(u + big, D->log_small, &ts); else u[big] = FR_R;
    fr_t ws = fr_pow_u64(&omega, small), L0 = fr_pow_u64(t, small); fr_sub(&L0, &L0, &ws);
    fr_t bws; fr_sqr(&bws, &ws);                       /* big_omega^small_m */
    fr_t *den = (fr_t *)malloc(big * sizeof *den); fr_t elt = FR_R;
    for (size_t i = 0; i < big; i++) { fr_sub(&den[i], &elt, &ws); fr_mul(&elt, &elt, &bws); }
    fr_batch_inv(den, big);
    for (size_t i = 0; i < big; i++) { fr_mul(&u[i], &u[i], &L0); fr_mul(&u[i], &u[i], &den[i]); }
    free(den);
    fr_t L1 = fr_pow_u64(t, big), d1 = fr_pow_u64(&omega, big); fr_sub(&L1, &L1, &FR_R); fr_sub(&d1, &d1, &FR_R); fr_inv(&d1, &d1); fr_mul(&L1, &L1, &d1);
    for (size_t i = 0; i < small; i++) fr_mul(&u[big + i], &u[big + i], &L1);
}
/* divide_by_Z_on_coset */
static void dom_divide_by_z_on_coset(const dom_t *D, fr_t *P, const fr_t *g) {
    if (!D->step) { fr_t zi = dom_vanishing(D, g); fr_inv(&zi, &zi); for (size_t i = 0; i < D->m; i++) fr_mul(&P[i], &P[i], &zi); return; }
    size_t big = D->big_m, small = D->small_m;
    fr_t omega = fr_omega(D->log_big + 1);
    fr_t Z0 = fr_pow_u64(g, big); fr_sub(&Z0, &Z0, &FR_R);
    fr_t cZ0 = fr_pow_u64(g, small); fr_mul(&cZ0, &cZ0, &Z0);
    fr_t w1 = fr_pow_u64(&omega, small), w2; fr_sqr(&w2, &w1);
    fr_t w1Z0; fr_mul(&w1Z0, &w1, &Z0);
    fr_t *den = (fr_t *)malloc(big * sizeof *den); fr_t elt = FR_R;
    for (size_t i = 0; i < big; i++) { fr_mul(&den[i], &cZ0, &elt); fr_sub(&den[i], &den[i], &w1Z0); fr_mul(&elt, &elt, &w2); }
    fr_batch_inv(den, big);
    for (size_t i = 0; i < big; i++) fr_mul(&P[i], &P[i], &den[i]);
    free(den);
    fr_t go; fr_mul(&go, g, &omega);
    fr_t Z1 = dom_vanishing(D, &go); fr_inv(&Z1, &Z1);
    for (size_t i = 0; i < small; i++) fr_mul(&P[big + i], &P[big + i], &Z1);
}

/* exported domain operations (canonical limbs): op 0 fft, 1 inverse fft, 2 coset fft, 3 inverse coset fft (g = aux),
 * 4 lagrange at t = aux (a receives m values), 5 divide_by_z_on_coset with g = aux.  Returns m (0 = no such domain). */
size_t ref_domain_size(size_t min_size) { dom_t D; return dom_make(&D, min_size) ? D.m : 0; }
int ref_domain_is_step(size_t min_size) { dom_t D; return dom_make(&D, min_size) ? D.step : -1; }
size_t ref_domain_op(size_t min_size, int op, uint64_t *a, const uint64_t *aux) {
    ref_init();
    dom_t D; if (!dom_make(&D, min_size)) return 0;
    fr_t *v = (fr_t *)malloc(D.m * sizeof *v);
    for (size_t i = 0; i < D.m; i++) fr_from_canon(&v[i], a + 4 * i);
    fr_t x; if (aux) fr_from_canon(&x, aux);
    switch (op) {
        case 0: dom_fft(&D, v); break; case 1: dom_ifft(&D, v); break;
        case 2: dom_coset_fft(&D, v, &x); break; case 3: dom_icoset_fft(&D, v, &x); break;
        case 4: dom_lagrange(&D, v, &x); break; case 5: dom_divide_by_z_on_coset(&D, v, &x); break;
    }
    for (size_t i = 0; i < D.m; i++) fr_to_canon(a + 4 * i, &v[i]);
    free(v);
    return D.m;
}
void ref_domain_element(size_t min_size, size_t idx, uint64_t *out) { ref_init(); dom_t D; dom_make(&D, min_size); fr_t e = dom_element(&D, idx); fr_to_canon(out, &e); }
void ref_domain_vanishing(size_t min_size, const uint64_t *t, uint64_t *out) { ref_init(); dom_t D; dom_make(&D, min_size); fr_t x; fr_from_canon(&x, t); fr_t z = dom_vanishing(&D, &x); fr_to_canon(out, &z); }

/* ------------------------------------------------------------------ R1CS (CSR), synthetic instance of SURVEY 8(d) cfg 4 */
typedef struct {
    size_t num_constraints, num_inputs, num_vars;   /* num_vars excludes the constant; column 0 is the constant 1 */
    /* three CSR matrices, columns in [0, num_vars] */
    uint32_t *rp[3]; uint32_t *ci[3]; fr_t *co[3];
} r1cs_t;

static uint64_t sm64(uint64_t *s) { uint64_t z = (*s += 0x9E3779B97F4A7C15ULL); z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL; z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL; return z ^ (z >> 31); }
static void rand_fr_canon(uint64_t *s, uint64_t out[4]) {
    for (int i = 0; i < 4; i++) out[i] = sm64(s);
    out[3] &= 0x7fffffffffffffffULL;                      /* 255 bits, then reduce */
    while (big_geq(out, FR_MOD, 4)) big_sub(out, out, FR_MOD, 4);
}

/* Synthetic satisfiable R1CS: variable k (1-based) beyond the inputs is defined by one constraint:
 *   boolean wire (90%):  z_k * z_k = z_k,  z_k in {0,1}
 *   product wire (10%):  z_a * z_b = z_k,  a,b < k
 * 3 sparse terms per row.  Fills the canonical witness z[1..num_vars] (4 limbs each). */
void *ref_r1cs_synth_ballot(size_t num_constraints, size_t num_inputs, uint64_t seed, size_t msg_size, size_t vote, uint64_t *witness_out);
void *ref_r1cs_synth(size_t num_constraints, size_t num_inputs, uint64_t seed, uint64_t *witness_out /* num_vars*4 */) {
    return ref_r1cs_synth_ballot(num_constraints, num_inputs, seed, 0, 0, witness_out);
}
/* the same instance with a BALLOT in the first msg_size public inputs: one-hot at `vote` (the reference's m_field, common.hpp:1029-1040:
 * m[vote] = true); the remaining inputs stay random field elements (eid / sn / rt are packed field elements there too) */
void *ref_r1cs_synth_ballot(size_t num_constraints, size_t num_inputs, uint64_t seed, size_t msg_size, size_t vote, uint64_t *witness_out /* num_vars*4 */) {
    ref_init();
    r1cs_t *cs = (r1cs_t *)calloc(1, sizeof *cs);
    cs->num_constraints = num_constraints; cs->num_inputs = num_inputs; cs->num_vars = num_inputs + num_constraints;
    for (int m = 0; m < 3; m++) {
        cs->rp[m] = (uint32_t *)malloc((num_constraints + 1) * 4);
        cs->ci[m] = (uint32_t *)malloc((num_constraints ? num_constraints : 1) * 4);
        cs->co[m] = (fr_t *)malloc((num_constraints ? num_constraints : 1) * sizeof(fr_t));
    }
    uint64_t s = seed;
    for (size_t k = 1; k <= num_inputs; k++) rand_fr_canon(&s, witness_out + 4 * (k - 1));
    for (size_t k = 1; k <= msg_size && k <= num_inputs; k++) { memset(witness_out + 4 * (k - 1), 0, 32); witness_out[4 * (k - 1)] = (k - 1 == vote) ? 1 : 0; }
    fr_t *z = (fr_t *)malloc((cs->num_vars + 1) * sizeof *z);
    z[0] = FR_R;
    for (size_t k = 1; k <= num_inputs; k++) fr_from_canon(&z[k], witness_out + 4 * (k - 1));
    for (size_t j = 0; j < num_constraints; j++) {
        size_t k = num_inputs + 1 + j;
        uint64_t h = sm64(&s);
        size_t a, b;
        if (h % 10 != 0 || k < 3) {            /* boolean wire */
            uint64_t bit = (h >> 32) & 1;
            z[k] = bit ? FR_R : (fr_t){{0, 0, 0, 0}};
            a = b = k;
        } else {
            a = 1 + (size_t)((h >> 8) % (k - 1)); b = 1 + (size_t)((h >> 36) % (k - 1));
            fr_mul(&z[k], &z[a], &z[b]);
        }
        fr_to_canon(witness_out + 4 * (k - 1), &z[k]);
        for (int m = 0; m < 3; m++) { cs->rp[m][j] = (uint32_t)j; cs->co[m][j] = FR_R; }
        cs->ci[0][j] = (uint32_t)a; cs->ci[1][j] = (uint32_t)b; cs->ci[2][j] = (uint32_t)k;
    }
    for (int m = 0; m < 3; m++) cs->rp[m][num_constraints] = (uint32_t)num_constraints;
    free(z);
    return cs;
}
/* generic CSR constructor (coefficients canonical) */
void *ref_r1cs_from_csr(size_t num_constraints, size_t num_inputs, size_t num_vars,
                        const uint32_t *rpA, const uint32_t *ciA, const uint64_t *coA,
                        const uint32_t *rpB, const uint32_t *ciB, const uint64_t *coB,
                        const uint32_t *rpC, const uint32_t *ciC, const uint64_t *coC) {
    ref_init();
    r1cs_t *cs = (r1cs_t *)calloc(1, sizeof *cs);
    cs->num_constraints = num_constraints; cs->num_inputs = num_inputs; cs->num_vars = num_vars;
    const uint32_t *rp[3] = {rpA, rpB, rpC}, *ci[3] = {ciA, ciB, ciC}; const uint64_t *co[3] = {coA, coB, coC};
    for (int m = 0; m < 3; m++) {
        size_t nnz = rp[m][num_constraints];
        cs->rp[m] = (uint32_t *)malloc((num_constraints + 1) * 4); memcpy(cs->rp[m], rp[m], (num_constraints + 1) * 4);
        cs->ci[m] = (uint32_t *)malloc((nnz ? nnz : 1) * 4); memcpy(cs->ci[m], ci[m], nnz * 4);
        cs->co[m] = (fr_t *)malloc((nnz ? nnz : 1) * sizeof(fr_t));
        for (size_t i = 0; i < nnz; i++) fr_from_canon(&cs->co[m][i], co[m] + 4 * i);
    }
    return cs;
}
void ref_r1cs_free(void *p) { r1cs_t *cs = (r1cs_t *)p; for (int m = 0; m < 3; m++) { free(cs->rp[m]); free(cs->ci[m]); free(cs->co[m]); } free(cs); }
size_t ref_r1cs_num_vars(void *p) { return ((r1cs_t *)p)->num_vars; }
/* copy out the CSR so the device path receives the very same instance */
void ref_r1cs_export(void *p, int m, uint32_t *rp, uint32_t *ci, uint64_t *co) {
    r1cs_t *cs = (r1cs_t *)p; size_t nnz = cs->rp[m][cs->num_constraints];
    memcpy(rp, cs->rp[m], (cs->num_constraints + 1) * 4); memcpy(ci, cs->ci[m], nnz * 4);
    for (size_t i = 0; i < nnz; i++) fr_to_canon(co + 4 * i, &cs->co[m][i]);
}
size_t ref_r1cs_nnz(void *p, int m) { r1cs_t *cs = (r1cs_t *)p; return cs->rp[m][cs->num_constraints]; }

/* r1cs_to_qap: domain = make_evaluation_domain(num_constraints + num_inputs + 1) */
static dom_t cs_domain(const r1cs_t *cs) { dom_t D; memset(&D, 0, sizeof D); dom_make(&D, cs->num_constraints + cs->num_inputs + 1); return D; }
size_t ref_r1cs_domain_size(void *p) { return cs_domain((r1cs_t *)p).m; }
int ref_r1cs_domain_is_step(void *p) { return cs_domain((r1cs_t *)p).step; }

static void csr_matvec(fr_t *out, const r1cs_t *cs, int m, const fr_t *zfull /* [0]=1 */) {
    for (size_t i = 0; i < cs->num_constraints; i++) {
        fr_t acc; fr_set_zero(&acc);
        for (uint32_t e = cs->rp[m][i]; e < cs->rp[m][i + 1]; e++) { fr_t t; fr_mul(&t, &cs->co[m][e], &zfull[cs->ci[m][e]]); fr_add(&acc, &acc, &t); }
        out[i] = acc;
    }
}

/* r1cs_to_qap::witness_map with d1 = d2 = d3 = 0 (the values r1cs_gg_ppzksnark's prover passes):
 * returns the m coefficients of H (coefficients_for_H[0..m-1]; the (m+1)-th is 0). */
static void witness_map_h(fr_t *H, const r1cs_t *cs, const fr_t *zfull) {
    dom_t D = cs_domain(cs); size_t m = D.m;
    fr_t *aA = (fr_t *)calloc(m, sizeof(fr_t)), *aB = (fr_t *)calloc(m, sizeof(fr_t)), *aC = (fr_t *)calloc(m, sizeof(fr_t));
    csr_matvec(aA, cs, 0, zfull); csr_matvec(aB, cs, 1, zfull); csr_matvec(aC, cs, 2, zfull);
    /* the additional constraints input_i * 0 = 0 */
    for (size_t i = 0; i <= cs->num_inputs; i++) aA[cs->num_constraints + i] = zfull[i];
    fr_t g = fr_from_u64(7);
    dom_ifft(&D, aA); dom_ifft(&D, aB);
    dom_coset_fft(&D, aA, &g); dom_coset_fft(&D, aB, &g);
    for (size_t i = 0; i < m; i++) fr_mul(&H[i], &aA[i], &aB[i]);
    dom_ifft(&D, aC); dom_coset_fft(&D, aC, &g);
    for (size_t i = 0; i < m; i++) fr_sub(&H[i], &H[i], &aC[i]);
    dom_divide_by_z_on_coset(&D, H, &g);
    dom_icoset_fft(&D, H, &g);
    free(aA); free(aB); free(aC);
}
/* exported: Az/Bz/Cz evaluation vectors (each m x 4, zero padded, with the input rows in A) and H */
void ref_witness_map(void *p, const uint64_t *witness /* num_vars*4 */, uint64_t *H_out /* m*4 */,
                     uint64_t *Az, uint64_t *Bz, uint64_t *Cz /* each m*4, may be NULL */) {
    r1cs_t *cs = (r1cs_t *)p; size_t m = cs_domain(cs).m;
    fr_t *z = (fr_t *)malloc((cs->num_vars + 1) * sizeof *z); z[0] = FR_R;
    for (size_t k = 1; k <= cs->num_vars; k++) fr_from_canon(&z[k], witness + 4 * (k - 1));
    if (Az && Bz && Cz) {
        fr_t *t = (fr_t *)calloc(m, sizeof(fr_t));
        uint64_t *outs[3] = {Az, Bz, Cz};
        for (int mm = 0; mm < 3; mm++) {
            memset(t, 0, m * sizeof(fr_t)); csr_matvec(t, cs, mm, z);
            if (mm == 0) for (size_t i = 0; i <= cs->num_inputs; i++) t[cs->num_constraints + i] = z[i];
            for (size_t i = 0; i < m; i++) fr_to_canon(outs[mm] + 4 * i, &t[i]);
        }
        free(t);
    }
    fr_t *H = (fr_t *)malloc(m * sizeof *H);
    witness_map_h(H, cs, z);
    for (size_t i = 0; i < m; i++) fr_to_canon(H_out + 4 * i, &H[i]);
    free(H); free(z);
}
int ref_r1cs_is_satisfied(void *p, const uint64_t *witness) {
    r1cs_t *cs = (r1cs_t *)p;
    fr_t *z = (fr_t *)malloc((cs->num_vars + 1) * sizeof *z); z[0] = FR_R;
    for (size_t k = 1; k <= cs->num_vars; k++) fr_from_canon(&z[k], witness + 4 * (k - 1));
    fr_t *a = (fr_t *)malloc(cs->num_constraints * sizeof *a), *b = (fr_t *)malloc(cs->num_constraints * sizeof *a), *c = (fr_t *)malloc(cs->num_constraints * sizeof *a);
    csr_matvec(a, cs, 0, z); csr_matvec(b, cs, 1, z); csr_matvec(c, cs, 2, z);
    int ok = 1;
    for (size_t i = 0; i < cs->num_constraints && ok; i++) { fr_t t; fr_mul(&t, &a[i], &b[i]); if (!fr_eq(&t, &c[i])) ok = 0; }
    free(a); free(b); free(c); free(z);
    return ok;
}

/* ------------------------------------------------------------------ Groth16 keys (a9), generator, prover (a8) */
typedef struct {
    size_t num_vars, num_inputs, m;
    g1_aff_t alpha_g1, beta_g1, delta_g1, gamma_g1; g2_aff_t beta_g2, delta_g2, gamma_g2;
    g1_aff_t *A_query;             /* num_vars+1 */
    g2_aff_t *B_query_g2; g1_aff_t *B_query_g1;   /* num_vars+1 each */
    g1_aff_t *H_query;             /* m-1 */
    g1_aff_t *L_query;             /* num_vars-num_inputs */
    g1_aff_t *gamma_ABC_g1;        /* num_inputs+1 (verification key) */
} keypair_t;

/* r1cs_gg_ppzksnark generator with explicit toxic waste (t, alpha, beta, gamma, delta canonical Fr) */
void *ref_groth16_generate(void *pcs, const uint64_t *toxic /* 5*4 */) {
    r1cs_t *cs = (r1cs_t *)pcs;
    keypair_t *kp = (keypair_t *)calloc(1, sizeof *kp);
    dom_t D = cs_domain(cs); size_t m = D.m; size_t nv = cs->num_vars, ni = cs->num_inputs;
    kp->num_vars = nv; kp->num_inputs = ni; kp->m = m;
    fr_t t, alpha, beta, gamma, delta;
    fr_from_canon(&t, toxic); fr_from_canon(&alpha, toxic + 4); fr_from_canon(&beta, toxic + 8);
    fr_from_canon(&gamma, toxic + 12); fr_from_canon(&delta, toxic + 16);
    fr_t *u = (fr_t *)malloc(m * sizeof *u); dom_lagrange(&D, u, &t);
    fr_t *At = (fr_t *)calloc(nv + 1, sizeof(fr_t)), *Bt = (fr_t *)calloc(nv + 1, sizeof(fr_t)), *Ct = (fr_t *)calloc(nv + 1, sizeof(fr_t));
    for (size_t i = 0; i <= ni; i++) At[i] = u[cs->num_constraints + i];
    fr_t *Xt[3] = {At, Bt, Ct};
    for (int mm = 0; mm < 3; mm++)
        for (size_t i = 0; i < cs->num_constraints; i++)
            for (uint32_t e = cs->rp[mm][i]; e < cs->rp[mm][i + 1]; e++) {
                fr_t x; fr_mul(&x, &u[i], &cs->co[mm][e]); fr_add(&Xt[mm][cs->ci[mm][e]], &Xt[mm][cs->ci[mm][e]], &x);
            }
    fr_t Zt = dom_vanishing(&D, &t);
    fr_t gi, di; fr_inv(&gi, &gamma); fr_inv(&di, &delta);
    /* scalars -> canonical arrays for the fixed-base batch */
    uint64_t *sc = (uint64_t *)malloc((nv + 1 + m) * 32);
    g1_aff_t *tmp1;
    /* A_query */
    for (size_t i = 0; i <= nv; i++) fr_to_canon(sc + 4 * i, &At[i]);
    kp->A_query = (g1_aff_t *)malloc((nv + 1) * sizeof(g1_aff_t)); g1_batch_mul_fixed(kp->A_query, &G1_GEN, sc, nv + 1);
    /* B_query */
    for (size_t i = 0; i <= nv; i++) fr_to_canon(sc + 4 * i, &Bt[i]);
    kp->B_query_g1 = (g1_aff_t *)malloc((nv + 1) * sizeof(g1_aff_t)); g1_batch_mul_fixed(kp->B_query_g1, &G1_GEN, sc, nv + 1);
    kp->B_query_g2 = (g2_aff_t *)malloc((nv + 1) * sizeof(g2_aff_t)); g2_batch_mul_fixed(kp->B_query_g2, &G2_GEN, sc, nv + 1);
    /* H_query: t^i * Z(t) / delta, i < m-1 */
    { fr_t ti = FR_R, zd; fr_mul(&zd, &Zt, &di);
      for (size_t i = 0; i + 1 < m; i++) { fr_t x; fr_mul(&x, &ti, &zd); fr_to_canon(sc + 4 * i, &x); fr_mul(&ti, &ti, &t); } }
    kp->H_query = (g1_aff_t *)malloc((m ? m : 1) * sizeof(g1_aff_t)); g1_batch_mul_fixed(kp->H_query, &G1_GEN, sc, m - 1);
    /* L_query: (beta*A_i + alpha*B_i + C_i)/delta for auxiliary i; gamma_ABC for inputs */
    for (size_t i = 0; i <= nv; i++) {
        fr_t x, y; fr_mul(&x, &beta, &At[i]); fr_mul(&y, &alpha, &Bt[i]); fr_add(&x, &x, &y); fr_add(&x, &x, &Ct[i]);
        fr_mul(&x, &x, i <= ni ? &gi : &di); fr_to_canon(sc + 4 * i, &x);
    }
    tmp1 = (g1_aff_t *)malloc((nv + 1) * sizeof(g1_aff_t)); g1_batch_mul_fixed(tmp1, &G1_GEN, sc, nv + 1);
    kp->gamma_ABC_g1 = (g1_aff_t *)malloc((ni + 1) * sizeof(g1_aff_t)); memcpy(kp->gamma_ABC_g1, tmp1, (ni + 1) * sizeof(g1_aff_t));
    kp->L_query = (g1_aff_t *)malloc((nv - ni ? nv - ni : 1) * sizeof(g1_aff_t)); memcpy(kp->L_query, tmp1 + ni + 1, (nv - ni) * sizeof(g1_aff_t));
    free(tmp1);
    /* single elements */
    uint64_t c4[4];
    fr_to_canon(c4, &alpha); g1_batch_mul_fixed(&kp->alpha_g1, &G1_GEN, c4, 1);
    fr_to_canon(c4, &beta); g1_batch_mul_fixed(&kp->beta_g1, &G1_GEN, c4, 1); g2_batch_mul_fixed(&kp->beta_g2, &G2_GEN, c4, 1);
    fr_to_canon(c4, &delta); g1_batch_mul_fixed(&kp->delta_g1, &G1_GEN, c4, 1); g2_batch_mul_fixed(&kp->delta_g2, &G2_GEN, c4, 1);
    fr_to_canon(c4, &gamma); g2_batch_mul_fixed(&kp->gamma_g2, &G2_GEN, c4, 1); g1_batch_mul_fixed(&kp->gamma_g1, &G1_GEN, c4, 1);
    free(u); free(At); free(Bt); free(Ct); free(sc);
    return kp;
}

/* The generator's exponents, canonical Fr, for building a key with an external batch exponentiation
 * (the GPU fixed-base kernel): A_sc, B_sc [num_vars+1]; H_sc [m-1]; L_sc [num_vars-num_inputs]; ABC_sc [num_inputs+1]. */
void ref_groth16_key_scalars(void *pcs, const uint64_t *toxic, uint64_t *A_sc, uint64_t *B_sc, uint64_t *H_sc, uint64_t *L_sc, uint64_t *ABC_sc) {
    r1cs_t *cs = (r1cs_t *)pcs;
    dom_t D = cs_domain(cs); size_t m = D.m; size_t nv = cs->num_vars, ni = cs->num_inputs;
    fr_t t, alpha, beta, gamma, delta;
    fr_from_canon(&t, toxic); fr_from_canon(&alpha, toxic + 4); fr_from_canon(&beta, toxic + 8);
    fr_from_canon(&gamma, toxic + 12); fr_from_canon(&delta, toxic + 16);
    fr_t *u = (fr_t *)malloc(m * sizeof *u); dom_lagrange(&D, u, &t);
    fr_t *At = (fr_t *)calloc(nv + 1, sizeof(fr_t)), *Bt = (fr_t *)calloc(nv + 1, sizeof(fr_t)), *Ct = (fr_t *)calloc(nv + 1, sizeof(fr_t));
    for (size_t i = 0; i <= ni; i++) At[i] = u[cs->num_constraints + i];
    fr_t *Xt[3] = {At, Bt, Ct};
    for (int mm = 0; mm < 3; mm++)
        for (size_t i = 0; i < cs->num_constraints; i++)
            for (uint32_t e = cs->rp[mm][i]; e < cs->rp[mm][i + 1]; e++) {
                fr_t x; fr_mul(&x, &u[i], &cs->co[mm][e]); fr_add(&Xt[mm][cs->ci[mm][e]], &Xt[mm][cs->ci[mm][e]], &x);
            }
    fr_t Zt = dom_vanishing(&D, &t);
    fr_t gi, di; fr_inv(&gi, &gamma); fr_inv(&di, &delta);
    for (size_t i = 0; i <= nv; i++) { fr_to_canon(A_sc + 4 * i, &At[i]); fr_to_canon(B_sc + 4 * i, &Bt[i]); }
    { fr_t ti = FR_R, zd; fr_mul(&zd, &Zt, &di);
      for (size_t i = 0; i + 1 < m; i++) { fr_t x; fr_mul(&x, &ti, &zd); fr_to_canon(H_sc + 4 * i, &x); fr_mul(&ti, &ti, &t); } }
    for (size_t i = 0; i <= nv; i++) {
        fr_t x, y; fr_mul(&x, &beta, &At[i]); fr_mul(&y, &alpha, &Bt[i]); fr_add(&x, &x, &y); fr_add(&x, &x, &Ct[i]);
        if (i <= ni) { fr_mul(&x, &x, &gi); fr_to_canon(ABC_sc + 4 * i, &x); }
        else { fr_mul(&x, &x, &di); fr_to_canon(L_sc + 4 * (i - ni - 1), &x); }
    }
    free(u); free(At); free(Bt); free(Ct);
}
void ref_keypair_free(void *p) { keypair_t *kp = (keypair_t *)p; free(kp->A_query); free(kp->B_query_g1); free(kp->B_query_g2); free(kp->H_query); free(kp->L_query); free(kp->gamma_ABC_g1); free(kp); }

/* export a proving-key / verification-key component as canonical limbs.
 * which: 0 A_query(G1), 1 B_query_g1, 2 B_query_g2(G2), 3 H_query, 4 L_query, 5 gamma_ABC_g1,
 *        6 alpha_g1, 7 beta_g1, 8 delta_g1, 9 beta_g2(G2), 10 delta_g2(G2), 11 gamma_g2(G2), 12 gamma_g1 (extended verification key) */
size_t ref_keypair_count(void *p, int which) {
    keypair_t *kp = (keypair_t *)p;
    switch (which) { case 0: case 1: case 2: return kp->num_vars + 1; case 3: return kp->m - 1; case 4: return kp->num_vars - kp->num_inputs;
                     case 5: return kp->num_inputs + 1; default: return 1; }
}
void ref_keypair_export(void *p, int which, uint64_t *out) {
    keypair_t *kp = (keypair_t *)p; size_t n = ref_keypair_count(p, which);
    const g1_aff_t *g1 = NULL; const g2_aff_t *g2 = NULL;
    switch (which) {
        case 0: g1 = kp->A_query; break; case 1: g1 = kp->B_query_g1; break; case 2: g2 = kp->B_query_g2; break;
        case 3: g1 = kp->H_query; break; case 4: g1 = kp->L_query; break; case 5: g1 = kp->gamma_ABC_g1; break;
        case 6: g1 = &kp->alpha_g1; break; case 7: g1 = &kp->beta_g1; break; case 8: g1 = &kp->delta_g1; break;
        case 9: g2 = &kp->beta_g2; break; case 10: g2 = &kp->delta_g2; break; case 11: g2 = &kp->gamma_g2; break;
        case 12: g1 = &kp->gamma_g1; break;
    }
    if (g1) for (size_t i = 0; i < n; i++) g1_aff_store(out + 12 * i, &g1[i]);
    if (g2) for (size_t i = 0; i < n; i++) g2_aff_store(out + 24 * i, &g2[i]);
}

/* r1cs_gg_ppzksnark prover (a8) with explicit r, s:  proof = (A in G1, B in G2, C in G1), affine canonical.
 * If P1 != NULL the SAVER term r_enc*P1 is added to C (encrypted-input mode, common.hpp:1132-1135). */
void ref_groth16_prove(void *pcs, void *pkp, const uint64_t *witness, const uint64_t *r4, const uint64_t *s4,
                       const uint64_t *P1 /* 12 or NULL */, const uint64_t *r_enc /* 4 or NULL */,
                       uint64_t *A_out /*12*/, uint64_t *B_out /*24*/, uint64_t *C_out /*12*/) {
    r1cs_t *cs = (r1cs_t *)pcs; keypair_t *kp = (keypair_t *)pkp;
    size_t nv = cs->num_vars, ni = cs->num_inputs, m = kp->m;
    fr_t *z = (fr_t *)malloc((nv + 1) * sizeof *z); z[0] = FR_R;
    for (size_t k = 1; k <= nv; k++) fr_from_canon(&z[k], witness + 4 * (k - 1));
    fr_t *H = (fr_t *)malloc(m * sizeof *H);
    witness_map_h(H, cs, z);
    uint64_t *zc = (uint64_t *)malloc((nv + 1) * 32), *hc = (uint64_t *)malloc(m * 32);
    zc[0] = 1; zc[1] = zc[2] = zc[3] = 0; memcpy(zc + 4, witness, nv * 32);
    for (size_t i = 0; i < m; i++) fr_to_canon(hc + 4 * i, &H[i]);
    g1_jac_t eA, eB1, eH, eL; g2_jac_t eB2;
    g1_multiexp_bdlo12(&eA, kp->A_query, zc, nv + 1);
    g1_multiexp_mixed(&eB1, kp->B_query_g1, zc, nv + 1);
    g2_multiexp_mixed(&eB2, kp->B_query_g2, zc, nv + 1);
    g1_multiexp_bdlo12(&eH, kp->H_query, hc, m - 1);
    g1_multiexp_mixed(&eL, kp->L_query, zc + 4 * (ni + 1), nv - ni);
    g1_jac_t t, dj, gA, gB1, gC; g2_jac_t t2, dj2, gB2;
    /* A = alpha + eA + r*delta */
    g1_jac_from_aff(&dj, &kp->delta_g1); g1_jac_mul(&t, &dj, r4);
    g1_jac_madd(&gA, &eA, &kp->alpha_g1); g1_jac_add(&gA, &gA, &t);
    /* B (G1) = beta + eB1 + s*delta ; B (G2) likewise */
    g1_jac_mul(&t, &dj, s4); g1_jac_madd(&gB1, &eB1, &kp->beta_g1); g1_jac_add(&gB1, &gB1, &t);
    g2_jac_from_aff(&dj2, &kp->delta_g2); g2_jac_mul(&t2, &dj2, s4);
    g2_jac_madd(&gB2, &eB2, &kp->beta_g2); g2_jac_add(&gB2, &gB2, &t2);
    /* C = eH + eL + s*A + r*B1 - (r*s)*delta */
    fr_t rr, ss, rs; fr_from_canon(&rr, r4); fr_from_canon(&ss, s4); fr_mul(&rs, &rr, &ss);
    uint64_t rs4[4]; fr_to_canon(rs4, &rs);
    g1_jac_add(&gC, &eH, &eL);
    g1_jac_mul(&t, &gA, s4); g1_jac_add(&gC, &gC, &t);
    g1_jac_mul(&t, &gB1, r4); g1_jac_add(&gC, &gC, &t);
    g1_jac_mul(&t, &dj, rs4); g1_jac_neg(&t, &t); g1_jac_add(&gC, &gC, &t);
    if (P1 && r_enc) { g1_aff_t p1; g1_jac_t pj; g1_aff_load(&p1, P1); g1_jac_from_aff(&pj, &p1); g1_jac_mul(&t, &pj, r_enc); g1_jac_add(&gC, &gC, &t); }
    g1_aff_t a; g2_aff_t b;
    g1_jac_to_aff(&a, &gA); g1_aff_store(A_out, &a);
    g2_jac_to_aff(&b, &gB2); g2_aff_store(B_out, &b);
    g1_jac_to_aff(&a, &gC); g1_aff_store(C_out, &a);
    free(z); free(H); free(zc); free(hc);
}


/* ------------------------------------------------------------------ SAVER: elgamal_verifiable around the prover
 * Restates crypto3-pubkey elgamal_verifiable.hpp (absent submodule, /root/reference/.gitmodules:50) = the SAVER scheme of Lee, Choi,
 * Kim, Oh ("SAVER: SNARK-friendly, Additively-homomorphic, and Verifiable Encryption and decryption with Rerandomization", fig. 3),
 * as the reference calls it: generate_keypair(rnd[3n+2], {gg_keypair, msg_size}) (common.hpp:921-931), encrypt(m, {r, pk, gg_keypair,
 * primary, auxiliary}) (:1131-1135), rerandomize(rnd[3], ct, {pk, gg_keypair, proof}) (:1138-1145).  [UPSTREAM-KNOWLEDGE: member
 * names and the order in which rnd is consumed follow upstream as remembered; the scheme itself is pinned by the paper and by the
 * verification equations oracle/saver.py checks with the pairing.]   n = msg_size, G_i = gamma_ABC_g1[i] (i = 1..n: message inputs).
 *   rnd   : s_1..s_n | v_1..v_n | t_0..t_n | rho
 *   pk    : delta_g1 | delta_s_g1[i] = s_i delta_g1 | t_g1[i] = t_i G_i | t_g2[j] = t_j H (j = 0..n) |
 *           delta_sum_s_g1 = (t_0 + sum t_j s_j) delta_g1 | gamma_inverse_sum_s_g1 = -(1 + sum s_j) gamma_g1
 *   sk    : rho          vk : rho_g2 = rho H | rho_sv_g2[i] = s_i v_i H | rho_rhov_g2[i] = rho v_i H
 *   ct    : c_0 = r delta_g1 | c_i = r delta_s_g1[i] + m_i G_i | psi = r delta_sum_s_g1 + sum m_i t_g1[i]          (n + 2 points)
 *   proof : Groth16 proof with C += r gamma_inverse_sum_s_g1 (ref_groth16_prove's P1 / r_enc arguments)
 * Flat layouts (uint64 canonical limbs): pk = 12 + 12n + 12n + 24(n+1) + 12 + 12 words, vk = 24 + 24n + 24n words. */
static void g1_mul_canon(g1_jac_t *r, const uint64_t *pt12, const uint64_t k[4]) { g1_aff_t a; g1_jac_t j; g1_aff_load(&a, pt12); g1_jac_from_aff(&j, &a); g1_jac_mul(r, &j, k); }
static void g1_store_jac(uint64_t *out12, const g1_jac_t *j) { g1_aff_t a; g1_jac_to_aff(&a, j); g1_aff_store(out12, &a); }
static void g2_store_jac(uint64_t *out24, const g2_jac_t *j) { g2_aff_t a; g2_jac_to_aff(&a, j); g2_aff_store(out24, &a); }
static void g1_add_canon(g1_jac_t *r, const g1_jac_t *p, const uint64_t *pt12) {       /* r = p + affine point (all-zero = infinity) */
    g1_aff_t a; g1_aff_load(&a, pt12); g1_jac_t j;
    if (g1_aff_is_inf(&a)) { *r = *p; return; }
    g1_jac_from_aff(&j, &a); g1_jac_add(r, p, &j);
}
size_t ref_saver_pk_words(size_t n) { return 12 + 12 * n + 12 * n + 24 * (n + 1) + 12 + 12; }
size_t ref_saver_vk_words(size_t n) { return 24 + 24 * n + 24 * n; }

void ref_saver_keygen(size_t n, const uint64_t *delta_g1, const uint64_t *gamma_g1, const uint64_t *gamma_abc /* (n+1) x 12 */,
                      const uint64_t *rnd /* (3n+2) x 4 */, uint64_t *pk, uint64_t *sk /* 4 */, uint64_t *vk) {
    ref_init();
    const uint64_t *s = rnd, *v = rnd + 4 * n, *t = rnd + 8 * n, *rho = rnd + 4 * (3 * n + 1);
    uint64_t *p_delta_s = pk + 12, *p_t_g1 = p_delta_s + 12 * n, *p_t_g2 = p_t_g1 + 12 * n, *p_dsum = p_t_g2 + 24 * (n + 1), *p_ginv = p_dsum + 12;
    memcpy(pk, delta_g1, 96);
    g1_jac_t j; g2_jac_t j2, h; g2_jac_from_aff(&h, &G2_GEN);
    fr_t sum_s = FR_R, sum_ts, x, y;                      /* 1 + sum s_j  (FR_R = one) ;  t_0 + sum t_j s_j */
    fr_from_canon(&sum_ts, t);
    for (size_t i = 0; i < n; i++) {
        g1_mul_canon(&j, delta_g1, s + 4 * i); g1_store_jac(p_delta_s + 12 * i, &j);
        g1_mul_canon(&j, gamma_abc + 12 * (i + 1), t + 4 * (i + 1)); g1_store_jac(p_t_g1 + 12 * i, &j);
        fr_from_canon(&x, s + 4 * i); fr_add(&sum_s, &sum_s, &x);
        fr_from_canon(&y, t + 4 * (i + 1)); fr_mul(&y, &y, &x); fr_add(&sum_ts, &sum_ts, &y);
    }
    for (size_t jx = 0; jx <= n; jx++) { g2_jac_mul(&j2, &h, t + 4 * jx); g2_store_jac(p_t_g2 + 24 * jx, &j2); }
    uint64_t c4[4];
    fr_to_canon(c4, &sum_ts); g1_mul_canon(&j, delta_g1, c4); g1_store_jac(p_dsum, &j);
    fr_neg(&sum_s, &sum_s); fr_to_canon(c4, &sum_s); g1_mul_canon(&j, gamma_g1, c4); g1_store_jac(p_ginv, &j);
    memcpy(sk, rho, 32);
    fr_t r_; fr_from_canon(&r_, rho);
    g2_jac_mul(&j2, &h, rho); g2_store_jac(vk, &j2);
    for (size_t i = 0; i < n; i++) {
        fr_t si, vi; fr_from_canon(&si, s + 4 * i); fr_from_canon(&vi, v + 4 * i);
        fr_mul(&x, &si, &vi); fr_to_canon(c4, &x); g2_jac_mul(&j2, &h, c4); g2_store_jac(vk + 24 + 24 * i, &j2);
        fr_mul(&x, &r_, &vi); fr_to_canon(c4, &x); g2_jac_mul(&j2, &h, c4); g2_store_jac(vk + 24 + 24 * n + 24 * i, &j2);
    }
}

/* ciphertext part of encrypt (the proof part is ref_groth16_prove with P1 = gamma_inverse_sum_s_g1, r_enc = r) */
void ref_saver_encrypt_ct(size_t n, const uint64_t *pk, const uint64_t *gamma_abc, const uint64_t *msg /* n x 4 */, const uint64_t *r, uint64_t *ct /* (n+2) x 12 */) {
    ref_init();
    const uint64_t *p_delta_s = pk + 12, *p_t_g1 = p_delta_s + 12 * n, *p_dsum = p_t_g1 + 12 * n + 24 * (n + 1);
    g1_jac_t c, m, psi;
    g1_mul_canon(&c, pk, r); g1_store_jac(ct, &c);
    g1_mul_canon(&psi, p_dsum, r);
    for (size_t i = 0; i < n; i++) {
        g1_mul_canon(&c, p_delta_s + 12 * i, r);
        g1_mul_canon(&m, gamma_abc + 12 * (i + 1), msg + 4 * i); g1_jac_add(&c, &c, &m);
        g1_store_jac(ct + 12 * (i + 1), &c);
        g1_mul_canon(&m, p_t_g1 + 12 * i, msg + 4 * i); g1_jac_add(&psi, &psi, &m);
    }
    g1_store_jac(ct + 12 * (n + 1), &psi);
}

/* rerandomize: rnd3 = (r', z1, z2);  ct_i += r' X_i, psi += r' delta_sum_s_g1;
 * A' = z1 A,  B' = z1^-1 B + z2 delta_g2,  C' = C + (z1 z2) A + r' gamma_inverse_sum_s_g1        (all in place) */
void ref_saver_rerandomize(size_t n, const uint64_t *pk, const uint64_t *delta_g2, const uint64_t *rnd3, uint64_t *ct, uint64_t *A, uint64_t *B, uint64_t *Cc) {
    ref_init();
    const uint64_t *rp = rnd3, *z1 = rnd3 + 4, *z2 = rnd3 + 8;
    const uint64_t *p_delta_s = pk + 12, *p_dsum = p_delta_s + 12 * n + 12 * n + 24 * (n + 1), *p_ginv = p_dsum + 12;
    g1_jac_t t;
    for (size_t i = 0; i <= n + 1; i++) {
        const uint64_t *X = i == 0 ? pk : (i <= n ? p_delta_s + 12 * (i - 1) : p_dsum);
        g1_mul_canon(&t, X, rp); g1_add_canon(&t, &t, ct + 12 * i); g1_store_jac(ct + 12 * i, &t);
    }
    fr_t a, b, zi, zz; fr_from_canon(&a, z1); fr_from_canon(&b, z2); fr_inv(&zi, &a); fr_mul(&zz, &a, &b);
    uint64_t zi4[4], zz4[4]; fr_to_canon(zi4, &zi); fr_to_canon(zz4, &zz);
    g1_jac_t nA, nC, u; g2_jac_t nB, w; g2_aff_t b2, d2;
    g1_mul_canon(&u, A, zz4);                          /* (z1 z2) A, from the ORIGINAL A */
    g1_mul_canon(&nC, p_ginv, rp); g1_jac_add(&nC, &nC, &u); g1_add_canon(&nC, &nC, Cc);
    g1_mul_canon(&nA, A, z1);
    g2_aff_load(&b2, B); g2_jac_from_aff(&nB, &b2); g2_jac_mul(&nB, &nB, zi4);
    g2_aff_load(&d2, delta_g2); g2_jac_from_aff(&w, &d2); g2_jac_mul(&w, &w, z2); g2_jac_add(&nB, &nB, &w);
    g1_store_jac(A, &nA); g2_store_jac(B, &nB); g1_store_jac(Cc, &nC);
}
