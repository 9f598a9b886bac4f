/* TEST INFRASTRUCTURE ONLY (see vsp_ref.c header).
 * Curve template: short Weierstrass, a4 = 0, Jacobian coordinates -- restates the group law of
 * crypto3-algebra curves/detail/forms/short_weierstrass/jacobian_with_a4_0 (absent submodule,
 * /root/reference/.gitmodules:8-9; coordinates named at bin/cli/include/nil/vote_saver/common.hpp:117-121).
 * Formulas are the libff-lineage ones: add-2007-bl, madd-2007-bl, dbl-2009-l.
 *
 * Instantiate with:  #define F(x) fp_##x / fp2_##x ,  #define FT fp_t / fp2_t,
 *                    #define C(x) g1_##x / g2_##x
 */

typedef struct { FT X, Y, Z; } C(jac_t);
typedef struct { FT x, y; } C(aff_t);        /* infinity: x = y = 0 (not on the curve since b != 0) */

static inline int C(aff_is_inf)(const C(aff_t) *p) { return F(is_zero)(&p->x) && F(is_zero)(&p->y); }
static inline int C(jac_is_inf)(const C(jac_t) *p) { return F(is_zero)(&p->Z); }
static inline void C(jac_set_inf)(C(jac_t) *p) { F(set_one)(&p->X); F(set_one)(&p->Y); F(set_zero)(&p->Z); }
static inline void C(jac_from_aff)(C(jac_t) *r, const C(aff_t) *p) {
    if (C(aff_is_inf)(p)) { C(jac_set_inf)(r); return; }
    r->X = p->x; r->Y = p->y; F(set_one)(&r->Z);
}

static void C(jac_dbl)(C(jac_t) *r, const C(jac_t) *p) {
    if (C(jac_is_inf)(p)) { *r = *p; return; }
    FT A, B, Cc, D, E, Fq, t, X3, Y3, Z3;
    F(sqr)(&A, &p->X);
    F(sqr)(&B, &p->Y);
    F(sqr)(&Cc, &B);
    F(add)(&t, &p->X, &B); F(sqr)(&t, &t); F(sub)(&t, &t, &A); F(sub)(&t, &t, &Cc);
    F(add)(&D, &t, &t);
    F(add)(&E, &A, &A); F(add)(&E, &E, &A);
    F(sqr)(&Fq, &E);
    F(sub)(&X3, &Fq, &D); F(sub)(&X3, &X3, &D);
    F(sub)(&t, &D, &X3); F(mul)(&Y3, &E, &t);
    F(add)(&t, &Cc, &Cc); F(add)(&t, &t, &t); F(add)(&t, &t, &t);
    F(sub)(&Y3, &Y3, &t);
    F(mul)(&Z3, &p->Y, &p->Z); F(add)(&Z3, &Z3, &Z3);
    r->X = X3; r->Y = Y3; r->Z = Z3;
}

static void C(jac_add)(C(jac_t) *r, const C(jac_t) *p, const C(jac_t) *q) {
    if (C(jac_is_inf)(p)) { *r = *q; return; }
    if (C(jac_is_inf)(q)) { *r = *p; return; }
    FT Z1Z1, Z2Z2, U1, U2, S1, S2, H, I, J, rr, V, t, X3, Y3, Z3;
    F(sqr)(&Z1Z1, &p->Z); F(sqr)(&Z2Z2, &q->Z);
    F(mul)(&U1, &p->X, &Z2Z2); F(mul)(&U2, &q->X, &Z1Z1);
    F(mul)(&S1, &p->Y, &q->Z); F(mul)(&S1, &S1, &Z2Z2);
    F(mul)(&S2, &q->Y, &p->Z); F(mul)(&S2, &S2, &Z1Z1);
    if (F(eq)(&U1, &U2)) {
        if (F(eq)(&S1, &S2)) { C(jac_dbl)(r, p); return; }
        C(jac_set_inf)(r); return;
    }
    F(sub)(&H, &U2, &U1);
    F(add)(&I, &H, &H); F(sqr)(&I, &I);
    F(mul)(&J, &H, &I);
    F(sub)(&rr, &S2, &S1); F(add)(&rr, &rr, &rr);
    F(mul)(&V, &U1, &I);
    F(sqr)(&X3, &rr); F(sub)(&X3, &X3, &J); F(sub)(&X3, &X3, &V); F(sub)(&X3, &X3, &V);
    F(sub)(&t, &V, &X3); F(mul)(&Y3, &rr, &t);
    F(mul)(&t, &S1, &J); F(add)(&t, &t, &t); F(sub)(&Y3, &Y3, &t);
    F(add)(&Z3, &p->Z, &q->Z); F(sqr)(&Z3, &Z3); F(sub)(&Z3, &Z3, &Z1Z1); F(sub)(&Z3, &Z3, &Z2Z2);
    F(mul)(&Z3, &Z3, &H);
    r->X = X3; r->Y = Y3; r->Z = Z3;
}

/* p Jacobian + q affine */
static void C(jac_madd)(C(jac_t) *r, const C(jac_t) *p, const C(aff_t) *q) {
    if (C(aff_is_inf)(q)) { *r = *p; return; }
    if (C(jac_is_inf)(p)) { C(jac_from_aff)(r, q); return; }
    FT Z1Z1, U2, S2, H, HH, I, J, rr, V, t, X3, Y3, Z3;
    F(sqr)(&Z1Z1, &p->Z);
    F(mul)(&U2, &q->x, &Z1Z1);
    F(mul)(&S2, &q->y, &p->Z); F(mul)(&S2, &S2, &Z1Z1);
    if (F(eq)(&p->X, &U2)) {
        if (F(eq)(&p->Y, &S2)) { C(jac_dbl)(r, p); return; }
        C(jac_set_inf)(r); return;
    }
    F(sub)(&H, &U2, &p->X);
    F(sqr)(&HH, &H);
    F(add)(&I, &HH, &HH); F(add)(&I, &I, &I);
    F(mul)(&J, &H, &I);
    F(sub)(&rr, &S2, &p->Y); F(add)(&rr, &rr, &rr);
    F(mul)(&V, &p->X, &I);
    F(sqr)(&X3, &rr); F(sub)(&X3, &X3, &J); F(sub)(&X3, &X3, &V); F(sub)(&X3, &X3, &V);
    F(sub)(&t, &V, &X3); F(mul)(&Y3, &rr, &t);
    F(mul)(&t, &p->Y, &J); F(add)(&t, &t, &t); F(sub)(&Y3, &Y3, &t);
    F(add)(&Z3, &p->Z, &H); F(sqr)(&Z3, &Z3); F(sub)(&Z3, &Z3, &Z1Z1); F(sub)(&Z3, &Z3, &HH);
    r->X = X3; r->Y = Y3; r->Z = Z3;
}

static void C(jac_neg)(C(jac_t) *r, const C(jac_t) *p) { r->X = p->X; r->Z = p->Z; F(neg)(&r->Y, &p->Y); }

static void C(jac_to_aff)(C(aff_t) *r, const C(jac_t) *p) {
    if (C(jac_is_inf)(p)) { F(set_zero)(&r->x); F(set_zero)(&r->y); return; }
    FT zi, zi2, zi3;
    F(inv)(&zi, &p->Z); F(sqr)(&zi2, &zi); F(mul)(&zi3, &zi2, &zi);
    F(mul)(&r->x, &p->X, &zi2); F(mul)(&r->y, &p->Y, &zi3);
}

/* batch_to_special: Montgomery's simultaneous inversion */
static void C(batch_to_aff)(C(aff_t) *out, const C(jac_t) *in, size_t n) {
    if (!n) return;
    FT *pre = (FT *)malloc(n * sizeof(FT));
    FT acc; F(set_one)(&acc);
    for (size_t i = 0; i < n; i++) {
        pre[i] = acc;
        if (!C(jac_is_inf)(&in[i])) F(mul)(&acc, &acc, &in[i].Z);
    }
    FT inv; F(inv)(&inv, &acc);
    for (size_t i = n; i-- > 0;) {
        if (C(jac_is_inf)(&in[i])) { F(set_zero)(&out[i].x); F(set_zero)(&out[i].y); continue; }
        FT zi, zi2, zi3;
        F(mul)(&zi, &inv, &pre[i]);
        F(mul)(&inv, &inv, &in[i].Z);
        F(sqr)(&zi2, &zi); F(mul)(&zi3, &zi2, &zi);
        F(mul)(&out[i].x, &in[i].X, &zi2); F(mul)(&out[i].y, &in[i].Y, &zi3);
    }
    free(pre);
}

/* scalar (canonical 4xu64) times Jacobian point, plain double-and-add MSB first */
static void C(jac_mul)(C(jac_t) *r, const C(jac_t) *p, const uint64_t k[4]) {
    C(jac_t) acc; C(jac_set_inf)(&acc);
    for (int i = 255; i >= 0; i--) {
        C(jac_dbl)(&acc, &acc);
        if ((k[i >> 6] >> (i & 63)) & 1) C(jac_add)(&acc, &acc, p);
    }
    *r = acc;
}

/* ---- multiexp, bucket method "BDLO12" -------------------------------------------------------
 * Restates algebra::multiexp<policies::multiexp_method_BDLO12> (crypto3-algebra
 * multiexp/detail/multiexp.hpp, absent; libff multiexp.tcc lineage; parameter table included at
 * common.hpp:38, reached from common.hpp:1132-1135):  c = L - (L/3 - 2), L = ceil(log2 n);
 * windows MSB->LSB; per window: c doublings of the result, buckets[id] += base, running sum. */
static void C(multiexp_bdlo12)(C(jac_t) *out, const C(aff_t) *bases, const uint64_t *scalars, size_t n) {
    C(jac_t) result; C(jac_set_inf)(&result);
    if (n == 0) { *out = result; return; }
    size_t L = 0; while (((size_t)1 << L) < n) L++;
    long cl = (long)L - ((long)(L / 3) - 2);
    size_t c = (size_t)(cl < 1 ? 1 : cl);
    size_t num_bits = 0;
    for (size_t i = 0; i < n; i++) {
        const uint64_t *k = scalars + 4 * i;
        for (int b = 255; b >= 0; b--) if ((k[b >> 6] >> (b & 63)) & 1) { if ((size_t)b + 1 > num_bits) num_bits = b + 1; break; }
    }
    size_t num_groups = (num_bits + c - 1) / c;
    size_t nb = (size_t)1 << c;
    C(jac_t) *bucket = (C(jac_t) *)malloc(nb * sizeof(C(jac_t)));
    int result_nonzero = 0;
    for (size_t k = num_groups; k-- > 0;) {
        if (result_nonzero) for (size_t i = 0; i < c; i++) C(jac_dbl)(&result, &result);
        for (size_t i = 0; i < nb; i++) C(jac_set_inf)(&bucket[i]);
        for (size_t i = 0; i < n; i++) {
            const uint64_t *s = scalars + 4 * i;
            size_t id = 0;
            for (size_t j = 0; j < c; j++) {
                size_t bit = k * c + j;
                if (bit < 256 && ((s[bit >> 6] >> (bit & 63)) & 1)) id |= (size_t)1 << j;
            }
            if (id == 0) continue;
            C(jac_madd)(&bucket[id], &bucket[id], &bases[i]);
        }
        C(jac_t) running; C(jac_set_inf)(&running);
        for (size_t i = nb - 1; i > 0; i--) {
            C(jac_add)(&running, &running, &bucket[i]);
            if (!C(jac_is_inf)(&running)) { C(jac_add)(&result, &result, &running); result_nonzero = 1; }
        }
    }
    free(bucket);
    *out = result;
}

/* multiexp_with_mixed_addition: skip 0, add the 1s directly, bucket method on the rest (a2) */
static void C(multiexp_mixed)(C(jac_t) *out, const C(aff_t) *bases, const uint64_t *scalars, size_t n) {
    C(jac_t) acc; C(jac_set_inf)(&acc);
    C(aff_t) *g = (C(aff_t) *)malloc((n ? n : 1) * sizeof(C(aff_t)));
    uint64_t *p = (uint64_t *)malloc((n ? n : 1) * 4 * sizeof(uint64_t));
    size_t m = 0;
    for (size_t i = 0; i < n; i++) {
        const uint64_t *s = scalars + 4 * i;
        if ((s[0] | s[1] | s[2] | s[3]) == 0) continue;
        if (s[0] == 1 && (s[1] | s[2] | s[3]) == 0) { C(jac_madd)(&acc, &acc, &bases[i]); continue; }
        g[m] = bases[i]; memcpy(p + 4 * m, s, 32); m++;
    }
    C(jac_t) rest; C(multiexp_bdlo12)(&rest, g, p, m);
    C(jac_add)(out, &acc, &rest);
    free(g); free(p);
}

/* fixed-base windowed batch exponentiation (libff get_window_table / batch_exp lineage;
 * the Groth16 generator's A/B/H/L queries, common.hpp:916-917): out[i] = scalars[i] * base.
 * 8-bit windows, 32 tables of 255 affine entries. */
static void C(batch_mul_fixed)(C(aff_t) *out, const C(aff_t) *base, const uint64_t *scalars, size_t n) {
    enum { WB = 8, NW = 32, TE = 255 };
    C(jac_t) *tj = (C(jac_t) *)malloc((size_t)NW * TE * sizeof(C(jac_t)));
    C(aff_t) *ta = (C(aff_t) *)malloc((size_t)NW * TE * sizeof(C(aff_t)));
    C(jac_t) wb; C(jac_from_aff)(&wb, base);
    for (int w = 0; w < NW; w++) {
        C(jac_t) acc = wb;
        for (int e = 0; e < TE; e++) {
            tj[w * TE + e] = acc;
            C(jac_add)(&acc, &acc, &wb);
        }
        wb = acc;            /* 256 * previous window base */
    }
    C(batch_to_aff)(ta, tj, (size_t)NW * TE);
    C(jac_t) *res = (C(jac_t) *)malloc((n ? n : 1) * sizeof(C(jac_t)));
    for (size_t i = 0; i < n; i++) {
        const uint64_t *s = scalars + 4 * i;
        C(jac_t) acc; C(jac_set_inf)(&acc);
        for (int w = 0; w < NW; w++) {
            unsigned d = (unsigned)((s[w >> 3] >> ((w & 7) * 8)) & 0xFF);
            if (d) C(jac_madd)(&acc, &acc, &ta[w * TE + d - 1]);
        }
        res[i] = acc;
    }
    C(batch_to_aff)(out, res, n);
    free(tj); free(ta); free(res);
}
