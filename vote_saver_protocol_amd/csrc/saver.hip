// SAVER wrapper around the prover: elgamal_verifiable over BLS12-381 as the reference's vote phase calls it
//     generate_keypair<elgamal_verifiable>(rnd[3 * msg_size + 2], {gg_keypair, msg_size})      bin/cli/include/nil/vote_saver/common.hpp:921-931
//     encrypt<...>(m_field, {d(), pk_eid, gg_keypair, primary_input, auxiliary_input})          common.hpp:1131-1135  <- the one statement
//                                                                                              under which the whole hot path runs
//     rerandomize<...>(rnd[3], cipher_text.first, {pk_eid, gg_keypair, cipher_text.second})     common.hpp:1138-1145
// (crypto3-pubkey elgamal_verifiable.hpp, absent submodule, /root/reference/.gitmodules:50; the scheme is SAVER, Lee-Choi-Kim-Oh,
// fig. 3 -- [UPSTREAM-KNOWLEDGE] for member names and the order random values are consumed in; the equations are checked by the
// pairing on the test side).  msg_size = 25 in the reference (common.hpp:163).
//
// Where the work runs.  The Groth16 proof inside encrypt is vsp_groth16_prove: GPU.  Everything else is O(msg_size) group operations:
// msg_size + 2 products of ONE scalar with the fixed points of the public key, a handful of one-off scalar multiplications of fresh
// points (the proof elements).  A serial chain of 255 doublings is what a GPU lane is ~50x slower at than a CPU core,
// so these run on the host -- over per-key fixed-base tables (4-bit windows, built once by vsp_saver_pk_load), and INSIDE the window
// in which the host otherwise only waits for the GPU proof (prove_queued's host-overlap hook): the ciphertext costs no wall time.
// Decryption and the two verifications are pairing work on the verifier / tally side -- outside the prover's hot path.
#include <future>

#include "common.h"

namespace vsp {

int prove_with_overlap(vsp_ctx *ctx, const vsp_r1cs *cs, const vsp_pk *pk, const uint64_t *witness, const uint64_t r[4], const uint64_t s[4],
                       const uint64_t *saver_P1, const uint64_t *saver_r_enc, uint64_t A_out[12], uint64_t B_out[24], uint64_t C_out[12],
                       uint8_t proof_out[192], const std::function<void()> *overlap);

// k * P for a fixed P: table[w][d - 1] = d * 16^w * P (affine), 64 windows x 15 entries; a product is <= 64 mixed additions
struct FixedBase {
    std::vector<Affine<HFp>> tab;
    bool inf = false;
    void build(const Affine<HFp> &p) {
        inf = is_inf(p);
        if (inf) return;
        std::vector<XYZZ<HFp>> pts(64 * 15);
        XYZZ<HFp> base = xyzz_from_affine(p);
        for (int w = 0; w < 64; w++) {
            XYZZ<HFp> acc = base;
            for (int d = 1; d <= 15; d++) { pts[w * 15 + d - 1] = acc; xyzz_add(acc, base); }
            base = acc;                                            // 16 * base
        }
        // batch normalisation (one inversion): prefix products of ZZZ
        std::vector<HFp> pre(pts.size());
        HFp run = HFp::one();
        for (size_t i = 0; i < pts.size(); i++) { pre[i] = run; run = mul(run, pts[i].ZZZ); }
        HFp ri = inv(run);
        tab.resize(pts.size());
        for (size_t i = pts.size(); i-- > 0;) {
            HFp zi3 = mul(ri, pre[i]);                               // 1 / ZZZ_i
            ri = mul(ri, pts[i].ZZZ);
            HFp zi = mul(zi3, pts[i].ZZ), zi2 = sqr(zi);
            tab[i].x = mul(pts[i].X, zi2); tab[i].y = mul(pts[i].Y, zi3);
        }
    }
    XYZZ<HFp> mul_scalar(const uint64_t k[4]) const {
        XYZZ<HFp> acc = XYZZ<HFp>::inf();
        if (inf) return acc;
        for (int w = 0; w < 64; w++) {
            unsigned d = (unsigned)(k[w >> 4] >> ((w & 15) * 4)) & 15u;
            if (d) xyzz_madd(acc, tab[w * 15 + d - 1]);
        }
        return acc;
    }
};

static bool fr_canon(const uint64_t *k) {
    for (int i = 3; i >= 0; i--) { if (k[i] < FrP64::MOD[i]) return true; if (k[i] > FrP64::MOD[i]) return false; }
    return false;
}
static int bit_length(const uint64_t k[4]) {
    for (int i = 3; i >= 0; i--) if (k[i]) return 64 * i + 64 - __builtin_clzll(k[i]);
    return 0;
}
// canonical coordinates below p and the point on the curve (or infinity)
static bool g1_valid(const uint64_t *p) {
    auto below_p = [](const uint64_t *l) { for (int i = 5; i >= 0; i--) { if (l[i] < FpP64::MOD[i]) return true; if (l[i] > FpP64::MOD[i]) return false; } return false; };
    if (!below_p(p) || !below_p(p + 6)) return false;
    Affine<HFp> a = host_load_g1(p);
    if (is_inf(a)) return true;
    HFp four = dbl(dbl(HFp::one()));
    return eq(sqr(a.y), add(mul(sqr(a.x), a.x), four));
}
static bool g2_valid(const uint64_t *p) {                      // coordinates below p, on y^2 = x^3 + 4 (1 + u) (all zero = infinity)
    auto below_p = [](const uint64_t *l) { for (int i = 5; i >= 0; i--) { if (l[i] < FpP64::MOD[i]) return true; if (l[i] > FpP64::MOD[i]) return false; } return false; };
    for (int k = 0; k < 4; k++) if (!below_p(p + 6 * k)) return false;
    Affine<HFp2> a = host_load_g2(p);
    if (is_inf(a)) return true;
    HFp2 b; b.c0 = dbl(dbl(HFp::one())); b.c1 = b.c0;
    return eq(sqr(a.y), add(mul(sqr(a.x), a.x), b));
}

}  // namespace vsp

using namespace vsp;

struct vsp_saver_pk {
    size_t n = 0;
    std::vector<uint64_t> words;                   // the flat public key
    std::vector<uint64_t> gabc;                    // G_0 .. G_n (canonical)
    std::vector<FixedBase> X;                      // delta_g1, delta_s_g1[0..n), delta_sum_s_g1   (n + 2 tables: the bases of the ciphertext)
    FixedBase P2;                                  // gamma_inverse_sum_s_g1
    std::vector<Affine<HFp>> G, Y;                 // message bases G_i (i = 1..n) and t_g1[i]
};

static size_t pk_words(size_t n) { return 12 + 12 * n + 12 * n + 24 * (n + 1) + 12 + 12; }
static size_t vk_words(size_t n) { return 24 + 24 * n + 24 * n; }

// one product m * P for a (usually tiny) message value
static void add_msg_term(XYZZ<HFp> &acc, const Affine<HFp> &p, const uint64_t m[4]) {
    int bits = bit_length(m);
    if (!bits) return;
    if (bits == 1) { xyzz_madd(acc, p); return; }
    XYZZ<HFp> t = xyzz_mul_scalar(xyzz_from_affine(p), m, bits);
    xyzz_add(acc, t);
}

extern "C" {

size_t vsp_saver_pk_words(size_t msg_size) { return pk_words(msg_size); }
size_t vsp_saver_vk_words(size_t msg_size) { return vk_words(msg_size); }

int vsp_saver_keygen(vsp_ctx *ctx, size_t n, const uint64_t delta_g1[12], const uint64_t gamma_g1[12], const uint64_t *gamma_abc_g1,
                     const uint64_t *rnd, uint64_t *pk_out, uint64_t sk_out[4], uint64_t *vk_out) {
    // host only: ctx may be NULL (then there is no error text, only the status)
    if (!n || !delta_g1 || !gamma_g1 || !gamma_abc_g1 || !rnd || !pk_out || !sk_out || !vk_out) return set_error(ctx, VSP_ERR_ARG, "saver_keygen: null argument");
    for (size_t i = 0; i < 3 * n + 2; i++) if (!fr_canon(rnd + 4 * i)) return set_error(ctx, VSP_ERR_ARG, "saver_keygen: a random value is not canonical (>= r)");
    if (!g1_valid(delta_g1) || !g1_valid(gamma_g1)) return set_error(ctx, VSP_ERR_ARG, "saver_keygen: delta_g1 / gamma_g1 is not a curve point");
    for (size_t i = 0; i <= n; i++) if (!g1_valid(gamma_abc_g1 + 12 * i)) return set_error(ctx, VSP_ERR_ARG, "saver_keygen: gamma_ABC_g1 entry is not a curve point");
    const uint64_t *s = rnd, *v = rnd + 4 * n, *t = rnd + 8 * n, *rho = rnd + 4 * (3 * n + 1);
    uint64_t *p_delta_s = pk_out + 12, *p_t_g1 = p_delta_s + 12 * n, *p_t_g2 = p_t_g1 + 12 * n, *p_dsum = p_t_g2 + 24 * (n + 1), *p_ginv = p_dsum + 12;
    memcpy(pk_out, delta_g1, 96);
    FixedBase fd, fg; fd.build(host_load_g1(delta_g1)); fg.build(host_load_g1(gamma_g1));
    // generator of G2 (public constant)
    static const uint64_t G2_GEN[24] = {0xd48056c8c121bdb8ULL, 0x0bac0326a805bbefULL, 0xb4510b647ae3d177ULL, 0xc6e47ad4fa403b02ULL, 0x260805272dc51051ULL, 0x024aa2b2f08f0a91ULL,
                                        0xe5ac7d055d042b7eULL, 0x334cf11213945d57ULL, 0xb5da61bbdc7f5049ULL, 0x596bd0d09920b61aULL, 0x7dacd3a088274f65ULL, 0x13e02b6052719f60ULL,
                                        0xe193548608b82801ULL, 0x923ac9cc3baca289ULL, 0x6d429a695160d12cULL, 0xadfd9baa8cbdd3a7ULL, 0x8cc9cdc6da2e351aULL, 0x0ce5d527727d6e11ULL,
                                        0xaaa9075ff05f79beULL, 0x3f370d275cec1da1ULL, 0x267492ab572e99abULL, 0xcb3e287e85a763afULL, 0x32acd2b02bc28b99ULL, 0x0606c4a02ea734ccULL};
    XYZZ<HFp2> h = xyzz_from_affine(host_load_g2(G2_GEN));
    auto g2mul = [&](const HFr &k, uint64_t *out) { uint64_t c[4]; host_store_canon(c, k); host_store_g2(out, xyzz_to_affine(xyzz_mul_scalar(h, c, 255))); };
    HFr sum_s = HFr::one(), sum_ts = host_load_canon<HFr>(t);
    for (size_t i = 0; i < n; i++) {
        host_store_g1(p_delta_s + 12 * i, xyzz_to_affine(fd.mul_scalar(s + 4 * i)));
        host_store_g1(p_t_g1 + 12 * i, xyzz_to_affine(xyzz_mul_scalar(xyzz_from_affine(host_load_g1(gamma_abc_g1 + 12 * (i + 1))), t + 4 * (i + 1), 255)));
        HFr si = host_load_canon<HFr>(s + 4 * i);
        sum_s = add(sum_s, si);
        sum_ts = add(sum_ts, mul(host_load_canon<HFr>(t + 4 * (i + 1)), si));
    }
    for (size_t j = 0; j <= n; j++) g2mul(host_load_canon<HFr>(t + 4 * j), p_t_g2 + 24 * j);
    uint64_t c4[4];
    host_store_canon(c4, sum_ts); host_store_g1(p_dsum, xyzz_to_affine(fd.mul_scalar(c4)));
    host_store_canon(c4, neg(sum_s)); host_store_g1(p_ginv, xyzz_to_affine(fg.mul_scalar(c4)));
    memcpy(sk_out, rho, 32);
    HFr r_ = host_load_canon<HFr>(rho);
    g2mul(r_, vk_out);
    for (size_t i = 0; i < n; i++) {
        HFr si = host_load_canon<HFr>(s + 4 * i), vi = host_load_canon<HFr>(v + 4 * i);
        g2mul(mul(si, vi), vk_out + 24 + 24 * i);
        g2mul(mul(r_, vi), vk_out + 24 + 24 * n + 24 * i);
    }
    return VSP_OK;
}

vsp_saver_pk *vsp_saver_pk_load(vsp_ctx *ctx, size_t n, const uint64_t *pk_in, const uint64_t *gamma_abc_g1) {
    if (!n || !pk_in || !gamma_abc_g1) { set_error(ctx, VSP_ERR_ARG, "saver_pk_load: null argument"); return nullptr; }
    const uint64_t *p_delta_s = pk_in + 12, *p_t_g1 = p_delta_s + 12 * n, *p_dsum = p_t_g1 + 12 * n + 24 * (n + 1), *p_ginv = p_dsum + 12;
    for (size_t i = 0; i < 2 * n + 1; i++) if (!g1_valid(pk_in + 12 * i)) { set_error(ctx, VSP_ERR_ARG, "saver_pk_load: a G1 element of the key is not a curve point"); return nullptr; }
    if (!g1_valid(p_dsum) || !g1_valid(p_ginv)) { set_error(ctx, VSP_ERR_ARG, "saver_pk_load: a G1 element of the key is not a curve point"); return nullptr; }
    for (size_t i = 0; i <= n; i++) if (!g1_valid(gamma_abc_g1 + 12 * i)) { set_error(ctx, VSP_ERR_ARG, "saver_pk_load: gamma_ABC_g1 entry is not a curve point"); return nullptr; }
    vsp_saver_pk *k = new vsp_saver_pk();
    k->n = n;
    k->words.assign(pk_in, pk_in + pk_words(n));
    k->gabc.assign(gamma_abc_g1, gamma_abc_g1 + 12 * (n + 1));
    k->X.resize(n + 2);
    k->X[0].build(host_load_g1(pk_in));
    for (size_t i = 0; i < n; i++) k->X[i + 1].build(host_load_g1(p_delta_s + 12 * i));
    k->X[n + 1].build(host_load_g1(p_dsum));
    k->P2.build(host_load_g1(p_ginv));
    k->G.resize(n); k->Y.resize(n);
    for (size_t i = 0; i < n; i++) { k->G[i] = host_load_g1(gamma_abc_g1 + 12 * (i + 1)); k->Y[i] = host_load_g1(p_t_g1 + 12 * i); }
    return k;
}
void vsp_saver_pk_free(vsp_ctx *, vsp_saver_pk *k) { delete k; }
size_t vsp_saver_pk_msg_size(const vsp_saver_pk *k) { return k ? k->n : 0; }

int vsp_saver_encrypt(vsp_ctx *ctx, const vsp_saver_pk *spk, const vsp_r1cs *cs, const vsp_pk *pk, const uint64_t *msg, const uint64_t *witness,
                      const uint64_t r_enc[4], const uint64_t r[4], const uint64_t s[4],
                      uint64_t *ct_out, uint64_t A_out[12], uint64_t B_out[24], uint64_t C_out[12], uint8_t proof_out[192]) {
    if (!ctx) return VSP_ERR_ARG;
    if (!spk || !cs || !pk || !msg || !witness || !r_enc || !r || !s || !ct_out) return set_error(ctx, VSP_ERR_ARG, "saver_encrypt: null argument");
    const size_t n = spk->n;
    if (cs->num_inputs < n) return set_error(ctx, VSP_ERR_ARG, "saver_encrypt: the constraint system has fewer public inputs than message blocks");
    if (!fr_canon(r_enc)) return set_error(ctx, VSP_ERR_ARG, "saver_encrypt: r_enc must be canonical (< r)");
    for (size_t i = 0; i < n; i++) {
        if (!fr_canon(msg + 4 * i)) return set_error(ctx, VSP_ERR_ARG, "saver_encrypt: a message block is not canonical (>= r)");
        // the message IS the first msg_size public inputs (common.hpp:1110-1135: m_block is allocated first)
        if (memcmp(msg + 4 * i, witness + 4 * i, 32) != 0) return set_error(ctx, VSP_ERR_ARG, "saver_encrypt: message differs from the first public inputs of the witness");
    }
    // ciphertext: c_0 = r X_0 | c_i = r X_i + m_i G_i | psi = r P_1 + sum m_i Y_i -- computed while the GPU proves
    std::function<void()> overlap = [&]() {
        XYZZ<HFp> psi = spk->X[n + 1].mul_scalar(r_enc);
        host_store_g1(ct_out, xyzz_to_affine(spk->X[0].mul_scalar(r_enc)));
        for (size_t i = 0; i < n; i++) {
            XYZZ<HFp> c = spk->X[i + 1].mul_scalar(r_enc);
            add_msg_term(c, spk->G[i], msg + 4 * i);
            host_store_g1(ct_out + 12 * (i + 1), xyzz_to_affine(c));
            add_msg_term(psi, spk->Y[i], msg + 4 * i);
        }
        host_store_g1(ct_out + 12 * (n + 1), xyzz_to_affine(psi));
    };
    const uint64_t *p_ginv = spk->words.data() + 12 + 12 * n + 12 * n + 24 * (n + 1) + 12;
    return prove_with_overlap(ctx, cs, pk, witness, r, s, p_ginv, r_enc, A_out, B_out, C_out, proof_out, &overlap);
}

int vsp_saver_rerandomize(vsp_ctx *ctx, const vsp_saver_pk *spk, const uint64_t delta_g2[24], const uint64_t rnd[12],
                          uint64_t *ct, uint64_t A[12], uint64_t B[24], uint64_t C[12], uint8_t proof_out[192]) {
    // host only: ctx may be NULL
    if (!spk || !delta_g2 || !rnd || !ct || !A || !B || !C) return set_error(ctx, VSP_ERR_ARG, "saver_rerandomize: null argument");
    const size_t n = spk->n;
    const uint64_t *rp = rnd, *z1 = rnd + 4, *z2 = rnd + 8;
    if (!fr_canon(rp) || !fr_canon(z1) || !fr_canon(z2)) return set_error(ctx, VSP_ERR_ARG, "saver_rerandomize: random values must be canonical (< r)");
    HFr a = host_load_canon<HFr>(z1), b = host_load_canon<HFr>(z2);
    if (is_zero(a)) return set_error(ctx, VSP_ERR_ARG, "saver_rerandomize: z1 must be invertible");
    for (size_t i = 0; i < n + 2; i++) if (!g1_valid(ct + 12 * i)) return set_error(ctx, VSP_ERR_ARG, "saver_rerandomize: a ciphertext element is not a curve point");
    if (!g1_valid(A) || !g1_valid(C) || !g2_valid(B)) return set_error(ctx, VSP_ERR_ARG, "saver_rerandomize: a proof element is not a curve point");
    if (!g2_valid(delta_g2)) return set_error(ctx, VSP_ERR_ARG, "saver_rerandomize: delta_g2 is not a curve point");
    uint64_t zi4[4], zz4[4];
    host_store_canon(zi4, inv(a)); host_store_canon(zz4, mul(a, b));
    // the four one-off scalar multiplications are independent: the two in G2 (the long ones) run beside the G1 work
    Affine<HFp2> b_in = host_load_g2(B), d2 = host_load_g2(delta_g2);
    auto fB = std::async(std::launch::async, [&]() { return xyzz_mul_scalar(xyzz_from_affine(b_in), zi4, 255); });
    auto fD = std::async(std::launch::async, [&]() { return xyzz_mul_scalar(xyzz_from_affine(d2), z2, 255); });
    Affine<HFp> a_in = host_load_g1(A);
    auto fA = std::async(std::launch::async, [&]() { return xyzz_mul_scalar(xyzz_from_affine(a_in), z1, 255); });
    // ct_i += r' X_i: n + 2 fixed-base multiplications (64 mixed additions each), the upper half on another thread; all the G1
    // results of the call (n + 2 ciphertext elements, A, C) share ONE field inversion (prefix products of ZZZ) -- a Fermat
    // inversion per element was a third of the call
    std::vector<XYZZ<HFp>> pts(n + 4);
    auto ct_range = [&](size_t lo, size_t hi) {
        for (size_t i = lo; i < hi; i++) { XYZZ<HFp> t = spk->X[i].mul_scalar(rp); xyzz_madd(t, host_load_g1(ct + 12 * i)); pts[i] = t; }
    };
    const size_t mid = (n + 2) / 2;
    auto fC = std::async(std::launch::async, [&]() { ct_range(mid, n + 2); });
    XYZZ<HFp> zzA = xyzz_mul_scalar(xyzz_from_affine(a_in), zz4, 255);
    ct_range(0, mid);
    XYZZ<HFp> nC = spk->P2.mul_scalar(rp);
    xyzz_add(nC, zzA);
    xyzz_madd(nC, host_load_g1(C));
    fC.get();
    pts[n + 2] = fA.get(); pts[n + 3] = nC;
    {
        std::vector<HFp> pre(pts.size());
        HFp run = HFp::one();
        for (size_t i = 0; i < pts.size(); i++) { pre[i] = run; if (!is_inf(pts[i])) run = mul(run, pts[i].ZZZ); }
        HFp ri = inv(run);
        for (size_t i = pts.size(); i-- > 0;) {
            Affine<HFp> q; q.x = HFp::zero(); q.y = HFp::zero();
            if (!is_inf(pts[i])) {
                HFp zi3 = mul(ri, pre[i]);                               // 1 / ZZZ_i
                ri = mul(ri, pts[i].ZZZ);
                HFp zi = mul(zi3, pts[i].ZZ), zi2 = sqr(zi);
                q.x = mul(pts[i].X, zi2); q.y = mul(pts[i].Y, zi3);
            }
            host_store_g1(i < n + 2 ? ct + 12 * i : (i == n + 2 ? A : C), q);
        }
    }
    XYZZ<HFp2> nB = fB.get(); { XYZZ<HFp2> w = fD.get(); xyzz_add(nB, w); }
    host_store_g2(B, xyzz_to_affine(nB));
    if (proof_out) { vsp_g1_compress(A, proof_out); vsp_g2_compress(B, proof_out + 48); vsp_g1_compress(C, proof_out + 144); }
    return VSP_OK;
}

}  // extern "C"
