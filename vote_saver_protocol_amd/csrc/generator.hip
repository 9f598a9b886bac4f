// Groth16 generator on the GPU: r1cs_gg_ppzksnark_generator with explicit toxic waste (t, alpha, beta, gamma, delta).
//
// Replaces, as the first "next" row of SURVEY.md 8(f), zk::generate<proof_system>(r1cs) of crypto3-zk (absent submodule,
// /root/reference/.gitmodules:11-12), called at bin/cli/include/nil/vote_saver/common.hpp:916-917 -- the multi-minute CPU
// setup of the reference.  Steps (libsnark r1cs_to_qap::instance_map_with_evaluation + r1cs_gg_ppzksnark_generator lineage):
//   u_j   = L_j(t) over the constraint system's domain, basic or step radix-2  (domain.hip: one batched inversion per 32 elements)
//   A_i(t), B_i(t), C_i(t) = sum_j coef_{j,i} u_j  (k_qap_columns: one thread per variable over a column-major copy)
//   exponents: A_i, B_i, t^i Z(t)/delta, (beta A_i + alpha B_i + C_i)/delta | /gamma
//   queries = exponent * generator               (vsp_fixed_base_mul: 8-bit windowed fixed-base, batch normalisation)
#include "common.h"

namespace vsp {
namespace {

static constexpr unsigned GEN_PW = 11;          // two-level power tables: x^i = lo[i & 2047] * hi[i >> 11]

struct GenConsts { Fr alpha, beta, gamma_inv, delta_inv, zt_delta_inv; };

// X_i(t) = sum over the column's entries of coef * u[row]   (+ u[nc + i] for the input rows of A)
__global__ __launch_bounds__(256) void k_qap_columns(const uint32_t *cp, const uint32_t *ri, const Fr *cot, const Fr *u, size_t ncols,
                                                     size_t nc, size_t ni, int is_a, Fr *out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ncols) return;
    Fr acc = Fr::zero();
    for (uint32_t e = cp[i]; e < cp[i + 1]; e++) acc = add(acc, mul(cot[e], u[ri[e]]));
    if (is_a && i <= ni) acc = add(acc, u[nc + i]);
    out[i] = acc;
}

// exponent vectors, canonical: A_sc = A, B_sc = B, L_sc / ABC_sc = (beta A + alpha B + C) / delta | / gamma
__global__ __launch_bounds__(256) void k_key_exponents(const Fr *At, const Fr *Bt, const Fr *Ct, const GenConsts *kc, size_t ncols, size_t ni,
                                                       Fr *A_sc, Fr *B_sc, Fr *L_sc, Fr *ABC_sc) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ncols) return;
    Fr a = At[i], b = Bt[i];
    A_sc[i] = from_mont(a);
    B_sc[i] = from_mont(b);
    Fr x = add(add(mul(kc->beta, a), mul(kc->alpha, b)), Ct[i]);
    if (i <= ni) ABC_sc[i] = from_mont(mul(x, kc->gamma_inv));
    else L_sc[i - ni - 1] = from_mont(mul(x, kc->delta_inv));
}
// H_sc[i] = t^i * Z(t) / delta, canonical
__global__ __launch_bounds__(256) void k_h_exponents(const Fr *t_lo, const Fr *t_hi, const GenConsts *kc, size_t count, Fr *H_sc) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    Fr ti = mul(t_lo[i & ((1u << GEN_PW) - 1u)], t_hi[i >> GEN_PW]);
    H_sc[i] = from_mont(mul(ti, kc->zt_delta_inv));
}
__global__ __launch_bounds__(256) void k_affine_from_mont_g1(const G1Affine *in, G1Affine *out, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    G1Affine p = in[i]; p.x = from_mont(p.x); p.y = from_mont(p.y); out[i] = p;
}
__global__ __launch_bounds__(256) void k_affine_from_mont_g2(const G2Affine *in, G2Affine *out, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    G2Affine p = in[i]; p.x = from_mont(p.x); p.y = from_mont(p.y); out[i] = p;
}

static Fr dev(const HFr &h) { Fr d; memcpy(&d, &h, sizeof(Fr)); return d; }
// the generator's scratch holds functions of the toxic waste (powers of t, exponent vectors): zeroed before it goes back to the allocator
static void free_dev(DevBuf &b) { if (b.p) { hipMemset(b.p, 0, b.cap); hipFree(b.p); } b.p = nullptr; b.cap = 0; }

static const uint64_t G1_GEN_L[12] = {0xfb3af00adb22c6bbULL, 0x6c55e83ff97a1aefULL, 0xa14e3a3f171bac58ULL, 0xc3688c4f9774b905ULL, 0x2695638c4fa9ac0fULL, 0x17f1d3a73197d794ULL,
                                      0x0caa232946c5e7e1ULL, 0xd03cc744a2888ae4ULL, 0x00db18cb2c04b3edULL, 0xfcf5e095d5d00af6ULL, 0xa09e30ed741d8ae4ULL, 0x08b3f481e3aaa0f1ULL};
static const uint64_t G2_GEN_L[24] = {0xd48056c8c121bdb8ULL, 0x0bac0326a805bbefULL, 0xb4510b647ae3d177ULL, 0xc6e47ad4fa403b02ULL, 0x260805272dc51051ULL, 0x024aa2b2f08f0a91ULL,
                                      0xe5ac7d055d042b7eULL, 0x334cf11213945d57ULL, 0xb5da61bbdc7f5049ULL, 0x596bd0d09920b61aULL, 0x7dacd3a088274f65ULL, 0x13e02b6052719f60ULL,
                                      0xe193548608b82801ULL, 0x923ac9cc3baca289ULL, 0x6d429a695160d12cULL, 0xadfd9baa8cbdd3a7ULL, 0x8cc9cdc6da2e351aULL, 0x0ce5d527727d6e11ULL,
                                      0xaaa9075ff05f79beULL, 0x3f370d275cec1da1ULL, 0x267492ab572e99abULL, 0xcb3e287e85a763afULL, 0x32acd2b02bc28b99ULL, 0x0606c4a02ea734ccULL};

}  // anonymous namespace
}  // namespace vsp

using namespace vsp;

extern "C" {

void vsp_keypair_free(vsp_ctx *ctx, vsp_keypair *kp) {
    if (!kp) return;
    if (kp->pk) vsp_pk_free(ctx, kp->pk);
    for (int i = 0; i < 6; i++) if (kp->q[i]) vsp_bases_free(ctx, kp->q[i]);
    delete kp;
}

vsp_keypair *vsp_groth16_generate(vsp_ctx *ctx, const vsp_r1cs *cs, const uint64_t toxic[20], int precompute) {
    if (!ctx) return nullptr;
    if (!cs || !toxic) { set_error(ctx, VSP_ERR_ARG, "generate: null argument"); return nullptr; }
    hipSetDevice(ctx->device);
    hipStream_t st = ctx->stream;
    const size_t nv = cs->num_vars, ni = cs->num_inputs, nc = cs->num_constraints, ncols = nv + 1;
    const size_t m = cs->dom.m;
    HFr t = host_load_canon<HFr>(toxic), alpha = host_load_canon<HFr>(toxic + 4), beta = host_load_canon<HFr>(toxic + 8);
    HFr gamma = host_load_canon<HFr>(toxic + 12), delta = host_load_canon<HFr>(toxic + 16);
    if (is_zero(gamma) || is_zero(delta)) { set_error(ctx, VSP_ERR_ARG, "generate: gamma and delta must be non-zero"); return nullptr; }
    HFr Zt = domain_vanishing(&cs->dom, t);
    GenConsts k;
    k.alpha = dev(alpha); k.beta = dev(beta);
    k.gamma_inv = dev(inv(gamma)); k.delta_inv = dev(inv(delta)); k.zt_delta_inv = dev(mul(Zt, inv(delta)));

    DevBuf t_lo, t_hi, u, At, Bt, Ct, A_sc, B_sc, H_sc, L_sc, ABC_sc, pts, kbuf;
    vsp_keypair *kp = new vsp_keypair();
    auto fail = [&](const char *msg) -> vsp_keypair * {
        if (msg) set_error(ctx, VSP_ERR_HIP, msg);
        DevBuf *all[] = {&t_lo, &t_hi, &u, &At, &Bt, &Ct, &A_sc, &B_sc, &H_sc, &L_sc, &ABC_sc, &pts, &kbuf};
        for (DevBuf *b : all) free_dev(*b);
        vsp_keypair_free(ctx, kp);
        return nullptr;
    };
    const size_t hi_count = (m + ((size_t)1 << GEN_PW) - 1) >> GEN_PW;     // m need not be a power of two (step domain)
    if (upload_power_tables(ctx, t, hi_count, t_lo, t_hi) != VSP_OK) return fail(nullptr);
    size_t sizes[] = {m, ncols, ncols, ncols, ncols, ncols, m, nv - ni + 1, ni + 1};
    DevBuf *bufs[] = {&u, &At, &Bt, &Ct, &A_sc, &B_sc, &H_sc, &L_sc, &ABC_sc};
    for (int i = 0; i < 9; i++) if (ensure(ctx, *bufs[i], sizes[i] * sizeof(Fr)) != VSP_OK) return fail(nullptr);

    if (ensure(ctx, kbuf, sizeof(GenConsts)) != VSP_OK) return fail(nullptr);
    if (hipMemcpyAsync(kbuf.p, &k, sizeof(GenConsts), hipMemcpyHostToDevice, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return fail("generate: constants upload failed");
    const GenConsts *kd = (const GenConsts *)kbuf.p;
    if (domain_lagrange_device(ctx, &cs->dom, t, (Fr *)u.p) != VSP_OK) return fail(nullptr);
    const unsigned cblk = (unsigned)((ncols + 255) / 256);
    Fr *Xt[3] = {(Fr *)At.p, (Fr *)Bt.p, (Fr *)Ct.p};
    for (int mm = 0; mm < 3; mm++)
        hipLaunchKernelGGL(k_qap_columns, dim3(cblk), dim3(256), 0, st, (const uint32_t *)cs->cp[mm], (const uint32_t *)cs->ri[mm], (const Fr *)cs->cot[mm],
                           (const Fr *)u.p, ncols, nc, ni, mm == 0 ? 1 : 0, Xt[mm]);
    hipLaunchKernelGGL(k_key_exponents, dim3(cblk), dim3(256), 0, st, (const Fr *)At.p, (const Fr *)Bt.p, (const Fr *)Ct.p, kd, ncols, ni,
                       (Fr *)A_sc.p, (Fr *)B_sc.p, (Fr *)L_sc.p, (Fr *)ABC_sc.p);
    if (m > 1) hipLaunchKernelGGL(k_h_exponents, dim3((unsigned)((m - 1 + 255) / 256)), dim3(256), 0, st, (const Fr *)t_lo.p, (const Fr *)t_hi.p, kd, m - 1, (Fr *)H_sc.p);
    if (hipGetLastError() != hipSuccess) return fail("generate: kernel launch failed");

    // batch exponentiation of the six queries
    struct Q { const Fr *sc; size_t n; int group; } qs[6] = {{(const Fr *)A_sc.p, ncols, 1}, {(const Fr *)B_sc.p, ncols, 1}, {(const Fr *)B_sc.p, ncols, 2},
                                                              {(const Fr *)H_sc.p, m - 1, 1}, {(const Fr *)L_sc.p, nv - ni, 1}, {(const Fr *)ABC_sc.p, ni + 1, 1}};
    for (int i = 0; i < 6; i++) {
        size_t esz = qs[i].group == 1 ? sizeof(G1Affine) : sizeof(G2Affine);
        if (ensure(ctx, pts, (qs[i].n ? qs[i].n : 1) * esz) != VSP_OK) return fail(nullptr);
        int rc = qs[i].group == 1 ? fixed_base_mul_g1(ctx, qs[i].sc, qs[i].n, pts.p) : fixed_base_mul_g2(ctx, qs[i].sc, qs[i].n, pts.p);
        if (rc != VSP_OK) return fail(nullptr);
        kp->q[i] = bases_create(ctx, qs[i].group, pts.p, true, qs[i].n, BASES_OWN);      // multiples of the generators: in the subgroup by construction
        if (!kp->q[i]) return fail(nullptr);
        // precompute: bit 0 = the recommended set A, B(G1), B(G2), L; bits 1..5 select A, B(G1), B(G2), H, L one by one.  H stays plain
        // in the recommended set: its scalars are dense, so the window size does not shrink, the bucket reduction over 16 window sets is
        // small beside 2^20 * 16 additions, and the plain 128 MB table is read out of the Infinity Cache where the 2 GB table of window
        // multiples misses it (2^20 constraints: 9.15 ms per proof against 9.45 with H precomputed too, and 2 GB less key memory)
        const bool pre_this = i < 5 && (((precompute & 1) && i != 3) || ((precompute >> (i + 1)) & 1));
        long pre_window = 0; { auto itw = ctx->opts.find("generate_precompute_window"); if (itw != ctx->opts.end() && itw->second >= 8 && itw->second <= 22) pre_window = itw->second; }
        if (pre_this && vsp_bases_precompute(ctx, kp->q[i], (unsigned)pre_window) != VSP_OK) return fail(nullptr);      // (0: by the query's size)
    }
    // single elements on the host
    Affine<HFp> g1 = host_load_g1(G1_GEN_L); Affine<HFp2> g2 = host_load_g2(G2_GEN_L);
    auto mul1 = [&](const uint64_t *sc, uint64_t *out) { host_store_g1(out, xyzz_to_affine(xyzz_mul_scalar(xyzz_from_affine(g1), sc, 255))); };
    auto mul2 = [&](const uint64_t *sc, uint64_t *out) { host_store_g2(out, xyzz_to_affine(xyzz_mul_scalar(xyzz_from_affine(g2), sc, 255))); };
    mul1(toxic + 4, kp->alpha_g1); mul1(toxic + 8, kp->beta_g1); mul1(toxic + 16, kp->delta_g1);
    mul2(toxic + 8, kp->beta_g2); mul2(toxic + 16, kp->delta_g2); mul2(toxic + 12, kp->gamma_g2); mul1(toxic + 12, kp->gamma_g1);
    kp->pk = vsp_pk_create(ctx, kp->alpha_g1, kp->beta_g1, kp->beta_g2, kp->delta_g1, kp->delta_g2, kp->q[0], kp->q[1], kp->q[2], kp->q[3], kp->q[4]);
    if (!kp->pk || hipStreamSynchronize(st) != hipSuccess) return fail("generate: failed");
    DevBuf *all[] = {&t_lo, &t_hi, &u, &At, &Bt, &Ct, &A_sc, &B_sc, &H_sc, &L_sc, &ABC_sc, &pts, &kbuf};
    for (DevBuf *b : all) free_dev(*b);
    return kp;
}

const vsp_pk *vsp_keypair_pk(const vsp_keypair *kp) { return kp ? kp->pk : nullptr; }

// which: 0 A_query, 1 B_query_g1, 2 B_query_g2, 3 H_query, 4 L_query, 5 gamma_ABC_g1, 6 alpha_g1, 7 beta_g1, 8 delta_g1,
//        9 beta_g2, 10 delta_g2, 11 gamma_g2, 12 gamma_g1
size_t vsp_keypair_device_bytes(const vsp_keypair *kp) {
    size_t t = 0;
    if (kp) for (int i = 0; i < 6; i++) t += vsp_bases_device_bytes(kp->q[i]);
    return t;
}
size_t vsp_keypair_count(const vsp_keypair *kp, int which) {
    if (!kp || which < 0 || which > 12) return 0;
    return which < 6 ? (kp->q[which] ? kp->q[which]->n : 0) : 1;       // a key loaded from a proving-key blob has no gamma_ABC_g1
}
int vsp_keypair_export(vsp_ctx *ctx, const vsp_keypair *kp, int which, uint64_t *out) {
    if (!ctx) return VSP_ERR_ARG;
    if (!kp || !out || which < 0 || which > 12) return set_error(ctx, VSP_ERR_ARG, "keypair_export: bad argument");
    if (which >= 6) {
        const uint64_t *src[] = {kp->alpha_g1, kp->beta_g1, kp->delta_g1, kp->beta_g2, kp->delta_g2, kp->gamma_g2, kp->gamma_g1};
        memcpy(out, src[which - 6], (which >= 9 && which <= 11) ? 192 : 96);
        return VSP_OK;
    }
    const vsp_bases *b = kp->q[which];
    if (!b || !b->n) return VSP_OK;
    VSP_HIP(hipSetDevice(ctx->device));
    size_t esz = b->group == 1 ? sizeof(G1Affine) : sizeof(G2Affine);
    DevBuf tmp;
    VSP_TRY(ensure(ctx, tmp, b->n * esz));
    unsigned blk = (unsigned)((b->n + 255) / 256);
    if (b->group == 1) hipLaunchKernelGGL(k_affine_from_mont_g1, dim3(blk), dim3(256), 0, ctx->stream, (const G1Affine *)b->d, (G1Affine *)tmp.p, b->n);
    else hipLaunchKernelGGL(k_affine_from_mont_g2, dim3(blk), dim3(256), 0, ctx->stream, (const G2Affine *)b->d, (G2Affine *)tmp.p, b->n);
    hipError_t e = hipMemcpyAsync(out, tmp.p, b->n * esz, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    hipFree(tmp.p);
    if (e != hipSuccess) return set_error(ctx, VSP_ERR_HIP, "keypair_export: copy failed");
    return VSP_OK;
}

}  // extern "C"
