// Groth16 prover orchestration (r1cs_gg_ppzksnark_prover::process), the sparse R1CS mat-vec that feeds
// r1cs_to_qap::witness_map, and the generator-side fixed-base batch exponentiation.
//
// Replaces, for the hot path, crypto3-zk r1cs_gg_ppzksnark/prover.hpp + reductions/r1cs_to_qap.hpp
// (absent submodule, /root/reference/.gitmodules:11-12; README.md:272-273 points at prover.hpp#L73),
// reached from bin/cli/include/nil/vote_saver/common.hpp:1132-1135; the generator's batch_exp is
// reached from common.hpp:916-917.
//
//   A = alpha + sum z_i A_i + r delta          (G1)
//   B = beta  + sum z_i B_i + s delta          (G2, and the same in G1 for C)
//   C = sum h_i H_i + sum_{aux} z_i L_i + s A + r B_g1 - r s delta  (+ r_enc P1 in SAVER mode)
#include <chrono>

#include "common.h"

namespace vsp {

// ---- sparse mat-vec: out[row] = sum_e coef[e] * z[col[e]],  z canonical, coef Montgomery -> canonical out
__global__ __launch_bounds__(256) void k_csr_matvec(const uint32_t *rp, const uint32_t *ci, const Fr *co, const Fr *z, size_t rows, Fr *out,
                                                    size_t z_stride = 0, size_t out_stride = 0) {
    z += (size_t)blockIdx.y * z_stride; out += (size_t)blockIdx.y * out_stride;      // grid.y: the witnesses of a batch
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows) return;
    Fr acc = Fr::zero();
    for (uint32_t e = rp[i]; e < rp[i + 1]; e++) acc = add(acc, mul(z[ci[e]], co[e]));
    out[i] = acc;
}
__global__ __launch_bounds__(256) void k_fr_to_mont(Fr *a, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) a[i] = to_mont(a[i]);
}

// canonical Fr check on the host (4 x uint64 little-endian limbs below r)
static bool fr_canonical(const uint64_t *k) {
    for (int i = 3; i >= 0; i--) { if (k[i] < FrP64::MOD[i]) return true; if (k[i] > FrP64::MOD[i]) return false; }
    return false;
}

}  // namespace vsp

using namespace vsp;

// ------------------------------------------------------------------------------------------------ C ABI: r1cs / pk / prove
extern "C" {

vsp_r1cs *vsp_r1cs_upload(vsp_ctx *ctx, size_t num_constraints, size_t num_inputs, size_t num_vars,
                          const uint32_t *row_ptr_a, const uint32_t *col_a, const uint64_t *coef_a,
                          const uint32_t *row_ptr_b, const uint32_t *col_b, const uint64_t *coef_b,
                          const uint32_t *row_ptr_c, const uint32_t *col_c, const uint64_t *coef_c) {
    if (!ctx) return nullptr;
    if (!row_ptr_a || !row_ptr_b || !row_ptr_c || num_inputs > num_vars) { set_error(ctx, VSP_ERR_ARG, "r1cs_upload: bad argument"); return nullptr; }
    hipSetDevice(ctx->device);
    vsp_r1cs *cs = new vsp_r1cs();
    cs->num_constraints = num_constraints; cs->num_inputs = num_inputs; cs->num_vars = num_vars;
    if (domain_init(ctx, &cs->dom, num_constraints + num_inputs + 1) != VSP_OK) { delete cs; return nullptr; }
    const uint32_t *rp[3] = {row_ptr_a, row_ptr_b, row_ptr_c}, *ci[3] = {col_a, col_b, col_c};
    const uint64_t *co[3] = {coef_a, coef_b, coef_c};
    for (int m = 0; m < 3; m++) {                             // every matrix before anything is allocated: row pointers monotone from 0, arrays present
        const size_t nnz = rp[m][num_constraints];
        bool mono = rp[m][0] == 0;
        for (size_t r = 0; r < num_constraints && mono; r++) mono = rp[m][r] <= rp[m][r + 1];
        if (!mono || (nnz && (!ci[m] || !co[m]))) {
            set_error(ctx, VSP_ERR_ARG, !mono ? "r1cs_upload: row pointers must start at 0 and not decrease" : "r1cs_upload: null column / coefficient array with nnz > 0");
            vsp_r1cs_free(ctx, cs); return nullptr;
        }
    }
    for (int m = 0; m < 3; m++) {
        size_t nnz = rp[m][num_constraints];
        for (size_t e = 0; e < nnz; e++) if (ci[m][e] > num_vars) { set_error(ctx, VSP_ERR_ARG, "r1cs_upload: column out of range"); vsp_r1cs_free(ctx, cs); return nullptr; }
        for (size_t e = 0; e < nnz; e++) if (!fr_canonical(co[m] + 4 * e)) { set_error(ctx, VSP_ERR_ARG, "r1cs_upload: a coefficient is not canonical (>= r)"); vsp_r1cs_free(ctx, cs); return nullptr; }
        bool ok = hipMalloc((void **)&cs->rp[m], (num_constraints + 1) * 4) == hipSuccess &&
                  hipMalloc((void **)&cs->ci[m], (nnz ? nnz : 1) * 4) == hipSuccess &&
                  hipMalloc(&cs->co[m], (nnz ? nnz : 1) * sizeof(Fr)) == hipSuccess;
        if (!ok) { set_error(ctx, VSP_ERR_NOMEM, "r1cs_upload: hipMalloc"); vsp_r1cs_free(ctx, cs); return nullptr; }
        if (hipMemcpyAsync(cs->rp[m], rp[m], (num_constraints + 1) * 4, hipMemcpyHostToDevice, ctx->stream) != hipSuccess ||
            hipMemcpyAsync(cs->ci[m], ci[m], nnz * 4, hipMemcpyHostToDevice, ctx->stream) != hipSuccess ||
            hipMemcpyAsync(cs->co[m], co[m], nnz * sizeof(Fr), hipMemcpyHostToDevice, ctx->stream) != hipSuccess) {
            set_error(ctx, VSP_ERR_HIP, "r1cs_upload: host-to-device copy failed"); vsp_r1cs_free(ctx, cs); return nullptr;
        }
        if (nnz) hipLaunchKernelGGL(k_fr_to_mont, dim3((unsigned)((nnz + 255) / 256)), dim3(256), 0, ctx->stream, (Fr *)cs->co[m], nnz);
        // column-major copy by a counting sort over the column index
        std::vector<uint32_t> colptr(num_vars + 2, 0), rows(nnz ? nnz : 1);
        std::vector<uint64_t> cot((nnz ? nnz : 1) * 4);
        for (size_t e = 0; e < nnz; e++) colptr[ci[m][e] + 1]++;
        for (size_t c = 0; c <= num_vars; c++) colptr[c + 1] += colptr[c];
        std::vector<uint32_t> cur(colptr.begin(), colptr.end() - 1);
        for (size_t r = 0; r < num_constraints; r++)
            for (uint32_t e = rp[m][r]; e < rp[m][r + 1]; e++) {
                uint32_t pos = cur[ci[m][e]]++;
                rows[pos] = (uint32_t)r; memcpy(&cot[4 * (size_t)pos], co[m] + 4 * (size_t)e, 32);
            }
        ok = hipMalloc((void **)&cs->cp[m], (num_vars + 2) * 4) == hipSuccess && hipMalloc((void **)&cs->ri[m], (nnz ? nnz : 1) * 4) == hipSuccess &&
             hipMalloc(&cs->cot[m], (nnz ? nnz : 1) * sizeof(Fr)) == hipSuccess;
        if (!ok) { set_error(ctx, VSP_ERR_NOMEM, "r1cs_upload: hipMalloc"); vsp_r1cs_free(ctx, cs); return nullptr; }
        if (hipMemcpy(cs->cp[m], colptr.data(), (num_vars + 2) * 4, hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(cs->ri[m], rows.data(), nnz * 4, hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(cs->cot[m], cot.data(), nnz * sizeof(Fr), hipMemcpyHostToDevice) != hipSuccess) {
            set_error(ctx, VSP_ERR_HIP, "r1cs_upload: host-to-device copy failed"); vsp_r1cs_free(ctx, cs); return nullptr;
        }
        if (nnz) hipLaunchKernelGGL(k_fr_to_mont, dim3((unsigned)((nnz + 255) / 256)), dim3(256), 0, ctx->stream, (Fr *)cs->cot[m], nnz);
    }
    if (hipStreamSynchronize(ctx->stream) != hipSuccess) { set_error(ctx, VSP_ERR_HIP, "r1cs_upload: sync"); vsp_r1cs_free(ctx, cs); return nullptr; }
    return cs;
}

void vsp_r1cs_free(vsp_ctx *ctx, vsp_r1cs *cs) {
    if (!cs) return;
    if (ctx) hipSetDevice(ctx->device);
    domain_release(&cs->dom);
    for (int m = 0; m < 3; m++) {
        if (cs->rp[m]) hipFree(cs->rp[m]); if (cs->ci[m]) hipFree(cs->ci[m]); if (cs->co[m]) hipFree(cs->co[m]);
        if (cs->cp[m]) hipFree(cs->cp[m]); if (cs->ri[m]) hipFree(cs->ri[m]); if (cs->cot[m]) hipFree(cs->cot[m]);
    }
    delete cs;
}

vsp_pk *vsp_pk_create(vsp_ctx *ctx, const uint64_t alpha_g1[12], const uint64_t beta_g1[12], const uint64_t beta_g2[24],
                      const uint64_t delta_g1[12], const uint64_t delta_g2[24],
                      const vsp_bases *A_query, const vsp_bases *B_query_g1, const vsp_bases *B_query_g2,
                      const vsp_bases *H_query, const vsp_bases *L_query) {
    if (!ctx) return nullptr;
    if (!alpha_g1 || !beta_g1 || !beta_g2 || !delta_g1 || !delta_g2 || !A_query || !B_query_g1 || !B_query_g2 || !H_query || !L_query ||
        A_query->group != 1 || B_query_g1->group != 1 || B_query_g2->group != 2 || H_query->group != 1 || L_query->group != 1) {
        set_error(ctx, VSP_ERR_ARG, "pk_create: bad argument"); return nullptr;
    }
    vsp_pk *pk = new vsp_pk();
    pk->alpha_g1 = host_load_g1(alpha_g1); pk->beta_g1 = host_load_g1(beta_g1); pk->delta_g1 = host_load_g1(delta_g1);
    pk->beta_g2 = host_load_g2(beta_g2); pk->delta_g2 = host_load_g2(delta_g2);
    pk->A = A_query; pk->B1 = B_query_g1; pk->B2 = B_query_g2; pk->H = H_query; pk->L = L_query;
    return pk;
}
void vsp_pk_free(vsp_ctx *, vsp_pk *pk) { delete pk; }
}  // extern "C"
namespace vsp {
// the fixed-base tables of delta (common.h vsp_pk): built by the first proof over the key, 32 x 255 additions per group (~5 ms G1, ~15 ms G2);
// option "prove_fixed_base" = 0 keeps the double-and-add multiplications of rounds 1-3
static bool delta_tables(vsp_ctx *ctx, const vsp_pk *pk) {
    { auto it = ctx->opts.find("prove_fixed_base"); if (it != ctx->opts.end() && it->second == 0) return false; }
    if (pk->tab_ready.load(std::memory_order_acquire)) return true;
    std::lock_guard<std::mutex> lock(pk->tab_mu);
    if (!pk->tab_ready.load(std::memory_order_relaxed)) {
        std::thread t2([&]() { xyzz_fixed_base_table(xyzz_from_affine(pk->delta_g2), pk->tab2); });
        xyzz_fixed_base_table(xyzz_from_affine(pk->delta_g1), pk->tab1);
        t2.join();
        pk->tab_ready.store(true, std::memory_order_release);
    }
    return true;
}
}  // namespace vsp
extern "C" {

// the witness as the caller hands it over: plain (num_vars x 4 canonical words), or packed (vsp_witness_pack)
struct WitnessSrc { const uint64_t *plain; const uint64_t *class_words; const uint32_t *word_offsets; const uint64_t *dense; size_t n_dense; };
static int prove_launch_impl(vsp_ctx *ctx, const vsp_r1cs *cs, const vsp_pk *pk, const WitnessSrc &w, const uint64_t r[4], const uint64_t s[4],
                             const uint64_t *saver_P1, const uint64_t *saver_r_enc);
static int prove_finish_impl(vsp_ctx *ctx, uint64_t A_out[12], uint64_t B_out[24], uint64_t C_out[12], uint8_t proof_out[192], const std::function<void()> *overlap);

}  // extern "C"
namespace vsp {
static void prove_cleanup(vsp_ctx *ctx, int rc) {
    if (rc != VSP_OK) {
        // an early return leaves multi-exponentiations in flight on their own streams, still reading the witness and H vectors:
        // wait for all of them before the caller (or the next call's workspace growth) can touch those buffers
        std::string keep = ctx->err;
        msm_drain_slots(ctx);
        ctx->err = keep;
    }
    for (unsigned k = 1; k <= 4; k++) (void)msm_slot_use_stream(ctx, k, nullptr);      // the slots go back to their own streams
    // the witness does not outlive the call in device memory
    if (ctx->pr_z.p) hipMemsetAsync(ctx->pr_z.p, 0, ctx->pr_z.cap, ctx->stream);
    ctx->prove.active = false;
}
static int prove_launch_checked(vsp_ctx *ctx, const vsp_r1cs *cs, const vsp_pk *pk, const WitnessSrc &w, const uint64_t r[4], const uint64_t s[4],
                                const uint64_t *saver_P1, const uint64_t *saver_r_enc) {
    if (!ctx) return VSP_ERR_ARG;
    if (!cs || !pk || !r || !s || (!w.plain && !(w.class_words && w.word_offsets && (w.dense || !w.n_dense)))) return set_error(ctx, VSP_ERR_ARG, "prove: null argument");
    if (ctx->prove.active || ctx->prove_batch.active) return set_error(ctx, VSP_ERR_ARG, "prove: a proof is already in flight on this context (finish it first)");
    if (!fr_canonical(r) || !fr_canonical(s) || (saver_r_enc && !fr_canonical(saver_r_enc)))
        return set_error(ctx, VSP_ERR_ARG, "prove: r, s and r_enc must be canonical (< r)");
    int rc = prove_launch_impl(ctx, cs, pk, w, r, s, saver_P1, saver_r_enc);
    if (rc != VSP_OK) prove_cleanup(ctx, rc);
    return rc;
}
// the prover with a hook: `overlap` runs on the host after every kernel is queued and before the first wait -- the window in which
// the host has nothing to do but wait for the GPU (vsp_saver_encrypt computes its ciphertext there)
int prove_with_overlap(vsp_ctx *ctx, const vsp_r1cs *cs, const vsp_pk *pk, const uint64_t *witness, const uint64_t r[4], const uint64_t s[4],
                       const uint64_t *saver_P1, const uint64_t *saver_r_enc, uint64_t A_out[12], uint64_t B_out[24], uint64_t C_out[12],
                       uint8_t proof_out[192], const std::function<void()> *overlap) {
    WitnessSrc w{witness, nullptr, nullptr, nullptr, 0};
    int rc = prove_launch_checked(ctx, cs, pk, w, r, s, saver_P1, saver_r_enc);
    if (rc != VSP_OK) return rc;
    rc = prove_finish_impl(ctx, A_out, B_out, C_out, proof_out, overlap);
    prove_cleanup(ctx, rc);
    return rc;
}
}  // namespace vsp
extern "C" {

int vsp_groth16_prove(vsp_ctx *ctx, const vsp_r1cs *cs, const vsp_pk *pk, const uint64_t *witness,
                      const uint64_t r[4], const uint64_t s[4], const uint64_t *saver_P1, const uint64_t *saver_r_enc,
                      uint64_t A_out[12], uint64_t B_out[24], uint64_t C_out[12], uint8_t proof_out[192]) {
    return prove_with_overlap(ctx, cs, pk, witness, r, s, saver_P1, saver_r_enc, A_out, B_out, C_out, proof_out, nullptr);
}
// The same call in two halves, so that ONE host thread keeps several proofs in flight (one per context, all over one resident key): launch
// queues every kernel of the proof and returns; finish does the host-side scalar multiplications, waits and assembles.
int vsp_groth16_prove_launch(vsp_ctx *ctx, const vsp_r1cs *cs, const vsp_pk *pk, const uint64_t *witness, const uint64_t r[4], const uint64_t s[4],
                             const uint64_t *saver_P1, const uint64_t *saver_r_enc) {
    WitnessSrc w{witness, nullptr, nullptr, nullptr, 0};
    return prove_launch_checked(ctx, cs, pk, w, r, s, saver_P1, saver_r_enc);
}
int vsp_groth16_prove_launch_packed(vsp_ctx *ctx, const vsp_r1cs *cs, const vsp_pk *pk, const uint64_t *class_words, const uint32_t *word_offsets,
                                    const uint64_t *dense, size_t n_dense, const uint64_t r[4], const uint64_t s[4],
                                    const uint64_t *saver_P1, const uint64_t *saver_r_enc) {
    if (!ctx) return VSP_ERR_ARG;
    if (!cs || !class_words || !word_offsets || (!dense && n_dense)) return set_error(ctx, VSP_ERR_ARG, "prove: null argument");
    // the class map indexes the dense values on the device: refuse a map whose offsets or count do not add up (a read outside the buffer)
    {
        const size_t nv = cs->num_vars, words = (nv + 31) / 32;
        size_t run = 0;
        for (size_t wd = 0; wd < words; wd++) {
            uint64_t cw = class_words[wd];
            if (wd == words - 1 && (nv & 31) && (cw >> (2 * (nv & 31)))) return set_error(ctx, VSP_ERR_ARG, "prove: packed witness has class bits beyond num_vars");
            if ((cw >> 1) & cw & 0x5555555555555555ull) return set_error(ctx, VSP_ERR_ARG, "prove: packed witness uses the reserved class 3");
            if (word_offsets[wd] != run) return set_error(ctx, VSP_ERR_ARG, "prove: packed witness offsets do not match its class map");
            run += (size_t)__builtin_popcountll((cw >> 1) & 0x5555555555555555ull);
        }
        if (run != n_dense) return set_error(ctx, VSP_ERR_ARG, "prove: packed witness dense count does not match its class map");
    }
    WitnessSrc w{nullptr, class_words, word_offsets, dense, n_dense};
    return prove_launch_checked(ctx, cs, pk, w, r, s, saver_P1, saver_r_enc);
}
int vsp_groth16_prove_finish(vsp_ctx *ctx, uint64_t A_out[12], uint64_t B_out[24], uint64_t C_out[12], uint8_t proof_out[192]) {
    if (!ctx) return VSP_ERR_ARG;
    if (!ctx->prove.active) return set_error(ctx, VSP_ERR_ARG, "prove_finish: no proof in flight on this context");
    int rc = prove_finish_impl(ctx, A_out, B_out, C_out, proof_out, nullptr);
    prove_cleanup(ctx, rc);
    return rc;
}
// Packed witness.  A Groth16 witness is mostly wires equal to 0 or 1, and its 32 bytes per wire cross PCIe in front of every proof with
// the GPU idle (0.8 ms at 2^20 constraints).  Packed form: two bits per wire (0 = zero, 1 = one, 2 = a dense value), 32 wires per 64-bit
// word; per word the index of its first dense value; the dense values (4 words each) in wire order.  A witness generator can emit this
// directly; vsp_witness_pack converts a plain witness (one pass over it on the host).  Sizes: words = (n + 31) / 32.
size_t vsp_witness_pack_words(size_t n) { return (n + 31) / 32; }
int vsp_witness_pack(const uint64_t *witness, size_t n, uint64_t *class_words, uint32_t *word_offsets, uint64_t *dense_out, size_t dense_capacity, size_t *n_dense_out) {
    if ((!witness && n) || !class_words || !word_offsets || !n_dense_out) return VSP_ERR_ARG;
    size_t nd = 0;
    const size_t words = (n + 31) / 32;
    for (size_t wd = 0; wd < words; wd++) {
        uint64_t cw = 0;
        word_offsets[wd] = (uint32_t)nd;
        const size_t lim = n - 32 * wd < 32 ? n - 32 * wd : 32;
        for (size_t j = 0; j < lim; j++) {
            const uint64_t *v = witness + 4 * (32 * wd + j);
            const bool small = (v[1] | v[2] | v[3]) == 0 && v[0] <= 1;
            if (small) cw |= (uint64_t)v[0] << (2 * j);
            else {
                cw |= (uint64_t)2 << (2 * j);
                if (dense_out) { if (nd >= dense_capacity) return VSP_ERR_ARG; memcpy(dense_out + 4 * nd, v, 32); }
                nd++;
            }
        }
        class_words[wd] = cw;
    }
    *n_dense_out = nd;
    return nd > 0xFFFFFFFFull ? VSP_ERR_UNSUPPORTED : VSP_OK;
}

// z[1 + i] of the packed witness: 0, 1 or the next dense value (canonical words either way)
__global__ __launch_bounds__(256) void k_witness_expand(const uint64_t *class_words, const uint32_t *word_offsets, const uint4 *dense, size_t n, uint4 *z) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t cw = class_words[i >> 5];
    const unsigned j = (unsigned)(i & 31), cls = (unsigned)(cw >> (2 * j)) & 3u;
    uint4 lo = make_uint4(cls == 1 ? 1u : 0u, 0, 0, 0), hi = make_uint4(0, 0, 0, 0);
    if (cls >= 2) {
        // dense values before this wire inside the word: the even bits of the class pairs above... class 2 = binary 10: count the high bits below j
        const uint64_t highs = (cw >> 1) & 0x5555555555555555ull & ((j ? ((uint64_t)1 << (2 * j)) : 1ull) - 1ull);
        const size_t k = (size_t)word_offsets[i >> 5] + (size_t)__popcll(highs);
        lo = dense[2 * k]; hi = dense[2 * k + 1];
    }
    z[2 * i] = lo; z[2 * i + 1] = hi;
}

static int prove_launch_impl(vsp_ctx *ctx, const vsp_r1cs *cs, const vsp_pk *pk, const WitnessSrc &wsrc, const uint64_t r[4], const uint64_t s[4],
                             const uint64_t *saver_P1, const uint64_t *saver_r_enc) {
    const size_t nv = cs->num_vars, ni = cs->num_inputs, nc = cs->num_constraints;
    const size_t m = cs->dom.m;
    if (pk->A->n != nv + 1 || pk->B1->n != nv + 1 || pk->B2->n != nv + 1 || pk->H->n + 1 != m || pk->L->n != nv - ni)
        return set_error(ctx, VSP_ERR_ARG, "prove: proving key does not match the constraint system");
    VSP_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    auto now = []() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double t_prev = now();
    auto lap = [&](const char *name) { double t = now(); ctx->stats[name] += t - t_prev; t_prev = t; };
    // z = (1, witness) canonical on device
    VSP_TRY(ensure(ctx, ctx->pr_z, (nv + 1) * sizeof(Fr)));
    VSP_TRY(ensure(ctx, ctx->pr_a, m * sizeof(Fr)));
    VSP_TRY(ensure(ctx, ctx->pr_b, m * sizeof(Fr)));
    VSP_TRY(ensure(ctx, ctx->pr_c, m * sizeof(Fr)));
    VSP_TRY(ensure(ctx, ctx->pr_h, m * sizeof(Fr)));
    Fr *dz = (Fr *)ctx->pr_z.p, *dA = (Fr *)ctx->pr_a.p, *dB = (Fr *)ctx->pr_b.p, *dC = (Fr *)ctx->pr_c.p, *dH = (Fr *)ctx->pr_h.p;
    const uint64_t one4[4] = {1, 0, 0, 0};
    VSP_HIP(hipMemcpyAsync(dz, one4, 32, hipMemcpyHostToDevice, st));
    if (wsrc.plain) VSP_HIP(hipMemcpyAsync(dz + 1, wsrc.plain, nv * 32, hipMemcpyHostToDevice, st));
    else {
        // packed witness: class map, per-word offsets and the dense values cross PCIe (a tenth of the plain witness for a 90 % boolean one); a kernel expands
        const size_t words = (nv + 31) / 32;
        VSP_TRY(ensure(ctx, ctx->pr_pack, words * 12 + wsrc.n_dense * 32 + 64));
        uint64_t *d_cw = (uint64_t *)ctx->pr_pack.p; uint32_t *d_off = (uint32_t *)(d_cw + words);
        uint4 *d_dense = (uint4 *)(((uintptr_t)(d_off + words) + 15) & ~(uintptr_t)15);
        VSP_HIP(hipMemcpyAsync(d_cw, wsrc.class_words, words * 8, hipMemcpyHostToDevice, st));
        VSP_HIP(hipMemcpyAsync(d_off, wsrc.word_offsets, words * 4, hipMemcpyHostToDevice, st));
        if (wsrc.n_dense) VSP_HIP(hipMemcpyAsync(d_dense, wsrc.dense, wsrc.n_dense * 32, hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(k_witness_expand, dim3((unsigned)((nv + 255) / 256)), dim3(256), 0, st, (const uint64_t *)d_cw, (const uint32_t *)d_off, (const uint4 *)d_dense, nv, (uint4 *)(dz + 1));
        VSP_LAUNCH_CHECK();
    }
    // evaluation vectors (witness_map part 1): A z, B z, C z, plus the rows "input_i * 0 = 0" in A
    VSP_HIP(hipMemsetAsync(dA, 0, m * sizeof(Fr), st));
    VSP_HIP(hipMemsetAsync(dB, 0, m * sizeof(Fr), st));
    VSP_HIP(hipMemsetAsync(dC, 0, m * sizeof(Fr), st));
    Fr *outs[3] = {dA, dB, dC};
    if (nc) for (int k = 0; k < 3; k++) {
        hipLaunchKernelGGL(k_csr_matvec, dim3((unsigned)((nc + 255) / 256)), dim3(256), 0, st, (const uint32_t *)cs->rp[k], (const uint32_t *)cs->ci[k],
                           (const Fr *)cs->co[k], (const Fr *)dz, nc, outs[k]);
        VSP_LAUNCH_CHECK();
    }
    VSP_HIP(hipMemcpyAsync(dA + nc, dz, (ni + 1) * sizeof(Fr), hipMemcpyDeviceToDevice, st));
    // Order of queueing: (1) the two scalar censuses, tiny, on the context's stream; (2) witness_map (7 NTTs) and the H
    // multi-exponentiation, the longest dependent chain, on the context's stream; (3) the four multi-exponentiations over the
    // witness -- independent of witness_map -- on the low-priority slots 1-4, filling the GPU around (2).  A_query, B_query(G1)
    // and B_query(G2) share one digit sort / bucket plan (same scalars) when their bases are precomputed alike.
    // The four witness multi-exponentiations run as TWO chains, on two streams the prover owns (lowest priority, like the work slots' own
    // streams, which stay free for callers that pipeline independent multi-exponentiations): A_query then B_query(G1) on one, B_query(G2)
    // -- which waits for A_query's digit sort, the plan the three share -- then L_query on the other.  Measured at 2^20 constraints on one
    // box: 6.5 ms per proof against 7.2-7.6 ms with one stream per multi-exponentiation, the same 170 / 190 proofs/s with two / three
    // contexts; the other pairings lose (A,B2 | B1,L: 7.8 ms; all four on one stream: 7.3 ms).  Fewer concurrent witness chains leave the
    // transforms and the H accumulation -- the critical chain -- alone for longer.  Option "prove_witness_streams" = 0: the
    // slots' own streams.
    long wstreams = 1; { auto it = ctx->opts.find("prove_witness_streams"); if (it != ctx->opts.end()) wstreams = it->second; }
    if (wstreams) {
        for (int k = 0; k < 2; k++) if (!ctx->prove_streams[k]) VSP_TRY(msm_make_slot_stream(ctx, &ctx->prove_streams[k]));
        VSP_TRY(msm_slot_use_stream(ctx, 1, ctx->prove_streams[0])); VSP_TRY(msm_slot_use_stream(ctx, 2, ctx->prove_streams[0]));
        VSP_TRY(msm_slot_use_stream(ctx, 3, ctx->prove_streams[1])); VSP_TRY(msm_slot_use_stream(ctx, 4, ctx->prove_streams[1]));
    }
    VSP_TRY(msm_slot_census(ctx, 1, dz, nv + 1));
    VSP_TRY(msm_slot_census(ctx, 4, dz + ni + 1, nv - ni));
    VSP_HIP(hipEventRecord(ctx->ev_aux, st));          // z resident and censuses queued
    // option "prove_h_first" (default 1): queue witness_map + H before the witness multi-exponentiations, or after (0)
    long h_first = 1; { auto it = ctx->opts.find("prove_h_first"); if (it != ctx->opts.end()) h_first = it->second; }
    {
        hipStream_t s1, s2, s3, s4;
        VSP_TRY(msm_slot_stream(ctx, 1, &s1)); VSP_TRY(msm_slot_stream(ctx, 2, &s2));
        VSP_TRY(msm_slot_stream(ctx, 3, &s3)); VSP_TRY(msm_slot_stream(ctx, 4, &s4));
        VSP_HIP(hipStreamWaitEvent(s1, ctx->ev_aux, 0)); VSP_HIP(hipStreamWaitEvent(s2, ctx->ev_aux, 0));
        VSP_HIP(hipStreamWaitEvent(s3, ctx->ev_aux, 0)); VSP_HIP(hipStreamWaitEvent(s4, ctx->ev_aux, 0));
    }
    // option "prove_plan_first" (default 0): queue the digit sorts and bucket plans of the two witness vectors BEFORE witness_map.
    // Their counting sort needs 128 KiB of LDS per workgroup; queued after the transforms it finds every CU's LDS taken by NTT tiles and
    // then every wave slot taken by the H accumulation, and waits ~2 ms (kernel timeline).  Queued first it runs while the
    // GPU is idle -- measured: no gain (9.10 against 9.02 ms): the proof is bound by the sum of its kernels, not by that wait.
    long plan_first = 0; { auto it = ctx->opts.find("prove_plan_first"); if (it != ctx->opts.end()) plan_first = it->second; }
    int planA = -1, planL = -1;
    if (h_first && plan_first) {
        VSP_TRY(launch_on_bases(ctx, 1, pk->A, 0, nv + 1, dz, VSP_MSM_PLAN_ONLY)); planA = 1;
        VSP_TRY(launch_on_bases(ctx, 4, pk->L, 0, nv - ni, dz + ni + 1, VSP_MSM_PLAN_ONLY)); planL = 4;
    }
    if (h_first) {
        VSP_TRY(witness_map_device(ctx, dA, dB, dC, &cs->dom, dH));
        VSP_TRY(launch_on_bases(ctx, 0, pk->H, 0, m - 1, dH, VSP_MSM_DENSE));   // H coefficients are dense
    }
    VSP_TRY(launch_on_bases(ctx, 1, pk->A, 0, nv + 1, dz, planA));
    VSP_TRY(launch_on_bases(ctx, 3, pk->B2, 0, nv + 1, dz, pk->B2->pre_c == pk->A->pre_c ? 1 : -1));
    VSP_TRY(launch_on_bases(ctx, 2, pk->B1, 0, nv + 1, dz, pk->B1->pre_c == pk->A->pre_c ? 1 : -1));
    VSP_TRY(launch_on_bases(ctx, 4, pk->L, 0, nv - ni, dz + ni + 1, planL));
    if (!h_first) {
        VSP_TRY(witness_map_device(ctx, dA, dB, dC, &cs->dom, dH));
        VSP_TRY(launch_on_bases(ctx, 0, pk->H, 0, m - 1, dH, VSP_MSM_DENSE));
    }
    lap("prove_launch_ms");
    // what the second half needs: the key, the randomness, the SAVER term
    ctx->prove.active = true; ctx->prove.pk = pk;
    memcpy(ctx->prove.r, r, 32); memcpy(ctx->prove.s, s, 32);
    ctx->prove.has_saver = saver_P1 && saver_r_enc;
    if (ctx->prove.has_saver) { memcpy(ctx->prove.P1, saver_P1, 96); memcpy(ctx->prove.r_enc, saver_r_enc, 32); }
    return VSP_OK;
}

static int prove_finish_impl(vsp_ctx *ctx, uint64_t A_out[12], uint64_t B_out[24], uint64_t C_out[12], uint8_t proof_out[192], const std::function<void()> *overlap) {
    const vsp_pk *pk = ctx->prove.pk;
    const uint64_t *r = ctx->prove.r, *s = ctx->prove.s;
    const uint64_t *saver_P1 = ctx->prove.has_saver ? ctx->prove.P1 : nullptr, *saver_r_enc = ctx->prove.has_saver ? ctx->prove.r_enc : nullptr;
    VSP_HIP(hipSetDevice(ctx->device));
    auto now = []() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double t_prev = now();
    auto lap = [&](const char *name) { double t = now(); ctx->stats[name] += t - t_prev; t_prev = t; };
    XYZZ<HFp> eA = XYZZ<HFp>::inf(), eB1 = XYZZ<HFp>::inf(), eH = XYZZ<HFp>::inf(), eL = XYZZ<HFp>::inf(); XYZZ<HFp2> eB2 = XYZZ<HFp2>::inf();
    // host work that needs no MSM result, done while the GPU runs: the four multiples of delta on four host threads (round 4: at the real
    // circuit's size the HOST was the critical path of a proof -- 0.85 ms of scalar multiplications here and ~1 ms of Horner chains in the
    // five finishes below, one after the other, against ~1.5 ms of GPU work; tools/prove_phases.py)
    XYZZ<HFp> dj = xyzz_from_affine(pk->delta_g1);
    XYZZ<HFp2> dj2 = xyzz_from_affine(pk->delta_g2);
    HFr rr = host_load_canon<HFr>(r), ss = host_load_canon<HFr>(s);
    uint64_t rs4[4]; host_store_canon(rs4, mul(rr, ss));
    XYZZ<HFp> r_delta, s_delta, neg_rs_delta, saver = XYZZ<HFp>::inf();
    XYZZ<HFp2> s_delta2;
    long hthreads = 1; { auto it = ctx->opts.find("prove_host_threads"); if (it != ctx->opts.end()) hthreads = it->second; }
    const unsigned T = hthreads ? 8u : 1u;
    const bool fixed = delta_tables(ctx, pk);                 // (round 4: at most 32 additions per multiple of delta instead of 255 doublings + ~127 additions)
    host_parallel_for(5, [&](size_t j) {
        if (j == 0) s_delta2 = fixed ? xyzz_mul_fixed(pk->tab2, s) : xyzz_mul_scalar(dj2, s, 255);                 // the G2 one is the longest: first
        else if (j == 1) r_delta = fixed ? xyzz_mul_fixed(pk->tab1, r) : xyzz_mul_scalar(dj, r, 255);
        else if (j == 2) s_delta = fixed ? xyzz_mul_fixed(pk->tab1, s) : xyzz_mul_scalar(dj, s, 255);
        else if (j == 3) neg_rs_delta = xyzz_neg(fixed ? xyzz_mul_fixed(pk->tab1, rs4) : xyzz_mul_scalar(dj, rs4, 255));
        else if (saver_P1 && saver_r_enc) saver = xyzz_mul_scalar_w4(xyzz_from_affine(host_load_g1(saver_P1)), saver_r_enc);
    }, T);
    if (overlap && *overlap) (*overlap)();
    lap("prove_host_overlap_ms");
    // The witness multi-exponentiations finish long before the H chain (witness_map, then the dense H query).  Each finish is a wait (this
    // thread: it touches the context) and a fold of the window results -- a Horner chain of a few hundred host group operations, 0.2 ms in G1,
    // 0.6 ms in G2 -- which runs on a thread of its own while this one waits for the next slot; s * A and r * B1, the two 255-bit scalar
    // multiplications of the assembly, follow their folds on the same threads.  Option "prove_host_threads" = 0: everything on this thread.
    XYZZ<HFp> gA, gB1, s_gA, r_gB1;
    XYZZ<HFp2> gB2;
    std::vector<std::thread> workers;
    auto run = [&](std::function<void()> f) { if (T > 1) workers.emplace_back(std::move(f)); else f(); };
    auto join_all = [&]() { for (auto &w : workers) w.join(); workers.clear(); };
    int rc = VSP_OK; bool empty = false;
    if ((rc = msm_g1_finish_wait(ctx, 1, &empty)) == VSP_OK) {
        const bool e = empty;
        run([&, e]() { if (!e) msm_g1_fold(ctx, 1, &eA); gA = eA; xyzz_madd(gA, pk->alpha_g1); xyzz_add(gA, r_delta); s_gA = xyzz_mul_scalar_w4(gA, s); });
    }
    if (rc == VSP_OK && (rc = msm_g1_finish_wait(ctx, 2, &empty)) == VSP_OK) {
        const bool e = empty;
        run([&, e]() { if (!e) msm_g1_fold(ctx, 2, &eB1); gB1 = eB1; xyzz_madd(gB1, pk->beta_g1); xyzz_add(gB1, s_delta); r_gB1 = xyzz_mul_scalar_w4(gB1, r); });
    }
    if (rc == VSP_OK && (rc = msm_g1_finish_wait(ctx, 4, &empty)) == VSP_OK) {
        const bool e = empty;
        run([&, e]() { if (!e) msm_g1_fold(ctx, 4, &eL); });
    }
    if (rc == VSP_OK && (rc = msm_g2_finish_wait(ctx, 3, &empty)) == VSP_OK) {
        const bool e = empty;
        run([&, e]() { if (!e) msm_g2_fold(ctx, 3, &eB2); gB2 = eB2; xyzz_madd(gB2, pk->beta_g2); xyzz_add(gB2, s_delta2); });
    }
    if (rc == VSP_OK && (rc = msm_g1_finish_wait(ctx, 0, &empty)) == VSP_OK && !empty) msm_g1_fold(ctx, 0, &eH);
    join_all();
    if (rc != VSP_OK) return rc;
    lap("prove_wait_ms");
    // assembly: a handful of group operations
    XYZZ<HFp> gC = eH; xyzz_add(gC, eL);
    xyzz_add(gC, s_gA);
    xyzz_add(gC, r_gB1);
    xyzz_add(gC, neg_rs_delta);
    xyzz_add(gC, saver);
    Affine<HFp> a = xyzz_to_affine(gA), c = xyzz_to_affine(gC);
    Affine<HFp2> b = xyzz_to_affine(gB2);
    uint64_t A12[12], B24[24], C12[12];
    host_store_g1(A12, a); host_store_g2(B24, b); host_store_g1(C12, c);
    if (A_out) memcpy(A_out, A12, sizeof A12);
    if (B_out) memcpy(B_out, B24, sizeof B24);
    if (C_out) memcpy(C_out, C12, sizeof C12);
    if (proof_out) { vsp_g1_compress(A12, proof_out); vsp_g2_compress(B24, proof_out + 48); vsp_g1_compress(C12, proof_out + 144); }
    lap("prove_assembly_ms");
    ctx->stats["prove_calls"] += 1;
    return VSP_OK;
}


// ---- a BATCH of proofs over one key (round 4).  Proofs of the real circuit's size (2^15..2^16 constraints) are bound by
// the latency of their dependent chains -- a few hundred small launches, each a fraction of the GPU -- not by work: 2.1 ms per proof, 465 / s
// from one context, ~900 / s from twelve.  K witnesses proved TOGETHER run the same number of launches K times as wide: one matvec, one
// witness_map over 3 K transforms, and each of the five multi-exponentiations once over K scalar vectors (MsmGeom.K: separate bucket sets
// per witness, the same base rows).  Every proof is byte-identical to vsp_groth16_prove's for the same (witness, r, s).
// PLAIN key (vsp_groth16_generate with precompute = 0, or vsp_pk_create over plain bases): a batch has no use for tables of window multiples.
static void prove_batch_cleanup(vsp_ctx *ctx, int rc) {
    if (rc != VSP_OK) { std::string keep = ctx->err; msm_drain_slots(ctx); ctx->err = keep; }
    for (unsigned k = 1; k <= 4; k++) (void)msm_slot_use_stream(ctx, k, nullptr);
    if (ctx->pr_bz.p) hipMemsetAsync(ctx->pr_bz.p, 0, ctx->pr_bz.cap, ctx->stream);      // the witnesses do not outlive the call in device memory
}
static int prove_batch_launch_impl(vsp_ctx *ctx, const vsp_r1cs *cs, const vsp_pk *pk, const uint64_t *witnesses, size_t K) {
    const size_t nv = cs->num_vars, ni = cs->num_inputs, nc = cs->num_constraints, m = cs->dom.m, zs = nv + 1;
    if (pk->A->n != nv + 1 || pk->B1->n != nv + 1 || pk->B2->n != nv + 1 || pk->H->n + 1 != m || pk->L->n != nv - ni)
        return set_error(ctx, VSP_ERR_ARG, "prove: proving key does not match the constraint system");
    // (a key with tables of window multiples: one bucket set per witness and query, windows of 16 bits -- worth it where the tables are small, i.e. at
    // the real circuit's size; option "msm_batch_tables" = 0 refuses such keys as rounds before the end of round 4 did)
    VSP_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    auto now = []() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double t_prev = now();
    auto lap = [&](const char *name) { double t = now(); ctx->stats[name] += t - t_prev; t_prev = t; };      // vsp_get_stat "prove_batch_*_ms": where a batch's time goes on the host
    VSP_TRY(ensure(ctx, ctx->pr_bz, K * zs * sizeof(Fr)));
    VSP_TRY(ensure(ctx, ctx->pr_babc, K * 3 * m * sizeof(Fr)));
    VSP_TRY(ensure(ctx, ctx->pr_bh, K * m * sizeof(Fr)));
    Fr *dz = (Fr *)ctx->pr_bz.p, *abc = (Fr *)ctx->pr_babc.p, *dH = (Fr *)ctx->pr_bh.p;
    // z_k = (1, witness_k), canonical: the K ones from a small host array, the witnesses by one strided copy
    std::vector<uint64_t> ones(K * 4, 0); for (size_t k = 0; k < K; k++) ones[4 * k] = 1;
    VSP_HIP(hipMemcpy2DAsync(dz, zs * 32, ones.data(), 32, 32, K, hipMemcpyHostToDevice, st));
    VSP_HIP(hipMemcpy2DAsync(dz + 1, zs * 32, witnesses, nv * 32, nv * 32, K, hipMemcpyHostToDevice, st));
    VSP_HIP(hipMemsetAsync(abc, 0, K * 3 * m * sizeof(Fr), st));
    if (nc) for (int j = 0; j < 3; j++) {
        hipLaunchKernelGGL(k_csr_matvec, dim3((unsigned)((nc + 255) / 256), (unsigned)K), dim3(256), 0, st, (const uint32_t *)cs->rp[j], (const uint32_t *)cs->ci[j],
                           (const Fr *)cs->co[j], (const Fr *)dz, nc, abc + (size_t)j * m, zs, 3 * m);
        VSP_LAUNCH_CHECK();
    }
    VSP_HIP(hipMemcpy2DAsync(abc + nc, 3 * m * sizeof(Fr), dz, zs * sizeof(Fr), (ni + 1) * sizeof(Fr), K, hipMemcpyDeviceToDevice, st));      // the rows "input_i * 0 = 0" of A
    VSP_HIP(hipStreamSynchronize(st));                       // `ones` goes out of scope; the copies above are queued from pageable memory anyway
    // the four witness multi-exponentiations as two chains on the prover's two low-priority streams, the H chain on the context's (prove_launch_impl)
    for (int k = 0; k < 2; k++) if (!ctx->prove_streams[k]) VSP_TRY(msm_make_slot_stream(ctx, &ctx->prove_streams[k]));
    VSP_TRY(msm_slot_use_stream(ctx, 1, ctx->prove_streams[0])); VSP_TRY(msm_slot_use_stream(ctx, 2, ctx->prove_streams[0]));
    VSP_TRY(msm_slot_use_stream(ctx, 3, ctx->prove_streams[1])); VSP_TRY(msm_slot_use_stream(ctx, 4, ctx->prove_streams[1]));
    VSP_TRY(witness_map_device_batch(ctx, abc, (unsigned)K, &cs->dom, dH));
    VSP_TRY(launch_on_bases_batch(ctx, 0, pk->H, 0, m - 1, dH, (unsigned)K, m, true));           // H coefficients are dense
    VSP_TRY(launch_on_bases_batch(ctx, 1, pk->A, 0, nv + 1, dz, (unsigned)K, zs, false));
    // A, B1 and B2 multiply by the same K witness vectors: one digit sort and bucket plan (A's) serves the three (option "prove_batch_share_plan")
    long share = 1; { auto it = ctx->opts.find("prove_batch_share_plan"); if (it != ctx->opts.end()) share = it->second; }
    const bool same_shape = pk->A->glv == pk->B1->glv && pk->A->glv == pk->B2->glv && (pk->A->d28 != nullptr) == (pk->B1->d28 != nullptr) && (pk->A->d28 != nullptr) == (pk->B2->d28 != nullptr) &&
                            pk->A->pre_c == pk->B1->pre_c && pk->A->pre_c == pk->B2->pre_c && pk->A->n == pk->B1->n && pk->A->n == pk->B2->n;
    const int from_a = share && same_shape && nv + 1 > 0 ? 1 : -1;
    VSP_TRY(launch_on_bases_batch(ctx, 3, pk->B2, 0, nv + 1, dz, (unsigned)K, zs, false, from_a));
    VSP_TRY(launch_on_bases_batch(ctx, 2, pk->B1, 0, nv + 1, dz, (unsigned)K, zs, false, from_a));
    VSP_TRY(launch_on_bases_batch(ctx, 4, pk->L, 0, nv - ni, dz + ni + 1, (unsigned)K, zs, false));
    lap("prove_batch_launch_ms");
    return VSP_OK;
}
static int prove_batch_finish_impl(vsp_ctx *ctx, const vsp_pk *pk, size_t K, const uint64_t *r, const uint64_t *s,
                                   uint64_t *A_out, uint64_t *B_out, uint64_t *C_out, uint8_t *proofs_out) {
    auto now = []() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double t_prev = now();
    auto lap = [&](const char *name) { double t = now(); ctx->stats[name] += t - t_prev; t_prev = t; };
    VSP_HIP(hipSetDevice(ctx->device));
    // host work that needs no result: the delta multiples of every proof
    XYZZ<HFp> dj = xyzz_from_affine(pk->delta_g1);
    XYZZ<HFp2> dj2 = xyzz_from_affine(pk->delta_g2);
    std::vector<XYZZ<HFp>> r_delta(K), s_delta(K), neg_rs_delta(K);
    std::vector<XYZZ<HFp2>> s_delta2(K);
    const bool fixed = delta_tables(ctx, pk);
    host_parallel_for(K, [&](size_t k) {
        const uint64_t *rk = r + 4 * k, *sk = s + 4 * k;
        uint64_t rs4[4]; host_store_canon(rs4, mul(host_load_canon<HFr>(rk), host_load_canon<HFr>(sk)));
        if (fixed) {
            r_delta[k] = xyzz_mul_fixed(pk->tab1, rk); s_delta[k] = xyzz_mul_fixed(pk->tab1, sk);
            neg_rs_delta[k] = xyzz_neg(xyzz_mul_fixed(pk->tab1, rs4));
            s_delta2[k] = xyzz_mul_fixed(pk->tab2, sk);
        } else {
            r_delta[k] = xyzz_mul_scalar(dj, rk, 255); s_delta[k] = xyzz_mul_scalar(dj, sk, 255);
            neg_rs_delta[k] = xyzz_neg(xyzz_mul_scalar(dj, rs4, 255));
            s_delta2[k] = xyzz_mul_scalar(dj2, sk, 255);
        }
    });
    lap("prove_batch_delta_ms");
    std::vector<XYZZ<HFp>> eA(K), eB1(K), eH(K), eL(K);
    std::vector<XYZZ<HFp2>> eB2(K);
    VSP_TRY(msm_g1_finish_batch(ctx, 1, eA.data(), (unsigned)K));
    VSP_TRY(msm_g1_finish_batch(ctx, 2, eB1.data(), (unsigned)K));
    VSP_TRY(msm_g1_finish_batch(ctx, 4, eL.data(), (unsigned)K));
    VSP_TRY(msm_g2_finish_batch(ctx, 3, eB2.data(), (unsigned)K));
    lap("prove_batch_witness_finishes_ms");
    // s * A and r * B1 of every proof inside the wait for the H chain (prove_finish_impl, prove_early_assembly)
    std::vector<XYZZ<HFp>> gA(K), s_gA(K), r_gB1(K);
    host_parallel_for(K, [&](size_t k) {
        gA[k] = eA[k]; xyzz_madd(gA[k], pk->alpha_g1); xyzz_add(gA[k], r_delta[k]);
        s_gA[k] = xyzz_mul_scalar_w4(gA[k], s + 4 * k);
        XYZZ<HFp> gB1 = eB1[k]; xyzz_madd(gB1, pk->beta_g1); xyzz_add(gB1, s_delta[k]);
        r_gB1[k] = xyzz_mul_scalar_w4(gB1, r + 4 * k);
    });
    lap("prove_batch_sA_rB1_ms");
    VSP_TRY(msm_g1_finish_batch(ctx, 0, eH.data(), (unsigned)K));
    lap("prove_batch_h_finish_ms");
    host_parallel_for(K, [&](size_t k) {
        XYZZ<HFp2> gB2 = eB2[k]; xyzz_madd(gB2, pk->beta_g2); xyzz_add(gB2, s_delta2[k]);
        XYZZ<HFp> gC = eH[k]; xyzz_add(gC, eL[k]);
        xyzz_add(gC, s_gA[k]); xyzz_add(gC, r_gB1[k]); xyzz_add(gC, neg_rs_delta[k]);
        Affine<HFp> a = xyzz_to_affine(gA[k]), c = xyzz_to_affine(gC);
        Affine<HFp2> b = xyzz_to_affine(gB2);
        uint64_t A12[12], B24[24], C12[12];
        host_store_g1(A12, a); host_store_g2(B24, b); host_store_g1(C12, c);
        if (A_out) memcpy(A_out + 12 * k, A12, sizeof A12);
        if (B_out) memcpy(B_out + 24 * k, B24, sizeof B24);
        if (C_out) memcpy(C_out + 12 * k, C12, sizeof C12);
        if (proofs_out) { vsp_g1_compress(A12, proofs_out + 192 * k); vsp_g2_compress(B24, proofs_out + 192 * k + 48); vsp_g1_compress(C12, proofs_out + 192 * k + 144); }
    });
    lap("prove_batch_assembly_ms");
    ctx->stats["prove_calls"] += (double)K;
    ctx->stats["prove_batches"] += 1;
    return VSP_OK;
}
static int prove_batch_launch_checked(vsp_ctx *ctx, const vsp_r1cs *cs, const vsp_pk *pk, const uint64_t *witnesses, size_t count, const uint64_t *r, const uint64_t *s) {
    if (!ctx) return VSP_ERR_ARG;
    if (!cs || !pk || !witnesses || !r || !s || count < 1 || count > 64) return set_error(ctx, VSP_ERR_ARG, "prove_batch: null argument or a batch outside 1..64");
    if (ctx->prove.active || ctx->prove_batch.active) return set_error(ctx, VSP_ERR_ARG, "prove: a proof is already in flight on this context (finish it first)");
    for (size_t k = 0; k < count; k++)
        if (!fr_canonical(r + 4 * k) || !fr_canonical(s + 4 * k)) return set_error(ctx, VSP_ERR_ARG, "prove: r and s must be canonical (< r)");
    int rc = prove_batch_launch_impl(ctx, cs, pk, witnesses, count);
    if (rc != VSP_OK) { prove_batch_cleanup(ctx, rc); return rc; }
    ctx->prove_batch.active = true; ctx->prove_batch.pk = pk; ctx->prove_batch.count = count;
    ctx->prove_batch.r.assign(r, r + 4 * count); ctx->prove_batch.s.assign(s, s + 4 * count);
    return VSP_OK;
}
int vsp_groth16_prove_batch_launch(vsp_ctx *ctx, const vsp_r1cs *cs, const vsp_pk *pk, const uint64_t *witnesses, size_t count, const uint64_t *r, const uint64_t *s) {
    return prove_batch_launch_checked(ctx, cs, pk, witnesses, count, r, s);
}
int vsp_groth16_prove_batch_finish(vsp_ctx *ctx, uint64_t *A_out, uint64_t *B_out, uint64_t *C_out, uint8_t *proofs_out) {
    if (!ctx) return VSP_ERR_ARG;
    if (!ctx->prove_batch.active) return set_error(ctx, VSP_ERR_ARG, "prove_batch_finish: no batch in flight on this context");
    int rc = prove_batch_finish_impl(ctx, ctx->prove_batch.pk, ctx->prove_batch.count, ctx->prove_batch.r.data(), ctx->prove_batch.s.data(), A_out, B_out, C_out, proofs_out);
    ctx->prove_batch.active = false;
    prove_batch_cleanup(ctx, rc);
    return rc;
}
int vsp_groth16_prove_batch(vsp_ctx *ctx, const vsp_r1cs *cs, const vsp_pk *pk, const uint64_t *witnesses, size_t count, const uint64_t *r, const uint64_t *s,
                            uint64_t *A_out, uint64_t *B_out, uint64_t *C_out, uint8_t *proofs_out) {
    int rc = prove_batch_launch_checked(ctx, cs, pk, witnesses, count, r, s);
    if (rc != VSP_OK) return rc;
    return vsp_groth16_prove_batch_finish(ctx, A_out, B_out, C_out, proofs_out);
}

}  // extern "C"
