// Wire formats of the artefacts that cross the prover's boundary: the big-endian blobs of the reference's
// marshaling_policy (bin/cli/include/nil/vote_saver/common.hpp:168-203: option::big_endian; proof / primary input / ciphertext
// written at :462-485, the proving key read INSIDE the timed vote phase at :749-754 and :1002-1004).
//
// The marshalling code itself lives in absent submodules (crypto3-marshalling-*, /root/reference/.gitmodules:35-44), so only what the
// reference's own files show is pinned; everything else is a stated guess kept in ONE place (this file):
//   pinned    * proof = A | B | C, ZCash-compressed, 48 + 96 + 48 bytes                      bin/cli/src/data.bin[0:192)
//             * scalar vectors = 8-byte count, then 32-byte field elements                     protocol_exec.ipynb cell 0 / 20 (fr_size = 32,
//               (eid / sn / rt blobs: "eid[std_size_t_size * 2:]", sizes eid_len * fr_size)    std_size_t_size = 8) -- big-endian per common.hpp:180
//             * extended verification key starts: 4 bytes, then a GT element as 12 x 48-byte   data.bin[192:196), [196:772): satisfies g^r = 1
//               LITTLE-endian Fp (tower order c0.c0.c0, c0.c0.c1, c0.c1.c0, ...), then two      under the test-side pairing; [772:868), [868:964)
//               compressed G2 points and one compressed G1 point                               decode as G2, [964:1012) as G1, all in the subgroup
//   guesses   * the roles gamma_g2, delta_g2, delta_g1 of those three points; the tail of the key (gamma_ABC_g1 as a counted vector of
//               compressed G1, then gamma_g1): data.bin is zero from byte 1012 on
//             * ciphertext = counted vector of compressed G1 (by analogy with the scalar vectors)
//             * "fast" proving key = UNCOMPRESSED big-endian affine points (ZCash uncompressed form: no square root on load, which is
//               what makes it fast), fixed elements alpha_g1 beta_g1 beta_g2 delta_g1 delta_g2, then A_query, B_query (pairs G2 | G1),
//               H_query, L_query as counted vectors.  Upstream's key also embeds the constraint system; here it travels separately
//               (vsp_r1cs_upload), because circuit construction is out of scope.
// The proving-key loader is the part with a cost: the reference parses ~0.6 GB per vote at 2^20 constraints.  vsp_pk_from_blob copies the
// raw bytes to the GPU once and converts there (byte order, infinity flags, curve check, Montgomery form, the 28-bit-limb table).
#include "common.h"

namespace vsp {

// 48 big-endian bytes (12 words) -> 12 little-endian 32-bit limbs; the three flag bits of the first byte are cleared
__device__ __forceinline__ Fp fp_from_be(const uint32_t *w, bool first) {
    Fp r;
#pragma unroll
    for (int j = 0; j < 12; j++) r.l[j] = __builtin_bswap32(w[11 - j]);
    if (first) r.l[11] &= 0x1FFFFFFFu;
    return r;
}
// ZCash uncompressed records -> canonical affine points (infinity flag -> all zero).  flag bit 2: a record is malformed -- it claims to be
// compressed (0x80), carries the sign bit that only the compressed form has (0x20), or is an infinity record (0x40) with a non-zero payload
__device__ __forceinline__ bool record_rest_zero(const uint32_t *w, unsigned words) {
    uint32_t acc = w[0] & ~0xFFu;                              // everything but the flag byte
    for (unsigned j = 1; j < words; j++) acc |= w[j];
    return acc == 0 && (w[0] & 0x3Fu) == 0;
}
__global__ __launch_bounds__(256) void k_g1_from_be(const uint8_t *src, size_t stride, size_t n, G1Affine *out, uint32_t *flag) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t *w = (const uint32_t *)(src + i * stride);
    const uint32_t head = w[0] & 0xFFu;                       // first byte of the record
    G1Affine p;
    if (head & 0xA0u) atomicOr(flag, 4u);
    if (head & 0x40u) { p.x = Fp::zero(); p.y = Fp::zero(); if (!record_rest_zero(w, 24)) atomicOr(flag, 4u); }
    else { p.x = fp_from_be(w, true); p.y = fp_from_be(w + 12, false); }
    out[i] = p;
}
__global__ __launch_bounds__(256) void k_g2_from_be(const uint8_t *src, size_t stride, size_t n, G2Affine *out, uint32_t *flag) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t *w = (const uint32_t *)(src + i * stride);
    const uint32_t head = w[0] & 0xFFu;
    G2Affine p;
    if (head & 0xA0u) atomicOr(flag, 4u);
    if (head & 0x40u) { p.x.c0 = p.x.c1 = p.y.c0 = p.y.c1 = Fp::zero(); if (!record_rest_zero(w, 48)) atomicOr(flag, 4u); }
    else { p.x.c1 = fp_from_be(w, true); p.x.c0 = fp_from_be(w + 12, false); p.y.c1 = fp_from_be(w + 24, false); p.y.c0 = fp_from_be(w + 36, false); }
    out[i] = p;
}

static void put_be64(uint8_t *o, uint64_t v) { for (int i = 0; i < 8; i++) o[i] = (uint8_t)(v >> (56 - 8 * i)); }
static uint64_t get_be64(const uint8_t *p) { uint64_t v = 0; for (int i = 0; i < 8; i++) v = (v << 8) | p[i]; return v; }
static void be_from_limbs(uint8_t *o, const uint64_t *l, int nl) { for (int i = 0; i < nl; i++) for (int b = 0; b < 8; b++) o[nl * 8 - 1 - (i * 8 + b)] = (uint8_t)(l[i] >> (8 * b)); }
static void limbs_from_be(uint64_t *l, const uint8_t *p, int nl) { for (int i = 0; i < nl; i++) { uint64_t v = 0; for (int b = 0; b < 8; b++) v |= (uint64_t)p[nl * 8 - 1 - (i * 8 + b)] << (8 * b); l[i] = v; } }
static bool all_zero(const uint64_t *l, int n) { uint64_t o = 0; for (int i = 0; i < n; i++) o |= l[i]; return o == 0; }
static void g1_uncompressed(uint8_t *o, const uint64_t *p) { if (all_zero(p, 12)) { memset(o, 0, 96); o[0] = 0x40; return; } be_from_limbs(o, p, 6); be_from_limbs(o + 48, p + 6, 6); }
static void g2_uncompressed(uint8_t *o, const uint64_t *p) {
    if (all_zero(p, 24)) { memset(o, 0, 192); o[0] = 0x40; return; }
    be_from_limbs(o, p + 6, 6); be_from_limbs(o + 48, p, 6); be_from_limbs(o + 96, p + 18, 6); be_from_limbs(o + 144, p + 12, 6);
}
static bool fr_below_r(const uint64_t *k) { for (int i = 3; i >= 0; i--) { if (k[i] < FrP64::MOD[i]) return true; if (k[i] > FrP64::MOD[i]) return false; } return false; }

}  // namespace vsp

using namespace vsp;

extern "C" {

// ---- scalar vectors (primary input, eid / sn / rt, voting result): common.hpp:476-485, 529-535 --------------------------------------
size_t vsp_fr_vector_blob_size(size_t count) { return 8 + 32 * count; }
int vsp_fr_vector_to_blob(const uint64_t *vals, size_t count, uint8_t *out) {
    if ((!vals && count) || !out) return VSP_ERR_ARG;
    for (size_t i = 0; i < count; i++) if (!fr_below_r(vals + 4 * i)) return VSP_ERR_ARG;
    put_be64(out, count);
    for (size_t i = 0; i < count; i++) be_from_limbs(out + 8 + 32 * i, vals + 4 * i, 4);
    return VSP_OK;
}
int vsp_fr_vector_from_blob(const uint8_t *blob, size_t len, uint64_t *vals_out, size_t capacity, size_t *count_out) {
    if (!blob || len < 8 || !count_out) return VSP_ERR_ARG;
    uint64_t n = get_be64(blob);
    if (n > (len - 8) / 32 || len != 8 + 32 * n) return VSP_ERR_ARG;
    *count_out = (size_t)n;
    if (!vals_out) return VSP_OK;                                  // size query
    if (capacity < n) return VSP_ERR_ARG;
    for (size_t i = 0; i < n; i++) { limbs_from_be(vals_out + 4 * i, blob + 8 + 32 * i, 4); if (!fr_below_r(vals_out + 4 * i)) return VSP_ERR_ARG; }
    return VSP_OK;
}

// ---- G1 vectors (the ciphertext, r1cs_gg_ppzksnark_encrypted_primary_input: common.hpp:471-474, 773-781) ----------------------------
size_t vsp_g1_vector_blob_size(size_t count) { return 8 + 48 * count; }
int vsp_g1_vector_to_blob(const uint64_t *pts, size_t count, uint8_t *out) {
    if ((!pts && count) || !out) return VSP_ERR_ARG;
    put_be64(out, count);
    for (size_t i = 0; i < count; i++) if (vsp_g1_compress(pts + 12 * i, out + 8 + 48 * i) != VSP_OK) return VSP_ERR_ARG;
    return VSP_OK;
}
int vsp_g1_vector_from_blob(const uint8_t *blob, size_t len, int check_subgroup, uint64_t *pts_out, size_t capacity, size_t *count_out) {
    if (!blob || len < 8 || !count_out) return VSP_ERR_ARG;
    uint64_t n = get_be64(blob);
    if (n > (len - 8) / 48 || len != 8 + 48 * n) return VSP_ERR_ARG;
    *count_out = (size_t)n;
    if (!pts_out) return VSP_OK;
    if (capacity < n) return VSP_ERR_ARG;
    for (size_t i = 0; i < n; i++) { int inf; if (vsp_g1_decompress(blob + 8 + 48 * i, check_subgroup, pts_out + 12 * i, &inf) != VSP_OK) return VSP_ERR_ARG; }
    return VSP_OK;
}

// ---- proof (r1cs_gg_ppzksnark_proof: common.hpp:467-469) -- the layout of data.bin[0:192) ------------------------------------------
int vsp_proof_from_blob(const uint8_t blob[192], int check_subgroup, uint64_t A[12], uint64_t B[24], uint64_t C[12]) {
    if (!blob || !A || !B || !C) return VSP_ERR_ARG;
    int inf;
    if (vsp_g1_decompress(blob, check_subgroup, A, &inf) != VSP_OK || vsp_g2_decompress(blob + 48, check_subgroup, B, &inf) != VSP_OK ||
        vsp_g1_decompress(blob + 144, check_subgroup, C, &inf) != VSP_OK) return VSP_ERR_ARG;
    return VSP_OK;
}
int vsp_proof_to_blob(const uint64_t A[12], const uint64_t B[24], const uint64_t C[12], uint8_t out[192]) {
    if (!A || !B || !C || !out) return VSP_ERR_ARG;
    if (vsp_g1_compress(A, out) != VSP_OK || vsp_g2_compress(B, out + 48) != VSP_OK || vsp_g1_compress(C, out + 144) != VSP_OK) return VSP_ERR_ARG;
    return VSP_OK;
}

// ---- extended verification key (r1cs_gg_ppzksnark_extended_verification_key: common.hpp:183-185) -----------------------------------
// head (4) | alpha_g1_beta_g2 (576, opaque here: the pairing lives on the verifier side) | gamma_g2 (96) | delta_g2 (96) | delta_g1 (48) |
// count (8) | gamma_ABC_g1 (count x 48) | gamma_g1 (48)
size_t vsp_vk_blob_size(size_t n_abc) { return 4 + 576 + 96 + 96 + 48 + 8 + 48 * n_abc + 48; }
int vsp_vk_to_blob(uint32_t head, const uint8_t gt[576], const uint64_t gamma_g2[24], const uint64_t delta_g2[24], const uint64_t delta_g1[12],
                   const uint64_t *gamma_abc_g1, size_t n_abc, const uint64_t gamma_g1[12], uint8_t *out) {
    if (!gt || !gamma_g2 || !delta_g2 || !delta_g1 || (!gamma_abc_g1 && n_abc) || !gamma_g1 || !out) return VSP_ERR_ARG;
    for (int i = 0; i < 4; i++) out[i] = (uint8_t)(head >> (24 - 8 * i));
    memcpy(out + 4, gt, 576);
    uint8_t *p = out + 580;
    if (vsp_g2_compress(gamma_g2, p) != VSP_OK || vsp_g2_compress(delta_g2, p + 96) != VSP_OK || vsp_g1_compress(delta_g1, p + 192) != VSP_OK) return VSP_ERR_ARG;
    put_be64(p + 240, n_abc);
    for (size_t i = 0; i < n_abc; i++) if (vsp_g1_compress(gamma_abc_g1 + 12 * i, p + 248 + 48 * i) != VSP_OK) return VSP_ERR_ARG;
    if (vsp_g1_compress(gamma_g1, p + 248 + 48 * n_abc) != VSP_OK) return VSP_ERR_ARG;
    return VSP_OK;
}
int vsp_vk_from_blob(const uint8_t *blob, size_t len, int check_subgroup, uint32_t *head, uint8_t gt[576], uint64_t gamma_g2[24], uint64_t delta_g2[24],
                     uint64_t delta_g1[12], uint64_t *gamma_abc_g1, size_t capacity, size_t *n_abc, uint64_t gamma_g1[12]) {
    if (!blob || len < vsp_vk_blob_size(0) || !n_abc) return VSP_ERR_ARG;
    const uint8_t *p = blob + 580;
    uint64_t n = get_be64(p + 240);
    if (n > (len - vsp_vk_blob_size(0)) / 48 || len != vsp_vk_blob_size((size_t)n)) return VSP_ERR_ARG;
    *n_abc = (size_t)n;
    if (head) *head = ((uint32_t)blob[0] << 24) | ((uint32_t)blob[1] << 16) | ((uint32_t)blob[2] << 8) | blob[3];
    if (gt) memcpy(gt, blob + 4, 576);
    int inf;
    if (gamma_g2 && vsp_g2_decompress(p, check_subgroup, gamma_g2, &inf) != VSP_OK) return VSP_ERR_ARG;
    if (delta_g2 && vsp_g2_decompress(p + 96, check_subgroup, delta_g2, &inf) != VSP_OK) return VSP_ERR_ARG;
    if (delta_g1 && vsp_g1_decompress(p + 192, check_subgroup, delta_g1, &inf) != VSP_OK) return VSP_ERR_ARG;
    if (gamma_abc_g1) {
        if (capacity < n) return VSP_ERR_ARG;
        for (size_t i = 0; i < n; i++) if (vsp_g1_decompress(p + 248 + 48 * i, check_subgroup, gamma_abc_g1 + 12 * i, &inf) != VSP_OK) return VSP_ERR_ARG;
    }
    if (gamma_g1 && vsp_g1_decompress(p + 248 + 48 * n, check_subgroup, gamma_g1, &inf) != VSP_OK) return VSP_ERR_ARG;
    return VSP_OK;
}

// ---- proving key (r1cs_gg_ppzksnark_fast_proving_key: common.hpp:186-188, read at :749-754 inside the timed vote phase) -------------
// alpha_g1 (96) beta_g1 (96) beta_g2 (192) delta_g1 (96) delta_g2 (192) | count A (8) A_query (x 96) | count B (8) B_query (x (192 + 96)) |
// count H (8) H_query (x 96) | count L (8) L_query (x 96)
static const size_t PK_FIXED = 96 + 96 + 192 + 96 + 192;
size_t vsp_pk_blob_size(const vsp_keypair *kp) {
    if (!kp) return 0;
    return PK_FIXED + 8 + 96 * kp->q[0]->n + 8 + 288 * kp->q[2]->n + 8 + 96 * kp->q[3]->n + 8 + 96 * kp->q[4]->n;
}
int vsp_pk_to_blob(vsp_ctx *ctx, const vsp_keypair *kp, uint8_t *out) {
    if (!ctx) return VSP_ERR_ARG;
    if (!kp || !out) return set_error(ctx, VSP_ERR_ARG, "pk_to_blob: null argument");
    if (kp->q[1]->n != kp->q[2]->n) return set_error(ctx, VSP_ERR_ARG, "pk_to_blob: B_query halves differ in length");
    uint8_t *p = out;
    g1_uncompressed(p, kp->alpha_g1); g1_uncompressed(p + 96, kp->beta_g1); g2_uncompressed(p + 192, kp->beta_g2);
    g1_uncompressed(p + 384, kp->delta_g1); g2_uncompressed(p + 480, kp->delta_g2);
    p += PK_FIXED;
    std::vector<uint64_t> a, b;
    auto g1_section = [&](int which) -> int {
        size_t n = kp->q[which]->n;
        a.resize(12 * (n ? n : 1));
        VSP_TRY(vsp_keypair_export(ctx, kp, which, a.data()));
        put_be64(p, n); p += 8;
        for (size_t i = 0; i < n; i++) g1_uncompressed(p + 96 * i, a.data() + 12 * i);
        p += 96 * n;
        return VSP_OK;
    };
    VSP_TRY(g1_section(0));
    {
        size_t n = kp->q[2]->n;
        a.resize(12 * (n ? n : 1)); b.resize(24 * (n ? n : 1));
        VSP_TRY(vsp_keypair_export(ctx, kp, 1, a.data())); VSP_TRY(vsp_keypair_export(ctx, kp, 2, b.data()));
        put_be64(p, n); p += 8;
        for (size_t i = 0; i < n; i++) { g2_uncompressed(p + 288 * i, b.data() + 24 * i); g1_uncompressed(p + 288 * i + 192, a.data() + 12 * i); }
        p += 288 * n;
    }
    VSP_TRY(g1_section(3));
    VSP_TRY(g1_section(4));
    return VSP_OK;
}

vsp_keypair *vsp_pk_from_blob(vsp_ctx *ctx, const uint8_t *blob, size_t len, int precompute) {
    if (!ctx) return nullptr;
    auto fail = [&](const char *msg) -> vsp_keypair * { if (msg) set_error(ctx, VSP_ERR_ARG, msg); return nullptr; };
    if (!blob || len < PK_FIXED + 32) return fail("pk_from_blob: blob too short");
    // section table from the counts (all offsets are multiples of 8)
    size_t off = PK_FIXED, cnt[4], at[4];
    const size_t rec[4] = {96, 288, 96, 96};
    for (int s = 0; s < 4; s++) {
        if (off + 8 > len) return fail("pk_from_blob: truncated");
        uint64_t n = get_be64(blob + off); off += 8;
        if (n > (len - off) / rec[s]) return fail("pk_from_blob: a count exceeds the blob");
        cnt[s] = (size_t)n; at[s] = off; off += rec[s] * cnt[s];
    }
    if (off != len) return fail("pk_from_blob: trailing bytes");
    if (hipSetDevice(ctx->device) != hipSuccess) return fail("pk_from_blob: device");
    hipStream_t st = ctx->stream;
    void *d_raw = nullptr, *d_pts = nullptr;
    size_t maxn = 1; for (int s = 0; s < 4; s++) if (cnt[s] > maxn) maxn = cnt[s];
    if (hipMalloc(&d_raw, len) != hipSuccess || hipMalloc(&d_pts, maxn * sizeof(G2Affine)) != hipSuccess || ensure(ctx, ctx->val_flag, 16) != VSP_OK) {
        if (d_raw) hipFree(d_raw); if (d_pts) hipFree(d_pts); set_error(ctx, VSP_ERR_NOMEM, "pk_from_blob: hipMalloc failed"); return nullptr;
    }
    vsp_keypair *kp = new vsp_keypair();
    memset(kp->gamma_g2, 0, sizeof kp->gamma_g2); memset(kp->gamma_g1, 0, sizeof kp->gamma_g1);
    bool ok = hipMemcpyAsync(d_raw, blob, len, hipMemcpyHostToDevice, st) == hipSuccess && hipMemsetAsync(ctx->val_flag.p, 0, 16, st) == hipSuccess;
    // which query each section feeds: A -> q0, B -> q2 (G2, record offset 0) and q1 (G1, record offset 192), H -> q3, L -> q4
    struct Part { int sec; size_t shift; int group; int q; } parts[5] = {{0, 0, 1, 0}, {1, 192, 1, 1}, {1, 0, 2, 2}, {2, 0, 1, 3}, {3, 0, 1, 4}};
    for (int k = 0; k < 5 && ok; k++) {
        const Part &pt = parts[k];
        size_t n = cnt[pt.sec];
        unsigned blk = (unsigned)((n + 255) / 256);
        const uint8_t *src = (const uint8_t *)d_raw + at[pt.sec] + pt.shift;
        if (n) {
            if (pt.group == 1) hipLaunchKernelGGL(k_g1_from_be, dim3(blk), dim3(256), 0, st, src, rec[pt.sec], n, (G1Affine *)d_pts, (uint32_t *)ctx->val_flag.p);
            else hipLaunchKernelGGL(k_g2_from_be, dim3(blk), dim3(256), 0, st, src, rec[pt.sec], n, (G2Affine *)d_pts, (uint32_t *)ctx->val_flag.p);
            uint32_t h_flag = 0;
            ok = hipMemcpyAsync(&h_flag, ctx->val_flag.p, 4, hipMemcpyDeviceToHost, st) == hipSuccess && hipStreamSynchronize(st) == hipSuccess;
            if (ok && h_flag) { set_error(ctx, VSP_ERR_ARG, "pk_from_blob: a point record is not a well-formed uncompressed record (compressed form, stray flag bits, or an infinity record with a payload)"); ok = false; break; }
        }
        if (ok) {
            // curve check, Montgomery form, 28-bit table; wire bytes are caller data: the endomorphism layout only after the subgroup check
            // (a query that gets window multiples below never uses that layout: BASES_TRANSIENT skips the layout and its check)
            const bool pre_this = ((precompute & 1) && pt.q != 3) || ((precompute >> (pt.q + 1)) & 1);
            kp->q[pt.q] = bases_create(ctx, pt.group, d_pts, true, n, pre_this ? BASES_TRANSIENT : BASES_CALLER);
            ok = kp->q[pt.q] != nullptr;
        }
    }
    hipStreamSynchronize(st);
    hipFree(d_raw); hipFree(d_pts);         // before the window multiples are built: the raw blob (0.6 GB at 2^20 constraints) is no longer needed
    // precompute: the bit mask of vsp_groth16_generate -- bit 0 = the recommended set A, B(G1), B(G2), L (H stays plain); bits 1..5 = A, B(G1), B(G2), H, L
    for (int i = 0; i < 5 && ok; i++) {
        const bool pre_this = ((precompute & 1) && i != 3) || ((precompute >> (i + 1)) & 1);
        if (pre_this) ok = vsp_bases_precompute(ctx, kp->q[i], 0) == VSP_OK;
    }
    if (ok) {
        // the five single elements on the host (uncompressed records; infinity is not a valid key element)
        auto g1 = [&](const uint8_t *p, uint64_t *o) { if (p[0] & 0xC0) return false; uint8_t t[96]; memcpy(t, p, 96); limbs_from_be(o, t, 6); limbs_from_be(o + 6, t + 48, 6); return true; };
        auto g2 = [&](const uint8_t *p, uint64_t *o) { if (p[0] & 0xC0) return false; limbs_from_be(o + 6, p, 6); limbs_from_be(o, p + 48, 6); limbs_from_be(o + 18, p + 96, 6); limbs_from_be(o + 12, p + 144, 6); return true; };
        ok = g1(blob, kp->alpha_g1) && g1(blob + 96, kp->beta_g1) && g2(blob + 192, kp->beta_g2) && g1(blob + 384, kp->delta_g1) && g2(blob + 480, kp->delta_g2);
        // validate them through the compression round trip (on the curve <=> decompress(compress(p)) == p)
        auto on_curve1 = [&](const uint64_t *p) { uint8_t c[48]; uint64_t q[12]; int inf; return vsp_g1_compress(p, c) == VSP_OK && vsp_g1_decompress(c, 0, q, &inf) == VSP_OK && !memcmp(p, q, 96); };
        auto on_curve2 = [&](const uint64_t *p) { uint8_t c[96]; uint64_t q[24]; int inf; return vsp_g2_compress(p, c) == VSP_OK && vsp_g2_decompress(c, 0, q, &inf) == VSP_OK && !memcmp(p, q, 192); };
        ok = ok && on_curve1(kp->alpha_g1) && on_curve1(kp->beta_g1) && on_curve2(kp->beta_g2) && on_curve1(kp->delta_g1) && on_curve2(kp->delta_g2);
        if (!ok) set_error(ctx, VSP_ERR_ARG, "pk_from_blob: a fixed key element is not a curve point in uncompressed form");
    }
    if (ok) {
        kp->pk = vsp_pk_create(ctx, kp->alpha_g1, kp->beta_g1, kp->beta_g2, kp->delta_g1, kp->delta_g2, kp->q[0], kp->q[1], kp->q[2], kp->q[3], kp->q[4]);
        ok = kp->pk != nullptr;
    }
    if (!ok) { vsp_keypair_free(ctx, kp); return nullptr; }
    return kp;
}

}  // extern "C"
