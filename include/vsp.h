/* vsp.h -- C ABI of the MI355X-native Groth16 prover hot path (libvsp_hip.so).
 *
 * Drop-in boundary for the one compute-heavy path of NilFoundation/vote-saver-protocol: the
 * r1cs_gg_ppzksnark prover that runs below
 *     bin/cli/include/nil/vote_saver/common.hpp:1132-1135   (encrypt<elgamal_verifiable<bls12_381>, ...>)
 * whose two hot loops are crypto3-algebra `multiexp` (included via common.hpp:38) and crypto3-math
 * `evaluation_domain` (absent submodules, .gitmodules:8-9,47-48).  The reference has no FFI for
 * this path (header-only C++ templates); its only C ABI convention is the WASM shell
 * (bin/cli/src/wasm.cpp:32-44,62-201): POD pointer+size buffers, blocking calls.  This header keeps
 * that convention, with two deliberate differences: every buffer is CALLER-owned,
 * and failures are returned as negative status codes instead of aborting the process
 * (reference: BOOST_ASSERT -> std::exit(1), bin/cli/src/main.cpp:24-33).
 *
 * Data layout at the boundary (host or device memory, see each function):
 *   Fr scalar        4 x uint64  canonical (non-Montgomery) little-endian limbs, value < r
 *   Fp element       6 x uint64  canonical little-endian limbs, value < p
 *   G1 affine point 12 x uint64  x, y                      -- infinity = all 96 bytes zero
 *   G2 affine point 24 x uint64  x.c0, x.c1, y.c0, y.c1    -- infinity = all 192 bytes zero
 *   G1 Jacobian     18 x uint64  X, Y, Z (x = X/Z^2, y = Y/Z^3; Z = 0 infinity); G2: 36 x uint64
 * No torch types, no C++ types: plain pointers and sizes.
 *
 * Threading: a vsp_ctx is used by one thread at a time (the reference is single-threaded,
 * bin/cli/CMakeLists.txt:114-116); different contexts are independent.  All calls block until the
 * result is in the caller's buffer unless the name ends in _async.
 */
#ifndef VSP_H
#define VSP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct vsp_ctx vsp_ctx;
typedef struct vsp_bases vsp_bases;     /* device-resident MSM bases (a slice of the proving key) */
typedef struct vsp_r1cs vsp_r1cs;       /* device-resident constraint system (three CSR matrices) */
typedef struct vsp_pk vsp_pk;           /* device-resident Groth16 proving key */

enum {
    VSP_OK = 0,
    VSP_ERR_ARG = -1,        /* null pointer, size out of range, scalar/coordinate layout violated: a scalar >= r (reported when the
                              * multi-exponentiation / proof finishes), a base coordinate >= p or a base off the curve (at upload) */
    VSP_ERR_HIP = -2,        /* a HIP runtime call failed; vsp_last_error() has the text */
    VSP_ERR_NOMEM = -3,
    VSP_ERR_UNSUPPORTED = -4 /* e.g. log_m > 28 */
};

/* ---- context ------------------------------------------------------------------------------- */
/* One context = one GPU (device_ordinal) + one HIP stream + grow-only device workspaces. */
vsp_ctx *vsp_create(int device_ordinal);
void vsp_destroy(vsp_ctx *ctx);
const char *vsp_last_error(vsp_ctx *ctx);
/* Run on a caller-provided hipStream_t (e.g. torch.cuda.Stream().cuda_stream); NULL = context's own. */
int vsp_set_stream(vsp_ctx *ctx, void *hip_stream);
int vsp_synchronize(vsp_ctx *ctx);
/* Named timing/diagnostic values of the last calls, e.g. "msm_accum_ms" (HIP-event time of the bucket
 * accumulation kernel, summed since the last vsp_stats_reset), "msm_accum_launches", "msm_window_bits". */
double vsp_get_stat(vsp_ctx *ctx, const char *name);
void vsp_stats_reset(vsp_ctx *ctx);
/* options: "bases_check_curve" (default 1: uploads verify y^2 = x^3 + b for every point; coordinates < p are always checked),
 * "bases_check_subgroup" (the endomorphism split below relies on phi(P) = lambda P, which holds only in the order-r subgroup, while
 * the curve equation also admits points with a cofactor component.  1, the default: an upload that would use the split checks
 * phi(P) = lambda P for every point on the GPU (126 doublings + 11 additions per point, 20-30 ms for 2^20 G1 points, once per key);
 * bases that fail keep the plain layout, over which sum k_i P_i is exact for ANY curve point -- as the reference's generic multiexp is
 * -- and vsp_get_stat("bases_outside_subgroup") counts them.  2: every upload is checked and one that fails is refused with
 * VSP_ERR_ARG.  0: never checked, and then never split unless "msm_glv" = 2.  Keys made by vsp_groth16_generate are multiples of the
 * generators and need no check; the bases of one vsp_msm_g1 / vsp_msm_g2 call are never split, hence never checked),
 * "msm_census_sync" (1: every multi-exponentiation waits for its own 0/1 census before planning; default 0: a slot plans from the
 * count it saw for the previous vector of the same length -- the count steers window size and part length, never the result);
 * "msm_glv" (default 1: plain resident bases keep phi(P) = (beta x, y) = lambda P beside every P in the 28-bit-limb table -- 2 x 128
 * (G1) / 2 x 256 (G2) bytes per point -- and every scalar is split k = k1 + k2 lambda into two signed 127-bit halves: half as many windows,
 * i.e. half the bucket sets to reduce -- applied while the doubled table stays within 256 MB for G1 (2^20 points) / 128 MB for G2
 * (2^18 points): beyond that the measurements favour the plain layout; 2 forces it for any size AND skips the subgroup check -- the caller
 * vouches that every base is in the order-r subgroup, else results are wrong; 0 before an upload keeps the plain layout);
 * tuning knobs: "msm_window_bits" (0 = automatic), "msm_split" (bucket split threshold), "prove_h_first" (1: queue witness_map and
 * the H multi-exponentiation before the witness ones), "msm_fp28" (1: bases are kept a second time on 14 x 28-bit limbs for the
 * accumulation kernel -- 128 (G1) / 256 (G2) bytes per point (cache-line rows) on top of the 96 / 192; 0 before an upload / precomputation leaves that copy out and the
 * 12 x 32-bit kernel runs), "prove_plan_first" (1: the witness vectors' digit sorts are queued before witness_map),
 * "prove_host_threads" (default 1: the prover's host steps -- the four multiples of delta, the Horner chain over each multi-exponentiation's
 * window results, s*A and r*B1 -- run on host threads of their own inside the wait for the GPU; 0: on the calling thread, one after the other),
 * "generate_precompute_window" (8..22; 0 = by the query's size: the window of the tables vsp_groth16_generate builds when `precompute` asks for them),
 * "prove_fixed_base" (default 1: the multiples of delta every proof needs come from fixed-base tables of 32 x 255 multiples per group,
 * built on the host by the first proof over a key -- at most 32 additions per multiple; 0: double-and-add),
 * "prove_batch_share_plan" (default 1: in vsp_groth16_prove_batch the B1 and B2 multi-exponentiations take A's digit sort and bucket plan --
 * the three multiply by the same witness vectors; 0: each sorts for itself),
 * "witness_map_batched" (default 1: the three transforms of every step of witness_map in one launch per pass and the pointwise step inside
 * the last transform's first pass, basic domains on the 29-bit butterflies; 0: transform by transform), "msm_dimsum_lanes" (8/16/32/64 lanes per bucket-digit sum; 0 = chosen by the library),
 * "msm_dimsum_maxw" (256..4096, default 1024: waves the per-digit lane plan of the bucket reduction may fill; 2048 = two per SIMD),
 * "msm_dimsum_prefetch" (1: the bucket reduction requests the next bucket before the current addition; default 0, it spills), "msm_dimbits" (1 / 0: the last
 * step of the bucket reduction as plain subset sums folded by the host's doubling chain / as weighted sums on the GPU; default by group),
 * "msm_slot_normal_priority" (1 before the first use of a work slot: its stream gets the context's priority instead of the lowest --
 * faster single proofs, slower independent multi-exponentiations in flight), "ntt_fr29" (default 1: butterflies on
 * 9 x 29-bit limbs; 0: the 8 x 32-bit kernel); diagnostics: every context checks its hand-laid-out field routines THROUGH the kernels
 * that use them, against the generic kernels, on data the library generates itself -- vsp_get_stat "msm_fp28_selfcheck_g1" / "_g2"
 * (first 28-bit table of a group) and "ntt_fr29_selfcheck" (first transform): 1 passed, -1 failed (the context then runs the generic
 * kernels for its lifetime and vsp_last_error says so), 0 could not run; "msm_fp28_selfcheck_detail_g1" / "_g2": which leg of the check
 * differed (bit 0 the default pipeline, 1 split buckets / the other last step, 4 / 5 the 6-bit / 12-bit window legs, 2 an unexpected
 * infinity, 3 a failed launch); "msm_fp28" = 2 (diagnostics only: bisecting a failed check) builds the 28-bit table WITHOUT the check;
 * "msm_debug_counts" (1: vsp_get_stat reports "msm_buckets", "msm_parts", "msm_medium_buckets", "msm_heavy_buckets" of the last
 * multi-exponentiation -- a blocking read-back); "msm_sort" (0: by size; 1: never the staged sort of large wide-window problems;
 * 2: the staged sort for every window of 12 bits and more), "msm_wide_windows" (0: never more than 16 bits per window), "msm_fold" (0: 255-bit
 * scalars are not folded to min(k, r - k) where the window width divides 255), "msm_fused_split" (0: the endomorphism split and the digit
 * extraction run as two kernels over the scalars instead of one), "msm_fused_scans" (0: the bucket scans take three kernels each).
 *
 * Runtime environment.  Results never depend on it.  GPU_MAX_HW_QUEUES (HIP runtime, read once when the runtime starts; default 4
 * hardware queues per stream priority): one proof's latency does not depend on it (the prover's two chains take their queues when the
 * context is created), but independent multi-exponentiations kept IN FLIGHT TOGETHER over the work slots overlap better with 8
 * (2^20 G1 points, four in flight: 2.86 ms per multi-exponentiation against 3.03 with the default 4; MI355X, ROCm 7.2).  A service that
 * pipelines proofs or multi-exponentiations should export GPU_MAX_HW_QUEUES=8 before the process's first HIP call;
 * vsp_get_stat("runtime_hw_queues_env") is the value this process saw (0: not set). */
int vsp_set_option(vsp_ctx *ctx, const char *name, long value);
/* Diagnostic build only (libvsp_hip_diag.so, `make -C vote_saver_protocol_amd/csrc diag`: the same sources with stamps around the G1
 * accumulation loop -- in the shipped library no stamp executes and this returns VSP_ERR_UNSUPPORTED): the clock the chip held inside
 * that loop since the last reset, delta s_memtime / delta s_memrealtime x 100 MHz summed over waves, and the number of waves. */
int vsp_diag_clock(vsp_ctx *ctx, int reset, double *ghz_out, double *waves_out);
/* The same for the passes of the radix-2 transform on 29-bit limbs (k_ntt29_pass; stamps around a whole pass of every wave). */
int vsp_diag_clock_ntt(vsp_ctx *ctx, int reset, double *ghz_out, double *waves_out);

/* ---- raw device memory helpers (for callers without torch) --------------------------------- */
void *vsp_dmalloc(vsp_ctx *ctx, size_t bytes);
void vsp_dfree(vsp_ctx *ctx, void *dptr);
int vsp_h2d(vsp_ctx *ctx, void *dst_dev, const void *src_host, size_t bytes);
int vsp_d2h(vsp_ctx *ctx, void *dst_host, const void *src_dev, size_t bytes);
/* Page-lock / release caller memory: host buffers the library copies from on every call (the witness of vsp_groth16_prove) then
 * travel by asynchronous DMA instead of the runtime's staged pageable path.  The reference keeps its witness in a std::vector
 * (common.hpp:1110-1128); registering that storage once costs nothing per proof. */
int vsp_host_register(vsp_ctx *ctx, void *ptr, size_t bytes);
int vsp_host_unregister(vsp_ctx *ctx, void *ptr);

/* ---- multi-scalar multiplication: algebra::multiexp<multiexp_method_BDLO12> (a1), ------------
 *      multiexp_with_mixed_addition (a2: zero scalars are skipped, equal-to-one scalars are summed
 *      directly -- both happen inside the same bucket pipeline), G2 half of kc_multiexp (a3).
 * result = sum_i scalars[i] * bases[i], returned as an affine point (out_is_inf = 1 for infinity). */
int vsp_msm_g1(vsp_ctx *ctx, const uint64_t *bases /* host n x 12 */, const uint64_t *scalars /* host n x 4 */,
               size_t n, uint64_t out_affine[12], int *out_is_inf);
int vsp_msm_g2(vsp_ctx *ctx, const uint64_t *bases /* host n x 24 */, const uint64_t *scalars /* host n x 4 */,
               size_t n, uint64_t out_affine[24], int *out_is_inf);

/* Resident bases (proving-key slices, a9): uploaded and converted once, reused by every proof. */
vsp_bases *vsp_bases_upload_g1(vsp_ctx *ctx, const uint64_t *bases /* host n x 12 */, size_t n);
vsp_bases *vsp_bases_upload_g2(vsp_ctx *ctx, const uint64_t *bases /* host n x 24 */, size_t n);
/* bases already in device memory (same canonical layout), e.g. produced by vsp_fixed_base_mul_* */
vsp_bases *vsp_bases_from_device_g1(vsp_ctx *ctx, const void *d_bases, size_t n);
vsp_bases *vsp_bases_from_device_g2(vsp_ctx *ctx, const void *d_bases, size_t n);
/* Optional preprocessing of resident bases (once per proving key): store 2^(c*w) * P for every c-bit window w (c = window_bits,
 * 0 = chosen from the number of bases: 16 from 2^19 bases up), 255/c + 1 slices = 16x the memory at c = 16.  Every multi-exponentiation over these bases then uses ONE shared set of
 * 2^(c-1) buckets for all windows: the bucket reduction shrinks by the number of windows and the work per thread is uniform.
 * Results are identical; sub-range calls (first, n) keep working. */
int vsp_bases_precompute(vsp_ctx *ctx, vsp_bases *b, unsigned window_bits);
/* The same table for bases whose scalars are DENSE (the H query; not witness vectors, whose scalars are mostly 0 / 1): beside every window
 * multiple its image under the curve's endomorphism, for the ceil(128 / c) windows of a split scalar k = k1 + k2 lambda -- the
 * endomorphism split of "msm_glv" over ONE bucket set.  Same memory as the plain table at c = 16 (8 windows x 2 rows instead of 16).
 * Needs the order-r subgroup: bases not yet checked are checked here ("bases_check_subgroup"); bases outside it get the plain table. */
int vsp_bases_precompute_split(vsp_ctx *ctx, vsp_bases *b, unsigned window_bits);
size_t vsp_bases_count(const vsp_bases *b);
/* device memory the handle holds: the Montgomery-form points (or the table of window multiples) plus the 28-bit-limb copy */
size_t vsp_bases_device_bytes(const vsp_bases *b);
void vsp_bases_free(vsp_ctx *ctx, vsp_bases *b);

/* MSM over resident bases [first, first+n) with scalars in DEVICE memory (n x 4 uint64, canonical).
 * out_affine is a host buffer (12 or 24 uint64 according to the group of `bases`). */
int vsp_msm_resident(vsp_ctx *ctx, const vsp_bases *bases, size_t first, size_t n,
                     const void *d_scalars, uint64_t *out_affine, int *out_is_inf);
/* Same, but the result is left as a Jacobian partial sum (X, Y, Z canonical: 18 / 36 uint64) in the host
 * buffer -- the fixed-size record ranks exchange in the sharded multi-GPU MSM. */
/* A BATCH of multi-exponentiations over the same resident bases: `batch` (1..64) scalar vectors of n canonical scalars each, vector k at
 * d_scalars + 32 * k * stride bytes (stride >= n, in scalars), out_affine[k] / out_is_inf[k] = sum_i scalars_k[i] * bases[first + i].
 * One digit sort, one bucket accumulation and one bucket reduction serve all vectors (separate buckets per vector, the same base rows):
 * what a prover of MANY small statements over one key needs -- small multi-exponentiations are bound by the latency of their dependent
 * chains, not by work, and a batch is as wide as its vectors together at the latency of one.  Plain bases: a bucket set per (vector, window);
 * bases with a table of window multiples (windows of at most 16 bits): ONE bucket set per vector -- fewer, larger windows, worth it where
 * the table is small (VSP_ERR_UNSUPPORTED for wider windows, or with option "msm_batch_tables" = 0). */
int vsp_msm_resident_batch(vsp_ctx *ctx, const vsp_bases *bases, size_t first, size_t n, const void *d_scalars, size_t batch, size_t stride,
                           uint64_t *out_affine /* batch x 12 or 24 */, int *out_is_inf /* batch, may be NULL */);
int vsp_msm_resident_jacobian(vsp_ctx *ctx, const vsp_bases *bases, size_t first, size_t n,
                              const void *d_scalars, uint64_t *out_jacobian);
/* Pipelined form: vsp_msm_launch enqueues the whole multi-exponentiation on work slot `slot` (0..5; slot 0 runs on the
 * context's stream, the others on streams of their own) and returns once the GPU work is queued; vsp_msm_finish_jacobian
 * waits for that slot and returns the Jacobian record.  Several slots may be in flight, so the latency-bound tail of one
 * multi-exponentiation overlaps the bulk of the next (the prover runs its five this way).  The scalars must be complete in
 * device memory before the launch and must not change until the finish. */
int vsp_msm_launch(vsp_ctx *ctx, unsigned slot, const vsp_bases *bases, size_t first, size_t n, const void *d_scalars);
int vsp_msm_finish_jacobian(vsp_ctx *ctx, unsigned slot, uint64_t *out_jacobian);
/* Same, but the record is left in DEVICE memory at d_out_jacobian (18 / 36 uint64) -- where the RCCL all-gather of the sharded
 * multi-exponentiation reads it.  The record is completed on the host (the last c * W doublings of a multi-exponentiation
 * are one dependent chain, ~50x faster on a CPU core than on a GPU lane), written to a pinned ring entry of the slot and copied by an
 * asynchronous DMA queued on hip_stream (a hipStream_t; NULL = the context's stream): the call returns without waiting for the copy, and
 * anything queued on hip_stream afterwards sees the record. */
int vsp_msm_finish_jacobian_device(vsp_ctx *ctx, unsigned slot, void *d_out_jacobian, void *hip_stream);

/* Fold `count` Jacobian records (host, canonical) into one affine point; group: 1 = G1, 2 = G2. */
int vsp_fold_jacobian(vsp_ctx *ctx, int group, const uint64_t *records, size_t count,
                      uint64_t *out_affine, int *out_is_inf);
/* Same over records in DEVICE memory (the all-gather's output): copied to a pinned buffer behind whatever hip_stream (NULL = the
 * context's stream) already holds, that stream is waited for, the fold runs on the host.  count <= 4096. */
int vsp_fold_jacobian_device(vsp_ctx *ctx, int group, const void *d_records, size_t count, void *hip_stream,
                             uint64_t *out_affine, int *out_is_inf);

/* ---- evaluation_domain<Fr> (a6): basic radix-2 domain of size m = 2^log_m ------------------
 * In-place on n = 2^log_m canonical Fr values.
 *   inverse = 0, coset_g = NULL : fft          a[k] <- sum_j a[j] w^(jk),  w = 7^((r-1)/2^32)^(2^(32-log_m))
 *   inverse = 1, coset_g = NULL : inverse_fft  (w^-1, then scale by m^-1)
 *   inverse = 0, coset_g = g    : coset fft    (a[j] *= g^j first)
 *   inverse = 1, coset_g = g    : inverse coset fft (inverse_fft, then a[j] *= g^-j) */
int vsp_ntt_fr(vsp_ctx *ctx, uint64_t *a /* host m x 4 */, unsigned log_m, int inverse, const uint64_t coset_g[4]);
int vsp_ntt_fr_device(vsp_ctx *ctx, void *d_a /* device m x 4 */, unsigned log_m, int inverse, const uint64_t coset_g[4]);

/* ---- r1cs_to_qap::witness_map (a7) -----------------------------------------------------------
 * From the three evaluation vectors Az, Bz, Cz over the domain (each m x 4, canonical, zero padded; the
 * caller has already placed the "input_i * 0 = 0" rows in Az) compute the m coefficients of
 * H = (A*B - C)/Z  (d1 = d2 = d3 = 0, as the r1cs_gg_ppzksnark prover calls it).  Inputs are overwritten. */
int vsp_witness_map_h(vsp_ctx *ctx, uint64_t *Az, uint64_t *Bz, uint64_t *Cz /* host */, unsigned log_m,
                      uint64_t *H /* host m x 4 */);
int vsp_witness_map_h_device(vsp_ctx *ctx, void *d_Az, void *d_Bz, void *d_Cz, unsigned log_m, void *d_H);

/* ---- evaluation_domain<Fr> handles (a6): make_evaluation_domain, basic_radix2_domain, step_radix2_domain -----------------
 * Replaces crypto3-math math/domains/evaluation_domain.hpp (abstract m, fft, inverse_fft, evaluate_all_lagrange_polynomials,
 * get_domain_element, compute_vanishing_polynomial, add_poly_z, divide_by_z_on_coset), basic_radix2_domain.hpp,
 * step_radix2_domain.hpp and math/algorithms/make_evaluation_domain.hpp (absent submodule, /root/reference/.gitmodules:47-48),
 * which r1cs_to_qap reaches from bin/cli/include/nil/vote_saver/common.hpp:916-917 and :1132-1135.
 * vsp_domain_create(min_size) makes the choice make_evaluation_domain makes: the basic radix-2 domain when min_size is a power of
 * two, else the step radix-2 domain of size big + small (big = 2^(ceil_log2(min_size)-1), small = the next power of two of
 * min_size - big), which is again basic when small == big.  extended_radix2 (m = 2^33 for this field) and the
 * arithmetic/geometric sequence domains (Fr defines no such generators) cannot be selected for any size up to 2^28.
 * Element order of a step domain: the big-th roots of unity w^(2i), then w * (small-th roots), w of order 2 big.
 * NULL is returned (vsp_last_error tells why) for min_size <= 1 or > 2^28.  A domain belongs to the context's device. */
typedef struct vsp_domain vsp_domain;
vsp_domain *vsp_domain_create(vsp_ctx *ctx, size_t min_size);
void vsp_domain_free(vsp_ctx *ctx, vsp_domain *dom);
size_t vsp_domain_size(const vsp_domain *dom);                 /* m */
int vsp_domain_kind(const vsp_domain *dom);                    /* 0 basic_radix2, 1 step_radix2 */
/* fft / inverse_fft / coset variants, in place on m canonical values (same flags as vsp_ntt_fr) */
int vsp_domain_fft(vsp_ctx *ctx, const vsp_domain *dom, uint64_t *a /* host m x 4 */, int inverse, const uint64_t coset_g[4]);
int vsp_domain_fft_device(vsp_ctx *ctx, const vsp_domain *dom, void *d_a /* device m x 4 */, int inverse, const uint64_t coset_g[4]);
/* evaluate_all_lagrange_polynomials(t): out[i] = L_i(t), m values (the indicator vector when t lies in the domain) */
int vsp_domain_lagrange(vsp_ctx *ctx, const vsp_domain *dom, const uint64_t t[4], uint64_t *out /* host m x 4 */);
int vsp_domain_element(vsp_ctx *ctx, const vsp_domain *dom, size_t idx, uint64_t out[4]);            /* get_domain_element */
int vsp_domain_vanishing(vsp_ctx *ctx, const vsp_domain *dom, const uint64_t t[4], uint64_t out[4]);  /* compute_vanishing_polynomial */
int vsp_domain_add_poly_z(vsp_ctx *ctx, const vsp_domain *dom, const uint64_t coeff[4], uint64_t *H /* host (m+1) x 4 */);
/* divide_by_z_on_coset: P[i] /= Z(g x_i) with g the field's multiplicative generator 7 */
int vsp_domain_divide_by_z_on_coset(vsp_ctx *ctx, const vsp_domain *dom, uint64_t *P /* host m x 4 */);
/* witness_map over a domain handle (vsp_witness_map_h is the basic radix-2 case) */
int vsp_domain_witness_map_h(vsp_ctx *ctx, const vsp_domain *dom, uint64_t *Az, uint64_t *Bz, uint64_t *Cz, uint64_t *H /* host m x 4 each */);

/* ---- constraint system + proving key + prover (a8, a9) -------------------------------------- */
/* Three CSR matrices over columns 0..num_vars (column 0 is the constant 1); coefficients canonical Fr. */
vsp_r1cs *vsp_r1cs_upload(vsp_ctx *ctx, size_t num_constraints, size_t num_inputs, size_t num_vars,
                          const uint32_t *row_ptr_a, const uint32_t *col_a, const uint64_t *coef_a,
                          const uint32_t *row_ptr_b, const uint32_t *col_b, const uint64_t *coef_b,
                          const uint32_t *row_ptr_c, const uint32_t *col_c, const uint64_t *coef_c);
void vsp_r1cs_free(vsp_ctx *ctx, vsp_r1cs *cs);
/* the domain r1cs_to_qap uses for this system: make_evaluation_domain(num_constraints + num_inputs + 1); H_query has size - 1 bases */
size_t vsp_r1cs_domain_size(const vsp_r1cs *cs);
int vsp_r1cs_domain_kind(const vsp_r1cs *cs);                  /* 0 basic_radix2, 1 step_radix2 */

/* Proving key = { alpha_g1, beta_g1, beta_g2, delta_g1, delta_g2, A_query[num_vars+1],
 * B_query (G2 and G1 halves, num_vars+1 each), H_query[m-1], L_query[num_vars-num_inputs] }
 * (proof_system::proving_key_type, common.hpp:173,749-754).  Query handles stay owned by the caller. */
vsp_pk *vsp_pk_create(vsp_ctx *ctx, const uint64_t alpha_g1[12], const uint64_t beta_g1[12], const uint64_t beta_g2[24],
                      const uint64_t delta_g1[12], const uint64_t delta_g2[24],
                      const vsp_bases *A_query, const vsp_bases *B_query_g1, const vsp_bases *B_query_g2,
                      const vsp_bases *H_query, const vsp_bases *L_query);
void vsp_pk_free(vsp_ctx *ctx, vsp_pk *pk);

/* r1cs_gg_ppzksnark_prover::process with explicit randomness r, s (upstream draws them from a
 * non-seedable device, common.hpp:1131).  witness = primary || auxiliary, num_vars x 4 canonical (host).
 * If saver_P1 != NULL, r_enc * P1 is added to C (encrypted-input / SAVER mode).
 * Outputs: affine A (G1), B (G2), C (G1) canonical, and the 192-byte ZCash-compressed proof A||B||C
 * (the format of the reference's bin/cli/src/data.bin[0:192]); any output pointer may be NULL. */
int vsp_groth16_prove(vsp_ctx *ctx, const vsp_r1cs *cs, const vsp_pk *pk, const uint64_t *witness,
                      const uint64_t r[4], const uint64_t s[4],
                      const uint64_t *saver_P1 /* 12 or NULL */, const uint64_t *saver_r_enc /* 4 or NULL */,
                      uint64_t A_out[12], uint64_t B_out[24], uint64_t C_out[12], uint8_t proof_out[192]);

/* A BATCH of proofs over one key: `count` (1..64) witnesses of num_vars canonical values each, one after the other; r, s: count x 4 words;
 * outputs count x 12 / 24 / 12 words and count x 192 bytes (any may be NULL).  Proof k is byte-identical to vsp_groth16_prove(witness_k,
 * r_k, s_k).  Proofs of a small circuit (2^15..2^16 constraints) are bound by the latency of their dependent kernel chains, not by work:
 * proved together, the same launches run `count` times as wide -- one witness_map over 3 x count transforms, every multi-exponentiation
 * once over `count` scalar vectors.  Over a PLAIN key (vsp_groth16_generate with precompute = 0) every (witness, window) pair has its own
 * bucket set; over a key with tables of window multiples (windows of at most 16 bits) every witness has ONE per query, which is ~12 %
 * faster where the tables are small -- 2^16 constraints, precompute = 17 (all five queries), option "generate_precompute_window" = 14:
 * 1.7 GB -- and no use at 2^20 (19 GB).  Option "msm_batch_tables" = 0 refuses table keys (VSP_ERR_UNSUPPORTED), as the first version did.
 * No SAVER addend (vsp_saver_encrypt proves one vote). */
int vsp_groth16_prove_batch(vsp_ctx *ctx, const vsp_r1cs *cs, const vsp_pk *pk, const uint64_t *witnesses, size_t count,
                            const uint64_t *r, const uint64_t *s, uint64_t *A_out, uint64_t *B_out, uint64_t *C_out, uint8_t *proofs_out);
/* The batch in two halves, as vsp_groth16_prove_launch / _finish below: launch copies the witnesses, r and s, queues every kernel of the
 * `count` proofs and returns; finish does the host-side scalar multiplications, waits and assembles.  One host thread with two contexts
 * over one key keeps the card busy while it assembles the other context's batch.  One batch (or one proof) in flight per context. */
int vsp_groth16_prove_batch_launch(vsp_ctx *ctx, const vsp_r1cs *cs, const vsp_pk *pk, const uint64_t *witnesses, size_t count,
                                   const uint64_t *r, const uint64_t *s);
int vsp_groth16_prove_batch_finish(vsp_ctx *ctx, uint64_t *A_out, uint64_t *B_out, uint64_t *C_out, uint8_t *proofs_out);
/* The same proof in two halves, so that ONE host thread keeps several proofs in flight over one resident key -- one per context:
 *     vsp_groth16_prove_launch(ctxA, ...); vsp_groth16_prove_launch(ctxB, ...); vsp_groth16_prove_finish(ctxA, ...); launch(ctxA, next) ...
 * launch queues every kernel of the proof on the context's streams and returns (it copies r, s and the SAVER term; a witness in
 * page-locked memory, vsp_host_register, must stay unchanged until the finish -- pageable memory is copied before launch returns);
 * finish does the host-side scalar multiplications, waits for the GPU and assembles the proof.  One proof in flight per context. */
int vsp_groth16_prove_launch(vsp_ctx *ctx, const vsp_r1cs *cs, const vsp_pk *pk, const uint64_t *witness, const uint64_t r[4], const uint64_t s[4],
                             const uint64_t *saver_P1 /* 12 or NULL */, const uint64_t *saver_r_enc /* 4 or NULL */);
int vsp_groth16_prove_finish(vsp_ctx *ctx, uint64_t A_out[12], uint64_t B_out[24], uint64_t C_out[12], uint8_t proof_out[192]);
/* Packed witness: a Groth16 witness is mostly wires equal to 0 or 1, and its 32 bytes per wire cross PCIe in front of every proof with the
 * GPU idle (0.8 ms at 2^20 constraints).  Packed form: class_words = two bits per wire (0 = zero, 1 = one, 2 = a dense value; 3 reserved),
 * 32 wires per uint64, (num_vars + 31) / 32 words; word_offsets[w] = the index, among the dense values, of word w's first one; dense = the
 * dense values in wire order, 4 canonical words each.  A witness generator can emit this form directly (the reference builds its witness
 * wire by wire, common.hpp:1110-1128); vsp_witness_pack converts a plain witness in one host pass.  The launch validates the map
 * against n_dense and expands it on the GPU; everything else is vsp_groth16_prove_launch. */
size_t vsp_witness_pack_words(size_t num_vars);
int vsp_witness_pack(const uint64_t *witness, size_t num_vars, uint64_t *class_words, uint32_t *word_offsets, uint64_t *dense_out /* or NULL: count only */,
                     size_t dense_capacity, size_t *n_dense_out);
int vsp_groth16_prove_launch_packed(vsp_ctx *ctx, const vsp_r1cs *cs, const vsp_pk *pk, const uint64_t *class_words, const uint32_t *word_offsets,
                                    const uint64_t *dense, size_t n_dense, const uint64_t r[4], const uint64_t s[4],
                                    const uint64_t *saver_P1, const uint64_t *saver_r_enc);

/* ---- Groth16 generator: zk::generate<proof_system>(constraint_system), bin/cli/.../common.hpp:916-917 ----
 * r1cs_gg_ppzksnark_generator with explicit toxic waste toxic[20] = (t, alpha, beta, gamma, delta), 4 limbs each, canonical
 * (upstream draws them from algebraic_random_device).  Builds the whole key on the GPU: Lagrange coefficients at t, the
 * per-variable QAP evaluations, the exponent vectors and the six batch exponentiations.  precompute additionally stores the window
 * multiples (vsp_bases_precompute) of proving-key queries: bit 0 = the recommended set (A, both halves of B, L; H stays plain: dense scalars
 * gain nothing from it and its 2 GB table would miss the cache); bits 1..5 select A_query, B_query (G1), B_query (G2), H_query, L_query one by one.  The key owns its queries.
 * WHICH TO CHOOSE (measured, 2^20 constraints, 90 % boolean witness, MI355X): precompute = 0, the PLAIN key, holds 1.95 GB and proves in
 * 6.65 ms (194 proofs/s from one thread with two proofs in flight); precompute = 1 holds 19.2 GB and proves in 6.26 ms (203 proofs/s).  The
 * tables buy ~5 % for ten times the memory: keep the key plain unless single-proof latency at that size is what matters -- and
 * vsp_groth16_prove_batch, the way to prove many statements of one circuit, takes plain keys only. */
typedef struct vsp_keypair vsp_keypair;
vsp_keypair *vsp_groth16_generate(vsp_ctx *ctx, const vsp_r1cs *cs, const uint64_t toxic[20], int precompute);
const vsp_pk *vsp_keypair_pk(const vsp_keypair *kp);
/* Key components as canonical affine points (host).  which: 0 A_query, 1 B_query (G1 half), 2 B_query (G2 half), 3 H_query,
 * 4 L_query, 5 gamma_ABC_g1 (verification key), 6 alpha_g1, 7 beta_g1, 8 delta_g1, 9 beta_g2, 10 delta_g2, 11 gamma_g2,
 * 12 gamma_g1 (extended verification key: the SAVER key generation needs it). */
size_t vsp_keypair_count(const vsp_keypair *kp, int which);
size_t vsp_keypair_device_bytes(const vsp_keypair *kp);     /* device memory of the whole key (the six queries) */
int vsp_keypair_export(vsp_ctx *ctx, const vsp_keypair *kp, int which, uint64_t *out);
void vsp_keypair_free(vsp_ctx *ctx, vsp_keypair *kp);

/* ---- SAVER wrapper around the prover: elgamal_verifiable<bls12_381> as the vote phase calls it ----------------
 *     generate_keypair<elgamal_verifiable>(rnd[3 n + 2], {gg_keypair, n})                common.hpp:921-931      n = msg_size = 25 (:163)
 *     encrypt<...>(m_field, {d(), pk_eid, gg_keypair, primary_input, auxiliary_input})   common.hpp:1131-1135    <- proves
 *     rerandomize<...>(rnd[3], ct, {pk_eid, gg_keypair, proof})                          common.hpp:1138-1145
 * (crypto3-pubkey, absent submodule: the SAVER scheme; member names below are upstream's.)  G_i = gamma_ABC_g1[i], H = G2 generator.
 *   rnd (3n + 2 scalars)   s_1..s_n | v_1..v_n | t_0..t_n | rho
 *   public key, flat       delta_g1 (12) | delta_s_g1[i] = s_i delta_g1 (n x 12) | t_g1[i] = t_i G_i (n x 12) | t_g2[j] = t_j H ((n+1) x 24) |
 *                          delta_sum_s_g1 = (t_0 + sum t_j s_j) delta_g1 (12) | gamma_inverse_sum_s_g1 = -(1 + sum s_j) gamma_g1 (12)
 *   secret key             rho (4)
 *   verification key, flat rho_g2 = rho H (24) | rho_sv_g2[i] = s_i v_i H (n x 24) | rho_rhov_g2[i] = rho v_i H (n x 24)
 *   ciphertext             c_0 = r delta_g1 | c_i = r delta_s_g1[i] + m_i G_i | psi = r delta_sum_s_g1 + sum m_i t_g1[i]     ((n + 2) x 12)
 * The random values upstream draws from algebraic_random_device (common.hpp:923, 1131, 1139) are explicit inputs here.
 * gamma_abc_g1 points at the first n + 1 entries of the verification key's accumulation vector (constant term, then the n
 * message inputs).  Decryption and the two verifications are pairing work on the tally / verifier side, outside this path.
 * vsp_saver_keygen, vsp_saver_pk_load and vsp_saver_rerandomize are host-only and accept ctx = NULL. */
typedef struct vsp_saver_pk vsp_saver_pk;
size_t vsp_saver_pk_words(size_t msg_size);     /* uint64 words of the flat public key */
size_t vsp_saver_vk_words(size_t msg_size);
int vsp_saver_keygen(vsp_ctx *ctx, size_t msg_size, const uint64_t delta_g1[12], const uint64_t gamma_g1[12], const uint64_t *gamma_abc_g1,
                     const uint64_t *rnd, uint64_t *pk_out, uint64_t sk_out[4], uint64_t *vk_out);
/* public key resident for many votes: validates the G1 elements and builds the fixed-base tables of its msg_size + 3 bases */
vsp_saver_pk *vsp_saver_pk_load(vsp_ctx *ctx, size_t msg_size, const uint64_t *pk_words, const uint64_t *gamma_abc_g1);
void vsp_saver_pk_free(vsp_ctx *ctx, vsp_saver_pk *spk);
size_t vsp_saver_pk_msg_size(const vsp_saver_pk *spk);
/* encrypt: ciphertext of msg (n x 4, must equal the first n entries of witness = primary || auxiliary) under randomness r_enc, and
 * the Groth16 proof of the statement with C += r_enc * gamma_inverse_sum_s_g1 (vsp_groth16_prove with explicit r, s).  The
 * ciphertext is computed on the host while the GPU proves.  ct_out: (n + 2) x 12. */
int vsp_saver_encrypt(vsp_ctx *ctx, const vsp_saver_pk *spk, const vsp_r1cs *cs, const vsp_pk *pk, const uint64_t *msg, const uint64_t *witness,
                      const uint64_t r_enc[4], const uint64_t r[4], const uint64_t s[4],
                      uint64_t *ct_out, uint64_t A_out[12], uint64_t B_out[24], uint64_t C_out[12], uint8_t proof_out[192]);
/* rerandomize in place with rnd = (r', z1, z2):  ct_i += r' X_i,  A' = z1 A,  B' = z1^-1 B + z2 delta_g2,
 * C' = C + (z1 z2) A + r' gamma_inverse_sum_s_g1;  proof_out (may be NULL) receives the 192 compressed bytes of the new proof */
int vsp_saver_rerandomize(vsp_ctx *ctx, const vsp_saver_pk *spk, const uint64_t delta_g2[24], const uint64_t rnd[12],
                          uint64_t *ct, uint64_t A[12], uint64_t B[24], uint64_t C[12], uint8_t proof_out[192]);

/* ---- generator-side batch exponentiation (section 8(f).1; also builds synthetic benchmark bases) ---
 * out[i] = scalars[i] * generator, written to DEVICE memory as canonical affine points. */
int vsp_fixed_base_mul_g1(vsp_ctx *ctx, const void *d_scalars, size_t n, void *d_out /* n x 12 u64 */);
int vsp_fixed_base_mul_g2(vsp_ctx *ctx, const void *d_scalars, size_t n, void *d_out /* n x 24 u64 */);

/* ---- diagnostics ---------------------------------------------------------------------------------
 * Elementwise field arithmetic on the GPU over n canonical values (host buffers): field 0 = Fp (6 limbs),
 * 1 = Fr (4 limbs); op 0 mul, 1 add, 2 sub, 3 sqr(a), 4 canonical-times-Montgomery product, 5 inverse(a); Fp only: ops 6..12
 * exercise the 14 x 28-bit lazy field of the G1 accumulation (round trip, product, the lazy subtractions and the negation).
 * Lets the tests check the kernels' Montgomery arithmetic directly against known-answer vectors. */
int vsp_selftest_field(vsp_ctx *ctx, int field, int op, const uint64_t *a, const uint64_t *b, uint64_t *out, size_t n);
/* The full addition of two bucket sums on its own: out[i] = a[i] + b[i], points as canonical (X, Y, ZZ, ZZZ) records with x = X / ZZ,
 * y = Y / ZZZ, ZZ^3 = ZZZ^2, all-zero = infinity (group 1: 4 x 6 limbs, group 2: 4 x 12 limbs; host buffers).  form 0 = the 12 x 32-bit
 * formulas, 1 = the 14 x 28-bit lazy form the merges and the bucket reduction run.  The result is some representation of the sum
 * (compare X / ZZ, Y / ZZZ).  Covers the exceptional cases -- doubling, cancellation, infinity on either side -- lane by lane. */
int vsp_selftest_xyzz_add(vsp_ctx *ctx, int group, int form, const uint64_t *a, const uint64_t *b, uint64_t *out, size_t n);

/* ---- wire format helpers (host only, tiny) ---------------------------------------------------
 * compress: VSP_ERR_ARG for a null pointer or a coordinate that is not canonical (>= p). */
int vsp_g1_compress(const uint64_t affine[12], uint8_t out[48]);
int vsp_g2_compress(const uint64_t affine[24], uint8_t out[96]);
/* Inverse: ZCash-compressed bytes -> canonical affine limbs (the 192-byte proof of bin/cli/src/data.bin[0:192] is A (48) | B (96) |
 * C (48)).  VSP_ERR_ARG for an encoding that is not compressed form, not canonical (x >= p, stray bits), not on the curve or --
 * with check_subgroup != 0 -- not in the order-r subgroup. */
int vsp_g1_decompress(const uint8_t in[48], int check_subgroup, uint64_t out_affine[12], int *out_is_inf);
int vsp_g2_decompress(const uint8_t in[96], int check_subgroup, uint64_t out_affine[24], int *out_is_inf);

/* ---- wire formats of the reference's marshaling_policy (common.hpp:168-203 option::big_endian) --------------------
 * PROVISIONAL where marked: the marshalling sources are absent submodules, and only the proof bytes, the scalar vectors and the head of
 * the verification key (4 bytes, the GT element, three points) are pinned by files of the reference.  The tail of the verification key
 * (vsp_vk_*: count + gamma_ABC_g1 + gamma_g1 and the roles of the three points), the ciphertext vector (vsp_g1_vector_*) and the whole
 * "fast" proving-key layout (vsp_pk_to_blob / vsp_pk_from_blob) are this library's stated guesses: they round-trip with themselves, a
 * blob written by the real upstream cli will most likely be refused.  They stay provisional until a reference-produced blob exists.
 * The marshalling sources are absent submodules; csrc/wire.hip lists, field by field, what the reference's own files pin
 * (proof bytes and the head of the extended verification key: bin/cli/src/data.bin; 8-byte counts and 32-byte scalars:
 * protocol_exec.ipynb) and what is a stated guess.  All host-only except the proving-key pair.
 *   scalar vector   8-byte big-endian count | count x 32-byte big-endian Fr          (primary input, eid, sn, rt, voting result)
 *   G1 vector       8-byte big-endian count | count x 48-byte compressed G1          (the ciphertext)
 *   proof           A (48) | B (96) | C (48), ZCash compressed                       (data.bin[0:192))
 *   verification    head (4) | alpha_g1_beta_g2: 12 x 48-byte LITTLE-endian Fp (576, opaque: the pairing lives on the verifier side) |
 *   key, extended   gamma_g2 (96) | delta_g2 (96) | delta_g1 (48) | count (8) | gamma_ABC_g1 (count x 48) | gamma_g1 (48)
 *   proving key     alpha_g1 beta_g1 (96 each) beta_g2 (192) delta_g1 (96) delta_g2 (192), ZCash UNCOMPRESSED big-endian, then counted
 *   ("fast")        vectors A_query (x 96), B_query (x 192 + 96: G2 | G1), H_query (x 96), L_query (x 96)
 * *_from_blob with a NULL output and a count pointer returns the element count (size query).  VSP_ERR_ARG for malformed input. */
size_t vsp_fr_vector_blob_size(size_t count);
int vsp_fr_vector_to_blob(const uint64_t *vals, size_t count, uint8_t *out);
int vsp_fr_vector_from_blob(const uint8_t *blob, size_t len, uint64_t *vals_out, size_t capacity, size_t *count_out);
size_t vsp_g1_vector_blob_size(size_t count);
int vsp_g1_vector_to_blob(const uint64_t *pts, size_t count, uint8_t *out);
int vsp_g1_vector_from_blob(const uint8_t *blob, size_t len, int check_subgroup, uint64_t *pts_out, size_t capacity, size_t *count_out);
int vsp_proof_to_blob(const uint64_t A[12], const uint64_t B[24], const uint64_t C[12], uint8_t out[192]);
int vsp_proof_from_blob(const uint8_t blob[192], int check_subgroup, uint64_t A[12], uint64_t B[24], uint64_t C[12]);
size_t vsp_vk_blob_size(size_t n_abc);
int vsp_vk_to_blob(uint32_t head, const uint8_t gt[576], const uint64_t gamma_g2[24], const uint64_t delta_g2[24], const uint64_t delta_g1[12],
                   const uint64_t *gamma_abc_g1, size_t n_abc, const uint64_t gamma_g1[12], uint8_t *out);
int vsp_vk_from_blob(const uint8_t *blob, size_t len, int check_subgroup, uint32_t *head, uint8_t gt[576], uint64_t gamma_g2[24], uint64_t delta_g2[24],
                     uint64_t delta_g1[12], uint64_t *gamma_abc_g1, size_t capacity, size_t *n_abc, uint64_t gamma_g1[12]);
/* Proving key: the reference deserialises it inside its timed vote phase (common.hpp:1002-1004; main.cpp:446-456).  vsp_pk_from_blob copies
 * the raw bytes to the GPU once and converts them there (byte order, infinity flags, curve check, Montgomery form, 28-bit-limb table,
 * optionally the window multiples); the result is a resident key (vsp_keypair_pk) without gamma_ABC_g1.  precompute is the bit mask of
 * vsp_groth16_generate (bit 0 = A, both halves of B, L; H stays plain; bits 1..5 one query each).  Records must be well-formed ZCash
 * uncompressed records (no compression or sign flag, an infinity record all zero otherwise), on the curve; queries that keep the plain
 * layout go through the subgroup policy of "bases_check_subgroup".  PROVISIONAL layout (see above). */
size_t vsp_pk_blob_size(const vsp_keypair *kp);
int vsp_pk_to_blob(vsp_ctx *ctx, const vsp_keypair *kp, uint8_t *out);
vsp_keypair *vsp_pk_from_blob(vsp_ctx *ctx, const uint8_t *blob, size_t len, int precompute);

#ifdef __cplusplus
}
#endif
#endif /* VSP_H */
