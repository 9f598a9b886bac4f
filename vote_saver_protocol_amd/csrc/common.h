// Internal definitions shared by the translation units of libvsp_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <functional>
#include <map>
#include <string>
#include <vector>

#include "../../include/vsp.h"
#include <atomic>
#include <mutex>
#include <thread>
#include <vector>
#include "curve.h"

namespace vsp {

struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
};

// stage [s0, s1) split of one NTT (see ntt.hip)
struct NttTables {
    DevBuf fwd, inv;        // omega^j and omega^-j, j < 2^(log-1), Montgomery form
    unsigned log = 0;       // domain log the tables were generated for (serves every smaller domain)
    // two-level coset power tables for the cached coset generator
    DevBuf pw_lo_f, pw_hi_f, pw_lo_i, pw_hi_i;
    unsigned pw_log = 0;    // log_m the hi tables were sized for
    uint64_t pw_g[4] = {0, 0, 0, 0};
    bool pw_valid = false;
    // the same tables for the 9 x 29-bit butterflies (fr29.h): Montgomery form for R' = 2^261, three planes per table (16 + 16 + 4 bytes per entry)
    DevBuf fwd29, inv29, pw29[4];   // pw29: lo_f, hi_f, lo_i, hi_i
    unsigned log29 = 0;             // log the fwd29 / inv29 tables were converted for (0 = none)
    unsigned pw29_log = 0; uint64_t pw29_g[4] = {0, 0, 0, 0}; bool pw29_valid = false;
};

// digit-window geometry of one MSM
struct MsmGeom {
    unsigned c;         // window bits
    unsigned W;         // windows
    unsigned B;         // buckets per window = 2^(c-1)
    unsigned q0, q1, q2;  // bucket index bit split, q0+q1+q2 = c-1
    unsigned T;         // split threshold (max points per bucket part)
    size_t n;
    size_t G;           // Wr * B buckets in total
    unsigned Wr;        // bucket sets: W (one per window) or 1 (precomputed window multiples, all windows share one set)
    unsigned single;    // 1 in the shared-set mode
    uint32_t idx_stride, idx_first;   // shared-set mode: sorted entry of digit w of scalar i = w * idx_stride + idx_first + i
    unsigned sbits;     // bits of a scalar the windows must cover: 255, or 128 for the two halves of an endomorphism-split scalar
    unsigned lb;        // windows wider than 16 bits: the low bits of the bucket index that the second sort pass orders (c - 16); 0 otherwise
    unsigned fold;      // 255-bit scalars are read as min(k, r - k) < 2^254 with the sign flipped: one window fewer where c divides 255 (c = 15, 17)
    unsigned bd;        // slots per digit in the k_dimbits result layout: 8 (digits of up to 8 bits, three of them) or 12 (two digits of up to 12 bits)
    // a BATCH of K scalar vectors over one set of bases (round 4: vsp_groth16_prove_batch): vector k starts kstride elements after vector
    // k - 1, its Wk windows are the windows k Wk .. (k + 1) Wk - 1 of the W = K Wk the rest of the pipeline sees -- separate bucket sets per
    // (vector, window), the same base rows: the sort, the accumulation and the bucket reduction run ONCE, K times as wide.  K = 1: Wk = W.
    unsigned K, Wk;
    size_t kstride;
};
// reference to the precomputed window multiples of resident bases
struct MsmPre { size_t stride; size_t first; unsigned c; const void *table28; bool glv; };   // table28: the same table on 14 x 28-bit limbs (fp28.h), or null;
                                                                                             // glv: table28 holds (2^(cw) P_i, phi(2^(cw) P_i)) interleaved for the 128 / c windows of split scalars

// one in-flight MSM: its stream, device workspaces (grow only) and the pinned landing buffer of its window results
struct MsmWork {
    static constexpr size_t PINNED_BYTES = 256 * 1024;       // window results land here (25 records of 192 / 384 bytes per window at most); a batch grows it (pinned_cap)
    bool inited = false, own_stream = false, active = false, empty = false;
    bool dimbits = false;                  // layout of the window results of the last launch (msm_impl.inc k_dimbits / k_dimweight)
    hipStream_t stream = nullptr;          // the stream launches queue on: the slot's own one, or a borrowed one (msm_slot_use_stream)
    hipStream_t own = nullptr;             // the slot's own stream (slots 1..), created when first needed
    hipEvent_t ev0 = nullptr, ev1 = nullptr, done = nullptr, plan_ready = nullptr;
    DevBuf cnt, off, cursor, nsub, suboff, blocksum, sorted, heavy, counters, digits, blockhist, partbucket, perm, sizehist;
    DevBuf buckets, partials, dims, winres, medium, redo;
    DevBuf buckets28, partials28;          // G1: bucket sums in the 14 x 28-bit form (fp28.h XYZZ<Fp28>)
    DevBuf pairs_a, pairs_b, ms_h1, ms_h1s, ms_h2, ms_h2s;      // staged sort of large wide-window problems (msm_impl.inc k_ms_*)
    DevBuf tmp_sorted, tmp_lo, off_hi, cnt_hi;      // windows wider than 16 bits: the entries ordered by the high 15 bits of the bucket index, their low bits, the segment offsets
    DevBuf glv_scalars;                  // endomorphism split: 2n half-length scalars k1_i, k2_i (interleaved)
    bool glv = false;                    // this launch runs over the split scalars and the interleaved (P, phi(P)) table
    void *h_pinned = nullptr;
    size_t pinned_cap = 0;
    // vsp_msm_finish_jacobian_device: the Jacobian record leaves through a small pinned ring (REC_RING entries of 288 bytes) so that the
    // copy into the caller's device buffer is an asynchronous DMA; rec_ev[k] marks the copy out of entry k as done
    static constexpr unsigned REC_RING = 4;
    void *h_rec = nullptr; hipEvent_t rec_ev[REC_RING] = {nullptr, nullptr, nullptr, nullptr}; unsigned rec_idx = 0;
    void *h_census = nullptr;            // pinned: count of scalars that are neither 0 nor 1
    hipEvent_t census_done = nullptr;
    bool census_pending = false; size_t census_n = 0; const void *census_scalars = nullptr;
    bool check_pending = false;          // this launch's census (count + non-canonical flag) is to be read at finish
    size_t neff_cache_n = 0, neff_cache = 0;   // count of the last census over a vector of neff_cache_n scalars
    MsmGeom g;
    size_t n = 0, n_eff = 0;
};
static constexpr unsigned VSP_MSM_SLOTS = 6;
static constexpr int VSP_MSM_DENSE = -2;      // plan_from_slot value: the scalars are known to be dense, skip the 0/1 census
static constexpr int VSP_MSM_PLAN_ONLY = -3;  // plan_from_slot value: queue only the digit sort and the bucket plan (up to plan_ready); a later launch on the
                                              // SAME slot with plan_from_slot = that slot queues the accumulation and the reduction over it

}  // namespace vsp

struct vsp_ctx {
    int device = 0;
    hipStream_t stream = nullptr;       // stream in use
    hipStream_t own_stream = nullptr;   // created by vsp_create
    hipStream_t prove_streams[2] = {nullptr, nullptr};      // the prover's two witness chains (prover.hip), created on first use
    hipEvent_t ev0 = nullptr, ev1 = nullptr, ev_aux = nullptr;
    std::string err;
    int err_code = 0;                   // the code set_error last returned (callers that receive a null handle report it instead of guessing from the text)
    std::map<std::string, double> stats;
    std::map<std::string, long> opts;
    vsp::NttTables ntt;
    vsp::DevBuf ntt_scratch;
    vsp::DevBuf dom_scratch;            // step-domain transforms: the d / partial-sum vectors
    // MSM work slots (slot 0 runs on the context's stream; the others own a stream each)
    vsp::MsmWork msm_work[vsp::VSP_MSM_SLOTS];
    int slot_group[vsp::VSP_MSM_SLOTS] = {1, 1, 1, 1, 1, 1};
    bool lds_attr_set[2] = {false, false};
    bool ntt_attr_set = false;
    int ntt29_checked = 0;              // known-answer check of k_ntt29_pass: 0 not yet, 1 passed, -1 failed (8 x 32-bit kernel in use), 2 running
    vsp::DevBuf msm_scalars;
    void *h_fold = nullptr; size_t h_fold_cap = 0;      // pinned landing buffer of vsp_fold_jacobian_device (the ranks' records)
    vsp::DevBuf val_flag;               // one word: validation result of the last bases upload
    int fp28_checked[2] = {0, 0};       // known-answer check of the 28-bit-limb accumulation kernels, per group: 0 not yet, 1 passed, -1 failed (kernel disabled)
    // fixed-base tables (generator multiples), built lazily
    vsp::DevBuf fb_g1, fb_g2, fb_tmp, fb_pre;
    // prover workspaces
    vsp::DevBuf pr_z, pr_a, pr_b, pr_c, pr_h, pr_pack;
    vsp::DevBuf pr_bz, pr_babc, pr_bh;      // vsp_groth16_prove_batch: [K][num_vars + 1], [K][3][m], [K][m]
    // a proof in flight between vsp_groth16_prove_launch and _finish (one per context)
    struct { bool active = false; const vsp_pk *pk = nullptr; uint64_t r[4], s[4], P1[12], r_enc[4]; bool has_saver = false; } prove;
    // a BATCH of proofs in flight between vsp_groth16_prove_batch_launch and _finish (one per context)
    struct { bool active = false; const vsp_pk *pk = nullptr; size_t count = 0; std::vector<uint64_t> r, s; } prove_batch;
};

struct vsp_bases {
    int group = 1;          // 1 = G1, 2 = G2
    size_t n = 0;
    void *d = nullptr;      // device array of Affine<Fp> / Affine<Fp2>, Montgomery form; with pre_c != 0 it is the table
                            // [W][n]: slice w holds 2^(pre_c * w) * P  (vsp_bases_precompute)
    unsigned pre_c = 0;
    int in_subgroup = 0;    // 1: every point satisfies phi(P) = lambda P (checked at upload, or the library's own multiples of a generator); -1: the check
                            // found a point that does not; 0: not checked.  The endomorphism layout below needs 1 (or option "msm_glv" = 2)
    bool glv = false;       // d28 holds (P_i, phi(P_i)) interleaved, phi(x, y) = (beta x, y) = lambda * P (the curve's endomorphism): a scalar
                            // k = k1 + k2 lambda then needs windows over 128 bits only.  Plain bases: 2n rows, half the bucket sets to reduce.
                            // Window multiples (pre_split): 2n rows for each of the ceil(128 / pre_c) windows -- the split over ONE bucket set
    bool pre_split = false; // vsp_bases_precompute_split: the table of window multiples is meant for dense scalars and carries the endomorphism rows
    void *d28 = nullptr;    // the same array (or table) once more on 14 x 28-bit limbs (fp28.h: 112-byte rows G1, 224-byte rows G2) for the accumulation kernel
};

// math::evaluation_domain<Fr>: the basic radix-2 domain (step = 0, m = big_m = 2^log_big) or the step radix-2 domain
// (m = big_m + small_m, both powers of two, small_m < big_m) that make_evaluation_domain(min_size) selects
struct vsp_domain {
    size_t m = 0, big_m = 0, small_m = 0;
    unsigned log_big = 0, log_small = 0;
    int step = 0;
    // divide_by_z_on_coset for the coset generator 7: 1 / Z(7 x_i).  Basic domain: one constant.  Step domain: a table of period
    // compr = big_m / small_m over the first big_m elements (device, Montgomery form) and one constant for the last small_m.
    vsp::DevBuf zinv;
    vsp::HFr zinv_const;
};

struct vsp_r1cs {
    size_t num_constraints = 0, num_inputs = 0, num_vars = 0;
    vsp_domain dom;          // make_evaluation_domain(num_constraints + num_inputs + 1)
    uint32_t *rp[3] = {nullptr, nullptr, nullptr};
    uint32_t *ci[3] = {nullptr, nullptr, nullptr};
    void *co[3] = {nullptr, nullptr, nullptr};      // Fr Montgomery
    // column-major copy (for the generator's per-variable accumulation): col_ptr [num_vars+2], row index, coefficient
    uint32_t *cp[3] = {nullptr, nullptr, nullptr};
    uint32_t *ri[3] = {nullptr, nullptr, nullptr};
    void *cot[3] = {nullptr, nullptr, nullptr};     // Fr Montgomery
};

struct vsp_keypair {
    vsp_pk *pk = nullptr;
    vsp_bases *q[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};   // A, B_g1, B_g2, H, L, gamma_ABC_g1
    uint64_t alpha_g1[12], beta_g1[12], delta_g1[12], beta_g2[24], delta_g2[24], gamma_g2[24];
    uint64_t gamma_g1[12];          // extended verification key (the SAVER key generation needs gamma in G1)
};

struct vsp_pk {
    vsp::Affine<vsp::HFp> alpha_g1, beta_g1, delta_g1;     // host, Montgomery
    vsp::Affine<vsp::HFp2> beta_g2, delta_g2;
    const vsp_bases *A = nullptr, *B1 = nullptr, *B2 = nullptr, *H = nullptr, *L = nullptr;
    // every proof multiplies delta (G1: by r, s, r s; G2: by s): fixed-base tables of d 2^(8 w) delta, d = 1..255, w = 0..31, built by the
    // first proof over this key (prover.hip delta_tables; contexts on several threads share a key: built once, under the mutex)
    mutable std::mutex tab_mu;
    mutable std::atomic<bool> tab_ready{false};
    mutable std::vector<vsp::XYZZ<vsp::HFp>> tab1;
    mutable std::vector<vsp::XYZZ<vsp::HFp2>> tab2;
};

namespace vsp {

int set_hip_error(vsp_ctx *ctx, hipError_t e, const char *what, const char *file, int line);
// f(0) .. f(n - 1) on up to `max_threads` host threads (the host steps of a BATCH: the Horner chains over the window results of K
// multi-exponentiations and the assembly of K proofs are independent pieces of a few hundred group operations each)
template <class Fn> inline void host_parallel_for(size_t n, Fn f, unsigned max_threads = 16) {
    unsigned hw = std::thread::hardware_concurrency(); if (hw == 0) hw = 4;
    size_t T = n < hw ? n : hw; if (T > max_threads) T = max_threads;
    if (T <= 1) { for (size_t i = 0; i < n; i++) f(i); return; }
    std::vector<std::thread> th;
    th.reserve(T - 1);
    for (size_t t = 1; t < T; t++) th.emplace_back([=]() { for (size_t i = t; i < n; i += T) f(i); });
    for (size_t i = 0; i < n; i += T) f(i);
    for (auto &x : th) x.join();
}
int set_error(vsp_ctx *ctx, int code, const char *msg);
int ensure(vsp_ctx *ctx, DevBuf &b, size_t bytes);

#define VSP_HIP(call)                                                                        \
    do {                                                                                     \
        hipError_t e_ = (call);                                                              \
        if (e_ != hipSuccess) return vsp::set_hip_error(ctx, e_, #call, __FILE__, __LINE__); \
    } while (0)
#define VSP_TRY(call)                \
    do {                             \
        int rc_ = (call);            \
        if (rc_ != VSP_OK) return rc_; \
    } while (0)
#define VSP_LAUNCH_CHECK() VSP_HIP(hipGetLastError())

// ---- internal entry points (each implemented in its own .hip) ----
int ntt_device(vsp_ctx *ctx, Fr *d_a, unsigned log_m, int inverse, const uint64_t *coset_g, const HFr *extra_scale);
bool ntt29_in_use(vsp_ctx *ctx);
int ntt_device_strided(vsp_ctx *ctx, Fr *base, unsigned count, size_t stride, unsigned log_m, int inverse, const uint64_t *coset_g, const HFr *extra_scale);
int ntt_device_fused_abc_strided(vsp_ctx *ctx, const Fr *d_a, size_t off_b, size_t off_c, size_t in_stride, Fr *d_h, size_t out_stride, unsigned count, unsigned log_m,
                                 int inverse, const uint64_t *coset_g, const HFr *extra_scale);
int witness_map_device_batch(vsp_ctx *ctx, Fr *abc, unsigned count, const vsp_domain *d, Fr *dH);
int ntt_device_batch(vsp_ctx *ctx, Fr *const *d_a, unsigned count, unsigned log_m, int inverse, const uint64_t *coset_g, const HFr *extra_scale);
int ntt_device_fused_abc(vsp_ctx *ctx, const Fr *d_a, const Fr *d_b, const Fr *d_c, Fr *d_h, unsigned log_m, int inverse, const uint64_t *coset_g, const HFr *extra_scale);
int ntt_ensure_twiddles(vsp_ctx *ctx, unsigned log_m);
void ntt_selfcheck_once(vsp_ctx *ctx);                          // known-answer check of the 29-bit butterfly kernel, before any table set-up
int ntt_ensure_coset_tables(vsp_ctx *ctx, unsigned log_m, const uint64_t *g4);
// evaluation domains (domain.hip)
int domain_init(vsp_ctx *ctx, vsp_domain *d, size_t min_size);      // make_evaluation_domain's choice + the coset divisors
void domain_release(vsp_domain *d);
void domain_basic(vsp_domain *d, unsigned log_m);                   // the basic radix-2 domain of size 2^log_m (log_m = 0 allowed)
int fr_from_mont_device(vsp_ctx *ctx, Fr *d_a, size_t n);
int domain_fft_device(vsp_ctx *ctx, const vsp_domain *d, Fr *d_a, int inverse, const uint64_t *coset_g, const HFr *extra_scale);
int domain_divide_by_z_device(vsp_ctx *ctx, const vsp_domain *d, Fr *d_p);
int domain_lagrange_device(vsp_ctx *ctx, const vsp_domain *d, const HFr &t, Fr *d_u /* m, Montgomery form */);
HFr domain_vanishing(const vsp_domain *d, const HFr &t);
HFr domain_element(const vsp_domain *d, size_t idx);
int witness_map_device(vsp_ctx *ctx, Fr *dA, Fr *dB, Fr *dC, const vsp_domain *d, Fr *dH);

// MSM on device-resident Montgomery bases; result as host XYZZ (Montgomery, 64-bit limbs)
int msm_g1_device(vsp_ctx *ctx, const G1Affine *d_bases, const Fr *d_scalars, size_t n, XYZZ<HFp> *out);
int msm_g2_device(vsp_ctx *ctx, const G2Affine *d_bases, const Fr *d_scalars, size_t n, XYZZ<HFp2> *out);
int msm_g1_launch(vsp_ctx *ctx, unsigned slot, const G1Affine *d_bases, const Fr *d_scalars, size_t n, int plan_from_slot, const MsmPre *pre,
                  const void *plain_table28 = nullptr, bool glv = false);   // plain bases (pre == null): the same points as Affine28 (glv: 2n rows, P and phi(P) interleaved), or null
int msm_g1_precompute(vsp_ctx *ctx, G1Affine *table, size_t n, unsigned c);
int msm_g2_precompute(vsp_ctx *ctx, G2Affine *table, size_t n, unsigned c);
int msm_g1_table28(vsp_ctx *ctx, const G1Affine *table, size_t count, void *d_out /* count (glv: 2 count) rows of 128 bytes */, bool glv);
int msm_g2_table28(vsp_ctx *ctx, const G2Affine *table, size_t count, void *d_out /* count (glv: 2 count) rows of 256 bytes */, bool glv);
int msm_g1_finish(vsp_ctx *ctx, unsigned slot, XYZZ<HFp> *out);
int msm_g2_launch(vsp_ctx *ctx, unsigned slot, const G2Affine *d_bases, const Fr *d_scalars, size_t n, int plan_from_slot, const MsmPre *pre,
                  const void *plain_table28 = nullptr, bool glv = false);
int msm_g2_finish(vsp_ctx *ctx, unsigned slot, XYZZ<HFp2> *out);
// a batch of scalar vectors over one set of PLAIN bases (MsmGeom.K): vector k at d_scalars + k * stride; one result per vector
int msm_g1_launch_batch(vsp_ctx *ctx, unsigned slot, const G1Affine *d_bases, const Fr *d_scalars, size_t n, unsigned batch, size_t stride, bool dense, const void *plain_table28, bool glv, int plan_from_slot = -1, const MsmPre *pre = nullptr);
int msm_g1_finish_batch(vsp_ctx *ctx, unsigned slot, XYZZ<HFp> *out, unsigned batch);
int msm_g2_launch_batch(vsp_ctx *ctx, unsigned slot, const G2Affine *d_bases, const Fr *d_scalars, size_t n, unsigned batch, size_t stride, bool dense, const void *plain_table28, bool glv, int plan_from_slot = -1, const MsmPre *pre = nullptr);
int msm_g2_finish_batch(vsp_ctx *ctx, unsigned slot, XYZZ<HFp2> *out, unsigned batch);
// a finish in two halves: the wait (context state: caller's thread) and the fold of the window results (pure host arithmetic over the slot: any thread)
int msm_g1_finish_wait(vsp_ctx *ctx, unsigned slot, bool *empty);
void msm_g1_fold(vsp_ctx *ctx, unsigned slot, XYZZ<HFp> *out);
int msm_g2_finish_wait(vsp_ctx *ctx, unsigned slot, bool *empty);
void msm_g2_fold(vsp_ctx *ctx, unsigned slot, XYZZ<HFp2> *out);
int launch_on_bases(vsp_ctx *ctx, unsigned slot, const vsp_bases *bases, size_t first, size_t n, const Fr *d_scalars, int plan_from_slot);
int launch_on_bases_batch(vsp_ctx *ctx, unsigned slot, const vsp_bases *bases, size_t first, size_t n, const Fr *d_scalars, unsigned batch, size_t stride, bool dense, int plan_from_slot = -1);
int msm_slot_stream(vsp_ctx *ctx, unsigned slot, hipStream_t *out);
int msm_slot_census(vsp_ctx *ctx, unsigned slot, const Fr *d_scalars, size_t n);
void msm_free_slots(vsp_ctx *ctx);
int msm_slot_use_stream(vsp_ctx *ctx, unsigned slot, hipStream_t stream_or_null);
int msm_make_slot_stream(vsp_ctx *ctx, hipStream_t *out);
void msm_drain_slots(vsp_ctx *ctx);
// d_flag: one device word, zeroed by the caller; bit 0 = coordinate >= p, bit 1 = point off the curve (only when check_curve)
int bases_to_mont_g1(vsp_ctx *ctx, const void *d_canon, G1Affine *d_out, size_t n, int check_curve, uint32_t *d_flag);
int bases_to_mont_g2(vsp_ctx *ctx, const void *d_canon, G2Affine *d_out, size_t n, int check_curve, uint32_t *d_flag);
// raises bit 2 of *d_flag when some point fails phi(P) = lambda P (the endomorphism split's precondition; msm_impl.inc k_subgroup_check)
int subgroup_check_g1(vsp_ctx *ctx, const G1Affine *d_mont, size_t n, uint32_t *d_flag);
int msm_diag_clock(vsp_ctx *ctx, int reset, double *ghz, double *waves);
int ntt_diag_clock(vsp_ctx *ctx, int reset, double *ghz, double *waves);
int subgroup_check_g2(vsp_ctx *ctx, const G2Affine *d_mont, size_t n, uint32_t *d_flag);
// resident bases from canonical points; trust: BASES_CALLER = caller data (validated; the split only after the subgroup check),
// BASES_OWN = points this library computed as multiples of a generator (in the subgroup by construction: no check),
// BASES_TRANSIENT = bases of one host-buffer call, or bases about to get window multiples (no split, so no check: exact for any curve point)
enum { BASES_CALLER = 0, BASES_OWN = 1, BASES_TRANSIENT = 2 };
vsp_bases *bases_create(vsp_ctx *ctx, int group, const void *src, bool src_on_device, size_t n, int trust);
int fixed_base_mul_g1(vsp_ctx *ctx, const Fr *d_scalars, size_t n, void *d_out);
int fixed_base_mul_g2(vsp_ctx *ctx, const Fr *d_scalars, size_t n, void *d_out);
int upload_power_tables(vsp_ctx *ctx, const HFr &base, size_t hi_count, DevBuf &lo, DevBuf &hi);
HFr host_omega(unsigned log_m);

// canonical <-> host Montgomery helpers
template <class F> inline F host_load_canon(const uint64_t *p) { F t; memcpy(&t, p, sizeof(F)); return to_mont(t); }
template <class F> inline void host_store_canon(uint64_t *p, const F &m) { F t = from_mont(m); memcpy(p, &t, sizeof(F)); }
inline Affine<HFp> host_load_g1(const uint64_t *p) { Affine<HFp> a; a.x = host_load_canon<HFp>(p); a.y = host_load_canon<HFp>(p + 6); return a; }
inline Affine<HFp2> host_load_g2(const uint64_t *p) {
    Affine<HFp2> a;
    a.x.c0 = host_load_canon<HFp>(p); a.x.c1 = host_load_canon<HFp>(p + 6);
    a.y.c0 = host_load_canon<HFp>(p + 12); a.y.c1 = host_load_canon<HFp>(p + 18);
    return a;
}
inline void host_store_g1(uint64_t *p, const Affine<HFp> &a) { host_store_canon(p, a.x); host_store_canon(p + 6, a.y); }
inline void host_store_g2(uint64_t *p, const Affine<HFp2> &a) {
    host_store_canon(p, a.x.c0); host_store_canon(p + 6, a.x.c1); host_store_canon(p + 12, a.y.c0); host_store_canon(p + 18, a.y.c1);
}

static inline unsigned ceil_log2(size_t n) { unsigned l = 0; while (((size_t)1 << l) < n) l++; return l; }

}  // namespace vsp
