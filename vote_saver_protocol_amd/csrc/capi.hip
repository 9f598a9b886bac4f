// extern "C" surface of libvsp_hip.so (declared in include/vsp.h): context, device memory, MSM / NTT /
// witness_map entry points, Jacobian record folding and the ZCash point compression.
#include "common.h"
#include "fp28.h"
#include "lane_view.h"

namespace vsp {

int set_hip_error(vsp_ctx *ctx, hipError_t e, const char *what, const char *file, int line) {
    char buf[512];
    snprintf(buf, sizeof buf, "%s failed: %s (%s:%d)", what, hipGetErrorString(e), file, line);
    if (ctx) ctx->err = buf;
    return VSP_ERR_HIP;
}
int set_error(vsp_ctx *ctx, int code, const char *msg) {
    if (ctx) { ctx->err = msg; ctx->err_code = code; }
    return code;
}
int ensure(vsp_ctx *ctx, DevBuf &b, size_t bytes) {
    if (bytes == 0) bytes = 16;
    if (b.cap >= bytes && b.p) return VSP_OK;
    if (b.p) { VSP_HIP(hipStreamSynchronize(ctx->stream)); hipFree(b.p); b.p = nullptr; b.cap = 0; }
    size_t want = bytes + bytes / 8;          // a little headroom so a slightly larger call does not realloc
    hipError_t e = hipMalloc(&b.p, want);
    if (e != hipSuccess) { b.p = nullptr; char m[128]; snprintf(m, sizeof m, "hipMalloc(%zu) failed: %s", want, hipGetErrorString(e)); return set_error(ctx, VSP_ERR_NOMEM, m); }
    b.cap = want;
    return VSP_OK;
}

static void free_buf(DevBuf &b) { if (b.p) hipFree(b.p); b.p = nullptr; b.cap = 0; }

static const uint64_t HALF_P[6] = {0xdcff7fffffffd555ULL, 0x0f55ffff58a9ffffULL, 0xb39869507b587b12ULL, 0xb23ba5c279c2895fULL, 0x258dd3db21a5d66bULL, 0x0d0088f51cbff34dULL};
static bool fp_lex_larger(const uint64_t *y) {      // y > (p-1)/2
    for (int i = 5; i >= 0; i--) { if (y[i] > HALF_P[i]) return true; if (y[i] < HALF_P[i]) return false; }
    return false;
}
static bool limbs_zero(const uint64_t *a, int n) { uint64_t o = 0; for (int i = 0; i < n; i++) o |= a[i]; return o == 0; }
static void be48(uint8_t *out, const uint64_t *l) { for (int i = 0; i < 6; i++) for (int b = 0; b < 8; b++) out[47 - (i * 8 + b)] = (uint8_t)(l[i] >> (8 * b)); }

template <class F, class HF>
static int finish_affine(const XYZZ<HF> &acc, uint64_t *out_affine, int *out_is_inf);
template <> int finish_affine<Fp, HFp>(const XYZZ<HFp> &acc, uint64_t *out_affine, int *out_is_inf) {
    Affine<HFp> a = xyzz_to_affine(acc);
    if (out_affine) host_store_g1(out_affine, a);
    if (out_is_inf) *out_is_inf = is_inf(acc) ? 1 : 0;
    return VSP_OK;
}
template <> int finish_affine<Fp2, HFp2>(const XYZZ<HFp2> &acc, uint64_t *out_affine, int *out_is_inf) {
    Affine<HFp2> a = xyzz_to_affine(acc);
    if (out_affine) host_store_g2(out_affine, a);
    if (out_is_inf) *out_is_inf = is_inf(acc) ? 1 : 0;
    return VSP_OK;
}

// elementwise field operations on canonical values (diagnostic entry point vsp_selftest_field)
template <class F> __global__ __launch_bounds__(64) void k_selftest_field(int op, const F *a, const F *b, F *out, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    F x = a[i], y = b[i], r;
    switch (op) {
        case 0: r = from_mont(mul(to_mont(x), to_mont(y))); break;
        case 1: r = add(x, y); break;
        case 2: r = sub(x, y); break;
        case 3: r = from_mont(sqr(to_mont(x))); break;
        case 4: r = mul(x, to_mont(y)); break;          // canonical x Montgomery -> canonical (the NTT butterfly product)
        default: r = from_mont(inv(to_mont(x))); break;
    }
    out[i] = r;
}

// the 14 x 28-bit lazy field of the G1 accumulation (fp28.h) against the 12 x 32-bit one, on canonical inputs:
//   op 6 round trip, 7 product, 8 (x - y)^2 through the K32 subtraction and a carry pass, 9 x - 3y through the K8 subtraction of a
//   lazy sum, 10 x (y - x) with a loose operand, 11 -y x through the negation used for signed digits, 12 x (y - x) - y x as one dual product
__global__ __launch_bounds__(64) void k_selftest_fp28(int op, const Fp *a, const Fp *b, Fp *out, size_t n) {
#if defined(__HIP_DEVICE_COMPILE__)
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Fp28 x = fp_to_fp28(to_mont(a[i])), y = fp_to_fp28(to_mont(b[i])), r;
    switch (op) {
        case 6: r = x; break;
        case 7: r = mul28(x, y); break;
        case 8: { Fp28 d = norm28(sub28(x, FP28_K32_L1, y)); r = sqr28(d); } break;          // through the dedicated squaring routine
        case 9: { Fp28 s3; for (int k = 0; k < 14; k++) s3.l[k] = y.l[k] + 2u * y.l[k]; r = norm28(sub28(x, FP28_K8_L4, s3)); } break;
        case 10: r = mul28(x, sub28(y, FP28_K32_L1, x)); break;
        case 11: r = mul28(neg28(FP28_K8_L1, y), x); break;
        default: r = mul28x2(x, sub28(y, FP28_K32_L1, x), neg28(FP28_K32_L1, y), x); break;    // op 12: x (y - x) - y x through the dual product
    }
    out[i] = from_mont(fp28_to_fp(r));
#endif
}

// The full addition of two bucket sums (curve.h / fp28.h xyzz_add) on its own, in each of the four forms the merges and the bucket
// reduction run it in (diagnostic entry point vsp_selftest_xyzz_add): canonical X, Y, ZZ, ZZZ in, canonical out.  Lanes of one wave take
// different paths (ordinary sum, doubling, cancellation, infinity on either side) as the test orders its cases -- the divergence the
// kernels meet.  form 0: 12 x 32-bit limbs; 1: 14 x 28-bit lazy limbs.  G2 runs on lane pairs (two lanes per point).
__device__ __forceinline__ Fp &fp_of(Fp &x) { return x; }
__device__ __forceinline__ Fp &fp_of(Fp2L &x) { return x.v; }
__device__ __forceinline__ Fp28 &fp28_of(Fp28 &x) { return x; }
__device__ __forceinline__ Fp28 &fp28_of(Fp28L &x) { return x.v; }
template <class M, class F28> __global__ __launch_bounds__(64) void k_selftest_xyzz_add(int form, const XYZZ<M> *a, const XYZZ<M> *b, XYZZ<M> *out, size_t n) {
#if defined(__HIP_DEVICE_COMPILE__)
    using LV = LaneView<M>; using E = typename LV::E; using E28 = typename LaneView<F28>::E;
    const size_t i = gid<M>();
    if (i >= n) return;                                           // both lanes of a pair leave together
    XYZZ<E> x = LV::load(&a[i]), y = LV::load(&b[i]);
    for (XYZZ<E> *v : {&x, &y}) { fp_of(v->X) = to_mont(fp_of(v->X)); fp_of(v->Y) = to_mont(fp_of(v->Y)); fp_of(v->ZZ) = to_mont(fp_of(v->ZZ)); fp_of(v->ZZZ) = to_mont(fp_of(v->ZZZ)); }
    if (form == 0) xyzz_add(x, y);
    else {
        XYZZ<E28> x28, y28;                                       // zero stays zero: infinity is all-zero in both forms
        fp28_of(x28.X) = fp_to_fp28(fp_of(x.X)); fp28_of(x28.Y) = fp_to_fp28(fp_of(x.Y)); fp28_of(x28.ZZ) = fp_to_fp28(fp_of(x.ZZ)); fp28_of(x28.ZZZ) = fp_to_fp28(fp_of(x.ZZZ));
        fp28_of(y28.X) = fp_to_fp28(fp_of(y.X)); fp28_of(y28.Y) = fp_to_fp28(fp_of(y.Y)); fp28_of(y28.ZZ) = fp_to_fp28(fp_of(y.ZZ)); fp28_of(y28.ZZZ) = fp_to_fp28(fp_of(y.ZZZ));
        xyzz_add(x28, y28);
        fp_of(x.X) = fp28_to_fp(fp28_of(x28.X)); fp_of(x.Y) = fp28_to_fp(fp28_of(x28.Y)); fp_of(x.ZZ) = fp28_to_fp(fp28_of(x28.ZZ)); fp_of(x.ZZZ) = fp28_to_fp(fp28_of(x28.ZZZ));
    }
    fp_of(x.X) = from_mont(fp_of(x.X)); fp_of(x.Y) = from_mont(fp_of(x.Y)); fp_of(x.ZZ) = from_mont(fp_of(x.ZZ)); fp_of(x.ZZZ) = from_mont(fp_of(x.ZZZ));
    LV::store(&out[i], x);
#endif
}

}  // namespace vsp

using namespace vsp;

extern "C" {

int vsp_selftest_field(vsp_ctx *ctx, int field, int op, const uint64_t *a, const uint64_t *b, uint64_t *out, size_t n) {
    if (!ctx) return VSP_ERR_ARG;
    if (!a || !b || !out || (field != 0 && field != 1) || op < 0 || op > 12 || (op > 5 && field != 0)) return set_error(ctx, VSP_ERR_ARG, "selftest: bad argument");
    VSP_HIP(hipSetDevice(ctx->device));
    size_t esz = field == 0 ? sizeof(Fp) : sizeof(Fr);
    DevBuf da, db, dc;
    int rc = ensure(ctx, da, n * esz); if (rc == VSP_OK) rc = ensure(ctx, db, n * esz); if (rc == VSP_OK) rc = ensure(ctx, dc, n * esz);
    if (rc == VSP_OK) {
        hipMemcpyAsync(da.p, a, n * esz, hipMemcpyHostToDevice, ctx->stream);
        hipMemcpyAsync(db.p, b, n * esz, hipMemcpyHostToDevice, ctx->stream);
        unsigned blocks = (unsigned)((n + 63) / 64);
        if (field == 0 && op > 5) hipLaunchKernelGGL(k_selftest_fp28, dim3(blocks), dim3(64), 0, ctx->stream, op, (const Fp *)da.p, (const Fp *)db.p, (Fp *)dc.p, n);
        else if (field == 0) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_selftest_field<Fp>), dim3(blocks), dim3(64), 0, ctx->stream, op, (const Fp *)da.p, (const Fp *)db.p, (Fp *)dc.p, n);
        else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_selftest_field<Fr>), dim3(blocks), dim3(64), 0, ctx->stream, op, (const Fr *)da.p, (const Fr *)db.p, (Fr *)dc.p, n);
        hipMemcpyAsync(out, dc.p, n * esz, hipMemcpyDeviceToHost, ctx->stream);
        if (hipStreamSynchronize(ctx->stream) != hipSuccess || hipGetLastError() != hipSuccess) rc = set_error(ctx, VSP_ERR_HIP, "selftest: kernel failed");
    }
    free_buf(da); free_buf(db); free_buf(dc);
    return rc;
}

int vsp_selftest_xyzz_add(vsp_ctx *ctx, int group, int form, const uint64_t *a, const uint64_t *b, uint64_t *out, size_t n) {
    if (!ctx) return VSP_ERR_ARG;
    if (!a || !b || !out || (group != 1 && group != 2) || (form != 0 && form != 1)) return set_error(ctx, VSP_ERR_ARG, "selftest: bad argument");
    VSP_HIP(hipSetDevice(ctx->device));
    const size_t esz = group == 1 ? sizeof(XYZZ<Fp>) : sizeof(XYZZ<Fp2>);
    DevBuf da, db, dc;
    int rc = ensure(ctx, da, n * esz); if (rc == VSP_OK) rc = ensure(ctx, db, n * esz); if (rc == VSP_OK) rc = ensure(ctx, dc, n * esz);
    if (rc == VSP_OK) {
        hipMemcpyAsync(da.p, a, n * esz, hipMemcpyHostToDevice, ctx->stream);
        hipMemcpyAsync(db.p, b, n * esz, hipMemcpyHostToDevice, ctx->stream);
        if (group == 1) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_selftest_xyzz_add<Fp, Fp28>), dim3((unsigned)((n + 63) / 64)), dim3(64), 0, ctx->stream, form, (const XYZZ<Fp> *)da.p, (const XYZZ<Fp> *)db.p, (XYZZ<Fp> *)dc.p, n);
        else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_selftest_xyzz_add<Fp2, Fp2x28>), dim3((unsigned)((2 * n + 63) / 64)), dim3(64), 0, ctx->stream, form, (const XYZZ<Fp2> *)da.p, (const XYZZ<Fp2> *)db.p, (XYZZ<Fp2> *)dc.p, n);
        hipMemcpyAsync(out, dc.p, n * esz, hipMemcpyDeviceToHost, ctx->stream);
        if (hipStreamSynchronize(ctx->stream) != hipSuccess || hipGetLastError() != hipSuccess) rc = set_error(ctx, VSP_ERR_HIP, "selftest: kernel failed");
    }
    free_buf(da); free_buf(db); free_buf(dc);
    return rc;
}

vsp_ctx *vsp_create(int device_ordinal) {
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || device_ordinal < 0 || device_ordinal >= count) return nullptr;
    if (hipSetDevice(device_ordinal) != hipSuccess) return nullptr;
    vsp_ctx *ctx = new vsp_ctx();
    ctx->device = device_ordinal;
    if (hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking) != hipSuccess) { delete ctx; return nullptr; }
    ctx->stream = ctx->own_stream;
    if (hipEventCreate(&ctx->ev0) != hipSuccess || hipEventCreate(&ctx->ev1) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev_aux, hipEventDisableTiming) != hipSuccess) { hipStreamDestroy(ctx->own_stream); delete ctx; return nullptr; }
    // The prover's two witness chains get their streams NOW, ahead of every work slot's: the runtime deals its hardware queues (4 per
    // priority unless GPU_MAX_HW_QUEUES says otherwise) to streams in creation order, and two chains that land on one queue run one after
    // the other (7.4 instead of 6.6 ms per proof in a process with a dozen streams).  Created first, they get a queue each whatever else
    // the process creates later -- the library does not depend on that environment variable.
    for (int k = 0; k < 2; k++) if (msm_make_slot_stream(ctx, &ctx->prove_streams[k]) != VSP_OK) ctx->prove_streams[k] = nullptr;
    ctx->err.clear();
    return ctx;
}

void vsp_destroy(vsp_ctx *ctx) {
    if (!ctx) return;
    hipSetDevice(ctx->device);
    if (ctx->prove.active || ctx->prove_batch.active) msm_drain_slots(ctx);      // a proof launched and never finished: wait its kernels out before their buffers go
    hipStreamSynchronize(ctx->stream);
    DevBuf *bufs[] = {&ctx->ntt.fwd, &ctx->ntt.inv, &ctx->ntt.pw_lo_f, &ctx->ntt.pw_hi_f, &ctx->ntt.pw_lo_i, &ctx->ntt.pw_hi_i, &ctx->ntt_scratch, &ctx->dom_scratch,
                      &ctx->ntt.fwd29, &ctx->ntt.inv29, &ctx->ntt.pw29[0], &ctx->ntt.pw29[1], &ctx->ntt.pw29[2], &ctx->ntt.pw29[3],
                      &ctx->msm_scalars, &ctx->val_flag, &ctx->fb_g1, &ctx->fb_g2, &ctx->fb_tmp, &ctx->fb_pre,
                      &ctx->pr_z, &ctx->pr_a, &ctx->pr_b, &ctx->pr_c, &ctx->pr_h, &ctx->pr_pack, &ctx->pr_bz, &ctx->pr_babc, &ctx->pr_bh};
    for (DevBuf *b : bufs) free_buf(*b);
    msm_free_slots(ctx);
    if (ctx->h_fold) hipHostFree(ctx->h_fold);
    hipEventDestroy(ctx->ev0); hipEventDestroy(ctx->ev1); hipEventDestroy(ctx->ev_aux);
    for (hipStream_t ps : ctx->prove_streams) if (ps) hipStreamDestroy(ps);
    hipStreamDestroy(ctx->own_stream);
    delete ctx;
}

const char *vsp_last_error(vsp_ctx *ctx) { return ctx ? ctx->err.c_str() : "null context"; }

int vsp_set_stream(vsp_ctx *ctx, void *hip_stream) {
    if (!ctx) return VSP_ERR_ARG;
    VSP_HIP(hipStreamSynchronize(ctx->stream));
    ctx->stream = hip_stream ? (hipStream_t)hip_stream : ctx->own_stream;
    return VSP_OK;
}
int vsp_synchronize(vsp_ctx *ctx) {
    if (!ctx) return VSP_ERR_ARG;
    VSP_HIP(hipStreamSynchronize(ctx->stream));
    return VSP_OK;
}
double vsp_get_stat(vsp_ctx *ctx, const char *name) {
    if (!ctx || !name) return 0.0;
    if (!strcmp(name, "runtime_hw_queues_env")) { const char *q = getenv("GPU_MAX_HW_QUEUES"); return q ? atof(q) : 0.0; }      // include/vsp.h "Runtime environment"
    auto it = ctx->stats.find(name);
    return it == ctx->stats.end() ? 0.0 : it->second;
}
void vsp_stats_reset(vsp_ctx *ctx) { if (ctx) ctx->stats.clear(); }
int vsp_diag_clock(vsp_ctx *ctx, int reset, double *ghz_out, double *waves_out) {
    if (!ctx) return VSP_ERR_ARG;
    VSP_HIP(hipSetDevice(ctx->device));
    return msm_diag_clock(ctx, reset, ghz_out, waves_out);
}
int vsp_diag_clock_ntt(vsp_ctx *ctx, int reset, double *ghz_out, double *waves_out) {
    if (!ctx) return VSP_ERR_ARG;
    VSP_HIP(hipSetDevice(ctx->device));
    return ntt_diag_clock(ctx, reset, ghz_out, waves_out);
}
int vsp_set_option(vsp_ctx *ctx, const char *name, long value) {
    if (!ctx || !name) return VSP_ERR_ARG;
    ctx->opts[name] = value;
    return VSP_OK;
}

void *vsp_dmalloc(vsp_ctx *ctx, size_t bytes) {
    if (!ctx) return nullptr;
    hipSetDevice(ctx->device);
    void *p = nullptr;
    if (hipMalloc(&p, bytes ? bytes : 16) != hipSuccess) { set_error(ctx, VSP_ERR_NOMEM, "vsp_dmalloc: hipMalloc failed"); return nullptr; }
    return p;
}
void vsp_dfree(vsp_ctx *ctx, void *dptr) { if (ctx && dptr) { hipSetDevice(ctx->device); hipStreamSynchronize(ctx->stream); hipFree(dptr); } }
int vsp_h2d(vsp_ctx *ctx, void *dst_dev, const void *src_host, size_t bytes) {
    if (!ctx || (!dst_dev && bytes) || (!src_host && bytes)) return VSP_ERR_ARG;
    VSP_HIP(hipSetDevice(ctx->device));
    VSP_HIP(hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, ctx->stream));
    VSP_HIP(hipStreamSynchronize(ctx->stream));
    return VSP_OK;
}
int vsp_d2h(vsp_ctx *ctx, void *dst_host, const void *src_dev, size_t bytes) {
    if (!ctx || (!dst_host && bytes) || (!src_dev && bytes)) return VSP_ERR_ARG;
    VSP_HIP(hipSetDevice(ctx->device));
    VSP_HIP(hipMemcpyAsync(dst_host, src_dev, bytes, hipMemcpyDeviceToHost, ctx->stream));
    VSP_HIP(hipStreamSynchronize(ctx->stream));
    return VSP_OK;
}

// page-lock caller memory (a witness vector, say) so that the copies vsp_groth16_prove / vsp_h2d queue from it are asynchronous DMA
// instead of the runtime's staged pageable path
int vsp_host_register(vsp_ctx *ctx, void *ptr, size_t bytes) {
    if (!ctx) return VSP_ERR_ARG;
    if (!ptr || !bytes) return set_error(ctx, VSP_ERR_ARG, "host_register: null pointer or zero size");
    VSP_HIP(hipSetDevice(ctx->device));
    VSP_HIP(hipHostRegister(ptr, bytes, hipHostRegisterDefault));
    return VSP_OK;
}
int vsp_host_unregister(vsp_ctx *ctx, void *ptr) {
    if (!ctx) return VSP_ERR_ARG;
    if (!ptr) return set_error(ctx, VSP_ERR_ARG, "host_unregister: null pointer");
    VSP_HIP(hipHostUnregister(ptr));
    return VSP_OK;
}

// ---- bases ------------------------------------------------------------------------------------
}  // extern "C"
// Known-answer check of the hand-laid-out field routines THROUGH the kernels that use them (their products are entered with a private
// calling convention the compiler's hazard recogniser and register allocator cannot see into; field-level selftests run them in another
// code arrangement).  Once per context and group, before the first 28-bit table is used, over points the LIBRARY generates (4096
// multiples of the generator: in the subgroup by construction, independent of whatever the caller uploads): the same multi-exponentiation
//   (a) through k_accum28 over the endomorphism layout, the 28-bit merges (k_merge_a, k_merge2: its scalars hold zeros and ones, so one
//       bucket is split in tens of parts), k_dimsum and k_dimbits / k_dimweight -- the default plan;
//   (b) the same with short bucket parts forced ("msm_split" = 6: every bucket is cut in several parts, the merges' full additions run
//       thousands of times) and the other last step of the bucket reduction;
//   (c) through the generic 12 x 32-bit kernels, no split;
//   (d), (e) the default plan again with 6-bit and with 12-bit windows: other digit splits (q0, q1) of the bucket reduction, other lane counts
//       per sum in k_dimsum_mixed, and -- the last 512 points being ONE point under 512 different scalars -- bucket sums that coincide all
//       over the reduction: the doubling branch of the full addition inside k_dimsum(_mixed), k_dimbits and the merges (round 4).
// All five affine results must be identical.  On a mismatch the 28-bit kernels are switched off for this context ("msm_fp28" = 0:
// every later multi-exponentiation takes the generic kernels).  Returns true when 28-bit tables may be used.
static bool fp28_known_answer_check(vsp_ctx *ctx, int group) {
    const int gi = group - 1;
    if (ctx->fp28_checked[gi] != 0) return ctx->fp28_checked[gi] > 0;
    if (ctx->msm_work[0].active) return true;                 // slot 0 busy (unusual): check at the next table instead
    const size_t n = 4096;
    std::vector<uint64_t> sc(2 * n * 4);                      // [0, n): the multiples that make the points; [n, 2n): the scalars of the check
    uint64_t x = 0x9E3779B97F4A7C15ULL ^ (uint64_t)group;
    auto next = [&]() { x += 0x9E3779B97F4A7C15ULL; uint64_t z = x; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL; z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL; return z ^ (z >> 31); };
    for (size_t i = 0; i < 2 * n; i++) {
        sc[4 * i] = next(); sc[4 * i + 1] = next(); sc[4 * i + 2] = next(); sc[4 * i + 3] = next() >> 2;     // < 2^254 < r
        if (i >= n && i % 7 == 3) { sc[4 * i] &= 1; sc[4 * i + 1] = sc[4 * i + 2] = sc[4 * i + 3] = 0; }     // zeros and ones among the check's scalars
    }
    // equal points and opposite points under equal scalars: the same bucket meets P + P (the doubling path of every addition routine, the
    // equal-x hand-back of the accumulation kernel) and P - P (the infinity paths) in all three pipelines
    static const uint64_t R64[4] = {0xffffffff00000001ULL, 0x53bda402fffe5bfeULL, 0x3339d80809a1d805ULL, 0x73eda753299d7d48ULL};
    for (size_t i = 1; i < n; i++) {
        if (i % 16 == 5) for (int j = 0; j < 4; j++) sc[4 * i + j] = sc[4 * (i - 1) + j];                    // the same point twice
        else if (i % 16 == 9) {                                                                             // a point and its negative: r - k
            unsigned __int128 borrow = 0;
            for (int j = 0; j < 4; j++) {
                const unsigned __int128 d = (unsigned __int128)R64[j] - sc[4 * (i - 1) + j] - borrow;
                sc[4 * i + j] = (uint64_t)d; borrow = (d >> 64) & 1;
            }
        } else continue;
        for (int j = 0; j < 4; j++) sc[4 * (n + i) + j] = sc[4 * (n + i - 1) + j];                           // ... under the same scalar
    }
    for (size_t i = n - 511; i < n; i++) for (int j = 0; j < 4; j++) sc[4 * i + j] = sc[4 * (n - 512) + j];  // one point 512 times, scalars as drawn
    const size_t esz = group == 1 ? sizeof(G1Affine) : sizeof(G2Affine), row = group == 1 ? sizeof(Affine28) : sizeof(Affine28x2);
    void *d_pts = nullptr, *t28 = nullptr;
    bool same = false, ran = false;
    const long saved_split = ctx->opts.count("msm_split") ? ctx->opts["msm_split"] : 0, saved_db = ctx->opts.count("msm_dimbits") ? ctx->opts["msm_dimbits"] : -1;
    const long saved_wb = ctx->opts.count("msm_window_bits") ? ctx->opts["msm_window_bits"] : 0;
    if (ensure(ctx, ctx->msm_scalars, 2 * n * 32) == VSP_OK && ensure(ctx, ctx->val_flag, 16) == VSP_OK &&
        hipMemcpyAsync(ctx->msm_scalars.p, sc.data(), 2 * n * 32, hipMemcpyHostToDevice, ctx->stream) == hipSuccess &&
        hipStreamSynchronize(ctx->stream) == hipSuccess && hipMalloc(&d_pts, n * esz) == hipSuccess && hipMalloc(&t28, 2 * n * row) == hipSuccess) {
        const Fr *dk = (const Fr *)ctx->msm_scalars.p, *ds = dk + n;
        int rc = group == 1 ? fixed_base_mul_g1(ctx, dk, n, d_pts) : fixed_base_mul_g2(ctx, dk, n, d_pts);
        if (rc == VSP_OK) rc = group == 1 ? bases_to_mont_g1(ctx, d_pts, (G1Affine *)d_pts, n, 0, (uint32_t *)ctx->val_flag.p)
                                          : bases_to_mont_g2(ctx, d_pts, (G2Affine *)d_pts, n, 0, (uint32_t *)ctx->val_flag.p);
        if (rc == VSP_OK) rc = group == 1 ? msm_g1_table28(ctx, (const G1Affine *)d_pts, n, t28, true) : msm_g2_table28(ctx, (const G2Affine *)d_pts, n, t28, true);
        if (rc == VSP_OK && hipStreamSynchronize(ctx->stream) == hipSuccess) {
            ran = true;
            auto run = [&](const void *table, bool glv, long split, long dimbits, uint64_t *aff, int *inf, long wbits = 0) -> bool {
                if (wbits) ctx->opts["msm_window_bits"] = wbits; else ctx->opts.erase("msm_window_bits");
                if (split) ctx->opts["msm_split"] = split; else ctx->opts.erase("msm_split");
                if (dimbits >= 0) ctx->opts["msm_dimbits"] = dimbits; else ctx->opts.erase("msm_dimbits");
                if (group == 1) {
                    XYZZ<HFp> r;
                    if (msm_g1_launch(ctx, 0, (const G1Affine *)d_pts, ds, n, VSP_MSM_DENSE, nullptr, table, glv) != VSP_OK || msm_g1_finish(ctx, 0, &r) != VSP_OK) return false;
                    Affine<HFp> a = xyzz_to_affine(r); host_store_g1(aff, a); *inf = is_inf(r);
                } else {
                    XYZZ<HFp2> r;
                    if (msm_g2_launch(ctx, 0, (const G2Affine *)d_pts, ds, n, VSP_MSM_DENSE, nullptr, table, glv) != VSP_OK || msm_g2_finish(ctx, 0, &r) != VSP_OK) return false;
                    Affine<HFp2> a = xyzz_to_affine(r); host_store_g2(aff, a); *inf = is_inf(r);
                }
                return true;
            };
            uint64_t ra[24], rb[24], rc3[24], rd[24], re[24]; int ia = 0, ib = 0, ic = 0, id = 0, ie = 0;
            memset(ra, 0, sizeof ra); memset(rb, 0, sizeof rb); memset(rc3, 0, sizeof rc3); memset(rd, 0, sizeof rd); memset(re, 0, sizeof re);
            const bool okr = run(t28, true, 0, -1, ra, &ia) && run(t28, true, 6, group == 1 ? 0 : 1, rb, &ib) && run(nullptr, false, 0, -1, rc3, &ic) &&
                             run(t28, true, 0, -1, rd, &id, 6) && run(t28, true, 0, -1, re, &ie, 12);
            same = okr && ia == ic && ib == ic && id == ic && ie == ic && !ic && memcmp(ra, rc3, sizeof ra) == 0 && memcmp(rb, rc3, sizeof rb) == 0 &&
                   memcmp(rd, rc3, sizeof rd) == 0 && memcmp(re, rc3, sizeof re) == 0;
            // which leg differed (stat "msm_fp28_selfcheck_detail_g1/2"): 1 = the default 28-bit pipeline, 2 = the split-bucket / other reduction
            // pipeline, 4 = a point at infinity where there should be none, 8 = a launch failed, 16 / 32 = the 6-bit / 12-bit window legs
            ctx->stats[group == 1 ? "msm_fp28_selfcheck_detail_g1" : "msm_fp28_selfcheck_detail_g2"] =
                (double)((okr ? 0 : 8) | ((ia != ic || memcmp(ra, rc3, sizeof ra)) ? 1 : 0) | ((ib != ic || memcmp(rb, rc3, sizeof rb)) ? 2 : 0) | (ic ? 4 : 0) |
                         ((id != ic || memcmp(rd, rc3, sizeof rd)) ? 16 : 0) | ((ie != ic || memcmp(re, rc3, sizeof re)) ? 32 : 0));
        }
    }
    if (saved_split) ctx->opts["msm_split"] = saved_split; else ctx->opts.erase("msm_split");
    if (saved_db >= 0) ctx->opts["msm_dimbits"] = saved_db; else ctx->opts.erase("msm_dimbits");
    if (saved_wb) ctx->opts["msm_window_bits"] = saved_wb; else ctx->opts.erase("msm_window_bits");
    if (d_pts) hipFree(d_pts);
    if (t28) hipFree(t28);
    hipGetLastError();
    if (!ran) { ctx->stats[group == 1 ? "msm_fp28_selfcheck_g1" : "msm_fp28_selfcheck_g2"] = 0.0; return false; }      // could not run (out of memory): no verdict, no 28-bit table this time
    { auto it = ctx->opts.find("msm_fp28_selfcheck_fault"); if (it != ctx->opts.end() && it->second) same = false; }   // test hook: exercise the fallback
    ctx->fp28_checked[gi] = same ? 1 : -1;
    ctx->stats[group == 1 ? "msm_fp28_selfcheck_g1" : "msm_fp28_selfcheck_g2"] = same ? 1.0 : -1.0;
    if (!same) { ctx->opts["msm_fp28"] = 0; ctx->err = "msm: the 28-bit-limb kernels failed their known-answer check; generic kernels in use"; }
    return same;
}
// the points once more on 14 x 28-bit limbs for the accumulation kernel (fp28.h); option "msm_fp28" = 0 switches it off.
// Plain bases (no window multiples) whose points are known to satisfy phi(P) = lambda P (vsp_bases.in_subgroup) get the endomorphism
// layout: 2 count rows, (P_i, phi(P_i)) interleaved (option "msm_glv" = 0: off; 2: on for any size and WITHOUT the check -- the caller vouches)
static bool glv_wanted(vsp_ctx *ctx, int group, size_t count, unsigned pre_c) {
    long want = 1; { auto it = ctx->opts.find("msm_fp28"); if (it != ctx->opts.end()) want = it->second; }
    long want_glv = 1; { auto it = ctx->opts.find("msm_glv"); if (it != ctx->opts.end()) want_glv = it->second; }
    const size_t row = group == 1 ? sizeof(Affine28) : sizeof(Affine28x2);
    // The split halves the bucket sets (and the host Horner chain) but doubles the table and the sort's input.  Measured
    // (tools/msm_sizes.py, bench.py; one in flight / three in flight, ms): G1 2^16 1.49 / 1.42 -> 1.35 / 0.80, G1 2^18 2.31 / 1.29 ->
    // 2.04 / 1.27, G1 2^20 4.24 / 3.35 -> 3.98 / 3.34, a 2^20-constraint proof 8.71 -> 8.45 ms (plain key: 9.98 -> 8.95);
    // G2 2^16 3.01 / 1.58 -> 2.67 / 1.40, G2 2^18 5.21 / 3.21 -> 5.56 / 2.90, G2 2^19 7.23 / 4.82 -> 8.28 / 5.58 (dense scalars:
    // the lane-pair merges of the split buckets cost more than the windows saved).  So: on while the doubled table is at most
    // 256 MB for G1 (2^20 points) and 128 MB for G2 (2^18 points); "msm_glv" = 2 forces it on, 0 switches it off.
    const size_t glv_limit = group == 1 ? ((size_t)256 << 20) : ((size_t)128 << 20);
    return want && want_glv && pre_c == 0 && count >= 1024 && count < ((size_t)1 << 30) && (want_glv >= 2 || 2 * count * row <= glv_limit);
}
static void build_table28(vsp_ctx *ctx, vsp_bases *b, size_t count) {
    if (b->d28) { hipFree(b->d28); b->d28 = nullptr; }
    b->glv = false;
    long want = 1; { auto it = ctx->opts.find("msm_fp28"); if (it != ctx->opts.end()) want = it->second; }
    long want_glv = 1; { auto it = ctx->opts.find("msm_glv"); if (it != ctx->opts.end()) want_glv = it->second; }
    // "msm_fp28" = 2 (diagnostics: bisecting a failed check with tools/fuzz_msm.py): the 28-bit kernels WITHOUT the context-time check
    if (!want || (want < 2 && !fp28_known_answer_check(ctx, b->group))) return;   // the check may have just switched "msm_fp28" off
    const size_t row = b->group == 1 ? sizeof(Affine28) : sizeof(Affine28x2);
    bool glv = glv_wanted(ctx, b->group, count, b->pre_c) && (b->in_subgroup > 0 || want_glv >= 2);
    if (b->pre_c && b->pre_split && want_glv && (b->in_subgroup > 0 || want_glv >= 2)) {
        // window multiples for dense scalars: the 128 / c windows of a split scalar, every row with its endomorphism image beside it
        glv = true;
        count = b->n * ((128 + b->pre_c - 1) / b->pre_c);
    }
    void *t28 = nullptr;
    if (hipMalloc(&t28, count * row * (glv ? 2 : 1)) != hipSuccess) { hipGetLastError(); return; }
    int rc = b->group == 1 ? msm_g1_table28(ctx, (const G1Affine *)b->d, count, t28, glv) : msm_g2_table28(ctx, (const G2Affine *)b->d, count, t28, glv);
    if (rc == VSP_OK && hipStreamSynchronize(ctx->stream) == hipSuccess) { b->d28 = t28; b->glv = glv; }
    else { hipFree(t28); hipGetLastError(); }
}
namespace vsp {
vsp_bases *bases_create(vsp_ctx *ctx, int group, const void *src, bool src_on_device, size_t n, int trust) {
    if (!ctx) return nullptr;
    if (!src && n) { set_error(ctx, VSP_ERR_ARG, "bases: null pointer"); return nullptr; }
    hipSetDevice(ctx->device);
    size_t esz = group == 1 ? sizeof(G1Affine) : sizeof(G2Affine);
    vsp_bases *b = new vsp_bases();
    b->group = group; b->n = n;
    if (hipMalloc(&b->d, n ? n * esz : 16) != hipSuccess) { set_error(ctx, VSP_ERR_NOMEM, "bases: hipMalloc failed"); delete b; return nullptr; }
    if (n) {
        int rc;
        if (!src_on_device) {
            if (hipMemcpyAsync(b->d, src, n * esz, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) { set_error(ctx, VSP_ERR_HIP, "bases: H2D failed"); hipFree(b->d); delete b; return nullptr; }
            src = b->d;                      // convert in place
        }
        // boundary validation (include/vsp.h): coordinates below p always; the curve equation unless option "bases_check_curve" = 0
        long check_curve = 1; { auto it = ctx->opts.find("bases_check_curve"); if (it != ctx->opts.end()) check_curve = it->second; }
        // the subgroup (option "bases_check_subgroup"): 1 (default) = checked where the endomorphism split would be used -- bases that fail
        // keep the plain layout, whose result is exact for ANY curve point (like the reference's generic multiexp); 2 = always checked, a
        // failing upload is refused; 0 = never checked, and then never split unless "msm_glv" = 2 (the caller vouches for the points)
        long check_sub = 1; { auto it = ctx->opts.find("bases_check_subgroup"); if (it != ctx->opts.end()) check_sub = it->second; }
        uint32_t h_flag = 0;
        rc = ensure(ctx, ctx->val_flag, 16);
        if (rc == VSP_OK && hipMemsetAsync(ctx->val_flag.p, 0, 16, ctx->stream) != hipSuccess) rc = set_error(ctx, VSP_ERR_HIP, "bases: memset failed");
        if (rc == VSP_OK)
            rc = group == 1 ? bases_to_mont_g1(ctx, src, (G1Affine *)b->d, n, (int)check_curve, (uint32_t *)ctx->val_flag.p)
                            : bases_to_mont_g2(ctx, src, (G2Affine *)b->d, n, (int)check_curve, (uint32_t *)ctx->val_flag.p);
        bool sub_checked = false;
        if (trust == BASES_OWN) b->in_subgroup = 1;
        // policy 2 covers every upload whose points the library did not make itself (caller's handles, one call's host buffers, key blobs),
        // with or without the curve check; policy 1 only the uploads that would get the endomorphism layout
        else if (rc == VSP_OK && (check_sub >= 2 || (trust == BASES_CALLER && check_curve && check_sub == 1 && glv_wanted(ctx, group, n, 0)))) {
            rc = group == 1 ? subgroup_check_g1(ctx, (const G1Affine *)b->d, n, (uint32_t *)ctx->val_flag.p)
                            : subgroup_check_g2(ctx, (const G2Affine *)b->d, n, (uint32_t *)ctx->val_flag.p);
            sub_checked = true;
        }
        if (rc == VSP_OK && (hipMemcpyAsync(&h_flag, ctx->val_flag.p, 4, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
                             hipStreamSynchronize(ctx->stream) != hipSuccess)) rc = set_error(ctx, VSP_ERR_HIP, "bases: conversion failed");
        if (rc == VSP_OK && (h_flag & 3u))
            rc = set_error(ctx, VSP_ERR_ARG, (h_flag & 1u) ? "bases: a coordinate is not canonical (>= p)" : "bases: a point is not on the curve");
        if (rc == VSP_OK && sub_checked) {
            ctx->stats["bases_subgroup_checks"] += 1;
            b->in_subgroup = (h_flag & 4u) ? -1 : 1;
            if (h_flag & 4u) {
                ctx->stats["bases_outside_subgroup"] += 1;
                if (check_sub >= 2) rc = set_error(ctx, VSP_ERR_ARG, "bases: a point is not in the order-r subgroup");
            }
        }
        if (rc != VSP_OK) { hipFree(b->d); delete b; return nullptr; }
        if (n >= 1024) build_table28(ctx, b, n);       // best effort: without it the 12 x 32-bit kernel runs
    }
    return b;
}
}  // namespace vsp
extern "C" {
vsp_bases *vsp_bases_upload_g1(vsp_ctx *ctx, const uint64_t *bases, size_t n) { return bases_create(ctx, 1, bases, false, n, BASES_CALLER); }
vsp_bases *vsp_bases_upload_g2(vsp_ctx *ctx, const uint64_t *bases, size_t n) { return bases_create(ctx, 2, bases, false, n, BASES_CALLER); }
vsp_bases *vsp_bases_from_device_g1(vsp_ctx *ctx, const void *d_bases, size_t n) { return bases_create(ctx, 1, d_bases, true, n, BASES_CALLER); }
vsp_bases *vsp_bases_from_device_g2(vsp_ctx *ctx, const void *d_bases, size_t n) { return bases_create(ctx, 2, d_bases, true, n, BASES_CALLER); }
size_t vsp_bases_count(const vsp_bases *b) { return b ? b->n : 0; }
size_t vsp_bases_device_bytes(const vsp_bases *b) {
    if (!b) return 0;
    const size_t slices = b->pre_c ? 255 / b->pre_c + 1 : 1, count = b->n * slices;
    const size_t esz = b->group == 1 ? sizeof(G1Affine) : sizeof(G2Affine), row = b->group == 1 ? sizeof(Affine28) : sizeof(Affine28x2);
    const size_t rows28 = (b->pre_c && b->glv) ? b->n * ((128 + b->pre_c - 1) / b->pre_c) * 2 : count * (b->glv ? 2 : 1);
    return (count ? count * esz : 16) + (b->d28 ? rows28 * row : 0);
}
void vsp_bases_free(vsp_ctx *ctx, vsp_bases *b) {
    if (!b) return;
    if (ctx) { hipSetDevice(ctx->device); hipStreamSynchronize(ctx->stream); }
    if (b->d) hipFree(b->d);
    if (b->d28) hipFree(b->d28);
    delete b;
}

// ---- MSM ----------------------------------------------------------------------------------------
}  // extern "C"
namespace vsp {
// queue the multi-exponentiation over points [first, first+n) of resident bases on a work slot (plain or precomputed bases)
int launch_on_bases(vsp_ctx *ctx, unsigned slot, const vsp_bases *bases, size_t first, size_t n, const Fr *d_scalars, int plan_from_slot) {
    if (slot < VSP_MSM_SLOTS) ctx->slot_group[slot] = bases->group;
    if (bases->pre_c) {
        MsmPre pre{bases->n, first, bases->pre_c, bases->d28, bases->glv};
        if (bases->group == 1) return msm_g1_launch(ctx, slot, (const G1Affine *)bases->d, d_scalars, n, plan_from_slot, &pre, nullptr, bases->glv);
        return msm_g2_launch(ctx, slot, (const G2Affine *)bases->d, d_scalars, n, plan_from_slot, &pre, nullptr, bases->glv);
    }
    if (bases->group == 1)
        return msm_g1_launch(ctx, slot, (const G1Affine *)bases->d + first, d_scalars, n, plan_from_slot, nullptr,
                             bases->d28 ? (const char *)bases->d28 + first * sizeof(Affine28) * (bases->glv ? 2 : 1) : nullptr, bases->glv);
    return msm_g2_launch(ctx, slot, (const G2Affine *)bases->d + first, d_scalars, n, plan_from_slot, nullptr,
                         bases->d28 ? (const char *)bases->d28 + first * sizeof(Affine28x2) * (bases->glv ? 2 : 1) : nullptr, bases->glv);
}
// the same for a batch of scalar vectors (vector k at d_scalars + k * stride) over PLAIN resident bases: one launch, one result per vector
int launch_on_bases_batch(vsp_ctx *ctx, unsigned slot, const vsp_bases *bases, size_t first, size_t n, const Fr *d_scalars, unsigned batch, size_t stride, bool dense, int plan_from_slot) {
    if (slot < VSP_MSM_SLOTS) ctx->slot_group[slot] = bases->group;
    if (bases->pre_c) {                                       // the table of window multiples: ONE bucket set per vector (round 4, last hours)
        long allow = 1; { auto it = ctx->opts.find("msm_batch_tables"); if (it != ctx->opts.end()) allow = it->second; }
        if (!allow || bases->pre_c > 16) return set_error(ctx, VSP_ERR_UNSUPPORTED, "msm: a batch over this table of window multiples is not supported (plain bases, or windows of at most 16 bits)");
        MsmPre pre{bases->n, first, bases->pre_c, bases->d28, bases->glv};
        if (bases->group == 1) return msm_g1_launch_batch(ctx, slot, (const G1Affine *)bases->d, d_scalars, n, batch, stride, dense, nullptr, bases->glv, plan_from_slot, &pre);
        return msm_g2_launch_batch(ctx, slot, (const G2Affine *)bases->d, d_scalars, n, batch, stride, dense, nullptr, bases->glv, plan_from_slot, &pre);
    }
    if (bases->group == 1)
        return msm_g1_launch_batch(ctx, slot, (const G1Affine *)bases->d + first, d_scalars, n, batch, stride, dense,
                                   bases->d28 ? (const char *)bases->d28 + first * sizeof(Affine28) * (bases->glv ? 2 : 1) : nullptr, bases->glv, plan_from_slot);
    return msm_g2_launch_batch(ctx, slot, (const G2Affine *)bases->d + first, d_scalars, n, batch, stride, dense,
                               bases->d28 ? (const char *)bases->d28 + first * sizeof(Affine28x2) * (bases->glv ? 2 : 1) : nullptr, bases->glv, plan_from_slot);
}
}  // namespace vsp
extern "C" {

static int bases_precompute(vsp_ctx *ctx, vsp_bases *b, unsigned window_bits, bool split);
int vsp_bases_precompute(vsp_ctx *ctx, vsp_bases *b, unsigned window_bits) { return bases_precompute(ctx, b, window_bits, false); }
int vsp_bases_precompute_split(vsp_ctx *ctx, vsp_bases *b, unsigned window_bits) { return bases_precompute(ctx, b, window_bits, true); }
static int bases_precompute(vsp_ctx *ctx, vsp_bases *b, unsigned window_bits, bool split) {
    if (!ctx) return VSP_ERR_ARG;
    if (!b) return set_error(ctx, VSP_ERR_ARG, "precompute: null bases");
    if (split && b->pre_c == 0) {
        // the endomorphism rows need the order-r subgroup (include/vsp.h "bases_check_subgroup"): bases that were not checked at upload are checked now
        long want_glv = 1; { auto it = ctx->opts.find("msm_glv"); if (it != ctx->opts.end()) want_glv = it->second; }
        if (b->in_subgroup == 0 && want_glv == 1 && b->n) {
            VSP_HIP(hipSetDevice(ctx->device));
            VSP_TRY(ensure(ctx, ctx->val_flag, 16));
            VSP_HIP(hipMemsetAsync(ctx->val_flag.p, 0, 4, ctx->stream));
            VSP_TRY(b->group == 1 ? subgroup_check_g1(ctx, (const G1Affine *)b->d, b->n, (uint32_t *)ctx->val_flag.p)
                                  : subgroup_check_g2(ctx, (const G2Affine *)b->d, b->n, (uint32_t *)ctx->val_flag.p));
            uint32_t h_flag = 0;
            VSP_HIP(hipMemcpyAsync(&h_flag, ctx->val_flag.p, 4, hipMemcpyDeviceToHost, ctx->stream));
            VSP_HIP(hipStreamSynchronize(ctx->stream));
            ctx->stats["bases_subgroup_checks"] += 1;
            b->in_subgroup = (h_flag & 4u) ? -1 : 1;
            if (h_flag & 4u) ctx->stats["bases_outside_subgroup"] += 1;
        }
    }
    if (window_bits == 0) {                     // automatic: about n * W / 2^(c-1) = 256 points per shared bucket
        unsigned lg = ceil_log2(b->n ? b->n : 1);
        window_bits = lg < 11 ? 8 : (lg - 3 > 16 ? 16 : lg - 3);
    }
    if (window_bits < 8 || window_bits > 22) return set_error(ctx, VSP_ERR_ARG, "precompute: window_bits must be 8..22");
    if (b->pre_c == window_bits) return VSP_OK;
    if (b->pre_c) return set_error(ctx, VSP_ERR_ARG, "precompute: bases already precomputed for another window size");
    if (b->n == 0) { b->pre_c = window_bits; b->pre_split = split; return VSP_OK; }
    VSP_HIP(hipSetDevice(ctx->device));
    const unsigned W = 255 / window_bits + 1;
    const size_t esz = b->group == 1 ? sizeof(G1Affine) : sizeof(G2Affine);
    if ((size_t)W * b->n >= ((size_t)1 << 31)) return set_error(ctx, VSP_ERR_UNSUPPORTED, "precompute: table too large to index");
    void *table = nullptr;
    if (hipMalloc(&table, (size_t)W * b->n * esz) != hipSuccess) return set_error(ctx, VSP_ERR_NOMEM, "precompute: hipMalloc failed");
    VSP_HIP(hipMemcpyAsync(table, b->d, b->n * esz, hipMemcpyDeviceToDevice, ctx->stream));
    int rc = b->group == 1 ? msm_g1_precompute(ctx, (G1Affine *)table, b->n, window_bits) : msm_g2_precompute(ctx, (G2Affine *)table, b->n, window_bits);
    if (rc != VSP_OK) { hipFree(table); return rc; }
    VSP_HIP(hipStreamSynchronize(ctx->stream));
    hipFree(b->d);
    b->d = table; b->pre_c = window_bits;
    b->pre_split = split;                                    // only now: a refused or failed call leaves the handle as it was (bases outside the subgroup get the ordinary table: build_table28 decides)
    build_table28(ctx, b, (size_t)W * b->n);
    return VSP_OK;
}

static int msm_resident_xyzz(vsp_ctx *ctx, const vsp_bases *bases, size_t first, size_t n, const void *d_scalars,
                             XYZZ<HFp> *o1, XYZZ<HFp2> *o2) {
    if (!ctx) return VSP_ERR_ARG;
    if (!bases || (!d_scalars && n)) return set_error(ctx, VSP_ERR_ARG, "msm: null argument");
    if (first > bases->n || n > bases->n - first) return set_error(ctx, VSP_ERR_ARG, "msm: range outside the resident bases");
    VSP_HIP(hipSetDevice(ctx->device));
    VSP_TRY(launch_on_bases(ctx, 0, bases, first, n, (const Fr *)d_scalars, -1));
    if (bases->group == 1) return msm_g1_finish(ctx, 0, o1);
    return msm_g2_finish(ctx, 0, o2);
}

int vsp_msm_resident(vsp_ctx *ctx, const vsp_bases *bases, size_t first, size_t n, const void *d_scalars, uint64_t *out_affine, int *out_is_inf) {
    XYZZ<HFp> a1; XYZZ<HFp2> a2;
    VSP_TRY(msm_resident_xyzz(ctx, bases, first, n, d_scalars, &a1, &a2));
    return bases->group == 1 ? finish_affine<Fp, HFp>(a1, out_affine, out_is_inf) : finish_affine<Fp2, HFp2>(a2, out_affine, out_is_inf);
}

int vsp_msm_resident_batch(vsp_ctx *ctx, const vsp_bases *bases, size_t first, size_t n, const void *d_scalars, size_t batch, size_t stride,
                           uint64_t *out_affine, int *out_is_inf) {
    if (!ctx) return VSP_ERR_ARG;
    if (!bases || (!d_scalars && n) || !out_affine || batch < 1 || batch > 64 || (batch > 1 && stride < n)) return set_error(ctx, VSP_ERR_ARG, "msm: bad batch argument");
    if (first > bases->n || n > bases->n - first) return set_error(ctx, VSP_ERR_ARG, "msm: range outside the resident bases");
    VSP_HIP(hipSetDevice(ctx->device));
    const size_t words = bases->group == 1 ? 12 : 24;
    VSP_TRY(launch_on_bases_batch(ctx, 0, bases, first, n, (const Fr *)d_scalars, (unsigned)batch, stride, false));
    if (bases->group == 1) {
        std::vector<XYZZ<HFp>> r(batch);
        VSP_TRY(msm_g1_finish_batch(ctx, 0, r.data(), (unsigned)batch));
        for (size_t k = 0; k < batch; k++) VSP_TRY((finish_affine<Fp, HFp>(r[k], out_affine + k * words, out_is_inf ? out_is_inf + k : nullptr)));
    } else {
        std::vector<XYZZ<HFp2>> r(batch);
        VSP_TRY(msm_g2_finish_batch(ctx, 0, r.data(), (unsigned)batch));
        for (size_t k = 0; k < batch; k++) VSP_TRY((finish_affine<Fp2, HFp2>(r[k], out_affine + k * words, out_is_inf ? out_is_inf + k : nullptr)));
    }
    return VSP_OK;
}

int vsp_msm_resident_jacobian(vsp_ctx *ctx, const vsp_bases *bases, size_t first, size_t n, const void *d_scalars, uint64_t *out_jacobian) {
    XYZZ<HFp> a1; XYZZ<HFp2> a2;
    if (!out_jacobian) return set_error(ctx, VSP_ERR_ARG, "msm: null output");
    VSP_TRY(msm_resident_xyzz(ctx, bases, first, n, d_scalars, &a1, &a2));
    if (bases->group == 1) {
        Jacobian<HFp> j = xyzz_to_jacobian(a1);
        host_store_canon(out_jacobian, j.X); host_store_canon(out_jacobian + 6, j.Y); host_store_canon(out_jacobian + 12, j.Z);
    } else {
        Jacobian<HFp2> j = xyzz_to_jacobian(a2);
        host_store_canon(out_jacobian, j.X.c0); host_store_canon(out_jacobian + 6, j.X.c1);
        host_store_canon(out_jacobian + 12, j.Y.c0); host_store_canon(out_jacobian + 18, j.Y.c1);
        host_store_canon(out_jacobian + 24, j.Z.c0); host_store_canon(out_jacobian + 30, j.Z.c1);
    }
    return VSP_OK;
}

// ---- pipelined form: up to VSP_MSM_SLOTS multi-exponentiations in flight, each on its own stream ----
int vsp_msm_launch(vsp_ctx *ctx, unsigned slot, const vsp_bases *bases, size_t first, size_t n, const void *d_scalars) {
    if (!ctx) return VSP_ERR_ARG;
    if (!bases || (!d_scalars && n)) return set_error(ctx, VSP_ERR_ARG, "msm: null argument");
    if (first > bases->n || n > bases->n - first) return set_error(ctx, VSP_ERR_ARG, "msm: range outside the resident bases");
    VSP_HIP(hipSetDevice(ctx->device));
    return launch_on_bases(ctx, slot, bases, first, n, (const Fr *)d_scalars, -1);
}
// the Jacobian record of a finished slot (18 / 36 canonical words); returns the words written
static int finish_record(vsp_ctx *ctx, unsigned slot, uint64_t *out_jacobian, size_t *words) {
    if (ctx->slot_group[slot] == 1) {
        XYZZ<HFp> a; VSP_TRY(msm_g1_finish(ctx, slot, &a));
        Jacobian<HFp> j = xyzz_to_jacobian(a);
        host_store_canon(out_jacobian, j.X); host_store_canon(out_jacobian + 6, j.Y); host_store_canon(out_jacobian + 12, j.Z);
        *words = 18;
    } else {
        XYZZ<HFp2> a; VSP_TRY(msm_g2_finish(ctx, slot, &a));
        Jacobian<HFp2> j = xyzz_to_jacobian(a);
        host_store_canon(out_jacobian, j.X.c0); host_store_canon(out_jacobian + 6, j.X.c1);
        host_store_canon(out_jacobian + 12, j.Y.c0); host_store_canon(out_jacobian + 18, j.Y.c1);
        host_store_canon(out_jacobian + 24, j.Z.c0); host_store_canon(out_jacobian + 30, j.Z.c1);
        *words = 36;
    }
    return VSP_OK;
}
int vsp_msm_finish_jacobian(vsp_ctx *ctx, unsigned slot, uint64_t *out_jacobian) {
    if (!ctx) return VSP_ERR_ARG;
    if (slot >= VSP_MSM_SLOTS || !out_jacobian) return set_error(ctx, VSP_ERR_ARG, "msm: bad slot or null output");
    size_t words = 0;
    return finish_record(ctx, slot, out_jacobian, &words);
}
// The exchange step of the sharded multi-exponentiation wants the record in DEVICE memory (the RCCL all-gather reads it there).  The
// last step of a multi-exponentiation is a chain of c * W dependent doublings, which the host runs ~50x faster than a GPU lane
// (DESIGN.md 3.4), so the record is born on the host: it is written into a pinned ring entry of the slot and copied to d_out_jacobian
// by an asynchronous DMA queued on hip_stream (NULL = the context's stream) -- the caller's host thread never waits for the copy, and
// work queued on hip_stream afterwards (the all-gather) sees the record.
int vsp_msm_finish_jacobian_device(vsp_ctx *ctx, unsigned slot, void *d_out_jacobian, void *hip_stream) {
    if (!ctx) return VSP_ERR_ARG;
    if (slot >= VSP_MSM_SLOTS || !d_out_jacobian) return set_error(ctx, VSP_ERR_ARG, "msm: bad slot or null output");
    VSP_HIP(hipSetDevice(ctx->device));
    MsmWork &wk = ctx->msm_work[slot];
    if (!wk.inited || !wk.h_rec) return set_error(ctx, VSP_ERR_ARG, "msm: finish without launch");
    const unsigned k = wk.rec_idx++ % MsmWork::REC_RING;
    VSP_HIP(hipEventSynchronize(wk.rec_ev[k]));                 // the copy that last read this ring entry (long done in practice; never-recorded events return at once)
    uint64_t *rec = (uint64_t *)((char *)wk.h_rec + (size_t)k * 288);
    size_t words = 0;
    VSP_TRY(finish_record(ctx, slot, rec, &words));
    hipStream_t st = hip_stream ? (hipStream_t)hip_stream : ctx->stream;
    VSP_HIP(hipMemcpyAsync(d_out_jacobian, rec, words * 8, hipMemcpyHostToDevice, st));
    VSP_HIP(hipEventRecord(wk.rec_ev[k], st));
    return VSP_OK;
}

int vsp_fold_jacobian(vsp_ctx *ctx, int group, const uint64_t *records, size_t count, uint64_t *out_affine, int *out_is_inf) {
    if (!records && count) return set_error(ctx, VSP_ERR_ARG, "fold: null records");
    if (group == 1) {
        XYZZ<HFp> acc = XYZZ<HFp>::inf();
        for (size_t i = 0; i < count; i++) {
            const uint64_t *p = records + 18 * i;
            Jacobian<HFp> j; j.X = host_load_canon<HFp>(p); j.Y = host_load_canon<HFp>(p + 6); j.Z = host_load_canon<HFp>(p + 12);
            xyzz_add(acc, jacobian_to_xyzz(j));
        }
        return finish_affine<Fp, HFp>(acc, out_affine, out_is_inf);
    } else if (group == 2) {
        XYZZ<HFp2> acc = XYZZ<HFp2>::inf();
        for (size_t i = 0; i < count; i++) {
            const uint64_t *p = records + 36 * i;
            Jacobian<HFp2> j;
            j.X.c0 = host_load_canon<HFp>(p); j.X.c1 = host_load_canon<HFp>(p + 6);
            j.Y.c0 = host_load_canon<HFp>(p + 12); j.Y.c1 = host_load_canon<HFp>(p + 18);
            j.Z.c0 = host_load_canon<HFp>(p + 24); j.Z.c1 = host_load_canon<HFp>(p + 30);
            xyzz_add(acc, jacobian_to_xyzz(j));
        }
        return finish_affine<Fp2, HFp2>(acc, out_affine, out_is_inf);
    }
    return set_error(ctx, VSP_ERR_ARG, "fold: group must be 1 or 2");
}

// Fold records that sit in DEVICE memory (the output of the all-gather): one asynchronous copy of count * 144 / 288 bytes into a pinned
// buffer on hip_stream (NULL = the context's stream) -- behind whatever produced the records on that stream --, a wait for that stream,
// then the host fold.  The result is wanted on the host (the prover's caller assembles the proof there).
int vsp_fold_jacobian_device(vsp_ctx *ctx, int group, const void *d_records, size_t count, void *hip_stream, uint64_t *out_affine, int *out_is_inf) {
    if (!ctx) return VSP_ERR_ARG;
    if (group != 1 && group != 2) return set_error(ctx, VSP_ERR_ARG, "fold: group must be 1 or 2");
    if (!d_records && count) return set_error(ctx, VSP_ERR_ARG, "fold: null records");
    if (count > 4096) return set_error(ctx, VSP_ERR_ARG, "fold: more than 4096 records");
    VSP_HIP(hipSetDevice(ctx->device));
    const size_t bytes = count * (size_t)(group == 1 ? 144 : 288);
    if (bytes > ctx->h_fold_cap) {
        if (ctx->h_fold) { hipHostFree(ctx->h_fold); ctx->h_fold = nullptr; ctx->h_fold_cap = 0; }
        const size_t want = bytes < 8192 ? 8192 : bytes;
        VSP_HIP(hipHostMalloc(&ctx->h_fold, want, hipHostMallocDefault));
        ctx->h_fold_cap = want;
    }
    hipStream_t st = hip_stream ? (hipStream_t)hip_stream : ctx->stream;
    if (bytes) {
        VSP_HIP(hipMemcpyAsync(ctx->h_fold, d_records, bytes, hipMemcpyDeviceToHost, st));
        VSP_HIP(hipStreamSynchronize(st));
    }
    return vsp_fold_jacobian(ctx, group, (const uint64_t *)ctx->h_fold, count, out_affine, out_is_inf);
}

static int msm_host(vsp_ctx *ctx, int group, const uint64_t *bases, const uint64_t *scalars, size_t n, uint64_t *out_affine, int *out_is_inf) {
    if (!ctx) return VSP_ERR_ARG;
    if ((!bases || !scalars) && n) return set_error(ctx, VSP_ERR_ARG, "msm: null argument");
    VSP_HIP(hipSetDevice(ctx->device));
    vsp_bases *b = bases_create(ctx, group, bases, false, n, BASES_TRANSIENT);      // one call's bases: no endomorphism split, hence no subgroup check
    if (!b) return ctx->err_code ? ctx->err_code : VSP_ERR_ARG;      // bases_create has said why (set_error)
    int rc = ensure(ctx, ctx->msm_scalars, n * 32);
    if (rc == VSP_OK && n) {
        if (hipMemcpyAsync(ctx->msm_scalars.p, scalars, n * 32, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) rc = set_error(ctx, VSP_ERR_HIP, "msm: H2D of scalars failed");
    }
    if (rc == VSP_OK) rc = vsp_msm_resident(ctx, b, 0, n, ctx->msm_scalars.p, out_affine, out_is_inf);
    vsp_bases_free(ctx, b);
    return rc;
}
int vsp_msm_g1(vsp_ctx *ctx, const uint64_t *bases, const uint64_t *scalars, size_t n, uint64_t out_affine[12], int *out_is_inf) {
    return msm_host(ctx, 1, bases, scalars, n, out_affine, out_is_inf);
}
int vsp_msm_g2(vsp_ctx *ctx, const uint64_t *bases, const uint64_t *scalars, size_t n, uint64_t out_affine[24], int *out_is_inf) {
    return msm_host(ctx, 2, bases, scalars, n, out_affine, out_is_inf);
}

// ---- NTT / witness_map ------------------------------------------------------------------------
int vsp_ntt_fr_device(vsp_ctx *ctx, void *d_a, unsigned log_m, int inverse, const uint64_t coset_g[4]) {
    if (!ctx) return VSP_ERR_ARG;
    if (!d_a) return set_error(ctx, VSP_ERR_ARG, "ntt: null pointer");
    VSP_HIP(hipSetDevice(ctx->device));
    return ntt_device(ctx, (Fr *)d_a, log_m, inverse, coset_g, nullptr);
}
int vsp_ntt_fr(vsp_ctx *ctx, uint64_t *a, unsigned log_m, int inverse, const uint64_t coset_g[4]) {
    if (!ctx) return VSP_ERR_ARG;
    if (!a) return set_error(ctx, VSP_ERR_ARG, "ntt: null pointer");
    if (log_m > 28) return set_error(ctx, VSP_ERR_UNSUPPORTED, "ntt: log_m > 28");
    VSP_HIP(hipSetDevice(ctx->device));
    size_t bytes = ((size_t)1 << log_m) * 32;
    VSP_TRY(ensure(ctx, ctx->pr_h, bytes));
    VSP_HIP(hipMemcpyAsync(ctx->pr_h.p, a, bytes, hipMemcpyHostToDevice, ctx->stream));
    VSP_TRY(ntt_device(ctx, (Fr *)ctx->pr_h.p, log_m, inverse, coset_g, nullptr));
    VSP_HIP(hipMemcpyAsync(a, ctx->pr_h.p, bytes, hipMemcpyDeviceToHost, ctx->stream));
    VSP_HIP(hipStreamSynchronize(ctx->stream));
    return VSP_OK;
}
int vsp_witness_map_h_device(vsp_ctx *ctx, void *d_Az, void *d_Bz, void *d_Cz, unsigned log_m, void *d_H) {
    if (!ctx) return VSP_ERR_ARG;
    if (!d_Az || !d_Bz || !d_Cz || !d_H) return set_error(ctx, VSP_ERR_ARG, "witness_map: null pointer");
    if (log_m > 28) return set_error(ctx, VSP_ERR_UNSUPPORTED, "witness_map: log_m > 28");
    VSP_HIP(hipSetDevice(ctx->device));
    vsp_domain d; domain_basic(&d, log_m);
    return witness_map_device(ctx, (Fr *)d_Az, (Fr *)d_Bz, (Fr *)d_Cz, &d, (Fr *)d_H);
}
static int witness_map_host(vsp_ctx *ctx, const vsp_domain *d, uint64_t *Az, uint64_t *Bz, uint64_t *Cz, uint64_t *H) {
    size_t bytes = d->m * 32;
    VSP_TRY(ensure(ctx, ctx->pr_a, bytes)); VSP_TRY(ensure(ctx, ctx->pr_b, bytes));
    VSP_TRY(ensure(ctx, ctx->pr_c, bytes)); VSP_TRY(ensure(ctx, ctx->pr_h, bytes));
    VSP_HIP(hipMemcpyAsync(ctx->pr_a.p, Az, bytes, hipMemcpyHostToDevice, ctx->stream));
    VSP_HIP(hipMemcpyAsync(ctx->pr_b.p, Bz, bytes, hipMemcpyHostToDevice, ctx->stream));
    VSP_HIP(hipMemcpyAsync(ctx->pr_c.p, Cz, bytes, hipMemcpyHostToDevice, ctx->stream));
    VSP_TRY(witness_map_device(ctx, (Fr *)ctx->pr_a.p, (Fr *)ctx->pr_b.p, (Fr *)ctx->pr_c.p, d, (Fr *)ctx->pr_h.p));
    VSP_HIP(hipMemcpyAsync(H, ctx->pr_h.p, bytes, hipMemcpyDeviceToHost, ctx->stream));
    VSP_HIP(hipStreamSynchronize(ctx->stream));
    return VSP_OK;
}
int vsp_witness_map_h(vsp_ctx *ctx, uint64_t *Az, uint64_t *Bz, uint64_t *Cz, unsigned log_m, uint64_t *H) {
    if (!ctx) return VSP_ERR_ARG;
    if (!Az || !Bz || !Cz || !H) return set_error(ctx, VSP_ERR_ARG, "witness_map: null pointer");
    if (log_m > 28) return set_error(ctx, VSP_ERR_UNSUPPORTED, "witness_map: log_m > 28");
    VSP_HIP(hipSetDevice(ctx->device));
    vsp_domain d; domain_basic(&d, log_m);
    return witness_map_host(ctx, &d, Az, Bz, Cz, H);
}

// ---- evaluation_domain<Fr> handles: make_evaluation_domain, basic and step radix-2 ---------------------------------
vsp_domain *vsp_domain_create(vsp_ctx *ctx, size_t min_size) {
    if (!ctx) return nullptr;
    hipSetDevice(ctx->device);
    vsp_domain *d = new vsp_domain();
    if (domain_init(ctx, d, min_size) != VSP_OK) { domain_release(d); delete d; return nullptr; }
    return d;
}
void vsp_domain_free(vsp_ctx *ctx, vsp_domain *d) {
    if (!d) return;
    if (ctx) hipSetDevice(ctx->device);
    domain_release(d);
    delete d;
}
size_t vsp_domain_size(const vsp_domain *d) { return d ? d->m : 0; }
int vsp_domain_kind(const vsp_domain *d) { return d ? d->step : -1; }
int vsp_domain_fft_device(vsp_ctx *ctx, const vsp_domain *d, void *d_a, int inverse, const uint64_t coset_g[4]) {
    if (!ctx) return VSP_ERR_ARG;
    if (!d || !d_a) return set_error(ctx, VSP_ERR_ARG, "domain_fft: null pointer");
    VSP_HIP(hipSetDevice(ctx->device));
    return domain_fft_device(ctx, d, (Fr *)d_a, inverse, coset_g, nullptr);
}
int vsp_domain_fft(vsp_ctx *ctx, const vsp_domain *d, uint64_t *a, int inverse, const uint64_t coset_g[4]) {
    if (!ctx) return VSP_ERR_ARG;
    if (!d || !a) return set_error(ctx, VSP_ERR_ARG, "domain_fft: null pointer");
    VSP_HIP(hipSetDevice(ctx->device));
    size_t bytes = d->m * 32;
    VSP_TRY(ensure(ctx, ctx->pr_h, bytes));
    VSP_HIP(hipMemcpyAsync(ctx->pr_h.p, a, bytes, hipMemcpyHostToDevice, ctx->stream));
    VSP_TRY(domain_fft_device(ctx, d, (Fr *)ctx->pr_h.p, inverse, coset_g, nullptr));
    VSP_HIP(hipMemcpyAsync(a, ctx->pr_h.p, bytes, hipMemcpyDeviceToHost, ctx->stream));
    VSP_HIP(hipStreamSynchronize(ctx->stream));
    return VSP_OK;
}
int vsp_domain_lagrange(vsp_ctx *ctx, const vsp_domain *d, const uint64_t t[4], uint64_t *out) {
    if (!ctx) return VSP_ERR_ARG;
    if (!d || !t || !out) return set_error(ctx, VSP_ERR_ARG, "domain_lagrange: null pointer");
    VSP_HIP(hipSetDevice(ctx->device));
    size_t bytes = d->m * 32;
    VSP_TRY(ensure(ctx, ctx->pr_h, bytes));
    VSP_TRY(domain_lagrange_device(ctx, d, host_load_canon<HFr>(t), (Fr *)ctx->pr_h.p));
    VSP_TRY(fr_from_mont_device(ctx, (Fr *)ctx->pr_h.p, d->m));
    VSP_HIP(hipMemcpyAsync(out, ctx->pr_h.p, bytes, hipMemcpyDeviceToHost, ctx->stream));
    VSP_HIP(hipStreamSynchronize(ctx->stream));
    return VSP_OK;
}
int vsp_domain_element(vsp_ctx *ctx, const vsp_domain *d, size_t idx, uint64_t out[4]) {
    if (!ctx) return VSP_ERR_ARG;
    if (!d || !out || idx >= d->m) return set_error(ctx, VSP_ERR_ARG, "domain_element: bad argument");
    host_store_canon(out, domain_element(d, idx));
    return VSP_OK;
}
int vsp_domain_vanishing(vsp_ctx *ctx, const vsp_domain *d, const uint64_t t[4], uint64_t out[4]) {
    if (!ctx) return VSP_ERR_ARG;
    if (!d || !t || !out) return set_error(ctx, VSP_ERR_ARG, "domain_vanishing: null pointer");
    host_store_canon(out, domain_vanishing(d, host_load_canon<HFr>(t)));
    return VSP_OK;
}
// H (m + 1 coefficients, host, canonical) += coeff * Z
int vsp_domain_add_poly_z(vsp_ctx *ctx, const vsp_domain *d, const uint64_t coeff[4], uint64_t *H) {
    if (!ctx) return VSP_ERR_ARG;
    if (!d || !coeff || !H) return set_error(ctx, VSP_ERR_ARG, "domain_add_poly_z: null pointer");
    HFr c = host_load_canon<HFr>(coeff);
    auto upd = [&](size_t i, const HFr &delta) { HFr v = host_load_canon<HFr>(H + 4 * i); host_store_canon(H + 4 * i, add(v, delta)); };
    if (!d->step) { upd(d->m, c); upd(0, neg(c)); return VSP_OK; }
    uint64_t e[1] = {(uint64_t)d->small_m};
    HFr cw = mul(c, pow_limbs(host_omega(d->log_big + 1), e, 1));
    upd(d->m, c); upd(d->big_m, neg(cw)); upd(d->small_m, neg(c)); upd(0, cw);
    return VSP_OK;
}
int vsp_domain_divide_by_z_on_coset(vsp_ctx *ctx, const vsp_domain *d, uint64_t *P) {
    if (!ctx) return VSP_ERR_ARG;
    if (!d || !P) return set_error(ctx, VSP_ERR_ARG, "domain_divide_by_z_on_coset: null pointer");
    VSP_HIP(hipSetDevice(ctx->device));
    size_t bytes = d->m * 32;
    VSP_TRY(ensure(ctx, ctx->pr_h, bytes));
    VSP_HIP(hipMemcpyAsync(ctx->pr_h.p, P, bytes, hipMemcpyHostToDevice, ctx->stream));
    VSP_TRY(domain_divide_by_z_device(ctx, d, (Fr *)ctx->pr_h.p));
    VSP_HIP(hipMemcpyAsync(P, ctx->pr_h.p, bytes, hipMemcpyDeviceToHost, ctx->stream));
    VSP_HIP(hipStreamSynchronize(ctx->stream));
    return VSP_OK;
}
int vsp_domain_witness_map_h(vsp_ctx *ctx, const vsp_domain *d, uint64_t *Az, uint64_t *Bz, uint64_t *Cz, uint64_t *H) {
    if (!ctx) return VSP_ERR_ARG;
    if (!d || !Az || !Bz || !Cz || !H) return set_error(ctx, VSP_ERR_ARG, "witness_map: null pointer");
    VSP_HIP(hipSetDevice(ctx->device));
    return witness_map_host(ctx, d, Az, Bz, Cz, H);
}
size_t vsp_r1cs_domain_size(const vsp_r1cs *cs) { return cs ? cs->dom.m : 0; }
int vsp_r1cs_domain_kind(const vsp_r1cs *cs) { return cs ? cs->dom.step : -1; }

// ---- generator-side batch exponentiation ---------------------------------------------------------
int vsp_fixed_base_mul_g1(vsp_ctx *ctx, const void *d_scalars, size_t n, void *d_out) {
    if (!ctx) return VSP_ERR_ARG;
    if ((!d_scalars || !d_out) && n) return set_error(ctx, VSP_ERR_ARG, "fixed_base_mul: null pointer");
    VSP_HIP(hipSetDevice(ctx->device));
    VSP_TRY(fixed_base_mul_g1(ctx, (const Fr *)d_scalars, n, d_out));
    VSP_HIP(hipStreamSynchronize(ctx->stream));
    return VSP_OK;
}
int vsp_fixed_base_mul_g2(vsp_ctx *ctx, const void *d_scalars, size_t n, void *d_out) {
    if (!ctx) return VSP_ERR_ARG;
    if ((!d_scalars || !d_out) && n) return set_error(ctx, VSP_ERR_ARG, "fixed_base_mul: null pointer");
    VSP_HIP(hipSetDevice(ctx->device));
    VSP_TRY(fixed_base_mul_g2(ctx, (const Fr *)d_scalars, n, d_out));
    VSP_HIP(hipStreamSynchronize(ctx->stream));
    return VSP_OK;
}

// ---- ZCash compressed encoding (the wire format of reference bin/cli/src/data.bin[0:192]) ----------
static bool canon_lt_p(const uint64_t *l) {
    for (int i = 5; i >= 0; i--) { if (l[i] < FpP64::MOD[i]) return true; if (l[i] > FpP64::MOD[i]) return false; }
    return false;
}
// VSP_ERR_ARG for a null pointer or a coordinate that is not canonical (>= p: its top bits would collide with the flag bits)
int vsp_g1_compress(const uint64_t affine[12], uint8_t out[48]) {
    if (!affine || !out) return VSP_ERR_ARG;
    if (limbs_zero(affine, 12)) { memset(out, 0, 48); out[0] = 0xC0; return VSP_OK; }
    if (!canon_lt_p(affine) || !canon_lt_p(affine + 6)) return VSP_ERR_ARG;
    be48(out, affine);
    out[0] |= 0x80;
    if (fp_lex_larger(affine + 6)) out[0] |= 0x20;
    return VSP_OK;
}
int vsp_g2_compress(const uint64_t affine[24], uint8_t out[96]) {
    if (!affine || !out) return VSP_ERR_ARG;
    if (limbs_zero(affine, 24)) { memset(out, 0, 96); out[0] = 0xC0; return VSP_OK; }
    for (int k = 0; k < 4; k++) if (!canon_lt_p(affine + 6 * k)) return VSP_ERR_ARG;
    be48(out, affine + 6);          // x.c1 first
    be48(out + 48, affine);         // then x.c0
    out[0] |= 0x80;
    const uint64_t *y0 = affine + 12, *y1 = affine + 18;
    bool larger = limbs_zero(y1, 6) ? fp_lex_larger(y0) : fp_lex_larger(y1);
    if (larger) out[0] |= 0x20;
    return VSP_OK;
}


// ---- decompression (the inverse of the above): x from the big-endian bytes, y = sqrt(x^3 + b) with the sign the flag names ----
static void from_be48(uint64_t *l, const uint8_t *in) {
    for (int i = 0; i < 6; i++) { uint64_t v = 0; for (int b = 0; b < 8; b++) v |= (uint64_t)in[47 - (i * 8 + b)] << (8 * b); l[i] = v; }
}
// a^((p+1)/4) -- the square root when a is a quadratic residue (p = 3 mod 4); returns false when it is not
static bool fp_sqrt(const HFp &a, HFp &out) {
    uint64_t e[6]; uint64_t carry = 1;                         // e = (p + 1) / 4
    for (int i = 0; i < 6; i++) { uint64_t v = FpP64::MOD[i] + carry; carry = (v < carry) ? 1 : 0; e[i] = v; }
    for (int i = 0; i < 6; i++) e[i] = (e[i] >> 2) | (i < 5 ? e[i + 1] << 62 : 0);
    out = pow_limbs(a, e, 6);
    return eq(sqr(out), a);
}
// square root in Fp2 = Fp[u]/(u^2+1) by the norm: returns false when a is not a square
static bool fp2_sqrt(const HFp2 &a, HFp2 &out) {
    if (is_zero(a.c1)) {
        HFp r;
        if (fp_sqrt(a.c0, r)) { out.c0 = r; out.c1 = HFp::zero(); return true; }
        if (fp_sqrt(neg(a.c0), r)) { out.c0 = HFp::zero(); out.c1 = r; return true; }   // (r u)^2 = -r^2
        return false;
    }
    HFp s;
    if (!fp_sqrt(add(sqr(a.c0), sqr(a.c1)), s)) return false;
    uint64_t two4[6] = {2, 0, 0, 0, 0, 0};
    HFp half = inv(host_load_canon<HFp>(two4));
    HFp t = mul(add(a.c0, s), half), x0;
    if (!fp_sqrt(t, x0)) { t = mul(sub(a.c0, s), half); if (!fp_sqrt(t, x0)) return false; }
    out.c0 = x0; out.c1 = mul(mul(a.c1, half), inv(x0));
    return eq(sqr(out), a);
}
static const uint64_t R_LIMBS[4] = {0xffffffff00000001ULL, 0x53bda402fffe5bfeULL, 0x3339d80809a1d805ULL, 0x73eda753299d7d48ULL};

int vsp_g1_decompress(const uint8_t in[48], int check_subgroup, uint64_t out_affine[12], int *out_is_inf) {
    if (!in || !out_affine) return VSP_ERR_ARG;
    if (!(in[0] & 0x80)) return VSP_ERR_ARG;                       // uncompressed form is not accepted
    uint8_t buf[48]; memcpy(buf, in, 48); buf[0] &= 0x1f;
    if (in[0] & 0x40) {                                            // infinity: every other bit must be clear
        for (int i = 0; i < 48; i++) if (buf[i]) return VSP_ERR_ARG;
        if (in[0] & 0x20) return VSP_ERR_ARG;
        memset(out_affine, 0, 96); if (out_is_inf) *out_is_inf = 1; return VSP_OK;
    }
    uint64_t x4[6]; from_be48(x4, buf);
    if (!canon_lt_p(x4)) return VSP_ERR_ARG;
    HFp x = host_load_canon<HFp>(x4), y;
    uint64_t four[6] = {4, 0, 0, 0, 0, 0};
    if (!fp_sqrt(add(mul(sqr(x), x), host_load_canon<HFp>(four)), y)) return VSP_ERR_ARG;      // not on the curve
    uint64_t y4[6]; host_store_canon(y4, y);
    if (fp_lex_larger(y4) != ((in[0] & 0x20) != 0)) { y = neg(y); host_store_canon(y4, y); }
    if (check_subgroup) {
        Affine<HFp> p; p.x = x; p.y = y;
        if (!is_inf(xyzz_mul_scalar(xyzz_from_affine(p), R_LIMBS, 255))) return VSP_ERR_ARG;
    }
    memcpy(out_affine, x4, 48); memcpy(out_affine + 6, y4, 48);
    if (out_is_inf) *out_is_inf = 0;
    return VSP_OK;
}
int vsp_g2_decompress(const uint8_t in[96], int check_subgroup, uint64_t out_affine[24], int *out_is_inf) {
    if (!in || !out_affine) return VSP_ERR_ARG;
    if (!(in[0] & 0x80)) return VSP_ERR_ARG;
    uint8_t buf[96]; memcpy(buf, in, 96); buf[0] &= 0x1f;
    if (in[0] & 0x40) {
        for (int i = 0; i < 96; i++) if (buf[i]) return VSP_ERR_ARG;
        if (in[0] & 0x20) return VSP_ERR_ARG;
        memset(out_affine, 0, 192); if (out_is_inf) *out_is_inf = 1; return VSP_OK;
    }
    uint64_t x1[6], x0[6]; from_be48(x1, buf); from_be48(x0, buf + 48);      // x.c1 first, then x.c0
    if (!canon_lt_p(x0) || !canon_lt_p(x1)) return VSP_ERR_ARG;
    HFp2 x, y; x.c0 = host_load_canon<HFp>(x0); x.c1 = host_load_canon<HFp>(x1);
    uint64_t four[6] = {4, 0, 0, 0, 0, 0};
    HFp2 b; b.c0 = host_load_canon<HFp>(four); b.c1 = b.c0;                   // 4 (1 + u)
    if (!fp2_sqrt(add(mul(sqr(x), x), b), y)) return VSP_ERR_ARG;
    uint64_t y0[6], y1[6]; host_store_canon(y0, y.c0); host_store_canon(y1, y.c1);
    bool larger = limbs_zero(y1, 6) ? fp_lex_larger(y0) : fp_lex_larger(y1);
    if (larger != ((in[0] & 0x20) != 0)) { y = neg(y); host_store_canon(y0, y.c0); host_store_canon(y1, y.c1); }
    if (check_subgroup) {
        Affine<HFp2> p; p.x = x; p.y = y;
        if (!is_inf(xyzz_mul_scalar(xyzz_from_affine(p), R_LIMBS, 255))) return VSP_ERR_ARG;
    }
    memcpy(out_affine, x0, 48); memcpy(out_affine + 6, x1, 48); memcpy(out_affine + 12, y0, 48); memcpy(out_affine + 18, y1, 48);
    if (out_is_inf) *out_is_inf = 0;
    return VSP_OK;
}

}  // extern "C"
