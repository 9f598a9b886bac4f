// math::evaluation_domain<Fr> beyond the plain power-of-two case: make_evaluation_domain's choice of domain, the step radix-2
// domain, evaluate_all_lagrange_polynomials, divide_by_z_on_coset, and r1cs_to_qap::witness_map over either domain.
//
// Replaces crypto3-math domains/{evaluation_domain,basic_radix2_domain,step_radix2_domain}.hpp and
// algorithms/make_evaluation_domain.hpp (absent submodule, /root/reference/.gitmodules:47-48; libfqfft lineage
// step_radix2_domain.tcc / get_evaluation_domain.tcc), reached from bin/cli/include/nil/vote_saver/common.hpp:916-917 (generator)
// and :1132-1135 (prover) through r1cs_to_qap.  The reference's circuit does not have a power-of-two constraint count, so its
// domain is a step domain: m = big_m + small_m, the big_m-th roots of unity followed by the coset omega * <small_m-th roots>,
// omega a primitive (2 big_m)-th root.
//
// A step transform is two radix-2 transforms (ntt.hip) plus O(m) glue:
//   fft:   c[i] = a[i] (+ a[i+big_m]),  d[i] = omega^i (a[i] (- a[i+big_m])),  e[i] = sum_j d[i + j small_m];  FFT_big(c) | FFT_small(e)
//   ifft:  U0 = iFFT_big, U1 = iFFT_small;  U1'[i] = (U1[i] - sum_{j>=1} omega^(i + j small_m) U0[i + j small_m]) omega^-i;
//          a[i] = (U0[i] + U1'[i]) / 2, a[big_m + i] = (U0[i] - U1'[i]) / 2 for i < small_m, a[i] = U0[i] otherwise
// The glue kernels are one pass over the data each (coalesced, one thread per element); the strided sums use a tree of
// 64-term partial sums so that a tiny small_m (m = 2^k + 1) still spreads over the whole chip.  Values stay canonical in HBM;
// constants and twiddles are in Montgomery form, so every product lands canonical (as in ntt.hip).
#include "common.h"

namespace vsp {
namespace {

static constexpr unsigned PW_LOG = 11;            // two-level power tables: x^i = lo[i & 2047] * hi[i >> 11]  (as ntt.hip)
static constexpr unsigned FOLD = 64;              // terms per thread and level of the strided sums
static constexpr unsigned LG_CHUNK = 32;          // denominators per batched inversion in the Lagrange kernel

__device__ __forceinline__ Fr pw(const Fr *lo, const Fr *hi, size_t i) { return mul(lo[i & ((1u << PW_LOG) - 1u)], hi[i >> PW_LOG]); }

// ---- forward glue: c in place, d to scratch (optionally after the coset shift a[i] *= g^i)
__global__ __launch_bounds__(256) void k_step_pre(Fr *a, Fr *d, const Fr *tw, unsigned tw_shift, size_t big, size_t small,
                                                  const Fr *g_lo, const Fr *g_hi) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= big) return;
    Fr x = a[i];
    if (g_lo) x = mul(x, pw(g_lo, g_hi, i));
    Fr t = x;
    if (i < small) {
        Fr y = a[i + big];
        if (g_lo) y = mul(y, pw(g_lo, g_hi, i + big));
        t = sub(x, y);
        x = add(x, y);
    }
    a[i] = x;
    d[i] = mul(t, tw[i << tw_shift]);
}

// out[s * small + i] = sum over j in {s, s + S, s + 2S, ...}, j_begin <= j < count, of in[i + j * small] (* tw[(i + j small) << shift])
__global__ __launch_bounds__(256) void k_fold(const Fr *in, Fr *out, size_t small, size_t count, size_t S, size_t j_begin,
                                              const Fr *tw, unsigned tw_shift) {
    size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= S * small) return;
    size_t s = idx / small;
    Fr acc = Fr::zero();
    size_t pos = idx;
    for (size_t j = s; j < count; j += S, pos += S * small) {
        if (j < j_begin) continue;
        Fr v = in[pos];
        if (tw) v = mul(v, tw[pos << tw_shift]);
        acc = add(acc, v);
    }
    out[idx] = acc;
}

struct PostConsts { Fr half, scale; };      // Montgomery: scale/2 and scale
// ---- inverse glue (S = the strided sums above), optionally followed by a[i] *= ginv^i; `scale` multiplies every output
__global__ __launch_bounds__(256) void k_step_post(Fr *a, const Fr *S, const Fr *tw_inv, unsigned tw_shift, size_t big, size_t small,
                                                   const Fr *gi_lo, const Fr *gi_hi, int have_scale, PostConsts k) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= big) return;
    Fr u0 = a[i];
    if (i < small) {
        Fr u1 = mul(sub(a[big + i], S[i]), tw_inv[i << tw_shift]);
        Fr lo = mul(add(u0, u1), k.half), hi = mul(sub(u0, u1), k.half);
        if (gi_lo) { lo = mul(lo, pw(gi_lo, gi_hi, i)); hi = mul(hi, pw(gi_lo, gi_hi, i + big)); }
        a[i] = lo; a[big + i] = hi;
    } else if (gi_lo || have_scale) {
        Fr f = k.scale;
        if (gi_lo) f = mul(f, pw(gi_lo, gi_hi, i));
        a[i] = mul(u0, f);
    }
}

// a[i] *= x^i * scale   (coset shift of the basic domain is fused into ntt.hip; this serves the public divide/scale helpers)
struct ZConsts { Fr c0, c1; };
// 1 / Z(7 x_i) over the first big_m elements of a step domain has period compr: tab[i] = 1 / (cZ0 * omega^(2 small i) - w1Z0)
__global__ __launch_bounds__(64) void k_zinv_table(Fr *tab, size_t compr, size_t small, size_t big, const Fr *tw, unsigned tw_shift, ZConsts k) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= compr) return;
    size_t e = 2 * small * i;               // < 2 big; the table holds omega^j for j < big, and omega^big = -1
    Fr elt = e < big ? tw[e << tw_shift] : neg(tw[(e - big) << tw_shift]);
    tab[i] = inv(sub(mul(k.c0, elt), k.c1));
}

// h[i] = (a[i] b[i] - c[i]) / R * zinv(i)      (canonical a, b, c; the missing factor R is folded into the next transform's scale)
__global__ __launch_bounds__(256) void k_ab_minus_c_div(Fr *h, const Fr *a, const Fr *b, const Fr *c, size_t n, size_t big, size_t compr_mask,
                                                        const Fr *tab, Fr z_small) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Fr x = sub(mul(a[i], b[i]), mul(c[i], Fr::raw_one()));
    h[i] = mul(x, i < big ? tab[i & compr_mask] : z_small);
}
// basic domain: Z is constant on the coset and is folded, with R, into the inverse coset transform
__global__ __launch_bounds__(256) void k_ab_minus_c(Fr *h, const Fr *a, const Fr *b, const Fr *c, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) h[i] = sub(mul(a[i], b[i]), mul(c[i], Fr::raw_one()));
}
// P[i] *= zinv(i)   (public divide_by_z_on_coset; tab == nullptr: constant)
__global__ __launch_bounds__(256) void k_mul_zinv(Fr *p, size_t n, size_t big, size_t compr_mask, const Fr *tab, Fr z_small) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    p[i] = mul(p[i], (tab && i < big) ? tab[i & compr_mask] : z_small);
}

// ---- evaluate_all_lagrange_polynomials for one radix-2 subgroup of size n with generator w (tables w_lo/w_hi), evaluated at x:
//   u[j] = coef * w^j / ((x - w^j) * extra_j),   extra_j = w^(stride j mod n) - shift  (step domain's big part) or 1
struct LagConsts { Fr x, coef, shift; };
__global__ __launch_bounds__(64) void k_lagrange(const Fr *w_lo, const Fr *w_hi, const LagConsts *kc, size_t n, size_t stride, Fr *pre, Fr *u) {
    const Fr k_x = kc->x;
    size_t th = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t b = th * LG_CHUNK, e = b + LG_CHUNK < n ? b + LG_CHUNK : n;
    if (b >= n) return;
    Fr acc = Fr::one();
    for (size_t j = b; j < e; j++) {
        Fr den = sub(k_x, pw(w_lo, w_hi, j));
        if (stride) den = mul(den, sub(pw(w_lo, w_hi, (stride * j) & (n - 1)), kc->shift));
        pre[j] = acc;
        acc = mul(acc, den);
    }
    Fr ai = inv(acc);
    for (size_t j = e; j-- > b;) {
        Fr w = pw(w_lo, w_hi, j);
        Fr den = sub(k_x, w);
        if (stride) den = mul(den, sub(pw(w_lo, w_hi, (stride * j) & (n - 1)), kc->shift));
        Fr di = mul(ai, pre[j]);
        ai = mul(ai, den);
        u[j] = mul(mul(kc->coef, w), di);
    }
}
// t lies in the domain: the indicator of  mult * w^j == t
__global__ __launch_bounds__(256) void k_lagrange_onehot(const Fr *w_lo, const Fr *w_hi, const LagConsts *kc, size_t n, Fr *u) {
    size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    u[j] = eq(mul(pw(w_lo, w_hi, j), kc->coef), kc->x) ? Fr::one() : Fr::zero();
}

__global__ __launch_bounds__(256) void k_fr_from_mont(Fr *a, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) a[i] = from_mont(a[i]);
}

static Fr dev(const HFr &h) { Fr d; memcpy(&d, &h, sizeof(Fr)); return d; }
static HFr h_u64(uint64_t v) { uint64_t c[4] = {v, 0, 0, 0}; return host_load_canon<HFr>(c); }
static HFr h_pow(const HFr &b, uint64_t e) { uint64_t ee[1] = {e}; return pow_limbs(b, ee, 1); }
static const uint64_t G7[4] = {7, 0, 0, 0};

static bool pick_basic(vsp_domain *d, size_t m) {
    if (m <= 1 || (m & (m - 1)) || ceil_log2(m) > 28) return false;
    d->m = d->big_m = m; d->small_m = 0; d->log_big = ceil_log2(m); d->log_small = 0; d->step = 0; return true;
}
static bool pick_step(vsp_domain *d, size_t m) {
    if (m <= 1 || ceil_log2(m) > 28) return false;
    size_t big = (size_t)1 << (ceil_log2(m) - 1), small = m - big;
    if (small == 0 || (small & (small - 1))) return false;
    d->m = m; d->big_m = big; d->small_m = small; d->log_big = ceil_log2(big); d->log_small = ceil_log2(small); d->step = 1; return true;
}

// strided sums: out[i] = sum_{j_begin <= j < count} term(i + j small), i < small
static int fold(vsp_ctx *ctx, const Fr *in, Fr *out, Fr *ping, Fr *pong, size_t small, size_t count, size_t j_begin, const Fr *tw, unsigned tw_shift) {
    const Fr *src = in;
    for (int level = 0;; level++) {
        size_t S = (count + FOLD - 1) / FOLD;
        Fr *dst = S == 1 ? out : ((level & 1) ? pong : ping);
        size_t threads = S * small;
        hipLaunchKernelGGL(k_fold, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, ctx->stream, src, dst, small, count, S, j_begin, tw, tw_shift);
        VSP_LAUNCH_CHECK();
        if (S == 1) return VSP_OK;
        src = dst; count = S; j_begin = 0; tw = nullptr;
    }
}

}  // anonymous namespace

// make_evaluation_domain(min_size): libfqfft get_evaluation_domain.tcc order -- basic_radix2(min_size), [extended_radix2: needs
// m = 2^33, never constructs here], step_radix2(min_size), then the same at big + rounded_small.
int domain_init(vsp_ctx *ctx, vsp_domain *d, size_t min_size) {
    if (min_size <= 1) return set_error(ctx, VSP_ERR_ARG, "make_evaluation_domain: min_size must exceed 1");
    if (ceil_log2(min_size) > 28) return set_error(ctx, VSP_ERR_UNSUPPORTED, "make_evaluation_domain: domain larger than 2^28");
    size_t big = (size_t)1 << (ceil_log2(min_size) - 1), small = min_size - big, rounded = (size_t)1 << ceil_log2(small);
    if (!(pick_basic(d, min_size) || pick_step(d, min_size) || pick_basic(d, big + rounded) || pick_step(d, big + rounded)))
        return set_error(ctx, VSP_ERR_UNSUPPORTED, "make_evaluation_domain: no radix-2 family domain of this size");
    // divisors of divide_by_z_on_coset, coset generator 7
    HFr g = h_u64(7);
    if (!d->step) { d->zinv_const = inv(domain_vanishing(d, g)); return VSP_OK; }
    VSP_HIP(hipSetDevice(ctx->device));
    VSP_TRY(ntt_ensure_twiddles(ctx, d->log_big + 1));
    HFr omega = host_omega(d->log_big + 1);
    HFr Z0 = sub(h_pow(g, d->big_m), HFr::one());
    HFr w1 = h_pow(omega, d->small_m);
    ZConsts k; k.c0 = dev(mul(h_pow(g, d->small_m), Z0)); k.c1 = dev(mul(w1, Z0));
    size_t compr = d->big_m / d->small_m;
    VSP_TRY(ensure(ctx, d->zinv, compr * sizeof(Fr)));
    hipLaunchKernelGGL(k_zinv_table, dim3((unsigned)((compr + 63) / 64)), dim3(64), 0, ctx->stream, (Fr *)d->zinv.p, compr, d->small_m, d->big_m,
                       (const Fr *)ctx->ntt.fwd.p, ctx->ntt.log - 1 - d->log_big, k);
    VSP_LAUNCH_CHECK();
    VSP_HIP(hipStreamSynchronize(ctx->stream));
    d->zinv_const = inv(domain_vanishing(d, mul(g, omega)));
    return VSP_OK;
}
void domain_basic(vsp_domain *d, unsigned log_m) {
    d->m = d->big_m = (size_t)1 << log_m; d->small_m = 0; d->log_big = log_m; d->log_small = 0; d->step = 0;
    d->zinv_const = inv(domain_vanishing(d, h_u64(7)));
}
int fr_from_mont_device(vsp_ctx *ctx, Fr *a, size_t n) {
    if (!n) return VSP_OK;
    hipLaunchKernelGGL(k_fr_from_mont, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, a, n);
    VSP_LAUNCH_CHECK();
    return VSP_OK;
}
void domain_release(vsp_domain *d) { if (d->zinv.p) hipFree(d->zinv.p); d->zinv.p = nullptr; d->zinv.cap = 0; }

HFr domain_vanishing(const vsp_domain *d, const HFr &t) {
    HFr a = sub(h_pow(t, d->big_m), HFr::one());
    if (!d->step) return a;
    HFr omega = host_omega(d->log_big + 1);
    return mul(a, sub(h_pow(t, d->small_m), h_pow(omega, d->small_m)));
}
HFr domain_element(const vsp_domain *d, size_t idx) {
    if (!d->step) return h_pow(host_omega(d->log_big), idx);
    HFr omega = host_omega(d->log_big + 1);
    if (idx < d->big_m) return h_pow(sqr(omega), idx);
    return mul(omega, h_pow(host_omega(d->log_small), idx - d->big_m));
}

int domain_fft_device(vsp_ctx *ctx, const vsp_domain *d, Fr *a, int inverse, const uint64_t *coset_g, const HFr *extra_scale) {
    if (!d->step) return ntt_device(ctx, a, d->log_big, inverse, coset_g, extra_scale);
    if (coset_g && !(coset_g[0] | coset_g[1] | coset_g[2] | coset_g[3])) return set_error(ctx, VSP_ERR_ARG, "fft: coset generator is zero");
    ntt_selfcheck_once(ctx);
    const size_t big = d->big_m, small = d->small_m, compr = big / small;
    VSP_TRY(ntt_ensure_twiddles(ctx, d->log_big + 1));
    if (coset_g) VSP_TRY(ntt_ensure_coset_tables(ctx, d->log_big + 1, coset_g));
    // scratch: d (big) | ping | pong (partial sums)
    const size_t part = big / FOLD + small + FOLD;
    VSP_TRY(ensure(ctx, ctx->dom_scratch, (big + 2 * part) * sizeof(Fr)));
    Fr *dv = (Fr *)ctx->dom_scratch.p, *ping = dv + big, *pong = ping + part;
    const unsigned blocks = (unsigned)((big + 255) / 256);
    if (!inverse) {
        const unsigned shift = ctx->ntt.log - 1 - d->log_big;
        hipLaunchKernelGGL(k_step_pre, dim3(blocks), dim3(256), 0, ctx->stream, a, dv, (const Fr *)ctx->ntt.fwd.p, shift, big, small,
                           coset_g ? (const Fr *)ctx->ntt.pw_lo_f.p : nullptr, coset_g ? (const Fr *)ctx->ntt.pw_hi_f.p : nullptr);
        VSP_LAUNCH_CHECK();
        VSP_TRY(fold(ctx, dv, a + big, ping, pong, small, compr, 0, nullptr, 0));
        VSP_TRY(ntt_device(ctx, a, d->log_big, 0, nullptr, extra_scale));
        VSP_TRY(ntt_device(ctx, a + big, d->log_small, 0, nullptr, extra_scale));
        return VSP_OK;
    }
    VSP_TRY(ntt_device(ctx, a, d->log_big, 1, nullptr, nullptr));
    VSP_TRY(ntt_device(ctx, a + big, d->log_small, 1, nullptr, nullptr));
    const unsigned shift = ctx->ntt.log - 1 - d->log_big;      // after the transforms: their table may have been regenerated larger
    VSP_TRY(fold(ctx, a, dv, ping, pong, small, compr, 1, (const Fr *)ctx->ntt.fwd.p, shift));
    PostConsts k;
    HFr scale = extra_scale ? *extra_scale : HFr::one();
    k.scale = dev(scale); k.half = dev(mul(scale, inv(h_u64(2))));
    hipLaunchKernelGGL(k_step_post, dim3(blocks), dim3(256), 0, ctx->stream, a, (const Fr *)dv, (const Fr *)ctx->ntt.inv.p, shift, big, small,
                       coset_g ? (const Fr *)ctx->ntt.pw_lo_i.p : nullptr, coset_g ? (const Fr *)ctx->ntt.pw_hi_i.p : nullptr, extra_scale ? 1 : 0, k);
    VSP_LAUNCH_CHECK();
    return VSP_OK;
}

int domain_divide_by_z_device(vsp_ctx *ctx, const vsp_domain *d, Fr *p) {
    size_t compr = d->step ? d->big_m / d->small_m : 1;
    hipLaunchKernelGGL(k_mul_zinv, dim3((unsigned)((d->m + 255) / 256)), dim3(256), 0, ctx->stream, p, d->m, d->step ? d->big_m : 0, compr - 1,
                       d->step ? (const Fr *)d->zinv.p : nullptr, dev(d->zinv_const));
    VSP_LAUNCH_CHECK();
    return VSP_OK;
}

// u[j] = L_j(t), Montgomery form, device
int domain_lagrange_device(vsp_ctx *ctx, const vsp_domain *d, const HFr &t, Fr *u) {
    hipStream_t st = ctx->stream;
    DevBuf lo, hi, pre, kbuf;
    auto done = [&](int rc) { hipStreamSynchronize(st); DevBuf *all[] = {&lo, &hi, &pre, &kbuf}; for (DevBuf *b : all) if (b->p) hipFree(b->p); return rc; };
    if (ensure(ctx, pre, d->m * sizeof(Fr)) != VSP_OK || ensure(ctx, kbuf, 2 * sizeof(LagConsts)) != VSP_OK) return done(VSP_ERR_NOMEM);
    const bool in_domain = is_zero(domain_vanishing(d, t));
    // the radix-2 subgroups to evaluate: (size, generator, point x, coefficient, offset, stride/shift of the extra denominators)
    struct Part { size_t n; HFr w, x, coef, shift, mult; size_t stride, off; } parts[2];
    int np = 0;
    if (!d->step) {
        HFr Z = sub(h_pow(t, d->m), HFr::one());
        parts[np++] = {d->m, host_omega(d->log_big), t, mul(Z, inv(h_u64(d->m))), HFr::zero(), HFr::one(), 0, 0};
    } else {
        HFr omega = host_omega(d->log_big + 1), ts = mul(t, inv(omega));
        HFr ws = h_pow(omega, d->small_m);
        HFr Zb = sub(h_pow(t, d->big_m), HFr::one()), Zs = sub(h_pow(ts, d->small_m), HFr::one());
        HFr L0 = sub(h_pow(t, d->small_m), ws);
        HFr L1 = mul(Zb, inv(sub(h_pow(omega, d->big_m), HFr::one())));
        parts[np++] = {d->big_m, sqr(omega), t, mul(mul(Zb, inv(h_u64(d->big_m))), L0), ws, HFr::one(), d->small_m, 0};
        parts[np++] = {d->small_m, host_omega(d->log_small), ts, mul(mul(Zs, inv(h_u64(d->small_m))), L1), HFr::zero(), omega, 0, d->big_m};
    }
    for (int k = 0; k < np; k++) {
        const Part &p = parts[k];
        const size_t hi_count = p.n > ((size_t)1 << PW_LOG) ? (p.n >> PW_LOG) : 1;
        int rc = upload_power_tables(ctx, p.w, hi_count, lo, hi);
        if (rc != VSP_OK) return done(rc);
        LagConsts kc; kc.x = dev(in_domain ? t : p.x); kc.coef = dev(in_domain ? p.mult : p.coef); kc.shift = dev(p.shift);
        LagConsts *kd = (LagConsts *)kbuf.p + k;
        if (hipMemcpyAsync(kd, &kc, sizeof kc, hipMemcpyHostToDevice, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess)
            return done(set_error(ctx, VSP_ERR_HIP, "lagrange: constants upload failed"));
        if (in_domain) {
            hipLaunchKernelGGL(k_lagrange_onehot, dim3((unsigned)((p.n + 255) / 256)), dim3(256), 0, st, (const Fr *)lo.p, (const Fr *)hi.p, kd, p.n, u + p.off);
        } else {
            size_t chunks = (p.n + LG_CHUNK - 1) / LG_CHUNK;
            hipLaunchKernelGGL(k_lagrange, dim3((unsigned)((chunks + 63) / 64)), dim3(64), 0, st, (const Fr *)lo.p, (const Fr *)hi.p, kd, p.n, p.stride,
                               (Fr *)pre.p + p.off, u + p.off);
        }
        if (hipGetLastError() != hipSuccess) return done(set_error(ctx, VSP_ERR_HIP, "lagrange: kernel launch failed"));
        if (hipStreamSynchronize(st) != hipSuccess) return done(set_error(ctx, VSP_ERR_HIP, "lagrange: kernel failed"));   // lo/hi are reused
    }
    return done(VSP_OK);
}

// r1cs_to_qap::witness_map, d1 = d2 = d3 = 0:  H = icosetFFT( (cosetFFT(iFFT(A)) * cosetFFT(iFFT(B)) - cosetFFT(iFFT(C))) / Z on the coset )
// dA, dB, dC: m canonical values each (overwritten); dH receives the m coefficients of H.
int witness_map_device(vsp_ctx *ctx, Fr *dA, Fr *dB, Fr *dC, const vsp_domain *d, Fr *dH) {
    const size_t m = d->m;
    Fr *v[3] = {dA, dB, dC};
    { long batched = 1; auto it = ctx->opts.find("witness_map_batched"); if (it != ctx->opts.end()) batched = it->second;
      if (batched && !d->step && m >= 2 && ntt29_in_use(ctx)) {
        // basic radix-2 domain, 29-bit butterflies: the three inverse transforms as ONE launch per pass, the three coset transforms likewise, and
        // the pointwise step A B - C inside the first pass of the last transform: 3 x passes launches instead of 7 x passes + 1, and the
        // 128 bytes per element the pointwise kernel moved stay in registers.  (Option "witness_map_batched" = 0: the sequence below.)
        unsigned log_m = 0; while (((size_t)1 << log_m) < m) log_m++;
        VSP_TRY(ntt_device_batch(ctx, v, 3, log_m, 1, nullptr, nullptr));
        VSP_TRY(ntt_device_batch(ctx, v, 3, log_m, 0, G7, nullptr));
        // the fused load leaves (a b - c) 2^-261: 2^261 = 32 R goes into the scale with 1 / Z(g), constant on the coset of a basic domain
        const uint64_t c32[4] = {32, 0, 0, 0};
        HFr extra = mul(mul(HFr::r2(), host_load_canon<HFr>(c32)), d->zinv_const);
        VSP_TRY(ntt_device_fused_abc(ctx, dA, dB, dC, dH, log_m, 1, G7, &extra));
        return VSP_OK;
      } }
    for (int k = 0; k < 3; k++) {
        VSP_TRY(domain_fft_device(ctx, d, v[k], 1, nullptr, nullptr));
        VSP_TRY(domain_fft_device(ctx, d, v[k], 0, G7, nullptr));
    }
    const unsigned blocks = (unsigned)((m + 255) / 256);
    HFr extra = HFr::r2();                    // the value R in Montgomery form: undoes the plain Montgomery products of the kernel
    if (!d->step) {
        hipLaunchKernelGGL(k_ab_minus_c, dim3(blocks), dim3(256), 0, ctx->stream, dH, (const Fr *)dA, (const Fr *)dB, (const Fr *)dC, m);
        extra = mul(d->zinv_const, extra);    // Z is constant on the coset of a basic domain: fold 1/Z(g) as well
    } else {
        hipLaunchKernelGGL(k_ab_minus_c_div, dim3(blocks), dim3(256), 0, ctx->stream, dH, (const Fr *)dA, (const Fr *)dB, (const Fr *)dC, m, d->big_m,
                           d->big_m / d->small_m - 1, (const Fr *)d->zinv.p, dev(d->zinv_const));
    }
    VSP_LAUNCH_CHECK();
    VSP_TRY(domain_fft_device(ctx, d, dH, 1, G7, &extra));
    return VSP_OK;
}

// the same for `count` witnesses: abc holds [count][3][m] evaluation vectors (A z, B z, C z of witness k at abc + 3 k m), dH [count][m].
// Basic domains on the 29-bit butterflies run 3 count transforms per launch; anything else takes the single form, witness by witness.
int witness_map_device_batch(vsp_ctx *ctx, Fr *abc, unsigned count, const vsp_domain *d, Fr *dH) {
    const size_t m = d->m;
    if (d->step || m < 2 || !ntt29_in_use(ctx)) {
        for (unsigned k = 0; k < count; k++) VSP_TRY(witness_map_device(ctx, abc + (size_t)3 * k * m, abc + (size_t)(3 * k + 1) * m, abc + (size_t)(3 * k + 2) * m, d, dH + (size_t)k * m));
        return VSP_OK;
    }
    unsigned log_m = 0; while (((size_t)1 << log_m) < m) log_m++;
    VSP_TRY(ntt_device_strided(ctx, abc, 3 * count, m, log_m, 1, nullptr, nullptr));
    VSP_TRY(ntt_device_strided(ctx, abc, 3 * count, m, log_m, 0, G7, nullptr));
    const uint64_t c32[4] = {32, 0, 0, 0};
    HFr extra = mul(mul(HFr::r2(), host_load_canon<HFr>(c32)), d->zinv_const);
    VSP_TRY(ntt_device_fused_abc_strided(ctx, abc, m, 2 * m, 3 * m, dH, m, count, log_m, 1, G7, &extra));
    return VSP_OK;
}

}  // namespace vsp
