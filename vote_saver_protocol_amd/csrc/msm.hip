// Pippenger (bucket-method) multi-scalar multiplication over BLS12-381 G1 / G2 for gfx950.
//
// Replaces algebra::multiexp<policies::multiexp_method_BDLO12>, multiexp_with_mixed_addition and the
// G2 half of kc_multiexp_with_mixed_addition (crypto3-algebra / crypto3-zk, absent submodules,
// /root/reference/.gitmodules:8-12; parameter table included at
// bin/cli/include/nil/vote_saver/common.hpp:38, reached from common.hpp:1132-1135).
// The reference algorithm is serial: for each c-bit window, add every base into bucket[digit], then a
// running sum over the buckets, then c doublings between windows.  The value sum_i k_i * P_i is unique
// as an affine point, so any bucket schedule gives bit-identical output after normalisation.
//
// MI355X pipeline (all on one stream, no host round trip until the last few hundred bytes):
//   1. k_count     signed c-bit digits of every scalar (halves the buckets), histogram per (window, bucket)
//   2. scan        exclusive prefix sums -> bucket offsets; k_plan splits buckets larger than T into parts
//   3. k_scatter   counting-sort of point indices (+ sign bit) into bucket order
//   4. k_accum     ONE THREAD PER BUCKET PART: gathers its affine points (96 B / 192 B rows, the bases fit
//                  in the Infinity Cache for n <= 2^21) and folds them with mixed XYZZ additions -- this is
//                  where >85 % of the time goes; it is VALU integer-multiply bound, not HBM bound
//   5. k_merge     buckets that were split (skewed scalars: the 0/1-heavy witnesses of
//                  multiexp_with_mixed_addition) are folded by a workgroup each, LDS tree
//   6. k_dimsum    the weighted bucket sum  sum_b (b+1) B_b  is decomposed over the 3 digits of the bucket
//                  index b = (v2, v1, v0):  plain sums along each digit (workgroup + LDS tree per sum),
//   7. k_dimweight then a <=256-term weighted sum per (window, digit) by suffix scan in LDS
//   8. host        Horner over W*4 points (c*W doublings) in 64-bit limbs and the affine normalisation
//
// Zero scalars produce no digit and are skipped; scalars equal to one land in one bucket of window 0
// and are summed by steps 4-5 -- the two special cases of multiexp_with_mixed_addition need no
// separate pre-pass.
#include "common.h"

namespace vsp {

static constexpr unsigned MSM_THREADS = 256;

// ------------------------------------------------------------------------------------------------
// digits
struct MsmGeom {
    unsigned c;         // window bits
    unsigned W;         // windows
    unsigned B;         // buckets per window = 2^(c-1)
    unsigned q0, q1, q2;  // bucket index bit split, q0+q1+q2 = c-1
    unsigned T;         // split threshold (max points per bucket part)
    size_t n;
    size_t G;           // W * B
};

__device__ __forceinline__ uint32_t scalar_bits(const uint32_t *k, unsigned pos, unsigned c) {
    unsigned limb = pos >> 5, sh = pos & 31;
    if (limb >= 8) return 0;
    uint64_t v = k[limb];
    if (limb + 1 < 8) v |= (uint64_t)k[limb + 1] << 32;
    return (uint32_t)(v >> sh) & ((1u << c) - 1u);
}

// calls f(window, bucket_index, negative) for every non-zero signed digit of scalar k
template <class Fn> __device__ __forceinline__ void for_each_digit(const uint32_t *k, const MsmGeom &g, Fn f) {
    uint32_t carry = 0;
    for (unsigned w = 0; w < g.W; w++) {
        uint32_t raw = scalar_bits(k, w * g.c, g.c) + carry;
        uint32_t mag; bool neg;
        if (raw > g.B) { mag = (1u << g.c) - raw; neg = true; carry = 1; }
        else { mag = raw; neg = false; carry = 0; }
        if (mag) f(w, mag - 1, neg);
    }
}

__global__ void k_count(const Fr *scalars, MsmGeom g, uint32_t *cnt) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= g.n) return;
    const uint32_t *k = scalars[i].l;
    for_each_digit(k, g, [&](unsigned w, uint32_t b, bool) { atomicAdd(&cnt[(size_t)w * g.B + b], 1u); });
}

__global__ void k_scatter(const Fr *scalars, MsmGeom g, uint32_t *cursor, uint32_t *sorted) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= g.n) return;
    const uint32_t *k = scalars[i].l;
    for_each_digit(k, g, [&](unsigned w, uint32_t b, bool neg) {
        uint32_t pos = atomicAdd(&cursor[(size_t)w * g.B + b], 1u);
        sorted[pos] = (uint32_t)i | (neg ? 0x80000000u : 0u);
    });
}

// ------------------------------------------------------------------------------------------------
// exclusive scan of uint32 (three small kernels; arrays are <= a few million entries)
static constexpr unsigned SCAN_ITEMS = 8;                       // per thread
static constexpr unsigned SCAN_TILE = MSM_THREADS * SCAN_ITEMS;  // 2048 per block

__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t v, uint32_t *total) {
    __shared__ uint32_t sh[MSM_THREADS];
    unsigned t = threadIdx.x;
    sh[t] = v;
    __syncthreads();
    for (unsigned s = 1; s < MSM_THREADS; s <<= 1) {
        uint32_t x = t >= s ? sh[t - s] : 0;
        __syncthreads();
        sh[t] += x;
        __syncthreads();
    }
    uint32_t incl = sh[t];
    if (total) *total = sh[MSM_THREADS - 1];
    __syncthreads();
    return incl - v;
}

__global__ __launch_bounds__(MSM_THREADS) void k_scan_reduce(const uint32_t *in, size_t n, uint32_t *blocksum) {
    size_t base = (size_t)blockIdx.x * SCAN_TILE + (size_t)threadIdx.x * SCAN_ITEMS;
    uint32_t s = 0;
    for (unsigned i = 0; i < SCAN_ITEMS; i++) if (base + i < n) s += in[base + i];
    uint32_t tot;
    block_exclusive_scan(s, &tot);
    if (threadIdx.x == 0) blocksum[blockIdx.x] = tot;
}
// single block: exclusive scan of blocksum[0..nb) in place; writes the grand total to blocksum[nb]
__global__ __launch_bounds__(MSM_THREADS) void k_scan_top(uint32_t *blocksum, size_t nb) {
    uint32_t carry = 0;
    for (size_t base = 0; base < nb; base += MSM_THREADS) {
        size_t i = base + threadIdx.x;
        uint32_t v = i < nb ? blocksum[i] : 0, tot;
        uint32_t ex = block_exclusive_scan(v, &tot);
        if (i < nb) blocksum[i] = carry + ex;
        carry += tot;
    }
    if (threadIdx.x == 0) blocksum[nb] = carry;
}
// out[i] = exclusive prefix; out2 (optional) gets a copy; out[n] = total
__global__ __launch_bounds__(MSM_THREADS) void k_scan_final(const uint32_t *in, size_t n, const uint32_t *blocksum, uint32_t *out, uint32_t *out2) {
    size_t base = (size_t)blockIdx.x * SCAN_TILE + (size_t)threadIdx.x * SCAN_ITEMS;
    uint32_t v[SCAN_ITEMS], s = 0;
    for (unsigned i = 0; i < SCAN_ITEMS; i++) { v[i] = base + i < n ? in[base + i] : 0; s += v[i]; }
    uint32_t ex = block_exclusive_scan(s, nullptr) + blocksum[blockIdx.x];
    for (unsigned i = 0; i < SCAN_ITEMS; i++) {
        if (base + i < n) { out[base + i] = ex; if (out2) out2[base + i] = ex; }
        ex += v[i];
    }
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == MSM_THREADS - 1) { out[n] = ex; if (out2) out2[n] = ex; }
}

static int exclusive_scan(vsp_ctx *ctx, const uint32_t *in, size_t n, uint32_t *out, uint32_t *out2) {
    size_t nb = (n + SCAN_TILE - 1) / SCAN_TILE;
    VSP_TRY(ensure(ctx, ctx->msm_blocksum, (nb + 1) * sizeof(uint32_t)));
    uint32_t *bs = (uint32_t *)ctx->msm_blocksum.p;
    hipLaunchKernelGGL(k_scan_reduce, dim3((unsigned)nb), dim3(MSM_THREADS), 0, ctx->stream, in, n, bs);
    hipLaunchKernelGGL(k_scan_top, dim3(1), dim3(MSM_THREADS), 0, ctx->stream, bs, nb);
    hipLaunchKernelGGL(k_scan_final, dim3((unsigned)nb), dim3(MSM_THREADS), 0, ctx->stream, in, n, (const uint32_t *)bs, out, out2);
    VSP_LAUNCH_CHECK();
    return VSP_OK;
}

// ------------------------------------------------------------------------------------------------
// plan: number of parts per bucket (>= 1), list of split ("heavy") buckets
// counters[0] = number of heavy buckets
__global__ void k_plan(const uint32_t *cnt, size_t G, unsigned T, uint32_t *nsub, uint32_t *heavy, uint32_t *counters) {
    size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= G) return;
    uint32_t s = cnt[g];
    uint32_t parts = s <= T ? 1u : (s + T - 1) / T;
    nsub[g] = parts;
    if (parts > 1) heavy[atomicAdd(&counters[0], 1u)] = (uint32_t)g;
}

// ------------------------------------------------------------------------------------------------
// accumulate: one thread per bucket part
template <class F>
__global__ __launch_bounds__(MSM_THREADS) void k_accum(const Affine<F> *__restrict__ bases, const uint32_t *__restrict__ sorted,
                                                       const uint32_t *__restrict__ off, const uint32_t *__restrict__ suboff,
                                                       size_t G, unsigned T, XYZZ<F> *buckets, XYZZ<F> *partials) {
    size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t S = suboff[G];
    if (q >= S) return;
    // largest g with suboff[g] <= q  (suboff is strictly increasing: every bucket has >= 1 part)
    size_t lo = 0, hi = G - 1;
    while (lo < hi) {
        size_t mid = (lo + hi + 1) >> 1;
        if (suboff[mid] <= q) lo = mid; else hi = mid - 1;
    }
    const size_t g = lo;
    const uint32_t part = (uint32_t)(q - suboff[g]);
    const uint32_t parts = suboff[g + 1] - suboff[g];
    const uint32_t b0 = off[g], b1 = off[g + 1];
    uint32_t start = b0 + part * T;
    uint32_t end = start + T < b1 ? start + T : b1;
    if (parts == 1) end = b1;
    XYZZ<F> acc = XYZZ<F>::inf();
    for (uint32_t i = start; i < end; i++) {
        uint32_t e = sorted[i];
        Affine<F> p = bases[e & 0x7fffffffu];
        xyzz_madd(acc, p, (e >> 31) != 0);
    }
    if (parts == 1) buckets[g] = acc; else partials[q] = acc;
}

// workgroup tree reduction of one XYZZ per thread; result valid in thread 0.  sh: NT entries.
template <class F, unsigned NT>
__device__ __forceinline__ XYZZ<F> block_sum(XYZZ<F> acc, XYZZ<F> *sh) {
    unsigned t = threadIdx.x;
    sh[t] = acc;
    __syncthreads();
    for (unsigned s = NT / 2; s > 0; s >>= 1) {
        if (t < s) { xyzz_add(acc, sh[t + s]); sh[t] = acc; }
        __syncthreads();
    }
    return acc;
}

template <class F> struct MsmBlock { static constexpr unsigned NT = sizeof(F) > 48 ? 128 : 256; };  // 48 KiB of LDS either way

// merge: one workgroup per heavy bucket folds its partials
template <class F>
__global__ __launch_bounds__(MsmBlock<F>::NT) void k_merge(const uint32_t *heavy, const uint32_t *counters, const uint32_t *suboff,
                                                            const XYZZ<F> *partials, XYZZ<F> *buckets) {
    constexpr unsigned NT = MsmBlock<F>::NT;
    __shared__ XYZZ<F> sh[NT];
    uint32_t nh = counters[0];
    for (uint32_t h = blockIdx.x; h < nh; h += gridDim.x) {
        uint32_t g = heavy[h];
        uint32_t q0 = suboff[g], q1 = suboff[g + 1];
        XYZZ<F> acc = XYZZ<F>::inf();
        for (uint32_t q = q0 + threadIdx.x; q < q1; q += NT) xyzz_add(acc, partials[q]);
        acc = block_sum<F, NT>(acc, sh);
        if (threadIdx.x == 0) buckets[g] = acc;
        __syncthreads();
    }
}

// dimension sums: grid = W * (2^q2 + 2^q1 + 2^q0) workgroups.  dims[w][slot], slot = d-offset + v
template <class F>
__global__ __launch_bounds__(MsmBlock<F>::NT) void k_dimsum(const XYZZ<F> *buckets, MsmGeom g, XYZZ<F> *dims) {
    constexpr unsigned NT = MsmBlock<F>::NT;
    __shared__ XYZZ<F> sh[NT];
    const unsigned n0 = 1u << g.q0, n1 = 1u << g.q1, n2 = 1u << g.q2;
    const unsigned per_w = n0 + n1 + n2;
    const unsigned w = blockIdx.x / per_w;
    unsigned slot = blockIdx.x % per_w;
    unsigned d, v;
    if (slot < n0) { d = 0; v = slot; } else if (slot < n0 + n1) { d = 1; v = slot - n0; } else { d = 2; v = slot - n0 - n1; }
    const unsigned qd = d == 0 ? g.q0 : (d == 1 ? g.q1 : g.q2);
    const unsigned items = g.B >> qd;
    const XYZZ<F> *bw = buckets + (size_t)w * g.B;
    XYZZ<F> acc = XYZZ<F>::inf();
    for (unsigned i = threadIdx.x; i < items; i += NT) {
        unsigned idx;
        if (d == 0) idx = (i << g.q0) | v;
        else if (d == 1) idx = ((i >> g.q0) << (g.q1 + g.q0)) | (v << g.q0) | (i & (n0 - 1));
        else idx = (v << (g.q1 + g.q0)) | i;
        xyzz_add(acc, bw[idx]);
    }
    acc = block_sum<F, NT>(acc, sh);
    if (threadIdx.x == 0) dims[(size_t)w * per_w + slot] = acc;
}

// per (window, digit): D = sum_v v * A[v]  and  Tot = sum_v A[v]  by suffix scan + tree sum in LDS.
// winres[w][d] = D_d (d = 0,1,2), winres[w][3] = Tot
template <class F>
__global__ __launch_bounds__(MsmBlock<F>::NT) void k_dimweight(const XYZZ<F> *dims, MsmGeom g, XYZZ<F> *winres) {
    constexpr unsigned NT = MsmBlock<F>::NT;
    __shared__ XYZZ<F> sh[NT];
    const unsigned n0 = 1u << g.q0, n1 = 1u << g.q1, n2 = 1u << g.q2;
    const unsigned per_w = n0 + n1 + n2;
    const unsigned w = blockIdx.x / 3, d = blockIdx.x % 3;
    const unsigned cntv = d == 0 ? n0 : (d == 1 ? n1 : n2);
    const unsigned doff = d == 0 ? 0 : (d == 1 ? n0 : n0 + n1);
    const unsigned t = threadIdx.x;
    XYZZ<F> x = t < cntv ? dims[(size_t)w * per_w + doff + t] : XYZZ<F>::inf();
    sh[t] = x;
    __syncthreads();
    for (unsigned s = 1; s < cntv; s <<= 1) {            // inclusive suffix scan: x[t] = sum_{u >= t} A[u]
        XYZZ<F> y = (t + s < NT) ? sh[t + s] : XYZZ<F>::inf();
        __syncthreads();
        xyzz_add(x, y);
        sh[t] = x;
        __syncthreads();
    }
    XYZZ<F> tot = sh[0];
    __syncthreads();
    XYZZ<F> acc = (t >= 1 && t < cntv) ? x : XYZZ<F>::inf();   // sum_{v>=1} suffix[v] = sum_v v*A[v]
    acc = block_sum<F, NT>(acc, sh);
    if (t == 0) {
        winres[(size_t)w * 4 + d] = acc;
        if (d == 0) winres[(size_t)w * 4 + 3] = tot;
    }
}

// canonical affine (host layout) -> Montgomery affine
template <class F> __global__ void k_bases_to_mont(const Affine<F> *in, Affine<F> *out, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Affine<F> p = in[i];
    p.x = to_mont(p.x); p.y = to_mont(p.y);
    out[i] = p;
}

// ------------------------------------------------------------------------------------------------
static unsigned pick_window_bits(vsp_ctx *ctx, size_t n) {
    long forced = 0;
    auto it = ctx->opts.find("msm_window_bits");
    if (it != ctx->opts.end()) forced = it->second;
    if (forced >= 2 && forced <= 20) return (unsigned)forced;
    unsigned L = ceil_log2(n ? n : 1);
    // mean bucket load n / 2^(c-1) of about 32 points balances the accumulation against the
    // bucket reduction, whose cost grows with 2^c
    int c = (int)L - 4;
    if (c < 4) c = 4;
    if (c > 16) c = 16;
    return (unsigned)c;
}

template <class F, class HF>
static int msm_device(vsp_ctx *ctx, const Affine<F> *d_bases, const Fr *d_scalars, size_t n, XYZZ<HF> *out) {
    static_assert(sizeof(F) == sizeof(HF), "host/device field layouts must match");
    *out = XYZZ<HF>::inf();
    if (n == 0) return VSP_OK;
    if (n >= ((size_t)1 << 31)) return set_error(ctx, VSP_ERR_UNSUPPORTED, "msm: n >= 2^31");
    MsmGeom g;
    g.c = pick_window_bits(ctx, n);
    g.W = 255 / g.c + 1;
    g.B = 1u << (g.c - 1);
    unsigned qb = g.c - 1;
    g.q0 = qb / 3; g.q1 = (qb - g.q0) / 2; g.q2 = qb - g.q0 - g.q1;
    g.n = n; g.G = (size_t)g.W * g.B;
    {
        long t = 0; auto it = ctx->opts.find("msm_split"); if (it != ctx->opts.end()) t = it->second;
        size_t mean = n / g.B + 1;
        g.T = t > 0 ? (unsigned)t : (unsigned)(mean * 4 < 128 ? 128 : mean * 4);
    }
    constexpr unsigned NT = MsmBlock<F>::NT;
    if ((1u << g.q2) > NT) return set_error(ctx, VSP_ERR_UNSUPPORTED, "msm: window too wide");
    const size_t M = n * g.W;                               // upper bound on sorted entries
    const size_t Smax = g.G + M / g.T + 1;                   // upper bound on bucket parts
    const unsigned per_w = (1u << g.q0) + (1u << g.q1) + (1u << g.q2);

    VSP_TRY(ensure(ctx, ctx->msm_cnt, (g.G + 1) * 4));
    VSP_TRY(ensure(ctx, ctx->msm_off, (g.G + 1) * 4));
    VSP_TRY(ensure(ctx, ctx->msm_cursor, (g.G + 1) * 4));
    VSP_TRY(ensure(ctx, ctx->msm_nsub, (g.G + 1) * 4));
    VSP_TRY(ensure(ctx, ctx->msm_suboff, (g.G + 1) * 4));
    VSP_TRY(ensure(ctx, ctx->msm_heavy, (g.G + 1) * 4));
    VSP_TRY(ensure(ctx, ctx->msm_counters, 64));
    VSP_TRY(ensure(ctx, ctx->msm_sorted, M * 4));
    VSP_TRY(ensure(ctx, ctx->msm_buckets, g.G * sizeof(XYZZ<F>)));
    VSP_TRY(ensure(ctx, ctx->msm_partials, Smax * sizeof(XYZZ<F>)));
    VSP_TRY(ensure(ctx, ctx->msm_dims, (size_t)g.W * per_w * sizeof(XYZZ<F>)));
    VSP_TRY(ensure(ctx, ctx->msm_winres, (size_t)g.W * 4 * sizeof(XYZZ<F>)));
    uint32_t *cnt = (uint32_t *)ctx->msm_cnt.p, *off = (uint32_t *)ctx->msm_off.p, *cursor = (uint32_t *)ctx->msm_cursor.p;
    uint32_t *nsub = (uint32_t *)ctx->msm_nsub.p, *suboff = (uint32_t *)ctx->msm_suboff.p, *heavy = (uint32_t *)ctx->msm_heavy.p;
    uint32_t *counters = (uint32_t *)ctx->msm_counters.p, *sorted = (uint32_t *)ctx->msm_sorted.p;
    XYZZ<F> *buckets = (XYZZ<F> *)ctx->msm_buckets.p, *partials = (XYZZ<F> *)ctx->msm_partials.p;
    XYZZ<F> *dims = (XYZZ<F> *)ctx->msm_dims.p, *winres = (XYZZ<F> *)ctx->msm_winres.p;
    hipStream_t st = ctx->stream;

    VSP_HIP(hipMemsetAsync(cnt, 0, (g.G + 1) * 4, st));
    VSP_HIP(hipMemsetAsync(counters, 0, 64, st));
    const unsigned nblk = (unsigned)((n + MSM_THREADS - 1) / MSM_THREADS);
    const unsigned gblk = (unsigned)((g.G + MSM_THREADS - 1) / MSM_THREADS);
    hipLaunchKernelGGL(k_count, dim3(nblk), dim3(MSM_THREADS), 0, st, d_scalars, g, cnt);
    VSP_LAUNCH_CHECK();
    VSP_TRY(exclusive_scan(ctx, cnt, g.G, off, cursor));
    hipLaunchKernelGGL(k_plan, dim3(gblk), dim3(MSM_THREADS), 0, st, (const uint32_t *)cnt, g.G, g.T, nsub, heavy, counters);
    VSP_LAUNCH_CHECK();
    VSP_TRY(exclusive_scan(ctx, nsub, g.G, suboff, nullptr));
    hipLaunchKernelGGL(k_scatter, dim3(nblk), dim3(MSM_THREADS), 0, st, d_scalars, g, cursor, sorted);
    VSP_LAUNCH_CHECK();

    const unsigned ablk = (unsigned)((Smax + MSM_THREADS - 1) / MSM_THREADS);
    VSP_HIP(hipEventRecord(ctx->ev0, st));
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_accum<F>), dim3(ablk), dim3(MSM_THREADS), 0, st, d_bases, (const uint32_t *)sorted,
                       (const uint32_t *)off, (const uint32_t *)suboff, g.G, g.T, buckets, partials);
    VSP_HIP(hipEventRecord(ctx->ev1, st));
    VSP_LAUNCH_CHECK();
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_merge<F>), dim3(1024), dim3(NT), 0, st, (const uint32_t *)heavy, (const uint32_t *)counters,
                       (const uint32_t *)suboff, (const XYZZ<F> *)partials, buckets);
    VSP_LAUNCH_CHECK();
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_dimsum<F>), dim3(g.W * per_w), dim3(NT), 0, st, (const XYZZ<F> *)buckets, g, dims);
    VSP_LAUNCH_CHECK();
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_dimweight<F>), dim3(g.W * 3), dim3(NT), 0, st, (const XYZZ<F> *)dims, g, winres);
    VSP_LAUNCH_CHECK();

    std::vector<XYZZ<HF>> wr((size_t)g.W * 4);
    VSP_HIP(hipMemcpyAsync(wr.data(), winres, wr.size() * sizeof(XYZZ<HF>), hipMemcpyDeviceToHost, st));
    VSP_HIP(hipStreamSynchronize(st));
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1) == hipSuccess) {
        ctx->stats["msm_accum_ms"] += ms;
        ctx->stats["msm_accum_launches"] += 1;
    }
    ctx->stats["msm_window_bits"] = g.c;
    ctx->stats["msm_windows"] = g.W;
    ctx->stats["msm_split"] = g.T;

    // Horner over bit positions: window w contributes D2 at c*w + q1 + q0, D1 at c*w + q0, D0 + Tot at c*w
    XYZZ<HF> acc = XYZZ<HF>::inf();
    for (int w = (int)g.W - 1; w >= 0; w--) {
        for (unsigned i = 0; i < g.q2 + 1; i++) acc = xyzz_dbl(acc);
        xyzz_add(acc, wr[(size_t)w * 4 + 2]);
        for (unsigned i = 0; i < g.q1; i++) acc = xyzz_dbl(acc);
        xyzz_add(acc, wr[(size_t)w * 4 + 1]);
        for (unsigned i = 0; i < g.q0; i++) acc = xyzz_dbl(acc);
        xyzz_add(acc, wr[(size_t)w * 4 + 0]);
        xyzz_add(acc, wr[(size_t)w * 4 + 3]);
    }
    *out = acc;
    return VSP_OK;
}

int msm_g1_device(vsp_ctx *ctx, const G1Affine *d_bases, const Fr *d_scalars, size_t n, XYZZ<HFp> *out) {
    return msm_device<Fp, HFp>(ctx, d_bases, d_scalars, n, out);
}
int msm_g2_device(vsp_ctx *ctx, const G2Affine *d_bases, const Fr *d_scalars, size_t n, XYZZ<HFp2> *out) {
    return msm_device<Fp2, HFp2>(ctx, d_bases, d_scalars, n, out);
}

int bases_to_mont_g1(vsp_ctx *ctx, const void *d_canon, G1Affine *d_out, size_t n) {
    if (!n) return VSP_OK;
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_bases_to_mont<Fp>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream,
                       (const G1Affine *)d_canon, d_out, n);
    VSP_LAUNCH_CHECK();
    return VSP_OK;
}
int bases_to_mont_g2(vsp_ctx *ctx, const void *d_canon, G2Affine *d_out, size_t n) {
    if (!n) return VSP_OK;
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_bases_to_mont<Fp2>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream,
                       (const G2Affine *)d_canon, d_out, n);
    VSP_LAUNCH_CHECK();
    return VSP_OK;
}

}  // namespace vsp
