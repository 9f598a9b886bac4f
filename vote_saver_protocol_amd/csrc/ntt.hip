// Radix-2 NTT over BLS12-381 Fr for gfx950 (the step radix-2 domain and witness_map built on it are in domain.hip).
//
// Replaces math::evaluation_domain<Fr> / basic_radix2_domain (fft, inverse_fft, coset variants,
// divide_by_z_on_coset) of crypto3-math -- absent submodule, /root/reference/.gitmodules:47-48;
// reached from bin/cli/include/nil/vote_saver/common.hpp:1132-1135 via the prover's witness_map.
// The serial reference algorithm is bit-reverse + log2(m) butterfly stages over one array
// (libfqfft _basic_serial_radix2_FFT); the output is the DFT, unique given omega.
//
// MI355X design:
//  * The log2(m) stages are grouped into passes of <= 8 stages.  One workgroup (512 threads) owns a
//    tile of 2048 elements (64 KiB of LDS, two workgroups per CU), runs all of the pass's stages out
//    of LDS and touches HBM exactly once for read and once for write: traffic per pass = 64 B/element,
//    2-4 passes per transform (m = 2^22: 3 passes, 8+8+6 stages).  Inside a pass the stages run as radix-4 steps: two stages per
//    LDS round trip and barrier.
//  * A tile is [2^K butterfly rows] x [C = 2048/2^K contiguous columns]; every global access is a run
//    of C*32 bytes >= 256 B.  The first pass reads through the bit-reversal permutation (runs of C
//    inputs) and writes natural order, so no separate permutation pass exists.
//  * Values stay CANONICAL in HBM and LDS: a Montgomery product of a canonical value with a
//    Montgomery-form twiddle is already the canonical product, so no domain conversion is ever done.
//  * LDS holds each 32-byte element as two 16-byte halves in separate planes: a wave's ds_read_b128 /
//    ds_write_b128 then covers 64 x 16 contiguous bytes (conflict-free), instead of a 32-byte stride.
//  * Twiddles come from one table omega^j, j < m/2 (64 MiB at m = 2^22, Infinity-Cache resident);
//    the table of the largest domain seen serves every smaller one by striding.
//  * Coset shift (a[j] *= g^j), inverse scaling (m^-1) and inverse coset shift are fused into the
//    first pass's load / the last pass's store.
//  * The path is integer-ALU bound (one 8-limb Montgomery product per butterfly), not HBM bound.
#include <algorithm>
#include "common.h"
#include <type_traits>
#include "fr29.h"

namespace vsp {

#ifndef VSP_NTT_TILE_LOG
#define VSP_NTT_TILE_LOG 11
#define VSP_NTT_THREADS 512
#define VSP_NTT_MAX_STAGES 8
#endif
static constexpr unsigned NTT_TILE_LOG = VSP_NTT_TILE_LOG;      // 2048 elements per workgroup
static constexpr unsigned NTT_THREADS = VSP_NTT_THREADS;     // 2 workgroups per CU (64 KiB tiles) -> 4 waves per SIMD (256: 0.81 ms, 512: 0.74 ms, 1024: 0.84 ms at 2^22)
static constexpr unsigned NTT_MAX_STAGES = VSP_NTT_MAX_STAGES;     // per pass (keeps C >= 8 columns = 256 B runs)
static constexpr unsigned PW_LOG = 11;            // two-level power tables: g^i = lo[i & 2047] * hi[i >> 11]

struct NttPassArgs {
    unsigned log_n, s0, s1, clog, tlog;
    int first, last;
    int premul;     // multiply input i by g^i (forward coset), first pass only
    int postmul;    // last pass only: 0 none, 1 constant `scale`, 2 scale * ginv^pos
    const Fr *tw;
    const Fr *pw_lo, *pw_hi;
    Fr scale;       // Montgomery form
};

__device__ __forceinline__ unsigned brev(unsigned v, unsigned bits) { return bits ? (__brev(v) >> (32 - bits)) : 0u; }

__device__ __forceinline__ Fr lds_load(const uint4 *pl0, const uint4 *pl1, unsigned e) {
    Fr v; uint4 a = pl0[e], b = pl1[e];
    v.l[0] = a.x; v.l[1] = a.y; v.l[2] = a.z; v.l[3] = a.w; v.l[4] = b.x; v.l[5] = b.y; v.l[6] = b.z; v.l[7] = b.w;
    return v;
}
__device__ __forceinline__ void lds_store(uint4 *pl0, uint4 *pl1, unsigned e, const Fr &v) {
    pl0[e] = make_uint4(v.l[0], v.l[1], v.l[2], v.l[3]);
    pl1[e] = make_uint4(v.l[4], v.l[5], v.l[6], v.l[7]);
}

__global__ __launch_bounds__(NTT_THREADS) void k_ntt_pass(const Fr *__restrict__ in, Fr *__restrict__ out, NttPassArgs p) {
    __shared__ uint4 lds[2u << NTT_TILE_LOG];
    const unsigned K = p.s1 - p.s0;
    const unsigned C = 1u << p.clog;
    const unsigned tile = 1u << (K + p.clog);
    uint4 *pl0 = lds, *pl1 = lds + tile;
    const unsigned tid = threadIdx.x;
    const unsigned wg = blockIdx.x;
    // position pieces for passes after the first: pos = hi<<s1 | mid<<s0 | lo_hi<<clog | c
    const unsigned lo_bits = p.s0 - p.clog;                   // only meaningful when !first
    const unsigned lo_hi = p.first ? 0u : (wg & ((1u << lo_bits) - 1u));
    const unsigned hi = p.first ? 0u : (wg >> lo_bits);
    const size_t base_pos = p.first ? 0 : (((size_t)hi << p.s1) | ((size_t)lo_hi << p.clog));

    // ---- load tile (LDS index e = mid*C + c; consecutive threads -> consecutive c -> contiguous HBM)
    for (unsigned e = tid; e < tile; e += NTT_THREADS) {
        unsigned mid = e >> p.clog, c = e & (C - 1);
        size_t src;
        if (p.first) src = ((size_t)brev(mid, p.s1) << (p.log_n - p.s1)) + (size_t)wg * C + c;
        else src = base_pos | ((size_t)mid << p.s0) | c;
        Fr v = in[src];
        if (p.premul) {
            Fr g = mul(p.pw_lo[src & ((1u << PW_LOG) - 1u)], p.pw_hi[src >> PW_LOG]);
            v = mul(v, g);
        }
        lds_store(pl0, pl1, e, v);
    }
    __syncthreads();

    // ---- K butterfly stages out of LDS: an odd K starts with one radix-2 stage, the rest run as radix-4 steps (two stages per
    //      LDS round trip and per barrier; a thread owns the four elements mid0 + {0, h, 2h, 3h} of one column)
    const unsigned lo_part = p.first ? 0u : ((lo_hi << p.clog));
    unsigned t = 0;
    if (K & 1) {
        const unsigned s = p.s0;              // global stage: half-size 2^s, twiddle omega_{2^(s+1)}^j
        for (unsigned bf = tid; bf < (tile >> 1); bf += NTT_THREADS) {
            unsigned c = bf & (C - 1), q = bf >> p.clog;
            unsigned e0 = (q << (1 + p.clog)) | c, e1 = e0 + C;
            Fr u = lds_load(pl0, pl1, e0), v = lds_load(pl0, pl1, e1);
            if (s > 0) v = mul(v, p.tw[(size_t)(p.first ? 0u : (lo_part | c)) << (p.tlog - 1 - s)]);
            lds_store(pl0, pl1, e0, add(u, v));
            lds_store(pl0, pl1, e1, sub(u, v));
        }
        __syncthreads();
        t = 1;
    }
    for (; t < K; t += 2) {
        const unsigned s = p.s0 + t;          // stages s and s + 1
        const unsigned h = 1u << t;
        for (unsigned g = tid; g < (tile >> 2); g += NTT_THREADS) {
            unsigned c = g & (C - 1), q = g >> p.clog;
            unsigned mid_lo = q & (h - 1);
            unsigned mid0 = ((q >> t) << (t + 2)) | mid_lo;
            unsigned e0 = (mid0 << p.clog) | c, e1 = e0 + (h << p.clog), e2 = e1 + (h << p.clog), e3 = e2 + (h << p.clog);
            const size_t off = p.first ? 0u : (lo_part | c);
            Fr x0 = lds_load(pl0, pl1, e0), x1 = lds_load(pl0, pl1, e1), x2 = lds_load(pl0, pl1, e2), x3 = lds_load(pl0, pl1, e3);
            if (s > 0) {
                const Fr w1 = p.tw[(((size_t)mid_lo << p.s0) | off) << (p.tlog - 1 - s)];
                x1 = mul(x1, w1); x3 = mul(x3, w1);
            }
            Fr a0 = add(x0, x1), a1 = sub(x0, x1), a2 = add(x2, x3), a3 = sub(x2, x3);
            a2 = mul(a2, p.tw[(((size_t)mid_lo << p.s0) | off) << (p.tlog - 2 - s)]);
            a3 = mul(a3, p.tw[(((size_t)(mid_lo + h) << p.s0) | off) << (p.tlog - 2 - s)]);
            lds_store(pl0, pl1, e0, add(a0, a2));
            lds_store(pl0, pl1, e1, add(a1, a3));
            lds_store(pl0, pl1, e2, sub(a0, a2));
            lds_store(pl0, pl1, e3, sub(a1, a3));
        }
        __syncthreads();
    }

    // ---- store tile
    for (unsigned i = tid; i < tile; i += NTT_THREADS) {
        unsigned mid, c; size_t pos;
        if (p.first) {              // natural-order output: runs of 2^K contiguous elements per column
            mid = i & ((1u << K) - 1u); c = i >> K;
            pos = ((size_t)brev(wg * C + c, p.log_n - p.s1) << p.s1) | mid;
        } else {
            mid = i >> p.clog; c = i & (C - 1);
            pos = base_pos | ((size_t)mid << p.s0) | c;
        }
        Fr v = lds_load(pl0, pl1, (mid << p.clog) | c);
        if (p.last && p.postmul) {
            Fr m = p.scale;
            if (p.postmul == 2) m = mul(m, mul(p.pw_lo[pos & ((1u << PW_LOG) - 1u)], p.pw_hi[pos >> PW_LOG]));
            v = mul(v, m);
        }
        out[pos] = v;
    }
}

// ================================================================================================
// The same pass on 9 x 29-bit limbs with lazy reduction (fr29.h): ~1.4 x fewer instructions per butterfly.  Data enters canonical
// (8 x 32-bit words), stays "lazy" (congruent, tight limbs, value below 48 r) in LDS and -- between passes -- in the scratch buffer
// (three planes: limbs 0-3, limbs 4-7, limb 8: 36 bytes per element), and leaves canonical from the last pass, where the product
// with the scale (1/m, the extra scale, the inverse coset power -- or the Montgomery one when there is none) brings it below 2r.
// Twiddles and coset powers come from tables in Montgomery form for R' = 2^261, canonical, in the same three planes.
struct Planes29 { const uint4 *p0, *p1; const uint32_t *p2; };
struct PlanesOut29 { uint4 *p0, *p1; uint32_t *p2; };
struct NttPass29Args {
    unsigned log_n, s0, s1, clog, tlog;
    int first, last, premul, postmul;
    Planes29 tw, pw_lo, pw_hi;
    Fr29 scale;             // Montgomery (R') form, canonical; the Montgomery one when no scaling is asked for
    // a BATCH of up to three transforms of one size in one launch (grid.y; r1cs_to_qap::witness_map runs three at every step): transform b
    // reads the canonical words of in[b] in its first pass, writes those of out[b] in its last, and keeps its lazy values between passes
    // in the slice [b 2^log_n, (b + 1) 2^log_n) of every scratch plane
    const Fr *in[3];
    Fr *out[3];
    size_t in_stride, out_stride;      // != 0: transform b reads in[0] + b in_stride and writes out[0] + b out_stride (any number of transforms: a batch of proofs)
    // fused load of the first pass (witness_map's last transform): the value transformed is in[0][i] * fuse_b[i] - fuse_c[i], left with the
    // factor 2^-261 of the two products (the caller folds 2^261 into the scale); nullptr: plain load
    const Fr *fuse_b, *fuse_c;
};
__device__ __forceinline__ Fr29 ld29(const Planes29 &t, size_t j) {
    Fr29 v; uint4 a = t.p0[j], b = t.p1[j];
    v.l[0] = a.x; v.l[1] = a.y; v.l[2] = a.z; v.l[3] = a.w; v.l[4] = b.x; v.l[5] = b.y; v.l[6] = b.z; v.l[7] = b.w; v.l[8] = t.p2[j];
    return v;
}
__device__ __forceinline__ Fr29 lds_load29(const uint4 *pl0, const uint4 *pl1, const uint32_t *pl2, unsigned e) {
    Fr29 v; uint4 a = pl0[e], b = pl1[e];
    v.l[0] = a.x; v.l[1] = a.y; v.l[2] = a.z; v.l[3] = a.w; v.l[4] = b.x; v.l[5] = b.y; v.l[6] = b.z; v.l[7] = b.w; v.l[8] = pl2[e];
    return v;
}
__device__ __forceinline__ void lds_store29(uint4 *pl0, uint4 *pl1, uint32_t *pl2, unsigned e, const Fr29 &v) {
    pl0[e] = make_uint4(v.l[0], v.l[1], v.l[2], v.l[3]);
    pl1[e] = make_uint4(v.l[4], v.l[5], v.l[6], v.l[7]);
    pl2[e] = v.l[8];
}

#ifdef VSP_DIAG_CLOCK
__device__ unsigned long long vsp_diag_ntt_sums[4096][4];    // per workgroup slot: shader cycles, 100 MHz ticks, waves (diagnostic build only; ONE address for all waves cost 390 us per pass)
#endif
__global__ __launch_bounds__(NTT_THREADS) void k_ntt29_pass(Planes29 in_lazy, PlanesOut29 out_lazy, NttPass29Args p) {
#if defined(__HIP_DEVICE_COMPILE__)
    __shared__ uint4 lds[2u << NTT_TILE_LOG];
    __shared__ uint32_t lds8[1u << NTT_TILE_LOG];
    const Fr *__restrict__ in_words = p.in_stride ? p.in[0] + (size_t)blockIdx.y * p.in_stride : p.in[blockIdx.y];
    Fr *__restrict__ out_words = p.out_stride ? p.out[0] + (size_t)blockIdx.y * p.out_stride : p.out[blockIdx.y];
    const size_t fuse_off = (size_t)blockIdx.y * p.in_stride;      // the fused load's other two operands follow their transform
    { const size_t boff = (size_t)blockIdx.y << p.log_n;          // this transform's slice of the scratch planes
      in_lazy.p0 += boff; in_lazy.p1 += boff; in_lazy.p2 += boff; out_lazy.p0 += boff; out_lazy.p1 += boff; out_lazy.p2 += boff; }
#ifdef VSP_DIAG_CLOCK
    // DIAGNOSTIC BUILD ONLY (libvsp_hip_diag.so): the clock the chip holds inside a pass, stamped once per wave around the whole pass
    unsigned long long dc_t0, dc_r0, dc_t1, dc_r1;
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(dc_t0), "=s"(dc_r0) :: "memory");
#endif
    const unsigned K = p.s1 - p.s0;
    const unsigned C = 1u << p.clog;
    const unsigned tile = 1u << (K + p.clog);
    uint4 *pl0 = lds, *pl1 = lds + tile;
    uint32_t *pl2 = lds8;
    const unsigned tid = threadIdx.x;
    const unsigned wg = blockIdx.x;
    const unsigned lo_bits = p.s0 - p.clog;                   // only meaningful when !first
    const unsigned lo_hi = p.first ? 0u : (wg & ((1u << lo_bits) - 1u));
    const unsigned hi = p.first ? 0u : (wg >> lo_bits);
    const size_t base_pos = p.first ? 0 : (((size_t)hi << p.s1) | ((size_t)lo_hi << p.clog));

    // ---- load tile
    for (unsigned e = tid; e < tile; e += NTT_THREADS) {
        unsigned mid = e >> p.clog, c = e & (C - 1);
        size_t src;
        if (p.first) src = ((size_t)brev(mid, p.s1) << (p.log_n - p.s1)) + (size_t)wg * C + c;
        else src = base_pos | ((size_t)mid << p.s0) | c;
        Fr29 v;
        if (p.first) {
            v = fr29_from_words(in_words[src]);
            if (p.fuse_b) {                                       // (a b - c) / 2^261: two products of canonical operands, one lazy subtraction, one carry pass
                Fr29 one; for (int k = 0; k < 9; k++) one.l[k] = k == 0 ? 1u : 0u;
                v = norm29(sub29(mul29(v, fr29_from_words(p.fuse_b[fuse_off + src])), mul29(fr29_from_words(p.fuse_c[fuse_off + src]), one)));
            }
            if (p.premul) v = mul29(v, mul29(ld29(p.pw_lo, src & ((1u << PW_LOG) - 1u)), ld29(p.pw_hi, src >> PW_LOG)));
        } else v = ld29(in_lazy, src);
        lds_store29(pl0, pl1, pl2, e, v);
    }
    __syncthreads();

    const unsigned lo_part = p.first ? 0u : ((lo_hi << p.clog));
    unsigned t = 0;
    if (K & 1) {
        const unsigned s = p.s0;
        for (unsigned bf = tid; bf < (tile >> 1); bf += NTT_THREADS) {
            unsigned c = bf & (C - 1), q = bf >> p.clog;
            unsigned e0 = (q << (1 + p.clog)) | c, e1 = e0 + C;
            Fr29 u = lds_load29(pl0, pl1, pl2, e0), v = lds_load29(pl0, pl1, pl2, e1);
            if (s > 0) v = mul29(v, ld29(p.tw, (size_t)(p.first ? 0u : (lo_part | c)) << (p.tlog - 1 - s)));
            lds_store29(pl0, pl1, pl2, e0, norm29(add29(u, v)));
            lds_store29(pl0, pl1, pl2, e1, norm29(sub29(u, v)));       // s == 0 only in the first pass: v is a canonical input there
        }
        __syncthreads();
        t = 1;
    }
    for (; t < K; t += 2) {
        const unsigned s = p.s0 + t;
        const unsigned h = 1u << t;
        // the first pair of products exists from stage 1 on (stage 0's twiddle is one).  Two copies of the loop, not a branch inside it:
        // a value that is "the loaded x1 or the product" is ONE register set for the compiler, which then loads x1 into the routine's
        // result registers and copies it to the operand registers before every call (nine copies per product, section 3.2 of DESIGN.md)
        auto group = [&](unsigned g, auto with_first_pair) {
            constexpr bool TW1 = decltype(with_first_pair)::value;
            unsigned c = g & (C - 1), q = g >> p.clog;
            unsigned mid_lo = q & (h - 1);
            unsigned mid0 = ((q >> t) << (t + 2)) | mid_lo;
            unsigned e0 = (mid0 << p.clog) | c, e1 = e0 + (h << p.clog), e2 = e1 + (h << p.clog), e3 = e2 + (h << p.clog);
            const size_t off = p.first ? 0u : (lo_part | c);
            Fr29 x1 = lds_load29(pl0, pl1, pl2, e1), x3 = lds_load29(pl0, pl1, pl2, e3);
            if constexpr (TW1) {
                const Fr29 w1 = ld29(p.tw, (((size_t)mid_lo << p.s0) | off) << (p.tlog - 1 - s));
                x1 = mul29(x1, w1); x3 = mul29q(x3, w1);
            }
            Fr29 x0 = lds_load29(pl0, pl1, pl2, e0), x2 = lds_load29(pl0, pl1, pl2, e2);
            Fr29 a0 = add29(x0, x1), a1 = sub29(x0, x1), a2 = add29(x2, x3), a3 = sub29(x2, x3);
            a2 = mul29(a2, ld29(p.tw, (((size_t)mid_lo << p.s0) | off) << (p.tlog - 2 - s)));
            a3 = mul29q(a3, ld29(p.tw, (((size_t)(mid_lo + h) << p.s0) | off) << (p.tlog - 2 - s)));
            lds_store29(pl0, pl1, pl2, e0, norm29(add29(a0, a2)));
            lds_store29(pl0, pl1, pl2, e1, norm29(add29(a1, a3)));
            lds_store29(pl0, pl1, pl2, e2, norm29(sub29(a0, a2)));
            lds_store29(pl0, pl1, pl2, e3, norm29(sub29(a1, a3)));
        };
        if (s > 0) for (unsigned g = tid; g < (tile >> 2); g += NTT_THREADS) group(g, std::true_type{});
        else for (unsigned g = tid; g < (tile >> 2); g += NTT_THREADS) group(g, std::false_type{});
        __syncthreads();
    }

    // ---- store tile
    for (unsigned i = tid; i < tile; i += NTT_THREADS) {
        unsigned mid, c; size_t pos;
        if (p.first) { mid = i & ((1u << K) - 1u); c = i >> K; pos = ((size_t)brev(wg * C + c, p.log_n - p.s1) << p.s1) | mid; }
        else { mid = i >> p.clog; c = i & (C - 1); pos = base_pos | ((size_t)mid << p.s0) | c; }
        Fr29 v = lds_load29(pl0, pl1, pl2, (mid << p.clog) | c);
        if (p.last) {
            Fr29 m = p.scale;
            if (p.postmul == 2) m = mul29(m, mul29(ld29(p.pw_lo, pos & ((1u << PW_LOG) - 1u)), ld29(p.pw_hi, pos >> PW_LOG)));
            out_words[pos] = fr29_to_words(csub29(mul29(v, m)));
        } else {
            out_lazy.p0[pos] = make_uint4(v.l[0], v.l[1], v.l[2], v.l[3]);
            out_lazy.p1[pos] = make_uint4(v.l[4], v.l[5], v.l[6], v.l[7]);
            out_lazy.p2[pos] = v.l[8];
        }
    }
#ifdef VSP_DIAG_CLOCK
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(dc_t1), "=s"(dc_r1) :: "memory");
    if ((threadIdx.x & 63u) == 0) {
        unsigned long long *slot = vsp_diag_ntt_sums[(blockIdx.x + 2048u * blockIdx.y) & 4095u];
        atomicAdd(&slot[0], dc_t1 - dc_t0); atomicAdd(&slot[1], dc_r1 - dc_r0); atomicAdd(&slot[2], 1ull);
    }
#endif
#endif
}

// 8 x 32-bit Montgomery table -> three planes in the R' = 2^261 form
__global__ __launch_bounds__(256) void k_table29(const Fr *in, size_t count, uint4 *p0, uint4 *p1, uint32_t *p2) {
#if defined(__HIP_DEVICE_COMPILE__)
    size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= count) return;
    Fr29 v = fr29_from_mont256(in[j]);
    p0[j] = make_uint4(v.l[0], v.l[1], v.l[2], v.l[3]); p1[j] = make_uint4(v.l[4], v.l[5], v.l[6], v.l[7]); p2[j] = v.l[8];
#endif
}

// T[j] = A[j & 2047] * B[j >> 11]  (all Montgomery)
__global__ __launch_bounds__(256) void k_fill_twiddles(Fr *T, const Fr *A, const Fr *B, size_t count) {
    size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j < count) T[j] = mul(A[j & ((1u << PW_LOG) - 1u)], B[j >> PW_LOG]);
}

// ---- host side ----------------------------------------------------------------------------------
// diagnostic build: clock held inside the passes of the 29-bit transform since the last reset
int ntt_diag_clock(vsp_ctx *ctx, int reset, double *ghz, double *waves) {
#ifdef VSP_DIAG_CLOCK
    std::vector<unsigned long long> all(4096 * 4, 0);
    VSP_HIP(hipStreamSynchronize(ctx->stream));
    VSP_HIP(hipMemcpyFromSymbol(all.data(), HIP_SYMBOL(vsp_diag_ntt_sums), all.size() * sizeof(unsigned long long)));
    unsigned long long h[4] = {0, 0, 0, 0};
    for (size_t i = 0; i < 4096; i++) for (int j = 0; j < 3; j++) h[j] += all[4 * i + j];
    if (ghz) *ghz = h[1] ? (double)h[0] / (double)h[1] * 0.1 : 0.0;
    if (waves) *waves = (double)h[2];
    if (reset) { std::fill(all.begin(), all.end(), 0ull); VSP_HIP(hipMemcpyToSymbol(HIP_SYMBOL(vsp_diag_ntt_sums), all.data(), all.size() * sizeof(unsigned long long))); }
    return VSP_OK;
#else
    (void)reset; if (ghz) *ghz = 0.0; if (waves) *waves = 0.0;
    return set_error(ctx, VSP_ERR_UNSUPPORTED, "diag_clock: this is not the diagnostic build (make diag -> libvsp_hip_diag.so)");
#endif
}
static const uint64_t FR_ROOT_2_32[4] = {0x3829971f439f0d2bULL, 0xb63683508c2280b9ULL, 0xd09b681922c813b4ULL, 0x16a2a19edfe81f20ULL};

HFr host_omega(unsigned log_m) {
    HFr w = host_load_canon<HFr>(FR_ROOT_2_32);
    for (unsigned i = log_m; i < 32; i++) w = sqr(w);
    return w;
}
static HFr host_from_u64(uint64_t v) { uint64_t c[4] = {v, 0, 0, 0}; return host_load_canon<HFr>(c); }

static Fr to_dev(const HFr &h) { Fr d; memcpy(&d, &h, sizeof(Fr)); return d; }

// lo[k] = base^k (k < 2^PW_LOG), hi[k] = base^(k << PW_LOG) (k < hi_count), Montgomery form
int upload_power_tables(vsp_ctx *ctx, const HFr &base, size_t hi_count, DevBuf &lo, DevBuf &hi) {
    const size_t L = (size_t)1 << PW_LOG;
    std::vector<HFr> a(L), b(hi_count);
    HFr acc = HFr::one();
    for (size_t k = 0; k < L; k++) { a[k] = acc; acc = mul(acc, base); }
    HFr step = acc;                      // base^(2^PW_LOG)
    acc = HFr::one();
    for (size_t k = 0; k < hi_count; k++) { b[k] = acc; acc = mul(acc, step); }
    VSP_TRY(ensure(ctx, lo, L * sizeof(Fr)));
    VSP_TRY(ensure(ctx, hi, hi_count * sizeof(Fr)));
    VSP_HIP(hipMemcpyAsync(lo.p, a.data(), L * sizeof(Fr), hipMemcpyHostToDevice, ctx->stream));
    VSP_HIP(hipMemcpyAsync(hi.p, b.data(), hi_count * sizeof(Fr), hipMemcpyHostToDevice, ctx->stream));
    VSP_HIP(hipStreamSynchronize(ctx->stream));     // a, b go out of scope
    return VSP_OK;
}

int ntt_ensure_twiddles(vsp_ctx *ctx, unsigned log_m) {
    NttTables &t = ctx->ntt;
    if (t.log >= log_m && t.fwd.p) return VSP_OK;
    unsigned lg = log_m < 1 ? 1 : log_m;
    size_t count = (size_t)1 << (lg - 1);
    size_t hi_count = count > ((size_t)1 << PW_LOG) ? (count >> PW_LOG) : 1;
    DevBuf lo, hi;
    for (int dir = 0; dir < 2; dir++) {
        HFr w = host_omega(lg);
        if (dir) w = inv(w);
        VSP_TRY(upload_power_tables(ctx, w, hi_count, lo, hi));
        DevBuf &dst = dir ? t.inv : t.fwd;
        VSP_TRY(ensure(ctx, dst, count * sizeof(Fr)));
        unsigned blocks = (unsigned)((count + 255) / 256);
        hipLaunchKernelGGL(k_fill_twiddles, dim3(blocks), dim3(256), 0, ctx->stream, (Fr *)dst.p, (const Fr *)lo.p, (const Fr *)hi.p, count);
        VSP_LAUNCH_CHECK();
    }
    VSP_HIP(hipStreamSynchronize(ctx->stream));
    hipFree(lo.p); hipFree(hi.p);
    t.log = lg;
    t.log29 = 0;                 // the 29-bit copies follow on demand (ntt29_ensure_tables)
    return VSP_OK;
}

int ntt_ensure_coset_tables(vsp_ctx *ctx, unsigned log_m, const uint64_t *g4) {
    NttTables &t = ctx->ntt;
    if (t.pw_valid && t.pw_log >= log_m && memcmp(t.pw_g, g4, 32) == 0) return VSP_OK;
    size_t n = (size_t)1 << log_m;
    size_t hi_count = n > ((size_t)1 << PW_LOG) ? (n >> PW_LOG) : 1;
    HFr g = host_load_canon<HFr>(g4);
    VSP_TRY(upload_power_tables(ctx, g, hi_count, t.pw_lo_f, t.pw_hi_f));
    VSP_TRY(upload_power_tables(ctx, inv(g), hi_count, t.pw_lo_i, t.pw_hi_i));
    memcpy(t.pw_g, g4, 32); t.pw_log = log_m; t.pw_valid = true;
    t.pw29_valid = false;
    return VSP_OK;
}

// the three planes of a 29-bit table of `count` entries stored at `base`
static Planes29 planes_of(const DevBuf &b, size_t count) {
    Planes29 pl; pl.p0 = (const uint4 *)b.p; pl.p1 = (const uint4 *)((const char *)b.p + 16 * count); pl.p2 = (const uint32_t *)((const char *)b.p + 32 * count);
    return pl;
}
static int convert_table29(vsp_ctx *ctx, const DevBuf &src, size_t count, DevBuf &dst) {
    VSP_TRY(ensure(ctx, dst, 36 * count));
    hipLaunchKernelGGL(k_table29, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, ctx->stream, (const Fr *)src.p, count,
                       (uint4 *)dst.p, (uint4 *)((char *)dst.p + 16 * count), (uint32_t *)((char *)dst.p + 32 * count));
    VSP_LAUNCH_CHECK();
    return VSP_OK;
}
// 29-bit copies of the twiddle tables (and of the coset power tables when a coset generator is in use), converted on the GPU
static int ntt29_ensure_tables(vsp_ctx *ctx, bool coset) {
    NttTables &t = ctx->ntt;
    if (t.log29 != t.log) {
        const size_t count = (size_t)1 << (t.log - 1);
        VSP_TRY(convert_table29(ctx, t.fwd, count, t.fwd29));
        VSP_TRY(convert_table29(ctx, t.inv, count, t.inv29));
        t.log29 = t.log;
    }
    if (coset && !(t.pw29_valid && t.pw29_log == t.pw_log && memcmp(t.pw29_g, t.pw_g, 32) == 0)) {
        const size_t L = (size_t)1 << PW_LOG, n = (size_t)1 << t.pw_log, H = n > L ? (n >> PW_LOG) : 1;
        VSP_TRY(convert_table29(ctx, t.pw_lo_f, L, t.pw29[0])); VSP_TRY(convert_table29(ctx, t.pw_hi_f, H, t.pw29[1]));
        VSP_TRY(convert_table29(ctx, t.pw_lo_i, L, t.pw29[2])); VSP_TRY(convert_table29(ctx, t.pw_hi_i, H, t.pw29[3]));
        memcpy(t.pw29_g, t.pw_g, 32); t.pw29_log = t.pw_log; t.pw29_valid = true;
    }
    return VSP_OK;
}
static Fr29 host_to_fr29_mont(const HFr &x) {          // host value (Montgomery, R = 2^256) -> canonical x * 2^261 mod r as 9 x 29-bit limbs
    uint64_t c[4]; host_store_canon(c, x);
    HFr sh = host_load_canon<HFr>(c);                   // x again, to multiply by 2^261 = 2^256 * 2^5 through host arithmetic
    HFr two = add(HFr::one(), HFr::one()), p32 = two;
    for (int i = 0; i < 4; i++) p32 = add(p32, p32);      // 2^5
    // 2^256 mod r in Montgomery form is R^2's reduction: to_mont(one's canonical R)...: simpler: multiply by 2 two hundred sixty-one times
    HFr acc = sh;
    for (int i = 0; i < 261; i++) acc = add(acc, acc);
    host_store_canon(c, acc);
    Fr29 r;
    for (int i = 0; i < 9; i++) {
        const int bit = 29 * i, w = bit >> 6, s2 = bit & 63;
        unsigned __int128 v = c[w];
        if (w + 1 < 4) v |= (unsigned __int128)c[w + 1] << 64;
        r.l[i] = (uint32_t)(v >> s2) & (i < 8 ? 0x1FFFFFFFu : 0xFFFFFFFFu);
    }
    (void)p32;
    return r;
}

// Known-answer check of the 29-bit-limb butterflies THROUGH k_ntt29_pass (its product, vsp_mm29, is a hand-laid-out routine with a private
// calling convention: see the note at capi.hip fp28_known_answer_check).  Once per context, before the first transform on that path: a
// 2^13-point vector (two passes: the lazy planes between passes, the first and the last pass) goes through a forward coset transform
// and an inverse coset transform on k_ntt29_pass and on the 8 x 32-bit k_ntt_pass; the outputs must agree word for word.  On a
// mismatch the context falls back to the 8 x 32-bit kernel ("ntt_fr29" = 0) for its lifetime.
static bool ntt29_known_answer_check(vsp_ctx *ctx) {
    if (ctx->ntt29_checked != 0) return ctx->ntt29_checked == 1;
    ctx->ntt29_checked = 2;                                   // in progress: the transforms below must not re-enter
    const unsigned lg = 13; const size_t n = (size_t)1 << lg, bytes = n * sizeof(Fr);
    std::vector<uint64_t> h(n * 4), o29(n * 4), o32(n * 4);
    uint64_t x = 0x243F6A8885A308D3ULL;
    auto next = [&]() { x += 0x9E3779B97F4A7C15ULL; uint64_t z = x; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL; z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL; return z ^ (z >> 31); };
    for (size_t i = 0; i < n; i++) { h[4 * i] = next(); h[4 * i + 1] = next(); h[4 * i + 2] = next(); h[4 * i + 3] = next() >> 2; }
    h[0] = h[1] = h[2] = h[3] = 0;                            // a zero and r - 1 among them
    h[4] = 0xffffffff00000000ULL; h[5] = 0x53bda402fffe5bfeULL; h[6] = 0x3339d80809a1d805ULL; h[7] = 0x73eda753299d7d48ULL;
    const uint64_t g7[4] = {7, 0, 0, 0};
    void *d = nullptr;
    bool ran = false, same = true;
    const bool had = ctx->opts.count("ntt_fr29") != 0; const long saved = had ? ctx->opts["ntt_fr29"] : 1;
    if (hipMalloc(&d, bytes) == hipSuccess) {
        ran = true;
        for (int inverse = 0; inverse < 2 && ran; inverse++) {
            for (int path = 0; path < 2 && ran; path++) {
                ctx->opts["ntt_fr29"] = path == 0 ? 1 : 0;
                std::vector<uint64_t> &out = path == 0 ? o29 : o32;
                ran = hipMemcpyAsync(d, h.data(), bytes, hipMemcpyHostToDevice, ctx->stream) == hipSuccess &&
                      ntt_device(ctx, (Fr *)d, lg, inverse, g7, nullptr) == VSP_OK &&
                      hipMemcpyAsync(out.data(), d, bytes, hipMemcpyDeviceToHost, ctx->stream) == hipSuccess &&
                      hipStreamSynchronize(ctx->stream) == hipSuccess;
            }
            if (ran && memcmp(o29.data(), o32.data(), bytes) != 0) same = false;
        }
        hipFree(d);
    }
    if (had) ctx->opts["ntt_fr29"] = saved; else ctx->opts.erase("ntt_fr29");
    hipGetLastError();
    if (!ran) { ctx->ntt29_checked = 0; ctx->stats["ntt_fr29_selfcheck"] = 0.0; return false; }      // could not run: no verdict, the 8 x 32-bit kernel this time
    { auto it = ctx->opts.find("ntt_fr29_selfcheck_fault"); if (it != ctx->opts.end() && it->second) same = false; }      // test hook: exercise the fallback
    ctx->ntt29_checked = same ? 1 : -1;
    ctx->stats["ntt_fr29_selfcheck"] = same ? 1.0 : -1.0;
    if (!same) { ctx->opts["ntt_fr29"] = 0; ctx->err = "ntt: the 29-bit-limb butterfly kernel failed its known-answer check; 8 x 32-bit kernel in use"; }
    return same;
}

// callers that set tables up before their first transform (the step-domain glue) run the check first: it rebuilds tables for its own use
void ntt_selfcheck_once(vsp_ctx *ctx) {
    long want29 = 1; auto it = ctx->opts.find("ntt_fr29"); if (it != ctx->opts.end()) want29 = it->second;
    if (want29 && ctx->ntt29_checked == 0) ntt29_known_answer_check(ctx);
}

// whether the 29-bit butterflies are in use on this context (runs their known-answer check when it has not run yet)
bool ntt29_in_use(vsp_ctx *ctx) {
    long want29 = 1; { auto it = ctx->opts.find("ntt_fr29"); if (it != ctx->opts.end()) want29 = it->second; }
    if (!want29) return false;
    if (ctx->ntt29_checked <= 0 && !ntt29_known_answer_check(ctx)) return false;
    { auto it = ctx->opts.find("ntt_fr29"); if (it != ctx->opts.end()) want29 = it->second; }
    return want29 != 0;
}
static int ntt_device_impl(vsp_ctx *ctx, const Fr *const *d_in, Fr *const *d_out, unsigned count, unsigned log_m, int inverse, const uint64_t *coset_g,
                           const HFr *extra_scale, const Fr *fuse_b, const Fr *fuse_c, size_t in_stride = 0, size_t out_stride = 0);
// d_a: n canonical Fr values in device memory, transformed in place.
// extra_scale (optional, host Montgomery): an additional constant multiplied into every output.
int ntt_device(vsp_ctx *ctx, Fr *d_a, unsigned log_m, int inverse, const uint64_t *coset_g, const HFr *extra_scale) {
    const Fr *in[1] = {d_a}; Fr *out[1] = {d_a};
    return ntt_device_impl(ctx, in, out, 1, log_m, inverse, coset_g, extra_scale, nullptr, nullptr);
}
// `count` (<= 3) transforms of one size in ONE launch per pass, in place (29-bit butterflies only: callers ask ntt29_in_use first)
int ntt_device_batch(vsp_ctx *ctx, Fr *const *d_a, unsigned count, unsigned log_m, int inverse, const uint64_t *coset_g, const HFr *extra_scale) {
    if (count < 1 || count > 3) return set_error(ctx, VSP_ERR_ARG, "ntt: batch of 1..3 transforms");
    const Fr *in[3] = {d_a[0], count > 1 ? d_a[1] : nullptr, count > 2 ? d_a[2] : nullptr};
    return ntt_device_impl(ctx, in, d_a, count, log_m, inverse, coset_g, extra_scale, nullptr, nullptr);
}
// d_h = transform of (a[i] b[i] - c[i]) 2^-261, the pointwise step fused into the first pass's load (29-bit butterflies only)
int ntt_device_fused_abc(vsp_ctx *ctx, const Fr *d_a, const Fr *d_b, const Fr *d_c, Fr *d_h, unsigned log_m, int inverse, const uint64_t *coset_g, const HFr *extra_scale) {
    const Fr *in[1] = {d_a}; Fr *out[1] = {d_h};
    return ntt_device_impl(ctx, in, out, 1, log_m, inverse, coset_g, extra_scale, d_b, d_c);
}
// `count` transforms at base + b * stride, in place, one launch per pass (a batch of proofs: 3 K transforms; 29-bit butterflies only)
int ntt_device_strided(vsp_ctx *ctx, Fr *base, unsigned count, size_t stride, unsigned log_m, int inverse, const uint64_t *coset_g, const HFr *extra_scale) {
    if (count < 1 || count > 65535) return set_error(ctx, VSP_ERR_ARG, "ntt: batch size");
    const Fr *in[1] = {base}; Fr *out[1] = {base};
    return ntt_device_impl(ctx, in, out, count, log_m, inverse, coset_g, extra_scale, nullptr, nullptr, stride, stride);
}
// d_h + b out_stride = transform of (a b - c) 2^-261 with a = d_a + b in_stride, b = a + off_b, c = a + off_c
int ntt_device_fused_abc_strided(vsp_ctx *ctx, const Fr *d_a, size_t off_b, size_t off_c, size_t in_stride, Fr *d_h, size_t out_stride, unsigned count, unsigned log_m,
                                 int inverse, const uint64_t *coset_g, const HFr *extra_scale) {
    if (count < 1 || count > 65535) return set_error(ctx, VSP_ERR_ARG, "ntt: batch size");
    const Fr *in[1] = {d_a}; Fr *out[1] = {d_h};
    return ntt_device_impl(ctx, in, out, count, log_m, inverse, coset_g, extra_scale, d_a + off_b, d_a + off_c, in_stride, out_stride);
}
static int ntt_device_impl(vsp_ctx *ctx, const Fr *const *d_in, Fr *const *d_out, unsigned count, unsigned log_m, int inverse, const uint64_t *coset_g,
                           const HFr *extra_scale, const Fr *fuse_b, const Fr *fuse_c, size_t in_stride, size_t out_stride) {
    Fr *d_a = d_out[0];
    if (log_m > 28) return set_error(ctx, VSP_ERR_UNSUPPORTED, "ntt: log_m > 28");
    if (coset_g) {
        uint64_t z = coset_g[0] | coset_g[1] | coset_g[2] | coset_g[3];
        if (!z) return set_error(ctx, VSP_ERR_ARG, "ntt: coset generator is zero");
    }
    // the 29-bit path's known-answer check comes first: it runs transforms of its own, which may rebuild the twiddle and coset tables
    bool check29_ok = true;
    { long want29 = 1; auto it = ctx->opts.find("ntt_fr29"); if (it != ctx->opts.end()) want29 = it->second;
      if (want29 && ctx->ntt29_checked <= 0) check29_ok = ntt29_known_answer_check(ctx); }      // (2 = the check itself is running: not re-entered)
    VSP_TRY(ntt_ensure_twiddles(ctx, log_m));
    if (coset_g) VSP_TRY(ntt_ensure_coset_tables(ctx, log_m, coset_g));
    const size_t n = (size_t)1 << log_m;

    // pass plan
    unsigned npass = log_m <= NTT_TILE_LOG ? 1 : (log_m + NTT_MAX_STAGES - 1) / NTT_MAX_STAGES;
    unsigned stages[8];
    for (unsigned i = 0; i < npass; i++) stages[i] = log_m / npass + (i < log_m % npass ? 1 : 0);
    // radix-4 steps take stages two at a time: trade a stage between two odd passes (7 + 7 -> 8 + 6), earlier pass the larger
    for (unsigned i = 0; i + 1 < npass; i++)
        if (stages[i] & 1)
            for (unsigned k = i + 1; k < npass; k++)
                if ((stages[k] & 1) && stages[i] < NTT_MAX_STAGES && stages[k] > 1) { stages[i]++; stages[k]--; break; }

    long use29 = 1; { auto it = ctx->opts.find("ntt_fr29"); if (it != ctx->opts.end()) use29 = it->second; }
    if (use29 && !check29_ok) use29 = 0;
    if (!use29 && (count > 1 || fuse_b || d_in[0] != d_out[0])) return set_error(ctx, VSP_ERR_UNSUPPORTED, "ntt: batched / fused transforms need the 29-bit butterflies");
    Fr *scratch = nullptr;
    if (npass > 1) { VSP_TRY(ensure(ctx, ctx->ntt_scratch, (size_t)count * n * (use29 ? 36 : sizeof(Fr)))); scratch = (Fr *)ctx->ntt_scratch.p; }

    HFr scale = HFr::one();
    bool have_scale = false;
    if (inverse) { scale = inv(host_from_u64((uint64_t)n)); have_scale = true; }
    if (extra_scale) { scale = mul(scale, *extra_scale); have_scale = true; }

    if (use29) {
        // butterflies on 9 x 29-bit limbs (fr29.h); the scratch buffer holds lazy values in three planes between passes
        VSP_TRY(ntt29_ensure_tables(ctx, coset_g != nullptr));
        const size_t tcount = (size_t)1 << (ctx->ntt.log - 1);
        const size_t L = (size_t)1 << PW_LOG, pn = (size_t)1 << ctx->ntt.pw_log, H = pn > L ? (pn >> PW_LOG) : 1;
        Planes29 lazy_in = planes_of(ctx->ntt_scratch, (size_t)count * n);
        PlanesOut29 lazy_out; lazy_out.p0 = (uint4 *)lazy_in.p0; lazy_out.p1 = (uint4 *)lazy_in.p1; lazy_out.p2 = (uint32_t *)lazy_in.p2;
        unsigned s0 = 0;
        for (unsigned i = 0; i < npass; i++) {
            NttPass29Args p;
            memset(&p, 0, sizeof p);
            p.log_n = log_m; p.s0 = s0; p.s1 = s0 + stages[i];
            p.tlog = ctx->ntt.log;
            p.first = (i == 0); p.last = (i == npass - 1);
            p.clog = npass == 1 ? 0 : NTT_TILE_LOG - stages[i];
            p.tw = planes_of(inverse ? ctx->ntt.inv29 : ctx->ntt.fwd29, tcount);
            p.premul = (p.first && coset_g && !inverse) ? 1 : 0;
            p.postmul = (p.last && inverse && coset_g) ? 2 : 1;                    // the last pass always multiplies: by the scale or by one
            if (p.premul) { p.pw_lo = planes_of(ctx->ntt.pw29[0], L); p.pw_hi = planes_of(ctx->ntt.pw29[1], H); }
            if (p.postmul == 2) { p.pw_lo = planes_of(ctx->ntt.pw29[2], L); p.pw_hi = planes_of(ctx->ntt.pw29[3], H); }
            p.scale = host_to_fr29_mont(scale);
            if (in_stride || out_stride) { p.in[0] = d_in[0]; p.out[0] = d_out[0]; p.in_stride = in_stride; p.out_stride = out_stride; }
            else for (unsigned b = 0; b < count; b++) { p.in[b] = d_in[b]; p.out[b] = d_out[b]; }
            if (p.first) { p.fuse_b = fuse_b; p.fuse_c = fuse_c; }
            if (!p.first && p.s0 < p.clog) return set_error(ctx, VSP_ERR_UNSUPPORTED, "ntt: pass plan");
            const unsigned tile_log = stages[i] + p.clog;
            hipLaunchKernelGGL(k_ntt29_pass, dim3((unsigned)(n >> tile_log), count), dim3(NTT_THREADS), 0, ctx->stream, lazy_in, lazy_out, p);
            VSP_LAUNCH_CHECK();
            s0 = p.s1;
        }
        ctx->stats["ntt_passes"] = (double)npass;
        ctx->stats["ntt_fr29"] = 1;
        return VSP_OK;
    }
    ctx->stats["ntt_fr29"] = 0;

    unsigned s0 = 0;
    for (unsigned i = 0; i < npass; i++) {
        NttPassArgs p;
        memset(&p, 0, sizeof p);
        p.log_n = log_m; p.s0 = s0; p.s1 = s0 + stages[i];
        p.tlog = ctx->ntt.log;
        p.first = (i == 0); p.last = (i == npass - 1);
        p.clog = npass == 1 ? 0 : NTT_TILE_LOG - stages[i];
        p.tw = (const Fr *)(inverse ? ctx->ntt.inv.p : ctx->ntt.fwd.p);
        p.premul = (p.first && coset_g && !inverse) ? 1 : 0;
        p.postmul = 0;
        if (p.last) {
            if (inverse && coset_g) p.postmul = 2;
            else if (have_scale) p.postmul = 1;
        }
        if (p.premul) { p.pw_lo = (const Fr *)ctx->ntt.pw_lo_f.p; p.pw_hi = (const Fr *)ctx->ntt.pw_hi_f.p; }
        if (p.postmul == 2) { p.pw_lo = (const Fr *)ctx->ntt.pw_lo_i.p; p.pw_hi = (const Fr *)ctx->ntt.pw_hi_i.p; }
        p.scale = to_dev(scale);
        if (!p.first && p.s0 < p.clog) return set_error(ctx, VSP_ERR_UNSUPPORTED, "ntt: pass plan");
        const Fr *src = (i == 0) ? d_a : scratch;
        Fr *dst = (npass == 1 || i == npass - 1) ? d_a : scratch;
        unsigned tile_log = stages[i] + p.clog;
        unsigned blocks = (unsigned)(n >> tile_log);
        hipLaunchKernelGGL(k_ntt_pass, dim3(blocks), dim3(NTT_THREADS), 0, ctx->stream, src, dst, p);
        VSP_LAUNCH_CHECK();
        s0 = p.s1;
    }
    if (log_m == 0 && have_scale) {
        // m = 1: the transform is the identity; only an explicit extra scale would matter (not used)
    }
    ctx->stats["ntt_passes"] = (double)npass;
    return VSP_OK;
}

}  // namespace vsp
