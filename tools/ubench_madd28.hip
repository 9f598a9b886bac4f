// How the REAL 28-bit routines scale with waves per SIMD, in shader cycles (s_memtime) -- the question behind the accumulation kernel's
// register budget: it runs at 2 waves per SIMD (238 VGPRs); would 3 (<= 168 VGPRs) or 4 (<= 128) issue more products per cycle?
//   (a) a dependent chain of products r = mul28(r, b)           -- vsp_mm28 alone (45 VGPRs: any occupancy)
//   (b) a dependent chain of mixed additions acc = madd28(acc, p) on registers, no memory in the loop -- the accumulation kernel's body
// One workgroup of 256 x W threads per CU = W waves on every SIMD; every wave stamps its own loop.  Reported per SIMD:
// cycles per wave-operation = (a wave's cycles per operation) / W, and the clock the chip held.
// Diagnostic tool, not part of the library:   hipcc --offload-arch=gfx950 -O3 -I vote_saver_protocol_amd/csrc tools/ubench_madd28.hip -o tools/ubench_madd28
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <algorithm>
#include <vector>
#include "fp28.h"
#include "accum28_asm_gfx950.h"

using namespace vsp;
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

struct Stamp { unsigned long long cyc, rt; };
static const uint32_t *g_sorted_small = nullptr, *g_sorted_big = nullptr; static const void *g_big_table = nullptr;
__device__ __forceinline__ void stamp(unsigned long long &t, unsigned long long &r) {
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t), "=s"(r) :: "memory");
}

template <int W> __global__ __launch_bounds__(256 * W) void k_mul(const uint32_t *in, uint32_t *out, Stamp *st, int iters) {
#if defined(__HIP_DEVICE_COMPILE__)
    const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    Fp28 a, b;
    for (int i = 0; i < 14; i++) { a.l[i] = in[g * 28 + i]; b.l[i] = in[g * 28 + 14 + i]; }
    unsigned long long t0, r0, t1, r1;
    stamp(t0, r0);
    for (int it = 0; it < iters; it++) a = mul28(a, b);
    stamp(t1, r1);
    for (int i = 0; i < 14; i++) out[g * 16 + i] = a.l[i];
    if ((threadIdx.x & 63) == 0) { Stamp s; s.cyc = t1 - t0; s.rt = r1 - r0; st[g >> 6] = s; }
#endif
}
template <int W> __global__ __launch_bounds__(256 * W) void k_madd(const Affine28 *pts, uint32_t *out, Stamp *st, int iters) {
#if defined(__HIP_DEVICE_COMPILE__)
    const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const Affine28 p = pts[g];
    XYZZ28 acc = xyzz28_inf();
    madd28(acc, pts[g + 1], false);                           // start from another point: the chain below never meets an equal x for random data
    unsigned long long t0, r0, t1, r1;
    stamp(t0, r0);
    bool ok = true;
    for (int it = 0; it < iters; it++) ok = madd28(acc, p, false) && ok;      // acc + p, + p, ...: acc = q + k p
    stamp(t1, r1);
    for (int i = 0; i < 14; i++) out[g * 16 + i] = acc.X.l[i] ^ acc.Y.l[i] ^ acc.ZZ.l[i] ^ acc.ZZZ.l[i];
    out[g * 16 + 15] = ok;
    if ((threadIdx.x & 63) == 0) { Stamp s; s.cyc = t1 - t0; s.rt = r1 - r0; st[g >> 6] = s; }
#endif
}

// (c) the generated whole-loop routine (accum28_asm_gfx950.h) over a table small enough to stay in the vector L1: its arithmetic and
//     instruction fetch without the gather's memory side.  GATHER = 1: every lane walks its own random rows of a 256 MiB table instead.
template <int W, int GATHER> __global__ __launch_bounds__(256 * W) void k_asm(const Affine28 *table, const uint32_t *sorted, uint32_t *out, Stamp *st, int iters) {
#if defined(__HIP_DEVICE_COMPILE__)
    const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t acc[56], flag;
    unsigned long long t0, r0, t1, r1;
    const uint32_t start = GATHER ? (uint32_t)(g * (size_t)iters) : 0u;
    stamp(t0, r0);
    accum28_asm(acc, flag, table, sorted, start, start + (uint32_t)iters);
    stamp(t1, r1);
    for (int i = 0; i < 14; i++) out[g * 16 + i] = acc[i] ^ acc[14 + i] ^ acc[28 + i] ^ acc[42 + i];
    out[g * 16 + 15] = flag;
    if ((threadIdx.x & 63) == 0) { Stamp s; s.cyc = t1 - t0; s.rt = r1 - r0; st[g >> 6] = s; }
#endif
}

template <int W> int run(const uint32_t *d_in, uint32_t *d_out, Stamp *d_st, int cus, int which) {
    const int iters = which == 0 ? 4000 : (which == 3 ? 64 : 600), threads = 256 * W, waves = cus * 4 * W;
    for (int rep = 0; rep < 2; rep++) {                       // first launch warms up
        if (which == 0) hipLaunchKernelGGL(k_mul<W>, dim3(cus), dim3(threads), 0, 0, d_in, d_out, d_st, iters);
        else if (which == 1) hipLaunchKernelGGL(k_madd<W>, dim3(cus), dim3(threads), 0, 0, (const Affine28 *)d_in, d_out, d_st, iters);
        else if (which == 2) hipLaunchKernelGGL((k_asm<W, 0>), dim3(cus), dim3(threads), 0, 0, (const Affine28 *)d_in, g_sorted_small, d_out, d_st, iters);
        else hipLaunchKernelGGL((k_asm<W, 1>), dim3(cus), dim3(threads), 0, 0, (const Affine28 *)g_big_table, g_sorted_big, d_out, d_st, iters);
        CHK(hipDeviceSynchronize());
    }
    std::vector<Stamp> h(waves);
    CHK(hipMemcpy(h.data(), d_st, waves * sizeof(Stamp), hipMemcpyDeviceToHost));
    std::vector<double> c(waves), ghz(waves);
    for (int i = 0; i < waves; i++) { c[i] = (double)h[i].cyc / iters; ghz[i] = h[i].rt ? (double)h[i].cyc / (double)h[i].rt * 0.1 : 0.0; }
    std::sort(c.begin(), c.end()); std::sort(ghz.begin(), ghz.end());
    // the SIMD's cost per wave-op = what its SLOWEST wave took, divided by the waves it ran: all waves start together and do equal work,
    // but not at equal speed (the arbiter favours one wave, which leaves early, and the others then speed up), so a per-wave median
    // divided by W under-states the cost -- the first reading of this tool did, and promised 1.4x from a third wave that is not there
    const double per_simd = c[waves - 1 - waves / 50] / W, clk = ghz[waves / 2];
    printf("%-34s %d wave/SIMD: %8.0f cycles per op and wave (p5 %.0f, p98 %.0f) -> %7.0f cycles of a SIMD per wave-op, clock held %.2f GHz -> %6.1f ns per wave-op per SIMD\n",
           which == 0 ? "mul28 chain (vsp_mm28)" : which == 1 ? "madd28 chain (mixed addition)" : which == 2 ? "accum28 asm loop, L1-resident rows" : "accum28 asm loop, random 256 MiB gather", W, c[waves / 2], c[waves / 20], c[waves - 1 - waves / 50], per_simd, clk, per_simd / clk);
    return 0;
}

int main() {
    hipDeviceProp_t pr; CHK(hipGetDeviceProperties(&pr, 0));
    const int cus = pr.multiProcessorCount;
    printf("device %s, %d CUs, nominal clock %.2f GHz\n", pr.name, cus, pr.clockRate * 1e-6);
    const size_t threads = (size_t)cus * 1024 + 64;
    uint32_t *d_in, *d_out; Stamp *d_st;
    CHK(hipMalloc(&d_in, threads * 32 * 4)); CHK(hipMalloc(&d_out, threads * 16 * 4)); CHK(hipMalloc(&d_st, (size_t)cus * 16 * sizeof(Stamp)));
    std::vector<uint32_t> h(threads * 32);
    uint64_t s = 88172645463325252ULL;
    for (auto &v : h) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; v = (uint32_t)s & 0x0FFFFFFFu; }
    for (size_t t = 0; t < threads; t++) { h[t * 32 + 13] &= 0xFFFF; h[t * 32 + 27] &= 0xFFFF; h[t * 32 + 28] = h[t * 32 + 29] = h[t * 32 + 30] = h[t * 32 + 31] = 0; }   // values below 2^380
    CHK(hipMemcpy(d_in, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    // (c): 600 entries walking over the first 64 rows (every lane the same: the loads coalesce and hit L1); signs mixed
    {
        std::vector<uint32_t> ent(1024);
        for (size_t i = 0; i < ent.size(); i++) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; ent[i] = (uint32_t)(s % 64) | ((uint32_t)(s >> 40) & 1u) << 31; }
        uint32_t *d; CHK(hipMalloc(&d, ent.size() * 4)); CHK(hipMemcpy(d, ent.data(), ent.size() * 4, hipMemcpyHostToDevice)); g_sorted_small = d;
        // (d): a 2^21-row table (256 MiB) of random limbs, and for every lane of the largest launch 64 random entries of its own
        const size_t rows = (size_t)1 << 21, lanes = (size_t)cus * 1024, per = 64;
        std::vector<uint32_t> tb(rows * 32), eb(lanes * per + 8);
        for (auto &v : tb) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; v = (uint32_t)s & 0x0FFFFFFFu; }
        for (size_t r = 0; r < rows; r++) { tb[r * 32 + 13] &= 0xFFFF; tb[r * 32 + 27] &= 0xFFFF; }
        for (auto &v : eb) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; v = (uint32_t)(s % rows) | ((uint32_t)(s >> 40) & 1u) << 31; }
        void *dt; uint32_t *de; CHK(hipMalloc(&dt, tb.size() * 4)); CHK(hipMemcpy(dt, tb.data(), tb.size() * 4, hipMemcpyHostToDevice));
        CHK(hipMalloc(&de, eb.size() * 4)); CHK(hipMemcpy(de, eb.data(), eb.size() * 4, hipMemcpyHostToDevice));
        g_big_table = dt; g_sorted_big = de;
    }
    for (int which = 0; which < 4; which++) {
        run<1>(d_in, d_out, d_st, cus, which); run<2>(d_in, d_out, d_st, cus, which); run<3>(d_in, d_out, d_st, cus, which); if (which < 2) run<4>(d_in, d_out, d_st, cus, which);
    }
    return 0;
}
