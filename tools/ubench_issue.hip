// Issue cost, in SHADER CYCLES, of the integer VALU instructions the 28-bit Montgomery routines are made of (gfx950) -- the unit costs
// behind bench.py's roofline_valu.  Unlike tools/ubench_valu.hip (wall time x the nominal clock, which over-states the cycles whenever the
// chip holds its clock below nominal under load), every wave stamps s_memtime around its own loop, so the result is in cycles whatever
// the clock does; the clock actually held is reported beside it (delta s_memtime / delta s_memrealtime x 100 MHz).
// W waves per SIMD run the same stream concurrently: per-SIMD issue cost = (a wave's cycles per instruction) / W.
// 16 independent accumulator chains per wave; the multiply-add writes its carry-out to an SGPR pair (as the product routines' do not
// need it, this is the cheapest legal form) or to VCC (the form ubench_valu.hip measured).
// Diagnostic tool, not part of the library:   hipcc --offload-arch=gfx950 -O3 tools/ubench_issue.hip -o tools/ubench_issue
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <algorithm>
#include <map>
#include <vector>

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

enum { OP_MAD_SGPR = 0, OP_MAD_VCC, OP_MAD_DEP, OP_MUL_LO, OP_LSHR64, OP_AND, OP_ADD, OP_MOV, OP_ADDC_PAIR, OP_LSHL_ADD64, OP_ADD3, OP_MM28_MIX, OP_COUNT };
static const char *NAMES[] = {"v_mad_u64_u32 (carry-out to an SGPR pair)", "v_mad_u64_u32 (carry-out to VCC)", "v_mad_u64_u32 (VCC, ONE dependent chain, as the routines)", "v_mul_lo_u32", "v_lshrrev_b64", "v_and_b32", "v_add_u32",
                              "v_mov_b32", "v_add_co_u32 + v_addc_co_u32", "v_lshl_add_u64", "v_add3_u32", "mm28 mix: 14 mad + 1 v_lshrrev_b64 + 1 v_and_b32"};
static const int PER_ITER[] = {16, 16, 16, 16, 16, 16, 16, 16, 16, 16, 16, 16};

struct Stamp { unsigned long long cyc, rt, t0, t1; unsigned hw, xcc; };

template <int OP> __global__ __launch_bounds__(256) void kern(Stamp *out, uint32_t *sink, int iters) {
    uint32_t x = threadIdx.x * 2654435761u + 12345u, y = blockIdx.x * 40503u + 977u;
    uint64_t a[16]; uint32_t b[16];
#pragma unroll
    for (int k = 0; k < 16; k++) { a[k] = (uint64_t)x * (k + 3) + y; b[k] = x * (2 * k + 1) ^ y; }
    unsigned long long t0, r0, t1, r1;
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(r0) :: "memory");
    // every variant is ONE asm statement of 16 instructions: between separate asm statements the compiler's hazard recogniser puts an
    // s_nop after each instruction that writes VCC / an SGPR (the first version of this tool, and tools/ubench_valu.hip, timed those too)
#define OPS8 : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7])
#define OPB8 : "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]), "+v"(b[4]), "+v"(b[5]), "+v"(b[6]), "+v"(b[7])
#define T8(I) I(0) I(1) I(2) I(3) I(4) I(5) I(6) I(7) I(0) I(1) I(2) I(3) I(4) I(5) I(6) I(7)
    for (int i = 0; i < iters; i++) {
        if (OP == OP_MAD_SGPR) {
#define I(k) "v_mad_u64_u32 %" #k ", s[20:21], %8, %9, %" #k "\n\t"
            asm volatile(T8(I) OPS8 : "v"(x), "v"(y) : "s20", "s21");
#undef I
        } else if (OP == OP_MAD_VCC) {
#define I(k) "v_mad_u64_u32 %" #k ", vcc, %8, %9, %" #k "\n\t"
            asm volatile(T8(I) OPS8 : "v"(x), "v"(y) : "vcc");
#undef I
        } else if (OP == OP_MAD_DEP) {
#define I(k) "v_mad_u64_u32 %0, vcc, %8, %9, %0\n\t"
            asm volatile(T8(I) OPS8 : "v"(x), "v"(y) : "vcc");
#undef I
        } else if (OP == OP_MUL_LO) {
#define I(k) "v_mul_lo_u32 %" #k ", %" #k ", %8\n\t"
            asm volatile(T8(I) OPB8 : "v"(x));
#undef I
        } else if (OP == OP_LSHR64) {
#define I(k) "v_lshrrev_b64 %" #k ", 28, %" #k "\n\t"
            asm volatile(T8(I) OPS8);
#undef I
        } else if (OP == OP_AND) {
#define I(k) "v_and_b32 %" #k ", 0xfffffff, %" #k "\n\t"
            asm volatile(T8(I) OPB8);
#undef I
        } else if (OP == OP_ADD) {
#define I(k) "v_add_u32 %" #k ", %" #k ", %8\n\t"
            asm volatile(T8(I) OPB8 : "v"(x));
#undef I
        } else if (OP == OP_MOV) {
#define I(k) "v_mov_b32 %" #k ", %8\n\t"
            asm volatile(T8(I) OPB8 : "v"(x));
#undef I
        } else if (OP == OP_ADDC_PAIR) {
#define I(k) "v_add_co_u32 %" #k ", vcc, %" #k ", %8\n\t"
#define J(k) "v_addc_co_u32 %" #k ", vcc, %" #k ", %8, vcc\n\t"
            asm volatile(I(0) J(1) I(2) J(3) I(4) J(5) I(6) J(7) I(0) J(1) I(2) J(3) I(4) J(5) I(6) J(7) OPB8 : "v"(y) : "vcc");
#undef I
#undef J
        } else if (OP == OP_LSHL_ADD64) {
#define I(k) "v_lshl_add_u64 %" #k ", %" #k ", 0, %7\n\t"
            asm volatile(T8(I) OPS8);
#undef I
        } else if (OP == OP_ADD3) {
#define I(k) "v_add3_u32 %" #k ", %" #k ", %8, %9\n\t"
            asm volatile(T8(I) OPB8 : "v"(x), "v"(y));
#undef I
        } else if (OP == OP_MM28_MIX) {
            // one column of the 28-bit product as the routine has it: 14 multiply-adds into ONE accumulator (half of them with an SGPR
            // operand), one 64-bit shift, one mask
            asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_mad_u64_u32 %0, vcc, %3, %4, %0\n\tv_mad_u64_u32 %0, vcc, %4, %5, %0\n\tv_mad_u64_u32 %0, vcc, %5, %2, %0\n\t"
                         "v_mad_u64_u32 %0, vcc, %2, %4, %0\n\tv_mad_u64_u32 %0, vcc, %3, %5, %0\n\tv_mad_u64_u32 %0, vcc, %2, %2, %0\n\t"
                         "v_mad_u64_u32 %0, vcc, %2, s20, %0\n\tv_mad_u64_u32 %0, vcc, %3, s21, %0\n\tv_mad_u64_u32 %0, vcc, %4, s20, %0\n\tv_mad_u64_u32 %0, vcc, %5, s21, %0\n\t"
                         "v_mad_u64_u32 %0, vcc, %2, s21, %0\n\tv_mad_u64_u32 %0, vcc, %3, s20, %0\n\tv_mad_u64_u32 %0, vcc, %4, s21, %0\n\t"
                         "v_lshrrev_b64 %0, 28, %0\n\tv_and_b32 %1, 0xfffffff, %1"
                         : "+v"(a[0]), "+v"(b[8]) : "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]) : "vcc", "s20", "s21");
        }
    }
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1) :: "memory");
    uint64_t s = 0; uint32_t t = x;
#pragma unroll
    for (int k = 0; k < 16; k++) { s ^= a[k]; t ^= b[k]; }
    sink[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)s ^ (uint32_t)(s >> 32) ^ t;
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)\n\ts_getreg_b32 %1, hwreg(HW_REG_XCC_ID)" : "=s"(hw), "=s"(xcc));
    if ((threadIdx.x & 63) == 0) { Stamp st; st.cyc = t1 - t0; st.rt = r1 - r0; st.t0 = t0; st.t1 = t1; st.hw = hw; st.xcc = xcc; out[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = st; }
}

template <int OP> int run(Stamp *d_st, uint32_t *d_sink, int cus) {
    const int iters = 20000;
    for (int wps = 1; wps <= 4; wps++) {
        const int blocks = cus * wps, waves = blocks * 4;              // 256-thread blocks: one wave per SIMD each, wps blocks per CU
        hipLaunchKernelGGL(kern<OP>, dim3(blocks), dim3(256), 0, 0, d_st, d_sink, 200);
        CHK(hipDeviceSynchronize());
        hipLaunchKernelGGL(kern<OP>, dim3(blocks), dim3(256), 0, 0, d_st, d_sink, iters);
        CHK(hipDeviceSynchronize());
        std::vector<Stamp> h(waves);
        CHK(hipMemcpy(h.data(), d_st, waves * sizeof(Stamp), hipMemcpyDeviceToHost));
        // group the waves by the SIMD they ran on (XCC, SE, SH, CU, SIMD of HW_ID).  The SIMD's issue cost is what it took the SIMD to get
        // through ALL its waves' instructions: (last wave's end - first wave's start) / instructions of all its waves.  (A wave's own
        // average is NOT that: the waves of a SIMD do not progress at equal speed -- the arbiter favours one, which then leaves early and
        // the others speed up -- so per-wave medians divided by the wave count under-state the cost by up to 2x; the first version of
        // this tool and the per-wave reading of ubench_madd28 made that mistake.)
        struct Acc { unsigned long long t0 = ~0ull, t1 = 0; size_t n = 0; double lone = 0; };
        std::map<unsigned long long, Acc> simd;
        std::vector<double> ghz(waves);
        for (int i = 0; i < waves; i++) {
            Acc &a = simd[((unsigned long long)(h[i].xcc & 0xF) << 32) | ((h[i].hw >> 4) & 0x3u) | (((h[i].hw >> 8) & 0xFFu) << 2)];
            a.t0 = std::min(a.t0, h[i].t0); a.t1 = std::max(a.t1, h[i].t1); a.n++;
            a.lone = std::max(a.lone, (double)h[i].cyc / ((double)iters * PER_ITER[OP]));
            ghz[i] = h[i].rt ? (double)h[i].cyc / (double)h[i].rt * 0.1 : 0.0;
        }
        std::sort(ghz.begin(), ghz.end());
        std::map<size_t, std::vector<double>> by_n;               // waves sharing the SIMD -> cycles of that SIMD per wave-instruction
        for (auto &kv : simd) by_n[kv.second.n].push_back((double)(kv.second.t1 - kv.second.t0) / ((double)kv.second.n * iters * PER_ITER[OP]));
        printf("%-52s %d wave/SIMD (clock held %.2f GHz):", NAMES[OP], wps, ghz[waves / 2]);
        for (auto &kv : by_n) { std::vector<double> &v = kv.second; std::sort(v.begin(), v.end()); printf("  [%zu SIMDs with %zu waves: %.2f cycles of the SIMD per wave-instruction]", v.size(), kv.first, v[v.size() / 2]); }
        printf("\n");
    }
    return 0;
}

int main() {
    hipDeviceProp_t p; CHK(hipGetDeviceProperties(&p, 0));
    int cus = p.multiProcessorCount;
    printf("device %s, %d CUs, nominal clock %.2f GHz; costs below are in shader cycles (s_memtime), independent of the clock held\n", p.name, cus, p.clockRate * 1e-6);
    Stamp *d_st; uint32_t *d_sink;
    CHK(hipMalloc(&d_st, (size_t)cus * 4 * 4 * sizeof(Stamp))); CHK(hipMalloc(&d_sink, (size_t)cus * 4 * 256 * 4));
    run<OP_MAD_SGPR>(d_st, d_sink, cus); run<OP_MAD_VCC>(d_st, d_sink, cus); run<OP_MAD_DEP>(d_st, d_sink, cus); run<OP_MUL_LO>(d_st, d_sink, cus); run<OP_LSHR64>(d_st, d_sink, cus);
    run<OP_AND>(d_st, d_sink, cus); run<OP_ADD>(d_st, d_sink, cus); run<OP_MOV>(d_st, d_sink, cus); run<OP_ADDC_PAIR>(d_st, d_sink, cus);
    run<OP_LSHL_ADD64>(d_st, d_sink, cus); run<OP_ADD3>(d_st, d_sink, cus); run<OP_MM28_MIX>(d_st, d_sink, cus);
    return 0;
}
