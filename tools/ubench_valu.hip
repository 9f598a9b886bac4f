// Instruction-rate microbenchmark for the integer VALU ops a big-integer Montgomery product can be built
// from on gfx950.  Prints ops per clock per CU (wave64 lanes * instructions / cycles) at 1, 2 and 4 waves/SIMD.
// Diagnostic tool only (not part of the library).   hipcc --offload-arch=gfx950 -O3 ubench_valu.hip -o ubench_valu
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

enum { OP_MAD64 = 0, OP_MUL_LO, OP_MUL_HI, OP_MAD_U24, OP_MUL_HI_U24, OP_ADD_CO_CHAIN, OP_LSHL_ADD64, OP_FMA64, OP_MOV, OP_ADD3, OP_MAD_U32_U16, OP_COUNT };
static const char *NAMES[] = {"v_mad_u64_u32", "v_mul_lo_u32", "v_mul_hi_u32", "v_mad_u32_u24", "v_mul_hi_u32_u24", "v_add_co+v_addc_co pair", "v_lshl_add_u64", "v_fma_f64", "v_mov_b32", "v_add3_u32", "v_mad_u32_u16"};

template <int OP> __global__ void kern(uint32_t *out, int iters) {
    uint32_t x = threadIdx.x * 2654435761u + 12345u, y = blockIdx.x * 40503u + 977u;
    uint64_t a0 = x, a1 = y, a2 = x ^ y, a3 = x + y, a4 = x * 3, a5 = y * 5, a6 = x * 7, a7 = y * 11;
    uint32_t b0 = x, b1 = y, b2 = x ^ y, b3 = x + y, b4 = x * 3, b5 = y * 5, b6 = x * 7, b7 = y * 11;
    double d0 = x, d1 = y, d2 = 1.5, d3 = 2.5, d4 = 3.5, d5 = 4.5, d6 = 5.5, d7 = 6.5, dx = 1.0000001, dy = 0.5;
    for (int i = 0; i < iters; i++) {
#define R8(S) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7)
        if (OP == OP_MAD64) {
#define S(k) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(a##k) : "v"(x), "v"(y) : "vcc");
            R8(S)
#undef S
        } else if (OP == OP_MUL_LO) {
#define S(k) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(b##k) : "v"(x));
            R8(S)
#undef S
        } else if (OP == OP_MUL_HI) {
#define S(k) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(b##k) : "v"(x));
            R8(S)
#undef S
        } else if (OP == OP_MAD_U24) {
#define S(k) asm volatile("v_mad_u32_u24 %0, %1, %2, %0" : "+v"(b##k) : "v"(x), "v"(y));
            R8(S)
#undef S
        } else if (OP == OP_MUL_HI_U24) {
#define S(k) asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(b##k) : "v"(x));
            R8(S)
#undef S
        } else if (OP == OP_ADD_CO_CHAIN) {
#define S(k) asm volatile("v_add_co_u32 %0, vcc, %0, %1\n\tv_addc_co_u32 %2, vcc, %2, %3, vcc" : "+v"(b##k), "+v"(x) : "v"(y), "v"(y) : "vcc");
            S(0) S(1) S(2) S(3)
#undef S
        } else if (OP == OP_LSHL_ADD64) {
#define S(k) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(a##k) : "v"(a7));
            S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(0)
#undef S
        } else if (OP == OP_FMA64) {
#define S(k) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d##k) : "v"(dx), "v"(dy));
            R8(S)
#undef S
        } else if (OP == OP_MOV) {
#define S(k) asm volatile("v_mov_b32 %0, %1" : "+v"(b##k) : "v"(x));
            R8(S)
#undef S
        } else if (OP == OP_ADD3) {
#define S(k) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(b##k) : "v"(x), "v"(y));
            R8(S)
#undef S
        } else if (OP == OP_MAD_U32_U16) {
#define S(k) asm volatile("v_mad_u32_u16 %0, %1, %2, %0" : "+v"(b##k) : "v"(x), "v"(y));
            R8(S)
#undef S
        }
    }
    uint64_t s = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
    uint32_t t = b0 ^ b1 ^ b2 ^ b3 ^ b4 ^ b5 ^ b6 ^ b7 ^ x;
    double d = d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7;
    out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)s ^ (uint32_t)(s >> 32) ^ t ^ (uint32_t)d;
}

template <int OP> int run(uint32_t *d_out, int cus, double clk_ghz) {
    const int iters = 20000;
    for (int wps = 1; wps <= 4; wps *= 2) {
        int threads = 256 * wps;         // wps waves per SIMD with one block per CU
        if (threads > 1024) { threads = 1024; }
        int blocks = cus * (wps == 4 ? 1 : 1);
        hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
        hipLaunchKernelGGL(kern<OP>, dim3(blocks), dim3(threads), 0, 0, d_out, 100);
        CHK(hipDeviceSynchronize());
        CHK(hipEventRecord(e0));
        hipLaunchKernelGGL(kern<OP>, dim3(blocks), dim3(threads), 0, 0, d_out, iters);
        CHK(hipEventRecord(e1));
        CHK(hipEventSynchronize(e1));
        float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
        double wave_instr = (double)iters * 8 * (threads / 64);        // per CU
        double instr_per_s_per_cu = wave_instr / (ms * 1e-3);
        double cyc_per_wave_instr_per_simd = clk_ghz * 1e9 / (instr_per_s_per_cu / 4);
        printf("%-26s %d wave/SIMD: %8.3f ms  %7.2f G wave-instr/s/CU  -> %5.2f cycles per wave-instruction per SIMD (at %.2f GHz)\n",
               NAMES[OP], threads / 256, ms, instr_per_s_per_cu * 1e-9, cyc_per_wave_instr_per_simd, clk_ghz);
    }
    return 0;
}

int main() {
    hipDeviceProp_t p; CHK(hipGetDeviceProperties(&p, 0));
    int cus = p.multiProcessorCount; double clk = p.clockRate * 1e-6;
    printf("device %s, %d CUs, clock %.2f GHz\n", p.name, cus, clk);
    uint32_t *d_out; CHK(hipMalloc(&d_out, (size_t)cus * 1024 * 4));
    run<OP_MAD64>(d_out, cus, clk); run<OP_MUL_LO>(d_out, cus, clk); run<OP_MUL_HI>(d_out, cus, clk);
    run<OP_MAD_U24>(d_out, cus, clk); run<OP_MUL_HI_U24>(d_out, cus, clk); run<OP_ADD_CO_CHAIN>(d_out, cus, clk);
    run<OP_LSHL_ADD64>(d_out, cus, clk); run<OP_FMA64>(d_out, cus, clk); run<OP_MOV>(d_out, cus, clk); run<OP_ADD3>(d_out, cus, clk);
    run<OP_MAD_U32_U16>(d_out, cus, clk);
    return 0;
}
