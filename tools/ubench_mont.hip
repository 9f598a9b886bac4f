// Microbenchmark: cycles per Fp Montgomery product of the library's 12 x 32-bit routine against the carry-free 14 x 28-bit
// prototype (tools/gen_mont28_proto.py), both as straight instruction streams in a dependent chain r = r * b.
// Also prints one lane's result of each so that the prototype can be checked against big-integer arithmetic offline.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_mont.hip -o tools/ubench_mont
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "mont28_proto.h"

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

// in: 12 (or 14) limbs of a then of b per thread; out: the limbs of the last result
__global__ __launch_bounds__(256) void k32(const uint32_t *in, uint32_t *out, int iters) {
    uint32_t a[12], b[12], r[12];
    const uint32_t *p = in + (size_t)(blockIdx.x * blockDim.x + threadIdx.x) * 24;
    for (int i = 0; i < 12; i++) { a[i] = p[i]; b[i] = p[12 + i]; }
    for (int it = 0; it < iters; it++) {
        uint32_t x[12], y[12];
        for (int i = 0; i < 12; i++) { x[i] = a[i]; y[i] = b[i]; }
        asm volatile(MONT32_BODY
            : "={v24}"(r[0]), "={v25}"(r[1]), "={v26}"(r[2]), "={v27}"(r[3]), "={v28}"(r[4]), "={v29}"(r[5]), "={v30}"(r[6]), "={v31}"(r[7]), "={v32}"(r[8]), "={v33}"(r[9]), "={v34}"(r[10]), "={v35}"(r[11]),
              "+{v0}"(x[0]), "+{v1}"(x[1]), "+{v2}"(x[2]), "+{v3}"(x[3]), "+{v4}"(x[4]), "+{v5}"(x[5]), "+{v6}"(x[6]), "+{v7}"(x[7]), "+{v8}"(x[8]), "+{v9}"(x[9]), "+{v10}"(x[10]), "+{v11}"(x[11]),
              "+{v12}"(y[0]), "+{v13}"(y[1]), "+{v14}"(y[2]), "+{v15}"(y[3]), "+{v16}"(y[4]), "+{v17}"(y[5]), "+{v18}"(y[6]), "+{v19}"(y[7]), "+{v20}"(y[8]), "+{v21}"(y[9]), "+{v22}"(y[10]), "+{v23}"(y[11])
            :
            : "vcc", "scc", "s0", "s1", "s2", "s3", "s4", "s5", "s6", "s7", "s8", "s9", "s10", "s11", "s12", "v36", "v37", "v38", "v39");
        for (int i = 0; i < 12; i++) a[i] = r[i];
    }
    uint32_t *q = out + (size_t)(blockIdx.x * blockDim.x + threadIdx.x) * 16;
    for (int i = 0; i < 12; i++) q[i] = a[i];
}
__global__ __launch_bounds__(256) void k28(const uint32_t *in, uint32_t *out, int iters) {
    uint32_t a[14], b[14], r[14];
    const uint32_t *p = in + (size_t)(blockIdx.x * blockDim.x + threadIdx.x) * 28;
    for (int i = 0; i < 14; i++) { a[i] = p[i]; b[i] = p[14 + i]; }
    for (int it = 0; it < iters; it++) {
        asm volatile(MONT28_BODY
            : "={v28}"(r[0]), "={v29}"(r[1]), "={v30}"(r[2]), "={v31}"(r[3]), "={v32}"(r[4]), "={v33}"(r[5]), "={v34}"(r[6]), "={v35}"(r[7]), "={v36}"(r[8]), "={v37}"(r[9]), "={v38}"(r[10]), "={v39}"(r[11]), "={v40}"(r[12]), "={v41}"(r[13])
            : "{v0}"(a[0]), "{v1}"(a[1]), "{v2}"(a[2]), "{v3}"(a[3]), "{v4}"(a[4]), "{v5}"(a[5]), "{v6}"(a[6]), "{v7}"(a[7]), "{v8}"(a[8]), "{v9}"(a[9]), "{v10}"(a[10]), "{v11}"(a[11]), "{v12}"(a[12]), "{v13}"(a[13]),
              "{v14}"(b[0]), "{v15}"(b[1]), "{v16}"(b[2]), "{v17}"(b[3]), "{v18}"(b[4]), "{v19}"(b[5]), "{v20}"(b[6]), "{v21}"(b[7]), "{v22}"(b[8]), "{v23}"(b[9]), "{v24}"(b[10]), "{v25}"(b[11]), "{v26}"(b[12]), "{v27}"(b[13])
            : "vcc", "scc", "s0", "s1", "s2", "s3", "s4", "s5", "s6", "s7", "s8", "s9", "s10", "s11", "s12", "s13", "s14", "s15", "v42", "v43", "v44");
        for (int i = 0; i < 14; i++) a[i] = r[i];
    }
    uint32_t *q = out + (size_t)(blockIdx.x * blockDim.x + threadIdx.x) * 16;
    for (int i = 0; i < 14; i++) q[i] = a[i];
}

int main() {
    hipDeviceProp_t pr; CHK(hipGetDeviceProperties(&pr, 0));
    const int cus = pr.multiProcessorCount; const double clk = pr.clockRate * 1e-6;
    printf("device %s, %d CUs, clock %.2f GHz\n", pr.name, cus, clk);
    const int maxthreads = cus * 1024;
    uint32_t *d_in, *d_out; CHK(hipMalloc(&d_in, (size_t)maxthreads * 28 * 4)); CHK(hipMalloc(&d_out, (size_t)maxthreads * 16 * 4));
    // thread 0's operands are read from stdin-free constants: a = 3 (limb 0), b = 5 (limb 0), everything else pseudo-random but < 2^28 per limb
    uint32_t *h = (uint32_t *)malloc((size_t)maxthreads * 28 * 4);
    uint64_t s = 88172645463325252ULL;
    for (size_t i = 0; i < (size_t)maxthreads * 28; i++) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; h[i] = (uint32_t)s & 0x0FFFFFFF; }
    for (int i = 0; i < 28; i++) h[i] = 0;                       // thread 0 (both layouts): a = 0x1234567, b = 0x7654321
    h[0] = 0x1234567;
    uint32_t h32[24] = {0}; h32[0] = 0x1234567; h32[12] = 0x7654321;
    h[14] = 0x7654321;
    const int iters = 2000;
    for (int variant = 0; variant < 2; variant++) {
        if (variant == 0) { CHK(hipMemcpy(d_in, h, (size_t)maxthreads * 28 * 4, hipMemcpyHostToDevice)); CHK(hipMemcpy(d_in, h32, sizeof h32, hipMemcpyHostToDevice)); }
        else CHK(hipMemcpy(d_in, h, (size_t)maxthreads * 28 * 4, hipMemcpyHostToDevice));
        for (int wps = 1; wps <= 4; wps *= 2) {
            const int threads = 256, blocks = cus * wps;            // wps workgroups of 4 waves per CU = wps waves per SIMD
            hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
            if (variant == 0) hipLaunchKernelGGL(k32, dim3(blocks), dim3(threads), 0, 0, d_in, d_out, 10); else hipLaunchKernelGGL(k28, dim3(blocks), dim3(threads), 0, 0, d_in, d_out, 10);
            CHK(hipDeviceSynchronize());
            CHK(hipEventRecord(e0));
            if (variant == 0) hipLaunchKernelGGL(k32, dim3(blocks), dim3(threads), 0, 0, d_in, d_out, iters); else hipLaunchKernelGGL(k28, dim3(blocks), dim3(threads), 0, 0, d_in, d_out, iters);
            CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
            float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
            double wave_products_per_simd = (double)iters * wps;     // each SIMD runs wps waves
            printf("%s  %d wave(s)/SIMD: %8.3f ms -> %7.0f cycles per wave-product per SIMD, %6.1f G products/s on the GPU\n",
                   variant == 0 ? "12 x 32-bit (library routine)" : "14 x 28-bit carry-free proto ", wps, ms, ms * 1e-3 * clk * 1e9 / wave_products_per_simd,
                   (double)iters * blocks * threads / (ms * 1e-3) * 1e-9);
        }
        // one product of thread 0 for the offline check
        if (variant == 0) hipLaunchKernelGGL(k32, dim3(1), dim3(64), 0, 0, d_in, d_out, 1); else hipLaunchKernelGGL(k28, dim3(1), dim3(64), 0, 0, d_in, d_out, 1);
        uint32_t r[16]; CHK(hipMemcpy(r, d_out, sizeof r, hipMemcpyDeviceToHost));
        printf("%s result limbs:", variant == 0 ? "r32" : "r28");
        for (int i = 0; i < (variant == 0 ? 12 : 14); i++) printf(" %08x", r[i]);
        printf("\n");
    }
    return 0;
}
