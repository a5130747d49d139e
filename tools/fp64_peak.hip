// Micro-benchmark: sustained FP64 MFMA (v_mfma_f64_16x16x4_f64) and FP64 VALU FMA rate on gfx950.
// Establishes the measured ceiling the Cholesky trailing update is priced against (the CDNA4 guide in this
// image lists no FP64 matrix number; the public spec figure is 78.6 TFLOP/s for both vector and matrix FP64).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double double4_t __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(256) void k_mfma(double *out, int iters, double a0, double b0) {
    double4_t acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; i++) acc[i] = double4_t{0, 0, 0, 0};
    double a = a0 + threadIdx.x * 1e-9, b = b0;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < NACC; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 12345.678) out[0] = s;
}

__global__ __launch_bounds__(256) void k_fma(double *out, int iters, double a0, double b0) {
    double acc[16];
#pragma unroll
    for (int i = 0; i < 16; i++) acc[i] = i;
    double a = a0 + threadIdx.x * 1e-9, b = b0;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 16; i++) acc[i] = __builtin_fma(a, acc[i], b);
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) s += acc[i];
    if (s == 12345.678) out[0] = s;
}

template <typename F>
static double time_ms(F f) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    f();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    f();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main() {
    double *out;
    hipMalloc(&out, 8);
    const int iters = 4000;
    for (int wg_per_cu = 1; wg_per_cu <= 2; wg_per_cu++) {
        const int grid = 256 * wg_per_cu;
        double ms = time_ms([&] { hipLaunchKernelGGL(k_mfma<16>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0, 1e-3); });
        double flops = (double)grid * 4 * iters * 16 * 2048.0;
        printf("mfma_f64_16x16x4 16 acc, %d waves/SIMD: %.2f TFLOP/s (%.3f ms)\n", wg_per_cu, flops / ms / 1e9, ms);
        ms = time_ms([&] { hipLaunchKernelGGL(k_mfma<4>, dim3(grid), dim3(256), 0, 0, out, iters * 4, 1.0, 1e-3); });
        printf("mfma_f64_16x16x4  4 acc, %d waves/SIMD: %.2f TFLOP/s (%.3f ms)\n", wg_per_cu, flops / ms / 1e9, ms);
    }
    for (int wg_per_cu = 1; wg_per_cu <= 4; wg_per_cu *= 2) {
        const int grid = 256 * wg_per_cu;
        double ms = time_ms([&] { hipLaunchKernelGGL(k_fma, dim3(grid), dim3(256), 0, 0, out, iters * 4, 1.0000001, 1e-3); });
        double flops = (double)grid * 256 * (iters * 4.0) * 16 * 2.0;
        printf("v_fma_f64 16 chains, %d waves/SIMD: %.2f TFLOP/s (%.3f ms)\n", wg_per_cu, flops / ms / 1e9, ms);
    }
    return 0;
}
