// Micro-benchmark: sustained FP64 MFMA (v_mfma_f64_16x16x4_f64) and FP64 VALU FMA rate on gfx950.
// Establishes the measured ceiling the Cholesky trailing update is priced against (the CDNA4 guide in this
// image lists no FP64 matrix number; the public spec figure is 78.6 TFLOP/s for both vector and matrix FP64).
// Every configuration is run repeatedly (>= 40 ms) after a warm-up: short runs right after idle measure the
// DVFS ramp, not the pipe.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double double4_t __attribute__((ext_vector_type(4)));

template <int NACC, int THREADS>
__global__ __launch_bounds__(THREADS) void k_mfma(double *out, int iters, double a0, double b0) {
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

// the same chains fed with random operands (hash of lane / register index in [-1,1]): the data-dependent switching
// power moves the sustained clock, so this -- not the constant-operand figure -- is what a GEMM on real data sees.
// out[1], out[2] = shader clocks / 100 MHz wall ticks of one wave -> sustained shader clock.
template <int NACC, int THREADS>
__global__ __launch_bounds__(THREADS) void k_mfma_rand(double *out, int iters, unsigned seed) {
    double4_t acc[NACC];
    double a[NACC], b[NACC];
#pragma unroll
    for (int i = 0; i < NACC; i++) {
        acc[i] = double4_t{0, 0, 0, 0};
        unsigned h = (threadIdx.x * 2654435761u) ^ (i * 40503u + seed) ^ (blockIdx.x * 97u);
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        a[i] = (double)(int)h * (1.0 / 2147483648.0);
        h *= 3266489917u; h ^= h >> 16;
        b[i] = (double)(int)h * (1.0 / 2147483648.0);
    }
    const long long c0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[i], acc[i], 0, 0, 0);
    }
    const long long c1 = clock64(), w1 = wall_clock64();
    double s = 0;
#pragma unroll
    for (int i = 0; i < NACC; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 12345.678) out[0] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) { out[1] = (double)(c1 - c0); out[2] = (double)(w1 - w0); }
}

template <int THREADS>
__global__ __launch_bounds__(THREADS) void k_fma(double *out, int iters, double a0, double b0) {
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

// half the waves of every SIMD run MFMA chains, the other half VALU FMA chains (are the two pipes additive?)
__global__ __launch_bounds__(512) void k_mixed(double *out, int iters_mfma, int iters_fma, double a0, double b0) {
    const int wave = threadIdx.x >> 6;
    double s = 0;
    if ((wave & 1) == 0) {
        double4_t acc[8];
#pragma unroll
        for (int i = 0; i < 8; i++) acc[i] = double4_t{0, 0, 0, 0};
        double a = a0 + threadIdx.x * 1e-9, b = b0;
        for (int it = 0; it < iters_mfma; it++) {
#pragma unroll
            for (int i = 0; i < 8; i++) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 8; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    } else {
        double acc[16];
#pragma unroll
        for (int i = 0; i < 16; i++) acc[i] = i;
        double a = a0 + threadIdx.x * 1e-9, b = b0;
        for (int it = 0; it < iters_fma; it++) {
#pragma unroll
            for (int i = 0; i < 16; i++) acc[i] = __builtin_fma(a, acc[i], b);
        }
#pragma unroll
        for (int i = 0; i < 16; i++) s += acc[i];
    }
    if (s == 12345.678) out[0] = s;
}

template <typename F>
static double time_ms(F f, int reps) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int i = 0; i < reps; i++) f();  // warm-up (clock ramp)
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < reps; i++) f();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms / reps;
}

template <int NACC, int THREADS>
static void run_mfma(double *out, int wg_per_cu) {
    const int iters = 64000 / NACC;
    const int grid = 256 * wg_per_cu;
    double ms = time_ms([&] { hipLaunchKernelGGL((k_mfma<NACC, THREADS>), dim3(grid), dim3(THREADS), 0, 0, out, iters, 1.0, 1e-3); }, 8);
    double flops = (double)grid * (THREADS / 64) * iters * NACC * 2048.0;
    printf("mfma_f64_16x16x4 %2d acc, WG %3d threads, %d WG/CU (%d waves/SIMD): %6.2f TFLOP/s (%.3f ms/launch)\n", NACC, THREADS,
           wg_per_cu, wg_per_cu * THREADS / 256, flops / ms / 1e9, ms);
}

template <int NACC, int THREADS>
static void run_mfma_rand(double *out, int wg_per_cu) {
    const int iters = 64000 / NACC;
    const int grid = 256 * wg_per_cu;
    double ms = time_ms([&] { hipLaunchKernelGGL((k_mfma_rand<NACC, THREADS>), dim3(grid), dim3(THREADS), 0, 0, out, iters, 12345u); }, 8);
    double flops = (double)grid * (THREADS / 64) * iters * NACC * 2048.0;
    double h[3];
    hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
    printf("mfma_f64_16x16x4 RANDOM operands %2d acc, WG %3d threads, %d WG/CU: %6.2f TFLOP/s (%.3f ms/launch), clock64/wall ratio %.3f "
           "(x100 MHz if clock64 ticks at the shader clock)\n", NACC, THREADS, wg_per_cu, flops / ms / 1e9, ms, h[1] / h[2]);
}

int main() {
    double *out;
    hipMalloc(&out, 64);
    if (getenv("EGX_PEAK_RANDOM_ONLY") == nullptr) {
    run_mfma<8, 256>(out, 1);
    run_mfma<8, 256>(out, 2);
    run_mfma<8, 256>(out, 3);
    run_mfma<8, 256>(out, 4);
    run_mfma<8, 256>(out, 8);
    run_mfma<8, 512>(out, 1);
    run_mfma<8, 512>(out, 2);
    run_mfma<8, 512>(out, 4);
    run_mfma<8, 1024>(out, 1);
    run_mfma<8, 128>(out, 2);
    run_mfma<8, 128>(out, 4);
    run_mfma<8, 64>(out, 8);
    run_mfma<16, 512>(out, 1);
    run_mfma<16, 512>(out, 2);
    for (int wg_per_cu = 1; wg_per_cu <= 4; wg_per_cu *= 2) {
        const int grid = 256 * wg_per_cu, iters = 40000;
        double ms = time_ms([&] { hipLaunchKernelGGL(k_fma<256>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0000001, 1e-3); }, 8);
        double flops = (double)grid * 256 * (double)iters * 16 * 2.0;
        printf("v_fma_f64 16 chains, %d waves/SIMD: %6.2f TFLOP/s (%.3f ms/launch)\n", wg_per_cu, flops / ms / 1e9, ms);
    }
    {
        const int im = 8000, iv = 16000;
        double ms = time_ms([&] { hipLaunchKernelGGL(k_mixed, dim3(256), dim3(512), 0, 0, out, im, iv, 1.0000001, 1e-3); }, 8);
        const double fm = 256.0 * 4 * im * 8 * 2048.0, fv = 256.0 * 4 * 64 * (double)iv * 16 * 2.0;
        printf("mixed (4 MFMA + 4 VALU waves/CU, one of each per SIMD): %.3f ms -> mfma part %.2f + valu part %.2f TFLOP/s (upper bounds: each part "
               "divided by the whole launch time)\n", ms, fm / ms / 1e9, fv / ms / 1e9);
    }
    }
    run_mfma<8, 512>(out, 2);
    run_mfma_rand<8, 512>(out, 1);
    run_mfma_rand<8, 512>(out, 2);
    run_mfma_rand<16, 512>(out, 2);
    run_mfma<8, 512>(out, 2);
    return 0;
}
