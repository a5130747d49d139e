// Dependent-issue latency of the FP64 VALU / DPP / transcendental instructions the diagonal-tile chain is made of, one
// wave alone on a CU (shader-clock cycles per instruction, 256 dependent instructions each), and the accuracy of
// v_rsq_f64 (what the Newton / Halley refinement has to start from).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include <random>
#define REP16(x) x x x x x x x x x x x x x x x x
#define REP256(x) REP16(REP16(x))
__global__ void k_lat(long long *out, double *sink, double seed) {
    double a = seed + threadIdx.x * 1e-9, b = 1.0000001, c = 1e-9;
    long long t0, t1;
    int n = 0;
#define MEASURE(stmt)                                   \
    __syncthreads();                                    \
    t0 = __builtin_readcyclecounter();                  \
    REP256(stmt)                                        \
    asm volatile("s_nop 0" : "+v"(a));                  \
    t1 = __builtin_readcyclecounter();                  \
    if (threadIdx.x == 0) out[n] = t1 - t0;             \
    n++;
    MEASURE(asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));)
    MEASURE(asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a) : "v"(b));)
    MEASURE(asm volatile("v_fmac_f64_e32 %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));)
    MEASURE(asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(a) : "v"(b), "v"(c));)
    MEASURE(asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %0 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(a));)
    MEASURE(asm volatile("v_rsq_f64 %0, %0" : "+v"(a));)
    MEASURE(asm volatile("s_nop 1\n\tv_rsq_f64_dpp %0, %0 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(a));)
    { int ia = (int)threadIdx.x, ib = 7;
    MEASURE(asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(ia) : "v"(ib));)
    c += ia; }
    MEASURE(asm volatile("v_fma_f64 %0, %1, %2, %2" : "=v"(c) : "v"(b), "v"(a));)  // independent: issue rate
    MEASURE(asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(c) : "v"(b), "v"(b));)  // DPP source old: no wait states
    sink[threadIdx.x] = a + c;
}
__global__ void k_rsq(const double *x, double *y, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = __builtin_amdgcn_rsq(x[i]);
}
int main() {
    long long *out; double *sink;
    hipMalloc(&out, 8 * 16); hipMalloc(&sink, 8 * 64);
    const char *names[] = {"v_fma_f64 dependent", "v_mul_f64 dependent", "v_fmac_f64 dependent", "s_nop 1 + v_fmac_f64_dpp dependent",
                           "s_nop 1 + v_mov_b64_dpp dependent", "v_rsq_f64 dependent", "s_nop 1 + v_rsq_f64_dpp dependent",
                           "v_cndmask_b32 dependent", "v_fma_f64 independent", "v_fmac_f64_dpp on one accumulator (no nop)"};
    for (int rep = 0; rep < 2; rep++) {
        hipLaunchKernelGGL(k_lat, dim3(1), dim3(64), 0, 0, out, sink, 1.0);
        hipDeviceSynchronize();
    }
    long long h[16];
    hipMemcpy(h, out, sizeof h, hipMemcpyDeviceToHost);
    for (int i = 0; i < 10; i++) printf("%-46s %6.1f cycles\n", names[i], h[i] / 256.0);
    const int n = 1 << 20;
    std::vector<double> x(n), y(n);
    std::mt19937_64 rng(1);
    std::uniform_real_distribution<double> u(-30.0, 30.0);
    for (auto &v : x) v = std::exp2(u(rng));
    double *dx, *dy;
    hipMalloc(&dx, 8 * n); hipMalloc(&dy, 8 * n);
    hipMemcpy(dx, x.data(), 8 * n, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_rsq, dim3(n / 256), dim3(256), 0, 0, dx, dy, n);
    hipMemcpy(y.data(), dy, 8 * n, hipMemcpyDeviceToHost);
    double worst = 0;
    for (int i = 0; i < n; i++) worst = std::fmax(worst, std::fabs(y[i] * std::sqrt(x[i]) - 1.0));
    printf("v_rsq_f64: max relative error %.3e = 2^%.1f over %d values in [2^-30, 2^30]\n", worst, std::log2(worst), n);
    return 0;
}
