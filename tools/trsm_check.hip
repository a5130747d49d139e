// Backward error of the solves after a factorisation (launch_trsm_rows: predict_var, theta-gradient) on an ill-conditioned
// kernel matrix: residual of X L^T = B with and without the refinement step of k_panel_trsm (EGX_TRSM_REFINE=0).
#include "../egobox_amd/csrc/kernels_chol.hip"
#include "../egobox_amd/csrc/kernels_pipe.hip"  // (launch_potrf refers to the chain launch)
#include <cstdio>
#include <vector>
#include <cmath>
#include <random>
namespace egx {
void set_error(const std::string &m) { fprintf(stderr, "error: %s\n", m.c_str()); }
hipError_t dev_malloc_bytes(void **p, size_t bytes) { return hipMalloc(p, bytes); }
}  // namespace egx
using namespace egx;
int main() {
    const int n = 1024, m = 128;
    std::mt19937_64 rng(5);
    std::uniform_real_distribution<double> u(0.0, 3.0);
    std::normal_distribution<double> nd;
    std::vector<double> t(n), a((size_t)n * n), b((size_t)m * n);
    for (auto &v : t) v = u(rng);
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) a[(size_t)i * n + j] = std::exp(-(t[i] - t[j]) * (t[i] - t[j]) / 0.02) + (i == j ? 1e-10 : 0.0);
    for (auto &v : b) v = nd(rng);
    double *dM, *dinv, *dB; int *info;
    hipMalloc(&dM, sizeof(double) * n * n); hipMalloc(&dinv, sizeof(double) * dinv_doubles(n)); hipMalloc(&dB, sizeof(double) * m * n);
    hipMalloc(&info, sizeof(double));
    hipMemcpy(dM, a.data(), sizeof(double) * n * n, hipMemcpyHostToDevice);
    hipMemcpy(dB, b.data(), sizeof(double) * m * n, hipMemcpyHostToDevice);
    hipMemset(info, 0, 4);
    if (launch_potrf(0, dM, n, n, n, dinv, info, nullptr, nullptr)) return 1;
    if (launch_trsm_rows(0, dM, n, n, dinv, dB, n, m, 0)) return 1;
    hipDeviceSynchronize();
    std::vector<double> l((size_t)n * n), x((size_t)m * n), fl(n / 64);
    int hinfo;
    hipMemcpy(l.data(), dM, sizeof(double) * n * n, hipMemcpyDeviceToHost);
    hipMemcpy(x.data(), dB, sizeof(double) * m * n, hipMemcpyDeviceToHost);
    hipMemcpy(fl.data(), dinv + (size_t)(n / 64) * 4096, sizeof(double) * (n / 64), hipMemcpyDeviceToHost);
    hipMemcpy(&hinfo, info, 4, hipMemcpyDeviceToHost);
    double worst = 0.0;  // componentwise backward error max_ij |X L^T - B|_ij / (|X| |L^T|)_ij
    for (int i = 0; i < m; i++)
        for (int j = 0; j < n; j++) {
            long double s = 0, sa = 0;
            for (int k = 0; k <= j; k++) {
                s += (long double)x[(size_t)i * n + k] * l[(size_t)j * n + k];
                sa += fabsl((long double)x[(size_t)i * n + k] * l[(size_t)j * n + k]);
            }
            worst = std::fmax(worst, (double)(fabsl(s - b[(size_t)i * n + j]) / (sa + fabsl(b[(size_t)i * n + j]))));
        }
    int nflag = 0;
    for (double f : fl) nflag += f != 0.0;
    printf("info %d, flagged 64x64 tiles %d of %d, componentwise backward error of X L^T = B: %.3e (%.1f eps)\n", hinfo, nflag, n / 64, worst,
           worst / 2.22e-16);
    return 0;
}
