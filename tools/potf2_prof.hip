// Per-step cycle profile of the diagonal-block kernel (k_potf2_block) and its neighbours, on one 256x256 SPD block.
#define EGX_POTF2_PROFILE 1
#include "../egobox_amd/csrc/kernels_chol.hip"
#include <cstdio>
#include <vector>
#include <cmath>
namespace egx { void set_error(const std::string &m) { fprintf(stderr, "error: %s\n", m.c_str()); } }
using namespace egx;
int main() {
    const int n = 256, ld = 256;
    std::vector<double> a(n * n);
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) a[i * n + j] = std::exp(-0.5 * (i - j) * (i - j) / 900.0) + (i == j ? 1e-3 : 0.0);
    double *dA, *dinv; int *info;
    hipMalloc(&dA, sizeof(double) * n * n); hipMalloc(&dinv, sizeof(double) * 4 * 4096); hipMalloc(&info, 4);
    chol_init();
    for (int rep = 0; rep < 3; rep++) {
        hipMemcpy(dA, a.data(), sizeof(double) * n * n, hipMemcpyHostToDevice);
        hipMemset(info, 0, 4);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_potf2_block<512>, dim3(1), dim3(512), POTF2_LDS_BYTES, 0, dA, (int64_t)ld, n, dinv, info, 0, n);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        long long st[4][6];
        hipMemcpyFromSymbol(st, HIP_SYMBOL(g_potf2_stamps), sizeof st);
        int hinfo; hipMemcpy(&hinfo, info, 4, hipMemcpyDeviceToHost);
        printf("rep %d: kernel %.1f us, info %d\n", rep, ms * 1e3, hinfo);
        for (int s = 0; s < 4; s++)
            printf("  s=%d: load+potf2 %lld  inv %lld  writeback %lld  mfma trsm %lld  mfma syrk %lld cycles\n", s,
                   st[s][5] - st[s][0], st[s][1] - st[s][5], st[s][2] - st[s][1], st[s][3] - st[s][2], st[s][4] - st[s][3]);
    }
    return 0;
}
