// Phase profile of the trailing-update kernel: per workgroup wall-clock stamps (100 MHz constant clock) at
// start / end of the K loop / end of the C read-modify-write, for one big lower-triangular update.
#define EGX_GEMM_PROFILE 1
#include "../egobox_amd/csrc/kernels_chol.hip"
#include <cstdio>
#include <vector>
#include <algorithm>
namespace egx { void set_error(const std::string &m) { fprintf(stderr, "error: %s\n", m.c_str()); } }
using namespace egx;
int main(int argc, char **argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 8192, K = argc > 3 ? atoi(argv[3]) : 256;
    const int64_t ld = n + K;
    double *M;
    hipMalloc(&M, sizeof(double) * (size_t)n * ld);
    hipMemset(M, 0, sizeof(double) * (size_t)n * ld);
    if (argc > 2) {  // random fill (DVFS: zero operands draw less power and clock higher)
        std::vector<double> h((size_t)n * ld);
        unsigned long long x = 88172645463325252ULL;
        for (auto &v : h) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; v = (double)(x >> 11) / 9007199254740992.0 - 0.5; }
        hipMemcpy(M, h.data(), sizeof(double) * h.size(), hipMemcpyHostToDevice);
    }
    chol_init();
    double *C = M + K, *P = M;  // C = columns [K, K+n), panel = columns [0, K)
    for (int rep = 0; rep < 6; rep++) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        launch_gemm_nt_sub(0, C, ld, P, ld, P, ld, n, n, K, 1);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double flops = 2.0 * K * ((double)n * (n + 128) / 2.0);
        printf("rep %d: n=%d K=%d  %.3f ms  %.2f TFLOP/s (lower incl. diagonal tiles)\n", rep, n, K, ms, flops / ms / 1e9);
    }
    static long long st[1 << 16][4];
    hipMemcpyFromSymbol(st, HIP_SYMBOL(g_gemm_stamps), sizeof st);
    std::vector<double> loop, epi; long long tmin = 1LL << 62, tmax = 0; int cnt = 0;
    for (int i = 0; i < (1 << 16); i++) {
        if (st[i][2] == 0) continue;
        loop.push_back((st[i][1] - st[i][0]) / 100.0);
        epi.push_back((st[i][2] - st[i][1]) / 100.0);
        tmin = std::min(tmin, st[i][0]); tmax = std::max(tmax, st[i][2]); cnt++;
    }
    static long long cy[1 << 16][2];
    hipMemcpyFromSymbol(cy, HIP_SYMBOL(g_gemm_cycles), sizeof cy);
    std::vector<double> mhz;
    for (int i = 0; i < (1 << 16); i++)
        if (st[i][2] != 0 && st[i][1] > st[i][0]) mhz.push_back((cy[i][1] - cy[i][0]) / ((st[i][1] - st[i][0]) / 100.0));
    std::sort(mhz.begin(), mhz.end());
    if (!mhz.empty())
        printf("shader clock during the K loop (s_memtime ticks / 100 MHz wall clock): p10 %.0f  p50 %.0f  p90 %.0f MHz\n",
               mhz[(size_t)(0.1 * (mhz.size() - 1))], mhz[(size_t)(0.5 * (mhz.size() - 1))], mhz[(size_t)(0.9 * (mhz.size() - 1))]);
    std::sort(loop.begin(), loop.end()); std::sort(epi.begin(), epi.end());
    auto q = [](std::vector<double> &v, double f) { return v[(size_t)(f * (v.size() - 1))]; };
    printf("%d tiles stamped; span %.1f us\n", cnt, (tmax - tmin) / 100.0);
    printf("K loop  (us): p10 %.1f  p50 %.1f  p90 %.1f\n", q(loop, .1), q(loop, .5), q(loop, .9));
    printf("C RMW   (us): p10 %.1f  p50 %.1f  p90 %.1f\n", q(epi, .1), q(epi, .5), q(epi, .9));
    // per-CU occupancy: how many tiles each SM id processed
    std::vector<int> per(4096, 0);
    for (int i = 0; i < (1 << 16); i++) if (st[i][2]) per[st[i][3] & 4095]++;
    int used = 0, mx = 0, mn = 1 << 30;
    for (int v : per) if (v) { used++; mx = std::max(mx, v); mn = std::min(mn, v); }
    printf("distinct SM ids %d, tiles per SM id min %d max %d\n", used, mn, mx);
    return 0;
}
