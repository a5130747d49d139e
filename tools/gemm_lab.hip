// A/B laboratory for the chip-filling trailing update: k_gemm_stream (LDS-DMA ring, persistent tile walk, store-only
// epilogue) against the register-staged wide k_gemm_nt_sub, same binary, same random operands.
//   gemm_lab [n=15872] [K=512]
// 1. correctness: both kernels on copies of the same matrix, LOWER and full rectangle, max |difference|
// 2. speed: TFLOP/s of each variant (several tiles-per-workgroup settings), interleaved repetitions
#define EGX_STREAM_PROFILE 1
#include "../egobox_amd/csrc/kernels_chol.hip"
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <vector>
namespace egx { void set_error(const std::string &m) { fprintf(stderr, "error: %s\n", m.c_str()); } }
using namespace egx;

static void fill_random(std::vector<double> &h, unsigned long long seed) {
    unsigned long long x = seed;
    for (auto &v : h) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; v = (double)(x >> 11) / 9007199254740992.0 - 0.5; }
}

int main(int argc, char **argv) {
    const int n_speed = argc > 1 ? atoi(argv[1]) : 15872, K = argc > 2 ? atoi(argv[2]) : 512;
    const int n_check = n_speed < 6144 ? n_speed : 6144;  // >= 512 wide tiles, host comparison stays quick
    int n = n_speed;
    const int64_t ld = n_speed + K;
    const size_t elems = (size_t)n * ld;
    double *M0, *M1, *M2;
    hipMalloc(&M0, sizeof(double) * elems);
    hipMalloc(&M1, sizeof(double) * elems);
    hipMalloc(&M2, sizeof(double) * elems);
    {
        std::vector<double> h(elems);
        fill_random(h, 88172645463325252ULL);
        hipMemcpy(M0, h.data(), sizeof(double) * elems, hipMemcpyHostToDevice);
    }
    chol_init();
    auto run = [&](double *M, int lower, int stream, int tpw, int wgs, int variant = 0) {
        g_gemm_stream = stream; g_stream_tpw = tpw; g_stream_wgs = wgs; g_stream_variant = variant;
        return launch_gemm_nt_sub(0, M + K, ld, M, ld, M, ld, n, n, K, lower);
    };
    // ---- correctness (on the leading n_check rows / columns so that the host comparison stays quick)
    n = n_check;
    const int check_variant = argc > 3 ? atoi(argv[3]) : 3;
    for (int lower = 1; lower >= 0; lower--)
        for (int tpw : {0, 1, 3}) {
            hipMemcpy(M1, M0, sizeof(double) * elems, hipMemcpyDeviceToDevice);
            hipMemcpy(M2, M0, sizeof(double) * elems, hipMemcpyDeviceToDevice);
            run(M1, lower, 0, 0, 256);
            run(M2, lower, 1, tpw, 256, check_variant);
            hipDeviceSynchronize();
            std::vector<double> a(elems), b(elems);
            hipMemcpy(a.data(), M1, sizeof(double) * elems, hipMemcpyDeviceToHost);
            hipMemcpy(b.data(), M2, sizeof(double) * elems, hipMemcpyDeviceToHost);
            double worst = 0.0; size_t bad = 0, cmp = 0;
            for (int i = 0; i < n; i++)
                for (int j = 0; j < n; j++) {
                    if (lower && (j / 256) * 256 > (i / 128) * 128 + 127) continue;  // tiles entirely above the diagonal
                    const double d = std::fabs(a[(size_t)i * ld + K + j] - b[(size_t)i * ld + K + j]);
                    cmp++;
                    if (!(d <= 1e-12)) bad++;
                    if (d > worst || d != d) worst = d;
                }
            // the panel columns must be untouched
            size_t pbad = 0;
            std::vector<double> h0(elems);
            hipMemcpy(h0.data(), M0, sizeof(double) * elems, hipMemcpyDeviceToHost);
            for (int i = 0; i < n; i++)
                for (int j = 0; j < K; j++) if (b[(size_t)i * ld + j] != h0[(size_t)i * ld + j]) pbad++;
            printf("check variant %d lower=%d tpw=%d: %zu elements compared, max |stream - wide| = %.3e, %zu beyond 1e-12, panel touched %zu\n",
                   check_variant, lower, tpw, cmp, worst, bad, pbad);
        }
    // ---- speed
    n = n_speed;
    struct V { const char *name; int stream, tpw, wgs, variant; };
    std::vector<V> vs = {{"wide (register staged)", 0, 0, 256, 0}, {"stream persistent v0", 1, 0, 256, 0},
                         {"stream persistent v1 (spread issue)", 1, 0, 256, 1}, {"stream persistent v2 (staggered halves)", 1, 0, 256, 2},
                         {"stream persistent v3 (mid-chunk barrier)", 1, 0, 256, 3}, {"stream persistent v4 (v3 + staggered issue)", 1, 0, 256, 4}, {"stream tpw=1 v4", 1, 1, 256, 4}, {"stream tpw=1 v0", 1, 1, 256, 0}, {"stream tpw=1 v2", 1, 1, 256, 2}, {"stream tpw=1 v3", 1, 1, 256, 3}};
    const double flops = 2.0 * K * ((double)n * (n + 128) / 2.0);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    std::vector<std::vector<double>> ms(vs.size());
    for (int rep = 0; rep < 5; rep++)
        for (size_t v = 0; v < vs.size(); v++) {
            hipEventRecord(e0);
            run(M1, 1, vs[v].stream, vs[v].tpw, vs[v].wgs, vs[v].variant);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float t; hipEventElapsedTime(&t, e0, e1);
            if (rep) ms[v].push_back(t);
        }
    for (size_t v = 0; v < vs.size(); v++) {
        std::sort(ms[v].begin(), ms[v].end());
        const double med = ms[v][ms[v].size() / 2];
        printf("n=%d K=%d LOWER  %-26s median %.3f ms  %.2f TFLOP/s  (min %.3f max %.3f)\n", n, K, vs[v].name, med,
               flops / med / 1e9, ms[v].front(), ms[v].back());
    }
    // ---- per-tile phase breakdown of the stream kernel (one tile per workgroup), shader-clock cycles
    for (int variant = 0; variant < 5; variant++) {
        static long long st[1 << 14][6];
        memset(st, 0, sizeof st);
        hipMemcpyToSymbol(HIP_SYMBOL(g_stream_stamps), st, sizeof st);
        run(M1, 1, 1, 1, 256, variant);
        hipDeviceSynchronize();
        hipMemcpyFromSymbol(st, HIP_SYMBOL(g_stream_stamps), sizeof st);
        std::vector<double> cl, kl, sto, tot, mhz;
        for (int i = 0; i < (1 << 14); i++) {
            if (st[i][5] == 0) continue;
            cl.push_back((double)(st[i][3] - st[i][2]));
            kl.push_back((double)(st[i][4] - st[i][3]));
            sto.push_back((double)(st[i][5] - st[i][4]));
            tot.push_back((double)(st[i][5] - st[i][2]));
            if (st[i][1] > st[i][0]) mhz.push_back((double)(st[i][5] - st[i][2]) / ((st[i][1] - st[i][0]) / 100.0));
        }
        auto med = [](std::vector<double> &v) { std::sort(v.begin(), v.end()); return v.empty() ? 0.0 : v[v.size() / 2]; };
        const double ideal = (double)K / 4.0 * 16.0 * 2.0 * 64.0;  // MFMAs per wave x 2 waves per SIMD x 64 cycles
        printf("variant %d, %zu tiles: C load %.0f  K loop %.0f (MFMA-bound %.0f -> %.3f)  store %.0f  tile %.0f cycles; clock p50 %.0f MHz\n",
               variant, tot.size(), med(cl), med(kl), ideal, ideal / med(kl), med(sto), med(tot), med(mhz));
    }
    return 0;
}
