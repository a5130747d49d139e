// Diagonal-block kernels on one SPD block: the register-resident k_potf2_reg (+ k_diag_tile_inverses) against the
// LDS-tile kernel of round 1 (k_potf2_block) and against a long-double host Cholesky; timings of each, and the
// per-step cycle profile of the round-1 kernel.
#define EGX_POTF2_PROFILE 1
#include "../egobox_amd/csrc/kernels_chol.hip"
#include <cstdio>
#include <vector>
#include <cmath>
#include <random>
namespace egx { void set_error(const std::string &m) { fprintf(stderr, "error: %s\n", m.c_str()); } }
using namespace egx;

static std::vector<double> host_chol(const std::vector<double> &a, int n, int ld, int *bad) {
    std::vector<long double> l((size_t)n * n, 0.0L);
    *bad = 0;
    for (int j = 0; j < n; j++) {
        long double s = a[(size_t)j * ld + j];
        for (int k = 0; k < j; k++) s -= l[j * n + k] * l[j * n + k];
        if (!(s > 0)) { *bad = j + 1; break; }
        const long double d = sqrtl(s);
        l[j * n + j] = d;
        for (int i = j + 1; i < n; i++) {
            long double t = a[(size_t)i * ld + j];
            for (int k = 0; k < j; k++) t -= l[i * n + k] * l[j * n + k];
            l[i * n + j] = t / d;
        }
    }
    std::vector<double> out((size_t)n * n);
    for (size_t e = 0; e < out.size(); e++) out[e] = (double)l[e];
    return out;
}

struct Case { const char *name; int n, ld, off; int kind; };
static void launch_reg(int nw, double *d, int64_t ld, int n, double *lin, int *info, int k0, int nv) {
    if (nw == 16) hipLaunchKernelGGL(k_potf2_reg<16>, dim3(1), dim3(1024), RB_LDS_BYTES, 0, d, ld, n, lin, info, k0, nv);
    else if (nw == 12) hipLaunchKernelGGL(k_potf2_reg<12>, dim3(1), dim3(768), RB_LDS_BYTES, 0, d, ld, n, lin, info, k0, nv);
    else hipLaunchKernelGGL(k_potf2_reg<8>, dim3(1), dim3(512), RB_LDS_BYTES, 0, d, ld, n, lin, info, k0, nv);
}

int main() {
    chol_init();
    const Case cases[] = {{"gauss kernel n=256", 256, 256, 0, 0}, {"random spd n=256 ld=640 off=128", 256, 640, 128, 1},
                          {"gauss kernel n=128", 128, 256, 0, 0}, {"random spd n=192 ld=384 off=128", 192, 384, 128, 1},
                          {"negative pivot at column 100", 256, 256, 0, 2}};
    double *dA, *dB, *dinvA, *dinvB; int *info;
    const int LDMAX = 640;
    hipMalloc(&dA, sizeof(double) * LDMAX * LDMAX); hipMalloc(&dB, sizeof(double) * LDMAX * LDMAX);
    hipMalloc(&dinvA, sizeof(double) * 4 * 4096); hipMalloc(&dinvB, sizeof(double) * 4 * 4096); hipMalloc(&info, 4);
    std::mt19937_64 rng(7);
    std::normal_distribution<double> nd;
    for (int nw : {8, 12, 16})
    for (const Case &c : cases) {
        const int n = c.n, ld = c.ld;
        std::vector<double> a((size_t)LDMAX * LDMAX, 0.0);
        double *blk = a.data() + (size_t)c.off * ld + c.off;
        if (c.kind == 1) {
            std::vector<double> g((size_t)n * n);
            for (auto &v : g) v = nd(rng);
            for (int i = 0; i < n; i++)
                for (int j = 0; j < n; j++) {
                    double s = 0;
                    for (int k = 0; k < n; k++) s += g[i * n + k] * g[j * n + k];
                    blk[(size_t)i * ld + j] = s / n + (i == j ? 0.05 : 0.0);
                }
        } else {
            for (int i = 0; i < n; i++)
                for (int j = 0; j < n; j++)
                    blk[(size_t)i * ld + j] = std::exp(-0.5 * (i - j) * (i - j) / 900.0) + (i == j ? 1e-3 : 0.0);
            if (c.kind == 2) blk[(size_t)100 * ld + 100] = -1.0;
        }
        // the upper triangle must not matter: poison it outside the diagonal 16x16 tiles' own upper parts
        std::vector<double> ap = a;
        double *bp = ap.data() + (size_t)c.off * ld + c.off;
        for (int i = 0; i < n; i++)
            for (int j = i + 1; j < n; j++) bp[(size_t)i * ld + j] = 1e30;
        int bad;
        const std::vector<double> ref = host_chol(std::vector<double>(blk, blk + (size_t)(n - 1) * ld + n), n, ld, &bad);
        std::vector<double> outA((size_t)LDMAX * LDMAX), outB((size_t)LDMAX * LDMAX), iA(4 * 4096), iB(4 * 4096);
        int infoA, infoB;
        // round-1 kernel
        hipMemcpy(dA, a.data(), sizeof(double) * a.size(), hipMemcpyHostToDevice);
        hipMemset(info, 0, 4);
        hipLaunchKernelGGL(k_potf2_block<512>, dim3(1), dim3(512), POTF2_LDS_BYTES, 0, dA + (size_t)c.off * ld + c.off, (int64_t)ld, n, dinvA, info, 0, n);
        hipMemcpy(outA.data(), dA, sizeof(double) * a.size(), hipMemcpyDeviceToHost);
        hipMemcpy(iA.data(), dinvA, sizeof(double) * 4 * 4096, hipMemcpyDeviceToHost);
        hipMemcpy(&infoA, info, 4, hipMemcpyDeviceToHost);
        // register-resident kernel (poisoned upper triangle)
        hipMemcpy(dB, ap.data(), sizeof(double) * a.size(), hipMemcpyHostToDevice);
        hipMemset(info, 0, 4);
        hipMemset(dinvB, 0, sizeof(double) * 4 * 4096);
        launch_reg(nw, dB + (size_t)c.off * ld + c.off, (int64_t)ld, n, dinvB, info, 0, n);
        hipLaunchKernelGGL(k_diag_tile_inverses, dim3(n / 64), dim3(256), 0, 0, (const double *)(dB + (size_t)c.off * ld + c.off), (int64_t)ld, dinvB, (double *)nullptr);
        hipError_t err = hipDeviceSynchronize();
        hipMemcpy(outB.data(), dB, sizeof(double) * a.size(), hipMemcpyDeviceToHost);
        hipMemcpy(iB.data(), dinvB, sizeof(double) * 4 * 4096, hipMemcpyDeviceToHost);
        hipMemcpy(&infoB, info, 4, hipMemcpyDeviceToHost);
        double eA = 0, eB = 0, eI = 0, scale = 0, up = 0;
        int outside = 0;
        if (!bad) {
            const double *oa = outA.data() + (size_t)c.off * ld + c.off, *ob = outB.data() + (size_t)c.off * ld + c.off;
            for (int i = 0; i < n; i++)
                for (int j = 0; j <= i; j++) {
                    scale = std::fmax(scale, std::fabs(ref[i * n + j]));
                    eA = std::fmax(eA, std::fabs(oa[(size_t)i * ld + j] - ref[i * n + j]));
                    eB = std::fmax(eB, std::fabs(ob[(size_t)i * ld + j] - ref[i * n + j]));
                }
            for (int i = 0; i < n; i++)  // strictly upper part inside the 64x64 diagonal tiles: zeros
                for (int j = i + 1; j < (i / 64 + 1) * 64; j++) up = std::fmax(up, std::fabs(ob[(size_t)i * ld + j]));
            for (int t = 0; t < n / 64; t++)
                for (int e = 0; e < 4096; e++) eI = std::fmax(eI, std::fabs(iA[t * 4096 + e] - iB[t * 4096 + e]) / (1.0 + std::fabs(iA[t * 4096 + e])));
            std::vector<char> inside(ap.size(), 0);  // nothing outside the block is touched
            for (int i = 0; i < n; i++)
                for (int j = 0; j < n; j++) inside[(size_t)(c.off + i) * ld + c.off + j] = 1;
            for (size_t e = 0; e < ap.size(); e++)
                if (!inside[e] && outB[e] != ap[e]) outside++;
        }
        printf("[%2d waves] %-36s hip %d  host first bad %d | info old %d new %d | max |L - ref| old %.2e new %.2e (max |L| %.2e) | upper %.1e | "
               "tile inverses new vs old %.2e | writes outside %d\n",
               nw, c.name, (int)err, bad, infoA, infoB, eA, eB, scale, up, eI, outside);
    }
    // ---- timings on the n = 256 kernel matrix
    {
        const int n = 256, ld = 256;
        std::vector<double> a((size_t)n * n);
        for (int i = 0; i < n; i++)
            for (int j = 0; j < n; j++) a[i * n + j] = std::exp(-0.5 * (i - j) * (i - j) / 900.0) + (i == j ? 1e-3 : 0.0);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int nw : {8, 12, 16})
        for (int which = (nw == 8 ? 0 : 1); which < 3; which++)
            for (int rep = 0; rep < 4; rep++) {
                hipMemcpy(dA, a.data(), sizeof(double) * n * n, hipMemcpyHostToDevice);
                hipMemset(info, 0, 4);
                hipDeviceSynchronize();
                hipEventRecord(e0);
                if (which == 0)
                    hipLaunchKernelGGL(k_potf2_block<512>, dim3(1), dim3(512), POTF2_LDS_BYTES, 0, dA, (int64_t)ld, n, dinvA, info, 0, n);
                else {
                    launch_reg(nw, dA, (int64_t)ld, n, dinvB, info, 0, n);
                    if (which == 2) hipLaunchKernelGGL(k_diag_tile_inverses, dim3(n / 64), dim3(256), 0, 0, (const double *)dA, (int64_t)ld, dinvB, (double *)nullptr);
                }
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (rep) printf("%s (%d waves) rep %d: %.1f us\n", which == 0 ? "k_potf2_block<512>" : which == 1 ? "k_potf2_reg" : "k_potf2_reg + k_diag_tile_inverses", which ? nw : 8, rep, ms * 1e3);
                if (which == 1 && rep == 3) {
                    static long long rb[2][16][8];
                    hipMemcpyFromSymbol(rb, HIP_SYMBOL(g_rb_stamps), sizeof rb);
                    const long long t0 = rb[0][0][0];
                    printf("  k_potf2_reg cycles (chain wave: start, length, barrier | update wave 1: phase A, wait, phase B, wait)\n");
                    for (int k = 0; k < 16; k++)
                        printf("  strip %2d: chain at %7lld len %5lld (+barrier %5lld) | A %5lld wait %5lld  B %5lld wait %5lld\n", k,
                               rb[0][k][0] - t0, rb[0][k][1] - rb[0][k][0], k ? rb[0][k][2] - rb[0][k][1] : 0LL, rb[1][k][1] - rb[1][k][0],
                               rb[1][k][2] - rb[1][k][1], rb[1][k][3] - rb[1][k][2], rb[1][k][4] - rb[1][k][3]);
                }
                if (which == 0 && rep == 3) {
                    long long st[4][6];
                    hipMemcpyFromSymbol(st, HIP_SYMBOL(g_potf2_stamps), sizeof st);
                    for (int s = 0; s < 4; s++)
                        printf("  s=%d: load+potf2 %lld  inv %lld  writeback %lld  mfma trsm %lld  mfma syrk %lld cycles\n", s,
                               st[s][5] - st[s][0], st[s][1] - st[s][5], st[s][2] - st[s][1], st[s][3] - st[s][2], st[s][4] - st[s][3]);
                }
            }
    }
    return 0;
}
