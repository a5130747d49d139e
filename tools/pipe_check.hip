// Checker + timer of the pipelined chain kernel (egobox_amd/csrc/kernels_pipe.hip) against the separate-launch chain
// (k_potf2_reg + k_panel_trsm16 + update launches) of launch_potrf, on kernel matrices built on the device:
//   bits      groups of ONE panel (potrf_group = 1): the chain launch holds a diagonal block and its panel solve only -- the
//             same arithmetic as the separate launches, so the factors must agree BIT FOR BIT (this checks every hand-off)
//   groups    the default groups, and the whole factorisation as ONE launch:
//             factor vs the separate launches (relative) and residual max |L L^T - A| / max |A|
//   batch     lock-step batches: every matrix gets the bits it gets alone
//   pivot     a matrix that loses a pivot: same `info` as the separate launches, and the launch ends
//   timeout   strips never published (pipe_stall): the launch drains within the bound and raises its abort word
//   time      median milliseconds of launch_potrf, separate launches vs chain launches
// usage: pipe_check [max_n]   (run under `timeout`: a bug here may spin until the in-kernel bound)
#include "../egobox_amd/csrc/kernels_chol.hip"
#include "../egobox_amd/csrc/kernels_pipe.hip"

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

namespace egx {
void set_error(const std::string &m) { fprintf(stderr, "egx error: %s\n", m.c_str()); }
hipError_t dev_malloc_bytes(void **p, size_t bytes) { return hipMalloc(p, bytes); }
}  // namespace egx
using namespace egx;

#define CK(x)                                                                          \
    do {                                                                               \
        hipError_t e_ = (x);                                                           \
        if (e_ != hipSuccess) {                                                        \
            fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
            exit(2);                                                                   \
        }                                                                              \
    } while (0)

__device__ __forceinline__ double hash01(unsigned a, unsigned b) {
    unsigned h = a * 2654435761u ^ (b + 0x9e3779b9u) * 40503u;
    h ^= h >> 15, h *= 2246822519u, h ^= h >> 13, h *= 3266489917u, h ^= h >> 16;
    return (h >> 8) * (1.0 / 16777216.0);
}
// rows [0, n): squared-exponential kernel matrix of n points in 3-D (+ nugget), identity padding up to n_pad; rows
// [n_pad, m_tot): right-hand sides
__global__ void k_build(double *M, int64_t ld, int n, int n_pad, int m_tot, double scale, double nugget, unsigned seed) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y;
    if (j >= n_pad || i >= m_tot) return;
    double v;
    if (i >= n_pad) v = (j < n) ? hash01(i + seed, j) - 0.5 : 0.0;
    else if (i >= n || j >= n) v = (i == j) ? 1.0 : 0.0;
    else {
        double s = 0.0;
        for (int d = 0; d < 3; d++) {
            const double a = hash01(i * 3 + d, seed), b = hash01(j * 3 + d, seed);
            s += (a - b) * (a - b);
        }
        v = exp(-scale * s) + (i == j ? nugget : 0.0);
    }
    M[(int64_t)i * ld + j] = v;
}
// max |(L L^T)_ij - A_ij| over the lower triangle of the leading n x n block (A rebuilt on the fly), per-thread atomicMax on bits
__global__ void k_resid(const double *L, int64_t ld, int n, double scale, double nugget, unsigned seed, unsigned long long *out) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y;
    if (j > i || i >= n) return;
    double s = 0.0;
    for (int k = 0; k <= j; k++) s += L[(int64_t)i * ld + k] * L[(int64_t)j * ld + k];
    double a = 0.0;
    for (int d = 0; d < 3; d++) {
        const double x = hash01(i * 3 + d, seed), y = hash01(j * 3 + d, seed);
        a += (x - y) * (x - y);
    }
    a = exp(-scale * a) + (i == j ? nugget : 0.0);
    const double e = fabs(s - a);
    atomicMax(out, (unsigned long long)__double_as_longlong(e));
}

static bool g_whole = false;  // PotrfBatch::whole of the next factorisations (the library decides it per handle)
static bool g_flow = false;   // PotrfBatch::flow: the whole factorisation as a FLOW launch (pipe_flow.h)
static int g_flow_tail = 0;    // PotrfBatch::flow_tail: the last columns of a right-looking factorisation as a flow launch
struct Problem {
    int n, n_pad, m_tot, nz;
    int64_t ld;
    size_t mat, dinv_n, sync_n;
    double *M = nullptr, *dinv = nullptr;
    int *info = nullptr, *sync = nullptr;
    PotrfLookahead lk;
    void create(int n_, int nz_, bool lookahead) {
        n = n_, nz = nz_;
        n_pad = (int)round_up(n, n >= 4096 ? 256 : 128);
        m_tot = n_pad + 128;
        ld = n_pad + (getenv("EGX_LD_PAD") ? atoi(getenv("EGX_LD_PAD")) : 0);  // A/B of a padded leading dimension
        mat = (size_t)m_tot * ld;
        dinv_n = (dinv_doubles(n_pad) + 63) / 64 * 64;
        sync_n = pipe_sync_ints(n_pad, m_tot);
        CK(hipMalloc(&M, sizeof(double) * mat * nz));
        CK(hipMalloc(&dinv, sizeof(double) * dinv_n * nz));
        CK(hipMalloc(&info, sizeof(int) * nz));
        CK(hipMalloc(&sync, sizeof(int) * sync_n * nz));
        if (lookahead) {
            int lo, hi;
            CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
            CK(hipStreamCreateWithPriority(&lk.s2, hipStreamNonBlocking, hi));
            CK(hipStreamCreateWithPriority(&lk.s3, hipStreamNonBlocking, hi));
            for (hipEvent_t *e : {&lk.ev_lu, &lk.ev_lur, &lk.ev_panel, &lk.ev_a, &lk.ev_b}) CK(hipEventCreateWithFlags(e, hipEventDisableTiming));
        }
    }
    void destroy() {
        hipFree(M), hipFree(dinv), hipFree(info), hipFree(sync);
        if (lk.s2) {
            hipStreamDestroy(lk.s2), hipStreamDestroy(lk.s3);
            for (hipEvent_t e : {lk.ev_lu, lk.ev_lur, lk.ev_panel, lk.ev_a, lk.ev_b}) hipEventDestroy(e);
        }
    }
    void build(hipStream_t s, double scale, double nugget, unsigned seed0) {
        for (int z = 0; z < nz; z++)
            hipLaunchKernelGGL(k_build, dim3((n_pad + 255) / 256, m_tot), dim3(256), 0, s, M + z * mat, ld, n, n_pad, m_tot, scale, nugget,
                               seed0 + 17u * z);
        CK(hipMemsetAsync(info, 0, sizeof(int) * nz, s));
    }
    int factor(hipStream_t s, bool pipe) {
        PotrfBatch pb;
        pb.count = nz;
        pb.sM = (int64_t)mat;
        pb.sD = (int64_t)dinv_n;
        pb.sI = 1;
        pb.sync = pipe ? sync : nullptr;
        pb.sS = (int64_t)sync_n;
        pb.pipe = pipe ? 1 : 0;
        pb.whole = g_whole ? 1 : 0;
        pb.flow = g_flow && pipe ? 1 : 0;
        pb.flow_tail = pipe ? g_flow_tail : 0;
        return launch_potrf(s, M, ld, n_pad, m_tot, dinv, info, lk.s2 ? &lk : nullptr, nullptr, &pb, nullptr);
    }
    std::vector<double> download(int z) {
        std::vector<double> h(mat);
        CK(hipMemcpy(h.data(), M + z * mat, sizeof(double) * mat, hipMemcpyDeviceToHost));
        return h;
    }
    int abort_word() {
        int v = 0;
        CK(hipMemcpy(&v, sync, sizeof(int), hipMemcpyDeviceToHost));
        return v;
    }
    std::vector<int> infos() {
        std::vector<int> v(nz);
        CK(hipMemcpy(v.data(), info, sizeof(int) * nz, hipMemcpyDeviceToHost));
        return v;
    }
};

static long g_ndiff = 0;
// lower triangle (by 128-tiles, as the factorisation defines it) + right-hand-side rows: max relative difference, and
// whether all those doubles are the same bits
static void compare(const Problem &P, const std::vector<double> &a, const std::vector<double> &b, double &rel, bool &same) {
    double num = 0.0, den = 0.0;
    same = true;
    for (int i = 0; i < P.m_tot; i++) {
        const int jmax = (i < P.n_pad) ? std::min(P.n_pad, (i / 128 + 1) * 128) : P.n_pad;
        for (int j = 0; j < jmax; j++) {
            if (i < P.n_pad && j > i && (j / 64) == (i / 64) && (j / 16) > (i / 16)) continue;  // (zeroed by both)
            if (i < P.n_pad && j > i) continue;
            const double x = a[(size_t)i * P.ld + j], y = b[(size_t)i * P.ld + j];
            if (std::memcmp(&x, &y, 8) != 0) {
                if (same && getenv("PIPE_CHECK_VERBOSE")) printf("   first difference at (%d, %d): %.17g vs %.17g\n", i, j, x, y);
                same = false;
                g_ndiff++;
                if (getenv("PIPE_CHECK_VERBOSE") && g_ndiff < 40 && (g_ndiff % 4) == 0) printf("   diff at (%d, %d): %.3e\n", i, j, x - y);
            }
            num = std::max(num, std::fabs(x - y));
            den = std::max(den, std::fabs(x));
        }
    }
    rel = den > 0 ? num / den : num;
}

static double residual(Problem &P, int z, double scale, double nugget, unsigned seed) {
    unsigned long long *d;
    CK(hipMalloc(&d, 8));
    CK(hipMemset(d, 0, 8));
    hipLaunchKernelGGL(k_resid, dim3((P.n + 255) / 256, P.n), dim3(256), 0, 0, P.M + z * P.mat, P.ld, P.n, scale, nugget, seed + 17u * z, d);
    unsigned long long h;
    CK(hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost));
    hipFree(d);
    double e;
    std::memcpy(&e, &h, 8);
    return e;
}

static int g_fail = 0;
static void verdict(bool ok, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    printf("%s ", ok ? "PASS" : "FAIL");
    vprintf(fmt, ap);
    printf("\n");
    va_end(ap);
    fflush(stdout);
    if (!ok) g_fail++;
}

static double time_factor(Problem &P, bool pipe, double scale, double nugget, int reps) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    std::vector<float> ms;
    for (int r = 0; r < reps + 1; r++) {
        P.build(0, scale, nugget, 1000);
        CK(hipEventRecord(e0, 0));
        if (P.factor(0, pipe)) exit(3);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float t;
        CK(hipEventElapsedTime(&t, e0, e1));
        if (r) ms.push_back(t);
    }
    hipEventDestroy(e0), hipEventDestroy(e1);
    std::sort(ms.begin(), ms.end());
    return ms[ms.size() / 2];
}

// the diagonal-block body in KERNEL context with and without its publishing form (no consumers): what PIPE itself costs
template <bool PIPE>
__global__ __launch_bounds__(1024, 1) void k_diag_alone(double *D, int64_t ld, double *lin, int *info, int *strips) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    RbPublish pub;
    pub.strips = strips;
    (void)rb_factor_block<16, PIPE>(D, ld, 256, lin, info, 0, 256, sm, pub);
}
static int diag_alone_main() {
    Problem P;
    P.create(256, 1, false);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_diag_alone<true>), hipFuncAttributeMaxDynamicSharedMemorySize, RB_LDS_BYTES));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_diag_alone<false>), hipFuncAttributeMaxDynamicSharedMemorySize, RB_LDS_BYTES));
    for (int pipe = 0; pipe < 2; pipe++) {
        std::vector<float> ms;
        for (int r = 0; r < 8; r++) {
            P.build(0, 6.0, 1e-8, 1000);
            CK(hipMemsetAsync(P.sync, 0, 64, 0));
            CK(hipEventRecord(e0, 0));
            if (pipe) hipLaunchKernelGGL(k_diag_alone<true>, dim3(1), dim3(1024), RB_LDS_BYTES, 0, P.M, P.ld, P.dinv, P.info, P.sync + 1);
            else hipLaunchKernelGGL(k_diag_alone<false>, dim3(1), dim3(1024), RB_LDS_BYTES, 0, P.M, P.ld, P.dinv, P.info, P.sync + 1);
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float t;
            CK(hipEventElapsedTime(&t, e0, e1));
            if (r) ms.push_back(t);
        }
        std::sort(ms.begin(), ms.end());
        printf("diag block alone, kernel context, %s: median %.1f us (min %.1f)\n", pipe ? "PIPE (write-through stores, drain, publish)" : "plain", ms[ms.size() / 2] * 1e3, ms[0] * 1e3);
    }
    P.destroy();
    return 0;
}

// one traced factorisation: every task of its chain launches with the times it was taken / became ready / finished
static void env_knobs() {
    if (const char *e = getenv("PIPE_KNOBS")) {
        std::string s(e);
        size_t i = 0;
        while (i < s.size()) {
            size_t j = s.find(',', i);
            if (j == std::string::npos) j = s.size();
            const std::string kv = s.substr(i, j - i);
            const size_t q = kv.find('=');
            if (q != std::string::npos) pipe_set_knob(kv.substr(0, q).c_str(), atoi(kv.substr(q + 1).c_str()));
            i = j + 1;
        }
    }
}
static int trace_main(int n, int whole, int nz) {
    const double scale = 6.0, nugget = 1e-8;
    pipe_set_knob("pipe_timeout_ms", 500);
    g_whole = (whole ? 1 << 30 : 0) != 0;
    env_knobs();
    Problem P;
    P.create(n, nz, n >= 8192);
    const size_t cap = 1 << 20;  // tickets
    long long *d_tr;
    CK(hipMalloc(&d_tr, cap * 64));
    for (int rep = 0; rep < 2; rep++) {
        P.build(0, scale, nugget, 3);
        CK(hipMemset(d_tr, 0, cap * 64));
        pipe_set_trace(rep ? d_tr : nullptr);
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0));
        CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0, 0));
        if (P.factor(0, true)) return 3;
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("# n=%d nz=%d %s: launch_potrf %.3f ms%s\n", n, nz, whole ? "WHOLE" : "per-group", ms, rep ? " (traced)" : "");
    }
    pipe_set_trace(nullptr);
    std::vector<long long> h(cap * 8);
    CK(hipMemcpy(h.data(), d_tr, cap * 64, hipMemcpyDeviceToHost));
    // NOTE: with per-group launches the tickets of later launches overwrite the earlier ones' (same buffer, tickets restart
    // at 0): trace per-group runs show the LAST group that used each ticket; use WHOLE for a complete picture
    struct Rec { long long take, ready, done, w0, w1, xcc, p6, p7; };
    std::vector<Rec> recs;
    long long t0 = -1;
    for (size_t i = 0; i < cap; i++) {
        const long long *r = &h[i * 8];
        if (r[2] == 0) continue;
        recs.push_back({r[2], r[3], r[4], r[0], r[1], r[5], r[6], r[7]});
        if (t0 < 0 || r[2] < t0) t0 = r[2];
    }
    std::sort(recs.begin(), recs.end(), [](const Rec &a, const Rec &b) { return a.take < b.take; });
    const char *names[] = {"TRSM", "FINE", "COARSE", "DIAG"};
    printf("# %zu tasks; times in us from the first ticket (100 MHz clock)\n#   taken    ready     done   (wait   work)  task\n", recs.size());
    double busy[4] = {0, 0, 0, 0}, waitt[4] = {0, 0, 0, 0};
    int cnt[4] = {0, 0, 0, 0};
    for (const Rec &r : recs) {
        const int type = (int)(r.w0 & 255), p = (int)((r.w0 >> 8) & 255), z = (int)(r.w0 >> 16);
        const double a = (r.take - t0) * 0.01, b = (r.ready - t0) * 0.01, c = (r.done - t0) * 0.01;
        cnt[type]++, busy[type] += c - b, waitt[type] += b - a;
        if (type == PT_DIAG || type == PT_TRSM || recs.size() < 600 || (type == PT_FINE && (int)(r.w1 >> 32) == 0 && ((int)r.w1 % 8) == 0))
            printf("%9.2f %8.2f %8.2f  (%6.2f %6.2f)  %-6s p=%d z=%d a=%d b=%d xcd=%d wg=%d%s\n", a, r.ready ? b : -1.0, r.done ? c : -1.0, b - a, c - b,
                   names[type], p, z, (int)(unsigned)r.w1, (int)(r.w1 >> 32), (int)(r.xcc & 15), (int)(r.xcc >> 8),
                   type == PT_DIAG ? (std::string("  first strip published ") + std::to_string((r.p6 - t0) * 0.01) + ", last " + std::to_string((r.p7 - t0) * 0.01)).c_str() : "");
    }
    for (int ty = 0; ty < 4; ty++)
        if (cnt[ty]) printf("# %-6s %5d tasks: mean wait %.2f us, mean work %.2f us\n", names[ty], cnt[ty], waitt[ty] / cnt[ty], busy[ty] / cnt[ty]);
    P.destroy();
    return 0;
}

// ---- FLOW launches (k_potrf_flow, pipe_flow.h): factor vs the separate launches, residual, info; median times of the forms
static int flow_main(int argc, char **argv) {
    const double scale = 6.0, nugget = 1e-8;
    pipe_set_knob("pipe_timeout_ms", 2000);
    std::vector<int> sizes;
    for (int i = 2; i < argc; i++) sizes.push_back(atoi(argv[i]));
    if (sizes.empty()) sizes = {2048, 4096, 6144, 8192};
    for (int n : sizes) {
        Problem P;
        P.create(n, 1, n >= 8192);
        if (P.n_pad % 256) {
            printf("SKIP n=%d: n_pad %d is not a multiple of 256\n", n, P.n_pad);
            P.destroy();
            continue;
        }
        g_flow = false, g_whole = false;
        P.build(0, scale, nugget, 11);
        if (P.factor(0, false)) return 3;
        CK(hipDeviceSynchronize());
        const std::vector<double> ref = P.download(0);
        const double res_ref = residual(P, 0, scale, nugget, 11);
        g_flow = true;
        P.build(0, scale, nugget, 11);
        if (P.factor(0, true)) return 3;
        CK(hipDeviceSynchronize());
        double rel;
        bool same;
        compare(P, ref, P.download(0), rel, same);
        const double res = residual(P, 0, scale, nugget, 11);
        std::vector<int> hdr(8);
        CK(hipMemcpy(hdr.data(), P.sync, 32, hipMemcpyDeviceToHost));
        verdict(rel < 2e-5 && res < 50 * std::max(res_ref, 1e-15) && P.abort_word() == 0 && P.infos()[0] == 0,
                "flow    n=%d: vs separate launches %.2e, residual %.2e (separate %.2e), info %d abort %d (who %x word %d want %d saw %d)", n, rel, res,
                res_ref, P.infos()[0], P.abort_word(), hdr[4], hdr[5], hdr[6], hdr[7]);
        if (P.abort_word() != 0) {
            P.destroy();
            continue;
        }
        // a second time: the same bits (the schedule is dynamic, the arithmetic is not)
        const std::vector<double> first = P.download(0);
        P.build(0, scale, nugget, 11);
        if (P.factor(0, true)) return 3;
        CK(hipDeviceSynchronize());
        compare(P, first, P.download(0), rel, same);
        verdict(same, "flow    n=%d: a second run gives %s", n, same ? "the same bits" : "DIFFERENT bits");
        // a lost pivot: info as the separate launches, the launch ends
        if (n <= 8192) {
            const int bad = n / 2 + 77;
            auto spoil = [&]() {
                const double v = -1.0;
                CK(hipMemcpy(P.M + (size_t)bad * P.ld + bad, &v, 8, hipMemcpyHostToDevice));
            };
            g_flow = false;
            P.build(0, scale, nugget, 11);
            CK(hipDeviceSynchronize());
            spoil();
            if (P.factor(0, false)) return 3;
            CK(hipDeviceSynchronize());
            const int info_ref = P.infos()[0];
            g_flow = true;
            P.build(0, scale, nugget, 11);
            CK(hipDeviceSynchronize());
            spoil();
            if (P.factor(0, true)) return 3;
            CK(hipDeviceSynchronize());
            CK(hipMemcpy(hdr.data(), P.sync, 32, hipMemcpyDeviceToHost));
            verdict(P.infos()[0] == info_ref && info_ref == bad + 1 && P.abort_word() == 0,
                    "flow    n=%d: lost pivot, info %d (separate %d), abort %d (who %x word %d want %d saw %d)", n, P.infos()[0], info_ref, P.abort_word(), hdr[4], hdr[5],
                    hdr[6], hdr[7]);
        }
        g_flow = false;
        const double t_sep = time_factor(P, false, scale, nugget, 5);
        double t_whole = 0.0;
        if (P.n_pad <= 8192) {
            g_whole = true;
            t_whole = time_factor(P, true, scale, nugget, 5);
            g_whole = false;
        }
        g_flow = true;
        const double t_flow = time_factor(P, true, scale, nugget, 5);
        g_flow = false;
        const double fl = (double)n * n * n / 3.0;
        printf("time    n=%d: separate %.3f ms, whole chain launch %.3f ms, FLOW %.3f ms = %.1f TFLOP/s (%.3f of 78.6)\n", n, t_sep, t_whole, t_flow,
               fl / (t_flow * 1e-3) / 1e12, fl / (t_flow * 1e-3) / 1e12 / 78.6);
        fflush(stdout);
        P.destroy();
    }
    printf("%s\n", g_fail ? "FLOW CHECK FAILED" : "flow check ok");
    return g_fail ? 1 : 0;
}

// ---- the flow TAIL: right-looking separate launches for the first columns, one flow launch for the last `tail` columns
static int tail_main(int argc, char **argv) {
    const double scale = 6.0, nugget = 1e-8;
    pipe_set_knob("pipe_timeout_ms", 2000);
    const int n = argc > 2 ? atoi(argv[2]) : 16384;
    Problem P;
    P.create(n, 1, true);
    P.build(0, scale, nugget, 11);
    if (P.factor(0, false)) return 3;
    CK(hipDeviceSynchronize());
    const std::vector<double> ref = P.download(0);
    const double res_ref = residual(P, 0, scale, nugget, 11);
    const double t_sep = time_factor(P, false, scale, nugget, 5);
    printf("time    n=%d: separate launches %.3f ms\n", n, t_sep);
    for (int i = 3; i < argc; i++) {
        g_flow_tail = atoi(argv[i]);
        P.build(0, scale, nugget, 11);
        if (P.factor(0, true)) return 3;
        CK(hipDeviceSynchronize());
        double rel;
        bool same;
        compare(P, ref, P.download(0), rel, same);
        const double res = residual(P, 0, scale, nugget, 11);
        verdict(rel < 2e-5 && res < 50 * std::max(res_ref, 1e-15) && P.abort_word() == 0 && P.infos()[0] == 0,
                "tail    n=%d last %d columns as a flow launch: vs separate launches %.2e, residual %.2e (separate %.2e), info %d abort %d", n, g_flow_tail, rel, res,
                res_ref, P.infos()[0], P.abort_word());
        const int bad = n - g_flow_tail / 2 + 33;  // a lost pivot inside the tail: the index of the whole matrix
        P.build(0, scale, nugget, 11);
        CK(hipDeviceSynchronize());
        const double v = -1.0;
        CK(hipMemcpy(P.M + (size_t)bad * P.ld + bad, &v, 8, hipMemcpyHostToDevice));
        if (P.factor(0, true)) return 3;
        CK(hipDeviceSynchronize());
        verdict(P.infos()[0] == bad + 1 && P.abort_word() == 0, "tail    n=%d: lost pivot in the tail, info %d (want %d), abort %d", n, P.infos()[0], bad + 1, P.abort_word());
        const double t = time_factor(P, true, scale, nugget, 5);
        printf("time    n=%d: flow tail of %d columns %.3f ms (separate %.3f)\n", n, g_flow_tail, t, t_sep);
        fflush(stdout);
    }
    g_flow_tail = 0;
    P.destroy();
    printf("%s\n", g_fail ? "TAIL CHECK FAILED" : "tail check ok");
    return g_fail ? 1 : 0;
}

// one traced flow factorisation: every task with the times it was taken / became ready / finished (100 MHz clock)
static int ftrace_main(int n, int verbose) {
    const double scale = 6.0, nugget = 1e-8;
    pipe_set_knob("pipe_timeout_ms", 2000);
    env_knobs();
    Problem P;
    P.create(n, 1, false);
    const size_t cap = 1 << 19;
    long long *d_tr;
    CK(hipMalloc(&d_tr, cap * 64));
    g_flow = true;
    float ms = 0;
    for (int rep = 0; rep < 2; rep++) {
        P.build(0, scale, nugget, 3);
        CK(hipMemset(d_tr, 0, cap * 64));
        pipe_set_trace(rep ? d_tr : nullptr);
        pipe_set_trace_cap((int)cap);
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0));
        CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0, 0));
        if (P.factor(0, true)) return 3;
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("# n=%d FLOW: launch_potrf %.3f ms%s, abort %d\n", n, ms, rep ? " (traced)" : "", P.abort_word());
    }
    pipe_set_trace(nullptr);
    std::vector<long long> h(cap * 8);
    CK(hipMemcpy(h.data(), d_tr, cap * 64, hipMemcpyDeviceToHost));
    struct Rec { long long take, ready, done, w0, w1, xcc, p6, p7; };
    std::vector<Rec> recs;
    long long t0 = -1, t1 = 0;
    for (size_t i = 0; i < cap; i++) {
        const long long *r = &h[i * 8];
        if (r[2] == 0) continue;
        recs.push_back({r[2], r[3], r[4], r[0], r[1], r[5], r[6], r[7]});
        if (t0 < 0 || r[2] < t0) t0 = r[2];
        if (r[4] > t1) t1 = r[4];
    }
    std::sort(recs.begin(), recs.end(), [](const Rec &a, const Rec &b) { return a.take < b.take; });
    const char *names[] = {"TRSM", "FINE", "COARSE", "DIAG", "LAST", "BULK"};
    double busy[6] = {0}, waitt[6] = {0};
    int cnt[6] = {0};
    printf("# %zu tasks in %.1f us; times in us from the first ticket\n", recs.size(), (t1 - t0) * 0.01);
    double prev_diag_done = 0;
    for (const Rec &r : recs) {
        const int type = (int)(r.w0 & 255), p = (int)((r.w0 >> 8) & 255);
        if (type > 5) continue;
        const double a = (r.take - t0) * 0.01, b = (r.ready - t0) * 0.01, c = (r.done - t0) * 0.01;
        cnt[type]++, busy[type] += c - b, waitt[type] += b - a;
        if (type == PT_DIAG) {
            printf("DIAG p=%2d taken %9.1f ready %9.1f done %9.1f (factor %6.1f, from the previous block's end to ready %7.1f) first strip %8.1f\n", p, a, b, c, c - b,
                   b - prev_diag_done, (r.p6 - t0) * 0.01);
            prev_diag_done = c;
        } else if (verbose) {
            printf("%9.2f %8.2f %8.2f (%6.2f %6.2f) %-6s p=%d a=%d b=%d q=%d xcd=%d wg=%d\n", a, b, c, b - a, c - b, names[type], p, (int)(unsigned)r.w1,
                   (int)(r.w1 >> 32), (int)((r.w0 >> 24) & 255), (int)(r.xcc & 15), (int)(r.xcc >> 8));
        }
    }
    double tot = 0;
    for (int ty = 0; ty < 6; ty++)
        if (cnt[ty]) {
            printf("# %-6s %6d tasks: mean wait %8.2f us, mean work %8.2f us, total work %10.1f us\n", names[ty], cnt[ty], waitt[ty] / cnt[ty], busy[ty] / cnt[ty],
                   busy[ty]);
            tot += busy[ty] + waitt[ty];
        }
    printf("# workgroup time accounted for by tasks (work + in-task wait): %.3f of 256 x %.1f us\n", tot / (256.0 * (t1 - t0) * 0.01), (t1 - t0) * 0.01);
    // occupancy over time: busy workgroups per 50-us slice, by class
    const double slice = ms > 4 ? 100.0 : 25.0;
    const int ns = (int)((t1 - t0) * 0.01 / slice) + 1;
    std::vector<double> occ_b(ns, 0.0), occ_c(ns, 0.0), occ_w(ns, 0.0);
    for (const Rec &r : recs) {
        const int type = (int)(r.w0 & 255);
        if (type > 5 || !r.done) continue;
        const double a = (r.take - t0) * 0.01, b = (r.ready - t0) * 0.01, c = (r.done - t0) * 0.01;
        auto add = [&](std::vector<double> &v, double x0, double x1) {
            for (int i = (int)(x0 / slice); i <= (int)(x1 / slice) && i < ns; i++) {
                const double lo = std::max(x0, i * slice), hi = std::min(x1, (i + 1) * slice);
                if (hi > lo) v[i] += (hi - lo) / slice;
            }
        };
        add(occ_w, a, b);
        add(type == PT_BULK ? occ_b : occ_c, b, c);
    }
    printf("# per %.0f-us slice: workgroups in BULK work | other work | waiting inside a task\n", slice);
    for (int i = 0; i < ns; i++) printf("occ %8.0f %6.1f %6.1f %6.1f\n", i * slice, occ_b[i], occ_c[i], occ_w[i]);
    P.destroy();
    return 0;
}

// flowk <n> <wgs> <k>: k matrices of size n as k CONCURRENT flow launches of `wgs` workgroups each, one stream per matrix
// (round 6, last session: would eight experts fare better as eight narrow flow launches than as one lock-step group?)
static int flowk_main(int n, int wgs, int k) {
    const double scale = 6.0, nugget = 1e-8;
    pipe_set_knob("pipe_timeout_ms", 2000);
    std::vector<Problem> P(k);
    std::vector<hipStream_t> st(k);
    for (int j = 0; j < k; j++) {
        P[j].create(n, 1, false);
        CK(hipStreamCreateWithFlags(&st[j], hipStreamNonBlocking));
    }
    g_flow = true;
    pipe_test_set_workgroups(wgs);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    std::vector<hipEvent_t> done(k);
    for (auto &e : done) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    std::vector<double> ms;
    int aborts = 0;
    for (int rep = 0; rep < 5; rep++) {
        for (int j = 0; j < k; j++) P[j].build(0, scale, nugget, 11 + j);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, 0));
        for (int j = 0; j < k; j++) {
            CK(hipStreamWaitEvent(st[j], e0, 0));
            if (P[j].factor(st[j], true)) return 3;
            CK(hipEventRecord(done[j], st[j]));
            CK(hipStreamWaitEvent(0, done[j], 0));
        }
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float t;
        CK(hipEventElapsedTime(&t, e0, e1));
        if (rep) ms.push_back(t);
        for (int j = 0; j < k; j++) aborts += P[j].abort_word() != 0;
    }
    std::sort(ms.begin(), ms.end());
    const double fl = (double)k * n * n * n / 3.0;
    printf("flowk   n=%d, %d matrices as %d concurrent flow launches of %d workgroups: median %.3f ms (min %.3f) = %.1f TFLOP/s (%.3f of 78.6), aborted launches %d, info %d\n",
           n, k, k, wgs, ms[ms.size() / 2], ms[0], fl / (ms[ms.size() / 2] * 1e-3) / 1e12, fl / (ms[ms.size() / 2] * 1e-3) / 1e12 / 78.6, aborts, P[0].infos()[0]);
    pipe_test_set_workgroups(0);
    g_flow = false;
    for (int j = 0; j < k; j++) P[j].destroy();
    return 0;
}

int main(int argc, char **argv) {
    if (chol_init()) return 1;
    if (argc > 1 && std::string(argv[1]) == "flow") return flow_main(argc, argv);
    if (argc > 4 && std::string(argv[1]) == "flowk") return flowk_main(atoi(argv[2]), atoi(argv[3]), atoi(argv[4]));
    if (argc > 1 && std::string(argv[1]) == "tail") return tail_main(argc, argv);
    if (argc > 1 && std::string(argv[1]) == "ftrace") return ftrace_main(argc > 2 ? atoi(argv[2]) : 4096, argc > 3 ? atoi(argv[3]) : 0);
    if (argc > 1 && std::string(argv[1]) == "diag") return diag_alone_main();
    if (argc > 1 && std::string(argv[1]) == "trace")
        return trace_main(argc > 2 ? atoi(argv[2]) : 1024, argc > 3 ? atoi(argv[3]) : 1, argc > 4 ? atoi(argv[4]) : 1);
    const int max_n = argc > 1 ? atoi(argv[1]) : 8192;
    const double scale = 6.0, nugget = 1e-8;  // cond ~ 1e8..1e10: the refinement step of the solves is exercised
    pipe_set_knob("pipe_timeout_ms", 500);
    // ---------------------------------------------------------------- bits: one panel per chain launch
    set_knob("potrf_group", 1);
    for (int n : {200, 512, 1000, 2048}) {
        for (int nz : {1, 3}) {
            Problem P;
            P.create(n, nz, false);
            P.build(0, scale, nugget, 7);
            if (P.factor(0, false)) return 3;
            CK(hipDeviceSynchronize());
            std::vector<std::vector<double>> ref;
            for (int z = 0; z < nz; z++) ref.push_back(P.download(z));
            P.build(0, scale, nugget, 7);
            if (P.factor(0, true)) return 3;
            CK(hipDeviceSynchronize());
            bool all_same = true;
            double worst = 0.0;
            for (int z = 0; z < nz; z++) {
                double rel;
                bool same;
                compare(P, ref[z], P.download(z), rel, same);
                all_same = all_same && same;
                worst = std::max(worst, rel);
            }
            verdict(all_same && P.abort_word() == 0, "bits    n=%d nz=%d one panel per launch: %s (max rel diff %.2e), abort word %d", n, nz,
                    all_same ? "bit-identical" : "DIFFERENT", worst, P.abort_word());
            P.destroy();
        }
    }
    set_knob("potrf_group", 0);
    // ---------------------------------------------------------------- groups / whole factorisation
    for (int n : {1000, 2048, 4096}) {
        if (n > max_n) continue;
        Problem P;
        P.create(n, 1, false);
        P.build(0, scale, nugget, 11);
        if (P.factor(0, false)) return 3;
        CK(hipDeviceSynchronize());
        const std::vector<double> ref = P.download(0);
        const double res_ref = residual(P, 0, scale, nugget, 11);
        for (int whole : {0, 1}) {
            g_whole = whole != 0;
            P.build(0, scale, nugget, 11);
            if (P.factor(0, true)) return 3;
            CK(hipDeviceSynchronize());
            double rel;
            bool same;
            compare(P, ref, P.download(0), rel, same);
            const double res = residual(P, 0, scale, nugget, 11);
            verdict(rel < 2e-5 && res < 50 * std::max(res_ref, 1e-15) && P.abort_word() == 0 && P.infos()[0] == 0,
                    "groups  n=%d %s: vs separate launches %.2e, residual %.2e (separate launches %.2e), info %d abort %d", n,
                    whole ? "WHOLE" : "per-group", rel, res, res_ref, P.infos()[0], P.abort_word());
        }
        g_whole = false;
        P.destroy();
    }
    // ---------------------------------------------------------------- lock-step batches: the bits a matrix gets alone
    for (int whole : {0, 1 << 30}) {
        g_whole = (whole) != 0;
        const int n = 1500, nz = 4;
        Problem B, S;
        B.create(n, nz, false);
        S.create(n, 1, false);
        B.build(0, scale, nugget, 23);
        if (B.factor(0, true)) return 3;
        CK(hipDeviceSynchronize());
        bool ok = true;
        for (int z = 0; z < nz; z++) {
            S.build(0, scale, nugget, 23 + 17u * z);
            if (S.factor(0, true)) return 3;
            CK(hipDeviceSynchronize());
            double rel;
            bool same;
            compare(S, S.download(0), B.download(z), rel, same);
            ok = ok && same;
        }
        verdict(ok && B.abort_word() == 0, "batch   n=%d nz=%d %s: every matrix %s its lone factor", n, nz, whole ? "WHOLE" : "per-group",
                ok ? "bit-identical to" : "DIFFERENT from");
        // smaller grids: the same bits on 3 and on 17 workgroups
        bool ok2 = true;
        const std::vector<double> full = B.download(1);
        for (int wgs : {3, 17}) {
            pipe_test_set_workgroups(wgs);
            B.build(0, scale, nugget, 23);
            if (B.factor(0, true)) return 3;
            CK(hipDeviceSynchronize());
            double rel;
            bool same;
            compare(B, full, B.download(1), rel, same);
            ok2 = ok2 && same && B.abort_word() == 0;
        }
        pipe_test_set_workgroups(0);
        verdict(ok2, "grid    n=%d nz=%d %s: the same bits on 3, 17 and all workgroups", n, nz, whole ? "WHOLE" : "per-group");
        B.destroy(), S.destroy();
    }
    g_whole = (0) != 0;
    // ---------------------------------------------------------------- a lost pivot
    for (int whole : {0, 1 << 30}) {
        g_whole = (whole) != 0;
        Problem P;
        P.create(1200, 2, false);
        int infos[2][2];
        for (int pipe = 0; pipe < 2; pipe++) {
            P.build(0, scale, nugget, 31);
            const double bad = -3.0;  // matrix 1 loses pivot 701 (1-based), matrix 0 stays fine
            CK(hipMemcpy(P.M + P.mat + (size_t)700 * P.ld + 700, &bad, 8, hipMemcpyHostToDevice));
            if (P.factor(0, pipe != 0)) return 3;
            CK(hipDeviceSynchronize());
            infos[pipe][0] = P.infos()[0], infos[pipe][1] = P.infos()[1];
        }
        verdict(infos[0][0] == 0 && infos[0][1] == 701 && infos[1][0] == 0 && infos[1][1] == 701 && P.abort_word() == 0,
                "pivot   %s: info separate launches (%d, %d), chain launches (%d, %d), abort %d", whole ? "WHOLE" : "per-group", infos[0][0],
                infos[0][1], infos[1][0], infos[1][1], P.abort_word());
        P.destroy();
    }
    g_whole = (0) != 0;
    // ---------------------------------------------------------------- strips that are never published
    {
        Problem P;
        P.create(1024, 1, false);
        pipe_set_knob("pipe_timeout_ms", 20);
        pipe_set_knob("pipe_stall", 1);
        P.build(0, scale, nugget, 5);
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0));
        CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0, 0));
        if (P.factor(0, true)) return 3;
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        verdict(P.abort_word() != 0 && ms < 2000.0f, "timeout strips never published: abort word %d, the factorisation's launches drained in %.1f ms (bound 20 ms per wait)",
                P.abort_word(), ms);
        pipe_set_knob("pipe_stall", 0);
        pipe_set_knob("pipe_timeout_ms", 500);
        // ... and the next factorisation on the same words is fine again
        P.build(0, scale, nugget, 5);
        if (P.factor(0, true)) return 3;
        CK(hipDeviceSynchronize());
        verdict(P.abort_word() == 0 && P.infos()[0] == 0, "timeout the next factorisation is clean: abort %d info %d", P.abort_word(), P.infos()[0]);
        P.destroy();
    }
    // ---------------------------------------------------------------- timings
    printf("\n# median ms of launch_potrf (one matrix unless nz is given); flops = n_pad^3 / 3\n");
    for (int n : {1024, 2048, 4096, 8192, 16384}) {
        if (n > max_n) continue;
        for (int nz : {1, 8}) {
            if (nz > 1 && n != 4096 && n != 8192) continue;
            Problem P;
            P.create(n, nz, n >= 8192);
            const double fl = nz * (double)P.n_pad * P.n_pad * P.n_pad / 3.0;
            const int reps = n >= 8192 ? 3 : 7;
            const double t_sep = time_factor(P, false, scale, nugget, reps);
            g_whole = (0) != 0;
            const double t_grp = time_factor(P, true, scale, nugget, reps);
            printf("time    n=%5d nz=%d separate %8.3f ms (%5.1f TFLOP/s) | chain per group %8.3f ms (%5.1f)", n, nz, t_sep, fl / t_sep * 1e-9,
                   t_grp, fl / t_grp * 1e-9);
            if (n <= 4096) {
                g_whole = true;
                const double t = time_factor(P, true, scale, nugget, reps);
                printf(" | whole %8.3f ms (%5.1f)", t, fl / t * 1e-9);
                g_whole = false;
            }
            printf(" abort %d\n", P.abort_word());
            fflush(stdout);
            P.destroy();
        }
    }
    printf("\n%s (%d failed)\n", g_fail ? "SOME CHECKS FAILED" : "ALL CHECKS PASSED", g_fail);
    return g_fail ? 1 : 0;
}
