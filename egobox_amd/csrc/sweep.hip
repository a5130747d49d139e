// Multi-GPU theta sweep behind the C ABI (include/egx_gp.h, egx_sweep_*): one process per GPU; candidate c goes to
// rank c mod world (static) or to whichever rank pulls it first from a node-wide counter in POSIX shared memory
// (dynamic: candidates that are not positive definite return ~10x sooner than the others); every rank evaluates its
// share on its own device through the handle's batched likelihood path, ONE RCCL all-gather of {likelihood, status}
// (16 B per candidate and rank) over xGMI assembles the result on every rank.
//
// Failure safety: the all-gather is ALWAYS reached.  A rank whose local work failed contributes a poisoned payload
// (sweep_shard.h) and returns its error afterwards; the other ranks return EGX_ERR_PEER with the failed rank in the
// message.  The staging buffers are allocated once in egx_sweep_create (payloads larger than them go in chunks), and
// the wait for the collective has a deadline (EGX_SWEEP_TIMEOUT_S, default 1800 s) after which the communicator is
// aborted instead of blocking for ever on a peer that died.
//
// Reference seam: the rayon multistart of crates/gp/src/algorithm.rs:928-945 (independent likelihood evaluations,
// arg-min reduce :942-945) and the serial expert loop of crates/moe/src/algorithm.rs:167-177.  Evaluations at
// different theta share nothing but the replicated (4 MiB) training set, so there is no data-path collective besides
// this gather.
//
// RCCL is bound at run time (dlopen of librccl.so.1 on the first sweep call): libegx_gp_hip.so itself keeps
// libamdhip64 as its only link-time dependency, and single-GPU users never load the collective library.
#include "gp_handle.h"
#include "sweep_shard.h"

#include <dlfcn.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <unistd.h>
#include <cstdio>
#include <rccl/rccl.h>

namespace egx {

struct RcclApi {
    void *lib = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclCommAbort) CommAbort = nullptr;  // optional
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    decltype(&ncclGetVersion) GetVersion = nullptr;
    std::string err;
};

static RcclApi &rccl() {
    static RcclApi api = [] {
        RcclApi a;
        const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char *nm : names) {
            a.lib = dlopen(nm, RTLD_NOW | RTLD_LOCAL);
            if (a.lib) break;
        }
        if (!a.lib) {
            const char *e = dlerror();
            a.err = std::string("cannot load librccl.so.1: ") + (e ? e : "unknown error");
            return a;
        }
#define EGX_SYM(field, name)                                                  \
    a.field = reinterpret_cast<decltype(a.field)>(dlsym(a.lib, name));        \
    if (!a.field && a.err.empty()) a.err = std::string("librccl: missing symbol ") + name;
        EGX_SYM(GetUniqueId, "ncclGetUniqueId");
        EGX_SYM(CommInitRank, "ncclCommInitRank");
        EGX_SYM(CommDestroy, "ncclCommDestroy");
        EGX_SYM(AllGather, "ncclAllGather");
        EGX_SYM(GetErrorString, "ncclGetErrorString");
        EGX_SYM(GetVersion, "ncclGetVersion");
#undef EGX_SYM
        a.CommAbort = reinterpret_cast<decltype(a.CommAbort)>(dlsym(a.lib, "ncclCommAbort"));
        return a;
    }();
    return api;
}

#define EGX_NCCL_CHECK(expr)                                                                   \
    do {                                                                                       \
        ncclResult_t _r = (expr);                                                              \
        if (_r != ncclSuccess) {                                                               \
            ::egx::set_error(std::string(#expr) + ": " + ::egx::rccl().GetErrorString(_r));    \
            return EGX_ERR_HIP;                                                                \
        }                                                                                      \
    } while (0)

}  // namespace egx

using namespace egx;

// node-wide candidate counters of the dynamic assignment: call number s of a sweep uses slot s % kSlots; rank 0 zeroes
// slot (s + kSlots / 2) % kSlots at the start of call s (nobody touches that slot for the next kSlots / 2 - 1 calls, and
// every call ends with an all-gather, which no rank leaves before every rank has stopped pulling)
struct SweepCounters {
    static constexpr int kSlots = 64;
    std::atomic<int64_t> next[kSlots];
    // HOST TRANSPORT of the all-gather (EGX_SWEEP_TRANSPORT=shm, see egx_sweep_create): two monotonic arrival counters
    // (a barrier = "add one, wait until the count reaches generation x world") and one slab per rank
    static constexpr int kMaxWorld = 16;
    static constexpr int64_t kSlab = 8192;  // doubles per rank and round
    std::atomic<int64_t> arrived_a, arrived_b;
    double gather[kMaxWorld * kSlab];
};
static_assert(std::atomic<int64_t>::is_always_lock_free, "the shared counters must be plain lock-free words");

struct egx_sweep {
    egx_gp *gp = nullptr;
    int rank = 0, world = 1;
    ncclComm_t comm = nullptr;
    hipStream_t stream = nullptr;
    // staging of the all-gather, allocated once: kChunk doubles per rank
    static constexpr int64_t kChunk = 262144;  // 2 MiB per rank: a 100 000-point mean + variance pair goes in one piece
    double *d_send = nullptr, *d_recv = nullptr;
    double *h_send = nullptr, *h_recv = nullptr;  // pinned
    std::mutex mu;
    int64_t n_allgathers = 0;
    // dynamic assignment
    int dynamic = 0;
    SweepCounters *counters = nullptr;  // shared memory (world > 1) or heap (world == 1)
    bool counters_shared = false;
    bool shm_transport = false;  // the all-gather goes through `counters->gather` instead of RCCL (test transport)
    int64_t shm_generation = 0;
    std::string shm_name;
    int64_t call_seq = 0;
    // balance of the last call
    std::vector<int64_t> last_per_rank;
    double last_eval_s = 0.0;
    double timeout_s = 1800.0;
};

namespace {

// candidates of one rank in the order it evaluates them
struct StaticSource final : egx::CandidateSource {
    int64_t k, next;
    int world;
    StaticSource(int64_t k_, int rank, int world_) : k(k_), next(rank), world(world_) {}
    int pull(int want, int64_t *out) override {
        int got = 0;
        while (got < want && next < k) {
            out[got++] = next;
            next += world;
        }
        return got;
    }
};
struct DynamicSource final : egx::CandidateSource {
    std::atomic<int64_t> *ctr;
    int64_t k;
    DynamicSource(std::atomic<int64_t> *c, int64_t k_) : ctr(c), k(k_) {}
    int pull(int want, int64_t *out) override {
        if (ctr->load(std::memory_order_relaxed) >= k) return 0;
        const int64_t c0 = ctr->fetch_add(want, std::memory_order_relaxed);
        int got = 0;
        for (int64_t c = c0; c < c0 + want && c < k; c++) out[got++] = c;
        return got;
    }
};

// wait for the sweep's stream with a deadline; on expiry the communicator is aborted (a peer died or never arrived)
int sweep_wait(egx_sweep *sw) {
    const auto t0 = std::chrono::steady_clock::now();
    for (int spins = 0;; spins++) {
        const hipError_t q = hipStreamQuery(sw->stream);
        if (q == hipSuccess) return EGX_SUCCESS;
        if (q != hipErrorNotReady) {
            set_error(std::string("sweep collective: ") + hipGetErrorString(q));
            (void)hipGetLastError();
            return EGX_ERR_HIP;
        }
        if (spins > 2000) std::this_thread::sleep_for(std::chrono::microseconds(50));
        if ((spins & 1023) == 1023) {
            const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            if (el > sw->timeout_s) {
                if (sw->comm && rccl().CommAbort) {
                    rccl().CommAbort(sw->comm);
                    sw->comm = nullptr;
                }
                set_error("sweep collective: no answer from the other ranks within " + std::to_string((int)sw->timeout_s) +
                          " s (EGX_SWEEP_TIMEOUT_S); communicator aborted");
                return EGX_ERR_PEER;
            }
        }
    }
}

}  // namespace

// The HOST transport (EGX_SWEEP_TRANSPORT=shm): the same all-gather through the ranks' shared-memory segment.  It
// exists so that the world > 1 logic of this file -- sharding, the dynamic counter, poisoned payloads, the deadline --
// can EXECUTE where RCCL cannot build a communicator: several ranks on ONE GPU (RCCL refuses duplicate devices), i.e. the
// one-GPU box the test-suite runs on (tests/test_gpu_configs.py::test_sweep_two_ranks_*).  The payload is 16 bytes per
// candidate either way; the product transport is RCCL (ncclAllGather over xGMI), which every run with a communicator uses.
static int shm_barrier(egx_sweep *sw, std::atomic<int64_t> &ctr, int64_t generation) {
    ctr.fetch_add(1, std::memory_order_acq_rel);
    const int64_t want = generation * sw->world;
    const auto t0 = std::chrono::steady_clock::now();
    for (int64_t spins = 0; ctr.load(std::memory_order_acquire) < want; spins++) {
        if (spins > 4000) std::this_thread::sleep_for(std::chrono::microseconds(20));
        if ((spins & 4095) == 4095 &&
            std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > sw->timeout_s) {
            set_error("sweep collective (host transport): no answer from the other ranks within " +
                      std::to_string((int)sw->timeout_s) + " s (EGX_SWEEP_TIMEOUT_S)");
            return EGX_ERR_PEER;
        }
    }
    return EGX_SUCCESS;
}
static int shm_allgather_doubles(egx_sweep *sw, const double *send, int64_t count, double *recv) {
    SweepCounters *sc = sw->counters;
    for (int64_t off = 0; off < count; off += SweepCounters::kSlab) {
        const int64_t len = std::min<int64_t>(SweepCounters::kSlab, count - off);
        const int64_t gen = ++sw->shm_generation;
        std::memcpy(sc->gather + (size_t)sw->rank * SweepCounters::kSlab, send + off, sizeof(double) * len);
        EGX_RC(shm_barrier(sw, sc->arrived_a, gen));  // every slab of this round is written
        for (int r = 0; r < sw->world; r++)
            std::memcpy(recv + (size_t)r * count + off, sc->gather + (size_t)r * SweepCounters::kSlab, sizeof(double) * len);
        EGX_RC(shm_barrier(sw, sc->arrived_b, gen));  // ... and read by everybody before the next round overwrites it
        sw->n_allgathers++;
    }
    return EGX_SUCCESS;
}

// all ranks: recv (world x count doubles) <- concatenation over ranks of send (count doubles); host buffers, pinned
// staging allocated in egx_sweep_create, one ncclAllGather per kChunk doubles on the sweep's stream
static int sweep_allgather_doubles(egx_sweep *sw, const double *send, int64_t count, double *recv) {
    if (sw->world == 1 && !sw->comm) {
        std::memcpy(recv, send, sizeof(double) * count);
        return EGX_SUCCESS;
    }
    if (sw->shm_transport) return shm_allgather_doubles(sw, send, count, recv);
    if (!sw->comm) {
        set_error("sweep collective: the communicator was aborted by an earlier failure");
        return EGX_ERR_PEER;
    }
    for (int64_t off = 0; off < count; off += egx_sweep::kChunk) {
        const int64_t len = std::min<int64_t>(egx_sweep::kChunk, count - off);
        std::memcpy(sw->h_send, send + off, sizeof(double) * len);
        EGX_HIP_CHECK(hipMemcpyAsync(sw->d_send, sw->h_send, sizeof(double) * len, hipMemcpyHostToDevice, sw->stream));
        EGX_NCCL_CHECK(rccl().AllGather(sw->d_send, sw->d_recv, (size_t)len, ncclDouble, sw->comm, sw->stream));
        EGX_HIP_CHECK(hipMemcpyAsync(sw->h_recv, sw->d_recv, sizeof(double) * len * sw->world, hipMemcpyDeviceToHost,
                                     sw->stream));
        EGX_RC(sweep_wait(sw));
        for (int r = 0; r < sw->world; r++)
            std::memcpy(recv + (size_t)r * count + off, sw->h_recv + (size_t)r * len, sizeof(double) * len);
        sw->n_allgathers++;
    }
    return EGX_SUCCESS;
}

namespace egx {
// Responsibilities of a Gaussian mixture at m points (GaussianMixture::predict_probas, crates/moe/src/gaussian_mixture.rs:
// 114-121, 231-283): one lane per point, the workgroup's 64 points staged in LDS (row stride d | 1: a lane walks its own
// row), the k x d x d scaled precision factors and the means read as wave-uniform operands.  par = [log w_c + log det_c -
// 0.5 d ln 2 pi] (k).  q_c = || (x - mu_c) P_c ||^2; weighted log probability, the sum of the exponentials above
// f64::MIN_10_EXP, its logarithm unless the sum is below epsilon (:236-251), exp of the difference (:119).
__global__ __launch_bounds__(64) void k_gmx_probas(const double *__restrict__ xq, int64_t m, int d, int k,
                                                  const double *__restrict__ means, const double *__restrict__ precs,
                                                  const double *__restrict__ par, double *__restrict__ probas) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const int64_t q0 = (int64_t)blockIdx.x * 64;
    const int lane = threadIdx.x, ds = d | 1;
    const int rows = (int)((m - q0 < 64) ? (m - q0) : 64);
    for (int e = lane; e < rows * d; e += 64) {
        const int i = e / d, j = e - i * d;
        sm[i * ds + j] = xq[q0 * d + e];
    }
    __syncthreads();
    if (lane >= rows) return;
    const double *x = sm + lane * ds;
    double *out = probas + (q0 + lane) * k;
    double s = 0.0;
    for (int c = 0; c < k; c++) {
        const double *mu = means + (size_t)c * d, *P = precs + (size_t)c * d * d;
        double q = 0.0;
        for (int j = 0; j < d; j++) {
            double acc = 0.0;
            for (int i = 0; i < d; i++) acc = __builtin_fma(x[i] - mu[i], P[(size_t)i * d + j], acc);
            q = __builtin_fma(acc, acc, q);
        }
        const double wlp = par[c] - 0.5 * q;
        out[c] = wlp;
        s += (wlp <= -307.0) ? 0.0 : exp(wlp);
    }
    const double norm = (fabs(s) < 2.220446049250313e-16) ? 0.0 : log(s);
    for (int c = 0; c < k; c++) out[c] = exp(out[c] - norm);
}

// GaussianMixture::predict_probas_derivatives (crates/moe/src/gaussian_mixture.rs:127-170), one lane per point:
//   u_c = w_c pdf_c(x),  v = sum_c u_c,  deriv_c = (x - mu_c) precisions_c / hf,  u'_c = -deriv_c u_c,  v' = sum_c u'_c
//   d p_c / d x = (u'_c v - u_c v') / v^2
// With the scaled factor P' = precisions_chol_c hf^-1/2 (what pdfs() itself uses, :253-283): z = (x - mu_c) P' gives both the
// quadratic form |z|^2 of the pdf and deriv_c = z P'^T (precisions = P P^T, :208-217).  Pass A writes u'_c to the output and
// accumulates v, v'; pass B finishes the output in place.  Per-lane scratch (x, z, v': d each; u: k) lives in LDS.
__global__ __launch_bounds__(64) void k_gmx_probas_deriv(const double *__restrict__ xq, int64_t m, int d, int k,
                                                        const double *__restrict__ means, const double *__restrict__ precs,
                                                        const double *__restrict__ par, double *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const int64_t q0 = (int64_t)blockIdx.x * 64;
    const int lane = threadIdx.x, ds = d | 1, ks = k | 1;
    const int rows = (int)((m - q0 < 64) ? (m - q0) : 64);
    double *xs = sm, *zs = sm + 64 * ds, *vps = zs + 64 * ds, *us = vps + 64 * ds;
    for (int e = lane; e < rows * d; e += 64) {
        const int i = e / d, j = e - i * d;
        xs[i * ds + j] = xq[q0 * d + e];
    }
    __syncthreads();
    if (lane >= rows) return;
    const double *x = xs + lane * ds;
    double *z = zs + lane * ds, *vp = vps + lane * ds, *u = us + lane * ks;
    double *o = out + (q0 + lane) * (int64_t)k * d;
    for (int l = 0; l < d; l++) vp[l] = 0.0;
    double v = 0.0;
    for (int c = 0; c < k; c++) {
        const double *mu = means + (size_t)c * d, *P = precs + (size_t)c * d * d;
        double q = 0.0;
        for (int j = 0; j < d; j++) {
            double acc = 0.0;
            for (int i = 0; i < d; i++) acc = __builtin_fma(x[i] - mu[i], P[(size_t)i * d + j], acc);
            z[j] = acc;
            q = __builtin_fma(acc, acc, q);
        }
        const double uc = exp(par[c] - 0.5 * q);  // w_c pdf_c(x)  (:136-139: no MIN_10_EXP guard on this path)
        u[c] = uc;
        v += uc;
        for (int l = 0; l < d; l++) {
            double acc = 0.0;
            for (int j = 0; j < d; j++) acc = __builtin_fma(z[j], P[(size_t)l * d + j], acc);
            const double up = -acc * uc;
            o[(int64_t)c * d + l] = up;
            vp[l] += up;
        }
    }
    const double v2 = v * v;
    for (int c = 0; c < k; c++)
        for (int l = 0; l < d; l++) o[(int64_t)c * d + l] = (o[(int64_t)c * d + l] * v - u[c] * vp[l]) / v2;
}
}  // namespace egx

extern "C" {

int32_t egx_sweep_unique_id(void *id_out) {
    if (!id_out) {
        set_error("egx_sweep_unique_id: NULL output");
        return EGX_ERR_INVALID_VALUE;
    }
    static_assert(sizeof(ncclUniqueId) == EGX_SWEEP_ID_BYTES, "EGX_SWEEP_ID_BYTES must match ncclUniqueId");
    RcclApi &api = rccl();
    if (!api.err.empty()) {
        set_error(api.err);
        return EGX_ERR_UNSUPPORTED;
    }
    ncclUniqueId id;
    EGX_NCCL_CHECK(api.GetUniqueId(&id));
    std::memcpy(id_out, &id, sizeof id);
    return EGX_SUCCESS;
}

int32_t egx_sweep_create(const egx_gp_config *cfg_in, const double *x, const double *y, int64_t n, int64_t d,
                         const void *nccl_id, int32_t rank, int32_t world, egx_sweep **out) {
    if (!out) {
        set_error("out handle pointer is NULL");
        return EGX_ERR_INVALID_VALUE;
    }
    *out = nullptr;
    if (world < 1 || rank < 0 || rank >= world) {
        set_error("egx_sweep_create: need 0 <= rank < world, got rank " + std::to_string(rank) + " world " +
                  std::to_string(world));
        return EGX_ERR_INVALID_VALUE;
    }
    if (world > 1 && !nccl_id) {
        set_error("egx_sweep_create: world > 1 needs the unique id of egx_sweep_unique_id (from rank 0)");
        return EGX_ERR_INVALID_VALUE;
    }
    egx_gp_config cfg;
    if (cfg_in) cfg = *cfg_in; else egx_gp_config_default(&cfg);
    if (cfg.n_workspaces < 2) cfg.n_workspaces = 2;  // two candidates in flight per GPU (panel latency hidden)
    egx_sweep *sw = new egx_sweep();
    sw->rank = rank;
    sw->world = world;
    if (const char *e = std::getenv("EGX_SWEEP_TIMEOUT_S")) {
        const double t = std::atof(e);
        if (t > 0.0) sw->timeout_s = t;
    }
    int rc = egx_gp_create(&cfg, x, y, n, d, &sw->gp);
    if (rc) {
        delete sw;
        return rc;
    }
    auto fail = [&](int r) {
        egx_sweep_destroy(sw);
        return r;
    };
    if (hipSetDevice(sw->gp->device) != hipSuccess ||
        hipStreamCreateWithFlags(&sw->stream, hipStreamNonBlocking) != hipSuccess) {
        set_error("egx_sweep_create: stream creation failed");
        (void)hipGetLastError();
        return fail(EGX_ERR_HIP);
    }
    {   // staging of the collective: allocated here, never inside a collective call
        const size_t one = sizeof(double) * (size_t)egx_sweep::kChunk;
        if (dev_malloc(&sw->d_send, one) != hipSuccess || dev_malloc(&sw->d_recv, one * world) != hipSuccess ||
            hipHostMalloc(&sw->h_send, one, hipHostMallocDefault) != hipSuccess ||
            hipHostMalloc(&sw->h_recv, one * world, hipHostMallocDefault) != hipSuccess) {
            set_error("egx_sweep_create: staging buffers");
            (void)hipGetLastError();
            return fail(EGX_ERR_HIP);
        }
    }
    // candidate counters of the dynamic assignment: shared by the ranks of this node (named after the unique id)
    if (world > 1) {
        uint64_t hsh = 1469598103934665603ull;  // FNV-1a of the id
        for (int i = 0; i < EGX_SWEEP_ID_BYTES; i++) hsh = (hsh ^ ((const unsigned char *)nccl_id)[i]) * 1099511628211ull;
        char nm[64];
        std::snprintf(nm, sizeof nm, "/egx_sweep_%016llx", (unsigned long long)hsh);
        sw->shm_name = nm;
        const int fd = shm_open(nm, O_CREAT | O_RDWR, 0600);
        void *mem = MAP_FAILED;
        if (fd >= 0 && ftruncate(fd, sizeof(SweepCounters)) == 0)  // a new object is zero filled
            mem = mmap(nullptr, sizeof(SweepCounters), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        if (fd >= 0) close(fd);
        if (mem != MAP_FAILED) {
            sw->counters = static_cast<SweepCounters *>(mem);
            sw->counters_shared = true;
        }  // else: only the static assignment is available (egx_sweep_set_assignment reports it)
    } else {
        sw->counters = new SweepCounters();
        for (auto &c : sw->counters->next) c.store(0);
    }
    {
        const char *tr = std::getenv("EGX_SWEEP_TRANSPORT");
        if (tr && std::string(tr) == "shm" && world > 1) {
            if (!sw->counters_shared || world > SweepCounters::kMaxWorld) {
                set_error("egx_sweep_create: the host transport needs the shared-memory segment and world <= 16");
                return fail(EGX_ERR_UNSUPPORTED);
            }
            sw->shm_transport = true;
        }
    }
    if (nccl_id && !sw->shm_transport) {
        RcclApi &api = rccl();
        if (!api.err.empty()) {
            set_error(api.err);
            return fail(EGX_ERR_UNSUPPORTED);
        }
        ncclUniqueId id;
        std::memcpy(&id, nccl_id, sizeof id);
        // RCCL >= 2.26 prints a version banner to STDOUT on the first communicator: keep the host's stdout clean (it
        // may carry a protocol, e.g. bench.py's single JSON line) by pointing fd 1 at stderr for the duration of the
        // init.  EGX_RCCL_BANNER=1 leaves it alone.
        const char *keep = std::getenv("EGX_RCCL_BANNER");
        int saved = -1;
        if (!(keep && keep[0] == '1')) {
            fflush(stdout);
            saved = dup(1);
            if (saved >= 0) dup2(2, 1);
        }
        ncclResult_t r = api.CommInitRank(&sw->comm, world, id, rank);
        if (saved >= 0) {
            fflush(stdout);
            dup2(saved, 1);
            close(saved);
        }
        if (r != ncclSuccess) {
            set_error(std::string("ncclCommInitRank: ") + api.GetErrorString(r));
            sw->comm = nullptr;
            return fail(EGX_ERR_HIP);
        }
    }
    *out = sw;
    return EGX_SUCCESS;
}

void egx_sweep_destroy(egx_sweep *sw) {
    if (!sw) return;
    if (sw->gp) hipSetDevice(sw->gp->device);
    if (sw->comm) rccl().CommDestroy(sw->comm);
    if (sw->d_send) hipFree(sw->d_send);
    if (sw->d_recv) hipFree(sw->d_recv);
    if (sw->h_send) hipHostFree(sw->h_send);
    if (sw->h_recv) hipHostFree(sw->h_recv);
    if (sw->stream) hipStreamDestroy(sw->stream);
    if (sw->counters) {
        if (sw->counters_shared) {
            munmap(sw->counters, sizeof(SweepCounters));
            if (sw->rank == 0) shm_unlink(sw->shm_name.c_str());
        } else {
            delete sw->counters;
        }
    }
    if (sw->gp) egx_gp_destroy(sw->gp);
    delete sw;
}

egx_gp *egx_sweep_handle(egx_sweep *sw) { return sw ? sw->gp : nullptr; }

int32_t egx_sweep_info(const egx_sweep *sw, int32_t *rank, int32_t *world, int32_t *rccl_ranks, int32_t *rccl_version,
                       int64_t *n_allgathers) {
    if (!sw) {
        set_error("NULL sweep handle");
        return EGX_ERR_INVALID_VALUE;
    }
    if (rank) *rank = sw->rank;
    if (world) *world = sw->world;
    if (rccl_ranks) *rccl_ranks = sw->comm ? sw->world : 0;
    if (rccl_version) {
        int v = 0;
        if (sw->comm && rccl().GetVersion) rccl().GetVersion(&v);
        *rccl_version = v;
    }
    if (n_allgathers) *n_allgathers = sw->n_allgathers;
    return EGX_SUCCESS;
}

int32_t egx_sweep_set_assignment(egx_sweep *sw, int32_t dynamic) {
    if (!sw) {
        set_error("NULL sweep handle");
        return EGX_ERR_INVALID_VALUE;
    }
    std::lock_guard<std::mutex> lock(sw->mu);
    if (dynamic && !sw->counters) {
        set_error("egx_sweep_set_assignment: no shared-memory counters on this node (shm_open failed): static only");
        return EGX_ERR_UNSUPPORTED;
    }
    sw->dynamic = dynamic ? 1 : 0;
    return EGX_SUCCESS;
}

int32_t egx_sweep_last_balance(const egx_sweep *sw, int64_t *per_rank, double *local_eval_s) {
    if (!sw) {
        set_error("NULL sweep handle");
        return EGX_ERR_INVALID_VALUE;
    }
    if (per_rank)
        for (int r = 0; r < sw->world; r++) per_rank[r] = r < (int)sw->last_per_rank.size() ? sw->last_per_rank[r] : 0;
    if (local_eval_s) *local_eval_s = sw->last_eval_s;
    return EGX_SUCCESS;
}

int32_t egx_sweep_likelihood(egx_sweep *sw, const double *thetas, int64_t k, int64_t theta_len, double *lkh,
                             int32_t *status) {
    if (!sw || k < 0 || (k > 0 && (!thetas || !lkh || !status)) || theta_len < 1) {
        set_error("egx_sweep_likelihood: NULL argument");
        return EGX_ERR_INVALID_VALUE;
    }
    std::lock_guard<std::mutex> lock(sw->mu);
    if (k == 0) return EGX_SUCCESS;
    const int world = sw->world, rank = sw->rank;
    // ---- local part: every failure from here on is CARRIED to the collective, never returned before it
    int local_rc = EGX_SUCCESS;
    std::string local_msg;
    std::vector<double> lk(k);
    std::vector<int32_t> st(k);
    std::vector<char> mine(k, 0);
    const int64_t seq = sw->call_seq++;
    const auto t0 = std::chrono::steady_clock::now();
    {
        std::atomic<int64_t> *ctr = nullptr;
        // The slot of call `seq` was zeroed by rank 0 half a ring earlier, when every rank had long left the slot's previous
        // use (the all-gather of each call is a barrier).  The sequence number advances on EVERY call, so the look-ahead slot
        // is recycled on every call too -- static ones included: a slot dirtied by a dynamic call must be clean again 64
        // calls later whatever mode the call half a ring in between was in.
        if (sw->counters && rank == 0)
            sw->counters->next[(seq + SweepCounters::kSlots / 2) % SweepCounters::kSlots].store(0);
        if (sw->dynamic && sw->counters) ctr = &sw->counters->next[seq % SweepCounters::kSlots];
        StaticSource ssrc(k, rank, world);
        DynamicSource dsrc(ctr, k);
        egx::CandidateSource *src = ctr ? static_cast<egx::CandidateSource *>(&dsrc) : &ssrc;
        local_rc = set_device(sw->gp);
        if (!local_rc) {
            std::unique_lock<std::shared_mutex> glock(sw->gp->mu);
            local_rc = likelihood_batch_core(sw->gp, thetas, k, theta_len, lk.data(), st.data(), src, mine.data());
        }
        if (local_rc) local_msg = last_error_string();
    }
    sw->last_eval_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    std::vector<double> send;
    if (local_rc) {
        send = sweep_poison(k, local_rc);
    } else {
        send = sweep_payload(k);
        for (int64_t c = 0; c < k; c++)
            if (mine[c]) sweep_put(send, c, lk[c], st[c]);
    }
    // ---- the collective
    std::vector<double> recv((size_t)k * 2 * world);
    const int coll_rc = sweep_allgather_doubles(sw, send.data(), k * 2, recv.data());
    if (coll_rc) {
        if (local_rc) set_error(local_msg + " (and the collective failed: " + last_error_string() + ")");
        return local_rc ? local_rc : coll_rc;
    }
    const SweepVerdict v = sweep_unpack(recv.data(), k, world, lkh, status, EGX_STATUS_RANK_FAILED);
    sw->last_per_rank = v.per_rank;
    if (local_rc) {
        set_error(local_msg);
        return local_rc;
    }
    if (v.failed_rank >= 0) {
        set_error("egx_sweep_likelihood: rank " + std::to_string(v.failed_rank) + " failed with egx_rc " +
                  std::to_string(v.failed_rc) + "; " + std::to_string(v.missing) + " of " + std::to_string(k) +
                  " candidates have no result (status EGX_STATUS_RANK_FAILED)");
        return EGX_ERR_PEER;
    }
    if (v.missing || v.duplicate) {
        set_error("egx_sweep_likelihood: assignment inconsistent across ranks (" + std::to_string(v.missing) + " missing, " +
                  std::to_string(v.duplicate) + " duplicated): did every rank select the same assignment mode?");
        return EGX_ERR_PEER;
    }
    return EGX_SUCCESS;
}

int32_t egx_sweep_allgather(egx_sweep *sw, const double *send, int64_t count, double *recv) {
    if (!sw || count < 0 || (count > 0 && (!send || !recv))) {
        set_error("egx_sweep_allgather: NULL argument");
        return EGX_ERR_INVALID_VALUE;
    }
    std::lock_guard<std::mutex> lock(sw->mu);
    if (count == 0) return EGX_SUCCESS;
    // (a failing set_device is carried into the collective as NaNs: every rank still arrives)
    const int dev_rc = set_device(sw->gp);
    const std::string dev_msg = dev_rc ? last_error_string() : std::string();
    std::vector<double> poisoned;
    if (dev_rc) poisoned.assign((size_t)count, std::numeric_limits<double>::quiet_NaN());
    const int rc = sweep_allgather_doubles(sw, dev_rc ? poisoned.data() : send, count, recv);
    if (dev_rc) {
        set_error(dev_msg);
        return dev_rc;
    }
    return rc;
}

// ---- mixture-of-experts recombination (BASELINE config 5: one expert per GPU) -----------------------
// GpMixture::predict_smooth / predict_var_smooth (crates/moe/src/algorithm.rs:411-423, 670-685): val = sum_e p_e y_e,
// var = sum_e p_e^2 v_e over ALL points; predict_hard / predict_var_hard (:879-935): every point is answered by the expert
// of its cluster, argmax_e p_e.  The reference calls the expert once per ROW in hard mode (a full n^2 triangular solve per
// point for the variance); here the points are routed once and every expert gets ONE batched call on its subset.
// Multi-rank: every rank owns some of the experts, forms the partial sums of its own and ONE all-gather of the partial
// (val, var) vectors + a sum in rank order (the same bits on every rank) replaces the reference's fold over experts.
int32_t egx_moe_predict_valvar(egx_sweep *sw, egx_gp *const *experts, const int32_t *expert_ids, int64_t n_local,
                               int64_t n_experts, const double *probas, const double *xq, int64_t m, int64_t d,
                               int32_t smooth, double *val, double *var) {
    if (n_local < 0 || n_experts < 1 || m < 0 || d < 1 || (m > 0 && (!probas || !xq)) || (n_local > 0 && (!experts || !expert_ids)) ||
        (!val && !var)) {
        set_error("egx_moe_predict_valvar: bad arguments");
        return EGX_ERR_INVALID_VALUE;
    }
    if (!sw && n_local != n_experts) {
        set_error("egx_moe_predict_valvar: without a sweep handle (single process) every expert must be local");
        return EGX_ERR_INVALID_VALUE;
    }
    if (m == 0) return EGX_SUCCESS;
    // ---- local part: failures are carried into the collective (a rank that left early would hang the others)
    int local_rc = EGX_SUCCESS;
    std::string local_msg;
    std::vector<double> part((size_t)2 * m + 1, 0.0);  // [status | val (m) | var (m)]
    double *pv = part.data() + 1, *pw = pv + m;
    for (int64_t e = 0; e < n_local && !local_rc; e++)
        if (!experts[e] || expert_ids[e] < 0 || expert_ids[e] >= n_experts) {
            local_msg = "egx_moe_predict_valvar: NULL expert handle or expert id out of range";
            local_rc = EGX_ERR_INVALID_VALUE;
        }
    std::vector<int32_t> cluster;
    if (!smooth && !local_rc) {  // gmx.predict: first maximum of the responsibilities (ndarray / numpy argmax)
        cluster.resize(m);
        for (int64_t a = 0; a < m; a++) {
            const double *pa = probas + a * n_experts;
            int32_t best = 0;
            for (int32_t j = 1; j < n_experts; j++)
                if (pa[j] > pa[best]) best = j;
            cluster[a] = best;
        }
    }
    if (!local_rc) {
        // experts of one rank: up to two in flight (each handle has its own streams: one's launch gaps hide in the other)
        std::atomic<int64_t> next{0};
        std::mutex acc_mu;
        auto worker = [&]() {
            std::vector<double> xs, ys, vs;
            std::vector<int64_t> idx;
            for (;;) {
                const int64_t e = next.fetch_add(1);
                if (e >= n_local) return;
                {
                    std::lock_guard<std::mutex> l(acc_mu);
                    if (local_rc) return;
                }
                const int32_t g = expert_ids[e];
                const double *xin = xq;
                int64_t me = m;
                if (!smooth) {
                    idx.clear();
                    for (int64_t a = 0; a < m; a++)
                        if (cluster[a] == g) idx.push_back(a);
                    me = (int64_t)idx.size();
                    if (me == 0) continue;
                    xs.resize((size_t)me * d);
                    for (int64_t i = 0; i < me; i++) std::memcpy(&xs[(size_t)i * d], xq + idx[i] * d, sizeof(double) * d);
                    xin = xs.data();
                }
                ys.resize(me);
                vs.resize(me);
                int rc;
                if (val && var) rc = egx_gp_predict_valvar(experts[e], xin, me, ys.data(), vs.data());
                else if (val) rc = egx_gp_predict(experts[e], xin, me, ys.data());
                else rc = egx_gp_predict_var(experts[e], xin, me, vs.data());
                std::lock_guard<std::mutex> l(acc_mu);
                if (rc) {
                    if (!local_rc) {
                        local_rc = rc;
                        local_msg = last_error_string();
                    }
                    return;
                }
                if (smooth) {
                    for (int64_t a = 0; a < m; a++) {
                        const double p = probas[a * n_experts + g];
                        if (val) pv[a] += p * ys[a];
                        if (var) pw[a] += p * p * vs[a];
                    }
                } else {
                    for (int64_t i = 0; i < me; i++) {
                        if (val) pv[idx[i]] = ys[i];
                        if (var) pw[idx[i]] = vs[i];
                    }
                }
            }
        };
        // (the accumulation of the smooth sums is ordered by completion, not by expert: addition of <= n_local terms per
        //  point in a run-dependent order would not be reproducible to the bit -- so with two workers each owns its own
        //  partial vectors and they are added in worker order below)
        if (n_local > 1 && smooth) {
            std::vector<double> part2((size_t)2 * m, 0.0);
            double *pv0 = pv, *pw0 = pw;
            // worker A takes the even local experts into part, worker B the odd ones into part2
            auto fixed_worker = [&](int64_t first, double *tv, double *tw) {
                std::vector<double> ys(m), vs(m);
                for (int64_t e = first; e < n_local; e += 2) {
                    {
                        std::lock_guard<std::mutex> l(acc_mu);
                        if (local_rc) return;
                    }
                    int rc;
                    if (val && var) rc = egx_gp_predict_valvar(experts[e], xq, m, ys.data(), vs.data());
                    else if (val) rc = egx_gp_predict(experts[e], xq, m, ys.data());
                    else rc = egx_gp_predict_var(experts[e], xq, m, vs.data());
                    if (rc) {
                        std::lock_guard<std::mutex> l(acc_mu);
                        if (!local_rc) {
                            local_rc = rc;
                            local_msg = last_error_string();
                        }
                        return;
                    }
                    const int32_t g = expert_ids[e];
                    for (int64_t a = 0; a < m; a++) {
                        const double p = probas[a * n_experts + g];
                        if (val) tv[a] += p * ys[a];
                        if (var) tw[a] += p * p * vs[a];
                    }
                }
            };
            std::thread tb(fixed_worker, (int64_t)1, part2.data(), part2.data() + m);
            fixed_worker(0, pv0, pw0);
            tb.join();
            for (int64_t a = 0; a < m; a++) {
                pv0[a] += part2[a];
                pw0[a] += part2[(size_t)m + a];
            }
        } else if (n_local > 1) {
            std::thread tb(worker);  // hard: disjoint subsets, no sums: completion order does not matter
            worker();
            tb.join();
        } else {
            worker();
        }
    }
    part[0] = local_rc ? -(double)(kSweepPoison + local_rc) : 0.0;
    // ---- the collective (multi-rank) and the sum over ranks, in rank order
    const int world = sw ? sw->world : 1;
    std::vector<double> all;
    const double *src = part.data();
    if (sw) {
        std::lock_guard<std::mutex> lock(sw->mu);
        all.resize(part.size() * (size_t)world);
        (void)set_device(sw->gp);
        const int coll_rc = sweep_allgather_doubles(sw, part.data(), (int64_t)part.size(), all.data());
        if (coll_rc) {
            if (local_rc) set_error(local_msg + " (and the collective failed: " + last_error_string() + ")");
            return local_rc ? local_rc : coll_rc;
        }
        src = all.data();
    }
    if (local_rc) {
        set_error(local_msg);
        return local_rc;
    }
    for (int r = 0; r < world; r++)
        if (src[(size_t)r * part.size()] != 0.0) {
            set_error("egx_moe_predict_valvar: rank " + std::to_string(r) + " failed with egx_rc " +
                      std::to_string((int)(-src[(size_t)r * part.size()]) - kSweepPoison));
            return EGX_ERR_PEER;
        }
    for (int64_t a = 0; a < m; a++) {
        double sv = 0.0, sw2 = 0.0;
        for (int r = 0; r < world; r++) {
            const double *pr = src + (size_t)r * part.size() + 1;
            sv += pr[a];
            sw2 += pr[m + a];
        }
        if (val) val[a] = sv;
        if (var) var[a] = sw2;
    }
    return EGX_SUCCESS;
}

// ---- tuned fit over the ranks of a sweep (SURVEY 8e x 8f rank 2) --------------------------------------------------
// GpValidParams::fit with ThetaTuning::Full (crates/gp/src/algorithm.rs:873-960): the reference runs its n_start + 1 COBYLA
// runs as rayon tasks on one host (:928-945).  Here start s belongs to rank s mod world; a rank advances the machines of
// its starts in lock-step on its own GPU (fit_run_starts), ONE all-gather carries every start's (objective, evaluations,
// minimiser) to every rank, all ranks reduce in start order (first minimum wins, :942-945) and factor the winner on their
// replica.  An evaluation returns the same bits wherever and in whichever batch it runs, so every start walks the trajectory
// it walks on one GPU and the fitted model is bit for bit the one egx_gp_fit returns.
int32_t egx_sweep_fit(egx_sweep *sw, const double *theta0s, int64_t n_starts, const double *lo, const double *hi,
                      int64_t bounds_len, int64_t max_eval, int64_t *n_evals_out) {
    if (!sw || !theta0s || !lo || !hi || n_starts < 1) {
        set_error("egx_sweep_fit: NULL argument / no start point");
        return EGX_ERR_INVALID_VALUE;
    }
    egx_gp *gp = sw->gp;
    const int h = gp->h, world = sw->world;
    std::vector<int> active(h);
    for (int i = 0; i < h; i++) active[i] = i;
    std::vector<StartResult> results;
    // one critical section on the replica from the first evaluation to the finalized model (the all-gather included: it has
    // its own deadline), as egx_gp_fit is on a plain handle.  Lock order as in egx_sweep_likelihood: the sweep, then its model.
    std::lock_guard<std::mutex> lock(sw->mu);
    std::unique_lock<std::shared_mutex> glock(gp->mu);
    int local_rc = fit_run_starts(gp, theta0s, active, theta0s, n_starts, lo, hi, bounds_len, max_eval, sw->rank, world, results);
    const std::string local_msg = local_rc ? last_error_string() : std::string();
    // payload: [status | per start: objective, evaluations, minimiser (h)], NaN objective for the starts of other ranks
    const size_t per = (size_t)h + 2;
    std::vector<double> part(1 + (size_t)n_starts * per, std::numeric_limits<double>::quiet_NaN());
    part[0] = local_rc ? -(double)(kSweepPoison + local_rc) : 0.0;
    if (!local_rc)
        for (int64_t s = sw->rank; s < n_starts; s += world) {
            double *q = part.data() + 1 + (size_t)s * per;
            q[0] = results[(size_t)s].f;
            q[1] = (double)results[(size_t)s].evals;
            for (int i = 0; i < h; i++) q[2 + i] = results[(size_t)s].x[(size_t)i];
        }
    std::vector<double> all(part.size() * (size_t)world);
    {
        (void)set_device(gp);
        const int coll_rc = sweep_allgather_doubles(sw, part.data(), (int64_t)part.size(), all.data());
        if (coll_rc) {
            if (local_rc) set_error(local_msg + " (and the collective failed: " + last_error_string() + ")");
            return local_rc ? local_rc : coll_rc;
        }
    }
    if (local_rc) {
        set_error(local_msg);
        return local_rc;
    }
    for (int r = 0; r < world; r++)
        if (all[(size_t)r * part.size()] != 0.0) {
            set_error("egx_sweep_fit: rank " + std::to_string(r) + " failed with egx_rc " +
                      std::to_string((int)(-all[(size_t)r * part.size()]) - kSweepPoison));
            return EGX_ERR_PEER;
        }
    results.assign((size_t)n_starts, StartResult{std::numeric_limits<double>::infinity(), std::vector<double>(h, 0.0), 0});
    for (int64_t s = 0; s < n_starts; s++) {
        const double *q = all.data() + (size_t)(s % world) * part.size() + 1 + (size_t)s * per;
        results[(size_t)s].f = q[0];
        results[(size_t)s].evals = (int64_t)q[1];
        results[(size_t)s].x.assign(q + 2, q + 2 + h);
    }
    return fit_reduce_finalize(gp, theta0s, active, theta0s, results, n_evals_out);
}

// ---- Gaussian mixture responsibilities ----------------------------------------------------------
// precisions_chol[c] = (chol(cov_c)^-1)^T, crates/moe/src/gaussian_mixture.rs:182-205: d x d host arithmetic.
int32_t egx_gmx_precisions_chol(const double *covariances, int64_t k, int64_t d, double *precisions_chol) {
    if (!covariances || !precisions_chol || k < 1 || d < 1) {
        set_error("egx_gmx_precisions_chol: bad arguments");
        return EGX_ERR_INVALID_VALUE;
    }
    std::vector<double> L((size_t)d * d), Li((size_t)d * d);
    for (int64_t c = 0; c < k; c++) {
        const double *A = covariances + (size_t)c * d * d;
        std::fill(L.begin(), L.end(), 0.0);
        for (int64_t j = 0; j < d; j++) {  // lower Cholesky factor, column by column
            double dj = A[j * d + j];
            for (int64_t l = 0; l < j; l++) dj -= L[j * d + l] * L[j * d + l];
            if (!(dj > 0.0) || !std::isfinite(dj)) {
                set_error("egx_gmx_precisions_chol: covariance " + std::to_string((long long)c) + " is not positive definite");
                return EGX_ERR_LINALG;
            }
            L[j * d + j] = std::sqrt(dj);
            for (int64_t i = j + 1; i < d; i++) {
                double v = A[i * d + j];
                for (int64_t l = 0; l < j; l++) v -= L[i * d + l] * L[j * d + l];
                L[i * d + j] = v / L[j * d + j];
            }
        }
        std::fill(Li.begin(), Li.end(), 0.0);  // L^-1 by forward substitution on the identity
        for (int64_t col = 0; col < d; col++)
            for (int64_t i = col; i < d; i++) {
                double v = (i == col) ? 1.0 : 0.0;
                for (int64_t l = col; l < i; l++) v -= L[i * d + l] * Li[l * d + col];
                Li[i * d + col] = v / L[i * d + i];
            }
        double *out = precisions_chol + (size_t)c * d * d;
        for (int64_t i = 0; i < d; i++)
            for (int64_t j = 0; j < d; j++) out[i * d + j] = Li[j * d + i];  // transposed: upper triangular
    }
    return EGX_SUCCESS;
}

int32_t egx_gmx_predict_probas(int32_t device, const double *weights, const double *means, const double *precisions_chol,
                               int64_t k, int64_t d, double heaviside_factor, const double *xq, int64_t m, double *probas) {
    if (!weights || !means || !precisions_chol || k < 1 || d < 1 || d > 4096 || m < 0 || (m > 0 && (!xq || !probas)) ||
        !(heaviside_factor > 0.0)) {
        set_error("egx_gmx_predict_probas: bad arguments");
        return EGX_ERR_INVALID_VALUE;
    }
    if (m == 0) return EGX_SUCCESS;
    if (k == 1) {  // gaussian_mixture.rs:115-116
        for (int64_t a = 0; a < m; a++) probas[a] = 1.0;
        return EGX_SUCCESS;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) {
        (void)hipGetLastError();
        set_error("egx_gmx_predict_probas: no HIP device");
        return EGX_ERR_NO_DEVICE;
    }
    if (device < 0 && hipGetDevice(&device) != hipSuccess) device = 0;  // the calling thread's current device
    if (device >= ndev) {
        set_error("egx_gmx_predict_probas: device out of range");
        return EGX_ERR_INVALID_VALUE;
    }
    EGX_HIP_CHECK(hipSetDevice(device));
    // scaled factors and the per-cluster constant (:105-110, 253-283): precs = P * hf^-0.5, log det = sum log diag(precs)
    const double factor = std::pow(heaviside_factor, -0.5);
    std::vector<double> precs((size_t)k * d * d), par(k);
    const double cst = (double)d * std::log(2.0 * M_PI);
    for (int64_t c = 0; c < k; c++) {
        double ld = 0.0;
        for (int64_t i = 0; i < d * d; i++) precs[(size_t)c * d * d + i] = precisions_chol[(size_t)c * d * d + i] * factor;
        for (int64_t i = 0; i < d; i++) ld += std::log(precs[(size_t)c * d * d + i * d + i]);
        par[c] = (-0.5 * cst + ld) + std::log(weights[c]);
    }
    egx::DevBuf d_x, d_mu, d_p, d_par, d_out;
    EGX_RC(d_x.alloc((size_t)m * d));
    EGX_RC(d_mu.alloc((size_t)k * d));
    EGX_RC(d_p.alloc(precs.size()));
    EGX_RC(d_par.alloc(k));
    EGX_RC(d_out.alloc((size_t)m * k));
    EGX_HIP_CHECK(hipMemcpy(d_x.p, xq, sizeof(double) * (size_t)m * d, hipMemcpyHostToDevice));
    EGX_HIP_CHECK(hipMemcpy(d_mu.p, means, sizeof(double) * (size_t)k * d, hipMemcpyHostToDevice));
    EGX_HIP_CHECK(hipMemcpy(d_p.p, precs.data(), sizeof(double) * precs.size(), hipMemcpyHostToDevice));
    EGX_HIP_CHECK(hipMemcpy(d_par.p, par.data(), sizeof(double) * k, hipMemcpyHostToDevice));
    const size_t lds = sizeof(double) * 64 * (size_t)(d | 1);
    if (lds > 160 * 1024) {
        set_error("egx_gmx_predict_probas: d too large for one workgroup's LDS");
        return EGX_ERR_INVALID_VALUE;
    }
    hipLaunchKernelGGL(egx::k_gmx_probas, dim3((unsigned)((m + 63) / 64)), dim3(64), lds, 0, d_x.p, m, (int)d, (int)k, d_mu.p,
                       d_p.p, d_par.p, d_out.p);
    EGX_HIP_CHECK(hipGetLastError());
    EGX_HIP_CHECK(hipMemcpy(probas, d_out.p, sizeof(double) * (size_t)m * k, hipMemcpyDeviceToHost));
    return EGX_SUCCESS;
}

int32_t egx_gmx_predict_probas_derivatives(int32_t device, const double *weights, const double *means,
                                           const double *precisions_chol, int64_t k, int64_t d, double heaviside_factor,
                                           const double *xq, int64_t m, double *dprobas) {
    if (!weights || !means || !precisions_chol || k < 1 || d < 1 || m < 0 || (m > 0 && (!xq || !dprobas)) ||
        !(heaviside_factor > 0.0)) {
        set_error("egx_gmx_predict_probas_derivatives: bad arguments");
        return EGX_ERR_INVALID_VALUE;
    }
    if (m == 0) return EGX_SUCCESS;
    const size_t lds = sizeof(double) * 64 * (size_t)(3 * (d | 1) + (k | 1));
    if (lds > 160 * 1024) {
        set_error("egx_gmx_predict_probas_derivatives: d / k too large for one workgroup's LDS (3 d + k <= 320)");
        return EGX_ERR_INVALID_VALUE;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) {
        (void)hipGetLastError();
        set_error("egx_gmx_predict_probas_derivatives: no HIP device");
        return EGX_ERR_NO_DEVICE;
    }
    if (device < 0 && hipGetDevice(&device) != hipSuccess) device = 0;
    if (device >= ndev) {
        set_error("egx_gmx_predict_probas_derivatives: device out of range");
        return EGX_ERR_INVALID_VALUE;
    }
    EGX_HIP_CHECK(hipSetDevice(device));
    const double factor = std::pow(heaviside_factor, -0.5);
    std::vector<double> precs((size_t)k * d * d), par(k);
    const double cst = (double)d * std::log(2.0 * M_PI);
    for (int64_t c = 0; c < k; c++) {
        double ld = 0.0;
        for (int64_t i = 0; i < d * d; i++) precs[(size_t)c * d * d + i] = precisions_chol[(size_t)c * d * d + i] * factor;
        for (int64_t i = 0; i < d; i++) ld += std::log(precs[(size_t)c * d * d + i * d + i]);
        par[c] = (-0.5 * cst + ld) + std::log(weights[c]);
    }
    egx::DevBuf d_x, d_mu, d_p, d_par, d_out;
    EGX_RC(d_x.alloc((size_t)m * d));
    EGX_RC(d_mu.alloc((size_t)k * d));
    EGX_RC(d_p.alloc(precs.size()));
    EGX_RC(d_par.alloc(k));
    EGX_RC(d_out.alloc((size_t)m * k * d));
    EGX_HIP_CHECK(hipMemcpy(d_x.p, xq, sizeof(double) * (size_t)m * d, hipMemcpyHostToDevice));
    EGX_HIP_CHECK(hipMemcpy(d_mu.p, means, sizeof(double) * (size_t)k * d, hipMemcpyHostToDevice));
    EGX_HIP_CHECK(hipMemcpy(d_p.p, precs.data(), sizeof(double) * precs.size(), hipMemcpyHostToDevice));
    EGX_HIP_CHECK(hipMemcpy(d_par.p, par.data(), sizeof(double) * k, hipMemcpyHostToDevice));
    if (lds > 64 * 1024)
        EGX_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&egx::k_gmx_probas_deriv),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(egx::k_gmx_probas_deriv, dim3((unsigned)((m + 63) / 64)), dim3(64), lds, 0, d_x.p, m, (int)d, (int)k,
                       d_mu.p, d_p.p, d_par.p, d_out.p);
    EGX_HIP_CHECK(hipGetLastError());
    EGX_HIP_CHECK(hipMemcpy(dprobas, d_out.p, sizeof(double) * (size_t)m * k * d, hipMemcpyDeviceToHost));
    return EGX_SUCCESS;
}

// GpMixture::predict_gradients_smooth / predict_var_gradients_smooth (crates/moe/src/algorithm.rs:691-783):
//     d val / dx = sum_e p_e grad y_e + p'_e y_e ,   d var / dx = sum_e p_e^2 grad v_e + 2 p_e p'_e v_e
// and predict_gradients_hard / predict_var_gradients_hard (:942-1010): the gradient of the expert of argmax_e p_e.  The
// reference calls every expert once per ROW; here every expert gets ONE batched call per quantity on its points (all of
// them in smooth mode, its cluster's in hard mode).  Sharded exactly like egx_moe_predict_valvar: a rank's partial
// (m x d) sums over ITS experts -- two experts in flight, even / odd local experts into separate partial sums that are
// added in a fixed order -- one all-gather, the sum over ranks in rank order (the same bits on every rank).
int32_t egx_moe_predict_valvar_gradients(egx_sweep *sw, egx_gp *const *experts, const int32_t *expert_ids, int64_t n_local,
                                         int64_t n_experts, const double *probas, const double *dprobas, const double *xq,
                                         int64_t m, int64_t d, int32_t smooth, double *grad_val, double *grad_var) {
    if (n_local < 0 || n_experts < 1 || m < 0 || d < 1 || (m > 0 && (!probas || !xq)) || (n_local > 0 && (!experts || !expert_ids)) ||
        (!grad_val && !grad_var) || (smooth && n_experts > 1 && m > 0 && !dprobas)) {
        set_error("egx_moe_predict_valvar_gradients: bad arguments (the smooth recombination of more than one expert needs dprobas)");
        return EGX_ERR_INVALID_VALUE;
    }
    if (!sw && n_local != n_experts) {
        set_error("egx_moe_predict_valvar_gradients: without a sweep handle (single process) every expert must be local");
        return EGX_ERR_INVALID_VALUE;
    }
    if (m == 0) return EGX_SUCCESS;
    const size_t md = (size_t)m * d;
    int local_rc = EGX_SUCCESS;
    std::string local_msg;
    std::vector<double> part(2 * md + 1, 0.0);  // [status | grad val (m x d) | grad var (m x d)]
    for (int64_t e = 0; e < n_local && !local_rc; e++)
        if (!experts[e] || expert_ids[e] < 0 || expert_ids[e] >= n_experts) {
            local_msg = "egx_moe_predict_valvar_gradients: NULL expert handle or expert id out of range";
            local_rc = EGX_ERR_INVALID_VALUE;
        }
    std::vector<int32_t> cluster;
    if (!smooth && !local_rc) {
        cluster.resize(m);
        for (int64_t a = 0; a < m; a++) {
            const double *pa = probas + a * n_experts;
            int32_t best = 0;
            for (int32_t j = 1; j < n_experts; j++)
                if (pa[j] > pa[best]) best = j;
            cluster[a] = best;
        }
    }
    if (!local_rc) {
        std::mutex acc_mu;
        // worker w takes the local experts w, w + 2, ... into its own partial sums (tv, tw)
        auto worker = [&](int64_t first, double *tv, double *tw) {
            std::vector<double> xs, gy, gv, ys, vs;
            std::vector<int64_t> idx;
            for (int64_t e = first; e < n_local; e += 2) {
                {
                    std::lock_guard<std::mutex> l(acc_mu);
                    if (local_rc) return;
                }
                const int32_t g = expert_ids[e];
                const double *xin = xq;
                int64_t me = m;
                if (!smooth) {
                    idx.clear();
                    for (int64_t a = 0; a < m; a++)
                        if (cluster[a] == g) idx.push_back(a);
                    me = (int64_t)idx.size();
                    if (me == 0) continue;
                    xs.resize((size_t)me * d);
                    for (int64_t i = 0; i < me; i++) std::memcpy(&xs[(size_t)i * d], xq + idx[i] * d, sizeof(double) * d);
                    xin = xs.data();
                }
                int rc = EGX_SUCCESS;
                if (grad_val) gy.resize((size_t)me * d);
                if (grad_var) gv.resize((size_t)me * d);
                if (grad_val && grad_var) rc = egx_gp_predict_valvar_gradients(experts[e], xin, me, gy.data(), gv.data());
                else if (grad_val) rc = egx_gp_predict_gradients(experts[e], xin, me, gy.data());
                else rc = egx_gp_predict_var_gradients(experts[e], xin, me, gv.data());
                const bool need_pp = smooth && n_experts > 1;  // the p' terms need the experts' values too
                if (!rc && need_pp) {
                    if (grad_val) ys.resize(me);
                    if (grad_var) vs.resize(me);
                    if (grad_val && grad_var) rc = egx_gp_predict_valvar(experts[e], xin, me, ys.data(), vs.data());
                    else if (grad_val) rc = egx_gp_predict(experts[e], xin, me, ys.data());
                    else rc = egx_gp_predict_var(experts[e], xin, me, vs.data());
                }
                if (rc) {
                    std::lock_guard<std::mutex> l(acc_mu);
                    if (!local_rc) {
                        local_rc = rc;
                        local_msg = last_error_string();
                    }
                    return;
                }
                if (smooth) {
                    for (int64_t a = 0; a < m; a++) {
                        const double p = probas[a * n_experts + g];
                        const double *pp = need_pp ? dprobas + ((size_t)a * n_experts + g) * d : nullptr;
                        for (int64_t j = 0; j < d; j++) {
                            if (grad_val) tv[a * d + j] += gy[a * d + j] * p + (pp ? pp[j] * ys[a] : 0.0);
                            if (grad_var) tw[a * d + j] += gv[a * d + j] * (p * p) + (pp ? 2.0 * p * pp[j] * vs[a] : 0.0);
                        }
                    }
                } else {
                    for (int64_t i = 0; i < me; i++) {  // disjoint rows: no sums
                        if (grad_val) std::memcpy(tv + idx[i] * d, &gy[(size_t)i * d], sizeof(double) * d);
                        if (grad_var) std::memcpy(tw + idx[i] * d, &gv[(size_t)i * d], sizeof(double) * d);
                    }
                }
            }
        };
        double *pv = part.data() + 1, *pw = pv + md;
        if (n_local > 1) {
            std::vector<double> part2(smooth ? 2 * md : 0, 0.0);
            // smooth: worker B accumulates into its own vectors (a fixed order of additions); hard: rows are disjoint
            double *bv = smooth ? part2.data() : pv, *bw = smooth ? part2.data() + md : pw;
            std::thread tb(worker, (int64_t)1, bv, bw);
            worker(0, pv, pw);
            tb.join();
            if (smooth)
                for (size_t i = 0; i < md; i++) {
                    pv[i] += part2[i];
                    pw[i] += part2[md + i];
                }
        } else {
            worker(0, pv, pw);
        }
    }
    part[0] = local_rc ? -(double)(kSweepPoison + local_rc) : 0.0;
    const int world = sw ? sw->world : 1;
    std::vector<double> all;
    const double *src = part.data();
    if (sw) {
        std::lock_guard<std::mutex> lock(sw->mu);
        all.resize(part.size() * (size_t)world);
        (void)set_device(sw->gp);
        const int coll_rc = sweep_allgather_doubles(sw, part.data(), (int64_t)part.size(), all.data());
        if (coll_rc) {
            if (local_rc) set_error(local_msg + " (and the collective failed: " + last_error_string() + ")");
            return local_rc ? local_rc : coll_rc;
        }
        src = all.data();
    }
    if (local_rc) {
        set_error(local_msg);
        return local_rc;
    }
    for (int r = 0; r < world; r++)
        if (src[(size_t)r * part.size()] != 0.0) {
            set_error("egx_moe_predict_valvar_gradients: rank " + std::to_string(r) + " failed with egx_rc " +
                      std::to_string((int)(-src[(size_t)r * part.size()]) - kSweepPoison));
            return EGX_ERR_PEER;
        }
    for (size_t i = 0; i < md; i++) {
        double sv = 0.0, sw2 = 0.0;
        for (int r = 0; r < world; r++) {
            const double *pr = src + (size_t)r * part.size() + 1;
            sv += pr[i];
            sw2 += pr[md + i];
        }
        if (grad_val) grad_val[i] = sv;
        if (grad_var) grad_var[i] = sw2;
    }
    return EGX_SUCCESS;
}

}  // extern "C"
