// Multi-GPU theta sweep behind the C ABI (include/egx_gp.h, egx_sweep_*): one process per GPU, candidate k -> rank
// k mod world, every rank evaluates its shard on its own device through egx_gp_likelihood_batch, ONE RCCL all-gather
// of {likelihood, status} (16 B per candidate) over xGMI assembles the result on every rank.
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
#include <unistd.h>
#include <cstdio>
#include <rccl/rccl.h>

namespace egx {

struct RcclApi {
    void *lib = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
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

struct egx_sweep {
    egx_gp *gp = nullptr;
    int rank = 0, world = 1;
    ncclComm_t comm = nullptr;
    hipStream_t stream = nullptr;
    double *d_send = nullptr, *d_recv = nullptr;  // per x 2 and world x per x 2 doubles
    double *h_send = nullptr, *h_recv = nullptr;  // pinned
    int64_t cap = 0;                              // entries per rank the buffers hold
    std::mutex mu;
    int64_t n_allgathers = 0;
};

static int sweep_reserve(egx_sweep *sw, int64_t per) {
    if (per <= sw->cap) return EGX_SUCCESS;
    if (sw->d_send) (void)hipFree(sw->d_send);
    if (sw->d_recv) (void)hipFree(sw->d_recv);
    if (sw->h_send) (void)hipHostFree(sw->h_send);
    if (sw->h_recv) (void)hipHostFree(sw->h_recv);
    sw->d_send = sw->d_recv = sw->h_send = sw->h_recv = nullptr;
    sw->cap = 0;
    const size_t one = sizeof(double) * 2 * (size_t)per;
    EGX_HIP_CHECK(hipMalloc(&sw->d_send, one));
    EGX_HIP_CHECK(hipMalloc(&sw->d_recv, one * sw->world));
    EGX_HIP_CHECK(hipHostMalloc(&sw->h_send, one, hipHostMallocDefault));
    EGX_HIP_CHECK(hipHostMalloc(&sw->h_recv, one * sw->world, hipHostMallocDefault));
    sw->cap = per;
    return EGX_SUCCESS;
}

// all ranks: recv (world x count doubles) <- concatenation over ranks of send (count doubles); host buffers, pinned
// staging, one ncclAllGather on the sweep's stream
static int sweep_allgather_doubles(egx_sweep *sw, const double *send, int64_t count, double *recv) {
    if (sw->world == 1 && !sw->comm) {
        std::memcpy(recv, send, sizeof(double) * count);
        return EGX_SUCCESS;
    }
    EGX_RC(sweep_reserve(sw, (count + 1) / 2));
    std::memcpy(sw->h_send, send, sizeof(double) * count);
    EGX_HIP_CHECK(hipMemcpyAsync(sw->d_send, sw->h_send, sizeof(double) * count, hipMemcpyHostToDevice, sw->stream));
    EGX_NCCL_CHECK(rccl().AllGather(sw->d_send, sw->d_recv, (size_t)count, ncclDouble, sw->comm, sw->stream));
    EGX_HIP_CHECK(hipMemcpyAsync(sw->h_recv, sw->d_recv, sizeof(double) * count * sw->world, hipMemcpyDeviceToHost,
                                 sw->stream));
    EGX_HIP_CHECK(hipStreamSynchronize(sw->stream));
    std::memcpy(recv, sw->h_recv, sizeof(double) * count * sw->world);
    sw->n_allgathers++;
    return EGX_SUCCESS;
}

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
    if (nccl_id) {
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

int32_t egx_sweep_likelihood(egx_sweep *sw, const double *thetas, int64_t k, int64_t theta_len, double *lkh,
                             int32_t *status) {
    if (!sw || k < 0 || (k > 0 && (!thetas || !lkh || !status)) || theta_len < 1) {
        set_error("egx_sweep_likelihood: NULL argument");
        return EGX_ERR_INVALID_VALUE;
    }
    std::lock_guard<std::mutex> lock(sw->mu);
    if (k == 0) return EGX_SUCCESS;
    EGX_RC(set_device(sw->gp));
    const int world = sw->world, rank = sw->rank;
    const int64_t per = sweep_slots_per_rank(k, world), mine = sweep_count_of_rank(k, rank, world);
    // this rank's shard: candidates rank, rank + world, ...
    std::vector<double> th((size_t)mine * theta_len), lk(mine);
    std::vector<int32_t> st(mine);
    for (int64_t j = 0; j < mine; j++)
        std::memcpy(&th[(size_t)j * theta_len], thetas + (size_t)sweep_candidate(rank, j, world) * theta_len,
                    sizeof(double) * theta_len);
    if (mine) EGX_RC(egx_gp_likelihood_batch(sw->gp, th.data(), mine, theta_len, lk.data(), st.data()));
    // fixed-size payload per rank: per x {likelihood, status}; unused slots NaN
    std::vector<double> send = sweep_pack(lk.data(), st.data(), mine, per), recv((size_t)per * 2 * world);
    EGX_RC(sweep_allgather_doubles(sw, send.data(), per * 2, recv.data()));
    sweep_unpack(recv.data(), k, world, lkh, status);
    return EGX_SUCCESS;
}

int32_t egx_sweep_allgather(egx_sweep *sw, const double *send, int64_t count, double *recv) {
    if (!sw || count < 0 || (count > 0 && (!send || !recv))) {
        set_error("egx_sweep_allgather: NULL argument");
        return EGX_ERR_INVALID_VALUE;
    }
    std::lock_guard<std::mutex> lock(sw->mu);
    if (count == 0) return EGX_SUCCESS;
    EGX_RC(set_device(sw->gp));
    return sweep_allgather_doubles(sw, send, count, recv);
}

}  // extern "C"
