// C ABI implementation (include/egx_gp.h), core: handle lifetime, one likelihood evaluation (device half + the small
// host-side GLS algebra), the fixed-theta fit, state download / upload, kernel-level entry points.  Mirrors, operation
// for operation, GpValidParams::fit and reduced_likelihood (crates/gp/src/algorithm.rs:785-1056); the O(n^2 d) / O(n^3)
// parts run in the HIP kernels of kernels_corr.hip / kernels_chol.hip.  Predictions: gp_predict.hip; optimiser drivers
// and the theta-gradient: gp_fit.hip; sparse GP: sgp_host.hip.  There is no CPU fallback.
#include "gp_handle.h"

namespace egx {

static thread_local std::string g_last_error;
// EGX_PIPE_RETRY (egx_set_tuning "pipe_retry"): an evaluation whose chain launch ran into its wait bound is enqueued once more by
// separate launches (1, default) or reported as EGX_ERR_HIP at once (0); the counters behind egx_chain_stats
static std::atomic<int> g_pipe_retry{[] { const char *e = std::getenv("EGX_PIPE_RETRY"); return e ? std::atoi(e) : 1; }()};
static std::atomic<int64_t> g_chain_aborts{0}, g_chain_retries{0};
void set_error(const std::string &msg) { g_last_error = msg; }
const std::string &last_error_string() { return g_last_error; }

}  // namespace egx

using namespace egx;

namespace egx {

int set_device(const egx_gp *gp) {
    EGX_HIP_CHECK(hipSetDevice(gp->device));
    return EGX_SUCCESS;
}

// The four streams of a workspace -- evaluation, C^-T rider, look-ahead chain, its side stream -- are the expensive part of a
// FIRST handle of a shape: the HIP runtime takes ~3 ms to create a stream and ~2.7 ms to destroy one, serialised over host
// threads (profiles/r06_first_handle_of_a_shape_costs.txt: 14.5 ms per workspace whatever n, 175-190 ms for the twelve workspaces
// of a tuned fit, against 0.05 ms for 400 events and 0.02 ms for the matrix).  They carry no shape, so idle workspaces -- pooled or
// freed -- hand their streams to a per-device free list as a SET (created together, in the order the runtime deals them to its
// hardware queues: section 4.4 of DESIGN.md) and any workspace that comes to life, of whatever shape, takes a set from there
// before it creates one: a model per cluster / fold / candidate specification with sizes all different (crates/moe/src/algorithm.rs:167-262,
// clustering.rs) pays the runtime once per stream, not once per shape.
struct StreamSet {
    hipStream_t stream = nullptr, inv_stream = nullptr, s2 = nullptr, s3 = nullptr;
};
static std::mutex g_stream_mu;
static std::vector<std::pair<int, StreamSet>> g_stream_sets;  // (device, idle set); the back is taken first
constexpr size_t kMaxIdleStreamSets = 96;
static void destroy_stream_set(StreamSet &ss) {
    if (ss.s2) (void)hipStreamDestroy(ss.s2);
    if (ss.s3) (void)hipStreamDestroy(ss.s3);
    if (ss.inv_stream) (void)hipStreamDestroy(ss.inv_stream);
    if (ss.stream) (void)hipStreamDestroy(ss.stream);
    ss = StreamSet();
}
// the workspace's streams (idle: every caller has synchronised them) go to the free list; a partial set is destroyed
static void give_streams(int device, Workspace &w) {
    StreamSet ss;
    ss.stream = w.stream, ss.inv_stream = w.inv_stream, ss.s2 = w.lk.s2, ss.s3 = w.lk.s3;
    w.stream = w.eval_stream = w.inv_stream = nullptr;
    w.lk.s2 = w.lk.s3 = nullptr;
    if (ss.stream && ss.inv_stream && ss.s2 && ss.s3) {
        std::lock_guard<std::mutex> lock(g_stream_mu);
        if (g_stream_sets.size() < kMaxIdleStreamSets) {
            g_stream_sets.emplace_back(device, ss);
            return;
        }
    }
    destroy_stream_set(ss);
}
// four streams for a workspace of `device`: an idle set, or new ones (evaluation, rider, then the two high-priority ones)
static hipError_t take_streams(int device, Workspace &w) {
    {
        std::lock_guard<std::mutex> lock(g_stream_mu);
        for (size_t i = g_stream_sets.size(); i-- > 0;)
            if (g_stream_sets[i].first == device) {
                const StreamSet ss = g_stream_sets[i].second;
                g_stream_sets.erase(g_stream_sets.begin() + (long)i);
                w.stream = w.eval_stream = ss.stream, w.inv_stream = ss.inv_stream, w.lk.s2 = ss.s2, w.lk.s3 = ss.s3;
                return hipSuccess;
            }
    }
    hipError_t e = hipStreamCreateWithFlags(&w.stream, hipStreamNonBlocking);
    w.eval_stream = w.stream;
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&w.inv_stream, hipStreamNonBlocking);
    int lo = 0, hi = 0;
    if (e == hipSuccess) e = hipDeviceGetStreamPriorityRange(&lo, &hi);
    if (e == hipSuccess) e = hipStreamCreateWithPriority(&w.lk.s2, hipStreamNonBlocking, hi);
    if (e == hipSuccess) e = hipStreamCreateWithPriority(&w.lk.s3, hipStreamNonBlocking, hi);
    return e;
}
// egx_trim: the idle sets of `device` (-1: of every device) are destroyed
static void destroy_idle_streams(int device) {
    std::vector<std::pair<int, StreamSet>> out;
    {
        std::lock_guard<std::mutex> lock(g_stream_mu);
        for (size_t i = g_stream_sets.size(); i-- > 0;)
            if (device < 0 || g_stream_sets[i].first == device) {
                out.push_back(g_stream_sets[i]);
                g_stream_sets.erase(g_stream_sets.begin() + (long)i);
            }
    }
    int cur = 0;
    const bool have_cur = hipGetDevice(&cur) == hipSuccess;
    for (auto &d : out) {
        (void)hipSetDevice(d.first);
        destroy_stream_set(d.second);
    }
    if (have_cur) (void)hipSetDevice(cur);
}

static void free_workspace(Workspace &w, int device) {
    // (M, dinv, d_info are views into the handle's slabs)
    if (w.dW) hipFree(w.dW);
    if (w.d_coef) hipFree(w.d_coef);
    if (w.d_xs) hipFree(w.d_xs);
    if (w.d_diag) hipFree(w.d_diag);
    if (w.d_vec) hipFree(w.d_vec);
    if (w.d_rhs) hipFree(w.d_rhs);
    for (double *q : {w.d_gneg, w.d_gram, w.d_gdinv, w.d_gramP, w.d_beta, w.d_part})
        if (q) hipFree(q);
    if (w.d_ginfo) hipFree(w.d_ginfo);
    for (double *q : {w.h_gram, w.h_part, w.h_beta})
        if (q) hipHostFree(q);
    if (w.h_ginfo) hipHostFree(w.h_ginfo);
    if (w.h_coef) hipHostFree(w.h_coef);
    if (w.h_rows) hipHostFree(w.h_rows);
    if (w.h_diag) hipHostFree(w.h_diag);
    if (w.h_vec) hipHostFree(w.h_vec);
    if (w.h_info) hipHostFree(w.h_info);
    if (w.d_gpart) hipFree(w.d_gpart);
    if (w.d_gout) hipFree(w.d_gout);
    if (w.h_gout) hipHostFree(w.h_gout);
    for (auto &e : w.ev)
        if (e) hipEventDestroy(e);
    if (w.trace.ready)
        for (int i = 0; i < GemmTrace::kMax; i++) {
            hipEventDestroy(w.trace.e0[i]);
            hipEventDestroy(w.trace.e1[i]);
        }
    for (hipEvent_t e : {w.lk.ev_lu, w.lk.ev_lur, w.lk.ev_panel, w.lk.ev_a, w.lk.ev_b})
        if (e) hipEventDestroy(e);
    if (w.ev_inv_grp) hipEventDestroy(w.ev_inv_grp);
    if (w.ev_inv_done) hipEventDestroy(w.ev_inv_done);
    give_streams(device, w);  // (idle: whoever frees a workspace has synchronised its streams)
    w = Workspace();
}

static int alloc_workspace(egx_gp *gp, Workspace &w, int index) {
    w.M = gp->slab_M + (int64_t)index * gp->stride_M;
    w.dinv = gp->slab_D + (int64_t)index * gp->stride_D;
    w.d_info = gp->slab_I + index;
    // evaluation stream, the C^-T rider's stream, the look-ahead chain's stream and the side stream of the split updates
    // (launch_potrf; the last two high priority): an idle set of the device, or new ones
    EGX_HIP_CHECK(take_streams(gp->device, w));
    EGX_HIP_CHECK(hipEventCreateWithFlags(&w.ev_inv_grp, hipEventDisableTiming));
    EGX_HIP_CHECK(hipEventCreateWithFlags(&w.ev_inv_done, hipEventDisableTiming));
    for (hipEvent_t *e : {&w.lk.ev_lu, &w.lk.ev_lur, &w.lk.ev_panel, &w.lk.ev_a, &w.lk.ev_b})
        EGX_HIP_CHECK(hipEventCreateWithFlags(e, hipEventDisableTiming));
    const int hmax = gp->has_w ? gp->h : 1;
    EGX_HIP_CHECK(dev_malloc(&w.d_coef, sizeof(double) * (size_t)gp->d * hmax));
    EGX_HIP_CHECK(dev_malloc(&w.d_xs, sizeof(double) * (size_t)gp->d * gp->n_pad));
    EGX_HIP_CHECK(dev_malloc(&w.d_diag, sizeof(double) * (size_t)gp->n_pad));
    EGX_HIP_CHECK(dev_malloc(&w.d_vec, sizeof(double) * (size_t)gp->n_pad));
    EGX_HIP_CHECK(dev_malloc(&w.d_rhs, sizeof(double) * (size_t)gp->n_pad));
    EGX_HIP_CHECK(hipHostMalloc(&w.h_coef, sizeof(double) * (size_t)gp->d * hmax, hipHostMallocDefault));
    EGX_HIP_CHECK(hipHostMalloc(&w.h_rows, sizeof(double) * (size_t)gp->q * gp->n_pad, hipHostMallocDefault));
    EGX_HIP_CHECK(hipHostMalloc(&w.h_diag, sizeof(double) * (size_t)gp->n_pad, hipHostMallocDefault));
    EGX_HIP_CHECK(hipHostMalloc(&w.h_vec, sizeof(double) * (size_t)gp->n_pad, hipHostMallocDefault));
    EGX_HIP_CHECK(hipHostMalloc(&w.h_info, 16 * sizeof(int), hipHostMallocDefault));
    std::memset(w.h_info, 0, 16 * sizeof(int));
    if (gp->gls_device) {
        const size_t g2 = (size_t)gp->rhs_pad * gp->rhs_pad;
        EGX_HIP_CHECK(dev_malloc(&w.d_gneg, sizeof(double) * g2));
        EGX_HIP_CHECK(dev_malloc(&w.d_gram, sizeof(double) * g2));
        EGX_HIP_CHECK(dev_malloc(&w.d_gdinv, sizeof(double) * dinv_doubles(gp->rhs_pad)));
        EGX_HIP_CHECK(dev_malloc(&w.d_gramP, sizeof(double) * gram_scratch_doubles(gp->rhs_pad, gp->n_pad)));
        EGX_HIP_CHECK(dev_malloc(&w.d_beta, sizeof(double) * (size_t)gp->rhs_pad));
        EGX_HIP_CHECK(dev_malloc(&w.d_part, sizeof(double) * (size_t)((gp->n_pad + 255) / 256)));
        EGX_HIP_CHECK(dev_malloc(&w.d_ginfo, sizeof(int)));
        EGX_HIP_CHECK(hipHostMalloc(&w.h_gram, sizeof(double) * g2, hipHostMallocDefault));
        EGX_HIP_CHECK(hipHostMalloc(&w.h_part, sizeof(double) * (size_t)((gp->n_pad + 255) / 256), hipHostMallocDefault));
        EGX_HIP_CHECK(hipHostMalloc(&w.h_beta, sizeof(double) * (size_t)gp->rhs_pad, hipHostMallocDefault));
        EGX_HIP_CHECK(hipHostMalloc(&w.h_ginfo, sizeof(int), hipHostMallocDefault));
    }
    for (auto &e : w.ev) EGX_HIP_CHECK(hipEventCreate(&e));
    for (int i = 0; i < GemmTrace::kMax; i++) {
        EGX_HIP_CHECK(hipEventCreate(&w.trace.e0[i]));
        EGX_HIP_CHECK(hipEventCreate(&w.trace.e1[i]));
    }
    w.trace.ready = true;
    return EGX_SUCCESS;
}

// ---------------------------------------------------------------------------------------------
// Resource pool.  The reference's `fit` is one-shot per data set (GpValidParams::fit, algorithm.rs:785-794) and its
// callers -- the expert loop of egobox-moe (crates/moe/src/algorithm.rs:167-177), EGO's surrogate refits -- create a new
// model per call.  A new handle used to cost a 2 GiB hipMalloc (n = 16384), ~400 event / 3 stream creations per
// workspace and a first factorisation at half speed on the never-touched allocation (round 2: 68 ms against 27 ms
// resident).  Destroyed handles therefore leave everything device-side they own -- slabs, workspaces with their
// streams / events / pinned buffers, the training-set buffers -- in a per-process pool keyed by the SHAPE of the handle;
// the next egx_gp_create of that shape adopts it and only uploads x and y.  Bounded (EGX_POOL_MAX_GB, default 48; least
// recently returned entries are freed first), emptied by egx_trim().  Nothing pooled is ever read before it is
// rewritten: every evaluation rebuilds R including its identity padding, the right-hand-side rows and the flags.
// ---------------------------------------------------------------------------------------------
struct PoolKey {
    int device = 0, n_pad = 0, d = 0, q = 0, hmax = 1, nws = 0;
    bool gls = false;
    bool member = false;  // a member of a group (egx_gp_create_group): everything but the slabs, which belong to the group
    bool operator==(const PoolKey &o) const {
        return device == o.device && n_pad == o.n_pad && d == o.d && q == o.q && hmax == o.hmax && nws == o.nws && gls == o.gls &&
               member == o.member;
    }
};
struct PoolEntry {
    PoolKey key;
    double *d_xT = nullptr, *d_rhsT = nullptr, *d_gamma = nullptr, *d_fit_coef = nullptr;
    double *slab_M = nullptr, *slab_D = nullptr;
    int *slab_I = nullptr;
    std::vector<Workspace> ws;
    size_t bytes = 0;
};
// the slabs of a destroyed GROUP (GroupSlabs: k matrices, tile inverses, flags + hand-off words of one shape): adopted by the
// next egx_gp_create_group whose three sizes match (round 6: eight n = 8192 experts were 107 ms to create and 100 ms to
// destroy -- a 4.4 GB hipMalloc / hipFree and eight workspaces' streams, ~400 events and pinned buffers each -- for a 33 ms fit)
struct SlabEntry {
    int device = 0;
    double *M = nullptr, *D = nullptr;
    int *I = nullptr;
    size_t bytes_M = 0, bytes_D = 0, bytes_I = 0;
    size_t bytes() const { return bytes_M + bytes_D + bytes_I; }
};
static std::mutex g_pool_mu;
static std::list<PoolEntry> g_pool;  // front = most recently returned
static std::list<SlabEntry> g_slab_pool;  // front = most recently returned
static int64_t g_pool_hits = 0, g_pool_misses = 0;

// bytes pooled on one device (the bound EGX_POOL_MAX_GB is PER DEVICE)
static size_t pool_bytes_on(int device) {
    size_t b = 0;
    for (const auto &e : g_pool)
        if (e.key.device == device) b += e.bytes;
    for (const auto &e : g_slab_pool)
        if (e.device == device) b += e.bytes();
    return b;
}
static void free_slab_entry(SlabEntry &e) {
    (void)hipSetDevice(e.device);
    if (e.M) (void)hipFree(e.M);
    if (e.D) (void)hipFree(e.D);
    if (e.I) (void)hipFree(e.I);
    e = SlabEntry();
}
static size_t pool_cap_bytes() {
    static const size_t cap = [] {
        double gb = 48.0;
        if (const char *e = std::getenv("EGX_POOL_MAX_GB")) gb = std::atof(e);
        return gb <= 0.0 ? (size_t)0 : (size_t)(gb * 1073741824.0);
    }();
    return cap;
}
static void free_entry(PoolEntry &e) {
    (void)hipSetDevice(e.key.device);
    for (auto &w : e.ws) free_workspace(w, e.key.device);
    for (double *q : {e.d_xT, e.d_rhsT, e.d_gamma, e.d_fit_coef, e.slab_M, e.slab_D})
        if (q) (void)hipFree(q);
    if (e.slab_I) (void)hipFree(e.slab_I);
    e = PoolEntry();
}
// everything pooled on `device` (-1: on every device) is freed; returns the bytes
static size_t pool_trim(int device) {
    std::list<PoolEntry> out;
    std::list<SlabEntry> out_slabs;
    size_t bytes = 0;
    {
        std::lock_guard<std::mutex> lock(g_pool_mu);
        for (auto it = g_pool.begin(); it != g_pool.end();) {
            auto next = std::next(it);
            if (device < 0 || it->key.device == device) {
                bytes += it->bytes;
                out.splice(out.end(), g_pool, it);
            }
            it = next;
        }
        for (auto it = g_slab_pool.begin(); it != g_slab_pool.end();) {
            auto next = std::next(it);
            if (device < 0 || it->device == device) {
                bytes += it->bytes();
                out_slabs.splice(out_slabs.end(), g_slab_pool, it);
            }
            it = next;
        }
    }
    int cur = 0;
    const bool have_cur = hipGetDevice(&cur) == hipSuccess;
    for (auto &e : out) free_entry(e);
    for (auto &e : out_slabs) free_slab_entry(e);
    if (have_cur) (void)hipSetDevice(cur);
    return bytes;
}
static PoolKey pool_key(const egx_gp *gp, int nws) {
    PoolKey k;
    k.device = gp->device;
    k.n_pad = gp->n_pad;
    k.d = gp->d;
    k.q = gp->q;
    k.hmax = gp->has_w ? gp->h : 1;
    k.nws = nws;
    k.gls = gp->gls_device;
    k.member = gp->group != nullptr;
    return k;
}
// device bytes a handle of this shape hands to the pool: the slabs, the training-set buffers and every workspace's own
// buffers (the lazily allocated ones -- block inverses, theta-gradient scratch -- counted when present)
static size_t pooled_bytes(const egx_gp *gp, const std::vector<Workspace> &ws) {
    const size_t n_pad = (size_t)gp->n_pad, d = (size_t)gp->d, hmax = gp->has_w ? (size_t)gp->h : 1;
    size_t b = gp->group ? 0  // (a member of a group: the slabs are the group's, pooled on their own)
                         : sizeof(double) * ((size_t)gp->stride_M + (size_t)gp->stride_D) * ws.size() + sizeof(int) * slab_I_ints(gp, (int)ws.size());
    b += sizeof(double) * (2 * d * n_pad + (size_t)gp->q * n_pad + n_pad + d * hmax + 2 * d);
    for (const auto &w : ws) {
        b += sizeof(double) * (d * hmax + d * n_pad + 3 * n_pad);
        if (w.dW) b += sizeof(double) * ((n_pad + kNB - 1) / kNB) * 65536;
        if (w.d_gram) {
            const size_t g = (size_t)gp->rhs_pad;
            b += sizeof(double) * (2 * g * g + dinv_doubles((int)g) + gram_scratch_doubles((int)g, (int)n_pad) + g + (n_pad + 255) / 256);
        }
        if (w.d_gpart) b += sizeof(double) * ((size_t)grad_partial_doubles((int)d) + 64 + d);
    }
    return b;
}
// adopt a pooled set of resources of this shape, if there is one
static bool pool_take(egx_gp *gp, int nws) {
    const PoolKey key = pool_key(gp, nws);
    std::lock_guard<std::mutex> lock(g_pool_mu);
    for (auto it = g_pool.begin(); it != g_pool.end(); ++it)
        if (it->key == key) {
            gp->d_xT = it->d_xT;
            gp->d_rhsT = it->d_rhsT;
            gp->d_gamma = it->d_gamma;
            gp->d_fit_coef = it->d_fit_coef;
            gp->slab_M = it->slab_M;
            gp->slab_D = it->slab_D;
            gp->slab_I = it->slab_I;
            gp->ws = std::move(it->ws);
            g_pool.erase(it);
            g_pool_hits++;
            return true;
        }
    g_pool_misses++;
    return false;
}
// hand a dying handle's resources to the pool (or free them when the pool is disabled / they alone exceed its bound)
static void pool_give(egx_gp *gp) {
    PoolEntry e;
    e.key = pool_key(gp, (int)gp->ws.size());
    e.d_xT = gp->d_xT;
    e.d_rhsT = gp->d_rhsT;
    e.d_gamma = gp->d_gamma;
    e.d_fit_coef = gp->d_fit_coef;
    e.slab_M = gp->slab_M;
    e.slab_D = gp->slab_D;
    e.slab_I = gp->slab_I;
    e.ws = std::move(gp->ws);
    e.bytes = pooled_bytes(gp, e.ws);
    gp->d_xT = gp->d_rhsT = gp->d_gamma = gp->d_fit_coef = gp->slab_M = gp->slab_D = nullptr;
    gp->slab_I = nullptr;
    gp->ws.clear();
    bool complete = (e.key.member || (e.slab_M && e.slab_D && e.slab_I)) && e.d_xT && e.d_rhsT && e.d_gamma && e.d_fit_coef && !e.ws.empty();
    for (auto &w : e.ws) {
        complete = complete && w.stream && w.trace.ready && w.inv_stream && w.ev_inv_grp && w.ev_inv_done;
        w.eval_stream = w.stream;
        w.trace.used = 0;
        w.gls_enqueued = false;
        w.block_inv_ready = false;
        w.retried = false;
        w.retry_W = nullptr;
        w.sync_lead = nullptr;
        if (e.key.member) w.M = w.dinv = nullptr, w.d_info = nullptr;  // (views of a group's slabs: set again by the adopting member)
    }
    const size_t cap = pool_cap_bytes();
    if (!complete || e.bytes > cap) {  // a handle whose creation failed half way, or a pool too small for it
        free_entry(e);
        return;
    }
    // the pooled workspaces keep everything but their streams: those are idle now and may serve a handle of any shape (the last
    // workspace's set goes first, so that a handle which adopts this entry next finds every set where it was)
    for (size_t i = e.ws.size(); i-- > 0;) give_streams(e.key.device, e.ws[i]);
    std::list<PoolEntry> evicted;
    std::list<SlabEntry> evicted_slabs;
    {
        std::lock_guard<std::mutex> lock(g_pool_mu);
        const int dev = e.key.device;
        g_pool.push_front(std::move(e));
        // least recently returned entries OF THIS DEVICE go first (the newcomer itself stays), groups' slabs before them
        while (pool_bytes_on(dev) > cap) {
            auto sv = g_slab_pool.end();
            for (auto it = g_slab_pool.begin(); it != g_slab_pool.end(); ++it)
                if (it->device == dev) sv = it;
            if (sv != g_slab_pool.end()) {
                evicted_slabs.splice(evicted_slabs.begin(), g_slab_pool, sv);
                continue;
            }
            auto victim = g_pool.end();
            for (auto it = std::next(g_pool.begin()); it != g_pool.end(); ++it)
                if (it->key.device == dev) victim = it;
            if (victim == g_pool.end()) break;
            evicted.splice(evicted.begin(), g_pool, victim);
        }
    }
    for (auto &v : evicted) free_entry(v);
    for (auto &v : evicted_slabs) free_slab_entry(v);
}
// a group's slabs: adopt pooled ones of exactly these sizes, or allocate
static int group_slabs_take(GroupSlabs &g) {
    {
        std::lock_guard<std::mutex> lock(g_pool_mu);
        for (auto it = g_slab_pool.begin(); it != g_slab_pool.end(); ++it)
            if (it->device == g.device && it->bytes_M == g.bytes_M && it->bytes_D == g.bytes_D && it->bytes_I == g.bytes_I) {
                g.M = it->M, g.D = it->D, g.I = it->I;
                g_slab_pool.erase(it);
                g_pool_hits++;
                return EGX_SUCCESS;
            }
        g_pool_misses++;
    }
    EGX_HIP_CHECK(dev_malloc_bytes(reinterpret_cast<void **>(&g.M), g.bytes_M));
    EGX_HIP_CHECK(dev_malloc_bytes(reinterpret_cast<void **>(&g.D), g.bytes_D));
    EGX_HIP_CHECK(dev_malloc_bytes(reinterpret_cast<void **>(&g.I), g.bytes_I));
    return EGX_SUCCESS;
}
}  // namespace egx
// the last member of a group is gone: its slabs go to the pool (least recently returned slabs of the device make room), or
// are freed when the pool is disabled, they alone exceed its bound, or the group was never completely allocated
egx::GroupSlabs::~GroupSlabs() {
    using namespace egx;
    SlabEntry e;
    e.device = device, e.M = M, e.D = D, e.I = I, e.bytes_M = bytes_M, e.bytes_D = bytes_D, e.bytes_I = bytes_I;
    M = D = nullptr, I = nullptr;
    const size_t cap = pool_cap_bytes();
    if (!(e.M && e.D && e.I) || e.bytes() > cap) {
        int cur = 0;
        const bool have_cur = hipGetDevice(&cur) == hipSuccess;
        free_slab_entry(e);
        if (have_cur) (void)hipSetDevice(cur);
        return;
    }
    std::list<PoolEntry> evicted;
    std::list<SlabEntry> evicted_slabs;
    {
        std::lock_guard<std::mutex> lock(g_pool_mu);
        const int dev = e.device;
        g_slab_pool.push_front(e);
        while (pool_bytes_on(dev) > cap) {
            auto sv = g_slab_pool.end();
            for (auto it = std::next(g_slab_pool.begin()); it != g_slab_pool.end(); ++it)
                if (it->device == dev) sv = it;
            if (sv != g_slab_pool.end()) {
                evicted_slabs.splice(evicted_slabs.begin(), g_slab_pool, sv);
                continue;
            }
            auto victim = g_pool.end();
            for (auto it = g_pool.begin(); it != g_pool.end(); ++it)
                if (it->key.device == dev) victim = it;
            if (victim == g_pool.end()) break;
            evicted.splice(evicted.begin(), g_pool, victim);
        }
    }
    int cur = 0;
    const bool have_cur = hipGetDevice(&cur) == hipSuccess;
    for (auto &v : evicted) free_entry(v);
    for (auto &v : evicted_slabs) free_slab_entry(v);
    if (have_cur) (void)hipSetDevice(cur);
}
namespace egx {

hipError_t dev_malloc_bytes(void **p, size_t bytes) {
    hipError_t e = hipMalloc(p, bytes);
    if (e == hipErrorOutOfMemory || e == hipErrorMemoryAllocation) {
        // the library itself may hold tens of GB of destroyed handles' buffers on this device: give them back, once
        (void)hipGetLastError();
        int dev = 0;
        if (hipGetDevice(&dev) == hipSuccess && pool_trim(dev) > 0) e = hipMalloc(p, bytes);
    }
    return e;
}

// theta (len 1 or h) -> per-dimension coefficient table (d x hcols), see kernels_corr.hip.
// correlation_models.rs:97-98 (sq-exp theta_w), :191 (abs-exp), :333 / :505 (Matern theta_w).
int make_coef(const egx_gp *gp, const double *theta, int64_t theta_len, std::vector<double> &coef,
                     int &hcols, std::vector<double> *theta_full) {
    if (theta_len != 1 && theta_len != gp->h) {
        set_error("theta should be either 1-dim or dim of xtrain (w_star.ncols()), got " + std::to_string(theta_len));
        return EGX_ERR_INVALID_VALUE;
    }
    std::vector<double> th(gp->h);
    for (int l = 0; l < gp->h; l++) th[l] = theta[theta_len == 1 ? 0 : l];
    if (theta_full) *theta_full = th;
    const int d = gp->d, h = gp->h;
    if (!gp->has_w) {
        hcols = 1;
        coef.assign(th.begin(), th.end());  // h == d
        return EGX_SUCCESS;
    }
    const double *w = gp->w_star.data();
    if (gp->corr == EGX_CORR_SQUARED_EXPONENTIAL) {
        hcols = 1;
        coef.assign(d, 0.0);
        for (int j = 0; j < d; j++) {
            double s = 0.0;
            for (int l = 0; l < h; l++) s += (th[l] * w[j * h + l]) * (th[l] * w[j * h + l]);
            coef[j] = std::sqrt(s);
        }
    } else if (gp->corr == EGX_CORR_ABSOLUTE_EXPONENTIAL) {
        hcols = 1;
        coef.assign(d, 0.0);
        for (int j = 0; j < d; j++) {
            double s = 0.0;
            for (int l = 0; l < h; l++) s += std::fabs(w[j * h + l]) * th[l];
            coef[j] = s;
        }
    } else {
        hcols = h;
        coef.assign((size_t)d * h, 0.0);
        for (int j = 0; j < d; j++)
            for (int l = 0; l < h; l++) coef[j * h + l] = th[l] * std::fabs(w[j * h + l]);
    }
    return EGX_SUCCESS;
}

// GPU half of `count` likelihood evaluations on the consecutive workspaces w0 ..: R assembly + RHS rows per candidate,
// ONE factorisation launch sequence for all of them (lock-step batch, kernels_chol.hip) with fused forward solves,
// diagonal gather, async download of (diag C, ft^T, yt^T, info) per candidate.  All asynchronous on the streams of
// workspace w0; every member's eval_stream is set to it.
// core: evaluation j uses the training set of owners[j] and the workspace wss[j]; the workspaces are consecutive slots of ONE
// slab (a handle's, or a group's: egx_gp_create_group), the launches run on the streams of the first
// separate: the chain by separate launches whatever the handle's schedule says (finish_eval's retry of an aborted chain launch)
static int enqueue_eval_core(egx_gp *const *owners, Workspace *const *wss, int count, const std::vector<double> *coefs, int hcols,
                             double *W0, bool separate = false) {
    egx_gp *gp = owners[0];
    Workspace &lead = *wss[0];
    hipStream_t st = lead.stream;
    PotrfInverse inv;
    if (W0) {  // theta-gradient: C^-T rides along (identity rows first, on the rider's own stream: it is idle here)
        inv.W = W0;
        inv.ldw = gp->n_pad;
        inv.sW = (int64_t)gp->n_pad * gp->n_pad;
        inv.sw = lead.inv_stream;
        inv.ev_grp = lead.ev_inv_grp;
        inv.ev_done = lead.ev_inv_done;
        EGX_RC(launch_identity_rows(inv.sw, W0, gp->n_pad, gp->n_pad, count, inv.sW));
    }
    for (int j = 0; j < count; j++) {
        Workspace &w = *wss[j];
        w.eval_stream = st;
        std::memcpy(w.h_coef, coefs[j].data(), sizeof(double) * coefs[j].size());
        EGX_HIP_CHECK(hipMemcpyAsync(w.d_coef, w.h_coef, sizeof(double) * coefs[j].size(), hipMemcpyHostToDevice, st));
    }
    EGX_HIP_CHECK(hipMemsetAsync(lead.d_info, 0, sizeof(int) * (size_t)count, st));
    EGX_HIP_CHECK(hipEventRecord(lead.ev[0], st));
    // the candidates' pointers for the batched front-end and tail launches (by value in the kernel arguments)
    EvalBatchPtrs bp;
    // (a lone candidate takes the same route: its tail is then one launch instead of a gather + four copies -- 30 us of a
    //  0.12-0.4 ms evaluation at n = 128 ... 1024)
    const bool batched = count >= 1 && count <= EvalBatchPtrs::kMax;
    if (batched)
        for (int j = 0; j < count; j++) {
            Workspace &w = *wss[j];
            bp.xT[j] = owners[j]->d_xT, bp.coef[j] = w.d_coef, bp.rhsT[j] = owners[j]->d_rhsT, bp.xs[j] = w.d_xs, bp.M[j] = w.M;
            bp.h_diag[j] = w.h_diag, bp.h_rows[j] = w.h_rows, bp.h_info[j] = w.h_info, bp.d_info[j] = w.d_info;
        }
    int front = batched ? launch_eval_front_batch(st, gp->corr, bp, count, gp->n_pad, gp->n, gp->d, hcols, gp->nugget, gp->ld,
                                                  gp->n_pad, gp->rhs_pad, gp->q)
                        : EGX_ERR_UNSUPPORTED;
    if (front != EGX_SUCCESS && front != EGX_ERR_UNSUPPORTED) return front;
    if (front == EGX_ERR_UNSUPPORTED)  // Matern with KPLS weights (hcols > 1) or d > 64: launch by launch
        for (int j = 0; j < count; j++) {
            Workspace &w = *wss[j];
            egx_gp *o = owners[j];
            EGX_RC(launch_corr_sym(st, gp->corr, o->d_xT, gp->n_pad, gp->n, gp->d, w.d_coef, hcols, gp->nugget, w.M, gp->ld,
                                   gp->n_pad, w.d_xs));
            EGX_RC(launch_fill_rows(st, w.M, gp->ld, gp->n_pad, gp->rhs_pad, o->d_rhsT, gp->n_pad, gp->q, gp->n_pad));
        }
    EGX_HIP_CHECK(hipEventRecord(lead.ev[1], st));
    int *lead_sync = dev_sync(gp, (int)(&lead - gp->ws.data()));
    PotrfBatch pb;
    pb.count = count;
    pb.sM = gp->stride_M;
    pb.sD = gp->stride_D;
    pb.sI = 1;
    pb.left = gp->sched.left;  // (the handle's schedule: schedule_for, egx_internal.h)
    pb.w_left = gp->sched.w_left;
    pb.pipe = separate ? 0 : gp->sched.pipe;
    pb.whole = separate ? 0 : gp->sched.whole;
    pb.flow = separate || count > 1 || gp->group ? 0 : gp->sched.flow;
    pb.flow_tail = separate || count > 1 || gp->group ? 0 : gp->sched.flow_tail;
    pb.group_panels = gp->sched.group_panels;
    pb.seqs = gp->lockstep > 0 ? ((int)gp->ws.size() + gp->lockstep - 1) / gp->lockstep : 1;
    pb.sync = pb.pipe || pb.flow || pb.flow_tail ? lead_sync : nullptr;  // the chain of a group of panels as one persistent launch (kernels_pipe.hip)
    pb.sS = gp->stride_S;
    for (int j = 0; j < count; j++) {  // what finish_eval needs to run this evaluation once more, alone, by separate launches
        wss[j]->retry_hcols = hcols;
        wss[j]->retry_W = W0 ? W0 + (size_t)j * (size_t)gp->n_pad * gp->n_pad : nullptr;
        wss[j]->retry_ncoef = (int)coefs[j].size();
        wss[j]->retried = separate;
    }
    EGX_RC(launch_potrf(st, lead.M, gp->ld, gp->n_pad, gp->m_tot, lead.dinv, lead.d_info, lead.lk.s2 ? &lead.lk : nullptr,
                        &lead.trace, &pb, W0 ? &inv : nullptr));
    EGX_HIP_CHECK(hipEventRecord(lead.ev[2], st));
    // what the host needs back: ONE launch writes every candidate's diagonal, solved right-hand-side rows, info and hand-off
    // diagnostics straight into its pinned host buffers (k_eval_tail); beyond 16 candidates (never today): a gather + copies each
    if (batched) EGX_RC(launch_eval_tail(st, bp, count, gp->ld, gp->n, gp->n_pad, gp->q, gp->gls_device ? 0 : 1, pb.sync));
    for (int j = 0; j < count; j++) {
        Workspace &w = *wss[j];
        if (!batched) {
            EGX_RC(launch_gather_diag(st, w.M, gp->ld, gp->n, w.d_diag));
            EGX_HIP_CHECK(hipMemcpyAsync(w.h_diag, w.d_diag, sizeof(double) * gp->n, hipMemcpyDeviceToHost, st));
        }
        w.gls_enqueued = gp->gls_device;
        if (gp->gls_device) {
            // p > 1: ft never leaves the device.  Gram matrix of the solved rows [ft | yt] (split-K MFMA, fixed-order
            // reduction), its Cholesky factor by the same blocked kernels (R = L^T is the reference's QR factor with a
            // positive diagonal, row p of L is z = L^-1 ft^T yt), and only that (rhs_pad x rhs_pad) factor goes to the host
            const int g = gp->rhs_pad;
            EGX_RC(launch_gram_lower(st, w.M + (size_t)gp->n_pad * gp->ld, gp->ld, g, gp->n_pad, w.d_gneg, g, w.d_gramP));
            EGX_RC(launch_gram_finish(st, w.d_gneg, w.d_gram, g, g, gp->q));
            EGX_HIP_CHECK(hipMemsetAsync(w.d_ginfo, 0, sizeof(int), st));
            EGX_RC(launch_potrf(st, w.d_gram, g, g, g, w.d_gdinv, w.d_ginfo));
            EGX_HIP_CHECK(hipMemcpyAsync(w.h_gram, w.d_gram, sizeof(double) * (size_t)g * g, hipMemcpyDeviceToHost, st));
            EGX_HIP_CHECK(hipMemcpyAsync(w.h_ginfo, w.d_ginfo, sizeof(int), hipMemcpyDeviceToHost, st));
        } else if (!batched) {
            EGX_HIP_CHECK(hipMemcpyAsync(w.h_rows, w.M + (size_t)gp->n_pad * gp->ld,
                                         sizeof(double) * (size_t)gp->q * gp->n_pad, hipMemcpyDeviceToHost, st));
        }
        w.sync_lead = pb.sync;  // word 0 of the LEAD's hand-off words: non-zero iff a bounded wait inside a chain launch ran out
        if (batched) continue;
        EGX_HIP_CHECK(hipMemcpyAsync(w.h_info, w.d_info, sizeof(int), hipMemcpyDeviceToHost, st));
        if (pb.sync)
            EGX_HIP_CHECK(hipMemcpyAsync(w.h_info + 1, pb.sync, 8 * sizeof(int), hipMemcpyDeviceToHost, st));
        else  // (a handle whose schedule has no chain launches never touches its hand-off words)
            w.h_info[1] = 0;
    }
    EGX_HIP_CHECK(hipEventRecord(lead.ev[3], st));
    return EGX_SUCCESS;
}

int enqueue_eval_group(egx_gp *gp, int w0, int count, const std::vector<double> *coefs, int hcols, double *W0) {
    std::vector<egx_gp *> owners((size_t)count, gp);
    std::vector<Workspace *> wss((size_t)count);
    for (int j = 0; j < count; j++) wss[(size_t)j] = &gp->ws[(size_t)(w0 + j)];
    return enqueue_eval_core(owners.data(), wss.data(), count, coefs, hcols, W0);
}
int enqueue_eval_members(egx_gp *const *gps, int count, const std::vector<double> *coefs, int hcols) {
    std::vector<Workspace *> wss((size_t)count);
    for (int j = 0; j < count; j++) wss[(size_t)j] = &gps[j]->ws[0];
    return enqueue_eval_core(gps, wss.data(), count, coefs, hcols, nullptr);
}

// one evaluation on one workspace (its own streams)
int enqueue_eval(egx_gp *gp, Workspace &w, const std::vector<double> &coef, int hcols) {
    return enqueue_eval_group(gp, (int)(&w - gp->ws.data()), 1, &coef, hcols);
}

// Device GLS route of finish_eval (p >= 2).  Returns 0 = evaluation finished (out filled), 1 = fall back to the host
// Householder route, < 0 = -(egx_rc) on a HIP failure.
//   A = ft^T ft = L L^T,  z = L^-1 ft^T yt  (device, enqueue_eval)  ->  R = L^T (algorithm.rs:1007 `qr`, positive
//   diagonal), beta = R^-1 z (:1030), rho = yt - ft beta and sum rho^2 on the device (:1031-1032).
// The conditioning test of :1010-1027 compares sigma_min / sigma_max of R with 1e-10.  For p <= 64 the singular values
// are computed exactly (Jacobi on the p x p factor); wider trends estimate the ratio in O(p^2) by power / inverse
// iteration and keep two decades of margin.  The Gram route is taken only when the ratio is >= 1e-7 (1e-6 estimated) and
// the factor's diagonal ratio >= 1e-4; anything closer to the threshold goes to the host Householder + SVD route, which
// decides exactly as for p = 1.
static int finish_eval_device_gls(egx_gp *gp, Workspace &w, EvalResult &out, int keep) {
    const int n = gp->n, p = gp->p, n_pad = gp->n_pad, g = gp->rhs_pad;
    if (*w.h_ginfo != 0) return 1;
    const double *L = w.h_gram;  // (g x g) row-major lower factor
    double dmin = std::numeric_limits<double>::infinity(), dmax = 0.0;
    for (int i = 0; i < p; i++) {
        const double v = L[(size_t)i * g + i];
        if (!(v > 0.0) || !std::isfinite(v)) return 1;
        dmin = std::fmin(dmin, v);
        dmax = std::fmax(dmax, v);
    }
    if (dmin / dmax < 1e-4) return 1;
    // row-oriented triangular kernels on the (g-strided) factor: contiguous inner loops
    auto mul_lt = [&](const std::vector<double> &v, std::vector<double> &u) {  // u = L^T v
        std::fill(u.begin(), u.end(), 0.0);
        for (int l = 0; l < p; l++) {
            const double *row = L + (size_t)l * g;
            const double vl = v[l];
            for (int i = 0; i <= l; i++) u[i] += row[i] * vl;
        }
    };
    auto mul_l = [&](const std::vector<double> &u, std::vector<double> &v) {  // v = L u
        for (int i = 0; i < p; i++) {
            const double *row = L + (size_t)i * g;
            double s = 0.0;
            for (int l = 0; l <= i; l++) s += row[l] * u[l];
            v[i] = s;
        }
    };
    auto solve_l = [&](std::vector<double> &v) {  // v <- L^-1 v
        for (int i = 0; i < p; i++) {
            const double *row = L + (size_t)i * g;
            double s = v[i];
            for (int l = 0; l < i; l++) s -= row[l] * v[l];
            v[i] = s / row[i];
        }
    };
    auto solve_lt = [&](std::vector<double> &v) {  // v <- L^-T v
        for (int i = p - 1; i >= 0; i--) {
            const double *row = L + (size_t)i * g;
            const double wi = v[i] / row[i];
            v[i] = wi;
            for (int l = 0; l < i; l++) v[l] -= row[l] * wi;
        }
    };
    if (p <= 64) {
        // the reference's own test, exactly (algorithm.rs:1010-1027): singular values of the p x p factor R = L^T by
        // one-sided Jacobi, O(p^3) on a tiny matrix.  A ratio below 1e-10 is the reference's error; the Gram route is
        // taken down to 1e-7 only (normal equations square the condition number), the Householder route decides below
        std::vector<double> Rm((size_t)p * p, 0.0);
        for (int i = 0; i < p; i++)
            for (int l = i; l < p; l++) Rm[(size_t)i * p + l] = L[(size_t)l * g + i];
        const std::vector<double> sv = hm::singular_values(Rm, p);
        if (!(sv[p - 1] / sv[0] >= 1e-7)) return 1;
    } else {
        // wide trends (quadratic in d = 32: p = 561): sigma_max^2 by power iteration on R^T R = L L^T (R = L^T),
        // 1 / sigma_min^2 by inverse iteration.  Both UNDER-estimate their target, so the ratio is OVER-estimated: the
        // Gram route is taken only with two decades of margin (>= 1e-6), anything closer to the reference's 1e-10
        // threshold goes to the host Householder + SVD route, which decides exactly as for p = 1
        std::vector<double> v(p), u(p);
        auto normalize = [&](std::vector<double> &t) {
            double s = 0.0;
            for (double e : t) s += e * e;
            s = std::sqrt(s);
            for (double &e : t) e /= s;
            return s;
        };
        for (int i = 0; i < p; i++) v[i] = 1.0 + 0.37 * ((i * 2654435761u) % 1000) / 1000.0;
        normalize(v);
        double smax2 = 0.0, sinv2 = 0.0;
        for (int it = 0; it < 12; it++) {
            mul_lt(v, u);
            mul_l(u, v);
            smax2 = normalize(v);
        }
        for (int i = 0; i < p; i++) v[i] = 1.0 - 0.29 * ((i * 40503u) % 1000) / 1000.0;
        normalize(v);
        for (int it = 0; it < 12; it++) {
            solve_l(v);
            solve_lt(v);
            sinv2 = normalize(v);
        }
        const double ratio = 1.0 / std::sqrt(sinv2 * smax2);  // ~ sigma_min / sigma_max
        if (!(ratio >= 1e-6)) return 1;
    }
    // beta = L^-T z   (z = row p of the factor)
    std::vector<double> beta(L + (size_t)p * g, L + (size_t)p * g + p);
    solve_lt(beta);
    std::memcpy(w.h_beta, beta.data(), sizeof(double) * p);
    if (hipMemcpyAsync(w.d_beta, w.h_beta, sizeof(double) * p, hipMemcpyHostToDevice, w.stream) != hipSuccess) return -EGX_ERR_HIP;
    const double *rows = w.M + (size_t)n_pad * gp->ld;
    int rc = launch_gls_residual(w.stream, rows, gp->ld, rows + (size_t)p * gp->ld, w.d_beta, p, n, n_pad, w.d_rhs, w.d_part);
    if (rc) return -rc;
    const int nblk = (n_pad + 255) / 256;
    if (hipMemcpyAsync(w.h_part, w.d_part, sizeof(double) * nblk, hipMemcpyDeviceToHost, w.stream) != hipSuccess ||
        hipStreamSynchronize(w.stream) != hipSuccess) {
        set_error("device GLS: copy / synchronise failed");
        (void)hipGetLastError();
        return -EGX_ERR_HIP;
    }
    double rho_sqr = 0.0;
    for (int b = 0; b < nblk; b++) rho_sqr += w.h_part[b];
    double slog = 0.0;
    slog = hm::sum_log10(w.h_diag, n);
    const double sigma2n = rho_sqr / (double)n;
    out.lkh = -(double)n * (std::log10(sigma2n) + slog * 2.0 / (double)n);  // algorithm.rs:1039-1043
    out.sigma2n = sigma2n;
    out.status = EGX_STATUS_OK;
    out.rho_on_device = true;
    if (keep == 1) {
        out.beta = beta;
        out.ft_qr_r.assign((size_t)p * p, 0.0);
        for (int i = 0; i < p; i++)
            for (int l = i; l < p; l++) out.ft_qr_r[(size_t)i * p + l] = L[(size_t)l * g + i];
        // the fitted state keeps a host copy of ft (serde schema, few-query prediction paths)
        std::vector<double> tmp((size_t)p * n_pad);
        if (hipMemcpy(tmp.data(), rows, sizeof(double) * tmp.size(), hipMemcpyDeviceToHost) != hipSuccess) {
            set_error("device GLS: download of ft failed");
            (void)hipGetLastError();
            return -EGX_ERR_HIP;
        }
        out.ft.assign((size_t)n * p, 0.0);
        for (int l = 0; l < p; l++)
            for (int i = 0; i < n; i++) out.ft[(size_t)i * p + l] = tmp[(size_t)l * n_pad + i];
    }
    return 0;
}

// Host half: algorithm.rs:1007-1043 on the downloaded ft, yt, diag(C).  keep: 0 = the scalars only, 1 = everything the
// fitted state needs (beta, rho, ft, its QR factor), 2 = rho only (the theta-gradient)
int finish_eval(egx_gp *gp, Workspace &w, EvalResult &out, int keep) {
    EGX_HIP_CHECK(hipStreamSynchronize(w.eval_stream));
    const int n = gp->n, p = gp->p, n_pad = gp->n_pad;
    out = EvalResult();
    if (w.h_info[1] != 0) g_chain_aborts++;
    if (w.h_info[1] != 0 && g_pipe_retry.load() != 0 && !w.retried) {
        // A bounded wait inside a chain launch ran out (EGX_PIPE_TIMEOUT_MS): waves pre-empted under multi-process sharing of the
        // GPU or a debugger are enough -- nothing numerical, and the reference's cholesky() (algorithm.rs:1004) never fails for
        // such a reason (the objective only ever sees numerical errors, :893-896).  The evaluation is enqueued ONCE more, alone,
        // on the separate-launch schedule of the same handle (no device-side wait anywhere); only if that fails too is it an
        // error.  (The factor then carries the separate-launch rounding, 1e-10 from the chain launch's: schedule.h.)
        g_chain_retries++;
        egx_gp *owner = gp;
        Workspace *wp = &w;
        const std::vector<double> coef(w.h_coef, w.h_coef + w.retry_ncoef);
        EGX_RC(enqueue_eval_core(&owner, &wp, 1, &coef, w.retry_hcols, w.retry_W, true));
        EGX_HIP_CHECK(hipStreamSynchronize(w.eval_stream));
    }
    if (w.h_info[1] != 0) {  // (never seen outside the test that forces it: a hand-off inside k_potrf_pipe did not arrive)
        std::string more;
        {   // the launches' tickets and this matrix' published strips, for the report
            const int np_all = (gp->n_pad + 255) / 256;
            std::vector<int> hdr((size_t)8 + 2 * np_all);
            int strips = -1;
            if (w.sync_lead && hipMemcpy(hdr.data(), w.sync_lead, sizeof(int) * hdr.size(), hipMemcpyDeviceToHost) == hipSuccess &&
                hipMemcpy(&strips, dev_sync(gp, (int)(&w - gp->ws.data())) + 1, sizeof(int), hipMemcpyDeviceToHost) == hipSuccess) {
                more = "; signal word " + std::to_string(hdr[3]) + ", strips of this matrix " + std::to_string(strips) + ", tickets task/start per group:";
                for (int P = 0; P < np_all; P++)
                    if (hdr[8 + P] || hdr[8 + np_all + P]) more += " [" + std::to_string(P) + "] " + std::to_string(hdr[8 + P]) + "/" + std::to_string(hdr[8 + np_all + P]);
            }
            (void)hipGetLastError();
        }
        set_error("factorisation aborted: a wait inside the pipelined chain kernel exceeded EGX_PIPE_TIMEOUT_MS (waiter " +
                  std::to_string((unsigned)w.h_info[5] >> 28) + " panel " + std::to_string((w.h_info[5] >> 20) & 255) + " arg " +
                  std::to_string(w.h_info[5] & 0xfffff) + ", word " + std::to_string(w.h_info[6]) + " wanted " +
                  std::to_string(w.h_info[7]) + " saw " + std::to_string(w.h_info[8]) + ", " + std::to_string(w.h_info[1]) + " waiters gave up" + more + ")");
        return EGX_ERR_HIP;
    }
    if (*w.h_info != 0) {
        out.status = EGX_STATUS_NOT_POSITIVE_DEFINITE;
        return EGX_SUCCESS;
    }
    for (int i = 0; i < n; i++)
        if (!(w.h_diag[i] > 0.0) || !std::isfinite(w.h_diag[i])) {
            out.status = EGX_STATUS_NOT_POSITIVE_DEFINITE;
            return EGX_SUCCESS;
        }
    if (w.gls_enqueued) {
        int route = finish_eval_device_gls(gp, w, out, keep);
        if (route <= 0) return route < 0 ? -route : EGX_SUCCESS;  // done (or a HIP error)
        // route 1: the Gram route is not trustworthy here (ft too ill conditioned for normal equations): the solved
        // rows come over after all and the Householder path below decides, exactly as for p = 1
        EGX_HIP_CHECK(hipMemcpyAsync(w.h_rows, w.M + (size_t)n_pad * gp->ld, sizeof(double) * (size_t)gp->q * n_pad,
                                     hipMemcpyDeviceToHost, w.stream));
        EGX_HIP_CHECK(hipStreamSynchronize(w.stream));
    }
    // ft (column-major copy for the QR), yt
    std::vector<double> a((size_t)n * p), qty(n);
    for (int l = 0; l < p; l++) std::memcpy(&a[(size_t)l * n], w.h_rows + (size_t)l * n_pad, sizeof(double) * n);
    std::memcpy(qty.data(), w.h_rows + (size_t)p * n_pad, sizeof(double) * n);
    std::vector<double> yt(qty);
    std::vector<double> ftcm;
    ftcm = a;                     // keep ft (column-major) for rho
    hm::qr_apply(a, n, p, &qty);  // :1007  a -> R (upper), qty -> Q^T yt
    std::vector<double> R((size_t)p * p, 0.0);
    for (int i = 0; i < p; i++)
        for (int l = i; l < p; l++) R[(size_t)i * p + l] = (i < n) ? a[(size_t)l * n + i] : 0.0;
    // :1010-1027 conditioning of the p x p factor
    std::vector<double> sv = hm::singular_values(R, p);
    const double cond_ft = sv[p - 1] / sv[0];
    if (!(cond_ft >= 1e-10)) {
        std::vector<double> fcm((size_t)n * p);
        for (int l = 0; l < p; l++)
            for (int i = 0; i < n; i++) fcm[(size_t)l * n + i] = gp->F[(size_t)i * p + l];
        hm::qr_apply(fcm, n, p, nullptr);
        std::vector<double> RF((size_t)p * p, 0.0);
        for (int i = 0; i < p; i++)
            for (int l = i; l < p; l++) RF[(size_t)i * p + l] = (i < n) ? fcm[(size_t)l * n + i] : 0.0;
        std::vector<double> svf = hm::singular_values(RF, p);
        out.status = (svf[0] / svf[p - 1] > 1e15) ? EGX_STATUS_ILL_CONDITIONED_F : EGX_STATUS_ILL_CONDITIONED_FT;
        return EGX_SUCCESS;
    }
    // :1030 beta = R^-1 (Q^T yt)[:p]
    std::vector<double> beta(p);
    for (int i = p - 1; i >= 0; i--) {
        double s = qty[i];
        for (int l = i + 1; l < p; l++) s -= R[(size_t)i * p + l] * beta[l];
        beta[i] = s / R[(size_t)i * p + i];
    }
    // :1031-1032 rho = yt - ft beta
    std::vector<double> rho(yt);
    for (int l = 0; l < p; l++) {
        const double b = beta[l];
        const double *col = &ftcm[(size_t)l * n];
        for (int i = 0; i < n; i++) rho[i] -= col[i] * b;
    }
    double rho_sqr = 0.0;
    for (int i = 0; i < n; i++) rho_sqr += rho[i] * rho[i];
    // :1039-1043
    double slog = 0.0;
    slog = hm::sum_log10(w.h_diag, n);
    const double logdet = slog * 2.0 / (double)n;
    const double sigma2n = rho_sqr / (double)n;
    out.lkh = -(double)n * (std::log10(sigma2n) + logdet);
    out.sigma2n = sigma2n;
    out.status = EGX_STATUS_OK;
    if (keep == 2) out.rho = rho;
    if (keep == 1) {
        out.beta = beta;
        out.rho = rho;
        out.ft_qr_r = R;
        out.ft.assign((size_t)n * p, 0.0);
        for (int l = 0; l < p; l++)
            for (int i = 0; i < n; i++) out.ft[(size_t)i * p + l] = ftcm[(size_t)l * n + i];
    }
    return EGX_SUCCESS;
}

void record_timings(egx_gp *gp, Workspace &w, double host_ms, double solve_ms) {
    float t01 = 0, t12 = 0, t03 = 0;
    hipEventElapsedTime(&t01, w.ev[0], w.ev[1]);
    hipEventElapsedTime(&t12, w.ev[1], w.ev[2]);
    hipEventElapsedTime(&t03, w.ev[0], w.ev[3]);
    egx_timings &t = gp->timings;
    t.corr_build_ms = t01;
    t.potrf_ms = t12;
    t.potrf_syrk_ms = 0.0;
    t.syrk_launches = 0;
    t.syrk_flops = 0;
    double fl = 0.0;
    for (int i = 0; i < w.trace.used; i++) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, w.trace.e0[i], w.trace.e1[i]) == hipSuccess) {
            t.potrf_syrk_ms += ms;
            fl += w.trace.flops[i];
            t.syrk_launches++;
        }
    }
    t.syrk_flops = (int64_t)fl;
    t.solve_ms = solve_ms;
    t.host_ms = host_ms;
    t.total_ms = t03 + host_ms + solve_ms;
    const double nn = (double)gp->n;
    t.potrf_flops = (int64_t)(nn * nn * nn / 3.0);
    t.corr_bytes = (int64_t)(8.0 * nn * gp->d + 8.0 * nn * (nn + 1.0) / 2.0);
}

bool has_nan(const double *theta, int64_t len) {
    for (int64_t i = 0; i < len; i++)
        if (std::isnan(theta[i])) return true;
    return false;
}

int eval_one(egx_gp *gp, int widx, const double *theta, int64_t theta_len, EvalResult &res, bool keep) {
    std::vector<double> coef;
    int hcols = 1;
    EGX_RC(make_coef(gp, theta, theta_len, coef, hcols, nullptr));
    if (has_nan(theta, theta_len)) {  // algorithm.rs:885-891
        res = EvalResult();
        res.status = EGX_STATUS_NAN_THETA;
        return EGX_SUCCESS;
    }
    if (widx == 0) gp->fitted = false;
    Workspace &w = gp->ws[widx];
    EGX_RC(enqueue_eval(gp, w, coef, hcols));
    auto t0 = std::chrono::steady_clock::now();
    EGX_RC(finish_eval(gp, w, res, keep ? 1 : 0));
    // host_ms includes the wait for the stream; subtract GPU time below
    auto t1 = std::chrono::steady_clock::now();
    float gpu = 0;
    hipEventElapsedTime(&gpu, w.ev[0], w.ev[3]);
    double host_ms = std::chrono::duration<double, std::milli>(t1 - t0).count() - gpu;
    if (host_ms < 0) host_ms = 0;
    if (widx == 0) record_timings(gp, w, host_ms, 0.0);
    return EGX_SUCCESS;
}

// w.d_vec <- C^-T w.d_rhs  (block inverses are rebuilt: the factor in w.M has just changed)
static int ensure_block_inverse_buffer(egx_gp *gp, Workspace &w) {
    if (!w.dW) EGX_HIP_CHECK(dev_malloc(&w.dW, sizeof(double) * block_inverse_doubles(gp->n_pad)));
    return EGX_SUCCESS;
}
// The inverse blocks only need the factor, not the host's half of the likelihood: a fit launches them right behind the
// evaluation, so that they run while the host waits for the read-back and does its GLS (84 us off a fit's critical path)
static int prelaunch_block_inverse(egx_gp *gp, Workspace &w, hipStream_t st) {
    EGX_RC(ensure_block_inverse_buffer(gp, w));
    EGX_RC(launch_block_inverse(st, w.M, gp->ld, gp->n_pad, w.dinv, w.dW));
    w.block_inv_ready = true;
    return EGX_SUCCESS;
}
int backward_solve(egx_gp *gp, Workspace &w) {
    if (!w.block_inv_ready) EGX_RC(prelaunch_block_inverse(gp, w, w.stream));
    w.block_inv_ready = false;
    EGX_RC(launch_trsv_t(w.stream, w.M, gp->ld, gp->n_pad, w.dW, w.d_rhs, w.d_vec));
    return EGX_SUCCESS;
}

// second half of a fit at fixed theta, behind the evaluation that was enqueued on workspace 0, in two steps so that several
// models' tails overlap (egx_gp_finalize_multi): (1) the host half of the likelihood, then gamma = C^-T rho and the fitted
// state's device copies ENQUEUED on the model's own stream; (2) wait for them, take over the fitted state
struct FinalizeTail {
    EvalResult res;
    std::chrono::steady_clock::time_point t0, t1;
};
static int finalize_tail_enqueue(egx_gp *gp, const std::vector<double> &coef, int hcols, FinalizeTail &ft) {
    Workspace &w = gp->ws[0];
    EvalResult &res = ft.res;
    ft.t0 = std::chrono::steady_clock::now();
    EGX_RC(finish_eval(gp, w, res, 1));
    ft.t1 = std::chrono::steady_clock::now();
    if (res.status == EGX_STATUS_NOT_POSITIVE_DEFINITE) {
        set_error("LinalgError: matrix is not positive definite (pivot " + std::to_string(*w.h_info) + ")");
        return EGX_ERR_LINALG;
    }
    if (res.status == EGX_STATUS_ILL_CONDITIONED_F) {
        set_error("LikelihoodComputation computation error: F is too ill conditioned. Poor combination of "
                  "regression model and observations.");
        return EGX_ERR_LIKELIHOOD;
    }
    if (res.status == EGX_STATUS_ILL_CONDITIONED_FT) {
        set_error("LikelihoodComputation computation error: ft is too ill conditioned, try another theta again");
        return EGX_ERR_LIKELIHOOD;
    }
    // gamma = C^-T rho   (algorithm.rs:1034)
    const int n = gp->n, n_pad = gp->n_pad;
    if (!res.rho_on_device) {
        std::memset(w.h_vec, 0, sizeof(double) * n_pad);
        std::memcpy(w.h_vec, res.rho.data(), sizeof(double) * n);
        EGX_HIP_CHECK(hipMemcpyAsync(w.d_rhs, w.h_vec, sizeof(double) * n_pad, hipMemcpyHostToDevice, w.stream));
    }
    EGX_HIP_CHECK(hipEventRecord(w.ev[4], w.stream));
    EGX_RC(backward_solve(gp, w));
    EGX_HIP_CHECK(hipMemcpyAsync(gp->d_gamma, w.d_vec, sizeof(double) * n_pad, hipMemcpyDeviceToDevice, w.stream));
    EGX_HIP_CHECK(hipMemcpyAsync(w.h_vec, w.d_vec, sizeof(double) * n_pad, hipMemcpyDeviceToHost, w.stream));
    EGX_HIP_CHECK(hipMemcpyAsync(gp->d_fit_coef, w.d_coef, sizeof(double) * coef.size(), hipMemcpyDeviceToDevice,
                                 w.stream));
    if (hcols == 1) EGX_RC(launch_scale_rows(w.stream, gp->d_xT, gp->n_pad, gp->d, gp->d_fit_coef, dev_xs_fit(gp)));
    return EGX_SUCCESS;
}
// ... of the members of a group that were evaluated in lock-step (egx_gp_finalize_multi): the host halves one after the other,
// then ONE launch sequence for all back-substitutions (launch_trsv_t_batch: 64 launches instead of 64 per model -- the
// command processor serialises such ~5 us launches however many streams or host threads issue them: eight n = 8192 experts
// 36.9 -> X ms, profiles/r05_expert_group_*.txt) on the stream the evaluation ran on, and the fitted state's device copies.
// rcs[j] / errs[j]: per-model outcome of the host half (a model that failed is left out of the launches).
static int finalize_tails_lockstep(egx_gp *const *gps, int len, const std::vector<double> *coefs, int hcols, FinalizeTail *tails,
                                   int *rcs, std::string *errs) {
    hipStream_t st = gps[0]->ws[0].eval_stream;
    SolveBatchPtrs bp;
    int live[SolveBatchPtrs::kMax], nlive = 0;
    // the host halves (log-determinant over the diagonal, GLS: ~0.1 ms per model at n = 8192, no HIP launches on the host-GLS
    // route) side by side on host threads; the device-GLS route (p > 1 trend columns) issues launches: one after the other
    std::vector<int> frc((size_t)len, EGX_SUCCESS);
    std::vector<std::string> ferr((size_t)len);
    auto host_half = [&](int j) {
        FinalizeTail &ft = tails[j];
        ft.t0 = std::chrono::steady_clock::now();
        frc[(size_t)j] = finish_eval(gps[j], gps[j]->ws[0], ft.res, 1);
        ft.t1 = std::chrono::steady_clock::now();
        if (frc[(size_t)j] != EGX_SUCCESS) ferr[(size_t)j] = last_error_string();  // (the error text is per thread)
    };
    if (!gps[0]->gls_device) {
        std::vector<std::thread> pool;
        for (int j = 1; j < len; j++)
            pool.emplace_back([&, j] {
                if (hipSetDevice(gps[j]->device) != hipSuccess) {
                    frc[(size_t)j] = EGX_ERR_HIP;
                    ferr[(size_t)j] = "hipSetDevice failed in a finalize worker";
                    return;
                }
                host_half(j);
            });
        host_half(0);
        for (auto &t : pool) t.join();
    } else {
        for (int j = 0; j < len; j++) host_half(j);
    }
    for (int j = 0; j < len; j++) {
        egx_gp *gp = gps[j];
        Workspace &w = gp->ws[0];
        FinalizeTail &ft = tails[j];
        int rc = frc[(size_t)j];
        if (rc != EGX_SUCCESS) set_error(ferr[(size_t)j]);
        if (rc == EGX_SUCCESS && ft.res.status != EGX_STATUS_OK) {
            rc = ft.res.status == EGX_STATUS_NOT_POSITIVE_DEFINITE ? EGX_ERR_LINALG : EGX_ERR_LIKELIHOOD;
            set_error(ft.res.status == EGX_STATUS_NOT_POSITIVE_DEFINITE
                          ? "LinalgError: matrix is not positive definite (pivot " + std::to_string(*w.h_info) + ")"
                          : ft.res.status == EGX_STATUS_ILL_CONDITIONED_F
                                ? std::string("LikelihoodComputation computation error: F is too ill conditioned. Poor combination of "
                                              "regression model and observations.")
                                : std::string("LikelihoodComputation computation error: ft is too ill conditioned, try another theta again"));
        }
        rcs[j] = rc;
        if (rc != EGX_SUCCESS) {
            errs[j] = last_error_string();
            continue;
        }
        if (!ft.res.rho_on_device) {
            std::memset(w.h_vec, 0, sizeof(double) * gp->n_pad);
            std::memcpy(w.h_vec, ft.res.rho.data(), sizeof(double) * gp->n);
            EGX_HIP_CHECK(hipMemcpyAsync(w.d_rhs, w.h_vec, sizeof(double) * gp->n_pad, hipMemcpyHostToDevice, st));
        }
        bp.M[nlive] = w.M, bp.dinv[nlive] = w.dinv, bp.dW[nlive] = w.dW, bp.rhs[nlive] = w.d_rhs, bp.vec[nlive] = w.d_vec;
        live[nlive++] = j;
    }
    if (nlive == 0) return EGX_SUCCESS;
    egx_gp *g0 = gps[live[0]];
    EGX_RC(launch_trsv_t_batch(st, bp, nlive, g0->ld, g0->n_pad));  // gamma = C^-T rho (algorithm.rs:1034); inverse blocks: multi_eval
    for (int q = 0; q < nlive; q++) {
        egx_gp *gp = gps[live[q]];
        Workspace &w = gp->ws[0];
        const size_t nb = sizeof(double) * gp->n_pad;
        EGX_HIP_CHECK(hipMemcpyAsync(gp->d_gamma, w.d_vec, nb, hipMemcpyDeviceToDevice, st));
        EGX_HIP_CHECK(hipMemcpyAsync(w.h_vec, w.d_vec, nb, hipMemcpyDeviceToHost, st));
        EGX_HIP_CHECK(hipMemcpyAsync(gp->d_fit_coef, w.d_coef, sizeof(double) * coefs[live[q]].size(), hipMemcpyDeviceToDevice, st));
        if (hcols == 1) EGX_RC(launch_scale_rows(st, gp->d_xT, gp->n_pad, gp->d, gp->d_fit_coef, dev_xs_fit(gp)));
    }
    EGX_HIP_CHECK(hipStreamSynchronize(st));
    return EGX_SUCCESS;
}
static int finalize_tail_complete(egx_gp *gp, const std::vector<double> &coef, int hcols, const std::vector<double> &thfull,
                                  FinalizeTail &ft) {
    Workspace &w = gp->ws[0];
    EvalResult &res = ft.res;
    EGX_HIP_CHECK(hipStreamSynchronize(w.stream));
    auto t2 = std::chrono::steady_clock::now();
    // The one-launch back-substitution leaves NaN in gamma when one of its bounded waits ran out (kernels_chol.hip): like an
    // aborted chain launch -- not a numerical event, the reference knows no such failure -- it is run ONCE more, launch per
    // block, from the right-hand side it left untouched (egx_chain_stats counts it; "pipe_retry" = 0: the error at once)
    auto finite = [&]() {
        for (int i = 0; i < gp->n; i++)
            if (!std::isfinite(w.h_vec[i])) return false;
        return true;
    };
    if (!finite()) {
        g_chain_aborts++;
        bool ok = false;
        if (g_pipe_retry.load() != 0 && w.dW != nullptr) {
            g_chain_retries++;
            EGX_RC(launch_trsv_t(w.stream, w.M, gp->ld, gp->n_pad, w.dW, w.d_rhs, w.d_vec, true));
            EGX_HIP_CHECK(hipMemcpyAsync(gp->d_gamma, w.d_vec, sizeof(double) * gp->n_pad, hipMemcpyDeviceToDevice, w.stream));
            EGX_HIP_CHECK(hipMemcpyAsync(w.h_vec, w.d_vec, sizeof(double) * gp->n_pad, hipMemcpyDeviceToHost, w.stream));
            EGX_HIP_CHECK(hipStreamSynchronize(w.stream));
            ok = finite();
        }
        if (!ok) {
            set_error("back-substitution gamma = C^-T rho: non-finite result (the launch ran into its wait bound, EGX_PIPE_TIMEOUT_MS, "
                      "or rho is not finite)");
            return EGX_ERR_HIP;
        }
    }
    gp->gamma.assign(w.h_vec, w.h_vec + gp->n);
    gp->theta = thfull;
    gp->likelihood = res.lkh;
    gp->sigma2 = res.sigma2n * gp->y_std * gp->y_std;  // algorithm.rs:1048
    gp->beta = res.beta;
    gp->ft = res.ft;
    gp->ft_qr_r = res.ft_qr_r;
    gp->fit_coef = coef;
    gp->fit_hcols = hcols;
    gp->fitted = true;
    gp->fit_epoch++;
    gp->small_var_calls = 0;
    if (w.eval_stream == w.stream) {  // (a model that rode in another model's launch sequence has no timings of its own)
        float gpu = 0;
        hipEventElapsedTime(&gpu, w.ev[0], w.ev[3]);
        double host_ms = std::chrono::duration<double, std::milli>(ft.t1 - ft.t0).count() - gpu;
        if (host_ms < 0) host_ms = 0;
        record_timings(gp, w, host_ms, std::chrono::duration<double, std::milli>(t2 - ft.t1).count());
    }
    return EGX_SUCCESS;
}
static int finalize_tail(egx_gp *gp, const std::vector<double> &coef, int hcols, const std::vector<double> &thfull) {
    FinalizeTail ft;
    EGX_RC(finalize_tail_enqueue(gp, coef, hcols, ft));
    return finalize_tail_complete(gp, coef, hcols, thfull, ft);
}
int do_finalize(egx_gp *gp, const double *theta, int64_t theta_len) {
    std::vector<double> coef, thfull;
    int hcols = 1;
    EGX_RC(make_coef(gp, theta, theta_len, coef, hcols, &thfull));
    if (has_nan(theta, theta_len)) {
        set_error("theta contains NaN");
        return EGX_ERR_INVALID_VALUE;
    }
    gp->fitted = false;
    EGX_RC(enqueue_eval(gp, gp->ws[0], coef, hcols));
    EGX_RC(prelaunch_block_inverse(gp, gp->ws[0], gp->ws[0].stream));
    return finalize_tail(gp, coef, hcols, thfull);
}
// Candidates pipelined over the handle's workspaces (caller holds gp->mu exclusively and has set the device).
// The workspaces form SLOTS of gp->lockstep consecutive ones; the candidates of a slot are factored in lock-step by one
// launch sequence (enqueue_eval_group), different slots run on their own stream sets, so that the exposed serial parts
// of one slot (first group's chain, the last ~3000 columns) overlap the trailing updates of the other.  A slot whose
// candidates have been read back takes the next ones the source hands out.
int likelihood_batch_core(egx_gp *gp, const double *thetas, int64_t k, int64_t theta_len, double *lkh, int32_t *status,
                          CandidateSource *src, char *evaluated) {
    struct Sequential final : CandidateSource {
        int64_t k, next = 0;
        explicit Sequential(int64_t k_) : k(k_) {}
        int pull(int want, int64_t *out) override {
            int got = 0;
            while (got < want && next < k) out[got++] = next++;
            return got;
        }
    } seq(k);
    if (!src) src = &seq;
    // a fitted model keeps its factor (workspace 0) as long as another workspace exists
    const int ws_lo = (gp->fitted && gp->ws.size() > 1) ? 1 : 0;
    const int nws = (int)gp->ws.size() - ws_lo;
    const int B = gp->lockstep < nws ? (gp->lockstep < 1 ? 1 : gp->lockstep) : nws;
    const int nslots = (nws + B - 1) / B;  // the last slot may hold fewer workspaces (11 starts: 4 + 4 + 3)
    auto slot_cap = [&](int i) { return std::min(B, nws - i * B); };
    std::vector<std::vector<int64_t>> cand(nslots);  // candidates in flight on slot i (workspaces ws_lo + i B ...)
    bool exhausted = false;
    int busy = 0;
    // next candidate with a usable theta (NaN thetas are answered at once, algorithm.rs:885-891)
    auto next_valid = [&](int64_t &c, std::vector<double> &coef, int &hcols) -> int {
        while (!exhausted) {
            if (src->pull(1, &c) == 0) {
                exhausted = true;
                break;
            }
            if (c < 0 || c >= k) {
                set_error("likelihood batch: the candidate source returned an index out of range");
                return EGX_ERR_INVALID_VALUE;
            }
            const double *th = thetas + c * theta_len;
            EGX_RC(make_coef(gp, th, theta_len, coef, hcols, nullptr));
            if (evaluated) evaluated[c] = 1;
            if (!has_nan(th, theta_len)) return EGX_SUCCESS;
            lkh[c] = -std::numeric_limits<double>::infinity();
            status[c] = EGX_STATUS_NAN_THETA;
        }
        c = -1;
        return EGX_SUCCESS;
    };
    auto run = [&]() -> int {
        std::vector<std::vector<double>> coefs(B);
        for (int i = 0;; i = (i + 1) % nslots) {
            const int w0 = ws_lo + i * B;
            if (!cand[i].empty()) {
                for (size_t j = 0; j < cand[i].size(); j++) {
                    EvalResult res;
                    EGX_RC(finish_eval(gp, gp->ws[w0 + (int)j], res, 0));
                    lkh[cand[i][j]] = res.lkh;
                    status[cand[i][j]] = res.status;
                }
                if (w0 == 0) record_timings(gp, gp->ws[0], 0.0, 0.0);
                cand[i].clear();
                busy--;
            }
            int hcols = 1;
            while (!exhausted && (int)cand[i].size() < slot_cap(i)) {
                int64_t c;
                EGX_RC(next_valid(c, coefs[cand[i].size()], hcols));
                if (c >= 0) cand[i].push_back(c);
            }
            if (!cand[i].empty()) {
                if (w0 == 0) gp->fitted = false;
                EGX_RC(enqueue_eval_group(gp, w0, (int)cand[i].size(), coefs.data(), hcols));
                busy++;
            }
            if (exhausted && busy == 0) return EGX_SUCCESS;
        }
    };
    const int rc = run();
    if (rc) {  // leave no work behind that still writes into the workspaces
        const std::string msg = last_error_string();
        for (int i = 0; i < nslots; i++)
            if (!cand[i].empty()) (void)hipStreamSynchronize(gp->ws[ws_lo + i * B].stream);
        (void)hipGetLastError();
        set_error(msg);
    }
    return rc;
}
}  // namespace egx

// =================================================================================================
// C ABI
// =================================================================================================
// Lock-step width of a handle with nws workspaces (egx_gp_set_lockstep changes it).  Large matrices: 8 wide from 16
// workspaces on, else 4; up to 12 for n_pad <= 4096, where an evaluation is bound by launch latency and the chain (a tuned
// fit's 11 COBYLA starts are then ONE launch sequence per round).  Round 3 (right-looking, n = 16384; in flight / width ->
// fits/s): 12 / 4 40.7, 24 / 6 40.8, 24 / 8 41.3, 32 / 8 40.8, 36 / 12 41.2 -- three groups in flight were the constant.  Round 4
// (handles of that size with width >= 8 factor left-looking, launch_potrf): 8 / 8 40.7, 16 / 8 42.2, 24 / 8 41.9, 32 / 16 42.2,
// 36 / 12 41.2 -- TWO groups in flight (profiles/r04_run6_left_looking_in_flight_and_width.txt).
// Round 6 (profiles/r06_lockstep_one_or_two_slots_ab.txt): below the left-looking sizes ONE slot of up to 12 is the robust choice --
// n = 8192: 12 workspaces 267 likelihoods/s as one slot, 240 as three slots of four (the default until then), 8 workspaces 260 / 240;
// n = 12288, 8 workspaces: 86.0 / 82.5; what two slots in flight gain or lose (-9 ... +5 %) is which of their streams share a
// hardware queue.
static int default_lockstep(int nws, int n_pad) {
    int ls = n_pad < 14336 ? 12 : (nws >= 16 ? 8 : 4);
    return ls < 1 ? 1 : (ls > nws ? nws : ls);
}

extern "C" {

int32_t egx_abi_version(void) { return EGX_GP_ABI_VERSION; }
const char *egx_last_error(void) { return g_last_error.c_str(); }

int32_t egx_device_count(void) {
    int c = 0;
    if (hipGetDeviceCount(&c) != hipSuccess) return 0;
    return c;
}

int64_t egx_trim(void) {
    const size_t bytes = pool_trim(-1) + pipe_release_plans();
    destroy_idle_streams(-1);  // (after the pool: its workspaces' streams were handed over when they were pooled / freed)
    return (int64_t)bytes;
}

void egx_chain_stats(int64_t *aborted, int64_t *retried) {
    if (aborted) *aborted = g_chain_aborts.load();
    if (retried) *retried = g_chain_retries.load();
}

int32_t egx_set_tuning(const char *knob, int32_t value, int32_t *previous) {
    if (!knob) {
        set_error("egx_set_tuning: NULL knob name");
        return EGX_ERR_INVALID_VALUE;
    }
    int old = set_knob(knob, value);
    if (old == -2147483647 - 1) old = pipe_set_knob(knob, value);  // the chain kernel's knobs (kernels_pipe.hip)
    if (old == -2147483647 - 1 && std::string(knob) == "pipe_retry") old = g_pipe_retry.exchange(value);
    if (old == -2147483647 - 1) {
        set_error(std::string("egx_set_tuning: unknown knob '") + knob + "'");
        return EGX_ERR_INVALID_VALUE;
    }
    if (previous) *previous = old;
    return EGX_SUCCESS;
}

void egx_pool_stats(int64_t *cached_bytes, int64_t *hits, int64_t *misses) {
    std::lock_guard<std::mutex> lock(g_pool_mu);
    if (cached_bytes) {
        size_t b = 0;
        for (const auto &e : g_pool) b += e.bytes;
        for (const auto &e : g_slab_pool) b += e.bytes();
        *cached_bytes = (int64_t)b;
    }
    if (hits) *hits = g_pool_hits;
    if (misses) *misses = g_pool_misses;
}

void egx_gp_config_default(egx_gp_config *cfg) {
    if (!cfg) return;
    cfg->corr = EGX_CORR_SQUARED_EXPONENTIAL;  // Kriging = ConstantMean + SquaredExponentialCorr, algorithm.rs:200-207
    cfg->mean = EGX_MEAN_CONSTANT;
    cfg->nugget = 100.0 * std::numeric_limits<double>::epsilon();  // parameters.rs:118
    cfg->device = -1;
    cfg->n_workspaces = 1;
    cfg->w_star = nullptr;
    cfg->kpls_dim = 0;
}

int32_t egx_normalize(const double *x, int64_t n, int64_t d, double *xnorm, double *mean, double *std_) {
    if (!x || !xnorm || !mean || !std_ || n < 2 || d < 1) {
        set_error("egx_normalize: need n >= 2, d >= 1 and non-null arrays");
        return EGX_ERR_INVALID_VALUE;
    }
    hm::normalize(x, n, d, xnorm, mean, std_);
    return EGX_SUCCESS;
}

int64_t egx_regression_ncols(int32_t mean, int64_t d) { return hm::regression_ncols(mean, d); }

int32_t egx_regression_basis(int32_t mean, const double *x, int64_t n, int64_t d, double *f) {
    const int64_t p = hm::regression_ncols(mean, d);
    if (p < 0 || !x || !f || n < 0 || d < 1) {
        set_error("egx_regression_basis: bad arguments");
        return EGX_ERR_INVALID_VALUE;
    }
    for (int64_t i = 0; i < n; i++) hm::regression_row(mean, x + i * d, d, f + i * p);
    return EGX_SUCCESS;
}

// set by egx_gp_create_group around the creation of its members: the member takes slot `slot` of these slabs
struct GroupCtx {
    std::shared_ptr<GroupSlabs> slabs;
    int slot = 0;
    int64_t sync_off = 0;
};
static thread_local GroupCtx *t_group_ctx = nullptr;

int32_t egx_gp_create(const egx_gp_config *cfg_in, const double *x, const double *y, int64_t n, int64_t d,
                      egx_gp **out) {
    if (!out) {
        set_error("out handle pointer is NULL");
        return EGX_ERR_INVALID_VALUE;
    }
    *out = nullptr;
    egx_gp_config cfg;
    if (cfg_in) cfg = *cfg_in; else egx_gp_config_default(&cfg);
    if (!x || !y) {
        set_error("x / y is NULL");
        return EGX_ERR_INVALID_VALUE;
    }
    if (n < 2) {
        set_error("need at least 2 training points (sample standard deviation, utils.rs:48), got " + std::to_string(n));
        return EGX_ERR_INVALID_VALUE;
    }
    if (d < 1 || d > 65535) {
        set_error("input dimension must be between 1 and 65535, got " + std::to_string(d));
        return EGX_ERR_INVALID_VALUE;
    }
    if (cfg.corr < 0 || cfg.corr > 3 || cfg.mean < 0 || cfg.mean > 2) {
        set_error("unknown correlation / regression model");
        return EGX_ERR_INVALID_VALUE;
    }
    if (!std::isfinite(cfg.nugget)) {  // the reference does not constrain the nugget's sign
        set_error("nugget must be finite");
        return EGX_ERR_INVALID_VALUE;
    }
    if (cfg.w_star) {
        if (cfg.kpls_dim == 0) {  // parameters.rs:290-294
            set_error("`kpls_dim` canot be 0!");
            return EGX_ERR_INVALID_VALUE;
        }
        if (cfg.kpls_dim < 0 || cfg.kpls_dim > d) {  // algorithm.rs:798-807
            set_error("Dimension reduction " + std::to_string(cfg.kpls_dim) +
                      " should be smaller than actual training input dimensions " + std::to_string(d));
            return EGX_ERR_INVALID_VALUE;
        }
    }
    for (int64_t i = 0; i < n * d; i++)
        if (!std::isfinite(x[i])) {
            set_error("x contains non-finite values");
            return EGX_ERR_INVALID_VALUE;
        }
    for (int64_t i = 0; i < n; i++)
        if (!std::isfinite(y[i])) {
            set_error("y contains non-finite values");
            return EGX_ERR_INVALID_VALUE;
        }
    const int64_t p = hm::regression_ncols(cfg.mean, d);
    if (p >= n) {
        // the reference would produce a rank-deficient GLS; refuse explicitly
        set_error("need more training points than regression basis columns");
        return EGX_ERR_INVALID_VALUE;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        set_error("no HIP device: libegx_gp_hip has no CPU fallback (needs an MI355X / gfx950 GPU)");
        return EGX_ERR_NO_DEVICE;
    }
    int dev = cfg.device;
    if (dev < 0) {
        if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    }
    if (dev >= ndev) {
        set_error("device ordinal out of range");
        return EGX_ERR_INVALID_VALUE;
    }
    egx_gp *gp = new egx_gp();
    gp->device = dev;
    gp->n = (int)n;
    gp->d = (int)d;
    gp->p = (int)p;
    gp->corr = cfg.corr;
    gp->mean = cfg.mean;
    gp->nugget = cfg.nugget;
    gp->has_w = cfg.w_star != nullptr;
    gp->h = gp->has_w ? (int)cfg.kpls_dim : (int)d;
    if (gp->has_w) gp->w_star.assign(cfg.w_star, cfg.w_star + d * cfg.kpls_dim);
    gp->q = gp->p + 1;
    {
        const char *e = std::getenv("EGX_GLS_DEVICE");  // 0 = always the host Householder route (A/B, debugging)
        gp->gls_device = gp->p >= 2 && !(e && e[0] == '0');
    }
    // 256-column granularity lets every trailing update of a large fit use the 128x256 tile (N % 256 == 0)
    gp->n_pad = (int)round_up(n, n >= 4096 ? kNB : kTile);
    gp->rhs_pad = (int)round_up(gp->q, kRhsPad);
    gp->m_tot = gp->n_pad + gp->rhs_pad;
    gp->ld = gp->n_pad;
    gp->x_raw.assign(x, x + n * d);
    gp->y_raw.assign(y, y + n);
    gp->xnorm.resize(n * d);
    gp->x_mean.resize(d);
    gp->x_std.resize(d);
    gp->ynorm.resize(n);
    hm::normalize(x, n, d, gp->xnorm.data(), gp->x_mean.data(), gp->x_std.data());  // algorithm.rs:840
    hm::normalize(y, n, 1, gp->ynorm.data(), &gp->y_mean, &gp->y_std);               // algorithm.rs:841
    gp->F.resize(n * p);
    for (int64_t i = 0; i < n; i++) hm::regression_row(gp->mean, &gp->xnorm[i * d], d, &gp->F[i * p]);  // :866

    auto fail = [&](int rc) {
        egx_gp_destroy(gp);
        return rc;
    };
    if (hipSetDevice(dev) != hipSuccess) {
        set_error("hipSetDevice failed");
        return fail(EGX_ERR_HIP);
    }
    int rc = chol_init();
    if (rc) return fail(rc);
    // k-major, zero padded copies
    std::vector<double> xT((size_t)d * gp->n_pad, 0.0), rhsT((size_t)gp->q * gp->n_pad, 0.0);
    for (int64_t i0 = 0; i0 < n; i0 += 64) {  // 64-point blocks: the d write streams stay within a few cache lines each
        const int64_t i1 = std::min<int64_t>(n, i0 + 64);
        for (int64_t j = 0; j < d; j++) {
            double *dst = &xT[(size_t)j * gp->n_pad];
            for (int64_t i = i0; i < i1; i++) dst[i] = gp->xnorm[i * d + j];
        }
    }
    for (int64_t l = 0; l < p; l++)
        for (int64_t i = 0; i < n; i++) rhsT[(size_t)l * gp->n_pad + i] = gp->F[i * p + l];
    for (int64_t i = 0; i < n; i++) rhsT[(size_t)p * gp->n_pad + i] = gp->ynorm[i];
#define EGX_HIPF(expr)                                                            \
    do {                                                                          \
        hipError_t _e = (expr);                                                   \
        if (_e != hipSuccess) {                                                   \
            set_error(std::string(#expr) + ": " + hipGetErrorString(_e));         \
            (void)hipGetLastError();                                              \
            return fail(EGX_ERR_HIP);                                             \
        }                                                                         \
    } while (0)
    int nws = cfg.n_workspaces < 1 ? 1 : cfg.n_workspaces;
    // all workspaces' matrices / tile inverses / failure flags at fixed strides in one allocation each (lock-step batches)
    gp->stride_M = (int64_t)gp->m_tot * gp->ld;
    gp->stride_D = round_up((int64_t)dinv_doubles(gp->n_pad), 64);
    gp->stride_S = (int64_t)pipe_sync_ints(gp->n_pad, gp->m_tot);
    gp->sync_off = round_up(nws, 64);
    if (t_group_ctx) {
        // member of a group (egx_gp_create_group): ONE workspace whose matrix, tile inverses, flag and hand-off words are slot
        // `slot` of the group's slabs -- the members' matrices sit at the strides a lock-step launch needs
        GroupCtx &g = *t_group_ctx;
        nws = 1;
        gp->group = g.slabs;
        gp->group_slot = g.slot;
        // (a destroyed member of a group of this shape left its workspace -- streams, events, pinned buffers -- and its
        //  training-set buffers in the pool: adopt them; the slab views are this group's)
        const bool pooled = pool_take(gp, 1);
        gp->slab_M = g.slabs->M + (int64_t)g.slot * gp->stride_M;
        gp->slab_D = g.slabs->D + (int64_t)g.slot * gp->stride_D;
        gp->slab_I = g.slabs->I + g.slot;
        gp->sync_off = g.sync_off + (int64_t)g.slot * gp->stride_S - g.slot;  // dev_sync(gp, 0) = the slot's words
        if (pooled) {
            Workspace &w = gp->ws[0];
            w.M = gp->slab_M, w.dinv = gp->slab_D, w.d_info = gp->slab_I;
            EGX_HIPF(take_streams(gp->device, w));  // (pooled workspaces are kept without streams)
        } else {
            EGX_HIPF(dev_malloc(&gp->d_xT, sizeof(double) * 2 * xT.size()));
            EGX_HIPF(dev_malloc(&gp->d_rhsT, sizeof(double) * rhsT.size()));
            EGX_HIPF(dev_malloc(&gp->d_gamma, sizeof(double) * gp->n_pad));
            EGX_HIPF(dev_malloc(&gp->d_fit_coef, sizeof(double) * ((size_t)d * (gp->has_w ? gp->h : 1) + 2 * (size_t)d)));
            gp->ws.resize(1);
            rc = alloc_workspace(gp, gp->ws[0], 0);
            if (rc) return fail(rc);
        }
    } else if (pool_take(gp, nws)) {  // a destroyed handle of this shape left its resources behind; its streams went to the free list
        for (auto &w : gp->ws) EGX_HIPF(take_streams(gp->device, w));
    } else {  // allocate
        EGX_HIPF(dev_malloc(&gp->d_xT, sizeof(double) * 2 * xT.size()));  // + dev_xs_fit()
        EGX_HIPF(dev_malloc(&gp->d_rhsT, sizeof(double) * rhsT.size()));
        EGX_HIPF(dev_malloc(&gp->d_gamma, sizeof(double) * gp->n_pad));
        EGX_HIPF(dev_malloc(&gp->d_fit_coef, sizeof(double) * ((size_t)d * (gp->has_w ? gp->h : 1) + 2 * (size_t)d)));
        EGX_HIPF(dev_malloc(&gp->slab_M, sizeof(double) * (size_t)gp->stride_M * nws));
        EGX_HIPF(dev_malloc(&gp->slab_D, sizeof(double) * (size_t)gp->stride_D * nws));
        EGX_HIPF(dev_malloc(&gp->slab_I, sizeof(int) * slab_I_ints(gp, nws)));
        gp->ws.resize(nws);
        for (int i = 0; i < nws; i++) {
            rc = alloc_workspace(gp, gp->ws[i], i);
            if (rc) return fail(rc);
        }
    }
    EGX_HIPF(hipMemcpy(gp->d_xT, xT.data(), sizeof(double) * xT.size(), hipMemcpyHostToDevice));
    EGX_HIPF(hipMemcpy(gp->d_rhsT, rhsT.data(), sizeof(double) * rhsT.size(), hipMemcpyHostToDevice));
    {  // normalisation parameters for the query-side kernel (gp_predict.hip upload_queries)
        std::vector<double> par(gp->x_mean);
        par.insert(par.end(), gp->x_std.begin(), gp->x_std.end());
        EGX_HIPF(hipMemcpy(dev_xnorm(gp), par.data(), sizeof(double) * par.size(), hipMemcpyHostToDevice));
    }
    gp->lockstep = default_lockstep(nws, gp->n_pad);  // candidates of a likelihood batch factored in lock-step: egx_gp_set_lockstep
    gp->sched = schedule_for(gp->n_pad, gp->lockstep, nws);
    if (t_group_ctx) gp->sched.flow = gp->sched.flow_tail = 0;  // (members of a group are factored in lock-step: the flow launch takes one matrix)
    rc = pipe_prepare(gp->n_pad, gp->m_tot, gp->sched);  // the chain launches' task lists: not inside the first evaluation
    if (rc) return fail(rc);
    *out = gp;
    return EGX_SUCCESS;
}

void egx_gp_destroy(egx_gp *gp) {
    if (!gp) return;
    hipSetDevice(gp->device);
    // nothing of this handle may still run when its resources change hands
    for (auto &w : gp->ws) {
        if (w.stream) (void)hipStreamSynchronize(w.stream);
        if (w.lk.s2) (void)hipStreamSynchronize(w.lk.s2);
        if (w.lk.s3) (void)hipStreamSynchronize(w.lk.s3);
        if (w.inv_stream) (void)hipStreamSynchronize(w.inv_stream);
    }
    (void)hipGetLastError();
    if (gp->group) gp->slab_M = gp->slab_D = nullptr, gp->slab_I = nullptr;  // (views of the group's slabs: freed with the last member)
    pool_give(gp);
    if (gp->d_W) hipFree(gp->d_W);
    if (gp->d_neg_invkf) hipFree(gp->d_neg_invkf);
    for (double *q : {gp->sp_R, gp->sp_P, gp->sp_y, gp->sp_z, gp->sp_wt, gp->sp_out, gp->sp_xq})
        if (q) hipFree(q);
    if (gp->slab_W) hipFree(gp->slab_W);
    if (gp->d_wabs) hipFree(gp->d_wabs);
    delete gp;
}

int32_t egx_gp_dims(const egx_gp *gp, int64_t *n, int64_t *d, int64_t *p, int64_t *h) {
    if (!gp) {
        set_error("NULL handle");
        return EGX_ERR_INVALID_VALUE;
    }
    if (n) *n = gp->n;
    if (d) *d = gp->d;
    if (p) *p = gp->p;
    if (h) *h = gp->h;
    return EGX_SUCCESS;
}

int32_t egx_gp_get_training_data(const egx_gp *gp, double *x_out, double *y_out) {
    if (!gp) {
        set_error("NULL handle");
        return EGX_ERR_INVALID_VALUE;
    }
    if (x_out) std::memcpy(x_out, gp->x_raw.data(), sizeof(double) * gp->x_raw.size());
    if (y_out) std::memcpy(y_out, gp->y_raw.data(), sizeof(double) * gp->y_raw.size());
    return EGX_SUCCESS;
}

int32_t egx_gp_likelihood(egx_gp *gp, const double *theta, int64_t theta_len, double *lkh, int32_t *status) {
    if (!gp || !theta || !lkh || !status) {
        set_error("NULL argument");
        return EGX_ERR_INVALID_VALUE;
    }
    std::shared_lock<std::shared_mutex> lock(gp->mu);
    EGX_RC(set_device(gp));
    // take a free workspace.  Workspace 0 holds the factor of a FITTED model: while the model is fitted and other
    // workspaces exist, callers wait for one of those instead of overwriting the fit (a handle with a single workspace
    // has no choice: the evaluation un-fits it, as documented in the header)
    int wi = -1;
    {
        std::unique_lock<std::mutex> pl(gp->pool_mu);
        if (gp->ws_busy.size() != gp->ws.size()) gp->ws_busy.assign(gp->ws.size(), 0);
        for (;;) {
            const int lowest = (gp->fitted && gp->ws.size() > 1) ? 1 : 0;
            for (int i = (int)gp->ws.size() - 1; i >= lowest && wi < 0; i--)
                if (!gp->ws_busy[i]) wi = i;
            if (wi >= 0) break;
            gp->pool_cv.wait(pl);
        }
        gp->ws_busy[wi] = 1;
    }
    EvalResult res;
    const int rc = eval_one(gp, wi, theta, theta_len, res, false);
    {
        std::lock_guard<std::mutex> pl(gp->pool_mu);
        gp->ws_busy[wi] = 0;
    }
    gp->pool_cv.notify_one();
    if (rc) return rc;
    *lkh = res.lkh;
    *status = res.status;
    return EGX_SUCCESS;
}

int32_t egx_gp_likelihood_batch(egx_gp *gp, const double *thetas, int64_t k, int64_t theta_len, double *lkh,
                                int32_t *status) {
    if (!gp || (k > 0 && (!thetas || !lkh || !status)) || k < 0) {
        set_error("NULL argument");
        return EGX_ERR_INVALID_VALUE;
    }
    std::unique_lock<std::shared_mutex> lock(gp->mu);
    EGX_RC(set_device(gp));
    return likelihood_batch_core(gp, thetas, k, theta_len, lkh, status);
}

int32_t egx_gp_set_lockstep(egx_gp *gp, int32_t width) {
    if (!gp || width < 0) {
        set_error("egx_gp_set_lockstep: NULL handle or negative width");
        return EGX_ERR_INVALID_VALUE;
    }
    std::unique_lock<std::shared_mutex> lock(gp->mu);
    const int nws = (int)gp->ws.size();
    gp->lockstep = width == 0 ? default_lockstep(nws, gp->n_pad) : (width > nws ? nws : width);
    gp->sched = schedule_for(gp->n_pad, gp->lockstep, nws);
    if (gp->group) gp->sched.flow = gp->sched.flow_tail = 0;
    EGX_RC(set_device(gp));
    return pipe_prepare(gp->n_pad, gp->m_tot, gp->sched);
}

int32_t egx_gp_get_lockstep(const egx_gp *gp) { return gp ? gp->lockstep : 0; }

int32_t egx_gp_shrink(egx_gp *gp, int32_t n_keep) {
    if (!gp || n_keep < 1) {
        set_error("egx_gp_shrink: NULL handle or n_keep < 1");
        return EGX_ERR_INVALID_VALUE;
    }
    std::unique_lock<std::shared_mutex> lock(gp->mu);
    EGX_RC(set_device(gp));
    const int nws = (int)gp->ws.size();
    for (auto &w : gp->ws) {  // nothing of this handle may still run
        if (w.stream) EGX_HIP_CHECK(hipStreamSynchronize(w.stream));
        if (w.lk.s2) EGX_HIP_CHECK(hipStreamSynchronize(w.lk.s2));
        if (w.lk.s3) EGX_HIP_CHECK(hipStreamSynchronize(w.lk.s3));
        if (w.inv_stream) EGX_HIP_CHECK(hipStreamSynchronize(w.inv_stream));
    }
    if (gp->slab_W) {  // the theta-gradient's C^-T buffers are scratch between calls
        (void)hipFree(gp->slab_W);
        gp->slab_W = nullptr;
        gp->slab_W_count = 0;
    }
    if (n_keep >= nws) return EGX_SUCCESS;
    // the first n_keep workspaces sit at the front of the slabs: smaller slabs, one device-to-device copy each
    double *nM = nullptr, *nD = nullptr;
    int *nI = nullptr;
    if (dev_malloc(&nM, sizeof(double) * (size_t)gp->stride_M * n_keep) != hipSuccess ||
        dev_malloc(&nD, sizeof(double) * (size_t)gp->stride_D * n_keep) != hipSuccess ||
        dev_malloc(&nI, sizeof(int) * slab_I_ints(gp, n_keep)) != hipSuccess) {
        (void)hipGetLastError();
        if (nM) (void)hipFree(nM);
        if (nD) (void)hipFree(nD);
        if (nI) (void)hipFree(nI);
        set_error("egx_gp_shrink: no memory for the smaller slabs (the handle is unchanged)");
        return EGX_ERR_HIP;
    }
    hipError_t ce = hipMemcpy(nM, gp->slab_M, sizeof(double) * (size_t)gp->stride_M * n_keep, hipMemcpyDeviceToDevice);
    if (ce == hipSuccess) ce = hipMemcpy(nD, gp->slab_D, sizeof(double) * (size_t)gp->stride_D * n_keep, hipMemcpyDeviceToDevice);
    if (ce == hipSuccess) ce = hipMemcpy(nI, gp->slab_I, sizeof(int) * (size_t)n_keep, hipMemcpyDeviceToDevice);
    if (ce != hipSuccess) {  // the handle keeps its slabs, the new ones go back
        (void)hipFree(nM);
        (void)hipFree(nD);
        (void)hipFree(nI);
        set_error(std::string("egx_gp_shrink: copying the kept workspaces failed (the handle is unchanged): ") + hipGetErrorString(ce));
        return EGX_ERR_HIP;
    }
    (void)hipFree(gp->slab_M);
    (void)hipFree(gp->slab_D);
    (void)hipFree(gp->slab_I);
    gp->slab_M = nM;
    gp->slab_D = nD;
    gp->slab_I = nI;
    gp->sync_off = round_up(n_keep, 64);  // (the hand-off words need no copy: every factorisation zeroes its own)
    for (int i = n_keep; i < nws; i++) free_workspace(gp->ws[i], gp->device);
    gp->ws.resize((size_t)n_keep);
    for (int i = 0; i < n_keep; i++) {
        gp->ws[i].M = gp->slab_M + (int64_t)i * gp->stride_M;
        gp->ws[i].dinv = gp->slab_D + (int64_t)i * gp->stride_D;
        gp->ws[i].d_info = gp->slab_I + i;
    }
    {
        std::lock_guard<std::mutex> pl(gp->pool_mu);
        gp->ws_busy.assign((size_t)n_keep, 0);
    }
    // the schedule stays what it was (a shrunk handle must keep giving the bits it gave before); the width is
    // capped by the workspaces that are left
    if (gp->lockstep > n_keep) gp->lockstep = n_keep;
    return EGX_SUCCESS;
}

int32_t egx_gp_get_schedule(const egx_gp *gp, int32_t *out, int32_t out_len) {
    if (!gp || !out || out_len < 6) {
        set_error("egx_gp_get_schedule: NULL argument or fewer than 6 slots");
        return EGX_ERR_INVALID_VALUE;
    }
    out[0] = gp->sched.left, out[1] = gp->sched.w_left, out[2] = gp->sched.pipe, out[3] = gp->sched.whole;
    out[4] = gp->sched.group_panels, out[5] = gp->lockstep;
    for (int32_t i = 6; i < out_len; i++) out[i] = 0;  // (reserved)
    if (out_len >= 7) out[6] = gp->sched.flow ? 1 : (gp->sched.flow_tail ? 2 : 0);
    return EGX_SUCCESS;
}

int32_t egx_gp_finalize(egx_gp *gp, const double *theta, int64_t theta_len) {
    if (!gp || !theta) {
        set_error("NULL argument");
        return EGX_ERR_INVALID_VALUE;
    }
    std::unique_lock<std::shared_mutex> lock(gp->mu);
    EGX_RC(set_device(gp));
    return do_finalize(gp, theta, theta_len);
}

// ---------------------------------------------------------------------------------------------
// Lock-step across MODELS (round 5): the reference's callers multiply independent models, not only candidates of one model --
// the expert loop of egobox-moe (crates/moe/src/algorithm.rs:167-177: one GP per cluster) and EGO's objective + constraint
// surrogates (crates/ego/src/solver/solver_impl.rs:370-391).  Models of ONE shape created into one group of slabs are
// factored by one launch sequence: the kernels' batch dimension never required the matrices to share a training set.
// ---------------------------------------------------------------------------------------------
int32_t egx_gp_create_group(const egx_gp_config *cfg_in, const double *x, const double *y, int64_t n, int64_t d, int32_t k,
                            egx_gp **out) {
    if (!out || k < 1 || k > 256) {
        set_error("egx_gp_create_group: NULL output or a member count outside 1..256");
        return EGX_ERR_INVALID_VALUE;
    }
    for (int32_t j = 0; j < k; j++) out[j] = nullptr;
    if (!x || !y || n < 2 || d < 1) {
        set_error("egx_gp_create_group: bad training sets");
        return EGX_ERR_INVALID_VALUE;
    }
    egx_gp_config cfg;
    if (cfg_in) cfg = *cfg_in; else egx_gp_config_default(&cfg);
    cfg.n_workspaces = 1;
    if (egx_device_count() <= 0) {
        set_error("no HIP device: libegx_gp_hip has no CPU fallback (needs an MI355X / gfx950 GPU)");
        return EGX_ERR_NO_DEVICE;
    }
    int dev = cfg.device;
    if (dev < 0 && hipGetDevice(&dev) != hipSuccess) dev = 0;
    EGX_HIP_CHECK(hipSetDevice(dev));
    // the members' geometry (the formulas of egx_gp_create)
    if (cfg.mean < 0 || cfg.mean > 2) {
        set_error("unknown regression model");
        return EGX_ERR_INVALID_VALUE;
    }
    const int64_t p = hm::regression_ncols(cfg.mean, d);
    const int n_pad = (int)round_up(n, n >= 4096 ? kNB : kTile);
    const int m_tot = n_pad + (int)round_up(p + 1, kRhsPad);
    const int64_t sM = (int64_t)m_tot * n_pad, sD = round_up((int64_t)dinv_doubles(n_pad), 64), sS = (int64_t)pipe_sync_ints(n_pad, m_tot);
    GroupCtx ctx;
    ctx.slabs = std::make_shared<GroupSlabs>();
    ctx.slabs->device = dev;
    ctx.slabs->k = k;
    ctx.sync_off = round_up(k, 64);
    ctx.slabs->bytes_M = sizeof(double) * (size_t)sM * k;
    ctx.slabs->bytes_D = sizeof(double) * (size_t)sD * k;
    ctx.slabs->bytes_I = sizeof(int) * ((size_t)ctx.sync_off + (size_t)sS * k);
    EGX_RC(group_slabs_take(*ctx.slabs));
    int rc = EGX_SUCCESS;
    for (int32_t j = 0; j < k && rc == EGX_SUCCESS; j++) {
        ctx.slot = j;
        t_group_ctx = &ctx;
        rc = egx_gp_create(&cfg, x + (size_t)j * n * d, y + (size_t)j * n, n, d, &out[j]);
        t_group_ctx = nullptr;
    }
    if (rc != EGX_SUCCESS)
        for (int32_t j = 0; j < k; j++) {
            if (out[j]) egx_gp_destroy(out[j]);
            out[j] = nullptr;
        }
    return rc;
}

// gps[0..k): distinct fitted-or-not models; runs of members of one group in consecutive slots are evaluated in lock-step
// (up to `cap` per launch sequence), everything else one by one: the results do not depend on which (a matrix gets the
// same arithmetic alone and in a batch)
static int multi_run_len(egx_gp *const *gps, int k, int i) {
    if (!gps[i]->group) return 1;
    int cap = 12;
    if (gps[i]->sched.whole) {  // (every diagonal block of a whole-factorisation launch has a workgroup of its own)
        const int np = (gps[i]->n_pad + kNB - 1) / kNB;
        cap = std::max(1, std::min(cap, 128 / np));
    }
    int len = 1;
    while (i + len < k && len < cap && gps[i + len]->group == gps[i]->group && gps[i + len]->group_slot == gps[i]->group_slot + len &&
           gps[i + len]->corr == gps[i]->corr && gps[i + len]->has_w == gps[i]->has_w)
        len++;
    return len;
}
// the exclusive locks of k DISTINCT handles, in address order (two such calls cannot deadlock)
static int multi_lock(egx_gp *const *gps, int32_t k, std::vector<std::unique_lock<std::shared_mutex>> &locks) {
    if (!gps || k < 1) {
        set_error("NULL argument");
        return EGX_ERR_INVALID_VALUE;
    }
    std::vector<egx_gp *> order(gps, gps + k);
    std::sort(order.begin(), order.end());
    if (std::adjacent_find(order.begin(), order.end()) != order.end() || !order[0]) {
        set_error("the models of a multi-model call must be distinct handles");
        return EGX_ERR_INVALID_VALUE;
    }
    for (egx_gp *g : order) locks.emplace_back(g->mu);
    return EGX_SUCCESS;
}
static int multi_eval_locked(egx_gp *const *gps, int32_t k, const double *thetas, int64_t theta_len, bool finalize, double *lkh,
                             int32_t *status);
static int multi_eval(egx_gp *const *gps, int32_t k, const double *thetas, int64_t theta_len, bool finalize, double *lkh,
                      int32_t *status) {
    if (!thetas) {
        set_error("NULL argument");
        return EGX_ERR_INVALID_VALUE;
    }
    std::vector<std::unique_lock<std::shared_mutex>> locks;
    EGX_RC(multi_lock(gps, k, locks));
    return multi_eval_locked(gps, k, thetas, theta_len, finalize, lkh, status);
}
// (the caller holds every member's lock)
static int multi_eval_locked(egx_gp *const *gps, int32_t k, const double *thetas, int64_t theta_len, bool finalize, double *lkh,
                             int32_t *status) {
    int first_rc = EGX_SUCCESS;
    for (int i = 0; i < k;) {
        const int len = multi_run_len(gps, k, i);
        EGX_RC(set_device(gps[i]));
        std::vector<std::vector<double>> coefs((size_t)len), thfull((size_t)len);
        int hcols = 1;
        bool nan = false;
        for (int j = 0; j < len; j++) {
            const double *th = thetas + (size_t)(i + j) * theta_len;
            EGX_RC(make_coef(gps[i + j], th, theta_len, coefs[(size_t)j], hcols, &thfull[(size_t)j]));
            nan = nan || has_nan(th, theta_len);
        }
        if (nan) {
            set_error("theta contains NaN");
            return EGX_ERR_INVALID_VALUE;
        }
        // (every member evaluates on its workspace 0 and is UN-FITTED by the call -- also by egx_gp_likelihood_multi, unlike
        //  egx_gp_likelihood, which keeps a fitted handle with spare workspaces fitted: include/egx_gp.h says so)
        for (int j = 0; j < len; j++) gps[i + j]->fitted = false;
        // An error between here and the members' host halves must not leave work in flight on the lead's stream that writes into
        // the other members' pinned buffers, nor their eval_stream pointing at a stream of a handle that may be destroyed first:
        // drain the lead's stream and hand every member its own stream back before returning.
        auto bail = [&](int rc_in) {
            const std::string msg = last_error_string();
            (void)hipStreamSynchronize(gps[i]->ws[0].stream);
            for (int j = 0; j < len; j++) gps[i + j]->ws[0].eval_stream = gps[i + j]->ws[0].stream;
            (void)hipGetLastError();
            set_error(msg);
            return rc_in;
        };
#define EGX_RC_BAIL(call)                    \
    do {                                     \
        const int _rc = (call);              \
        if (_rc) return bail(_rc);           \
    } while (0)
        if (len > 1) EGX_RC_BAIL(enqueue_eval_members(gps + i, len, coefs.data(), hcols));
        else EGX_RC_BAIL(enqueue_eval(gps[i], gps[i]->ws[0], coefs[0], hcols));
        if (finalize && len > 1 && len <= SolveBatchPtrs::kMax) {
            // the members' inverse blocks (for gamma's back-substitution) right behind the evaluation, in lock-step: they only need
            // the factors and run while the host waits for the read-back and does the models' GLS
            SolveBatchPtrs ib;
            for (int j = 0; j < len; j++) {
                Workspace &w = gps[i + j]->ws[0];
                EGX_RC_BAIL(ensure_block_inverse_buffer(gps[i + j], w));
                ib.M[j] = w.M, ib.dinv[j] = w.dinv, ib.dW[j] = w.dW, ib.rhs[j] = w.d_rhs, ib.vec[j] = w.d_vec;
            }
            EGX_RC_BAIL(launch_block_inverse_batch(gps[i]->ws[0].eval_stream, ib, len, gps[i]->ld, gps[i]->n_pad));
        } else if (finalize) {
            for (int j = 0; j < len; j++) EGX_RC_BAIL(prelaunch_block_inverse(gps[i + j], gps[i + j]->ws[0], gps[i + j]->ws[0].stream));
        }
        std::vector<FinalizeTail> tails((size_t)(finalize ? len : 0));
        std::vector<int> trc((size_t)len, EGX_SUCCESS);
        std::vector<std::string> terr((size_t)len);
        if (finalize && len > 1 && len <= SolveBatchPtrs::kMax) {
            EGX_RC_BAIL(finalize_tails_lockstep(gps + i, len, coefs.data(), hcols, tails.data(), trc.data(), terr.data()));
#undef EGX_RC_BAIL
        } else if (finalize) {
            for (int j = 0; j < len; j++) {
                trc[(size_t)j] = finalize_tail_enqueue(gps[i + j], coefs[(size_t)j], hcols, tails[(size_t)j]);
                if (trc[(size_t)j] != EGX_SUCCESS) terr[(size_t)j] = last_error_string();
            }
        }
        // (likelihoods only: the members' host halves side by side as well, see finalize_tails_lockstep)
        std::vector<EvalResult> lres((size_t)(finalize ? 0 : len));
        std::vector<int> lrc((size_t)len, EGX_SUCCESS);
        if (!finalize) {
            auto host_half = [&](int j) {
                lrc[(size_t)j] = finish_eval(gps[i + j], gps[i + j]->ws[0], lres[(size_t)j], 0);
                if (lrc[(size_t)j] != EGX_SUCCESS) terr[(size_t)j] = last_error_string();
            };
            if (len > 1 && !gps[i]->gls_device) {
                std::vector<std::thread> pool;
                for (int j = 1; j < len; j++)
                    pool.emplace_back([&, j] {
                        if (hipSetDevice(gps[i + j]->device) == hipSuccess) host_half(j);
                        else lrc[(size_t)j] = EGX_ERR_HIP, terr[(size_t)j] = "hipSetDevice failed in a likelihood worker";
                    });
                host_half(0);
                for (auto &t : pool) t.join();
            } else {
                for (int j = 0; j < len; j++) host_half(j);
            }
        }
        for (int j = 0; j < len; j++) {
            egx_gp *g = gps[i + j];
            int rc;
            if (finalize) {
                rc = trc[(size_t)j];
                if (rc == EGX_SUCCESS) rc = finalize_tail_complete(g, coefs[(size_t)j], hcols, thfull[(size_t)j], tails[(size_t)j]);
                else {
                    if (!terr[(size_t)j].empty()) set_error(terr[(size_t)j]);
                    (void)hipStreamSynchronize(g->ws[0].stream);
                }
            } else {
                rc = lrc[(size_t)j];
                if (rc == EGX_SUCCESS) {
                    if (lkh) lkh[i + j] = lres[(size_t)j].lkh;
                    if (status) status[i + j] = lres[(size_t)j].status;
                } else if (!terr[(size_t)j].empty()) {
                    set_error(terr[(size_t)j]);
                }
            }
            if (rc != EGX_SUCCESS && first_rc == EGX_SUCCESS) first_rc = rc;  // (the others still finish: their streams are drained)
        }
        i += len;
    }
    return first_rc;
}
int32_t egx_gp_finalize_multi(egx_gp *const *gps, int32_t k, const double *thetas, int64_t theta_len) {
    return multi_eval(gps, k, thetas, theta_len, true, nullptr, nullptr);
}
int32_t egx_gp_likelihood_multi(egx_gp *const *gps, int32_t k, const double *thetas, int64_t theta_len, double *lkh,
                                int32_t *status) {
    if (!lkh || !status) {
        set_error("NULL output");
        return EGX_ERR_INVALID_VALUE;
    }
    return multi_eval(gps, k, thetas, theta_len, false, lkh, status);
}

// ThetaTuning::Full across MODELS: the tuned fit of every expert of a mixture (crates/moe/src/algorithm.rs:167-177 -> :209-262 ->
// GpValidParams::fit -> the multistart COBYLA of crates/gp/src/algorithm.rs:921-945), the k x n_starts COBYLA machines advanced
// in lock-step: per round and start index, the trial points of all models whose machine still asks are ONE lock-step
// evaluation across the group's slab (what egx_gp_likelihood_multi does).  A machine only ever sees the values of its own
// model, and a model's evaluation does not depend on its companions, so every member walks the trajectory it walks under
// egx_gp_fit on a one-workspace handle: theta*, likelihood and predictions are the same bits.
int32_t egx_gp_fit_multi(egx_gp *const *gps, int32_t k, const double *theta0s, int64_t n_starts, const double *lo, const double *hi,
                         int64_t bounds_len, int64_t max_eval, int64_t *n_evals_out) {
    if (!gps || k < 1 || !theta0s || !lo || !hi || n_starts < 1) {
        set_error("NULL argument / no start point");
        return EGX_ERR_INVALID_VALUE;
    }
    std::vector<std::unique_lock<std::shared_mutex>> locks;
    EGX_RC(multi_lock(gps, k, locks));
    const int h = gps[0]->h;
    for (int32_t m = 0; m < k; m++)
        if (gps[m]->h != h) {
            set_error("egx_gp_fit_multi: the models must have the same number of hyperparameters");
            return EGX_ERR_INVALID_VALUE;
        }
    if (bounds_len != 1 && bounds_len != h) {  // algorithm.rs:901-912
        set_error("Bounds for theta should be either 1-dim or dim of xtrain (" + std::to_string(h) + "), got " + std::to_string(bounds_len));
        return EGX_ERR_INVALID_VALUE;
    }
    std::vector<double> blo((size_t)h), bhi((size_t)h);
    for (int i = 0; i < h; i++) {
        const double l = lo[bounds_len == 1 ? 0 : i], u = hi[bounds_len == 1 ? 0 : i];
        if (!(l > 0.0) || !(u >= l)) {
            set_error("theta bounds must satisfy 0 < lo <= hi");
            return EGX_ERR_INVALID_VALUE;
        }
        blo[(size_t)i] = std::log10(l), bhi[(size_t)i] = std::log10(u);  // optimization.rs:32-35
    }
    for (int64_t e = 0; e < (int64_t)k * n_starts * h; e++)
        if (!(theta0s[e] > 0.0)) {
            set_error("theta start points must be > 0");
            return EGX_ERR_INVALID_VALUE;
        }
    int64_t per_start = 10 * (int64_t)h;  // maxeval = clamp(10 h, 25, max_eval), algorithm.rs:933-936
    if (per_start < 25) per_start = 25;
    if (max_eval >= 25 && per_start > max_eval) per_start = max_eval;
    std::vector<std::vector<CobylaBox>> mach((size_t)k);
    for (int32_t m = 0; m < k; m++) {
        mach[(size_t)m].reserve((size_t)n_starts);
        for (int64_t st = 0; st < n_starts; st++) {
            std::vector<double> x0((size_t)h);
            for (int i = 0; i < h; i++) x0[(size_t)i] = std::log10(theta0s[((size_t)m * n_starts + st) * h + i]);
            mach[(size_t)m].emplace_back(x0, blo, bhi, 0.5, 1e-4, per_start);  // rhobeg / ftol_rel: optimization.rs:16-24
        }
    }
    std::vector<egx_gp *> sub;
    std::vector<int32_t> who, stt;
    std::vector<double> thetas, lk, x;
    for (bool any = true; any;) {
        any = false;
        for (int64_t st = 0; st < n_starts; st++) {  // start index st of every model whose machine still asks: one lock-step evaluation
            sub.clear(), who.clear(), thetas.clear();
            for (int32_t m = 0; m < k; m++)
                if (mach[(size_t)m][(size_t)st].ask(x)) {
                    sub.push_back(gps[m]);
                    who.push_back(m);
                    for (int i = 0; i < h; i++) thetas.push_back(std::pow(10.0, x[(size_t)i]));
                }
            if (sub.empty()) continue;
            any = true;
            lk.assign(sub.size(), 0.0), stt.assign(sub.size(), 0);
            EGX_RC(multi_eval_locked(sub.data(), (int32_t)sub.size(), thetas.data(), h, false, lk.data(), stt.data()));
            for (size_t q = 0; q < who.size(); q++) {
                const bool ok = stt[q] == EGX_STATUS_OK && !std::isnan(lk[q]);
                mach[(size_t)who[q]][(size_t)st].tell(ok ? -lk[q] : std::numeric_limits<double>::infinity());  // algorithm.rs:893-896
            }
        }
    }
    // the reduction over every model's starts (first wins ties, algorithm.rs:942-945), then ONE lock-step finalize
    std::vector<double> best((size_t)k * h);
    for (int32_t m = 0; m < k; m++) {
        double best_f = std::numeric_limits<double>::infinity();
        std::vector<double> bx((size_t)h, 0.0);
        int64_t evals = 0;
        for (int64_t st = 0; st < n_starts; st++) {
            const CobylaBox &c = mach[(size_t)m][(size_t)st];
            double fb = c.best_f();
            if (std::isnan(fb) || fb >= 1e30) fb = std::numeric_limits<double>::infinity();  // optimization.rs:153-157
            evals += c.evals();
            if (fb < best_f) best_f = fb, bx = c.best_x();
        }
        if (n_evals_out) n_evals_out[m] = evals;
        for (int i = 0; i < h; i++)
            best[(size_t)m * h + i] = std::isfinite(best_f) ? std::pow(10.0, bx[(size_t)i]) : theta0s[((size_t)m * n_starts) * h + i];
    }
    return multi_eval_locked(gps, k, best.data(), h, true, nullptr, nullptr);
}

int32_t egx_gp_get_inner(egx_gp *gp, const egx_gp_inner_view *v) {
    if (!gp || !v) {
        set_error("NULL argument");
        return EGX_ERR_INVALID_VALUE;
    }
    std::unique_lock<std::shared_mutex> lock(gp->mu);
    if (!gp->fitted) {
        set_error("model is not fitted");
        return EGX_ERR_NOT_FITTED;
    }
    EGX_RC(set_device(gp));
    const int n = gp->n, p = gp->p, d = gp->d;
    if (v->theta) std::memcpy(v->theta, gp->theta.data(), sizeof(double) * gp->h);
    if (v->likelihood) *v->likelihood = gp->likelihood;
    if (v->sigma2) *v->sigma2 = gp->sigma2;
    if (v->beta) std::memcpy(v->beta, gp->beta.data(), sizeof(double) * p);
    if (v->gamma) std::memcpy(v->gamma, gp->gamma.data(), sizeof(double) * n);
    if (v->ft) std::memcpy(v->ft, gp->ft.data(), sizeof(double) * (size_t)n * p);
    if (v->ft_qr_r) std::memcpy(v->ft_qr_r, gp->ft_qr_r.data(), sizeof(double) * (size_t)p * p);
    if (v->x_mean) std::memcpy(v->x_mean, gp->x_mean.data(), sizeof(double) * d);
    if (v->x_std) std::memcpy(v->x_std, gp->x_std.data(), sizeof(double) * d);
    if (v->y_mean) *v->y_mean = gp->y_mean;
    if (v->y_std) *v->y_std = gp->y_std;
    if (v->xt_norm) std::memcpy(v->xt_norm, gp->xnorm.data(), sizeof(double) * (size_t)n * d);
    if (v->yt_norm) std::memcpy(v->yt_norm, gp->ynorm.data(), sizeof(double) * n);
    if (v->r_chol) {
        Workspace &w = gp->ws[0];
        EGX_HIP_CHECK(hipMemcpy2DAsync(v->r_chol, sizeof(double) * n, w.M, sizeof(double) * gp->ld, sizeof(double) * n,
                                       n, hipMemcpyDeviceToHost, w.stream));
        EGX_HIP_CHECK(hipStreamSynchronize(w.stream));
        for (int i = 0; i < n; i++)
            for (int j = i + 1; j < n; j++) v->r_chol[(size_t)i * n + j] = 0.0;
    }
    return EGX_SUCCESS;
}

int32_t egx_gp_set_inner(egx_gp *gp, const egx_gp_inner_view *v) {
    if (!gp || !v || !v->theta || !v->likelihood || !v->sigma2 || !v->beta || !v->gamma || !v->r_chol || !v->ft ||
        !v->ft_qr_r) {
        set_error("egx_gp_set_inner: theta, likelihood, sigma2, beta, gamma, r_chol, ft and ft_qr_r are required");
        return EGX_ERR_INVALID_VALUE;
    }
    std::unique_lock<std::shared_mutex> lock(gp->mu);
    EGX_RC(set_device(gp));
    const int n = gp->n, p = gp->p, n_pad = gp->n_pad;
    for (int i = 0; i < n; i++)
        if (!(v->r_chol[(size_t)i * n + i] > 0.0)) {
            set_error("egx_gp_set_inner: r_chol must have a positive diagonal");
            return EGX_ERR_INVALID_VALUE;
        }
    std::vector<double> coef, thfull;
    int hcols = 1;
    EGX_RC(make_coef(gp, v->theta, gp->h, coef, hcols, &thfull));
    gp->fitted = false;
    Workspace &w = gp->ws[0];
    // factor (identity padded, lower) + ft^T rows below it, exactly the layout a factorisation leaves behind
    std::vector<double> hm((size_t)gp->m_tot * gp->ld, 0.0);
    for (int i = 0; i < n; i++)
        for (int j = 0; j <= i; j++) hm[(size_t)i * gp->ld + j] = v->r_chol[(size_t)i * n + j];
    for (int i = n; i < n_pad; i++) hm[(size_t)i * gp->ld + i] = 1.0;
    for (int l = 0; l < p; l++)
        for (int i = 0; i < n; i++) hm[(size_t)(n_pad + l) * gp->ld + i] = v->ft[(size_t)i * p + l];
    EGX_HIP_CHECK(hipMemcpyAsync(w.M, hm.data(), sizeof(double) * hm.size(), hipMemcpyHostToDevice, w.stream));
    EGX_RC(launch_diag_tile_inverses(w.stream, w.M, gp->ld, n_pad, w.dinv));
    std::memset(w.h_vec, 0, sizeof(double) * n_pad);
    std::memcpy(w.h_vec, v->gamma, sizeof(double) * n);
    EGX_HIP_CHECK(hipMemcpyAsync(gp->d_gamma, w.h_vec, sizeof(double) * n_pad, hipMemcpyHostToDevice, w.stream));
    std::memcpy(w.h_coef, coef.data(), sizeof(double) * coef.size());
    EGX_HIP_CHECK(hipMemcpyAsync(gp->d_fit_coef, w.h_coef, sizeof(double) * coef.size(), hipMemcpyHostToDevice, w.stream));
    if (hcols == 1) EGX_RC(launch_scale_rows(w.stream, gp->d_xT, gp->n_pad, gp->d, gp->d_fit_coef, dev_xs_fit(gp)));
    EGX_HIP_CHECK(hipStreamSynchronize(w.stream));
    gp->theta = thfull;
    gp->likelihood = *v->likelihood;
    gp->sigma2 = *v->sigma2;
    gp->beta.assign(v->beta, v->beta + p);
    gp->gamma.assign(v->gamma, v->gamma + n);
    gp->ft.assign(v->ft, v->ft + (size_t)n * p);
    gp->ft_qr_r.assign(v->ft_qr_r, v->ft_qr_r + (size_t)p * p);
    gp->fit_coef = coef;
    gp->fit_hcols = hcols;
    gp->fitted = true;
    gp->fit_epoch++;
    gp->small_var_calls = 0;
    return EGX_SUCCESS;
}

}  // extern "C"

namespace egx {
static int upload_kmajor(const double *x, int64_t n, int64_t d, int n_pad, DevBuf &buf) {
    std::vector<double> xT((size_t)d * n_pad, 0.0);
    for (int64_t i = 0; i < n; i++)
        for (int64_t j = 0; j < d; j++) xT[(size_t)j * n_pad + i] = x[i * d + j];
    EGX_RC(buf.alloc(xT.size()));
    EGX_HIP_CHECK(hipMemcpy(buf.p, xT.data(), sizeof(double) * xT.size(), hipMemcpyHostToDevice));
    return EGX_SUCCESS;
}
static int require_device() {
    if (egx_device_count() <= 0) {
        set_error("no HIP device: libegx_gp_hip has no CPU fallback (needs an MI355X / gfx950 GPU)");
        return EGX_ERR_NO_DEVICE;
    }
    return EGX_SUCCESS;
}
}  // namespace egx

extern "C" {

int32_t egx_corr_matrix(int32_t corr, const double *xnorm, int64_t n, int64_t d, const double *theta, double nugget,
                        double *r) {
    if (!xnorm || !theta || !r || n < 1 || d < 1 || corr < 0 || corr > 3) {
        set_error("egx_corr_matrix: bad arguments");
        return EGX_ERR_INVALID_VALUE;
    }
    EGX_RC(require_device());
    const int n_pad = (int)round_up(n, kTile);
    DevBuf d_xT, d_coef, d_M;
    EGX_RC(upload_kmajor(xnorm, n, d, n_pad, d_xT));
    EGX_RC(d_coef.alloc(d));
    EGX_HIP_CHECK(hipMemcpy(d_coef.p, theta, sizeof(double) * d, hipMemcpyHostToDevice));
    EGX_RC(d_M.alloc((size_t)n_pad * n_pad));
    EGX_RC(launch_corr_sym(0, corr, d_xT.p, n_pad, (int)n, (int)d, d_coef.p, 1, nugget, d_M.p, n_pad, n_pad));
    EGX_HIP_CHECK(hipMemcpy2D(r, sizeof(double) * n, d_M.p, sizeof(double) * n_pad, sizeof(double) * n, n,
                              hipMemcpyDeviceToHost));
    for (int64_t i = 0; i < n; i++)  // mirror the lower triangle (the scatter loop writes both, algorithm.rs:999-1000)
        for (int64_t j = i + 1; j < n; j++) r[i * n + j] = r[j * n + i];
    return EGX_SUCCESS;
}

int32_t egx_cross_corr(int32_t corr, const double *xq_norm, int64_t m, const double *xt_norm, int64_t n, int64_t d,
                       const double *theta, double *r) {
    if (!xq_norm || !xt_norm || !theta || !r || n < 1 || m < 1 || d < 1 || corr < 0 || corr > 3) {
        set_error("egx_cross_corr: bad arguments");
        return EGX_ERR_INVALID_VALUE;
    }
    EGX_RC(require_device());
    const int n_pad = (int)round_up(n, kTile), m_pad = (int)round_up(m, kTile);
    DevBuf d_xT, d_qT, d_coef, d_R;
    EGX_RC(upload_kmajor(xt_norm, n, d, n_pad, d_xT));
    EGX_RC(upload_kmajor(xq_norm, m, d, m_pad, d_qT));
    EGX_RC(d_coef.alloc(d));
    EGX_HIP_CHECK(hipMemcpy(d_coef.p, theta, sizeof(double) * d, hipMemcpyHostToDevice));
    EGX_RC(d_R.alloc((size_t)m_pad * n_pad));
    EGX_RC(launch_cross_corr(0, corr, d_qT.p, m_pad, m_pad, d_xT.p, n_pad, n_pad, (int)d, d_coef.p, 1, d_R.p, n_pad));
    EGX_HIP_CHECK(hipMemcpy2D(r, sizeof(double) * n, d_R.p, sizeof(double) * n_pad, sizeof(double) * n, m,
                              hipMemcpyDeviceToHost));
    return EGX_SUCCESS;
}

int32_t egx_potrf(double *a, int64_t n, int32_t *info) {
    if (!a || !info || n < 1) {
        set_error("egx_potrf: bad arguments");
        return EGX_ERR_INVALID_VALUE;
    }
    EGX_RC(require_device());
    const int n_pad = (int)round_up(n, kTile);
    std::vector<double> hp((size_t)n_pad * n_pad, 0.0);
    for (int64_t i = 0; i < n; i++) std::memcpy(&hp[(size_t)i * n_pad], a + i * n, sizeof(double) * n);
    for (int64_t i = n; i < n_pad; i++) hp[(size_t)i * n_pad + i] = 1.0;
    DevBuf d_M, d_dinv, d_info;  // d_info: one int stored in a double-sized slot
    EGX_RC(d_M.alloc(hp.size()));
    EGX_RC(d_dinv.alloc(dinv_doubles(n_pad)));
    EGX_RC(d_info.alloc(1));
    EGX_HIP_CHECK(hipMemset(d_info.p, 0, sizeof(double)));
    EGX_HIP_CHECK(hipMemcpy(d_M.p, hp.data(), sizeof(double) * hp.size(), hipMemcpyHostToDevice));
    // the schedule a one-workspace handle of this size gets (schedule.h): up to 7168 columns one chain launch
    const PotrfSchedule sch = schedule_for(n_pad, 1, 1);
    DevBuf d_sync;  // (ints in double-sized slots)
    PotrfBatch pb;
    pb.left = sch.left, pb.pipe = sch.pipe, pb.whole = sch.whole, pb.flow = sch.flow, pb.flow_tail = sch.flow_tail, pb.group_panels = sch.group_panels;
    if (sch.pipe || sch.flow || sch.flow_tail) {
        EGX_RC(d_sync.alloc((pipe_sync_ints(n_pad, n_pad) + 1) / 2));
        pb.sync = reinterpret_cast<int *>(d_sync.p);
    }
    EGX_RC(launch_potrf(0, d_M.p, n_pad, n_pad, n_pad, d_dinv.p, reinterpret_cast<int *>(d_info.p), nullptr, nullptr, &pb));
    if (pb.sync) {
        int aborted = 0;
        EGX_HIP_CHECK(hipMemcpy(&aborted, pb.sync, sizeof(int), hipMemcpyDeviceToHost));
        if (aborted) g_chain_aborts++;
        if (aborted && g_pipe_retry.load() != 0) {  // once more by separate launches (see finish_eval)
            g_chain_retries++;
            aborted = 0;
            pb.pipe = pb.whole = pb.flow = pb.flow_tail = 0, pb.sync = nullptr;
            EGX_HIP_CHECK(hipMemset(d_info.p, 0, sizeof(double)));
            EGX_HIP_CHECK(hipMemcpy(d_M.p, hp.data(), sizeof(double) * hp.size(), hipMemcpyHostToDevice));
            EGX_RC(launch_potrf(0, d_M.p, n_pad, n_pad, n_pad, d_dinv.p, reinterpret_cast<int *>(d_info.p), nullptr, nullptr, &pb));
        }
        if (aborted) {
            set_error("egx_potrf: a wait inside the pipelined chain kernel exceeded EGX_PIPE_TIMEOUT_MS");
            return EGX_ERR_HIP;
        }
    }
    EGX_HIP_CHECK(hipMemcpy(hp.data(), d_M.p, sizeof(double) * hp.size(), hipMemcpyDeviceToHost));
    EGX_HIP_CHECK(hipMemcpy(info, d_info.p, sizeof(int), hipMemcpyDeviceToHost));
    for (int64_t i = 0; i < n; i++)
        for (int64_t j = 0; j < n; j++) a[i * n + j] = (j <= i) ? hp[(size_t)i * n_pad + j] : 0.0;
    return EGX_SUCCESS;
}

int32_t egx_gp_last_timings(const egx_gp *gp, egx_timings *t) {
    if (!gp || !t) {
        set_error("NULL argument");
        return EGX_ERR_INVALID_VALUE;
    }
    *t = gp->timings;
    return EGX_SUCCESS;
}

int32_t egx_mfma_probe(double *max_abs_err) {
    if (!max_abs_err) return EGX_ERR_INVALID_VALUE;
    if (egx_device_count() <= 0) {
        set_error("no HIP device");
        return EGX_ERR_NO_DEVICE;
    }
    return mfma_probe(max_abs_err);
}

}  // extern "C"
