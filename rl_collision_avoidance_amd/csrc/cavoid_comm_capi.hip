// cavoid_comm_capi.hip -- the multi-GPU hand-over of include/cavoid.h: ONE ncclAllGather (RCCL over xGMI) of every
// rank's packed (obs | reward | done) shard per step, on the communicator's own stream so that gather(t) overlaps
// step(t+1).  Host side only.  The reference moves the same records between OS processes
// through mp.Queue (ga3c/GA3C/ProcessAgent.py:221,238); there is no collective in the reference to translate.
//
// Stream protocol (no host synchronisation anywhere):
// with slot = t % 2:
//   producer stream:  cavoid_gather_wait(slot) [gather(t-2) has left send[slot]] -> step(t) writes send[slot] -> record ev_ready[slot]
//   comm stream:      wait ev_ready[slot] -> ncclAllGather(send[slot] -> recv[slot]) -> record ev_done[slot]
//   consumer stream:  cavoid_gather_wait(slot) = wait ev_done[slot]  (the trainer side reads recv[slot])
// The caller double-buffers send / recv; step(t+1) is enqueued on the producer stream right after
// cavoid_gather_begin(t) returns and runs while gather(t) is on the wire.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>      // types and constants only: the library itself is bound lazily (see rccl())

#include <cstdlib>
#include <cstring>
#include <new>

#include "cavoid.h"
#include "cavoid_host.hpp"

static_assert(CAVOID_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "unique id size");

thread_local int g_last_comm_error = 0;

// RCCL is bound at the first multi-rank call, not at load time: the single-GPU env, the policy kernels and the host-only
// tests must load on a box without librccl on the loader path, and inside a PyTorch process the communicator must use the
// librccl PyTorch has already mapped (RTLD_NOLOAD first), not a second copy.
struct RcclApi {
    ncclResult_t (*get_unique_id)(ncclUniqueId *);
    ncclResult_t (*comm_init_rank)(ncclComm_t *, int, ncclUniqueId, int);
    ncclResult_t (*comm_destroy)(ncclComm_t);
    ncclResult_t (*all_gather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t);
    ncclResult_t (*gather_send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t);
    ncclResult_t (*gather_recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t);
    ncclResult_t (*group_start)(void);
    ncclResult_t (*group_end)(void);
    ncclResult_t (*get_version)(int *);
    ncclResult_t (*comm_count)(const ncclComm_t, int *);
    ncclResult_t (*comm_user_rank)(const ncclComm_t, int *);
};
static const RcclApi *rccl() {
    static RcclApi api{};
    static int state = 0;                                   // 0 untried, 1 bound, -1 unavailable
    if (state == 0) {
        void *h = nullptr;
        const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
        for (int pass = 0; pass < 2 && !h; ++pass)
            for (const char *n : names) {
                h = dlopen(n, RTLD_NOW | RTLD_GLOBAL | (pass == 0 ? RTLD_NOLOAD : 0));
                if (h) break;
            }
        bool ok = h != nullptr;
        auto bind = [&](auto &fn, const char *sym) { if (ok) { fn = reinterpret_cast<decltype(+fn)>(dlsym(h, sym)); ok = fn != nullptr; } };
        bind(api.get_unique_id, "ncclGetUniqueId"); bind(api.comm_init_rank, "ncclCommInitRank"); bind(api.comm_destroy, "ncclCommDestroy");
        bind(api.all_gather, "ncclAllGather"); bind(api.gather_send, "ncclSend"); bind(api.gather_recv, "ncclRecv");
        bind(api.group_start, "ncclGroupStart"); bind(api.group_end, "ncclGroupEnd");
        bind(api.get_version, "ncclGetVersion"); bind(api.comm_count, "ncclCommCount"); bind(api.comm_user_rank, "ncclCommUserRank");
        state = ok ? 1 : -1;
    }
    return state == 1 ? &api : nullptr;
}
#define RCCL_OR_FAIL(var)                                                        \
    const RcclApi *var = rccl();                                                 \
    if (!var) { g_last_comm_error = (int)ncclSystemError; return CAVOID_ECOMM; }

#define COMM_TRY(expr)                                 \
    do {                                               \
        ncclResult_t _r = (expr);                      \
        if (_r != ncclSuccess) {                       \
            g_last_comm_error = (int)_r;               \
            return CAVOID_ECOMM;                       \
        }                                              \
    } while (0)

struct cavoid_comm {
    int device = 0;
    int32_t nranks = 1, rank = 0;
    ncclComm_t comm = nullptr;      // null when nranks == 1 and RCCL was not forced (the gather is then a device copy)
    hipStream_t stream = nullptr;   // the communicator's own stream
    hipEvent_t ev_ready[CAVOID_COMM_SLOTS] = {}, ev_done[CAVOID_COMM_SLOTS] = {};
    bool pending[CAVOID_COMM_SLOTS] = {};
};

extern "C" int cavoid_last_comm_error(void) { return g_last_comm_error; }

extern "C" int cavoid_comm_unique_id(void *id_out) {
    if (!id_out) return CAVOID_EINVAL;
    RCCL_OR_FAIL(api);
    ncclUniqueId id;
    COMM_TRY(api->get_unique_id(&id));
    std::memcpy(id_out, &id, sizeof(id));
    return CAVOID_OK;
}

// CAVOID_COMM_FORCE_RCCL (flag, or the environment variable of the same name for cavoid_comm_create): a ONE-rank communicator goes
// through ncclCommInitRank / ncclAllGather / the grouped self send-recv too instead of the device copy -- the same calls a
// multi-rank communicator makes, so that a 1-GPU box executes the binding, the stream protocol and the RCCL entry points.
static bool force_rccl_env() {
    const char *v = std::getenv("CAVOID_COMM_FORCE_RCCL");
    return v && v[0] && !(v[0] == '0' && !v[1]);
}

extern "C" int cavoid_comm_create_ex(const void *unique_id, int32_t nranks, int32_t rank, int device, uint32_t flags, cavoid_comm **out) {
    if (!out) return CAVOID_EINVAL;
    *out = nullptr;
    if (flags & ~(uint32_t)CAVOID_COMM_FORCE_RCCL) return CAVOID_EINVAL;
    const bool with_rccl = nranks > 1 || (flags & CAVOID_COMM_FORCE_RCCL);
    if (nranks < 1 || rank < 0 || rank >= nranks || (nranks > 1 && !unique_id)) return CAVOID_EINVAL;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) return CAVOID_ENODEVICE;
    HIP_TRY(hipSetDevice(device));
    cavoid_comm *c = new (std::nothrow) cavoid_comm();
    if (!c) return CAVOID_ENOMEM;
    c->device = device; c->nranks = nranks; c->rank = rank;
    bool ok = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) == hipSuccess;
    for (int k = 0; k < CAVOID_COMM_SLOTS && ok; ++k)
        ok = hipEventCreateWithFlags(&c->ev_ready[k], hipEventDisableTiming) == hipSuccess &&
             hipEventCreateWithFlags(&c->ev_done[k], hipEventDisableTiming) == hipSuccess;
    if (!ok) {
        g_last_hip_error = (int)hipGetLastError();
        cavoid_comm_destroy(c);
        return CAVOID_EHIP;
    }
    if (with_rccl) {
        const RcclApi *api = rccl();
        ncclUniqueId id;
        ncclResult_t r = api ? ncclSuccess : ncclSystemError;
        if (r == ncclSuccess) {
            if (unique_id) std::memcpy(&id, unique_id, sizeof(id));
            else r = api->get_unique_id(&id);                  // (forced one-rank communicator without an id: make one here)
        }
        if (r == ncclSuccess) r = api->comm_init_rank(&c->comm, nranks, id, rank);
        int n = -1, me = -1;
        if (r == ncclSuccess) r = api->comm_count(c->comm, &n);
        if (r == ncclSuccess) r = api->comm_user_rank(c->comm, &me);
        if (r == ncclSuccess && (n != nranks || me != rank)) r = ncclInternalError;     // the communicator is not the one asked for
        if (r != ncclSuccess) {
            g_last_comm_error = (int)r;
            if (c->comm) (void)api->comm_destroy(c->comm);
            c->comm = nullptr;
            cavoid_comm_destroy(c);
            return CAVOID_ECOMM;
        }
    }
    *out = c;
    return CAVOID_OK;
}

extern "C" int cavoid_comm_create(const void *unique_id, int32_t nranks, int32_t rank, int device, cavoid_comm **out) {
    return cavoid_comm_create_ex(unique_id, nranks, rank, device, force_rccl_env() ? CAVOID_COMM_FORCE_RCCL : 0u, out);
}

extern "C" int cavoid_comm_info(const cavoid_comm *c, int32_t *nranks, int32_t *rank, int32_t *uses_rccl, int32_t *rccl_version) {
    if (!c) return CAVOID_EINVAL;
    if (nranks) *nranks = c->nranks;
    if (rank) *rank = c->rank;
    if (uses_rccl) *uses_rccl = c->comm != nullptr;
    if (rccl_version) {
        *rccl_version = 0;
        if (c->comm) {
            RCCL_OR_FAIL(api);
            int v = 0;
            COMM_TRY(api->get_version(&v));
            *rccl_version = v;
        }
    }
    return CAVOID_OK;
}

extern "C" void cavoid_comm_destroy(cavoid_comm *c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->comm && rccl()) (void)rccl()->comm_destroy(c->comm);
    for (int k = 0; k < CAVOID_COMM_SLOTS; ++k) {
        if (c->ev_ready[k]) (void)hipEventDestroy(c->ev_ready[k]);
        if (c->ev_done[k]) (void)hipEventDestroy(c->ev_done[k]);
    }
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

extern "C" int cavoid_gather_begin(cavoid_comm *c, int32_t slot, const float *send, float *recv, int64_t floats_per_rank, void *producer_stream) {
    if (!c || !send || !recv || floats_per_rank < 0 || slot < 0 || slot >= CAVOID_COMM_SLOTS) return CAVOID_EINVAL;
    hipStream_t prod = static_cast<hipStream_t>(producer_stream);
    HIP_TRY(hipEventRecord(c->ev_ready[slot], prod));
    HIP_TRY(hipStreamWaitEvent(c->stream, c->ev_ready[slot], 0));
    if (!c->comm) {
        if (floats_per_rank > 0 && send != recv)
            HIP_TRY(hipMemcpyAsync(recv, send, (size_t)floats_per_rank * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
    } else {
        RCCL_OR_FAIL(api);
        COMM_TRY(api->all_gather(send, recv, (size_t)floats_per_rank, ncclFloat, c->comm, c->stream));
    }
    HIP_TRY(hipEventRecord(c->ev_done[slot], c->stream));
    c->pending[slot] = true;
    return CAVOID_OK;
}

// Ragged shards / gather to one rank: point-to-point over the xGMI mesh inside ONE RCCL group -- every pair of GPUs has its own
// link, so "each rank sends its shard straight to whoever wants it" is the direct all-gather (no ring, no padding).
// counts[r] = floats of rank r's shard (host array, identical on every rank); recv is laid out in rank order, rank r's shard at
// offset sum(counts[0..r)).  root < 0: every rank receives every shard (all-gather-v); root >= 0: only that rank does (recv may be
// NULL elsewhere) -- the trainer-rank variant of SURVEY.md section 8e.
extern "C" int cavoid_gatherv_begin(cavoid_comm *c, int32_t slot, const float *send, float *recv, const int64_t *counts, int32_t root,
                                    void *producer_stream) {
    if (!c || !send || !counts || slot < 0 || slot >= CAVOID_COMM_SLOTS || root >= c->nranks) return CAVOID_EINVAL;
    const bool receiver = root < 0 || root == c->rank;
    if (receiver && !recv) return CAVOID_EINVAL;
    int64_t my_off = 0;
    for (int r = 0; r < c->nranks; ++r) {
        if (counts[r] < 0) return CAVOID_EINVAL;
        if (r < c->rank) my_off += counts[r];
    }
    hipStream_t prod = static_cast<hipStream_t>(producer_stream);
    HIP_TRY(hipEventRecord(c->ev_ready[slot], prod));
    HIP_TRY(hipStreamWaitEvent(c->stream, c->ev_ready[slot], 0));
    const int64_t mine = counts[c->rank];
    // my own shard: a device copy -- except on a forced one-rank communicator, where it travels as a grouped ncclSend / ncclRecv to
    // myself so that the point-to-point entry points really run
    const bool self_p2p = c->comm && c->nranks == 1;
    if (receiver && mine > 0 && send != recv + my_off && !self_p2p)
        HIP_TRY(hipMemcpyAsync(recv + my_off, send, (size_t)mine * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
    if (c->comm) {
        RCCL_OR_FAIL(api);
        COMM_TRY(api->group_start());
        ncclResult_t bad = ncclSuccess;
        int64_t off = 0;
        for (int p = 0; p < c->nranks; ++p) {
            if (p != c->rank || self_p2p) {
                if (mine > 0 && (root < 0 || root == p)) {               // p wants my shard
                    ncclResult_t r = api->gather_send(send, (size_t)mine, ncclFloat, p, c->comm, c->stream);
                    if (r != ncclSuccess) bad = r;
                }
                if (receiver && counts[p] > 0) {
                    ncclResult_t r = api->gather_recv(recv + off, (size_t)counts[p], ncclFloat, p, c->comm, c->stream);
                    if (r != ncclSuccess) bad = r;
                }
            }
            off += counts[p];
        }
        ncclResult_t end = api->group_end();                          // (always close the group, even after a failed call)
        if (bad != ncclSuccess || end != ncclSuccess) { g_last_comm_error = (int)(bad != ncclSuccess ? bad : end); return CAVOID_ECOMM; }
    }
    HIP_TRY(hipEventRecord(c->ev_done[slot], c->stream));
    c->pending[slot] = true;
    return CAVOID_OK;
}

extern "C" int cavoid_gather_wait(cavoid_comm *c, int32_t slot, void *consumer_stream) {
    if (!c || slot < 0 || slot >= CAVOID_COMM_SLOTS) return CAVOID_EINVAL;
    if (!c->pending[slot]) return CAVOID_OK;
    HIP_TRY(hipStreamWaitEvent(static_cast<hipStream_t>(consumer_stream), c->ev_done[slot], 0));
    return CAVOID_OK;
}
