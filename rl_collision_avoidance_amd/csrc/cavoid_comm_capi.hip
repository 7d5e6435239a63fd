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
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstring>
#include <new>

#include "cavoid.h"
#include "cavoid_host.hpp"

static_assert(CAVOID_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "unique id size");

thread_local int g_last_comm_error = 0;

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
    ncclComm_t comm = nullptr;      // null when nranks == 1 (the gather is a device copy)
    hipStream_t stream = nullptr;   // the communicator's own stream
    hipEvent_t ev_ready[CAVOID_COMM_SLOTS] = {}, ev_done[CAVOID_COMM_SLOTS] = {};
    bool pending[CAVOID_COMM_SLOTS] = {};
};

extern "C" int cavoid_last_comm_error(void) { return g_last_comm_error; }

extern "C" int cavoid_comm_unique_id(void *id_out) {
    if (!id_out) return CAVOID_EINVAL;
    ncclUniqueId id;
    COMM_TRY(ncclGetUniqueId(&id));
    std::memcpy(id_out, &id, sizeof(id));
    return CAVOID_OK;
}

extern "C" int cavoid_comm_create(const void *unique_id, int32_t nranks, int32_t rank, int device, cavoid_comm **out) {
    if (!out) return CAVOID_EINVAL;
    *out = nullptr;
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
    if (nranks > 1) {
        ncclUniqueId id;
        std::memcpy(&id, unique_id, sizeof(id));
        ncclResult_t r = ncclCommInitRank(&c->comm, nranks, id, rank);
        if (r != ncclSuccess) {
            g_last_comm_error = (int)r;
            c->comm = nullptr;
            cavoid_comm_destroy(c);
            return CAVOID_ECOMM;
        }
    }
    *out = c;
    return CAVOID_OK;
}

extern "C" void cavoid_comm_destroy(cavoid_comm *c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->comm) (void)ncclCommDestroy(c->comm);
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
    if (c->nranks == 1) {
        if (floats_per_rank > 0 && send != recv)
            HIP_TRY(hipMemcpyAsync(recv, send, (size_t)floats_per_rank * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
    } else {
        COMM_TRY(ncclAllGather(send, recv, (size_t)floats_per_rank, ncclFloat, c->comm, c->stream));
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
