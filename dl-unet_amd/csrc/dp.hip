// dp.hip — batch-sharded data parallelism behind the C ABI: one process per GPU, ONE exchange step,
// the gradient all-reduce, done with RCCL over xGMI on a communicator stream of its own.
//
//   unet_dp_unique_id   rank 0 makes the 128-byte rendezvous id; the host (Python) only carries these bytes to the
//                       other ranks (any channel: torch.distributed store, a file, MPI)
//   unet_dp_init        ncclCommInitRank + a non-blocking communicator stream + fork/join events
//   unet_dp_allreduce   in-place SUM all-reduce of a bucket: the communicator stream first waits for everything enqueued
//                       on the caller's stream so far (the bucket's gradients are final there), then runs the collective —
//                       the caller's stream goes on with the next backward stage meanwhile
//   unet_dp_join        the caller's stream waits for every collective issued so far (before the optimizer reads grads)
//   unet_dp_broadcast   parameters from rank 0 at start-up
//
// The reference has no distributed code (single cuda:0, main_main.py:157-158): this is new work named by the
// north_star, so there is no NCCL call pattern to follow.  librccl is bound at run time (dlopen) — the copy the process
// already has loaded (PyTorch-ROCm ships one) is reused, otherwise /opt/rocm/lib/librccl.so.1 — so libunet_hip.so itself
// loads on machines without it and single-GPU use never touches it.
#include "common.hpp"
#include "../../include/unet_hip.h"

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstdio>
#include <cstring>
#include <mutex>

namespace unet {

struct Rccl {
    void *so = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclBroadcast) Broadcast = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    decltype(&ncclGetVersion) GetVersion = nullptr;
};
static Rccl g_rccl;
static std::mutex g_rccl_mu;

static int rccl_load()
{
    std::lock_guard<std::mutex> lk(g_rccl_mu);
    if (g_rccl.so) return 0;
    void *so = nullptr;
    const char *names[] = {"librccl.so.1", "librccl.so"};
    for (const char *n : names) if (!so) so = dlopen(n, RTLD_NOW | RTLD_NOLOAD);       // a copy this process already has
    for (const char *n : names) if (!so) so = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    if (!so) so = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!so) { set_error("data parallel: cannot load librccl (%s)", dlerror()); return UNET_E_UNSUPPORTED; }
#define BIND(f)                                                                                   \
    g_rccl.f = (decltype(g_rccl.f))dlsym(so, "nccl" #f);                                          \
    if (!g_rccl.f) { set_error("data parallel: librccl lacks nccl" #f); return UNET_E_UNSUPPORTED; }
    BIND(GetUniqueId) BIND(CommInitRank) BIND(CommDestroy) BIND(AllReduce) BIND(Broadcast) BIND(GetErrorString) BIND(GetVersion)
#undef BIND
    g_rccl.so = so;
    return 0;
}

#define NCCL_TRY(expr)                                                                     \
    do {                                                                                   \
        ncclResult_t r_ = (expr);                                                          \
        if (r_ != ncclSuccess) {                                                           \
            unet::set_error("%s failed: %s", #expr, g_rccl.GetErrorString(r_));            \
            return UNET_E_COMM;                                                            \
        }                                                                                  \
    } while (0)

}  // namespace unet

using namespace unet;

constexpr int DP_EVENTS = 8;
struct unet_dp {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1, device = 0;
    hipStream_t cs = nullptr;                   // communicator stream
    hipEvent_t ev[DP_EVENTS] = {nullptr};       // fork events, used round-robin
    hipEvent_t done = nullptr;                  // join event
    unsigned next = 0;
};

int unet_dp_free(unet_dp *d)
{
    if (!d) return 0;
    // collectives of this communicator may still be queued (a second enable_data_parallel, or a second module on the
    // handle's device, re-initialises the slot): drain its stream before the communicator goes away
    if (d->cs) (void)hipStreamSynchronize(d->cs);
    if (d->comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(d->comm);
    for (auto &e : d->ev) if (e) (void)hipEventDestroy(e);
    if (d->done) (void)hipEventDestroy(d->done);
    if (d->cs) (void)hipStreamDestroy(d->cs);
    delete d;
    return 0;
}

// handle layout is private to net.hip: these two accessors are all dp.hip needs
unet_dp **unet_handle_dp_slot(unet_handle *h);
int unet_handle_device(const unet_handle *h);

// stream calls run on the communicator's device: the events and the communicator stream belong to it
#define DP_CHECK_DEVICE(d, what)                                                                                       \
    do {                                                                                                               \
        int cur_ = -1;                                                                                                 \
        HIP_TRY(hipGetDevice(&cur_));                                                                                  \
        ARG_CHECK(cur_ == (d)->device, what ": the communicator belongs to device %d but device %d is current", (d)->device, cur_); \
    } while (0)

extern "C" {

int unet_dp_unique_id(void *id_out)
{
    ARG_CHECK(id_out, "unet_dp_unique_id: null argument");
    if (int rc = rccl_load()) return rc;
    static_assert(sizeof(ncclUniqueId) == UNET_DP_ID_BYTES, "rendezvous id size");
    ncclUniqueId id;
    NCCL_TRY(g_rccl.GetUniqueId(&id));
    memcpy(id_out, &id, sizeof(id));
    return 0;
}

int unet_dp_init(unet_handle *h, int rank, int world, const void *id_bytes)
{
    ARG_CHECK(h && id_bytes, "unet_dp_init: null argument");
    ARG_CHECK(world >= 1 && rank >= 0 && rank < world, "unet_dp_init: bad rank %d of %d", rank, world);
    if (int rc = rccl_load()) return rc;
    unet_dp **slot = unet_handle_dp_slot(h);
    if (*slot) { (void)unet_dp_free(*slot); *slot = nullptr; }      // drains the old communicator's stream first
    int cur = -1;
    HIP_TRY(hipGetDevice(&cur));
    ARG_CHECK(cur == unet_handle_device(h), "unet_dp_init: the handle belongs to device %d but device %d is current", unet_handle_device(h), cur);
    unet_dp *d = new unet_dp();
    d->rank = rank; d->world = world; d->device = cur;
    ncclUniqueId id;
    memcpy(&id, id_bytes, sizeof(id));
    ncclResult_t r = g_rccl.CommInitRank(&d->comm, world, id, rank);
    if (r != ncclSuccess) { set_error("ncclCommInitRank(rank %d of %d) failed: %s", rank, world, g_rccl.GetErrorString(r)); d->comm = nullptr; (void)unet_dp_free(d); return UNET_E_COMM; }
    hipError_t e = hipStreamCreateWithFlags(&d->cs, hipStreamNonBlocking);
    for (int i = 0; i < DP_EVENTS && e == hipSuccess; ++i) e = hipEventCreateWithFlags(&d->ev[i], hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&d->done, hipEventDisableTiming);
    if (e != hipSuccess) { set_error("unet_dp_init: %s", hipGetErrorString(e)); (void)unet_dp_free(d); return (int)e; }
    *slot = d;
    return 0;
}

int unet_dp_destroy(unet_handle *h)
{
    ARG_CHECK(h, "unet_dp_destroy: null handle");
    unet_dp **slot = unet_handle_dp_slot(h);
    if (*slot) { (void)unet_dp_free(*slot); *slot = nullptr; }
    return 0;
}

int unet_dp_world(unet_handle *h)
{
    if (!h) return 0;
    unet_dp *d = *unet_handle_dp_slot(h);
    return d ? d->world : 0;
}

int unet_dp_rccl_version(void)
{
    if (rccl_load()) return 0;
    int v = 0;
    return g_rccl.GetVersion(&v) == ncclSuccess ? v : 0;
}

int unet_dp_allreduce(unet_handle *h, void *buf, size_t count, void *stream)
{
    ARG_CHECK(h && buf, "unet_dp_allreduce: null argument");
    unet_dp *d = *unet_handle_dp_slot(h);
    ARG_CHECK(d, "unet_dp_allreduce: unet_dp_init has not been called on this handle");
    DP_CHECK_DEVICE(d, "unet_dp_allreduce");
    if (count == 0) return 0;
    hipEvent_t ev = d->ev[d->next++ % DP_EVENTS];
    HIP_TRY(hipEventRecord(ev, (hipStream_t)stream));           // the bucket is final on the caller's stream here
    HIP_TRY(hipStreamWaitEvent(d->cs, ev, 0));
    char tag[64];
    snprintf(tag, sizeof(tag), "allreduce n=%zu world=%d", count, d->world);
    ProfScope ps("allreduce");
    // ring all-reduce moves 2 (w-1)/w of the message over each link (reduce-scatter + all-gather)
    prof_begin(PK_COMM, tag, d->cs, 0.0, 0.0, 4.0 * (double)count * 2.0 * (d->world - 1) / d->world);
    ncclResult_t r = g_rccl.AllReduce(buf, buf, count, ncclFloat32, ncclSum, d->comm, d->cs);
    prof_end(d->cs);
    if (r != ncclSuccess) { set_error("ncclAllReduce failed: %s", g_rccl.GetErrorString(r)); return UNET_E_COMM; }
    return 0;
}

int unet_dp_broadcast(unet_handle *h, void *buf, size_t count, int root, void *stream)
{
    ARG_CHECK(h && buf, "unet_dp_broadcast: null argument");
    unet_dp *d = *unet_handle_dp_slot(h);
    ARG_CHECK(d, "unet_dp_broadcast: unet_dp_init has not been called on this handle");
    ARG_CHECK(root >= 0 && root < d->world, "unet_dp_broadcast: bad root %d", root);
    DP_CHECK_DEVICE(d, "unet_dp_broadcast");
    if (count == 0) return 0;
    hipEvent_t ev = d->ev[d->next++ % DP_EVENTS];
    HIP_TRY(hipEventRecord(ev, (hipStream_t)stream));
    HIP_TRY(hipStreamWaitEvent(d->cs, ev, 0));
    NCCL_TRY(g_rccl.Broadcast(buf, buf, count, ncclFloat32, root, d->comm, d->cs));
    return 0;
}

int unet_dp_join(unet_handle *h, void *stream)
{
    ARG_CHECK(h, "unet_dp_join: null handle");
    unet_dp *d = *unet_handle_dp_slot(h);
    ARG_CHECK(d, "unet_dp_join: unet_dp_init has not been called on this handle");
    DP_CHECK_DEVICE(d, "unet_dp_join");
    HIP_TRY(hipEventRecord(d->done, d->cs));
    // under profiling the wait is bracketed on the CALLER's stream: its duration is the exposed (non-overlapped) part
    // of the step's collectives
    ProfScope ps("allreduce");
    prof_begin(PK_COMM, "join (exposed wait of the compute stream)", (hipStream_t)stream, 0.0, 0.0, 0.0);
    HIP_TRY(hipStreamWaitEvent((hipStream_t)stream, d->done, 0));
    prof_end((hipStream_t)stream);
    return 0;
}

}  // extern "C"
