// comm_capi.hip -- the multi-GPU entry points of the C ABI (include/neddf_hip.h, section "multi-GPU"): one RCCL
// communicator per context, an all-gather of rendered pixels on a communication stream of its own.
//
// The path shards by construction (rays are independent), so there is exactly one collective per view and no
// all-reduce; xGMI is point-to-point, but a 12.8 MB all-gather per 800x800 view is three orders of magnitude below the
// render time -- the design goal is not bandwidth but keeping the exchange OFF the compute stream, so view i's pixels
// travel while view i+1 renders.
//
// RCCL is bound with dlopen/dlsym at first use: the library keeps loading on hosts without RCCL, and inside a torch
// process "librccl.so.1" resolves to the copy torch already mapped (same SONAME), so there is one RCCL per process.
#include "../../include/neddf_hip.h"
#include "kernels.h"

#include <dlfcn.h>
#include <string.h>
#include <rccl/rccl.h>

#include <chrono>
#include <string>
#include <thread>

#include "capi_internal.h"

namespace {

struct Rccl {
    void *handle = nullptr;
    ncclResult_t (*GetVersion)(int *) = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;
    ncclResult_t (*CommGetAsyncError)(ncclComm_t, ncclResult_t *) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    std::string why;
};

Rccl *rccl()
{
    static Rccl r;
    static bool tried = false;
    if (tried) return &r;
    tried = true;
    for (const char *name : { "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1" }) {
        r.handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
        if (r.handle) break;
    }
    if (!r.handle) { r.why = std::string("cannot load librccl.so.1: ") + dlerror(); return &r; }
    bool ok = true;
    auto sym = [&](const char *n) { void *p = dlsym(r.handle, n); if (!p) { ok = false; r.why = std::string("librccl lacks ") + n; } return p; };
    r.GetVersion = (decltype(r.GetVersion))sym("ncclGetVersion");
    r.GetUniqueId = (decltype(r.GetUniqueId))sym("ncclGetUniqueId");
    r.CommInitRank = (decltype(r.CommInitRank))sym("ncclCommInitRank");
    r.CommDestroy = (decltype(r.CommDestroy))sym("ncclCommDestroy");
    r.CommAbort = (decltype(r.CommAbort))sym("ncclCommAbort");
    r.CommGetAsyncError = (decltype(r.CommGetAsyncError))sym("ncclCommGetAsyncError");
    r.AllGather = (decltype(r.AllGather))sym("ncclAllGather");
    r.GetErrorString = (decltype(r.GetErrorString))sym("ncclGetErrorString");
    if (!ok) { dlclose(r.handle); r.handle = nullptr; }
    return &r;
}

int need_rccl(neddf_ctx *ctx, Rccl *&r)
{
    r = rccl();
    if (!r->handle) return fail(ctx, NEDDF_ECOMM, r->why);
    return 0;
}

#define RCCLCHK(call)                                                                                   \
    do {                                                                                                \
        ncclResult_t e_ = (call);                                                                       \
        if (e_ != ncclSuccess) return fail(ctx, NEDDF_ECOMM, std::string(#call) + ": " + r->GetErrorString(e_)); \
    } while (0)

}  // namespace

// called by neddf_destroy (neddf_capi.hip)
void neddf_comm_release(neddf_ctx *ctx)
{
    CommState &c = ctx->comm;
    if (c.comm) {
        Rccl *r = rccl();
        if (c.stream) (void)hipStreamSynchronize(c.stream);
        if (r->handle) (void)r->CommDestroy((ncclComm_t)c.comm);
        c.comm = nullptr;
    }
    if (c.ready) { (void)hipEventDestroy(c.ready); c.ready = nullptr; }
    if (c.done) { (void)hipEventDestroy(c.done); c.done = nullptr; }
    if (c.stream) { (void)hipStreamDestroy(c.stream); c.stream = nullptr; }
    if (c.pad.p) { (void)hipFree(c.pad.p); c.pad = DevBuf{}; }
    c.rank = c.nranks = 0;
    c.pending = false;
}

extern "C" {

void neddf_shard_range(int64_t n_total, int rank, int nranks, int64_t *lo, int64_t *hi)
{
    if (nranks < 1) nranks = 1;
    const int64_t base = n_total / nranks, rem = n_total % nranks;
    const int64_t l = rank * base + (rank < rem ? rank : rem);
    if (lo) *lo = l;
    if (hi) *hi = l + base + (rank < rem ? 1 : 0);
}

int neddf_comm_unique_id(neddf_ctx *ctx, void *h_id)
{
    if (!ctx || !h_id) return NEDDF_EINVAL;
    static_assert(sizeof(ncclUniqueId) == NEDDF_COMM_ID_BYTES, "NEDDF_COMM_ID_BYTES must match ncclUniqueId");
    Rccl *r;
    if (int rc = need_rccl(ctx, r)) return rc;
    DeviceGuard guard_(ctx->device);
    ncclUniqueId id;
    RCCLCHK(r->GetUniqueId(&id));
    memcpy(h_id, &id, sizeof(id));
    return 0;
}

int neddf_comm_init(neddf_ctx *ctx, int rank, int nranks, const void *h_id)
{
    if (!ctx || !h_id) return NEDDF_EINVAL;
    if (nranks < 1 || rank < 0 || rank >= nranks) return fail(ctx, NEDDF_EINVAL, "comm_init: rank must be in [0, nranks)");
    if (ctx->comm.comm) return fail(ctx, NEDDF_EINVAL, "comm_init: this context already has a communicator (neddf_comm_destroy first)");
    Rccl *r;
    if (int rc = need_rccl(ctx, r)) return rc;
    DeviceGuard guard_(ctx->device);
    CommState &c = ctx->comm;
    ncclUniqueId id;
    memcpy(&id, h_id, sizeof(id));
    ncclComm_t comm = nullptr;
    RCCLCHK(r->CommInitRank(&comm, nranks, id, rank));
    c.comm = comm; c.rank = rank; c.nranks = nranks;
    HIPCHK(hipStreamCreateWithFlags(&c.stream, hipStreamNonBlocking));
    HIPCHK(hipEventCreateWithFlags(&c.ready, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&c.done, hipEventDisableTiming));
    return 0;
}

int neddf_comm_info(neddf_ctx *ctx, int *rank, int *nranks, int *version)
{
    if (!ctx) return NEDDF_EINVAL;
    if (rank) *rank = ctx->comm.rank;
    if (nranks) *nranks = ctx->comm.nranks;
    if (version) {
        *version = 0;
        Rccl *r = rccl();
        if (r->handle) (void)r->GetVersion(version);
    }
    return 0;
}

int neddf_comm_destroy(neddf_ctx *ctx)
{
    if (!ctx) return NEDDF_EINVAL;
    DeviceGuard guard_(ctx->device);
    neddf_comm_release(ctx);
    return 0;
}

int neddf_gather_pixels(neddf_ctx *ctx, const float *d_local, int64_t n_total, int channels, float *d_all, void *stream)
{
    if (!ctx || !d_all || n_total < 0 || channels < 1) return NEDDF_EINVAL;
    CommState &c = ctx->comm;
    if (!c.comm) return fail(ctx, NEDDF_ECOMM, "gather_pixels: no communicator (neddf_comm_init)");
    if (n_total == 0) return 0;
    Rccl *r;
    if (int rc = need_rccl(ctx, r)) return rc;
    DeviceGuard guard_(ctx->device);
    int64_t lo, hi;
    neddf_shard_range(n_total, c.rank, c.nranks, &lo, &hi);
    if (hi > lo && !d_local) return NEDDF_EINVAL;
    const int64_t pad = (n_total + c.nranks - 1) / c.nranks;         // rows of the largest slab
    const bool ragged = n_total % c.nranks != 0;
    const size_t row = (size_t)channels * sizeof(float);
    if (ragged)
        if (int rc = ensure(ctx, c.pad, (size_t)(c.nranks + 1) * pad * row)) return rc;
    // the communication stream picks up after what `stream` has enqueued so far (the render of this slab) ...
    HIPCHK(hipEventRecord(c.ready, (hipStream_t)stream));
    HIPCHK(hipStreamWaitEvent(c.stream, c.ready, 0));
    // ... and after the previous gather's consumers: the caller waited (neddf_comm_wait) before touching its buffers
    tick(ctx, c.stream, NEDDF_STAGE_GATHER, true);
    if (!ragged) {
        RCCLCHK(r->AllGather(d_local, d_all, (size_t)pad * channels, ncclFloat, (ncclComm_t)c.comm, c.stream));
    } else {
        // equal-count all-gather through [send: pad rows | recv: nranks * pad rows], then one compaction copy per rank
        char *send = (char *)c.pad.p, *recv = send + pad * row;
        if (hi > lo) HIPCHK(hipMemcpyAsync(send, d_local, (size_t)(hi - lo) * row, hipMemcpyDeviceToDevice, c.stream));
        RCCLCHK(r->AllGather(send, recv, (size_t)pad * channels, ncclFloat, (ncclComm_t)c.comm, c.stream));
        for (int q = 0; q < c.nranks; ++q) {
            int64_t l, h;
            neddf_shard_range(n_total, q, c.nranks, &l, &h);
            if (h > l)
                HIPCHK(hipMemcpyAsync((char *)d_all + l * row, recv + (size_t)q * pad * row, (size_t)(h - l) * row,
                                      hipMemcpyDeviceToDevice, c.stream));
        }
    }
    tick(ctx, c.stream, NEDDF_STAGE_GATHER, false);
    HIPCHK(hipEventRecord(c.done, c.stream));
    c.pending = true;
    return 0;
}

int neddf_comm_wait(neddf_ctx *ctx, void *stream)
{
    if (!ctx) return NEDDF_EINVAL;
    CommState &c = ctx->comm;
    if (!c.comm || !c.pending) return 0;
    DeviceGuard guard_(ctx->device);
    HIPCHK(hipStreamWaitEvent((hipStream_t)stream, c.done, 0));
    return 0;
}

int neddf_comm_wait_host(neddf_ctx *ctx, int timeout_ms)
{
    if (!ctx) return NEDDF_EINVAL;
    CommState &c = ctx->comm;
    if (!c.comm || !c.pending) return 0;
    Rccl *r;
    if (int rc = need_rccl(ctx, r)) return rc;
    DeviceGuard guard_(ctx->device);
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
        hipError_t q = hipEventQuery(c.done);
        if (q == hipSuccess) { c.pending = false; return 0; }
        if (q != hipErrorNotReady) { ctx->err = std::string("comm_wait_host: ") + hipGetErrorString(q); return NEDDF_EHIP; }
        ncclResult_t async = ncclSuccess;
        RCCLCHK(r->CommGetAsyncError((ncclComm_t)c.comm, &async));
        if (async != ncclSuccess && async != ncclInProgress)
            return fail(ctx, NEDDF_ECOMM, std::string("asynchronous RCCL error: ") + r->GetErrorString(async));
        const auto ms = std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t0).count();
        if (timeout_ms >= 0 && ms > timeout_ms) {
            (void)r->CommAbort((ncclComm_t)c.comm);      // frees the communicator; a peer is gone or stuck
            c.comm = nullptr; c.pending = false;
            return fail(ctx, NEDDF_ETIMEOUT, "comm_wait_host: pixel gather did not complete in " + std::to_string(timeout_ms) + " ms; communicator aborted");
        }
        std::this_thread::sleep_for(std::chrono::microseconds(200));
    }
}

}  // extern "C"
