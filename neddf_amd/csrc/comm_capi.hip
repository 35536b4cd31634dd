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
#include <stdlib.h>
#include <string.h>
// RCCL is bound at run time (dlopen below); its development headers are used when the build host has them and otherwise
// replaced by the handful of declarations this file needs (RCCL keeps NCCL's public ABI: opaque communicator, 128-byte id)
#if __has_include(<rccl/rccl.h>)
#include <rccl/rccl.h>
#else
extern "C" {
typedef struct ncclComm *ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef enum { ncclSuccess = 0, ncclUnhandledCudaError = 1, ncclSystemError = 2, ncclInternalError = 3, ncclInvalidArgument = 4,
               ncclInvalidUsage = 5, ncclRemoteError = 6, ncclInProgress = 7 } ncclResult_t;
typedef enum { ncclInt8 = 0, ncclUint8 = 1, ncclInt32 = 2, ncclUint32 = 3, ncclInt64 = 4, ncclUint64 = 5, ncclFloat16 = 6,
               ncclFloat32 = 7, ncclFloat = 7, ncclFloat64 = 8 } ncclDataType_t;
}
#endif

#include <chrono>
#include <string>
#include <thread>
#include <vector>

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
    // optional (ragged slabs gathered in place): absent symbols select the padded staging route
    ncclResult_t (*Broadcast)(const void *, void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
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
    if (!ok) { dlclose(r.handle); r.handle = nullptr; return &r; }
    r.Broadcast = (decltype(r.Broadcast))dlsym(r.handle, "ncclBroadcast");
    r.GroupStart = (decltype(r.GroupStart))dlsym(r.handle, "ncclGroupStart");
    r.GroupEnd = (decltype(r.GroupEnd))dlsym(r.handle, "ncclGroupEnd");
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

// everything of the communicator state except the communicator itself: stream, events, staging, rank bookkeeping.
// `lost` marks a gather that was in flight when the communicator went away (neddf_comm_wait then reports it instead of
// letting a consumer read a partially gathered buffer)
static void comm_reset(CommState &c, bool lost)
{
    if (c.ready) { (void)hipEventDestroy(c.ready); c.ready = nullptr; }
    if (c.done) { (void)hipEventDestroy(c.done); c.done = nullptr; }
    if (c.stream) { (void)hipStreamDestroy(c.stream); c.stream = nullptr; }
    if (c.pad.p) { (void)hipFree(c.pad.p); c.pad = DevBuf{}; }
    c.rank = c.nranks = 0;
    c.pending = false;
    c.lost = lost;
}

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
    comm_reset(c, false);
}

extern "C" {

void neddf_shard_range(int64_t n_total, int rank, int nranks, int64_t *lo, int64_t *hi)
{
    neddf_shard_range_granular(n_total, 1, rank, nranks, lo, hi);
}

void neddf_shard_range_granular(int64_t n_total, int64_t granule, int rank, int nranks, int64_t *lo, int64_t *hi)
{
    if (nranks < 1) nranks = 1;
    if (granule < 1) granule = 1;
    if (n_total < 0) n_total = 0;
    // the units are the granules (the last one may be short); unit counts per rank differ by at most one
    const int64_t units = (n_total + granule - 1) / granule;
    const int64_t base = units / nranks, rem = units % nranks;
    const int64_t ul = rank * base + (rank < rem ? rank : rem), uh = ul + base + (rank < rem ? 1 : 0);
    const int64_t l = ul * granule < n_total ? ul * granule : n_total, h = uh * granule < n_total ? uh * granule : n_total;
    if (lo) *lo = l;
    if (hi) *hi = h;
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
    c.comm = comm; c.rank = rank; c.nranks = nranks; c.lost = false;
    hipError_t e = hipStreamCreateWithFlags(&c.stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c.ready, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c.done, hipEventDisableTiming);
    if (e != hipSuccess) {          // no half-built communicator state: unwind to "no communicator"
        const std::string why = std::string("comm_init: ") + hipGetErrorString(e);
        neddf_comm_release(ctx);
        return fail(ctx, NEDDF_EHIP, why);
    }
    // The route of a ragged gather is a property of the COMMUNICATOR, not of a call: ranks that chose differently (an environment
    // variable set on one of them, an RCCL without the group calls on another) would issue mismatched collectives and hang.  EVERY
    // rank of a communicator with more than one rank joins ONE 4-byte all-gather here, unconditionally -- a rank that skipped it
    // because it lacks the wish or the symbols would leave its peers inside it -- and contributes "I can and want to gather in
    // place" (NEDDF_GATHER_INPLACE=1 + the three optional symbols); the route is in place only if ALL of them said so, otherwise
    // every rank takes the padded staging route (equal-count all-gather + compaction copies).  It is also the communicator's first
    // collective: a broken fabric shows at start-up, not under the first frame.
    const char *ip = getenv("NEDDF_GATHER_INPLACE"), *fr = getenv("NEDDF_GATHER_FORCE_RAGGED");
    const int mine = (ip && atoi(ip) != 0 && r->Broadcast && r->GroupStart && r->GroupEnd) ? 1 : 0;
    c.force_ragged = fr && atoi(fr) != 0;
    c.in_place = mine != 0;
    if (nranks > 1) {
        int rc = ensure(ctx, c.pad, sizeof(int) * (size_t)(nranks + 1));
        std::vector<int> all((size_t)nranks, 0);
        if (!rc) {
            int *d = (int *)c.pad.p;
            e = hipMemcpyAsync(d, &mine, sizeof(int), hipMemcpyHostToDevice, c.stream);
            ncclResult_t ne = ncclSuccess;
            if (e == hipSuccess) ne = r->AllGather(d, d + 1, 1, ncclInt32, comm, c.stream);
            if (e == hipSuccess && ne == ncclSuccess) e = hipMemcpyAsync(all.data(), d + 1, sizeof(int) * (size_t)nranks, hipMemcpyDeviceToHost, c.stream);
            if (e == hipSuccess && ne == ncclSuccess) e = hipStreamSynchronize(c.stream);
            if (e != hipSuccess || ne != ncclSuccess) {
                const std::string why = std::string("comm_init: agreeing on the gather route: ") + (ne != ncclSuccess ? r->GetErrorString(ne) : hipGetErrorString(e));
                neddf_comm_release(ctx);
                return fail(ctx, NEDDF_ECOMM, why);
            }
        } else { neddf_comm_release(ctx); return rc; }
        for (int v : all) if (!v) c.in_place = false;
    }
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
    return neddf_gather_pixels_granular(ctx, d_local, n_total, 1, channels, d_all, stream);
}

int neddf_gather_pixels_granular(neddf_ctx *ctx, const float *d_local, int64_t n_total, int64_t granule, int channels, float *d_all,
                                 void *stream)
{
    if (!ctx || !d_all || n_total < 0 || channels < 1 || granule < 1) return NEDDF_EINVAL;
    CommState &c = ctx->comm;
    if (!c.comm) return fail(ctx, NEDDF_ECOMM, "gather_pixels: no communicator (neddf_comm_init)");
    if (n_total == 0) return 0;
    Rccl *r;
    if (int rc = need_rccl(ctx, r)) return rc;
    DeviceGuard guard_(ctx->device);
    int64_t lo, hi;
    neddf_shard_range_granular(n_total, granule, c.rank, c.nranks, &lo, &hi);
    if (hi > lo && !d_local) return NEDDF_EINVAL;
    int64_t pad = 0;                                                 // rows of the largest slab
    bool ragged = false;
    for (int q = 0; q < c.nranks; ++q) {
        int64_t l, h;
        neddf_shard_range_granular(n_total, granule, q, c.nranks, &l, &h);
        if (q && h - l != pad) ragged = true;
        if (h - l > pad) pad = h - l;
    }
    const size_t row = (size_t)channels * sizeof(float);
    if (c.force_ragged) ragged = true;          // test hook: one rank (or equal slabs) through the ragged routes
    // Ragged slabs (chunk-granular shards: 1 250 chunks of an 800 x 800 frame over 8 ranks are 157 or 156 each), default route:
    // equal-count all-gather through a padded staging buffer + one compaction copy per rank.  Opt-in (NEDDF_GATHER_INPLACE=1 on
    // EVERY rank, agreed at neddf_comm_init): one grouped set of broadcasts, rank q's slab from its own buffer straight to its
    // offset of every rank's d_all -- the all-gather-v idiom; no staging copy, no padding, no compaction.
    const bool in_place = ragged && c.in_place;
    if (ragged && !in_place)
        if (int rc = ensure(ctx, c.pad, (size_t)(c.nranks + 1) * pad * row)) return rc;
    // the communication stream picks up after what `stream` has enqueued so far (the render of this slab) ...
    HIPCHK(hipEventRecord(c.ready, (hipStream_t)stream));
    HIPCHK(hipStreamWaitEvent(c.stream, c.ready, 0));
    // ... and after the previous gather's consumers: the caller waited (neddf_comm_wait) before touching its buffers
    tick(ctx, c.stream, NEDDF_STAGE_GATHER, true);
    // the collective(s); an error leaves through `rc` so that the timing pair opened above is always closed
    int rc = 0;
    auto nccl_fail = [&](const char *what, ncclResult_t e_) { if (!rc) rc = fail(ctx, NEDDF_ECOMM, std::string(what) + ": " + r->GetErrorString(e_)); };
    auto hip_fail = [&](const char *what, hipError_t e_) { if (!rc) { ctx->err = std::string(what) + ": " + hipGetErrorString(e_); rc = NEDDF_EHIP; } };
    if (!ragged) {
        const ncclResult_t e_ = r->AllGather(d_local, d_all, (size_t)pad * channels, ncclFloat, (ncclComm_t)c.comm, c.stream);
        if (e_ != ncclSuccess) nccl_fail("ncclAllGather", e_);
    } else if (in_place) {
        ncclResult_t first = r->GroupStart();
        if (first == ncclSuccess) {
            for (int q = 0; q < c.nranks; ++q) {
                int64_t l, h;
                neddf_shard_range_granular(n_total, granule, q, c.nranks, &l, &h);
                if (h <= l) continue;               // more ranks than chunks: that rank contributes nothing
                char *dst = (char *)d_all + l * row;
                const ncclResult_t e_ = r->Broadcast(q == c.rank ? (const void *)d_local : (const void *)dst, dst, (size_t)(h - l) * channels, ncclFloat, q,
                                                     (ncclComm_t)c.comm, c.stream);
                if (e_ != ncclSuccess && first == ncclSuccess) first = e_;
            }
            const ncclResult_t ge = r->GroupEnd();          // always closed, also after a failed member
            if (first != ncclSuccess) nccl_fail("ncclBroadcast (grouped gather)", first);
            else if (ge != ncclSuccess) nccl_fail("ncclGroupEnd", ge);
        } else nccl_fail("ncclGroupStart", first);
    } else {
        // equal-count all-gather through [send: pad rows | recv: nranks * pad rows], then one compaction copy per rank
        char *send = (char *)c.pad.p, *recv = send + pad * row;
        hipError_t he = hipSuccess;
        if (hi > lo) he = hipMemcpyAsync(send, d_local, (size_t)(hi - lo) * row, hipMemcpyDeviceToDevice, c.stream);
        if (he != hipSuccess) hip_fail("hipMemcpyAsync (gather staging)", he);
        if (!rc) {
            const ncclResult_t e_ = r->AllGather(send, recv, (size_t)pad * channels, ncclFloat, (ncclComm_t)c.comm, c.stream);
            if (e_ != ncclSuccess) nccl_fail("ncclAllGather (staged)", e_);
        }
        for (int q = 0; q < c.nranks && !rc; ++q) {
            int64_t l, h;
            neddf_shard_range_granular(n_total, granule, q, c.nranks, &l, &h);
            if (h > l) {
                he = hipMemcpyAsync((char *)d_all + l * row, recv + (size_t)q * pad * row, (size_t)(h - l) * row, hipMemcpyDeviceToDevice, c.stream);
                if (he != hipSuccess) hip_fail("hipMemcpyAsync (gather compaction)", he);
            }
        }
    }
    tick(ctx, c.stream, NEDDF_STAGE_GATHER, false);
    if (rc) return rc;
    HIPCHK(hipEventRecord(c.done, c.stream));
    c.pending = true;
    return 0;
}

int neddf_comm_wait(neddf_ctx *ctx, void *stream)
{
    if (!ctx) return NEDDF_EINVAL;
    CommState &c = ctx->comm;
    if (c.lost) { c.lost = false; return fail(ctx, NEDDF_ECOMM, "comm_wait: the communicator was aborted while a pixel gather was in flight; its output is incomplete"); }
    if (!c.comm || !c.pending) return 0;
    DeviceGuard guard_(ctx->device);
    HIPCHK(hipStreamWaitEvent((hipStream_t)stream, c.done, 0));
    return 0;
}

int neddf_comm_wait_host(neddf_ctx *ctx, int timeout_ms)
{
    if (!ctx) return NEDDF_EINVAL;
    CommState &c = ctx->comm;
    if (c.lost) { c.lost = false; return fail(ctx, NEDDF_ECOMM, "comm_wait_host: the communicator was aborted while a pixel gather was in flight; its output is incomplete"); }
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
            c.comm = nullptr;
            // back to "no communicator" (neddf_comm_info reports 0 ranks, so a caller re-initialises instead of finding every
            // later gather refused), remembering that the gather in flight never completed
            comm_reset(c, true);
            return fail(ctx, NEDDF_ETIMEOUT, "comm_wait_host: pixel gather did not complete in " + std::to_string(timeout_ms) + " ms; communicator aborted");
        }
        std::this_thread::sleep_for(std::chrono::microseconds(200));
    }
}

}  // extern "C"
