// capi_internal.h -- context layout shared by the C-ABI translation units
// (neddf_capi.hip: inference path, train_capi.hip: training step).  Not installed.
#pragma once
#include "../../include/neddf_hip.h"
#include "kernels.h"

#include <stdlib.h>
#include <string>
#include <vector>

using namespace neddf;

struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    void *base = nullptr;       // NEDDF_GUARD=1: the allocation starts one guard band in front of p
};

// NEDDF_GUARD=1 (a debug mode, tests/test_gpu_multi.py::test_workspace_guard_bands): every workspace of a context is allocated at its
// EXACT requested size between two poisoned guard bands, every carve of the render arena is followed by one, and
// neddf_debug_check_guards() counts the band bytes a kernel has overwritten -- the bounds probe that stands in for a GPU-side
// AddressSanitizer run (ROCm's ASan runtime does not start beside an uninstrumented python on this image, tools/asan_probe.sh).
constexpr size_t kGuardBytes = 4096, kCarveGuardBytes = 256;
constexpr int kGuardByte = 0xA5;
static inline bool guard_mode()
{
    static const bool on = [] { const char *e = getenv("NEDDF_GUARD"); return e && atoi(e) != 0; }();
    return on;
}
struct GuardBand { const void *p; size_t bytes; };

struct Field {
    bool valid = false;
    neddf_field_desc d{};
    float aux_grad_scale = 1.1f, distance_range_max = 2.0f;
    float lowpass[10];
    DevBuf blob;
    hipEvent_t last_use = nullptr;       // recorded after the last launch that reads `blob` (neddf_set_field waits on it)
    bool last_use_recorded = false;      // ... and the stream that recorded it last: another stream waits for it before re-recording
    hipStream_t last_use_stream = nullptr;
    DdfArgs ddf{};
    ColArgs col{};
    NerfArgs nerf{};
};

struct EventPair {
    hipEvent_t a, b;
    int which;                   // NEDDF_STAGE_*
};

// Multi-GPU state of a context (comm_capi.hip): one RCCL communicator, a communication stream of its own and the event
// pair that orders it against the caller's compute stream.
struct CommState {
    void *comm = nullptr;        // ncclComm_t
    int rank = 0, nranks = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ready = nullptr;  // recorded on the compute stream: the local slab is complete
    hipEvent_t done = nullptr;   // recorded on the communication stream: the gathered pixels are complete
    bool pending = false;
    bool lost = false;           // the communicator was aborted with a gather in flight: the next wait reports it
    DevBuf pad;                  // [nranks * pad_rows * channels] staging for ragged slabs
    bool in_place = false;       // ragged slabs by one grouped set of broadcasts instead of the padded staging route: decided ONCE
                                 // at neddf_comm_init from every rank's own capability / NEDDF_GATHER_INPLACE (all ranks must agree)
    bool force_ragged = false;   // NEDDF_GATHER_FORCE_RAGGED=1 (test hook): equal slabs take the ragged route too
};

// Every entry point that launches or allocates runs on the ctx's device and leaves the caller's current device as it
// found it (a torch process may have another device current).
struct DeviceGuard {
    int prev = -1;
    bool switched = false;
    explicit DeviceGuard(int dev)
    {
        if (hipGetDevice(&prev) == hipSuccess && prev != dev) switched = hipSetDevice(dev) == hipSuccess;
    }
    ~DeviceGuard() { if (switched) (void)hipSetDevice(prev); }
    DeviceGuard(const DeviceGuard &) = delete;
    DeviceGuard &operator=(const DeviceGuard &) = delete;
};

struct neddf_ctx {
    int device = 0;
    int cus = 256;
    std::string err;
    Field field[NEDDF_NUM_SLOTS];
    DevBuf features, ptaux, scratch, arena, flags, sched;
    int64_t handoff_chunk = 0;   // field_forward: launch size last settled for a hand-off of `handoff_row` bytes per point (0: none yet) --
    size_t handoff_row = 0;      // the free-memory probe runs when the hand-off would have to GROW, not on every call
    DevBuf rflags;               // importance resampling: one NaN-fallback flag per group of rays
    DevBuf rev_scratch;          // reverse-mode distance kernel: per-workgroup y' of every layer + encoding Jacobian
    DevBuf tpack, ttmp;          // training step: packed weights of the layer in flight, gradient ping-pong buffers
    DevBuf tamax;                // training step: max |dZ| of every gradient matrix of a backward pass (split-fp16 operand range)
    std::vector<GuardBand> carve_guards;      // NEDDF_GUARD=1: the bands behind the carves of the last render call
    bool timing = false;
    std::vector<EventPair> events;
    std::vector<EventPair> pool;
    CommState comm;
};

// hipEvent pair around one stage launch (only while neddf_set_timing is on); `which` = NEDDF_STAGE_*
static inline void tick(neddf_ctx *ctx, hipStream_t s, int which, bool begin)
{
    if (!ctx->timing) return;
    if (begin) {
        EventPair e;
        if (!ctx->pool.empty()) { e = ctx->pool.back(); ctx->pool.pop_back(); }
        else { (void)hipEventCreate(&e.a); (void)hipEventCreate(&e.b); }
        e.which = which;
        (void)hipEventRecord(e.a, s);
        ctx->events.push_back(e);
    } else {
        (void)hipEventRecord(ctx->events.back().b, s);
    }
}

#define HIPCHK(call)                                                                          \
    do {                                                                                      \
        hipError_t e_ = (call);                                                               \
        if (e_ != hipSuccess) {                                                               \
            ctx->err = std::string(#call) + ": " + hipGetErrorString(e_);                     \
            return NEDDF_EHIP;                                                                \
        }                                                                                     \
    } while (0)

static inline int fail(neddf_ctx *ctx, int code, const std::string &msg)
{
    ctx->err = msg;
    return code;
}

static inline int ensure(neddf_ctx *ctx, DevBuf &b, size_t bytes)
{
    if (b.cap >= bytes && !(guard_mode() && b.cap != ((bytes + 15) & ~(size_t)15))) return 0;
    if (b.p) {
        HIPCHK(hipDeviceSynchronize());      // nothing in flight may still use the old block
        HIPCHK(hipFree(b.base ? b.base : b.p));
        b.p = nullptr; b.cap = 0; b.base = nullptr;
        if (&b == &ctx->arena) ctx->carve_guards.clear();      // the bands behind the last render's carves lived in the freed block
    }
    if (guard_mode()) {         // exact size (rounded to 16 B) between two poisoned bands: one element past either end lands in a band
        const size_t want = (bytes + 15) & ~(size_t)15;
        HIPCHK(hipMalloc(&b.base, want + 2 * kGuardBytes));
        HIPCHK(hipMemset(b.base, kGuardByte, kGuardBytes));
        HIPCHK(hipMemset((char *)b.base + kGuardBytes + want, kGuardByte, kGuardBytes));
        b.p = (char *)b.base + kGuardBytes;
        b.cap = want;
        return 0;
    }
    size_t want = bytes + bytes / 8;
    HIPCHK(hipMalloc(&b.p, want));
    b.cap = want;
    return 0;
}


static inline int roundup(int x, int m) { return (x + m - 1) / m * m; }

static inline bool in_skips(const neddf_field_desc &d, int id)
{
    for (int i = 0; i < d.n_skips; ++i) if (d.skips[i] == id) return true;
    return false;
}

static inline void fill_enc(EncodeDesc &e, const Field &f)
{
    e.E = f.d.embed_pos_rank; e.Ed = f.d.embed_dir_rank;
    e.KH = roundup(3 * e.E, 4); e.KD = roundup(3 * e.Ed, 4);
    for (int i = 0; i < 10; ++i) e.lowpass[i] = f.lowpass[i];
}
