// Multi-GPU exchange of the training path: RCCL over xGMI behind the C ABI, one communicator per
// context, one process per GPU.  Replaces what detectron2's DefaultTrainer gets from torch DDP + NCCL under
// ampis/data_utils.py:135 (gradient all-reduce) and comm.synchronize() at ampis/data_utils.py:107 (barrier).
//
// librccl is opened at run time (dlopen), not linked: the library must load on hosts without a GPU (the build check), and a
// process that already holds a copy of RCCL (torch bundles one under the same soname) shares that copy instead of mapping a
// second one.  Collectives run on a stream of their own so that they overlap the backward kernels of the context's stream;
// ordering between the two streams is by HIP events only (no host synchronisation in the data path).
#include <dlfcn.h>
#include <string.h>
#include <unistd.h>

#include <rccl/rccl.h>

#include "common.h"

namespace {

struct RcclApi {
    void* handle = nullptr;
    ncclResult_t (*GetVersion)(int*) = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
RcclApi g_rccl;

int load_rccl() {
    if (g_rccl.handle) return AMP_OK;
    // AMP_RCCL_LIB names the library outright (another RCCL build; the tests' shared-memory stand-in that lets two ranks share one
    // card, tests/fake_rccl).  Otherwise: a copy this process already mapped (same soname) first, then the loader's search path, then
    // the ROCm tree.
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void* h = nullptr;
    const char* forced = getenv("AMP_RCCL_LIB");
    if (forced && *forced) {
        h = dlopen(forced, RTLD_NOW | RTLD_LOCAL);
        if (!h) { amp::set_error("amp_comm: AMP_RCCL_LIB=%s could not be opened: %s", forced, dlerror()); return AMP_ERR_STATE; }
        fprintf(stderr, "ampis_hip: AMP_RCCL_LIB=%s overrides librccl for this process (a test hook: the collectives are NOT RCCL's unless this is an RCCL build)\n", forced);
    }
    if (!h) for (const char* n : names) if ((h = dlopen(n, RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD))) break;
    if (!h) for (const char* n : names) if ((h = dlopen(n, RTLD_NOW | RTLD_LOCAL))) break;
    if (!h) { amp::set_error("amp_comm: librccl.so.1 could not be opened: %s", dlerror()); return AMP_ERR_STATE; }
    RcclApi a;
    a.handle = h;
#define AMP_SYM(field, name)                                                                     \
    a.field = reinterpret_cast<decltype(a.field)>(dlsym(h, name));                                \
    if (!a.field) { amp::set_error("amp_comm: %s missing from librccl", name); return AMP_ERR_STATE; }
    AMP_SYM(GetVersion, "ncclGetVersion")
    AMP_SYM(GetUniqueId, "ncclGetUniqueId")
    AMP_SYM(CommInitRank, "ncclCommInitRank")
    AMP_SYM(CommDestroy, "ncclCommDestroy")
    AMP_SYM(AllReduce, "ncclAllReduce")
    AMP_SYM(Broadcast, "ncclBroadcast")
    AMP_SYM(GroupStart, "ncclGroupStart")
    AMP_SYM(GroupEnd, "ncclGroupEnd")
    AMP_SYM(GetErrorString, "ncclGetErrorString")
#undef AMP_SYM
    g_rccl = a;
    return AMP_OK;
}

#define AMP_NCCL_CHECK(expr)                                                                     \
    do {                                                                                          \
        ncclResult_t _r = (expr);                                                                 \
        if (_r != ncclSuccess) {                                                                  \
            amp::set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, g_rccl.GetErrorString(_r)); \
            return AMP_ERR_HIP;                                                                   \
        }                                                                                         \
    } while (0)

}  // namespace

struct amp_comm {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1, version = 0;
    hipStream_t stream = nullptr;        // collectives run here
    hipEvent_t ev_ready = nullptr;       // compute stream -> comm stream (re-recorded per bucket; the wait is enqueued at once)
    hipEvent_t ev_done = nullptr;        // comm stream -> compute stream: everything issued so far has completed
    hipEvent_t ev_first = nullptr;       // comm stream: the first collective of the current round may start (its inputs are ready)
    hipEvent_t ev_mark = nullptr;        // compute stream: the producer finished (end of the backward pass)
    hipEvent_t ev_last = nullptr;        // comm stream: the last collective of the round has completed
    hipEvent_t ev_b0[AMP_GRAD_BUCKETS] = {}, ev_b1[AMP_GRAD_BUCKETS] = {};   // comm stream: around each bucket's grouped all-reduce
    unsigned bucket_seen = 0;            // buckets whose event pair belongs to the last round
    int* d_token = nullptr;              // barrier payload
    bool round_open = false;             // ev_first recorded for the current round
    bool pending = false;                // collectives issued since the last wait_done
    bool stats_valid = false;
};

namespace amp {

amp_comm* comm_of(amp_ctx* ctx) { return ctx ? ctx->comm : nullptr; }

// Everything the compute stream has been given so far happens-before the collectives issued after this call.
static int comm_follow_compute(amp_ctx* ctx) {
    amp_comm* c = ctx->comm;
    AMP_HIP_CHECK(hipEventRecord(c->ev_ready, ctx->stream));
    AMP_HIP_CHECK(hipStreamWaitEvent(c->stream, c->ev_ready, 0));
    if (!c->round_open) {
        AMP_HIP_CHECK(hipEventRecord(c->ev_first, c->stream));
        c->round_open = true;
    }
    return AMP_OK;
}

// In-place SUM all-reduce of `nr` float ranges of `base` as ONE grouped RCCL operation, ordered after the compute stream's work
// so far, running on the communication stream.  Nothing waits for it until comm_wait_done.
int comm_allreduce_ranges(amp_ctx* ctx, float* base, const size_t* off, const size_t* n, int nr, int slot) {
    amp_comm* c = ctx->comm;
    AMP_REQUIRE(c, "amp_comm: no communicator on this context (amp_comm_init)");
    if (nr <= 0) return AMP_OK;
    const bool first_of_round = !c->round_open;
    AMP_TRY_STATUS(comm_follow_compute(ctx));
    if (first_of_round) c->bucket_seen = 0;
    const bool timed = slot >= 0 && slot < AMP_GRAD_BUCKETS && !(c->bucket_seen >> slot & 1);
    if (timed) AMP_HIP_CHECK(hipEventRecord(c->ev_b0[slot], c->stream));
    AMP_NCCL_CHECK(g_rccl.GroupStart());
    for (int i = 0; i < nr; ++i) {
        ncclResult_t r = g_rccl.AllReduce(base + off[i], base + off[i], n[i], ncclFloat, ncclSum, c->comm, c->stream);
        if (r != ncclSuccess) {
            (void)g_rccl.GroupEnd();
            amp::set_error("amp_comm: ncclAllReduce of range %d (%zu floats) -> %s", i, n[i], g_rccl.GetErrorString(r));
            return AMP_ERR_HIP;
        }
    }
    AMP_NCCL_CHECK(g_rccl.GroupEnd());
    if (timed) { AMP_HIP_CHECK(hipEventRecord(c->ev_b1[slot], c->stream)); c->bucket_seen |= 1u << slot; }
    c->pending = true;
    return AMP_OK;
}

// The producer (the backward pass) is complete on the compute stream: reference point of the exposed-communication time.
int comm_mark_producer_end(amp_ctx* ctx) {
    amp_comm* c = ctx->comm;
    if (!c) return AMP_OK;
    AMP_HIP_CHECK(hipEventRecord(c->ev_mark, ctx->stream));
    AMP_HIP_CHECK(hipEventRecord(c->ev_last, c->stream));
    c->stats_valid = c->round_open;
    c->round_open = false;
    return AMP_OK;
}

// The compute stream waits (on the device, not the host) for every collective issued so far.
int comm_wait_done(amp_ctx* ctx) {
    amp_comm* c = ctx->comm;
    if (!c || !c->pending) return AMP_OK;
    AMP_HIP_CHECK(hipEventRecord(c->ev_done, c->stream));
    AMP_HIP_CHECK(hipStreamWaitEvent(ctx->stream, c->ev_done, 0));
    c->pending = false;
    return AMP_OK;
}

// MAX of one device int over the ranks, result visible to the host after return (the f16x3 range flag: every rank must take
// the same re-run decision, or the collectives of the re-run would not match).
int comm_agree_flag(amp_ctx* ctx, int* d_flag) {
    amp_comm* c = ctx->comm;
    if (!c || c->world == 1) return AMP_OK;
    AMP_HIP_CHECK(hipEventRecord(c->ev_ready, ctx->stream));
    AMP_HIP_CHECK(hipStreamWaitEvent(c->stream, c->ev_ready, 0));
    AMP_NCCL_CHECK(g_rccl.AllReduce(d_flag, d_flag, 1, ncclInt, ncclMax, c->comm, c->stream));
    AMP_HIP_CHECK(hipStreamSynchronize(c->stream));
    return AMP_OK;
}

}  // namespace amp

extern "C" {

int amp_comm_unique_id(unsigned char* id_h) {
    AMP_REQUIRE(id_h, "amp_comm_unique_id: null argument");
    static_assert(sizeof(ncclUniqueId) == AMP_COMM_ID_BYTES, "AMP_COMM_ID_BYTES must equal sizeof(ncclUniqueId)");
    AMP_TRY_STATUS(load_rccl());
    ncclUniqueId id;
    AMP_NCCL_CHECK(g_rccl.GetUniqueId(&id));
    memcpy(id_h, &id, sizeof(id));
    return AMP_OK;
}

int amp_comm_init(amp_ctx* ctx, int rank, int world, const unsigned char* id_h) {
    AMP_REQUIRE(ctx && id_h && world >= 1 && rank >= 0 && rank < world, "amp_comm_init: bad argument (rank %d of %d)", rank, world);
    AMP_REQUIRE(!ctx->comm, "amp_comm_init: this context already has a communicator");
    AMP_TRY_STATUS(load_rccl());
    AMP_HIP_CHECK(hipSetDevice(ctx->device));
    amp_comm* c = new amp_comm();
    c->rank = rank; c->world = world;
    (void)g_rccl.GetVersion(&c->version);
    ncclUniqueId id;
    memcpy(&id, id_h, sizeof(id));
    // RCCL prints a version banner ("RCCL version : ...", HIP / ROCm versions, host, library path) on STDOUT when the first
    // communicator comes up.  A host that owns stdout (bench.py prints exactly one JSON line there) must not get foreign lines on it:
    // while RCCL initialises, file descriptor 1 points at stderr.
    fflush(stdout);
    const int saved_stdout = dup(1);
    if (saved_stdout >= 0) (void)dup2(2, 1);
    ncclResult_t r = g_rccl.CommInitRank(&c->comm, world, id, rank);
    fflush(stdout);
    if (saved_stdout >= 0) { (void)dup2(saved_stdout, 1); (void)close(saved_stdout); }
    if (r != ncclSuccess) {
        amp::set_error("amp_comm_init: ncclCommInitRank(rank %d of %d, device %d) -> %s", rank, world, ctx->device, g_rccl.GetErrorString(r));
        delete c;
        return AMP_ERR_HIP;
    }
    bool ok = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) == hipSuccess;
    ok = ok && hipEventCreateWithFlags(&c->ev_ready, hipEventDisableTiming) == hipSuccess;
    ok = ok && hipEventCreate(&c->ev_done) == hipSuccess && hipEventCreate(&c->ev_first) == hipSuccess && hipEventCreate(&c->ev_mark) == hipSuccess && hipEventCreate(&c->ev_last) == hipSuccess;
    for (int b = 0; b < AMP_GRAD_BUCKETS && ok; ++b) ok = hipEventCreate(&c->ev_b0[b]) == hipSuccess && hipEventCreate(&c->ev_b1[b]) == hipSuccess;
    ok = ok && hipMalloc(&c->d_token, 16) == hipSuccess && hipMemset(c->d_token, 0, 16) == hipSuccess;
    if (!ok) {
        amp::set_error("amp_comm_init: stream / event / token allocation failed");
        (void)g_rccl.CommDestroy(c->comm);
        delete c;
        return AMP_ERR_HIP;
    }
    ctx->comm = c;
    return AMP_OK;
}

int amp_comm_destroy(amp_ctx* ctx) {
    AMP_REQUIRE(ctx, "amp_comm_destroy: null ctx");
    amp_comm* c = ctx->comm;
    if (!c) return AMP_OK;
    (void)hipStreamSynchronize(c->stream);
    (void)g_rccl.CommDestroy(c->comm);
    (void)hipEventDestroy(c->ev_ready); (void)hipEventDestroy(c->ev_done); (void)hipEventDestroy(c->ev_first); (void)hipEventDestroy(c->ev_mark); (void)hipEventDestroy(c->ev_last);
    for (int b = 0; b < AMP_GRAD_BUCKETS; ++b) { if (c->ev_b0[b]) (void)hipEventDestroy(c->ev_b0[b]); if (c->ev_b1[b]) (void)hipEventDestroy(c->ev_b1[b]); }
    (void)hipFree(c->d_token);
    (void)hipStreamDestroy(c->stream);
    delete c;
    ctx->comm = nullptr;
    return AMP_OK;
}

int amp_comm_info(amp_ctx* ctx, int* rank, int* world, int* rccl_version) {
    AMP_REQUIRE(ctx, "amp_comm_info: null ctx");
    const amp_comm* c = ctx->comm;
    if (rank) *rank = c ? c->rank : 0;
    if (world) *world = c ? c->world : 0;          /* 0 = no communicator */
    if (rccl_version) *rccl_version = c ? c->version : 0;
    return AMP_OK;
}

int amp_barrier(amp_ctx* ctx) {
    AMP_REQUIRE(ctx && ctx->comm, "amp_barrier: no communicator on this context (amp_comm_init)");
    amp_comm* c = ctx->comm;
    AMP_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    AMP_NCCL_CHECK(g_rccl.AllReduce(c->d_token, c->d_token, 1, ncclInt, ncclSum, c->comm, c->stream));
    AMP_HIP_CHECK(hipStreamSynchronize(c->stream));
    AMP_HIP_CHECK(hipMemsetAsync(c->d_token, 0, 16, c->stream));
    return AMP_OK;
}

int amp_allreduce(amp_ctx* ctx, void* buf, size_t count, int dtype, int op) {
    AMP_REQUIRE(ctx && ctx->comm && buf, "amp_allreduce: no communicator on this context, or null buffer");
    AMP_REQUIRE(dtype >= AMP_F32 && dtype <= AMP_I32 && (op == AMP_SUM || op == AMP_MAX), "amp_allreduce: bad dtype / op");
    amp_comm* c = ctx->comm;
    const ncclDataType_t dt = dtype == AMP_F32 ? ncclFloat : dtype == AMP_F64 ? ncclDouble : ncclInt;
    AMP_HIP_CHECK(hipEventRecord(c->ev_ready, ctx->stream));
    AMP_HIP_CHECK(hipStreamWaitEvent(c->stream, c->ev_ready, 0));
    AMP_NCCL_CHECK(g_rccl.AllReduce(buf, buf, count, dt, op == AMP_SUM ? ncclSum : ncclMax, c->comm, c->stream));
    c->pending = true;
    return amp::comm_wait_done(ctx);
}

int amp_comm_stats(amp_ctx* ctx, float* exposed_ms, float* span_ms) {
    AMP_REQUIRE(ctx && ctx->comm && exposed_ms && span_ms, "amp_comm_stats: no communicator on this context, or null argument");
    amp_comm* c = ctx->comm;
    *exposed_ms = *span_ms = 0.f;
    if (!c->stats_valid) return AMP_OK;
    AMP_HIP_CHECK(hipEventSynchronize(c->ev_last));
    AMP_HIP_CHECK(hipEventSynchronize(c->ev_mark));
    float tail = 0.f;
    // ev_mark (compute stream: the backward pass has finished) -> ev_last (comm stream: the last bucket has been reduced)
    hipError_t e = hipEventElapsedTime(&tail, c->ev_mark, c->ev_last);
    if (e != hipSuccess) { (void)hipGetLastError(); tail = 0.f; }
    *exposed_ms = tail > 0.f ? tail : 0.f;
    AMP_HIP_CHECK(hipEventElapsedTime(span_ms, c->ev_first, c->ev_last));
    return AMP_OK;
}

int amp_comm_wait(amp_ctx* ctx) {
    AMP_REQUIRE(ctx, "amp_comm_wait: null ctx");
    return amp::comm_wait_done(ctx);
}

int amp_comm_bucket_stats(amp_ctx* ctx, float us[AMP_GRAD_BUCKETS]) {
    AMP_REQUIRE(ctx && ctx->comm && us, "amp_comm_bucket_stats: no communicator on this context, or null argument");
    amp_comm* c = ctx->comm;
    for (int b = 0; b < AMP_GRAD_BUCKETS; ++b) {
        us[b] = -1.f;                                   /* not exchanged in the last round */
        if (!(c->bucket_seen >> b & 1)) continue;
        AMP_HIP_CHECK(hipEventSynchronize(c->ev_b1[b]));
        float ms = 0.f;
        AMP_HIP_CHECK(hipEventElapsedTime(&ms, c->ev_b0[b], c->ev_b1[b]));
        us[b] = ms * 1000.f;
    }
    return AMP_OK;
}

int amp_comm_broadcast(amp_ctx* ctx, void* buf, size_t bytes, int root) {
    AMP_REQUIRE(ctx && ctx->comm && buf, "amp_comm_broadcast: no communicator on this context, or null buffer");
    amp_comm* c = ctx->comm;
    AMP_REQUIRE(root >= 0 && root < c->world, "amp_comm_broadcast: root %d of %d ranks", root, c->world);
    if (bytes == 0) return AMP_OK;
    AMP_HIP_CHECK(hipEventRecord(c->ev_ready, ctx->stream));
    AMP_HIP_CHECK(hipStreamWaitEvent(c->stream, c->ev_ready, 0));
    AMP_NCCL_CHECK(g_rccl.Broadcast(buf, buf, bytes, ncclChar, root, c->comm, c->stream));
    c->pending = true;
    return amp::comm_wait_done(ctx);
}

/* ---- the bucket plan: host-only, no device call (tests drive it without a GPU) ---- */

int amp_grad_bucket_of(const char* name) {
    if (!name) return -1;
    auto starts = [&](const char* p) { return strncmp(name, p, strlen(p)) == 0; };
    if (starts("roi_heads.mask_head")) return 0;                                             // first gradients the backward pass completes
    if (starts("roi_heads.box_head") || starts("roi_heads.box_predictor")) return 1;
    if (starts("proposal_generator")) return 2;
    if (starts("backbone.fpn_")) return 3;
    if (starts("backbone.bottom_up.res5")) return 4;
    if (starts("backbone.bottom_up.res4")) return 5;
    if (starts("backbone.bottom_up.res3")) return 6;                                         // last
    return -1;                                                                               // stem, res2: frozen (FREEZE_AT = 2)
}

int amp_plan_grad_buckets(int ntensors, const int* bucket, const size_t* off, const size_t* n, size_t max_gap, int cap,
                          int* out_bucket, size_t* out_off, size_t* out_n, int* out_count) {
    AMP_REQUIRE(ntensors >= 0 && bucket && off && n && out_bucket && out_off && out_n && out_count && cap >= 0, "amp_plan_grad_buckets: bad argument");
    int cnt = 0;
    std::vector<std::pair<size_t, size_t>> r;
    for (int b = 0; b < AMP_GRAD_BUCKETS; ++b) {
        r.clear();
        for (int t = 0; t < ntensors; ++t) if (bucket[t] == b && n[t] > 0) r.push_back({off[t], n[t]});
        std::sort(r.begin(), r.end());
        size_t i = 0;
        while (i < r.size()) {
            size_t lo = r[i].first, hi = r[i].first + r[i].second;
            size_t j = i + 1;
            while (j < r.size() && r[j].first <= hi + max_gap) {
                hi = std::max(hi, r[j].first + r[j].second);
                ++j;
            }
            AMP_REQUIRE(cnt < cap, "amp_plan_grad_buckets: more than %d ranges", cap);
            out_bucket[cnt] = b; out_off[cnt] = lo; out_n[cnt] = hi - lo;
            ++cnt;
            i = j;
        }
    }
    // a gap that was bridged must not belong to a tensor of ANOTHER bucket (it would be reduced twice)
    for (int a = 0; a < cnt; ++a)
        for (int t = 0; t < ntensors; ++t)
            if (bucket[t] >= 0 && bucket[t] != out_bucket[a] && n[t] > 0 && off[t] < out_off[a] + out_n[a] && off[t] + n[t] > out_off[a]) {
                amp::set_error("amp_plan_grad_buckets: range %d of bucket %d overlaps tensor %d of bucket %d (max_gap too large for this layout)",
                               a, out_bucket[a], t, bucket[t]);
                return AMP_ERR_ARG;
            }
    *out_count = cnt;
    return AMP_OK;
}

}  // extern "C"
