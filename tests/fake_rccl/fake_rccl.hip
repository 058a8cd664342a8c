// Test stand-in for librccl (selected with AMP_RCCL_LIB): the nine entry points ampis_amd/csrc/comm.hip binds, implemented over POSIX
// shared memory so that TWO OR MORE RANKS CAN SHARE ONE GPU (RCCL itself refuses two ranks on one device, and the GPU box has one card).
// It keeps RCCL's stream semantics, which is what the library's event / stream ordering is tested against:
//   * a collective is ENQUEUED: the call returns at once; the data is read only when the communication stream reaches the operation
//     (hipStreamWriteValue32 tells a worker thread) and the stream proceeds only when the result is in place (hipStreamWaitValue32);
//   * collectives of a communicator complete in issue order; every rank must issue the same sequence (a mismatch in kind, count or
//     type is detected and reported through ncclGetErrorString / FAKE_RCCL errors on stderr);
//   * sums run in rank order 0..N-1 on every rank: results are bitwise identical across the ranks.
// Not a performance model: the exchange is D2H -> shared memory -> H2D.  Nothing in ampis_amd/ knows about this file.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <mutex>
#include <thread>
#include <vector>

#include <fcntl.h>
#include <sys/mman.h>
#include <unistd.h>

namespace {

constexpr int MAX_RANKS = 8;
constexpr size_t CHUNK = (size_t)8 << 20;       // bytes per rank and round
constexpr int TIMEOUT_S = 60;

struct Header {
    std::atomic<unsigned long long> barrier;   // monotonic: barrier k is passed when the counter reaches k * world
    std::atomic<int> error;
    std::atomic<unsigned long long> sig[MAX_RANKS];     // what each rank thinks operation #seq is (mismatch detection)
};

struct Job { unsigned seq; int kind; const void* send; void* recv; size_t count; ncclDataType_t dt; int op_or_root; };

size_t dt_size(ncclDataType_t dt) {
    switch (dt) {
        case ncclChar: case ncclUint8: return 1;
        case ncclFloat: case ncclInt: case ncclUint32: return 4;
        case ncclDouble: case ncclInt64: case ncclUint64: return 8;
        default: return 0;
    }
}

struct IdPayload { char magic[8]; unsigned long long token; };

}  // namespace

struct ncclComm {
    int rank = 0, world = 1, device = 0;
    char shm_name[64] = {0};
    unsigned char* base = nullptr;
    size_t shm_bytes = 0;
    Header* hdr = nullptr;
    unsigned long long barriers_done = 0;
    bool use_stream_ops = true;
    unsigned* flags = nullptr;                  // pinned host: [0] ready (written by the stream), [1] done (written by the worker)
    hipStream_t copy_stream = nullptr;
    std::thread worker;
    std::mutex mu;
    std::condition_variable cv;
    std::deque<Job> jobs;
    bool stop = false;
    unsigned seq = 0;
    std::atomic<int> failed{0};

    unsigned char* slot(int r) { return base + 4096 + (size_t)r * CHUNK; }

    bool barrier() {
        ++barriers_done;
        hdr->barrier.fetch_add(1, std::memory_order_acq_rel);
        const unsigned long long want = barriers_done * (unsigned long long)world;
        const auto t0 = std::chrono::steady_clock::now();
        while (hdr->barrier.load(std::memory_order_acquire) < want) {
            if (hdr->error.load()) return false;
            if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(TIMEOUT_S)) {
                fprintf(stderr, "[fake_rccl] rank %d: a peer did not arrive within %d s (collective sequences differ?)\n", rank, TIMEOUT_S);
                hdr->error.store(1);
                return false;
            }
            std::this_thread::yield();
        }
        return true;
    }

    template <typename T> static void reduce(T* acc, const T* x, size_t n, bool is_max) {
        if (is_max) for (size_t i = 0; i < n; ++i) acc[i] = x[i] > acc[i] ? x[i] : acc[i];
        else for (size_t i = 0; i < n; ++i) acc[i] += x[i];
    }

    bool d2h(void* h, const void* d, size_t n) {
        return hipMemcpyAsync(h, d, n, hipMemcpyDeviceToHost, copy_stream) == hipSuccess && hipStreamSynchronize(copy_stream) == hipSuccess;
    }
    bool h2d(void* d, const void* h, size_t n) {
        return hipMemcpyAsync(d, h, n, hipMemcpyHostToDevice, copy_stream) == hipSuccess && hipStreamSynchronize(copy_stream) == hipSuccess;
    }

    bool run(const Job& j) {
        const size_t es = dt_size(j.dt), bytes = j.count * es;
        // every rank must be executing the same operation
        const unsigned long long sig = ((unsigned long long)j.kind << 60) ^ ((unsigned long long)j.dt << 52) ^ ((unsigned long long)(j.op_or_root & 0xff) << 44) ^ (unsigned long long)j.count;
        hdr->sig[rank].store(sig);
        if (!barrier()) return false;
        for (int r = 0; r < world; ++r)
            if (hdr->sig[r].load() != sig) {
                fprintf(stderr, "[fake_rccl] rank %d: operation #%u differs from rank %d's (kind %d, %zu elements here)\n", rank, j.seq, r, j.kind, j.count);
                hdr->error.store(2);
            }
        if (!barrier()) return false;
        if (hdr->error.load()) return false;
        std::vector<unsigned char> acc;
        for (size_t o = 0; o < bytes; o += CHUNK) {
            const size_t nb = bytes - o < CHUNK ? bytes - o : CHUNK;
            const unsigned char* src = static_cast<const unsigned char*>(j.send) + o;
            unsigned char* dst = static_cast<unsigned char*>(j.recv) + o;
            if (j.kind == 0) {            // all-reduce
                if (!d2h(slot(rank), src, nb)) return false;
                if (!barrier()) return false;
                acc.assign(slot(0), slot(0) + nb);
                const size_t n = nb / es;
                const bool mx = j.op_or_root == (int)ncclMax;
                for (int r = 1; r < world; ++r) {
                    if (j.dt == ncclFloat) reduce(reinterpret_cast<float*>(acc.data()), reinterpret_cast<const float*>(slot(r)), n, mx);
                    else if (j.dt == ncclDouble) reduce(reinterpret_cast<double*>(acc.data()), reinterpret_cast<const double*>(slot(r)), n, mx);
                    else if (j.dt == ncclInt) reduce(reinterpret_cast<int*>(acc.data()), reinterpret_cast<const int*>(slot(r)), n, mx);
                    else return false;
                }
                if (!h2d(dst, acc.data(), nb)) return false;
                if (!barrier()) return false;          // the slots may be overwritten by the next round
            } else {                      // broadcast
                if (rank == j.op_or_root && !d2h(slot(rank), src, nb)) return false;
                if (!barrier()) return false;
                if (rank != j.op_or_root || src != dst) { if (!h2d(dst, slot(j.op_or_root), nb)) return false; }
                if (!barrier()) return false;
            }
        }
        return true;
    }

    void loop() {
        (void)hipSetDevice(device);
        for (;;) {
            Job j;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return stop || !jobs.empty(); });
                if (jobs.empty()) return;
                j = jobs.front();
                jobs.pop_front();
            }
            if (use_stream_ops) {         // wait until the communication stream has reached this operation
                const auto t0 = std::chrono::steady_clock::now();
                while ((int)(__atomic_load_n(&flags[0], __ATOMIC_ACQUIRE) - j.seq) < 0) {
                    if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(4 * TIMEOUT_S)) { failed.store(1); break; }
                    std::this_thread::yield();
                }
            }
            if (!failed.load() && !run(j)) { failed.store(1); fprintf(stderr, "[fake_rccl] rank %d: operation #%u failed\n", rank, j.seq); }
            __atomic_store_n(&flags[1], j.seq, __ATOMIC_RELEASE);   // ALWAYS release the stream, also after a failure
        }
    }
};

extern "C" {

ncclResult_t ncclGetVersion(int* v) { if (v) *v = 99999; return ncclSuccess; }      // 9.99.99: recognisably not a real RCCL

const char* ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "no error" : "fake_rccl: operation failed (see stderr)"; }

ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
    static_assert(sizeof(IdPayload) <= sizeof(ncclUniqueId), "id payload");
    memset(id, 0, sizeof(*id));
    IdPayload p;
    memcpy(p.magic, "AMPFAKE", 8);
    p.token = ((unsigned long long)getpid() << 32) ^ (unsigned long long)std::chrono::steady_clock::now().time_since_epoch().count();
    memcpy(id, &p, sizeof(p));
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t* out, int world, ncclUniqueId id, int rank) {
    if (!out || world < 1 || world > MAX_RANKS || rank < 0 || rank >= world) return ncclInvalidArgument;
    IdPayload p;
    memcpy(&p, &id, sizeof(p));
    if (memcmp(p.magic, "AMPFAKE", 8) != 0) return ncclInvalidArgument;
    ncclComm* c = new ncclComm();
    c->rank = rank; c->world = world;
    if (hipGetDevice(&c->device) != hipSuccess) { delete c; return ncclUnhandledCudaError; }
    snprintf(c->shm_name, sizeof(c->shm_name), "/amp_fake_rccl_%llx", p.token);
    c->shm_bytes = 4096 + (size_t)world * CHUNK;
    int fd = -1;
    if (rank == 0) {
        fd = shm_open(c->shm_name, O_CREAT | O_EXCL | O_RDWR, 0600);
        if (fd >= 0 && ftruncate(fd, (off_t)c->shm_bytes) != 0) { close(fd); fd = -1; }
    } else {
        for (int tries = 0; tries < TIMEOUT_S * 100 && fd < 0; ++tries) {
            fd = shm_open(c->shm_name, O_RDWR, 0600);
            if (fd >= 0) {       // rank 0 may not have sized it yet
                off_t sz = lseek(fd, 0, SEEK_END);
                if (sz < (off_t)c->shm_bytes) { close(fd); fd = -1; }
            }
            if (fd < 0) usleep(10000);
        }
    }
    if (fd < 0) { fprintf(stderr, "[fake_rccl] rank %d: shared memory %s unavailable\n", rank, c->shm_name); delete c; return ncclSystemError; }
    c->base = static_cast<unsigned char*>(mmap(nullptr, c->shm_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0));
    close(fd);
    if (c->base == MAP_FAILED) { delete c; return ncclSystemError; }
    c->hdr = reinterpret_cast<Header*>(c->base);      // a fresh shm object is zero-filled: barrier 0, error 0
    int can = 0;
    (void)hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, c->device);
    c->use_stream_ops = can != 0 && getenv("FAKE_RCCL_BLOCKING") == nullptr;
    if (hipHostMalloc(reinterpret_cast<void**>(&c->flags), 64, hipHostMallocCoherent) != hipSuccess ||
        hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking) != hipSuccess) { delete c; return ncclUnhandledCudaError; }
    c->flags[0] = c->flags[1] = 0;
    if (!c->barrier()) { delete c; return ncclSystemError; }       // everybody has mapped the segment
    if (rank == 0) shm_unlink(c->shm_name);                         // the mappings keep it alive; nothing is left behind in /dev/shm
    c->worker = std::thread([c] { c->loop(); });
    *out = c;
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t c) {
    if (!c) return ncclSuccess;
    { std::lock_guard<std::mutex> lk(c->mu); c->stop = true; }
    c->cv.notify_all();
    if (c->worker.joinable()) c->worker.join();
    if (c->copy_stream) (void)hipStreamDestroy(c->copy_stream);
    if (c->flags) (void)hipHostFree(c->flags);
    if (c->base) munmap(c->base, c->shm_bytes);
    delete c;
    return ncclSuccess;
}

static ncclResult_t enqueue(ncclComm_t c, int kind, const void* send, void* recv, size_t count, ncclDataType_t dt, int op_or_root, hipStream_t stream) {
    if (!c || !send || !recv || dt_size(dt) == 0) return ncclInvalidArgument;
    if (c->failed.load()) return ncclInternalError;
    Job j{++c->seq, kind, send, recv, count, dt, op_or_root};
    if (c->use_stream_ops) {
        if (hipStreamWriteValue32(stream, &c->flags[0], j.seq, 0) != hipSuccess) return ncclUnhandledCudaError;
        { std::lock_guard<std::mutex> lk(c->mu); c->jobs.push_back(j); }
        c->cv.notify_all();
        if (hipStreamWaitValue32(stream, &c->flags[1], j.seq, hipStreamWaitValueGte, 0xffffffffu) != hipSuccess) return ncclUnhandledCudaError;
        return ncclSuccess;
    }
    // blocking variant (no stream memory operations on this device): the host waits for the stream, then runs the operation
    if (hipStreamSynchronize(stream) != hipSuccess) return ncclUnhandledCudaError;
    return c->run(j) ? ncclSuccess : ncclInternalError;
}

ncclResult_t ncclAllReduce(const void* send, void* recv, size_t count, ncclDataType_t dt, ncclRedOp_t op, ncclComm_t c, hipStream_t stream) {
    if (op != ncclSum && op != ncclMax) return ncclInvalidArgument;
    return enqueue(c, 0, send, recv, count, dt, (int)op, stream);
}

ncclResult_t ncclBroadcast(const void* send, void* recv, size_t count, ncclDataType_t dt, int root, ncclComm_t c, hipStream_t stream) {
    if (!c || root < 0 || root >= c->world) return ncclInvalidArgument;
    return enqueue(c, 1, send, recv, count, dt, root, stream);
}

ncclResult_t ncclGroupStart() { return ncclSuccess; }
ncclResult_t ncclGroupEnd() { return ncclSuccess; }

}  // extern "C"
