// LoopbackComm: the multi-rank TEST transport (interface and rationale in dml_comm.h).
//
// It serves the same collectives the RCCL transport does - gather / scatter through a root
// (/root/reference/part2/2a/main.py:101,114), all_reduce (/root/reference/part2/2b/main.py:106)
// and broadcast (/root/reference/part3/main.py:137) - for ranks that are processes on one
// machine, possibly sharing one GPU.
//
// No GPU call is made off the caller's stream: the payload is copied device -> pinned staging ON
// that stream before the "ready" flag kernel and back AFTER the "wait" kernel, and the helper
// thread only moves host memory.  (A helper-side copy on a stream of its own would need a free
// hardware queue; HIP maps streams onto four, and a queue occupied by a spinning wait kernel
// would never run it.)
//
// Shared segment /dev/shm/dml_loop_<name>:  Header | world x kChunk bytes (one slot per rank).
// A collective moves its payload through the slots chunk by chunk, two cross-process barriers
// per chunk (slots written -> everyone reads -> slots free again).  Sums are formed in rank
// order from the slots, so every rank computes the same bits.
#include <fcntl.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <map>
#include <mutex>
#include <stdexcept>
#include <thread>

#include "dml_comm.h"

namespace dml {

namespace {

constexpr size_t kChunk = 4u << 20;       // one rank's slot in the shared segment
constexpr size_t kStage = 48u << 20;      // pinned staging per process
constexpr unsigned kRing = 4096;
constexpr uint64_t kMagic = 0x646d6c6c6f6f7031ull;

using Clock = std::chrono::steady_clock;

double since(Clock::time_point t0) {
    return std::chrono::duration<double>(Clock::now() - t0).count();
}

void check_hip(hipError_t e, const char* what) {
    if (e != hipSuccess)
        throw std::runtime_error(std::string("HIP ") + what + " failed: " + hipGetErrorString(e));
}

struct Header {
    std::atomic<uint64_t> magic;
    std::atomic<uint64_t> arrivals;     // barrier counter, monotonic
    std::atomic<uint32_t> abort;
    int64_t created_s;                  // CLOCK_REALTIME seconds at creation
    int32_t world;
};

float bf16_to_f32(uint16_t v) {
    uint32_t u = (uint32_t)v << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}

uint16_t f32_to_bf16(float f) {      // round to nearest even, NaN kept quiet
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}

enum Kind { ALL_REDUCE, BROADCAST, GATHER, SCATTER, STOP };

// one piece of a collective: `count` elements that fit the staging buffer
struct Op {
    Kind kind = STOP;
    size_t count = 0;
    DType dt = DT_F32;
    int root = 0;
    bool average = false;
    unsigned seq = 0;
    char* stage = nullptr;      // the issuing stream's staging buffer
};

}  // namespace

struct LoopbackComm::Impl {
    int rank, world, device;
    double timeout_s;
    std::string shm_name;
    Header* hdr = nullptr;
    char* slots = nullptr;
    size_t map_bytes = 0;
    uint64_t epoch = 0;                 // barriers passed by this rank
    unsigned* flags = nullptr;          // coherent host memory: ready[kRing] | done[kRing] | timed_out
    // pinned staging, kStage bytes per issuing stream: pieces issued on one stream follow each
    // other in stream order, pieces on different streams must not share a buffer
    std::map<hipStream_t, char*> stages;
    char* scratch = nullptr;            // host, kChunk bytes: a reduced chunk
    long long wait_ticks = 0;
    unsigned issued = 0;
    std::mutex mu;
    std::condition_variable cv;
    std::deque<Op> queue;
    std::thread helper;
    std::atomic<bool> failed{false};
    std::string error;
    bool destroyed = false;

    unsigned* ready(unsigned seq) { return flags + (seq % kRing); }
    unsigned* done(unsigned seq) { return flags + kRing + (seq % kRing); }
    unsigned* timed_out() { return flags + 2 * kRing; }
    char* slot(int r) { return slots + (size_t)r * kChunk; }

    void fail(const std::string& why) {
        if (!failed.exchange(true)) error = why;
        if (hdr) hdr->abort.store(1);
    }

    void barrier() {
        const uint64_t target = (++epoch) * (uint64_t)world;
        hdr->arrivals.fetch_add(1);
        const auto t0 = Clock::now();
        while (hdr->arrivals.load() < target) {
            if (hdr->abort.load()) throw std::runtime_error("loopback: a peer aborted");
            if (since(t0) > timeout_s)
                throw std::runtime_error("loopback: barrier timed out (peer lost?)");
            ::usleep(20);
        }
    }

    // out[i] = sum over ranks (rank order) of slot[r][i]
    void reduce_chunk(char* out_bytes, size_t n, DType dt, bool average) {
        if (dt == DT_F32) {
            float* out = (float*)out_bytes;
            for (size_t i = 0; i < n; ++i) {
                float acc = ((const float*)slot(0))[i];
                for (int r = 1; r < world; ++r) acc += ((const float*)slot(r))[i];
                out[i] = average ? acc / (float)world : acc;
            }
        } else if (dt == DT_BF16) {
            uint16_t* out = (uint16_t*)out_bytes;
            for (size_t i = 0; i < n; ++i) {
                float acc = bf16_to_f32(((const uint16_t*)slot(0))[i]);
                for (int r = 1; r < world; ++r) acc += bf16_to_f32(((const uint16_t*)slot(r))[i]);
                out[i] = f32_to_bf16(average ? acc / (float)world : acc);
            }
        } else if (dt == DT_I64) {
            int64_t* out = (int64_t*)out_bytes;
            for (size_t i = 0; i < n; ++i) {
                int64_t acc = 0;
                for (int r = 0; r < world; ++r) acc += ((const int64_t*)slot(r))[i];
                out[i] = average ? acc / world : acc;
            }
        } else {
            uint8_t* out = (uint8_t*)out_bytes;
            for (size_t i = 0; i < n; ++i) {
                unsigned acc = 0;
                for (int r = 0; r < world; ++r) acc += ((const uint8_t*)slot(r))[i];
                out[i] = (uint8_t)(average ? acc / (unsigned)world : acc);
            }
        }
    }

    // Host side of one piece.  On entry the staging buffer holds what this rank contributes
    // (layout per kind, see submit()); on exit it holds what this rank receives.
    void execute(const Op& op) {
        char* stage = op.stage;
        const size_t esz = dtype_bytes(op.dt);
        const size_t bytes_total = op.count * esz;
        for (size_t off = 0; off < bytes_total; off += kChunk) {
            const size_t bytes = bytes_total - off < kChunk ? bytes_total - off : kChunk;
            switch (op.kind) {
                case ALL_REDUCE:
                    memcpy(slot(rank), stage + off, bytes);
                    barrier();
                    reduce_chunk(scratch, bytes / esz, op.dt, op.average);
                    memcpy(stage + off, scratch, bytes);
                    barrier();
                    break;
                case BROADCAST:
                    if (rank == op.root) memcpy(slot(op.root), stage + off, bytes);
                    barrier();
                    if (rank != op.root) memcpy(stage + off, slot(op.root), bytes);
                    barrier();
                    break;
                case GATHER:        // root's staging: world regions of bytes_total each
                    if (rank != op.root) memcpy(slot(rank), stage + off, bytes);
                    barrier();
                    if (rank == op.root)
                        for (int r = 0; r < world; ++r)
                            if (r != op.root)
                                memcpy(stage + (size_t)r * bytes_total + off, slot(r), bytes);
                    barrier();
                    break;
                case SCATTER:
                    if (rank == op.root)
                        for (int r = 0; r < world; ++r)
                            if (r != op.root)
                                memcpy(slot(r), stage + (size_t)r * bytes_total + off, bytes);
                    barrier();
                    if (rank != op.root) memcpy(stage + off, slot(rank), bytes);
                    barrier();
                    break;
                case STOP:
                    break;
            }
        }
    }

    void serve() {
        (void)hipSetDevice(device);
        while (true) {
            Op op;
            {
                std::unique_lock<std::mutex> lock(mu);
                cv.wait(lock, [this] { return !queue.empty(); });
                op = queue.front();
                queue.pop_front();
            }
            if (op.kind == STOP) return;
            // wait until the issuing stream has reached the operation
            const auto t0 = Clock::now();
            volatile unsigned* rdy = ready(op.seq);
            bool ok = true;
            while ((int)(*rdy - op.seq) < 0) {
                if (failed.load() || since(t0) > timeout_s) { ok = false; break; }
                ::usleep(5);
            }
            if (!ok && !failed.load()) fail("loopback: stream never reached an enqueued collective");
            if (!failed.load()) {
                try {
                    execute(op);
                } catch (const std::exception& e) {
                    fail(e.what());
                }
            }
            // release the stream in every case (after a failure its results are garbage, and
            // wait_stream reports the error)
            __atomic_store_n(done(op.seq), op.seq, __ATOMIC_RELEASE);
        }
    }

    void copy(void* dst, const void* src, size_t bytes, hipMemcpyKind kind, hipStream_t stream) {
        check_hip(hipMemcpyAsync(dst, src, bytes, kind, stream), "hipMemcpyAsync(loopback)");
    }

    // Enqueue one piece: [copies out, issued by the caller before this] ready flag, wait kernel.
    char* stage_for(hipStream_t stream) {
        auto it = stages.find(stream);
        if (it != stages.end()) return it->second;
        char* p = nullptr;
        check_hip(hipHostMalloc((void**)&p, kStage, hipHostMallocDefault), "hipHostMalloc(stage)");
        stages[stream] = p;
        return p;
    }

    void post(Op op, hipStream_t stream) {
        op.seq = ++issued;
        op.stage = stage_for(stream);
        loopback_enqueue_signal(ready(op.seq), op.seq, stream);
        loopback_enqueue_wait(done(op.seq), op.seq, timed_out(), wait_ticks, stream);
        {
            std::lock_guard<std::mutex> lock(mu);
            queue.push_back(op);
        }
        cv.notify_one();
    }

    void check_usable() {
        if (destroyed) throw std::logic_error("LoopbackComm: used after destroy()");
        if (failed.load()) throw std::runtime_error("LoopbackComm: " + error);
    }
};

LoopbackComm::LoopbackComm(const std::string& name, int rank, int world_size, int device,
                           double timeout_s)
    : Comm(rank, world_size), impl_(new Impl) {
    if (rank < 0 || rank >= world_size) throw std::invalid_argument("LoopbackComm: bad rank");
    Impl& s = *impl_;
    s.rank = rank; s.world = world_size; s.device = device; s.timeout_s = timeout_s;
    s.shm_name = "/dml_loop_" + name;
    s.map_bytes = 4096 + (size_t)world_size * kChunk;
    check_hip(hipSetDevice(device), "hipSetDevice");
    int fd = -1;
    const auto t0 = Clock::now();
    if (rank == 0) {
        ::shm_unlink(s.shm_name.c_str());       // left behind by a crashed run
        fd = ::shm_open(s.shm_name.c_str(), O_CREAT | O_EXCL | O_RDWR, 0600);
        if (fd < 0) throw std::runtime_error("LoopbackComm: shm_open(create) failed");
        if (::ftruncate(fd, (off_t)s.map_bytes) != 0) {
            ::close(fd);
            throw std::runtime_error("LoopbackComm: ftruncate failed");
        }
    }
    while (true) {
        if (rank != 0) fd = ::shm_open(s.shm_name.c_str(), O_RDWR, 0600);
        if (fd >= 0) {
            struct stat st;
            if (::fstat(fd, &st) == 0 && (size_t)st.st_size >= s.map_bytes) {
                void* p = ::mmap(nullptr, s.map_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
                if (p != MAP_FAILED) {
                    Header* h = (Header*)p;
                    if (rank == 0) {
                        h->arrivals.store(0);
                        h->abort.store(0);
                        h->created_s = (int64_t)::time(nullptr);
                        h->world = world_size;
                        h->magic.store(kMagic);
                        s.hdr = h;
                    } else if (h->magic.load() == kMagic && h->world == world_size &&
                               (int64_t)::time(nullptr) - h->created_s <= (int64_t)timeout_s + 5) {
                        s.hdr = h;
                    } else {
                        ::munmap(p, s.map_bytes);     // not initialised yet, or a stale segment
                    }
                }
            }
            ::close(fd);
            fd = -1;
        }
        if (s.hdr) break;
        if (since(t0) > timeout_s)
            throw std::runtime_error("LoopbackComm: timed out waiting for rank 0's segment");
        ::usleep(2000);
    }
    s.slots = (char*)s.hdr + 4096;
    check_hip(hipHostMalloc((void**)&s.flags, (2 * kRing + 16) * sizeof(unsigned),
                            hipHostMallocCoherent), "hipHostMalloc(flags)");
    memset(s.flags, 0, (2 * kRing + 16) * sizeof(unsigned));
    s.scratch = new char[kChunk];
    int khz = 0;
    if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, device) != hipSuccess || khz <= 0)
        khz = 100000;
    s.wait_ticks = (long long)(timeout_s * 1e3 * khz);
    try {
        s.barrier();                      // everyone has mapped the segment
    } catch (...) {
        ::munmap(s.hdr, s.map_bytes);
        throw;
    }
    if (rank == 0) ::shm_unlink(s.shm_name.c_str());    // the mappings keep it alive
    s.helper = std::thread([this] { impl_->serve(); });
}

LoopbackComm::~LoopbackComm() {
    try {
        destroy();
    } catch (...) {
    }
    delete impl_;
}

void LoopbackComm::destroy() {
    Impl& s = *impl_;
    if (s.destroyed) return;
    s.destroyed = true;
    {
        std::lock_guard<std::mutex> lock(s.mu);
        s.queue.push_back(Op());        // STOP
    }
    s.cv.notify_one();
    if (s.helper.joinable()) s.helper.join();
    for (auto& kv : s.stages) (void)hipHostFree(kv.second);
    s.stages.clear();
    delete[] s.scratch;
    s.scratch = nullptr;
    if (s.flags) (void)hipHostFree(s.flags);
    if (s.hdr) {
        ::munmap(s.hdr, s.map_bytes);
        s.hdr = nullptr;
    }
}

int LoopbackComm::peer_count() { return impl_->hdr ? impl_->hdr->world : -1; }

void LoopbackComm::all_reduce(void* buf, size_t count, DType dt, bool average, hipStream_t stream) {
    if (count == 0) return;
    account(COLL_ALL_REDUCE, count, dt, rank_);
    Impl& s = *impl_;
    s.check_usable();
    const size_t esz = dtype_bytes(dt), per = kStage / esz;
    char* stage = s.stage_for(stream);
    for (size_t at = 0; at < count; at += per) {
        const size_t n = count - at < per ? count - at : per;
        char* p = (char*)buf + at * esz;
        s.copy(stage, p, n * esz, hipMemcpyDeviceToHost, stream);
        Op op;
        op.kind = ALL_REDUCE; op.count = n; op.dt = dt; op.average = average;
        s.post(op, stream);
        s.copy(p, stage, n * esz, hipMemcpyHostToDevice, stream);
    }
}

void LoopbackComm::broadcast(const void* send, void* recv, size_t count, DType dt, int root,
                             hipStream_t stream) {
    if (count == 0) return;
    account(COLL_BROADCAST, count, dt, root);
    Impl& s = *impl_;
    s.check_usable();
    const size_t esz = dtype_bytes(dt), per = kStage / esz;
    char* stage = s.stage_for(stream);
    for (size_t at = 0; at < count; at += per) {
        const size_t n = count - at < per ? count - at : per;
        if (rank_ == root)
            s.copy(stage, (const char*)send + at * esz, n * esz, hipMemcpyDeviceToHost, stream);
        Op op;
        op.kind = BROADCAST; op.count = n; op.dt = dt; op.root = root;
        s.post(op, stream);
        if (rank_ != root)
            s.copy((char*)recv + at * esz, stage, n * esz, hipMemcpyHostToDevice, stream);
        else if (recv != send)
            s.copy((char*)recv + at * esz, (const char*)send + at * esz, n * esz,
                   hipMemcpyDeviceToDevice, stream);
    }
}

void LoopbackComm::gather(const void* send, const std::vector<void*>& recv, size_t count, DType dt,
                          int root, hipStream_t stream) {
    if (count == 0) return;
    account(COLL_GATHER, count, dt, root);
    if (rank_ == root && (int)recv.size() != world_)
        throw std::invalid_argument("LoopbackComm::gather: root needs world_size receive slabs");
    Impl& s = *impl_;
    s.check_usable();
    const size_t esz = dtype_bytes(dt), per = kStage / esz / (size_t)world_;
    char* stage = s.stage_for(stream);
    for (size_t at = 0; at < count; at += per) {
        const size_t n = count - at < per ? count - at : per;
        if (rank_ != root)
            s.copy(stage, (const char*)send + at * esz, n * esz, hipMemcpyDeviceToHost, stream);
        Op op;
        op.kind = GATHER; op.count = n; op.dt = dt; op.root = root;
        s.post(op, stream);
        if (rank_ == root) {
            for (int r = 0; r < world_; ++r) {
                if (r != root)
                    s.copy((char*)recv[r] + at * esz, stage + (size_t)r * n * esz, n * esz,
                           hipMemcpyHostToDevice, stream);
                else if (recv[r] != send)
                    s.copy((char*)recv[r] + at * esz, (const char*)send + at * esz, n * esz,
                           hipMemcpyDeviceToDevice, stream);
            }
        }
    }
}

void LoopbackComm::scatter(void* out, const std::vector<const void*>& chunks, size_t count,
                           DType dt, int root, hipStream_t stream) {
    if (count == 0) return;
    account(COLL_SCATTER, count, dt, root);
    if (rank_ == root && (int)chunks.size() != world_)
        throw std::invalid_argument("LoopbackComm::scatter: root needs world_size chunks");
    Impl& s = *impl_;
    s.check_usable();
    const size_t esz = dtype_bytes(dt), per = kStage / esz / (size_t)world_;
    char* stage = s.stage_for(stream);
    for (size_t at = 0; at < count; at += per) {
        const size_t n = count - at < per ? count - at : per;
        if (rank_ == root)
            for (int r = 0; r < world_; ++r)
                if (r != root)
                    s.copy(stage + (size_t)r * n * esz, (const char*)chunks[r] + at * esz,
                           n * esz, hipMemcpyDeviceToHost, stream);
        Op op;
        op.kind = SCATTER; op.count = n; op.dt = dt; op.root = root;
        s.post(op, stream);
        if (rank_ != root)
            s.copy((char*)out + at * esz, stage, n * esz, hipMemcpyHostToDevice, stream);
        else if (chunks[root] != out)
            s.copy((char*)out + at * esz, (const char*)chunks[root] + at * esz, n * esz,
                   hipMemcpyDeviceToDevice, stream);
    }
}

void LoopbackComm::wait_stream(hipStream_t stream, double timeout_s) {
    Impl& s = *impl_;
    const auto t0 = Clock::now();
    while (true) {
        const hipError_t q = hipStreamQuery(stream);
        if (q == hipSuccess) break;
        if (q != hipErrorNotReady) check_hip(q, "hipStreamQuery");
        if (since(t0) > timeout_s) {
            s.fail("collective did not complete within " + std::to_string(timeout_s) +
                   " s (peer lost?)");
            throw std::runtime_error("LoopbackComm: " + s.error);
        }
        ::usleep(200);
    }
    if (s.flags && *(volatile unsigned*)s.timed_out())
        s.fail("a stream gave up waiting for its collective");
    if (s.failed.load()) throw std::runtime_error("LoopbackComm: " + s.error);
}

}  // namespace dml
