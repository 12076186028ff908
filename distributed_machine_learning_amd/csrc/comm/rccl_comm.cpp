// RCCL communicator: the collectives of the three sync modes on xGMI.
//
// Call sites being served: dist.gather / dist.scatter through rank 0
// (/root/reference/part2/2a/main.py:101,114), dist.all_reduce(SUM)
// (/root/reference/part2/2b/main.py:106), and DDP's broadcast + bucket all-reduce + mean
// (/root/reference/part3/main.py:137).  Everything is enqueued on a caller-provided HIP
// stream; gather and scatter are grouped send/recv so the root's transfers to its seven
// peers proceed concurrently over their own xGMI links.
#include <rccl/rccl.h>
#include <string.h>
#include <unistd.h>

#include <chrono>
#include <stdexcept>

#include "dml_comm.h"
#include "dml_kernels.h"

namespace dml {

namespace {

void check_nccl(ncclResult_t r, const char* what) {
    if (r != ncclSuccess && r != ncclInProgress)
        throw std::runtime_error(std::string("RCCL ") + what + " failed: " + ncclGetErrorString(r));
}

void check_hip(hipError_t e, const char* what) {
    if (e != hipSuccess)
        throw std::runtime_error(std::string("HIP ") + what + " failed: " + hipGetErrorString(e));
}

ncclDataType_t nccl_type(DType dt) {
    switch (dt) {
        case DT_F32: return ncclFloat32;
        case DT_BF16: return ncclBfloat16;
        case DT_I64: return ncclInt64;
        case DT_U8: return ncclUint8;
    }
    throw std::invalid_argument("unknown dtype code");
}

}  // namespace

size_t dtype_bytes(DType dt) {
    switch (dt) {
        case DT_F32: return 4;
        case DT_BF16: return 2;
        case DT_I64: return 8;
        case DT_U8: return 1;
    }
    throw std::invalid_argument("unknown dtype code");
}

void link_bytes(CollKind kind, uint64_t payload_bytes, int world, bool is_root, uint64_t* tx,
                uint64_t* rx) {
    *tx = *rx = 0;
    if (world <= 1) return;                 // a group of one moves nothing
    const uint64_t w = (uint64_t)world;
    switch (kind) {
        case COLL_ALL_REDUCE:               // ring: reduce-scatter + all-gather
            *tx = *rx = 2 * (w - 1) * payload_bytes / w;
            return;
        case COLL_BROADCAST:                // the payload leaves the root once and arrives once
            if (is_root) *tx = payload_bytes; else *rx = payload_bytes;
            return;
        case COLL_GATHER:                   // every other rank sends its payload to the root
            if (is_root) *rx = (w - 1) * payload_bytes; else *tx = payload_bytes;
            return;
        case COLL_SCATTER:
            if (is_root) *tx = (w - 1) * payload_bytes; else *rx = payload_bytes;
            return;
    }
    throw std::invalid_argument("link_bytes: unknown collective");
}

std::string RcclComm::unique_id() {
    ncclUniqueId id;
    check_nccl(ncclGetUniqueId(&id), "ncclGetUniqueId");
    return std::string((const char*)&id, sizeof(id));
}

int RcclComm::version() {
    int v = 0;
    check_nccl(ncclGetVersion(&v), "ncclGetVersion");
    return v;
}

void RcclComm::self_exchange(const void* src, void* dst, size_t count, DType dt,
                             hipStream_t stream) {
    if (!comm_) throw std::logic_error("RcclComm: communicator destroyed");
    if (count == 0) return;
    const ncclDataType_t t = nccl_type(dt);
    group_start();
    check_nccl(ncclSend(src, count, t, rank_, (ncclComm_t)comm_, stream), "ncclSend(self)");
    check_nccl(ncclRecv(dst, count, t, rank_, (ncclComm_t)comm_, stream), "ncclRecv(self)");
    group_end();
}

RcclComm::RcclComm(const std::string& unique_id, int rank, int world_size, int device)
    : Comm(rank, world_size), device_(device) {
    if (unique_id.size() != sizeof(ncclUniqueId))
        throw std::invalid_argument("RcclComm: unique id must be " +
                                    std::to_string(sizeof(ncclUniqueId)) + " bytes");
    if (rank < 0 || rank >= world_size) throw std::invalid_argument("RcclComm: bad rank");
    check_hip(hipSetDevice(device), "hipSetDevice");
    ncclUniqueId id;
    memcpy(&id, unique_id.data(), sizeof(id));
    ncclComm_t c = nullptr;
    check_nccl(ncclCommInitRank(&c, world_size, id, rank), "ncclCommInitRank");
    comm_ = c;
}

RcclComm::~RcclComm() {
    try {
        destroy();
    } catch (...) {
    }
}

void RcclComm::destroy() {
    if (comm_) {
        ncclCommDestroy((ncclComm_t)comm_);
        comm_ = nullptr;
    }
}

int RcclComm::peer_count() {
    int n = -1;
    if (!comm_) throw std::logic_error("RcclComm: communicator destroyed");
    check_nccl(ncclCommCount((ncclComm_t)comm_, &n), "ncclCommCount");
    return n;
}

void RcclComm::group_start() {
    check_nccl(ncclGroupStart(), "ncclGroupStart");
    ++group_depth_;
}

void RcclComm::group_end() {
    if (group_depth_ <= 0) throw std::logic_error("RcclComm: group_end without group_start");
    --group_depth_;
    check_nccl(ncclGroupEnd(), "ncclGroupEnd");
    if (group_depth_ == 0) {
        // the grouped transfers are enqueued now; the root's own copies follow them in stream
        // order (they touch buffers no transfer of the group reads or writes)
        // ... batched: one launch per run of copies on the same stream instead of one copy
        // kernel per tensor
        CopyJobs jobs;
        hipStream_t on = nullptr;
        auto flush = [&]() {
            if (jobs.n) copy_batch(jobs, on);
            jobs.n = 0;
        };
        for (const SelfCopy& c : deferred_copies_) {
            if (jobs.n && (c.stream != on || jobs.n == CopyJobs::kMax)) flush();
            on = c.stream;
            jobs.dst[jobs.n] = c.dst; jobs.src[jobs.n] = c.src; jobs.bytes[jobs.n] = c.bytes;
            ++jobs.n;
        }
        flush();
        deferred_copies_.clear();
    }
}

void RcclComm::all_reduce(void* buf, size_t count, DType dt, bool average, hipStream_t stream) {
    if (count == 0) return;
    account(COLL_ALL_REDUCE, count, dt, rank_);
    check_nccl(ncclAllReduce(buf, buf, count, nccl_type(dt), average ? ncclAvg : ncclSum,
                             (ncclComm_t)comm_, stream),
               "ncclAllReduce");
}

void RcclComm::broadcast(const void* send, void* recv, size_t count, DType dt, int root,
                         hipStream_t stream) {
    if (count == 0) return;
    account(COLL_BROADCAST, count, dt, root);
    check_nccl(ncclBroadcast(send, recv, count, nccl_type(dt), root, (ncclComm_t)comm_, stream),
               "ncclBroadcast");
}

void RcclComm::gather(const void* send, const std::vector<void*>& recv, size_t count, DType dt,
                      int root, hipStream_t stream) {
    if (count == 0) return;
    account(COLL_GATHER, count, dt, root);
    const ncclDataType_t t = nccl_type(dt);
    if (rank_ == root) {
        if ((int)recv.size() != world_)
            throw std::invalid_argument("RcclComm::gather: root needs world_size receive slabs");
        group_start();
        for (int r = 0; r < world_; ++r)
            if (r != root)
                check_nccl(ncclRecv(recv[r], count, t, r, (ncclComm_t)comm_, stream), "ncclRecv");
        if (recv[root] != send)
            deferred_copies_.push_back({recv[root], send, count * dtype_bytes(dt), stream});
        group_end();
    } else {
        check_nccl(ncclSend(send, count, t, root, (ncclComm_t)comm_, stream), "ncclSend");
    }
}

void RcclComm::scatter(void* out, const std::vector<const void*>& chunks, size_t count, DType dt,
                       int root, hipStream_t stream) {
    if (count == 0) return;
    account(COLL_SCATTER, count, dt, root);
    const ncclDataType_t t = nccl_type(dt);
    if (rank_ == root) {
        if ((int)chunks.size() != world_)
            throw std::invalid_argument("RcclComm::scatter: root needs world_size chunks");
        group_start();
        for (int r = 0; r < world_; ++r)
            if (r != root)
                check_nccl(ncclSend(chunks[r], count, t, r, (ncclComm_t)comm_, stream), "ncclSend");
        if (chunks[root] != out)
            deferred_copies_.push_back({out, chunks[root], count * dtype_bytes(dt), stream});
        group_end();
    } else {
        check_nccl(ncclRecv(out, count, t, root, (ncclComm_t)comm_, stream), "ncclRecv");
    }
}

void RcclComm::wait_stream(hipStream_t stream, double timeout_s) {
    const auto t0 = std::chrono::steady_clock::now();
    while (true) {
        const hipError_t q = hipStreamQuery(stream);
        if (q == hipSuccess) return;
        if (q != hipErrorNotReady) check_hip(q, "hipStreamQuery");
        ncclResult_t async = ncclSuccess;
        if (comm_ && ncclCommGetAsyncError((ncclComm_t)comm_, &async) == ncclSuccess &&
            async != ncclSuccess && async != ncclInProgress) {
            ncclCommAbort((ncclComm_t)comm_);
            comm_ = nullptr;
            throw std::runtime_error(std::string("RCCL asynchronous error: ") +
                                     ncclGetErrorString(async));
        }
        const double dt =
            std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (dt > timeout_s) {
            if (comm_) {
                ncclCommAbort((ncclComm_t)comm_);
                comm_ = nullptr;
            }
            throw std::runtime_error("collective did not complete within " +
                                     std::to_string(timeout_s) + " s (peer lost?)");
        }
        ::usleep(200);
    }
}

}  // namespace dml
