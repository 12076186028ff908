// Communication runtime: rendezvous store, RCCL communicator, bucket reducer.
//
// Replaces what the reference borrows from torch.distributed: the TCP rendezvous of
// init_process_group("gloo", "tcp://host:port") (/root/reference/part2/2a/main.py:197),
// the gather / scatter / all_reduce collectives (part2/2a/main.py:101,114,
// part2/2b/main.py:106) and DistributedDataParallel's bucketed, backward-overlapped
// all-reduce with its construction-time broadcast (part3/main.py:137).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>
#include <map>
#include <mutex>
#include <functional>
#include <string>
#include <thread>
#include <vector>

namespace pybind11 { class module_; }

namespace dml {

// ------------------------------------------------------------------------ TcpStore
// Minimal key-value store over TCP.  The process with is_server=true runs the server in
// a background thread (and talks to it over loopback like any other client); `get`
// blocks until the key exists or the timeout expires.
class TcpStore {
  public:
    TcpStore(const std::string& host, int port, bool is_server, int world_size, double timeout_s);
    ~TcpStore();
    TcpStore(const TcpStore&) = delete;
    TcpStore& operator=(const TcpStore&) = delete;

    void set(const std::string& key, const std::string& value);
    std::string get(const std::string& key);
    int64_t add(const std::string& key, int64_t delta);
    // all `world_size` ranks call this with the same name; returns when everyone arrived
    void barrier(const std::string& name);
    int port() const { return port_; }

  private:
    void serve();
    void request(uint8_t op, const std::string& key, const std::string& val, std::string* out);

    std::string host_;
    int port_;
    bool is_server_;
    int world_size_;
    double timeout_s_;
    int listen_fd_ = -1;
    int client_fd_ = -1;
    std::thread server_;
    std::atomic<bool> stop_{false};
    std::atomic<int> live_clients_{0};
    std::mutex client_mu_;
};

// ---------------------------------------------------------------------------- Comm
enum DType : int { DT_F32 = 0, DT_BF16 = 1, DT_I64 = 2, DT_U8 = 3 };
size_t dtype_bytes(DType dt);

// Bytes a rank puts on (tx) and takes off (rx) its links, by the algorithms the transports run:
// all-reduce as a ring (2 (W-1)/W of the payload each way), gather / scatter as direct transfers
// to / from the root ((W-1) payloads at the root, one at every other rank), broadcast as a chain
// from the root.  The reference's report samples exactly this per node (group25.pdf p4, p6, p8:
// node 0 moves ~4.4 GB each way in 40 iterations of gather/scatter against ~1.5 GB elsewhere,
// ~2.2 GB per node with all-reduce); the counters below reproduce it from what was issued.
enum CollKind : int { COLL_ALL_REDUCE = 0, COLL_BROADCAST = 1, COLL_GATHER = 2, COLL_SCATTER = 3 };
void link_bytes(CollKind kind, uint64_t payload_bytes, int world, bool is_root, uint64_t* tx,
                uint64_t* rx);
struct CommVolume {
    uint64_t tx_bytes = 0, rx_bytes = 0;
    uint64_t calls[4] = {0, 0, 0, 0};       // indexed by CollKind
    uint64_t payload_bytes[4] = {0, 0, 0, 0};
};

// The collective surface of the three sync modes.  Every call enqueues on the caller's HIP
// stream and returns; calls on one communicator must be issued in the same order on every rank.
//
// Two transports implement it.  RcclComm is the product path (RCCL over xGMI, one process per
// GPU).  LoopbackComm exists for tests only: RCCL refuses two ranks on one device ("Duplicate
// GPU detected"), so on a single-GPU machine it is the only way to run the engine, the bucket
// reducer and the sync strategies with more than one rank.
class Comm {
  public:
    virtual ~Comm() {}
    int rank() const { return rank_; }
    int world_size() const { return world_; }
    virtual const char* transport() const = 0;
    // number of ranks the transport itself reports for this communicator (ncclCommCount)
    virtual int peer_count() = 0;

    virtual void all_reduce(void* buf, size_t count, DType dt, bool average,
                            hipStream_t stream) = 0;
    // root sends `send`, every rank (root included) receives into `recv` (may alias `send`)
    virtual void broadcast(const void* send, void* recv, size_t count, DType dt, int root,
                           hipStream_t stream) = 0;
    void broadcast(void* buf, size_t count, DType dt, int root, hipStream_t stream) {
        broadcast(buf, buf, count, dt, root, stream);
    }
    // root receives rank r's `send` into recv[r]; its own contribution is a device copy
    virtual void gather(const void* send, const std::vector<void*>& recv, size_t count, DType dt,
                        int root, hipStream_t stream) = 0;
    // rank r receives root's chunks[r] into `out`
    virtual void scatter(void* out, const std::vector<const void*>& chunks, size_t count,
                         DType dt, int root, hipStream_t stream) = 0;
    // calls between group_start / group_end may be fused into one launch by the transport
    virtual void group_start() {}
    virtual void group_end() {}
    // Host-side wait for `stream` with a deadline; on expiry or an asynchronous transport error
    // the communicator is aborted and std::runtime_error is thrown, so a dead peer turns
    // into a clean non-zero exit instead of a hang.
    virtual void wait_stream(hipStream_t stream, double timeout_s) = 0;
    virtual void destroy() = 0;
    // what this rank has issued since construction / the last reset (host-side bookkeeping)
    const CommVolume& volume() const { return volume_; }
    void reset_volume() { volume_ = CommVolume{}; }

  protected:
    Comm(int rank, int world) : rank_(rank), world_(world) {}
    // every transport calls this once per collective it issues
    void account(CollKind kind, size_t count, DType dt, int root) {
        uint64_t tx = 0, rx = 0;
        const uint64_t bytes = (uint64_t)count * dtype_bytes(dt);
        link_bytes(kind, bytes, world_, rank_ == root, &tx, &rx);
        volume_.tx_bytes += tx;
        volume_.rx_bytes += rx;
        volume_.calls[kind] += 1;
        volume_.payload_bytes[kind] += bytes;
    }
    int rank_, world_;
    CommVolume volume_;
};

// ------------------------------------------------------------------------ RcclComm
class RcclComm : public Comm {
  public:
    static std::string unique_id();     // 128 opaque bytes, created on rank 0
    static int version();               // ncclGetVersion
    // One grouped ncclSend + ncclRecv of this rank with ITSELF (dst := src through the transport):
    // the point-to-point calls gather / scatter are built from, executable with a single rank.
    void self_exchange(const void* src, void* dst, size_t count, DType dt, hipStream_t stream);
    RcclComm(const std::string& unique_id, int rank, int world_size, int device);
    ~RcclComm() override;

    const char* transport() const override { return "rccl"; }
    int peer_count() override;
    void all_reduce(void* buf, size_t count, DType dt, bool average, hipStream_t stream) override;
    using Comm::broadcast;
    void broadcast(const void* send, void* recv, size_t count, DType dt, int root,
                   hipStream_t stream) override;
    void gather(const void* send, const std::vector<void*>& recv, size_t count, DType dt,
                int root, hipStream_t stream) override;
    void scatter(void* out, const std::vector<const void*>& chunks, size_t count, DType dt,
                 int root, hipStream_t stream) override;
    void group_start() override;
    void group_end() override;
    void wait_stream(hipStream_t stream, double timeout_s) override;
    void destroy() override;
    void* raw() const { return comm_; }
    int device() const { return device_; }

  private:
    void* comm_ = nullptr;      // ncclComm_t
    int device_;                // the GPU the communicator was created on
    int group_depth_ = 0;
    // root-side self copies of gather / scatter calls made inside an open group: issued when
    // the group closes, so they never interleave with the grouped transfers
    struct SelfCopy { void* dst; const void* src; size_t bytes; hipStream_t stream; };
    std::vector<SelfCopy> deferred_copies_;
};

// -------------------------------------------------------------------- LoopbackComm
// TEST TRANSPORT.  Ranks are processes that may all sit on ONE GPU; a collective stages its
// payload through a POSIX shared-memory segment in chunks.  It keeps the property the engine's
// choreography depends on - a collective is stream-ordered and asynchronous to the host: the
// call enqueues a "ready" flag kernel and a "wait" kernel on the caller's stream and hands the
// operation to a helper thread, which waits for the flag, copies device -> host, exchanges and
// reduces with its peers (in rank order: every rank computes bit-identical sums), copies the
// result back and releases the wait kernel.  Operations are served in issue order whatever
// stream they were issued on, which is also what RCCL does for one communicator.
class LoopbackComm : public Comm {
  public:
    // `name` identifies the job (all ranks pass the same string, e.g. the rendezvous port)
    LoopbackComm(const std::string& name, int rank, int world_size, int device, double timeout_s);
    ~LoopbackComm() override;

    const char* transport() const override { return "loopback"; }
    int peer_count() override;
    void all_reduce(void* buf, size_t count, DType dt, bool average, hipStream_t stream) override;
    using Comm::broadcast;
    void broadcast(const void* send, void* recv, size_t count, DType dt, int root,
                   hipStream_t stream) override;
    void gather(const void* send, const std::vector<void*>& recv, size_t count, DType dt,
                int root, hipStream_t stream) override;
    void scatter(void* out, const std::vector<const void*>& chunks, size_t count, DType dt,
                 int root, hipStream_t stream) override;
    void wait_stream(hipStream_t stream, double timeout_s) override;
    void destroy() override;

  private:
    struct Impl;
    Impl* impl_;
};

// stream-side flag kernels of the loopback transport (loopback_flags.hip)
void loopback_enqueue_signal(unsigned* flag, unsigned value, hipStream_t stream);
// "everything enqueued on `stream` before this has completed": raises *flag to `value`
void enqueue_flag_raise(unsigned* flag, unsigned value, hipStream_t stream);
void loopback_enqueue_wait(const unsigned* flag, unsigned value, unsigned* timed_out,
                           long long max_ticks, hipStream_t stream);

// --------------------------------------------------------------------- ManualSync
// The two gradient synchronisations the reference runs between backward and optimizer.step(),
// driven from C++ over the flat gradient buffer (no per-tensor Python in the step):
//   GATHER_SCATTER  per parameter tensor: gather on rank `root`, sum there, scatter the sum back
//                   (/root/reference/part2/2a/main.py:89-116)
//   ALL_REDUCE      per parameter tensor: all_reduce(SUM)  (/root/reference/part2/2b/main.py:101-106)
// Both keep one collective per tensor (fused into one launch by a group) and leave the SUM in
// the gradient buffer; a mean is folded into the optimizer's gradient scale by the caller.
class ManualSync {
  public:
    enum Mode : int { GATHER_SCATTER = 0, ALL_REDUCE = 1 };
    struct Slice { size_t offset, count; };
    // `root_slabs`: on the root of GATHER_SCATTER, world_size * slab_stride floats of landing
    // area (slab r holds rank r's gradients, tensors packed back to back); ignored otherwise.
    ManualSync(Comm* comm, Mode mode, float* flat_grads, std::vector<Slice> tensors, int root,
               float* root_slabs, size_t slab_stride);
    void run(hipStream_t stream);
    Mode mode() const { return mode_; }
    static size_t packed_elems(const std::vector<Slice>& tensors);    // slab_stride needed

  private:
    Comm* comm_;
    Mode mode_;
    float* grads_;
    std::vector<Slice> tensors_;
    int root_;
    float* slabs_;
    size_t stride_;
    bool in_place_ = false;
  public:
    bool in_place() const { return in_place_; }
};

// ------------------------------------------------------------------- bucket planning
struct BucketRange { int lo, hi; };     // parameter index range [lo, hi), forward order
// Walks the tensors back to front; a bucket closes once it has reached its cap (the first
// bucket uses first_cap_bytes so communication starts early).  Returned in completion order.
std::vector<BucketRange> plan_buckets(const std::vector<size_t>& sizes_bytes, size_t cap_bytes,
                                      size_t first_cap_bytes);

// ------------------------------------------------------------------- BucketReducer
// Overlaps gradient all-reduce with the rest of backward.  Gradients live in one flat
// fp32 buffer; a bucket is a contiguous element range of it.  `bucket_ready(b, compute)`
// orders the communication stream behind the compute stream's work so far (flag word or event
// pair, see departure_mode) and enqueues the all-reduce of that slice in place;
// `wait_all(compute)` makes the compute stream wait for every outstanding bucket (called right
// before the optimizer kernel).
// No host synchronisation, no staging copies.
//
// Module buffers (BatchNorm running statistics) follow rank 0 the way
// DistributedDataParallel(broadcast_buffers=True) makes them: every forward starts from
// rank 0's values.  Rank 0's buffers are final once its forward has run; they ride inside a
// gradient bucket (staging floats at the end of that bucket's slice: rank 0's copy there,
// zeros on every other rank, so the SUM all-reduce delivers rank 0's values) and the next step
// copies the staging area over the live buffers before its forward.  averaging must be off.
class BucketReducer {
  public:
    struct Slice { size_t offset, count; };
    // compute_stream: the stream backward runs on (the communication stream is chosen to
    // execute concurrently with it)
    BucketReducer(Comm* comm, float* flat_grads, std::vector<Slice> slices, bool average,
                  bool serialize, hipStream_t compute_stream = nullptr);
    ~BucketReducer();
    int num_buckets() const { return (int)slices_.size(); }
    void begin_step();
    // `after` (optional) is enqueued behind the bucket's all-reduce on the stream that ran it,
    // before the bucket counts as done - e.g. the optimizer update of the bucket's parameters
    // `in_line`: run the all-reduce on `compute` itself instead of the communication stream
    void bucket_ready(int b, hipStream_t compute,
                      const std::function<void(hipStream_t)>& after = nullptr,
                      bool in_line = false);
    void wait_all(hipStream_t compute);
    hipStream_t comm_stream() const { return comm_stream_; }
    // "value": flag word + wait-value packet, "event": event pair
    const char* departure_mode() const { return by_value_ ? "value" : "event"; }
    // Between steps, with both streams idle: switch the form of departure (the same schedule and
    // arithmetic either way, so runs from one snapshot must agree bit for bit).
    void set_departure_by_value(bool on);
    bool can_depart_by_value() const { return can_wait_value_ != 0 && !serialize_; }
    // Send gradients as bf16: `wire` is a bf16 buffer as long as the flat gradient buffer (same
    // element offsets).  A bucket is rounded into it, summed there and widened back into the
    // fp32 gradients, all on the stream that runs its collective.  nullptr: fp32 on the wire.
    void set_bf16_payload(uint16_t* wire);
    // The consumer (the engine's fused SGD) reads the reduced bf16 values from the wire buffer
    // itself: no widening pass behind the collective.  The fp32 gradients then keep the LOCAL
    // values.  Needs every slice's bf16 part to be whole groups of four (only BatchNorm staging
    // floats may follow it).
    void set_keep_on_wire(bool on);
    bool keeps_on_wire() const { return wire_ && keep_on_wire_; }
    const uint16_t* wire() const { return wire_; }
    // Whether a bucket marked `in_line` really runs on the compute stream (default) or goes
    // through the communication stream like every other bucket.
    void set_inline_last(bool on) { inline_last_ = on; }
    bool inline_last() const { return inline_last_; }

    void set_buffer_sync(float* buffers, float* staging, size_t count);
    void buffers_updated(hipStream_t compute);      // call after the forward pass
    void apply_buffers(hipStream_t compute);        // call before the next forward pass
    void drop_pending_buffers() { buffers_pending_ = buffers_staged_ = false; }
    // After wait_all: hands the copy "staging -> live buffers" that apply_buffers would issue at
    // the next step to the caller (who can let it ride in a launch it makes anyway); false if
    // there is nothing to copy.
    bool take_pending_buffers(const float** src, float** dst, size_t* count) {
        if (!buffers_count_ || !buffers_pending_) return false;
        buffers_pending_ = false;
        *src = buffers_staging_; *dst = buffers_; *count = buffers_count_;
        return true;
    }

  private:
    void reduce_slice(int b, hipStream_t stream);
    Comm* comm_;
    float* grads_;
    std::vector<Slice> slices_;
    bool average_, serialize_;
    bool inline_last_ = true;
    uint16_t* wire_ = nullptr;
    bool keep_on_wire_ = false;
    hipStream_t comm_stream_ = nullptr;
    std::vector<hipEvent_t> ready_, done_;
    // Departures through flag words instead of the "ready" event pair (see bucket_ready): word b
    // is raised to the step's sequence number by a one-wave kernel behind bucket b's gradients.
    unsigned* ready_flags_ = nullptr;
    unsigned step_seq_ = 0;
    bool by_value_ = false;
    int can_wait_value_ = 0;                // hipDeviceAttributeCanUseStreamWaitValue
    bool done_by_value_ = false;            // how the most recent bucket reports completion
    std::vector<char> launched_;
    int last_ready_ = -1;
    float* buffers_ = nullptr;
    float* buffers_staging_ = nullptr;
    size_t buffers_count_ = 0;
    bool buffers_pending_ = false, buffers_staged_ = false;
};

void bind_comm(pybind11::module_& m);

}  // namespace dml
