// Bucket planner, stream-overlapped bucket reducer, and the Python bindings of the
// communication runtime.
//
// The reducer is the native counterpart of what DistributedDataParallel(model,
// bucket_cap_mb=25) gives the reference (/root/reference/part3/main.py:137): gradients
// are grouped into buckets in reverse parameter order and each bucket's all-reduce is
// started as soon as backward has produced its last gradient, overlapping the rest of
// backward.  Here a bucket is a slice of the flat gradient buffer, the all-reduce runs in
// place on a dedicated HIP stream, and a bucket departs / reports back either through a flag
// word and a wait-value packet or through an event pair (see BucketReducer's constructor).
#include <cstring>
#include <cstdlib>
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>

#include <stdexcept>

#include "dml_comm.h"
#include "dml_kernels.h"

namespace py = pybind11;

namespace dml {

std::vector<BucketRange> plan_buckets(const std::vector<size_t>& sizes_bytes, size_t cap_bytes,
                                      size_t first_cap_bytes) {
    std::vector<BucketRange> out;
    int hi = (int)sizes_bytes.size();
    size_t acc = 0;
    size_t cap = first_cap_bytes > 0 ? first_cap_bytes : cap_bytes;
    for (int i = (int)sizes_bytes.size() - 1; i >= 0; --i) {
        acc += sizes_bytes[i];
        if (acc >= cap) {
            out.push_back({i, hi});
            hi = i;
            acc = 0;
            cap = cap_bytes;
        }
    }
    if (hi > 0) out.push_back({0, hi});
    return out;
}

namespace {
// Event-pair departures keep the default (system) release scope on purpose, no
// hipEventReleaseToDevice: with more than one rank, peers read a bucket straight out of this
// GPU's memory over xGMI, and the "ready" event's release is what is documented to make this
// GPU's writes visible beyond the device.  (The flag-word departure relies on the release every
// kernel performs as it ends instead; see the constructor for where each form is the default.)
constexpr unsigned kEventFlags = hipEventDisableTiming;

void check_hip(hipError_t e, const char* what) {
    if (e != hipSuccess)
        throw std::runtime_error(std::string("HIP ") + what + " failed: " + hipGetErrorString(e));
}
}  // namespace

BucketReducer::BucketReducer(Comm* comm, float* flat_grads, std::vector<Slice> slices,
                             bool average, bool serialize, hipStream_t compute_stream)
    : comm_(comm), grads_(flat_grads), slices_(std::move(slices)), average_(average),
      serialize_(serialize) {
    if (!comm_) throw std::invalid_argument("BucketReducer: communicator missing");
    // Default priority on purpose: with a high-priority communication stream the
    // event hand-off with the compute stream was observed to stall on MI355X / ROCm 7.
    // The stream is probed to sit on a hardware queue of its own: on a queue shared with the
    // compute stream the collectives would simply run in line with backward.
    if (!serialize_) comm_stream_ = create_concurrent_stream({compute_stream});
    ready_.resize(slices_.size());
    done_.resize(slices_.size());
    launched_.assign(slices_.size(), 0);
    unsigned ready_flags = kEventFlags, done_flags = kEventFlags;
    if (const char* e = getenv("DML_REDUCER_EVENT_SCOPE")) {        // measurement knob
        if (strstr(e, "ready-nofence")) ready_flags |= hipEventDisableSystemFence;
        if (strstr(e, "done-nofence")) done_flags |= hipEventDisableSystemFence;
    }
    for (size_t i = 0; i < slices_.size(); ++i) {
        check_hip(hipEventCreateWithFlags(&ready_[i], ready_flags), "hipEventCreate");
        check_hip(hipEventCreateWithFlags(&done_[i], done_flags), "hipEventCreate");
    }
    // A stream that waits for an EVENT of another queue costs the step ~6.5 us per bucket on
    // MI355X / ROCm 7 (measured at one rank by leaving the wait out: the marker alone costs
    // 1.7 us, marker + wait 8 us, tools/ab_reducer_probe.sh).  Where the device can wait for a
    // value in memory, a bucket can depart through a flag word instead: a one-wave kernel behind
    // the bucket's last gradient kernel raises it - every earlier kernel of that queue has then
    // completed and released its writes - and the communication stream holds a wait-value packet.
    //
    // Which form is the default.  Inside one GPU (one rank, or the loopback test transport, whose
    // payload travels through the host) the flag word has been shown bit-identical to the event
    // pair.  Over xGMI the peers of an RCCL all-reduce read this GPU's gradients remotely, and that
    // the end-of-kernel release covers such a reader has not been observed on a multi-GPU node
    // from this code yet: with RCCL and more than one rank the event pair (system-scope release)
    // stays the default until a run has compared the two - bench.py does that comparison at the
    // start of every multi-rank run and switches to flag words when they reproduce the event
    // pair's parameters bit for bit (set_departure_by_value).  DML_REDUCER_HANDOFF=event | value
    // forces one form.
    int device = 0;
    if (!serialize_ && hipGetDevice(&device) == hipSuccess)
        (void)hipDeviceGetAttribute(&can_wait_value_, hipDeviceAttributeCanUseStreamWaitValue, device);
    const char* mode = getenv("DML_REDUCER_HANDOFF");
    const bool remote_readers = comm_->world_size() > 1 && !strcmp(comm_->transport(), "rccl");
    bool by_value = can_wait_value_ && !remote_readers;
    if (mode && !strcmp(mode, "event")) by_value = false;
    if (mode && !strcmp(mode, "value")) by_value = can_wait_value_ != 0;
    set_departure_by_value(by_value);
}

void BucketReducer::set_departure_by_value(bool on) {
    if (on && (!can_wait_value_ || serialize_))
        throw std::logic_error("BucketReducer: flag-word departures need wait-value support and "
                               "a communication stream");
    if (on && !ready_flags_) {
        // (word b: bucket b's gradients are complete; word n + b: its all-reduce is complete)
        const size_t bytes = 2 * slices_.size() * sizeof(unsigned);
        check_hip(hipMalloc((void**)&ready_flags_, bytes), "hipMalloc(bucket flags)");
        check_hip(hipMemset(ready_flags_, 0, bytes), "hipMemset(bucket flags)");
        check_hip(hipDeviceSynchronize(), "hipDeviceSynchronize(bucket flags)");
    }
    by_value_ = on;
}

BucketReducer::~BucketReducer() {
    for (auto e : ready_) (void)hipEventDestroy(e);
    for (auto e : done_) (void)hipEventDestroy(e);
    if (comm_stream_) (void)hipStreamDestroy(comm_stream_);
    if (ready_flags_) (void)hipFree(ready_flags_);
}

void BucketReducer::begin_step() {
    launched_.assign(slices_.size(), 0);
    last_ready_ = -1;
    if (++step_seq_ == 0 && ready_flags_) {
        // the 32-bit sequence wrapped (2^32 steps): start the flag words over
        check_hip(hipDeviceSynchronize(), "hipDeviceSynchronize(sequence wrap)");
        check_hip(hipMemset(ready_flags_, 0, 2 * slices_.size() * sizeof(unsigned)), "hipMemset(bucket flags)");
        step_seq_ = 1;
    }
}

void BucketReducer::bucket_ready(int b, hipStream_t compute,
                                 const std::function<void(hipStream_t)>& after, bool in_line) {
    if (b < 0 || b >= (int)slices_.size()) throw std::out_of_range("BucketReducer: bucket index");
    if (launched_[b]) throw std::logic_error("BucketReducer: bucket marked ready twice");
    launched_[b] = 1;
    // in_line: the bucket that completes with the end of backward has nothing left to overlap
    // with; run on the compute stream itself it skips two cross-queue hand-offs (~20 us).  The
    // transport orders it behind the collectives still running on the communication stream
    // (one communicator serves its launches in issue order).
    if (serialize_ || (in_line && inline_last_)) {       // same stream, no events
        reduce_slice(b, compute);
        if (after) after(compute);
        return;
    }
    // (measurement knob, one rank only: which part of a bucket's departure costs the step time)
    static const char* probe = getenv("DML_REDUCER_PROBE");
    if (probe && comm_->world_size() == 1) {
        if (!strcmp(probe, "skip")) return;                          // no event, no hop
        if (!strcmp(probe, "marker")) {                              // the marker alone
            check_hip(hipEventRecord(ready_[b], compute), "hipEventRecord(ready)");
            return;
        }
        if (!strcmp(probe, "nodone")) {                              // marker + wait, no "done"
            check_hip(hipEventRecord(ready_[b], compute), "hipEventRecord(ready)");
            check_hip(hipStreamWaitEvent(comm_stream_, ready_[b], 0), "hipStreamWaitEvent(comm)");
            return;
        }
    }
    last_ready_ = b;
    hipStreamCaptureStatus capturing = hipStreamCaptureStatusNone;
    (void)hipStreamIsCapturing(compute, &capturing);
    if (by_value_ && capturing == hipStreamCaptureStatusNone) {
        enqueue_flag_raise(ready_flags_ + b, step_seq_, compute);
        check_hip(hipStreamWaitValue32(comm_stream_, ready_flags_ + b, step_seq_,
                                       hipStreamWaitValueGte, 0xffffffffu),
                  "hipStreamWaitValue32(comm)");
    } else {
        check_hip(hipEventRecord(ready_[b], compute), "hipEventRecord(ready)");
        check_hip(hipStreamWaitEvent(comm_stream_, ready_[b], 0), "hipStreamWaitEvent(comm)");
    }
    reduce_slice(b, comm_stream_);
    if (after) after(comm_stream_);
    done_by_value_ = by_value_ && capturing == hipStreamCaptureStatusNone;
    if (done_by_value_)
        enqueue_flag_raise(ready_flags_ + slices_.size() + b, step_seq_, comm_stream_);
    else
        check_hip(hipEventRecord(done_[b], comm_stream_), "hipEventRecord(done)");
}

void BucketReducer::set_bf16_payload(bf16_raw* wire) {
    if (wire && average_)
        throw std::invalid_argument("BucketReducer: the bf16 payload is summed; fold the mean "
                                    "into the optimizer's gradient scale");
    wire_ = wire;
}

void BucketReducer::reduce_slice(int b, hipStream_t stream) {
    const Slice& s = slices_[b];
    float* base = grads_ + s.offset;
    if (!wire_) {
        comm_->all_reduce(base, s.count, DT_F32, average_, stream);
        return;
    }
    // bf16 on the wire: half the bytes per link.  Gradients are rounded once before the sum
    // (the sum itself is formed by the transport in bf16), widened again behind it; the
    // BatchNorm statistics a bucket may carry at its end stay fp32 in the same launch.
    size_t n = s.count, tail = 0;
    if (buffers_count_ && buffers_staging_ >= base && buffers_staging_ < base + s.count) {
        n = (size_t)(buffers_staging_ - base);
        tail = s.count - n;
    }
    // (the conversion kernels move four elements per thread: up to three gradients at the end of
    // a slice of any other length join the fp32 part)
    tail += n & 3;
    n -= n & 3;
    if (n) cast_f32_to_bf16(base, wire_ + s.offset, n, stream);
    comm_->group_start();
    if (n) comm_->all_reduce(wire_ + s.offset, n, DT_BF16, false, stream);
    if (tail) comm_->all_reduce(base + n, tail, DT_F32, false, stream);
    comm_->group_end();
    if (n && !keep_on_wire_) cast_bf16_to_f32(wire_ + s.offset, base, n, stream);
}

void BucketReducer::set_keep_on_wire(bool on) {
    if (on) {
        if (!wire_) throw std::logic_error("BucketReducer: no bf16 payload to keep results on");
        for (size_t b = 0; b < slices_.size(); ++b) {
            size_t n = slices_[b].count;
            const float* base = grads_ + slices_[b].offset;
            if (buffers_count_ && buffers_staging_ >= base && buffers_staging_ < base + n)
                n = (size_t)(buffers_staging_ - base);
            if ((n & 3) || (slices_[b].offset & 3))
                throw std::invalid_argument("BucketReducer: a slice that is not whole groups of four "
                                            "cannot stay on the bf16 wire");
        }
    }
    keep_on_wire_ = on;
}

void BucketReducer::wait_all(hipStream_t compute) {
    for (size_t b = 0; b < slices_.size(); ++b)
        if (!launched_[b])
            throw std::logic_error("BucketReducer: backward ended with bucket " +
                                   std::to_string(b) + " never marked ready");
    // The communication stream runs the buckets in the order they were marked ready, so the
    // event of the most recent one covers them all: one cross-stream wait, not one per bucket.
    if (!serialize_ && last_ready_ >= 0) {
        if (done_by_value_)
            check_hip(hipStreamWaitValue32(compute, ready_flags_ + slices_.size() + last_ready_,
                                           step_seq_, hipStreamWaitValueGte, 0xffffffffu),
                      "hipStreamWaitValue32(compute)");
        else
            check_hip(hipStreamWaitEvent(compute, done_[last_ready_], 0), "hipStreamWaitEvent(compute)");
    }
    if (buffers_staged_) {
        buffers_staged_ = false;
        buffers_pending_ = true;        // the staging area now holds rank 0's values everywhere
    }
}

void BucketReducer::set_buffer_sync(float* buffers, float* staging, size_t count) {
    if (count && (!buffers || !staging))
        throw std::invalid_argument("BucketReducer: buffer sync needs both pointers");
    if (count && average_)
        throw std::invalid_argument("BucketReducer: buffers ride in a SUM all-reduce; fold the "
                                    "mean into the optimizer's gradient scale");
    if (count) {
        // the staging area must be the tail of one bucket's slice
        bool inside = false;
        for (const Slice& s : slices_)
            if (staging >= grads_ + s.offset && staging + count <= grads_ + s.offset + s.count)
                inside = true;
        if (!inside)
            throw std::invalid_argument("BucketReducer: the buffer staging area must lie inside a "
                                        "bucket slice");
    }
    buffers_ = buffers;
    buffers_staging_ = staging;
    buffers_count_ = count;
    buffers_pending_ = buffers_staged_ = false;
}

void BucketReducer::buffers_updated(hipStream_t compute) {
    // Forward is done, so rank 0's statistics are final.  They travel inside a gradient
    // bucket: rank 0 copies them into the staging floats that bucket's slice ends with, every
    // other rank zeroes its copy, and the bucket's SUM all-reduce leaves rank 0's values on
    // all ranks - no extra collective, no extra event.
    if (!buffers_count_) return;
    const size_t bytes = buffers_count_ * sizeof(float);
    if (comm_->rank() == 0)
        check_hip(hipMemcpyAsync(buffers_staging_, buffers_, bytes, hipMemcpyDeviceToDevice, compute),
                  "hipMemcpyAsync(stage buffers)");
    else
        check_hip(hipMemsetAsync(buffers_staging_, 0, bytes, compute), "hipMemsetAsync(staging)");
    buffers_staged_ = true;
}

void BucketReducer::apply_buffers(hipStream_t compute) {
    // ordered after the previous step's wait_all on this stream
    if (!buffers_count_ || !buffers_pending_) return;
    buffers_pending_ = false;
    check_hip(hipMemcpyAsync(buffers_, buffers_staging_, buffers_count_ * sizeof(float),
                             hipMemcpyDeviceToDevice, compute),
              "hipMemcpyAsync(buffers)");
}

// ----------------------------------------------------------------------- bindings
void bind_comm(py::module_& m) {
    py::class_<TcpStore>(m, "TcpStore")
        .def(py::init<const std::string&, int, bool, int, double>(), py::arg("host"),
             py::arg("port"), py::arg("is_server"), py::arg("world_size"), py::arg("timeout_s"),
             py::call_guard<py::gil_scoped_release>())
        .def("set", [](TcpStore& s, const std::string& k, py::bytes v) {
            std::string val = v;
            py::gil_scoped_release rel;
            s.set(k, val);
        })
        .def("get", [](TcpStore& s, const std::string& k) {
            std::string out;
            {
                py::gil_scoped_release rel;
                out = s.get(k);
            }
            return py::bytes(out);
        })
        .def("add", &TcpStore::add, py::call_guard<py::gil_scoped_release>())
        .def("barrier", &TcpStore::barrier, py::call_guard<py::gil_scoped_release>())
        .def_property_readonly("port", &TcpStore::port);

    m.def("rccl_unique_id", []() { return py::bytes(RcclComm::unique_id()); });
    m.def("rccl_version", &RcclComm::version);

    m.def("link_bytes", [](int kind, uint64_t payload, int world, bool is_root) {
        uint64_t tx = 0, rx = 0;
        link_bytes((CollKind)kind, payload, world, is_root, &tx, &rx);
        return py::make_tuple(tx, rx);
    }, py::arg("kind"), py::arg("payload_bytes"), py::arg("world"), py::arg("is_root"));

    py::class_<Comm>(m, "Comm")
        .def("volume", [](const Comm& c) {
            const CommVolume& v = c.volume();
            py::dict d;
            d["tx_bytes"] = v.tx_bytes;
            d["rx_bytes"] = v.rx_bytes;
            const char* names[4] = {"all_reduce", "broadcast", "gather", "scatter"};
            py::dict calls, payload;
            for (int k = 0; k < 4; ++k) {
                calls[names[k]] = v.calls[k];
                payload[names[k]] = v.payload_bytes[k];
            }
            d["calls"] = calls;
            d["payload_bytes"] = payload;
            return d;
        })
        .def("reset_volume", &Comm::reset_volume)
        .def_property_readonly("rank", &Comm::rank)
        .def_property_readonly("world_size", &Comm::world_size)
        .def_property_readonly("transport", [](Comm& c) { return std::string(c.transport()); })
        .def_property_readonly("peer_count", &Comm::peer_count)
        .def("all_reduce", [](Comm& c, uintptr_t buf, size_t n, int dt, bool avg, uintptr_t s) {
            c.all_reduce((void*)buf, n, (DType)dt, avg, (hipStream_t)s);
        })
        .def("broadcast", [](Comm& c, uintptr_t buf, size_t n, int dt, int root, uintptr_t s) {
            c.broadcast((void*)buf, n, (DType)dt, root, (hipStream_t)s);
        })
        .def("gather", [](Comm& c, uintptr_t send, std::vector<uintptr_t> recv, size_t n,
                          int dt, int root, uintptr_t s) {
            std::vector<void*> r;
            for (auto p : recv) r.push_back((void*)p);
            c.gather((const void*)send, r, n, (DType)dt, root, (hipStream_t)s);
        })
        .def("scatter", [](Comm& c, uintptr_t out, std::vector<uintptr_t> chunks, size_t n,
                           int dt, int root, uintptr_t s) {
            std::vector<const void*> r;
            for (auto p : chunks) r.push_back((const void*)p);
            c.scatter((void*)out, r, n, (DType)dt, root, (hipStream_t)s);
        })
        .def("group_start", &Comm::group_start)
        .def("group_end", &Comm::group_end)
        .def("wait_stream", [](Comm& c, uintptr_t s, double t) {
            py::gil_scoped_release rel;
            c.wait_stream((hipStream_t)s, t);
        })
        .def("destroy", [](Comm& c) {
            py::gil_scoped_release rel;
            c.destroy();
        });

    py::class_<RcclComm, Comm>(m, "RcclComm")
        .def(py::init([](py::bytes uid, int rank, int world, int device) {
            std::string id = uid;
            py::gil_scoped_release rel;
            return new RcclComm(id, rank, world, device);
        }))
        .def("self_exchange", [](RcclComm& c, uintptr_t src, uintptr_t dst, size_t n, int dt,
                                 uintptr_t s) {
            c.self_exchange((const void*)src, (void*)dst, n, (DType)dt, (hipStream_t)s);
        });

    py::class_<LoopbackComm, Comm>(m, "LoopbackComm")
        .def(py::init([](const std::string& name, int rank, int world, int device, double timeout) {
            py::gil_scoped_release rel;
            return new LoopbackComm(name, rank, world, device, timeout);
        }));

    py::class_<ManualSync>(m, "ManualSync")
        .def(py::init([](Comm* comm, int mode, uintptr_t grads,
                         std::vector<std::pair<size_t, size_t>> tensors, int root,
                         uintptr_t slabs, size_t stride) {
            std::vector<ManualSync::Slice> t;
            for (auto& p : tensors) t.push_back({p.first, p.second});
            return new ManualSync(comm, (ManualSync::Mode)mode, (float*)grads, t, root,
                                  (float*)slabs, stride);
        }), py::arg("comm"), py::arg("mode"), py::arg("grads"), py::arg("tensors"),
            py::arg("root") = 0, py::arg("slabs") = 0, py::arg("stride") = 0,
            py::keep_alive<1, 2>())
        .def("run", [](ManualSync& m, uintptr_t s) { m.run((hipStream_t)s); })
        .def_static("packed_elems", [](std::vector<std::pair<size_t, size_t>> tensors) {
            std::vector<ManualSync::Slice> t;
            for (auto& p : tensors) t.push_back({p.first, p.second});
            return ManualSync::packed_elems(t);
        });

    m.def("plan_buckets", [](std::vector<size_t> sizes, size_t cap, size_t first_cap) {
        std::vector<std::pair<int, int>> out;
        for (auto& b : plan_buckets(sizes, cap, first_cap)) out.push_back({b.lo, b.hi});
        return out;
    });

    py::class_<BucketReducer>(m, "BucketReducer")
        .def(py::init([](Comm* comm, uintptr_t grads,
                         std::vector<std::pair<size_t, size_t>> slices, bool average,
                         bool serialize, uintptr_t compute_stream) {
            std::vector<BucketReducer::Slice> s;
            for (auto& p : slices) s.push_back({p.first, p.second});
            return new BucketReducer(comm, (float*)grads, s, average, serialize,
                                     (hipStream_t)compute_stream);
        }), py::arg("comm"), py::arg("grads"), py::arg("slices"), py::arg("average"),
            py::arg("serialize"), py::arg("compute_stream") = 0, py::keep_alive<1, 2>())
        .def_property_readonly("num_buckets", &BucketReducer::num_buckets)
        .def("begin_step", &BucketReducer::begin_step)
        .def("bucket_ready", [](BucketReducer& r, int b, uintptr_t s) {
            r.bucket_ready(b, (hipStream_t)s);
        })
        .def("wait_all", [](BucketReducer& r, uintptr_t s) { r.wait_all((hipStream_t)s); })
        .def("set_buffer_sync", [](BucketReducer& r, uintptr_t buf, uintptr_t staging, size_t n) {
            r.set_buffer_sync((float*)buf, (float*)staging, n);
        })
        .def("buffers_updated", [](BucketReducer& r, uintptr_t s) {
            r.buffers_updated((hipStream_t)s);
        })
        .def("apply_buffers", [](BucketReducer& r, uintptr_t s) { r.apply_buffers((hipStream_t)s); })
        .def("drop_pending_buffers", &BucketReducer::drop_pending_buffers)
        .def("set_bf16_payload", [](BucketReducer& r, uintptr_t wire) {
            r.set_bf16_payload((bf16_raw*)wire);
        })
        .def("set_keep_on_wire", &BucketReducer::set_keep_on_wire)
        .def_property_readonly("keeps_on_wire", &BucketReducer::keeps_on_wire)
        .def_property("inline_last", &BucketReducer::inline_last, &BucketReducer::set_inline_last)
        .def_property_readonly("departure_mode",
                               [](BucketReducer& r) { return std::string(r.departure_mode()); })
        .def_property_readonly("can_depart_by_value", &BucketReducer::can_depart_by_value)
        .def("set_departure_by_value", &BucketReducer::set_departure_by_value)
        .def_property_readonly("comm_stream",
                               [](BucketReducer& r) { return (uintptr_t)r.comm_stream(); });
}

}  // namespace dml
