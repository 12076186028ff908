// ManualSync: the reference's two hand-written gradient synchronisations, over the flat
// gradient buffer, issued from C++ (interface in dml_comm.h).
//
//   gatherAndScatter(model)   /root/reference/part2/2a/main.py:89-116
//   allReduce(model)          /root/reference/part2/2b/main.py:101-106
//
// The reference issues one blocking collective per parameter tensor.  The per-tensor
// collectives are kept (18 gathers + 18 scatters, or 18 all-reduces, for VGG-11), but each run
// of them sits inside one transport group, so RCCL launches it as one kernel, and the root's
// (W-1) additions per tensor are one pass of one kernel over all tensors.
#include <stdexcept>

#include "dml_comm.h"
#include "dml_kernels.h"

namespace dml {

size_t ManualSync::packed_elems(const std::vector<Slice>& tensors) {
    size_t n = 0;
    for (const Slice& t : tensors) n += t.count;
    return (n + 3) / 4 * 4;         // the slab-sum kernel moves 16-byte vectors
}

ManualSync::ManualSync(Comm* comm, Mode mode, float* flat_grads, std::vector<Slice> tensors,
                       int root, float* root_slabs, size_t slab_stride)
    : comm_(comm), mode_(mode), grads_(flat_grads), tensors_(std::move(tensors)), root_(root),
      slabs_(root_slabs), stride_(slab_stride) {
    if (!comm_ || !grads_) throw std::invalid_argument("ManualSync: communicator / gradients missing");
    if (root < 0 || root >= comm_->world_size()) throw std::invalid_argument("ManualSync: bad root");
    if (mode_ == GATHER_SCATTER && comm_->rank() == root_) {
        if (!slabs_) throw std::invalid_argument("ManualSync: the root needs its landing slabs");
        if (stride_ < packed_elems(tensors_) || stride_ % 4)
            throw std::invalid_argument("ManualSync: slab stride too small or not a multiple of 4");
    }
}

void ManualSync::run(hipStream_t stream) {
    Comm& c = *comm_;
    const int w = c.world_size();
    if (mode_ == ALL_REDUCE) {
        c.group_start();
        for (const Slice& t : tensors_) c.all_reduce(grads_ + t.offset, t.count, DT_F32, false, stream);
        c.group_end();
        return;
    }
    const bool is_root = c.rank() == root_;
    // ---- gather: rank r's tensor i lands in slab r at the tensor's packed position
    std::vector<void*> recv(is_root ? w : 0);
    c.group_start();
    size_t packed = 0;
    for (const Slice& t : tensors_) {
        if (is_root)
            for (int r = 0; r < w; ++r) recv[r] = slabs_ + (size_t)r * stride_ + packed;
        c.gather(grads_ + t.offset, recv, t.count, DT_F32, root_, stream);
        packed += t.count;
    }
    c.group_end();
    // ---- the root adds slabs 1 .. W-1 into slab 0, every tensor in one pass
    if (is_root) sum_slabs(slabs_, w, stride_, stream);
    // ---- scatter: every rank receives the same summed tensor (part2/2a/main.py:109-114)
    std::vector<const void*> chunks(is_root ? w : 0);
    c.group_start();
    packed = 0;
    for (const Slice& t : tensors_) {
        if (is_root)
            for (int r = 0; r < w; ++r) chunks[r] = slabs_ + packed;
        c.scatter(grads_ + t.offset, chunks, t.count, DT_F32, root_, stream);
        packed += t.count;
    }
    c.group_end();
}

}  // namespace dml
