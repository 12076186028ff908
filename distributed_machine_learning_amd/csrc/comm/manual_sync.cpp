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
#include <cstdlib>
#include <cstring>
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
    // tensors back to back, first one on a 16-byte boundary: the root sums in place (see run)
    in_place_ = !tensors_.empty() && ((uintptr_t)(grads_ + tensors_[0].offset) & 15) == 0 &&
                !(getenv("DML_GATHER_IN_PLACE") && !strcmp(getenv("DML_GATHER_IN_PLACE"), "0"));
    for (size_t i = 1; i < tensors_.size(); ++i)
        if (tensors_[i].offset != tensors_[i - 1].offset + tensors_[i - 1].count) in_place_ = false;
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
    // The root's own contribution never moves when the tensors lie back to back in the gradient
    // buffer (they do: it is one flat allocation): its gradients ARE its entry of the gathered
    // list, the peers' land in slabs 0 .. W-2, the sum is formed in the gradient buffer and the
    // scatter sends from there.  (The reference's root gathers its own tensor into list[0], adds
    // the others to it and scatters it back to itself, part2/2a/main.py:101-114 - two copies of
    // the whole model per step that change nothing.)
    if (in_place_) {
        const size_t base = tensors_.empty() ? 0 : tensors_[0].offset;
        std::vector<void*> recv(is_root ? w : 0);
        c.group_start();
        size_t packed = 0;
        for (const Slice& t : tensors_) {
            if (is_root)
                for (int r = 0, p = 0; r < w; ++r)
                    recv[r] = r == root_ ? (void*)(grads_ + t.offset)
                                         : (void*)(slabs_ + (size_t)(p++) * stride_ + packed);
            c.gather(grads_ + t.offset, recv, t.count, DT_F32, root_, stream);
            packed += t.count;
        }
        c.group_end();
        if (is_root) sum_slabs_into(grads_ + base, slabs_, w - 1, stride_, packed, stream);
        std::vector<const void*> chunks(is_root ? w : 0);
        c.group_start();
        for (const Slice& t : tensors_) {
            if (is_root)
                for (int r = 0; r < w; ++r) chunks[r] = grads_ + t.offset;
            c.scatter(grads_ + t.offset, chunks, t.count, DT_F32, root_, stream);
        }
        c.group_end();
        return;
    }
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
