// Static-plan training engine for the VGG family.
//
// The reference drives every iteration from Python through torch.nn modules and the
// autograd engine (/root/reference/part1/main.py:38-46: zero_grad, forward, loss,
// backward, step).  At MI355X speeds an iteration is about a millisecond, so the per-op
// Python/autograd cost would dominate.  This engine instead walks a fixed layer plan in
// C++ and enqueues every kernel of forward, loss, backward (with the bucket reducer's
// hooks between layers) and the optimizer on the caller's HIP stream plus, for the weight
// gradients, a side stream of its own; nothing in a step touches the host allocator or
// synchronises, so a whole step can also be captured in a hipGraph.
//
// Memory is provided by the caller as six flat device buffers (so torch owns the
// allocations and can view them as tensors):
//   params / grads / momentum : fp32, parameter slots in named_parameters() order; conv
//                               weights stored [cout][3][3][cin] (KRSC), fc [10][512]
//   bn_buffers                : fp32 running_mean / running_var slots
//   arena                     : bf16 - weight shadows, dgrad-layout weights, activations,
//                               one gradient buffer per layer + a transit buffer
//   workspace                 : fp32 - logits, dlogits, loss / accuracy accumulators,
//                               BatchNorm saved statistics, split-K slabs
#pragma once

#include <stdexcept>

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>
#include <array>
#include <vector>

#include "dml_comm.h"
#include "dml_kernels.h"

namespace pybind11 { class module_; }

namespace dml {

struct SlotInfo {
    std::string name;          // state_dict key, e.g. "layers.3.weight"
    size_t offset;             // elements into the flat buffer
    size_t numel;
    std::vector<int> shape;    // internal layout ([cout,3,3,cin] for conv weights)
    int layer;                 // conv layer index, or n_conv for the classifier
};

// Parameter and buffer slots of a VGG-family network in named_parameters() order - the
// state_dict contract of /root/reference/part1/model.py:11-46 - shared by both engines.
struct ParamLayout {
    struct Conv {
        int cin, cout, hw;
        bool pool_after;
        size_t w_off, b_off, gamma_off, beta_off;     // flat params
        size_t rm_off, rv_off;                        // bn_buffers
    };
    std::vector<Conv> convs;
    std::vector<SlotInfo> slots, bn_slots;
    size_t fcw_off = 0, fcb_off = 0, param_elems = 0, bn_elems = 0;
    int fc_in = 0;
};
ParamLayout plan_parameters(const std::vector<int>& widths, bool batch_norm, int num_classes);

class VggEngine {
  public:
    // widths: output channels per conv layer, -1 for a 2x2 max-pool (the VGG width table)
    VggEngine(const std::vector<int>& widths, int max_batch, bool batch_norm, int num_classes);

    size_t param_elems() const { return param_elems_; }
    size_t bn_elems() const { return bn_elems_; }
    size_t arena_elems() const { return arena_elems_; }
    size_t workspace_elems() const { return ws_elems_; }
    const std::vector<SlotInfo>& param_slots() const { return slots_; }
    const std::vector<SlotInfo>& buffer_slots() const { return bn_slots_; }

    void bind(float* params, float* grads, float* momentum, float* bn_buffers, uint16_t* arena,
              float* workspace);
    void set_reducer(BucketReducer* reducer, const std::vector<BucketRange>& buckets);
    // Run the weight-gradient kernels of backward (with their partial-result sums, and the
    // gradient buckets' departure) on a second HIP stream, next to the input-gradient /
    // BatchNorm chain they do not depend on.  Results are identical: every kernel is
    // deterministic and the hand-offs are event-ordered.  Work is handed over in groups of
    // `group` layers because a cross-stream dependency costs ~15 us of host time on ROCm.
    void set_wgrad_overlap(bool on, int group = 2, hipStream_t compute_stream = nullptr);
    // Apply the SGD update inside backward (side-stream mode only), slice by slice, as soon as
    // a slice's gradients are final: behind a group's slab sums, or behind a bucket's
    // all-reduce on the communication stream.  Every element sees exactly the arithmetic of
    // the single launch in optimizer_step(), which then only waits and refreshes the repacks.
    void set_fused_optimizer(bool on, float lr = 0.f, float momentum = 0.f,
                             float weight_decay = 0.f, float grad_scale = 1.f);
    bool wgrad_overlap() const { return wgrad_overlap_; }
    // "value": flag word + wait-value packet, "event": event pair (see StreamSignal)
    const char* hand_off_mode() const { return value_hand_off_ ? "value" : "event"; }
    // Between steps, with the device idle: hand-offs through flag words (true; needs the words
    // allocated when the side stream was created) or through event pairs (false) - the same
    // schedule either way, so two runs from one snapshot must agree bit for bit.
    void set_value_hand_offs(bool on) {
        if (on && !hand_off_flags_)
            throw std::logic_error("engine: flag-word hand-offs are not available on this device");
        value_hand_off_ = on;
    }
    hipStream_t wgrad_stream() const { return wgrad_stream_; }
    // Overlap the parameter update of the deep layers with the first layers of the NEXT forward
    // (side-stream mode only).  optimizer_step() then updates and repacks the conv layers below
    // `first_late_layer` on the caller's stream and everything from that layer on (most of the
    // parameters: the wide layers and the classifier) on the side stream; the next forward - or
    // any other reader, see join_update() - waits for the side stream right before it reaches
    // that layer.  Same kernels on the same values: results are bit-identical.  0 = off.
    void set_update_overlap(int first_late_layer);
    int update_overlap() const { return late_layer_; }
    // make `stream` wait for a deep-layer update still running on the side stream (a no-op when
    // none is pending); every engine entry point that touches parameters does this itself, callers
    // that read the bound parameter / momentum buffers directly call it first
    void join_update(hipStream_t stream);

    // bf16 shadow + conv1 / dgrad repacks from the fp32 master weights
    void refresh_weights(hipStream_t stream);
    // zero grads, forward (training), loss, backward; reducer hooks fire between layers
    void forward_backward(const uint16_t* x_nhwc4, const int64_t* target, int batch,
                          hipStream_t stream);
    // waits for the reducer (if any), fused SGD over the flat buffers, refresh_weights
    void optimizer_step(float lr, float momentum, float weight_decay, float grad_scale,
                        hipStream_t stream);
    // One whole iteration in one call: forward_backward, the manual gradient synchronisation
    // if one is attached (set_manual_sync), optimizer_step.
    void train_step(const uint16_t* x_nhwc4, const int64_t* target, int batch, float lr,
                    float momentum, float weight_decay, float grad_scale, hipStream_t stream);
    void set_manual_sync(ManualSync* sync) { manual_sync_ = sync; }
    // forward only (evaluation statistics); accumulates loss / correct into the eval slots
    void eval_step(const uint16_t* x_nhwc4, const int64_t* target, int batch, hipStream_t stream);
    void forward(const uint16_t* x_nhwc4, int batch, bool training, hipStream_t stream);

    // hipGraph capture of forward_backward + optimizer_step on fixed input buffers
    void capture_train_step(const uint16_t* x_nhwc4, const int64_t* target, int batch, float lr,
                            float momentum, float weight_decay, float grad_scale,
                            hipStream_t stream);
    void replay_train_step(hipStream_t stream);
    bool has_graph() const { return graph_exec_ != nullptr; }

    // workspace addresses the Python side reads
    float* logits() const { return ws_ + logits_off_; }
    float* loss() const { return ws_ + loss_off_; }             // mean loss of the last step
    float* eval_loss() const { return ws_ + loss_off_ + 1; }    // sum of per-batch means
    int32_t* eval_correct() const { return (int32_t*)(ws_ + loss_off_ + 2); }
    size_t logits_offset() const { return logits_off_; }
    size_t loss_offset() const { return loss_off_; }   // [loss, eval_loss, eval_correct(int32)]
    long long bn_steps() const { return bn_steps_; }
    void set_bn_steps(long long v) { bn_steps_ = v; }
    int num_conv() const { return (int)convs_.size(); }
    // per conv layer: {cin, cout, hw, pool_after, pre_off, out_off, pool_off} (arena elements)
    std::vector<std::vector<long long>> layer_table() const;
    std::string debug_tap_bytes() const;
    std::vector<size_t> debug_tap_sections() const { return debug_tap_sections_; }     // host copy of the tap (device synchronised), see DML_DEBUG_TAP
    // workspace offset of each layer's saved [mean | rstd] (2 * cout floats), -1 without BatchNorm
    std::vector<long long> bn_save_offsets() const;
    ~VggEngine();

  private:
    struct Conv {
        int cin, cout, hw;
        bool pool_after;
        size_t w_off, b_off, gamma_off, beta_off;     // flat params
        size_t rm_off, rv_off;                        // bn_buffers
        size_t save_off;                              // workspace: mean[c], rstd[c]
        size_t wd_off;                                // arena: dgrad-layout weights
        size_t wslab_off;                             // workspace: wgrad partial results per split
        size_t pre_off, out_off, pool_off;            // arena: activations
        size_t gout_off;                              // arena: dL/d(conv output), read by dgrad + wgrad
    };
    void backward(const uint16_t* x_nhwc4, const int64_t* target, int batch, hipStream_t stream);
    void fire_buckets(int layer, hipStream_t stream);
    void flush_wgrad_sums(hipStream_t stream);
    void join_wgrad_stream(hipStream_t stream);
    void hand_off_wgrads(int batch, bool last);
    void launch_wgrad(int layer, int batch, hipStream_t stream);
    // Blocks the weight-gradient kernels aim for.  Under BatchNorm the main chain of the
    // two-stream backward is the longer one, and fewer splits (less slab traffic competing with
    // it) make the step faster (288: 0.667 -> 0.656 ms); without BatchNorm the kernel's own
    // optimum (432) does.  Independent of the stream mode on purpose: the single-stream run
    // must add the same partial sums in the same order (it is the race detector's reference).
    int wgrad_split_target() const { return bn_ ? 288 : 0; }
    size_t layer_begin(int layer) const;    // flat offset of the layer's first parameter slot
    void sgd_slice(size_t begin, size_t end, hipStream_t stream);
    void sgd_range(size_t begin, size_t n, float lr, float momentum, float weight_decay,
                   float grad_scale, hipStream_t stream);
    void repack_all(hipStream_t stream, const float* copy_src = nullptr, float* copy_dst = nullptr,
                    size_t copy_n = 0);
    // dgrad layouts of conv layers [lo, hi) (lo >= 1); with `conv1` also the first layer's padded
    // layout, and a small fp32 copy riding in the first launch
    void repack_layers(int lo, int hi, bool conv1, hipStream_t stream, const float* copy_src = nullptr,
                       float* copy_dst = nullptr, size_t copy_n = 0);

    std::vector<Conv> convs_;
    int max_batch_, num_classes_, fc_in_;
    bool bn_;
    size_t fcw_off_ = 0, fcb_off_ = 0;
    size_t param_elems_ = 0, bn_elems_ = 0, arena_elems_ = 0, ws_elems_ = 0;
    size_t shadow_off_ = 0, w1_off_ = 0, gin_off_ = 0;
    size_t logits_off_ = 0, dlogits_off_ = 0, loss_off_ = 0, stats_off_ = 0, slab_off_ = 0,
           cstats_off_ = 0, fc_ws_off_ = 0;
    bool plain_epi_sums_ = true;   // BatchNorm backward sums of un-pooled layers from the dgrad epilogue
    SlabSumJobs wgrad_sums_;      // partial-result reductions not yet launched
    size_t slab_elems_ = 0;
    std::vector<SlotInfo> slots_, bn_slots_;
    float *params_ = nullptr, *grads_ = nullptr, *mom_ = nullptr, *bnbuf_ = nullptr, *ws_ = nullptr;
    uint16_t* arena_ = nullptr;
    BucketReducer* reducer_ = nullptr;
    ManualSync* manual_sync_ = nullptr;
    std::vector<std::vector<int>> buckets_by_layer_;   // layer -> bucket ids to fire
    long long bn_steps_ = 0;
    bool wgrad_overlap_ = false;
    struct { bool on = false; float lr = 0, mu = 0, wd = 0, gscale = 1; } opt_;
    bool opt_applied_ = false;      // this step's update already happened inside backward
    int opt_slice_from_ = 0;        // fused update: layers >= this get a slice right behind their sums
    std::vector<BucketRange> bucket_ranges_;
    hipStream_t wgrad_stream_ = nullptr;
    int wgrad_group_ = 2;
    std::vector<hipEvent_t> hand_off_;      // per layer: main-stream position the side stream waits for
    hipEvent_t side_done_ = nullptr;
    // hand-offs without a marker in the main queue (see StreamSignal): one flag word per layer,
    // raised to the backward pass's sequence number by the layer's backward-data kernel
    unsigned* hand_off_flags_ = nullptr;
    unsigned backward_seq_ = 0;
    bool value_hand_off_ = false;
    bool join_by_value_ = true;             // the side stream's "done" through a flag word as well
    bool side_done_by_value_ = false;       // ... and this backward pass has raised it that way
    std::vector<int> wgrad_queue_;          // layers whose gradient is ready, wgrad not yet launched
    bool tuned_wgrad_splits_ = true;        // use conv3x3_wgrad_tuned_splits where it has an entry
    int experiment_no_wgrad_ = 0;           // timing experiment (wrong gradients): see the constructor
    // debugging tap (DML_DEBUG_TAP=<layer>): device copies of the split slabs and of the gradient
    // that layer's backward-data pass produced, taken in stream order right behind it
    int debug_tap_layer_ = -1;
    char* debug_tap_ = nullptr;
    size_t debug_tap_bytes_ = 0;
    std::vector<size_t> debug_tap_sections_;
    void debug_tap(const void* src, size_t bytes, hipStream_t stream);
    // tuning (DML_DGRAD_PLANS): backward-data plan {bm, bn, splits, pm, waves} per conv layer, bm 0 = default
    std::vector<std::array<int, 5>> dgrad_force_plan_;
    std::vector<std::array<int, 5>> streaming_dgrad_plan_;    // tuned defaults of the streaming plan policy
    std::vector<std::array<int, 5>> fwd_force_plan_;      // DML_FWD_PLANS (tuning): forward plans, same format
    // forward of the 2x2 layers under BatchNorm: split-K finish + statistics + apply in one launch
    // (DML_FUSED_FINISH_BN=1; measured equal to the two launches it replaces, so off by default)
    bool fused_finish_bn_ = false;
    std::vector<int> wgrad_force_splits_;   // tuning (DML_WGRAD_SPLITS): pixel splits per conv layer, 0 = planned
    // tuning (DML_WGRAD_RESIDENT): "bco:stages" per conv layer to run the tap-resident kernel there
    // (with the layer's DML_WGRAD_SPLITS entry as its split count), "0" for the per-tap kernel
    std::vector<std::array<int, 2>> wgrad_resident_;
    bool side_done_recorded_ = false;
    bool side_busy_ = false;                // side-stream work the main stream has not waited for
    hipGraph_t graph_ = nullptr;
    hipGraphExec_t graph_exec_ = nullptr;
    int late_layer_ = 0;                    // first layer updated on the side stream (0: none)
    hipEvent_t grads_final_ = nullptr, late_done_ = nullptr;
    bool late_pending_ = false;             // a side-stream update the main stream has not waited for
    bool capturing_ = false;
};

// The same network with every tensor in fp32 (the reference's precision); see
// csrc/engine/vgg_engine_f32.cpp.  Buffers as for VggEngine, except that the arena is fp32 and
// holds no weight shadow (the kernels read the master weights).
class VggEngineF32 {
  public:
    VggEngineF32(const std::vector<int>& widths, int max_batch, bool batch_norm, int num_classes);

    size_t param_elems() const { return lay_.param_elems; }
    size_t bn_elems() const { return lay_.bn_elems; }
    size_t arena_elems() const { return arena_elems_; }
    size_t workspace_elems() const { return ws_elems_; }
    const std::vector<SlotInfo>& param_slots() const { return lay_.slots; }
    const std::vector<SlotInfo>& buffer_slots() const { return lay_.bn_slots; }
    int num_conv() const { return (int)lay_.convs.size(); }
    long long bn_steps() const { return bn_steps_; }
    void set_bn_steps(long long v) { bn_steps_ = v; }
    size_t logits_offset() const { return logits_off_; }
    size_t loss_offset() const { return loss_off_; }   // [loss, eval_loss, eval_correct(int32)]
    // per conv layer: {cin, cout, hw, pool_after, pre_off, out_off, pool_off, gout_off} (arena
    // elements; gout = dL/d(conv output) of the last backward pass)
    std::vector<std::vector<long long>> layer_table() const;
    std::vector<long long> bn_save_offsets() const;

    void bind(float* params, float* grads, float* momentum, float* bn_buffers, float* arena,
              float* workspace);
    void set_reducer(BucketReducer* reducer, const std::vector<BucketRange>& buckets);
    void set_manual_sync(ManualSync* sync) { manual_sync_ = sync; }
    void refresh_weights(hipStream_t stream);       // backward-data weight layouts
    void forward(const float* x_nhwc4, int batch, bool training, hipStream_t stream);
    void forward_backward(const float* x_nhwc4, const int64_t* target, int batch,
                          hipStream_t stream);
    void optimizer_step(float lr, float momentum, float weight_decay, float grad_scale,
                        hipStream_t stream);
    void train_step(const float* x_nhwc4, const int64_t* target, int batch, float lr,
                    float momentum, float weight_decay, float grad_scale, hipStream_t stream);
    void eval_step(const float* x_nhwc4, const int64_t* target, int batch, hipStream_t stream);
    // Weight gradients (with bias gradients, slab sums and the buckets they complete) on a second
    // HIP stream next to the pool / BatchNorm / backward-data chain, one event per layer; the
    // main stream joins it at the end of backward.  Same kernels on the same values.
    void set_wgrad_overlap(bool on, hipStream_t compute_stream = nullptr);
    bool wgrad_overlap() const { return overlap_; }
    const char* hand_off_mode() const { return value_hand_off_ ? "value" : "event"; }
    // Between steps, with the device idle: hand-offs through flag words (true; needs the words
    // allocated when the side stream was created) or through event pairs (false) - the same
    // schedule either way, so two runs from one snapshot must agree bit for bit.
    void set_value_hand_offs(bool on) {
        if (on && !hand_off_flags_)
            throw std::logic_error("engine: flag-word hand-offs are not available on this device");
        value_hand_off_ = on;
    }
    hipStream_t wgrad_stream() const { return side_; }
    ~VggEngineF32();

  private:
    struct Buffers {
        size_t wd_off = 0, pre_off = 0, out_off = 0, pool_off = 0, gout_off = 0;   // arena
        // arena: a transit buffer of the layer's own for the gradient arriving from the next
        // layer, where that layer's backward-data kernel may split its taps (kNone: shared one)
        size_t gin_off = kNone;
        size_t save_off = 0, wslab_off = 0;                                        // workspace
    };
    static constexpr size_t kNone = ~(size_t)0;
    void backward(const float* x, const int64_t* target, int batch, hipStream_t stream);
    void fire_buckets(int layer, hipStream_t stream);
    void flush_sums(hipStream_t stream);

    ParamLayout lay_;
    std::vector<Buffers> buf_;
    int max_batch_, num_classes_;
    bool bn_;
    size_t arena_elems_ = 0, ws_elems_ = 0, ta_off_ = 0, tb_off_ = 0;
    size_t logits_off_ = 0, loss_off_ = 0, stats_off_ = 0, fc_ws_off_ = 0, c1_ws_off_ = 0;
    float *params_ = nullptr, *grads_ = nullptr, *mom_ = nullptr, *bnbuf_ = nullptr,
          *arena_ = nullptr, *ws_ = nullptr;
    BucketReducer* reducer_ = nullptr;
    ManualSync* manual_sync_ = nullptr;
    std::vector<std::vector<int>> buckets_by_layer_;
    SlabSumJobs sums_;
    long long bn_steps_ = 0;
    bool overlap_ = false;
    hipStream_t side_ = nullptr;
    std::vector<hipEvent_t> grad_ready_;    // per layer: dL/d(conv output) complete on the main stream
    // the same hand-offs through flag words and wait-value packets (see StreamSignal)
    unsigned* hand_off_flags_ = nullptr;
    unsigned backward_seq_ = 0;
    bool value_hand_off_ = false;
    // ... and the side stream's three joins with the main stream
    enum { kJoinSideDone = 0, kJoinRepack = 1, kJoinCleared = 2 };
    unsigned join_seq_[3] = {0, 0, 0};
    void side_raise(int word, hipEvent_t event);
    void main_await(int word, hipEvent_t event, hipStream_t stream);
    hipEvent_t side_done_ = nullptr, updated_ = nullptr, repack_done_ = nullptr;
    bool repack_pending_ = false;           // weight layouts of the last update still on the side stream
    void join_repack(hipStream_t stream);
    // Outputs of the tap-split convolutions (two blocks per position meet through atomics, so the
    // output starts at zero): all of a step's are cleared by one launch at the start of forward -
    // on the side stream when there is one - instead of a memset in front of each convolution.
    std::vector<int> wgrad_force_splits_;   // tuning (DML_WGRAD_F32_SPLITS): pixel splits per conv layer, 0 = planned
    std::vector<char> fwd_cleared_, bwd_cleared_;   // per layer: forward / backward-data output
    hipEvent_t fwd_begin_ = nullptr, cleared_ = nullptr;
    bool clear_pending_ = false;
    void clear_split_outputs(int batch, bool training, hipStream_t stream);
    void join_clear(hipStream_t stream);
    float* arriving(int layer) const;
    bool direct(int layer) const { return !bn_ && !lay_.convs[layer].pool_after; }
};

void bind_engine(pybind11::module_& m);
void bind_engine_f32(pybind11::module_& m);

}  // namespace dml
