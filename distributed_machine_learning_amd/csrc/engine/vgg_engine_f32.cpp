// VggEngineF32: the training step at the reference's precision.
//
// The reference trains in fp32 end to end (/root/reference/part1/main.py:15-16,38-46): fp32
// weights, activations, gradients and optimizer state.  This engine runs the same fixed layer
// plan as VggEngine with every tensor in fp32 - activations NHWC fp32, the master weights used
// directly by the kernels (no low-precision shadow), convolutions on the fp32-input matrix
// instruction (exact fp32 products and sums).  At 1/16 of the bf16 matrix rate the step is
// bound by the convolutions' arithmetic, so the schedule is the plain one: a single stream,
// unfused BatchNorm / pool kernels (the templates VggEngine's kernels are instantiated from),
// bucket hooks between layers.
//
//   forward : conv(+bias[+ReLU]) -> [BatchNorm(train stats)+ReLU] -> [max-pool]   per layer
//   loss    : classifier rows + batch reductions (logits, loss, dlogits, dx, dW, db)
//   backward: [pool bwd] -> [BatchNorm+ReLU bwd] -> bias grad -> wgrad (+ slab sum) -> dgrad
//   update  : reducer wait / manual sync -> fused SGD -> backward-data weight repacks
#include <cstring>
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>

#include <cstdlib>
#include <stdexcept>

#include "dml_engine.h"
#include "dml_kernels.h"
#include "dml_trace.h"

namespace py = pybind11;

namespace dml {

namespace {
constexpr size_t kAlign = 64;
size_t align_up(size_t v) { return (v + kAlign - 1) / kAlign * kAlign; }

void check_hip(hipError_t e, const char* what) {
    if (e != hipSuccess)
        throw std::runtime_error(std::string("HIP ") + what + " failed: " + hipGetErrorString(e));
}
}  // namespace

VggEngineF32::VggEngineF32(const std::vector<int>& widths, int max_batch, bool batch_norm,
                           int num_classes)
    : max_batch_(max_batch), num_classes_(num_classes), bn_(batch_norm) {
    if (max_batch <= 0) throw std::invalid_argument("VggEngineF32: batch must be positive");
    lay_ = plan_parameters(widths, batch_norm, num_classes);
    const size_t B = (size_t)max_batch_;
    size_t a = 0, max_act = 0;
    for (size_t i = 0; i < lay_.convs.size(); ++i) {
        const ParamLayout::Conv& c = lay_.convs[i];
        if (i > 0 && c.cout % 128)
            throw std::invalid_argument("VggEngineF32: layers after the first need a multiple of "
                                        "128 output channels");
        Buffers b;
        const size_t act = B * c.hw * c.hw * c.cout;
        max_act = act > max_act ? act : max_act;
        b.wd_off = a;
        if (i > 0) a = align_up(a + (size_t)c.cout * 9 * c.cin);
        b.pre_off = a;
        if (bn_) a = align_up(a + act);
        b.out_off = a; a = align_up(a + act);
        b.pool_off = a;
        if (c.pool_after) a = align_up(a + act / 4);
        b.gout_off = a; a = align_up(a + act);
        // (position-major backward-data tiles, the ones that may split, exist up to 8x8 maps)
        if (i + 1 < lay_.convs.size() && (bn_ || c.pool_after) && lay_.convs[i + 1].hw <= 8) {
            const ParamLayout::Conv& nx = lay_.convs[i + 1];
            b.gin_off = a; a = align_up(a + B * nx.hw * nx.hw * nx.cin);
        }
        buf_.push_back(b);
    }
    ta_off_ = a; a = align_up(a + max_act);
    tb_off_ = a; a = align_up(a + max_act);
    arena_elems_ = a;

    wgrad_force_splits_.assign(lay_.convs.size(), 0);
    if (const char* e = getenv("DML_WGRAD_F32_SPLITS")) {  // tuning: "s1,s2,..." for conv layers 1..
        size_t i = 1;
        for (const char* q = e; *q && i < lay_.convs.size(); ++i) {
            wgrad_force_splits_[i] = atoi(q);
            while (*q && *q != ',') ++q;
            if (*q == ',') ++q;
        }
    }
    size_t w = 0;
    logits_off_ = w; w = align_up(w + B * 16);
    loss_off_ = w;   w = align_up(w + 64);
    stats_off_ = w;  w = align_up(w + bn_workspace_elems());
    fc_ws_off_ = w;  w = align_up(w + fc_train_workspace_elems(max_batch_, lay_.fc_in, num_classes_));
    c1_ws_off_ = w;  w = align_up(w + conv1_wgrad_f32_workspace_elems(max_batch_));
    for (size_t i = 0; i < lay_.convs.size(); ++i) {
        const ParamLayout::Conv& c = lay_.convs[i];
        buf_[i].save_off = w;
        if (bn_) w = align_up(w + 2 * (size_t)c.cout);
        buf_[i].wslab_off = w;
        if (i > 0) {
            const ConvShape full = {max_batch_, c.hw, c.cin, c.cout};
            int splits = conv3x3_wgrad_f32_splits(full);
            if (wgrad_force_splits_[i] > 0) {
                const int f = conv3x3_wgrad_f32_splits(full, wgrad_force_splits_[i]);
                splits = f > splits ? f : splits;
            }
            if (splits > 1) w = align_up(w + (size_t)splits * c.cout * 9 * c.cin);
        }
    }
    ws_elems_ = w;
    buckets_by_layer_.assign(lay_.convs.size() + 1, {});
}

void VggEngineF32::bind(float* params, float* grads, float* momentum, float* bn_buffers,
                        float* arena, float* workspace) {
    if (!params || !grads || !momentum || !arena || !workspace || (bn_ && !bn_buffers))
        throw std::invalid_argument("VggEngineF32::bind: null buffer");
    params_ = params; grads_ = grads; mom_ = momentum; bnbuf_ = bn_buffers;
    arena_ = arena; ws_ = workspace;
}

VggEngineF32::~VggEngineF32() {
    for (hipEvent_t e : grad_ready_) (void)hipEventDestroy(e);
    if (side_done_) (void)hipEventDestroy(side_done_);
    if (updated_) (void)hipEventDestroy(updated_);
    if (repack_done_) (void)hipEventDestroy(repack_done_);
    if (fwd_begin_) (void)hipEventDestroy(fwd_begin_);
    if (cleared_) (void)hipEventDestroy(cleared_);
    if (side_) (void)hipStreamDestroy(side_);
    if (hand_off_flags_) (void)hipFree(hand_off_flags_);
}

void VggEngineF32::set_wgrad_overlap(bool on, hipStream_t compute_stream) {
    if (on && !side_) {
        // a hardware queue of its own, or the two streams run in line (see stream_probe.hip)
        std::vector<hipStream_t> avoid = {compute_stream};
        if (reducer_ && reducer_->comm_stream()) avoid.push_back(reducer_->comm_stream());
        side_ = create_concurrent_stream(avoid);
        grad_ready_.resize(lay_.convs.size());
        // (events between two queues of one GPU: no system-scope fence of their own, the kernels
        // release their writes to the device as they end; DML_EVENT_SCOPE=system restores it)
        unsigned flags = hipEventDisableTiming | hipEventDisableSystemFence;
        if (const char* e = getenv("DML_EVENT_SCOPE"))
            if (!strcmp(e, "system")) flags = hipEventDisableTiming;
        for (hipEvent_t& e : grad_ready_)
            check_hip(hipEventCreateWithFlags(&e, flags), "hipEventCreate");
        for (hipEvent_t* e : {&side_done_, &updated_, &repack_done_, &fwd_begin_, &cleared_})
            check_hip(hipEventCreateWithFlags(e, flags), "hipEventCreate");
        int can_wait_value = 0, device = 0;
        check_hip(hipGetDevice(&device), "hipGetDevice");
        (void)hipDeviceGetAttribute(&can_wait_value, hipDeviceAttributeCanUseStreamWaitValue, device);
        const char* mode = getenv("DML_HANDOFF");
        value_hand_off_ = can_wait_value && !(mode && !strcmp(mode, "event"));
        if (value_hand_off_) {
            // one word per layer + three for the side stream's joins (see side_raise)
            const size_t bytes = (lay_.convs.size() + 3) * sizeof(unsigned);
            check_hip(hipMalloc((void**)&hand_off_flags_, bytes), "hipMalloc(hand-off flags)");
            check_hip(hipMemset(hand_off_flags_, 0, bytes), "hipMemset(hand-off flags)");
            check_hip(hipDeviceSynchronize(), "hipDeviceSynchronize(hand-off flags)");
        }
    }
    overlap_ = on;
}

void VggEngineF32::set_reducer(BucketReducer* reducer, const std::vector<BucketRange>& buckets) {
    reducer_ = reducer;
    buckets_by_layer_.assign(lay_.convs.size() + 1, {});
    if (!reducer) return;
    if ((int)buckets.size() != reducer->num_buckets())
        throw std::invalid_argument("VggEngineF32::set_reducer: bucket count mismatch");
    for (size_t b = 0; b < buckets.size(); ++b) {
        const int lo = buckets[b].lo;
        if (lo < 0 || lo >= (int)lay_.slots.size())
            throw std::invalid_argument("VggEngineF32::set_reducer: bucket range outside the slots");
        // complete once the layer owning its first (earliest) slot has run backward
        buckets_by_layer_[lay_.slots[lo].layer].push_back((int)b);
    }
}

void VggEngineF32::fire_buckets(int layer, hipStream_t stream) {
    if (!reducer_ || buckets_by_layer_[layer].empty()) return;
    TraceRange range("backward/buckets");
    flush_sums(stream);
    for (int b : buckets_by_layer_[layer]) reducer_->bucket_ready(b, stream);
}

void VggEngineF32::flush_sums(hipStream_t stream) {
    if (sums_.n == 0) return;
    sum_slabs_batch(ws_, grads_, sums_, stream);
    sums_.n = 0;
}

// The side stream tells the main stream "done up to here": through a flag word raised by a
// one-wave kernel and a wait-value packet where the device has them (a queue that waits for
// another queue's EVENT pays ~6 us for it, for a word in memory ~3), else through the event.
void VggEngineF32::side_raise(int word, hipEvent_t event) {
    if (value_hand_off_)
        enqueue_flag_raise(hand_off_flags_ + lay_.convs.size() + word, ++join_seq_[word], side_);
    else
        check_hip(hipEventRecord(event, side_), "hipEventRecord(side)");
}

void VggEngineF32::main_await(int word, hipEvent_t event, hipStream_t stream) {
    if (value_hand_off_)
        check_hip(hipStreamWaitValue32(stream, hand_off_flags_ + lay_.convs.size() + word,
                                       join_seq_[word], hipStreamWaitValueGte, 0xffffffffu),
                  "hipStreamWaitValue32(side)");
    else
        check_hip(hipStreamWaitEvent(stream, event, 0), "hipStreamWaitEvent(side)");
}

void VggEngineF32::join_repack(hipStream_t stream) {
    if (!repack_pending_) return;
    main_await(kJoinRepack, repack_done_, stream);
    repack_pending_ = false;
}

// where the gradient arriving at a layer is written: straight into the layer's own buffer when it
// already is dL/d(conv output) (no BatchNorm, no pool: the producer masks with the layer's ReLU
// output), else a transit buffer
float* VggEngineF32::arriving(int layer) const {
    if (direct(layer)) return arena_ + buf_[layer].gout_off;
    return arena_ + (buf_[layer].gin_off != kNone ? buf_[layer].gin_off : ta_off_);
}

void VggEngineF32::clear_split_outputs(int batch, bool training, hipStream_t stream) {
    const int nconv = (int)lay_.convs.size();
    fwd_cleared_.assign(nconv, 0);
    bwd_cleared_.assign(nconv, 0);
    ZeroJobs jobs;
    for (int i = 1; i < nconv; ++i) {
        const ParamLayout::Conv& c = lay_.convs[i];
        const size_t pixels = (size_t)batch * c.hw * c.hw;
        if (bn_ && jobs.n < ZeroJobs::kMax &&
            conv3x3_igemm_f32_tap_splits({batch, c.hw, c.cin, c.cout}, EPI_BIAS) == 2) {
            jobs.ptr[jobs.n] = arena_ + buf_[i].pre_off;
            jobs.count[jobs.n++] = pixels * c.cout;
            fwd_cleared_[i] = 1;
        }
        // (the shared transit buffer carries other layers' gradients in between: not cleared ahead)
        const bool own = direct(i - 1) || buf_[i - 1].gin_off != kNone;
        if (training && own && jobs.n < ZeroJobs::kMax &&
            conv3x3_igemm_f32_tap_splits({batch, c.hw, c.cout, c.cin},
                                         direct(i - 1) ? EPI_MASK : EPI_NONE) == 2) {
            jobs.ptr[jobs.n] = arriving(i - 1);
            jobs.count[jobs.n++] = pixels * c.cin;
            bwd_cleared_[i] = 1;
        }
    }
    if (jobs.n == 0) return;
    if (overlap_) {
        // the side stream idles through forward; the first split convolution is far down the net
        check_hip(hipEventRecord(fwd_begin_, stream), "hipEventRecord(forward begin)");
        check_hip(hipStreamWaitEvent(side_, fwd_begin_, 0), "hipStreamWaitEvent(forward begin)");
        zero_buffers(jobs, side_);
        side_raise(kJoinCleared, cleared_);
        clear_pending_ = true;
    } else {
        zero_buffers(jobs, stream);
    }
}

void VggEngineF32::join_clear(hipStream_t stream) {
    if (!clear_pending_) return;
    main_await(kJoinCleared, cleared_, stream);
    clear_pending_ = false;
}

void VggEngineF32::refresh_weights(hipStream_t stream) {
    if (!params_) throw std::logic_error("VggEngineF32: bind() first");
    join_repack(stream);        // (a repack of the previous update may still run on the side stream)
    for (size_t i = 1; i < lay_.convs.size(); ++i) {
        const ParamLayout::Conv& c = lay_.convs[i];
        repack_dgrad_weights_f32(params_ + c.w_off, arena_ + buf_[i].wd_off, c.cout, c.cin, stream);
    }
}

std::vector<std::vector<long long>> VggEngineF32::layer_table() const {
    std::vector<std::vector<long long>> out;
    for (size_t i = 0; i < lay_.convs.size(); ++i) {
        const ParamLayout::Conv& c = lay_.convs[i];
        out.push_back({c.cin, c.cout, c.hw, c.pool_after ? 1 : 0, (long long)buf_[i].pre_off,
                       (long long)buf_[i].out_off, (long long)buf_[i].pool_off,
                       (long long)buf_[i].gout_off});
    }
    return out;
}

std::vector<long long> VggEngineF32::bn_save_offsets() const {
    std::vector<long long> out;
    for (const Buffers& b : buf_) out.push_back(bn_ ? (long long)b.save_off : -1);
    return out;
}

void VggEngineF32::forward(const float* x, int batch, bool training, hipStream_t stream) {
    if (!params_) throw std::logic_error("VggEngineF32: bind() first");
    if (batch <= 0 || batch > max_batch_) throw std::invalid_argument("VggEngineF32: bad batch size");
    TraceRange range(training ? "forward" : "forward(eval)");
    const float* in = x;
    clear_split_outputs(batch, training, stream);
    for (size_t i = 0; i < lay_.convs.size(); ++i) {
        const ParamLayout::Conv& c = lay_.convs[i];
        const Buffers& b = buf_[i];
        float* dst = arena_ + (bn_ ? b.pre_off : b.out_off);
        if (fwd_cleared_[i]) join_clear(stream);
        if (i == 0)
            conv1_fprop_f32(in, params_ + c.w_off, params_ + c.b_off, dst, batch, !bn_, stream);
        else
            conv3x3_igemm_f32(in, params_ + c.w_off, dst, params_ + c.b_off, nullptr,
                              {batch, c.hw, c.cin, c.cout}, bn_ ? EPI_BIAS : EPI_BIAS_RELU, stream,
                              0, -1, -1, fwd_cleared_[i] != 0);
        if (bn_ && c.pool_after) {
            // BatchNorm + ReLU + pool in one pass: only the pooled activation is stored (backward
            // re-derives the window winners from the conv output)
            if (training)
                bn_relu_pool_fwd_train(dst, arena_ + b.pool_off, params_ + c.gamma_off,
                                       params_ + c.beta_off, bnbuf_ + c.rm_off, bnbuf_ + c.rv_off,
                                       ws_ + b.save_off, ws_ + b.save_off + c.cout,
                                       ws_ + stats_off_, batch, c.hw, c.cout, 0.1f, 1e-5f, stream);
            else
                bn_relu_pool_fwd_eval(dst, arena_ + b.pool_off, params_ + c.gamma_off,
                                      params_ + c.beta_off, bnbuf_ + c.rm_off, bnbuf_ + c.rv_off,
                                      batch, c.hw, c.cout, 1e-5f, stream);
            in = arena_ + b.pool_off;
            continue;
        }
        if (bn_) {
            const int rows = batch * c.hw * c.hw;
            if (training)
                bn_relu_fwd_train(dst, arena_ + b.out_off, params_ + c.gamma_off,
                                  params_ + c.beta_off, bnbuf_ + c.rm_off, bnbuf_ + c.rv_off,
                                  ws_ + b.save_off, ws_ + b.save_off + c.cout, ws_ + stats_off_,
                                  rows, c.cout, 0.1f, 1e-5f, stream);
            else
                bn_relu_fwd_eval(dst, arena_ + b.out_off, params_ + c.gamma_off,
                                 params_ + c.beta_off, bnbuf_ + c.rm_off, bnbuf_ + c.rv_off, rows,
                                 c.cout, 1e-5f, stream);
        }
        in = arena_ + b.out_off;
        if (c.pool_after) {
            maxpool2x2_fwd(in, arena_ + b.pool_off, batch, c.hw, c.cout, stream);
            in = arena_ + b.pool_off;
        }
    }
    if (!training)
        linear_fwd(in, params_ + lay_.fcw_off, params_ + lay_.fcb_off, ws_ + logits_off_, batch,
                   lay_.fc_in, num_classes_, stream);
    if (training && bn_) ++bn_steps_;
}

void VggEngineF32::backward(const float* x, const int64_t* target, int batch, hipStream_t stream) {
    TraceRange range("backward");
    const int nconv = (int)lay_.convs.size();
    // the backward-data weight layouts of the last update were written on the side stream while
    // forward ran (see optimizer_step)
    join_repack(stream);
    join_clear(stream);
    sums_.n = 0;
    float* tb = arena_ + tb_off_;
    const bool by_value = overlap_ && value_hand_off_;
    // (a join word is raised up to twice per step and so runs ahead of the backward count: whichever
    // counter is within a step's worth of 2^32 starts all the words over)
    const unsigned top = 0xfffffff0u;
    const bool joins_near_wrap = join_seq_[0] >= top || join_seq_[1] >= top || join_seq_[2] >= top;
    if (by_value && (++backward_seq_ == 0 || joins_near_wrap)) {
        // the 32-bit sequence wrapped (2^32 backward passes): start the words over
        check_hip(hipDeviceSynchronize(), "hipDeviceSynchronize(sequence wrap)");
        check_hip(hipMemset(hand_off_flags_, 0, (lay_.convs.size() + 3) * sizeof(unsigned)), "hipMemset(hand-off flags)");
        backward_seq_ = 1;
        join_seq_[0] = join_seq_[1] = join_seq_[2] = 0;
    }

    const ParamLayout::Conv& last = lay_.convs[nconv - 1];
    const float* fc_in = arena_ + (last.pool_after ? buf_[nconv - 1].pool_off : buf_[nconv - 1].out_off);
    float* fc_dx = direct(nconv - 1) ? tb : arriving(nconv - 1);
    fc_train_rows(fc_in, params_ + lay_.fcw_off, params_ + lay_.fcb_off, target, ws_ + logits_off_,
                  fc_dx, ws_ + fc_ws_off_, batch, lay_.fc_in, num_classes_, stream);
    fc_train_grads(fc_in, ws_ + fc_ws_off_, ws_ + loss_off_, nullptr, grads_ + lay_.fcw_off,
                   grads_ + lay_.fcb_off, batch, lay_.fc_in, num_classes_, stream);
    if (direct(nconv - 1))      // a classifier straight behind an un-pooled ReLU layer
        relu_mask_f32(tb, arena_ + buf_[nconv - 1].out_off, arena_ + buf_[nconv - 1].gout_off,
                      (size_t)batch * last.hw * last.hw * last.cout, stream);
    fire_buckets(nconv, stream);

    for (int i = nconv - 1; i >= 0; --i) {
        const ParamLayout::Conv& c = lay_.convs[i];
        const Buffers& b = buf_[i];
        const int rows = batch * c.hw * c.hw;
        float* g = arena_ + b.gout_off;             // dL/d(conv output)
        float* src = arriving(i);
        if (bn_ && c.pool_after)
            // pool routing, ReLU mask and BatchNorm backward in one reduction and one apply pass
            bn_relu_pool_bwd(arena_ + b.pre_off, src, g, params_ + c.gamma_off,
                             params_ + c.beta_off, ws_ + b.save_off, ws_ + b.save_off + c.cout,
                             grads_ + c.gamma_off, grads_ + c.beta_off, ws_ + stats_off_, batch,
                             c.hw, c.cout, stream);
        else if (c.pool_after)
            // without BatchNorm the ReLU mask rides in the pool's routing (out is post-ReLU)
            maxpool2x2_bwd(arena_ + b.out_off, src, g, batch, c.hw, c.cout, true, stream);
        else if (bn_)
            bn_relu_bwd(arena_ + b.pre_off, src, g, params_ + c.gamma_off, params_ + c.beta_off,
                        ws_ + b.save_off, ws_ + b.save_off + c.cout, grads_ + c.gamma_off,
                        grads_ + c.beta_off, ws_ + stats_off_, rows, c.cout, stream);
        // g is complete: from here the weight gradient (side stream when overlapping) and the
        // input gradient (main stream) are independent.  The main chain is enqueued first.
        // (the first layer's weight gradient stays on the main stream: by then the main chain has
        // nothing else left and the side stream still has the second layer's to finish)
        hipStream_t wstream = stream;
        if (overlap_ && i > 0) {
            if (!by_value)
                check_hip(hipEventRecord(grad_ready_[i], stream), "hipEventRecord(grad ready)");
            wstream = side_;
        }
        // by value: the backward-data kernel, next in the main queue, raises the layer's flag
        const StreamSignal raise = {hand_off_flags_ ? hand_off_flags_ + i : nullptr, backward_seq_};
        if (i > 0) {
            // input gradient; the producer's ReLU is folded in when nothing sits in between
            const bool fold = direct(i - 1);
            conv3x3_igemm_f32(g, arena_ + b.wd_off, arriving(i - 1), nullptr,
                              fold ? arena_ + buf_[i - 1].out_off : nullptr,
                              {batch, c.hw, c.cout, c.cin}, fold ? EPI_MASK : EPI_NONE, stream,
                              0, -1, -1, bwd_cleared_[i] != 0, by_value ? &raise : nullptr);
        }
        if (by_value && i > 0)
            check_hip(hipStreamWaitValue32(side_, hand_off_flags_ + i, backward_seq_,
                                           hipStreamWaitValueGte, 0xffffffffu),
                      "hipStreamWaitValue32(grad ready)");
        else if (overlap_ && i > 0)
            check_hip(hipStreamWaitEvent(side_, grad_ready_[i], 0), "hipStreamWaitEvent(grad ready)");
        if (i == 0) {
            // Conv bias gradient under BatchNorm: exactly zero in exact arithmetic (the
            // normalisation removes any per-channel offset), left at the buffer's zero.
            conv1_wgrad_f32(x, g, grads_ + c.w_off, bn_ ? nullptr : grads_ + c.b_off, batch,
                            ws_ + c1_ws_off_, wstream);
        } else {
            const ParamLayout::Conv& prev = lay_.convs[i - 1];
            const float* in = arena_ + (prev.pool_after ? buf_[i - 1].pool_off : buf_[i - 1].out_off);
            // (the reduction scratch is free: BatchNorm, its other user, is off in this case)
            if (!bn_) column_sum(g, grads_ + c.b_off, rows, c.cout, ws_ + stats_off_, wstream);
            const ConvShape s = {batch, c.hw, c.cin, c.cout};
            const int forced = wgrad_force_splits_[i];
            conv3x3_wgrad_f32(in, g, grads_ + c.w_off, s, ws_ + b.wslab_off, wstream, forced);
            const int splits = conv3x3_wgrad_f32_splits(s, forced);
            if (splits > 1) {
                if (sums_.n == SlabSumJobs::kMax) flush_sums(wstream);
                const int j = sums_.n++;
                sums_.src_off[j] = b.wslab_off;
                sums_.dst_off[j] = c.w_off;
                sums_.count[j] = (size_t)c.cout * 9 * c.cin;
                sums_.splits[j] = splits;
            }
        }
        // The side stream has time to spare (it idles through forward): its slab sums follow
        // each weight gradient at once instead of piling up behind the last one.
        if (overlap_ && i > 0) flush_sums(side_);
        // a bucket leaves from the stream that completed it (the BatchNorm gradients it may hold
        // were written on the main stream before grad_ready_[i]); the last one only after the side
        // stream's share of it is complete
        if (overlap_ && i == 0 && reducer_ && !buckets_by_layer_[0].empty()) {
            side_raise(kJoinSideDone, side_done_);
            main_await(kJoinSideDone, side_done_, stream);
        }
        fire_buckets(i, wstream);
    }
    flush_sums(stream);
    if (overlap_) {
        side_raise(kJoinSideDone, side_done_);
        main_await(kJoinSideDone, side_done_, stream);
    }
}

void VggEngineF32::forward_backward(const float* x, const int64_t* target, int batch,
                                    hipStream_t stream) {
    if (!params_) throw std::logic_error("VggEngineF32: bind() first");
    if (reducer_) {
        reducer_->begin_step();
        reducer_->apply_buffers(stream);        // running statistics := rank 0's
    }
    forward(x, batch, true, stream);
    if (reducer_ && bn_) reducer_->buffers_updated(stream);
    backward(x, target, batch, stream);
}

void VggEngineF32::optimizer_step(float lr, float momentum, float weight_decay, float grad_scale,
                                  hipStream_t stream) {
    TraceRange range("optimizer");
    if (reducer_) reducer_->wait_all(stream);
    sgd_momentum(params_, grads_, mom_, nullptr, lay_.param_elems, lr, momentum, weight_decay,
                 grad_scale, false, stream);
    const float* bsrc = nullptr; float* bdst = nullptr; size_t bn = 0;
    if (reducer_ && reducer_->take_pending_buffers(&bsrc, &bdst, &bn))
        check_hip(hipMemcpyAsync(bdst, bsrc, bn * sizeof(float), hipMemcpyDeviceToDevice, stream),
                  "hipMemcpyAsync(buffers)");
    if (overlap_) {
        // Only backward reads the repacked weights: lay them out on the side stream, next to
        // the forward pass of the following step.
        check_hip(hipEventRecord(updated_, stream), "hipEventRecord(updated)");
        check_hip(hipStreamWaitEvent(side_, updated_, 0), "hipStreamWaitEvent(updated)");
        for (size_t i = 1; i < lay_.convs.size(); ++i) {
            const ParamLayout::Conv& c = lay_.convs[i];
            repack_dgrad_weights_f32(params_ + c.w_off, arena_ + buf_[i].wd_off, c.cout, c.cin, side_);
        }
        side_raise(kJoinRepack, repack_done_);
        repack_pending_ = true;
    } else {
        refresh_weights(stream);
    }
}

void VggEngineF32::train_step(const float* x, const int64_t* target, int batch, float lr,
                              float momentum, float weight_decay, float grad_scale,
                              hipStream_t stream) {
    TraceRange range("train_step");
    forward_backward(x, target, batch, stream);
    if (manual_sync_) {
        TraceRange sync(manual_sync_->mode() == ManualSync::ALL_REDUCE ? "sync/all_reduce"
                                                                        : "sync/gather_scatter");
        manual_sync_->run(stream);
    }
    optimizer_step(lr, momentum, weight_decay, grad_scale, stream);
}

void VggEngineF32::eval_step(const float* x, const int64_t* target, int batch, hipStream_t stream) {
    forward(x, batch, false, stream);
    softmax_xent(ws_ + logits_off_, target, ws_ + loss_off_ + 1, nullptr,
                 (int32_t*)(ws_ + loss_off_ + 2), batch, num_classes_, 1.0f, stream);
}

void bind_engine_f32(py::module_& m) {
    py::class_<VggEngineF32>(m, "VggEngineF32")
        .def(py::init<const std::vector<int>&, int, bool, int>())
        .def_property_readonly("param_elems", &VggEngineF32::param_elems)
        .def_property_readonly("bn_elems", &VggEngineF32::bn_elems)
        .def_property_readonly("arena_elems", &VggEngineF32::arena_elems)
        .def_property_readonly("workspace_elems", &VggEngineF32::workspace_elems)
        .def_property_readonly("param_slots", &VggEngineF32::param_slots)
        .def_property_readonly("buffer_slots", &VggEngineF32::buffer_slots)
        .def_property_readonly("num_conv", &VggEngineF32::num_conv)
        .def_property_readonly("layer_table", &VggEngineF32::layer_table)
        .def_property_readonly("bn_save_offsets", &VggEngineF32::bn_save_offsets)
        .def_property("bn_steps", &VggEngineF32::bn_steps, &VggEngineF32::set_bn_steps)
        .def_property_readonly("logits_offset", &VggEngineF32::logits_offset)
        .def_property_readonly("loss_offset", &VggEngineF32::loss_offset)
        .def("bind", [](VggEngineF32& e, uintptr_t p, uintptr_t g, uintptr_t mo, uintptr_t bn,
                        uintptr_t arena, uintptr_t ws) {
            e.bind((float*)p, (float*)g, (float*)mo, (float*)bn, (float*)arena, (float*)ws);
        })
        .def("set_reducer", [](VggEngineF32& e, BucketReducer* r,
                               std::vector<std::pair<int, int>> buckets) {
            std::vector<BucketRange> b;
            for (auto& p : buckets) b.push_back({p.first, p.second});
            e.set_reducer(r, b);
        }, py::keep_alive<1, 2>())
        .def("set_manual_sync", [](VggEngineF32& e, ManualSync* m) { e.set_manual_sync(m); },
             py::keep_alive<1, 2>())
        .def("set_value_hand_offs", &VggEngineF32::set_value_hand_offs)
        .def("set_wgrad_overlap", [](VggEngineF32& e, bool on, uintptr_t compute_stream) {
            e.set_wgrad_overlap(on, (hipStream_t)compute_stream);
        }, py::arg("on"), py::arg("compute_stream") = 0)
        .def_property_readonly("wgrad_overlap", &VggEngineF32::wgrad_overlap)
        .def_property_readonly("hand_off_mode", [](const VggEngineF32& e) { return std::string(e.hand_off_mode()); })
        .def_property_readonly("wgrad_stream", [](VggEngineF32& e) { return (uintptr_t)e.wgrad_stream(); })
        .def("refresh_weights", [](VggEngineF32& e, uintptr_t s) { e.refresh_weights((hipStream_t)s); })
        .def("forward", [](VggEngineF32& e, uintptr_t x, int batch, bool training, uintptr_t s) {
            e.forward((const float*)x, batch, training, (hipStream_t)s);
        })
        .def("forward_backward", [](VggEngineF32& e, uintptr_t x, uintptr_t t, int batch, uintptr_t s) {
            e.forward_backward((const float*)x, (const int64_t*)t, batch, (hipStream_t)s);
        })
        .def("optimizer_step", [](VggEngineF32& e, float lr, float mu, float wd, float gs, uintptr_t s) {
            e.optimizer_step(lr, mu, wd, gs, (hipStream_t)s);
        })
        .def("train_step", [](VggEngineF32& e, uintptr_t x, uintptr_t t, int batch, float lr, float mu,
                              float wd, float gs, uintptr_t s) {
            e.train_step((const float*)x, (const int64_t*)t, batch, lr, mu, wd, gs, (hipStream_t)s);
        })
        .def("eval_step", [](VggEngineF32& e, uintptr_t x, uintptr_t t, int batch, uintptr_t s) {
            e.eval_step((const float*)x, (const int64_t*)t, batch, (hipStream_t)s);
        });
}

}  // namespace dml
