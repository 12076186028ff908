// VggEngine: layout planning and the per-step kernel schedule.
//
// Schedule of one training step (reference order, /root/reference/part1/main.py:38-46):
//   (nothing to zero: every gradient is overwritten)                optimizer.zero_grad()
//   conv (+ BN statistics) -> [bn+relu(+pool)] | [pool] ...         model(data)
//   classifier rows (logits, loss terms, dlogits, dx)               criterion / backward
//   then per conv layer back to front, on the MAIN stream:
//     [bn+relu(+pool) bwd | pool bwd] -> dgrad (+ the next layer's BatchNorm sums)
//   and on the SIDE stream, handed over in groups of layers by one event each:
//     classifier batch reductions, bias grad, wgrad, partial-slab sums,
//     reducer.bucket_ready(...) for every bucket the group completes
//   (the first layer's wgrad, which forms its gradient on the fly, and the last bucket stay on
//   the main stream; without set_wgrad_overlap everything runs on the main stream)
//   join -> reducer.wait_all -> fused SGD -> weight shadow / repack  optimizer.step()
#include "dml_engine.h"

#include <cstring>
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>

#include <cstdlib>
#include <functional>
#include <stdexcept>

#include "dml_kernels.h"
#include "dml_trace.h"

namespace py = pybind11;

namespace dml {

namespace {

constexpr size_t kAlign = 64;     // elements; keeps every slot 128/256-byte aligned

size_t align_up(size_t v) { return (v + kAlign - 1) / kAlign * kAlign; }

void check_hip(hipError_t e, const char* what) {
    if (e != hipSuccess)
        throw std::runtime_error(std::string("HIP ") + what + " failed: " + hipGetErrorString(e));
}

}  // namespace

ParamLayout plan_parameters(const std::vector<int>& widths, bool batch_norm, int num_classes) {
    if (num_classes < 1 || num_classes > 16)
        throw std::invalid_argument("VggEngine: num_classes must be in [1,16]");
    ParamLayout out;
    int cin = 3, hw = 32, module_index = 0;
    size_t p = 0, b = 0;
    for (size_t i = 0; i < widths.size(); ++i) {
        const int w = widths[i];
        if (w < 0) {
            if (out.convs.empty() || out.convs.back().pool_after || hw < 2)
                throw std::invalid_argument("VggEngine: misplaced pool in width table");
            out.convs.back().pool_after = true;
            hw /= 2;
            ++module_index;
            continue;
        }
        if (w % 64) throw std::invalid_argument("VggEngine: channel widths must be multiples of 64");
        if (out.convs.empty() && w != 64)
            throw std::invalid_argument("VggEngine: the first conv must have 64 channels");
        ParamLayout::Conv c;
        c.cin = cin; c.cout = w; c.hw = hw; c.pool_after = false;
        c.gamma_off = c.beta_off = c.rm_off = c.rv_off = 0;
        const int layer = (int)out.convs.size();
        const std::string key = "layers." + std::to_string(module_index);
        c.w_off = p;
        out.slots.push_back({key + ".weight", p, (size_t)w * 9 * cin, {w, 3, 3, cin}, layer});
        p = align_up(p + (size_t)w * 9 * cin);
        c.b_off = p;
        out.slots.push_back({key + ".bias", p, (size_t)w, {w}, layer});
        p = align_up(p + w);
        ++module_index;
        if (batch_norm) {
            const std::string bkey = "layers." + std::to_string(module_index);
            c.gamma_off = p;
            out.slots.push_back({bkey + ".weight", p, (size_t)w, {w}, layer});
            p = align_up(p + w);
            c.beta_off = p;
            out.slots.push_back({bkey + ".bias", p, (size_t)w, {w}, layer});
            p = align_up(p + w);
            c.rm_off = b;
            out.bn_slots.push_back({bkey + ".running_mean", b, (size_t)w, {w}, layer});
            b = align_up(b + w);
            c.rv_off = b;
            out.bn_slots.push_back({bkey + ".running_var", b, (size_t)w, {w}, layer});
            b = align_up(b + w);
            ++module_index;
        }
        ++module_index;   // the ReLU
        out.convs.push_back(c);
        cin = w;
    }
    if (out.convs.empty()) throw std::invalid_argument("VggEngine: empty width table");
    out.fc_in = cin * hw * hw;
    if (hw != 1) throw std::invalid_argument("VggEngine: the feature map must be 1x1 at the classifier");
    const int fc_layer = (int)out.convs.size();
    out.fcw_off = p;
    out.slots.push_back({"fc1.weight", p, (size_t)num_classes * out.fc_in, {num_classes, out.fc_in},
                         fc_layer});
    p = align_up(p + (size_t)num_classes * out.fc_in);
    out.fcb_off = p;
    out.slots.push_back({"fc1.bias", p, (size_t)num_classes, {num_classes}, fc_layer});
    p = align_up(p + num_classes);
    out.param_elems = p;
    out.bn_elems = b;
    return out;
}

VggEngine::VggEngine(const std::vector<int>& widths, int max_batch, bool batch_norm,
                     int num_classes)
    : max_batch_(max_batch), num_classes_(num_classes), bn_(batch_norm) {
    if (max_batch <= 0) throw std::invalid_argument("VggEngine: batch must be positive");
    if (const char* e = getenv("DML_BN_EPI_SUMS")) plain_epi_sums_ = atoi(e) != 0;      // tuning
    if (const char* e = getenv("DML_WGRAD_TUNED")) tuned_wgrad_splits_ = atoi(e) != 0;  // tuning
    // Timing experiment, never a training mode: the weight-gradient kernels (and slab sums) of the
    // conv layers whose bit is set (bit i = layer i, 1 = all) are not launched and their gradients
    // stay stale, which shows what the main chain takes without them.
    if (const char* e = getenv("DML_EXPERIMENT_NO_WGRAD")) experiment_no_wgrad_ = atoi(e);
    if (const char* e = getenv("DML_DEBUG_TAP")) debug_tap_layer_ = atoi(e);
    const ParamLayout lay = plan_parameters(widths, batch_norm, num_classes);
    for (const ParamLayout::Conv& l : lay.convs) {
        Conv c;
        c.cin = l.cin; c.cout = l.cout; c.hw = l.hw; c.pool_after = l.pool_after;
        c.w_off = l.w_off; c.b_off = l.b_off; c.gamma_off = l.gamma_off; c.beta_off = l.beta_off;
        c.rm_off = l.rm_off; c.rv_off = l.rv_off; c.save_off = 0;
        convs_.push_back(c);
    }
    slots_ = lay.slots;
    bn_slots_ = lay.bn_slots;
    fc_in_ = lay.fc_in;
    fcw_off_ = lay.fcw_off;
    fcb_off_ = lay.fcb_off;
    param_elems_ = lay.param_elems;
    bn_elems_ = lay.bn_elems;

    // ---- bf16 arena
    size_t a = 0;
    shadow_off_ = a; a = align_up(a + param_elems_);
    w1_off_ = a;     a = align_up(a + 64 * 64);
    size_t max_act = 0;
    const size_t B = (size_t)max_batch_;
    for (size_t i = 0; i < convs_.size(); ++i) {
        Conv& c = convs_[i];
        const size_t act = B * c.hw * c.hw * c.cout;
        max_act = act > max_act ? act : max_act;
        c.wd_off = a;
        if (i > 0) a = align_up(a + (size_t)c.cout * 9 * c.cin);
        c.pre_off = a;
        if (bn_) a = align_up(a + act);
        c.out_off = a;
        // under BatchNorm a pooled layer never stores its full-resolution activation
        if (!(bn_ && c.pool_after)) a = align_up(a + act);
        c.pool_off = a;
        if (c.pool_after) a = align_up(a + act / 4);
    }
    // Gradients: one buffer per layer for dL/d(conv output) (read by that layer's dgrad and,
    // possibly much later on the side stream, by its wgrad), plus one transit buffer for the
    // gradient arriving at a layer that still has pool / BatchNorm / ReLU backward to apply.
    for (Conv& c : convs_) {
        c.gout_off = a;
        a = align_up(a + B * c.hw * c.hw * c.cout);
    }
    gin_off_ = a; a = align_up(a + max_act);
    arena_elems_ = a;

    // ---- fp32 workspace
    size_t w = 0;
    logits_off_ = w;  w = align_up(w + B * 16);
    dlogits_off_ = w; w = align_up(w + B * 16);
    loss_off_ = w;    w = align_up(w + 64);
    stats_off_ = w;   w = align_up(w + bn_workspace_elems());   // column-reduction scratch
    for (auto& c : convs_) {
        c.save_off = w;
        if (bn_) w = align_up(w + 2 * (size_t)c.cout);
    }
    slab_elems_ = 0;
    for (size_t i = 1; i < convs_.size(); ++i) {
        const Conv& c = convs_[i];
        ConvShape f = {max_batch_, c.hw, c.cin, c.cout};
        ConvShape d = {max_batch_, c.hw, c.cout, c.cin};
        const size_t need_f = conv3x3_igemm_workspace_elems(f);
        const size_t need_d = conv3x3_igemm_workspace_elems(d);
        slab_elems_ = need_f > slab_elems_ ? need_f : slab_elems_;
        slab_elems_ = need_d > slab_elems_ ? need_d : slab_elems_;
    }
    {
        const size_t need_c1 = conv1_wgrad_workspace_elems(max_batch_);
        slab_elems_ = need_c1 > slab_elems_ ? need_c1 : slab_elems_;
    }
    slab_off_ = w; w = align_up(w + slab_elems_);
    fc_ws_off_ = w; w = align_up(w + fc_train_workspace_elems(max_batch_, fc_in_, num_classes_));
    // BatchNorm statistics partials written by the conv epilogues: rows x [2][cout]
    cstats_off_ = w;
    if (bn_) {
        size_t need = (size_t)conv1_fprop_stats_rows(max_batch_) * 2 * convs_[0].cout;
        for (size_t i = 1; i < convs_.size(); ++i) {
            const Conv& c = convs_[i];
            // (one row per 64 pixels is the most any plan writes, forced ones included)
            const size_t n = ((size_t)max_batch_ * c.hw * c.hw + 63) / 64 * 2 * c.cout;
            need = n > need ? n : need;
            // backward: the dgrad of this layer may leave the previous layer's BatchNorm sums,
            // one row per pixel tile of at least 64 rows
            const size_t nbwd = ((size_t)max_batch_ * c.hw * c.hw + 63) / 64 * 2 * c.cin;
            need = nbwd > need ? nbwd : need;
        }
        w = align_up(w + need);
    }
    // wgrad partial results, one slab per split (summed by a batched kernel: deterministic,
    // and plain stores run ~4x faster than float atomics)
    wgrad_force_splits_.assign(convs_.size(), 0);
    if (const char* e = getenv("DML_WGRAD_SPLITS")) {      // tuning: "s1,s2,..." for conv layers 1..
        size_t i = 1;
        for (const char* q = e; *q && i < convs_.size(); ++i) {
            wgrad_force_splits_[i] = atoi(q);
            while (*q && *q != ',') ++q;
            if (*q == ',') ++q;
        }
    }
    wgrad_resident_.assign(convs_.size(), std::array<int, 2>{0, 0});
    if (const char* e = getenv("DML_WGRAD_RESIDENT")) {     // tuning: "bco:stages,..." for layers 1..
        size_t i = 1;
        for (const char* q = e; *q && i < convs_.size(); ++i) {
            wgrad_resident_[i][0] = atoi(q);
            while (*q && *q != ':' && *q != ',') ++q;
            if (*q == ':') { ++q; wgrad_resident_[i][1] = atoi(q); }
            while (*q && *q != ',') ++q;
            if (*q == ',') ++q;
        }
    }
    dgrad_force_plan_.assign(convs_.size(), std::array<int, 5>{0, 0, 0, -1, 0});
    if (const char* e = getenv("DML_DGRAD_PLANS")) {   // tuning: "bm:bn:splits:pm:waves,..." for layers 1..
        size_t i = 1;
        for (const char* q = e; *q && i < convs_.size(); ++i) {
            int k = 0;
            while (*q && *q != ',') {
                if (k < 5) dgrad_force_plan_[i][k++] = atoi(q);
                while (*q && *q != ':' && *q != ',') ++q;
                if (*q == ':') ++q;
            }
            if (*q == ',') ++q;
        }
    }
    fwd_force_plan_.assign(convs_.size(), std::array<int, 5>{0, 0, 0, -1, 0});
    if (const char* e = getenv("DML_FWD_PLANS")) {     // tuning: same format, forward convolutions
        size_t i = 1;
        for (const char* q = e; *q && i < convs_.size(); ++i) {
            int k = 0;
            while (*q && *q != ',') {
                if (k < 5) fwd_force_plan_[i][k++] = atoi(q);
                while (*q && *q != ':' && *q != ',') ++q;
                if (*q == ':') ++q;
            }
            if (*q == ',') ++q;
        }
    }
    if (const char* e = getenv("DML_FUSED_FINISH_BN")) fused_finish_bn_ = atoi(e) != 0;
    // (with the tall patch-resident tiles every backward-data pass of the flagship shapes owns its
    // pixel tiles already; the override below belongs to the streaming plan policy and is looked up
    // per call, since the policy can change between steps)
    streaming_dgrad_plan_.assign(convs_.size(), std::array<int, 5>{0, 0, 0, -1, 0});
    if (!getenv("DML_DGRAD_PLANS") && bn_ && max_batch_ == 256) {
        // Tuned inside the step (tools/tune_dgrad_plans.py, tools/ab_dgrad_l4.sh): the backward-data
        // pass of a 256 -> 512 layer on 4x4 maps runs 2.5 us per step faster on the 64 x 64
        // patch-resident kernel than on the default tap-split plan - a block that owns its pixel
        // tile leaves the previous layer's BatchNorm backward sums in its epilogue, which saves that
        // layer's stand-alone reduction pass.
        for (size_t i = 1; i < convs_.size(); ++i)
            if (convs_[i].hw == 4 && convs_[i].cin == 256 && convs_[i].cout == 512)
                streaming_dgrad_plan_[i] = {64, 64, 1, 0, 16};
    }
    for (size_t i = 0; i < convs_.size(); ++i) {
        Conv& c = convs_[i];
        c.wslab_off = w;
        if (i == 0) continue;
        // (room for either block target, see wgrad_split_target())
        const ConvShape ws_shape = {max_batch_, c.hw, c.cin, c.cout};
        int splits = conv3x3_wgrad_max_splits(ws_shape);
        if (wgrad_force_splits_[i] > splits) splits = wgrad_force_splits_[i];
        if (wgrad_resident_[i][0] > 0) {
            const int r = conv3x3_wgrad9_splits(ws_shape, wgrad_resident_[i][0], wgrad_force_splits_[i]);
            splits = r > splits ? r : splits;
        }
        const int tuned = conv3x3_wgrad_tuned_splits({256, c.hw, c.cin, c.cout}, bn_);
        if (max_batch_ >= 256 && tuned > splits) splits = tuned;
        if (splits > 1) w = align_up(w + (size_t)splits * c.cout * 9 * c.cin);
    }
    ws_elems_ = w;
    buckets_by_layer_.assign(convs_.size() + 1, {});
}

VggEngine::~VggEngine() {
    if (graph_exec_) (void)hipGraphExecDestroy(graph_exec_);
    if (graph_) (void)hipGraphDestroy(graph_);
    for (hipEvent_t e : hand_off_) (void)hipEventDestroy(e);
    if (side_done_) (void)hipEventDestroy(side_done_);
    if (grads_final_) (void)hipEventDestroy(grads_final_);
    if (late_done_) (void)hipEventDestroy(late_done_);
    if (wgrad_stream_) (void)hipStreamDestroy(wgrad_stream_);
    if (hand_off_flags_) (void)hipFree(hand_off_flags_);
}

void VggEngine::set_wgrad_overlap(bool on, int group, hipStream_t compute_stream) {
    if (group < 1) throw std::invalid_argument("VggEngine: wgrad group must be >= 1");
    if (on && !wgrad_stream_) {
        // on a hardware queue shared with the compute (or communication) stream the weight
        // gradients would run in line with it: probe for a queue of its own
        std::vector<hipStream_t> avoid = {compute_stream};
        if (reducer_ && reducer_->comm_stream()) avoid.push_back(reducer_->comm_stream());
        wgrad_stream_ = create_concurrent_stream(avoid);
        hand_off_.resize(convs_.size());
        // These events order two queues of ONE GPU, and every kernel releases its writes to the
        // whole device as it ends, so the events need no system-scope fence of their own
        // (-1.6 % step time; DML_EVENT_SCOPE=system brings the default fence back).
        unsigned flags = hipEventDisableTiming | hipEventDisableSystemFence;
        if (const char* e = getenv("DML_EVENT_SCOPE"))
            if (!strcmp(e, "system")) flags = hipEventDisableTiming;
        // Hand-offs main -> side through a flag word and a wait-value packet where the device
        // supports it (DML_HANDOFF=event: the event pair instead).
        int can_wait_value = 0, device = 0;
        check_hip(hipGetDevice(&device), "hipGetDevice");
        (void)hipDeviceGetAttribute(&can_wait_value, hipDeviceAttributeCanUseStreamWaitValue, device);
        const char* mode = getenv("DML_HANDOFF");
        value_hand_off_ = can_wait_value && !(mode && !strcmp(mode, "event"));
        join_by_value_ = !(getenv("DML_JOIN") && !strcmp(getenv("DML_JOIN"), "event"));
        if (value_hand_off_) {
            // one word per layer + one for the side stream's "done with this backward pass"
            const size_t bytes = (convs_.size() + 1) * sizeof(unsigned);
            check_hip(hipMalloc((void**)&hand_off_flags_, bytes), "hipMalloc(hand-off flags)");
            check_hip(hipMemset(hand_off_flags_, 0, bytes), "hipMemset(hand-off flags)");
            check_hip(hipDeviceSynchronize(), "hipDeviceSynchronize(hand-off flags)");
        }
        for (hipEvent_t& e : hand_off_)
            check_hip(hipEventCreateWithFlags(&e, flags), "hipEventCreate");
        check_hip(hipEventCreateWithFlags(&side_done_, flags), "hipEventCreate");
        check_hip(hipEventCreateWithFlags(&grads_final_, flags), "hipEventCreate");
        check_hip(hipEventCreateWithFlags(&late_done_, flags), "hipEventCreate");
    }
    if ((on != wgrad_overlap_ || group != wgrad_group_) && graph_exec_) {
        (void)hipGraphExecDestroy(graph_exec_);     // a captured step has the old schedule
        graph_exec_ = nullptr;
    }
    wgrad_overlap_ = on;
    wgrad_group_ = group;
}

void VggEngine::set_update_overlap(int first_late_layer) {
    if (first_late_layer < 0 || first_late_layer > (int)convs_.size())
        throw std::invalid_argument("VggEngine: update overlap layer outside the network");
    if (late_pending_) throw std::logic_error("VggEngine: join_update() before changing the overlap");
    late_layer_ = first_late_layer;
}

void VggEngine::join_update(hipStream_t stream) {
    if (!late_pending_) return;
    check_hip(hipStreamWaitEvent(stream, late_done_, 0), "hipStreamWaitEvent(late update)");
    late_pending_ = false;
}

void VggEngine::set_fused_optimizer(bool on, float lr, float momentum, float weight_decay,
                                    float grad_scale) {
    // per-layer slices from the first 4x4 layer on (see backward)
    opt_slice_from_ = (int)convs_.size();
    for (size_t i = 0; i < convs_.size(); ++i)
        if (convs_[i].hw <= 4) { opt_slice_from_ = (int)i; break; }
    if (const char* e = getenv("DML_OPT_SLICE_FROM")) opt_slice_from_ = atoi(e);   // tuning
    opt_.on = on; opt_.lr = lr; opt_.mu = momentum; opt_.wd = weight_decay; opt_.gscale = grad_scale;
}

size_t VggEngine::layer_begin(int layer) const {
    for (const SlotInfo& s : slots_)
        if (s.layer >= layer) return s.offset;
    return param_elems_;
}

void VggEngine::sgd_range(size_t begin, size_t n, float lr, float momentum, float weight_decay,
                          float grad_scale, hipStream_t stream) {
    // bf16 on the wire with the results left there: the update reads them in place
    if (reducer_ && reducer_->keeps_on_wire())
        sgd_momentum_bf16_grads(params_ + begin, reducer_->wire() + begin, mom_ + begin,
                                arena_ + shadow_off_ + begin, n, lr, momentum, weight_decay,
                                grad_scale, false, stream);
    else
        sgd_momentum(params_ + begin, grads_ + begin, mom_ + begin, arena_ + shadow_off_ + begin, n,
                     lr, momentum, weight_decay, grad_scale, false, stream);
}

void VggEngine::sgd_slice(size_t begin, size_t end, hipStream_t stream) {
    if (end <= begin) return;
    sgd_range(begin, end - begin, opt_.lr, opt_.mu, opt_.wd, opt_.gscale, stream);
}

void VggEngine::join_wgrad_stream(hipStream_t stream) {
    if (!side_busy_) return;
    // side_done_ was recorded right behind the side stream's last launch (see backward), so
    // that it has usually fired by now and this wait costs the main stream nothing; recording
    // it here would send a fresh marker through the other queue (~10 us) on the critical path
    if (side_done_by_value_) {
        // (a queue waiting for another queue's EVENT costs ~6 us; for a word in memory, ~3)
        check_hip(hipStreamWaitValue32(stream, hand_off_flags_ + convs_.size(), backward_seq_,
                                       hipStreamWaitValueGte, 0xffffffffu),
                  "hipStreamWaitValue32(side done)");
    } else {
        if (!side_done_recorded_)
            check_hip(hipEventRecord(side_done_, wgrad_stream_), "hipEventRecord(side done)");
        check_hip(hipStreamWaitEvent(stream, side_done_, 0), "hipStreamWaitEvent(side done)");
    }
    side_done_by_value_ = false;
    side_busy_ = false;
    side_done_recorded_ = false;
}

// Weight (and, without BatchNorm, bias) gradient of one conv layer after the first.
void VggEngine::launch_wgrad(int i, int batch, hipStream_t stream) {
    const Conv& c = convs_[i];
    const Conv& prev = convs_[i - 1];
    const uint16_t* in = arena_ + (prev.pool_after ? prev.pool_off : prev.out_off);
    const uint16_t* g = arena_ + c.gout_off;
    const ConvShape s = {batch, c.hw, c.cin, c.cout};
    // Conv bias gradient.  Under BatchNorm it is exactly zero in exact arithmetic (the
    // normalisation removes any per-channel offset; a framework's value there is rounding
    // noise), so it is left at the zero the buffer was allocated with.  Without BatchNorm
    // nothing else in a backward pass touches the reduction scratch.
    if (!bn_) column_sum(g, grads_ + c.b_off, batch * c.hw * c.hw, c.cout, ws_ + stats_off_, stream);
    if (experiment_no_wgrad_ == 1 || ((experiment_no_wgrad_ >> i) & 1)) return;
    const int target = wgrad_split_target();
    // pixel splits: the tuning override, else the list tuned inside the step for this shape, else
    // the kernel's plan for the block target
    int forced = wgrad_force_splits_[i];
    if (wgrad_resident_[i][0] > 0) {        // tuning: the tap-resident kernel for this layer
        const int bco = wgrad_resident_[i][0];
        const int n = conv3x3_wgrad9_splits(s, bco, forced, wgrad_split_target());
        conv3x3_wgrad9(in, g, grads_ + c.w_off, s, stream, ws_ + c.wslab_off, bco, n,
                       wgrad_resident_[i][1] == 3 ? 3 : 2);
        if (n > 1) {
            if (wgrad_sums_.n == SlabSumJobs::kMax) flush_wgrad_sums(stream);
            const int j = wgrad_sums_.n++;
            wgrad_sums_.src_off[j] = c.wslab_off;
            wgrad_sums_.dst_off[j] = c.w_off;
            wgrad_sums_.count[j] = (size_t)c.cout * 9 * c.cin;
            wgrad_sums_.splits[j] = n;
        }
        return;
    }
    if (forced == 0 && tuned_wgrad_splits_) forced = conv3x3_wgrad_tuned_splits(s, bn_);
    conv3x3_wgrad(in, g, grads_ + c.w_off, s, stream, forced, ws_ + c.wslab_off, target);
    const int splits = forced > 0 ? conv3x3_wgrad_forced_splits(s, forced)
                                  : conv3x3_wgrad_splits(s, target);
    if (splits > 1) {       // partial results per split, summed by flush_wgrad_sums
        if (wgrad_sums_.n == SlabSumJobs::kMax) flush_wgrad_sums(stream);
        const int j = wgrad_sums_.n++;
        wgrad_sums_.src_off[j] = c.wslab_off;
        wgrad_sums_.dst_off[j] = c.w_off;
        wgrad_sums_.count[j] = (size_t)c.cout * 9 * c.cin;
        wgrad_sums_.splits[j] = splits;
    }
}

// Side-stream mode: everything the main stream had produced when hand_off_[layer] was recorded
// (the queued layers' gradients, the BatchNorm / classifier gradients of a departing bucket)
// becomes visible to the side stream, which then runs the queued weight gradients and their sums.
void VggEngine::hand_off_wgrads(int batch, bool last) {
    // (the caller has ordered the side stream behind the hand-off event)
    TraceRange range("backward/wgrad-side");
    for (int i : wgrad_queue_) launch_wgrad(i, batch, wgrad_stream_);
    wgrad_queue_.clear();
    // The slab sums follow their weight gradients at once when a bucket may be waiting for them,
    // and under BatchNorm (measured faster); without either, one batched launch at the end of
    // the side stream's work interferes least with the main chain (0.574 -> 0.567 ms).  (An
    // optimizer slice that follows this hand-off reads the summed gradients too.)
    if (reducer_ || bn_ || last || opt_.on) flush_wgrad_sums(wgrad_stream_);
    side_busy_ = true;
}

void VggEngine::bind(float* params, float* grads, float* momentum, float* bn_buffers,
                     uint16_t* arena, float* workspace) {
    if (!params || !grads || !momentum || !arena || !workspace || (bn_ && !bn_buffers))
        throw std::invalid_argument("VggEngine::bind: null buffer");
    params_ = params; grads_ = grads; mom_ = momentum; bnbuf_ = bn_buffers;
    arena_ = arena; ws_ = workspace;
}

void VggEngine::set_reducer(BucketReducer* reducer, const std::vector<BucketRange>& buckets) {
    reducer_ = reducer;
    buckets_by_layer_.assign(convs_.size() + 1, {});
    bucket_ranges_.clear();
    if (!reducer) return;
    bucket_ranges_ = buckets;
    if ((int)buckets.size() != reducer->num_buckets())
        throw std::invalid_argument("VggEngine::set_reducer: bucket count mismatch");
    for (size_t b = 0; b < buckets.size(); ++b) {
        const int lo = buckets[b].lo;
        if (lo < 0 || lo >= (int)slots_.size())
            throw std::invalid_argument("VggEngine::set_reducer: bucket range outside the slots");
        // a bucket is complete once the layer owning its first (earliest) slot has run backward
        buckets_by_layer_[slots_[lo].layer].push_back((int)b);
    }
}

void VggEngine::fire_buckets(int layer, hipStream_t stream) {
    if (!reducer_ || buckets_by_layer_[layer].empty()) return;
    TraceRange range("backward/buckets");
    flush_wgrad_sums(stream);       // a bucket leaves with complete gradients
    for (int b : buckets_by_layer_[layer]) reducer_->bucket_ready(b, stream);
}

void VggEngine::flush_wgrad_sums(hipStream_t stream) {
    if (wgrad_sums_.n == 0) return;
    sum_slabs_batch(ws_, grads_, wgrad_sums_, stream);
    wgrad_sums_.n = 0;
}

void VggEngine::refresh_weights(hipStream_t stream) {
    if (!params_) throw std::logic_error("VggEngine: bind() first");
    join_update(stream);
    // Full refresh: needed when the master weights were written from outside (initialisation,
    // load_state_dict); during training the optimizer kernel keeps the shadow current.
    cast_f32_to_bf16(params_, arena_ + shadow_off_, param_elems_, stream);
    repack_all(stream);
}

std::vector<long long> VggEngine::bn_save_offsets() const {
    std::vector<long long> out;
    for (const Conv& c : convs_) out.push_back(bn_ ? (long long)c.save_off : -1);
    return out;
}

void VggEngine::debug_tap(const void* src, size_t bytes, hipStream_t stream) {
    constexpr size_t kRoom = 16u << 20;
    if (!debug_tap_) check_hip(hipMalloc((void**)&debug_tap_, kRoom), "hipMalloc(debug tap)");
    if (debug_tap_bytes_ + bytes > kRoom) throw std::logic_error("VggEngine: debug tap overflow");
    check_hip(hipMemcpyAsync(debug_tap_ + debug_tap_bytes_, src, bytes, hipMemcpyDeviceToDevice, stream),
              "hipMemcpyAsync(debug tap)");
    debug_tap_bytes_ += bytes;
    debug_tap_sections_.push_back(bytes);
}

std::string VggEngine::debug_tap_bytes() const {
    std::string out(debug_tap_bytes_, '\0');
    if (debug_tap_) {
        check_hip(hipDeviceSynchronize(), "hipDeviceSynchronize(debug tap)");
        check_hip(hipMemcpy(&out[0], debug_tap_, debug_tap_bytes_, hipMemcpyDeviceToHost), "hipMemcpy(debug tap)");
    }
    return out;
}

std::vector<std::vector<long long>> VggEngine::layer_table() const {
    std::vector<std::vector<long long>> out;
    for (const Conv& c : convs_)
        out.push_back({c.cin, c.cout, c.hw, c.pool_after ? 1 : 0, (long long)c.pre_off,
                       (long long)c.out_off, (long long)c.pool_off, (long long)c.gout_off,
                       (long long)c.wd_off});
    return out;
}

void VggEngine::repack_all(hipStream_t stream, const float* copy_src, float* copy_dst,
                           size_t copy_n) {
    repack_layers(1, (int)convs_.size(), true, stream, copy_src, copy_dst, copy_n);
}

void VggEngine::repack_layers(int lo, int hi, bool conv1, hipStream_t stream, const float* copy_src,
                              float* copy_dst, size_t copy_n) {
    // backward-data weight layout for conv layers [lo, hi), 16 layers per launch; the first
    // layer's padded bf16 layout and the copy ride in the first launch
    const float* c1 = conv1 ? params_ + convs_[0].w_off : nullptr;
    if (lo >= hi) {
        if (c1) repack_conv1_weights(c1, arena_ + w1_off_, stream);
        if (copy_n)
            check_hip(hipMemcpyAsync(copy_dst, copy_src, copy_n * sizeof(float),
                                     hipMemcpyDeviceToDevice, stream), "hipMemcpyAsync(buffers)");
        return;
    }
    RepackJobs jobs;
    bool first = true;
    for (int i = lo; i < hi; ++i) {
        const Conv& c = convs_[i];
        jobs.cout[jobs.n] = c.cout;
        jobs.cin[jobs.n] = c.cin;
        jobs.src_off[jobs.n] = shadow_off_ + c.w_off;
        jobs.dst_off[jobs.n] = c.wd_off;
        if (++jobs.n == RepackJobs::kMax || i + 1 == hi) {
            repack_dgrad_weights_batch(arena_, arena_, jobs, stream, first ? c1 : nullptr,
                                       arena_ + w1_off_, first ? copy_src : nullptr,
                                       first ? copy_dst : nullptr, first ? copy_n : 0);
            first = false;
            jobs.n = 0;
        }
    }
}

void VggEngine::forward(const uint16_t* x, int batch, bool training, hipStream_t stream) {
    if (!params_) throw std::logic_error("VggEngine: bind() first");
    if (batch <= 0 || batch > max_batch_) throw std::invalid_argument("VggEngine: bad batch size");
    TraceRange range(training ? "forward" : "forward(eval)");
    const uint16_t* in = x;
    float* slab = ws_ + slab_off_;
    for (size_t i = 0; i < convs_.size(); ++i) {
        const Conv& c = convs_[i];
        // the deep layers' weights may still be on their way (see set_update_overlap)
        if (late_pending_ && (int)i >= late_layer_) join_update(stream);
        uint16_t* dst = arena_ + (bn_ ? c.pre_off : c.out_off);
        // in training the conv epilogue also leaves the BatchNorm statistics partials
        BnPartials part;
        float* cstats = (bn_ && training) ? ws_ + cstats_off_ : nullptr;
        if (i > 0 && cstats && c.hw == 2 && fused_finish_bn_ && batch <= 256 && c.cout % 16 == 0) {
            // 2x2 maps: the tap-split convolution leaves its slabs, and one launch sums them,
            // takes the statistics and applies BatchNorm + ReLU (+ pool), see splitk_finish_bn_relu
            ConvShape s = {batch, c.hw, c.cin, c.cout};
            SplitSlabs pend;
            conv3x3_igemm(in, arena_ + shadow_off_ + c.w_off, dst, nullptr, nullptr, slab, slab_elems_,
                          s, EPI_NONE, stream, nullptr, nullptr, &pend);
            uint16_t* act = arena_ + (c.pool_after ? c.pool_off : c.out_off);
            if (pend.splits > 1 &&
                splitk_finish_bn_relu(pend, params_ + c.b_off, dst, act, params_ + c.gamma_off,
                                      params_ + c.beta_off, bnbuf_ + c.rm_off, bnbuf_ + c.rv_off,
                                      ws_ + c.save_off, ws_ + c.save_off + c.cout, batch, c.hw, c.cout,
                                      c.pool_after, 0.1f, 1e-5f, stream)) {
                in = act;
                continue;
            }
            // (an un-split plan, or a shape the fused kernel does not cover: the usual path)
            conv3x3_igemm(in, arena_ + shadow_off_ + c.w_off, dst, params_ + c.b_off, nullptr, slab,
                          slab_elems_, s, EPI_BIAS, stream, nullptr, cstats);
            part.rows = conv3x3_igemm_stats_rows(s);
        } else if (i == 0) {
            conv1_fprop(in, arena_ + w1_off_, params_ + c.b_off, dst, batch, !bn_, stream, cstats);
            part.rows = conv1_fprop_stats_rows(batch);
        } else {
            ConvShape s = {batch, c.hw, c.cin, c.cout};
            const int* force = fwd_force_plan_[i][0] > 0 ? fwd_force_plan_[i].data() : nullptr;
            conv3x3_igemm(in, arena_ + shadow_off_ + c.w_off, dst, params_ + c.b_off, nullptr, slab,
                          slab_elems_, s, bn_ ? EPI_BIAS : EPI_BIAS_RELU, stream, force, cstats);
            if (cstats) part.rows = conv3x3_igemm_stats_rows(s, force);
        }
        if (cstats) {
            part.data = cstats;
            part.shift = params_ + c.b_off;
        }
        if (bn_ && c.pool_after) {
            // BatchNorm + ReLU + max-pool in one pass; only the pooled tensor is written
            if (training)
                bn_relu_pool_fwd_train(dst, arena_ + c.pool_off, params_ + c.gamma_off,
                                       params_ + c.beta_off, bnbuf_ + c.rm_off, bnbuf_ + c.rv_off,
                                       ws_ + c.save_off, ws_ + c.save_off + c.cout,
                                       ws_ + stats_off_, batch, c.hw, c.cout, 0.1f, 1e-5f, stream,
                                       part);
            else
                bn_relu_pool_fwd_eval(dst, arena_ + c.pool_off, params_ + c.gamma_off,
                                      params_ + c.beta_off, bnbuf_ + c.rm_off, bnbuf_ + c.rv_off,
                                      batch, c.hw, c.cout, 1e-5f, stream);
            in = arena_ + c.pool_off;
            continue;
        }
        if (bn_) {
            const int rows = batch * c.hw * c.hw;
            if (training)
                bn_relu_fwd_train(dst, arena_ + c.out_off, params_ + c.gamma_off,
                                  params_ + c.beta_off, bnbuf_ + c.rm_off, bnbuf_ + c.rv_off,
                                  ws_ + c.save_off, ws_ + c.save_off + c.cout, ws_ + stats_off_,
                                  rows, c.cout, 0.1f, 1e-5f, stream, part);
            else
                bn_relu_fwd_eval(dst, arena_ + c.out_off, params_ + c.gamma_off,
                                 params_ + c.beta_off, bnbuf_ + c.rm_off, bnbuf_ + c.rv_off, rows,
                                 c.cout, 1e-5f, stream);
        }
        in = arena_ + c.out_off;
        if (c.pool_after) {
            maxpool2x2_fwd(in, arena_ + c.pool_off, batch, c.hw, c.cout, stream);
            in = arena_ + c.pool_off;
        }
    }
    join_update(stream);        // (a network whose late part is only the classifier)
    // in training the classifier runs fused with the loss and its backward (see backward())
    if (!training)
        linear_fwd(in, params_ + fcw_off_, params_ + fcb_off_, ws_ + logits_off_, batch, fc_in_,
                   num_classes_, stream);
    if (training && bn_) ++bn_steps_;
}

void VggEngine::backward(const uint16_t* x, const int64_t* target, int batch, hipStream_t stream) {
    TraceRange range("backward");
    const int nconv = (int)convs_.size();
    float* slab = ws_ + slab_off_;
    const Conv& last = convs_[nconv - 1];
    const uint16_t* fc_in = arena_ + (last.pool_after ? last.pool_off : last.out_off);
    const bool side = wgrad_overlap_;
    wgrad_queue_.clear();       // (left-overs of a step that threw)
    wgrad_sums_.n = 0;
    side_done_recorded_ = false;
    side_done_by_value_ = false;
    // (a captured graph keeps the event pair: its edges come from the capture)
    const bool by_value = side && value_hand_off_ && !capturing_;
    if (by_value && ++backward_seq_ == 0) {
        // the 32-bit sequence wrapped (2^32 backward passes): start the words over
        check_hip(hipDeviceSynchronize(), "hipDeviceSynchronize(sequence wrap)");
        check_hip(hipMemset(hand_off_flags_, 0, (convs_.size() + 1) * sizeof(unsigned)), "hipMemset(hand-off flags)");
        backward_seq_ = 1;
    }
    const bool fuse_opt = side && opt_.on;
    opt_applied_ = fuse_opt;
    int updated_above = nconv + 1;      // layers >= this already had their parameters updated
    // behind a bucket's all-reduce, on the stream that ran it: the update of its parameters
    auto bucket_update = [&](int b) -> std::function<void(hipStream_t)> {
        if (!fuse_opt) return nullptr;
        const BucketRange r = bucket_ranges_[b];
        const size_t begin = slots_[r.lo].offset;
        const size_t end = r.hi < (int)slots_.size() ? slots_[r.hi].offset : param_elems_;
        return [this, begin, end](hipStream_t s) { sgd_slice(begin, end, s); };
    };
    uint16_t* transit = arena_ + gin_off_;
    // where the gradient arriving at layer i is written: its own buffer when it already is
    // dL/d(conv output) (no BatchNorm, no pool: the producer applied the ReLU mask), else transit
    auto staged = [&](int i) { return bn_ || convs_[i].pool_after; };
    auto arriving = [&](int i) { return staged(i) ? transit : arena_ + convs_[i].gout_off; };
    auto bucket_leaves = [&](int layer) { return reducer_ && !buckets_by_layer_[layer].empty(); };

    // classifier + loss, forward and backward, in two launches; every output is overwritten.
    // Only the per-row half (dlogits, dx) is on the way to the conv layers; the batch reductions
    // (dW, db, loss) go to the side stream with the first hand-off when there is one.
    fc_train_rows(fc_in, params_ + fcw_off_, params_ + fcb_off_, target, ws_ + logits_off_,
                  arriving(nconv - 1), ws_ + fc_ws_off_, batch, fc_in_, num_classes_, stream);
    auto fc_grads = [this, fc_in, batch](hipStream_t s) {
        fc_train_grads(fc_in, ws_ + fc_ws_off_, ws_ + loss_off_, nullptr, grads_ + fcw_off_,
                       grads_ + fcb_off_, batch, fc_in_, num_classes_, s);
    };
    bool fc_grads_pending = side;
    if (!side) fc_grads(stream);
    bool stage_buffers_pending = side && reducer_ && bn_;
    // side mode: a completed bucket leaves (from the side stream) with the next hand-off
    std::vector<int> owed;          // layers whose buckets are complete but not yet sent off
    if (!side) fire_buckets(nconv, stream);
    else if (bucket_leaves(nconv)) owed.push_back(nconv);
    SplitSlabs pending;     // the arriving gradient is still a split convolution's unsummed slabs
    BnPartials bn_sums;     // the arriving gradient's BatchNorm backward sums came with it
    for (int i = nconv - 1; i >= 0; --i) {
        const Conv& c = convs_[i];
        const int rows = batch * c.hw * c.hw;
        uint16_t* g = arena_ + c.gout_off;       // dL/d(conv output), ReLU mask applied
        // first layer under BatchNorm + pool: its weight-gradient kernel forms g on the fly
        const bool g_on_the_fly = i == 0 && c.pool_after;
        const bool tap_bn = bn_ && debug_tap_layer_ > 0 && i == debug_tap_layer_ - 1;
        if (tap_bn) {
            debug_tap(arena_ + c.pre_off, (size_t)rows * c.cout * 2, stream);
            debug_tap(params_ + c.gamma_off, (size_t)c.cout * 4, stream);
            debug_tap(params_ + c.beta_off, (size_t)c.cout * 4, stream);
            debug_tap(ws_ + c.save_off, (size_t)c.cout * 8, stream);
            debug_tap(transit, (size_t)rows * c.cout * 2 / (c.pool_after ? 4 : 1), stream);
        }
        if (bn_ && c.pool_after)
            bn_relu_pool_bwd(arena_ + c.pre_off, transit, g_on_the_fly ? nullptr : g,
                             params_ + c.gamma_off,
                             params_ + c.beta_off, ws_ + c.save_off, ws_ + c.save_off + c.cout,
                             grads_ + c.gamma_off, grads_ + c.beta_off, ws_ + stats_off_, batch,
                             c.hw, c.cout, stream, pending, bn_sums);
        else if (c.pool_after && !g_on_the_fly)
            maxpool2x2_bwd(arena_ + c.out_off, transit, g, batch, c.hw, c.cout, /*relu_mask=*/true,
                           stream);
        else if (bn_)
            bn_relu_bwd(arena_ + c.pre_off, transit, g, params_ + c.gamma_off, params_ + c.beta_off,
                        ws_ + c.save_off, ws_ + c.save_off + c.cout, grads_ + c.gamma_off,
                        grads_ + c.beta_off, ws_ + stats_off_, rows, c.cout, stream, pending, bn_sums);
        if (tap_bn) {
            debug_tap(g, (size_t)rows * c.cout * 2, stream);
            debug_tap(grads_ + c.gamma_off, (size_t)c.cout * 4, stream);
            debug_tap(grads_ + c.beta_off, (size_t)c.cout * 4, stream);
        }
        if (g_on_the_fly && !bn_) {
            conv1_wgrad_relu_pool(x, arena_ + c.out_off, transit, grads_ + c.w_off,
                                  grads_ + c.b_off, batch, slab, stream);
        } else if (g_on_the_fly) {
            conv1_wgrad_bn_pool(x, arena_ + c.pre_off, transit, params_ + c.gamma_off,
                                params_ + c.beta_off, ws_ + c.save_off, ws_ + c.save_off + c.cout,
                                grads_ + c.gamma_off, grads_ + c.beta_off, grads_ + c.w_off, batch,
                                slab, stream);
        } else if (i == 0) {
            // shares the split-K slab with the input-gradient kernels: always on the main stream
            conv1_wgrad(x, g, grads_ + c.w_off, bn_ ? nullptr : grads_ + c.b_off, batch, slab,
                        stream);
        } else if (side) {
            wgrad_queue_.push_back(i);
        } else {
            launch_wgrad(i, batch, stream);
        }
        // Side mode: one hand-off serves a whole group of layers and every bucket completed
        // since the last one.  The event is recorded before this layer's dgrad (the side stream
        // need not wait for it), but the side stream's launches are issued after it so that
        // the main chain - the critical path - is enqueued first.
        bool hand = false;
        if (side) {
            if (bucket_leaves(i)) owed.push_back(i);
            const bool full = (int)wgrad_queue_.size() >= wgrad_group_ ||
                              (i <= 1 && !wgrad_queue_.empty());
            // (layer 0 finishes on the main stream: its weight gradient runs there, and a last
            // bucket that leaves from the main stream saves a cross-queue hop on the critical path)
            hand = i > 0 && (full || bucket_leaves(i));
            if (hand && !by_value)
                check_hip(hipEventRecord(hand_off_[i], stream), "hipEventRecord(hand-off)");
        }
        // by value: this layer's backward-data kernel, next in the main queue, raises the flag
        const StreamSignal raise = {hand_off_flags_ ? hand_off_flags_ + i : nullptr, backward_seq_};
        if (i > 0) {
            // input gradient; the producer's ReLU is folded in when nothing sits in between
            const Conv& prev = convs_[i - 1];
            ConvShape d = {batch, c.hw, c.cout, c.cin};
            const bool fold_relu = !staged(i - 1);
            // under BatchNorm the consumer is a reduction pass that can sum split slabs itself
            // ... and when that consumer is a pooled layer, the sums themselves can come out of
            // this kernel's epilogue (the partial rows reuse the forward statistics scratch)
            pending = SplitSlabs{};
            bn_sums = BnPartials{};
            // (a consumer small enough for the one-launch BatchNorm backward wants a stored tensor)
            const bool defer_sum = bn_ && !bn_bwd_single_launch(
                batch * c.hw * c.hw, c.cin);
            BnPoolReduce red;
            // (without a pool behind the previous layer the same epilogue leaves the sums of the
            // plain BatchNorm backward reduction; DML_BN_EPI_SUMS=0 keeps that in its own pass)
            const bool want_sums = bn_ && (prev.pool_after || plain_epi_sums_);
            if (want_sums) {
                red.pooled = prev.pool_after;
                red.pre = arena_ + prev.pre_off;
                red.gamma = params_ + prev.gamma_off; red.beta = params_ + prev.beta_off;
                red.mean = ws_ + prev.save_off; red.rstd = ws_ + prev.save_off + prev.cout;
                red.partials = ws_ + cstats_off_;
            }
            conv3x3_igemm(g, arena_ + c.wd_off, arriving(i - 1), nullptr,
                          fold_relu ? arena_ + prev.out_off : nullptr, slab, slab_elems_, d,
                          fold_relu ? EPI_MASK : EPI_NONE, stream,
                          dgrad_force_plan_[i][0] > 0 ? dgrad_force_plan_[i].data()
                          : (!conv_tall_tiles() && batch == 256 && streaming_dgrad_plan_[i][0] > 0)
                              ? streaming_dgrad_plan_[i].data() : nullptr, nullptr,
                          defer_sum ? &pending : nullptr, want_sums ? &red : nullptr,
                          want_sums ? &bn_sums : nullptr, hand && by_value ? &raise : nullptr);
            if (i == debug_tap_layer_) {
                const size_t out_bytes = (size_t)batch * c.hw * c.hw * c.cin * 2;
                debug_tap_sections_.clear();
                debug_tap_bytes_ = 0;
                if (9 * out_bytes * 2 <= slab_elems_ * sizeof(float)) debug_tap(slab, 9 * out_bytes * 2, stream);
                debug_tap(arriving(i - 1), out_bytes, stream);
            }
        }
        if (hand) {
            if (by_value)
                check_hip(hipStreamWaitValue32(wgrad_stream_, hand_off_flags_ + i, backward_seq_,
                                               hipStreamWaitValueGte, 0xffffffffu),
                          "hipStreamWaitValue32(hand-off)");
            else
                check_hip(hipStreamWaitEvent(wgrad_stream_, hand_off_[i], 0), "hipStreamWaitEvent(hand-off)");
            if (fc_grads_pending) {
                fc_grads(wgrad_stream_);
                fc_grads_pending = false;
            }
            if (stage_buffers_pending) {    // forward is over: rank 0's running statistics are final
                reducer_->buffers_updated(wgrad_stream_);
                stage_buffers_pending = false;
            }
            hand_off_wgrads(batch, i <= 1);
            if (!owed.empty()) {
                TraceRange buckets("backward/buckets");
                for (int l : owed)
                    for (int b : buckets_by_layer_[l])
                        reducer_->bucket_ready(b, wgrad_stream_, bucket_update(b));
                owed.clear();
            }
            if (fuse_opt && !reducer_ && (i >= opt_slice_from_ || i <= 1)) {
                // Behind this hand-off's sums the gradients of layers i.. are final.  The deep
                // layers (most of the parameters) are updated at once: early in backward the side
                // stream waits for hand-offs most of the time and the update fills those gaps.
                // From opt_slice_from_ down the side stream is as busy as the main chain, so the
                // remaining (small) layers wait for its last hand-off, where it would idle while
                // the main stream finishes the first layer.
                sgd_slice(layer_begin(i), layer_begin(updated_above), wgrad_stream_);
                updated_above = i;
            }
            if (i <= 1) {                   // the side stream's last work of this step
                if (by_value && join_by_value_) {
                    enqueue_flag_raise(hand_off_flags_ + nconv, backward_seq_, wgrad_stream_);
                    side_done_by_value_ = true;
                } else {
                    check_hip(hipEventRecord(side_done_, wgrad_stream_), "hipEventRecord(side done)");
                }
                side_done_recorded_ = true;
            }
        }
        if (i == 0 && fc_grads_pending) {    // a network with a single conv layer: no hand-off
            fc_grads(stream);
            fc_grads_pending = false;
        }
        if (i == 0 && stage_buffers_pending) {
            reducer_->buffers_updated(stream);
            stage_buffers_pending = false;
        }
        if (side && i == 0 && !owed.empty()) {
            join_wgrad_stream(stream);      // usually fired long ago (recorded at the last hand-off)
            TraceRange buckets("backward/buckets(last)");
            for (int l : owed)
                for (int b : buckets_by_layer_[l])
                    reducer_->bucket_ready(b, stream, bucket_update(b), /*in_line=*/true);
            owed.clear();
        }
        if (!side) fire_buckets(i, stream);
    }
    if (side) {
        // the main stream waits for the side stream (this also rejoins it into a graph capture)
        join_wgrad_stream(stream);
        if (fuse_opt && !reducer_) sgd_slice(0, layer_begin(updated_above), stream);
    } else {
        flush_wgrad_sums(stream);
    }
}

void VggEngine::forward_backward(const uint16_t* x, const int64_t* target, int batch,
                                 hipStream_t stream) {
    if (!params_) throw std::logic_error("VggEngine: bind() first");
    // Nothing to zero: every gradient is overwritten (conv biases under BatchNorm are never
    // written and keep the zero they were allocated with).
    if (reducer_) {
        reducer_->begin_step();
        reducer_->apply_buffers(stream);        // running statistics := rank 0's
    }
    forward(x, batch, true, stream);
    // (side-stream mode: the staging copy of the running statistics leaves the main chain and
    // goes with the first hand-off, see backward)
    if (reducer_ && bn_ && !wgrad_overlap_) reducer_->buffers_updated(stream);
    backward(x, target, batch, stream);
}

void VggEngine::optimizer_step(float lr, float momentum, float weight_decay, float grad_scale,
                               hipStream_t stream) {
    TraceRange range("optimizer");
    if (reducer_) reducer_->wait_all(stream);
    // the running statistics every rank adopts from rank 0 (see BucketReducer) arrived with the
    // gradients: their copy into the live buffers rides in the repack launch
    const float* bsrc = nullptr; float* bdst = nullptr; size_t bn = 0;
    if (reducer_) (void)reducer_->take_pending_buffers(&bsrc, &bdst, &bn);
    const int nconv = (int)convs_.size();
    if (opt_applied_) {
        if (lr != opt_.lr || momentum != opt_.mu || weight_decay != opt_.wd || grad_scale != opt_.gscale)
            throw std::logic_error("VggEngine: optimizer_step() disagrees with the hyper-parameters "
                                   "the fused update already applied in backward");
        opt_applied_ = false;
    } else if (late_layer_ > 0 && late_layer_ <= nconv && wgrad_overlap_ && wgrad_stream_ &&
               !capturing_) {
        // Early layers here, deep layers (and the classifier) on the side stream: the next forward
        // needs the early ones first and reaches the deep ones tens of microseconds later.
        join_update(stream);
        const size_t cut = layer_begin(late_layer_);
        check_hip(hipEventRecord(grads_final_, stream), "hipEventRecord(grads final)");
        sgd_range(0, cut, lr, momentum, weight_decay, grad_scale, stream);
        repack_layers(1, late_layer_ < nconv ? late_layer_ : nconv, true, stream, bsrc, bdst, bn);
        check_hip(hipStreamWaitEvent(wgrad_stream_, grads_final_, 0), "hipStreamWaitEvent(grads final)");
        sgd_range(cut, param_elems_ - cut, lr, momentum, weight_decay, grad_scale, wgrad_stream_);
        repack_layers(late_layer_ < 1 ? 1 : late_layer_, nconv, false, wgrad_stream_);
        check_hip(hipEventRecord(late_done_, wgrad_stream_), "hipEventRecord(late update)");
        late_pending_ = true;
        return;
    } else {
        sgd_range(0, param_elems_, lr, momentum, weight_decay, grad_scale, stream);
    }
    repack_all(stream, bsrc, bdst, bn);
}

void VggEngine::train_step(const uint16_t* x, const int64_t* target, int batch, float lr,
                           float momentum, float weight_decay, float grad_scale,
                           hipStream_t stream) {
    TraceRange range("train_step");
    forward_backward(x, target, batch, stream);
    if (manual_sync_) {
        // the two hand-written modes synchronise between backward and the update
        // (/root/reference/part2/2a/main.py:63, part2/2b/main.py:60)
        TraceRange sync(manual_sync_->mode() == ManualSync::ALL_REDUCE ? "sync/all_reduce"
                                                                        : "sync/gather_scatter");
        manual_sync_->run(stream);
    }
    optimizer_step(lr, momentum, weight_decay, grad_scale, stream);
}

void VggEngine::eval_step(const uint16_t* x, const int64_t* target, int batch, hipStream_t stream) {
    forward(x, batch, false, stream);
    softmax_xent(ws_ + logits_off_, target, eval_loss(), nullptr, eval_correct(), batch,
                 num_classes_, 1.0f, stream);
}

void VggEngine::capture_train_step(const uint16_t* x, const int64_t* target, int batch, float lr,
                                   float momentum, float weight_decay, float grad_scale,
                                   hipStream_t stream) {
    if (graph_exec_) { (void)hipGraphExecDestroy(graph_exec_); graph_exec_ = nullptr; }
    if (graph_) { (void)hipGraphDestroy(graph_); graph_ = nullptr; }
    const long long saved = bn_steps_;
    join_update(stream);
    check_hip(hipStreamBeginCapture(stream, hipStreamCaptureModeThreadLocal), "BeginCapture");
    capturing_ = true;      // (a captured step updates everything inside the graph)
    try {
        forward_backward(x, target, batch, stream);
        optimizer_step(lr, momentum, weight_decay, grad_scale, stream);
    } catch (...) {
        hipGraph_t g = nullptr;
        (void)hipStreamEndCapture(stream, &g);
        if (g) (void)hipGraphDestroy(g);
        bn_steps_ = saved;
        capturing_ = false;
        throw;
    }
    capturing_ = false;
    check_hip(hipStreamEndCapture(stream, &graph_), "EndCapture");
    check_hip(hipGraphInstantiate(&graph_exec_, graph_, nullptr, nullptr, 0), "GraphInstantiate");
    bn_steps_ = saved;      // capturing enqueues nothing
}

void VggEngine::replay_train_step(hipStream_t stream) {
    if (!graph_exec_) throw std::logic_error("VggEngine: capture_train_step() first");
    join_update(stream);
    check_hip(hipGraphLaunch(graph_exec_, stream), "GraphLaunch");
    if (bn_) ++bn_steps_;
}

// ------------------------------------------------------------------------- bindings
void bind_engine(py::module_& m) {
    m.def("set_trace_ranges", &set_trace_ranges);
    m.def("trace_ranges_enabled", &trace_ranges_enabled);
    py::class_<SlotInfo>(m, "SlotInfo")
        .def_readonly("name", &SlotInfo::name)
        .def_readonly("offset", &SlotInfo::offset)
        .def_readonly("numel", &SlotInfo::numel)
        .def_readonly("shape", &SlotInfo::shape)
        .def_readonly("layer", &SlotInfo::layer);

    py::class_<VggEngine>(m, "VggEngine")
        .def(py::init<const std::vector<int>&, int, bool, int>())
        .def_property_readonly("param_elems", &VggEngine::param_elems)
        .def_property_readonly("bn_elems", &VggEngine::bn_elems)
        .def_property_readonly("arena_elems", &VggEngine::arena_elems)
        .def_property_readonly("workspace_elems", &VggEngine::workspace_elems)
        .def_property_readonly("param_slots", &VggEngine::param_slots)
        .def_property_readonly("buffer_slots", &VggEngine::buffer_slots)
        .def_property_readonly("num_conv", &VggEngine::num_conv)
        .def_property_readonly("layer_table", &VggEngine::layer_table)
        .def_property_readonly("bn_save_offsets", &VggEngine::bn_save_offsets)
        .def("debug_tap_bytes", [](const VggEngine& e) { return py::bytes(e.debug_tap_bytes()); })
        .def_property_readonly("debug_tap_sections", &VggEngine::debug_tap_sections)
        .def_property("bn_steps", &VggEngine::bn_steps, &VggEngine::set_bn_steps)
        .def("bind", [](VggEngine& e, uintptr_t p, uintptr_t g, uintptr_t mo, uintptr_t bn,
                        uintptr_t arena, uintptr_t ws) {
            e.bind((float*)p, (float*)g, (float*)mo, (float*)bn, (uint16_t*)arena, (float*)ws);
        })
        .def("set_value_hand_offs", &VggEngine::set_value_hand_offs)
        .def("set_wgrad_overlap", [](VggEngine& e, bool on, int group, uintptr_t compute_stream) {
            e.set_wgrad_overlap(on, group, (hipStream_t)compute_stream);
        }, py::arg("on"), py::arg("group") = 2, py::arg("compute_stream") = 0)
        .def("set_fused_optimizer", &VggEngine::set_fused_optimizer, py::arg("on"),
             py::arg("lr") = 0.f, py::arg("momentum") = 0.f, py::arg("weight_decay") = 0.f,
             py::arg("grad_scale") = 1.f)
        .def("set_update_overlap", &VggEngine::set_update_overlap)
        .def_property_readonly("update_overlap", &VggEngine::update_overlap)
        .def("join_update", [](VggEngine& e, uintptr_t s) { e.join_update((hipStream_t)s); })
        .def_property_readonly("wgrad_overlap", &VggEngine::wgrad_overlap)
        .def_property_readonly("hand_off_mode", [](const VggEngine& e) { return std::string(e.hand_off_mode()); })
        .def_property_readonly("wgrad_stream", [](VggEngine& e) { return (uintptr_t)e.wgrad_stream(); })
        .def("set_reducer", [](VggEngine& e, BucketReducer* r,
                               std::vector<std::pair<int, int>> buckets) {
            std::vector<BucketRange> b;
            for (auto& p : buckets) b.push_back({p.first, p.second});
            e.set_reducer(r, b);
        }, py::keep_alive<1, 2>())
        .def("refresh_weights", [](VggEngine& e, uintptr_t s) { e.refresh_weights((hipStream_t)s); })
        .def("forward", [](VggEngine& e, uintptr_t x, int batch, bool training, uintptr_t s) {
            e.forward((const uint16_t*)x, batch, training, (hipStream_t)s);
        })
        .def("forward_backward", [](VggEngine& e, uintptr_t x, uintptr_t t, int batch, uintptr_t s) {
            e.forward_backward((const uint16_t*)x, (const int64_t*)t, batch, (hipStream_t)s);
        })
        .def("train_step", [](VggEngine& e, uintptr_t x, uintptr_t t, int batch, float lr, float mu,
                              float wd, float gs, uintptr_t s) {
            e.train_step((const uint16_t*)x, (const int64_t*)t, batch, lr, mu, wd, gs,
                         (hipStream_t)s);
        })
        .def("set_manual_sync", [](VggEngine& e, ManualSync* m) { e.set_manual_sync(m); },
             py::keep_alive<1, 2>())
        .def("optimizer_step", [](VggEngine& e, float lr, float mu, float wd, float gs, uintptr_t s) {
            e.optimizer_step(lr, mu, wd, gs, (hipStream_t)s);
        })
        .def("eval_step", [](VggEngine& e, uintptr_t x, uintptr_t t, int batch, uintptr_t s) {
            e.eval_step((const uint16_t*)x, (const int64_t*)t, batch, (hipStream_t)s);
        })
        .def("capture_train_step", [](VggEngine& e, uintptr_t x, uintptr_t t, int batch, float lr,
                                      float mu, float wd, float gs, uintptr_t s) {
            e.capture_train_step((const uint16_t*)x, (const int64_t*)t, batch, lr, mu, wd, gs,
                                 (hipStream_t)s);
        })
        .def("replay_train_step", [](VggEngine& e, uintptr_t s) {
            e.replay_train_step((hipStream_t)s);
        })
        .def_property_readonly("has_graph", &VggEngine::has_graph)
        .def_property_readonly("logits_offset", &VggEngine::logits_offset)
        .def_property_readonly("loss_offset", &VggEngine::loss_offset)
        .def_property_readonly("logits_ptr", [](VggEngine& e) { return (uintptr_t)e.logits(); })
        .def_property_readonly("loss_ptr", [](VggEngine& e) { return (uintptr_t)e.loss(); })
        .def_property_readonly("eval_loss_ptr", [](VggEngine& e) { return (uintptr_t)e.eval_loss(); })
        .def_property_readonly("eval_correct_ptr",
                               [](VggEngine& e) { return (uintptr_t)e.eval_correct(); });
}

}  // namespace dml
