// Host-side launch API of the gfx950 kernels.  Every function validates its shapes on
// the host (throws std::invalid_argument), enqueues on `stream` and returns; nothing
// here allocates or synchronises, so all of it can be captured into a hipGraph.
//
// Kernel inventory (call sites in the reference, /root/reference/part1/model.py and
// /root/reference/part1/main.py):
//   conv3x3 fwd / bwd-data / bwd-weight  <- nn.Conv2d(k=3, s=1, p=1)   model.py:18-23
//   BatchNorm+ReLU fwd / bwd             <- nn.BatchNorm2d + nn.ReLU    part3/model.py:24-25
//   maxpool 2x2 fwd / bwd                <- nn.MaxPool2d(2, 2)          model.py:16
//   linear fwd / bwd                     <- nn.Linear(512, 10)          model.py:40,45
//   softmax cross-entropy fwd+bwd, top-1 <- nn.CrossEntropyLoss, argmax main.py:43,70-72
//   SGD momentum / weight decay          <- optim.SGD(...).step()       main.py:120-121,46
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <vector>

namespace dml {

typedef uint16_t bf16_raw;   // bf16 bit pattern in host-visible signatures

// ---------------------------------------------------------------- conv3x3 (stride 1, pad 1)
enum ConvEpilogue : int {
    EPI_NONE = 0,       // y = acc
    EPI_BIAS = 1,       // y = acc + bias[n]
    EPI_BIAS_RELU = 2,  // y = max(acc + bias[n], 0)
    EPI_MASK = 3,       // y = mask[m][n] > 0 ? acc : 0     (ReLU backward folded into dgrad)
};

struct ConvShape {
    int batch;   // images
    int hw;      // square spatial size, power of two (32 .. 1)
    int cin;     // input channels of the GEMM's reduction (multiple of 64)
    int cout;    // output channels of the GEMM (multiple of 64)
};

// Column partial sums handed from a producer kernel to BatchNorm: `rows` rows of [2][c]
// floats; the mean is shifted by `shift[c]` (the conv bias the partials leave out).
struct BnPartials {
    const float* data = nullptr;
    int rows = 0;
    const float* shift = nullptr;
};

// A hand-off to another queue of the same GPU without a marker packet in the producing queue:
// the kernel that FOLLOWS the producer stores `value` to `flag` (device memory) as its first
// block starts - the queue runs its kernels in order, so everything before it has completed and
// released its writes - and the consuming queue waits with hipStreamWaitValue32(flag >= value).
// Measured on MI355X: an event record costs the producing queue 2-3 us per hand-off and the
// consumer starts 8-10 us after the producer's end; this way the producing queue pays nothing
// and the consumer starts after 3 us (tools/probes/waitvalue_probe.hip).
struct StreamSignal {
    unsigned* flag = nullptr;
    unsigned value = 0;
};

// Partial results of a convolution that split its reduction: `splits` fp32 slabs of the
// output, `stride` elements apart, still to be summed (splits == 0: nothing pending).
struct SplitSlabs {
    const float* data = nullptr;
    int splits = 0;
    size_t stride = 0;
};
// Backward of "conv -> BatchNorm -> ReLU -> 2x2 max-pool" for the layer whose POOLED output
// gradient this convolution (a backward-data pass) produces: with `bn_pool` the kernel's
// epilogue also computes that layer's BatchNorm backward sums (dbeta = sum g, dgamma = sum g*xh
// over the pooling winners) from its fp32 accumulators and the layer's stored conv output, one
// partial row per pixel tile in `partials` ([tile][2][N] floats) - the reduction pass of
// bn_relu_pool_bwd, which would re-read this gradient and the conv output, is then not needed.
// Applied only by plans that keep a pixel tile in one block (reported through *bn_done).
// With `pooled == false` the layer has no pool: `pre` has the shape of this convolution's output
// and every element is its own "winner" (the sums of bn_relu_bwd's reduction pass).
struct BnPoolReduce {
    const bf16_raw* pre = nullptr;      // the layer's conv output [batch][2hw][2hw][N] ([hw][hw] unpooled)
    const float *gamma = nullptr, *beta = nullptr, *mean = nullptr, *rstd = nullptr;
    float* partials = nullptr;
    bool pooled = true;
};

// Implicit GEMM  y[m][n] = sum_{tap,c} x[m (+) tap][c] * w[n][tap][c],  m = (b,h,w) NHWC.
// Forward: x = activations, w = weights [cout][9][cin].  Backward-data: x = dy,
// w = repacked weights [cin][9 flipped][cout] (see repack_dgrad_weights), roles of
// cin/cout swapped by the caller.  `workspace` (fp32, >= splits*M*N elements) is used
// when the launcher decides to split the reduction over taps.
// `defer` (EPI_NONE only): when the launcher splits the reduction, leave the sum of the slabs to
// the consumer (a BatchNorm backward reduction reads the result anyway) and describe them in
// *defer instead of launching the finishing kernel; y is then NOT written.
void conv3x3_igemm(const bf16_raw* x, const bf16_raw* w, bf16_raw* y, const float* bias,
                   const bf16_raw* mask, float* workspace, size_t workspace_elems,
                   ConvShape s, ConvEpilogue epi, hipStream_t stream,
                   const int* force_plan = nullptr /* tuning: {BM, BN, splits, position-major (-1 auto), waves} */,
                   float* stats = nullptr, SplitSlabs* defer = nullptr,
                   const BnPoolReduce* bn_pool = nullptr, BnPartials* bn_done = nullptr,
                   const StreamSignal* signal = nullptr);
size_t conv3x3_igemm_workspace_elems(ConvShape s);
// With `stats` (EPI_BIAS, default plan) the kernel also leaves per-tile column sums and sums
// of squares of the BIAS-FREE output: stats[row][0][n], stats[row][1][n] for
// conv3x3_igemm_stats_rows(s) rows - the BatchNorm statistics without another pass over y
// (see BnPartials).
int conv3x3_igemm_stats_rows(ConvShape s, const int* force_plan = nullptr);
// plan policy of the bf16 convolutions: tall patch-resident tiles where they cover the chip
// (default), or streaming tiles only; switchable between steps
bool conv_tall_tiles();
void set_conv_tall_tiles(bool on);
int conv_tall_max_rows();
void set_conv_tall_max_rows(int rows);       // 256 (default) or 128


// First layer: x is [B,32,32,4] bf16 (3 channels + zero pad), w1 is [64][64] bf16 with
// k = tap*4 + c (k >= 36 zero), y = relu?(conv + bias) as [B,32,32,64] bf16.
void conv1_fprop(const bf16_raw* x, const bf16_raw* w1, const float* bias, bf16_raw* y,
                 int batch, bool relu, hipStream_t stream, float* stats = nullptr);
int conv1_fprop_stats_rows(int batch);     // rows of [2][64] floats written to `stats`
// dW1[64][9][3] = sum_p dy[p][co] * x[p (+) tap][c],  db[64] = sum_p dy[p][co]  (overwritten).
// `ws`: conv1_wgrad_workspace_elems(batch) floats of scratch for the per-block partials.
void conv1_wgrad(const bf16_raw* x, const bf16_raw* dy, float* dw, float* db, int batch,
                 float* ws, hipStream_t stream);
size_t conv1_wgrad_workspace_elems(int batch);
// The same for a first layer followed by BatchNorm + ReLU + 2x2 max-pool, from the gradient of
// the POOLED output: dL/d(conv output) is formed on the fly from the conv output `pre` and the
// BatchNorm constants (dgamma / dbeta from bn_relu_pool_bwd with dx == nullptr) and never stored.
void conv1_wgrad_bn_pool(const bf16_raw* x, const bf16_raw* pre, const bf16_raw* dy_pooled,
                         const float* gamma, const float* beta, const float* save_mean,
                         const float* save_rstd, const float* dgamma, const float* dbeta, float* dw,
                         int batch, float* ws, hipStream_t stream);
// First layer followed by ReLU + 2x2 max-pool (no BatchNorm): `act` is the stored ReLU output.
void conv1_wgrad_relu_pool(const bf16_raw* x, const bf16_raw* act, const bf16_raw* dy_pooled,
                           float* dw, float* db, int batch, float* ws, hipStream_t stream);

// dw[cout][9][cin] (+)= sum_p dy[p][cout] * x[p (+) tap][cin]   (fp32).  The pixel range is
// split over several blocks per tile.  Without `slab` the splits accumulate into `dw` with
// atomics (zero it first).  With `slab` (splits * cout*9*cin floats) every
// split stores its partial result there instead - plain stores, deterministic - and the
// caller sums the conv3x3_wgrad_splits(s) partials (sum_slabs_batch); with a single split `dw`
// is written directly.
void conv3x3_wgrad(const bf16_raw* x, const bf16_raw* dy, float* dw, ConvShape s,
                   hipStream_t stream, int force_splits = 0 /* tuning */, float* slab = nullptr,
                   int split_target = 0 /* blocks to aim for, 0: the kernel's own optimum */);
int conv3x3_wgrad_splits(ConvShape s, int split_target = 0);
int conv3x3_wgrad_forced_splits(ConvShape s, int force_splits);
// pixel splits tuned inside the training step for the shapes of the flagship run (0: none)
int conv3x3_wgrad_tuned_splits(ConvShape s, bool batch_norm);
// the largest split count any kernel choice / block target can ask for at any batch size up to
// s.batch (slab sizing)
int conv3x3_wgrad_max_splits(ConvShape s);
// Kernel choice behind conv3x3_wgrad: 0 / 1 = one tap per block (default: the better neighbour
// of a concurrent backward-data chain), 2 = tap-resident (conv3x3_wgrad9) wherever it applies.
// A tuning / test switch; set it before engines are built.
void set_wgrad_variant(int variant);
// Tuning: blocks every conv3x3_wgrad launch aims for, overriding the caller's split_target
// (0 = no override).  Like the variant, set it before engines are built (slab sizing).
void set_wgrad_block_target(int blocks);
void set_wgrad_waves(int waves);       // per-tap kernel: waves per block, 4 (default) or 8
void set_wgrad_kstep(int pixels);      // per-tap kernel: pixels per K step, 64 (default) or 32
// Tuning: co tile (0 = 128 where it divides cout) and LDS ring depth of the tap-resident kernel
// when conv3x3_wgrad dispatches to it.
void set_wgrad9_config(int bco, int stages);
int wgrad_variant();

// The tap-resident kernel on its own: one block holds a dy tile (64 pixels x bco channels) and a
// zero-bordered x tile in LDS and produces all nine taps of a bco x 64 tile.  `splits` comes from
// conv3x3_wgrad9_splits (pixel-range splits, one slab of cout*9*cin floats each when > 1).
bool conv3x3_wgrad9_supported(ConvShape s);
int conv3x3_wgrad9_bco(ConvShape s, int force_bco = 0);
int conv3x3_wgrad9_splits(ConvShape s, int bco, int force_splits = 0, int block_target = 0);
int conv3x3_wgrad9_splits_bound(ConvShape s, int bco, int block_target = 0);
void conv3x3_wgrad9(const bf16_raw* x, const bf16_raw* dy, float* dw, ConvShape s,
                    hipStream_t stream, float* slab, int bco, int splits, int stages = 3);

// dst[job][i] = sum_k src[job][k * count + i] for up to 32 jobs in one launch (offsets in
// floats from the two bases, counts multiples of 4): the reduction behind the slab mode above.
struct SlabSumJobs {
    static constexpr int kMax = 32;
    int n = 0;
    size_t src_off[kMax], dst_off[kMax], count[kMax];
    int splits[kMax];
    unsigned block_start[kMax + 1];     // filled by the launcher
};
void sum_slabs_batch(const float* src_base, float* dst_base, SlabSumJobs jobs, hipStream_t stream);

// wd[cin][8 - tap][cout] = wf[cout][tap][cin]   (bf16 -> bf16)
void repack_dgrad_weights(const bf16_raw* wf, bf16_raw* wd, int cout, int cin, hipStream_t stream);
// the same for up to 16 layers in one launch; offsets are elements from the two bases
struct RepackJobs {
    static constexpr int kMax = 16;
    int n = 0;
    int cout[kMax], cin[kMax];
    size_t src_off[kMax], dst_off[kMax];
    int tile_start[kMax + 1];     // filled by the launcher
};
void repack_dgrad_weights_batch(const bf16_raw* src_base, bf16_raw* dst_base, RepackJobs jobs,
                                hipStream_t stream, const float* conv1_w = nullptr,
                                bf16_raw* conv1_w1 = nullptr, const float* copy_src = nullptr,
                                float* copy_dst = nullptr, size_t copy_n = 0);
// (copy_*: a small fp32 device copy that rides in the same launch)
// w1[64][64] bf16 (k = tap*4 + c) from master fp32 [64][9][3]
void repack_conv1_weights(const float* w, bf16_raw* w1, hipStream_t stream);

// out[c] = sum_m x[m][c] (overwritten).  `ws`: column_sum_workspace_elems() floats of scratch for the
// per-block partial sums.
void column_sum(const bf16_raw* x, float* out, int rows, int cols, float* ws, hipStream_t stream);
size_t column_sum_workspace_elems();

// ---------------------------------------------------------------- maxpool 2x2 stride 2
void maxpool2x2_fwd(const bf16_raw* x, bf16_raw* y, int batch, int hw, int c, hipStream_t stream);
// dx = dy routed to the first maximum of each window; with relu_mask the route is also
// closed where x <= 0 (x is a post-ReLU tensor).
void maxpool2x2_bwd(const bf16_raw* x, const bf16_raw* dy, bf16_raw* dx, int batch, int hw, int c,
                    bool relu_mask, hipStream_t stream);

// ---------------------------------------------------------------- BatchNorm + ReLU (NHWC)
// Training forward: batch statistics over rows, y = relu(gamma * (x - mean) * rstd + beta),
// running stats updated with `momentum` (unbiased variance), save_mean/save_rstd kept for
// backward.  `stats_ws` is bn_workspace_elems() floats of scratch.  With `from` the statistics
// pass is skipped and the producer's partial sums are folded instead.
void bn_relu_fwd_train(const bf16_raw* x, bf16_raw* y, const float* gamma, const float* beta,
                       float* running_mean, float* running_var, float* save_mean,
                       float* save_rstd, float* stats_ws, int rows, int c, float momentum,
                       float eps, hipStream_t stream, BnPartials from = BnPartials());
void bn_relu_fwd_eval(const bf16_raw* x, bf16_raw* y, const float* gamma, const float* beta,
                      const float* running_mean, const float* running_var, int rows, int c,
                      float eps, hipStream_t stream);
// dx, dgamma, dbeta from dy (grad of the ReLU output), the saved input x and statistics.
// `stats_ws` as above; dgamma/dbeta are overwritten.
size_t bn_workspace_elems();
// With `dy_slabs` (splits > 0) the incoming gradient is still the unsummed partial results of
// the convolution that produced it: the reduction pass sums them, WRITES the bf16 result to
// dy (the apply pass reads it) and works on the rounded values, i.e. exactly what a separate
// finishing kernel followed by this call would compute.
void bn_relu_bwd(const bf16_raw* x, bf16_raw* dy, bf16_raw* dx, const float* gamma,
                 const float* beta, const float* save_mean, const float* save_rstd,
                 float* dgamma, float* dbeta, float* stats_ws, int rows, int c,
                 hipStream_t stream, SplitSlabs dy_slabs = {}, BnPartials sums = {});
// (`sums`: the dbeta / dgamma partial rows were already produced by the convolution that wrote
// dy, see BnPoolReduce - the reduction pass is skipped)

// True when bn_relu_bwd / bn_relu_pool_bwd handle a gradient of `dy_rows` rows x c channels in a
// single launch (small layers) - provided dy is a stored tensor, not unsummed split slabs.
bool bn_bwd_single_launch(int dy_rows, int c);

// BatchNorm + ReLU + 2x2/2 max-pool in one pass each way (x is the conv output
// [batch][hw][hw][c], y / dy the pooled tensor [batch][hw/2][hw/2][c]).  The full-resolution
// activation and its gradient are never materialised: backward recomputes the winning
// window position from x.
// Split-K finish, batch statistics, BatchNorm + ReLU (+ 2x2 max-pool) of a layer on 2x2 maps in one
// launch (a block owns 16 channels and all 4 * batch rows).  `slabs`: the convolution's unsummed
// bias-free fp32 slabs; writes the bf16 conv output `pre` (bias added) and the activation `out`
// ([4 * batch][c], or [batch][c] pooled).  false: shape outside the kernel's range, nothing launched.
bool splitk_finish_bn_relu(const SplitSlabs& slabs, const float* bias, bf16_raw* pre, bf16_raw* out,
                           const float* gamma, const float* beta, float* running_mean,
                           float* running_var, float* save_mean, float* save_rstd, int batch, int hw,
                           int c, bool pool, float momentum, float eps, hipStream_t stream);
void bn_relu_pool_fwd_train(const bf16_raw* x, bf16_raw* y, const float* gamma, const float* beta,
                            float* running_mean, float* running_var, float* save_mean,
                            float* save_rstd, float* ws, int batch, int hw, int c, float momentum,
                            float eps, hipStream_t stream, BnPartials from = BnPartials());
void bn_relu_pool_fwd_eval(const bf16_raw* x, bf16_raw* y, const float* gamma, const float* beta,
                           const float* running_mean, const float* running_var, int batch, int hw,
                           int c, float eps, hipStream_t stream);
void bn_relu_pool_bwd(const bf16_raw* x, bf16_raw* dy, bf16_raw* dx, const float* gamma,
                      const float* beta, const float* save_mean, const float* save_rstd,
                      float* dgamma, float* dbeta, float* ws, int batch, int hw, int c,
                      hipStream_t stream, SplitSlabs dy_slabs = {}, BnPartials sums = {});
// (`sums`: the dbeta / dgamma partial rows were already produced by the convolution that wrote
// dy, see BnPoolReduce - the reduction pass is skipped)
// fp32 activations: the same passes
void bn_relu_pool_fwd_train(const float* x, float* y, const float* gamma, const float* beta,
                            float* running_mean, float* running_var, float* save_mean,
                            float* save_rstd, float* ws, int batch, int hw, int c, float momentum,
                            float eps, hipStream_t stream);
void bn_relu_pool_fwd_eval(const float* x, float* y, const float* gamma, const float* beta,
                           const float* running_mean, const float* running_var, int batch, int hw,
                           int c, float eps, hipStream_t stream);
void bn_relu_pool_bwd(const float* x, float* dy, float* dx, const float* gamma, const float* beta,
                      const float* save_mean, const float* save_rstd, float* dgamma, float* dbeta,
                      float* ws, int batch, int hw, int c, hipStream_t stream);

// ---------------------------------------------------------------- classifier + loss
// logits[b][n] = sum_k x[b][k] * w[n][k] + bias[n]      (x bf16, w/bias/logits fp32, n <= 16)
void linear_fwd(const bf16_raw* x, const float* w, const float* bias, float* logits, int batch,
                int in_features, int out_features, hipStream_t stream);
// dw += dlogits^T x, db += sum dlogits (fp32 atomics: zero them first), dx = dlogits w (bf16).
void linear_bwd(const bf16_raw* x, const float* w, const float* dlogits, float* dw, float* db,
                bf16_raw* dx, int batch, int in_features, int out_features, hipStream_t stream);
// Mean cross-entropy over the batch: loss_sum[0] += sum_b loss_b / batch,
// dlogits = (softmax - onehot) * grad_scale / batch (may be null), correct[0] += #(argmax == target).
void softmax_xent(const float* logits, const int64_t* target, float* loss_sum, float* dlogits,
                  int32_t* correct, int batch, int classes, float grad_scale, hipStream_t stream);

// One training step of the classifier head in two launches: logits = x w^T + b, mean
// cross-entropy `loss[0]`, top-1 count `correct[0]` (may be null), dx = dlogits w,
// dw = dlogits^T x, db = sum dlogits - all overwritten, no atomics (bit-reproducible).
// in_features must be 512, out_features <= 16; `ws` holds fc_train_workspace_elems floats.
void fc_train(const bf16_raw* x, const float* w, const float* bias, const int64_t* target,
              float* logits, float* loss, int32_t* correct, bf16_raw* dx, float* dw, float* db,
              float* ws, int batch, int in_features, int out_features, hipStream_t stream);
size_t fc_train_workspace_elems(int batch, int in_features, int out_features);
// The two halves of fc_train: everything that needs only its own batch row (logits, dlogits
// and per-row loss terms into `ws`, dx), and the reductions over the batch (dW, db, loss,
// accuracy), which nothing in the rest of backward waits for.
void fc_train_rows(const bf16_raw* x, const float* w, const float* bias, const int64_t* target,
                   float* logits, bf16_raw* dx, float* ws, int batch, int in_features,
                   int out_features, hipStream_t stream);
void fc_train_grads(const bf16_raw* x, const float* ws, float* loss, int32_t* correct, float* dw,
                    float* db, int batch, int in_features, int out_features, hipStream_t stream);

// ---------------------------------------------------------------- optimiser
// g' = g * grad_scale + wd * p;  m = first ? g' : mu * m + g';  p -= lr * m.
// Also writes the bf16 shadow copy of p when `shadow` is non-null.
void sgd_momentum(float* p, const float* g, float* m, bf16_raw* shadow, size_t n, float lr,
                  float mu, float wd, float grad_scale, bool first_step, hipStream_t stream);
// the same update from bf16 gradients (a bf16 all-reduce's result, read in place)
void sgd_momentum_bf16_grads(float* p, const bf16_raw* g, float* m, bf16_raw* shadow, size_t n,
                             float lr, float mu, float wd, float grad_scale, bool first_step,
                             hipStream_t stream);

// out[i] = bf16(x[i])
void cast_f32_to_bf16(const float* x, bf16_raw* out, size_t n, hipStream_t stream);
// out[i] = float(x[i])   (n a multiple of 4)
void cast_bf16_to_f32(const bf16_raw* x, float* out, size_t n, hipStream_t stream);

// ---------------------------------------------------------------- data / layout / comm helpers
// uint8 [B,32,32,3] -> bf16 [B,32,32,4], random-crop(pad 4)+flip per aug row (may be null),
// /255, normalise.  aug = int32 [B,3] (row offset, col offset, flip).
void augment_normalize(const uint8_t* images, const int32_t* aug, bf16_raw* out, int batch,
                       const float mean[3], const float stdv[3], hipStream_t stream);
// fp32 NCHW [B,3,32,32] -> bf16 NHWC4
void nchw_to_nhwc4(const float* x, bf16_raw* out, int batch, hipStream_t stream);
// fp32 NCHW <-> bf16 NHWC for generic tensors
void nchw_f32_to_nhwc_bf16(const float* x, bf16_raw* out, int batch, int c, int hw, hipStream_t s);
void nhwc_bf16_to_nchw_f32(const bf16_raw* x, float* out, int batch, int c, int hw, hipStream_t s);
// slabs[0][i] += slabs[1][i] + ... + slabs[w-1][i]   (root-side reduction of gather/scatter)
void sum_slabs(float* slabs, int world, size_t n, hipStream_t stream);
// dst[i] += slabs[0][i] + ... + slabs[k-1][i], slab r at slabs + r * stride, i < n
void sum_slabs_into(float* dst, const float* slabs, int k, size_t stride, size_t n,
                    hipStream_t stream);
// fill a bf16 buffer with uniform values in [-1, 1) from a counter hash (tests / benches)
void fill_uniform_bf16(bf16_raw* out, size_t n, uint32_t seed, float scale, hipStream_t stream);

// ---------------------------------------------------------------- fp32 activations
// The reference trains in fp32 end to end (/root/reference/part1/main.py:15-16: default dtype
// on the CPU).  The same operations on NHWC fp32 tensors, for --dtype fp32: overloads of the
// memory-bound kernels above (instantiations of the same templates), and fp32-input MFMA
// (v_mfma_f32_32x32x2_f32: exact fp32 products and sums) convolutions.
void maxpool2x2_fwd(const float* x, float* y, int batch, int hw, int c, hipStream_t stream);
void maxpool2x2_bwd(const float* x, const float* dy, float* dx, int batch, int hw, int c,
                    bool relu_mask, hipStream_t stream);
void column_sum(const float* x, float* out, int rows, int cols, float* ws, hipStream_t stream);
void bn_relu_fwd_train(const float* x, float* y, const float* gamma, const float* beta,
                       float* running_mean, float* running_var, float* save_mean,
                       float* save_rstd, float* stats_ws, int rows, int c, float momentum,
                       float eps, hipStream_t stream);
void bn_relu_fwd_eval(const float* x, float* y, const float* gamma, const float* beta,
                      const float* running_mean, const float* running_var, int rows, int c,
                      float eps, hipStream_t stream);
void bn_relu_bwd(const float* x, float* dy, float* dx, const float* gamma, const float* beta,
                 const float* save_mean, const float* save_rstd, float* dgamma, float* dbeta,
                 float* stats_ws, int rows, int c, hipStream_t stream);
void linear_fwd(const float* x, const float* w, const float* bias, float* logits, int batch,
                int in_features, int out_features, hipStream_t stream);
void fc_train_rows(const float* x, const float* w, const float* bias, const int64_t* target,
                   float* logits, float* dx, float* ws, int batch, int in_features,
                   int out_features, hipStream_t stream);
void fc_train_grads(const float* x, const float* ws, float* loss, int32_t* correct, float* dw,
                    float* db, int batch, int in_features, int out_features, hipStream_t stream);
void augment_normalize(const uint8_t* images, const int32_t* aug, float* out, int batch,
                       const float mean[3], const float stdv[3], hipStream_t stream);
void nchw_to_nhwc4(const float* x, float* out, int batch, hipStream_t stream);

// Implicit GEMM on fp32 NHWC tensors, same contract as conv3x3_igemm (forward with
// w = [cout][9][cin], backward-data with w = repack_dgrad_weights_f32 and cin/cout swapped).
// `mask` (EPI_MASK) has the output's shape.  No workspace: the reduction is never split.
void conv3x3_igemm_f32(const float* x, const float* w, float* y, const float* bias,
                       const float* mask, ConvShape s, ConvEpilogue epi, hipStream_t stream,
                       int force_tile = 0 /* tuning: 128 or 64 */,
                       int force_pm = -1 /* tuning: position-major tiles off / on */,
                       int force_split = -1 /* tuning: two-way tap split off / on (-1: by grid size) */,
                       bool pre_zeroed = false /* y is already zero: skip the tap split's memset */,
                       const StreamSignal* signal = nullptr /* raised as the kernel starts */);
// 2 when the default plan divides the taps of a position between two blocks that meet in y through
// atomics - y must then be zero before the launch (the launcher clears it unless pre_zeroed)
int conv3x3_igemm_f32_tap_splits(ConvShape s, ConvEpilogue epi);
// several device-to-device copies in one launch (any alignment; 16-byte vectors when it allows)
struct CopyJobs {
    static constexpr int kMax = 24;
    int n = 0;
    void* dst[kMax];
    const void* src[kMax];
    size_t bytes[kMax];
};
void copy_batch(const CopyJobs& jobs, hipStream_t stream);
// several fp32 buffers cleared in one launch (counts in floats, multiples of 4, 16-byte aligned)
struct ZeroJobs {
    static constexpr int kMax = 16;
    int n = 0;
    float* ptr[kMax];
    size_t count[kMax];
};
void zero_buffers(const ZeroJobs& jobs, hipStream_t stream);
// wd[cin][8 - tap][cout] = wf[cout][tap][cin]
void repack_dgrad_weights_f32(const float* wf, float* wd, int cout, int cin, hipStream_t stream);
// dw[cout][9][cin] = sum_p dy[p][cout] * x[p (+) tap][cin].  The pixel range is split over
// conv3x3_wgrad_f32_splits(s) blocks per tile; with more than one split every split stores its
// partial result to `slab` (splits * cout*9*cin floats) and the caller sums them
// (sum_slabs_batch), with one split `dw` is written directly.
void conv3x3_wgrad_f32(const float* x, const float* dy, float* dw, ConvShape s, float* slab,
                       hipStream_t stream, int force_splits = 0, int force_pm = -1);
int conv3x3_wgrad_f32_splits(ConvShape s, int force_splits = 0, int force_pm = -1);
// First layer on fp32: x [B,32,32,4] (3 channels + zero pad), w the master weights
// [64][9][3], y = relu?(conv + bias) [B,32,32,64].
void conv1_fprop_f32(const float* x, const float* w, const float* bias, float* y, int batch,
                     bool relu, hipStream_t stream);
// dW1[64][9][3] and (db != nullptr) db[64] from dy [B,32,32,64]; `ws` holds
// conv1_wgrad_f32_workspace_elems(batch) floats of per-block partials.
void conv1_wgrad_f32(const float* x, const float* dy, float* dw, float* db, int batch, float* ws,
                     hipStream_t stream);
size_t conv1_wgrad_f32_workspace_elems(int batch);
// y = mask > 0 ? x : 0  (ReLU backward where no other kernel can fold it; n % 8 == 0)
void relu_mask_f32(const float* x, const float* mask, float* y, size_t n, hipStream_t stream);
void relu_mask(const bf16_raw* x, const bf16_raw* mask, bf16_raw* y, size_t n, hipStream_t stream);

// ---- streams on hardware queues of their own (stream_probe.hip)
// true if a kernel on `candidate` starts while a kernel on `busy` is still running
bool streams_run_concurrently(hipStream_t busy, hipStream_t candidate);
// a non-blocking stream that runs concurrently with every stream in `avoid` (falls back to
// the last candidate, with a warning, when the hardware-queue pool is exhausted)
hipStream_t create_concurrent_stream(const std::vector<hipStream_t>& avoid, int max_tries = 8);

}  // namespace dml
