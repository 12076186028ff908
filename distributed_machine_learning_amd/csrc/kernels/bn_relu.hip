// Fused BatchNorm2d + ReLU on NHWC bf16 activations (training, evaluation, backward).
//
// Reference ops: nn.BatchNorm2d(c) followed by nn.ReLU(inplace=True), enabled in the
// part3 model (/root/reference/part3/model.py:24-25); defaults momentum 0.1, eps 1e-5.
// Training normalises with the biased batch variance and moves the running variance
// with the unbiased one; evaluation folds the running statistics into a scale/shift.
//
// In NHWC a channel is a column of the [rows = B*H*W][c] matrix, so the statistics are
// column reductions (dml_column_reduce.h): blocks stream row slabs with 16-byte loads (8
// channels per thread) and store partial sums, a small second kernel folds the partials
// and finalises (mean / rstd / running statistics forward, dgamma / dbeta backward) - no
// memset, no result atomics.  The apply kernels then stream the tensor once with
// per-channel constants held in registers.  All statistics stay fp32.
#include <cstdlib>
#include <type_traits>
#include <stdexcept>
#include <string>

#include "dml_column_reduce.h"
#include <vector>

#include "dml_common.h"
#include "dml_kernels.h"

namespace dml {

namespace {

void check(int rows, int c, const char* who) {
    if (c % 8 || c < 8 || c > 2048 || (256 % (c / 8)) != 0 || rows <= 0)
        throw std::invalid_argument(std::string(who) + ": c must be 8*2^k <= 2048 and rows > 0");
}

unsigned apply_blocks(size_t total8) {
    size_t blocks = (total8 + 255) / 256;
    return (unsigned)(blocks > 2048 ? 2048 : blocks);
}

}  // namespace

size_t bn_workspace_elems() { return kColReduceWsFloats; }

__device__ __forceinline__ void load_consts(const float* __restrict__ p, int col, float (&o)[8]) {
    const float4 a = *(const float4*)(p + col), b = *(const float4*)(p + col + 4);
    o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w; o[4] = b.x; o[5] = b.y; o[6] = b.z; o[7] = b.w;
}


// ----------------------------------------------------------------- forward, training
// Sums of (x - s) and (x - s)^2 with s = the column's value in row 0: a sample of the
// distribution, so both sums stay of the order of the spread and var = E[(x-s)^2] - E[x-s]^2
// does not cancel when |mean| >> std (the plain E[x^2] - E[x]^2 loses log2(mean^2/var) bits).
// Every block derives the same s; block 0 leaves it behind the slab for the fold.
template <typename T>
__global__ __launch_bounds__(256) void bn_stats_kernel(const T* __restrict__ x,
                                                       float* __restrict__ ws, int rows, int c,
                                                       int rows_per_block) {
    __shared__ float red[4096];
    float s[8];
    const int cg0 = threadIdx.x % (c / 8);
    load8(x + cg0 * 8, s);
    if (blockIdx.x == 0 && (int)threadIdx.x < c / 8) {
#pragma unroll
        for (int i = 0; i < 8; ++i) ws[kColReduceSlabFloats + cg0 * 8 + i] = s[i];
    }
    column_partials(
        [&](int r, int cg, float (&a)[8], float (&b)[8]) {
            float v[8];
            load8(x + (size_t)r * c + cg * 8, v);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float d = v[i] - s[i];
                a[i] += d;
                b[i] += d * d;
            }
        },
        ws, rows, c, rows_per_block, red);
}

__global__ __launch_bounds__(256) void bn_stats_finish_kernel(
    const float* __restrict__ ws, float* __restrict__ save_mean, float* __restrict__ save_rstd,
    float* __restrict__ running_mean, float* __restrict__ running_var, int nb, int rows, int c,
    float momentum, float eps, const float* __restrict__ shift) {
    const float inv_n = 1.f / (float)rows;
    column_fold(
        [&](int col, float s1, float s2) {
            const float m0 = s1 * inv_n;
            const float var = fmaxf(s2 * inv_n - m0 * m0, 0.f);
            // the sums are of x - shift (a conv epilogue leaves the bias out, the statistics
            // pass subtracts row 0): a shift only moves the mean
            const float mean = shift ? m0 + shift[col] : m0;
            save_mean[col] = mean;
            save_rstd[col] = rsqrtf(var + eps);
            if (running_mean) {
                const float unbiased = rows > 1 ? var * ((float)rows / (float)(rows - 1)) : var;
                running_mean[col] = (1.f - momentum) * running_mean[col] + momentum * mean;
                running_var[col] = (1.f - momentum) * running_var[col] + momentum * unbiased;
            }
        },
        ws, nb, c);
}

namespace {
// The partial statistics rows of x: the ones a producer kernel already left behind, or those of
// a pass over x (into ws, with the shift vector behind the slab).
template <typename T>
BnPartials statistics_partials(const T* x, float* ws, int rows, int c, const BnPartials& from,
                               hipStream_t stream) {
    if (from.data) {
        if (from.rows <= 0) throw std::invalid_argument("BatchNorm: empty partial sums");
        return from;
    }
    const ColReducePlan pl = plan_col_reduce(rows, c);
    hipLaunchKernelGGL(bn_stats_kernel<T>, dim3(pl.blocks), dim3(256), 0, stream, x, ws, rows, c,
                       pl.rows_per_block);
    BnPartials out;
    out.data = ws;
    out.rows = pl.blocks;
    out.shift = ws + kColReduceSlabFloats;
    return out;
}

// mean / rstd (+ running statistics) from the partial rows, as a launch of its own
void finish_statistics(const BnPartials& part, int rows, int c, float* save_mean, float* save_rstd,
                       float* running_mean, float* running_var, float momentum, float eps,
                       hipStream_t stream) {
    hipLaunchKernelGGL(bn_stats_finish_kernel, dim3(fold_blocks(c)), dim3(256), 0, stream, part.data,
                       save_mean, save_rstd, running_mean, running_var, part.rows, rows, c, momentum,
                       eps, part.shift);
}

// fold + apply in one launch when the partial rows are few enough (defined with its kernel below)
template <typename T>
bool fold_apply_plain(const T* x, T* y, const float* gamma, const float* beta, const BnPartials& part,
                      float* save_mean, float* save_rstd, float* running_mean, float* running_var,
                      int rows, int c, float momentum, float eps, hipStream_t stream);
}  // namespace

// y = relu(x * scale + shift) with scale = gamma * rstd, shift = beta - mean * scale.  A
// thread's 8 channels are the same in every grid-stride iteration (the stride is a multiple
// of c/8), so the per-channel constants are loaded once as 16-byte vectors; the streaming
// loop is load - fma - max - store with no branches.
template <bool FROM_VAR, typename T>
__global__ __launch_bounds__(256) void bn_apply_kernel(
    const T* __restrict__ x, T* __restrict__ y, const float* __restrict__ gamma,
    const float* __restrict__ beta, const float* __restrict__ mean,
    const float* __restrict__ rstd_or_var, int c, float eps, size_t total8) {
    const int c8 = c / 8;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const int col = (int)(i % c8) * 8;
    float sc[8], sh[8];
    {
        const float4 g0 = *(const float4*)(gamma + col), g1 = *(const float4*)(gamma + col + 4);
        const float4 b0 = *(const float4*)(beta + col), b1 = *(const float4*)(beta + col + 4);
        const float4 m0 = *(const float4*)(mean + col), m1 = *(const float4*)(mean + col + 4);
        const float4 r0 = *(const float4*)(rstd_or_var + col),
                     r1 = *(const float4*)(rstd_or_var + col + 4);
        const float g[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
        const float bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
        const float m[8] = {m0.x, m0.y, m0.z, m0.w, m1.x, m1.y, m1.z, m1.w};
        const float rr[8] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w};
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float rs = FROM_VAR ? rsqrtf(rr[k] + eps) : rr[k];
            sc[k] = g[k] * rs;
            sh[k] = bb[k] - m[k] * sc[k];
        }
    }
    for (; i < total8; i += stride) {
        float v[8], o[8];
        load8(x + i * 8, v);
#pragma unroll
        for (int k = 0; k < 8; ++k) o[k] = fmaxf(__builtin_fmaf(v[k], sc[k], sh[k]), 0.f);
        store8(y + i * 8, o);
    }
}

namespace {
template <typename T>
void bn_fwd_train_launch(const T* x, T* y, const float* gamma, const float* beta,
                         float* running_mean, float* running_var, float* save_mean,
                         float* save_rstd, float* ws, int rows, int c, float momentum, float eps,
                         hipStream_t stream, BnPartials from) {
    check(rows, c, "bn_relu_fwd_train");
    const BnPartials part = statistics_partials(x, ws, rows, c, from, stream);
    if (fold_apply_plain(x, y, gamma, beta, part, save_mean, save_rstd, running_mean, running_var,
                         rows, c, momentum, eps, stream))
        return;
    finish_statistics(part, rows, c, save_mean, save_rstd, running_mean, running_var, momentum, eps,
                      stream);
    const size_t total8 = (size_t)rows * (c / 8);
    hipLaunchKernelGGL((bn_apply_kernel<false, T>), dim3(apply_blocks(total8)), dim3(256), 0, stream,
                       x, y, gamma, beta, (const float*)save_mean, (const float*)save_rstd, c, eps,
                       total8);
}

template <typename T>
void bn_fwd_eval_launch(const T* x, T* y, const float* gamma, const float* beta,
                        const float* running_mean, const float* running_var, int rows, int c,
                        float eps, hipStream_t stream) {
    check(rows, c, "bn_relu_fwd_eval");
    const size_t total8 = (size_t)rows * (c / 8);
    hipLaunchKernelGGL((bn_apply_kernel<true, T>), dim3(apply_blocks(total8)), dim3(256), 0, stream,
                       x, y, gamma, beta, running_mean, running_var, c, eps, total8);
}
}  // namespace

void bn_relu_fwd_train(const bf16_raw* x, bf16_raw* y, const float* gamma, const float* beta,
                       float* running_mean, float* running_var, float* save_mean,
                       float* save_rstd, float* ws, int rows, int c, float momentum,
                       float eps, hipStream_t stream, BnPartials from) {
    bn_fwd_train_launch(x, y, gamma, beta, running_mean, running_var, save_mean, save_rstd, ws,
                        rows, c, momentum, eps, stream, from);
}
void bn_relu_fwd_train(const float* x, float* y, const float* gamma, const float* beta,
                       float* running_mean, float* running_var, float* save_mean,
                       float* save_rstd, float* ws, int rows, int c, float momentum,
                       float eps, hipStream_t stream) {
    bn_fwd_train_launch(x, y, gamma, beta, running_mean, running_var, save_mean, save_rstd, ws,
                        rows, c, momentum, eps, stream, BnPartials());
}

void bn_relu_fwd_eval(const bf16_raw* x, bf16_raw* y, const float* gamma, const float* beta,
                      const float* running_mean, const float* running_var, int rows, int c,
                      float eps, hipStream_t stream) {
    bn_fwd_eval_launch(x, y, gamma, beta, running_mean, running_var, rows, c, eps, stream);
}
void bn_relu_fwd_eval(const float* x, float* y, const float* gamma, const float* beta,
                      const float* running_mean, const float* running_var, int rows, int c,
                      float eps, hipStream_t stream) {
    bn_fwd_eval_launch(x, y, gamma, beta, running_mean, running_var, rows, c, eps, stream);
}

// The incoming gradient of row r, column group cg: either the stored bf16 tensor, or the sum
// of a convolution's split slabs - rounded to bf16, stored for the apply pass, and used rounded.
template <bool SLABS>
__device__ __forceinline__ void load_dy8(float* __restrict__ dy, SplitSlabs, size_t off,
                                         float (&g)[8]) {
    static_assert(!SLABS, "split slabs feed the bf16 path only");
    load8(dy + off, g);
}

template <bool SLABS>
__device__ __forceinline__ void load_dy8(uint16_t* __restrict__ dy, SplitSlabs sl, size_t off,
                                         float (&g)[8]) {
    if (SLABS) {
        float4 lo = *(const float4*)(sl.data + off), hi = *(const float4*)(sl.data + off + 4);
        for (int s = 1; s < sl.splits; ++s) {
            const float4 l2 = *(const float4*)(sl.data + s * sl.stride + off);
            const float4 h2 = *(const float4*)(sl.data + s * sl.stride + off + 4);
            lo.x += l2.x; lo.y += l2.y; lo.z += l2.z; lo.w += l2.w;
            hi.x += h2.x; hi.y += h2.y; hi.z += h2.z; hi.w += h2.w;
        }
        const float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
        const uint4 pk = pack8(v);
        *(uint4*)(dy + off) = pk;
        unpack8(pk, g);
    } else {
        unpack8(*(const uint4*)(dy + off), g);
    }
}

// -------------------------------------------------------------------------- backward
// With xh = (x - mean) * rstd, y = gamma*xh + beta and g = dy * [y > 0]:
//   dbeta = sum g,  dgamma = sum g*xh,  dx = gamma*rstd * (g - dbeta/n - xh*dgamma/n).
template <bool SLABS, typename T>
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(
    const T* __restrict__ x, T* __restrict__ dy,
    const float* __restrict__ gamma, const float* __restrict__ beta,
    const float* __restrict__ mean, const float* __restrict__ rstd, float* __restrict__ ws,
    int rows, int c, int rows_per_block, SplitSlabs sl) {
    __shared__ float red[4096];
    // this thread's 8 channels are fixed: y > 0  <=>  x * ysc + ysh > 0, xh = x * rs - mrs
    float rs[8], mrs[8], ysc[8], ysh[8];
    {
        const int col0 = (threadIdx.x % (c / 8)) * 8;
        float gm[8], bt[8], mn[8];
        load_consts(rstd, col0, rs); load_consts(mean, col0, mn);
        load_consts(gamma, col0, gm); load_consts(beta, col0, bt);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            mrs[i] = mn[i] * rs[i];
            // exactly forward's scale / shift, so the mask is the ReLU forward applied
            ysc[i] = gm[i] * rs[i];
            ysh[i] = bt[i] - mn[i] * ysc[i];
        }
    }
    column_partials(
        [&](int r, int cg, float (&a)[8], float (&b)[8]) {
            float v[8], g[8];
            load8(x + (size_t)r * c + cg * 8, v);
            load_dy8<SLABS>(dy, sl, (size_t)r * c + cg * 8, g);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float gi = (__builtin_fmaf(v[i], ysc[i], ysh[i]) > 0.f) ? g[i] : 0.f;
                a[i] += gi;
                b[i] += gi * (v[i] * rs[i] - mrs[i]);
            }
        },
        ws, rows, c, rows_per_block, red);
}

__global__ __launch_bounds__(256) void bn_bwd_finish_kernel(const float* __restrict__ ws,
                                                            float* __restrict__ dgamma,
                                                            float* __restrict__ dbeta, int nb,
                                                            int c) {
    column_fold(
        [&](int col, float s1, float s2) {
            dbeta[col] = s1;
            dgamma[col] = s2;
        },
        ws, nb, c);
}

template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(
    const T* __restrict__ x, const T* __restrict__ dy, T* __restrict__ dx,
    const float* __restrict__ gamma, const float* __restrict__ beta,
    const float* __restrict__ mean, const float* __restrict__ rstd,
    const float* __restrict__ dgamma, const float* __restrict__ dbeta, int rows, int c,
    size_t total8) {
    const float inv_n = 1.f / (float)rows;
    const int c8 = c / 8;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    // per-thread channel constants (fixed across the grid-stride loop):
    //   dx = k1 * g - k2 - k3 * x   with xh = x*rs - mrs folded in
    float rs[8], ysc[8], ysh[8], k1[8], k2[8], k3[8];
    {
        const int col0 = (int)(i % c8) * 8;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int col = col0 + k;
            rs[k] = rstd[col];
            const float mrs = mean[col] * rs[k];
            ysc[k] = gamma[col] * rs[k];
            ysh[k] = beta[col] - mean[col] * ysc[k];      // forward's shift, bit for bit
            k1[k] = gamma[col] * rs[k];
            const float dg = dgamma[col] * inv_n, db = dbeta[col] * inv_n;
            // k1 * (g - db - (x*rs - mrs) * dg) = k1*g - k1*(db - mrs*dg) - (k1*rs*dg) * x
            k2[k] = k1[k] * (db - mrs * dg);
            k3[k] = k1[k] * rs[k] * dg;
        }
    }
    for (; i < total8; i += stride) {
        float v[8], g[8], o[8];
        load8(x + i * 8, v);
        load8(dy + i * 8, g);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float gi = (__builtin_fmaf(v[k], ysc[k], ysh[k]) > 0.f) ? g[k] : 0.f;
            o[k] = k1[k] * gi - k2[k] - k3[k] * v[k];
        }
        store8(dx + i * 8, o);
    }
}

// (small layers: one launch, defined with its kernel further down)
bool bn_relu_bwd_small(const bf16_raw* x, bf16_raw* dy, bf16_raw* dx, const float* gamma,
                       const float* beta, const float* save_mean, const float* save_rstd,
                       float* dgamma, float* dbeta, int rows, int c, hipStream_t stream,
                       SplitSlabs sl);

// (fold + apply in one launch when the reduction left few enough partial rows; defined with its
// kernel further down)
template <typename T>
bool bn_bwd_fold_apply_plain(const T* x, const T* dy, T* dx, const float* gamma,
                             const float* beta, const float* save_mean, const float* save_rstd,
                             const float* partials, int nb, float* dgamma, float* dbeta, int rows,
                             int c, hipStream_t stream);

void bn_relu_bwd(const bf16_raw* x, bf16_raw* dy, bf16_raw* dx, const float* gamma,
                 const float* beta, const float* save_mean, const float* save_rstd,
                 float* dgamma, float* dbeta, float* ws, int rows, int c, hipStream_t stream,
                 SplitSlabs sl, BnPartials sums) {
    check(rows, c, "bn_relu_bwd");
    if (sums.data) {            // the producer of dy already left the sums, one partial row per tile
        if (sl.splits > 0 || sums.rows <= 0)
            throw std::invalid_argument("bn_relu_bwd: inconsistent partial sums");
        if (bn_bwd_fold_apply_plain<uint16_t>(x, dy, dx, gamma, beta, save_mean, save_rstd, sums.data,
                                              sums.rows, dgamma, dbeta, rows, c, stream))
            return;
        hipLaunchKernelGGL(bn_bwd_finish_kernel, dim3(fold_blocks(c)), dim3(256), 0, stream,
                           (const float*)sums.data, dgamma, dbeta, sums.rows, c);
        const size_t total8 = (size_t)rows * (c / 8);
        hipLaunchKernelGGL(bn_bwd_apply_kernel<uint16_t>, dim3(apply_blocks(total8)), dim3(256), 0,
                           stream, x, (const uint16_t*)dy, dx, gamma, beta, save_mean, save_rstd,
                           (const float*)dgamma, (const float*)dbeta, rows, c, total8);
        return;
    }
    if (bn_relu_bwd_small(x, dy, dx, gamma, beta, save_mean, save_rstd, dgamma, dbeta, rows, c,
                          stream, sl))
        return;
    const ColReducePlan pl = plan_col_reduce(rows, c);
    if (sl.splits > 0)
        hipLaunchKernelGGL((bn_bwd_reduce_kernel<true, uint16_t>), dim3(pl.blocks), dim3(256), 0,
                           stream, x, dy, gamma, beta, save_mean, save_rstd, ws, rows, c,
                           pl.rows_per_block, sl);
    else
        hipLaunchKernelGGL((bn_bwd_reduce_kernel<false, uint16_t>), dim3(pl.blocks), dim3(256), 0,
                           stream, x, dy, gamma, beta, save_mean, save_rstd, ws, rows, c,
                           pl.rows_per_block, sl);
    if (bn_bwd_fold_apply_plain<uint16_t>(x, dy, dx, gamma, beta, save_mean, save_rstd, ws, pl.blocks,
                                          dgamma, dbeta, rows, c, stream))
        return;
    hipLaunchKernelGGL(bn_bwd_finish_kernel, dim3(fold_blocks(c)), dim3(256), 0, stream,
                       (const float*)ws, dgamma, dbeta, pl.blocks, c);
    const size_t total8 = (size_t)rows * (c / 8);
    hipLaunchKernelGGL(bn_bwd_apply_kernel<uint16_t>, dim3(apply_blocks(total8)), dim3(256), 0,
                       stream, x, (const uint16_t*)dy, dx, gamma, beta, save_mean, save_rstd,
                       (const float*)dgamma, (const float*)dbeta, rows, c, total8);
}

// fp32 activations: the same launches (no split slabs, no one-launch variant)
void bn_relu_bwd(const float* x, float* dy, float* dx, const float* gamma, const float* beta,
                 const float* save_mean, const float* save_rstd, float* dgamma, float* dbeta,
                 float* ws, int rows, int c, hipStream_t stream) {
    check(rows, c, "bn_relu_bwd");
    const ColReducePlan pl = plan_col_reduce(rows, c);
    hipLaunchKernelGGL((bn_bwd_reduce_kernel<false, float>), dim3(pl.blocks), dim3(256), 0, stream,
                       x, dy, gamma, beta, save_mean, save_rstd, ws, rows, c, pl.rows_per_block,
                       SplitSlabs());
    if (bn_bwd_fold_apply_plain<float>(x, dy, dx, gamma, beta, save_mean, save_rstd, ws, pl.blocks,
                                       dgamma, dbeta, rows, c, stream))
        return;
    hipLaunchKernelGGL(bn_bwd_finish_kernel, dim3(fold_blocks(c)), dim3(256), 0, stream,
                       (const float*)ws, dgamma, dbeta, pl.blocks, c);
    const size_t total8 = (size_t)rows * (c / 8);
    hipLaunchKernelGGL(bn_bwd_apply_kernel<float>, dim3(apply_blocks(total8)), dim3(256), 0, stream,
                       x, (const float*)dy, dx, gamma, beta, save_mean, save_rstd,
                       (const float*)dgamma, (const float*)dbeta, rows, c, total8);
}


// ------------------------------------------------ BatchNorm + ReLU + 2x2 max-pool fused
// When a 2x2/2 max-pool follows (nn.MaxPool2d(kernel_size=2, stride=2),
// /root/reference/part1/model.py:16) the full-resolution activation is never stored:
// forward reads the four window positions of the conv output, applies scale/shift/ReLU and
// writes only the pooled maximum; backward re-derives the winning position (first maximum
// in row-major window order, strict comparisons, the same fma as forward) from the conv
// output, so neither the activation nor its full-resolution gradient ever touch memory.
struct PoolGeom {
    int c8, log_oh, hw;      // c / 8, log2(output side), input side
    __device__ __forceinline__ size_t in_pixel(size_t opix) const {
        const int oh = 1 << log_oh;
        const size_t ow_ = opix & (oh - 1);
        const size_t oh_ = (opix >> log_oh) & (oh - 1);
        const size_t b = opix >> (2 * log_oh);
        return (b * hw + 2 * oh_) * hw + 2 * ow_;
    }
};

__device__ __forceinline__ void load_window(const uint16_t* __restrict__ x, size_t ipix, int cg,
                                            int c, int hw, float (&v)[4][8]) {
    const uint16_t* p = x + ipix * c + cg * 8;
    const uint4 q0 = *(const uint4*)(p);
    const uint4 q1 = *(const uint4*)(p + c);
    const uint4 q2 = *(const uint4*)(p + (size_t)hw * c);
    const uint4 q3 = *(const uint4*)(p + (size_t)hw * c + c);
    unpack8(q0, v[0]); unpack8(q1, v[1]); unpack8(q2, v[2]); unpack8(q3, v[3]);
}
__device__ __forceinline__ void load_window(const float* __restrict__ x, size_t ipix, int cg,
                                            int c, int hw, float (&v)[4][8]) {
    const float* p = x + ipix * c + cg * 8;
    load8(p, v[0]); load8(p + c, v[1]);
    load8(p + (size_t)hw * c, v[2]); load8(p + (size_t)hw * c + c, v[3]);
}
// the four positions of a window written back (backward)
template <typename T>
__device__ __forceinline__ void store_window(T* __restrict__ dx, size_t ipix, int col, int c, int hw,
                                             const float (&o)[4][8]) {
    T* p = dx + ipix * c + col;
    store8(p, o[0]); store8(p + c, o[1]);
    store8(p + (size_t)hw * c, o[2]); store8(p + (size_t)hw * c + c, o[3]);
}


template <bool FROM_VAR, typename T>
__global__ __launch_bounds__(256) void bn_apply_pool_kernel(
    const T* __restrict__ x, T* __restrict__ y, const float* __restrict__ gamma,
    const float* __restrict__ beta, const float* __restrict__ mean,
    const float* __restrict__ rstd_or_var, int c, float eps, PoolGeom geo, size_t total8) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const int cg = (int)(i % geo.c8);
    float sc[8], sh[8];
    {
        float g[8], bb[8], m[8], rr[8];
        load_consts(gamma, cg * 8, g); load_consts(beta, cg * 8, bb);
        load_consts(mean, cg * 8, m); load_consts(rstd_or_var, cg * 8, rr);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float rs = FROM_VAR ? rsqrtf(rr[k] + eps) : rr[k];
            sc[k] = g[k] * rs;
            sh[k] = bb[k] - m[k] * sc[k];
        }
    }
    for (; i < total8; i += stride) {
        float v[4][8], o[8];
        load_window(x, geo.in_pixel(i / geo.c8), cg, c, geo.hw, v);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float z0 = __builtin_fmaf(v[0][k], sc[k], sh[k]);
            const float z1 = __builtin_fmaf(v[1][k], sc[k], sh[k]);
            const float z2 = __builtin_fmaf(v[2][k], sc[k], sh[k]);
            const float z3 = __builtin_fmaf(v[3][k], sc[k], sh[k]);
            o[k] = fmaxf(fmaxf(fmaxf(z0, z1), fmaxf(z2, z3)), 0.f);
        }
        store8(y + i * 8, o);
    }
}

// Winner of the window for one channel: index of the first maximum of z, and whether the
// pooled activation was positive (otherwise ReLU blocks the gradient).
__device__ __forceinline__ int window_winner(float z0, float z1, float z2, float z3, bool& live) {
    int w = 0;
    float m = z0;
    if (z1 > m) { m = z1; w = 1; }
    if (z2 > m) { m = z2; w = 2; }
    if (z3 > m) { m = z3; w = 3; }
    live = m > 0.f;
    return w;
}

template <bool SLABS, typename T>
__global__ __launch_bounds__(256) void bn_pool_bwd_reduce_kernel(
    const T* __restrict__ x, T* __restrict__ dy,
    const float* __restrict__ gamma, const float* __restrict__ beta,
    const float* __restrict__ mean, const float* __restrict__ rstd, float* __restrict__ ws,
    int out_rows, int c, int rows_per_block, PoolGeom geo, SplitSlabs sl) {
    __shared__ float red[4096];
    float rs[8], mrs[8], ysc[8], ysh[8];
    {
        const int col0 = (threadIdx.x % (c / 8)) * 8;
        float gm[8], bt[8], mn[8];
        load_consts(rstd, col0, rs); load_consts(mean, col0, mn);
        load_consts(gamma, col0, gm); load_consts(beta, col0, bt);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            mrs[i] = mn[i] * rs[i];
            ysc[i] = gm[i] * rs[i];
            ysh[i] = bt[i] - gm[i] * mrs[i];
        }
    }
    column_partials(
        [&](int r, int cg, float (&a)[8], float (&b)[8]) {
            float v[4][8], g[8];
            load_window(x, geo.in_pixel((size_t)r), cg, c, geo.hw, v);
            load_dy8<SLABS>(dy, sl, (size_t)r * c + cg * 8, g);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                bool live;
                const int w = window_winner(__builtin_fmaf(v[0][i], ysc[i], ysh[i]),
                                            __builtin_fmaf(v[1][i], ysc[i], ysh[i]),
                                            __builtin_fmaf(v[2][i], ysc[i], ysh[i]),
                                            __builtin_fmaf(v[3][i], ysc[i], ysh[i]), live);
                const float xv = w == 0 ? v[0][i] : w == 1 ? v[1][i] : w == 2 ? v[2][i] : v[3][i];
                const float gi = live ? g[i] : 0.f;
                a[i] += gi;
                b[i] += gi * (xv * rs[i] - mrs[i]);
            }
        },
        ws, out_rows, c, rows_per_block, red);
}

template <typename T>
__global__ __launch_bounds__(256) void bn_pool_bwd_apply_kernel(
    const T* __restrict__ x, const T* __restrict__ dy, T* __restrict__ dx,
    const float* __restrict__ gamma, const float* __restrict__ beta,
    const float* __restrict__ mean, const float* __restrict__ rstd,
    const float* __restrict__ dgamma, const float* __restrict__ dbeta, int in_rows, int c,
    PoolGeom geo, size_t total8) {
    const float inv_n = 1.f / (float)in_rows;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const int cg = (int)(i % geo.c8);
    float ysc[8], ysh[8], k1[8], k2[8], k3[8];
    {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int col = cg * 8 + k;
            const float rs = rstd[col];
            const float mrs = mean[col] * rs;
            ysc[k] = gamma[col] * rs;
            ysh[k] = __builtin_fmaf(-gamma[col], mrs, beta[col]);
            k1[k] = gamma[col] * rs;
            const float dg = dgamma[col] * inv_n, db = dbeta[col] * inv_n;
            k2[k] = k1[k] * __builtin_fmaf(-mrs, dg, db);
            k3[k] = (k1[k] * rs) * dg;
        }
    }
    for (; i < total8; i += stride) {
        float v[4][8], g[8], o[4][8];
        const size_t ipix = geo.in_pixel(i / geo.c8);
        load_window(x, ipix, cg, c, geo.hw, v);
        load8(dy + i * 8, g);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            bool live;
            const int w = window_winner(__builtin_fmaf(v[0][k], ysc[k], ysh[k]),
                                        __builtin_fmaf(v[1][k], ysc[k], ysh[k]),
                                        __builtin_fmaf(v[2][k], ysc[k], ysh[k]),
                                        __builtin_fmaf(v[3][k], ysc[k], ysh[k]), live);
            const float gk = live ? k1[k] * g[k] : 0.f;
#pragma unroll
            for (int q = 0; q < 4; ++q)
                o[q][k] = __builtin_fmaf(-k3[k], v[q][k], (w == q ? gk : 0.f) - k2[k]);
        }
        store_window(dx, ipix, cg * 8, c, geo.hw, o);
    }
}

// ------------------------------------------- small layers: backward in ONE launch
// The three-kernel scheme (partial sums, fold, apply) exists because a column has more rows
// than one block can stream.  On the last layers (<= 256 gradient rows: a fraction of a MB)
// the data is nothing and the three dependent launches are everything, so a block
// takes 16 channels and ALL rows instead: wave = 8 channels x 64 row lanes, sums by DPP and a
// 4-entry LDS exchange - the block then HAS dgamma / dbeta - and applies straight away (the
// second read of x / dy comes out of cache).  Fixed summation order: bit-reproducible.
template <bool POOL>
__global__ __launch_bounds__(256) void bn_bwd_slab_kernel(
    const uint16_t* __restrict__ x, const uint16_t* __restrict__ dy, uint16_t* __restrict__ dx,
    const float* __restrict__ gamma, const float* __restrict__ beta,
    const float* __restrict__ mean, const float* __restrict__ rstd, float* __restrict__ dgamma,
    float* __restrict__ dbeta, int rows, int c, float inv_n, PoolGeom geo) {
    constexpr int R = 2;                // rows per lane: rows <= 256, everything stays in registers
    constexpr int W = POOL ? 4 : 1;     // conv-output positions per gradient row
    __shared__ float xch[4][16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int rl = threadIdx.x & 127;                       // row lane 0..127
    const int cg = blockIdx.x * 2 + (wave >> 1);            // 8-channel group
    const int col0 = cg * 8;
    // every global load of the kernel is issued here, before anything waits
    uint4 gq[R], vq[R][W];
    size_t xoff[R];
#pragma unroll
    for (int k = 0; k < R; ++k) {
        const int r = rl + 128 * k;
        const int rc = r < rows ? r : 0;
        gq[k] = *(const uint4*)(dy + (size_t)rc * c + col0);
        xoff[k] = (POOL ? geo.in_pixel((size_t)rc) : (size_t)rc) * c + col0;
        vq[k][0] = *(const uint4*)(x + xoff[k]);
        if (POOL) {
            vq[k][1] = *(const uint4*)(x + xoff[k] + c);
            vq[k][2] = *(const uint4*)(x + xoff[k] + (size_t)geo.hw * c);
            vq[k][3] = *(const uint4*)(x + xoff[k] + (size_t)geo.hw * c + c);
        }
    }
    float rs[8], mrs[8], ysc[8], ysh[8], gm[8];
    {
        float bt[8], mn[8];
        load_consts(rstd, col0, rs); load_consts(mean, col0, mn);
        load_consts(gamma, col0, gm); load_consts(beta, col0, bt);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            mrs[i] = mn[i] * rs[i];
            ysc[i] = gm[i] * rs[i];
            ysh[i] = __builtin_fmaf(-gm[i], mrs[i], bt[i]);
        }
    }
    float g[R][8], v[R][W][8];
    int win[R][8];                      // winning window position, -1: nothing passes
    float a[8], b[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = b[i] = 0.f;
#pragma unroll
    for (int k = 0; k < R; ++k) {
        const bool valid = rl + 128 * k < rows;
        unpack8(gq[k], g[k]);
#pragma unroll
        for (int q = 0; q < W; ++q) unpack8(vq[k][q], v[k][q]);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            int w = 0;
            float m = __builtin_fmaf(v[k][0][i], ysc[i], ysh[i]), xv = v[k][0][i];
#pragma unroll
            for (int q = 1; q < W; ++q) {
                const float z = __builtin_fmaf(v[k][q][i], ysc[i], ysh[i]);
                if (z > m) { m = z; w = q; xv = v[k][q][i]; }
            }
            win[k][i] = (valid && m > 0.f) ? w : -1;
            const float gi = win[k][i] >= 0 ? g[k][i] : 0.f;
            a[i] += gi;
            b[i] += gi * (xv * rs[i] - mrs[i]);
        }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        a[i] = wave_sum_dpp(a[i]);
        b[i] = wave_sum_dpp(b[i]);
    }
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            xch[wave][i] = a[i];
            xch[wave][8 + i] = b[i];
        }
    }
    __syncthreads();
    const int w0 = wave & 2;            // the two waves of this channel group
    float k1[8], k2[8], k3[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const float db = xch[w0][i] + xch[w0 + 1][i];
        const float dg = xch[w0][8 + i] + xch[w0 + 1][8 + i];
        if (threadIdx.x == w0 * 64) {
            dbeta[col0 + i] = db;
            dgamma[col0 + i] = dg;
        }
        k1[i] = gm[i] * rs[i];
        const float dgn = dg * inv_n, dbn = db * inv_n;
        k2[i] = k1[i] * __builtin_fmaf(-mrs[i], dgn, dbn);
        k3[i] = (k1[i] * rs[i]) * dgn;
    }
#pragma unroll
    for (int k = 0; k < R; ++k) {
        if (rl + 128 * k >= rows) continue;
        float o[W][8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float gk = win[k][i] >= 0 ? k1[i] * g[k][i] : 0.f;
#pragma unroll
            for (int q = 0; q < W; ++q)
                o[q][i] = __builtin_fmaf(-k3[i], v[k][q][i], (win[k][i] == q ? gk : 0.f) - k2[i]);
        }
        uint16_t* p = dx + xoff[k];
        *(uint4*)(p) = pack8(o[0]);
        if (POOL) {
            *(uint4*)(p + c) = pack8(o[1]);
            *(uint4*)(p + (size_t)geo.hw * c) = pack8(o[2]);
            *(uint4*)(p + (size_t)geo.hw * c + c) = pack8(o[3]);
        }
    }
}

// rows of the incoming gradient up to which the one-launch kernel is used (measured at 512
// channels: a win at 256 rows, a loss at 1024 - 32 blocks do not stream 2-4 MB fast enough)
constexpr int kSlabKernelMaxRows = 256;
inline bool use_slab_kernel(int dy_rows, int c) { return dy_rows <= kSlabKernelMaxRows && c % 16 == 0 && c >= 256; }

bool bn_relu_bwd_small(const bf16_raw* x, bf16_raw* dy, bf16_raw* dx, const float* gamma,
                       const float* beta, const float* save_mean, const float* save_rstd,
                       float* dgamma, float* dbeta, int rows, int c, hipStream_t stream,
                       SplitSlabs sl) {
    // (not for unsummed split slabs: 9 fp32 slabs are real bytes, 32 blocks cannot stream them)
    if (!use_slab_kernel(rows, c) || sl.splits > 0) return false;
    const PoolGeom none = {c / 8, 0, 0};
    hipLaunchKernelGGL(bn_bwd_slab_kernel<false>, dim3(c / 16), dim3(256), 0, stream, x,
                       (const uint16_t*)dy, dx, gamma, beta, save_mean, save_rstd, dgamma, dbeta, rows,
                       c, 1.f / (float)rows, none);
    return true;
}

bool bn_bwd_single_launch(int dy_rows, int c) { return use_slab_kernel(dy_rows, c); }

// ------------------------------------- forward, training: fold + apply in ONE launch
// The fold of the partial statistics rows used to be a launch of its own between the producer
// and the apply pass - a few microseconds of kernel plus a dependent-launch gap on the critical
// path of every layer.  Here every block of the apply pass folds the rows itself, for the 32
// channels it works on only (grid.y = channel slices, so a block reads nb x 64 floats, not
// nb x 2c): 64 columns x 4 row parts per block, summed in a fixed order.  The blocks of the
// first row chunk also store mean / rstd for backward and move the running statistics.
constexpr int kFoldSlice = 32;          // channels per block
constexpr int kFoldMaxRows = 128;       // partial rows up to which the fold rides in the apply pass

template <bool POOL, typename T>
__global__ __launch_bounds__(256) void bn_fold_apply_kernel(
    const T* __restrict__ x, T* __restrict__ y, const float* __restrict__ gamma,
    const float* __restrict__ beta, const float* __restrict__ partials, int nb,
    const float* __restrict__ shift, float* __restrict__ save_mean, float* __restrict__ save_rstd,
    float* __restrict__ running_mean, float* __restrict__ running_var, int stat_rows,
    float momentum, float eps, int c, PoolGeom geo, int out_rows) {
    __shared__ float part[4][2 * kFoldSlice];
    const int tid = threadIdx.x;
    const int col0 = blockIdx.y * kFoldSlice;
    {
        const int j = tid & 63, pt = tid >> 6;          // column j = stat * 32 + channel
        const float* src = partials + (size_t)(j >> 5) * c + col0 + (j & 31);
        // every row of this thread in flight at once: the rows were written by another kernel,
        // possibly through another XCD's L2, and a dependent round trip costs more than the rest
        float v[kFoldMaxRows / 4];
#pragma unroll
        for (int q = 0; q < kFoldMaxRows / 4; ++q) {
            const int k = pt + 4 * q;
            v[q] = k < nb ? src[(size_t)k * 2 * c] : 0.f;
        }
        float acc = 0.f;
#pragma unroll
        for (int q = 0; q < kFoldMaxRows / 4; ++q) acc += v[q];
        part[pt][j] = acc;
    }
    __syncthreads();
    const int cgl = tid & 3, rl = tid >> 2;             // 4 column groups x 64 row lanes
    const int col = col0 + cgl * 8;
    float sc[8], sh[8];
    {
        const float inv_n = 1.f / (float)stat_rows;
        float g[8], bb[8];
        load_consts(gamma, col, g);
        load_consts(beta, col, bb);
        const bool writer = blockIdx.x == 0 && rl == 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int ch = cgl * 8 + k;
            const float s1 = (part[0][ch] + part[1][ch]) + (part[2][ch] + part[3][ch]);
            const float s2 = (part[0][kFoldSlice + ch] + part[1][kFoldSlice + ch]) +
                             (part[2][kFoldSlice + ch] + part[3][kFoldSlice + ch]);
            const float m0 = s1 * inv_n;
            const float var = fmaxf(s2 * inv_n - m0 * m0, 0.f);
            // the sums are of x - shift: a shift only moves the mean
            const float mean = shift ? m0 + shift[col + k] : m0;
            const float rs = rsqrtf(var + eps);
            sc[k] = g[k] * rs;
            sh[k] = bb[k] - mean * sc[k];
            if (writer) {
                save_mean[col + k] = mean;
                save_rstd[col + k] = rs;
                if (running_mean) {
                    const float unbiased =
                        stat_rows > 1 ? var * ((float)stat_rows / (float)(stat_rows - 1)) : var;
                    running_mean[col + k] = (1.f - momentum) * running_mean[col + k] + momentum * mean;
                    running_var[col + k] = (1.f - momentum) * running_var[col + k] + momentum * unbiased;
                }
            }
        }
    }
    for (int r = blockIdx.x * 64 + rl; r < out_rows; r += gridDim.x * 64) {
        float o[8];
        if constexpr (POOL) {
            float v[4][8];
            load_window(x, geo.in_pixel((size_t)r), col >> 3, c, geo.hw, v);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float z0 = __builtin_fmaf(v[0][k], sc[k], sh[k]);
                const float z1 = __builtin_fmaf(v[1][k], sc[k], sh[k]);
                const float z2 = __builtin_fmaf(v[2][k], sc[k], sh[k]);
                const float z3 = __builtin_fmaf(v[3][k], sc[k], sh[k]);
                o[k] = fmaxf(fmaxf(fmaxf(z0, z1), fmaxf(z2, z3)), 0.f);
            }
        } else {
            float v[8];
            load8(x + (size_t)r * c + col, v);
#pragma unroll
            for (int k = 0; k < 8; ++k) o[k] = fmaxf(__builtin_fmaf(v[k], sc[k], sh[k]), 0.f);
        }
        store8(y + (size_t)r * c + col, o);
    }
}

namespace {
// grid of the fused pass: channel slices x row chunks, 1-4 rows per thread, ~512 blocks
dim3 fold_apply_grid(int out_rows, int c) {
    const int slices = c / kFoldSlice;
    long long per_thread = (long long)out_rows * slices / (64LL * 512);
    per_thread = per_thread < 1 ? 1 : per_thread > 4 ? 4 : per_thread;
    const int chunks = (int)((out_rows + 64 * per_thread - 1) / (64 * per_thread));
    return dim3(chunks < 1 ? 1 : chunks, slices);
}
bool fold_in_apply(int nb, int c) { return nb > 0 && nb <= kFoldMaxRows && c % kFoldSlice == 0; }

template <typename T>
bool fold_apply_plain(const T* x, T* y, const float* gamma, const float* beta, const BnPartials& part,
                      float* save_mean, float* save_rstd, float* running_mean, float* running_var,
                      int rows, int c, float momentum, float eps, hipStream_t stream) {
    if (!fold_in_apply(part.rows, c)) return false;
    const PoolGeom none = {c / 8, 0, 0};
    hipLaunchKernelGGL((bn_fold_apply_kernel<false, T>), fold_apply_grid(rows, c), dim3(256), 0,
                       stream, x, y, gamma, beta, part.data, part.rows, part.shift, save_mean,
                       save_rstd, running_mean, running_var, rows, momentum, eps, c, none, rows);
    return true;
}
template bool fold_apply_plain<uint16_t>(const uint16_t*, uint16_t*, const float*, const float*,
                                         const BnPartials&, float*, float*, float*, float*, int, int,
                                         float, float, hipStream_t);
template bool fold_apply_plain<float>(const float*, float*, const float*, const float*,
                                      const BnPartials&, float*, float*, float*, float*, int, int,
                                      float, float, hipStream_t);
}  // namespace

namespace {
PoolGeom pool_geom(int batch, int hw, int c, const char* who) {
    if (batch <= 0 || hw < 2 || hw > 64 || (hw & (hw - 1)))
        throw std::invalid_argument(std::string(who) + ": hw must be a power of two in [2, 64]");
    int l = 0;
    while ((1 << l) < hw / 2) ++l;
    return {c / 8, l, hw};
}
}  // namespace

// ------------------------------------------ backward: fold + apply in ONE launch
// The mirror image of bn_fold_apply_kernel: every block of the apply pass folds the partial rows
// of dbeta / dgamma for its own 32 channels (up to 256 rows: 64 loads per thread, all in flight),
// the blocks of the first row chunk store the two gradients, then the tensor is streamed once.
constexpr int kBwdFoldMaxRows = 256;

template <bool POOL, typename T>
__global__ __launch_bounds__(256) void bn_bwd_fold_apply_kernel(
    const T* __restrict__ x, const T* __restrict__ dy, T* __restrict__ dx,
    const float* __restrict__ gamma, const float* __restrict__ beta,
    const float* __restrict__ mean, const float* __restrict__ rstd,
    const float* __restrict__ partials, int nb, float* __restrict__ dgamma,
    float* __restrict__ dbeta, int stat_rows, int c, PoolGeom geo, int out_rows) {
    __shared__ float part[4][2 * kFoldSlice];
    const int tid = threadIdx.x;
    const int col0 = blockIdx.y * kFoldSlice;
    {
        const int j = tid & 63, pt = tid >> 6;
        const float* src = partials + (size_t)(j >> 5) * c + col0 + (j & 31);
        float v[kBwdFoldMaxRows / 4];
#pragma unroll
        for (int q = 0; q < kBwdFoldMaxRows / 4; ++q) {
            const int k = pt + 4 * q;
            v[q] = k < nb ? src[(size_t)k * 2 * c] : 0.f;
        }
        float acc = 0.f;
#pragma unroll
        for (int q = 0; q < kBwdFoldMaxRows / 4; ++q) acc += v[q];
        part[pt][j] = acc;
    }
    __syncthreads();
    const int cgl = tid & 3, rl = tid >> 2;
    const int col = col0 + cgl * 8;
    const float inv_n = 1.f / (float)stat_rows;
    float ysc[8], ysh[8], k1[8], k2[8], k3[8];
    {
        float gm[8], bt[8], mn[8], rs[8];
        load_consts(gamma, col, gm); load_consts(beta, col, bt);
        load_consts(mean, col, mn); load_consts(rstd, col, rs);
        const bool writer = blockIdx.x == 0 && rl == 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int ch = cgl * 8 + k;
            const float db_sum = (part[0][ch] + part[1][ch]) + (part[2][ch] + part[3][ch]);
            const float dg_sum = (part[0][kFoldSlice + ch] + part[1][kFoldSlice + ch]) +
                                 (part[2][kFoldSlice + ch] + part[3][kFoldSlice + ch]);
            if (writer) {
                dbeta[col + k] = db_sum;
                dgamma[col + k] = dg_sum;
            }
            const float mrs = mn[k] * rs[k];
            const float dg = dg_sum * inv_n, db = db_sum * inv_n;
            ysc[k] = gm[k] * rs[k];
            k1[k] = gm[k] * rs[k];
            if (POOL) {     // (the pooled kernels' forms of the same constants, bit for bit)
                ysh[k] = __builtin_fmaf(-gm[k], mrs, bt[k]);
                k2[k] = k1[k] * __builtin_fmaf(-mrs, dg, db);
                k3[k] = (k1[k] * rs[k]) * dg;
            } else {
                ysh[k] = bt[k] - mn[k] * ysc[k];
                k2[k] = k1[k] * (db - mrs * dg);
                k3[k] = k1[k] * rs[k] * dg;
            }
        }
    }
    for (int r = blockIdx.x * 64 + rl; r < out_rows; r += gridDim.x * 64) {
        float g[8];
        load8(dy + (size_t)r * c + col, g);
        if constexpr (POOL) {
            float v[4][8], o[4][8];
            const size_t ipix = geo.in_pixel((size_t)r);
            load_window(x, ipix, col >> 3, c, geo.hw, v);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                bool live;
                const int w = window_winner(__builtin_fmaf(v[0][k], ysc[k], ysh[k]),
                                            __builtin_fmaf(v[1][k], ysc[k], ysh[k]),
                                            __builtin_fmaf(v[2][k], ysc[k], ysh[k]),
                                            __builtin_fmaf(v[3][k], ysc[k], ysh[k]), live);
                const float gk = live ? k1[k] * g[k] : 0.f;
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    o[q][k] = __builtin_fmaf(-k3[k], v[q][k], (w == q ? gk : 0.f) - k2[k]);
            }
            store_window(dx, ipix, col, c, geo.hw, o);
        } else {
            float v[8], o[8];
            load8(x + (size_t)r * c + col, v);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float gi = (__builtin_fmaf(v[k], ysc[k], ysh[k]) > 0.f) ? g[k] : 0.f;
                o[k] = k1[k] * gi - k2[k] - k3[k] * v[k];
            }
            store8(dx + (size_t)r * c + col, o);
        }
    }
}

namespace {
bool fold_in_bwd_apply(int nb, int c) {
    return nb > 0 && nb <= kBwdFoldMaxRows && c % kFoldSlice == 0;
}
}  // namespace

template <typename T>
bool bn_bwd_fold_apply_plain(const T* x, const T* dy, T* dx, const float* gamma,
                             const float* beta, const float* save_mean, const float* save_rstd,
                             const float* partials, int nb, float* dgamma, float* dbeta, int rows,
                             int c, hipStream_t stream) {
    if (!fold_in_bwd_apply(nb, c)) return false;
    const PoolGeom none = {c / 8, 0, 0};
    hipLaunchKernelGGL((bn_bwd_fold_apply_kernel<false, T>), fold_apply_grid(rows, c), dim3(256), 0,
                       stream, x, dy, dx, gamma, beta, save_mean, save_rstd, partials, nb, dgamma,
                       dbeta, rows, c, none, rows);
    return true;
}
template bool bn_bwd_fold_apply_plain<uint16_t>(const uint16_t*, const uint16_t*, uint16_t*,
                                                const float*, const float*, const float*,
                                                const float*, const float*, int, float*, float*, int,
                                                int, hipStream_t);
template bool bn_bwd_fold_apply_plain<float>(const float*, const float*, float*, const float*,
                                             const float*, const float*, const float*, const float*,
                                             int, float*, float*, int, int, hipStream_t);

namespace {
template <typename T>
void pool_fwd_train_launch(const T* x, T* y, const float* gamma, const float* beta,
                           float* running_mean, float* running_var, float* save_mean,
                           float* save_rstd, float* ws, int batch, int hw, int c, float momentum,
                           float eps, hipStream_t stream, BnPartials from) {
    const int rows = batch * hw * hw;
    check(rows, c, "bn_relu_pool_fwd_train");
    const PoolGeom geo = pool_geom(batch, hw, c, "bn_relu_pool_fwd_train");
    const BnPartials part = statistics_partials(x, ws, rows, c, from, stream);
    if (fold_in_apply(part.rows, c)) {
        hipLaunchKernelGGL((bn_fold_apply_kernel<true, T>), fold_apply_grid(rows / 4, c),
                           dim3(256), 0, stream, x, y, gamma, beta, part.data, part.rows, part.shift,
                           save_mean, save_rstd, running_mean, running_var, rows, momentum, eps, c,
                           geo, rows / 4);
        return;
    }
    finish_statistics(part, rows, c, save_mean, save_rstd, running_mean, running_var, momentum, eps,
                      stream);
    const size_t total8 = (size_t)(rows / 4) * (c / 8);
    hipLaunchKernelGGL((bn_apply_pool_kernel<false, T>), dim3(apply_blocks(total8)), dim3(256), 0,
                       stream, x, y, gamma, beta, (const float*)save_mean,
                       (const float*)save_rstd, c, eps, geo, total8);
}

template <typename T>
void pool_fwd_eval_launch(const T* x, T* y, const float* gamma, const float* beta,
                          const float* running_mean, const float* running_var, int batch, int hw,
                          int c, float eps, hipStream_t stream) {
    const int rows = batch * hw * hw;
    check(rows, c, "bn_relu_pool_fwd_eval");
    const PoolGeom geo = pool_geom(batch, hw, c, "bn_relu_pool_fwd_eval");
    const size_t total8 = (size_t)(rows / 4) * (c / 8);
    hipLaunchKernelGGL((bn_apply_pool_kernel<true, T>), dim3(apply_blocks(total8)), dim3(256), 0,
                       stream, x, y, gamma, beta, running_mean, running_var, c, eps, geo, total8);
}
}  // namespace

void bn_relu_pool_fwd_train(const bf16_raw* x, bf16_raw* y, const float* gamma, const float* beta,
                            float* running_mean, float* running_var, float* save_mean,
                            float* save_rstd, float* ws, int batch, int hw, int c, float momentum,
                            float eps, hipStream_t stream, BnPartials from) {
    pool_fwd_train_launch(x, y, gamma, beta, running_mean, running_var, save_mean, save_rstd, ws,
                          batch, hw, c, momentum, eps, stream, from);
}
void bn_relu_pool_fwd_train(const float* x, float* y, const float* gamma, const float* beta,
                            float* running_mean, float* running_var, float* save_mean,
                            float* save_rstd, float* ws, int batch, int hw, int c, float momentum,
                            float eps, hipStream_t stream) {
    pool_fwd_train_launch(x, y, gamma, beta, running_mean, running_var, save_mean, save_rstd, ws,
                          batch, hw, c, momentum, eps, stream, BnPartials());
}

void bn_relu_pool_fwd_eval(const bf16_raw* x, bf16_raw* y, const float* gamma, const float* beta,
                           const float* running_mean, const float* running_var, int batch, int hw,
                           int c, float eps, hipStream_t stream) {
    pool_fwd_eval_launch(x, y, gamma, beta, running_mean, running_var, batch, hw, c, eps, stream);
}
void bn_relu_pool_fwd_eval(const float* x, float* y, const float* gamma, const float* beta,
                           const float* running_mean, const float* running_var, int batch, int hw,
                           int c, float eps, hipStream_t stream) {
    pool_fwd_eval_launch(x, y, gamma, beta, running_mean, running_var, batch, hw, c, eps, stream);
}

void bn_relu_pool_bwd(const bf16_raw* x, bf16_raw* dy, bf16_raw* dx, const float* gamma,
                      const float* beta, const float* save_mean, const float* save_rstd,
                      float* dgamma, float* dbeta, float* ws, int batch, int hw, int c,
                      hipStream_t stream, SplitSlabs sl, BnPartials sums) {
    const int rows = batch * hw * hw;
    check(rows, c, "bn_relu_pool_bwd");
    const PoolGeom geo = pool_geom(batch, hw, c, "bn_relu_pool_bwd");
    const int out_rows = rows / 4;
    if (!sums.data && dx && sl.splits == 0 && use_slab_kernel(out_rows, c)) {
        hipLaunchKernelGGL(bn_bwd_slab_kernel<true>, dim3(c / 16), dim3(256), 0, stream, x,
                           (const uint16_t*)dy, dx, gamma, beta, save_mean, save_rstd, dgamma, dbeta,
                           out_rows, c, 1.f / (float)rows, geo);
        return;
    }
    const ColReducePlan pl = plan_col_reduce(out_rows, c);
    int nb = pl.blocks;
    const float* partials = ws;
    if (sums.data) {            // the producer of dy already left the sums, one partial row per tile
        if (sl.splits > 0 || sums.rows <= 0)
            throw std::invalid_argument("bn_relu_pool_bwd: inconsistent partial sums");
        nb = sums.rows;
        partials = sums.data;
    } else if (sl.splits > 0) {
        hipLaunchKernelGGL((bn_pool_bwd_reduce_kernel<true, uint16_t>), dim3(pl.blocks), dim3(256), 0,
                           stream, x, dy, gamma, beta, save_mean, save_rstd, ws, out_rows, c,
                           pl.rows_per_block, geo, sl);
    } else {
        hipLaunchKernelGGL((bn_pool_bwd_reduce_kernel<false, uint16_t>), dim3(pl.blocks), dim3(256), 0,
                           stream, x, dy, gamma, beta, save_mean, save_rstd, ws, out_rows, c,
                           pl.rows_per_block, geo, sl);
    }
    if (dx && fold_in_bwd_apply(nb, c)) {
        hipLaunchKernelGGL((bn_bwd_fold_apply_kernel<true, uint16_t>), fold_apply_grid(out_rows, c),
                           dim3(256), 0, stream, x, (const uint16_t*)dy, dx, gamma, beta, save_mean,
                           save_rstd, partials, nb, dgamma, dbeta, rows, c, geo, out_rows);
        return;
    }
    hipLaunchKernelGGL(bn_bwd_finish_kernel, dim3(fold_blocks(c)), dim3(256), 0, stream, partials,
                       dgamma, dbeta, nb, c);
    if (!dx) return;        // the caller's next kernel produces dx on the fly (conv1_wgrad_bn_pool)
    const size_t total8 = (size_t)out_rows * (c / 8);
    hipLaunchKernelGGL(bn_pool_bwd_apply_kernel<uint16_t>, dim3(apply_blocks(total8)), dim3(256), 0,
                       stream, x, (const uint16_t*)dy, dx, gamma, beta, save_mean, save_rstd,
                       (const float*)dgamma, (const float*)dbeta, rows, c, geo, total8);
}

// fp32 activations: partial sums, then fold + apply (or fold and apply) - no split slabs, no
// one-launch variant, no sums from a producer's epilogue
void bn_relu_pool_bwd(const float* x, float* dy, float* dx, const float* gamma, const float* beta,
                      const float* save_mean, const float* save_rstd, float* dgamma, float* dbeta,
                      float* ws, int batch, int hw, int c, hipStream_t stream) {
    const int rows = batch * hw * hw;
    check(rows, c, "bn_relu_pool_bwd");
    if (!dx) throw std::invalid_argument("bn_relu_pool_bwd: dx missing");
    const PoolGeom geo = pool_geom(batch, hw, c, "bn_relu_pool_bwd");
    const int out_rows = rows / 4;
    const ColReducePlan pl = plan_col_reduce(out_rows, c);
    hipLaunchKernelGGL((bn_pool_bwd_reduce_kernel<false, float>), dim3(pl.blocks), dim3(256), 0,
                       stream, x, dy, gamma, beta, save_mean, save_rstd, ws, out_rows, c,
                       pl.rows_per_block, geo, SplitSlabs());
    if (fold_in_bwd_apply(pl.blocks, c)) {
        hipLaunchKernelGGL((bn_bwd_fold_apply_kernel<true, float>), fold_apply_grid(out_rows, c),
                           dim3(256), 0, stream, x, (const float*)dy, dx, gamma, beta, save_mean,
                           save_rstd, (const float*)ws, pl.blocks, dgamma, dbeta, rows, c, geo,
                           out_rows);
        return;
    }
    hipLaunchKernelGGL(bn_bwd_finish_kernel, dim3(fold_blocks(c)), dim3(256), 0, stream,
                       (const float*)ws, dgamma, dbeta, pl.blocks, c);
    const size_t total8 = (size_t)out_rows * (c / 8);
    hipLaunchKernelGGL(bn_pool_bwd_apply_kernel<float>, dim3(apply_blocks(total8)), dim3(256), 0,
                       stream, x, (const float*)dy, dx, gamma, beta, save_mean, save_rstd,
                       (const float*)dgamma, (const float*)dbeta, rows, c, geo, total8);
}

// ------------------------------------------------ split-K finish + BatchNorm + ReLU (+ pool), 2x2 maps
// The two deepest VGG layers run on 2x2 maps: 4 * batch rows, so their convolutions split the taps
// and leave fp32 slabs.  Summing the slabs, the batch statistics and the normalisation then are
// three dependent launches of ~6 us for a few MB.  Here one block owns 16 channels and ALL rows
// (16 rows x 4 channels per thread, in registers), so the statistics are complete inside the block
// and everything happens in one launch: slab sum (+ bias) -> bf16 conv output for backward,
// bias-free column sums -> mean / rstd / running statistics, scale / shift / ReLU (and the 2x2
// max-pool: a window is 4 consecutive rows, all in one thread) -> bf16 activation.
// The activation is formed from the bf16-rounded conv output, like the separate apply pass does.
// CPB channels per block: 16 (32 blocks at 512 channels) or 8 (64 blocks, half the rows per thread;
// a row piece is then 32 bytes per slab, but the slabs sit in L2 and the launch is bound by its
// dependent round trips, not by bytes)
template <bool POOL, int CPB>
__global__ __launch_bounds__(256) void splitk_finish_bn_kernel(
    const float* __restrict__ slabs, int splits, size_t slab_stride, const float* __restrict__ bias,
    uint16_t* __restrict__ pre, uint16_t* __restrict__ out, const float* __restrict__ gamma,
    const float* __restrict__ beta, float* __restrict__ running_mean, float* __restrict__ running_var,
    float* __restrict__ save_mean, float* __restrict__ save_rstd, int batch, int c, float momentum,
    float eps) {
    constexpr int QB = CPB / 4;             // channel quads per block
    constexpr int RL = 256 / QB;            // row lanes (one image = one 4-row window each)
    constexpr int W = 256 / RL;             // windows per thread at batch 256
    constexpr int COLS = 2 * CPB;           // statistics columns: sums, sums of squares
    __shared__ float part[RL][COLS + 1];
    __shared__ float part2[8][COLS];
    __shared__ float stat[2][CPB];          // mean, rstd
    const int tid = threadIdx.x, q = tid % QB, rl = tid / QB;
    const int col = blockIdx.x * CPB + q * 4;
    const int rows = batch * 4;
    const float4 b4 = *(const float4*)(bias + col);
    float v[W][4][4];                       // [window][row of the window][channel], bias-free sums
#pragma unroll
    for (int w = 0; w < W; ++w)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int k = 0; k < 4; ++k) v[w][j][k] = 0.f;
    // All loads of a thread in flight together: the launch is bound by its dependent round trips
    // to L2, not by bytes.  (With the run-time loop over the splits INSIDE the row loops every row
    // became a chain of round trips: 16 us per launch; split by split: 3 round trips; with the
    // usual three splits unrolled, where the registers allow it: one.)
    auto load_splits = [&](auto nsplit_tag) {
        constexpr int NS = decltype(nsplit_tag)::value;
        float4 t[NS][W][4];
#pragma unroll
        for (int sidx = 0; sidx < NS; ++sidx)
#pragma unroll
            for (int w = 0; w < W; ++w) {
                const int img = w * RL + rl;
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    t[sidx][w][j] = img < batch
                        ? *(const float4*)(slabs + sidx * slab_stride + col + (size_t)(img * 4 + j) * c)
                        : float4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
        for (int sidx = 0; sidx < NS; ++sidx)       // (fixed order: split 0 first)
#pragma unroll
            for (int w = 0; w < W; ++w)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    v[w][j][0] += t[sidx][w][j].x; v[w][j][1] += t[sidx][w][j].y;
                    v[w][j][2] += t[sidx][w][j].z; v[w][j][3] += t[sidx][w][j].w;
                }
    };
    if (splits == 3 && W <= 2) {
        load_splits(std::integral_constant<int, 3>{});
    } else {
        for (int sidx = 0; sidx < splits; ++sidx) {
            const float* sl = slabs + sidx * slab_stride + col;
            float4 t[W][4];
#pragma unroll
            for (int w = 0; w < W; ++w) {
                const int img = w * RL + rl;
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    t[w][j] = img < batch ? *(const float4*)(sl + (size_t)(img * 4 + j) * c)
                                          : float4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int w = 0; w < W; ++w)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    v[w][j][0] += t[w][j].x; v[w][j][1] += t[w][j].y;
                    v[w][j][2] += t[w][j].z; v[w][j][3] += t[w][j].w;
                }
        }
    }
    float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int w = 0; w < W; ++w)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                s1[k] += v[w][j][k];
                s2[k] += v[w][j][k] * v[w][j][k];
            }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        part[rl][q * 4 + k] = s1[k];
        part[rl][CPB + q * 4 + k] = s2[k];
    }
    __syncthreads();
    if (tid < 8 * COLS) {                   // eight row segments per column, then the eight sums
        const int cidx = tid % COLS, seg = tid / COLS;
        float acc = 0.f;
        for (int r = seg * (RL / 8); r < (seg + 1) * (RL / 8); ++r) acc += part[r][cidx];
        part2[seg][cidx] = acc;
    }
    __syncthreads();
    if (tid < CPB) {
        float t1 = 0.f, t2 = 0.f;
#pragma unroll
        for (int seg = 0; seg < 8; ++seg) {
            t1 += part2[seg][tid];
            t2 += part2[seg][CPB + tid];
        }
        const int ch = blockIdx.x * CPB + tid;
        const float inv_n = 1.f / (float)rows;
        const float m0 = t1 * inv_n;
        const float var = fmaxf(t2 * inv_n - m0 * m0, 0.f);
        const float mean = m0 + bias[ch];           // the sums are bias-free: a shift moves the mean only
        const float rs = rsqrtf(var + eps);
        stat[0][tid] = mean;
        stat[1][tid] = rs;
        save_mean[ch] = mean;
        save_rstd[ch] = rs;
        if (running_mean) {
            const float unbiased = rows > 1 ? var * ((float)rows / (float)(rows - 1)) : var;
            running_mean[ch] = (1.f - momentum) * running_mean[ch] + momentum * mean;
            running_var[ch] = (1.f - momentum) * running_var[ch] + momentum * unbiased;
        }
    }
    __syncthreads();
    float sc[4], sh[4];
    const float bb[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        sc[k] = gamma[col + k] * stat[1][q * 4 + k];
        sh[k] = beta[col + k] - stat[0][q * 4 + k] * sc[k];
    }
#pragma unroll
    for (int w = 0; w < W; ++w) {
        const int img = w * RL + rl;
        if (img >= batch) continue;
        float best[4] = {0.f, 0.f, 0.f, 0.f};       // ReLU = max with 0
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            uint16_t pb[4];
            float z[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                pb[k] = float_to_bf16_bits(v[w][j][k] + bb[k]);
                z[k] = fmaxf(__builtin_fmaf(bf16_bits_to_float(pb[k]), sc[k], sh[k]), 0.f);
                best[k] = fmaxf(best[k], z[k]);
            }
            const size_t at = (size_t)(img * 4 + j) * c + col;
            *(uint2*)(pre + at) = make_uint2((uint32_t)pb[0] | ((uint32_t)pb[1] << 16),
                                             (uint32_t)pb[2] | ((uint32_t)pb[3] << 16));
            if (!POOL) *(uint2*)(out + at) = make_uint2(pack2(z[0], z[1]), pack2(z[2], z[3]));
        }
        if (POOL)
            *(uint2*)(out + (size_t)img * c + col) =
                make_uint2(pack2(best[0], best[1]), pack2(best[2], best[3]));
    }
}

bool splitk_finish_bn_relu(const SplitSlabs& slabs, const float* bias, bf16_raw* pre, bf16_raw* out,
                           const float* gamma, const float* beta, float* running_mean,
                           float* running_var, float* save_mean, float* save_rstd, int batch, int hw,
                           int c, bool pool, float momentum, float eps, hipStream_t stream) {
    if (slabs.splits < 1 || !slabs.data || hw != 2 || c % 16 || batch < 1 || batch > 256 ||
        slabs.stride % 4)
        return false;
    static const int cpb = getenv("DML_FINISH_BN_CPB") ? atoi(getenv("DML_FINISH_BN_CPB")) : 16;
#define DML_LAUNCH_FINISH_BN(POOL, CPB)                                                              \
    hipLaunchKernelGGL((splitk_finish_bn_kernel<POOL, CPB>), dim3(c / CPB), dim3(256), 0, stream,    \
                       slabs.data, slabs.splits, slabs.stride, bias, pre, out, gamma, beta,          \
                       running_mean, running_var, save_mean, save_rstd, batch, c, momentum, eps)
    if (cpb == 16) {
        if (pool) DML_LAUNCH_FINISH_BN(true, 16); else DML_LAUNCH_FINISH_BN(false, 16);
    } else {
        if (pool) DML_LAUNCH_FINISH_BN(true, 8); else DML_LAUNCH_FINISH_BN(false, 8);
    }
#undef DML_LAUNCH_FINISH_BN
    return true;
}

}  // namespace dml
