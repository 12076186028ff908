// Fused BatchNorm2d + ReLU on NHWC bf16 activations (training, evaluation, backward).
//
// Reference ops: nn.BatchNorm2d(c) followed by nn.ReLU(inplace=True), enabled in the
// part3 model (/root/reference/part3/model.py:24-25); defaults momentum 0.1, eps 1e-5.
// Training normalises with the biased batch variance and moves the running variance
// with the unbiased one; evaluation folds the running statistics into a scale/shift.
//
// In NHWC a channel is a column of the [rows = B*H*W][c] matrix, so the statistics are
// column reductions: each block walks a strided slab of rows with 16-byte loads (8
// channels per thread), reduces its row lanes through LDS and adds one fp32 partial per
// column with an atomic.  Apply kernels re-derive scale/shift per thread from the tiny
// per-channel vectors and stream the tensor once.  All statistics stay fp32.
#include <stdexcept>
#include <string>

#include "dml_common.h"
#include "dml_kernels.h"

namespace dml {

namespace {

constexpr int kMaxBlocks = 1024;

struct RowSplit { int blocks, rows_per_block; };

// Enough blocks to fill the chip, but at least 4 rows per row lane (256 / (c/8) lanes per
// block) so the LDS reduction and the atomics stay a small fraction of the work.
RowSplit split_rows(int rows, int c) {
    const int lanes = 256 / (c / 8);
    int rpb = (rows + kMaxBlocks - 1) / kMaxBlocks;
    if (rpb < 4 * lanes) rpb = 4 * lanes;
    return {(rows + rpb - 1) / rpb, rpb};
}

void check(int rows, int c, const char* who) {
    if (c % 8 || c < 8 || c > 2048 || (256 % (c / 8)) != 0 || rows <= 0)
        throw std::invalid_argument(std::string(who) + ": c must be 8*2^k <= 2048 and rows > 0");
}

// two column reductions at once: ws[col] += sum a, ws[c + col] += sum b
template <typename F>
__device__ __forceinline__ void reduce_columns2(F&& row_values, float* ws, int rows, int c,
                                                int rows_per_block, float* red) {
    const int c8 = c / 8, lanes = 256 / c8;
    const int cg = threadIdx.x % c8, rl = threadIdx.x / c8;
    const int r0 = blockIdx.x * rows_per_block;
    const int r1 = min(rows, r0 + rows_per_block);
    float a[8], b[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = b[i] = 0.f;
    int r = r0 + rl;
    for (; r + lanes < r1; r += 2 * lanes) {       // two rows in flight per thread
        row_values(r, cg, a, b);
        row_values(r + lanes, cg, a, b);
    }
    for (; r < r1; r += lanes) row_values(r, cg, a, b);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        red[threadIdx.x * 8 + i] = a[i];
        red[2048 + threadIdx.x * 8 + i] = b[i];
    }
    __syncthreads();
    for (int col = threadIdx.x; col < 2 * c; col += 256) {
        const int which = col >= c ? 1 : 0;
        const int cc = col - which * c;
        const int g = cc / 8, i = cc % 8;
        float s = 0.f;
        for (int l = 0; l < lanes; ++l) s += red[which * 2048 + (l * c8 + g) * 8 + i];
        atomicAdd(ws + col, s);
    }
}

}  // namespace

// ----------------------------------------------------------------- forward, training
__global__ __launch_bounds__(256) void bn_stats_kernel(const uint16_t* __restrict__ x,
                                                       float* __restrict__ ws, int rows, int c,
                                                       int rows_per_block) {
    __shared__ float red[4096];
    reduce_columns2(
        [&](int r, int cg, float (&a)[8], float (&b)[8]) {
            float v[8];
            unpack8(*(const uint4*)(x + (size_t)r * c + cg * 8), v);
#pragma unroll
            for (int i = 0; i < 8; ++i) { a[i] += v[i]; b[i] += v[i] * v[i]; }
        },
        ws, rows, c, rows_per_block, red);
}

__global__ __launch_bounds__(256) void bn_apply_train_kernel(
    const uint16_t* __restrict__ x, uint16_t* __restrict__ y, const float* __restrict__ gamma,
    const float* __restrict__ beta, float* __restrict__ running_mean,
    float* __restrict__ running_var, float* __restrict__ save_mean, float* __restrict__ save_rstd,
    const float* __restrict__ ws, int rows, int c, float momentum, float eps, size_t total8) {
    const float inv_n = 1.f / (float)rows;
    if (blockIdx.x == 0) {
        for (int col = threadIdx.x; col < c; col += 256) {
            const float mean = ws[col] * inv_n;
            const float var = fmaxf(ws[c + col] * inv_n - mean * mean, 0.f);
            save_mean[col] = mean;
            save_rstd[col] = rsqrtf(var + eps);
            if (running_mean) {
                const float unbiased = rows > 1 ? var * ((float)rows / (float)(rows - 1)) : var;
                running_mean[col] = (1.f - momentum) * running_mean[col] + momentum * mean;
                running_var[col] = (1.f - momentum) * running_var[col] + momentum * unbiased;
            }
        }
    }
    const int c8 = c / 8;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < total8; i += stride) {
        const int col = (int)(i % c8) * 8;
        float v[8], o[8];
        unpack8(*(const uint4*)(x + i * 8), v);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float mean = ws[col + k] * inv_n;
            const float var = fmaxf(ws[c + col + k] * inv_n - mean * mean, 0.f);
            const float sc = gamma[col + k] * rsqrtf(var + eps);
            o[k] = fmaxf((v[k] - mean) * sc + beta[col + k], 0.f);
        }
        *(uint4*)(y + i * 8) = pack8(o);
    }
}

void bn_relu_fwd_train(const bf16_raw* x, bf16_raw* y, const float* gamma, const float* beta,
                       float* running_mean, float* running_var, float* save_mean,
                       float* save_rstd, float* stats_ws, int rows, int c, float momentum,
                       float eps, hipStream_t stream) {
    check(rows, c, "bn_relu_fwd_train");
    (void)hipMemsetAsync(stats_ws, 0, sizeof(float) * 2 * c, stream);
    const RowSplit rs = split_rows(rows, c);
    hipLaunchKernelGGL(bn_stats_kernel, dim3(rs.blocks), dim3(256), 0, stream, x, stats_ws, rows,
                       c, rs.rows_per_block);
    const size_t total8 = (size_t)rows * (c / 8);
    size_t blocks = (total8 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(bn_apply_train_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, x, y,
                       gamma, beta, running_mean, running_var, save_mean, save_rstd, stats_ws, rows,
                       c, momentum, eps, total8);
}

// --------------------------------------------------------------- forward, evaluation
__global__ __launch_bounds__(256) void bn_apply_eval_kernel(
    const uint16_t* __restrict__ x, uint16_t* __restrict__ y, const float* __restrict__ gamma,
    const float* __restrict__ beta, const float* __restrict__ rm, const float* __restrict__ rv,
    int c, float eps, size_t total8) {
    const int c8 = c / 8;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < total8; i += stride) {
        const int col = (int)(i % c8) * 8;
        float v[8], o[8];
        unpack8(*(const uint4*)(x + i * 8), v);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float sc = gamma[col + k] * rsqrtf(rv[col + k] + eps);
            o[k] = fmaxf((v[k] - rm[col + k]) * sc + beta[col + k], 0.f);
        }
        *(uint4*)(y + i * 8) = pack8(o);
    }
}

void bn_relu_fwd_eval(const bf16_raw* x, bf16_raw* y, const float* gamma, const float* beta,
                      const float* running_mean, const float* running_var, int rows, int c,
                      float eps, hipStream_t stream) {
    check(rows, c, "bn_relu_fwd_eval");
    const size_t total8 = (size_t)rows * (c / 8);
    size_t blocks = (total8 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(bn_apply_eval_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, x, y,
                       gamma, beta, running_mean, running_var, c, eps, total8);
}

// -------------------------------------------------------------------------- backward
// With xh = (x - mean) * rstd, y = gamma*xh + beta and g = dy * [y > 0]:
//   dbeta = sum g,  dgamma = sum g*xh,  dx = gamma*rstd * (g - dbeta/n - xh*dgamma/n).
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(
    const uint16_t* __restrict__ x, const uint16_t* __restrict__ dy,
    const float* __restrict__ gamma, const float* __restrict__ beta,
    const float* __restrict__ mean, const float* __restrict__ rstd, float* __restrict__ ws,
    int rows, int c, int rows_per_block) {
    __shared__ float red[4096];
    reduce_columns2(
        [&](int r, int cg, float (&a)[8], float (&b)[8]) {
            float v[8], g[8];
            unpack8(*(const uint4*)(x + (size_t)r * c + cg * 8), v);
            unpack8(*(const uint4*)(dy + (size_t)r * c + cg * 8), g);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int col = cg * 8 + i;
                const float xh = (v[i] - mean[col]) * rstd[col];
                const float gi = (gamma[col] * xh + beta[col] > 0.f) ? g[i] : 0.f;
                a[i] += gi;
                b[i] += gi * xh;
            }
        },
        ws, rows, c, rows_per_block, red);
}

__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(
    const uint16_t* __restrict__ x, const uint16_t* __restrict__ dy, uint16_t* __restrict__ dx,
    const float* __restrict__ gamma, const float* __restrict__ beta,
    const float* __restrict__ mean, const float* __restrict__ rstd, const float* __restrict__ ws,
    float* __restrict__ dgamma, float* __restrict__ dbeta, int rows, int c, size_t total8) {
    if (blockIdx.x == 0) {
        for (int col = threadIdx.x; col < c; col += 256) {
            dbeta[col] = ws[col];
            dgamma[col] = ws[c + col];
        }
    }
    const float inv_n = 1.f / (float)rows;
    const int c8 = c / 8;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < total8; i += stride) {
        const int col0 = (int)(i % c8) * 8;
        float v[8], g[8], o[8];
        unpack8(*(const uint4*)(x + i * 8), v);
        unpack8(*(const uint4*)(dy + i * 8), g);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int col = col0 + k;
            const float xh = (v[k] - mean[col]) * rstd[col];
            const float gi = (gamma[col] * xh + beta[col] > 0.f) ? g[k] : 0.f;
            o[k] = gamma[col] * rstd[col] * (gi - ws[col] * inv_n - xh * ws[c + col] * inv_n);
        }
        *(uint4*)(dx + i * 8) = pack8(o);
    }
}

void bn_relu_bwd(const bf16_raw* x, const bf16_raw* dy, bf16_raw* dx, const float* gamma,
                 const float* beta, const float* save_mean, const float* save_rstd,
                 float* dgamma, float* dbeta, float* stats_ws, int rows, int c,
                 hipStream_t stream) {
    check(rows, c, "bn_relu_bwd");
    (void)hipMemsetAsync(stats_ws, 0, sizeof(float) * 2 * c, stream);
    const RowSplit rs = split_rows(rows, c);
    hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3(rs.blocks), dim3(256), 0, stream, x, dy, gamma,
                       beta, save_mean, save_rstd, stats_ws, rows, c, rs.rows_per_block);
    const size_t total8 = (size_t)rows * (c / 8);
    size_t blocks = (total8 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, x, dy, dx,
                       gamma, beta, save_mean, save_rstd, stats_ws, dgamma, dbeta, rows, c, total8);
}

}  // namespace dml
