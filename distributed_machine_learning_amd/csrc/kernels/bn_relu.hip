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
#include <stdexcept>
#include <string>

#include "dml_column_reduce.h"
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

// ----------------------------------------------------------------- forward, training
__global__ __launch_bounds__(256) void bn_stats_kernel(const uint16_t* __restrict__ x,
                                                       float* __restrict__ ws, int rows, int c,
                                                       int rows_per_block) {
    __shared__ float red[4096];
    column_partials(
        [&](int r, int cg, float (&a)[8], float (&b)[8]) {
            float v[8];
            unpack8(*(const uint4*)(x + (size_t)r * c + cg * 8), v);
#pragma unroll
            for (int i = 0; i < 8; ++i) { a[i] += v[i]; b[i] += v[i] * v[i]; }
        },
        ws, rows, c, rows_per_block, red);
}

__global__ __launch_bounds__(256) void bn_stats_finish_kernel(
    const float* __restrict__ ws, float* __restrict__ save_mean, float* __restrict__ save_rstd,
    float* __restrict__ running_mean, float* __restrict__ running_var, int nb, int rows, int c,
    float momentum, float eps) {
    const float inv_n = 1.f / (float)rows;
    column_fold(
        [&](int col, float s1, float s2) {
            const float mean = s1 * inv_n;
            const float var = fmaxf(s2 * inv_n - mean * mean, 0.f);
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

// y = relu(x * scale + shift) with scale = gamma * rstd, shift = beta - mean * scale.  A
// thread's 8 channels are the same in every grid-stride iteration (the stride is a multiple
// of c/8), so the per-channel constants are loaded once as 16-byte vectors; the streaming
// loop is load - fma - max - store with no branches.
template <bool FROM_VAR>
__global__ __launch_bounds__(256) void bn_apply_kernel(
    const uint16_t* __restrict__ x, uint16_t* __restrict__ y, const float* __restrict__ gamma,
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
        unpack8(*(const uint4*)(x + i * 8), v);
#pragma unroll
        for (int k = 0; k < 8; ++k) o[k] = fmaxf(v[k] * sc[k] + sh[k], 0.f);
        *(uint4*)(y + i * 8) = pack8(o);
    }
}

void bn_relu_fwd_train(const bf16_raw* x, bf16_raw* y, const float* gamma, const float* beta,
                       float* running_mean, float* running_var, float* save_mean,
                       float* save_rstd, float* ws, int rows, int c, float momentum,
                       float eps, hipStream_t stream) {
    check(rows, c, "bn_relu_fwd_train");
    const ColReducePlan pl = plan_col_reduce(rows, c);
    hipLaunchKernelGGL(bn_stats_kernel, dim3(pl.blocks), dim3(256), 0, stream, x, ws, rows, c,
                       pl.rows_per_block);
    hipLaunchKernelGGL(bn_stats_finish_kernel, dim3((c + 63) / 64), dim3(256), 0, stream,
                       (const float*)ws, save_mean, save_rstd, running_mean, running_var, pl.blocks,
                       rows, c, momentum, eps);
    const size_t total8 = (size_t)rows * (c / 8);
    hipLaunchKernelGGL(bn_apply_kernel<false>, dim3(apply_blocks(total8)), dim3(256), 0, stream, x,
                       y, gamma, beta, (const float*)save_mean, (const float*)save_rstd, c, eps,
                       total8);
}

void bn_relu_fwd_eval(const bf16_raw* x, bf16_raw* y, const float* gamma, const float* beta,
                      const float* running_mean, const float* running_var, int rows, int c,
                      float eps, hipStream_t stream) {
    check(rows, c, "bn_relu_fwd_eval");
    const size_t total8 = (size_t)rows * (c / 8);
    hipLaunchKernelGGL(bn_apply_kernel<true>, dim3(apply_blocks(total8)), dim3(256), 0, stream, x,
                       y, gamma, beta, running_mean, running_var, c, eps, total8);
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
    // this thread's 8 channels are fixed: y > 0  <=>  x * ysc + ysh > 0, xh = x * rs - mrs
    float rs[8], mrs[8], ysc[8], ysh[8];
    {
        const int col0 = (threadIdx.x % (c / 8)) * 8;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            rs[i] = rstd[col0 + i];
            mrs[i] = mean[col0 + i] * rs[i];
            ysc[i] = gamma[col0 + i] * rs[i];
            ysh[i] = beta[col0 + i] - gamma[col0 + i] * mrs[i];
        }
    }
    column_partials(
        [&](int r, int cg, float (&a)[8], float (&b)[8]) {
            float v[8], g[8];
            unpack8(*(const uint4*)(x + (size_t)r * c + cg * 8), v);
            unpack8(*(const uint4*)(dy + (size_t)r * c + cg * 8), g);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float gi = (v[i] * ysc[i] + ysh[i] > 0.f) ? g[i] : 0.f;
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

__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(
    const uint16_t* __restrict__ x, const uint16_t* __restrict__ dy, uint16_t* __restrict__ dx,
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
            ysh[k] = beta[col] - gamma[col] * mrs;
            k1[k] = gamma[col] * rs[k];
            const float dg = dgamma[col] * inv_n, db = dbeta[col] * inv_n;
            // k1 * (g - db - (x*rs - mrs) * dg) = k1*g - k1*(db - mrs*dg) - (k1*rs*dg) * x
            k2[k] = k1[k] * (db - mrs * dg);
            k3[k] = k1[k] * rs[k] * dg;
        }
    }
    for (; i < total8; i += stride) {
        float v[8], g[8], o[8];
        unpack8(*(const uint4*)(x + i * 8), v);
        unpack8(*(const uint4*)(dy + i * 8), g);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float gi = (v[k] * ysc[k] + ysh[k] > 0.f) ? g[k] : 0.f;
            o[k] = k1[k] * gi - k2[k] - k3[k] * v[k];
        }
        *(uint4*)(dx + i * 8) = pack8(o);
    }
}

void bn_relu_bwd(const bf16_raw* x, const bf16_raw* dy, bf16_raw* dx, const float* gamma,
                 const float* beta, const float* save_mean, const float* save_rstd,
                 float* dgamma, float* dbeta, float* ws, int rows, int c, hipStream_t stream) {
    check(rows, c, "bn_relu_bwd");
    const ColReducePlan pl = plan_col_reduce(rows, c);
    hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3(pl.blocks), dim3(256), 0, stream, x, dy, gamma,
                       beta, save_mean, save_rstd, ws, rows, c, pl.rows_per_block);
    hipLaunchKernelGGL(bn_bwd_finish_kernel, dim3((c + 63) / 64), dim3(256), 0, stream,
                       (const float*)ws, dgamma, dbeta, pl.blocks, c);
    const size_t total8 = (size_t)rows * (c / 8);
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(apply_blocks(total8)), dim3(256), 0, stream, x, dy,
                       dx, gamma, beta, save_mean, save_rstd, (const float*)dgamma,
                       (const float*)dbeta, rows, c, total8);
}

}  // namespace dml
