// Fused SGD (momentum + weight decay), the linear classifier and softmax cross-entropy.
//
// Reference call sites: optim.SGD(lr=0.1, momentum=0.9, weight_decay=1e-4).step()
// (/root/reference/part1/main.py:120-121,46), nn.Linear(512, 10)
// (/root/reference/part1/model.py:40,45), nn.CrossEntropyLoss and the top-1 count in
// test_model (/root/reference/part1/main.py:115,43,70-72).
#include <stdexcept>

#include "dml_common.h"
#include "dml_kernels.h"

namespace dml {

// ------------------------------------------------------------------------------- SGD
// One launch updates the whole model: p, g, m are the flat fp32 buffers.  Memory-bound:
// 16 B/lane loads and stores, grid capped at 2048 blocks with a grid-stride loop.
// GW: the gradients are bf16 (the reduced values of a bf16 all-reduce, read where it left them).
template <bool GW>
__global__ __launch_bounds__(256) void sgd_momentum_kernel(
    float4* __restrict__ p, const void* __restrict__ g_any, float4* __restrict__ m,
    uint2* __restrict__ shadow, size_t n4, float lr, float mu, float wd, float gscale, int first) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n4; i += stride) {
        float4 pv = p[i], gv, mv;
        if (GW) {
            const uint2 w = ((const uint2*)g_any)[i];
            gv.x = __builtin_bit_cast(float, w.x << 16);
            gv.y = __builtin_bit_cast(float, w.x & 0xffff0000u);
            gv.z = __builtin_bit_cast(float, w.y << 16);
            gv.w = __builtin_bit_cast(float, w.y & 0xffff0000u);
        } else {
            gv = ((const float4*)g_any)[i];
        }
        gv.x = gv.x * gscale + wd * pv.x;
        gv.y = gv.y * gscale + wd * pv.y;
        gv.z = gv.z * gscale + wd * pv.z;
        gv.w = gv.w * gscale + wd * pv.w;
        if (first) {
            mv = gv;
        } else {
            mv = m[i];
            mv.x = mu * mv.x + gv.x;
            mv.y = mu * mv.y + gv.y;
            mv.z = mu * mv.z + gv.z;
            mv.w = mu * mv.w + gv.w;
        }
        pv.x -= lr * mv.x;
        pv.y -= lr * mv.y;
        pv.z -= lr * mv.z;
        pv.w -= lr * mv.w;
        m[i] = mv;
        p[i] = pv;
        if (shadow) {
            uint2 s;
            s.x = pack2(pv.x, pv.y);
            s.y = pack2(pv.z, pv.w);
            shadow[i] = s;
        }
    }
}

namespace {
void sgd_launch(float* p, const void* g, bool g_bf16, float* m, bf16_raw* shadow, size_t n, float lr,
                float mu, float wd, float grad_scale, bool first_step, hipStream_t stream) {
    if (n % 4 != 0) throw std::invalid_argument("sgd_momentum: n must be a multiple of 4");
    if ((((uintptr_t)p | (uintptr_t)m) & 15) || ((uintptr_t)g & (g_bf16 ? 7 : 15)))
        throw std::invalid_argument("sgd_momentum: buffers must be 16-byte aligned");
    if (n == 0) return;
    const size_t n4 = n / 4;
    const int block = 256;
    size_t blocks = (n4 + block - 1) / block;
    if (blocks > 2048) blocks = 2048;
    if (g_bf16)
        hipLaunchKernelGGL(sgd_momentum_kernel<true>, dim3((unsigned)blocks), dim3(block), 0, stream,
                           (float4*)p, g, (float4*)m, (uint2*)shadow, n4, lr, mu, wd, grad_scale,
                           first_step ? 1 : 0);
    else
        hipLaunchKernelGGL(sgd_momentum_kernel<false>, dim3((unsigned)blocks), dim3(block), 0, stream,
                           (float4*)p, g, (float4*)m, (uint2*)shadow, n4, lr, mu, wd, grad_scale,
                           first_step ? 1 : 0);
}
}  // namespace

void sgd_momentum(float* p, const float* g, float* m, bf16_raw* shadow, size_t n, float lr,
                  float mu, float wd, float grad_scale, bool first_step, hipStream_t stream) {
    sgd_launch(p, g, false, m, shadow, n, lr, mu, wd, grad_scale, first_step, stream);
}

void sgd_momentum_bf16_grads(float* p, const bf16_raw* g, float* m, bf16_raw* shadow, size_t n,
                             float lr, float mu, float wd, float grad_scale, bool first_step,
                             hipStream_t stream) {
    sgd_launch(p, g, true, m, shadow, n, lr, mu, wd, grad_scale, first_step, stream);
}

__global__ __launch_bounds__(256) void cast_f32_to_bf16_kernel(const float4* __restrict__ x,
                                                               uint2* __restrict__ out, size_t n4) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n4; i += stride) {
        const float4 v = x[i];
        uint2 o;
        o.x = pack2(v.x, v.y);
        o.y = pack2(v.z, v.w);
        out[i] = o;
    }
}

void cast_f32_to_bf16(const float* x, bf16_raw* out, size_t n, hipStream_t stream) {
    if (n % 4 != 0) throw std::invalid_argument("cast_f32_to_bf16: n must be a multiple of 4");
    if (n == 0) return;
    const size_t n4 = n / 4;
    size_t blocks = (n4 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(cast_f32_to_bf16_kernel, dim3((unsigned)blocks), dim3(256), 0, stream,
                       (const float4*)x, (uint2*)out, n4);
}

__global__ __launch_bounds__(256) void cast_bf16_to_f32_kernel(const uint2* __restrict__ x,
                                                               float4* __restrict__ out, size_t n4) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n4; i += stride) {
        const uint2 v = x[i];
        float4 o;
        o.x = __uint_as_float(v.x << 16);
        o.y = __uint_as_float(v.x & 0xffff0000u);
        o.z = __uint_as_float(v.y << 16);
        o.w = __uint_as_float(v.y & 0xffff0000u);
        out[i] = o;
    }
}

void cast_bf16_to_f32(const bf16_raw* x, float* out, size_t n, hipStream_t stream) {
    if (n % 4 != 0) throw std::invalid_argument("cast_bf16_to_f32: n must be a multiple of 4");
    if (n == 0) return;
    const size_t n4 = n / 4;
    size_t blocks = (n4 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(cast_bf16_to_f32_kernel, dim3((unsigned)blocks), dim3(256), 0, stream,
                       (const uint2*)x, (float4*)out, n4);
}

// ---------------------------------------------------------------------------- linear
// out_features <= 16 (10 here): far too narrow for MFMA tiles.  One wave per batch row;
// each lane owns a contiguous slice of K, keeps `out_features` partial dots in registers
// and the wave reduces them with xor-shuffles.
template <int NOUT, typename T>
__global__ __launch_bounds__(256) void linear_fwd_kernel(
    const T* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
    float* __restrict__ logits, int batch, int k, int nout) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= batch) return;
    float acc[NOUT];
#pragma unroll
    for (int n = 0; n < NOUT; ++n) acc[n] = 0.f;
    for (int k0 = lane * 8; k0 < k; k0 += 64 * 8) {
        float xv[8];
        load8(x + (size_t)row * k + k0, xv);
#pragma unroll
        for (int n = 0; n < NOUT; ++n) {
            if (n < nout) {
                const float4 w0 = *(const float4*)(w + (size_t)n * k + k0);
                const float4 w1 = *(const float4*)(w + (size_t)n * k + k0 + 4);
                acc[n] += xv[0] * w0.x + xv[1] * w0.y + xv[2] * w0.z + xv[3] * w0.w +
                          xv[4] * w1.x + xv[5] * w1.y + xv[6] * w1.z + xv[7] * w1.w;
            }
        }
    }
#pragma unroll
    for (int n = 0; n < NOUT; ++n) {
        const float s = wave_sum(acc[n]);
        if (lane == 0 && n < nout) logits[(size_t)row * nout + n] = s + bias[n];
    }
}

namespace {
template <typename T>
void linear_fwd_launch(const T* x, const float* w, const float* bias, float* logits, int batch,
                       int in_features, int out_features, hipStream_t stream) {
    if (out_features < 1 || out_features > 16)
        throw std::invalid_argument("linear_fwd: out_features must be in [1,16]");
    if (in_features % 8 != 0) throw std::invalid_argument("linear_fwd: in_features % 8 != 0");
    if (batch <= 0) return;
    hipLaunchKernelGGL((linear_fwd_kernel<16, T>), dim3((batch + 3) / 4), dim3(256), 0, stream, x,
                       w, bias, logits, batch, in_features, out_features);
}
}  // namespace

void linear_fwd(const bf16_raw* x, const float* w, const float* bias, float* logits, int batch,
                int in_features, int out_features, hipStream_t stream) {
    linear_fwd_launch(x, w, bias, logits, batch, in_features, out_features, stream);
}
void linear_fwd(const float* x, const float* w, const float* bias, float* logits, int batch,
                int in_features, int out_features, hipStream_t stream) {
    linear_fwd_launch(x, w, bias, logits, batch, in_features, out_features, stream);
}

// dx[b][k] = sum_n dl[b][n] w[n][k]: one thread per (b, 8 k's).
__global__ __launch_bounds__(256) void linear_bwd_dx_kernel(
    const float* __restrict__ w, const float* __restrict__ dl, uint16_t* __restrict__ dx,
    int batch, int k, int nout) {
    const int kc = k / 8;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)batch * kc) return;
    const int b = (int)(idx / kc), k0 = (int)(idx % kc) * 8;
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int n = 0; n < nout; ++n) {
        const float d = dl[(size_t)b * nout + n];
        const float4 w0 = *(const float4*)(w + (size_t)n * k + k0);
        const float4 w1 = *(const float4*)(w + (size_t)n * k + k0 + 4);
        acc[0] += d * w0.x; acc[1] += d * w0.y; acc[2] += d * w0.z; acc[3] += d * w0.w;
        acc[4] += d * w1.x; acc[5] += d * w1.y; acc[6] += d * w1.z; acc[7] += d * w1.w;
    }
    *(uint4*)(dx + (size_t)b * k + k0) = pack8(acc);
}

// dw[n][k] += sum_b dl[b][n] x[b][k]; db[n] += sum_b dl[b][n].  Grid = (k / 64, batch / 32):
// a block owns 64 input features x 32 batch rows; thread = (feature, row lane), 16 output
// accumulators in registers, row lanes combined through LDS, one atomic per (n, k) per block.
__global__ __launch_bounds__(256) void linear_bwd_dw_kernel(
    const uint16_t* __restrict__ x, const float* __restrict__ dl, float* __restrict__ dw,
    float* __restrict__ db, int batch, int k, int nout) {
    __shared__ float red[4][64][17];
    const int kl = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int kk = blockIdx.x * 64 + kl;
    const int b0 = blockIdx.y * 32;
    float acc[16], accb[16];
#pragma unroll
    for (int n = 0; n < 16; ++n) acc[n] = accb[n] = 0.f;
    for (int b = b0 + rl; b < min(batch, b0 + 32); b += 4) {
        const float xv = kk < k ? bf16_bits_to_float(x[(size_t)b * k + kk]) : 0.f;
#pragma unroll
        for (int n = 0; n < 16; ++n) {
            const float d = n < nout ? dl[(size_t)b * nout + n] : 0.f;
            acc[n] += d * xv;
            accb[n] += d;
        }
    }
#pragma unroll
    for (int n = 0; n < 16; ++n) red[rl][kl][n] = acc[n];
    __syncthreads();
    for (int i = threadIdx.x; i < 64 * 16; i += 256) {
        const int n = i >> 6, kc = i & 63;
        if (n < nout && blockIdx.x * 64 + kc < k)
            atomicAdd(dw + (size_t)n * k + blockIdx.x * 64 + kc,
                      red[0][kc][n] + red[1][kc][n] + red[2][kc][n] + red[3][kc][n]);
    }
    if (blockIdx.x == 0 && db) {
        __syncthreads();
        if (kl == 0)
#pragma unroll
            for (int n = 0; n < 16; ++n) red[rl][0][n] = accb[n];
        __syncthreads();
        if (threadIdx.x < nout)
            atomicAdd(db + threadIdx.x, red[0][0][threadIdx.x] + red[1][0][threadIdx.x] +
                                            red[2][0][threadIdx.x] + red[3][0][threadIdx.x]);
    }
}

void linear_bwd(const bf16_raw* x, const float* w, const float* dlogits, float* dw, float* db,
                bf16_raw* dx, int batch, int in_features, int out_features, hipStream_t stream) {
    if (in_features % 8 != 0) throw std::invalid_argument("linear_bwd: in_features % 8 != 0");
    if (batch <= 0) return;
    if (dx) {
        const size_t total = (size_t)batch * (in_features / 8);
        hipLaunchKernelGGL(linear_bwd_dx_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256),
                           0, stream, w, dlogits, dx, batch, in_features, out_features);
    }
    if (dw) {
        if (out_features > 16) throw std::invalid_argument("linear_bwd: out_features > 16");
        hipLaunchKernelGGL(linear_bwd_dw_kernel, dim3((in_features + 63) / 64, (batch + 31) / 32),
                           dim3(256), 0, stream, x, dlogits, dw, db, batch, in_features,
                           out_features);
    }
}

// --------------------------------------------------------------- softmax cross-entropy
// One thread per row (classes <= 32): max, sum-exp, loss, gradient and arg-max in
// registers; the per-block loss / correct partials are wave-reduced, then one atomic each.
__global__ __launch_bounds__(256) void softmax_xent_kernel(
    const float* __restrict__ logits, const int64_t* __restrict__ target,
    float* __restrict__ loss_sum, float* __restrict__ dlogits, int32_t* __restrict__ correct,
    int batch, int classes, float inv_batch, float gscale) {
    const int row = blockIdx.x * blockDim.x + threadIdx.x;
    float loss = 0.f;
    int ok = 0;
    if (row < batch) {
        const float* l = logits + (size_t)row * classes;
        const int t = (int)target[row];
        float mx = l[0];
        int arg = 0;
        for (int c = 1; c < classes; ++c) {
            const float v = l[c];
            if (v > mx) { mx = v; arg = c; }
        }
        float se = 0.f;
        for (int c = 0; c < classes; ++c) se += __expf(l[c] - mx);
        const float lse = mx + __logf(se);
        loss = (lse - l[t]) * inv_batch;
        ok = (arg == t) ? 1 : 0;
        if (dlogits) {
            const float inv = 1.f / se;
            for (int c = 0; c < classes; ++c) {
                const float pr = __expf(l[c] - mx) * inv;
                dlogits[(size_t)row * classes + c] = (pr - (c == t ? 1.f : 0.f)) * inv_batch * gscale;
            }
        }
    }
    loss = wave_sum(loss);
    float okf = wave_sum((float)ok);
    if ((threadIdx.x & 63) == 0) {
        if (loss_sum) atomicAdd(loss_sum, loss);
        if (correct && okf != 0.f) atomicAdd(correct, (int)(okf + 0.5f));
    }
}

void softmax_xent(const float* logits, const int64_t* target, float* loss_sum, float* dlogits,
                  int32_t* correct, int batch, int classes, float grad_scale, hipStream_t stream) {
    if (classes < 1 || classes > 1024) throw std::invalid_argument("softmax_xent: bad class count");
    if (batch <= 0) return;
    hipLaunchKernelGGL(softmax_xent_kernel, dim3((batch + 255) / 256), dim3(256), 0, stream, logits,
                       target, loss_sum, dlogits, correct, batch, classes, 1.0f / (float)batch,
                       grad_scale);
}


// ------------------------------------------------ fused classifier step for training
// nn.Linear(512, 10) + nn.CrossEntropyLoss forward and backward in two launches
// (/root/reference/part1/model.py:40,45 and /root/reference/part1/main.py:43-44).
//   rows kernel : one wave per batch row, one 16-byte load of the row per lane - logits,
//                 softmax, the row's loss term, dlogits and dx (everything that needs only
//                 its own row)
//   grads kernel: dW = dlogits^T x, db, loss and accuracy - reductions over the batch; a
//                 block owns 32 feature columns, its 8 row groups are folded in LDS in a fixed order
// No atomics, no memsets: the result is bit-reproducible.
constexpr int kFcMaxClasses = 16;
constexpr int kFcCols = 32;          // feature columns per block of the grads kernel
constexpr int kFcGroups = 8;         // row groups per block (256 threads = 32 columns x 8 groups)

template <typename T>
__global__ __launch_bounds__(256) void fc_train_rows_kernel(
    const T* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
    const int64_t* __restrict__ target, float* __restrict__ logits, T* __restrict__ dx,
    float* __restrict__ dlogits, float* __restrict__ rowstat, int batch, int k_dim, int n,
    float inv_batch) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row = blockIdx.x * 4 + wave;
    if (row >= batch) return;
    const int kb = lane * 8;                 // k_dim == 512: one 16-byte chunk per lane
    float xv[8];
    load8(x + (size_t)row * k_dim + kb, xv);
    const int t = (int)target[row];
    float wreg[kFcMaxClasses][8];
#pragma unroll
    for (int c = 0; c < kFcMaxClasses; ++c)
        if (c < n) {
            const float4 a = *(const float4*)(w + (size_t)c * k_dim + kb);
            const float4 b = *(const float4*)(w + (size_t)c * k_dim + kb + 4);
            wreg[c][0] = a.x; wreg[c][1] = a.y; wreg[c][2] = a.z; wreg[c][3] = a.w;
            wreg[c][4] = b.x; wreg[c][5] = b.y; wreg[c][6] = b.z; wreg[c][7] = b.w;
        }
    float lg[kFcMaxClasses];
    float mx = -3.0e38f;
    int arg = 0;
#pragma unroll
    for (int c = 0; c < kFcMaxClasses; ++c) {
        lg[c] = 0.f;
        if (c < n) {
            float p = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) p += xv[i] * wreg[c][i];
            lg[c] = wave_sum_dpp(p) + bias[c];
            if (lg[c] > mx) { mx = lg[c]; arg = c; }
        }
    }
    float se = 0.f;
#pragma unroll
    for (int c = 0; c < kFcMaxClasses; ++c)
        if (c < n) se += __expf(lg[c] - mx);
    const float inv = 1.f / se;
    float lt = 0.f, o[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = 0.f;
#pragma unroll
    for (int c = 0; c < kFcMaxClasses; ++c) {
        float dl = 0.f;
        if (c < n) {
            if (c == t) lt = lg[c];
            dl = (__expf(lg[c] - mx) * inv - (c == t ? 1.f : 0.f)) * inv_batch;
#pragma unroll
            for (int i = 0; i < 8; ++i) o[i] += dl * wreg[c][i];
            if (lane == c) logits[(size_t)row * n + c] = lg[c];
        }
        if (lane == c) dlogits[(size_t)row * kFcMaxClasses + c] = dl;     // zero beyond n
    }
    if (lane == 0) {
        rowstat[(size_t)row * 2] = (mx + __logf(se) - lt) * inv_batch;
        rowstat[(size_t)row * 2 + 1] = (arg == t) ? 1.f : 0.f;
    }
    if (dx) store8(dx + (size_t)row * k_dim + kb, o);
}

template <typename T>
__global__ __launch_bounds__(256) void fc_train_grads_kernel(
    const T* __restrict__ x, const float* __restrict__ dlogits,
    const float* __restrict__ rowstat, float* __restrict__ dw, float* __restrict__ db,
    float* __restrict__ loss, int32_t* __restrict__ correct, int batch, int k_dim, int n) {
    // The kernel is a latency chain, not a bandwidth problem: 256 rows of its 32 columns and
    // of dlogits are staged in LDS with every global load issued up front (one round trip per
    // 256 rows), and the sums run out of LDS.
    constexpr int kTile = 256;
    // the staged x tile keeps the storage type; the fold scratch `red` reuses its bytes once the
    // last tile has been consumed
    __shared__ __attribute__((aligned(16))) char sx_bytes[kTile * kFcCols * 4];
    __shared__ __attribute__((aligned(16))) float sd[kTile][kFcMaxClasses];
    __shared__ float2 sr[kTile];
    static_assert(sizeof(float) * kFcGroups * kFcMaxClasses * (kFcCols + 1) <= sizeof(sx_bytes),
                  "fold scratch must fit the x tile");
    T (*sx)[kFcCols] = reinterpret_cast<T (*)[kFcCols]>(sx_bytes);
    float (*red)[kFcMaxClasses][kFcCols + 1] =
        reinterpret_cast<float (*)[kFcMaxClasses][kFcCols + 1]>(sx_bytes);
    const int col = threadIdx.x % kFcCols, grp = threadIdx.x / kFcCols;
    const int f0 = blockIdx.x * kFcCols;
    // batch sums of dlogits (db, block 0) and of the loss terms / hits (block 1) come out of
    // the same staged tiles: thread = (column c16, one of 16 row groups g16)
    const int c16 = threadIdx.x % 16, g16 = threadIdx.x / 16;
    float tail = 0.f;
    float acc[kFcMaxClasses];
#pragma unroll
    for (int c = 0; c < kFcMaxClasses; ++c) acc[c] = 0.f;
    for (int base = 0; base < batch; base += kTile) {
        Raw8<T> xr[4];
        float4 dr[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int id = threadIdx.x + 256 * i;
            const int rx = base + (id >> 2), rd = base + (id >> 2);
            xr[i] = rx < batch ? load_raw8(x + (size_t)rx * k_dim + f0 + (id & 3) * 8) : Raw8<T>();
            dr[i] = rd < batch ? *(const float4*)(dlogits + (size_t)rd * kFcMaxClasses + (id & 3) * 4)
                               : float4{0.f, 0.f, 0.f, 0.f};
        }
        const int rs = base + (int)threadIdx.x;
        const float2 st = rs < batch ? *(const float2*)(rowstat + (size_t)rs * 2) : float2{0.f, 0.f};
        if (base) __syncthreads();          // the previous tile has been consumed
        sr[threadIdx.x] = st;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int id = threadIdx.x + 256 * i;
            store_raw8(&sx[id >> 2][(id & 3) * 8], xr[i]);
            *(float4*)(&sd[id >> 2][(id & 3) * 4]) = dr[i];
        }
        __syncthreads();
#pragma unroll 4
        for (int r = grp; r < kTile; r += kFcGroups) {
            const float xv = elem_to_float(sx[r][col]);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 d = *(const float4*)(&sd[r][4 * q]);
                acc[4 * q + 0] += d.x * xv;
                acc[4 * q + 1] += d.y * xv;
                acc[4 * q + 2] += d.z * xv;
                acc[4 * q + 3] += d.w * xv;
            }
        }
        if (blockIdx.x == 0) {
            for (int r = g16; r < kTile; r += 16) tail += sd[r][c16];
        } else if (blockIdx.x == 1 && c16 < 2) {
            for (int r = g16; r < kTile; r += 16) tail += c16 ? sr[r].y : sr[r].x;
        }
    }
    __syncthreads();        // every thread is done with the staged tiles: `red` may take over
#pragma unroll
    for (int c = 0; c < kFcMaxClasses; ++c) red[grp][c][col] = acc[c];
    __syncthreads();
    for (int idx = threadIdx.x; idx < n * kFcCols; idx += 256) {
        const int c = idx / kFcCols, cc = idx % kFcCols;
        float s = 0.f;
#pragma unroll
        for (int g = 0; g < kFcGroups; ++g) s += red[g][c][cc];
        dw[(size_t)c * k_dim + f0 + cc] = s;
    }
    if (blockIdx.x < 2) {
        __syncthreads();
        const int width = blockIdx.x == 0 ? kFcMaxClasses : 2;
        float* flat = &red[0][0][0];
        flat[g16 * 16 + c16] = tail;
        __syncthreads();
        if ((int)threadIdx.x < width) {
            float tsum = 0.f;
#pragma unroll
            for (int q = 0; q < 16; ++q) tsum += flat[q * 16 + threadIdx.x];
            if (blockIdx.x == 0) {
                if ((int)threadIdx.x < n) db[threadIdx.x] = tsum;
            } else if (threadIdx.x == 0) {
                loss[0] = tsum;
            } else if (correct) {
                correct[0] = (int)(tsum + 0.5f);
            }
        }
    }
}

size_t fc_train_workspace_elems(int batch, int in_features, int out_features) {
    (void)in_features; (void)out_features;
    return (size_t)batch * (kFcMaxClasses + 2);      // dlogits [batch][16] + (loss term, hit) per row
}

namespace {
void fc_check(int batch, int in_features, int out_features) {
    if (in_features != 512)
        throw std::invalid_argument("fc_train: the fused classifier step handles 512 input features");
    if (out_features < 1 || out_features > kFcMaxClasses)
        throw std::invalid_argument("fc_train: 1..16 classes");
    (void)batch;
}
}  // namespace

namespace {
template <typename T>
void fc_rows_launch(const T* x, const float* w, const float* bias, const int64_t* target,
                    float* logits, T* dx, float* ws, int batch, int in_features, int out_features,
                    hipStream_t stream) {
    fc_check(batch, in_features, out_features);
    if (batch <= 0) return;
    float* dlogits = ws;
    float* rowstat = ws + (size_t)batch * kFcMaxClasses;
    hipLaunchKernelGGL(fc_train_rows_kernel<T>, dim3((batch + 3) / 4), dim3(256), 0, stream, x, w,
                       bias, target, logits, dx, dlogits, rowstat, batch, in_features, out_features,
                       1.0f / (float)batch);
}

template <typename T>
void fc_grads_launch(const T* x, const float* ws, float* loss, int32_t* correct, float* dw,
                     float* db, int batch, int in_features, int out_features, hipStream_t stream) {
    fc_check(batch, in_features, out_features);
    if (batch <= 0) return;
    const float* dlogits = ws;
    const float* rowstat = ws + (size_t)batch * kFcMaxClasses;
    hipLaunchKernelGGL(fc_train_grads_kernel<T>, dim3(in_features / kFcCols), dim3(256), 0, stream,
                       x, dlogits, rowstat, dw, db, loss, correct, batch, in_features, out_features);
}
}  // namespace

void fc_train_rows(const bf16_raw* x, const float* w, const float* bias, const int64_t* target,
                   float* logits, bf16_raw* dx, float* ws, int batch, int in_features,
                   int out_features, hipStream_t stream) {
    fc_rows_launch(x, w, bias, target, logits, dx, ws, batch, in_features, out_features, stream);
}
void fc_train_rows(const float* x, const float* w, const float* bias, const int64_t* target,
                   float* logits, float* dx, float* ws, int batch, int in_features,
                   int out_features, hipStream_t stream) {
    fc_rows_launch(x, w, bias, target, logits, dx, ws, batch, in_features, out_features, stream);
}

void fc_train_grads(const bf16_raw* x, const float* ws, float* loss, int32_t* correct, float* dw,
                    float* db, int batch, int in_features, int out_features, hipStream_t stream) {
    fc_grads_launch(x, ws, loss, correct, dw, db, batch, in_features, out_features, stream);
}
void fc_train_grads(const float* x, const float* ws, float* loss, int32_t* correct, float* dw,
                    float* db, int batch, int in_features, int out_features, hipStream_t stream) {
    fc_grads_launch(x, ws, loss, correct, dw, db, batch, in_features, out_features, stream);
}

void fc_train(const bf16_raw* x, const float* w, const float* bias, const int64_t* target,
              float* logits, float* loss, int32_t* correct, bf16_raw* dx, float* dw, float* db,
              float* ws, int batch, int in_features, int out_features, hipStream_t stream) {
    fc_train_rows(x, w, bias, target, logits, dx, ws, batch, in_features, out_features, stream);
    fc_train_grads(x, ws, loss, correct, dw, db, batch, in_features, out_features, stream);
}

}  // namespace dml
