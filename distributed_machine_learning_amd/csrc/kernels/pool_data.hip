// Max-pool 2x2 (NHWC bf16), column sums, data preparation and layout conversion.
//
// Reference call sites: nn.MaxPool2d(kernel_size=2, stride=2)
// (/root/reference/part1/model.py:16); the torchvision transform pipeline
// RandomCrop(32, padding=4) / RandomHorizontalFlip / ToTensor / Normalize
// (/root/reference/part1/main.py:84-93); the root-side gradient sum of the
// gather/scatter mode (/root/reference/part2/2a/main.py:104-107).
//
// All of these are memory-bound: every thread moves 16-byte vectors (8 bf16 channels).
#include <stdexcept>

#include "dml_column_reduce.h"
#include "dml_common.h"
#include "dml_kernels.h"

namespace dml {

__device__ __forceinline__ void max8(float (&m)[8], const float (&v)[8]) {
#pragma unroll
    for (int i = 0; i < 8; ++i) m[i] = fmaxf(m[i], v[i]);
}

// ---------------------------------------------------------------------- maxpool fwd
template <typename T>
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const T* __restrict__ x,
                                                          T* __restrict__ y, int hw, int c,
                                                          size_t total) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int c8 = c / 8, ohw = hw / 2;
    const int cc = (int)(idx % c8);
    size_t r = idx / c8;
    const int ow = (int)(r % ohw); r /= ohw;
    const int oh = (int)(r % ohw);
    const size_t b = r / ohw;
    const T* src = x + (((b * hw + 2 * oh) * hw + 2 * ow) * c) + cc * 8;
    float m[8], v[8];
    load8(src, m);
    load8(src + c, v); max8(m, v);
    load8(src + (size_t)hw * c, v); max8(m, v);
    load8(src + (size_t)hw * c + c, v); max8(m, v);
    store8(y + idx * 8, m);
}

namespace {
template <typename T>
void maxpool_fwd_launch(const T* x, T* y, int batch, int hw, int c, hipStream_t stream) {
    if (hw < 2 || (hw & 1) || c % 8) throw std::invalid_argument("maxpool2x2_fwd: bad shape");
    const size_t total = (size_t)batch * (hw / 2) * (hw / 2) * (c / 8);
    if (total == 0) return;
    hipLaunchKernelGGL(maxpool_fwd_kernel<T>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                       stream, x, y, hw, c, total);
}
}  // namespace

void maxpool2x2_fwd(const bf16_raw* x, bf16_raw* y, int batch, int hw, int c, hipStream_t stream) {
    maxpool_fwd_launch(x, y, batch, hw, c, stream);
}
void maxpool2x2_fwd(const float* x, float* y, int batch, int hw, int c, hipStream_t stream) {
    maxpool_fwd_launch(x, y, batch, hw, c, stream);
}

// ---------------------------------------------------------------------- maxpool bwd
// One thread per pooled position x 8 channels: re-derives the arg-max from the saved
// input (first maximum in row-major window order, like the index the forward of a
// framework pool records), writes all four input positions (zeros off the route).
template <typename T>
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const T* __restrict__ x,
                                                          const T* __restrict__ dy,
                                                          T* __restrict__ dx, int hw, int c,
                                                          size_t total, int relu_mask) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int c8 = c / 8, ohw = hw / 2;
    const int cc = (int)(idx % c8);
    size_t r = idx / c8;
    const int ow = (int)(r % ohw); r /= ohw;
    const int oh = (int)(r % ohw);
    const size_t b = r / ohw;
    const size_t base = (((b * hw + 2 * oh) * hw + 2 * ow) * c) + cc * 8;
    const size_t off[4] = {0, (size_t)c, (size_t)hw * c, (size_t)hw * c + c};
    float v[4][8], g[8];
#pragma unroll
    for (int q = 0; q < 4; ++q) load8(x + base + off[q], v[q]);
    load8(dy + idx * 8, g);
    float o[4][8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        int arg = 0;
        float m = v[0][i];
#pragma unroll
        for (int q = 1; q < 4; ++q)
            if (v[q][i] > m) { m = v[q][i]; arg = q; }
        const float gi = (relu_mask && !(m > 0.f)) ? 0.f : g[i];
#pragma unroll
        for (int q = 0; q < 4; ++q) o[q][i] = (q == arg) ? gi : 0.f;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) store8(dx + base + off[q], o[q]);
}

namespace {
template <typename T>
void maxpool_bwd_launch(const T* x, const T* dy, T* dx, int batch, int hw, int c, bool relu_mask,
                        hipStream_t stream) {
    if (hw < 2 || (hw & 1) || c % 8) throw std::invalid_argument("maxpool2x2_bwd: bad shape");
    const size_t total = (size_t)batch * (hw / 2) * (hw / 2) * (c / 8);
    if (total == 0) return;
    hipLaunchKernelGGL(maxpool_bwd_kernel<T>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                       stream, x, dy, dx, hw, c, total, relu_mask ? 1 : 0);
}
}  // namespace

void maxpool2x2_bwd(const bf16_raw* x, const bf16_raw* dy, bf16_raw* dx, int batch, int hw, int c,
                    bool relu_mask, hipStream_t stream) {
    maxpool_bwd_launch(x, dy, dx, batch, hw, c, relu_mask, stream);
}
void maxpool2x2_bwd(const float* x, const float* dy, float* dx, int batch, int hw, int c,
                    bool relu_mask, hipStream_t stream) {
    maxpool_bwd_launch(x, dy, dx, batch, hw, c, relu_mask, stream);
}

// ------------------------------------------------------------------------ ReLU backward
// y = mask > 0 ? x : 0 with `mask` the ReLU's output: nn.ReLU backward
// (/root/reference/part1/model.py:25) where no neighbouring kernel folds it (the engines fold it
// into the backward-data epilogue or the pool's routing).
template <typename T>
__global__ __launch_bounds__(256) void relu_mask_kernel(const T* __restrict__ x,
                                                        const T* __restrict__ mask,
                                                        T* __restrict__ y, size_t n8) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n8; i += stride) {
        float v[8], m[8];
        load8(x + i * 8, v);
        load8(mask + i * 8, m);
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = m[k] > 0.f ? v[k] : 0.f;
        store8(y + i * 8, v);
    }
}

namespace {
template <typename T>
void relu_mask_launch(const T* x, const T* mask, T* y, size_t n, hipStream_t stream) {
    if (n % 8) throw std::invalid_argument("relu_mask: n must be a multiple of 8");
    if (n == 0) return;
    size_t blocks = (n / 8 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(relu_mask_kernel<T>, dim3((unsigned)blocks), dim3(256), 0, stream, x, mask, y,
                       n / 8);
}
}  // namespace

void relu_mask(const bf16_raw* x, const bf16_raw* mask, bf16_raw* y, size_t n, hipStream_t stream) {
    relu_mask_launch(x, mask, y, n, stream);
}
void relu_mask_f32(const float* x, const float* mask, float* y, size_t n, hipStream_t stream) {
    relu_mask_launch(x, mask, y, n, stream);
}

// ----------------------------------------------------------------------- column sum
// out[c] += sum over rows (bias gradients): partial rows per block, then a small fold
// kernel (dml_column_reduce.h); no atomics on `out`.
template <typename T>
__global__ __launch_bounds__(256) void column_sum_kernel(const T* __restrict__ x,
                                                         float* __restrict__ ws, int rows, int c,
                                                         int rows_per_block) {
    __shared__ float red[4096];
    column_partials(
        [&](int r, int cg, float (&a)[8], float (&b)[8]) {
            float v[8];
            load8(x + (size_t)r * c + cg * 8, v);
#pragma unroll
            for (int i = 0; i < 8; ++i) a[i] += v[i];
        },
        ws, rows, c, rows_per_block, red);
}

__global__ __launch_bounds__(256) void column_sum_finish_kernel(const float* __restrict__ ws,
                                                                float* __restrict__ out, int nb,
                                                                int c) {
    column_fold([&](int col, float s1, float) { out[col] = s1; }, ws, nb, c);
}

size_t column_sum_workspace_elems() { return kColReduceWsFloats; }

namespace {
template <typename T>
void column_sum_launch(const T* x, float* out, int rows, int cols, float* ws, hipStream_t stream) {
    if (cols % 8 || cols < 8 || cols > 2048 || (256 % (cols / 8)) != 0)
        throw std::invalid_argument("column_sum: cols must be 8*2^k, <= 2048");
    if (!ws) throw std::invalid_argument("column_sum: workspace missing");
    if (rows <= 0) return;
    const ColReducePlan pl = plan_col_reduce(rows, cols);
    hipLaunchKernelGGL(column_sum_kernel<T>, dim3(pl.blocks), dim3(256), 0, stream, x, ws, rows,
                       cols, pl.rows_per_block);
    hipLaunchKernelGGL(column_sum_finish_kernel, dim3(fold_blocks(cols)), dim3(256), 0, stream,
                       (const float*)ws, out, pl.blocks, cols);
}
}  // namespace

void column_sum(const bf16_raw* x, float* out, int rows, int cols, float* ws, hipStream_t stream) {
    column_sum_launch(x, out, rows, cols, ws, stream);
}
void column_sum(const float* x, float* out, int rows, int cols, float* ws, hipStream_t stream) {
    column_sum_launch(x, out, rows, cols, ws, stream);
}

// ------------------------------------------------------------- augmentation + normalise
// One thread per output pixel: crop/flip source lookup on the uint8 HWC image, zero padding
// applied before normalisation, 8-byte bf16 NHWC4 store.
struct Norm3 { float mean[3]; float inv_std[3]; };

__device__ __forceinline__ void store_pixel4(uint16_t* p, float a, float b, float c) {
    uint2 o;
    o.x = pack2(a, b);
    o.y = pack2(c, 0.f);
    *(uint2*)p = o;
}
__device__ __forceinline__ void store_pixel4(float* p, float a, float b, float c) {
    *(float4*)p = float4{a, b, c, 0.f};
}

template <typename T>
__global__ __launch_bounds__(256) void augment_kernel(const uint8_t* __restrict__ img,
                                                      const int32_t* __restrict__ aug,
                                                      T* __restrict__ out, size_t total,
                                                      Norm3 nm) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int xo = (int)(idx & 31), yo = (int)((idx >> 5) & 31);
    const size_t b = idx >> 10;
    int sy = yo, sx = xo;
    if (aug) {
        const int oy = aug[b * 3], ox = aug[b * 3 + 1], fl = aug[b * 3 + 2];
        sy = yo + oy - 4;
        sx = (fl ? 31 - xo : xo) + ox - 4;
    }
    float v[3] = {0.f, 0.f, 0.f};
    if ((unsigned)sy < 32u && (unsigned)sx < 32u) {
        const uint8_t* p = img + ((b * 32 + sy) * 32 + sx) * 3;
        v[0] = p[0] * (1.f / 255.f); v[1] = p[1] * (1.f / 255.f); v[2] = p[2] * (1.f / 255.f);
    }
    store_pixel4(out + idx * 4, (v[0] - nm.mean[0]) * nm.inv_std[0],
                 (v[1] - nm.mean[1]) * nm.inv_std[1], (v[2] - nm.mean[2]) * nm.inv_std[2]);
}

namespace {
template <typename T>
void augment_launch(const uint8_t* images, const int32_t* aug, T* out, int batch,
                    const float mean[3], const float stdv[3], hipStream_t stream) {
    if (batch <= 0) return;
    Norm3 nm;
    for (int i = 0; i < 3; ++i) { nm.mean[i] = mean[i]; nm.inv_std[i] = 1.f / stdv[i]; }
    const size_t total = (size_t)batch * 1024;
    hipLaunchKernelGGL(augment_kernel<T>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                       stream, images, aug, out, total, nm);
}
}  // namespace

void augment_normalize(const uint8_t* images, const int32_t* aug, bf16_raw* out, int batch,
                       const float mean[3], const float stdv[3], hipStream_t stream) {
    augment_launch(images, aug, out, batch, mean, stdv, stream);
}
void augment_normalize(const uint8_t* images, const int32_t* aug, float* out, int batch,
                       const float mean[3], const float stdv[3], hipStream_t stream) {
    augment_launch(images, aug, out, batch, mean, stdv, stream);
}

template <typename T>
__global__ __launch_bounds__(256) void nchw_to_nhwc4_kernel(const float* __restrict__ x,
                                                            T* __restrict__ out, size_t total) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const size_t b = idx >> 10, pix = idx & 1023;
    const float* p = x + b * 3072 + pix;
    store_pixel4(out + idx * 4, p[0], p[1024], p[2048]);
}

void nchw_to_nhwc4(const float* x, bf16_raw* out, int batch, hipStream_t stream) {
    if (batch <= 0) return;
    const size_t total = (size_t)batch * 1024;
    hipLaunchKernelGGL(nchw_to_nhwc4_kernel<uint16_t>, dim3((unsigned)((total + 255) / 256)),
                       dim3(256), 0, stream, x, out, total);
}
void nchw_to_nhwc4(const float* x, float* out, int batch, hipStream_t stream) {
    if (batch <= 0) return;
    const size_t total = (size_t)batch * 1024;
    hipLaunchKernelGGL(nchw_to_nhwc4_kernel<float>, dim3((unsigned)((total + 255) / 256)),
                       dim3(256), 0, stream, x, out, total);
}

// Generic layout converters (tests, eager API boundary; not on the training hot path).
__global__ __launch_bounds__(256) void nchw_f32_to_nhwc_bf16_kernel(const float* __restrict__ x,
                                                                    uint16_t* __restrict__ out,
                                                                    int c, int hw2, size_t total) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;   // over NHWC elements
    if (idx >= total) return;
    const int ch = (int)(idx % c);
    const size_t r = idx / c;
    const int pix = (int)(r % hw2);
    const size_t b = r / hw2;
    out[idx] = float_to_bf16_bits(x[(b * c + ch) * hw2 + pix]);
}

__global__ __launch_bounds__(256) void nhwc_bf16_to_nchw_f32_kernel(const uint16_t* __restrict__ x,
                                                                    float* __restrict__ out,
                                                                    int c, int hw2, size_t total) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;   // over NCHW elements
    if (idx >= total) return;
    const int pix = (int)(idx % hw2);
    const size_t r = idx / hw2;
    const int ch = (int)(r % c);
    const size_t b = r / c;
    out[idx] = bf16_bits_to_float(x[(b * hw2 + pix) * c + ch]);
}

void nchw_f32_to_nhwc_bf16(const float* x, bf16_raw* out, int batch, int c, int hw, hipStream_t s) {
    const size_t total = (size_t)batch * c * hw * hw;
    if (total == 0) return;
    hipLaunchKernelGGL(nchw_f32_to_nhwc_bf16_kernel, dim3((unsigned)((total + 255) / 256)),
                       dim3(256), 0, s, x, out, c, hw * hw, total);
}

void nhwc_bf16_to_nchw_f32(const bf16_raw* x, float* out, int batch, int c, int hw, hipStream_t s) {
    const size_t total = (size_t)batch * c * hw * hw;
    if (total == 0) return;
    hipLaunchKernelGGL(nhwc_bf16_to_nchw_f32_kernel, dim3((unsigned)((total + 255) / 256)),
                       dim3(256), 0, s, x, out, c, hw * hw, total);
}

// ---------------------------------------------------------------- W-way slab reduction
__global__ __launch_bounds__(256) void sum_slabs_kernel(float4* __restrict__ slabs, int world,
                                                        size_t n4) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n4; i += stride) {
        float4 a = slabs[i];
        for (int r = 1; r < world; ++r) {
            const float4 v = slabs[(size_t)r * n4 + i];
            a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
        }
        slabs[i] = a;
    }
}

// dst[i] += slabs[0][i] + ... + slabs[k-1][i]: the root's own gradients are the first addend and
// stay where they are (same order of additions as sum_slabs on a landing area that holds the
// root's copy in slab 0)
__global__ __launch_bounds__(256) void sum_slabs_into_kernel(float* __restrict__ dst,
                                                             const float* __restrict__ slabs,
                                                             int k, size_t stride, size_t n) {
    const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t step = (size_t)gridDim.x * blockDim.x;
    const size_t n4 = n / 4;
    for (size_t i = tid; i < n4; i += step) {
        float4 a = ((const float4*)dst)[i];
        for (int r = 0; r < k; ++r) {
            const float4 v = ((const float4*)(slabs + (size_t)r * stride))[i];
            a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
        }
        ((float4*)dst)[i] = a;
    }
    for (size_t i = n4 * 4 + tid; i < n; i += step) {       // at most three elements
        float a = dst[i];
        for (int r = 0; r < k; ++r) a += slabs[(size_t)r * stride + i];
        dst[i] = a;
    }
}

void sum_slabs_into(float* dst, const float* slabs, int k, size_t stride, size_t n,
                    hipStream_t stream) {
    if (k <= 0 || n == 0) return;
    if (((uintptr_t)dst & 15) || ((uintptr_t)slabs & 15) || stride % 4)
        throw std::invalid_argument("sum_slabs_into: 16-byte aligned buffers, stride % 4 == 0");
    size_t blocks = (n / 4 + 255) / 256;
    if (blocks < 1) blocks = 1;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(sum_slabs_into_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, dst,
                       slabs, k, stride, n);
}

// Batched form for the wgrad partial results: job j owns blocks [block_start[j],
// block_start[j+1]), a thread sums one float4 column over the job's splits with eight loads in
// flight (the kernel is a pure stream: splits * count floats in, count out).
__global__ __launch_bounds__(256) void sum_slabs_batch_kernel(const float* __restrict__ src_base,
                                                              float* __restrict__ dst_base,
                                                              SlabSumJobs jobs) {
    int j = 0;
    while (j + 1 < jobs.n && blockIdx.x >= jobs.block_start[j + 1]) ++j;
    const size_t n4 = jobs.count[j] / 4;
    const size_t i = (size_t)(blockIdx.x - jobs.block_start[j]) * 256 + threadIdx.x;
    if (i >= n4) return;
    const float4* src = (const float4*)(src_base + jobs.src_off[j]) + i;
    const int splits = jobs.splits[j];
    float4 a = {0.f, 0.f, 0.f, 0.f};
    int k = 0;
    for (; k + 8 <= splits; k += 8) {
        float4 v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = src[(size_t)(k + q) * n4];
#pragma unroll
        for (int q = 0; q < 8; ++q) { a.x += v[q].x; a.y += v[q].y; a.z += v[q].z; a.w += v[q].w; }
    }
    for (; k < splits; ++k) {
        const float4 v = src[(size_t)k * n4];
        a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
    }
    ((float4*)(dst_base + jobs.dst_off[j]))[i] = a;
}

void sum_slabs_batch(const float* src_base, float* dst_base, SlabSumJobs jobs, hipStream_t stream) {
    if (jobs.n <= 0) return;
    if (jobs.n > SlabSumJobs::kMax) throw std::invalid_argument("sum_slabs_batch: too many jobs");
    unsigned blocks = 0;
    for (int j = 0; j < jobs.n; ++j) {
        if (jobs.count[j] % 4 || jobs.src_off[j] % 4 || jobs.dst_off[j] % 4 || jobs.splits[j] < 1)
            throw std::invalid_argument("sum_slabs_batch: counts and offsets must be multiples of 4");
        jobs.block_start[j] = blocks;
        blocks += (unsigned)((jobs.count[j] / 4 + 255) / 256);
    }
    jobs.block_start[jobs.n] = blocks;
    hipLaunchKernelGGL(sum_slabs_batch_kernel, dim3(blocks), dim3(256), 0, stream, src_base,
                       dst_base, jobs);
}

void sum_slabs(float* slabs, int world, size_t n, hipStream_t stream) {
    if (n % 4) throw std::invalid_argument("sum_slabs: n must be a multiple of 4");
    if (world <= 1 || n == 0) return;
    const size_t n4 = n / 4;
    size_t blocks = (n4 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(sum_slabs_kernel, dim3((unsigned)blocks), dim3(256), 0, stream,
                       (float4*)slabs, world, n4);
}

// ------------------------------------------------------------------- batched copies
// Several device-to-device copies in ONE launch: blockIdx.y picks the job, blockIdx.x strides
// over it.  The root of the gather / scatter mode copies its own 18 gradient tensors into and
// out of its landing slab each step; as 36 separate hipMemcpyAsync calls those were 36 copy
// kernels of ~2.6 us each on the compute stream.
__global__ __launch_bounds__(256) void copy_batch_kernel(CopyJobs jobs) {
    const int j = blockIdx.y;
    const size_t bytes = jobs.bytes[j];
    char* dst = (char*)jobs.dst[j];
    const char* src = (const char*)jobs.src[j];
    const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    if ((((uintptr_t)dst | (uintptr_t)src | bytes) & 15) == 0) {
        for (size_t i = tid; i < bytes / 16; i += stride) ((uint4*)dst)[i] = ((const uint4*)src)[i];
    } else if ((((uintptr_t)dst | (uintptr_t)src | bytes) & 3) == 0) {
        for (size_t i = tid; i < bytes / 4; i += stride) ((uint32_t*)dst)[i] = ((const uint32_t*)src)[i];
    } else {
        for (size_t i = tid; i < bytes; i += stride) dst[i] = src[i];
    }
}

void copy_batch(const CopyJobs& jobs, hipStream_t stream) {
    if (jobs.n <= 0) return;
    if (jobs.n > CopyJobs::kMax) throw std::invalid_argument("copy_batch: too many jobs");
    size_t largest = 0;
    for (int j = 0; j < jobs.n; ++j) {
        if (!jobs.dst[j] || !jobs.src[j]) throw std::invalid_argument("copy_batch: null pointer");
        if (jobs.bytes[j] > largest) largest = jobs.bytes[j];
    }
    if (largest == 0) return;
    // 16 bytes per thread and round; enough blocks for the largest job, small jobs end at once
    size_t bx = (largest / 16 + 255) / 256;
    if (bx < 1) bx = 1;
    if (bx > 128) bx = 128;
    hipLaunchKernelGGL(copy_batch_kernel, dim3((unsigned)bx, (unsigned)jobs.n), dim3(256), 0, stream,
                       jobs);
}

// ------------------------------------------------------------------- deterministic fill
__global__ __launch_bounds__(256) void fill_uniform_kernel(uint16_t* __restrict__ out, size_t n,
                                                           uint32_t seed, float scale) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        uint32_t h = (uint32_t)i * 0x9E3779B1u ^ seed;
        h ^= h >> 16; h *= 0x7feb352dU; h ^= h >> 15; h *= 0x846ca68bU; h ^= h >> 16;
        const float u = (float)(h >> 8) * (1.0f / 8388608.0f) - 1.0f;   // [-1, 1)
        out[i] = float_to_bf16_bits(u * scale);
    }
}

void fill_uniform_bf16(bf16_raw* out, size_t n, uint32_t seed, float scale, hipStream_t stream) {
    if (n == 0) return;
    size_t blocks = (n + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(fill_uniform_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, out, n,
                       seed, scale);
}

}  // namespace dml
