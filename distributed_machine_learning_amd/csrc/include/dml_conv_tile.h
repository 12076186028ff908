// Device-side pieces shared by the implicit-GEMM convolution kernels (conv_igemm.hip: streaming and
// first patch-resident form; conv_patch.hip: the patch-resident tiles with a row of taps per stage):
// the kernel argument block, the epilogue (bias / ReLU / ReLU mask, bf16 pack, 16-byte NHWC stores in
// the permuted channel order) and the BatchNorm backward sums an epilogue can leave behind.
//
// Reference op: nn.Conv2d(cin, cout, kernel_size=3, stride=1, padding=1, bias=True)
// (/root/reference/part1/model.py:18-23), BatchNorm / ReLU / max-pool behind it
// (/root/reference/part3/model.py:16,24-25).
#pragma once

#include "dml_column_reduce.h"
#include "dml_common.h"
#include "dml_kernels.h"

namespace dml {
namespace igemm {

struct IgemmArgs {
    const uint16_t* x;
    const uint16_t* w;
    uint16_t* y;
    const float* bias;
    const uint16_t* mask;
    float* slab;          // fp32 [splits][M][N] when splitting
    float* stats;         // optional per-tile column sums of the bias-free output, [tile_m][2][N]
    int M, N, C;          // pixels, output channels, reduction channels
    int hw, log_hw;       // spatial size
    int log_chunks;       // log2(C / 64)
    int tiles_n;
    int taps_per_split;
    int batch, img_blocks;   // position-major tiling: images, and tiles per spatial position
    unsigned x_bytes;     // extents for the buffer descriptors (out-of-range loads return 0)
    unsigned w_bytes;
    // optional BatchNorm+pool backward sums in the epilogue (see BnPoolReduce): the pooled
    // layer's conv output and constants; the partial rows go to `stats`
    const uint16_t* bn_pre;
    const float *bn_gamma, *bn_beta, *bn_mean, *bn_rstd;
    int bn_pooled;        // 0: the layer has no pool, bn_pre has this output's shape
    // optional stream signal (see StreamSignal): the first block stores `signal_value` here as
    // it starts, which tells a waiting queue that everything before this kernel has completed
    unsigned* signal;
    unsigned signal_value;
};

__device__ __forceinline__ void raise_stream_signal(const IgemmArgs& p) {
    if (p.signal && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0)
        __hip_atomic_store(p.signal, p.signal_value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
constexpr int kOutOfRange = (int)0x80000000;   // voffset beyond any tensor: hardware returns zeros

__device__ __forceinline__ float4 epilogue4(f32x4 a, int epi, const float* bias, int n,
                                            const uint16_t* mask, size_t off) {
    float4 v = {a[0], a[1], a[2], a[3]};
    if (epi == EPI_BIAS || epi == EPI_BIAS_RELU) {
        const float4 b = *(const float4*)(bias + n);
        v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
        if (epi == EPI_BIAS_RELU) {
            v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f);
            v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
        }
    } else if (epi == EPI_MASK) {
        const uint2 mk = *(const uint2*)(mask + off);
        // post-ReLU activations are >= 0: "passes" means a non-zero, positive bf16
        if (!(bf16_bits_to_float((uint16_t)(mk.x & 0xffff)) > 0.f)) v.x = 0.f;
        if (!(bf16_bits_to_float((uint16_t)(mk.x >> 16)) > 0.f)) v.y = 0.f;
        if (!(bf16_bits_to_float((uint16_t)(mk.y & 0xffff)) > 0.f)) v.z = 0.f;
        if (!(bf16_bits_to_float((uint16_t)(mk.y >> 16)) > 0.f)) v.w = 0.f;
    }
    return v;
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

typedef __attribute__((address_space(3))) void* lds_ptr;

// cache-policy bits of the weight loads (0 = default; marking them non-temporal to keep
// the pixel lines in L1 measured 1-5 % slower)
#ifndef DML_W_AUX
#define DML_W_AUX 0
#endif

// Output-channel order inside a wave's WTN-channel share of the weight tile.  An MFMA
// accumulator gives lane group lq = lane >> 4 rows 4*lq .. 4*lq+3 of every 16-row tile; with
// tile i / row q holding channel  lq*(4*TN) + 4*i + (q & 3)  those rows are 4*TN CONSECUTIVE
// channels, so a pixel's result leaves in 16-byte stores and the four lane groups complete
// whole 128-byte lines (with the natural order, row q = channel q, a lane's channels are 32
// bytes apart and every store instruction touches 16 rows x 4 separate 8-byte pieces).
// Only the global -> LDS row addresses and the epilogue's channel numbers know about it.
template <int WTN>
__device__ __forceinline__ int permuted_channel(int r) {
    constexpr int TN = WTN / 16;
    const int q = r & 15, i = (r % WTN) >> 4;
    return (r / WTN) * WTN + (q >> 2) * (4 * TN) + 4 * i + (q & 3);
}

// One pixel's share of a lane's accumulators -> memory: TN float4 (split-K slabs) or TN packed
// 4-channel groups, consecutive in memory (see permuted_channel), paired into 16-byte stores.
template <int TN, int EPI>
__device__ __forceinline__ void store_lane_channels(const f32x4 (&a)[TN], uint16_t* y,
                                                    const float* bias, const uint16_t* mask,
                                                    size_t off, int n) {
    uint32_t o[2 * TN];
#pragma unroll
    for (int i = 0; i < TN; ++i) {
        const float4 v = epilogue4(a[i], EPI, bias, n + 4 * i, mask, off + 4 * i);
        o[2 * i] = pack2(v.x, v.y);
        o[2 * i + 1] = pack2(v.z, v.w);
    }
    if constexpr (TN % 2 == 0) {
#pragma unroll
        for (int i = 0; i < TN; i += 2)
            *(uint4*)(y + off + 4 * i) = make_uint4(o[2 * i], o[2 * i + 1], o[2 * i + 2], o[2 * i + 3]);
    } else {
#pragma unroll
        for (int i = 0; i < TN; ++i) *(uint2*)(y + off + 4 * i) = make_uint2(o[2 * i], o[2 * i + 1]);
    }
}

// BatchNorm + ReLU + pool backward sums of one wave's accumulator tile (see BnPoolReduce): the
// tile holds the gradient of the POOLED activation; lane = pooled pixel m_first + 16*j and the
// 4*TN consecutive channels from nl.  Per element the pooling winner is re-derived from the
// layer's conv output (first maximum, strict comparisons, the fma of the forward pass) and the
// bf16-rounded gradient counts if the winner is positive.  Returns the sums over the tile's 16*TM
// pixels in every lane of a 16-lane row group: s1 = sum g, s2 = sum g * xh.
template <int TN, int TM>
__device__ __forceinline__ void bn_pool_bwd_tile_sums(const f32x4 (&acc)[TN][TM], const IgemmArgs& p,
                                                      int m_first, int nl, float (&s1)[TN][4],
                                                      float (&s2)[TN][4]) {
    const int hw = p.hw, in_hw = 2 * p.hw;
    float ysc[TN][4], ysh[TN][4], rs[TN][4], mrs[TN][4];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int col = nl + 4 * i + c;
            rs[i][c] = p.bn_rstd[col];
            mrs[i][c] = p.bn_mean[col] * rs[i][c];
            ysc[i][c] = p.bn_gamma[col] * rs[i][c];
            ysh[i][c] = __builtin_fmaf(-p.bn_gamma[col], mrs[i][c], p.bn_beta[col]);
            s1[i][c] = s2[i][c] = 0.f;
        }
#pragma unroll
    for (int j = 0; j < TM; ++j) {
        const int m = m_first + j * 16;          // pooled pixel
        if (m >= p.M) continue;
        const int ox = m & (hw - 1), oy = (m >> p.log_hw) & (hw - 1);
        const size_t b = (size_t)(m >> (2 * p.log_hw));
        const uint16_t* win = p.bn_pre + (((b * in_hw + 2 * oy) * in_hw + 2 * ox) * p.N + nl);
        // the lane's 4*TN consecutive channels at the four window positions, 16-byte loads
        uint32_t wv[4][2 * TN];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint16_t* wk = win + ((k >> 1) ? (size_t)in_hw * p.N : 0) + ((k & 1) ? p.N : 0);
            if constexpr (TN % 2 == 0) {
#pragma unroll
                for (int h = 0; h < TN / 2; ++h) {
                    const uint4 t = *(const uint4*)(wk + 8 * h);
                    wv[k][4 * h] = t.x; wv[k][4 * h + 1] = t.y;
                    wv[k][4 * h + 2] = t.z; wv[k][4 * h + 3] = t.w;
                }
            } else {
#pragma unroll
                for (int i = 0; i < TN; ++i) {
                    const uint2 t = *(const uint2*)(wk + 4 * i);
                    wv[k][2 * i] = t.x; wv[k][2 * i + 1] = t.y;
                }
            }
        }
#pragma unroll
        for (int i = 0; i < TN; ++i) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                float best = 0.f, xw = 0.f;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const uint32_t word = wv[k][2 * i + (c >> 1)];
                    const float v =
                        bf16_bits_to_float((uint16_t)((c & 1) ? (word >> 16) : (word & 0xffff)));
                    const float z = __builtin_fmaf(v, ysc[i][c], ysh[i][c]);
                    if (k == 0 || z > best) { best = z; xw = v; }
                }
                const float d = bf16_bits_to_float(float_to_bf16_bits(acc[i][j][c]));
                const float gi = best > 0.f ? d : 0.f;
                s1[i][c] += gi;
                s2[i][c] += gi * __builtin_fmaf(xw, rs[i][c], -mrs[i][c]);
            }
        }
    }
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            s1[i][c] = row16_sum(s1[i][c]);
            s2[i][c] = row16_sum(s2[i][c]);
        }
}

// The same sums for a layer WITHOUT a pool behind its ReLU: the tile holds the gradient of the
// layer's activation itself, element (m, col) pairs with conv output element (m, col); the mask is
// forward's ReLU decision re-derived with forward's scale / shift (as in bn_bwd_reduce_kernel).
// The conv output values of the tile are requested a few K steps before the loop ends
// (bn_plain_bwd_prefetch), so their round trip is over when the accumulators are.
template <int TN, int TM>
__device__ __forceinline__ void bn_plain_bwd_prefetch(const IgemmArgs& p, int m_first, int nl,
                                                      uint32_t (&xv)[TM][2 * TN]) {
#pragma unroll
    for (int j = 0; j < TM; ++j) {
        const int m = m_first + j * 16;
        const bool in = m < p.M;
        const uint16_t* px = p.bn_pre + ((size_t)(in ? m : 0) * p.N + nl);
        if constexpr (TN % 2 == 0) {
#pragma unroll
            for (int h = 0; h < TN / 2; ++h) {
                const uint4 t = *(const uint4*)(px + 8 * h);
                xv[j][4 * h] = t.x; xv[j][4 * h + 1] = t.y;
                xv[j][4 * h + 2] = t.z; xv[j][4 * h + 3] = t.w;
            }
        } else {
#pragma unroll
            for (int i = 0; i < TN; ++i) {
                const uint2 t = *(const uint2*)(px + 4 * i);
                xv[j][2 * i] = t.x; xv[j][2 * i + 1] = t.y;
            }
        }
    }
}

template <int TN, int TM>
__device__ __forceinline__ void bn_plain_bwd_tile_sums(const f32x4 (&acc)[TN][TM], const IgemmArgs& p,
                                                       const uint32_t (&xv)[TM][2 * TN], int m_first,
                                                       int nl, float (&s1)[TN][4], float (&s2)[TN][4]) {
    float ysc[TN][4], ysh[TN][4], rs[TN][4], mrs[TN][4];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int col = nl + 4 * i + c;
            rs[i][c] = p.bn_rstd[col];
            mrs[i][c] = p.bn_mean[col] * rs[i][c];
            ysc[i][c] = p.bn_gamma[col] * rs[i][c];
            ysh[i][c] = p.bn_beta[col] - p.bn_mean[col] * ysc[i][c];
            s1[i][c] = s2[i][c] = 0.f;
        }
#pragma unroll
    for (int j = 0; j < TM; ++j) {
        if (m_first + j * 16 >= p.M) continue;      // (the row loaded in its place adds nothing)
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const uint32_t word = xv[j][2 * i + (c >> 1)];
                const float v = bf16_bits_to_float((uint16_t)((c & 1) ? (word >> 16) : (word & 0xffff)));
                const float d = bf16_bits_to_float(float_to_bf16_bits(acc[i][j][c]));
                const float gi = (__builtin_fmaf(v, ysc[i][c], ysh[i][c]) > 0.f) ? d : 0.f;
                s1[i][c] += gi;
                s2[i][c] += gi * (v * rs[i][c] - mrs[i][c]);
            }
    }
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            s1[i][c] = row16_sum(s1[i][c]);
            s2[i][c] = row16_sum(s2[i][c]);
        }
}

// The patch-resident tall-tile kernels (conv_patch.hip): BM x 64 tiles, BM = 256 or 128, eight
// waves, hw 2 .. 16.  variant 0: the shipped wave layout of the tile height, 1: the other layout,
// 2 .. 4: ablation builds for timing only, 5 (128 rows): four waves.
bool conv3x3_pr_supported(int bm, int bn, int hw, int variant);
// splits > 1 (a power of two dividing cin / 64): the channel chunks are divided between that many
// blocks per tile, each storing an fp32 slab [M][N] at a.slab + split * M * N (plain epilogue).
void launch_pr(const IgemmArgs& a, int bm, int bn, int variant, int epi, int splits,
               hipStream_t stream);

}  // namespace igemm
}  // namespace dml
