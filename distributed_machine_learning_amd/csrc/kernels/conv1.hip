// First layer: conv3x3 with 3 input channels (forward and backward-weight).
//
// Reference op: the first nn.Conv2d(3, 64, 3, padding=1) of the Sequential
// (/root/reference/part1/model.py:13,18-23).  Its input needs no gradient, so there is
// no backward-data kernel.
//
// With C = 3 the reduction is only 27 long and the layer is bound by the 64-channel
// 32x32 output (the largest activation of the network), not by math.  The input is kept
// as NHWC with the channel padded to 4 (one 8-byte load per tap); the im2col row of a
// pixel is k = tap*4 + c for 9 taps (36 values) padded with zeros to K = 64, built by the
// thread that owns the pixel and written to LDS once.  The product itself runs on the
// matrix cores (v_mfma_f32_16x16x32_bf16 forward, 32x32x16 with transposing LDS reads
// backward) so the VALU is left for the gather and the epilogue.
//
// Backward-weight trick: im2col column k = 36 is set to 1.0, so the same MFMA that
// produces dW[co][k] yields the bias gradient sum_p dy[p][co] in column 36 for free.
#include <cstdlib>
#include <cstring>
#include <stdexcept>

#include "dml_common.h"
#include "dml_kernels.h"

namespace dml {

namespace {

typedef __attribute__((address_space(3))) s16x4 lds_s16x4_c1;

}  // namespace

// Output channel held in accumulator element c of MFMA tile i by lane group lq (rows 4*lq ..
// 4*lq+3 of every 16-row tile).  Tiles 0/1 give a lane the eight consecutive channels 8*lq ..
// 8*lq+7 and tiles 2/3 the eight channels 32 + 8*lq ..: each of the two 16-byte stores of a
// pixel then lands next to its neighbours' - the four lane groups write one contiguous 64-byte
// half of the pixel's 128-byte row per store instruction.
// (HALVES = false: the earlier order, 16 consecutive channels per lane group, whose two stores
// interleave 16-byte pieces of the row; kept for comparison, DML_CONV1_STORE=interleaved)
template <bool HALVES>
__device__ __forceinline__ int conv1_out_channel(int i, int lq, int c) {
    return HALVES ? (i >> 1) * 32 + lq * 8 + (i & 1) * 4 + c : lq * 16 + i * 4 + c;
}

// ------------------------------------------------------------------------- forward
// Block: 256 pixels x 64 output channels, K = 64 in one step.  Wave w owns pixels
// [64w, 64w+64).  Accumulator layout D[co][pixel] as in the generic kernel.
template <bool HALVES>
__global__ __launch_bounds__(256) void conv1_fprop_kernel(const uint16_t* __restrict__ x,
                                                          const uint16_t* __restrict__ w1,
                                                          const float* __restrict__ bias,
                                                          uint16_t* __restrict__ y, int M,
                                                          int relu, float* __restrict__ stats) {
    __shared__ __attribute__((aligned(16))) char smem[(256 + 64) * 128];
    char* sA = smem;                 // [256 px][64 k]
    char* sB = smem + 256 * 128;     // [64 co][64 k]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int pix = blockIdx.x * 256 + tid;
    const bool row_ok = pix < M;
    const int w_ = pix & 31, h_ = (pix >> 5) & 31;
    const int sw_row = (tid >> 1) & 7;
    // nine 8-byte tap loads through a raw buffer descriptor: out-of-image taps get an
    // out-of-range offset and come back as zeros, so all nine are in flight together
    const __amdgpu_buffer_rsrc_t xr =
        __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, (unsigned)M * 8u, 0x00020000);
    typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
    u32x2 taps[10];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const int dh = t / 3 - 1, dw = t % 3 - 1;
        const bool ok = row_ok && (unsigned)(h_ + dh) < 32u && (unsigned)(w_ + dw) < 32u;
        taps[t] = __builtin_amdgcn_raw_buffer_load_b64(
            xr, ok ? (pix + dh * 32 + dw) * 8 : (int)0x80000000, 0, 0);
    }
    taps[9] = u32x2{0u, 0u};
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        uint4 v = make_uint4(0, 0, 0, 0);
        if (c < 5) v = make_uint4(taps[2 * c][0], taps[2 * c][1], taps[2 * c + 1][0], taps[2 * c + 1][1]);
        *(uint4*)(sA + tid * 128 + ((c ^ sw_row) << 4)) = v;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int chunk = tid + 256 * i;
        const int r = chunk >> 3, c = chunk & 7;
        // output-channel order of the four 16-row MFMA tiles: see conv1_out_channel
        const int ch = conv1_out_channel<HALVES>(r >> 4, (r & 15) >> 2, r & 3);
        *(uint4*)(sB + r * 128 + ((c ^ ((r >> 1) & 7)) << 4)) = *(const uint4*)(w1 + ch * 64 + c * 8);
    }
    __syncthreads();

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int l15 = lane & 15, lq = lane >> 4, sw = l15 >> 1;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        const int koff = ((kk * 4 + lq) ^ sw) << 4;
        bf16x8 wf[4], xf[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) wf[i] = *(const bf16x8*)(sB + (i * 16 + l15) * 128 + koff);
#pragma unroll
        for (int j = 0; j < 4; ++j)
            xf[j] = *(const bf16x8*)(sA + (wave * 64 + j * 16 + l15) * 128 + koff);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i], xf[j], acc[i][j], 0, 0, 0);
    }
    // optional BatchNorm statistics: column sums / sums of squares of the bias-free output
    // of this block's 256 pixels (rows past the end multiplied zeros), one row per block
    if (stats) {
        float s1[4][4], s2[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                float a = 0.f, b = 0.f;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float v = acc[i][j][c];
                    a += v;
                    b += v * v;
                }
                s1[i][c] = row16_sum(a);
                s2[i][c] = row16_sum(b);
            }
        __syncthreads();
        float* red = (float*)smem;      // [4 waves][2][64]
        if (l15 == 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    red[(wave * 2 + 0) * 64 + conv1_out_channel<HALVES>(i, lq, c)] = s1[i][c];
                    red[(wave * 2 + 1) * 64 + conv1_out_channel<HALVES>(i, lq, c)] = s2[i][c];
                }
        }
        __syncthreads();
        if (tid < 128) {
            const int which = tid >> 6, col = tid & 63;
            stats[((size_t)blockIdx.x * 2 + which) * 64 + col] =
                red[(0 * 2 + which) * 64 + col] + red[(1 * 2 + which) * 64 + col] +
                red[(2 * 2 + which) * 64 + col] + red[(3 * 2 + which) * 64 + col];
        }
    }
    float bv[16];
#pragma unroll
    for (int q = 0; q < 4; ++q) {       // (tile q's four channels are consecutive)
        const float4 b = *(const float4*)(bias + conv1_out_channel<HALVES>(q, lq, 0));
        bv[q * 4 + 0] = b.x; bv[q * 4 + 1] = b.y; bv[q * 4 + 2] = b.z; bv[q * 4 + 3] = b.w;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int m = blockIdx.x * 256 + wave * 64 + j * 16 + l15;
        if (m >= M) continue;
        uint32_t o[8];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float v0 = acc[i][j][0] + bv[i * 4 + 0], v1 = acc[i][j][1] + bv[i * 4 + 1],
                  v2 = acc[i][j][2] + bv[i * 4 + 2], v3 = acc[i][j][3] + bv[i * 4 + 3];
            if (relu) {
                v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); v2 = fmaxf(v2, 0.f); v3 = fmaxf(v3, 0.f);
            }
            o[2 * i] = pack2(v0, v1);
            o[2 * i + 1] = pack2(v2, v3);
        }
        uint16_t* dst = y + (size_t)m * 64 + (HALVES ? lq * 8 : lq * 16);
        *(uint4*)dst = make_uint4(o[0], o[1], o[2], o[3]);           // channels 8*lq ..
        *(uint4*)(dst + (HALVES ? 32 : 8)) = make_uint4(o[4], o[5], o[6], o[7]);    // 32 + 8*lq ..
    }
}

int conv1_fprop_stats_rows(int batch) { return (batch * 1024 + 255) / 256; }

void conv1_fprop(const bf16_raw* x, const bf16_raw* w1, const float* bias, bf16_raw* y, int batch,
                 bool relu, hipStream_t stream, float* stats) {
    if (batch <= 0) return;
    if ((long long)batch * 1024 * 128 >= (1LL << 31))
        throw std::invalid_argument("conv1_fprop: batch too large");
    const int M = batch * 1024;
    static const bool halves = !(getenv("DML_CONV1_STORE") &&
                                 !strcmp(getenv("DML_CONV1_STORE"), "interleaved"));
    if (halves)
        hipLaunchKernelGGL(conv1_fprop_kernel<true>, dim3((M + 255) / 256), dim3(256), 0, stream, x,
                           w1, bias, y, M, relu ? 1 : 0, stats);
    else
        hipLaunchKernelGGL(conv1_fprop_kernel<false>, dim3((M + 255) / 256), dim3(256), 0, stream, x,
                           w1, bias, y, M, relu ? 1 : 0, stats);
}

// ------------------------------------------------------------------ backward-weight
// D[co][k] = sum_p dy[p][co] * col[p][k] over this block's pixel range; 4 waves = the four
// 32x32 tiles of the 64x64 result.  The kernel is bound by streaming dy (the largest
// gradient tensor of the network), so it is organised for bytes in flight: 128-pixel steps,
// all loads of step s+1 issued (raw buffer loads; out-of-image taps and past-the-end rows
// return zeros from the range check, no branches) before the MFMAs of step s, one LDS stage.
// Row stride 192 B keeps the transposing reads conflict-free.  Results are added to
// dw[64][9][3] / db[64] with fp32 atomics.
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4c1;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2c1;

// FUSED: the layer is followed by BatchNorm + ReLU + 2x2 max-pool and `dy` is the gradient of
// the POOLED output.  dL/d(conv output) is then produced on the fly, exactly as
// bn_pool_bwd_apply_kernel would have stored it (same formula, same bf16 rounding, so the
// result is bit-identical): a 128-pixel step is four image rows = 32 complete pooling windows,
// thread = (window, 8 channels) re-derives the window's winner from the conv output `pre`
// and writes the four pixels' gradients straight into the LDS tile.  The 33.5 MB gradient
// tensor of the first layer (batch 256) is never written or read back.
struct Conv1BnPool {
    const uint16_t* pre;        // conv output [M][64]
    const float *gamma, *beta, *mean, *rstd, *dgamma, *dbeta;
};

// MODE 0: dy is dL/d(conv output).  MODE 1: BatchNorm + ReLU + pool follow (see above).
// MODE 2: ReLU + pool follow, `bn.pre` is the stored ReLU output: the pooled gradient goes to
// the window's first maximum if that is positive (what maxpool2x2_bwd with relu_mask stores).
template <int MODE>
__global__ __launch_bounds__(256) void conv1_wgrad_kernel(const uint16_t* __restrict__ x,
                                                          const uint16_t* __restrict__ dy,
                                                          float* __restrict__ slab, int M,
                                                          int steps_per_block, Conv1BnPool bn) {
    constexpr bool FUSED = MODE != 0;
    constexpr int S = 192;
    constexpr int PX = 128;                      // pixels per step
    constexpr int kOOB = (int)0x80000000;
    __shared__ __attribute__((aligned(16))) char smem[2 * PX * S];
    char* sD = smem;               // dy tile  [128 px][64 co]
    char* sC = smem + PX * S;      // im2col   [128 px][64 k]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tco = wave & 1, tk = wave >> 1;
    const __amdgpu_buffer_rsrc_t xr =
        __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, (unsigned)M * 8u, 0x00020000);
    const __amdgpu_buffer_rsrc_t dr = __builtin_amdgcn_make_buffer_rsrc(
        (void*)dy, 0, FUSED ? (unsigned)(M / 4) * 128u : (unsigned)M * 128u, 0x00020000);
    const __amdgpu_buffer_rsrc_t pr = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(FUSED ? bn.pre : dy), 0, (unsigned)M * 128u, 0x00020000);
    // FUSED: thread = (window widx of the step's 32, channel group cgr); per-channel constants
    //   y = x*ysc + ysh (winner / live test),  dx = [winner] k1*g - k2 - k3*x
    const int widx = tid >> 3, cgr = tid & 7;
    float ysc[8], ysh[8], k1[8], k2[8], k3[8];
    if (MODE == 1) {
        const float inv_n = 1.f / (float)M;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int col = cgr * 8 + k;
            const float rs = bn.rstd[col];
            const float mrs = bn.mean[col] * rs;
            ysc[k] = bn.gamma[col] * rs;
            ysh[k] = __builtin_fmaf(-bn.gamma[col], mrs, bn.beta[col]);
            k1[k] = bn.gamma[col] * rs;
            const float dg = bn.dgamma[col] * inv_n, db = bn.dbeta[col] * inv_n;
            k2[k] = k1[k] * __builtin_fmaf(-mrs, dg, db);
            k3[k] = (k1[k] * rs) * dg;
        }
    }

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;

    const int g = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3;
    const int frag_row = 8 * (g >> 1) + q, frag_col = 16 * (g & 1) + 4 * pp;
    const int a_off = frag_row * S + (tco * 32 + frag_col) * 2;
    const int b_off = frag_row * S + (tk * 32 + frag_col) * 2;

    const int total_steps = (M + PX - 1) / PX;
    const int s0 = blockIdx.x * steps_per_block;
    const int s1 = min(total_steps, s0 + steps_per_block);
    if (s0 >= s1) return;

    // im2col staging: thread = (pixel row, half); half 0 owns taps 0..4, half 1 taps 5..8 and
    // the constant column k = 36.  Chunks 5..7 of every row (k >= 40) stay zero.
    const int crow = tid >> 1, chalf = tid & 1;
    if (chalf) {
#pragma unroll
        for (int c = 5; c < 8; ++c) *(uint4*)(sC + crow * S + c * 16) = make_uint4(0, 0, 0, 0);
    }
    u32x4c1 rd[4], rg;
    u32x2c1 rt[5];
    // local pixel rows of this thread's window: top-left, +1, +32, +33
    const int wrow0 = (widx >> 4) * 64 + (widx & 15) * 2;
    auto load_step = [&](int s) {
        const int pbase = s * PX;
        if (FUSED) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int pix = pbase + wrow0 + (i >> 1) * 32 + (i & 1);
                rd[i] = __builtin_amdgcn_raw_buffer_load_b128(
                    pr, pix < M ? pix * 128 + cgr * 16 : kOOB, 0, 0);
            }
            const int opix = s * 32 + widx;         // pooled pixel of this window
            rg = __builtin_amdgcn_raw_buffer_load_b128(
                dr, opix < M / 4 ? opix * 128 + cgr * 16 : kOOB, 0, 0);
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int chunk = tid + 256 * i;
                const int pix = pbase + (chunk >> 3);
                rd[i] = __builtin_amdgcn_raw_buffer_load_b128(
                    dr, pix < M ? pix * 128 + (chunk & 7) * 16 : kOOB, 0, 0);
            }
        }
        const int pix = pbase + crow;
        const int w_ = pix & 31, h_ = (pix >> 5) & 31;
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            const int tap = chalf * 5 + j;
            const int dh = tap / 3 - 1, dw_ = tap % 3 - 1;
            const bool ok = pix < M && tap < 9 && (unsigned)(h_ + dh) < 32u &&
                            (unsigned)(w_ + dw_) < 32u;
            rt[j] = __builtin_amdgcn_raw_buffer_load_b64(xr, ok ? (pix + dh * 32 + dw_) * 8 : kOOB,
                                                         0, 0);
        }
        if (chalf && pix < M) rt[4][0] = 0x3f80u;     // tap slot 9 = k 36: bf16 1.0 (bias column)
    };
    auto store_step = [&]() {
        if (FUSED) {
            float v[4][8], g[8], o[4][8];
#pragma unroll
            for (int i = 0; i < 4; ++i)
                unpack8(make_uint4(rd[i][0], rd[i][1], rd[i][2], rd[i][3]), v[i]);
            unpack8(make_uint4(rg[0], rg[1], rg[2], rg[3]), g);
            if (MODE == 2) {
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    int w = 0;
                    float m = v[0][k];
#pragma unroll
                    for (int i = 1; i < 4; ++i)
                        if (v[i][k] > m) { m = v[i][k]; w = i; }
                    const float gk = m > 0.f ? g[k] : 0.f;
#pragma unroll
                    for (int i = 0; i < 4; ++i) o[i][k] = w == i ? gk : 0.f;
                }
            } else {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                // first maximum in row-major window order, strict comparisons, same fma as forward
                int w = 0;
                float m = __builtin_fmaf(v[0][k], ysc[k], ysh[k]);
#pragma unroll
                for (int i = 1; i < 4; ++i) {
                    const float z = __builtin_fmaf(v[i][k], ysc[k], ysh[k]);
                    if (z > m) { m = z; w = i; }
                }
                const float gk = m > 0.f ? k1[k] * g[k] : 0.f;
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    o[i][k] = __builtin_fmaf(-k3[k], v[i][k], (w == i ? gk : 0.f) - k2[k]);
            }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const uint4 pk = pack8(o[i]);
                *(uint4*)(sD + (wrow0 + (i >> 1) * 32 + (i & 1)) * S + cgr * 16) = pk;
            }
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int chunk = tid + 256 * i;
                *(u32x4c1*)(sD + (chunk >> 3) * S + (chunk & 7) * 16) = rd[i];
            }
        }
        // half 0: k 0..19 (taps 0..4), half 1: k 20..39 (taps 5..8 + bias column): 40 bytes each
        char* dst = sC + crow * S + chalf * 40;
#pragma unroll
        for (int j = 0; j < 5; ++j) *(u32x2c1*)(dst + j * 8) = rt[j];
    };

    load_step(s0);
    store_step();
    __syncthreads();
    for (int s = s0; s < s1; ++s) {
        const bool more = s + 1 < s1;
        if (more) load_step(s + 1);
#pragma unroll
        for (int ks = 0; ks < PX / 16; ++ks) {
            const char* pa = sD + a_off + ks * 16 * S;
            const char* pb = sC + b_off + ks * 16 * S;
            const s16x4 alo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_c1*)pa);
            const s16x4 ahi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_c1*)(pa + 4 * S));
            const s16x4 blo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_c1*)pb);
            const s16x4 bhi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_c1*)(pb + 4 * S));
            const s16x8 av = {alo[0], alo[1], alo[2], alo[3], ahi[0], ahi[1], ahi[2], ahi[3]};
            const s16x8 bv = {blo[0], blo[1], blo[2], blo[3], bhi[0], bhi[1], bhi[2], bhi[3]};
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, av),
                                                          __builtin_bit_cast(bf16x8, bv), acc, 0, 0,
                                                          0);
        }
        __syncthreads();
        if (more) {
            store_step();
            __syncthreads();
        }
    }
    // D[co][k]: lane -> k = tk*32 + (l & 31); register r -> co.  Each block stores its
    // partial [64 co][40 k] tile; conv1_wgrad_finish_kernel folds the blocks (all blocks
    // adding into the same 1792 addresses with atomics would serialise at the memory side).
    const int k = tk * 32 + (lane & 31), lh = lane >> 5;
    if (k < 40) {
        float* dst = slab + (size_t)blockIdx.x * (64 * 40) + k;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = tco * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            dst[co * 40] = acc[r];
        }
    }
}

// dw[co][tap][c] = sum_blocks slab[b][co][tap*4+c] (c < 3), db[co] = sum_blocks slab[b][co][36].
// A latency chain, not a bandwidth problem (5 MB): 160 blocks of 16 outputs x 16 parts; a part
// folds every 16th slab row with up to 32 independent loads in flight, the parts meet in LDS.
__global__ __launch_bounds__(256) void conv1_wgrad_finish_kernel(const float* __restrict__ slab,
                                                                 float* __restrict__ dw,
                                                                 float* __restrict__ db,
                                                                 int nblocks) {
    __shared__ float red[16][17];
    const int col = threadIdx.x & 15, part = threadIdx.x >> 4;
    const int o = blockIdx.x * 16 + col;                     // 0 .. 2559 = co * 40 + k
    float s = 0.f;
    int b = part;
    // (512 slabs at batch 256: one round trip of 32 loads per thread instead of four of eight -
    // the kernel sits on the tail of backward's critical path)
    for (; b + 31 * 16 < nblocks; b += 32 * 16) {
        float v[32];
#pragma unroll
        for (int u = 0; u < 32; ++u) v[u] = slab[(size_t)(b + 16 * u) * 2560 + o];
#pragma unroll
        for (int u = 0; u < 32; ++u) s += v[u];
    }
    for (; b + 7 * 16 < nblocks; b += 8 * 16) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = slab[(size_t)(b + 16 * u) * 2560 + o];
#pragma unroll
        for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; b < nblocks; b += 16) s += slab[(size_t)b * 2560 + o];
    red[part][col] = s;
    __syncthreads();
    if (part == 0) {
        float v = 0.f;
#pragma unroll
        for (int q = 0; q < 16; ++q) v += red[q][col];
        const int co = o / 40, k = o % 40;
        const int tap = k >> 2, c = k & 3;
        if (tap < 9 && c < 3) dw[co * 27 + tap * 3 + c] = v;
        else if (k == 36 && db) db[co] = v;
    }
}

namespace {
int conv1_wgrad_blocks(int M, int* steps_per_block) {
    const int total_steps = (M + 127) / 128;
    int blocks = total_steps < 512 ? total_steps : 512;
    const int spb = (total_steps + blocks - 1) / blocks;
    *steps_per_block = spb;
    return (total_steps + spb - 1) / spb;
}
}  // namespace

size_t conv1_wgrad_workspace_elems(int batch) {
    int spb;
    return (size_t)conv1_wgrad_blocks(batch * 1024, &spb) * 2560;
}

void conv1_wgrad(const bf16_raw* x, const bf16_raw* dy, float* dw, float* db, int batch,
                 float* ws, hipStream_t stream) {
    if (batch <= 0) return;
    if ((long long)batch * 1024 * 128 >= (1LL << 31))
        throw std::invalid_argument("conv1_wgrad: batch too large");
    if (!ws) throw std::invalid_argument("conv1_wgrad: workspace missing");
    const int M = batch * 1024;
    int spb;
    const int blocks = conv1_wgrad_blocks(M, &spb);
    hipLaunchKernelGGL(conv1_wgrad_kernel<0>, dim3(blocks), dim3(256), 0, stream, x, dy, ws, M,
                       spb, Conv1BnPool{});
    hipLaunchKernelGGL(conv1_wgrad_finish_kernel, dim3(160), dim3(256), 0, stream,
                       (const float*)ws, dw, db, blocks);
}

void conv1_wgrad_bn_pool(const bf16_raw* x, const bf16_raw* pre, const bf16_raw* dy_pooled,
                         const float* gamma, const float* beta, const float* save_mean,
                         const float* save_rstd, const float* dgamma, const float* dbeta, float* dw,
                         int batch, float* ws, hipStream_t stream) {
    if (batch <= 0) return;
    if ((long long)batch * 1024 * 128 >= (1LL << 31))
        throw std::invalid_argument("conv1_wgrad_bn_pool: batch too large");
    const int M = batch * 1024;
    int spb;
    const int blocks = conv1_wgrad_blocks(M, &spb);
    const Conv1BnPool bn = {pre, gamma, beta, save_mean, save_rstd, dgamma, dbeta};
    hipLaunchKernelGGL(conv1_wgrad_kernel<1>, dim3(blocks), dim3(256), 0, stream, x, dy_pooled,
                       ws, M, spb, bn);
    hipLaunchKernelGGL(conv1_wgrad_finish_kernel, dim3(160), dim3(256), 0, stream,
                       (const float*)ws, dw, (float*)nullptr, blocks);
}

void conv1_wgrad_relu_pool(const bf16_raw* x, const bf16_raw* act, const bf16_raw* dy_pooled,
                           float* dw, float* db, int batch, float* ws, hipStream_t stream) {
    if (batch <= 0) return;
    if ((long long)batch * 1024 * 128 >= (1LL << 31))
        throw std::invalid_argument("conv1_wgrad_relu_pool: batch too large");
    const int M = batch * 1024;
    int spb;
    const int blocks = conv1_wgrad_blocks(M, &spb);
    Conv1BnPool a = {};
    a.pre = act;
    hipLaunchKernelGGL(conv1_wgrad_kernel<2>, dim3(blocks), dim3(256), 0, stream, x, dy_pooled, ws,
                       M, spb, a);
    hipLaunchKernelGGL(conv1_wgrad_finish_kernel, dim3(160), dim3(256), 0, stream,
                       (const float*)ws, dw, db, blocks);
}

}  // namespace dml
