// fp32 convolutions for --dtype fp32: the reference's precision (its nn.Conv2d layers run in
// fp32, /root/reference/part1/model.py:18-23 with /root/reference/part1/main.py:15-16) on the
// matrix cores.
//
// gfx950 has an fp32-input MFMA, v_mfma_f32_32x32x2_f32: exact fp32 products, fp32 sums, no
// reduced-precision (tf32-like) path.  It runs at 1/16 of the bf16 MFMA rate, i.e. at the vector
// ALU's peak, but leaves the vector ALU and the registers free and needs one VGPR per operand.
// At that rate everything around the MFMAs (LDS reads, staging, the epilogue) is cheap, so these
// kernels use a plain structure: register-staged global loads one K chunk ahead, double-buffered
// LDS tiles, one barrier per chunk, 4 waves per block each owning a (32*MT) x (32*NT) output
// tile.
//
// Operand maps of v_mfma_f32_32x32x2_f32 (lane l, r = l & 31, h = l >> 5):
//   A[i = r][k = h], B[k = h][j = r] - one float each;  D[i][j]: j = r, i = (g&3) + 8*(g>>2) + 4*h
//   for accumulator register g of 16.
#include <stdexcept>

#include "dml_common.h"
#include "dml_kernels.h"

namespace dml {

namespace {

__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ int acc_row(int g, int h) { return (g & 3) + 8 * (g >> 2) + 4 * h; }

constexpr int kBK = 32;      // reduction elements per LDS chunk

int log2_exact(int v) {
    int l = 0;
    while ((1 << l) < v) ++l;
    if ((1 << l) != v) throw std::invalid_argument("fp32 conv: the spatial size must be a power of two");
    return l;
}

// ------------------------------------------------------------------ forward / backward-data
// y[m][n] = sum_{tap,c} x[m (+) tap][c] * w[n][tap][c].  Both LDS tiles are [row][32 floats]
// with the row's eight 16-byte slots XOR-permuted by (row & 7); a lane reads its operand for
// four MFMA steps as one ds_read_b128: lane half h takes reduction elements 8q+4h .. 8q+4h+3,
// so step t of group q sums elements 8q+t and 8q+4+t - the same pairing on both operands.
struct IgemmParams {
    const float* x;
    const float* w;
    float* y;
    const float* bias;
    const float* mask;
    int M, N, cin, hw, log_hw, epi;
};

template <int MT, int NT>
__global__ __launch_bounds__(256) void conv3x3_igemm_f32_kernel(IgemmParams p) {
    constexpr int BM = 64 * MT, BN = 64 * NT;
    constexpr int LA = BM * 8 / 256, LB = BN * 8 / 256;     // float4 loads per thread and chunk
    __shared__ __attribute__((aligned(16))) float As[2][BM * kBK];
    __shared__ __attribute__((aligned(16))) float Bs[2][BN * kBK];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    const int ntiles = p.N / BN;
    const int m0 = (blockIdx.x / ntiles) * BM, n0 = (blockIdx.x % ntiles) * BN;
    const int hw = p.hw, cin = p.cin;

    // the rows this thread stages are the same for every chunk
    int a_h[LA], a_w[LA];
    long a_base[LA];
    bool a_ok[LA];
#pragma unroll
    for (int i = 0; i < LA; ++i) {
        const int idx = tid + 256 * i, row = idx >> 3;
        const int m = m0 + row;
        a_ok[i] = m < p.M;
        a_w[i] = m & (hw - 1);
        a_h[i] = (m >> p.log_hw) & (hw - 1);
        a_base[i] = (long)m * cin + (idx & 7) * 4;
    }
    long b_base[LB];
#pragma unroll
    for (int i = 0; i < LB; ++i) {
        const int idx = tid + 256 * i, row = idx >> 3;
        b_base[i] = (long)(n0 + row) * 9 * cin + (idx & 7) * 4;
    }
    const int chunks = cin / kBK, steps = 9 * chunks;
    float4 ra[LA], rb[LB];
    auto fetch = [&](int step) {
        const int tap = step / chunks, c0 = (step - tap * chunks) * kBK;
        const int dh = tap / 3 - 1, dw = tap % 3 - 1;
#pragma unroll
        for (int i = 0; i < LA; ++i) {
            const bool ok = a_ok[i] && (unsigned)(a_h[i] + dh) < (unsigned)hw &&
                            (unsigned)(a_w[i] + dw) < (unsigned)hw;
            ra[i] = ok ? *(const float4*)(p.x + a_base[i] + (long)(dh * hw + dw) * cin + c0)
                       : float4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int i = 0; i < LB; ++i) rb[i] = *(const float4*)(p.w + b_base[i] + (long)tap * cin + c0);
    };
    auto stash = [&](int buf) {
#pragma unroll
        for (int i = 0; i < LA; ++i) {
            const int idx = tid + 256 * i, row = idx >> 3, slot = (idx & 7) ^ (row & 7);
            *(float4*)(&As[buf][row * kBK + slot * 4]) = ra[i];
        }
#pragma unroll
        for (int i = 0; i < LB; ++i) {
            const int idx = tid + 256 * i, row = idx >> 3, slot = (idx & 7) ^ (row & 7);
            *(float4*)(&Bs[buf][row * kBK + slot * 4]) = rb[i];
        }
    };

    f32x16 acc[MT][NT];
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b)
#pragma unroll
            for (int g = 0; g < 16; ++g) acc[a][b][g] = 0.f;

    fetch(0);
    stash(0);
    __syncthreads();
    for (int step = 0; step < steps; ++step) {
        const int buf = step & 1;
        if (step + 1 < steps) fetch(step + 1);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float4 fa[MT], fb[NT];
#pragma unroll
            for (int a = 0; a < MT; ++a) {
                const int row = wm * 32 * MT + a * 32 + r;
                fa[a] = *(const float4*)(&As[buf][row * kBK + (((2 * q + h) ^ (row & 7)) * 4)]);
            }
#pragma unroll
            for (int b = 0; b < NT; ++b) {
                const int row = wn * 32 * NT + b * 32 + r;
                fb[b] = *(const float4*)(&Bs[buf][row * kBK + (((2 * q + h) ^ (row & 7)) * 4)]);
            }
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int a = 0; a < MT; ++a)
#pragma unroll
                    for (int b = 0; b < NT; ++b)
                        acc[a][b] = mfma32(((const float*)&fa[a])[t], ((const float*)&fb[b])[t],
                                           acc[a][b]);
        }
        if (step + 1 < steps) stash(buf ^ 1);
        __syncthreads();
    }

    // epilogue: a lane owns column n = .. + r and 16 rows per tile
#pragma unroll
    for (int b = 0; b < NT; ++b) {
        const int n = n0 + wn * 32 * NT + b * 32 + r;
        const float bias = (p.epi == EPI_BIAS || p.epi == EPI_BIAS_RELU) ? p.bias[n] : 0.f;
#pragma unroll
        for (int a = 0; a < MT; ++a) {
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                const int m = m0 + wm * 32 * MT + a * 32 + acc_row(g, h);
                if (m >= p.M) continue;
                float v = acc[a][b][g] + bias;
                if (p.epi == EPI_BIAS_RELU) v = fmaxf(v, 0.f);
                if (p.epi == EPI_MASK && !(p.mask[(long)m * p.N + n] > 0.f)) v = 0.f;
                p.y[(long)m * p.N + n] = v;
            }
        }
    }
}

// ------------------------------------------------------------------------ backward-weight
// Per tap a TN GEMM dw[co][ci] = sum_p dy[p][co] * x[p (+) tap][ci]; both operands are
// pixel-major in memory and stay so in LDS ([32 pixels][channels]): a lane's A element
// dy[p = 2s+h][co = r] and B element x[p][ci = r] are plain 4-byte LDS reads, consecutive lanes
// on consecutive banks.  blockIdx.x = (co tile, ci tile, tap), blockIdx.y = split of the pixels.
struct WgradParams {
    const float* x;
    const float* dy;
    float* out;          // dw, or the slab of this launch (split s at out + s * slab_stride)
    long slab_stride;
    int M, cin, cout, hw, log_hw, chunks_per_split;
};

template <int MT, int NT>
__global__ __launch_bounds__(256) void conv3x3_wgrad_f32_kernel(WgradParams p) {
    constexpr int BM = 64 * MT, BN = 64 * NT;
    constexpr int LA = kBK * BM / 4 / 256, LB = kBK * BN / 4 / 256;
    __shared__ __attribute__((aligned(16))) float Ds[2][kBK * BM];
    __shared__ __attribute__((aligned(16))) float Xs[2][kBK * BN];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    const int ci_tiles = p.cin / BN;
    const int tap = blockIdx.x % 9;
    const int tile = blockIdx.x / 9;
    const int co0 = (tile / ci_tiles) * BM, ci0 = (tile % ci_tiles) * BN;
    const int dh = tap / 3 - 1, dw = tap % 3 - 1;
    const int hw = p.hw;
    const int total_chunks = (p.M + kBK - 1) / kBK;
    const int c_begin = blockIdx.y * p.chunks_per_split;
    int c_end = c_begin + p.chunks_per_split;
    if (c_end > total_chunks) c_end = total_chunks;

    float4 rd[LA], rx[LB];
    auto fetch = [&](int chunk) {
        const int p0 = chunk * kBK;
#pragma unroll
        for (int i = 0; i < LA; ++i) {
            const int idx = tid + 256 * i, px = idx / (BM / 4), c4 = idx % (BM / 4);
            const int m = p0 + px;
            rd[i] = m < p.M ? *(const float4*)(p.dy + (long)m * p.cout + co0 + c4 * 4)
                            : float4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int i = 0; i < LB; ++i) {
            const int idx = tid + 256 * i, px = idx / (BN / 4), c4 = idx % (BN / 4);
            const int m = p0 + px;
            const int w_ = m & (hw - 1), h_ = (m >> p.log_hw) & (hw - 1);
            const bool ok = m < p.M && (unsigned)(h_ + dh) < (unsigned)hw &&
                            (unsigned)(w_ + dw) < (unsigned)hw;
            rx[i] = ok ? *(const float4*)(p.x + (long)(m + dh * hw + dw) * p.cin + ci0 + c4 * 4)
                       : float4{0.f, 0.f, 0.f, 0.f};
        }
    };
    auto stash = [&](int buf) {
#pragma unroll
        for (int i = 0; i < LA; ++i) *(float4*)(&Ds[buf][(tid + 256 * i) * 4]) = rd[i];
#pragma unroll
        for (int i = 0; i < LB; ++i) *(float4*)(&Xs[buf][(tid + 256 * i) * 4]) = rx[i];
    };

    f32x16 acc[MT][NT];
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b)
#pragma unroll
            for (int g = 0; g < 16; ++g) acc[a][b][g] = 0.f;

    if (c_begin < c_end) {
        fetch(c_begin);
        stash(0);
    }
    __syncthreads();
    for (int chunk = c_begin; chunk < c_end; ++chunk) {
        const int buf = (chunk - c_begin) & 1;
        if (chunk + 1 < c_end) fetch(chunk + 1);
#pragma unroll
        for (int s = 0; s < kBK / 2; ++s) {
            float fa[MT], fb[NT];
#pragma unroll
            for (int a = 0; a < MT; ++a) fa[a] = Ds[buf][(2 * s + h) * BM + wm * 32 * MT + a * 32 + r];
#pragma unroll
            for (int b = 0; b < NT; ++b) fb[b] = Xs[buf][(2 * s + h) * BN + wn * 32 * NT + b * 32 + r];
#pragma unroll
            for (int a = 0; a < MT; ++a)
#pragma unroll
                for (int b = 0; b < NT; ++b) acc[a][b] = mfma32(fa[a], fb[b], acc[a][b]);
        }
        if (chunk + 1 < c_end) stash(buf ^ 1);
        __syncthreads();
    }
    float* out = p.out + (long)blockIdx.y * p.slab_stride;
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b) {
            const int ci = ci0 + wn * 32 * NT + b * 32 + r;
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                const int co = co0 + wm * 32 * MT + a * 32 + acc_row(g, h);
                out[((long)co * 9 + tap) * p.cin + ci] = acc[a][b][g];
            }
        }
}

// --------------------------------------------------------------------------- first layer
// Cin = 3: the reduction is 27 long.  A block builds the im2col patch of 128 pixels in LDS
// (P[pixel][k = tap*3 + c], k = 27 zero) - the master weights [64][9][3] are already [n][k] -
// and each wave multiplies its 32 pixels with all 64 filters: 14 MFMA steps.
constexpr int kC1K = 28, kC1Stride = 29;

__device__ __forceinline__ void conv1_build_patch(const float* __restrict__ x, int m0, int M,
                                                  float* P, int stride, int k_one) {
    // 128 pixels x 9 taps; pixel = idx % 128 keeps a wave on consecutive pixels
    for (int idx = threadIdx.x; idx < 128 * 9; idx += 256) {
        const int px = idx & 127, tap = idx >> 7;
        const int m = m0 + px;
        const int w_ = m & 31, h_ = (m >> 5) & 31;
        const int dh = tap / 3 - 1, dw = tap % 3 - 1;
        const bool ok = m < M && (unsigned)(h_ + dh) < 32u && (unsigned)(w_ + dw) < 32u;
        const float4 v = ok ? *(const float4*)(x + (long)(m + dh * 32 + dw) * 4)
                            : float4{0.f, 0.f, 0.f, 0.f};
        float* dst = P + px * stride + tap * 3;
        dst[0] = v.x; dst[1] = v.y; dst[2] = v.z;
    }
    // the padding columns: zero, except column k_one (if >= 0) which is 1 for real pixels
    if (threadIdx.x < 128) {
        const int px = threadIdx.x;
        for (int k = 27; k < stride; ++k)
            P[px * stride + k] = (k == k_one && m0 + px < M) ? 1.f : 0.f;
    }
}

__global__ __launch_bounds__(256) void conv1_fprop_f32_kernel(
    const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
    float* __restrict__ y, int M, int relu) {
    __shared__ float P[128 * kC1Stride];
    __shared__ float W[64 * kC1Stride];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int m0 = blockIdx.x * 128;
    conv1_build_patch(x, m0, M, P, kC1Stride, -1);
    for (int idx = tid; idx < 64 * kC1Stride; idx += 256) {
        const int n = idx / kC1Stride, k = idx % kC1Stride;
        W[idx] = k < 27 ? w[n * 27 + k] : 0.f;
    }
    __syncthreads();
    f32x16 acc[2];
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int g = 0; g < 16; ++g) acc[b][g] = 0.f;
#pragma unroll
    for (int s = 0; s < kC1K / 2; ++s) {
        const float a = P[(wave * 32 + r) * kC1Stride + 2 * s + h];
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[b] = mfma32(a, W[(b * 32 + r) * kC1Stride + 2 * s + h], acc[b]);
    }
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const int n = b * 32 + r;
        const float bv = bias[n];
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            const int m = m0 + wave * 32 + acc_row(g, h);
            if (m >= M) continue;
            float v = acc[b][g] + bv;
            if (relu) v = fmaxf(v, 0.f);
            y[(long)m * 64 + n] = v;
        }
    }
}

// dW1[co][k] = sum_p dy[p][co] * P[p][k]; patch column 27 is 1, so out[co][27] = db[co].
// A block walks chunks of 128 pixels (wave = 32 of them), keeps [64 co][32 k] per wave in two
// accumulators, folds its four waves through LDS and stores one partial per block; a second
// kernel sums the blocks' partials in a fixed order.
constexpr int kC1WStride = 33;

__global__ __launch_bounds__(256) void conv1_wgrad_f32_kernel(
    const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ ws, int M) {
    __shared__ float P[128 * kC1WStride];
    __shared__ __attribute__((aligned(16))) float D[128 * 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int chunks = (M + 127) / 128;
    f32x16 acc[2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int g = 0; g < 16; ++g) acc[a][g] = 0.f;
    for (int chunk = blockIdx.x; chunk < chunks; chunk += gridDim.x) {
        const int m0 = chunk * 128;
        __syncthreads();                    // the previous chunk's tiles have been consumed
        conv1_build_patch(x, m0, M, P, kC1WStride, 27);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int idx = tid + 256 * i, px = idx >> 4, c4 = idx & 15;
            const int m = m0 + px;
            *(float4*)(&D[px * 64 + c4 * 4]) =
                m < M ? *(const float4*)(dy + (long)m * 64 + c4 * 4) : float4{0.f, 0.f, 0.f, 0.f};
        }
        __syncthreads();
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const int px = wave * 32 + 2 * s + h;
            const float b = P[px * kC1WStride + r];
#pragma unroll
            for (int a = 0; a < 2; ++a) acc[a] = mfma32(D[px * 64 + a * 32 + r], b, acc[a]);
        }
    }
    __syncthreads();
    float* red = D;                         // [4 waves][64 co][32 k]
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int g = 0; g < 16; ++g) red[(wave * 64 + a * 32 + acc_row(g, h)) * 32 + r] = acc[a][g];
    __syncthreads();
    float* out = ws + (long)blockIdx.x * 2048;
    for (int i = tid; i < 2048; i += 256)
        out[i] = red[i] + red[2048 + i] + red[4096 + i] + red[6144 + i];
}

__global__ __launch_bounds__(256) void conv1_wgrad_f32_fold_kernel(
    const float* __restrict__ ws, float* __restrict__ dw, float* __restrict__ db, int nblocks) {
    const int i = blockIdx.x * 256 + threadIdx.x;       // (co, k) of 64 x 32
    if (i >= 2048) return;
    float s = 0.f;
    for (int b = 0; b < nblocks; ++b) s += ws[(long)b * 2048 + i];
    const int co = i >> 5, k = i & 31;
    if (k < 27) dw[co * 27 + k] = s;
    else if (k == 27 && db) db[co] = s;
}

// ----------------------------------------------------------------------------- helpers
__global__ __launch_bounds__(256) void repack_dgrad_f32_kernel(const float* __restrict__ wf,
                                                               float* __restrict__ wd, int cout,
                                                               int cin, size_t total) {
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;      // over wd elements
    if (idx >= total) return;
    const int co = (int)(idx % cout);
    const size_t q = idx / cout;
    const int t = (int)(q % 9), ci = (int)(q / 9);
    wd[idx] = wf[((size_t)co * 9 + (8 - t)) * cin + ci];
}

__global__ __launch_bounds__(256) void relu_mask_f32_kernel(const float4* __restrict__ x,
                                                            const float4* __restrict__ mask,
                                                            float4* __restrict__ y, size_t n4) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n4; i += stride) {
        const float4 v = x[i], m = mask[i];
        y[i] = float4{m.x > 0.f ? v.x : 0.f, m.y > 0.f ? v.y : 0.f, m.z > 0.f ? v.z : 0.f,
                      m.w > 0.f ? v.w : 0.f};
    }
}

void check_shape(const ConvShape& s, const char* who) {
    if (s.batch <= 0 || s.hw < 1 || s.hw > 32 || s.cin % 64 || s.cout % 64 || s.cin < 64 ||
        s.cout < 64)
        throw std::invalid_argument(std::string(who) + ": channels must be multiples of 64, "
                                                       "1 <= hw <= 32, batch > 0");
}

}  // namespace

void conv3x3_igemm_f32(const float* x, const float* w, float* y, const float* bias,
                       const float* mask, ConvShape s, ConvEpilogue epi, hipStream_t stream,
                       int force_tile) {
    check_shape(s, "conv3x3_igemm_f32");
    if (!x || !w || !y) throw std::invalid_argument("conv3x3_igemm_f32: null tensor");
    if ((epi == EPI_BIAS || epi == EPI_BIAS_RELU) && !bias)
        throw std::invalid_argument("conv3x3_igemm_f32: bias missing");
    if (epi == EPI_MASK && !mask) throw std::invalid_argument("conv3x3_igemm_f32: mask missing");
    IgemmParams p;
    p.x = x; p.w = w; p.y = y; p.bias = bias; p.mask = mask;
    p.M = s.batch * s.hw * s.hw; p.N = s.cout; p.cin = s.cin; p.hw = s.hw;
    p.log_hw = log2_exact(s.hw); p.epi = (int)epi;
    // 128 x 128 tiles where they still give every CU a block, 64 x 64 tiles otherwise
    const long big_blocks = (long)((p.M + 127) / 128) * (p.N / 128);
    bool big = p.N % 128 == 0 && big_blocks >= 256;
    if (force_tile == 128) {
        if (p.N % 128) throw std::invalid_argument("conv3x3_igemm_f32: 128 tiles need cout % 128 == 0");
        big = true;
    } else if (force_tile == 64) {
        big = false;
    } else if (force_tile != 0) {
        throw std::invalid_argument("conv3x3_igemm_f32: tile must be 64 or 128");
    }
    if (big)
        hipLaunchKernelGGL((conv3x3_igemm_f32_kernel<2, 2>), dim3((unsigned)big_blocks), dim3(256), 0,
                           stream, p);
    else
        hipLaunchKernelGGL((conv3x3_igemm_f32_kernel<1, 1>),
                           dim3((unsigned)(((p.M + 63) / 64) * (p.N / 64))), dim3(256), 0, stream, p);
}

void repack_dgrad_weights_f32(const float* wf, float* wd, int cout, int cin, hipStream_t stream) {
    if (cout <= 0 || cin <= 0) throw std::invalid_argument("repack_dgrad_weights_f32: bad shape");
    const size_t total = (size_t)cout * 9 * cin;
    hipLaunchKernelGGL(repack_dgrad_f32_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                       stream, wf, wd, cout, cin, total);
}

namespace {
struct WgradPlan { int tiles, splits, chunks_per_split; bool wide; };

WgradPlan plan_wgrad(const ConvShape& s, int force_splits) {
    WgradPlan pl;
    pl.wide = s.cin % 128 == 0;
    if (s.cout % 128) throw std::invalid_argument("conv3x3_wgrad_f32: cout must be a multiple of 128");
    pl.tiles = (s.cout / 128) * (s.cin / (pl.wide ? 128 : 64));
    const int chunks = (s.batch * s.hw * s.hw + kBK - 1) / kBK;
    // about two blocks per CU, at least four chunks of pixels per block
    int splits = force_splits > 0 ? force_splits : (512 + pl.tiles * 9 - 1) / (pl.tiles * 9);
    const int max_splits = chunks / 4 > 0 ? chunks / 4 : 1;
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
    pl.chunks_per_split = (chunks + splits - 1) / splits;
    pl.splits = (chunks + pl.chunks_per_split - 1) / pl.chunks_per_split;
    return pl;
}
}  // namespace

int conv3x3_wgrad_f32_splits(ConvShape s, int force_splits) {
    check_shape(s, "conv3x3_wgrad_f32");
    return plan_wgrad(s, force_splits).splits;
}

void conv3x3_wgrad_f32(const float* x, const float* dy, float* dw, ConvShape s, float* slab,
                       hipStream_t stream, int force_splits) {
    check_shape(s, "conv3x3_wgrad_f32");
    if (!x || !dy || !dw) throw std::invalid_argument("conv3x3_wgrad_f32: null tensor");
    const WgradPlan pl = plan_wgrad(s, force_splits);
    if (pl.splits > 1 && !slab)
        throw std::invalid_argument("conv3x3_wgrad_f32: this shape splits the pixels and needs a slab");
    WgradParams p;
    p.x = x; p.dy = dy;
    p.out = pl.splits > 1 ? slab : dw;
    p.slab_stride = (long)s.cout * 9 * s.cin;
    p.M = s.batch * s.hw * s.hw; p.cin = s.cin; p.cout = s.cout; p.hw = s.hw;
    p.log_hw = log2_exact(s.hw); p.chunks_per_split = pl.chunks_per_split;
    const dim3 grid((unsigned)(pl.tiles * 9), (unsigned)pl.splits);
    if (pl.wide)
        hipLaunchKernelGGL((conv3x3_wgrad_f32_kernel<2, 2>), grid, dim3(256), 0, stream, p);
    else
        hipLaunchKernelGGL((conv3x3_wgrad_f32_kernel<2, 1>), grid, dim3(256), 0, stream, p);
}

void conv1_fprop_f32(const float* x, const float* w, const float* bias, float* y, int batch,
                     bool relu, hipStream_t stream) {
    if (batch <= 0) throw std::invalid_argument("conv1_fprop_f32: batch must be positive");
    if (!x || !w || !bias || !y) throw std::invalid_argument("conv1_fprop_f32: null tensor");
    const int M = batch * 1024;
    hipLaunchKernelGGL(conv1_fprop_f32_kernel, dim3((M + 127) / 128), dim3(256), 0, stream, x, w, bias,
                       y, M, relu ? 1 : 0);
}

namespace {
int conv1_wgrad_blocks(int batch) {
    const int chunks = batch * 8;       // 1024 pixels per image / 128
    return chunks < 512 ? chunks : 512;
}
}  // namespace

size_t conv1_wgrad_f32_workspace_elems(int batch) { return (size_t)conv1_wgrad_blocks(batch) * 2048; }

void conv1_wgrad_f32(const float* x, const float* dy, float* dw, float* db, int batch, float* ws,
                     hipStream_t stream) {
    if (batch <= 0) throw std::invalid_argument("conv1_wgrad_f32: batch must be positive");
    if (!x || !dy || !dw || !ws) throw std::invalid_argument("conv1_wgrad_f32: null tensor");
    const int nb = conv1_wgrad_blocks(batch);
    hipLaunchKernelGGL(conv1_wgrad_f32_kernel, dim3(nb), dim3(256), 0, stream, x, dy, ws,
                       batch * 1024);
    hipLaunchKernelGGL(conv1_wgrad_f32_fold_kernel, dim3(8), dim3(256), 0, stream, (const float*)ws,
                       dw, db, nb);
}

void relu_mask_f32(const float* x, const float* mask, float* y, size_t n, hipStream_t stream) {
    if (n % 4) throw std::invalid_argument("relu_mask_f32: n must be a multiple of 4");
    if (n == 0) return;
    size_t blocks = (n / 4 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(relu_mask_f32_kernel, dim3((unsigned)blocks), dim3(256), 0, stream,
                       (const float4*)x, (const float4*)mask, (float4*)y, n / 4);
}

}  // namespace dml
