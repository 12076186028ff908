// conv3x3 (stride 1, pad 1) forward and backward-data as one MFMA implicit GEMM.
//
// Reference op: nn.Conv2d(cin, cout, kernel_size=3, stride=1, padding=1, bias=True)
// (/root/reference/part1/model.py:18-23) and its autograd input gradient
// (loss.backward(), /root/reference/part1/main.py:44).
//
//   y[m][n] = sum_{tap, c} x[m (+) tap][c] * w[n][tap][c]        m = (b, h, w), NHWC
//
// GEMM view: M = B*H*W pixels, N = output channels, K = 9 * C ordered tap-major so one
// 64-wide K step lies inside one tap.  The "A" side (pixels) is gathered on the fly
// (im2col never exists in memory): every thread owns fixed pixel rows of the tile and per
// tap only re-derives the halo predicate and a constant offset.  Backward-data is the
// same kernel run on dy with the tap-flipped, channel-transposed weights.
//
// CDNA4 mapping
//   * 256 threads = 4 wave64; block tile BM x BN, K step 64, wave tile (BM/WM) x (BN/WN)
//     built from v_mfma_f32_16x16x32_bf16 with fp32 accumulators.
//   * The weight tile is fed as the MFMA *A* operand and the pixel tile as *B*, so the
//     accumulator layout is D[n][m]: a lane holds 4 consecutive output channels of one
//     pixel, which makes the epilogue (bias, ReLU, ReLU-mask, bf16 pack) an 8-byte
//     vector store into NHWC with no shuffles.
//   * Operand tiles live in LDS as 128-byte rows (64 bf16 of K).  16-byte chunks are
//     XOR-swizzled with (row >> 1) & 7 so the 16 rows x 2 chunks touched by one
//     ds_read_b128 lane group fall on 16 distinct 16-byte bank slots.
//   * Global -> register -> LDS staging with two LDS stages: the loads of step s+1 are
//     issued before the MFMAs of step s and written to the other stage after them; one
//     barrier per K step.
//   * Layers whose M x N grid cannot fill 256 CUs split the reduction over taps
//     (blockIdx.y); partial tiles go to an fp32 slab per split with plain stores and a
//     small finishing kernel sums the slabs and applies the epilogue (deterministic,
//     no atomics).
#include <stdexcept>
#include <string>

#include "dml_common.h"
#include "dml_kernels.h"

namespace dml {

namespace {

struct IgemmArgs {
    const uint16_t* x;
    const uint16_t* w;
    uint16_t* y;
    const float* bias;
    const uint16_t* mask;
    float* slab;          // fp32 [splits][M][N] when splitting
    int M, N, C;          // pixels, output channels, reduction channels
    int hw, log_hw;       // spatial size
    int log_chunks;       // log2(C / 64)
    int tiles_n;
    int taps_per_split;
};

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

template <int BM, int BN, int WAVES_M, int EPI, bool SPLIT>
__global__ __launch_bounds__(256) void conv3x3_igemm_kernel(IgemmArgs p) {
    constexpr int WAVES_N = 4 / WAVES_M;
    constexpr int WTM = BM / WAVES_M, WTN = BN / WAVES_N;
    constexpr int TM = WTM / 16, TN = WTN / 16;
    constexpr int A_IT = BM / 32, B_IT = BN / 32;
    constexpr int STAGE_BYTES = (BM + BN) * 128;
    static_assert(WTM % 16 == 0 && WTN % 16 == 0, "wave tile must be a multiple of 16");

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave % WAVES_M, wn = wave / WAVES_M;

    const int tile_n = blockIdx.x % p.tiles_n;
    const int tile_m = blockIdx.x / p.tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int tap_begin = SPLIT ? blockIdx.y * p.taps_per_split : 0;
    const int tap_end = SPLIT ? min(9, tap_begin + p.taps_per_split) : 9;

    const int C = p.C, hw = p.hw;
    const int chunks_mask = (1 << p.log_chunks) - 1;
    const int steps = (tap_end - tap_begin) << p.log_chunks;

    // ---- staging ownership: thread -> (row r0 + 32 i, 16-byte chunk c16) of both tiles
    const int c16 = tid & 7;
    const int r0 = tid >> 3;
    int pix_off[A_IT];       // element offset of (b,h,w,0) or -1 when the row is past M
    int pix_hw[A_IT];        // (h << 16) | w
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
        const int m = m0 + r0 + 32 * i;
        const int w_ = m & (hw - 1);
        const int h_ = (m >> p.log_hw) & (hw - 1);
        pix_hw[i] = (h_ << 16) | w_;
        pix_off[i] = (m < p.M) ? m * C : -1;
    }
    const uint16_t* wrow = p.w + (size_t)(n0 + r0) * 9 * C + c16 * 8;

    uint4 ra[A_IT], rb[B_IT];
    auto load_step = [&](int s) {
        const int tap = tap_begin + (s >> p.log_chunks);
        const int cc = s & chunks_mask;
        const int dh = tap / 3 - 1, dw = tap % 3 - 1;
        const int shift = (dh * hw + dw) * C + cc * 64 + c16 * 8;
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
            const int hh = (pix_hw[i] >> 16) + dh, ww = (pix_hw[i] & 0xffff) + dw;
            const bool ok = pix_off[i] >= 0 && (unsigned)hh < (unsigned)hw &&
                            (unsigned)ww < (unsigned)hw;
            ra[i] = ok ? *(const uint4*)(p.x + (pix_off[i] + shift)) : make_uint4(0, 0, 0, 0);
        }
        const uint16_t* wsrc = wrow + tap * C + cc * 64;
#pragma unroll
        for (int i = 0; i < B_IT; ++i) rb[i] = *(const uint4*)(wsrc + (size_t)(32 * i) * 9 * C);
    };
    auto store_step = [&](int stage) {
        char* sA = smem + stage * STAGE_BYTES;
        char* sB = sA + BM * 128;
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
            const int r = r0 + 32 * i;
            *(uint4*)(sA + r * 128 + ((c16 ^ ((r >> 1) & 7)) << 4)) = ra[i];
        }
#pragma unroll
        for (int i = 0; i < B_IT; ++i) {
            const int r = r0 + 32 * i;
            *(uint4*)(sB + r * 128 + ((c16 ^ ((r >> 1) & 7)) << 4)) = rb[i];
        }
    };

    f32x4 acc[TN][TM];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // fragment addressing: lane reads row (l & 15) of a 16-row tile, K chunk (kk*4 + l>>4)
    const int l15 = lane & 15, lq = lane >> 4;
    const int sw = l15 >> 1;
    const int koff0 = ((0 + lq) ^ sw) << 4, koff1 = ((4 + lq) ^ sw) << 4;
    const int a_row_byte = (wm * WTM + l15) * 128;             // pixel tile (MFMA B operand)
    const int b_row_byte = (BM + wn * WTN + l15) * 128;        // weight tile (MFMA A operand)

    auto compute = [&](int stage) {
        const char* base = smem + stage * STAGE_BYTES;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int koff = kk ? koff1 : koff0;
            bf16x8 wf[TN], xf[TM];
#pragma unroll
            for (int i = 0; i < TN; ++i)
                wf[i] = *(const bf16x8*)(base + b_row_byte + i * 16 * 128 + koff);
#pragma unroll
            for (int j = 0; j < TM; ++j)
                xf[j] = *(const bf16x8*)(base + a_row_byte + j * 16 * 128 + koff);
#pragma unroll
            for (int i = 0; i < TN; ++i)
#pragma unroll
                for (int j = 0; j < TM; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i], xf[j], acc[i][j],
                                                                        0, 0, 0);
        }
    };

    if (steps > 0) {
        load_step(0);
        store_step(0);
        __syncthreads();
        for (int s = 0; s < steps; ++s) {
            const bool more = s + 1 < steps;
            if (more) load_step(s + 1);
            compute(s & 1);
            if (more) store_step((s + 1) & 1);
            __syncthreads();
        }
    }

    // ---- epilogue: D[n][m]; lane holds n = nb + 4*lq + {0..3}, m = mb + l15
#pragma unroll
    for (int j = 0; j < TM; ++j) {
        const int m = m0 + wm * WTM + j * 16 + l15;
        if (m >= p.M) continue;
#pragma unroll
        for (int i = 0; i < TN; ++i) {
            const int n = n0 + wn * WTN + i * 16 + lq * 4;
            const size_t off = (size_t)m * p.N + n;
            if (SPLIT) {
                float* dst = p.slab + (size_t)blockIdx.y * p.M * p.N + off;
                *(float4*)dst = float4{acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
            } else {
                const float4 v = epilogue4(acc[i][j], EPI, p.bias, n, p.mask, off);
                uint2 o;
                o.x = pack2(v.x, v.y);
                o.y = pack2(v.z, v.w);
                *(uint2*)(p.y + off) = o;
            }
        }
    }
}

// Sums the per-split fp32 slabs and applies the epilogue; 4 outputs per thread.
__global__ __launch_bounds__(256) void splitk_finish_kernel(const float* __restrict__ slab,
                                                            uint16_t* __restrict__ y,
                                                            const float* __restrict__ bias,
                                                            const uint16_t* __restrict__ mask,
                                                            size_t total4, int N, int splits,
                                                            size_t slab_stride, int epi) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total4) return;
    const size_t off = i * 4;
    float4 a = *(const float4*)(slab + off);
    for (int s = 1; s < splits; ++s) {
        const float4 v = *(const float4*)(slab + s * slab_stride + off);
        a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
    }
    const int n = (int)(off % N);
    const float4 v = epilogue4(f32x4{a.x, a.y, a.z, a.w}, epi, bias, n, mask, off);
    uint2 o;
    o.x = pack2(v.x, v.y);
    o.y = pack2(v.z, v.w);
    *(uint2*)(y + off) = o;
}

int ilog2(int v) {
    int l = 0;
    while ((1 << l) < v) ++l;
    return l;
}

struct Plan {
    int bm, bn, splits;
};

// Tile / split choice: biggest tile that still gives >= ~0.8 blocks per CU; otherwise the
// smallest tile plus a tap split (3 or 9 ways) until the grid fills the chip.
Plan choose_plan(int M, int N) {
    const int kCUs = 256;
    auto blocks = [&](int bm, int bn) { return ((M + bm - 1) / bm) * (N / bn); };
    if (N % 128 == 0 && blocks(128, 128) >= 200) return {128, 128, 1};
    if (blocks(128, 64) >= 200) return {128, 64, 1};
    if (blocks(64, 64) >= 200) return {64, 64, 1};
    const int b = blocks(64, 64);
    if (b * 3 >= kCUs / 2) return {64, 64, 3};
    return {64, 64, 9};
}

template <int BM, int BN, int WAVES_M>
void launch_cfg(const IgemmArgs& a, int epi, int splits, hipStream_t stream) {
    const int tiles_m = (a.M + BM - 1) / BM;
    const dim3 grid(tiles_m * a.tiles_n, splits), block(256);
    const size_t lds = 2 * (BM + BN) * 128;
    if (splits > 1) {
        hipLaunchKernelGGL((conv3x3_igemm_kernel<BM, BN, WAVES_M, EPI_NONE, true>), grid, block,
                           lds, stream, a);
        return;
    }
    switch (epi) {
        case EPI_NONE:
            hipLaunchKernelGGL((conv3x3_igemm_kernel<BM, BN, WAVES_M, EPI_NONE, false>), grid,
                               block, lds, stream, a);
            break;
        case EPI_BIAS:
            hipLaunchKernelGGL((conv3x3_igemm_kernel<BM, BN, WAVES_M, EPI_BIAS, false>), grid,
                               block, lds, stream, a);
            break;
        case EPI_BIAS_RELU:
            hipLaunchKernelGGL((conv3x3_igemm_kernel<BM, BN, WAVES_M, EPI_BIAS_RELU, false>), grid,
                               block, lds, stream, a);
            break;
        default:
            hipLaunchKernelGGL((conv3x3_igemm_kernel<BM, BN, WAVES_M, EPI_MASK, false>), grid,
                               block, lds, stream, a);
            break;
    }
}

void check_shape(const ConvShape& s, const char* who) {
    if (s.batch <= 0 || s.hw < 1 || s.hw > 32 || (s.hw & (s.hw - 1)) || s.cin < 64 ||
        s.cin % 64 || s.cout < 64 || s.cout % 64 || (s.cin & (s.cin - 1)))
        throw std::invalid_argument(std::string(who) +
                                    ": needs hw in {1..32} power of two, cin a power of two >= 64, "
                                    "cout a multiple of 64");
    if ((long long)s.batch * s.hw * s.hw * (long long)(s.cin > s.cout ? s.cin : s.cout) >=
        (1LL << 31))
        throw std::invalid_argument(std::string(who) + ": tensor exceeds 2^31 elements");
}

}  // namespace

size_t conv3x3_igemm_workspace_elems(ConvShape s) {
    const int M = s.batch * s.hw * s.hw;
    const Plan pl = choose_plan(M, s.cout);
    return pl.splits > 1 ? (size_t)pl.splits * M * s.cout : 0;
}

void conv3x3_igemm(const bf16_raw* x, const bf16_raw* w, bf16_raw* y, const float* bias,
                   const bf16_raw* mask, float* workspace, size_t workspace_elems, ConvShape s,
                   ConvEpilogue epi, hipStream_t stream) {
    check_shape(s, "conv3x3_igemm");
    if ((epi == EPI_BIAS || epi == EPI_BIAS_RELU) && !bias)
        throw std::invalid_argument("conv3x3_igemm: bias epilogue without bias");
    if (epi == EPI_MASK && !mask) throw std::invalid_argument("conv3x3_igemm: mask missing");
    IgemmArgs a;
    a.x = x; a.w = w; a.y = y; a.bias = bias; a.mask = mask; a.slab = workspace;
    a.M = s.batch * s.hw * s.hw; a.N = s.cout; a.C = s.cin;
    a.hw = s.hw; a.log_hw = ilog2(s.hw); a.log_chunks = ilog2(s.cin / 64);
    const Plan pl = choose_plan(a.M, a.N);
    a.tiles_n = a.N / pl.bn;
    a.taps_per_split = 9 / pl.splits;
    if (pl.splits > 1) {
        const size_t need = (size_t)pl.splits * a.M * a.N;
        if (!workspace || workspace_elems < need)
            throw std::invalid_argument("conv3x3_igemm: workspace too small for the tap split");
    }
    if (pl.bm == 128 && pl.bn == 128) launch_cfg<128, 128, 2>(a, epi, pl.splits, stream);
    else if (pl.bm == 128 && pl.bn == 64) launch_cfg<128, 64, 2>(a, epi, pl.splits, stream);
    else launch_cfg<64, 64, 2>(a, epi, pl.splits, stream);
    if (pl.splits > 1) {
        const size_t total4 = (size_t)a.M * a.N / 4;
        hipLaunchKernelGGL(splitk_finish_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256),
                           0, stream, workspace, y, bias, mask, total4, a.N, pl.splits,
                           (size_t)a.M * a.N, (int)epi);
    }
}

}  // namespace dml
