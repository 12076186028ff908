// conv3x3 (stride 1, pad 1) forward and backward-data as one MFMA implicit GEMM.
//
// Reference op: nn.Conv2d(cin, cout, kernel_size=3, stride=1, padding=1, bias=True)
// (/root/reference/part1/model.py:18-23) and its autograd input gradient
// (loss.backward(), /root/reference/part1/main.py:44).
//
//   y[m][n] = sum_{tap, c} x[m (+) tap][c] * w[n][tap][c]        m = (b, h, w), NHWC
//
// GEMM view: M = B*H*W pixels, N = output channels, K = 9 * C walked chunk-outer /
// tap-inner in 64-channel steps (consecutive steps re-read neighbouring pixels of the same
// 64 channels, which stay in L2).  The "A" side (pixels) is gathered on the fly - im2col
// never exists in memory: every thread owns fixed pixel rows of the tile and per tap only
// re-derives the halo predicate and a constant offset.  Backward-data is the same kernel
// run on dy with the tap-flipped, channel-transposed weights.
//
// CDNA4 mapping
//   * 256 threads = 4 wave64; block tile BM x BN (128x128, 128x64 or 64x64), K step 64,
//     wave tiles built from v_mfma_f32_16x16x32_bf16 with fp32 accumulators.
//   * The weight tile is fed as the MFMA *A* operand and the pixel tile as *B*, so the
//     accumulator layout is D[n][m]: a lane holds 4 consecutive output channels of one
//     pixel, which makes the epilogue (bias, ReLU, ReLU-mask, bf16 pack) an 8-byte
//     vector store into NHWC with no shuffles.
//   * Global -> LDS without registers: buffer loads with the LDS-destination form write a
//     wave's 64 x 16 bytes as one linear 1 KiB LDS image.  Operand tiles are 128-byte rows
//     (64 bf16 of K); the bank swizzle chunk ^ ((row >> 1) & 7) is applied on the SOURCE
//     side (each lane fetches the global chunk that belongs in its fixed LDS slot), so the
//     ds_read_b128 fragment reads stay conflict-free.
//   * Halo pixels and rows past the end of the tensor are not branched on: their lanes use
//     a voffset beyond the buffer descriptor's extent and the hardware returns zeros.
//   * NS-stage LDS ring (3 or 4 stages by tile size) with counted waits: step s waits with
//     s_waitcnt vmcnt((D-1) * loads_per_stage) for its own stage only, one raw s_barrier,
//     then issues the loads of step s+D-1... and runs the MFMAs of step s, so D-1 stages of
//     HBM/L2 latency are always in flight behind the matrix cores.
//   * Layers whose M x N grid cannot fill 256 CUs split the reduction over taps
//     (blockIdx.y); partial tiles go to an fp32 slab per split with plain stores and a
//     small finishing kernel sums the slabs and applies the epilogue (deterministic,
//     no atomics).
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <stdexcept>
#include <string>

#include "dml_column_reduce.h"
#include "dml_common.h"
#include "dml_kernels.h"
#include "dml_conv_tile.h"

namespace dml {

namespace igemm {


// PM ("position-major") tiling: the BM rows of a tile are BM consecutive IMAGES at ONE
// spatial position instead of BM consecutive NHWC pixels.  Whether a tap falls into the
// padding then is the same for every row of the tile, so padded taps are skipped as whole K
// steps (56 % of the work at 2x2, 31 % at 4x4, 16 % at 8x8) instead of multiplying zeros.
// Rows are 128-byte K chunks fetched by per-lane addresses either way, so the memory layout
// stays NHWC.
//
// NW = waves per block (4 or 8).  Eight waves put two waves on every SIMD of the CU a large
// tile occupies alone, so one wave's waits (LDS fragments, the stage barrier) hide behind the
// other's MFMAs without shrinking the tile (the L2 -> LDS feed rate, ~28 B/clk/CU, is what
// bounds small tiles: bytes per flop fall with tile size).
template <int BM, int BN, int WAVES_M, int EPI, bool SPLIT, int NS, bool PM, int NW>
__global__ __launch_bounds__(NW * 64) void conv3x3_igemm_kernel(IgemmArgs p) {
    raise_stream_signal(p);
    constexpr int WAVES_N = NW / WAVES_M;
    constexpr int WTM = BM / WAVES_M, WTN = BN / WAVES_N;
    constexpr int TM = WTM / 16, TN = WTN / 16;
    constexpr int A_IT = BM / (8 * NW), B_IT = BN / (8 * NW);   // 1 KiB wave-loads per step and wave
    static_assert(A_IT >= 1 && B_IT >= 1, "tile too small for this many waves");
    constexpr int LPS = A_IT + B_IT;                  // loads per step per wave
    constexpr int D = NS - 1;                         // prefetch distance in K steps
    constexpr int STAGE_BYTES = (BM + BN) * 128;
    static_assert(WTM % 16 == 0 && WTN % 16 == 0, "wave tile must be a multiple of 16");
    static_assert(NS >= 2 && NS <= 4, "2..4 LDS stages");

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave % WAVES_M, wn = wave / WAVES_M;

    // Workgroups are dealt round-robin to the 8 XCDs, each with its own L2.  Renumber so that
    // an XCD works on a contiguous range of tiles: the tile_n blocks of one pixel tile (same
    // input lines) and neighbouring pixel tiles (shared halo lines) then hit the same L2.
    int bid = blockIdx.x;
    if ((gridDim.x & 7) == 0) bid = (bid & 7) * (gridDim.x >> 3) + (bid >> 3);
    const int tile_n = bid % p.tiles_n;
    const int tile_m = bid / p.tiles_n;
    const int n0 = tile_n * BN;
    const int C = p.C, hw = p.hw;
    // NHWC tiling: rows m0.. are consecutive pixels.  PM tiling: row r is image img0 + r at
    // position pos (tiles of neighbouring positions are adjacent in the grid, they share
    // input lines in L2).
    const int P = hw * hw;
    const int pos = PM ? tile_m % P : 0;
    const int img0 = PM ? (tile_m / P) * BM : 0;
    const int m0 = PM ? 0 : tile_m * BM;
    const int tap_begin = SPLIT ? blockIdx.y * p.taps_per_split : 0;
    const int tap_end = SPLIT ? min(9, tap_begin + p.taps_per_split) : 9;
    // taps this block multiplies, 4 bits each (PM: only those inside the image)
    unsigned long long tap_list = 0;
    int ntaps = 0;
    for (int t = tap_begin; t < tap_end; ++t) {
        bool valid = true;
        if (PM) {
            const int hh = (pos >> p.log_hw) + t / 3 - 1, ww = (pos & (hw - 1)) + t % 3 - 1;
            valid = (unsigned)hh < (unsigned)hw && (unsigned)ww < (unsigned)hw;
        }
        if (valid) tap_list |= (unsigned long long)t << (4 * ntaps++);
    }
    const int steps = ntaps << p.log_chunks;

    // ---- staging.  Wave w streams rows [w*BM/4, (w+1)*BM/4) of the pixel tile and the same
    // share of the weight tile straight into LDS (buffer_load ... lds: 16 B per lane, 8 rows =
    // 1 KiB per wave instruction, no VGPR round trip).  The LDS image of such a load is
    // lane-linear, so the bank swizzle is applied on the SOURCE side: lane (row, slot)
    // fetches the 16-byte chunk slot ^ ((row >> 1) & 7) of its row.  Halo taps and rows past
    // M get an out-of-range offset, for which the buffer range check writes zeros.
    const int lrow = lane >> 3, slot = lane & 7;
    // Per lane only a fixed byte offset and a 9-bit tap mask; everything that changes from
    // step to step (tap shift, channel chunk) is wave-uniform and travels in the scalar offset
    // of the buffer load, so a step costs two VALU instructions per load (mask test + select).
    int pix_voff[A_IT];      // byte offset of (b,h,w, this lane's chunk slot), chunk 0
    unsigned tapmask[A_IT];  // bit t: tap t of this row reads inside the image (0: row past M)
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
        const int r = wave * (BM / NW) + i * 8 + lrow;
        const int m = PM ? (img0 + r) * P + pos : m0 + r;
        const bool row_ok = PM ? (img0 + r < p.batch) : (m < p.M);
        const int w_ = m & (hw - 1);
        const int h_ = (m >> p.log_hw) & (hw - 1);
        unsigned bits = 0;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int hh = h_ + t / 3 - 1, ww = w_ + t % 3 - 1;
            if ((unsigned)hh < (unsigned)hw && (unsigned)ww < (unsigned)hw) bits |= 1u << t;
        }
        tapmask[i] = row_ok ? bits : 0u;
        pix_voff[i] = (m * C + ((slot ^ ((r >> 1) & 7)) << 3)) * 2;
    }
    int w_off[B_IT];         // byte offset of (n, tap 0, chunk) in the weight matrix
#pragma unroll
    for (int i = 0; i < B_IT; ++i) {
        const int r = wave * (BN / NW) + i * 8 + lrow;
        w_off[i] = ((n0 + permuted_channel<WTN>(r)) * 9 * C + ((slot ^ ((r >> 1) & 7)) << 3)) * 2;
    }
    // The pixel descriptor starts (hw + 1) pixels BEFORE the tensor so that the scalar offset
    // (tap shift + that bias) is never negative; lanes whose tap is padding, or whose row is
    // past the end, use an offset beyond the descriptor's extent and receive zeros.
    const int halo_bytes = (hw + 1) * C * 2;
    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)((const char*)p.x - halo_bytes), 0, p.x_bytes + 2 * (unsigned)halo_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wrsrc =
        __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, p.w_bytes, 0x00020000);

    // K order: channel chunk outer, tap inner - nine consecutive steps re-read the same
    // 128-byte lines of neighbouring pixels, so all but the first come from L1/L2.
    int issue_cc = 0, issue_ti = 0;      // issue() is called for s = 0, 1, 2, ... in order
    auto issue = [&](int s) {
        const int cc = issue_cc;
        const int tap = (int)((tap_list >> (4 * issue_ti)) & 15);
        if (++issue_ti == ntaps) { issue_ti = 0; ++issue_cc; }
        const int dh = tap / 3 - 1, dw = tap % 3 - 1;
        const int xs = ((dh * hw + dw) * C + cc * 64) * 2 + halo_bytes;     // uniform, >= 0
        char* sA = smem + (s % NS) * STAGE_BYTES + wave * (BM / NW) * 128;
        char* sB = smem + (s % NS) * STAGE_BYTES + BM * 128 + wave * (BN / NW) * 128;
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
            const int voff = ((tapmask[i] >> tap) & 1u) ? pix_voff[i] : kOutOfRange;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xrsrc, (lds_ptr)(sA + i * 1024), 16, voff, xs,
                                                     0, 0);
        }
        const int wshift = (tap * C + cc * 64) * 2;
#pragma unroll
        for (int i = 0; i < B_IT; ++i) {
            // (the offset must be a named value: hipcc's host pass rejects the whole kernel
            // template when an expression is passed in this argument position)
            const int voff = w_off[i];
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc, (lds_ptr)(sB + i * 1024), 16, voff,
                                                     wshift, 0, DML_W_AUX);
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

    auto load_frags = [&](int stage, int kk, bf16x8 (&wf)[TN], bf16x8 (&xf)[TM]) {
        const char* base = smem + stage * STAGE_BYTES;
        const int koff = kk ? koff1 : koff0;
#pragma unroll
        for (int i = 0; i < TN; ++i)
            wf[i] = *(const bf16x8*)(base + b_row_byte + i * 16 * 128 + koff);
#pragma unroll
        for (int j = 0; j < TM; ++j)
            xf[j] = *(const bf16x8*)(base + a_row_byte + j * 16 * 128 + koff);
    };
    auto mma = [&](const bf16x8 (&wf)[TN], const bf16x8 (&xf)[TM]) {
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
            for (int j = 0; j < TM; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i], xf[j], acc[i][j], 0, 0, 0);
    };
    auto wait_step = [&](int s) {       // this wave's loads of step s have landed
        const int ahead = min(D - 1, steps - 1 - s);
        if (ahead >= 2) wait_vmcnt<2 * LPS>();
        else if (ahead == 1) wait_vmcnt<LPS>();
        else wait_vmcnt<0>();
    };

    // Pipeline.  Global -> LDS: the loads of steps s+1 .. s+D-1 stay in flight across the
    // barrier while step s is multiplied (counted vmcnt, raw s_barrier).  LDS -> registers:
    // a K step is two 32-wide halves; the fragments of the second half are requested before the
    // MFMAs of the first, and the fragments of the NEXT step's first half right after the
    // barrier, before the MFMAs of this step's second half - so every LDS round trip hides
    // behind matrix work instead of following the barrier, where all waves would wait for it
    // together.  Before the barrier a wave drains its LDS reads (lgkmcnt(0)): the stage it
    // just read is the one the loads issued after the barrier refill.
#pragma unroll
    for (int s = 0; s < D; ++s)
        if (s < steps) issue(s);
    // (the un-pooled BatchNorm backward sums of the epilogue want the layer's conv output values
    // of this tile: requested four K steps before the end, see bn_plain_bwd_prefetch)
    constexpr bool kBnBwd = EPI == EPI_NONE && !SPLIT && !PM;
    uint32_t bn_xv[kBnBwd ? TM : 1][2 * TN];
    const bool bn_plain = kBnBwd && p.bn_pre != nullptr && !p.bn_pooled;
    auto bn_prefetch = [&]() {
        if constexpr (kBnBwd)
            bn_plain_bwd_prefetch<TN, TM>(p, m0 + wm * WTM + (lane & 15),
                                          n0 + wn * WTN + (lane >> 4) * 4 * TN, bn_xv);
    };
    const int bn_fetch_step = steps > 5 ? steps - 5 : 0;
    if (steps > 0) {
        bf16x8 wfa[TN], xfa[TM], wfb[TN], xfb[TM];
        wait_step(0);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (D < steps) issue(D);
        load_frags(0, 0, wfa, xfa);
        // (the last step is peeled: with a conditional barrier half inside the loop, the
        // compiler's wait-count bookkeeping merges both paths and makes the second-half MFMAs
        // wait for the reads issued after the barrier)
        for (int s = 0; s + 1 < steps; ++s) {
            load_frags(s % NS, 1, wfb, xfb);
            mma(wfa, xfa);
            // lgkmcnt(0), vmcnt / expcnt untouched: the stage just read is refilled below
            __builtin_amdgcn_s_waitcnt(0xC07F);
            wait_step(s + 1);
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (s + 1 + D < steps) issue(s + 1 + D);
            if (bn_plain && s == bn_fetch_step) bn_prefetch();
            load_frags((s + 1) % NS, 0, wfa, xfa);
            mma(wfb, xfb);
        }
        if (bn_plain && steps < 2) bn_prefetch();      // (the loop above did not run)
        load_frags((steps - 1) % NS, 1, wfb, xfb);
        mma(wfa, xfa);
        mma(wfb, xfb);
    }

    // ---- BatchNorm sums for free, one partial row [2][N] per pixel tile (a fold kernel sums
    // the tiles).  Forward (p.stats, EPI_BIAS): column sums and sums of squares of this tile's
    // fp32 accumulators, i.e. of the output without its bias (the consumer shifts the mean by
    // the bias; the variance does not care).  Backward (p.bn_pre, a dgrad pass whose output is
    // the gradient of a BatchNorm + ReLU + pool layer's POOLED activation): that layer's dbeta /
    // dgamma sums - per output element the pooling winner is re-derived from the layer's conv
    // output (first maximum, strict comparisons, the fma of the forward pass), the bf16-rounded
    // gradient goes to it if it is positive.  Rows past the end add nothing.
    const bool bn_bwd = EPI == EPI_NONE && !SPLIT && !PM && p.bn_pre != nullptr;
    if (!SPLIT && p.stats && (EPI == EPI_BIAS || bn_bwd)) {
        float s1[TN][4], s2[TN][4];
        if (bn_bwd) {
            if constexpr (EPI == EPI_NONE && !SPLIT && !PM) {
                if (p.bn_pooled)
                    bn_pool_bwd_tile_sums<TN, TM>(acc, p, m0 + wm * WTM + l15,
                                                  n0 + wn * WTN + lq * 4 * TN, s1, s2);
                else
                    bn_plain_bwd_tile_sums<TN, TM>(acc, p, bn_xv, m0 + wm * WTM + l15,
                                                   n0 + wn * WTN + lq * 4 * TN, s1, s2);
            }
        } else {
#pragma unroll
            for (int i = 0; i < TN; ++i)
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    float a = 0.f, b = 0.f;
#pragma unroll
                    for (int j = 0; j < TM; ++j) {
                        const float v = acc[i][j][c];
                        a += v;
                        b += v * v;
                    }
                    s1[i][c] = row16_sum(a);
                    s2[i][c] = row16_sum(b);
                }
        }
        __syncthreads();                       // every wave is done with the operand stages
        float* red = (float*)smem;             // [WAVES_M][2][BN]
        if (l15 == 0) {
#pragma unroll
            for (int i = 0; i < TN; ++i)
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int col = wn * WTN + lq * 4 * TN + 4 * i + c;
                    red[(wm * 2 + 0) * BN + col] = s1[i][c];
                    red[(wm * 2 + 1) * BN + col] = s2[i][c];
                }
        }
        __syncthreads();
        for (int idx = tid; idx < 2 * BN; idx += NW * 64) {
            const int which = idx / BN, col = idx - which * BN;
            float t = 0.f;
#pragma unroll
            for (int q = 0; q < WAVES_M; ++q) t += red[(q * 2 + which) * BN + col];
            p.stats[((size_t)tile_m * 2 + which) * p.N + n0 + col] = t;
        }
    }

    // ---- epilogue: lane holds pixel m = mb + l15 and its 4*TN consecutive channels
    // n0 + wn*WTN + lq*4*TN + {0 .. 4*TN-1}, tile i contributing channels 4*i .. 4*i+3 of those
#pragma unroll
    for (int j = 0; j < TM; ++j) {
        const int row = wm * WTM + j * 16 + l15;
        const int m = PM ? (img0 + row) * P + pos : m0 + row;
        if (PM ? (img0 + row >= p.batch) : (m >= p.M)) continue;
        const int n = n0 + wn * WTN + lq * 4 * TN;
        const size_t off = (size_t)m * p.N + n;
        if (SPLIT) {
            float* dst = p.slab + (size_t)blockIdx.y * p.M * p.N + off;
#pragma unroll
            for (int i = 0; i < TN; ++i)
                *(float4*)(dst + 4 * i) =
                    float4{acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
        } else {
            f32x4 a[TN];
#pragma unroll
            for (int i = 0; i < TN; ++i) a[i] = acc[i][j];
            store_lane_channels<TN, EPI>(a, p.y, p.bias, p.mask, off, n);
        }
    }
}

// Explicit instantiations of every (tile, epilogue) variant the launcher can pick.
#define DML_IGEMM_INSTANTIATE(BM, BN, NS, PM, NW)                                                \
    template __global__ void conv3x3_igemm_kernel<BM, BN, 2, EPI_NONE, true, NS, PM, NW>(        \
        IgemmArgs);                                                                              \
    template __global__ void conv3x3_igemm_kernel<BM, BN, 2, EPI_NONE, false, NS, PM, NW>(       \
        IgemmArgs);                                                                              \
    template __global__ void conv3x3_igemm_kernel<BM, BN, 2, EPI_BIAS, false, NS, PM, NW>(       \
        IgemmArgs);                                                                              \
    template __global__ void conv3x3_igemm_kernel<BM, BN, 2, EPI_BIAS_RELU, false, NS, PM, NW>(  \
        IgemmArgs);                                                                              \
    template __global__ void conv3x3_igemm_kernel<BM, BN, 2, EPI_MASK, false, NS, PM, NW>(       \
        IgemmArgs);
DML_IGEMM_INSTANTIATE(128, 128, 3, false, 4)
DML_IGEMM_INSTANTIATE(128, 64, 3, false, 4)
DML_IGEMM_INSTANTIATE(64, 64, 4, false, 4)
DML_IGEMM_INSTANTIATE(128, 128, 3, true, 4)
DML_IGEMM_INSTANTIATE(128, 64, 3, true, 4)
DML_IGEMM_INSTANTIATE(64, 64, 4, true, 4)
DML_IGEMM_INSTANTIATE(128, 128, 3, false, 8)
DML_IGEMM_INSTANTIATE(128, 128, 4, false, 8)
DML_IGEMM_INSTANTIATE(128, 128, 3, true, 8)
DML_IGEMM_INSTANTIATE(128, 128, 4, true, 8)
DML_IGEMM_INSTANTIATE(128, 128, 2, false, 8)
DML_IGEMM_INSTANTIATE(128, 128, 2, true, 8)
#undef DML_IGEMM_INSTANTIATE

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

// Same sum over the split slabs, organised as a column reduction so the BatchNorm statistics
// of the (bias-free) result fall out of the pass: thread = 8 columns x strided rows.
__global__ __launch_bounds__(256) void splitk_finish_stats_kernel(
    const float* __restrict__ slab, uint16_t* __restrict__ y, const float* __restrict__ bias,
    float* __restrict__ stats, int M, int N, int splits, size_t slab_stride, int rows_per_block) {
    __shared__ float red[4096];
    float bv[8];
    {
        const int col0 = (threadIdx.x % (N / 8)) * 8;
#pragma unroll
        for (int i = 0; i < 8; ++i) bv[i] = bias[col0 + i];
    }
    column_partials(
        [&](int r, int cg, float (&a)[8], float (&b)[8]) {
            const size_t off = (size_t)r * N + cg * 8;
            float4 lo = *(const float4*)(slab + off), hi = *(const float4*)(slab + off + 4);
            for (int s = 1; s < splits; ++s) {
                const float4 l2 = *(const float4*)(slab + s * slab_stride + off);
                const float4 h2 = *(const float4*)(slab + s * slab_stride + off + 4);
                lo.x += l2.x; lo.y += l2.y; lo.z += l2.z; lo.w += l2.w;
                hi.x += h2.x; hi.y += h2.y; hi.z += h2.z; hi.w += h2.w;
            }
            const float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
            float o[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                a[i] += v[i];
                b[i] += v[i] * v[i];
                o[i] = v[i] + bv[i];
            }
            *(uint4*)(y + off) = pack8(o);
        },
        stats, M, N, rows_per_block, red);
}

int ilog2(int v) {
    int l = 0;
    while ((1 << l) < v) ++l;
    return l;
}

// rows of `stats` the split path's finishing kernel writes: two rows per row lane and block
// (the layers that split are small - the kernel is latency-bound, so many short blocks)
ColReducePlan finish_stats_plan(int M, int N) {
    const int lanes = 256 / (N / 8);
    int rpb = lanes * 2;
    int blocks = (M + rpb - 1) / rpb;
    if (blocks > 512) {
        rpb = ((M + 511) / 512 + lanes - 1) / lanes * lanes;
        blocks = (M + rpb - 1) / rpb;
    }
    return {blocks, rpb};
}

// Plan policy: the tall patch-resident tiles (default; DML_TALL_TILES=0 starts without them), or the
// streaming tiles only.  A run-time switch because the better one depends on what else occupies the
// compute units: a tall tile wants a CU's LDS almost to itself, so a communication library's
// persistent blocks can push a 256-block grid into a second round (bench.py measures both policies on
// the live job when there is more than one rank).
std::atomic<int> g_tall_tiles{-1};
std::atomic<int> g_tall_max_rows{-1};        // (-1: DML_TALL_MAX_ROWS, else 256)
}  // namespace igemm
bool conv_tall_tiles() {
    int v = igemm::g_tall_tiles.load(std::memory_order_relaxed);
    if (v < 0) {
        const char* e = getenv("DML_TALL_TILES");
        v = !(e && e[0] == '0');
        igemm::g_tall_tiles.store(v, std::memory_order_relaxed);
    }
    return v != 0;
}
void set_conv_tall_tiles(bool on) { igemm::g_tall_tiles.store(on ? 1 : 0, std::memory_order_relaxed); }
// tallest tall tile the launcher picks: 256 (default) or 128 pixels - twice the blocks per layer, so
// a grid loses less when some compute units are not available to it
int conv_tall_max_rows() {
    int v = igemm::g_tall_max_rows.load(std::memory_order_relaxed);
    if (v < 0) {
        const char* e = getenv("DML_TALL_MAX_ROWS");
        v = (e && atoi(e) == 128) ? 128 : 256;
        igemm::g_tall_max_rows.store(v, std::memory_order_relaxed);
    }
    return v;
}
void set_conv_tall_max_rows(int rows) {
    if (rows != 128 && rows != 256) throw std::invalid_argument("set_conv_tall_max_rows: 128 or 256");
    igemm::g_tall_max_rows.store(rows, std::memory_order_relaxed);
}
namespace igemm {

struct Plan {
    int bm, bn, splits;
    bool pm;
    int waves;      // 4, 8 (3 LDS stages), 9 = eight waves with 4 LDS stages (tuning),
                    // 10 = eight waves with 2 LDS stages (64 KiB: two blocks per CU),
                    // 16 = patch-resident kernel (eight waves for 128-row tiles, four for 64),
                    // 20 + v = tall patch-resident tiles of conv_patch.hip (BM x 64), variant v
};

// Tile / split choice, from per-layer timings on MI355X at batch 256 (tools/bench_conv.py).
// The kernels are bound by the L2 -> LDS feed (~22-28 B/clk/CU), so bytes per flop decide:
// the largest tile that still covers the chip wins - 128x128 run by eight waves once it
// gives one block per CU, else 128x64 / 64x64 with two resident blocks per CU; layers that
// cannot reach one block per CU split the taps.  Position-major tiles (see the kernel) only
// pay off together with the tap split.  The patch-resident kernel fetches 1.7x fewer bytes
// but measures within +-10 % of the streaming one (LDS traffic, reads plus DMA writes, is
// the next limit); it is used where it wins, the 64x64 plans.
Plan choose_plan(int batch, int hw, int N, int cin, const int* force = nullptr) {
    const int M = batch * hw * hw;
    // position-major tiles pay off where padding is a visible share of the taps and the
    // batch fills whole tiles
    auto pm_ok = [&](int bm) { return hw <= 8 && batch % bm == 0; };
    if (force && force[0] > 0) {
        const bool pm = force[3] < 0 ? pm_ok(force[0]) : force[3] > 0;
        return {force[0], force[1], force[2] > 0 ? force[2] : 1, pm, force[4] > 0 ? force[4] : 4};
    }
    auto blocks = [&](int bm, int bn) { return (N % bn) ? 0 : ((M + bm - 1) / bm) * (N / bn); };
    // Tall patch-resident tiles (conv_patch.hip) wherever they give every CU a block: they fetch
    // 2.7x (256 x 64) or 1.9x (128 x 64) fewer bytes per flop than the streaming 128 x 128 tile,
    // and the L2 -> LDS feed is what bounds these layers.  Measured alone at batch 256 (us, old ->
    // new): 64->128@16 15.0 -> 15.3 / dgrad 18.5 -> 11.3, 128->256@8 14.9 -> 12.0 / 16.6 -> 11.7,
    // 256->256@8 24.4 -> 20.0 / 25.7 -> 16.9, 256->512@4 17.1 -> 11.5 / 22.0 -> 16.6,
    // 512->512@4 28.8 -> 18.8 / 26.4 -> 18.2.  DML_TALL_TILES=0 restores the older plans.
    const bool tall_ok = conv_tall_tiles();
    // (tuning: DML_TALL_VARIANT="v256,v128" picks the kernel variant per tile height)
    static const int v256 = [] { const char* e = getenv("DML_TALL_VARIANT"); return e ? atoi(e) : 0; }();
    static const int v128 = [] {
        const char* e = getenv("DML_TALL_VARIANT");
        const char* c = e ? strchr(e, ',') : nullptr;
        return c ? atoi(c + 1) : 0;
    }();
    if (tall_ok && hw <= 16 && hw >= 2) {
        if (conv_tall_max_rows() >= 256 && blocks(256, 64) >= 256) return {256, 64, 1, false, 20 + v256};
        if (blocks(128, 64) >= 256) return {128, 64, 1, false, 20 + v128};
        // Half a block per CU (256 -> 512 @ 4x4 backward-data): still one block per tile.  Dividing
        // the channel chunks between two blocks per tile (fp32 slabs summed by the consumer, splits =
        // 2, kept as a forced plan) is 2 us faster alone and 3-7 us slower inside the step
        // (tools/tune_dgrad_plans.py): the unsplit block owns its pixel tile and leaves the previous
        // layer's BatchNorm backward sums in its epilogue, which saves that layer's reduction pass.
        if (blocks(128, 64) * 2 >= 256 && cin >= 256) return {128, 64, 1, false, 20 + v128};
    }
    // (two LDS stages and two blocks per CU beat three stages and one: a second block's main
    // loop covers the first one's prologue and epilogue, which are a third of a block's life at
    // 9-36 K steps - 64 -> 128 @ 16x16 forward 17.9 -> 14.6 us, 256 -> 256 @ 8x8 24.0 -> 22.1 us)
    if (blocks(128, 128) >= 256) return {128, 128, 1, false, 10};
    if (blocks(128, 64) >= 512) return {128, 64, 1, false, 4};
    const int b = blocks(64, 64);
    if (b > 256) return {64, 64, 1, false, 16};     // patch-resident: 5-12 % faster at this size
    if (blocks(128, 128) * 3 >= 192) return {128, 128, 3, pm_ok(128), 10};
    if (b >= 256) return {64, 64, 1, false, 4};
    if (b * 3 >= 128) return {64, 64, 3, pm_ok(64), 4};
    return {64, 64, 9, pm_ok(64), 4};
}

template <int BM, int BN, int WAVES_M, int EPI, bool SPLIT, int NS, bool PM, int NW>
void launch_one(const IgemmArgs& a, dim3 grid, hipStream_t stream) {
    constexpr int lds = NS * (BM + BN) * 128;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(
            (const void*)conv3x3_igemm_kernel<BM, BN, WAVES_M, EPI, SPLIT, NS, PM, NW>,
            hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        attr_set = true;
    }
    hipLaunchKernelGGL((conv3x3_igemm_kernel<BM, BN, WAVES_M, EPI, SPLIT, NS, PM, NW>), grid,
                       dim3(NW * 64), lds, stream, a);
}

// LDS stages per tile: 128x128 -> 3 x 32 KiB (one block per CU), 128x64 -> 3 x 24 KiB and
// 64x64 -> 4 x 16 KiB (two blocks per CU).
template <int BM, int BN, int WAVES_M, int NS, bool PM, int NW>
void launch_cfg(IgemmArgs a, int epi, int splits, hipStream_t stream) {
    a.img_blocks = (a.batch + BM - 1) / BM;
    const int tiles_m = PM ? a.hw * a.hw * a.img_blocks : (a.M + BM - 1) / BM;
    const dim3 grid(tiles_m * a.tiles_n, splits);
    if (splits > 1) {
        launch_one<BM, BN, WAVES_M, EPI_NONE, true, NS, PM, NW>(a, grid, stream);
        return;
    }
    switch (epi) {
        case EPI_NONE:
            launch_one<BM, BN, WAVES_M, EPI_NONE, false, NS, PM, NW>(a, grid, stream);
            break;
        case EPI_BIAS:
            launch_one<BM, BN, WAVES_M, EPI_BIAS, false, NS, PM, NW>(a, grid, stream);
            break;
        case EPI_BIAS_RELU:
            launch_one<BM, BN, WAVES_M, EPI_BIAS_RELU, false, NS, PM, NW>(a, grid, stream);
            break;
        default: launch_one<BM, BN, WAVES_M, EPI_MASK, false, NS, PM, NW>(a, grid, stream); break;
    }
}

template <int BM, int BN, int WAVES_M, int NS, int NW>
void launch_layout(const IgemmArgs& a, int epi, int splits, bool pm, hipStream_t stream) {
    if (pm) launch_cfg<BM, BN, WAVES_M, NS, true, NW>(a, epi, splits, stream);
    else launch_cfg<BM, BN, WAVES_M, NS, false, NW>(a, epi, splits, stream);
}


// ------------------------------------------------------------------ patch-resident variant
// The kernel above re-fetches every input pixel once per tap: nine of the ten K steps of a
// channel chunk read pixels some other step of the same block has read already, and the
// L2 -> LDS feed (not the matrix cores) is what bounds it.  Here the block keeps, per
// 64-channel chunk, ONE copy of all the pixels its BM outputs touch - the tile's pixels plus
// hw+1 pixels on either side, BM + 2*hw + 2 rows of 128 bytes - and the nine taps read it at
// nine row shifts (in NHWC the (dh, dw) neighbour of pixel m is pixel m + dh*hw + dw).
// Only the weights still stream per tap, so the bytes fetched per chunk drop from
// 9*(BM+BN)*128 to (BM + 2hw + 2 + 9*BN)*128 (1.7x fewer at 128x128).
//
// A shifted row that falls outside the image (left/right wrap, another image above/below)
// holds some other pixel's data, so each lane carries a 9-bit validity mask per 16-row tile
// and zeroes its fragment for the taps that are padding.
//
// Pipeline: weights in a 3-stage ring exactly as above (prefetch distance 2); the patch of
// chunk c+1 is requested during tap 0 of chunk c into the other of two patch buffers.  The
// nine taps of a chunk are unrolled, so every s_waitcnt count is a compile-time constant even
// though tap 0 issues extra loads.
template <int BM, int BN, int NW, int EPI>
__global__ __launch_bounds__(NW * 64) void conv3x3_patch_kernel(IgemmArgs p) {
    raise_stream_signal(p);
    constexpr int WAVES_M = 2, WAVES_N = NW / 2;
    constexpr int WTM = BM / WAVES_M, WTN = BN / WAVES_N;
    constexpr int TM = WTM / 16, TN = WTN / 16;
    constexpr int B_IT = BN / (8 * NW);                 // weight wave-loads per step and wave
    constexpr int PGROUPS = (BM + 66 + 7) / 8;          // 8-row groups of the largest patch (hw 32)
    constexpr int P_IT = (PGROUPS + NW - 1) / NW;       // patch wave-loads per chunk and wave
    constexpr int PBUF = P_IT * NW * 1024;              // bytes per patch buffer
    constexpr int WSTAGE = BN * 128;
    static_assert(B_IT >= 1 && TM >= 1 && TN >= 1, "tile too small for this many waves");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const patch0 = smem;
    char* const wring = smem + 2 * PBUF;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave % WAVES_M, wn = wave / WAVES_M;
    int bid = blockIdx.x;       // XCD-contiguous tile ranges, as in the streaming kernel
    if ((gridDim.x & 7) == 0) bid = (bid & 7) * (gridDim.x >> 3) + (bid >> 3);
    const int tile_n = bid % p.tiles_n, tile_m = bid / p.tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int C = p.C, hw = p.hw;
    const int chunks = 1 << p.log_chunks;
    const int halo = hw + 1;

    const int lrow = lane >> 3, slot = lane & 7;
    int w_off[B_IT];
#pragma unroll
    for (int i = 0; i < B_IT; ++i) {
        const int r = wave * (BN / NW) + i * 8 + lrow;
        w_off[i] = ((n0 + permuted_channel<WTN>(r)) * 9 * C + ((slot ^ ((r >> 1) & 7)) << 3)) * 2;
    }
    int p_off[P_IT];        // byte offset of this lane's chunk of patch row (group, lrow), chunk 0
#pragma unroll
    for (int i = 0; i < P_IT; ++i) {
        const int prow = (wave + NW * i) * 8 + lrow;
        const int pix = m0 - halo + prow;
        const bool ok = prow < BM + 2 * halo && pix >= 0 && pix < p.M;
        p_off[i] = ok ? (pix * C + ((slot ^ ((prow >> 1) & 7)) << 3)) * 2 : kOutOfRange;
    }
    const __amdgpu_buffer_rsrc_t xrsrc =
        __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wrsrc =
        __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, p.w_bytes, 0x00020000);

    auto issue_patch = [&](int cc) {
        char* dst = patch0 + (cc & 1) * PBUF + wave * 1024;
#pragma unroll
        for (int i = 0; i < P_IT; ++i) {
            const int voff = p_off[i] == kOutOfRange ? kOutOfRange : p_off[i] + cc * 128;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xrsrc, (lds_ptr)(dst + i * NW * 1024), 16, voff,
                                                     0, 0, 0);
        }
    };
    auto issue_w = [&](int cc, int tap, int stage) {
        char* dst = wring + stage * WSTAGE + wave * (BN / NW) * 128;
        const int wshift = (tap * C + cc * 64) * 2;
#pragma unroll
        for (int i = 0; i < B_IT; ++i) {
            const int voff = w_off[i] + wshift;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc, (lds_ptr)(dst + i * 1024), 16, voff, 0, 0,
                                                     0);
        }
    };

    f32x4 acc[TN][TM];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int l15 = lane & 15, lq = lane >> 4;
    const int sw = l15 >> 1;
    const int koff0 = ((0 + lq) ^ sw) << 4, koff1 = ((4 + lq) ^ sw) << 4;
    const int b_row_byte = (wn * WTN + l15) * 128;          // weight tile (MFMA A operand)
    int prow0[TM];           // patch row of this lane's pixel for the centre tap
    unsigned tapmask[TM];    // bit t: tap t of this pixel lies inside the image
#pragma unroll
    for (int j = 0; j < TM; ++j) {
        const int r = wm * WTM + j * 16 + l15;
        prow0[j] = r + halo;
        const int m = m0 + r;
        const int w_ = m & (hw - 1), h_ = (m >> p.log_hw) & (hw - 1);
        unsigned bits = 0;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int hh = h_ + t / 3 - 1, ww = w_ + t % 3 - 1;
            if ((unsigned)hh < (unsigned)hw && (unsigned)ww < (unsigned)hw) bits |= 1u << t;
        }
        tapmask[j] = m < p.M ? bits : 0u;     // rows past the end multiply zeros (statistics!)
    }

    auto compute = [&](int stage, int cc, int tap) {
        const char* wbase = wring + stage * WSTAGE;
        const char* pbase = patch0 + (cc & 1) * PBUF;
        const int shift = (tap / 3 - 1) * hw + (tap % 3 - 1);
        int rowb[TM], key[TM];
        bool live[TM];
#pragma unroll
        for (int j = 0; j < TM; ++j) {
            const int row = prow0[j] + shift;
            rowb[j] = row * 128;
            key[j] = (row >> 1) & 7;
            live[j] = (tapmask[j] >> tap) & 1u;
        }
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int koff = kk ? koff1 : koff0;
            bf16x8 wf[TN], xf[TM];
#pragma unroll
            for (int i = 0; i < TN; ++i)
                wf[i] = *(const bf16x8*)(wbase + b_row_byte + i * 16 * 128 + koff);
#pragma unroll
            for (int j = 0; j < TM; ++j) {
                const int ko = ((kk * 4 + lq) ^ key[j]) << 4;
                const u32x4 raw = *(const u32x4*)(pbase + rowb[j] + ko);
                const u32x4 z = live[j] ? raw : u32x4{0u, 0u, 0u, 0u};
                xf[j] = __builtin_bit_cast(bf16x8, z);
            }
#pragma unroll
            for (int i = 0; i < TN; ++i)
#pragma unroll
                for (int j = 0; j < TM; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i], xf[j], acc[i][j],
                                                                        0, 0, 0);
        }
    };

    // (un-pooled BatchNorm backward sums in the epilogue: the layer's conv output values of this
    // tile are requested at the start of the last chunk, see bn_plain_bwd_prefetch)
    constexpr bool kBnBwd = EPI == EPI_NONE;
    uint32_t bn_xv[kBnBwd ? TM : 1][2 * TN];
    const bool bn_plain = kBnBwd && p.bn_pre != nullptr && !p.bn_pooled;
    issue_patch(0);
    issue_w(0, 0, 0);
    issue_w(0, 1, 1);
    for (int cc = 0; cc < chunks; ++cc) {
        const bool last = cc + 1 == chunks;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            // loads issued after the weights of this step: the next step's weights, and the next
            // chunk's patch when tap 0 of this chunk lies in between
            constexpr int kPatchAfter = P_IT;
            if (last) {
                if (t == 8) wait_vmcnt<0>();
                else wait_vmcnt<B_IT>();
            } else {
                if (t == 1 || t == 2) wait_vmcnt<B_IT + kPatchAfter>();
                else wait_vmcnt<B_IT>();
            }
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            {   // weights of step s + 2
                const int t2 = (t + 2) % 9, c2 = cc + (t + 2) / 9;
                if (c2 < chunks) issue_w(c2, t2, (t + 2) % 3);
            }
            if (t == 0 && !last) issue_patch(cc + 1);
            if constexpr (kBnBwd)
                if (bn_plain && t == 0 && last)
                    bn_plain_bwd_prefetch<TN, TM>(p, m0 + wm * WTM + l15,
                                                  n0 + wn * WTN + lq * 4 * TN, bn_xv);
            compute(t % 3, cc, t);
        }
    }

    const bool bn_bwd = EPI == EPI_NONE && p.bn_pre != nullptr;
    if (p.stats && (EPI == EPI_BIAS || bn_bwd)) {
        float s1[TN][4], s2[TN][4];
        if (bn_bwd) {
            if constexpr (EPI == EPI_NONE) {
                if (p.bn_pooled)
                    bn_pool_bwd_tile_sums<TN, TM>(acc, p, m0 + wm * WTM + l15,
                                                  n0 + wn * WTN + lq * 4 * TN, s1, s2);
                else
                    bn_plain_bwd_tile_sums<TN, TM>(acc, p, bn_xv, m0 + wm * WTM + l15,
                                                   n0 + wn * WTN + lq * 4 * TN, s1, s2);
            }
        } else {
#pragma unroll
            for (int i = 0; i < TN; ++i)
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    float a = 0.f, b = 0.f;
#pragma unroll
                    for (int j = 0; j < TM; ++j) {
                        const float v = acc[i][j][c];
                        a += v;
                        b += v * v;
                    }
                    s1[i][c] = row16_sum(a);
                    s2[i][c] = row16_sum(b);
                }
        }
        __syncthreads();
        float* red = (float*)smem;             // [WAVES_M][2][BN]
        if (l15 == 0) {
#pragma unroll
            for (int i = 0; i < TN; ++i)
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int col = wn * WTN + lq * 4 * TN + 4 * i + c;
                    red[(wm * 2 + 0) * BN + col] = s1[i][c];
                    red[(wm * 2 + 1) * BN + col] = s2[i][c];
                }
        }
        __syncthreads();
        for (int idx = tid; idx < 2 * BN; idx += NW * 64) {
            const int which = idx / BN, col = idx - which * BN;
            float t = 0.f;
#pragma unroll
            for (int q = 0; q < WAVES_M; ++q) t += red[(q * 2 + which) * BN + col];
            p.stats[((size_t)tile_m * 2 + which) * p.N + n0 + col] = t;
        }
    }

#pragma unroll
    for (int j = 0; j < TM; ++j) {
        const int m = m0 + wm * WTM + j * 16 + l15;
        if (m >= p.M) continue;
        const int n = n0 + wn * WTN + lq * 4 * TN;
        f32x4 a[TN];
#pragma unroll
        for (int i = 0; i < TN; ++i) a[i] = acc[i][j];
        store_lane_channels<TN, EPI>(a, p.y, p.bias, p.mask, (size_t)m * p.N + n, n);
    }
}

#define DML_PATCH_INSTANTIATE(BM, BN, NW)                                                  \
    template __global__ void conv3x3_patch_kernel<BM, BN, NW, EPI_NONE>(IgemmArgs);        \
    template __global__ void conv3x3_patch_kernel<BM, BN, NW, EPI_BIAS>(IgemmArgs);        \
    template __global__ void conv3x3_patch_kernel<BM, BN, NW, EPI_BIAS_RELU>(IgemmArgs);   \
    template __global__ void conv3x3_patch_kernel<BM, BN, NW, EPI_MASK>(IgemmArgs);
DML_PATCH_INSTANTIATE(128, 128, 8)
DML_PATCH_INSTANTIATE(128, 64, 8)
DML_PATCH_INSTANTIATE(64, 64, 4)
#undef DML_PATCH_INSTANTIATE

template <int BM, int BN, int NW, int EPI>
void launch_patch_one(const IgemmArgs& a, hipStream_t stream) {
    constexpr int pgroups = (BM + 66 + 7) / 8;
    constexpr int p_it = (pgroups + NW - 1) / NW;
    constexpr int lds = 2 * p_it * NW * 1024 + 3 * BN * 128;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)conv3x3_patch_kernel<BM, BN, NW, EPI>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        attr_set = true;
    }
    const dim3 grid(((a.M + BM - 1) / BM) * a.tiles_n);
    hipLaunchKernelGGL((conv3x3_patch_kernel<BM, BN, NW, EPI>), grid, dim3(NW * 64), lds, stream, a);
}

template <int BM, int BN, int NW>
void launch_patch(const IgemmArgs& a, int epi, hipStream_t stream) {
    switch (epi) {
        case EPI_NONE: launch_patch_one<BM, BN, NW, EPI_NONE>(a, stream); break;
        case EPI_BIAS: launch_patch_one<BM, BN, NW, EPI_BIAS>(a, stream); break;
        case EPI_BIAS_RELU: launch_patch_one<BM, BN, NW, EPI_BIAS_RELU>(a, stream); break;
        default: launch_patch_one<BM, BN, NW, EPI_MASK>(a, stream); break;
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

}  // namespace igemm
using namespace igemm;

size_t conv3x3_igemm_workspace_elems(ConvShape s) {
    const int M = s.batch * s.hw * s.hw;
    // under either plan policy (it can change at run time), and room for the widest split so a
    // forced plan (tuning) never outgrows the buffer
    const bool was = conv_tall_tiles();
    bool splits = false;
    for (bool tall : {true, false}) {
        set_conv_tall_tiles(tall);
        splits = splits || choose_plan(s.batch, s.hw, s.cout, s.cin).splits > 1;
    }
    set_conv_tall_tiles(was);
    return splits ? (size_t)9 * M * s.cout : 0;
}

int conv3x3_igemm_stats_rows(ConvShape s, const int* force_plan) {
    check_shape(s, "conv3x3_igemm_stats_rows");
    const int M = s.batch * s.hw * s.hw;
    const Plan pl = choose_plan(s.batch, s.hw, s.cout, s.cin, force_plan);
    if (pl.splits > 1) return finish_stats_plan(M, s.cout).blocks;
    return pl.pm ? s.hw * s.hw * ((s.batch + pl.bm - 1) / pl.bm) : (M + pl.bm - 1) / pl.bm;
}

void conv3x3_igemm(const bf16_raw* x, const bf16_raw* w, bf16_raw* y, const float* bias,
                   const bf16_raw* mask, float* workspace, size_t workspace_elems, ConvShape s,
                   ConvEpilogue epi, hipStream_t stream, const int* force_plan, float* stats,
                   SplitSlabs* defer, const BnPoolReduce* bn_pool, BnPartials* bn_done,
                   const StreamSignal* signal) {
    check_shape(s, "conv3x3_igemm");
    if (bn_done) *bn_done = BnPartials{};
    if (bn_pool && (epi != EPI_NONE || stats || !bn_done || !bn_pool->pre || !bn_pool->partials))
        throw std::invalid_argument("conv3x3_igemm: BatchNorm backward sums need the plain epilogue, "
                                    "a partials buffer and bn_done");
    if (defer) {
        if (epi != EPI_NONE || stats)
            throw std::invalid_argument("conv3x3_igemm: a deferred slab sum needs the plain epilogue");
        *defer = SplitSlabs{};
    }
    if ((epi == EPI_BIAS || epi == EPI_BIAS_RELU) && !bias)
        throw std::invalid_argument("conv3x3_igemm: bias epilogue without bias");
    if (epi == EPI_MASK && !mask) throw std::invalid_argument("conv3x3_igemm: mask missing");
    IgemmArgs a;
    a.x = x; a.w = w; a.y = y; a.bias = bias; a.mask = mask; a.slab = workspace;
    a.stats = stats;
    a.bn_pre = nullptr; a.bn_gamma = a.bn_beta = a.bn_mean = a.bn_rstd = nullptr;
    a.bn_pooled = 1;
    a.signal = signal ? signal->flag : nullptr;
    a.signal_value = signal ? signal->value : 0u;
    if (stats && epi != EPI_BIAS)
        throw std::invalid_argument("conv3x3_igemm: statistics come with the bias-only epilogue");
    a.M = s.batch * s.hw * s.hw; a.N = s.cout; a.C = s.cin;
    a.hw = s.hw; a.log_hw = ilog2(s.hw); a.log_chunks = ilog2(s.cin / 64);
    a.batch = s.batch; a.img_blocks = 0;
    Plan pl = choose_plan(s.batch, s.hw, a.N, s.cin, force_plan);
    // The two-stage form is a forward choice: backward-data runs next to the weight-gradient
    // kernels of the side stream, and there one block per CU with a deeper ring measures better
    // inside the step (0.637 against 0.648 ms) although two blocks per CU win stand-alone.
    if (!force_plan && pl.waves == 10 && (epi == EPI_NONE || epi == EPI_MASK))
        pl.waves = pl.splits > 1 ? 9 : 8;
    const bool tall = pl.waves >= 20;       // the patch-resident tall tiles of conv_patch.hip
    if (tall) {
        if (pl.splits < 1 || (pl.splits & (pl.splits - 1)) || pl.splits > s.cin / 64 || pl.pm ||
            a.N % pl.bn || !conv3x3_pr_supported(pl.bm, pl.bn, s.hw, pl.waves - 20))
            throw std::invalid_argument("conv3x3_igemm: unsupported patch-resident tile plan");
    } else if (!((pl.bm == 128 || pl.bm == 64) && (pl.bn == 128 || pl.bn == 64) && a.N % pl.bn == 0 &&
          (pl.splits == 1 || pl.splits == 3 || pl.splits == 9)) || (pl.bm == 64 && pl.bn == 128))
        throw std::invalid_argument("conv3x3_igemm: unsupported tile plan");
    a.x_bytes = (unsigned)((size_t)a.M * a.C * 2);
    a.w_bytes = (unsigned)((size_t)a.N * 9 * a.C * 2);
    a.tiles_n = a.N / pl.bn;
    a.taps_per_split = 9 / pl.splits;
    if (pl.splits > 1) {
        const size_t need = (size_t)pl.splits * a.M * a.N;
        if (!workspace || workspace_elems < need)
            throw std::invalid_argument("conv3x3_igemm: workspace too small for the tap split");
    }
    // BatchNorm backward sums in the epilogue: plans whose block owns a whole pixel tile, NHWC rows
    if (bn_pool && pl.splits == 1 && !pl.pm) {
        a.bn_pre = bn_pool->pre;
        a.bn_gamma = bn_pool->gamma; a.bn_beta = bn_pool->beta;
        a.bn_mean = bn_pool->mean; a.bn_rstd = bn_pool->rstd;
        a.bn_pooled = bn_pool->pooled ? 1 : 0;
        a.stats = bn_pool->partials;
        *bn_done = BnPartials{bn_pool->partials, (a.M + pl.bm - 1) / pl.bm, nullptr};
    }
    if (tall) {
        launch_pr(a, pl.bm, pl.bn, pl.waves - 20, epi, pl.splits, stream);
    } else if (pl.waves == 16) {
        if (pl.splits != 1 || pl.pm || pl.bm == pl.bn - 64)
            throw std::invalid_argument("conv3x3_igemm: the patch kernel has no split / "
                                        "position-major form");
        if (pl.bm == 128 && pl.bn == 128) launch_patch<128, 128, 8>(a, epi, stream);
        else if (pl.bm == 128 && pl.bn == 64) launch_patch<128, 64, 8>(a, epi, stream);
        else launch_patch<64, 64, 4>(a, epi, stream);
    } else if (pl.bm == 128 && pl.bn == 128 && pl.waves == 8)
        launch_layout<128, 128, 2, 3, 8>(a, epi, pl.splits, pl.pm, stream);
    else if (pl.bm == 128 && pl.bn == 128 && pl.waves == 9)
        launch_layout<128, 128, 2, 4, 8>(a, epi, pl.splits, pl.pm, stream);
    else if (pl.bm == 128 && pl.bn == 128 && pl.waves == 10)    // two LDS stages: two blocks per CU
        launch_layout<128, 128, 2, 2, 8>(a, epi, pl.splits, pl.pm, stream);
    else if (pl.bm == 128 && pl.bn == 128)
        launch_layout<128, 128, 2, 3, 4>(a, epi, pl.splits, pl.pm, stream);
    else if (pl.bm == 128 && pl.bn == 64)
        launch_layout<128, 64, 2, 3, 4>(a, epi, pl.splits, pl.pm, stream);
    else
        launch_layout<64, 64, 2, 4, 4>(a, epi, pl.splits, pl.pm, stream);
    if (pl.splits > 1 && stats) {
        const ColReducePlan fp = finish_stats_plan(a.M, a.N);
        hipLaunchKernelGGL(splitk_finish_stats_kernel, dim3(fp.blocks), dim3(256), 0, stream,
                           (const float*)workspace, y, bias, stats, a.M, a.N, pl.splits,
                           (size_t)a.M * a.N, fp.rows_per_block);
    } else if (pl.splits > 1 && defer) {
        *defer = SplitSlabs{workspace, pl.splits, (size_t)a.M * a.N};
    } else if (pl.splits > 1) {
        const size_t total4 = (size_t)a.M * a.N / 4;
        hipLaunchKernelGGL(splitk_finish_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256),
                           0, stream, workspace, y, bias, mask, total4, a.N, pl.splits,
                           (size_t)a.M * a.N, (int)epi);
    }
}

}  // namespace dml
