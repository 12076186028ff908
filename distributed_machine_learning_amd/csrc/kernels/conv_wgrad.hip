// conv3x3 backward-weight as an MFMA implicit GEMM (one tap per block), the dispatcher between
// it and the tap-resident kernel (conv_wgrad9.hip), and the weight repack for dgrad.
//
// Reference op: the weight / bias gradient autograd computes for nn.Conv2d
// (/root/reference/part1/model.py:18-23) inside loss.backward()
// (/root/reference/part1/main.py:44).
//
//   dw[co][tap][ci] = sum_p dy[p][co] * x[p (+) tap][ci]            p = (b, h, w)
//
// GEMM view per tap: D[co][ci] = dyT[co][p] * xs[p][ci]; the reduction runs over pixels,
// and both operands are stored pixel-major ([p][channel], NHWC), i.e. K-strided for the
// matrix cores.  Tiles are staged in LDS exactly as they sit in memory ([64 px][channels])
// and the MFMA fragments are fetched with ds_read_b64_tr_b16, gfx950's transposing LDS
// read: per 16 lanes it takes a 4 (pixels) x 16 (channels) block and hands each lane the
// four pixels of one channel, so two reads give the 8 consecutive-K values a lane needs.
//
// CDNA4 mapping of the per-tap kernel
//   * block = 256 threads (4 wave64, 2x2), tile BCO x BCI channels of ONE tap, K step
//     64 pixels = 4 x v_mfma_f32_32x32x16_bf16 per 32x32 tile.
//   * 32x32 accumulators put 32 consecutive ci on lanes 0..31 of one register, so the
//     fp32 result leaves as 128-byte row segments.
//   * LDS rows are padded by 64 bytes (row stride = 2*channels + 64): the four pixel rows
//     and two 16-channel groups touched by one half-wave of a transposing read then hit
//     disjoint banks.
//   * Two LDS stages filled through registers (buffer loads, out-of-range offsets for the
//     padding), one barrier per K step.
//   * The pixel range is split over blockIdx.y until the grid covers the chip.  With a slab
//     every split stores its partial tile with plain stores and sum_slabs_batch adds them in a
//     fixed order (what the engines use: deterministic); without one the splits accumulate
//     into the pre-zeroed gradient with fp32 atomics.
#include <stdexcept>
#include <string>

#include "dml_common.h"
#include "dml_kernels.h"

namespace dml {

namespace {

typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

__device__ __forceinline__ bf16x8 tr_frag(const char* row_ptr, int row_stride) {
    // rows k..k+3 then k+4..k+7 of the block this lane addresses
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(row_ptr));
    const s16x4 hi =
        __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(row_ptr + 4 * row_stride));
    const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
}

struct WgradArgs {
    const uint16_t* x;
    const uint16_t* dy;
    float* dw;
    int M, Cin, Cout;
    int hw, log_hw;
    int tiles_ci, tiles_co;
    int steps_per_split;   // K steps (64 pixels each) per blockIdx.y
    int atomic;
    size_t split_stride;   // elements between the partial results of consecutive splits (slab mode)
    unsigned x_bytes, dy_bytes;   // extents for the buffer descriptors
};

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4w;
constexpr int kOutOfRangeW = (int)0x80000000;

// KS = pixels per K step: 64, or 32 - half the LDS (40 KB for a 128 x 128 tile), so that more
// blocks, or a block of another kernel, fit next to one on a CU.
// NW = waves per block: 4 (2 x 2 wave tiles), or 8 - two waves per SIMD of the CU a block mostly
// has to itself, so one wave's LDS and barrier waits hide behind the other's MFMAs (the wave tile
// shrinks to a quarter of the longer tile side: 50 % more LDS fragment reads per step).
template <int BCO, int BCI, int KS, int NW>
__global__ __launch_bounds__(NW * 64) void conv3x3_wgrad_kernel(WgradArgs p) {
    constexpr int SA = BCO * 2 + 64;          // dy tile row stride (bytes)
    constexpr int SB = BCI * 2 + 64;          // x  tile row stride
    constexpr int STAGE = KS * (SA + SB);
    constexpr int NT = NW * 64;
    // wave grid: 2 x 2, or with eight waves 2 x 4 (4 x 2 when the ci tile is the narrow one)
    constexpr int WAVES_CO = NW == 4 ? 2 : (BCI >= BCO ? 2 : 4), WAVES_CI = NW / WAVES_CO;
    constexpr int WCO = BCO / WAVES_CO, WCI = BCI / WAVES_CI;      // wave tile
    constexpr int TCO = WCO / 32, TCI = WCI / 32;
    constexpr int A_CH = BCO / 8, B_CH = BCI / 8;    // 16-byte chunks per row
    constexpr int A_IT = KS * A_CH / NT, B_IT = KS * B_CH / NT;
    static_assert(TCO >= 1 && TCI >= 1, "wave tile too small");
    static_assert(A_IT >= 1 && B_IT >= 1 && (KS == 32 || KS == 64), "K step too short for this tile");

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wco = wave % WAVES_CO, wci = wave / WAVES_CO;

    // Workgroups go round-robin to the 8 XCDs (own L2 each) in launch order.  Renumber so
    // that an XCD owns a contiguous range of pixel splits for ALL tiles: the dy / x pixels of a
    // split are then fetched into one L2 instead of eight.
    int lin = blockIdx.y * gridDim.x + blockIdx.x;
    const int nblk = gridDim.x * gridDim.y;
    if ((nblk & 7) == 0) lin = (lin & 7) * (nblk >> 3) + (lin >> 3);
    const int split = lin / gridDim.x;
    int t = lin - split * gridDim.x;
    const int tile_ci = t % p.tiles_ci; t /= p.tiles_ci;
    const int tile_co = t % p.tiles_co; t /= p.tiles_co;
    const int tap = t;
    const int co0 = tile_co * BCO, ci0 = tile_ci * BCI;
    const int dh = tap / 3 - 1, dw_ = tap % 3 - 1;
    const int hw = p.hw;

    const int total_steps = (p.M + KS - 1) / KS;
    const int s_begin = split * p.steps_per_split;
    const int s_end = min(total_steps, s_begin + p.steps_per_split);

    const int a_c = tid % A_CH, a_r = tid / A_CH;       // dy staging ownership
    const int b_c = tid % B_CH, b_r = tid / B_CH;       // x staging ownership
    constexpr int A_RSTEP = NT / A_CH, B_RSTEP = NT / B_CH;

    // Both operands are fetched through raw buffer descriptors; rows past the end and taps
    // that fall into the padding get an out-of-range offset, for which the hardware returns
    // zeros - no branches, no selects, loads stay in flight across the MFMAs.
    const __amdgpu_buffer_rsrc_t xrsrc =
        __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t dyrsrc =
        __builtin_amdgcn_make_buffer_rsrc((void*)p.dy, 0, p.dy_bytes, 0x00020000);
    u32x4w ra[A_IT], rb[B_IT];
    auto load_step = [&](int s) {
        const int pbase = s * KS;
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
            const int pix = pbase + a_r + i * A_RSTEP;
            const int voff = (pix < p.M) ? (pix * p.Cout + co0 + a_c * 8) * 2 : kOutOfRangeW;
            ra[i] = __builtin_amdgcn_raw_buffer_load_b128(dyrsrc, voff, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < B_IT; ++i) {
            const int pix = pbase + b_r + i * B_RSTEP;
            const int ww = (pix & (hw - 1)) + dw_;
            const int hh = ((pix >> p.log_hw) & (hw - 1)) + dh;
            const bool ok = pix < p.M && (unsigned)hh < (unsigned)hw && (unsigned)ww < (unsigned)hw;
            const int voff = ok ? ((pix + dh * hw + dw_) * p.Cin + ci0 + b_c * 8) * 2 : kOutOfRangeW;
            rb[i] = __builtin_amdgcn_raw_buffer_load_b128(xrsrc, voff, 0, 0);
        }
    };
    auto store_step = [&](int stage) {
        char* sA = smem + stage * STAGE;
        char* sB = sA + KS * SA;
#pragma unroll
        for (int i = 0; i < A_IT; ++i)
            *(u32x4w*)(sA + (a_r + i * A_RSTEP) * SA + a_c * 16) = ra[i];
#pragma unroll
        for (int i = 0; i < B_IT; ++i)
            *(u32x4w*)(sB + (b_r + i * B_RSTEP) * SB + b_c * 16) = rb[i];
    };

    f32x16 acc[TCO][TCI];
#pragma unroll
    for (int i = 0; i < TCO; ++i)
#pragma unroll
        for (int j = 0; j < TCI; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // transposing-read addressing: 16-lane group g = lane>>4 covers channels 16*(g&1).. of
    // the 32-wide tile and K rows 8*(g>>1)..; inside the group lane 4q+pp addresses row q,
    // channels 4pp..4pp+3.
    const int g = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3;
    const int frag_row = 8 * (g >> 1) + q;
    const int frag_col = 16 * (g & 1) + 4 * pp;
    const int a_off = frag_row * SA + (wco * WCO + frag_col) * 2;
    const int b_off = KS * SA + frag_row * SB + (wci * WCI + frag_col) * 2;

    auto compute = [&](int stage) {
        const char* base = smem + stage * STAGE;
#pragma unroll
        for (int ks = 0; ks < KS / 16; ++ks) {
            bf16x8 af[TCO], bfr[TCI];
#pragma unroll
            for (int i = 0; i < TCO; ++i)
                af[i] = tr_frag(base + a_off + ks * 16 * SA + i * 64, SA);
#pragma unroll
            for (int j = 0; j < TCI; ++j)
                bfr[j] = tr_frag(base + b_off + ks * 16 * SB + j * 64, SB);
#pragma unroll
            for (int i = 0; i < TCO; ++i)
#pragma unroll
                for (int j = 0; j < TCI; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j],
                                                                        0, 0, 0);
        }
    };

    if (s_begin < s_end) {
        load_step(s_begin);
        store_step(0);
        __syncthreads();
        for (int s = s_begin; s < s_end; ++s) {
            const bool more = s + 1 < s_end;
            const int st = (s - s_begin) & 1;
            if (more) load_step(s + 1);
            compute(st);
            if (more) store_step(st ^ 1);
            __syncthreads();
        }
    } else {
        return;
    }

    // D[co][ci]: lane -> ci = l & 31, register r -> co = (r&3) + 8*(r>>2) + 4*(l>>5)
    const int l31 = lane & 31, lh = lane >> 5;
#pragma unroll
    for (int i = 0; i < TCO; ++i) {
#pragma unroll
        for (int j = 0; j < TCI; ++j) {
            const int ci = ci0 + wci * WCI + j * 32 + l31;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co0 + wco * WCO + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                float* dst = p.dw + split * p.split_stride + ((size_t)co * 9 + tap) * p.Cin + ci;
                if (p.atomic) atomicAdd(dst, acc[i][j][r]);
                else *dst = acc[i][j][r];
            }
        }
    }
}

int ilog2w(int v) {
    int l = 0;
    while ((1 << l) < v) ++l;
    return l;
}

// Pixel-range splits, keeping >= 4 K steps per block.  Measured optimum on MI355X (batch
// 256, tools/bench_conv.py): with atomic accumulation ~1.1 blocks per CU (every further split
// adds |dw| of atomic traffic); with per-split slabs, where a split only costs plain stores,
// ~1.7 blocks per CU (432 blocks: two 80 KB-LDS blocks share most CUs).  `split_target`
// overrides the block target: when the kernel runs on a side stream next to a longer main
// chain, fewer splits (less slab traffic competing with the main chain) serve the step better.
struct SplitPlan { int splits, steps_per_split; };
// the split count before it is rounded to whole steps per split: an upper bound for every
// batch size up to the one M stands for
int plan_splits_bound(int M, int tiles, bool slab_mode, int split_target) {
    const int total_steps = (M + 63) / 64;
    const int target = split_target > 0 ? split_target : slab_mode ? 432 : 288;
    int splits = slab_mode ? (target + tiles / 2) / tiles : (target + tiles - 1) / tiles;
    // a short reduction under tiles that already cover half the chip (the 2x2 layers): the
    // partial slabs and their sum would cost more than the split saves
    if (slab_mode && tiles >= 128 && total_steps <= 16) splits = 1;
    const int max_splits = total_steps >= 4 ? total_steps / 4 : 1;
    if (splits > max_splits) splits = max_splits;
    return splits < 1 ? 1 : splits;
}
SplitPlan plan_splits(int M, int tiles, int force_splits, bool slab_mode = true,
                      int split_target = 0) {
    const int total_steps = (M + 63) / 64;
    int splits = plan_splits_bound(M, tiles, slab_mode, split_target);
    if (force_splits > 0) splits = force_splits < total_steps ? force_splits : total_steps;
    if (splits < 1) splits = 1;
    const int sps = (total_steps + splits - 1) / splits;
    return {(total_steps + sps - 1) / sps, sps};
}

int wgrad_tiles(const ConvShape& s) {
    const int bco = s.cout % 128 == 0 ? 128 : 64, bci = s.cin % 128 == 0 ? 128 : 64;
    return (s.cout / bco) * (s.cin / bci) * 9;
}

template <int BCO, int BCI, int KS, int NW>
void launch_wgrad_ks(WgradArgs a, hipStream_t stream, int force_splits, float* slab,
                     int split_target) {
    a.tiles_co = a.Cout / BCO;
    a.tiles_ci = a.Cin / BCI;
    const int tiles = a.tiles_co * a.tiles_ci * 9;
    const SplitPlan sp = plan_splits(a.M, tiles, force_splits, slab != nullptr, split_target);
    a.steps_per_split = sp.steps_per_split * (64 / KS);     // (the plan counts 64-pixel steps)
    a.split_stride = 0;
    a.atomic = sp.splits > 1 ? 1 : 0;
    if (slab && sp.splits > 1) {        // deterministic: one partial result per split, no atomics
        a.dw = slab;
        a.split_stride = (size_t)a.Cout * 9 * a.Cin;
        a.atomic = 0;
    }
    constexpr int lds = 2 * KS * ((BCO * 2 + 64) + (BCI * 2 + 64));
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)conv3x3_wgrad_kernel<BCO, BCI, KS, NW>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        attr_set = true;
    }
    hipLaunchKernelGGL((conv3x3_wgrad_kernel<BCO, BCI, KS, NW>), dim3(tiles, sp.splits),
                       dim3(NW * 64), lds, stream, a);
}

int g_kstep = 64;       // pixels per K step of the per-tap kernel (tuning: 32)
// waves per block of the per-tap kernel for tiles with a 128 side: alone on the GPU eight waves
// are 10-17 % faster wherever a CU holds one block, and inside the training step they are worth
// ~1 % together with the tuned splits below (tools/ab_waves.sh)
int g_waves = 8;

template <int BCO, int BCI>
void launch_wgrad(WgradArgs a, hipStream_t stream, int force_splits, float* slab,
                  int split_target) {
    if constexpr (BCO == 128 || BCI == 128) {
        if (g_waves == 8 && g_kstep == 64) {
            launch_wgrad_ks<BCO, BCI, 64, 8>(a, stream, force_splits, slab, split_target);
            return;
        }
    }
    if (g_kstep == 32) launch_wgrad_ks<BCO, BCI, 32, 4>(a, stream, force_splits, slab, split_target);
    else launch_wgrad_ks<BCO, BCI, 64, 4>(a, stream, force_splits, slab, split_target);
}

}  // namespace

namespace {
int g_variant = 0;
int g_target = 0;       // block target override (tuning), 0: the caller's
int g_bco = 0, g_stages = 2;    // tap-resident kernel: co tile (0: 128 where it divides), ring depth

// Which kernel serves a call of conv3x3_wgrad.  The engines run weight gradients on a side
// stream NEXT TO the backward-data chain, and there the per-tap kernel is the better neighbour:
// measured inside the VGG-11+BN step at batch 256 it gives 0.657 ms/step against 0.667-0.68 with
// the tap-resident kernel (any co tile, ring depth or block target), although that kernel is
// 18 % faster when it has the GPU to itself (121 against 148 us over the seven layers) - sharing
// CUs it keeps one wave per SIMD and loses more than the per-tap kernel's many small blocks.
// So the default is the per-tap kernel and the tap-resident one is opt-in (variant 2) or called
// directly (conv3x3_wgrad9, which the stand-alone op uses).
bool use_tap_resident(const ConvShape& s) {
    return g_variant == 2 && conv3x3_wgrad9_supported(s);
}
}  // namespace

void set_wgrad_variant(int variant) {
    if (variant < 0 || variant > 2) throw std::invalid_argument("set_wgrad_variant: 0, 1 or 2");
    g_variant = variant;
}
int wgrad_variant() { return g_variant; }
void set_wgrad9_config(int bco, int stages) {
    if ((bco != 0 && bco != 64 && bco != 128) || (stages != 2 && stages != 3))
        throw std::invalid_argument("set_wgrad9_config: bco in {0, 64, 128}, stages in {2, 3}");
    g_bco = bco;
    g_stages = stages;
}
void set_wgrad_kstep(int pixels) {
    if (pixels != 32 && pixels != 64) throw std::invalid_argument("set_wgrad_kstep: 32 or 64");
    g_kstep = pixels;
}
void set_wgrad_waves(int waves) {
    if (waves != 4 && waves != 8) throw std::invalid_argument("set_wgrad_waves: 4 or 8");
    g_waves = waves;
}
void set_wgrad_block_target(int blocks) {
    if (blocks < 0) throw std::invalid_argument("set_wgrad_block_target: negative");
    g_target = blocks;
}

int conv3x3_wgrad_splits(ConvShape s, int split_target) {
    if (g_target > 0) split_target = g_target;
    if (use_tap_resident(s))
        return conv3x3_wgrad9_splits(s, conv3x3_wgrad9_bco(s, g_bco), 0, split_target);
    return plan_splits(s.batch * s.hw * s.hw, wgrad_tiles(s), 0, true, split_target).splits;
}

// Pixel splits tuned INSIDE the training step (tools/tune_wgrad_splits.py: coordinate descent over
// the seven layers of VGG-11 at batch 256 on an MI355X, one bench run per candidate).  In the
// two-stream backward the side stream's chain of weight-gradient kernels sets the pace, each of
// them sharing the CUs with whatever the main chain runs at that moment, and the best split count
// per layer is not a function of the block count alone: the tuned lists give 0.638 against
// 0.653 ms/step with BatchNorm and 0.557 against 0.564 ms without.  Re-tuned with the tall conv tiles
// on the main chain (their neighbours changed): 0.5520 -> 0.5508 ms with BatchNorm, 0.4975 -> 0.4855
// without.  0: no entry for this shape.
int conv3x3_wgrad_tuned_splits(ConvShape s, bool batch_norm) {
    if (s.batch != 256) return 0;
    struct Entry { int hw, cin, cout, bn, plain; };
    static const Entry table[] = {
        {16, 64, 128, 40, 33}, {8, 128, 256, 10, 14}, {8, 256, 256, 10, 12},
        {4, 256, 512, 3, 5},   {4, 512, 512, 3, 3},   {2, 512, 512, 1, 1},
    };
    for (const Entry& e : table)
        if (e.hw == s.hw && e.cin == s.cin && e.cout == s.cout) return batch_norm ? e.bn : e.plain;
    return 0;
}

// the split count a forced value turns into (whole 64-pixel steps per split)
int conv3x3_wgrad_forced_splits(ConvShape s, int force_splits) {
    return plan_splits(s.batch * s.hw * s.hw, wgrad_tiles(s), force_splits, true, 0).splits;
}

int conv3x3_wgrad_max_splits(ConvShape s) {
    int m = 1;
    const int M = s.batch * s.hw * s.hw;
    const int steps = (M + 63) / 64, cap = steps >= 4 ? steps / 4 : 1;
    const int tiles = wgrad_tiles(s);
    for (int target : {0, 288, g_target}) {
        // per-tap plan without its short-reduction rule (which depends on the batch size)
        int a = ((target > 0 ? target : 432) + tiles / 2) / tiles;
        a = a > cap ? cap : a;
        m = a > m ? a : m;
        if (conv3x3_wgrad9_supported(s)) {
            for (int bco : {64, 128}) {
                if (s.cout % bco) continue;
                const int b = conv3x3_wgrad9_splits_bound(s, bco, target);
                m = b > m ? b : m;
            }
        }
    }
    return m;
}

void conv3x3_wgrad(const bf16_raw* x, const bf16_raw* dy, float* dw, ConvShape s,
                   hipStream_t stream, int force_splits, float* slab, int split_target) {
    if (g_target > 0) split_target = g_target;
    if (slab && force_splits == 0 && use_tap_resident(s)) {
        const int bco = conv3x3_wgrad9_bco(s, g_bco);
        conv3x3_wgrad9(x, dy, dw, s, stream, slab, bco,
                       conv3x3_wgrad9_splits(s, bco, 0, split_target), g_stages);
        return;
    }
    if (s.batch <= 0 || s.hw < 1 || s.hw > 32 || (s.hw & (s.hw - 1)) || s.cin % 64 ||
        s.cout % 64 || s.cin < 64 || s.cout < 64)
        throw std::invalid_argument("conv3x3_wgrad: needs hw power of two <= 32, cin % 64 == 0, "
                                    "cout % 64 == 0");
    WgradArgs a;
    a.x = x; a.dy = dy; a.dw = dw;
    a.M = s.batch * s.hw * s.hw; a.Cin = s.cin; a.Cout = s.cout;
    a.hw = s.hw; a.log_hw = ilog2w(s.hw);
    a.tiles_ci = a.tiles_co = 0; a.steps_per_split = 0; a.atomic = 0; a.split_stride = 0;
    if ((long long)a.M * (s.cin > s.cout ? s.cin : s.cout) >= (1LL << 30))
        throw std::invalid_argument("conv3x3_wgrad: tensor exceeds 2^31 bytes");
    a.x_bytes = (unsigned)((size_t)a.M * s.cin * 2);
    a.dy_bytes = (unsigned)((size_t)a.M * s.cout * 2);
    if (s.cout % 128 == 0) {
        if (s.cin % 128 == 0) launch_wgrad<128, 128>(a, stream, force_splits, slab, split_target);
        else launch_wgrad<128, 64>(a, stream, force_splits, slab, split_target);
    } else {        // 64-channel layers of the deeper VGG variants (64 -> 64 at 32x32)
        if (s.cin % 128 == 0) launch_wgrad<64, 128>(a, stream, force_splits, slab, split_target);
        else launch_wgrad<64, 64>(a, stream, force_splits, slab, split_target);
    }
}

// ------------------------------------------------------------------ weight repacks
// wd[ci][8 - tap][co] = wf[co][tap][ci]: a 64x64 bf16 transpose per (tap, tile) through LDS.
__device__ __forceinline__ void repack_tile(const uint16_t* __restrict__ wf,
                                            uint16_t* __restrict__ wd, int cout, int cin,
                                            int ci0, int co0, int tap, uint16_t (*tile)[66]) {
    const int tid = threadIdx.x;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int chunk = tid + 256 * i;          // 512 chunks of 8 elements
        const int r = chunk >> 3, c = (chunk & 7) * 8;
        const uint4 v = *(const uint4*)(wf + ((size_t)(co0 + r) * 9 + tap) * cin + ci0 + c);
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            tile[r][c + 2 * e] = (uint16_t)(w[e] & 0xffff);
            tile[r][c + 2 * e + 1] = (uint16_t)(w[e] >> 16);
        }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int chunk = tid + 256 * i;
        const int r = chunk >> 3, c = (chunk & 7) * 8;   // r = ci row, c = co offset
        uint32_t w[4];
#pragma unroll
        for (int e = 0; e < 4; ++e)
            w[e] = (uint32_t)tile[c + 2 * e][r] | ((uint32_t)tile[c + 2 * e + 1][r] << 16);
        *(uint4*)(wd + ((size_t)(ci0 + r) * 9 + (8 - tap)) * cout + co0 + c) =
            make_uint4(w[0], w[1], w[2], w[3]);
    }
}

__global__ __launch_bounds__(256) void repack_dgrad_kernel(const uint16_t* __restrict__ wf,
                                                           uint16_t* __restrict__ wd, int cout,
                                                           int cin) {
    __shared__ uint16_t tile[64][66];
    repack_tile(wf, wd, cout, cin, blockIdx.x * 64, blockIdx.y * 64, blockIdx.z, tile);
}

void repack_dgrad_weights(const bf16_raw* wf, bf16_raw* wd, int cout, int cin, hipStream_t stream) {
    if (cout % 64 || cin % 64) throw std::invalid_argument("repack_dgrad_weights: dims % 64 != 0");
    hipLaunchKernelGGL(repack_dgrad_kernel, dim3(cin / 64, cout / 64, 9), dim3(256), 0, stream, wf,
                       wd, cout, cin);
}

void repack_conv1_weights(const float* w, bf16_raw* w1, hipStream_t stream);

// All layers of a model in one launch: block -> (layer, ci tile, co tile, tap); optionally 16
// more blocks that lay out the first layer's weights (fp32 [64][27] -> bf16 [64][64], K padded).
__device__ __forceinline__ void repack_conv1_block(const float* __restrict__ w,
                                                   uint16_t* __restrict__ w1, int block) {
    const int idx = block * 256 + threadIdx.x;      // 64 * 64
    if (idx >= 4096) return;
    const int co = idx >> 6, k = idx & 63;
    const int tap = k >> 2, c = k & 3;
    float v = 0.f;
    if (tap < 9 && c < 3) v = w[co * 27 + tap * 3 + c];
    w1[idx] = float_to_bf16_bits(v);
}

__global__ __launch_bounds__(256) void repack_dgrad_batch_kernel(const uint16_t* __restrict__ src,
                                                                 uint16_t* __restrict__ dst,
                                                                 RepackJobs jobs,
                                                                 const float* __restrict__ conv1_w,
                                                                 uint16_t* __restrict__ conv1_w1,
                                                                 const float* __restrict__ copy_src,
                                                                 float* __restrict__ copy_dst,
                                                                 int copy_n) {
    __shared__ uint16_t tile[64][66];
    const int total = jobs.tile_start[jobs.n];
    if ((int)blockIdx.x >= total) {
        const int extra = blockIdx.x - total;
        if (conv1_w && extra < 16) {        // 16 trailing blocks: first-layer weight layout
            repack_conv1_block(conv1_w, conv1_w1, extra);
        } else {                            // then: a small fp32 copy riding in this launch
            const int i = (extra - (conv1_w ? 16 : 0)) * 256 + threadIdx.x;
            if (i < copy_n) copy_dst[i] = copy_src[i];
        }
        return;
    }
    int j = 0;
    while (j + 1 < jobs.n && (int)blockIdx.x >= jobs.tile_start[j + 1]) ++j;
    int t = blockIdx.x - jobs.tile_start[j];
    const int cin = jobs.cin[j], cout = jobs.cout[j];
    const int tci = cin / 64, tco = cout / 64;
    const int ci_t = t % tci; t /= tci;
    const int co_t = t % tco; t /= tco;
    repack_tile(src + jobs.src_off[j], dst + jobs.dst_off[j], cout, cin, ci_t * 64, co_t * 64, t,
                tile);
}

void repack_dgrad_weights_batch(const bf16_raw* src_base, bf16_raw* dst_base, RepackJobs jobs,
                                hipStream_t stream, const float* conv1_w, bf16_raw* conv1_w1,
                                const float* copy_src, float* copy_dst, size_t copy_n) {
    if (jobs.n <= 0) {
        if (conv1_w) repack_conv1_weights(conv1_w, conv1_w1, stream);
        if (copy_n)
            (void)hipMemcpyAsync(copy_dst, copy_src, copy_n * sizeof(float), hipMemcpyDeviceToDevice,
                                 stream);
        return;
    }
    if (jobs.n > RepackJobs::kMax) throw std::invalid_argument("repack batch: too many layers");
    int total = 0;
    for (int j = 0; j < jobs.n; ++j) {
        if (jobs.cout[j] % 64 || jobs.cin[j] % 64)
            throw std::invalid_argument("repack batch: dims % 64 != 0");
        jobs.tile_start[j] = total;
        total += (jobs.cin[j] / 64) * (jobs.cout[j] / 64) * 9;
    }
    jobs.tile_start[jobs.n] = total;
    const int copy_blocks = (int)((copy_n + 255) / 256);
    hipLaunchKernelGGL(repack_dgrad_batch_kernel, dim3(total + (conv1_w ? 16 : 0) + copy_blocks),
                       dim3(256), 0, stream, src_base, dst_base, jobs, conv1_w, conv1_w1, copy_src,
                       copy_dst, (int)copy_n);
}

__global__ __launch_bounds__(256) void repack_conv1_kernel(const float* __restrict__ w,
                                                           uint16_t* __restrict__ w1) {
    repack_conv1_block(w, w1, blockIdx.x);
}

void repack_conv1_weights(const float* w, bf16_raw* w1, hipStream_t stream) {
    hipLaunchKernelGGL(repack_conv1_kernel, dim3(16), dim3(256), 0, stream, w, w1);
}

}  // namespace dml
