// conv3x3 backward-weight, tap-resident: one block keeps a dy tile and an x tile in LDS and
// produces ALL NINE taps of a (BCO x 64) weight-gradient tile from them.
//
// Reference op: the weight gradient autograd computes for nn.Conv2d
// (/root/reference/part1/model.py:18-23) inside loss.backward()
// (/root/reference/part1/main.py:44).
//
//   dw[co][tap][ci] = sum_p dy[p][co] * x[p (+) tap][ci]            p = (b, h, w), NHWC
//
// The reduction runs over pixels.  For a step of 64 consecutive pixels the nine taps read the
// SAME dy rows and nine shifted views of (nearly) the same x rows, so a per-tap kernel fetches
// every operand nine times.  Here both tiles are fetched once per step and the shift is done
// in LDS addressing:
//
//   * the x tile is stored with a zero border - every image (hw <= 8) or image-row band
//     (hw >= 16) of the step is laid out as a (rows + 2) x (hw + 2) grid - so "pixel (h + dh,
//     w + dw) or zero" is the row (h + dh + 1) * (hw + 2) + (w + dw + 1) for every tap: one
//     per-lane base address, the tap and the K sub-step are instruction immediates;
//   * fragments come from ds_read_b64_tr_b16 (each lane names the LDS row it contributes, so
//     the row-major [pixel][channel] tiles feed the K-major MFMA operands directly);
//   * a wave owns 32 co x 32 ci x 9 taps = nine v_mfma_f32_32x32x16_bf16 accumulators; the dy
//     fragment of a K sub-step is read once and used by nine MFMAs;
//   * 2x2 maps (hw = 2) take the position-major route instead: with the 16 images of a step
//     as the K sub-step, a tap only multiplies the output positions whose input lies inside
//     the image - 16 MFMAs per step instead of 36, four fragments per operand, no border.
//
// Global -> LDS goes through LDS-destination buffer loads (a wave writes 64 x 16 bytes as one
// linear 1 KiB image; each lane fetches the global chunk that belongs in its slot, border and
// padding slots use an out-of-range offset and receive zeros) into an NSTAGE ring with
// counted vmcnt waits and one raw s_barrier per step.  LDS rows are padded by 64 bytes (80 in
// the position-major layout, where a lane group's four rows are 4 pixels apart) so the four
// rows a half-wave touches in one transposing read fall into disjoint banks.
//
// The pixel range is split over blockIdx.y; every split stores its partial tile into a slab
// of its own with plain stores and sum_slabs_batch adds them in a fixed order.
#include <stdexcept>
#include <string>

#include "dml_common.h"
#include "dml_kernels.h"

namespace dml {

namespace wg9 {

typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
typedef __attribute__((address_space(3))) void* lds_ptr;
constexpr int kOutOfRange = (int)0x80000000;    // beyond any descriptor extent: zeros come back

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

struct Args {
    const uint16_t* x;
    const uint16_t* dy;
    float* dw;
    int M, Cin, Cout;
    int tiles_ci, tiles_co;
    int steps_per_split;
    size_t split_stride;
    unsigned x_bytes, dy_bytes;
};

// Geometry of the x tile of one 64-pixel step.
template <int HW>
struct Geo {
    static constexpr bool PM = HW == 2;
    static constexpr int SR = HW <= 8 ? HW : 64 / HW;          // image rows per segment
    static constexpr int NSEG = HW <= 8 ? 64 / (HW * HW) : 1;   // segments (images) per step
    static constexpr int PW = PM ? HW : HW + 2, PH = PM ? HW : SR + 2;
    static constexpr int XR = PM ? 64 : NSEG * PH * PW;         // LDS rows
    static constexpr int XS = PM ? 208 : 192;                   // x row stride (bytes)
    static constexpr int XCPR = XS / 16;
    static constexpr bool HALO = HW > 8;      // neighbouring image rows above / below the band
    // row of the lane's pixel, K sub-step ks, upper half (hi): lane part + these immediates
    static constexpr int ks_rows(int ks) {
        return HW == 4 ? ks * 36 : HW == 8 ? ks * 20 : HW == 16 ? ks * 18
                                                                : (ks >> 1) * 34 + (ks & 1) * 16;
    }
    static constexpr int hi_rows = HW == 4 ? 6 : 4;
};

template <int HW, int BCO, int NSTAGE>
__global__ __launch_bounds__(BCO * 4) void conv3x3_wgrad9_kernel(Args p) {
    using G = Geo<HW>;
    constexpr int NW = BCO / 16;                    // waves: (BCO / 32) x 2
    constexpr int WCO = BCO / 32;
    constexpr int AS = G::PM ? BCO * 2 + 80 : BCO * 2 + 64;     // dy row stride (bytes)
    constexpr int ACPR = AS / 16;
    constexpr int A_INSTR = (64 * ACPR + 63) / 64;
    constexpr int X_INSTR = (G::XR * G::XCPR + 63) / 64;
    constexpr int A_IT = (A_INSTR + NW - 1) / NW, X_IT = (X_INSTR + NW - 1) / NW;
    constexpr int A_BYTES = A_INSTR * 1024, X_BYTES = X_INSTR * 1024;
    constexpr int STAGE = A_BYTES + X_BYTES;
    constexpr int DUMP = NSTAGE * STAGE;            // 1 KiB that absorbs the padding wave-loads
    constexpr int LOADS = A_IT + X_IT;              // wave-loads per step and wave

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wco = wave % WCO, wci = wave / WCO;

    // XCD-contiguous pixel ranges: the blocks of one split (all tiles) share an L2
    int lin = blockIdx.y * gridDim.x + blockIdx.x;
    const int nblk = gridDim.x * gridDim.y;
    if ((nblk & 7) == 0) lin = (lin & 7) * (nblk >> 3) + (lin >> 3);
    const int split = lin / gridDim.x;
    int t = lin - split * gridDim.x;
    const int tile_ci = t % p.tiles_ci;
    const int tile_co = t / p.tiles_ci;
    const int co0 = tile_co * BCO, ci0 = tile_ci * 64;

    const int total_steps = (p.M + 63) / 64;
    const int s_begin = split * p.steps_per_split;
    const int s_end = min(total_steps, s_begin + p.steps_per_split);
    if (s_begin >= s_end) return;

    const __amdgpu_buffer_rsrc_t xrsrc =
        __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t dyrsrc =
        __builtin_amdgcn_make_buffer_rsrc((void*)p.dy, 0, p.dy_bytes, 0x00020000);

    // ---- global -> LDS slots of this lane (step 0 of the tensor; a step adds 64 pixels)
    int a_off[A_IT], x_off[X_IT];
    int a_dst[A_IT], x_dst[X_IT];       // wave-uniform LDS byte offsets inside a stage
    unsigned halo_bits = 0;             // 2 bits per x slot: 1 = row above the band, 2 = below
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
        const int instr = i * NW + wave;
        const int L = instr * 64 + lane;
        const int r = L / ACPR, c = L - r * ACPR;
        const bool ok = instr < A_INSTR && r < 64 && c < BCO / 8;
        a_off[i] = ok ? (r * p.Cout + co0 + c * 8) * 2 : kOutOfRange;
        a_dst[i] = instr < A_INSTR ? instr * 1024 : DUMP;
    }
#pragma unroll
    for (int i = 0; i < X_IT; ++i) {
        const int instr = i * NW + wave;
        const int L = instr * 64 + lane;
        const int r = L / G::XCPR, c = L - r * G::XCPR;
        bool ok = instr < X_INSTR && r < G::XR && c < 8;
        int pix = r;
        if (!G::PM) {
            const int seg = r / (G::PH * G::PW), pr = r - seg * (G::PH * G::PW);
            const int ph = pr / G::PW, pw = pr - ph * G::PW;
            const int lr = ph - 1, w = pw - 1;
            ok = ok && w >= 0 && w < HW;
            if (G::HALO) {
                if (lr < 0) halo_bits |= 1u << (2 * i);
                if (lr >= G::SR) halo_bits |= 2u << (2 * i);
            } else {
                ok = ok && lr >= 0 && lr < G::SR;
            }
            pix = seg * G::SR * HW + lr * HW + w;
        }
        x_off[i] = ok ? (pix * p.Cin + ci0 + c * 8) * 2 : kOutOfRange;
        x_dst[i] = instr < X_INSTR ? A_BYTES + instr * 1024 : DUMP;
    }

    auto issue = [&](int s, int stage) {
        char* base = smem + stage * STAGE;
        const int a_step = s * 64 * p.Cout * 2, x_step = s * 64 * p.Cin * 2;
        // first image row of the band (hw >= 16): the rows above / below exist inside the image?
        const int h0 = G::HALO ? (s * G::SR) & (HW - 1) : 0;
        const bool top_ok = h0 > 0, bot_ok = h0 + G::SR < HW;
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
            // (an out-of-range slot stays out of range: the step offset is < 2^31)
            char* dst = a_dst[i] == DUMP ? smem + DUMP : base + a_dst[i];
            const int voff = a_off[i] + a_step;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(dyrsrc, (lds_ptr)dst, 16, voff, 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < X_IT; ++i) {
            int voff = x_off[i] + x_step;
            if (G::HALO) {
                const unsigned cls = (halo_bits >> (2 * i)) & 3u;
                if ((cls == 1u && !top_ok) || (cls == 2u && !bot_ok)) voff = kOutOfRange;
            }
            char* dst = x_dst[i] == DUMP ? smem + DUMP : base + x_dst[i];
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xrsrc, (lds_ptr)dst, 16, voff, 0, 0, 0);
        }
    };

    // ---- fragment addressing.  Transposing read: 16-lane group g covers channels 16*(g&1)..
    // of a 32-wide tile and K rows 8*(g>>1)..; inside the group lane 4q+pp names row q (and,
    // second read, q + 4) and channels 4pp..4pp+3.
    const int g = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3;
    const int colb = (16 * (g & 1) + 4 * pp) * 2;
    int a_base, b_base;
    if (G::PM) {
        // K = the 16 images of the step at one position: image 8*(g>>1) + q (+4), row = 4*image
        a_base = (8 * (g >> 1) + q) * 4 * AS + wco * 64 + colb;
        b_base = A_BYTES + (8 * (g >> 1) + q) * 4 * G::XS + wci * 64 + colb;
    } else {
        a_base = (8 * (g >> 1) + q) * AS + wco * 64 + colb;
        // lane part of the padded row of pixel (K sub-step 0, lower half), shifted by one
        // grid row and column up-left so that every tap offset is non-negative
        int row;
        if (HW == 4) row = (2 * (g >> 1) + 1) * 6 + q + 1;            // h = 2*(g>>1) (+1), w = q
        else if (HW == 8) row = ((g >> 1) + 1) * 10 + q + 1;          // h = g>>1, w = q (+4)
        else row = G::PW + 8 * (g >> 1) + q + 1;                      // w = 8*(g>>1) + q (+4)
        b_base = A_BYTES + (row - G::PW - 1) * G::XS + wci * 64 + colb;
    }

    f32x16 acc[9];
#pragma unroll
    for (int tp = 0; tp < 9; ++tp)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[tp][r] = 0.f;

    auto tr2 = [&](const char* ptr, int second) {
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(ptr));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(ptr + second));
        const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        return __builtin_bit_cast(bf16x8, v);
    };

    auto compute = [&](int stage) {
        const char* ab = smem + stage * STAGE + a_base;
        const char* bb = smem + stage * STAGE + b_base;
        if constexpr (G::PM) {
            bf16x8 af[4], bf[4];
#pragma unroll
            for (int pos = 0; pos < 4; ++pos) {
                af[pos] = tr2(ab + pos * AS, 16 * AS);
                bf[pos] = tr2(bb + pos * G::XS, 16 * G::XS);
            }
#pragma unroll
            for (int tp = 0; tp < 9; ++tp) {
                const int dh = tp / 3 - 1, dw_ = tp % 3 - 1;
#pragma unroll
                for (int pos = 0; pos < 4; ++pos) {
                    const int hh = (pos >> 1) + dh, ww = (pos & 1) + dw_;
                    if (hh < 0 || hh > 1 || ww < 0 || ww > 1) continue;
                    acc[tp] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[pos], bf[hh * 2 + ww],
                                                                      acc[tp], 0, 0, 0);
                }
            }
        } else {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const bf16x8 af = tr2(ab + ks * 16 * AS, 4 * AS);
#pragma unroll
                for (int tp = 0; tp < 9; ++tp) {
                    const int shift = (tp / 3) * G::PW + (tp % 3);      // (dh + 1, dw + 1)
                    const bf16x8 bf = tr2(bb + (G::ks_rows(ks) + shift) * G::XS,
                                          G::hi_rows * G::XS);
                    acc[tp] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bf, acc[tp], 0, 0, 0);
                }
            }
        }
    };

    // ---- NSTAGE ring, prefetch distance NSTAGE - 1
    constexpr int D = NSTAGE - 1;
#pragma unroll
    for (int d = 0; d < D; ++d)
        if (s_begin + d < s_end) issue(s_begin + d, d);
    int stage = 0;
    for (int s = s_begin; s < s_end; ++s) {
        // loads issued after those of step s: the steps s+1 .. s+D-1 that exist
        const int later = min(D - 1, s_end - 1 - s);
        if (later >= 2) wait_vmcnt<2 * LOADS>();
        else if (later == 1) wait_vmcnt<LOADS>();
        else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (s + D < s_end) {
            int st = stage + D;
            if (st >= NSTAGE) st -= NSTAGE;
            issue(s + D, st);
        }
        compute(stage);
        stage = stage + 1 == NSTAGE ? 0 : stage + 1;
    }

    // D[co][ci]: lane -> ci = l & 31, register r -> co = (r&3) + 8*(r>>2) + 4*(l>>5)
    const int l31 = lane & 31, lh = lane >> 5;
    float* out = p.dw + split * p.split_stride;
    const int ci = ci0 + wci * 32 + l31;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int co = co0 + wco * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        float* dst = out + (size_t)co * 9 * p.Cin + ci;
#pragma unroll
        for (int tp = 0; tp < 9; ++tp) dst[tp * p.Cin] = acc[tp][r];
    }
}

#define DML_WG9_INSTANTIATE(HW)                                                 \
    template __global__ void conv3x3_wgrad9_kernel<HW, 64, 2>(Args);            \
    template __global__ void conv3x3_wgrad9_kernel<HW, 64, 3>(Args);            \
    template __global__ void conv3x3_wgrad9_kernel<HW, 128, 2>(Args);           \
    template __global__ void conv3x3_wgrad9_kernel<HW, 128, 3>(Args);
DML_WG9_INSTANTIATE(2)
DML_WG9_INSTANTIATE(4)
DML_WG9_INSTANTIATE(8)
DML_WG9_INSTANTIATE(16)
DML_WG9_INSTANTIATE(32)
#undef DML_WG9_INSTANTIATE

template <int HW, int BCO, int NSTAGE>
constexpr int lds_bytes() {
    using G = Geo<HW>;
    constexpr int AS = G::PM ? BCO * 2 + 80 : BCO * 2 + 64;
    constexpr int a_instr = (64 * (AS / 16) + 63) / 64;
    constexpr int x_instr = (G::XR * G::XCPR + 63) / 64;
    return NSTAGE * (a_instr + x_instr) * 1024 + 1024;
}

template <int HW, int BCO, int NSTAGE>
void launch(Args a, int splits, hipStream_t stream) {
    constexpr int lds = lds_bytes<HW, BCO, NSTAGE>();
    static_assert(lds <= 160 * 1024, "tile does not fit the 160 KiB LDS");
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)conv3x3_wgrad9_kernel<HW, BCO, NSTAGE>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        attr_set = true;
    }
    a.tiles_co = a.Cout / BCO;
    a.tiles_ci = a.Cin / 64;
    hipLaunchKernelGGL((conv3x3_wgrad9_kernel<HW, BCO, NSTAGE>),
                       dim3(a.tiles_co * a.tiles_ci, splits), dim3(BCO * 4), lds, stream, a);
}

template <int BCO, int NSTAGE>
void launch_hw(int hw, const Args& a, int splits, hipStream_t stream) {
    switch (hw) {
        case 2: launch<2, BCO, NSTAGE>(a, splits, stream); break;
        case 4: launch<4, BCO, NSTAGE>(a, splits, stream); break;
        case 8: launch<8, BCO, NSTAGE>(a, splits, stream); break;
        case 16: launch<16, BCO, NSTAGE>(a, splits, stream); break;
        case 32: launch<32, BCO, NSTAGE>(a, splits, stream); break;
        default: throw std::invalid_argument("conv3x3_wgrad9: hw must be 2, 4, 8, 16 or 32");
    }
}

}  // namespace wg9

bool conv3x3_wgrad9_supported(ConvShape s) {
    return s.batch > 0 && (s.hw == 2 || s.hw == 4 || s.hw == 8 || s.hw == 16 || s.hw == 32) &&
           s.cin >= 64 && s.cin % 64 == 0 && s.cout >= 64 && s.cout % 64 == 0;
}

int conv3x3_wgrad9_bco(ConvShape s, int force_bco) {
    if (force_bco == 64 || force_bco == 128) return s.cout % force_bco == 0 ? force_bco : 64;
    return s.cout % 128 == 0 ? 128 : 64;
}

// Pixel-range splits: every block costs a slab of BCO x 9 x 64 floats (written, then read by the
// sum), so the block count aimed for is well below what the per-tap kernel used.  The bound is
// what the plan can reach for ANY batch up to s.batch (the rounding to whole steps per split
// below can only lower it, by different amounts for different batch sizes).
int conv3x3_wgrad9_splits_bound(ConvShape s, int bco, int block_target) {
    const int total_steps = (s.batch * s.hw * s.hw + 63) / 64;
    const int tiles = (s.cout / bco) * (s.cin / 64);
    const int target = block_target > 0 ? block_target : 256;
    int splits = (target + tiles / 2) / tiles;
    const int max_splits = total_steps >= 4 ? total_steps / 4 : 1;
    if (splits > max_splits) splits = max_splits;
    return splits < 1 ? 1 : splits;
}

int conv3x3_wgrad9_splits(ConvShape s, int bco, int force_splits, int block_target) {
    const int total_steps = (s.batch * s.hw * s.hw + 63) / 64;
    int splits = conv3x3_wgrad9_splits_bound(s, bco, block_target);
    if (force_splits > 0) splits = force_splits < total_steps ? force_splits : total_steps;
    const int sps = (total_steps + splits - 1) / splits;
    return (total_steps + sps - 1) / sps;
}

void conv3x3_wgrad9(const bf16_raw* x, const bf16_raw* dy, float* dw, ConvShape s,
                    hipStream_t stream, float* slab, int bco, int splits, int stages) {
    if (!conv3x3_wgrad9_supported(s))
        throw std::invalid_argument("conv3x3_wgrad9: needs hw in {2,4,8,16,32}, cin % 64 == 0, "
                                    "cout % 64 == 0");
    if ((bco != 64 && bco != 128) || s.cout % bco)
        throw std::invalid_argument("conv3x3_wgrad9: bad co tile");
    const long long M = (long long)s.batch * s.hw * s.hw;
    if (M * (s.cin > s.cout ? s.cin : s.cout) >= (1LL << 30))
        throw std::invalid_argument("conv3x3_wgrad9: tensor exceeds 2^31 bytes");
    const int total_steps = (int)((M + 63) / 64);
    if (splits < 1 || splits > total_steps) throw std::invalid_argument("conv3x3_wgrad9: bad splits");
    if (splits > 1 && !slab) throw std::invalid_argument("conv3x3_wgrad9: splits need a slab");
    wg9::Args a;
    a.x = x; a.dy = dy;
    a.dw = splits > 1 ? slab : dw;
    a.M = (int)M; a.Cin = s.cin; a.Cout = s.cout;
    a.tiles_ci = a.tiles_co = 0;
    a.steps_per_split = (total_steps + splits - 1) / splits;
    if ((total_steps + a.steps_per_split - 1) / a.steps_per_split != splits)
        throw std::invalid_argument("conv3x3_wgrad9: splits must come from conv3x3_wgrad9_splits");
    a.split_stride = splits > 1 ? (size_t)s.cout * 9 * s.cin : 0;
    a.x_bytes = (unsigned)((size_t)M * s.cin * 2);
    a.dy_bytes = (unsigned)((size_t)M * s.cout * 2);
    if (bco == 128) {
        if (stages == 2) wg9::launch_hw<128, 2>(s.hw, a, splits, stream);
        else wg9::launch_hw<128, 3>(s.hw, a, splits, stream);
    } else {
        if (stages == 2) wg9::launch_hw<64, 2>(s.hw, a, splits, stream);
        else wg9::launch_hw<64, 3>(s.hw, a, splits, stream);
    }
}

}  // namespace dml
