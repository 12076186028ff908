// conv3x3 (stride 1, pad 1) forward and backward-data: patch-resident MFMA implicit GEMM with
// tall pixel tiles.
//
// Reference op: nn.Conv2d(cin, cout, kernel_size=3, stride=1, padding=1, bias=True)
// (/root/reference/part1/model.py:18-23) and its autograd input gradient
// (loss.backward(), /root/reference/part1/main.py:44).
//
// Why this form.  A CU takes in about 29 bytes per clock from L2 into LDS (LDS-DMA, measured), and
// its matrix cores finish a 64-deep K step of a 128 x 128 tile in 512 clocks.  The streaming kernel
// of conv_igemm.hip fetches (128 + 128) rows x 128 bytes for that step - 1130 clocks of feed - so it
// runs at the feed rate, not at the matrix rate.  Two things cut the bytes per flop:
//   * the pixels a tile touches are fetched ONCE per 64-channel chunk (the tile's BM pixels plus
//     hw + 1 pixels of halo on either side) and the nine taps read that one LDS copy at nine row
//     shifts (in NHWC the (dh, dw) neighbour of pixel m is pixel m + dh*hw + dw);
//   * the weights, which then dominate, are amortised over a TALL tile: BM = 256 pixels x BN = 64
//     channels has the area of 128 x 128 but fetches (BM + 2*hw + 2 + 9*BN) rows per chunk instead
//     of 9 * 256, i.e. 12 KB instead of 32 KB per K step - under the matrix time.
//
// LDS image.  Rows are 128 bytes (64 channels); the 16-byte group c of row r sits in slot
//   slot(r, c) = 4 * (c & 1) + ((c >> 1) ^ ((r >> 1) & 3)).
// A ds_read_b128 is served in groups of 16 lanes that hold rows a .. a+3, a+12 .. a+15 of one K
// group and rows a+4 .. a+11 of the next one (c differs in bit 0): the two sets land in different
// 64-byte halves, and each set covers every row residue mod 8 once, so (row & 1, slot) is a
// bijection onto the 16 slots of the 256-byte bank row for EVERY start row a - the nine shifted
// reads are conflict-free, not only the aligned one.  The global -> LDS copies are lane-linear
// (LDS-DMA), so the permutation is applied on the source side: lane (row, slot) fetches the group
// that belongs in its slot.
//
// Padding without touching the data: every patch buffer starts with two zero rows, and a lane
// whose (pixel, tap) falls outside the image reads the zero row of its row's parity at its own
// slot (same banks as the real read).  The 9 per-tap addresses of a lane's TM pixels are computed
// once; a K step costs no address arithmetic beyond one XOR per fragment for the second K half.
//
// Pipeline.  Weights stream through a 3-stage ring, one stage = one filter row (3 taps x BN rows),
// two stages in flight; the patch of chunk c+1 is requested during the first stage of chunk c into
// the other patch buffer.  One raw s_barrier per stage (1536 matrix clocks at 256 x 64), counted
// vmcnt waits - every count is a compile-time constant because a chunk (9 taps = 3 stages) is
// unrolled; fragments are double-buffered in registers and the first fragments of a stage are
// requested right after its barrier, under the MFMAs of the previous stage's last half step.
#include <stdexcept>
#include <string>
#include <type_traits>
#include <utility>

#include "dml_conv_tile.h"

namespace dml {
namespace igemm {

namespace {

__device__ __forceinline__ int slot_of(int r, int c) {
    return ((c & 1) << 2) | ((c >> 1) ^ ((r >> 1) & 3));
}
__device__ __forceinline__ int group_in_slot(int r, int s) {
    return (((s & 3) ^ ((r >> 1) & 3)) << 1) | (s >> 2);
}

template <int V> using ic = std::integral_constant<int, V>;

constexpr int kTapsPerStage = 3, kStages = 3;

template <int BM, int NW>
struct PatchGeom {
    static constexpr int kMaxRows = BM + 2 * 17;                  // hw <= 16
    static constexpr int kGroups = (kMaxRows + 7) / 8;
    static constexpr int kIters = (kGroups + NW - 1) / NW;        // wave-loads per chunk and wave
    static constexpr int kBytes = 256 + kIters * NW * 1024;       // two zero rows + the rows
};

}  // namespace

template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F& f, std::integer_sequence<int, I...>) {
    (f(ic<I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    static_for_impl(f, std::make_integer_sequence<int, N>{});
}

// WAVES_M waves divide the tile's pixel rows; WAVES_K = 2 puts TWO waves on every 64-row slice, one
// per 32-channel half of each K step, so that a wave's tile is (BM / WAVES_M) x 64 at half the K:
// the fragments it reads per MFMA halve against two waves that split the rows instead (LDS reads
// per K step of a 256 x 64 tile: 8 waves x 96 rows x 128 B -> 8 waves x 128 rows x 64 B, -33 %;
// with no copies and no MFMAs in the loop the kernel still took 70 % of its time, i.e. the fragment
// reads set the pace).  The two partial tiles meet once, in the epilogue, through LDS.
// OPT: ablation builds for timing only - 4: no copies inside the loop, 8: no MFMAs.
template <int BM, int BN, int WAVES_M, int WAVES_K, int EPI, int OPT>
__global__ __launch_bounds__(WAVES_M * WAVES_K * 64) void conv3x3_pr_kernel(IgemmArgs p) {
    raise_stream_signal(p);
    static_assert(WAVES_K == 1 || WAVES_K == 2, "one or two waves per row slice");
    constexpr int NW = WAVES_M * WAVES_K;
    constexpr int WTM = BM / WAVES_M, WTN = BN;
    constexpr int TM = WTM / 16, TN = WTN / 16;
    constexpr int TMH = TM / WAVES_K;                   // 16-row groups a wave finishes
    constexpr int KSUB = 2 / WAVES_K;                   // 32-deep sub-steps a wave runs per tap
    constexpr int TPS = kTapsPerStage, WS = kStages;
    constexpr int SUBS = TPS * KSUB;                    // sub-steps per stage
    constexpr int WSTAGE = TPS * BN * 128;
    // (Dividing the copies between the two waves of a SIMD - waves 0 .. 3 the weights, waves 4 .. 7
    // the patches, so that one of the two always has MFMAs to issue while the other pays the
    // 60-180 cycles an LDS-DMA instruction costs its wave - measured 3-10 % slower on every layer
    // than every wave issuing its share of all copies.)
    constexpr int NLOAD = NW;
    constexpr int W_IT = TPS * BN / 8 / NLOAD;          // weight wave-loads per stage and wave
    static_assert((TPS * BN / 8) % NLOAD == 0, "stage must divide over the waves");
    static_assert(TM >= 1 && TN >= 1 && TM % WAVES_K == 0, "tile too small for this many waves");
    using PG = PatchGeom<BM, NLOAD>;
    constexpr int P_IT = PG::kIters, PBUF = PG::kBytes;
    constexpr int RING = WS * WSTAGE;

    extern __shared__ __attribute__((aligned(256))) char smem[];      // [ring][patch 0][patch 1]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave % WAVES_M, wk = wave / WAVES_M;
    int bid = blockIdx.x;       // XCD-contiguous tile ranges (blocks are dealt round-robin to 8 XCDs)
    if ((gridDim.x & 7) == 0) bid = (bid & 7) * (gridDim.x >> 3) + (bid >> 3);
    const int tile_n = bid % p.tiles_n, tile_m = bid / p.tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int C = p.C, hw = p.hw;
    // (gridDim.y > 1: the 64-channel chunks of the reduction are divided between gridDim.y blocks
    // per tile, each leaving an fp32 slab for the consumer to sum; plain epilogue only)
    const int chunks = (1 << p.log_chunks) / (int)gridDim.y;      // this block's share
    const int c0 = blockIdx.y * chunks;
    const int halo = hw + 1;

    // the zero rows of both patch buffers
    if (tid < 32) {
        *(uint4*)(smem + RING + (tid >> 4) * PBUF + (tid & 15) * 16) = make_uint4(0u, 0u, 0u, 0u);
    }

    // ---- global -> LDS: per lane a fixed byte offset, everything that changes with the tap row
    // and the channel chunk is wave-uniform and travels in the scalar offset
    const int lrow = lane >> 3, slot = lane & 7;
    const int lw = wave;
    int w_off[W_IT];
#pragma unroll
    for (int i = 0; i < W_IT; ++i) {
        const int q = i * NLOAD + lw;                    // 1 KiB piece of the stage
        const int tap_in_stage = q / (BN / 8), r = (q % (BN / 8)) * 8 + lrow;
        w_off[i] = ((n0 + permuted_channel<WTN>(r)) * 9 * C + tap_in_stage * C +
                    group_in_slot(r, slot) * 8) * 2;
    }
    int p_off[P_IT];
#pragma unroll
    for (int i = 0; i < P_IT; ++i) {
        const int prow = (lw + NLOAD * i) * 8 + lrow;
        const int pix = m0 - halo + prow;
        const bool ok = prow < BM + 2 * halo && pix >= 0 && pix < p.M;
        p_off[i] = ok ? (pix * C + group_in_slot(prow, slot) * 8) * 2 : kOutOfRange;
    }
    const __amdgpu_buffer_rsrc_t xrsrc =
        __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wrsrc =
        __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, p.w_bytes, 0x00020000);

    auto issue_patch = [&](int cc) {
        char* dst = smem + RING + (cc & 1) * PBUF + 256 + lw * 1024;
        const int soff = (c0 + cc) * 128;
#pragma unroll
        for (int i = 0; i < P_IT; ++i) {
            const int voff = p_off[i];
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xrsrc, (lds_ptr)(dst + i * NLOAD * 1024), 16,
                                                     voff, soff, 0, 0);
        }
    };
    auto issue_w = [&](int cc, int row /* filter row = stage */) {
        char* dst = smem + row * WSTAGE + lw * 1024;
        const int soff = (3 * row * C + (c0 + cc) * 64) * 2;
#pragma unroll
        for (int i = 0; i < W_IT; ++i) {
            const int voff = w_off[i];
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc, (lds_ptr)(dst + i * NLOAD * 1024), 16,
                                                     voff, soff, 0, 0);
        }
    };

    // ---- fragment addresses (bytes from smem).  Lane (l15, lq) reads row l15 of a 16-row group and
    // the 16-byte channel group 4 * kk + lq of K half kk: with one wave per slice the address of
    // half 0 is kept and half 1 is that ^ 32, with two the wave's own half is folded in here.
    const int l15 = lane & 15, lq = lane >> 4;
    const int kgroup = (WAVES_K == 2 ? 4 * wk : 0) + lq;
    int wa[TN];
#pragma unroll
    for (int i = 0; i < TN; ++i) {
        const int r = i * 16 + l15;
        wa[i] = r * 128 + slot_of(r, kgroup) * 16;
    }
    int pa[TM][9];
#pragma unroll
    for (int j = 0; j < TM; ++j) {
        const int rt = wm * WTM + j * 16 + l15;
        const int m = m0 + rt;
        const int w_ = m & (hw - 1), h_ = (m >> p.log_hw) & (hw - 1);
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int dh = t / 3 - 1, dw = t % 3 - 1;
            const bool valid = m < p.M && (unsigned)(h_ + dh) < (unsigned)hw &&
                               (unsigned)(w_ + dw) < (unsigned)hw;
            const int pr = rt + halo + dh * hw + dw;
            const int in_row = slot_of(pr, kgroup) * 16;
            pa[j][t] = RING + (valid ? 256 + pr * 128 : (pr & 1) * 128) + in_row;
        }
    }

    f32x4 acc[TN][TM];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    struct Frags {
        bf16x8 w[TN];
        bf16x8 x[TM];
    };
    Frags fr[2];        // register double buffer, always indexed by compile-time constants

    // sub-step u of a chunk: tap u / KSUB (and K half u & 1 with one wave per slice); PAR = patch
    // buffer of the chunk.  The pixel fragments first: the MFMAs walk i outer / j inner, so the
    // first of them needs every pixel fragment but only the first weight fragment.
    auto load_frags = [&](auto par_c, auto u_c, Frags& f) {
        constexpr int PAR = decltype(par_c)::value, U = decltype(u_c)::value;
        constexpr int T = U / KSUB, KX = KSUB == 2 ? (U & 1) * 32 : 0;
        constexpr int woff = (T / TPS) * WSTAGE + (T % TPS) * BN * 128;
#pragma unroll
        for (int j = 0; j < TM; ++j) f.x[j] = *(const bf16x8*)(smem + ((pa[j][T] ^ KX) + PAR * PBUF));
#pragma unroll
        for (int i = 0; i < TN; ++i) f.w[i] = *(const bf16x8*)(smem + ((wa[i] ^ KX) + woff));
    };
    auto mma = [&](const Frags& f) {
        if constexpr (OPT & 8) {
            // (ablation build: fragments are read and kept alive, nothing is multiplied)
#pragma unroll
            for (int i = 0; i < TN; ++i) asm volatile("" ::"v"(f.w[i]));
#pragma unroll
            for (int j = 0; j < TM; ++j) asm volatile("" ::"v"(f.x[j]));
            return;
        }
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
            for (int j = 0; j < TM; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.w[i], f.x[j], acc[i][j], 0, 0, 0);
    };

    // Issue order of a sub-step, pinned for the compiler's scheduler (left alone it sinks every
    // fragment read down to its first use and the register double buffer is gone): the reads of
    // the NEXT sub-step alternate with the MFMAs of this one.
    auto interleave = [&]() {
        if constexpr (OPT & 8) return;
        constexpr int per = (TN * TM) / (TN + TM);
#pragma unroll
        for (int k = 0; k < TN + TM; ++k) {
            __builtin_amdgcn_sched_group_barrier(0x008, per, 0);      // MFMA
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);        // DS read
        }
        __builtin_amdgcn_sched_group_barrier(0x008, TN * TM - (TN + TM) * per, 0);
    };

    // (un-pooled BatchNorm backward sums in the epilogue want the layer's conv output values of
    // the rows this wave finishes: requested during the last chunk, see bn_plain_bwd_prefetch)
    constexpr bool kBnBwd = EPI == EPI_NONE;
    const int row0 = m0 + wm * WTM + wk * TMH * 16;         // first row this wave finishes
    const int nl = n0 + lq * 4 * TN;                        // this lane's first output channel
    uint32_t bn_xv[kBnBwd ? TMH : 1][2 * TN];
    const bool bn_plain = kBnBwd && p.bn_pre != nullptr && !p.bn_pooled;
    auto bn_prefetch = [&]() {
        if constexpr (kBnBwd) bn_plain_bwd_prefetch<TN, TMH>(p, row0 + l15, nl, bn_xv);
    };

    // One chunk = 3 stages of SUBS sub-steps.  On entry the fragments of sub-step 0 are in their
    // buffer and the barrier of the chunk's first stage has been passed.  Loads issued after the
    // barrier of stage (cc, i): the weights two stages ahead (filter row (i + 2) % 3 of this or the
    // next chunk) and, in the first stage, the next chunk's patch; the wait in front of the NEXT
    // barrier lets exactly those stay in flight.  (The buffer of sub-step u: with 18 sub-steps per
    // chunk u & 1; with 9 the parity carries over from chunk to chunk, and chunk parity = PAR.)
    auto chunk = [&](auto par_c, auto last_c, int cc) {
        constexpr int PAR = decltype(par_c)::value;
        constexpr bool LAST = decltype(last_c)::value != 0;
        constexpr int B0 = KSUB == 1 ? PAR : 0;
        auto after_barrier = [&](auto i_c) {          // loads issued behind the barrier of stage i
            constexpr int I = decltype(i_c)::value;
            if constexpr (OPT & 4) {
                // (ablation build: no copies in the loop - the operands are whatever LDS holds)
            } else if constexpr (!LAST) {
                if constexpr (I == 0) { issue_w(cc, 2); issue_patch(cc + 1); }
                else issue_w(cc + 1, I - 1);
            } else {
                if constexpr (I == 0) issue_w(cc, 2);
                if constexpr (I == 1)
                    if (bn_plain) bn_prefetch();
            }
        };
        auto wait_for_next = [&](auto i_c) {          // before the barrier of the stage after (cc, i)
            constexpr int I = decltype(i_c)::value;
            if constexpr (OPT & 4) {
                wait_vmcnt<0>();
            } else if constexpr (!LAST) {
                if constexpr (I < 2) wait_vmcnt<W_IT + P_IT>();
                else wait_vmcnt<W_IT>();
            } else {
                if constexpr (I == 0) wait_vmcnt<W_IT>();
                else wait_vmcnt<0>();
            }
        };
        auto stage = [&](auto i_c) {
            constexpr int I = decltype(i_c)::value;
            constexpr int U0 = SUBS * I;
            if constexpr (I == 0) after_barrier(ic<0>{});   // (stages 1, 2: issued at their barrier below)
            static_for<SUBS - 1>([&](auto v_c) {
                constexpr int U = U0 + decltype(v_c)::value;
                load_frags(ic<PAR>{}, ic<U + 1>{}, fr[(B0 + U + 1) & 1]);
                mma(fr[(B0 + U) & 1]);
                interleave();
            });
            constexpr int UL = U0 + SUBS - 1;               // the stage's last sub-step
            if constexpr (LAST && I == 2) {
                mma(fr[(B0 + UL) & 1]);
            } else {
                // lgkmcnt(0), vmcnt / expcnt untouched: every LDS read of this wave has returned
                // before another wave's copies may overwrite the stage / patch it read
                __builtin_amdgcn_s_waitcnt(0xC07F);
                wait_for_next(ic<I>{});
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                if constexpr (I < 2) {
                    after_barrier(ic<I + 1>{});
                    load_frags(ic<PAR>{}, ic<UL + 1>{}, fr[(B0 + UL + 1) & 1]);
                } else {
                    // (the next chunk issues its own first-stage loads on entry)
                    load_frags(ic<PAR ^ 1>{}, ic<0>{}, fr[(B0 + UL + 1) & 1]);
                }
                mma(fr[(B0 + UL) & 1]);
                interleave();
            }
        };
        stage(ic<0>{});
        stage(ic<1>{});
        stage(ic<2>{});
    };

    issue_patch(0);
    issue_w(0, 0);
    issue_w(0, 1);
    __builtin_amdgcn_s_waitcnt(0xC07F);        // the zero rows
    wait_vmcnt<W_IT>();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    load_frags(ic<0>{}, ic<0>{}, fr[0]);
    if (chunks == 1) {
        chunk(ic<0>{}, ic<1>{}, 0);
    } else {
        for (int cc = 0; cc + 2 < chunks; cc += 2) {
            chunk(ic<0>{}, ic<0>{}, cc);
            chunk(ic<1>{}, ic<0>{}, cc + 1);
        }
        chunk(ic<0>{}, ic<0>{}, chunks - 2);
        chunk(ic<1>{}, ic<1>{}, chunks - 1);
    }

    // ---- everything after the loop works on the TMH row groups this wave finishes
    auto finish = [&](const f32x4 (&fin)[TN][TMH]) {
        // BatchNorm sums for free, one partial row [2][N] per pixel tile (see conv_igemm.hip)
        const bool bn_bwd = EPI == EPI_NONE && p.bn_pre != nullptr && gridDim.y == 1;
        if (p.stats && (EPI == EPI_BIAS || bn_bwd)) {
            float s1[TN][4], s2[TN][4];
            if (bn_bwd) {
                if constexpr (EPI == EPI_NONE) {
                    if (p.bn_pooled) bn_pool_bwd_tile_sums<TN, TMH>(fin, p, row0 + l15, nl, s1, s2);
                    else bn_plain_bwd_tile_sums<TN, TMH>(fin, p, bn_xv, row0 + l15, nl, s1, s2);
                }
            } else {
#pragma unroll
                for (int i = 0; i < TN; ++i)
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        float a = 0.f, b = 0.f;
#pragma unroll
                        for (int j = 0; j < TMH; ++j) {
                            const float v = fin[i][j][c];
                            a += v;
                            b += v * v;
                        }
                        s1[i][c] = row16_sum(a);
                        s2[i][c] = row16_sum(b);
                    }
            }
            __syncthreads();                       // every wave is done with what LDS held before
            float* red = (float*)smem;             // [NW][2][BN]
            if (l15 == 0) {
#pragma unroll
                for (int i = 0; i < TN; ++i)
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const int col = lq * 4 * TN + 4 * i + c;
                        red[(wave * 2 + 0) * BN + col] = s1[i][c];
                        red[(wave * 2 + 1) * BN + col] = s2[i][c];
                    }
            }
            __syncthreads();
            for (int idx = tid; idx < 2 * BN; idx += NW * 64) {
                const int which = idx / BN, col = idx - which * BN;
                float t = 0.f;
                // (row slices in order, the two halves of a slice next to each other)
#pragma unroll
                for (int q = 0; q < WAVES_M; ++q)
#pragma unroll
                    for (int h = 0; h < WAVES_K; ++h)
                        t += red[((h * WAVES_M + q) * 2 + which) * BN + col];
                p.stats[((size_t)tile_m * 2 + which) * p.N + n0 + col] = t;
            }
        }
        // epilogue: lane holds pixel m and its 4*TN consecutive channels
#pragma unroll
        for (int j = 0; j < TMH; ++j) {
            const int m = row0 + j * 16 + l15;
            if (m >= p.M) continue;
            if constexpr (EPI == EPI_NONE) {
                if (gridDim.y > 1) {
                    float* dst = p.slab + ((size_t)blockIdx.y * p.M + m) * p.N + nl;
#pragma unroll
                    for (int i = 0; i < TN; ++i)
                        *(float4*)(dst + 4 * i) =
                            float4{fin[i][j][0], fin[i][j][1], fin[i][j][2], fin[i][j][3]};
                    continue;
                }
            }
            f32x4 a[TN];
#pragma unroll
            for (int i = 0; i < TN; ++i) a[i] = fin[i][j];
            store_lane_channels<TN, EPI>(a, p.y, p.bias, p.mask, (size_t)m * p.N + nl, nl);
        }
    };

    if constexpr (WAVES_K == 1) {
        finish(acc);
    } else {
        // The two waves of a row slice hold partial sums over their K halves.  Each keeps the row
        // groups it finishes and hands the others to its partner through LDS (16-byte values,
        // lane-linear: [receiving wave][i][j][lane]); own + partner's, always in that order.
        __syncthreads();                           // the ring and the patches are no longer read
        f32x4* xch = (f32x4*)smem;
        const int partner = wm + WAVES_M * (wk ^ 1);
        f32x4 fin[TN][TMH];
        auto exchange = [&](auto k_c) {            // k_c: this wave's K half, as a constant
            constexpr int K = decltype(k_c)::value;
#pragma unroll
            for (int i = 0; i < TN; ++i)
#pragma unroll
                for (int j = 0; j < TMH; ++j)
                    xch[((partner * TN + i) * TMH + j) * 64 + lane] = acc[i][(K ^ 1) * TMH + j];
            __syncthreads();
#pragma unroll
            for (int i = 0; i < TN; ++i)
#pragma unroll
                for (int j = 0; j < TMH; ++j)
                    fin[i][j] = acc[i][K * TMH + j] + xch[((wave * TN + i) * TMH + j) * 64 + lane];
        };
        if (wk == 0) exchange(ic<0>{});
        else exchange(ic<1>{});
        finish(fin);
    }
}

namespace {

template <int BM, int BN, int WAVES_M, int WAVES_K, int EPI, int OPT>
void launch_pr_one(const IgemmArgs& a, int splits, hipStream_t stream) {
    constexpr int NW = WAVES_M * WAVES_K;
    constexpr int lds = kStages * kTapsPerStage * BN * 128 + 2 * PatchGeom<BM, NW>::kBytes;
    static_assert(lds <= 160 * 1024, "tile does not fit the CU's LDS");
    static_assert(WAVES_K == 1 || NW * (BN / 16) * (BM / WAVES_M / 16 / WAVES_K) * 1024 <= lds,
                  "the partial tiles' exchange does not fit");
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)conv3x3_pr_kernel<BM, BN, WAVES_M, WAVES_K, EPI, OPT>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        attr_set = true;
    }
    const dim3 grid(((a.M + BM - 1) / BM) * a.tiles_n, splits);
    hipLaunchKernelGGL((conv3x3_pr_kernel<BM, BN, WAVES_M, WAVES_K, EPI, OPT>), grid, dim3(NW * 64), lds,
                       stream, a);
}

template <int BM, int BN, int WAVES_M, int WAVES_K, int OPT>
void launch_pr_epi(const IgemmArgs& a, int epi, int splits, hipStream_t stream) {
    if (splits > 1) {
        launch_pr_one<BM, BN, WAVES_M, WAVES_K, EPI_NONE, OPT>(a, splits, stream);
        return;
    }
    switch (epi) {
        case EPI_NONE: launch_pr_one<BM, BN, WAVES_M, WAVES_K, EPI_NONE, OPT>(a, 1, stream); break;
        case EPI_BIAS: launch_pr_one<BM, BN, WAVES_M, WAVES_K, EPI_BIAS, OPT>(a, 1, stream); break;
        case EPI_BIAS_RELU: launch_pr_one<BM, BN, WAVES_M, WAVES_K, EPI_BIAS_RELU, OPT>(a, 1, stream); break;
        default: launch_pr_one<BM, BN, WAVES_M, WAVES_K, EPI_MASK, OPT>(a, 1, stream); break;
    }
}

}  // namespace

// variant 0 = the shipped form of each tile height: 256 x 64 by eight waves of 32 x 64, 128 x 64 by
// eight waves as four 32-row slices x two K halves (measured alone at batch 256, shipped form
// against the other: 256 -> 256 @ 8x8 16.9 against 17.6 us, 512 -> 512 @ 4x4 16.7 against 18.7 us);
// 1 = the other form; 2 / 3 / 4 = ablation builds of variant 0 (no copies in the loop / no MFMAs /
// neither) for timing only; 5 = 128 x 64 by four waves of 32 x 64
bool conv3x3_pr_supported(int bm, int bn, int hw, int variant) {
    if (hw > 16 || hw < 2 || bn != 64) return false;
    return (bm == 256 && variant >= 0 && variant <= 4) || (bm == 128 && variant >= 0 && variant <= 5);
}

void launch_pr(const IgemmArgs& a, int bm, int bn, int variant, int epi, int splits,
               hipStream_t stream) {
    if (!conv3x3_pr_supported(bm, bn, a.hw, variant) || splits < 1 || (splits & (splits - 1)) ||
        splits > (1 << a.log_chunks) || (splits > 1 && !a.slab))
        throw std::invalid_argument("conv3x3_igemm: unsupported patch-resident tile");
    if (bm == 256) {
        switch (variant) {
            case 0: launch_pr_epi<256, 64, 8, 1, 0>(a, epi, splits, stream); break;
            case 1: launch_pr_epi<256, 64, 4, 2, 0>(a, epi, splits, stream); break;
            case 2: launch_pr_epi<256, 64, 8, 1, 4>(a, epi, splits, stream); break;
            case 3: launch_pr_epi<256, 64, 8, 1, 8>(a, epi, splits, stream); break;
            default: launch_pr_epi<256, 64, 8, 1, 12>(a, epi, splits, stream); break;
        }
    } else {
        switch (variant) {
            case 0: launch_pr_epi<128, 64, 4, 2, 0>(a, epi, splits, stream); break;
            case 1: launch_pr_epi<128, 64, 8, 1, 0>(a, epi, splits, stream); break;
            case 2: launch_pr_epi<128, 64, 4, 2, 4>(a, epi, splits, stream); break;
            case 3: launch_pr_epi<128, 64, 4, 2, 8>(a, epi, splits, stream); break;
            case 4: launch_pr_epi<128, 64, 4, 2, 12>(a, epi, splits, stream); break;
            default: launch_pr_epi<128, 64, 4, 1, 0>(a, epi, splits, stream); break;
        }
    }
}

}  // namespace igemm
}  // namespace dml
