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
//
// Two row orders.  Pixel-major: a tile is BM consecutive pixels (b,h,w) and every tap is
// multiplied, padded rows contributing zeros.  Position-major (PM, small maps): a tile is BM
// images at ONE spatial position, so a tap that falls outside the map does so for the whole
// tile and its K steps are skipped - at 2x2 that is 5 of 9 taps, at 4x4 31 % and at 8x8 16 %
// of the multiply work, which is what this kernel's time consists of.
struct IgemmParams {
    const float* x;
    const float* w;
    float* y;
    const float* bias;
    const float* mask;
    int M, N, cin, hw, log_hw, batch;
    // position-major only: 2 = the taps of a position are divided between blockIdx.y = 0 / 1 and
    // the two partial tiles meet in y (zeroed by the launcher) through float atomics.  With
    // exactly two addends onto zero the result does not depend on which arrives first
    // (0 + a = a exactly, a + b = b + a), so the kernel stays bit-reproducible; bias and ReLU mask
    // distribute over the two parts, a ReLU would not (EPI_BIAS_RELU is never split).
    int tap_splits;
    // optional hand-off to another queue (see StreamSignal): the first block stores the value
    // as it starts
    unsigned* signal;
    unsigned signal_value;
};

// (waves_per_eu: without the cap hipcc aims for 7 waves per SIMD on the 64 x 64 variant and
// spills the staging registers to scratch to get there; the LDS tiles allow 5 blocks per CU)
template <int MT, int NT, bool PM, int EPI>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 4)))
void conv3x3_igemm_f32_kernel(IgemmParams p) {
    if (p.signal && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0)
        __hip_atomic_store(p.signal, p.signal_value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    constexpr int BM = 64 * MT, BN = 64 * NT;
    constexpr int LA = BM * 8 / 256, LB = BN * 8 / 256;     // float4 loads per thread and chunk
    __shared__ __attribute__((aligned(16))) float As[2][BM * kBK];
    __shared__ __attribute__((aligned(16))) float Bs[2][BN * kBK];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    const int ntiles = p.N / BN;
    const int mt = blockIdx.x / ntiles, n0 = (blockIdx.x % ntiles) * BN;
    const int hw = p.hw, cin = p.cin, npix = hw * hw;
    // PM: block = (position, image tile); pixel-major: block = pixel tile
    const int btiles = PM ? (p.batch + BM - 1) / BM : 1;
    const int pos = PM ? mt / btiles : 0;
    const int row0 = PM ? (mt % btiles) * BM : mt * BM;      // first image / first pixel
    auto row_m = [&](int row) { return PM ? (row0 + row) * npix + pos : row0 + row; };
    auto row_ok = [&](int row) { return PM ? row0 + row < p.batch : row0 + row < p.M; };

    // taps this block multiplies, 4 bits each
    unsigned long long taps = 0x876543210ull;
    int ntaps = 9;
    if (PM) {
        const int ph = pos >> p.log_hw, pw = pos & (hw - 1);
        taps = 0;
        ntaps = 0;
        for (int t = 0; t < 9; ++t) {
            const int dh = t / 3 - 1, dw = t % 3 - 1;
            if ((unsigned)(ph + dh) < (unsigned)hw && (unsigned)(pw + dw) < (unsigned)hw) {
                taps |= (unsigned long long)t << (4 * ntaps);
                ++ntaps;
            }
        }
        if (p.tap_splits == 2) {
            const int half = ntaps >> 1;
            if (blockIdx.y == 0) {
                ntaps = half;
            } else {
                taps >>= 4 * half;
                ntaps -= half;
            }
        }
    }
    const bool split = PM && p.tap_splits == 2;

    // the rows this thread stages are the same for every chunk
    int a_h[LA], a_w[LA];
    long a_base[LA];
    bool a_ok[LA];
#pragma unroll
    for (int i = 0; i < LA; ++i) {
        const int idx = tid + 256 * i, row = idx >> 3;
        const int m = row_m(row);
        a_ok[i] = row_ok(row);
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
    const int chunks = cin / kBK, steps = ntaps * chunks;

    f32x16 acc[MT][NT];
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b)
#pragma unroll
            for (int g = 0; g < 16; ++g) acc[a][b][g] = 0.f;

    // Staging registers of the chunk after the one being multiplied.  Loads are branch-free
    // (a row outside the map or the tensor reads a safe address and is zeroed) and issued
    // unconditionally - the last iteration re-loads the last chunk and nobody reads it.
    // Two sets of them, used alternately: the loads of chunk s + 2 are issued at the top of chunk
    // s and written to LDS at the bottom of chunk s + 1, so they have two chunks of MFMAs (~2000
    // cycles for a 64 x 64 tile) to land instead of one - under load a global load takes longer
    // than one chunk.
    f32x4 ra0[LA], rb0[LB], ra1[LA], rb1[LB];
    int ld_tap = 0, ld_chunk = 0;           // position of the next chunk to load
#define DML_F32_FETCH(ra, rb)                                                                  \
    {                                                                                          \
        const int ti = ld_tap < ntaps ? ld_tap : ntaps - 1;                                    \
        const int tap = (int)((taps >> (4 * ti)) & 15);                                        \
        const int dh = tap / 3 - 1, dw = tap % 3 - 1;                                          \
        const int c0 = ld_chunk * kBK;                                                         \
        const long shift = (long)(dh * hw + dw) * cin + c0;                                    \
        _Pragma("unroll") for (int i = 0; i < LA; ++i) {                                       \
            const bool ok = a_ok[i] && (PM || ((unsigned)(a_h[i] + dh) < (unsigned)hw &&       \
                                               (unsigned)(a_w[i] + dw) < (unsigned)hw));       \
            const f32x4 v = *(const f32x4*)(ok ? p.x + a_base[i] + shift : p.x);               \
            ra[i] = ok ? v : f32x4{0.f, 0.f, 0.f, 0.f};                                        \
        }                                                                                      \
        _Pragma("unroll") for (int i = 0; i < LB; ++i)                                         \
            rb[i] = *(const f32x4*)(p.w + b_base[i] + (long)tap * cin + c0);                   \
        if (++ld_chunk == chunks) { ld_chunk = 0; ++ld_tap; }                                  \
    }
#define DML_F32_STASH(buf, ra, rb)                                                             \
    {                                                                                          \
        _Pragma("unroll") for (int i = 0; i < LA; ++i) {                                       \
            const int idx = tid + 256 * i, row = idx >> 3, slot = (idx & 7) ^ (row & 7);       \
            *(f32x4*)(&As[buf][row * kBK + slot * 4]) = ra[i];                                 \
        }                                                                                      \
        _Pragma("unroll") for (int i = 0; i < LB; ++i) {                                       \
            const int idx = tid + 256 * i, row = idx >> 3, slot = (idx & 7) ^ (row & 7);       \
            *(f32x4*)(&Bs[buf][row * kBK + slot * 4]) = rb[i];                                 \
        }                                                                                      \
    }
#define DML_F32_FRAGS(buf, q, FA, FB)                                                          \
    {                                                                                          \
        _Pragma("unroll") for (int a = 0; a < MT; ++a) {                                       \
            const int row = wm * 32 * MT + a * 32 + r;                                         \
            FA[a] = *(const f32x4*)(&As[buf][row * kBK + (((2 * (q) + h) ^ (row & 7)) * 4)]);  \
        }                                                                                      \
        _Pragma("unroll") for (int b = 0; b < NT; ++b) {                                       \
            const int row = wn * 32 * NT + b * 32 + r;                                         \
            FB[b] = *(const f32x4*)(&Bs[buf][row * kBK + (((2 * (q) + h) ^ (row & 7)) * 4)]);  \
        }                                                                                      \
    }

    // Operand fragments one group ahead of the MFMAs that use them.  The scheduling barriers pin
    // that order: left alone, hipcc sinks the global loads to the end of the chunk and reads each
    // fragment right before its first use, exposing both latencies on every group.
#define DML_F32_CHUNK(buf)                                                                     \
    {                                                                                          \
        f32x4 fa[2][MT], fb[2][NT];                                                            \
        DML_F32_FRAGS(buf, 0, fa[0], fb[0]);                                                   \
        __builtin_amdgcn_sched_barrier(0);                                                     \
        _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                        \
            if (q < 3) DML_F32_FRAGS(buf, q + 1, fa[(q + 1) & 1], fb[(q + 1) & 1]);            \
            __builtin_amdgcn_sched_barrier(0);                                                 \
            _Pragma("unroll") for (int t = 0; t < 4; ++t)                                      \
                _Pragma("unroll") for (int a = 0; a < MT; ++a)                                 \
                    _Pragma("unroll") for (int b = 0; b < NT; ++b)                             \
                        acc[a][b] = mfma32(fa[q & 1][a][t], fb[q & 1][b][t], acc[a][b]);       \
            __builtin_amdgcn_sched_barrier(0);                                                 \
        }                                                                                      \
    }

    DML_F32_FETCH(ra0, rb0);                // chunk 0
    DML_F32_STASH(0, ra0, rb0);
    DML_F32_FETCH(ra1, rb1);                // chunk 1
    __syncthreads();
    // chunk `step` is in LDS buffer step & 1; chunk step + 1 is in flight in set (step + 1) & 1
    int step = 0;
    for (; step + 1 < steps; step += 2) {
        DML_F32_FETCH(ra0, rb0);            // chunk step + 2
        DML_F32_CHUNK(0);
        DML_F32_STASH(1, ra1, rb1);         // chunk step + 1
        __syncthreads();
        DML_F32_FETCH(ra1, rb1);            // chunk step + 3
        DML_F32_CHUNK(1);
        DML_F32_STASH(0, ra0, rb0);         // chunk step + 2
        __syncthreads();
    }
    if (step < steps) DML_F32_CHUNK(0);     // odd count: the last chunk sits in buffer 0
#undef DML_F32_CHUNK
#undef DML_F32_FETCH
#undef DML_F32_STASH
#undef DML_F32_FRAGS

    // epilogue: a lane owns column n = .. + r and 16 rows per tile
#pragma unroll
    for (int b = 0; b < NT; ++b) {
        const int n = n0 + wn * 32 * NT + b * 32 + r;
        const float bias =
            ((EPI == EPI_BIAS || EPI == EPI_BIAS_RELU) && !(split && blockIdx.y != 0)) ? p.bias[n] : 0.f;
#pragma unroll
        for (int a = 0; a < MT; ++a) {
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                const int row = wm * 32 * MT + a * 32 + acc_row(g, h);
                const bool ok = row_ok(row);
                const long at = (long)(ok ? row_m(row) : 0) * p.N + n;
                float v = acc[a][b][g] + bias;
                if (EPI == EPI_BIAS_RELU) v = fmaxf(v, 0.f);
                if (EPI == EPI_MASK) v = p.mask[at] > 0.f ? v : 0.f;
                if (ok) {
                    if (split) atomicAdd(p.y + at, v);
                    else p.y[at] = v;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------ backward-weight
// Per tap a TN GEMM dw[co][ci] = sum_p dy[p][co] * x[p (+) tap][ci]; both operands are
// pixel-major in memory and stay so in LDS ([32 pixels][channels]): a lane's A element
// dy[p = 2s+h][co = r] and B element x[p][ci = r] are plain 4-byte LDS reads, consecutive lanes
// on consecutive banks.  blockIdx.x = (co tile, ci tile, tap), blockIdx.y = split of the pixels.
// The pixels are walked in chunks of 32; with PM (small maps) a chunk is 32 images at one
// spatial position, and the chunks whose position this block's tap shifts off the map are
// skipped (same saving as in the forward kernel).
struct WgradParams {
    const float* x;
    const float* dy;
    float* out;          // dw, or the slab of this launch (split s at out + s * slab_stride)
    long slab_stride;
    int M, cin, cout, hw, log_hw, chunks_per_split, total_chunks, batch;
};

template <int MT, int NT, bool PM>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 4)))
void conv3x3_wgrad_f32_kernel(WgradParams p) {
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
    const int hw = p.hw, npix = hw * hw;
    const int c_begin = blockIdx.y * p.chunks_per_split;
    int c_end = c_begin + p.chunks_per_split;
    if (c_end > p.total_chunks) c_end = p.total_chunks;
    // PM: chunk c = 32 images (c / npix) at position c % npix
    auto chunk_live = [&](int c) {
        if (!PM) return true;
        const int pos = c % npix;
        return (unsigned)((pos >> p.log_hw) + dh) < (unsigned)hw &&
               (unsigned)((pos & (hw - 1)) + dw) < (unsigned)hw;
    };
    auto next_live = [&](int c) {
        while (c < c_end && !chunk_live(c)) ++c;
        return c;
    };

    f32x16 acc[MT][NT];
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b)
#pragma unroll
            for (int g = 0; g < 16; ++g) acc[a][b][g] = 0.f;

    float4 rd[LA], rx[LB];
#define DML_F32_WFETCH(chunk)                                                                  \
    {                                                                                          \
        const int cc = (chunk) < c_end ? (chunk) : c_begin;      /* unused past the end */     \
        _Pragma("unroll") for (int i = 0; i < LA; ++i) {                                       \
            const int idx = tid + 256 * i, px = idx / (BM / 4), c4 = idx % (BM / 4);           \
            const int m = PM ? ((cc / npix) * kBK + px) * npix + cc % npix : cc * kBK + px;    \
            const bool ok = PM ? (cc / npix) * kBK + px < p.batch : m < p.M;                   \
            const float4 v = *(const float4*)(ok ? p.dy + (long)m * p.cout + co0 + c4 * 4 : p.dy); \
            rd[i] = ok ? v : float4{0.f, 0.f, 0.f, 0.f};                                       \
        }                                                                                      \
        _Pragma("unroll") for (int i = 0; i < LB; ++i) {                                       \
            const int idx = tid + 256 * i, px = idx / (BN / 4), c4 = idx % (BN / 4);           \
            const int m = PM ? ((cc / npix) * kBK + px) * npix + cc % npix : cc * kBK + px;    \
            const int w_ = m & (hw - 1), h_ = (m >> p.log_hw) & (hw - 1);                      \
            const bool ok = (PM ? (cc / npix) * kBK + px < p.batch : m < p.M) &&               \
                            (unsigned)(h_ + dh) < (unsigned)hw && (unsigned)(w_ + dw) < (unsigned)hw; \
            const float4 v = *(const float4*)(                                                 \
                ok ? p.x + (long)(m + dh * hw + dw) * p.cin + ci0 + c4 * 4 : p.x);             \
            rx[i] = ok ? v : float4{0.f, 0.f, 0.f, 0.f};                                       \
        }                                                                                      \
    }
#define DML_F32_WSTASH(buf)                                                                    \
    {                                                                                          \
        _Pragma("unroll") for (int i = 0; i < LA; ++i)                                         \
            *(float4*)(&Ds[buf][(tid + 256 * i) * 4]) = rd[i];                                 \
        _Pragma("unroll") for (int i = 0; i < LB; ++i)                                         \
            *(float4*)(&Xs[buf][(tid + 256 * i) * 4]) = rx[i];                                 \
    }

    int chunk = next_live(c_begin);
    DML_F32_WFETCH(chunk);
    DML_F32_WSTASH(0);
    __syncthreads();
    int buf = 0;
    while (chunk < c_end) {
        const int nxt = next_live(chunk + 1);
        DML_F32_WFETCH(nxt);
        // fragments of four MFMA steps at a time, one group ahead (order pinned, see igemm)
        float fa[2][4][MT], fb[2][4][NT];
#define DML_F32_WFRAGS(grp, F)                                                                 \
        _Pragma("unroll") for (int u = 0; u < 4; ++u) {                                        \
            const int prow = 2 * (4 * (grp) + u) + h;                                          \
            _Pragma("unroll") for (int a = 0; a < MT; ++a)                                     \
                fa[F][u][a] = Ds[buf][prow * BM + wm * 32 * MT + a * 32 + r];                  \
            _Pragma("unroll") for (int b = 0; b < NT; ++b)                                     \
                fb[F][u][b] = Xs[buf][prow * BN + wn * 32 * NT + b * 32 + r];                  \
        }
        DML_F32_WFRAGS(0, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int grp = 0; grp < kBK / 8; ++grp) {
            if (grp + 1 < kBK / 8) DML_F32_WFRAGS(grp + 1, (grp + 1) & 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int a = 0; a < MT; ++a)
#pragma unroll
                    for (int b = 0; b < NT; ++b)
                        acc[a][b] = mfma32(fa[grp & 1][u][a], fb[grp & 1][u][b], acc[a][b]);
            __builtin_amdgcn_sched_barrier(0);
        }
#undef DML_F32_WFRAGS
        DML_F32_WSTASH(buf ^ 1);
        __syncthreads();
        buf ^= 1;
        chunk = nxt;
    }
#undef DML_F32_WFETCH
#undef DML_F32_WSTASH
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

// out[i] = sum over the blocks' partials, i = (co, k) of 64 x 32: a block of 256 threads owns
// 16 outputs, 16 threads per output each summing every 16th partial, folded through LDS in a
// fixed order.
__global__ __launch_bounds__(256) void conv1_wgrad_f32_fold_kernel(
    const float* __restrict__ ws, float* __restrict__ dw, float* __restrict__ db, int nblocks) {
    __shared__ float red[16][17];
    const int ol = threadIdx.x & 15, part = threadIdx.x >> 4;
    const int i = blockIdx.x * 16 + ol;
    float s = 0.f;
    for (int b = part; b < nblocks; b += 16) s += ws[(long)b * 2048 + i];
    red[part][ol] = s;
    __syncthreads();
    if (part == 0) {
        float t = 0.f;
#pragma unroll
        for (int q = 0; q < 16; ++q) t += red[q][ol];
        const int co = i >> 5, k = i & 31;
        if (k < 27) dw[co * 27 + k] = t;
        else if (k == 27 && db) db[co] = t;
    }
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

__global__ __launch_bounds__(256) void zero_buffers_kernel(ZeroJobs jobs) {
    float4* p = (float4*)jobs.ptr[blockIdx.y];
    const size_t n4 = jobs.count[blockIdx.y] / 4;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256)
        p[i] = float4{0.f, 0.f, 0.f, 0.f};
}

void check_shape(const ConvShape& s, const char* who) {
    if (s.batch <= 0 || s.hw < 1 || s.hw > 32 || s.cin % 64 || s.cout % 64 || s.cin < 64 ||
        s.cout < 64)
        throw std::invalid_argument(std::string(who) + ": channels must be multiples of 64, "
                                                       "1 <= hw <= 32, batch > 0");
}

}  // namespace

namespace {
template <int MT, int NT, bool PM>
void launch_igemm(const IgemmParams& p, ConvEpilogue epi, unsigned nblocks, hipStream_t stream) {
    const dim3 blocks(nblocks, PM ? p.tap_splits : 1);
    switch (epi) {
        case EPI_NONE:
            hipLaunchKernelGGL((conv3x3_igemm_f32_kernel<MT, NT, PM, EPI_NONE>), blocks,
                               dim3(256), 0, stream, p);
            break;
        case EPI_BIAS:
            hipLaunchKernelGGL((conv3x3_igemm_f32_kernel<MT, NT, PM, EPI_BIAS>), blocks,
                               dim3(256), 0, stream, p);
            break;
        case EPI_BIAS_RELU:
            hipLaunchKernelGGL((conv3x3_igemm_f32_kernel<MT, NT, PM, EPI_BIAS_RELU>), blocks,
                               dim3(256), 0, stream, p);
            break;
        case EPI_MASK:
            hipLaunchKernelGGL((conv3x3_igemm_f32_kernel<MT, NT, PM, EPI_MASK>), blocks,
                               dim3(256), 0, stream, p);
            break;
    }
}
}  // namespace

void zero_buffers(const ZeroJobs& jobs, hipStream_t stream) {
    if (jobs.n <= 0) return;
    if (jobs.n > ZeroJobs::kMax) throw std::invalid_argument("zero_buffers: too many buffers");
    for (int j = 0; j < jobs.n; ++j)
        if (!jobs.ptr[j] || jobs.count[j] % 4 || ((uintptr_t)jobs.ptr[j] & 15))
            throw std::invalid_argument("zero_buffers: buffers must be 16-byte aligned, counts % 4 == 0");
    hipLaunchKernelGGL(zero_buffers_kernel, dim3(256, jobs.n), dim3(256), 0, stream, jobs);
}

namespace {
// the launch decisions of conv3x3_igemm_f32, shared with conv3x3_igemm_f32_tap_splits
struct IgemmPlan { bool pm, big; unsigned blocks; int tap_splits; };
IgemmPlan plan_igemm(const ConvShape& s, ConvEpilogue epi, int force_tile, int force_pm,
                     int force_split);
}  // namespace

int conv3x3_igemm_f32_tap_splits(ConvShape s, ConvEpilogue epi) {
    check_shape(s, "conv3x3_igemm_f32");
    return plan_igemm(s, epi, 0, -1, -1).tap_splits;
}

void conv3x3_igemm_f32(const float* x, const float* w, float* y, const float* bias,
                       const float* mask, ConvShape s, ConvEpilogue epi, hipStream_t stream,
                       int force_tile, int force_pm, int force_split, bool pre_zeroed,
                       const StreamSignal* signal) {
    check_shape(s, "conv3x3_igemm_f32");
    if (!x || !w || !y) throw std::invalid_argument("conv3x3_igemm_f32: null tensor");
    if ((epi == EPI_BIAS || epi == EPI_BIAS_RELU) && !bias)
        throw std::invalid_argument("conv3x3_igemm_f32: bias missing");
    if (epi == EPI_MASK && !mask) throw std::invalid_argument("conv3x3_igemm_f32: mask missing");
    if ((int)epi < 0 || (int)epi > 3) throw std::invalid_argument("conv3x3_igemm_f32: bad epilogue");
    IgemmParams p;
    p.x = x; p.w = w; p.y = y; p.bias = bias; p.mask = mask;
    p.M = s.batch * s.hw * s.hw; p.N = s.cout; p.cin = s.cin; p.hw = s.hw;
    p.log_hw = log2_exact(s.hw); p.batch = s.batch;
    p.signal = signal ? signal->flag : nullptr;
    p.signal_value = signal ? signal->value : 0u;
    const IgemmPlan pl = plan_igemm(s, epi, force_tile, force_pm, force_split);
    p.tap_splits = pl.tap_splits;
    if (pl.tap_splits == 2 && !pre_zeroed)
        if (hipMemsetAsync(y, 0, (size_t)p.M * p.N * sizeof(float), stream) != hipSuccess)
            throw std::runtime_error("conv3x3_igemm_f32: hipMemsetAsync failed");
    if (pl.big && pl.pm) launch_igemm<2, 2, true>(p, epi, pl.blocks, stream);
    else if (pl.big) launch_igemm<2, 2, false>(p, epi, pl.blocks, stream);
    else if (pl.pm) launch_igemm<1, 1, true>(p, epi, pl.blocks, stream);
    else launch_igemm<1, 1, false>(p, epi, pl.blocks, stream);
}

namespace {
IgemmPlan plan_igemm(const ConvShape& s, ConvEpilogue epi, int force_tile, int force_pm,
                     int force_split) {
    const long M = (long)s.batch * s.hw * s.hw;
    const int N = s.cout;
    // position-major tiles (padded taps skipped) on the small maps, when a tile of images fills
    bool pm = s.hw <= 8 && s.hw >= 2 && s.batch >= 64;
    if (force_pm == 0) pm = false;
    else if (force_pm == 1) pm = s.hw >= 2;
    else if (force_pm != -1) throw std::invalid_argument("conv3x3_igemm_f32: pm must be -1, 0 or 1");
    // 128 x 128 tiles where they still give every CU two blocks, 64 x 64 tiles otherwise
    auto blocks_for = [&](int tile) {
        const long mt = pm ? (long)s.hw * s.hw * ((s.batch + tile - 1) / tile) : (M + tile - 1) / tile;
        return mt * (N / tile);
    };
    // (measured at batch 256, tools/bench_conv_f32.py: on the position-major layers 64 x 64 tiles
    // win by 8-13 % even where 128 x 128 tiles would still cover the chip, and on the 16x16 layer
    // 128 x 128 tiles win by 5 %)
    bool big = N % 128 == 0 && !pm && blocks_for(128) >= 512;
    if (force_tile == 128) {
        if (N % 128) throw std::invalid_argument("conv3x3_igemm_f32: 128 tiles need cout % 128 == 0");
        big = true;
    } else if (force_tile == 64) {
        big = false;
    } else if (force_tile != 0) {
        throw std::invalid_argument("conv3x3_igemm_f32: tile must be 64 or 128");
    }
    IgemmPlan pl;
    pl.pm = pm;
    pl.big = big;
    pl.blocks = (unsigned)blocks_for(big ? 128 : 64);
    // Position-major tiles on a map too small to give every CU a block (2x2 at batch 256: 128
    // blocks): the taps of a position go to two blocks, see IgemmParams::tap_splits.  Also worth
    // it on the 4x4 maps whose grid does cover the chip (512 -> 512 @ 4x4: 184 against 206 us
    // forward, 174 against 197 us backward-data); on 8x8 it loses 3-5 %.
    pl.tap_splits = 1;
    if (pm && epi != EPI_BIAS_RELU && (pl.blocks < 256 || s.hw <= 4) && force_split != 0)
        pl.tap_splits = 2;
    if (force_split == 1 && pm && epi != EPI_BIAS_RELU) pl.tap_splits = 2;
    return pl;
}
}  // namespace

void repack_dgrad_weights_f32(const float* wf, float* wd, int cout, int cin, hipStream_t stream) {
    if (cout <= 0 || cin <= 0) throw std::invalid_argument("repack_dgrad_weights_f32: bad shape");
    const size_t total = (size_t)cout * 9 * cin;
    hipLaunchKernelGGL(repack_dgrad_f32_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                       stream, wf, wd, cout, cin, total);
}

namespace {
struct WgradPlan { int tiles, splits, chunks_per_split, total_chunks; bool wide, pm; };

WgradPlan plan_wgrad(const ConvShape& s, int force_splits, int force_pm) {
    WgradPlan pl;
    pl.wide = s.cin % 128 == 0;
    if (s.cout % 128) throw std::invalid_argument("conv3x3_wgrad_f32: cout must be a multiple of 128");
    pl.tiles = (s.cout / 128) * (s.cin / (pl.wide ? 128 : 64));
    // (position-major chunks also pay at 16x16: 122 against 154 us at 64 -> 128, batch 256)
    pl.pm = s.hw <= 16 && s.hw >= 2 && s.batch >= kBK;
    if (force_pm == 0) pl.pm = false;
    else if (force_pm == 1) pl.pm = s.hw >= 2;
    const int npix = s.hw * s.hw;
    const int chunks = pl.pm ? npix * ((s.batch + kBK - 1) / kBK) : (s.batch * npix + kBK - 1) / kBK;
    pl.total_chunks = chunks;
    // about two blocks per CU, at least four chunks of pixels per block; three times as many on
    // the 4x4 maps, where position-major blocks skip most of their chunks (512 -> 512 @ 4x4,
    // batch 256: 174 us with 12 splits against 217 us with 4)
    const int target = pl.pm && s.hw <= 4 ? 1536 : 512;
    int splits = force_splits > 0 ? force_splits : (target + pl.tiles * 9 - 1) / (pl.tiles * 9);
    const int max_splits = chunks / 4 > 0 ? chunks / 4 : 1;
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
    int cps = (chunks + splits - 1) / splits;
    // PM: whole sweeps over the positions per split, so that every split of a tap skips alike
    if (pl.pm && cps > npix) cps = (cps + npix - 1) / npix * npix;
    pl.chunks_per_split = cps;
    pl.splits = (chunks + cps - 1) / cps;
    return pl;
}
}  // namespace

int conv3x3_wgrad_f32_splits(ConvShape s, int force_splits, int force_pm) {
    check_shape(s, "conv3x3_wgrad_f32");
    return plan_wgrad(s, force_splits, force_pm).splits;
}

void conv3x3_wgrad_f32(const float* x, const float* dy, float* dw, ConvShape s, float* slab,
                       hipStream_t stream, int force_splits, int force_pm) {
    check_shape(s, "conv3x3_wgrad_f32");
    if (!x || !dy || !dw) throw std::invalid_argument("conv3x3_wgrad_f32: null tensor");
    const WgradPlan pl = plan_wgrad(s, force_splits, force_pm);
    if (pl.splits > 1 && !slab)
        throw std::invalid_argument("conv3x3_wgrad_f32: this shape splits the pixels and needs a slab");
    WgradParams p;
    p.x = x; p.dy = dy;
    p.out = pl.splits > 1 ? slab : dw;
    p.slab_stride = (long)s.cout * 9 * s.cin;
    p.M = s.batch * s.hw * s.hw; p.cin = s.cin; p.cout = s.cout; p.hw = s.hw;
    p.log_hw = log2_exact(s.hw); p.chunks_per_split = pl.chunks_per_split;
    p.total_chunks = pl.total_chunks; p.batch = s.batch;
    const dim3 grid((unsigned)(pl.tiles * 9), (unsigned)pl.splits);
    if (pl.wide && pl.pm)
        hipLaunchKernelGGL((conv3x3_wgrad_f32_kernel<2, 2, true>), grid, dim3(256), 0, stream, p);
    else if (pl.wide)
        hipLaunchKernelGGL((conv3x3_wgrad_f32_kernel<2, 2, false>), grid, dim3(256), 0, stream, p);
    else if (pl.pm)
        hipLaunchKernelGGL((conv3x3_wgrad_f32_kernel<2, 1, true>), grid, dim3(256), 0, stream, p);
    else
        hipLaunchKernelGGL((conv3x3_wgrad_f32_kernel<2, 1, false>), grid, dim3(256), 0, stream, p);
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
    hipLaunchKernelGGL(conv1_wgrad_f32_fold_kernel, dim3(128), dim3(256), 0, stream,
                       (const float*)ws, dw, db, nb);
}

}  // namespace dml
