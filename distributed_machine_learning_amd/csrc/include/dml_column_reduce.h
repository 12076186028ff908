// Two-level column reduction over a [rows][c] matrix without atomics on the results.
//
// Column sums over activations (BatchNorm statistics, BatchNorm backward sums, bias
// gradients) have thousands of rows per column.  Adding every block's partial with a
// float atomic makes all blocks hit the same few addresses, and same-address atomics
// serialise at the memory side (measured here: ~15 us for a 2 MB tensor); a ticket-based
// in-launch fold is no better because every block's release fence has to write back the
// L2 lines the producing kernel just dirtied.  So: kernel 1 streams the rows and stores one
// partial row per block into a small slab (plain stores), kernel 2 (column_fold) sums the
// slab with 4 threads per column and runs the caller's finalisation (mean / rstd / running
// statistics, dgamma / dbeta, bias gradient).  The kernel boundary is the hand-off.
#pragma once

#include "dml_common.h"

namespace dml {

constexpr int kColReduceSlabFloats = 131072;                       // 512 KiB
constexpr int kColReduceShiftFloats = 2048;    // one per-column vector behind the slab
constexpr int kColReduceWsFloats = kColReduceSlabFloats + kColReduceShiftFloats;

struct ColReducePlan {
    int blocks, rows_per_block;
};

// Two rows per thread where the caps below allow - all of a thread's loads in flight at once,
// these kernels are latency-bound on every layer but the first - at most 512 blocks (two per CU), slab
// (blocks * 2c floats) within 512 KiB, rows_per_block a multiple of the row lanes of a block.
inline ColReducePlan plan_col_reduce(int rows, int c) {
    const int lanes = 256 / (c / 8);
    long long want = (rows + lanes * 2 - 1) / (lanes * 2);
    if (want < 1) want = 1;
    if (want > 512) want = 512;
    const int cap = kColReduceSlabFloats / (2 * c);
    if (want > cap) want = cap;
    int rpb = (int)((rows + want - 1) / want);
    rpb = (rpb + lanes - 1) / lanes * lanes;
    return {(rows + rpb - 1) / rpb, rpb};
}

// Kernel-1 body.  row_values(r, cg, a, b): accumulate row r, column group cg (8 columns)
// into a[8], b[8].  `red` is 4096 floats of LDS; `ws` is kColReduceWsFloats floats of global
// scratch; block b writes ws[b * 2c .. b * 2c + 2c).
template <typename RowFn>
__device__ __forceinline__ void column_partials(RowFn&& row_values, float* ws, int rows, int c,
                                                int rows_per_block, float* red) {
    const int c8 = c / 8, lanes = 256 / c8;
    const int cg = threadIdx.x % c8, rl = threadIdx.x / c8;
    const int r0 = blockIdx.x * rows_per_block;
    const int r1 = min(rows, r0 + rows_per_block);
    float a[8], b[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = b[i] = 0.f;
    int r = r0 + rl;
    for (; r + 3 * lanes < r1; r += 4 * lanes) {   // four rows in flight per thread
        row_values(r, cg, a, b);
        row_values(r + lanes, cg, a, b);
        row_values(r + 2 * lanes, cg, a, b);
        row_values(r + 3 * lanes, cg, a, b);
    }
    for (; r < r1; r += lanes) row_values(r, cg, a, b);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        red[threadIdx.x * 8 + i] = a[i];
        red[2048 + threadIdx.x * 8 + i] = b[i];
    }
    __syncthreads();
    float* slab = ws + (size_t)blockIdx.x * 2 * c;
    for (int col = threadIdx.x; col < 2 * c; col += 256) {
        const int which = col >= c ? 1 : 0;
        const int cc = col - which * c;
        const int g = cc / 8, i = cc % 8;
        float s = 0.f;
        for (int l = 0; l < lanes; ++l) s += red[which * 2048 + (l * c8 + g) * 8 + i];
        slab[col] = s;
    }
}

// Kernel-2 body: grid = fold_blocks(c) blocks of 256 threads = 16 columns x 16 row parts (a
// fold is latency-bound: many short dependent chains beat few long ones).
// finalize(col, sum_a, sum_b) runs once per column.
inline int fold_blocks(int c) { return (c + 15) / 16; }

template <typename FinalFn>
__device__ __forceinline__ void column_fold(FinalFn&& finalize, const float* ws, int nb, int c) {
    __shared__ float fold[2][16][17];
    const int cl = threadIdx.x & 15, part = threadIdx.x >> 4;
    const int col = blockIdx.x * 16 + cl;
    float sa = 0.f, sb = 0.f;
    if (col < c) {
        int k = part;
        for (; k + 7 * 16 < nb; k += 8 * 16) {        // 16 loads in flight
            float va[8], vb[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                va[q] = ws[(size_t)(k + q * 16) * 2 * c + col];
                vb[q] = ws[(size_t)(k + q * 16) * 2 * c + c + col];
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) { sa += va[q]; sb += vb[q]; }
        }
        for (; k < nb; k += 16) {
            sa += ws[(size_t)k * 2 * c + col];
            sb += ws[(size_t)k * 2 * c + c + col];
        }
    }
    fold[0][part][cl] = sa;
    fold[1][part][cl] = sb;
    __syncthreads();
    if (part == 0 && col < c) {
        float ta = 0.f, tb = 0.f;
#pragma unroll
        for (int q = 0; q < 16; ++q) { ta += fold[0][q][cl]; tb += fold[1][q][cl]; }
        finalize(col, ta, tb);
    }
}

}  // namespace dml
