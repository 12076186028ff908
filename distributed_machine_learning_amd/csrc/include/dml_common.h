// Shared device-side helpers for the gfx950 (CDNA4) kernels.
//
// Conventions used by every kernel in this directory:
//   * wavefront = 64 lanes, workgroups are multiples of 64 threads;
//   * activations are NHWC bf16, master parameters / gradients / statistics are fp32;
//   * bf16 travels through memory as 16-byte (8-element) vectors wherever possible.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

namespace dml {

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

constexpr int kWave = 64;

__device__ __forceinline__ float bf16_bits_to_float(uint16_t b) {
    return __builtin_bit_cast(float, (uint32_t)b << 16);
}

__device__ __forceinline__ uint16_t float_to_bf16_bits(float f) {
    // plain cast: hipcc emits v_cvt_pk_bf16_f32 (round-to-nearest-even, NaN preserved)
    return __builtin_bit_cast(uint16_t, (bf16_t)f);
}

// 8 bf16 packed in a uint4 -> 8 floats
__device__ __forceinline__ void unpack8(const uint4& v, float (&f)[8]) {
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        f[2 * i] = __builtin_bit_cast(float, w[i] << 16);
        f[2 * i + 1] = __builtin_bit_cast(float, w[i] & 0xffff0000u);
    }
}

__device__ __forceinline__ uint32_t pack2(float lo, float hi) {
    return (uint32_t)float_to_bf16_bits(lo) | ((uint32_t)float_to_bf16_bits(hi) << 16);
}

__device__ __forceinline__ uint4 pack8(const float (&f)[8]) {
    uint4 v;
    v.x = pack2(f[0], f[1]);
    v.y = pack2(f[2], f[3]);
    v.z = pack2(f[4], f[5]);
    v.w = pack2(f[6], f[7]);
    return v;
}

// Eight consecutive activation elements <-> eight floats, for both storage types the kernels
// are instantiated with: uint16_t = bf16 bit patterns (one 16-byte vector), float (two).
__device__ __forceinline__ void load8(const uint16_t* p, float (&f)[8]) {
    unpack8(*(const uint4*)p, f);
}
__device__ __forceinline__ void load8(const float* p, float (&f)[8]) {
    const float4 a = *(const float4*)p, b = *(const float4*)(p + 4);
    f[0] = a.x; f[1] = a.y; f[2] = a.z; f[3] = a.w; f[4] = b.x; f[5] = b.y; f[6] = b.z; f[7] = b.w;
}
__device__ __forceinline__ void store8(uint16_t* p, const float (&f)[8]) {
    *(uint4*)p = pack8(f);
}
__device__ __forceinline__ void store8(float* p, const float (&f)[8]) {
    *(float4*)p = float4{f[0], f[1], f[2], f[3]};
    *(float4*)(p + 4) = float4{f[4], f[5], f[6], f[7]};
}
// the same eight elements moved without conversion
template <typename T> struct Raw8;
template <> struct Raw8<uint16_t> { uint4 v = {0, 0, 0, 0}; };
template <> struct Raw8<float> { float4 a = {0.f, 0.f, 0.f, 0.f}, b = {0.f, 0.f, 0.f, 0.f}; };
__device__ __forceinline__ Raw8<uint16_t> load_raw8(const uint16_t* p) {
    Raw8<uint16_t> r;
    r.v = *(const uint4*)p;
    return r;
}
__device__ __forceinline__ Raw8<float> load_raw8(const float* p) {
    Raw8<float> r;
    r.a = *(const float4*)p;
    r.b = *(const float4*)(p + 4);
    return r;
}
__device__ __forceinline__ void store_raw8(uint16_t* p, const Raw8<uint16_t>& r) { *(uint4*)p = r.v; }
__device__ __forceinline__ void store_raw8(float* p, const Raw8<float>& r) {
    *(float4*)p = r.a;
    *(float4*)(p + 4) = r.b;
}
__device__ __forceinline__ float elem_to_float(uint16_t v) { return bf16_bits_to_float(v); }
__device__ __forceinline__ float elem_to_float(float v) { return v; }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// Sum over each aligned group of 16 lanes (one DPP row), result in all 16 lanes: two quad
// permutes, then half-row and row mirrors - four VALU ops, no LDS traffic.
__device__ __forceinline__ float row16_sum(float v) {
    auto dpp_add = [](float x, auto ctrl) {
        const int y = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), decltype(ctrl)::value,
                                                  0xf, 0xf, true);
        return x + __builtin_bit_cast(float, y);
    };
    v = dpp_add(v, std::integral_constant<int, 0xB1>{});     // quad_perm [1,0,3,2]
    v = dpp_add(v, std::integral_constant<int, 0x4E>{});     // quad_perm [2,3,0,1]
    v = dpp_add(v, std::integral_constant<int, 0x141>{});    // row_half_mirror
    v = dpp_add(v, std::integral_constant<int, 0x140>{});    // row_mirror
    return v;
}

// Wave-wide sum without LDS traffic: DPP inside each 16-lane row, then the four row totals
// through scalar lane reads (the result is wave-uniform).
__device__ __forceinline__ float wave_sum_dpp(float v) {
    const int r = __builtin_bit_cast(int, row16_sum(v));
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(r, 0)) +
           __builtin_bit_cast(float, __builtin_amdgcn_readlane(r, 16)) +
           __builtin_bit_cast(float, __builtin_amdgcn_readlane(r, 32)) +
           __builtin_bit_cast(float, __builtin_amdgcn_readlane(r, 48));
}

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
    return v;
}

}  // namespace dml
