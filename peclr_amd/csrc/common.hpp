// Shared device helpers for libpeclr_hip (gfx950 only: wave64, MFMA, 160 KiB LDS).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/peclr_hip.h"

namespace peclr {

constexpr int kWave = 64;

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// Row of a 32x32 MFMA accumulator register: acc[r] of lane l holds C[row][l & 31].
__device__ __forceinline__ int mfma32_row(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, kWave);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, kWave));
    return v;
}
__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, kWave));
    return v;
}

// ---- exact three-way bf16 split of fp32 numbers (the "x6" GEMMs): x == h + m + l with h = bf16(x), m = bf16(x - h),
// l = x - h - m, every step round-to-nearest-even (v_cvt_pk_bf16_f32 converts two numbers per instruction) and every
// residual exact in fp32 (h keeps 8 significand bits, x - h fits 16, m keeps 8 of those, l the last 8).  Rounding to
// nearest instead of truncating makes the three dropped partial products (m.l, l.m, l.l <= 2^-27 |a.b|) zero-mean: their sum
// does not grow like a bias with the length of the contraction.  Two numbers in, three packed pairs out (element 0 in the
// low half, as the MFMA fragments want consecutive k).
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pk_bf16(float a, float b) {
    const f32x2_t v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
}
// (the subtractions go through one-instruction asm statements: left to itself hipcc pairs them into v_pk_add_f32, whose
// aligned 64-bit operands cost more v_mov than the packing saves -- the weight-gradient kernel ran 2x slower that way)
__device__ __forceinline__ float sub_f32(float a, float b) {
    float r;
    asm("v_sub_f32_e32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ void split3_pk(float x0, float x1, unsigned& h, unsigned& m, unsigned& l) {
    h = pk_bf16(x0, x1);
    const float r0 = sub_f32(x0, __uint_as_float(h << 16)), r1 = sub_f32(x1, __uint_as_float(h & 0xFFFF0000u));
    m = pk_bf16(r0, r1);
    l = pk_bf16(sub_f32(r0, __uint_as_float(m << 16)), sub_f32(r1, __uint_as_float(m & 0xFFFF0000u)));
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// Returns the hipError_t of the most recent launch as a positive int (0 = ok).
inline int launch_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? PECLR_OK : static_cast<int>(e);
}

}  // namespace peclr
