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

// ---- two-way fp16 split of SCALED fp32 numbers (the "pair" GEMMs, round 6): with s a power of two chosen per tensor so that
// max |s x| lies in [2^14, 2^15) (pair_scale below; fp16 overflows at 65 504), s x == hi + lo + e with hi = fp16(s x), lo =
// fp16(s x - hi) (the residual is exact in fp32), |e| <= 2^-23 |s x|: 11 + 1 (lo's sign) + 11 of fp32's 24 significand bits.
// Three of the four partial products (hi.hi, hi.lo, lo.hi; lo.lo <= 2^-22 of the product is dropped) on v_mfma_f32_32x32x16_f16
// with fp32 accumulation, the accumulators multiplied by 1 / (s_a s_b) -- exact, both are powers of two -- afterwards.  Elements
// below 2^-18 of the tensor's maximum lose low bits of `lo` gradually (fp16 subnormals; absolute error <= 2^-40 max |x|).
// Measured on real layer tensors (tools/exp/fp16_pair_probe.py, tools/exp/pair_probe.py): the error class of an fp32 BLAS GEMM,
// 1.2 - 2.4 x the six-product scheme's, a fraction of a k-ordered fp32 chain's (what v_mfma_f32 computes).
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pk_f16(float a, float b) {
    const f32x2_t v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, f16x2_t));      // round to nearest even
}
__device__ __forceinline__ float f16_lo(unsigned w) { return (float)__builtin_bit_cast(_Float16, (unsigned short)w); }
__device__ __forceinline__ float f16_hi(unsigned w) { return (float)__builtin_bit_cast(_Float16, (unsigned short)(w >> 16)); }
__device__ __forceinline__ void split2_pk(float x0, float x1, float s, unsigned& h, unsigned& l) {
    const float y0 = x0 * s, y1 = x1 * s;                    // exact: s is a power of two
    h = pk_f16(y0, y1);
    l = pk_f16(sub_f32(y0, f16_lo(h)), sub_f32(y1, f16_hi(h)));
}
// the power of two s with amax * s in [2^14, 2^15); exponent clamped to +-50 (amax = 0 / subnormal / inf / nan: any s will do --
// zeros stay zeros, non-finite operands give non-finite results as they would in fp32)
__host__ __device__ __forceinline__ float pair_scale(float amax) {
    unsigned bits;
    __builtin_memcpy(&bits, &amax, 4);
    int k = 14 - ((int)((bits >> 23) & 0xFFu) - 127);
    k = k < -50 ? -50 : (k > 50 ? 50 : k);
    bits = (unsigned)(k + 127) << 23;
    float s;
    __builtin_memcpy(&s, &bits, 4);
    return s;
}

// The entry of a pack table (8 x int64 per entry, field 6 = the entry's first chunk) that chunk `blockIdx.x` of a launch belongs to:
// the LAST entry whose first chunk is <= blockIdx.x.  Every thread tests a few entries and the one that holds the chunk says so
// through LDS -- a workgroup-uniform linear walk over the table was ~100 dependent loads per workgroup (tables of 106 ... 310
// matrices): 88 us for a pass that moves 94 MB.
__device__ __forceinline__ int pack_entry_of_chunk(const int64_t* table, int count) {
    __shared__ int sel;
    const int64_t c = (int64_t)blockIdx.x;
    for (int i = threadIdx.x; i < count; i += blockDim.x) {
        const int64_t b = table[8 * i + 6];
        const bool last = i + 1 == count || c < table[8 * (i + 1) + 6];
        if (c >= b && last) sel = i;                          // (entries that own no chunk share their successor's first chunk: never chosen)
    }
    __syncthreads();
    return sel;
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// Returns the hipError_t of the most recent launch as a positive int (0 = ok).
inline int launch_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? PECLR_OK : static_cast<int>(e);
}

}  // namespace peclr
