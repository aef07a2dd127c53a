// Fused BatchNorm2d (+ residual add) (+ ReLU) for the NHWC ResNet backbone, forward and backward,
// fp32 or bf16 activations (statistics, parameters and all arithmetic in fp32).
//
// Scope: SURVEY.md section 8f rank 4 ("fused ... backbone epilogue").  The convolutions stay on
// PyTorch-ROCm/MIOpen; what moves here is everything BETWEEN them: nn.BatchNorm2d (train/eval),
// the block-end `out += identity` and nn.ReLU of torchvision's BasicBlock/Bottleneck
// (resnet_model.py:15 builds them with norm_layer=nn.BatchNorm2d).  On MI355X that glue is 40 % of
// the fp32 ResNet-50 step (profiles/r01c: MIOpen BN fwd/bwd 22.7 % + add/threshold/clamp 14 %),
// all of it HBM-bound streaming with per-channel parameters.
//
// Layout: activations are NHWC (torch.channels_last), i.e. a row-major [R = N*H*W, C] matrix with
// the channel contiguous, so a channel is a COLUMN: the same "column statistics over rows" shape as
// the head's BatchNorm1d.  A thread owns one 16-byte word of channels (4 fp32 / 8 bf16) for the whole
// kernel (tid -> (row lane, column group) with the column group fastest, so consecutive lanes read
// consecutive 16-byte words: every wave-instruction is a 1 KiB contiguous burst) and walks rows with
// U independent loads per stream in flight.
//
//   forward  = stats (partial sum / sum-of-squares per row slice, shifted by row 0 to avoid
//              cancellation)  ->  finalize (fixed-order combine in float64, running stats)
//              ->  apply: y = relu(x*scale + shift + residual)
//   backward = reduce (partial dbeta, dgamma with the ReLU mask recomputed from x, or read from y
//              when a residual was added)  ->  finalize  ->  apply: dx (and d_residual = masked dy)
//
// When a residual is added before the ReLU the mask cannot be recomputed from x: the forward then
// writes a 1-bit-per-element mask ([R][C/32] words, 1/8 byte per element) that both backward passes read
// instead of re-reading y.
// Algorithmic bytes per element (e = 4 fp32 / 2 bf16): fwd e (stats) + 2e (apply) [+e residual];
// bwd 2e (reduce) + 3e (apply) [+1/8 for the mask, twice, and +e d_residual when a residual was added].
// The stock path moves 2e+3e (BN) + 2e (ReLU) + 3e (add) forward and 3e (ReLU) + 5e (BN) backward.
// Everything is bit-reproducible (no atomics; fixed combine order).
#include <initializer_list>

#include "common.hpp"

namespace peclr {
namespace {

constexpr int T = 256;

// ---- W floats held in registers + the 16-byte global word they travel as
template <int W> struct Fv { float v[W]; };

// every activation word of the streaming kernels travels through these two.  PECLR_BN2D_NT (compile time; A/B builds): bit 0 = loads,
// bit 1 = stores carry the non-temporal hint.  Loads do by default: a linear read of 822 MB runs at 6.8 TB/s with the hint and at 4.3
// without (tools/exp/rw_mix.hip, cold caches); in the step `bn2d_apply` 96 -> 86 us, `bn2d_bwd_apply` 118 -> 110, `bn2d_bwd_reduce`
// 158 -> 122 (same box, C2 fp32).  Stores gain nothing (5.2 - 5.5 TB/s either way; the consumer finds the tail of the tensor in the
// memory-side cache).  The pooled stem kernels re-read rows across windows and LOSE with the hint (283 -> 380 us): `Quad` loads
// stay plain.
#ifndef PECLR_BN2D_NT
#define PECLR_BN2D_NT 1
#endif
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ u32x4_t ld128(const void* p) {
#if PECLR_BN2D_NT & 1
    return __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(p));
#else
    return *reinterpret_cast<const u32x4_t*>(p);
#endif
}
__device__ __forceinline__ u32x4_t ld128_cached(const void* p) { return *reinterpret_cast<const u32x4_t*>(p); }   // rows read again soon
__device__ __forceinline__ void st128(void* p, u32x4_t v) {
#if PECLR_BN2D_NT & 2
    __builtin_nontemporal_store(v, reinterpret_cast<u32x4_t*>(p));
#else
    *reinterpret_cast<u32x4_t*>(p) = v;
#endif
}

template <typename IO> struct Word;
template <> struct Word<float> {
    static constexpr int W = 4, U = 8;  // U = rows in flight per stream
    typedef float4 Raw;                 // the 16-byte word as it travels (kept raw while in flight: 4 registers)
    static __device__ __forceinline__ Raw load_raw(const float* p) {
        const u32x4_t t = ld128(p);
        return make_float4(__uint_as_float(t[0]), __uint_as_float(t[1]), __uint_as_float(t[2]), __uint_as_float(t[3]));
    }
    static __device__ __forceinline__ Fv<4> expand(const Raw& t) { return {{t.x, t.y, t.z, t.w}}; }
    static __device__ __forceinline__ Fv<4> load(const float* p) { return expand(load_raw(p)); }
    static __device__ __forceinline__ Fv<4> load_cached(const float* p) {
        const u32x4_t t = ld128_cached(p);
        return {{__uint_as_float(t[0]), __uint_as_float(t[1]), __uint_as_float(t[2]), __uint_as_float(t[3])}};
    }
    static __device__ __forceinline__ void store(float* p, const Fv<4>& a) {
        const u32x4_t t = {__float_as_uint(a.v[0]), __float_as_uint(a.v[1]), __float_as_uint(a.v[2]), __float_as_uint(a.v[3])};
        st128(p, t);
    }
    static __device__ __forceinline__ float round(float f) { return f; }       // the value a store + load would hand back
};
typedef uint16_t bf16_t;
template <> struct Word<bf16_t> {
    static constexpr int W = 8, U = 8;
    typedef uint4 Raw;
    static __device__ __forceinline__ Raw load_raw(const bf16_t* p) {
        const u32x4_t t = ld128(p);
        return make_uint4(t[0], t[1], t[2], t[3]);
    }
    static __device__ __forceinline__ Fv<8> expand(const Raw& t) {
        const unsigned w[4] = {t.x, t.y, t.z, t.w};
        Fv<8> r;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            r.v[2 * k] = __uint_as_float(w[k] << 16);
            r.v[2 * k + 1] = __uint_as_float(w[k] & 0xFFFF0000u);
        }
        return r;
    }
    static __device__ __forceinline__ Fv<8> load(const bf16_t* p) { return expand(load_raw(p)); }
    static __device__ __forceinline__ Fv<8> load_cached(const bf16_t* p) {
        const u32x4_t t = ld128_cached(p);
        return expand(make_uint4(t[0], t[1], t[2], t[3]));
    }
    static __device__ __forceinline__ unsigned rne(float f) {  // fp32 -> bf16, round to nearest even
        const unsigned u = __float_as_uint(f);
        return (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;
    }
    static __device__ __forceinline__ void store(bf16_t* p, const Fv<8>& a) {
        unsigned w[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) w[k] = rne(a.v[2 * k]) | (rne(a.v[2 * k + 1]) << 16);
        const u32x4_t t = {w[0], w[1], w[2], w[3]};
        st128(p, t);
    }
    static __device__ __forceinline__ float round(float f) { return __uint_as_float(rne(f) << 16); }
};
// fp16 activations (the reference's default precision=16 = native AMP): same 8-per-word geometry as bf16
typedef _Float16 f16_t;
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
template <> struct Word<f16_t> {
    static constexpr int W = 8, U = 8;
    typedef f16x8 Raw;
    static __device__ __forceinline__ Raw load_raw(const f16_t* p) { return __builtin_bit_cast(f16x8, ld128(p)); }
    static __device__ __forceinline__ Fv<8> expand(const Raw& t) {
        Fv<8> r;
#pragma unroll
        for (int k = 0; k < 8; ++k) r.v[k] = (float)t[k];
        return r;
    }
    static __device__ __forceinline__ Fv<8> load(const f16_t* p) { return expand(load_raw(p)); }
    static __device__ __forceinline__ Fv<8> load_cached(const f16_t* p) { return expand(__builtin_bit_cast(f16x8, ld128_cached(p))); }
    static __device__ __forceinline__ void store(f16_t* p, const Fv<8>& a) {
        f16x8 t;
#pragma unroll
        for (int k = 0; k < 8; ++k) t[k] = (_Float16)a.v[k];   // round to nearest even; overflow -> inf (GradScaler's job)
        st128(p, __builtin_bit_cast(u32x4_t, t));
    }
    static __device__ __forceinline__ float round(float f) { return (float)(_Float16)f; }
};
// per-channel fp32 parameters: W consecutive floats
template <int W> __device__ __forceinline__ Fv<W> loadp(const float* p) {
    Fv<W> r;
#pragma unroll
    for (int k = 0; k < W; k += 4) {
        const float4 t = *reinterpret_cast<const float4*>(p + k);
        r.v[k] = t.x; r.v[k + 1] = t.y; r.v[k + 2] = t.z; r.v[k + 3] = t.w;
    }
    return r;
}
template <int W> __device__ __forceinline__ void storep(float* p, const Fv<W>& a) {
#pragma unroll
    for (int k = 0; k < W; k += 4) *reinterpret_cast<float4*>(p + k) = make_float4(a.v[k], a.v[k + 1], a.v[k + 2], a.v[k + 3]);
}
template <int W> __device__ __forceinline__ Fv<W> zero() {
    Fv<W> r;
#pragma unroll
    for (int k = 0; k < W; ++k) r.v[k] = 0.f;
    return r;
}
// ---- FOUR elements per lane whatever the storage type (16-bit: 8-byte words).  For gather kernels whose register budget
// is set by the number of words in flight, not by the width of one (bn2d_pool_bwd_apply: four candidate windows per element).
template <typename IO> struct Quad;
template <> struct Quad<float> {
    static __device__ __forceinline__ Fv<4> load(const float* p) {
        const float4 t = *reinterpret_cast<const float4*>(p);
        return {{t.x, t.y, t.z, t.w}};
    }
    static __device__ __forceinline__ void store(float* p, const Fv<4>& a) { Word<float>::store(p, a); }
};
template <> struct Quad<bf16_t> {
    static __device__ __forceinline__ Fv<4> load(const bf16_t* p) {
        const uint2 t = *reinterpret_cast<const uint2*>(p);
        return {{__uint_as_float(t.x << 16), __uint_as_float(t.x & 0xFFFF0000u), __uint_as_float(t.y << 16), __uint_as_float(t.y & 0xFFFF0000u)}};
    }
    static __device__ __forceinline__ void store(bf16_t* p, const Fv<4>& a) {
        *reinterpret_cast<uint2*>(p) = make_uint2(Word<bf16_t>::rne(a.v[0]) | (Word<bf16_t>::rne(a.v[1]) << 16),
                                                  Word<bf16_t>::rne(a.v[2]) | (Word<bf16_t>::rne(a.v[3]) << 16));
    }
};
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
template <> struct Quad<f16_t> {
    static __device__ __forceinline__ Fv<4> load(const f16_t* p) {
        const f16x4 t = *reinterpret_cast<const f16x4*>(p);
        return {{(float)t[0], (float)t[1], (float)t[2], (float)t[3]}};
    }
    static __device__ __forceinline__ void store(f16_t* p, const Fv<4>& a) {
        f16x4 t;
#pragma unroll
        for (int k = 0; k < 4; ++k) t[k] = (_Float16)a.v[k];
        *reinterpret_cast<f16x4*>(p) = t;
    }
};

// Geometry shared by every kernel: CW = C/W column groups; a block covers CGB = min(CW, 256) of them
// (blockIdx.x = column block) and RPP = 256/CGB rows per pass; blockIdx.y = row slice.
struct Geo {
    int R, C, CGB, RPP, rows_per_block;
    int rev;   // row slices are taken back to front: the workgroups dispatched first own the LAST rows -- the ones the GEMM that
               // produced the tensor wrote last and the 256 MB memory-side cache still holds (a 411 MB tensor read back to front
               // right after it was written front to back: 4.6 instead of 3.4 TB/s, tools/exp/mall_order.hip), and the GEMM
               // that consumes this kernel's output (front to back) starts on the rows written last.  Results do not change:
               // slices keep their logical index.
};
__device__ __forceinline__ int slice_y(const Geo& g) { return g.rev ? (int)(gridDim.y - 1 - blockIdx.y) : (int)blockIdx.y; }
template <int W>
__device__ __forceinline__ void thread_geo(const Geo& g, int& col, int& r_begin, int& r_end, int& rl) {
    const int cgl = threadIdx.x % g.CGB;
    rl = threadIdx.x / g.CGB;
    col = (blockIdx.x * g.CGB + cgl) * W;
    r_begin = slice_y(g) * g.rows_per_block;
    r_end = min(g.R, r_begin + g.rows_per_block);
}

// Sum over the block's row lanes (threads with equal column group); valid in row lane 0.
template <int W>
__device__ __forceinline__ Fv<W> lane_reduce(Fv<W> v, float* red, const Geo& g, int rl) {
    if (g.RPP == 1) return v;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < W; ++k) red[k * T + threadIdx.x] = v.v[k];
    __syncthreads();
    Fv<W> t = zero<W>();
    if (rl == 0)
        for (int l = 0; l < g.RPP; ++l)
#pragma unroll
            for (int k = 0; k < W; ++k) t.v[k] += red[k * T + l * g.CGB + threadIdx.x];
    return t;
}

// ------------------------------------------------------------------ forward: statistics
// partial: [n_split][2][C] sums, followed by one extra row [C] = the shift (row 0 of x as fp32).
template <typename IO>
__global__ __launch_bounds__(T) void bn2d_stats_kernel(const IO* __restrict__ x, const float* __restrict__ shift, Geo g,
                                                       int n_split, float* __restrict__ partial) {
    constexpr int W = Word<IO>::W, U = Word<IO>::U;
    __shared__ float red[W * T];
    int col, r0, r1, rl;
    thread_geo<W>(g, col, r0, r1, rl);
    // shift: row 0 of x (same for every block), or a caller-provided vector that is identical on every
    // rank (synchronised statistics: partial sums of different ranks must share their shift)
    const Fv<W> k0 = shift ? loadp<W>(shift + col) : Word<IO>::load(x + col);
    Fv<W> s = zero<W>(), q = zero<W>();
    auto acc = [&](const Fv<W>& v) {
#pragma unroll
        for (int k = 0; k < W; ++k) {
            const float d = v.v[k] - k0.v[k];
            s.v[k] += d;
            q.v[k] = fmaf(d, d, q.v[k]);
        }
    };
    int r = r0 + rl;
    for (; r + (U - 1) * g.RPP < r1; r += U * g.RPP) {
        Fv<W> v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = Word<IO>::load(x + (size_t)(r + u * g.RPP) * g.C + col);
#pragma unroll
        for (int u = 0; u < U; ++u) acc(v[u]);
    }
    for (; r < r1; r += g.RPP) acc(Word<IO>::load(x + (size_t)r * g.C + col));
    s = lane_reduce<W>(s, red, g, rl);
    q = lane_reduce<W>(q, red, g, rl);
    if (rl == 0) {
        float* o = partial + (size_t)slice_y(g) * 2 * g.C;
        storep<W>(o + col, s);
        storep<W>(o + g.C + col, q);
        if (slice_y(g) == 0) storep<W>(partial + (size_t)n_split * 2 * g.C + col, k0);
    }
}

// Finalize kernels: a 1024-thread workgroup owns 32 channels; its 32 row lanes each combine every 32nd
// row-slice partial in float64 (coalesced 128-byte reads), then lane 0 adds the 32 lane sums in a fixed order.
constexpr int FT = 1024, FC = 32, FL = FT / FC;
// rs: floats between consecutive rows of the table (2 C, or a multiple of it when the table was folded: see
// bn2d_fold_partials_kernel)
__device__ __forceinline__ void combine_partials(const float* __restrict__ partial, int n_split, int C, int c, int lane,
                                                 double (*red)[2][FC], double& a, double& b, size_t rs) {
    if (C % 4 == 0 && n_split >= 256) {
        // long tables (128-row GEMM tiles of the first layers: 3 000 - 12 000 rows; 19 - 26 us per launch with 32 row lanes of
        // 4-byte loads): 128 row lanes x 8 channel quads, 16-byte loads, eight in flight per thread; the eight row lanes of a wave
        // fold by shuffles (a fixed tree), the sixteen waves through `red` in order.  (Fewer channels per workgroup -- more CUs
        // loading -- was measured too: 9 us at 64 channels, 29 at 256: 32-byte pieces of 128-byte lines.)
        const int cq = threadIdx.x & 7, rl = threadIdx.x >> 3;
        const int c0 = blockIdx.x * FC + 4 * cq;
        double s4[4] = {0.0, 0.0, 0.0, 0.0}, q4[4] = {0.0, 0.0, 0.0, 0.0};
        if (c0 < C) {
            const float* p = partial + c0;
            int k = rl;
            for (; k + 3 * 128 < n_split; k += 4 * 128) {
                float4 sv[4], qv[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    sv[u] = *reinterpret_cast<const float4*>(p + (size_t)(k + 128 * u) * rs);
                    qv[u] = *reinterpret_cast<const float4*>(p + (size_t)(k + 128 * u) * rs + C);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    s4[0] += (double)sv[u].x; s4[1] += (double)sv[u].y; s4[2] += (double)sv[u].z; s4[3] += (double)sv[u].w;
                    q4[0] += (double)qv[u].x; q4[1] += (double)qv[u].y; q4[2] += (double)qv[u].z; q4[3] += (double)qv[u].w;
                }
            }
            for (; k < n_split; k += 128) {
                const float4 sv = *reinterpret_cast<const float4*>(p + (size_t)k * rs);
                const float4 qv = *reinterpret_cast<const float4*>(p + (size_t)k * rs + C);
                s4[0] += (double)sv.x; s4[1] += (double)sv.y; s4[2] += (double)sv.z; s4[3] += (double)sv.w;
                q4[0] += (double)qv.x; q4[1] += (double)qv.y; q4[2] += (double)qv.z; q4[3] += (double)qv.w;
            }
        }
#pragma unroll
        for (int off = 32; off >= 8; off >>= 1)
#pragma unroll
            for (int j = 0; j < 4; ++j) { s4[j] += __shfl_down(s4[j], off, 64); q4[j] += __shfl_down(q4[j], off, 64); }
        const int wv = threadIdx.x >> 6;
        if ((threadIdx.x & 63) < 8) {
#pragma unroll
            for (int j = 0; j < 4; ++j) { red[wv][0][4 * cq + j] = s4[j]; red[wv][1][4 * cq + j] = q4[j]; }
        }
        __syncthreads();
        a = b = 0.0;
        if (lane == 0)
            for (int w = 0; w < FT / 64; ++w) {
                a += red[w][0][threadIdx.x % FC];
                b += red[w][1][threadIdx.x % FC];
            }
        return;
    }
    double s = 0.0, q = 0.0;
    if (c < C) {
        int k = lane;
        for (; k + 3 * FL < n_split; k += 4 * FL) {  // 8 independent loads in flight
            const float s0 = partial[(size_t)k * rs + c], q0 = partial[(size_t)k * rs + C + c];
            const float s1 = partial[(size_t)(k + FL) * rs + c], q1 = partial[(size_t)(k + FL) * rs + C + c];
            const float s2 = partial[(size_t)(k + 2 * FL) * rs + c], q2 = partial[(size_t)(k + 2 * FL) * rs + C + c];
            const float s3 = partial[(size_t)(k + 3 * FL) * rs + c], q3 = partial[(size_t)(k + 3 * FL) * rs + C + c];
            s += (double)s0; s += (double)s1; s += (double)s2; s += (double)s3;
            q += (double)q0; q += (double)q1; q += (double)q2; q += (double)q3;
        }
        for (; k < n_split; k += FL) {
            s += (double)partial[(size_t)k * rs + c];
            q += (double)partial[(size_t)k * rs + C + c];
        }
    }
    red[lane][0][threadIdx.x % FC] = s;
    red[lane][1][threadIdx.x % FC] = q;
    __syncthreads();
    a = b = 0.0;
    if (lane == 0)
        for (int l = 0; l < FL; ++l) {
            a += red[l][0][threadIdx.x % FC];
            b += red[l][1][threadIdx.x % FC];
        }
}


// Long tables are folded first: workgroup (channel block, slice) sums `chunk` consecutive rows of its 32 channels and leaves the
// result IN the first row of its slice and zeros in the others (its own rows, its own columns: no other workgroup touches them;
// the table keeps its totals, so finalizing it again is harmless); the finalize kernels
// then combine the slices' first rows (row stride chunk).  One CU loads ~115 GB/s: the 12 544-row tables of 448 x 448 inputs
// took 25 - 50 us per finalize launch through one workgroup per channel block.
__global__ __launch_bounds__(FT) void bn2d_fold_partials_kernel(float* partial, int n_split, int C, int chunk) {
    __shared__ double red[FL][2][FC];
    const int c = blockIdx.x * FC + threadIdx.x % FC, lane = threadIdx.x / FC;
    const int k0 = blockIdx.y * chunk;
    const int n = n_split - k0 < chunk ? n_split - k0 : chunk;
    float* base = partial + (size_t)k0 * 2 * C;
    double a, b;
    combine_partials(base, n, C, c, lane, red, a, b, (size_t)2 * C);       // (ends behind a barrier: every row has been read)
    if (c >= C) return;
    // the slice's other rows become zeros, so that the table still SUMS to the same totals: finalizing it a second time (a
    // re-run, a debugging comparison, a synchronised-statistics combine after a finalize) gives the same result instead of
    // counting every slice twice
    for (int k = 1 + lane; k < n; k += FL) base[(size_t)k * 2 * C + c] = 0.f, base[(size_t)k * 2 * C + C + c] = 0.f;
    if (lane != 0) return;
    base[c] = (float)a;
    base[C + c] = (float)b;
}
constexpr int kFoldRows = 2048, kFoldChunk = 512;       // tables of >= kFoldRows rows are folded in slices of kFoldChunk
inline bool fold_on() { static const int on = getenv("PECLR_BN_FOLD") ? atoi(getenv("PECLR_BN_FOLD")) : 1; return on != 0; }

__global__ __launch_bounds__(FT) void bn2d_stats_finalize_kernel(const float* __restrict__ partial, int n_split, int R, int C,
                                                                 float eps, float momentum, const float* __restrict__ gamma,
                                                                 const float* __restrict__ beta, float* running_mean,
                                                                 float* running_var, int64_t* nbt, float* __restrict__ save_mean,
                                                                 float* __restrict__ save_invstd, float* __restrict__ scale_shift,
                                                                 int fold, int shift_row) {
    __shared__ double red[FL][2][FC];
    const int c = blockIdx.x * FC + threadIdx.x % FC, lane = threadIdx.x / FC;
    if (blockIdx.x == 0 && threadIdx.x == 0 && nbt) *nbt += 1;
    double s, q;
    combine_partials(partial, n_split, C, c, lane, red, s, q, (size_t)fold * 2 * C);      // (n_split slices of `fold` rows each)
    if (lane != 0 || c >= C) return;
    const double k0 = (double)partial[(size_t)shift_row * 2 * C + c];
    const double ms = s / R;                 // mean of (x - k0)
    double var = q / R - ms * ms;            // biased variance
    if (var < 0.0) var = 0.0;
    const float mean = (float)(ms + k0);
    const float invstd = (float)(1.0 / sqrt(var + (double)eps));
    save_mean[c] = mean;
    save_invstd[c] = invstd;
    const float sc = gamma[c] * invstd;
    scale_shift[c] = sc;
    scale_shift[C + c] = beta[c] - mean * sc;
    if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean;
    if (running_var) {
        const float unbiased = (float)(var * ((double)R / (double)(R > 1 ? R - 1 : 1)));
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * unbiased;
    }
}

// ---- synchronised statistics (data parallel): combine -> [all-reduce on the host side] -> finalize
// totals: double [2*C + 1] = per-channel sums, then the row count (set by the host before the all-reduce)
__global__ __launch_bounds__(FT) void bn2d_combine_kernel(const float* __restrict__ partial, int n_split, int C,
                                                          double* __restrict__ totals) {
    __shared__ double red[FL][2][FC];
    const int c = blockIdx.x * FC + threadIdx.x % FC, lane = threadIdx.x / FC;
    double a, b;
    combine_partials(partial, n_split, C, c, lane, red, a, b, (size_t)2 * C);
    if (lane != 0 || c >= C) return;
    totals[c] = a;
    totals[C + c] = b;
}

__global__ __launch_bounds__(T) void bn2d_finalize_totals_kernel(const double* __restrict__ totals, const float* __restrict__ shift,
                                                                 int C, float eps, float momentum,
                                                                 const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                 float* running_mean, float* running_var, int64_t* nbt,
                                                                 float* __restrict__ save_mean, float* __restrict__ save_invstd,
                                                                 float* __restrict__ scale_shift) {
    const int c = blockIdx.x * T + threadIdx.x;
    if (c == 0 && nbt) *nbt += 1;
    if (c >= C) return;
    const double count = totals[2 * C];
    const double ms = totals[c] / count;
    double var = totals[C + c] / count - ms * ms;
    if (var < 0.0) var = 0.0;
    const float mean = (float)(ms + (double)shift[c]);
    const float invstd = (float)(1.0 / sqrt(var + (double)eps));
    save_mean[c] = mean;
    save_invstd[c] = invstd;
    const float sc = gamma[c] * invstd;
    scale_shift[c] = sc;
    scale_shift[C + c] = beta[c] - mean * sc;
    if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean;
    if (running_var) {
        const float unbiased = (float)(var * (count / (count > 1.0 ? count - 1.0 : 1.0)));
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * unbiased;
    }
}

// dgamma / dbeta are this rank's LOCAL sums (parameter gradients are summed across ranks later by the
// gradient all-reduce); the dx coefficients use the GLOBAL sums and the global row count.
__global__ __launch_bounds__(T) void bn2d_bwd_finalize_totals_kernel(const double* __restrict__ local_totals,
                                                                     const double* __restrict__ global_totals,
                                                                     int C, int training, const float* __restrict__ scale_shift,
                                                                     float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                                     float* __restrict__ coef) {
    const int c = blockIdx.x * T + threadIdx.x;
    if (c >= C) return;
    const double count = global_totals[2 * C];
    dbeta[c] = (float)local_totals[c];
    dgamma[c] = (float)local_totals[C + c];
    const double k1 = (double)scale_shift[c];
    coef[c] = training ? (float)(-k1 * global_totals[c] / count) : 0.f;
    coef[C + c] = training ? (float)(-k1 * global_totals[C + c] / count) : 0.f;
}

// eval mode: scale/shift from the running statistics
__global__ __launch_bounds__(T) void bn2d_eval_params_kernel(int C, float eps, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, const float* __restrict__ rm,
                                                             const float* __restrict__ rv, float* __restrict__ save_mean,
                                                             float* __restrict__ save_invstd, float* __restrict__ scale_shift) {
    const int c = blockIdx.x * T + threadIdx.x;
    if (c >= C) return;
    const float invstd = 1.0f / sqrtf(rv[c] + eps);
    save_mean[c] = rm[c];
    save_invstd[c] = invstd;
    const float sc = gamma[c] * invstd;
    scale_shift[c] = sc;
    scale_shift[C + c] = beta[c] - rm[c] * sc;
}

// ------------------------------------------------------------------ forward: apply
// ReLU bit mask, 1 bit per element, layout [R][C/32] uint32 (bit c%32 of word (r, c/32)): independent
// of the launch geometry.  The 32/W lanes that own one word are adjacent lanes of one wave in the same
// row (column group is the fastest thread index and CGB is a multiple of 32/W).
template <int W>
__device__ __forceinline__ void mask_store(unsigned* __restrict__ mask, int row, int col, int C, unsigned bits) {
    constexpr int LPW = 32 / W;  // lanes per word
    unsigned word = bits << ((threadIdx.x % LPW) * W);
#pragma unroll
    for (int o = 1; o < LPW; o <<= 1) word |= __shfl_xor(word, o, kWave);
    if (threadIdx.x % LPW == 0) mask[(size_t)row * (C / 32) + col / 32] = word;
}
template <int W>
__device__ __forceinline__ unsigned mask_load(const unsigned* __restrict__ mask, int row, int col, int C) {
    constexpr int LPW = 32 / W;
    return (mask[(size_t)row * (C / 32) + col / 32] >> ((threadIdx.x % LPW) * W)) & ((1u << W) - 1u);
}

// absmax (nullable): the pass also leaves max |output| over everything it wrote in *absmax (one atomic per wave onto a slot its
// caller zeroed: non-negative floats order like their bit patterns, and a maximum does not depend on the order) -- what the "pair"
// GEMMs that consume the tensor derive its power of two from (include/peclr_hip.h peclr_x6_pair); a NaN output is not seen here
// (fmaxf drops it) and still poisons the products it enters, as it would in fp32
__device__ __forceinline__ void absmax_commit(float* absmax, float m) {
    if (!absmax) return;
    m = wave_max(m);
    if ((threadIdx.x & (kWave - 1)) == 0) {
        // thousands of waves, one address: look first (a relaxed atomic load, served by the L2 where the atomics execute) and
        // only send the atomic when this wave raises the maximum -- a handful of times per launch instead of once per wave
        // (unconditional atomics cost the streaming passes 15 - 20 us each: serialised at one L2 channel)
        unsigned* p = reinterpret_cast<unsigned*>(absmax);
        const unsigned mine = __float_as_uint(m);
        if (mine > __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(p, mine);
    }
}

// RES: 0 no residual; 1 a residual tensor; 2 `res` is the INPUT of the shortcut's BatchNorm2d (the downsample branch of a
// layer's first block: conv1x1 -> bn, torchvision Bottleneck.downsample behind resnet_model.py:15) and res_ss its [2][C] scale /
// shift: the residual is fmaf(res, scale, shift) rounded to the storage format -- the value peclr_bn2d_apply would have written
// and this pass read back -- so that layer's apply pass and its output tensor disappear
template <typename IO, int RES, bool RELU>
__global__ __launch_bounds__(T) void bn2d_apply_kernel(const IO* __restrict__ x, const IO* __restrict__ res, Geo g,
                                                       const float* __restrict__ scale_shift, IO* __restrict__ y,
                                                       unsigned* __restrict__ relu_mask, const float* __restrict__ res_ss,
                                                       float* __restrict__ absmax) {
    constexpr int W = Word<IO>::W, U = Word<IO>::U;
    int col, r0, r1, rl;
    thread_geo<W>(g, col, r0, r1, rl);
    const Fv<W> sc = loadp<W>(scale_shift + col), sh = loadp<W>(scale_shift + g.C + col);
    float amax = 0.f;
    Fv<W> rsc = sc, rsh = sh;
    if (RES == 2) { rsc = loadp<W>(res_ss + col); rsh = loadp<W>(res_ss + g.C + col); }
    auto emit = [&](int row, const Fv<W>& v, const Fv<W>& w) {
        const size_t o = (size_t)row * g.C + col;
        Fv<W> t;
        unsigned bits = 0;
#pragma unroll
        for (int k = 0; k < W; ++k) {
            float a = fmaf(v.v[k], sc.v[k], sh.v[k]);
            if (RES == 1) a += w.v[k];
            if (RES == 2) a += Word<IO>::round(fmaf(w.v[k], rsc.v[k], rsh.v[k]));
            if (RELU) bits |= (a > 0.f ? 1u : 0u) << k;
            t.v[k] = RELU ? fmaxf(a, 0.f) : a;
            if constexpr (sizeof(IO) == 4) amax = fmaxf(amax, fabsf(t.v[k]));      // (fp32 tensors only: the pair GEMMs' operands)
        }
        Word<IO>::store(y + o, t);
        if (RELU && relu_mask) mask_store<W>(relu_mask, row, col, g.C, bits);  // the word's lanes share `row`
    };
    int r = r0 + rl;
    for (; r + (U - 1) * g.RPP < r1; r += U * g.RPP) {
        Fv<W> v[U], w[RES ? U : 1];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t o = (size_t)(r + u * g.RPP) * g.C + col;
            v[u] = Word<IO>::load(x + o);
            if (RES) w[u] = Word<IO>::load(res + o);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) emit(r + u * g.RPP, v[u], RES ? w[RES ? u : 0] : v[u]);
    }
    for (; r < r1; r += g.RPP) {
        const size_t o = (size_t)r * g.C + col;
        const Fv<W> v = Word<IO>::load(x + o);
        emit(r, v, RES ? Word<IO>::load(res + o) : v);
    }
    if constexpr (sizeof(IO) == 4) absmax_commit(absmax, amax);
}

// Last block of layer4: BatchNorm2d + identity + ReLU + AdaptiveAvgPool2d((1, 1)) + flatten in one pass
// (resnet_model.py:24-26 features[7][-1] .. features[8], then `.flatten(1)`): the [N, HW, C] activation is
// never written -- no later kernel reads it (the pool's backward is a broadcast, this BatchNorm's backward
// reads x and the 1-bit mask) -- only its per-image channel means, as the fp32 [N, C] encoder output the
// projection head's first GEMM consumes.  grid = (column blocks, images); the block's row lanes walk the
// image's HW rows.
template <typename IO>
__global__ __launch_bounds__(T) void bn2d_apply_avgpool_kernel(const IO* __restrict__ x, const IO* __restrict__ res, Geo g,
                                                               const float* __restrict__ scale_shift, float inv_hw,
                                                               float* __restrict__ pooled, unsigned* __restrict__ relu_mask) {
    constexpr int W = Word<IO>::W, U = Word<IO>::U;
    __shared__ float red[W * T];
    int col, r0, r1, rl;
    thread_geo<W>(g, col, r0, r1, rl);   // rows_per_block = HW: [r0, r1) = this image's rows
    const Fv<W> sc = loadp<W>(scale_shift + col), sh = loadp<W>(scale_shift + g.C + col);
    Fv<W> sum = zero<W>();
    auto emit = [&](int row, const Fv<W>& v, const Fv<W>& w) {
        unsigned bits = 0;
#pragma unroll
        for (int k = 0; k < W; ++k) {
            const float a = fmaf(v.v[k], sc.v[k], sh.v[k]) + w.v[k];
            bits |= (a > 0.f ? 1u : 0u) << k;
            sum.v[k] += fmaxf(a, 0.f);
        }
        mask_store<W>(relu_mask, row, col, g.C, bits);
    };
    int r = r0 + rl;
    for (; r + (U - 1) * g.RPP < r1; r += U * g.RPP) {
        Fv<W> v[U], w[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t o = (size_t)(r + u * g.RPP) * g.C + col;
            v[u] = Word<IO>::load(x + o);
            w[u] = Word<IO>::load(res + o);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) emit(r + u * g.RPP, v[u], w[u]);
    }
    for (; r < r1; r += g.RPP) {
        const size_t o = (size_t)r * g.C + col;
        emit(r, Word<IO>::load(x + o), Word<IO>::load(res + o));
    }
    sum = lane_reduce<W>(sum, red, g, rl);
    if (rl == 0) {
#pragma unroll
        for (int k = 0; k < W; ++k) sum.v[k] *= inv_hw;
        storep<W>(pooled + (size_t)slice_y(g) * g.C + col, sum);
    }
}

// ------------------------------------------------------------------ backward
// MASK: 0 = no ReLU, 1 = ReLU mask recomputed from x (no residual), 2 = ReLU mask read from y,
//       3 = ReLU mask read from the 1-bit-per-element mask the forward wrote.
template <int W, int MASK>
__device__ __forceinline__ Fv<W> masked(const Fv<W>& d, const Fv<W>& xv, const Fv<W>& yv, unsigned bits, const Fv<W>& sc,
                                        const Fv<W>& sh) {
    Fv<W> r;
#pragma unroll
    for (int k = 0; k < W; ++k) {
        bool on = true;
        if (MASK == 1) on = fmaf(xv.v[k], sc.v[k], sh.v[k]) > 0.f;
        if (MASK == 2) on = yv.v[k] > 0.f;
        if (MASK == 3) on = (bits >> k) & 1u;
        r.v[k] = on ? d.v[k] : 0.f;
    }
    return r;
}

// POOL: the incoming gradient is that of the fused average pool: dy[row] = d_pooled[row / hw] / hw, an
// fp32 [N, C] matrix broadcast over the image's rows (no [R, C] gradient tensor exists).
template <int W>
__device__ __forceinline__ Fv<W> pooled_dy(const float* __restrict__ d_pooled, int row, int hw, float inv_hw, int C, int col) {
    Fv<W> d = loadp<W>(d_pooled + (size_t)(row / hw) * C + col);
#pragma unroll
    for (int k = 0; k < W; ++k) d.v[k] *= inv_hw;
    return d;
}

template <typename IO, int MASK, bool POOL = false>
__global__ __launch_bounds__(T) void bn2d_bwd_reduce_kernel(const IO* __restrict__ dy, const IO* __restrict__ x,
                                                            const IO* __restrict__ y, const unsigned* __restrict__ relu_mask, Geo g,
                                                            const float* __restrict__ save_mean, const float* __restrict__ save_invstd,
                                                            const float* __restrict__ scale_shift, float* __restrict__ partial,
                                                            const float* __restrict__ d_pooled = nullptr, int hw = 1, float inv_hw = 1.f) {
    constexpr int W = Word<IO>::W, U = Word<IO>::U;
    __shared__ float red[W * T];
    int col, r0, r1, rl;
    thread_geo<W>(g, col, r0, r1, rl);
    auto load_dy = [&](int row) -> Fv<W> {
        if constexpr (POOL) return pooled_dy<W>(d_pooled, row, hw, inv_hw, g.C, col);
        else return Word<IO>::load(dy + (size_t)row * g.C + col);
    };
    const Fv<W> mean = loadp<W>(save_mean + col), invstd = loadp<W>(save_invstd + col);
    const Fv<W> sc = loadp<W>(scale_shift + col), sh = loadp<W>(scale_shift + g.C + col);
    Fv<W> sb = zero<W>(), sg = zero<W>();
    auto acc = [&](const Fv<W>& d0, const Fv<W>& xv, const Fv<W>& yv, unsigned bits) {
        const Fv<W> d = masked<W, MASK>(d0, xv, yv, bits, sc, sh);
#pragma unroll
        for (int k = 0; k < W; ++k) {
            sb.v[k] += d.v[k];
            sg.v[k] = fmaf(d.v[k], (xv.v[k] - mean.v[k]) * invstd.v[k], sg.v[k]);
        }
    };
    int r = r0 + rl;
    for (; r + (U - 1) * g.RPP < r1; r += U * g.RPP) {
        Fv<W> d[U], xv[U], yv[MASK == 2 ? U : 1];
        unsigned mb[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t o = (size_t)(r + u * g.RPP) * g.C + col;
            d[u] = load_dy(r + u * g.RPP);
            xv[u] = Word<IO>::load(x + o);
            if (MASK == 2) yv[u] = Word<IO>::load(y + o);
            mb[u] = MASK == 3 ? mask_load<W>(relu_mask, r + u * g.RPP, col, g.C) : 0u;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) acc(d[u], xv[u], MASK == 2 ? yv[MASK == 2 ? u : 0] : xv[u], mb[u]);
    }
    for (; r < r1; r += g.RPP) {
        const size_t o = (size_t)r * g.C + col;
        const Fv<W> xv = Word<IO>::load(x + o);
        acc(load_dy(r), xv, MASK == 2 ? Word<IO>::load(y + o) : xv,
            MASK == 3 ? mask_load<W>(relu_mask, r, col, g.C) : 0u);
    }
    sb = lane_reduce<W>(sb, red, g, rl);
    sg = lane_reduce<W>(sg, red, g, rl);
    if (rl == 0) {
        float* o = partial + (size_t)slice_y(g) * 2 * g.C;
        storep<W>(o + col, sb);
        storep<W>(o + g.C + col, sg);
    }
}

// dbeta, dgamma, and the two per-channel coefficients of the dx pass:
//   training: dx = k1*dy' + (k2 + k3*xhat)  with k1 = gamma*invstd, k2 = -k1*dbeta/R, k3 = -k1*dgamma/R
//   eval    : dx = k1*dy'
__global__ __launch_bounds__(FT) void bn2d_bwd_finalize_kernel(const float* __restrict__ partial, int n_split, int R, int C,
                                                               int training, const float* __restrict__ scale_shift,
                                                               float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                               float* __restrict__ coef, int fold) {
    __shared__ double red[FL][2][FC];
    const int c = blockIdx.x * FC + threadIdx.x % FC, lane = threadIdx.x / FC;
    double sb, sg;
    combine_partials(partial, n_split, C, c, lane, red, sb, sg, (size_t)fold * 2 * C);
    if (lane != 0 || c >= C) return;
    dbeta[c] = (float)sb;
    dgamma[c] = (float)sg;
    const double k1 = (double)scale_shift[c];
    coef[c] = training ? (float)(-k1 * sb / R) : 0.f;
    coef[C + c] = training ? (float)(-k1 * sg / R) : 0.f;
}

template <typename IO, int MASK, bool DRES, bool POOL = false>
__global__ __launch_bounds__(T) void bn2d_bwd_apply_kernel(const IO* __restrict__ dy, const IO* __restrict__ x,
                                                           const IO* __restrict__ y, const unsigned* __restrict__ relu_mask, Geo g,
                                                           const float* __restrict__ save_mean, const float* __restrict__ save_invstd,
                                                           const float* __restrict__ scale_shift, const float* __restrict__ coef,
                                                           IO* __restrict__ dx, IO* __restrict__ dres,
                                                           const float* __restrict__ d_pooled = nullptr, int hw = 1, float inv_hw = 1.f,
                                                           float* __restrict__ absmax = nullptr) {
    constexpr int W = Word<IO>::W, U = Word<IO>::U;
    int col, r0, r1, rl;
    thread_geo<W>(g, col, r0, r1, rl);
    float amax = 0.f;
    auto load_dy = [&](int row) -> Fv<W> {
        if constexpr (POOL) return pooled_dy<W>(d_pooled, row, hw, inv_hw, g.C, col);
        else return Word<IO>::load(dy + (size_t)row * g.C + col);
    };
    // dx = d' scale + xhat k3 + k2 with xhat = (x - mean) invstd.  16-bit activations (eight channels per thread: the per-channel
    // constants alone were 48 registers, the kernel ran at two waves per SIMD): folded into dx = d' scale + x a + b with
    // a = invstd k3, b = k2 - mean a -- the extra fp32 rounding sits five orders below the 16-bit rounding of dx.  fp32: the
    // unfolded form (cancellation-free around x = mean), as before.
    constexpr bool FOLD = sizeof(IO) == 2;
    Fv<W> mean = loadp<W>(save_mean + col), invstd = loadp<W>(save_invstd + col);
    const Fv<W> sc = loadp<W>(scale_shift + col), sh = loadp<W>(scale_shift + g.C + col);
    Fv<W> k2 = loadp<W>(coef + col), k3 = loadp<W>(coef + g.C + col);
    if constexpr (FOLD) {
#pragma unroll
        for (int k = 0; k < W; ++k) {
            k3.v[k] *= invstd.v[k];                    // a
            k2.v[k] = fmaf(-mean.v[k], k3.v[k], k2.v[k]);   // b
        }
    }
    auto emit = [&](size_t o, const Fv<W>& d0, const Fv<W>& xv, const Fv<W>& yv, unsigned bits) {
        const Fv<W> d = masked<W, MASK>(d0, xv, yv, bits, sc, sh);
        if (DRES) Word<IO>::store(dres + o, d);
        Fv<W> t;
#pragma unroll
        for (int k = 0; k < W; ++k) {
            if constexpr (FOLD) t.v[k] = fmaf(d.v[k], sc.v[k], fmaf(xv.v[k], k3.v[k], k2.v[k]));
            else {
                const float xh = (xv.v[k] - mean.v[k]) * invstd.v[k];
                t.v[k] = fmaf(d.v[k], sc.v[k], fmaf(xh, k3.v[k], k2.v[k]));
            }
            if constexpr (sizeof(IO) == 4) amax = fmaxf(amax, fabsf(t.v[k]));
        }
        Word<IO>::store(dx + o, t);
    };
    typedef typename Word<IO>::Raw Raw;
    int r = r0 + rl;
    for (; r + (U - 1) * g.RPP < r1; r += U * g.RPP) {
        // the words stay raw (4 registers each) while in flight and are widened one row at a time
        Fv<W> dp[POOL ? U : 1];
        Raw d[POOL ? 1 : U], xv[U], yv[MASK == 2 ? U : 1];
        unsigned mb[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t o = (size_t)(r + u * g.RPP) * g.C + col;
            if constexpr (POOL) dp[u] = load_dy(r + u * g.RPP);
            else d[u] = Word<IO>::load_raw(dy + o);
            xv[u] = Word<IO>::load_raw(x + o);
            if (MASK == 2) yv[u] = Word<IO>::load_raw(y + o);
            mb[u] = MASK == 3 ? mask_load<W>(relu_mask, r + u * g.RPP, col, g.C) : 0u;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const Fv<W> xe = Word<IO>::expand(xv[u]);
            emit((size_t)(r + u * g.RPP) * g.C + col, POOL ? dp[POOL ? u : 0] : Word<IO>::expand(d[POOL ? 0 : u]), xe,
                 MASK == 2 ? Word<IO>::expand(yv[MASK == 2 ? u : 0]) : xe, mb[u]);
        }
    }
    for (; r < r1; r += g.RPP) {
        const size_t o = (size_t)r * g.C + col;
        const Fv<W> xv = Word<IO>::load(x + o);
        emit(o, load_dy(r), xv, MASK == 2 ? Word<IO>::load(y + o) : xv,
             MASK == 3 ? mask_load<W>(relu_mask, r, col, g.C) : 0u);
    }
    if constexpr (sizeof(IO) == 4) absmax_commit(absmax, amax);
}

// ------------------------------------------------------------------ host

// ------------------------------------------------------------------ stem: BN + ReLU + 3x3/2 max-pool in one pass
// torchvision's stem is conv7x7/2 -> BatchNorm2d -> ReLU -> MaxPool2d(3, stride 2, padding 1)
// (resnet_model.py:15 wraps it as features[0..3]).  Normalising, rectifying and pooling in ONE pass never
// materialises the 112x112 activation (822 MB at 2x128 views): forward reads x (L2 absorbs the 2.25x
// window overlap) and writes the pooled tensor plus a 1-byte tap code per pooled element; backward
// rebuilds the sparse pre-pool gradient from the codes on the fly, in the reduce pass (over pooled
// elements) and in the apply pass (every pre-pool element gathers from the <= 4 windows that hold it).
// Tap code = 3*dh + dw of the FIRST maximum in row-major window order (strict >), torch's rule.
struct PoolGeo {
    int N, H, Wd, C, PH, PW, CW, PPB;  // CW = C / W column groups, PPB = 256 / CW pixels per block pass
    int rev;                           // walk the tensor back to front (as Geo::rev: these tensors are 411 - 1 644 MB)
};
// Workgroups are handed to the 8 XCDs round-robin and every XCD has its own L2.  The 3x3/2 windows of
// neighbouring pixels overlap, so the pixels of ONE image should meet in ONE L2: hardware block b works on
// image n = 8 * (j / bpi) + b % 8 and on run j % bpi of that image's pixels, with j = b / 8 and bpi = runs
// per image.  At any moment the 8 XCDs walk 8 consecutive images in step (a few MB apart, so their streams
// spread over the HBM channels; handing each XCD a contiguous EIGHTH of the tensor instead put all eight
// streams on the same channels and ran 20 % slower).  Images past N idle.
constexpr int kXcd = 8;
struct XcdSlot {
    int n, run;
};
__device__ __forceinline__ XcdSlot xcd_slot(int runs_per_image, int rev) {
    int j = blockIdx.x / kXcd;
    if (rev) j = (int)(gridDim.x / kXcd) - 1 - j;        // (same XCD, the image groups and runs in reverse order)
    return {kXcd * (j / runs_per_image) + (int)(blockIdx.x % kXcd), j % runs_per_image};
}

constexpr int kPoolIter = 4;  // pixels per thread in the two apply kernels (fewer, longer workgroups)

template <typename IO>
__global__ __launch_bounds__(T) void bn2d_pool_apply_kernel(const IO* __restrict__ x, PoolGeo g,
                                                            const float* __restrict__ scale_shift, IO* __restrict__ y,
                                                            IO* __restrict__ x_at_max, uint8_t* __restrict__ code,
                                                            float* __restrict__ absmax) {
    constexpr int W = Word<IO>::W;
    float amax = 0.f;
    const int cg = threadIdx.x % g.CW, pl = threadIdx.x / g.CW, col = cg * W;
    const Fv<W> sc = loadp<W>(scale_shift + col), sh = loadp<W>(scale_shift + g.C + col);
    const int per_image = g.PH * g.PW, span = g.PPB * kPoolIter;
    const XcdSlot slot = xcd_slot((per_image + span - 1) / span, g.rev);
    if (slot.n >= g.N) return;
    const int n = slot.n;
    for (int it = 0; it < kPoolIter; ++it) {
        const int q = slot.run * span + it * g.PPB + pl;
        if (q >= per_image) break;
        const int pw = q % g.PW, ph = q / g.PW;
        const long long p = (long long)n * per_image + q;
        Fv<W> best, xb;
        unsigned char bc[W];
        bool first = true;
#pragma unroll
        for (int dh = 0; dh < 3; ++dh) {
            const int h = 2 * ph - 1 + dh;
            if (h < 0 || h >= g.H) continue;
#pragma unroll
            for (int dw = 0; dw < 3; ++dw) {
                const int w = 2 * pw - 1 + dw;
                if (w < 0 || w >= g.Wd) continue;
                const Fv<W> v = Word<IO>::load_cached(x + (((size_t)n * g.H + h) * g.Wd + w) * g.C + col);   // (nine windows share a pixel)
#pragma unroll
                for (int k = 0; k < W; ++k) {
                    const float t = fmaxf(fmaf(v.v[k], sc.v[k], sh.v[k]), 0.f);
                    if (first || t > best.v[k]) best.v[k] = t, xb.v[k] = v.v[k], bc[k] = (unsigned char)(3 * dh + dw);
                }
                first = false;
            }
        }
        Word<IO>::store(y + (size_t)p * g.C + col, best);
#pragma unroll
        for (int k = 0; k < W; ++k)
            if constexpr (sizeof(IO) == 4) amax = fmaxf(amax, best.v[k]);    // (rectified: non-negative)
        Word<IO>::store(x_at_max + (size_t)p * g.C + col, xb);   // exact: x is an IO value already
        unsigned packed[W / 4];
#pragma unroll
        for (int k = 0; k < W; k += 4) packed[k / 4] = bc[k] | (bc[k + 1] << 8) | (bc[k + 2] << 16) | ((unsigned)bc[k + 3] << 24);
#pragma unroll
        for (int k = 0; k < W / 4; ++k) reinterpret_cast<unsigned*>(code + (size_t)p * g.C + col)[k] = packed[k];
    }
    if constexpr (sizeof(IO) == 4) absmax_commit(absmax, amax);
}

// partial: [gridDim.x][2][C] = (sum of masked dy, sum of masked dy * xhat) over the block's pooled pixels
template <typename IO>
__global__ __launch_bounds__(T) void bn2d_pool_bwd_reduce_kernel(const IO* __restrict__ dyp, const IO* __restrict__ x_at_max,
                                                                 PoolGeo g, const float* __restrict__ save_mean,
                                                                 const float* __restrict__ save_invstd,
                                                                 const float* __restrict__ scale_shift,
                                                                 float* __restrict__ partial) {
    constexpr int W = Word<IO>::W;
    __shared__ float red[2 * W * T];
    const int cg = threadIdx.x % g.CW, pl = threadIdx.x / g.CW, col = cg * W;
    const Fv<W> sc = loadp<W>(scale_shift + col), sh = loadp<W>(scale_shift + g.C + col);
    const Fv<W> mean = loadp<W>(save_mean + col), inv = loadp<W>(save_invstd + col);
    Fv<W> s = zero<W>(), q = zero<W>();
    // every pooled element's gradient lands on ONE un-pooled element, whose x the forward kept: a plain
    // streaming reduction over the pooled grid.  gridDim.x blocks, contiguous chunks, 4 loads in flight.
    const long long P = (long long)g.N * g.PH * g.PW;
    const long long chunk = ((P + gridDim.x - 1) / gridDim.x + g.PPB - 1) / g.PPB * g.PPB;
    const long long lb = g.rev ? (long long)gridDim.x - 1 - blockIdx.x : (long long)blockIdx.x;       // logical block (slot of its sums)
    const long long p_begin = lb * chunk, p_end = p_begin + chunk < P ? p_begin + chunk : P;
    auto acc = [&](const Fv<W>& gy, const Fv<W>& xv) {
#pragma unroll
        for (int k = 0; k < W; ++k) {
            const float d = fmaf(xv.v[k], sc.v[k], sh.v[k]) > 0.f ? gy.v[k] : 0.f;
            s.v[k] += d;
            q.v[k] += d * ((xv.v[k] - mean.v[k]) * inv.v[k]);
        }
    };
    long long p = p_begin + pl;
    for (; p + 3 * g.PPB < p_end; p += 4 * g.PPB) {
        Fv<W> gy[4], xv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            gy[u] = Word<IO>::load(dyp + (size_t)(p + u * g.PPB) * g.C + col);
            xv[u] = Word<IO>::load(x_at_max + (size_t)(p + u * g.PPB) * g.C + col);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) acc(gy[u], xv[u]);
    }
    for (; p < p_end; p += g.PPB)
        acc(Word<IO>::load(dyp + (size_t)p * g.C + col), Word<IO>::load(x_at_max + (size_t)p * g.C + col));
    // block reduction over the pixel lanes that share a column group (fixed order)
#pragma unroll
    for (int k = 0; k < W; ++k) red[k * T + threadIdx.x] = s.v[k], red[(W + k) * T + threadIdx.x] = q.v[k];
    __syncthreads();
    if (pl == 0) {
        Fv<W> ts = zero<W>(), tq = zero<W>();
        for (int l = 0; l < g.PPB; ++l)
#pragma unroll
            for (int k = 0; k < W; ++k) ts.v[k] += red[k * T + l * g.CW + cg], tq.v[k] += red[(W + k) * T + l * g.CW + cg];
        storep<W>(partial + (size_t)lb * 2 * g.C + col, ts);      // slot order is irrelevant to the
        storep<W>(partial + (size_t)lb * 2 * g.C + g.C + col, tq);  // fixed-order combine that follows
    }
}

template <typename IO>
__global__ __launch_bounds__(T) void bn2d_pool_bwd_apply_kernel(const IO* __restrict__ dyp, const IO* __restrict__ x,
                                                                const uint8_t* __restrict__ code, PoolGeo g,
                                                                const float* __restrict__ save_mean,
                                                                const float* __restrict__ save_invstd,
                                                                const float* __restrict__ scale_shift,
                                                                const float* __restrict__ coef, IO* __restrict__ dx) {
    // four channels per lane for every storage type (`Quad`): with the 16-bit types' eight, the four candidate windows in flight
    // cost 130 VGPRs (3 waves per SIMD; 412 us at 2.4 TB/s against the fp32 instance's 344 us at 5.9 TB/s for twice the bytes)
    constexpr int W = 4;
    const int cg = threadIdx.x % g.CW, pl = threadIdx.x / g.CW, col = cg * W;
    const Fv<W> sc = loadp<W>(scale_shift + col), sh = loadp<W>(scale_shift + g.C + col);
    const Fv<W> mean = loadp<W>(save_mean + col), inv = loadp<W>(save_invstd + col);
    const Fv<W> c0 = loadp<W>(coef + col), c1 = loadp<W>(coef + g.C + col);
    const int per_image = g.H * g.Wd, span = g.PPB * 2 * kPoolIter;
    const XcdSlot slot = xcd_slot((per_image + span - 1) / span, g.rev);
    if (slot.n >= g.N) return;
    const int n = slot.n;
    for (int it = 0; it < 2 * kPoolIter; ++it) {
        const int q = slot.run * span + it * g.PPB + pl;
        if (q >= per_image) break;
        const int w = q % g.Wd, h = q / g.Wd;
        const long long r = (long long)n * per_image + q;
        const Fv<W> v = Quad<IO>::load(x + (size_t)r * g.C + col);
        Fv<W> d = zero<W>();
        // windows holding (h, w): ph in {h/2, (h+1)/2}, pw in {w/2, (w+1)/2} (one or two each); all four
        // candidates are loaded together (predicated) so the loads overlap
        const int phs[2] = {h >> 1, (h + 1) >> 1}, pws[2] = {w >> 1, (w + 1) >> 1};
        Fv<W> gy[4];
        unsigned cw[4][W / 4];
        bool on[4];
        unsigned char mine[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int ph = phs[i >> 1], pw = pws[i & 1];
            on[i] = ph < g.PH && pw < g.PW && !((i >> 1) && phs[1] == phs[0]) && !((i & 1) && pws[1] == pws[0]);
            mine[i] = (unsigned char)(3 * (h - (2 * ph - 1)) + (w - (2 * pw - 1)));
            const size_t p = on[i] ? (((size_t)n * g.PH + ph) * g.PW + pw) * g.C + col : 0;
            gy[i] = on[i] ? Quad<IO>::load(dyp + p) : zero<W>();
#pragma unroll
            for (int k = 0; k < W / 4; ++k) cw[i][k] = on[i] ? reinterpret_cast<const unsigned*>(code + p)[k] : 0xFFFFFFFFu;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int k = 0; k < W / 4; ++k)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (((cw[i][k] >> (8 * j)) & 255u) == mine[i]) d.v[4 * k + j] += gy[i].v[4 * k + j];
        Fv<W> o;
#pragma unroll
        for (int k = 0; k < W; ++k) {
            const float dy = fmaf(v.v[k], sc.v[k], sh.v[k]) > 0.f ? d.v[k] : 0.f;
            o.v[k] = fmaf(dy, sc.v[k], fmaf((v.v[k] - mean.v[k]) * inv.v[k], c1.v[k], c0.v[k]));
        }
        Quad<IO>::store(dx + (size_t)r * g.C + col, o);
    }
}

// 16-bit storage, EIGHT channels per lane (16-byte accesses: the four-channel form above issues as many memory instructions per
// pixel as the fp32 instance does and takes as long for half the bytes -- 345 against 355 us at 256 x 112 x 112 x 64).  What kept that
// form from working before was registers (130 VGPRs with everything expanded to fp32): here the four candidate windows stay
// packed until they are added (4 registers each), and the six per-channel constants live in LDS, fetched through an index the
// compiler cannot hoist.  C <= 512.
template <typename IO>
__global__ __launch_bounds__(T) void bn2d_pool_bwd_apply16_kernel(const IO* __restrict__ dyp, const IO* __restrict__ x,
                                                                  const uint8_t* __restrict__ code, PoolGeo g,
                                                                  const float* __restrict__ save_mean,
                                                                  const float* __restrict__ save_invstd,
                                                                  const float* __restrict__ scale_shift,
                                                                  const float* __restrict__ coef, IO* __restrict__ dx) {
    constexpr int W = 8;
    typedef typename Word<IO>::Raw Raw;
    __shared__ __attribute__((aligned(16))) float cst[6 * 512];     // [scale | shift | mean | invstd | c0 | c1][C]
    for (int c = threadIdx.x; c < g.C; c += T) {
        cst[c] = scale_shift[c]; cst[g.C + c] = scale_shift[g.C + c];
        cst[2 * g.C + c] = save_mean[c]; cst[3 * g.C + c] = save_invstd[c];
        cst[4 * g.C + c] = coef[c]; cst[5 * g.C + c] = coef[g.C + c];
    }
    __syncthreads();
    const int cg = threadIdx.x % g.CW, pl = threadIdx.x / g.CW, col = cg * W;
    const int per_image = g.H * g.Wd, span = g.PPB * 2 * kPoolIter;
    const XcdSlot slot = xcd_slot((per_image + span - 1) / span, g.rev);
    if (slot.n >= g.N) return;
    const int n = slot.n;
    for (int it = 0; it < 2 * kPoolIter; ++it) {
        const int q = slot.run * span + it * g.PPB + pl;
        if (q >= per_image) break;
        const int w = q % g.Wd, h = q / g.Wd;
        const long long r = (long long)n * per_image + q;
        const u32x4_t xr = ld128_cached(x + (size_t)r * g.C + col);
        const int phs[2] = {h >> 1, (h + 1) >> 1}, pws[2] = {w >> 1, (w + 1) >> 1};
        u32x4_t gr[4];
        unsigned cw[4][2];
        unsigned mine[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int ph = phs[i >> 1], pw = pws[i & 1];
            const bool on = ph < g.PH && pw < g.PW && !((i >> 1) && phs[1] == phs[0]) && !((i & 1) && pws[1] == pws[0]);
            mine[i] = (unsigned)(3 * (h - (2 * ph - 1)) + (w - (2 * pw - 1)));
            const size_t p = on ? (((size_t)n * g.PH + ph) * g.PW + pw) * g.C + col : 0;
            const u32x4_t z = {0u, 0u, 0u, 0u};
            gr[i] = on ? ld128_cached(dyp + p) : z;
            const uint2 cc = on ? *reinterpret_cast<const uint2*>(code + p) : make_uint2(0xFFFFFFFFu, 0xFFFFFFFFu);
            cw[i][0] = cc.x; cw[i][1] = cc.y;
        }
        Fv<W> d = zero<W>();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const Fv<W> gy = Word<IO>::expand(__builtin_bit_cast(Raw, gr[i]));
#pragma unroll
            for (int k = 0; k < W; ++k)
                if (((cw[i][k >> 2] >> (8 * (k & 3))) & 255u) == mine[i]) d.v[k] += gy.v[k];
        }
        const Fv<W> v = Word<IO>::expand(__builtin_bit_cast(Raw, xr));
        int colo = col;
        asm volatile("" : "+v"(colo));                    // (the constants are re-read per pixel: 48 registers they do not occupy)
        const float* cp = cst + colo;
        Fv<W> o;
#pragma unroll
        for (int k4 = 0; k4 < W; k4 += 4) {
            const float4 sc = *reinterpret_cast<const float4*>(cp + k4), sh = *reinterpret_cast<const float4*>(cp + g.C + k4);
            const float4 mean = *reinterpret_cast<const float4*>(cp + 2 * g.C + k4), inv = *reinterpret_cast<const float4*>(cp + 3 * g.C + k4);
            const float4 c0 = *reinterpret_cast<const float4*>(cp + 4 * g.C + k4), c1 = *reinterpret_cast<const float4*>(cp + 5 * g.C + k4);
            const float scv[4] = {sc.x, sc.y, sc.z, sc.w}, shv[4] = {sh.x, sh.y, sh.z, sh.w}, mv[4] = {mean.x, mean.y, mean.z, mean.w};
            const float iv[4] = {inv.x, inv.y, inv.z, inv.w}, c0v[4] = {c0.x, c0.y, c0.z, c0.w}, c1v[4] = {c1.x, c1.y, c1.z, c1.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int k = k4 + j;
                const float dy = fmaf(v.v[k], scv[j], shv[j]) > 0.f ? d.v[k] : 0.f;
                o.v[k] = fmaf(dy, scv[j], fmaf((v.v[k] - mv[j]) * iv[j], c1v[j], c0v[j]));
            }
        }
        Word<IO>::store(dx + (size_t)r * g.C + col, o);
    }
}

struct Plan {
    Geo g;
    dim3 grid;
    int n_split;
};
inline bool make_plan(int R, int C, int W, int U, int want_split, Plan& p) {
    if (R <= 0 || C <= 0 || C % W) return false;
    const int cw = C / W;
    const int cgb = cw < T ? cw : T;
    if (T % cgb || cw % cgb) return false;  // C/W a divisor of 256 or a multiple of it (every ResNet width)
    const int rpp = T / cgb;
    const int ncb = cw / cgb;
    int split, rows;
    if (want_split > 0) {  // caller-fixed (partials buffer already sized): trailing blocks may be empty
        split = want_split;
        rows = ((R + split - 1) / split + rpp - 1) / rpp * rpp;
    } else {               // aim for ~1024 workgroups (4 per CU), at least one unrolled pass per block
        split = 1024 / ncb;
        const int max_split = (R + rpp * U - 1) / (rpp * U);
        if (split > max_split) split = max_split;
        if (split < 1) split = 1;
        rows = ((R + split - 1) / split + rpp - 1) / rpp * rpp;
        split = (R + rows - 1) / rows;
    }
    static const int rev = getenv("PECLR_BN_REVERSE") ? atoi(getenv("PECLR_BN_REVERSE")) != 0 : 1;     // (0: A/B runs)
    p.g = {R, C, cgb, rpp, rows, rev};
    p.grid = dim3(ncb, split);
    p.n_split = split;
    return true;
}
inline bool plan_for(int io_dtype, int R, int C, int want_split, Plan& p) {
    if (io_dtype == PECLR_DTYPE_F32) return make_plan(R, C, Word<float>::W, Word<float>::U, want_split, p);
    if (io_dtype == PECLR_DTYPE_BF16) return make_plan(R, C, Word<bf16_t>::W, Word<bf16_t>::U, want_split, p);
    if (io_dtype == PECLR_DTYPE_F16) return make_plan(R, C, Word<f16_t>::W, Word<f16_t>::U, want_split, p);
    return false;
}
// Runs the statement(s) with `IO` = the activation type of `io` (callers have validated `io` through plan_for / pool_geo).
#define PECLR_IO_SWITCH(io, ...)                                                  \
    do {                                                                          \
        if ((io) == PECLR_DTYPE_F32) { using IO = float; __VA_ARGS__; }           \
        else if ((io) == PECLR_DTYPE_BF16) { using IO = bf16_t; __VA_ARGS__; }    \
        else { using IO = f16_t; __VA_ARGS__; }                                   \
    } while (0)
inline bool all_aligned(std::initializer_list<const void*> ps) {
    for (const void* q : ps)
        if (q && !aligned16(q)) return false;
    return true;
}

template <typename IO>
void launch_apply(const Plan& p, hipStream_t s, const void* x, const void* res, const float* ss, int relu, void* y,
                  unsigned* mask, const float* res_ss, float* absmax) {
    const IO* xp = static_cast<const IO*>(x);
    const IO* rp = static_cast<const IO*>(res);
    IO* yp = static_cast<IO*>(y);
    if (res && res_ss && relu) hipLaunchKernelGGL((bn2d_apply_kernel<IO, 2, true>), p.grid, dim3(T), 0, s, xp, rp, p.g, ss, yp, mask, res_ss, absmax);
    else if (res && res_ss) hipLaunchKernelGGL((bn2d_apply_kernel<IO, 2, false>), p.grid, dim3(T), 0, s, xp, rp, p.g, ss, yp, mask, res_ss, absmax);
    else if (res && relu) hipLaunchKernelGGL((bn2d_apply_kernel<IO, 1, true>), p.grid, dim3(T), 0, s, xp, rp, p.g, ss, yp, mask, res_ss, absmax);
    else if (res) hipLaunchKernelGGL((bn2d_apply_kernel<IO, 1, false>), p.grid, dim3(T), 0, s, xp, rp, p.g, ss, yp, mask, res_ss, absmax);
    else if (relu) hipLaunchKernelGGL((bn2d_apply_kernel<IO, 0, true>), p.grid, dim3(T), 0, s, xp, rp, p.g, ss, yp, mask, res_ss, absmax);
    else hipLaunchKernelGGL((bn2d_apply_kernel<IO, 0, false>), p.grid, dim3(T), 0, s, xp, rp, p.g, ss, yp, mask, res_ss, absmax);
}

inline int mask_mode(int relu, const void* y, const void* mask) { return !relu ? 0 : (mask ? 3 : (y ? 2 : 1)); }

template <typename IO>
void launch_reduce(const Plan& p, hipStream_t s, int mm, const void* dy, const void* x, const void* y, const unsigned* mask,
                   const float* mean, const float* invstd, const float* ss, float* partial) {
    const IO *d = static_cast<const IO*>(dy), *xp = static_cast<const IO*>(x), *yp = static_cast<const IO*>(y);
#define PECLR_LAUNCH(M) hipLaunchKernelGGL((bn2d_bwd_reduce_kernel<IO, M>), p.grid, dim3(T), 0, s, d, xp, yp, mask, p.g, mean, invstd, ss, partial)
    if (mm == 0) PECLR_LAUNCH(0); else if (mm == 1) PECLR_LAUNCH(1); else if (mm == 2) PECLR_LAUNCH(2); else PECLR_LAUNCH(3);
#undef PECLR_LAUNCH
}

template <typename IO>
void launch_bwd_apply(const Plan& p, hipStream_t s, int mm, const void* dy, const void* x, const void* y, const unsigned* mask,
                      const float* mean, const float* invstd, const float* ss, const float* coef, void* dx, void* dres, float* absmax) {
    const IO *d = static_cast<const IO*>(dy), *xp = static_cast<const IO*>(x), *yp = static_cast<const IO*>(y);
    IO *o = static_cast<IO*>(dx), *r = static_cast<IO*>(dres);
#define PECLR_LAUNCH(M, D) hipLaunchKernelGGL((bn2d_bwd_apply_kernel<IO, M, D>), p.grid, dim3(T), 0, s, d, xp, yp, mask, p.g, mean, invstd, ss, coef, o, r, \
                                              (const float*)nullptr, 1, 1.f, absmax)
    if (dres) {
        if (mm == 0) PECLR_LAUNCH(0, true); else if (mm == 1) PECLR_LAUNCH(1, true); else if (mm == 2) PECLR_LAUNCH(2, true); else PECLR_LAUNCH(3, true);
    } else {
        if (mm == 0) PECLR_LAUNCH(0, false); else if (mm == 1) PECLR_LAUNCH(1, false); else if (mm == 2) PECLR_LAUNCH(2, false); else PECLR_LAUNCH(3, false);
    }
#undef PECLR_LAUNCH
}

}  // namespace
}  // namespace peclr

using namespace peclr;

extern "C" int peclr_bn2d_n_split(int R, int C, int io_dtype) {
    Plan p;
    return plan_for(io_dtype, R, C, 0, p) ? p.n_split : 0;
}

extern "C" int peclr_bn2d_stats(const void* x, int io_dtype, int R, int C, const float* shift, float* partial,
                                int n_split, peclr_stream_t stream) {
    if (!x || !partial) return PECLR_ERR_NULL;
    Plan p;
    if (n_split < 1 || !plan_for(io_dtype, R, C, n_split, p)) return PECLR_ERR_SHAPE;
    if (!all_aligned({x, partial, shift})) return PECLR_ERR_ALIGN;
    hipStream_t s = static_cast<hipStream_t>(stream);
    PECLR_IO_SWITCH(io_dtype, hipLaunchKernelGGL((bn2d_stats_kernel<IO>), p.grid, dim3(T), 0, s, static_cast<const IO*>(x), shift, p.g, n_split, partial));
    return launch_status();
}

extern "C" int peclr_bn2d_combine_f64(const float* partial, int n_split, int C, double* totals, peclr_stream_t stream) {
    if (!partial || !totals) return PECLR_ERR_NULL;
    if (n_split < 1 || C <= 0) return PECLR_ERR_SHAPE;
    hipLaunchKernelGGL(bn2d_combine_kernel, dim3((C + FC - 1) / FC), dim3(FT), 0, static_cast<hipStream_t>(stream), partial,
                       n_split, C, totals);
    return launch_status();
}

extern "C" int peclr_bn2d_finalize_totals_f32(const double* totals, const float* shift, int C, float eps,
                                              float momentum, const float* gamma, const float* beta, float* running_mean,
                                              float* running_var, int64_t* num_batches_tracked, float* save_mean,
                                              float* save_invstd, float* scale_shift, peclr_stream_t stream) {
    if (!totals || !shift || !gamma || !beta || !save_mean || !save_invstd || !scale_shift) return PECLR_ERR_NULL;
    if (C <= 0) return PECLR_ERR_SHAPE;
    hipLaunchKernelGGL(bn2d_finalize_totals_kernel, dim3((C + T - 1) / T), dim3(T), 0, static_cast<hipStream_t>(stream), totals,
                       shift, C, eps, momentum, gamma, beta, running_mean, running_var, num_batches_tracked, save_mean,
                       save_invstd, scale_shift);
    return launch_status();
}

extern "C" int peclr_bn2d_bwd_finalize_totals_f32(const double* local_totals, const double* global_totals,
                                                  int C, int training, const float* scale_shift, float* dgamma,
                                                  float* dbeta, float* coef, peclr_stream_t stream) {
    if (!local_totals || !global_totals || !scale_shift || !dgamma || !dbeta || !coef) return PECLR_ERR_NULL;
    if (C <= 0) return PECLR_ERR_SHAPE;
    hipLaunchKernelGGL(bn2d_bwd_finalize_totals_kernel, dim3((C + T - 1) / T), dim3(T), 0, static_cast<hipStream_t>(stream),
                       local_totals, global_totals, C, training, scale_shift, dgamma, dbeta, coef);
    return launch_status();
}

extern "C" int peclr_bn2d_finalize_f32(float* partial, int n_split, int R, int C, int training, float eps,
                                       float momentum, const float* gamma, const float* beta, float* running_mean,
                                       float* running_var, int64_t* num_batches_tracked, float* save_mean,
                                       float* save_invstd, float* scale_shift, peclr_stream_t stream) {
    if (!gamma || !beta || !save_mean || !save_invstd || !scale_shift) return PECLR_ERR_NULL;
    if (R <= 0 || C <= 0) return PECLR_ERR_SHAPE;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (training) {
        if (!partial || n_split < 1) return PECLR_ERR_NULL;
        int rows = n_split, fold = 1;
        if (fold_on() && n_split >= kFoldRows && C % 4 == 0) {         // (folds the table IN PLACE: its rows are not read again)
            fold = kFoldChunk;
            rows = (n_split + fold - 1) / fold;
            hipLaunchKernelGGL(bn2d_fold_partials_kernel, dim3((C + FC - 1) / FC, rows), dim3(FT), 0, s, partial,
                               n_split, C, fold);
        }
        hipLaunchKernelGGL(bn2d_stats_finalize_kernel, dim3((C + FC - 1) / FC), dim3(FT), 0, s, partial, rows, R, C, eps,
                           momentum, gamma, beta, running_mean, running_var, num_batches_tracked, save_mean, save_invstd,
                           scale_shift, fold, n_split);
    } else {
        if (!running_mean || !running_var) return PECLR_ERR_NULL;
        hipLaunchKernelGGL(bn2d_eval_params_kernel, dim3((C + T - 1) / T), dim3(T), 0, s, C, eps, gamma, beta, running_mean,
                           running_var, save_mean, save_invstd, scale_shift);
    }
    return launch_status();
}

extern "C" int peclr_bn2d_apply(const void* x, const void* residual, int io_dtype, int R, int C,
                                const float* scale_shift, int relu, void* y, uint32_t* relu_mask, float* absmax_out,
                                peclr_stream_t stream) {
    if (!x || !scale_shift || !y) return PECLR_ERR_NULL;
    Plan p;
    if (!plan_for(io_dtype, R, C, 0, p)) return PECLR_ERR_SHAPE;
    if (relu_mask && (C % 32 || !relu)) return PECLR_ERR_SHAPE;
    if (!all_aligned({x, y, scale_shift, residual})) return PECLR_ERR_ALIGN;
    hipStream_t s = static_cast<hipStream_t>(stream);
    PECLR_IO_SWITCH(io_dtype, launch_apply<IO>(p, s, x, residual, scale_shift, relu, y, relu_mask, nullptr, absmax_out));
    return launch_status();
}

// y = (relu)(bn(x) + bn_s(res_x)): the last pass of a layer's FIRST block, whose shortcut is conv1x1 -> BatchNorm2d.  res_x is
// that BatchNorm's input, res_scale_shift its [2][C] table; the shortcut's own apply pass and output tensor are not needed.
extern "C" int peclr_bn2d_apply_res_bn(const void* x, const void* res_x, const float* res_scale_shift, int io_dtype, int R, int C,
                                       const float* scale_shift, int relu, void* y, uint32_t* relu_mask, float* absmax_out,
                                       peclr_stream_t stream) {
    if (!x || !res_x || !res_scale_shift || !scale_shift || !y) return PECLR_ERR_NULL;
    Plan p;
    if (!plan_for(io_dtype, R, C, 0, p)) return PECLR_ERR_SHAPE;
    if (relu_mask && (C % 32 || !relu)) return PECLR_ERR_SHAPE;
    if (!all_aligned({x, y, scale_shift, res_x, res_scale_shift})) return PECLR_ERR_ALIGN;
    hipStream_t s = static_cast<hipStream_t>(stream);
    PECLR_IO_SWITCH(io_dtype, launch_apply<IO>(p, s, x, res_x, scale_shift, relu, y, relu_mask, res_scale_shift, absmax_out));
    return launch_status();
}

extern "C" int peclr_bn2d_bwd_reduce(const void* dy, const void* x, const void* y, const uint32_t* relu_mask,
                                     int io_dtype, int R, int C, int relu, const float* save_mean, const float* save_invstd,
                                     const float* scale_shift, float* partial, int n_split, peclr_stream_t stream) {
    if (!dy || !x || !save_mean || !save_invstd || !scale_shift || !partial) return PECLR_ERR_NULL;
    Plan p;
    if (n_split < 1 || !plan_for(io_dtype, R, C, n_split, p)) return PECLR_ERR_SHAPE;
    if (!all_aligned({dy, x, y, partial})) return PECLR_ERR_ALIGN;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (relu_mask && C % 32) return PECLR_ERR_SHAPE;
    const int mm = mask_mode(relu, y, relu_mask);
    PECLR_IO_SWITCH(io_dtype, launch_reduce<IO>(p, s, mm, dy, x, y, relu_mask, save_mean, save_invstd, scale_shift, partial));
    return launch_status();
}

extern "C" int peclr_bn2d_bwd_finalize_f32(float* partial, int n_split, int R, int C, int training,
                                           const float* scale_shift, float* dgamma, float* dbeta, float* coef,
                                           peclr_stream_t stream) {
    if (!partial || !scale_shift || !dgamma || !dbeta || !coef) return PECLR_ERR_NULL;
    if (n_split < 1 || R <= 0 || C <= 0) return PECLR_ERR_SHAPE;
    int rows = n_split, fold = 1;
    if (fold_on() && n_split >= kFoldRows && C % 4 == 0) {             // (as peclr_bn2d_finalize_f32)
        fold = kFoldChunk;
        rows = (n_split + fold - 1) / fold;
        hipLaunchKernelGGL(bn2d_fold_partials_kernel, dim3((C + FC - 1) / FC, rows), dim3(FT), 0, static_cast<hipStream_t>(stream),
                           partial, n_split, C, fold);
    }
    hipLaunchKernelGGL(bn2d_bwd_finalize_kernel, dim3((C + FC - 1) / FC), dim3(FT), 0, static_cast<hipStream_t>(stream),
                       partial, rows, R, C, training, scale_shift, dgamma, dbeta, coef, fold);
    return launch_status();
}

extern "C" int peclr_bn2d_bwd_apply(const void* dy, const void* x, const void* y, const uint32_t* relu_mask, int io_dtype,
                                    int R, int C, int relu,
                                    const float* save_mean, const float* save_invstd, const float* scale_shift,
                                    const float* coef, void* dx, void* d_residual, float* absmax_out, peclr_stream_t stream) {
    if (!dy || !x || !save_mean || !save_invstd || !scale_shift || !coef || !dx) return PECLR_ERR_NULL;
    Plan p;
    if (!plan_for(io_dtype, R, C, 0, p)) return PECLR_ERR_SHAPE;
    if (!all_aligned({dy, x, y, dx, d_residual})) return PECLR_ERR_ALIGN;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (relu_mask && C % 32) return PECLR_ERR_SHAPE;
    const int mm = mask_mode(relu, y, relu_mask);
    PECLR_IO_SWITCH(io_dtype, launch_bwd_apply<IO>(p, s, mm, dy, x, y, relu_mask, save_mean, save_invstd, scale_shift, coef, dx, d_residual, absmax_out));
    return launch_status();
}

// ---- stem: BN + ReLU + max-pool(3, 2, 1)
namespace {
bool pool_geo(int io_dtype, int N, int H, int W_, int C, peclr::PoolGeo& g) {
    const int w = io_dtype == PECLR_DTYPE_F32 ? 4 : (io_dtype == PECLR_DTYPE_BF16 || io_dtype == PECLR_DTYPE_F16) ? 8 : 0;
    if (!w || N <= 0 || H <= 0 || W_ <= 0 || C <= 0 || C % w) return false;
    const int cw = C / w;
    if (cw > T || T % cw) return false;
    static const int rev = getenv("PECLR_BN_REVERSE") ? atoi(getenv("PECLR_BN_REVERSE")) == 1 : 1;    // (2: the residual blocks' kernels only)
    g = {N, H, W_, C, (H - 1) / 2 + 1, (W_ - 1) / 2 + 1, cw, T / cw, rev};
    return true;
}
// grid = 8 * ceil(N / 8) * runs-per-image (see xcd_slot); cap > 0 bounds the total (the reduce's partials)
int pool_blocks(const peclr::PoolGeo& g, int pixels_per_image, int iters) {
    const int span = g.PPB * iters;
    return 8 * ((g.N + 7) / 8) * ((pixels_per_image + span - 1) / span);
}
}  // namespace

extern "C" int peclr_bn2d_pool_n_split(int N, int H, int W, int C, int io_dtype) {
    PoolGeo g;
    if (!pool_geo(io_dtype, N, H, W, C, g)) return PECLR_ERR_SHAPE;
    // ~2048 workgroups, at least four passes (one unrolled batch of loads) each
    const long long pixels = (long long)N * g.PH * g.PW;
    long long n = (pixels + 4LL * g.PPB - 1) / (4LL * g.PPB);
    if (n > 2048) n = 2048;
    return (int)(n < 1 ? 1 : n);
}

extern "C" int peclr_bn2d_pool_apply(const void* x, int io_dtype, int N, int H, int W, int C, const float* scale_shift,
                                     void* y, void* x_at_max, uint8_t* code, float* absmax_out, peclr_stream_t stream) {
    if (!x || !scale_shift || !y || !x_at_max || !code) return PECLR_ERR_NULL;
    PoolGeo g;
    if (!pool_geo(io_dtype, N, H, W, C, g)) return PECLR_ERR_SHAPE;
    if (!all_aligned({x, y, x_at_max, code, scale_shift})) return PECLR_ERR_ALIGN;
    const int blocks = pool_blocks(g, g.PH * g.PW, kPoolIter);
    hipStream_t s = static_cast<hipStream_t>(stream);
    PECLR_IO_SWITCH(io_dtype, hipLaunchKernelGGL((bn2d_pool_apply_kernel<IO>), dim3(blocks), dim3(T), 0, s, static_cast<const IO*>(x), g,
                           scale_shift, static_cast<IO*>(y), static_cast<IO*>(x_at_max), code, absmax_out));
    return launch_status();
}

extern "C" int peclr_bn2d_pool_bwd_reduce(const void* dy_pool, const void* x_at_max, int io_dtype, int N, int H, int W, int C,
                                          const float* save_mean, const float* save_invstd, const float* scale_shift,
                                          float* partial, int n_split, peclr_stream_t stream) {
    if (!dy_pool || !x_at_max || !save_mean || !save_invstd || !scale_shift || !partial) return PECLR_ERR_NULL;
    PoolGeo g;
    if (!pool_geo(io_dtype, N, H, W, C, g) || n_split < 1) return PECLR_ERR_SHAPE;
    if (!all_aligned({dy_pool, x_at_max, partial})) return PECLR_ERR_ALIGN;
    hipStream_t s = static_cast<hipStream_t>(stream);
    PECLR_IO_SWITCH(io_dtype, hipLaunchKernelGGL((bn2d_pool_bwd_reduce_kernel<IO>), dim3(n_split), dim3(T), 0, s, static_cast<const IO*>(dy_pool),
                           static_cast<const IO*>(x_at_max), g, save_mean, save_invstd, scale_shift, partial));
    return launch_status();
}

extern "C" int peclr_bn2d_pool_bwd_apply(const void* dy_pool, const void* x, const uint8_t* code, int io_dtype, int N, int H,
                                         int W, int C, const float* save_mean, const float* save_invstd,
                                         const float* scale_shift, const float* coef, void* dx, peclr_stream_t stream) {
    if (!dy_pool || !x || !code || !save_mean || !save_invstd || !scale_shift || !coef || !dx) return PECLR_ERR_NULL;
    PoolGeo g;
    if (!pool_geo(io_dtype, N, H, W, C, g)) return PECLR_ERR_SHAPE;
    if (!all_aligned({dy_pool, x, code, dx})) return PECLR_ERR_ALIGN;
    hipStream_t s = static_cast<hipStream_t>(stream);
    static const int wide16 = getenv("PECLR_POOL_BWD16") ? atoi(getenv("PECLR_POOL_BWD16")) : 1;      // (0: the four-channel form, A/B)
    if (wide16 && io_dtype != PECLR_DTYPE_F32 && C <= 512) {      // 16-bit storage: eight channels per lane (pool_geo's own geometry)
        const int blocks16 = pool_blocks(g, H * W, 2 * kPoolIter);
        if (io_dtype == PECLR_DTYPE_BF16)
            hipLaunchKernelGGL((bn2d_pool_bwd_apply16_kernel<bf16_t>), dim3(blocks16), dim3(T), 0, s, static_cast<const bf16_t*>(dy_pool),
                               static_cast<const bf16_t*>(x), code, g, save_mean, save_invstd, scale_shift, coef, static_cast<bf16_t*>(dx));
        else
            hipLaunchKernelGGL((bn2d_pool_bwd_apply16_kernel<f16_t>), dim3(blocks16), dim3(T), 0, s, static_cast<const f16_t*>(dy_pool),
                               static_cast<const f16_t*>(x), code, g, save_mean, save_invstd, scale_shift, coef, static_cast<f16_t*>(dx));
        return launch_status();
    }
    if (C % 4 || C / 4 > T || T % (C / 4)) return PECLR_ERR_SHAPE;
    g.CW = C / 4;                 // this kernel moves four channels per lane whatever the storage type
    g.PPB = T / g.CW;
    const int blocks = pool_blocks(g, H * W, 2 * kPoolIter);
    PECLR_IO_SWITCH(io_dtype, hipLaunchKernelGGL((bn2d_pool_bwd_apply_kernel<IO>), dim3(blocks), dim3(T), 0, s, static_cast<const IO*>(dy_pool),
                           static_cast<const IO*>(x), code, g, save_mean, save_invstd, scale_shift, coef,
                           static_cast<IO*>(dx)));
    return launch_status();
}

// ---- layer4's last block: BN + identity + ReLU + global average pool (+ flatten)
extern "C" int peclr_bn2d_apply_avgpool(const void* x, const void* residual, int io_dtype, int N, int HW, int C,
                                        const float* scale_shift, float* pooled, uint32_t* relu_mask, peclr_stream_t stream) {
    if (!x || !residual || !scale_shift || !pooled || !relu_mask) return PECLR_ERR_NULL;
    if (N <= 0 || HW <= 0 || C % 32 || (long long)N * HW > 0x7fffffffLL) return PECLR_ERR_SHAPE;
    Plan p;
    if (!plan_for(io_dtype, N * HW, C, 1, p)) return PECLR_ERR_SHAPE;
    if (!all_aligned({x, residual, scale_shift, pooled})) return PECLR_ERR_ALIGN;
    p.g.rows_per_block = HW;                       // one image per block row
    const dim3 grid(p.grid.x, N);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const float inv = 1.0f / (float)HW;
    PECLR_IO_SWITCH(io_dtype, hipLaunchKernelGGL((bn2d_apply_avgpool_kernel<IO>), grid, dim3(T), 0, s, static_cast<const IO*>(x),
                           static_cast<const IO*>(residual), p.g, scale_shift, inv, pooled, relu_mask));
    return launch_status();
}

extern "C" int peclr_bn2d_bwd_reduce_avgpool(const float* d_pooled, const void* x, const uint32_t* relu_mask, int io_dtype,
                                             int N, int HW, int C, const float* save_mean, const float* save_invstd,
                                             const float* scale_shift, float* partial, int n_split, peclr_stream_t stream) {
    if (!d_pooled || !x || !relu_mask || !save_mean || !save_invstd || !scale_shift || !partial) return PECLR_ERR_NULL;
    if (N <= 0 || HW <= 0 || C % 32 || (long long)N * HW > 0x7fffffffLL) return PECLR_ERR_SHAPE;
    Plan p;
    if (n_split < 1 || !plan_for(io_dtype, N * HW, C, n_split, p)) return PECLR_ERR_SHAPE;
    if (!all_aligned({d_pooled, x, partial})) return PECLR_ERR_ALIGN;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const float inv = 1.0f / (float)HW;
    PECLR_IO_SWITCH(io_dtype, hipLaunchKernelGGL((bn2d_bwd_reduce_kernel<IO, 3, true>), p.grid, dim3(T), 0, s, (const IO*)nullptr,
                           static_cast<const IO*>(x), (const IO*)nullptr, relu_mask, p.g, save_mean, save_invstd, scale_shift,
                           partial, d_pooled, HW, inv));
    return launch_status();
}

extern "C" int peclr_bn2d_bwd_apply_avgpool(const float* d_pooled, const void* x, const uint32_t* relu_mask, int io_dtype,
                                            int N, int HW, int C, const float* save_mean, const float* save_invstd,
                                            const float* scale_shift, const float* coef, void* dx, void* d_residual,
                                            float* absmax_out, peclr_stream_t stream) {
    if (!d_pooled || !x || !relu_mask || !save_mean || !save_invstd || !scale_shift || !coef || !dx || !d_residual)
        return PECLR_ERR_NULL;
    if (N <= 0 || HW <= 0 || C % 32 || (long long)N * HW > 0x7fffffffLL) return PECLR_ERR_SHAPE;
    Plan p;
    if (!plan_for(io_dtype, N * HW, C, 0, p)) return PECLR_ERR_SHAPE;
    if (!all_aligned({d_pooled, x, dx, d_residual})) return PECLR_ERR_ALIGN;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const float inv = 1.0f / (float)HW;
    PECLR_IO_SWITCH(io_dtype, hipLaunchKernelGGL((bn2d_bwd_apply_kernel<IO, 3, true, true>), p.grid, dim3(T), 0, s, (const IO*)nullptr,
                           static_cast<const IO*>(x), (const IO*)nullptr, relu_mask, p.g, save_mean, save_invstd, scale_shift,
                           coef, static_cast<IO*>(dx), static_cast<IO*>(d_residual), d_pooled, HW, inv, absmax_out));
    return launch_status();
}
