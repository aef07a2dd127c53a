// K2: BatchNorm1d (train/eval) + ReLU of the projection head, forward and backward
// (nn.BatchNorm1d(H) + nn.ReLU at simclr_model.py:27-28), with the split-K reduction of the
// first Linear and its bias fused into the load.
//
// Roofline: HBM (really L2 / Infinity Cache at these sizes): algorithmic bytes
// fwd = 4*M*H*(n_slabs + 2), bwd = 4*M*H*3.  At M = 256 that is 5 MB -- ~1 us of HBM time -- so
// the kernel is latency-bound and is organised to pay the memory latency ONCE:
//   - a workgroup owns 16 feature columns and all M rows: 256 threads = 64 row slices x 4 float4
//     column groups (each quarter-wave reads one 64-byte row segment);
//   - when M <= 64*RPT (RPT <= 16) every thread keeps its rows in REGISTERS: all its loads (rows x
//     slabs) are issued back to back, and the mean / variance / normalise passes (forward) or the
//     dbeta,dgamma / dx passes (backward) run out of registers -- one trip to memory, no re-read;
//   - larger M falls back to a strided loop that re-reads the workgroup's own column block.
// Column reductions: 4 wave64 butterfly steps across the 16 slices of a wave, then a 4-wave LDS
// step, always in the same order (bit-reproducible, no atomics).  Statistics follow torch:
// biased variance for the normalisation, unbiased for running_var, eps inside the sqrt.
#include "common.hpp"

namespace peclr {
namespace {

constexpr int TW = 4;            // float4 column groups per workgroup
constexpr int COLS = 4 * TW;     // 16 columns per workgroup
constexpr int SLICES = 256 / TW; // 64 row slices

__device__ __forceinline__ float4 f4(float v) { return make_float4(v, v, v, v); }
__device__ __forceinline__ float4 operator+(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 operator-(float4 a, float4 b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }
__device__ __forceinline__ float4 operator*(float4 a, float4 b) { return make_float4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w); }
__device__ __forceinline__ float4 operator*(float4 a, float s) { return make_float4(a.x * s, a.y * s, a.z * s, a.w * s); }

// sum_k slabs[k][o .. o+3]: loads are issued in batches of 8 before any add, so a thread pays the
// memory latency once per batch instead of once per slab.
__device__ __forceinline__ float4 slab_sum(const float* __restrict__ p, size_t slab, int n_slabs) {
    float4 v = f4(0.f);
    for (int k0 = 0; k0 < n_slabs; k0 += 8) {
        float4 t[8];
#pragma unroll
        for (int j = 0; j < 8; ++j)
            t[j] = (k0 + j < n_slabs) ? *reinterpret_cast<const float4*>(p + (size_t)(k0 + j) * slab) : f4(0.f);
#pragma unroll
        for (int j = 0; j < 8; ++j) v = v + t[j];
    }
    return v;
}

// Sum over the 64 row slices of the workgroup, result broadcast to every thread of the column group.
// tid = slice * TW + tc: a wave holds 16 slices x 4 column groups.
__device__ __forceinline__ float4 slice_sum(float4 v, float4 (*red)[TW], int tc) {
#pragma unroll
    for (int o = TW; o < 64; o <<= 1) {
        v.x += __shfl_xor(v.x, o, kWave);
        v.y += __shfl_xor(v.y, o, kWave);
        v.z += __shfl_xor(v.z, o, kWave);
        v.w += __shfl_xor(v.w, o, kWave);
    }
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) < TW) red[wave][tc] = v;
    __syncthreads();
    const float4 t = (red[0][tc] + red[1][tc]) + (red[2][tc] + red[3][tc]);
    __syncthreads();
    return t;
}

struct BnFwdArgs {
    const float* slabs; const float* bias; const float* gamma; const float* beta;
    float* running_mean; float* running_var; int64_t* nbt;
    float* a_pre; float* a_out; float* save_mean; float* save_invstd;
    int n_slabs, M, H, training;
    float eps, momentum;
};

// RPT > 0: rows held in registers (M <= SLICES*RPT).  RPT == 0: strided loop, re-reads a_pre.
template <int RPT>
__global__ __launch_bounds__(256) void bn_relu_fwd_kernel(BnFwdArgs g) {
    __shared__ float4 red[4][TW];
    const int tc = threadIdx.x & (TW - 1), s = threadIdx.x / TW;
    const int col = blockIdx.x * COLS + 4 * tc;
    const bool ok = col < g.H;  // H % 4 == 0: a float4 is all-in or all-out
    const size_t slab = (size_t)g.M * g.H;
    const float4 bv = (ok && g.bias) ? *reinterpret_cast<const float4*>(g.bias + col) : f4(0.f);
    constexpr int R = RPT > 0 ? RPT : 1;
    float4 x[R];

    float4 sum = f4(0.f);
    if (RPT > 0) {
#pragma unroll
        for (int j = 0; j < R; ++j) {
            const int r = s + SLICES * j;
            x[j] = f4(0.f);
            if (ok && r < g.M) {
                const size_t o = (size_t)r * g.H + col;
                const float4 v = slab_sum(g.slabs + o, slab, g.n_slabs) + bv;
                x[j] = v;
                *reinterpret_cast<float4*>(g.a_pre + o) = v;
                sum = sum + v;
            }
        }
    } else if (ok) {
        for (int r = s; r < g.M; r += SLICES) {
            const size_t o = (size_t)r * g.H + col;
            const float4 v = slab_sum(g.slabs + o, slab, g.n_slabs) + bv;
            *reinterpret_cast<float4*>(g.a_pre + o) = v;  // re-read below by the SAME thread
            sum = sum + v;
        }
    }

    float4 mean, invstd;
    if (g.training) {
        const float inv_m = 1.0f / (float)g.M;
        mean = slice_sum(sum, red, tc) * inv_m;
        float4 ss = f4(0.f);
        if (RPT > 0) {
#pragma unroll
            for (int j = 0; j < R; ++j)
                if (ok && s + SLICES * j < g.M) {
                    const float4 d = x[j] - mean;
                    ss = ss + d * d;
                }
        } else if (ok) {
            for (int r = s; r < g.M; r += SLICES) {
                const float4 d = *reinterpret_cast<const float4*>(g.a_pre + (size_t)r * g.H + col) - mean;
                ss = ss + d * d;
            }
        }
        const float4 var = slice_sum(ss, red, tc) * inv_m;
        invstd = make_float4(1.0f / sqrtf(var.x + g.eps), 1.0f / sqrtf(var.y + g.eps),
                             1.0f / sqrtf(var.z + g.eps), 1.0f / sqrtf(var.w + g.eps));
        if (ok && s == 0) {
            const float mom = g.momentum;
            if (g.running_mean) {
                float4* rm = reinterpret_cast<float4*>(g.running_mean + col);
                *rm = *rm * (1.f - mom) + mean * mom;
            }
            if (g.running_var) {
                const float ub = (float)g.M / (float)(g.M > 1 ? g.M - 1 : 1);
                float4* rv = reinterpret_cast<float4*>(g.running_var + col);
                *rv = *rv * (1.f - mom) + var * (ub * mom);
            }
        }
        if (g.nbt && blockIdx.x == 0 && threadIdx.x == 0) *g.nbt += 1;
    } else {
        mean = ok ? *reinterpret_cast<const float4*>(g.running_mean + col) : f4(0.f);
        const float4 rv = ok ? *reinterpret_cast<const float4*>(g.running_var + col) : f4(1.f);
        invstd = make_float4(1.0f / sqrtf(rv.x + g.eps), 1.0f / sqrtf(rv.y + g.eps), 1.0f / sqrtf(rv.z + g.eps),
                             1.0f / sqrtf(rv.w + g.eps));
    }
    if (!ok) return;
    if (s == 0) {
        *reinterpret_cast<float4*>(g.save_mean + col) = mean;
        *reinterpret_cast<float4*>(g.save_invstd + col) = invstd;
    }
    const float4 gm = *reinterpret_cast<const float4*>(g.gamma + col);
    const float4 bt = *reinterpret_cast<const float4*>(g.beta + col);
    auto emit = [&](int r, float4 v) {
        const float4 y = (v - mean) * invstd * gm + bt;
        *reinterpret_cast<float4*>(g.a_out + (size_t)r * g.H + col) =
            make_float4(fmaxf(y.x, 0.f), fmaxf(y.y, 0.f), fmaxf(y.z, 0.f), fmaxf(y.w, 0.f));
    };
    if (RPT > 0) {
#pragma unroll
        for (int j = 0; j < R; ++j)
            if (s + SLICES * j < g.M) emit(s + SLICES * j, x[j]);
    } else {
        for (int r = s; r < g.M; r += SLICES) emit(r, *reinterpret_cast<const float4*>(g.a_pre + (size_t)r * g.H + col));
    }
}

struct BnBwdArgs {
    const float* da; const float* a_pre; const float* save_mean; const float* save_invstd;
    const float* gamma; const float* beta;
    float* d_a_pre; float* dgamma; float* dbeta; float* dbias;
    int M, H, training;
};

__device__ __forceinline__ float4 relu_mask(float4 y, float4 d) {
    return make_float4(y.x > 0.f ? d.x : 0.f, y.y > 0.f ? d.y : 0.f, y.z > 0.f ? d.z : 0.f, y.w > 0.f ? d.w : 0.f);
}

template <int RPT>
__global__ __launch_bounds__(256) void bn_relu_bwd_kernel(BnBwdArgs g) {
    __shared__ float4 red[4][TW];
    const int tc = threadIdx.x & (TW - 1), s = threadIdx.x / TW;
    const int col = blockIdx.x * COLS + 4 * tc;
    const bool ok = col < g.H;
    const float4 mean = ok ? *reinterpret_cast<const float4*>(g.save_mean + col) : f4(0.f);
    const float4 invstd = ok ? *reinterpret_cast<const float4*>(g.save_invstd + col) : f4(0.f);
    const float4 gm = ok ? *reinterpret_cast<const float4*>(g.gamma + col) : f4(0.f);
    const float4 bt = ok ? *reinterpret_cast<const float4*>(g.beta + col) : f4(0.f);
    constexpr int R = RPT > 0 ? RPT : 1;
    float4 xh[R], dy[R];

    auto load = [&](int r, float4& xhat, float4& d) {
        const size_t o = (size_t)r * g.H + col;
        xhat = (*reinterpret_cast<const float4*>(g.a_pre + o) - mean) * invstd;
        d = relu_mask(xhat * gm + bt, *reinterpret_cast<const float4*>(g.da + o));
    };
    float4 sb = f4(0.f), sg = f4(0.f);
    if (RPT > 0) {
#pragma unroll
        for (int j = 0; j < R; ++j) {
            xh[j] = dy[j] = f4(0.f);
            if (ok && s + SLICES * j < g.M) {
                load(s + SLICES * j, xh[j], dy[j]);
                sb = sb + dy[j];
                sg = sg + dy[j] * xh[j];
            }
        }
    } else if (ok) {
        for (int r = s; r < g.M; r += SLICES) {
            float4 a, d;
            load(r, a, d);
            sb = sb + d;
            sg = sg + d * a;
        }
    }
    const float4 db = slice_sum(sb, red, tc);
    const float4 dg = slice_sum(sg, red, tc);
    const float fm = (float)g.M;
    const float4 k = gm * invstd * (1.0f / fm);
    float4 sbias = f4(0.f);
    auto emit = [&](int r, float4 xhat, float4 d) {
        const float4 dx = g.training ? k * (d * fm - db - xhat * dg) : gm * invstd * d;
        *reinterpret_cast<float4*>(g.d_a_pre + (size_t)r * g.H + col) = dx;
        sbias = sbias + dx;
    };
    if (RPT > 0) {
#pragma unroll
        for (int j = 0; j < R; ++j)
            if (ok && s + SLICES * j < g.M) emit(s + SLICES * j, xh[j], dy[j]);
    } else if (ok) {
        for (int r = s; r < g.M; r += SLICES) {
            float4 a, d;
            load(r, a, d);
            emit(r, a, d);
        }
    }
    const float4 dbi = slice_sum(sbias, red, tc);
    if (ok && s == 0) {
        *reinterpret_cast<float4*>(g.dgamma + col) = dg;
        *reinterpret_cast<float4*>(g.dbeta + col) = db;
        if (g.dbias) *reinterpret_cast<float4*>(g.dbias + col) = dbi;
    }
}

inline int rows_per_thread(int M) {
    const int need = (M + SLICES - 1) / SLICES;
    for (int r = 1; r <= 16; r <<= 1)
        if (need <= r) return r;
    return 0;  // strided path
}

template <template <int> class Launcher, typename Args>
void dispatch(int rpt, dim3 grid, hipStream_t s, const Args& a) {
    switch (rpt) {
        case 1: Launcher<1>::run(grid, s, a); break;
        case 2: Launcher<2>::run(grid, s, a); break;
        case 4: Launcher<4>::run(grid, s, a); break;
        case 8: Launcher<8>::run(grid, s, a); break;
        case 16: Launcher<16>::run(grid, s, a); break;
        default: Launcher<0>::run(grid, s, a); break;
    }
}
template <int RPT> struct FwdLauncher {
    static void run(dim3 grid, hipStream_t s, const BnFwdArgs& a) {
        hipLaunchKernelGGL((bn_relu_fwd_kernel<RPT>), grid, dim3(256), 0, s, a);
    }
};
template <int RPT> struct BwdLauncher {
    static void run(dim3 grid, hipStream_t s, const BnBwdArgs& a) {
        hipLaunchKernelGGL((bn_relu_bwd_kernel<RPT>), grid, dim3(256), 0, s, a);
    }
};

}  // namespace
}  // namespace peclr

using namespace peclr;

extern "C" int peclr_bn_relu_fwd_f32(const float* a_slabs, int n_slabs, const float* bias, int M, int H,
                                     const float* gamma, const float* beta, float eps, float momentum,
                                     int training, float* running_mean, float* running_var,
                                     int64_t* num_batches_tracked, float* a_pre, float* a_out,
                                     float* save_mean, float* save_invstd, peclr_stream_t stream) {
    if (!a_slabs || !gamma || !beta || !a_pre || !a_out || !save_mean || !save_invstd) return PECLR_ERR_NULL;
    if (!training && (!running_mean || !running_var)) return PECLR_ERR_NULL;
    if (M <= 0 || H <= 0 || n_slabs < 1) return PECLR_ERR_SHAPE;
    if (H % 4) return PECLR_ERR_ALIGN;
    const void* ptrs[] = {a_slabs, bias, gamma, beta, running_mean, running_var, a_pre, a_out, save_mean, save_invstd};
    for (const void* p : ptrs)
        if (p && !aligned16(p)) return PECLR_ERR_ALIGN;
    BnFwdArgs g = {a_slabs, bias, gamma, beta, running_mean, running_var, num_batches_tracked, a_pre, a_out,
                   save_mean, save_invstd, n_slabs, M, H, training, eps, momentum};
    dispatch<FwdLauncher>(rows_per_thread(M), dim3((H + COLS - 1) / COLS), static_cast<hipStream_t>(stream), g);
    return launch_status();
}

extern "C" int peclr_bn_relu_bwd_f32(const float* d_a_out, const float* a_pre, const float* save_mean,
                                     const float* save_invstd, const float* gamma, const float* beta, int M,
                                     int H, int training, float* d_a_pre, float* dgamma, float* dbeta,
                                     float* dbias, peclr_stream_t stream) {
    if (!d_a_out || !a_pre || !save_mean || !save_invstd || !gamma || !beta || !d_a_pre || !dgamma || !dbeta)
        return PECLR_ERR_NULL;
    if (M <= 0 || H <= 0) return PECLR_ERR_SHAPE;
    if (H % 4) return PECLR_ERR_ALIGN;
    const void* ptrs[] = {d_a_out, a_pre, save_mean, save_invstd, gamma, beta, d_a_pre, dgamma, dbeta, dbias};
    for (const void* p : ptrs)
        if (p && !aligned16(p)) return PECLR_ERR_ALIGN;
    BnBwdArgs g = {d_a_out, a_pre, save_mean, save_invstd, gamma, beta, d_a_pre, dgamma, dbeta, dbias, M, H, training};
    dispatch<BwdLauncher>(rows_per_thread(M), dim3((H + COLS - 1) / COLS), static_cast<hipStream_t>(stream), g);
    return launch_status();
}
