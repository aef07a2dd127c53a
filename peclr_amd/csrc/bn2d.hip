// Fused BatchNorm2d (+ residual add) (+ ReLU) for the NHWC ResNet backbone, forward and backward.
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
// the head's BatchNorm1d.  Threads own one float4 of 4 channels for the whole kernel
// (tid -> (row lane, column group) with the column group fastest, so consecutive lanes read
// consecutive 16-byte words: every wave-instruction is a 1 KiB contiguous burst) and walk rows with
// 8 independent loads in flight.
//
//   forward  = stats (partial sum / sum-of-squares per row slice, shifted by row 0 to avoid
//              cancellation)  ->  finalize (fixed-order combine in float64, running stats)
//              ->  apply: y = relu(x*scale + shift + residual)
//   backward = reduce (partial dbeta, dgamma with the ReLU mask recomputed from x, or read from y
//              when a residual was added)  ->  finalize  ->  apply: dx (and d_residual = masked dy)
//
// Algorithmic bytes per element (fp32): fwd 4 (stats) + 8 (apply) [+4 residual];
// bwd 8 (reduce) + 12 (apply) [+4 y, twice, and +4 d_residual when a residual was added].
// The stock path moves 8+12 (BN) + 8 (ReLU) + 12 (add) forward and 12 (ReLU) + 20 (BN) backward.
// Everything is bit-reproducible (no atomics; fixed combine order).
#include "common.hpp"

namespace peclr {
namespace {

constexpr int T = 256;
constexpr int UNROLL = 8;

__device__ __forceinline__ float4 f4(float v) { return make_float4(v, v, v, v); }
__device__ __forceinline__ float4 operator+(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 operator-(float4 a, float4 b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }
__device__ __forceinline__ float4 operator*(float4 a, float4 b) { return make_float4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w); }
__device__ __forceinline__ float4 fma4(float4 a, float4 b, float4 c) {
    return make_float4(fmaf(a.x, b.x, c.x), fmaf(a.y, b.y, c.y), fmaf(a.z, b.z, c.z), fmaf(a.w, b.w, c.w));
}
__device__ __forceinline__ float4 relu4(float4 a) { return make_float4(fmaxf(a.x, 0.f), fmaxf(a.y, 0.f), fmaxf(a.z, 0.f), fmaxf(a.w, 0.f)); }
__device__ __forceinline__ float4 mask4(float4 y, float4 d) {
    return make_float4(y.x > 0.f ? d.x : 0.f, y.y > 0.f ? d.y : 0.f, y.z > 0.f ? d.z : 0.f, y.w > 0.f ? d.w : 0.f);
}
typedef float v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store_stream(float* p, float4 x) {  // written once, read by a later kernel
    *reinterpret_cast<float4*>(p) = x;
}

// Geometry shared by every kernel: C4 = C/4 column groups; a block covers CGB = min(C4, 256) of them
// (blockIdx.x = column block) and RPP = 256/CGB rows per pass; blockIdx.y = row slice.
struct Geo {
    int R, C, CGB, RPP, rows_per_block;
};
__device__ __forceinline__ void thread_geo(const Geo& g, int& col, int& r_begin, int& r_end, int& rl) {
    const int cgl = threadIdx.x % g.CGB;
    rl = threadIdx.x / g.CGB;
    col = (blockIdx.x * g.CGB + cgl) * 4;
    r_begin = blockIdx.y * g.rows_per_block;
    r_end = min(g.R, r_begin + g.rows_per_block);
}

// Sum over the block's row lanes (threads with equal column group); valid in row lane 0.
__device__ __forceinline__ float4 lane_reduce(float4 v, float4* red, const Geo& g, int rl) {
    if (g.RPP == 1) return v;
    __syncthreads();
    red[threadIdx.x] = v;
    __syncthreads();
    float4 t = f4(0.f);
    if (rl == 0)
        for (int k = 0; k < g.RPP; ++k) t = t + red[k * g.CGB + threadIdx.x];
    return t;
}

// ------------------------------------------------------------------ forward: statistics
__global__ __launch_bounds__(T) void bn2d_stats_kernel(const float* __restrict__ x, Geo g, float* __restrict__ partial) {
    __shared__ float4 red[T];
    int col, r0, r1, rl;
    thread_geo(g, col, r0, r1, rl);
    const float4 k0 = *reinterpret_cast<const float4*>(x + col);  // shift = row 0 (same for every block)
    float4 s = f4(0.f), q = f4(0.f);
    int r = r0 + rl;
    for (; r + (UNROLL - 1) * g.RPP < r1; r += UNROLL * g.RPP) {
        float4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) v[u] = *reinterpret_cast<const float4*>(x + (size_t)(r + u * g.RPP) * g.C + col);
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const float4 d = v[u] - k0;
            s = s + d;
            q = fma4(d, d, q);
        }
    }
    for (; r < r1; r += g.RPP) {
        const float4 d = *reinterpret_cast<const float4*>(x + (size_t)r * g.C + col) - k0;
        s = s + d;
        q = fma4(d, d, q);
    }
    s = lane_reduce(s, red, g, rl);
    q = lane_reduce(q, red, g, rl);
    if (rl == 0) {
        float* o = partial + (size_t)blockIdx.y * 2 * g.C;
        *reinterpret_cast<float4*>(o + col) = s;
        *reinterpret_cast<float4*>(o + g.C + col) = q;
    }
}

// Finalize kernels: a 1024-thread workgroup owns 32 channels; its 32 row lanes each combine every 32nd row-slice
// partial in float64 (coalesced 128-byte reads), then lane 0 adds the 32 lane sums in a fixed order.
constexpr int FT = 1024, FC = 32, FL = FT / FC;
__device__ __forceinline__ void combine_partials(const float* __restrict__ partial, int n_split, int C, int c, int lane,
                                                 double (*red)[2][FC], double& a, double& b) {
    double s = 0.0, q = 0.0;
    if (c < C) {
        int k = lane;
        for (; k + 3 * FL < n_split; k += 4 * FL) {  // 8 independent loads in flight
            const float s0 = partial[(size_t)k * 2 * C + c], q0 = partial[(size_t)k * 2 * C + C + c];
            const float s1 = partial[(size_t)(k + FL) * 2 * C + c], q1 = partial[(size_t)(k + FL) * 2 * C + C + c];
            const float s2 = partial[(size_t)(k + 2 * FL) * 2 * C + c], q2 = partial[(size_t)(k + 2 * FL) * 2 * C + C + c];
            const float s3 = partial[(size_t)(k + 3 * FL) * 2 * C + c], q3 = partial[(size_t)(k + 3 * FL) * 2 * C + C + c];
            s += (double)s0; s += (double)s1; s += (double)s2; s += (double)s3;
            q += (double)q0; q += (double)q1; q += (double)q2; q += (double)q3;
        }
        for (; k < n_split; k += FL) {
            s += (double)partial[(size_t)k * 2 * C + c];
            q += (double)partial[(size_t)k * 2 * C + C + c];
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

__global__ __launch_bounds__(FT) void bn2d_stats_finalize_kernel(const float* __restrict__ x, const float* __restrict__ partial,
                                                                int n_split, int R, int C, float eps, float momentum,
                                                                const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                float* running_mean, float* running_var, int64_t* nbt,
                                                                float* __restrict__ save_mean, float* __restrict__ save_invstd,
                                                                float* __restrict__ scale_shift) {
    __shared__ double red[FL][2][FC];
    const int c = blockIdx.x * FC + threadIdx.x % FC, lane = threadIdx.x / FC;
    if (blockIdx.x == 0 && threadIdx.x == 0 && nbt) *nbt += 1;
    double s, q;
    combine_partials(partial, n_split, C, c, lane, red, s, q);
    if (lane != 0 || c >= C) return;
    const double k0 = (double)x[c];
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
template <bool RES, bool RELU>
__global__ __launch_bounds__(T) void bn2d_apply_kernel(const float* __restrict__ x, const float* __restrict__ res, Geo g,
                                                       const float* __restrict__ scale_shift, float* __restrict__ y) {
    int col, r0, r1, rl;
    thread_geo(g, col, r0, r1, rl);
    const float4 sc = *reinterpret_cast<const float4*>(scale_shift + col);
    const float4 sh = *reinterpret_cast<const float4*>(scale_shift + g.C + col);
    int r = r0 + rl;
    for (; r + (UNROLL - 1) * g.RPP < r1; r += UNROLL * g.RPP) {
        float4 v[UNROLL], w[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const size_t o = (size_t)(r + u * g.RPP) * g.C + col;
            v[u] = *reinterpret_cast<const float4*>(x + o);
            if (RES) w[u] = *reinterpret_cast<const float4*>(res + o);
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            float4 t = fma4(v[u], sc, sh);
            if (RES) t = t + w[u];
            if (RELU) t = relu4(t);
            store_stream(y + (size_t)(r + u * g.RPP) * g.C + col, t);
        }
    }
    for (; r < r1; r += g.RPP) {
        const size_t o = (size_t)r * g.C + col;
        float4 t = fma4(*reinterpret_cast<const float4*>(x + o), sc, sh);
        if (RES) t = t + *reinterpret_cast<const float4*>(res + o);
        if (RELU) t = relu4(t);
        store_stream(y + o, t);
    }
}

// ------------------------------------------------------------------ backward: reduce
// MASK: 0 = no ReLU, 1 = ReLU mask recomputed from x (no residual), 2 = ReLU mask read from y.
template <int MASK>
__global__ __launch_bounds__(T) void bn2d_bwd_reduce_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                            const float* __restrict__ y, Geo g,
                                                            const float* __restrict__ save_mean, const float* __restrict__ save_invstd,
                                                            const float* __restrict__ scale_shift, float* __restrict__ partial) {
    __shared__ float4 red[T];
    int col, r0, r1, rl;
    thread_geo(g, col, r0, r1, rl);
    const float4 mean = *reinterpret_cast<const float4*>(save_mean + col);
    const float4 invstd = *reinterpret_cast<const float4*>(save_invstd + col);
    const float4 sc = *reinterpret_cast<const float4*>(scale_shift + col);
    const float4 sh = *reinterpret_cast<const float4*>(scale_shift + g.C + col);
    float4 sb = f4(0.f), sg = f4(0.f);
    int r = r0 + rl;
    auto acc = [&](float4 d, float4 xv, float4 yv) {
        if (MASK == 1) d = mask4(fma4(xv, sc, sh), d);
        if (MASK == 2) d = mask4(yv, d);
        sb = sb + d;
        sg = fma4(d, (xv - mean) * invstd, sg);
    };
    for (; r + (UNROLL - 1) * g.RPP < r1; r += UNROLL * g.RPP) {
        float4 d[UNROLL], xv[UNROLL], yv[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const size_t o = (size_t)(r + u * g.RPP) * g.C + col;
            d[u] = *reinterpret_cast<const float4*>(dy + o);
            xv[u] = *reinterpret_cast<const float4*>(x + o);
            if (MASK == 2) yv[u] = *reinterpret_cast<const float4*>(y + o);
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) acc(d[u], xv[u], MASK == 2 ? yv[u] : f4(0.f));
    }
    for (; r < r1; r += g.RPP) {
        const size_t o = (size_t)r * g.C + col;
        acc(*reinterpret_cast<const float4*>(dy + o), *reinterpret_cast<const float4*>(x + o),
            MASK == 2 ? *reinterpret_cast<const float4*>(y + o) : f4(0.f));
    }
    sb = lane_reduce(sb, red, g, rl);
    sg = lane_reduce(sg, red, g, rl);
    if (rl == 0) {
        float* o = partial + (size_t)blockIdx.y * 2 * g.C;
        *reinterpret_cast<float4*>(o + col) = sb;
        *reinterpret_cast<float4*>(o + g.C + col) = sg;
    }
}

// dbeta, dgamma, and the two per-channel coefficients of the dx pass:
//   training: dx = k1*dy' + (k2 + k3*xhat)  with k1 = gamma*invstd, k2 = -k1*dbeta/R, k3 = -k1*dgamma/R
//   eval    : dx = k1*dy'
__global__ __launch_bounds__(FT) void bn2d_bwd_finalize_kernel(const float* __restrict__ partial, int n_split, int R, int C,
                                                              int training, const float* __restrict__ scale_shift,
                                                              float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                              float* __restrict__ coef) {
    __shared__ double red[FL][2][FC];
    const int c = blockIdx.x * FC + threadIdx.x % FC, lane = threadIdx.x / FC;
    double sb, sg;
    combine_partials(partial, n_split, C, c, lane, red, sb, sg);
    if (lane != 0 || c >= C) return;
    dbeta[c] = (float)sb;
    dgamma[c] = (float)sg;
    const double k1 = (double)scale_shift[c];
    coef[c] = training ? (float)(-k1 * sb / R) : 0.f;
    coef[C + c] = training ? (float)(-k1 * sg / R) : 0.f;
}

template <int MASK, bool DRES>
__global__ __launch_bounds__(T) void bn2d_bwd_apply_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                           const float* __restrict__ y, Geo g,
                                                           const float* __restrict__ save_mean, const float* __restrict__ save_invstd,
                                                           const float* __restrict__ scale_shift, const float* __restrict__ coef,
                                                           float* __restrict__ dx, float* __restrict__ dres) {
    int col, r0, r1, rl;
    thread_geo(g, col, r0, r1, rl);
    const float4 mean = *reinterpret_cast<const float4*>(save_mean + col);
    const float4 invstd = *reinterpret_cast<const float4*>(save_invstd + col);
    const float4 sc = *reinterpret_cast<const float4*>(scale_shift + col);
    const float4 sh = *reinterpret_cast<const float4*>(scale_shift + g.C + col);
    const float4 k2 = *reinterpret_cast<const float4*>(coef + col);
    const float4 k3 = *reinterpret_cast<const float4*>(coef + g.C + col);
    auto emit = [&](size_t o, float4 d, float4 xv, float4 yv) {
        if (MASK == 1) d = mask4(fma4(xv, sc, sh), d);
        if (MASK == 2) d = mask4(yv, d);
        if (DRES) store_stream(dres + o, d);
        const float4 xh = (xv - mean) * invstd;
        store_stream(dx + o, fma4(d, sc, fma4(xh, k3, k2)));
    };
    int r = r0 + rl;
    for (; r + (UNROLL - 1) * g.RPP < r1; r += UNROLL * g.RPP) {
        float4 d[UNROLL], xv[UNROLL], yv[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const size_t o = (size_t)(r + u * g.RPP) * g.C + col;
            d[u] = *reinterpret_cast<const float4*>(dy + o);
            xv[u] = *reinterpret_cast<const float4*>(x + o);
            if (MASK == 2) yv[u] = *reinterpret_cast<const float4*>(y + o);
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) emit((size_t)(r + u * g.RPP) * g.C + col, d[u], xv[u], MASK == 2 ? yv[u] : f4(0.f));
    }
    for (; r < r1; r += g.RPP) {
        const size_t o = (size_t)r * g.C + col;
        emit(o, *reinterpret_cast<const float4*>(dy + o), *reinterpret_cast<const float4*>(x + o),
             MASK == 2 ? *reinterpret_cast<const float4*>(y + o) : f4(0.f));
    }
}

// ------------------------------------------------------------------ host
struct Plan {
    Geo g;
    dim3 grid;
    int n_split;
};
inline bool make_plan(int R, int C, int want_split, Plan& p) {
    if (R <= 0 || C <= 0 || C % 4) return false;
    const int c4 = C / 4;
    int cgb = c4 < T ? c4 : T;
    if (T % cgb || c4 % cgb) return false;  // C/4 must be a power-of-two-ish divisor layout (true for every ResNet width)
    const int rpp = T / cgb;
    const int ncb = c4 / cgb;
    int split, rows;
    if (want_split > 0) {  // caller-fixed (partials buffer already sized): trailing blocks may be empty
        split = want_split;
        rows = ((R + split - 1) / split + rpp - 1) / rpp * rpp;
    } else {               // aim for ~1024 workgroups (4 per CU), at least one unrolled pass per block
        split = 1024 / ncb;
        const int max_split = (R + rpp * UNROLL - 1) / (rpp * UNROLL);
        if (split > max_split) split = max_split;
        if (split < 1) split = 1;
        rows = ((R + split - 1) / split + rpp - 1) / rpp * rpp;
        split = (R + rows - 1) / rows;
    }
    p.g = {R, C, cgb, rpp, rows};
    p.grid = dim3(ncb, split);
    p.n_split = split;
    return true;
}

}  // namespace
}  // namespace peclr

using namespace peclr;

extern "C" int peclr_bn2d_n_split(int R, int C) {
    Plan p;
    return make_plan(R, C, 0, p) ? p.n_split : 0;
}

extern "C" int peclr_bn2d_stats_f32(const float* x, int R, int C, float* partial, int n_split, peclr_stream_t stream) {
    if (!x || !partial) return PECLR_ERR_NULL;
    Plan p;
    if (n_split < 1 || !make_plan(R, C, n_split, p)) return PECLR_ERR_SHAPE;
    if (!aligned16(x) || !aligned16(partial)) return PECLR_ERR_ALIGN;
    hipLaunchKernelGGL(bn2d_stats_kernel, p.grid, dim3(T), 0, static_cast<hipStream_t>(stream), x, p.g, partial);
    return launch_status();
}

extern "C" int peclr_bn2d_finalize_f32(const float* x, const float* partial, int n_split, int R, int C, int training,
                                       float eps, float momentum, const float* gamma, const float* beta,
                                       float* running_mean, float* running_var, int64_t* num_batches_tracked,
                                       float* save_mean, float* save_invstd, float* scale_shift, peclr_stream_t stream) {
    if (!gamma || !beta || !save_mean || !save_invstd || !scale_shift) return PECLR_ERR_NULL;
    if (R <= 0 || C <= 0) return PECLR_ERR_SHAPE;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (training) {
        if (!x || !partial || n_split < 1) return PECLR_ERR_NULL;
        hipLaunchKernelGGL(bn2d_stats_finalize_kernel, dim3((C + FC - 1) / FC), dim3(FT), 0, s, x, partial, n_split, R, C, eps,
                           momentum, gamma, beta, running_mean, running_var, num_batches_tracked, save_mean, save_invstd,
                           scale_shift);
    } else {
        if (!running_mean || !running_var) return PECLR_ERR_NULL;
        hipLaunchKernelGGL(bn2d_eval_params_kernel, dim3((C + T - 1) / T), dim3(T), 0, s, C, eps, gamma, beta, running_mean,
                           running_var, save_mean, save_invstd, scale_shift);
    }
    return launch_status();
}

extern "C" int peclr_bn2d_apply_f32(const float* x, const float* residual, int R, int C, const float* scale_shift,
                                    int relu, float* y, peclr_stream_t stream) {
    if (!x || !scale_shift || !y) return PECLR_ERR_NULL;
    Plan p;
    if (!make_plan(R, C, 0, p)) return PECLR_ERR_SHAPE;
    if (!aligned16(x) || !aligned16(y) || !aligned16(scale_shift) || (residual && !aligned16(residual))) return PECLR_ERR_ALIGN;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (residual && relu) hipLaunchKernelGGL((bn2d_apply_kernel<true, true>), p.grid, dim3(T), 0, s, x, residual, p.g, scale_shift, y);
    else if (residual) hipLaunchKernelGGL((bn2d_apply_kernel<true, false>), p.grid, dim3(T), 0, s, x, residual, p.g, scale_shift, y);
    else if (relu) hipLaunchKernelGGL((bn2d_apply_kernel<false, true>), p.grid, dim3(T), 0, s, x, residual, p.g, scale_shift, y);
    else hipLaunchKernelGGL((bn2d_apply_kernel<false, false>), p.grid, dim3(T), 0, s, x, residual, p.g, scale_shift, y);
    return launch_status();
}

static int mask_mode(int relu, const float* y) { return !relu ? 0 : (y ? 2 : 1); }

extern "C" int peclr_bn2d_bwd_reduce_f32(const float* dy, const float* x, const float* y, int R, int C, int relu,
                                         const float* save_mean, const float* save_invstd, const float* scale_shift,
                                         float* partial, int n_split, peclr_stream_t stream) {
    if (!dy || !x || !save_mean || !save_invstd || !scale_shift || !partial) return PECLR_ERR_NULL;
    Plan p;
    if (n_split < 1 || !make_plan(R, C, n_split, p)) return PECLR_ERR_SHAPE;
    if (!aligned16(dy) || !aligned16(x) || (y && !aligned16(y)) || !aligned16(partial)) return PECLR_ERR_ALIGN;
    hipStream_t s = static_cast<hipStream_t>(stream);
    switch (mask_mode(relu, y)) {
        case 0: hipLaunchKernelGGL((bn2d_bwd_reduce_kernel<0>), p.grid, dim3(T), 0, s, dy, x, y, p.g, save_mean, save_invstd, scale_shift, partial); break;
        case 1: hipLaunchKernelGGL((bn2d_bwd_reduce_kernel<1>), p.grid, dim3(T), 0, s, dy, x, y, p.g, save_mean, save_invstd, scale_shift, partial); break;
        default: hipLaunchKernelGGL((bn2d_bwd_reduce_kernel<2>), p.grid, dim3(T), 0, s, dy, x, y, p.g, save_mean, save_invstd, scale_shift, partial); break;
    }
    return launch_status();
}

extern "C" int peclr_bn2d_bwd_finalize_f32(const float* partial, int n_split, int R, int C, int training,
                                           const float* scale_shift, float* dgamma, float* dbeta, float* coef,
                                           peclr_stream_t stream) {
    if (!partial || !scale_shift || !dgamma || !dbeta || !coef) return PECLR_ERR_NULL;
    if (n_split < 1 || R <= 0 || C <= 0) return PECLR_ERR_SHAPE;
    hipLaunchKernelGGL(bn2d_bwd_finalize_kernel, dim3((C + FC - 1) / FC), dim3(FT), 0, static_cast<hipStream_t>(stream), partial,
                       n_split, R, C, training, scale_shift, dgamma, dbeta, coef);
    return launch_status();
}

extern "C" int peclr_bn2d_bwd_apply_f32(const float* dy, const float* x, const float* y, int R, int C, int relu,
                                        const float* save_mean, const float* save_invstd, const float* scale_shift,
                                        const float* coef, float* dx, float* d_residual, peclr_stream_t stream) {
    if (!dy || !x || !save_mean || !save_invstd || !scale_shift || !coef || !dx) return PECLR_ERR_NULL;
    Plan p;
    if (!make_plan(R, C, 0, p)) return PECLR_ERR_SHAPE;
    if (!aligned16(dy) || !aligned16(x) || (y && !aligned16(y)) || !aligned16(dx) || (d_residual && !aligned16(d_residual)))
        return PECLR_ERR_ALIGN;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int mm = mask_mode(relu, y);
#define LAUNCH(M, D) hipLaunchKernelGGL((bn2d_bwd_apply_kernel<M, D>), p.grid, dim3(T), 0, s, dy, x, y, p.g, save_mean, save_invstd, scale_shift, coef, dx, d_residual)
    if (d_residual) {
        if (mm == 0) LAUNCH(0, true); else if (mm == 1) LAUNCH(1, true); else LAUNCH(2, true);
    } else {
        if (mm == 0) LAUNCH(0, false); else if (mm == 1) LAUNCH(1, false); else LAUNCH(2, false);
    }
#undef LAUNCH
    return launch_status();
}
