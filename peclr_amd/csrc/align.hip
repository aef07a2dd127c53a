// K3..K7 fused: projection stats + F.normalize + translate_encodings + rotate_encoding +
// F.normalize, forward and backward (hybrid2_model.py:40-85, utils.py:271-346), with the
// split-K reduction of the second Linear fused into the load.
//
// The reference spends ~60 launches, an M x 3 x 2 matrix built ON THE CPU and a blocking
// D2H/H2D round trip here (utils.py:290,316).  This kernel is one launch: one wave per
// sample, lane l owns 2-D point l (float2 = 8 B/lane, a 512-byte coalesced row), every
// reduction over the 64 points is a wave64 butterfly, the median is a 21-stage bitonic sort
// across the lanes, and sin/cos of the (integer-degree, float64) angle are computed
// on device in float64 exactly as get_rotation_2D_matrix does.
//
// Roofline: HBM.  Algorithmic bytes per row: 512*n_slabs read + 512 (p) + 512 (z) written
// + 48 of parameters/outputs = ~1.5 KiB (n_slabs = 1), i.e. pure latency at M = 256.
#include "common.hpp"

namespace peclr {
namespace {

constexpr int D = 128;
constexpr int ROWS_PER_WG = 4;
constexpr float F_EPS = 1e-12f;
constexpr double PI = 3.14159265358979323846;

// Lower median (torch.median: sorted[(64-1)//2]) of one value per lane: bitonic sort across the 64
// lanes (21 compare-exchange stages), lane 31 then holds sorted[31].
__device__ __forceinline__ float wave_lower_median(float v, int lane) {
#pragma unroll
    for (int k = 2; k <= 64; k <<= 1) {
#pragma unroll
        for (int j = k >> 1; j > 0; j >>= 1) {
            const float o = __shfl_xor(v, j, kWave);
            const bool take_min = ((lane & k) == 0) == ((lane & j) == 0);
            v = take_min ? fminf(v, o) : fmaxf(v, o);
        }
    }
    return __shfl(v, 31, kWave);
}

// get_rotation_2D_matrix (utils.py:287-296) for rotate_encoding(projections, -angles):
// theta = (-angle) * pi / 180 in float64, entries rounded to float32.
__device__ __forceinline__ void rotation_f64(double angle_deg, double& alpha, double& beta) {
    const double theta = (-angle_deg) * PI / 180.0;
    alpha = cos(theta);
    beta = sin(theta);
}

// ROWS: rows per workgroup.  4 = one row per wave (small M: all the parallelism there is);
// 64 = each wave walks 16 rows and the float64 sin/cos of the 64 rows are computed ONCE, one row per
// lane of the first wave, and shared through LDS (they are ~40 % of the per-row instruction count).
template <int ROWS>
__global__ __launch_bounds__(256) void align_fwd_kernel(
    const float* __restrict__ p_slabs, int n_slabs, int M, int n_pairs, int flags,
    const int64_t* __restrict__ jx1, const int64_t* __restrict__ jx2, const int64_t* __restrict__ jy1,
    const int64_t* __restrict__ jy2, float extent_x, float extent_y, const double* __restrict__ ang1,
    const double* __restrict__ ang2, float* __restrict__ p_out, float* __restrict__ z_out,
    float* __restrict__ norms, float* __restrict__ row_stats) {
    __shared__ double rot[ROWS > 4 ? ROWS : 1][2];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row0 = blockIdx.x * ROWS;
    const bool rotate = (flags & PECLR_ALIGN_ROTATE) && !(flags & PECLR_ALIGN_SINGLE_NORM);
    if (ROWS > 4) {
        if (rotate && threadIdx.x < ROWS && row0 + threadIdx.x < M) {
            const int r = row0 + threadIdx.x;
            rotation_f64(r >= n_pairs ? ang2[r - n_pairs] : ang1[r], rot[threadIdx.x][0], rot[threadIdx.x][1]);
        }
        __syncthreads();
    }
    const size_t slab = (size_t)M * D;
    for (int rr = wave; rr < ROWS; rr += 4) {
        const int row = row0 + rr;
        if (row >= M) break;  // wave-uniform
        const size_t off = (size_t)row * D + 2 * lane;
        float2 v = *reinterpret_cast<const float2*>(p_slabs + off);
        for (int k = 1; k < n_slabs; ++k) {
            const float2 t = *reinterpret_cast<const float2*>(p_slabs + k * slab + off);
            v.x += t.x;
            v.y += t.y;
        }
        *reinterpret_cast<float2*>(p_out + off) = v;

        // reductions over the 64 points of the RAW projection; min/max/sum are reused below
        const float sx = wave_sum(v.x), sy = wave_sum(v.y);
        const float mnx = wave_min(v.x), mny = wave_min(v.y);
        const float mxx = wave_max(v.x), mxy = wave_max(v.y);
        if (row_stats) {  // hybrid2_model.py:92-106, per-sample part
            const float mdx = wave_lower_median(v.x, lane), mdy = wave_lower_median(v.y, lane);
            if (lane == 0) {
                float4* o = reinterpret_cast<float4*>(row_stats + (size_t)row * 8);
                o[0] = make_float4(sx * (1.f / 64.f), mdx, mnx, mxx);
                o[1] = make_float4(sy * (1.f / 64.f), mdy, mny, mxy);
            }
        }

        // first F.normalize over all 128 dims (hybrid2_model.py:48-49)
        const float n1 = fmaxf(sqrtf(wave_sum(v.x * v.x + v.y * v.y)), F_EPS);
        float x = v.x / n1, y = v.y / n1;
        if (lane == 0) norms[row] = n1;

        if (flags & PECLR_ALIGN_SINGLE_NORM) {
            *reinterpret_cast<float2*>(z_out + off) = make_float2(x, y);
            if (lane == 0) norms[M + row] = 1.f;
            continue;
        }
        const bool second = row >= n_pairs;
        const int s = second ? row - n_pairs : row;

        float tx = 0.f, ty = 0.f;
        if (flags & PECLR_ALIGN_CROP) {  // translate_encodings(q, -jx/H, -jy/W) (utils.py:338-346)
            // q = p / n1 with n1 > 0 is monotone, so max(q) = fl(max(p) / n1) exactly: the range of the
            // normalised points comes from the raw min/max without another pair of reductions
            tx = -((float)(second ? jx2[s] : jx1[s]) / extent_x) * (mxx / n1 - mnx / n1);
            ty = -((float)(second ? jy2[s] : jy1[s]) / extent_y) * (mxy / n1 - mny / n1);
            x += tx;
            y += ty;
        }
        if (rotate) {  // rotate_encoding(q, -angle) (utils.py:312-320), about the centroid AFTER the shift
            const float cx = wave_sum(x) * (1.f / 64.f), cy = wave_sum(y) * (1.f / 64.f);
            double alpha, beta;
            if (ROWS > 4) {
                alpha = rot[rr][0];
                beta = rot[rr][1];
            } else {
                rotation_f64(second ? ang2[s] : ang1[s], alpha, beta);
            }
            const float r00 = (float)alpha, r10 = (float)beta, r01 = (float)(-beta), r11 = (float)alpha;
            const float r20 = (float)((1.0 - alpha) * (double)cx - beta * (double)cy);
            const float r21 = (float)((1.0 - alpha) * (double)cy + beta * (double)cx);
            const float nx = fmaf(y, r10, x * r00) + r20;
            const float ny = fmaf(y, r11, x * r01) + r21;
            x = nx;
            y = ny;
        }
        // second F.normalize (hybrid2_model.py:83-84)
        const float n2 = fmaxf(sqrtf(wave_sum(x * x + y * y)), F_EPS);
        *reinterpret_cast<float2*>(z_out + off) = make_float2(x / n2, y / n2);
        if (lane == 0) norms[M + row] = n2;
    }
}

// y = x / max(|x|, eps) backward for one row held one float2 per lane.
__device__ __forceinline__ float2 normalize_bwd(float2 dy, float2 yv, float nc) {
    if (nc <= F_EPS) return make_float2(dy.x / nc, dy.y / nc);
    const float dot = wave_sum(dy.x * yv.x + dy.y * yv.y);
    return make_float2((dy.x - yv.x * dot) / nc, (dy.y - yv.y * dot) / nc);
}

__global__ __launch_bounds__(256) void align_bwd_kernel(
    const float* __restrict__ dz, const float* __restrict__ p, const float* __restrict__ z,
    const float* __restrict__ norms, int M, int n_pairs, int flags, const double* __restrict__ ang1,
    const double* __restrict__ ang2, float* __restrict__ dp) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * ROWS_PER_WG + (threadIdx.x >> 6);
    if (row >= M) return;
    const size_t off = (size_t)row * D + 2 * lane;
    float2 g = *reinterpret_cast<const float2*>(dz + off);
    const float2 pv = *reinterpret_cast<const float2*>(p + off);
    const float n1 = norms[row];
    const float2 q = make_float2(pv.x / n1, pv.y / n1);
    if (!(flags & PECLR_ALIGN_SINGLE_NORM)) {
        const float2 zv = *reinterpret_cast<const float2*>(z + off);
        g = normalize_bwd(g, zv, norms[M + row]);
        if (flags & PECLR_ALIGN_ROTATE) {  // centroid is a constant: only the 2x2 part
            const bool second = row >= n_pairs;
            double alpha, beta;
            rotation_f64(second ? ang2[row - n_pairs] : ang1[row], alpha, beta);
            const float a = (float)alpha, b = (float)beta;
            g = make_float2(a * g.x - b * g.y, b * g.x + a * g.y);
        }
        // translate: identity (range is a constant)
    }
    g = normalize_bwd(g, q, n1);
    *reinterpret_cast<float2*>(dp + off) = g;
}

}  // namespace
}  // namespace peclr

using namespace peclr;

extern "C" int peclr_align_fwd_f32(const float* p_slabs, int n_slabs, int M, int Dd, int n_pairs, int flags,
                                   const int64_t* jitter_x1, const int64_t* jitter_x2,
                                   const int64_t* jitter_y1, const int64_t* jitter_y2, float extent_x,
                                   float extent_y, const double* angle1, const double* angle2, float* p_out,
                                   float* z_out, float* norms, float* row_stats, peclr_stream_t stream) {
    if (!p_slabs || !p_out || !z_out || !norms) return PECLR_ERR_NULL;
    if (Dd != D || M <= 0 || n_slabs < 1 || n_pairs < 0 || n_pairs > M) return PECLR_ERR_SHAPE;
    if (flags & ~(PECLR_ALIGN_CROP | PECLR_ALIGN_ROTATE | PECLR_ALIGN_SINGLE_NORM)) return PECLR_ERR_UNSUPPORTED;
    if (!(flags & PECLR_ALIGN_SINGLE_NORM)) {
        if ((flags & PECLR_ALIGN_CROP) && (!jitter_x1 || !jitter_x2 || !jitter_y1 || !jitter_y2))
            return PECLR_ERR_NULL;
        if ((flags & PECLR_ALIGN_CROP) && (!(extent_x > 0.f) || !(extent_y > 0.f))) return PECLR_ERR_SHAPE;
        if ((flags & PECLR_ALIGN_ROTATE) && (!angle1 || !angle2)) return PECLR_ERR_NULL;
        if ((flags & (PECLR_ALIGN_CROP | PECLR_ALIGN_ROTATE)) && M != 2 * n_pairs) return PECLR_ERR_SHAPE;
    }
    if (!aligned16(p_slabs) || !aligned16(p_out) || !aligned16(z_out) || (row_stats && !aligned16(row_stats)))
        return PECLR_ERR_ALIGN;
    hipStream_t st = static_cast<hipStream_t>(stream);
    // share the float64 sin/cos across a workgroup's rows only while >= 2048 workgroups remain
    if (M >= 131072)
        hipLaunchKernelGGL((align_fwd_kernel<64>), dim3((M + 63) / 64), dim3(256), 0, st, p_slabs, n_slabs, M, n_pairs,
                           flags, jitter_x1, jitter_x2, jitter_y1, jitter_y2, extent_x, extent_y, angle1, angle2,
                           p_out, z_out, norms, row_stats);
    else if (M >= 32768)
        hipLaunchKernelGGL((align_fwd_kernel<16>), dim3((M + 15) / 16), dim3(256), 0, st, p_slabs, n_slabs, M, n_pairs,
                           flags, jitter_x1, jitter_x2, jitter_y1, jitter_y2, extent_x, extent_y, angle1, angle2,
                           p_out, z_out, norms, row_stats);
    else
        hipLaunchKernelGGL((align_fwd_kernel<ROWS_PER_WG>), dim3((M + ROWS_PER_WG - 1) / ROWS_PER_WG), dim3(256), 0,
                           st, p_slabs, n_slabs, M, n_pairs, flags, jitter_x1, jitter_x2, jitter_y1, jitter_y2,
                           extent_x, extent_y, angle1, angle2, p_out, z_out, norms, row_stats);
    return launch_status();
}

extern "C" int peclr_align_bwd_f32(const float* dz, const float* p, const float* z, const float* norms, int M,
                                   int Dd, int n_pairs, int flags, const double* angle1, const double* angle2,
                                   float* dp, peclr_stream_t stream) {
    if (!dz || !p || !z || !norms || !dp) return PECLR_ERR_NULL;
    if (Dd != D || M <= 0) return PECLR_ERR_SHAPE;
    if (!(flags & PECLR_ALIGN_SINGLE_NORM) && (flags & PECLR_ALIGN_ROTATE)) {
        if (!angle1 || !angle2) return PECLR_ERR_NULL;
        if (M != 2 * n_pairs) return PECLR_ERR_SHAPE;
    }
    if (!aligned16(dz) || !aligned16(p) || !aligned16(z) || !aligned16(dp)) return PECLR_ERR_ALIGN;
    hipLaunchKernelGGL(align_bwd_kernel, dim3((M + ROWS_PER_WG - 1) / ROWS_PER_WG), dim3(256), 0,
                       static_cast<hipStream_t>(stream), dz, p, z, norms, M, n_pairs, flags, angle1, angle2, dp);
    return launch_status();
}
