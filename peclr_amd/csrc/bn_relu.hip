// K2: BatchNorm1d (train/eval) + ReLU of the projection head, forward and backward
// (nn.BatchNorm1d(H) + nn.ReLU at simclr_model.py:27-28), with the split-K reduction of the
// first Linear and its bias fused into the load.
//
// Roofline: HBM (really L2 at these sizes): algorithmic bytes fwd = 4*M*H*(n_slabs + 2),
// bwd = 4*M*H*3.  One workgroup owns 32 feature columns and all M rows: 256 threads =
// 32 columns x 8 row slices, so each half-wave reads one 128-byte row segment (coalesced),
// column statistics are an in-register strided sum + an 8-way LDS tree in a FIXED order
// (deterministic), and the two-pass variance re-reads the workgroup's own a_pre column
// block from L2.  Statistics follow torch: biased variance for normalisation, unbiased for
// running_var, momentum update, eps inside the sqrt.
#include "common.hpp"

namespace peclr {
namespace {

constexpr int COLS = 32, SLICES = 8;

__device__ __forceinline__ float slice_reduce(float v, float (*red)[COLS], int c, int s) {
    red[s][c] = v;
    __syncthreads();
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < SLICES; ++k) t += red[k][c];
    __syncthreads();
    return t;
}

__global__ __launch_bounds__(256) void bn_relu_fwd_kernel(
    const float* __restrict__ slabs, int n_slabs, const float* __restrict__ bias, int M, int H,
    const float* __restrict__ gamma, const float* __restrict__ beta, float eps, float momentum, int training,
    float* running_mean, float* running_var, int64_t* nbt, float* __restrict__ a_pre,
    float* __restrict__ a_out, float* __restrict__ save_mean, float* __restrict__ save_invstd) {
    __shared__ float red[SLICES][COLS];
    const int c = threadIdx.x & (COLS - 1), s = threadIdx.x >> 5;
    const int col = blockIdx.x * COLS + c;
    const bool ok = col < H;
    const size_t slab = (size_t)M * H;
    const float bv = (ok && bias) ? bias[col] : 0.f;

    // pass 1: reduce slabs (+bias) -> a_pre, column sum
    float sum = 0.f;
    if (ok) {
        for (int r = s; r < M; r += SLICES) {
            const size_t o = (size_t)r * H + col;
            float v = slabs[o];
            for (int k = 1; k < n_slabs; ++k) v += slabs[k * slab + o];
            v += bv;
            a_pre[o] = v;
            sum += v;
        }
    }
    float mean, invstd;
    if (training) {
        mean = slice_reduce(sum, red, c, s) / (float)M;
        float ss = 0.f;
        if (ok)
            for (int r = s; r < M; r += SLICES) {  // each thread re-reads its own writes
                const float d = a_pre[(size_t)r * H + col] - mean;
                ss += d * d;
            }
        const float var = slice_reduce(ss, red, c, s) / (float)M;
        invstd = 1.0f / sqrtf(var + eps);
        if (ok && s == 0) {
            if (running_mean) running_mean[col] = (1.f - momentum) * running_mean[col] + momentum * mean;
            if (running_var) {
                const float unbiased = var * ((float)M / (float)(M > 1 ? M - 1 : 1));
                running_var[col] = (1.f - momentum) * running_var[col] + momentum * unbiased;
            }
        }
        if (nbt && blockIdx.x == 0 && threadIdx.x == 0) *nbt += 1;
    } else {
        mean = ok ? running_mean[col] : 0.f;
        invstd = ok ? 1.0f / sqrtf(running_var[col] + eps) : 0.f;
    }
    if (!ok) return;
    if (s == 0) {
        save_mean[col] = mean;
        save_invstd[col] = invstd;
    }
    const float gm = gamma[col], bt = beta[col];
    for (int r = s; r < M; r += SLICES) {
        const size_t o = (size_t)r * H + col;
        const float y = (a_pre[o] - mean) * invstd * gm + bt;
        a_out[o] = fmaxf(y, 0.f);
    }
}

__global__ __launch_bounds__(256) void bn_relu_bwd_kernel(
    const float* __restrict__ da, const float* __restrict__ a_pre, const float* __restrict__ save_mean,
    const float* __restrict__ save_invstd, const float* __restrict__ gamma, const float* __restrict__ beta,
    int M, int H, int training, float* __restrict__ d_a_pre, float* __restrict__ dgamma,
    float* __restrict__ dbeta, float* __restrict__ dbias) {
    __shared__ float red[SLICES][COLS];
    const int c = threadIdx.x & (COLS - 1), s = threadIdx.x >> 5;
    const int col = blockIdx.x * COLS + c;
    const bool ok = col < H;
    const float mean = ok ? save_mean[col] : 0.f, invstd = ok ? save_invstd[col] : 0.f;
    const float gm = ok ? gamma[col] : 0.f, bt = ok ? beta[col] : 0.f;

    float sb = 0.f, sg = 0.f;
    if (ok)
        for (int r = s; r < M; r += SLICES) {
            const size_t o = (size_t)r * H + col;
            const float xhat = (a_pre[o] - mean) * invstd;
            const float dy = (xhat * gm + bt) > 0.f ? da[o] : 0.f;
            sb += dy;
            sg += dy * xhat;
        }
    const float db = slice_reduce(sb, red, c, s);
    const float dg = slice_reduce(sg, red, c, s);
    const float k = gm * invstd / (float)M;
    float sbias = 0.f;
    if (ok)
        for (int r = s; r < M; r += SLICES) {
            const size_t o = (size_t)r * H + col;
            const float xhat = (a_pre[o] - mean) * invstd;
            const float dy = (xhat * gm + bt) > 0.f ? da[o] : 0.f;
            const float dx = training ? k * ((float)M * dy - db - xhat * dg) : gm * invstd * dy;
            d_a_pre[o] = dx;
            sbias += dx;
        }
    const float dbi = slice_reduce(sbias, red, c, s);
    if (ok && s == 0) {
        dgamma[col] = dg;
        dbeta[col] = db;
        if (dbias) dbias[col] = dbi;
    }
}

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
    hipLaunchKernelGGL(bn_relu_fwd_kernel, dim3((H + COLS - 1) / COLS), dim3(256), 0,
                       static_cast<hipStream_t>(stream), a_slabs, n_slabs, bias, M, H, gamma, beta, eps,
                       momentum, training, running_mean, running_var, num_batches_tracked, a_pre, a_out,
                       save_mean, save_invstd);
    return launch_status();
}

extern "C" int peclr_bn_relu_bwd_f32(const float* d_a_out, const float* a_pre, const float* save_mean,
                                     const float* save_invstd, const float* gamma, const float* beta, int M,
                                     int H, int training, float* d_a_pre, float* dgamma, float* dbeta,
                                     float* dbias, peclr_stream_t stream) {
    if (!d_a_out || !a_pre || !save_mean || !save_invstd || !gamma || !beta || !d_a_pre || !dgamma || !dbeta)
        return PECLR_ERR_NULL;
    if (M <= 0 || H <= 0) return PECLR_ERR_SHAPE;
    hipLaunchKernelGGL(bn_relu_bwd_kernel, dim3((H + COLS - 1) / COLS), dim3(256), 0,
                       static_cast<hipStream_t>(stream), d_a_out, a_pre, save_mean, save_invstd, gamma, beta,
                       M, H, training, d_a_pre, dgamma, dbeta, dbias);
    return launch_status();
}
