// Fused multi-tensor LARSWrapper(Adam) step (BaseModel.configure_optimizers, base_model.py:57-104;
// pl_bolts 0.2.2 LARSWrapper.step/update_p + torch.optim.Adam).
//
// The reference walks ~160 tensors in Python, 2 torch.norm + 2 host syncs (`if p_norm != 0`)
// + ~12 elementwise launches per tensor.  Here one parameter group is two launches:
//   1. per-chunk sums of squares of p and g  -> norms_ws[2][n_chunks]
//   2. per-chunk update: combine the tensor's chunk sums in a fixed order (bit-reproducible),
//      LARS trust ratio, weight decay, Adam moments and the parameter update in registers.
// Roofline: HBM.  Algorithmic bytes per parameter: 8 (norm pass) + 16 read + 12 written = 36 B.
//
// precision=16 (the reference's default: Lightning native AMP = torch GradScaler around the optimiser,
// peclr_training.py:78-79): the `_amp` entry points fold the scaler into the same two launches -- pass 1 also
// looks for a non-finite gradient, pass 2 multiplies by 1/scale on load, returns early when pass 1 found one
// (GradScaler.step's skip) and takes the bias corrections from a DEVICE count of the steps actually taken; a
// one-thread third launch is GradScaler.update.  No host branch: the step stays capturable in a hipGraph.
#include "common.hpp"

namespace peclr {
namespace {

constexpr int CHUNK = PECLR_OPT_CHUNK;  // 256 threads x 16 elements

__device__ __forceinline__ bool aligned16_dev(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) red[wave] = v;
    __syncthreads();
    const float t = red[0] + red[1] + red[2] + red[3];
    __syncthreads();
    return t;
}

__device__ __forceinline__ int non_finite(float x) { return (__float_as_uint(x) & 0x7f800000u) == 0x7f800000u; }

// AMP: the gradients hold scale * g.  The norms are those of the unscaled g (what LARS sees after
// GradScaler.unscale_), and amp->found_inf is raised if any element is inf / nan (torch's
// _amp_foreach_non_finite_check_and_unscale_: element test, not a test of the sum).
template <bool AMP>
__global__ __launch_bounds__(256) void sumsq_kernel(float* const* __restrict__ ptrs,
                                                    const int64_t* __restrict__ sizes, int n_tensors,
                                                    const int32_t* __restrict__ chunk_tensor,
                                                    const int64_t* __restrict__ chunk_offset, int n_chunks,
                                                    float* __restrict__ norms_ws, peclr_amp_state* amp) {
    __shared__ float red[4];
    const float inv = AMP ? 1.f / amp->scale : 1.f;
    int bad = 0;
    const int c = blockIdx.x;
    const int t = chunk_tensor[c];
    const int64_t off = chunk_offset[c];
    const int64_t n = min((int64_t)CHUNK, sizes[t] - off);
    const float* p = ptrs[t] + off;
    const float* g = ptrs[n_tensors + t] + off;
    float sp = 0.f, sg = 0.f;
    if (n == CHUNK && aligned16_dev(p) && aligned16_dev(g)) {
        // full chunk: 4 float4 per thread per tensor, all 8 loads in flight before the first FMA
        float4 a[4], b[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            a[j] = reinterpret_cast<const float4*>(p)[threadIdx.x + 256 * j];
            b[j] = reinterpret_cast<const float4*>(g)[threadIdx.x + 256 * j];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            sp += a[j].x * a[j].x + a[j].y * a[j].y + a[j].z * a[j].z + a[j].w * a[j].w;
            if (AMP) {
                bad |= non_finite(b[j].x) | non_finite(b[j].y) | non_finite(b[j].z) | non_finite(b[j].w);
                b[j].x *= inv; b[j].y *= inv; b[j].z *= inv; b[j].w *= inv;
            }
            sg += b[j].x * b[j].x + b[j].y * b[j].y + b[j].z * b[j].z + b[j].w * b[j].w;
        }
    } else {
        for (int64_t k = threadIdx.x; k < n; k += 256) {
            const float a = p[k];
            float b = g[k];
            if (AMP) {
                bad |= non_finite(b);
                b *= inv;
            }
            sp += a * a;
            sg += b * b;
        }
    }
    sp = block_sum(sp, red);
    sg = block_sum(sg, red);
    if (AMP) bad = __syncthreads_or(bad);
    if (threadIdx.x == 0) {
        norms_ws[c] = sp;
        norms_ws[n_chunks + c] = sg;
        if (AMP && bad) amp->found_inf = 1.f;   // every writer stores the same value
    }
}

constexpr int MAX_GROUPS = 8;
struct OptArgs {
    float lr[MAX_GROUPS], weight_decay[MAX_GROUPS];  // per parameter group
    float beta1, beta2, adam_eps, bias_corr1, bias_corr2, lars_eta, lars_eps;
    int use_lars, lars_clip;
    double beta1_d, beta2_d;  // AMP: bias corrections are computed on the device, in the host's precision
};

// Streaming stores: the moments and the parameter are written once per step and not re-read by this
// kernel; on MI355X a 4-read/3-write stream reaches 6.5-6.6 TB/s with non-temporal stores vs 5.2 TB/s
// with plain ones (tools/exp/stream_4r3w.hip, random data).
typedef float v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store_nt(float4* a, float4 x) {
    v4f t = {x.x, x.y, x.z, x.w};
    __builtin_nontemporal_store(t, reinterpret_cast<v4f*>(a));
}

template <bool AMP>
__global__ __launch_bounds__(256) void lars_adam_kernel(float* const* __restrict__ ptrs,
                                                        const int64_t* __restrict__ sizes, int n_tensors,
                                                        const int32_t* __restrict__ chunk_tensor,
                                                        const int64_t* __restrict__ chunk_offset,
                                                        const int32_t* __restrict__ tensor_chunk_begin,
                                                        const int32_t* __restrict__ tensor_group,
                                                        int n_chunks, const float* __restrict__ norms_ws,
                                                        const float* __restrict__ device_hyper, OptArgs a,
                                                        const peclr_amp_state* amp) {
    const int c = blockIdx.x;
    const int t = chunk_tensor[c];
    const int grp = tensor_group ? tensor_group[t] : 0;
    // per-step scalars either by value (kernel arguments) or from device memory, so that a captured
    // hipGraph can be replayed with a new learning rate / bias correction every step
    // (read into locals: writing the by-value argument struct would demote it to scratch memory)
    const float lr = device_hyper ? device_hyper[grp] : a.lr[grp];
    const float wd_group = device_hyper ? device_hyper[MAX_GROUPS + grp] : a.weight_decay[grp];
    float bias_corr1 = device_hyper ? device_hyper[2 * MAX_GROUPS] : a.bias_corr1;
    float bias_corr2 = device_hyper ? device_hyper[2 * MAX_GROUPS + 1] : a.bias_corr2;
    const int64_t off = chunk_offset[c];
    const int64_t n = min((int64_t)CHUNK, sizes[t] - off);
    float* p = ptrs[t] + off;
    const float* g = ptrs[n_tensors + t] + off;
    float* m = ptrs[2 * n_tensors + t] + off;
    float* v = ptrs[3 * n_tensors + t] + off;
    const float inv = AMP ? 1.f / amp->scale : 1.f;
    if (AMP) {
        if (amp->found_inf != 0.f) {
            // GradScaler.step: no optimiser step.  The reference's wrapper has unscaled p.grad in place by
            // then; only the write-back mode exposes that, so only it pays for the pass.
            if (a.use_lars == 2) {
                float* gw = const_cast<float*>(g);
                for (int64_t k = threadIdx.x; k < n; k += 256) gw[k] *= inv;
            }
            return;
        }
        // the host cannot know how many steps were skipped: the count of steps TAKEN lives next to the scale
        const double step = (double)(amp->good_steps + 1);
        bias_corr1 = (float)(1.0 - pow(a.beta1_d, step));
        bias_corr2 = (float)(1.0 - pow(a.beta2_d, step));
    }

    float trust = 1.f, wd = wd_group;
    if (a.use_lars) {
        // the workgroup combines the tensor's chunk sums cooperatively, always in the same tree order
        // (every chunk of a tensor gets bit-identical norms); a serial per-thread loop over up to
        // ~600 chunks of a big conv weight cost more than the update itself
        __shared__ float red[4];
        float sp = 0.f, sg = 0.f;
        for (int k = tensor_chunk_begin[t] + threadIdx.x; k < tensor_chunk_begin[t + 1]; k += 256) {
            sp += norms_ws[k];
            sg += norms_ws[n_chunks + k];
        }
        sp = block_sum(sp, red);
        sg = block_sum(sg, red);
        const float pn = sqrtf(sp), gn = sqrtf(sg);
        if (pn != 0.f && gn != 0.f) {
            trust = a.lars_eta * pn / (gn + pn * wd + a.lars_eps);
            if (a.lars_clip) trust = fminf(trust / lr, 1.f);
        } else {
            wd = 0.f;  // update_p leaves the gradient untouched
        }
    }
    const float step_size = lr / bias_corr1;
    const float inv_bc2_sqrt = 1.f / sqrtf(bias_corr2);
    const float b1 = a.beta1, b2 = a.beta2, eps = a.adam_eps;
    const bool write_back = a.use_lars == 2;   // LARSWrapper.update_p mutates p.grad in place
    float* gw = const_cast<float*>(g);
    auto upd = [&](float& pk, float& gk, float& mk, float& vk) {
        if (AMP) gk *= inv;
        gk = (gk + wd * pk) * trust;
        mk = b1 * mk + (1.f - b1) * gk;
        vk = b2 * vk + (1.f - b2) * gk * gk;
        pk = pk - step_size * (mk / (sqrtf(vk) * inv_bc2_sqrt + eps));
    };
    if (n == CHUNK && aligned16_dev(p) && aligned16_dev(g) && aligned16_dev(m) && aligned16_dev(v)) {
        // full chunk: 16 x 16-byte loads in flight per thread (1 KiB per wave-instruction), then the
        // update in registers and 12 x 16-byte stores
        float4 pp[4], gg[4], mm[4], vv[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int k = threadIdx.x + 256 * j;
            pp[j] = reinterpret_cast<const float4*>(p)[k];
            gg[j] = reinterpret_cast<const float4*>(g)[k];
            mm[j] = reinterpret_cast<const float4*>(m)[k];
            vv[j] = reinterpret_cast<const float4*>(v)[k];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int k = threadIdx.x + 256 * j;
            upd(pp[j].x, gg[j].x, mm[j].x, vv[j].x);
            upd(pp[j].y, gg[j].y, mm[j].y, vv[j].y);
            upd(pp[j].z, gg[j].z, mm[j].z, vv[j].z);
            upd(pp[j].w, gg[j].w, mm[j].w, vv[j].w);
            store_nt(reinterpret_cast<float4*>(m) + k, mm[j]);
            store_nt(reinterpret_cast<float4*>(v) + k, vv[j]);
            store_nt(reinterpret_cast<float4*>(p) + k, pp[j]);
            if (write_back) store_nt(reinterpret_cast<float4*>(gw) + k, gg[j]);
        }
    } else {
        for (int64_t k = threadIdx.x; k < n; k += 256) {
            float pk = p[k], gk = g[k], mk = m[k], vk = v[k];
            upd(pk, gk, mk, vk);
            if (write_back) gw[k] = gk;
            m[k] = mk;
            v[k] = vk;
            p[k] = pk;
        }
    }
}

// torch.amp.GradScaler.update (= _amp_update_scale_) + the count of optimiser steps actually taken
__global__ void amp_update_kernel(peclr_amp_state* amp, float growth_factor, float backoff_factor,
                                  int growth_interval) {
    if (amp->found_inf != 0.f) {
        amp->scale *= backoff_factor;
        amp->growth_tracker = 0;
    } else {
        amp->good_steps += 1;
        const int ok = amp->growth_tracker + 1;
        if (ok == growth_interval) {
            const float grown = amp->scale * growth_factor;
            if (!non_finite(grown)) amp->scale = grown;
            amp->growth_tracker = 0;
        } else {
            amp->growth_tracker = ok;
        }
    }
    amp->found_inf = 0.f;
}

}  // namespace
}  // namespace peclr

using namespace peclr;

static int sumsq_launch(float* const* ptrs, const int64_t* sizes, int n_tensors, const int32_t* chunk_tensor,
                        const int64_t* chunk_offset, int n_chunks, float* norms_ws, peclr_amp_state* amp,
                        peclr_stream_t stream) {
    if (!ptrs || !sizes || !chunk_tensor || !chunk_offset || !norms_ws) return PECLR_ERR_NULL;
    if (n_tensors <= 0 || n_chunks <= 0) return PECLR_ERR_SHAPE;
    if (amp)
        hipLaunchKernelGGL(sumsq_kernel<true>, dim3(n_chunks), dim3(256), 0, static_cast<hipStream_t>(stream), ptrs,
                           sizes, n_tensors, chunk_tensor, chunk_offset, n_chunks, norms_ws, amp);
    else
        hipLaunchKernelGGL(sumsq_kernel<false>, dim3(n_chunks), dim3(256), 0, static_cast<hipStream_t>(stream), ptrs,
                           sizes, n_tensors, chunk_tensor, chunk_offset, n_chunks, norms_ws, amp);
    return launch_status();
}

extern "C" int peclr_lars_sumsq_f32(float* const* ptrs, const int64_t* sizes, int n_tensors,
                                    const int32_t* chunk_tensor, const int64_t* chunk_offset, int n_chunks,
                                    float* norms_ws, peclr_stream_t stream) {
    return sumsq_launch(ptrs, sizes, n_tensors, chunk_tensor, chunk_offset, n_chunks, norms_ws, nullptr, stream);
}

extern "C" int peclr_lars_sumsq_amp_f32(float* const* ptrs, const int64_t* sizes, int n_tensors,
                                        const int32_t* chunk_tensor, const int64_t* chunk_offset, int n_chunks,
                                        float* norms_ws, peclr_amp_state* amp, peclr_stream_t stream) {
    if (!amp) return PECLR_ERR_NULL;
    return sumsq_launch(ptrs, sizes, n_tensors, chunk_tensor, chunk_offset, n_chunks, norms_ws, amp, stream);
}

extern "C" int peclr_amp_update(peclr_amp_state* amp, float growth_factor, float backoff_factor,
                                int growth_interval, peclr_stream_t stream) {
    if (!amp) return PECLR_ERR_NULL;
    if (!(growth_factor > 1.f) || !(backoff_factor > 0.f && backoff_factor < 1.f) || growth_interval < 1)
        return PECLR_ERR_SHAPE;
    hipLaunchKernelGGL(amp_update_kernel, dim3(1), dim3(1), 0, static_cast<hipStream_t>(stream), amp, growth_factor,
                       backoff_factor, growth_interval);
    return launch_status();
}

static int update_launch(float* const* ptrs, const int64_t* sizes, int n_tensors, const int32_t* chunk_tensor,
                         const int64_t* chunk_offset, const int32_t* tensor_chunk_begin, const int32_t* tensor_group,
                         int n_chunks, const float* norms_ws, const float* device_hyper, const float* group_lr,
                         const float* group_weight_decay, int n_groups, double beta1, double beta2, float adam_eps,
                         float bias_corr1, float bias_corr2, int use_lars, float lars_eta, float lars_eps,
                         int lars_clip, const peclr_amp_state* amp, peclr_stream_t stream) {
    if (!ptrs || !sizes || !chunk_tensor || !chunk_offset || !tensor_chunk_begin) return PECLR_ERR_NULL;
    if (!device_hyper && (!group_lr || !group_weight_decay)) return PECLR_ERR_NULL;
    if (use_lars && !norms_ws) return PECLR_ERR_NULL;
    if (n_tensors <= 0 || n_chunks <= 0 || n_groups < 1 || n_groups > MAX_GROUPS) return PECLR_ERR_SHAPE;
    if (n_groups > 1 && !tensor_group) return PECLR_ERR_NULL;
    if (!device_hyper && !amp && (!(bias_corr1 > 0.f) || !(bias_corr2 > 0.f))) return PECLR_ERR_SHAPE;
    OptArgs a = {};
    for (int g = 0; g < n_groups && !device_hyper; ++g) {
        a.lr[g] = group_lr[g];
        a.weight_decay[g] = group_weight_decay[g];
    }
    a.beta1 = (float)beta1; a.beta2 = (float)beta2; a.beta1_d = beta1; a.beta2_d = beta2; a.adam_eps = adam_eps; a.bias_corr1 = bias_corr1; a.bias_corr2 = bias_corr2;
    a.lars_eta = lars_eta; a.lars_eps = lars_eps; a.use_lars = use_lars; a.lars_clip = lars_clip;
    if (amp)
        hipLaunchKernelGGL(lars_adam_kernel<true>, dim3(n_chunks), dim3(256), 0, static_cast<hipStream_t>(stream),
                           ptrs, sizes, n_tensors, chunk_tensor, chunk_offset, tensor_chunk_begin, tensor_group,
                           n_chunks, norms_ws, device_hyper, a, amp);
    else
        hipLaunchKernelGGL(lars_adam_kernel<false>, dim3(n_chunks), dim3(256), 0, static_cast<hipStream_t>(stream),
                           ptrs, sizes, n_tensors, chunk_tensor, chunk_offset, tensor_chunk_begin, tensor_group,
                           n_chunks, norms_ws, device_hyper, a, amp);
    return launch_status();
}

extern "C" int peclr_lars_adam_update_f32(float* const* ptrs, const int64_t* sizes, int n_tensors,
                                          const int32_t* chunk_tensor, const int64_t* chunk_offset,
                                          const int32_t* tensor_chunk_begin, const int32_t* tensor_group,
                                          int n_chunks, const float* norms_ws, const float* device_hyper,
                                          const float* group_lr, const float* group_weight_decay, int n_groups,
                                          float beta1,
                                          float beta2, float adam_eps, float bias_corr1, float bias_corr2,
                                          int use_lars, float lars_eta, float lars_eps, int lars_clip,
                                          peclr_stream_t stream) {
    return update_launch(ptrs, sizes, n_tensors, chunk_tensor, chunk_offset, tensor_chunk_begin, tensor_group, n_chunks,
                         norms_ws, device_hyper, group_lr, group_weight_decay, n_groups, beta1, beta2, adam_eps,
                         bias_corr1, bias_corr2, use_lars, lars_eta, lars_eps, lars_clip, nullptr, stream);
}

extern "C" int peclr_lars_adam_update_amp_f32(float* const* ptrs, const int64_t* sizes, int n_tensors,
                                              const int32_t* chunk_tensor, const int64_t* chunk_offset,
                                              const int32_t* tensor_chunk_begin, const int32_t* tensor_group,
                                              int n_chunks, const float* norms_ws, const float* device_hyper,
                                              const float* group_lr, const float* group_weight_decay, int n_groups,
                                              double beta1, double beta2, float adam_eps, int use_lars,
                                              float lars_eta, float lars_eps, int lars_clip,
                                              const peclr_amp_state* amp,
                                              peclr_stream_t stream) {
    if (!amp) return PECLR_ERR_NULL;
    return update_launch(ptrs, sizes, n_tensors, chunk_tensor, chunk_offset, tensor_chunk_begin, tensor_group, n_chunks,
                         norms_ws, device_hyper, group_lr, group_weight_decay, n_groups, beta1, beta2, adam_eps, 1.f,
                         1.f, use_lars, lars_eta, lars_eps, lars_clip, amp, stream);
}
