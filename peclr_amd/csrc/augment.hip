// Two-view augmentation on the device: the pixel side of
// SampleAugmenter.transform_sample (reference src/data_loader/sample_augmenter.py:47-129) for the
// published recipe -- rotate (cv2.warpAffine), crop, resize (cv2.resize INTER_AREA), colour jitter
// (cv2 BGR<->HSV) -- followed by ToTensor + Normalize (src/data_loader/utils.py:283-293).
//
// The reference runs this per sample on CPU workers; here a whole batch and both views are two
// launches.  8-bit intermediate images are kept between the stages exactly where the reference has
// them (after the rotation, after the resize, around the HSV round trip), so the arithmetic is the
// integer / float32 arithmetic of oracle/augment_oracle.py and the outputs are bit-identical to it.
//
//   warp_crop_kernel          source images [B][H][W][3] u8 -> the crop window of the rotated image,
//                             [V][B][H][W][3] u8 scratch (only the window is written)
//   resize_color_norm_kernel  crop window -> out_h x out_w (area / integer-box / area-mode bilinear,
//                             chosen per sample like cv::resize does) -> HSV jitter -> normalised
//                             float32, NCHW or NHWC
//
// Work per batch is tiny (B=128: ~40 MB read, ~50 MB written): these kernels exist to take the
// cv2-on-CPU producer off the critical path, not to approach a roofline.  One thread per pixel,
// consecutive lanes = consecutive x, so the float32 NHWC / NCHW stores coalesce.
#include "common.hpp"

#pragma clang fp contract(off)  // products and sums round separately, as in the scalar restatement

namespace peclr {
namespace {

constexpr int NP = PECLR_AUG_PARAM_DOUBLES;
constexpr int BX = 64, BY = 4;

struct ViewParam {
    double minv[6];
    bool rotate;
    int x0, y0, cw, ch;
    bool color;
    double h, s, a, b;
};

__device__ __forceinline__ ViewParam load_param(const double* __restrict__ params, int n) {
    const double* p = params + (size_t)n * NP;
    ViewParam v;
#pragma unroll
    for (int i = 0; i < 6; ++i) v.minv[i] = p[i];
    v.rotate = p[6] != 0.0;
    v.x0 = (int)p[7], v.y0 = (int)p[8], v.cw = (int)p[9], v.ch = (int)p[10];
    v.color = p[11] != 0.0;
    v.h = p[12], v.s = p[13], v.a = p[14], v.b = p[15];
    return v;
}

__device__ __forceinline__ long long round_ll(double x) { return (long long)rint(x); }  // half to even

// ---- stage 1: rotation (8-bit warpAffine, bilinear, zero border), evaluated on the crop window only
__global__ __launch_bounds__(BX* BY) void warp_crop_kernel(const uint8_t* __restrict__ images, int B, int H, int W,
                                                            const double* __restrict__ params,
                                                            uint8_t* __restrict__ crops) {
    const int n = blockIdx.z;  // view * B + sample
    const ViewParam v = load_param(params, n);
    const int cx = blockIdx.x * BX + threadIdx.x, cy = blockIdx.y * BY + threadIdx.y;
    if (cx >= v.cw || cy >= v.ch) return;
    const uint8_t* src = images + (size_t)(n % B) * H * W * 3;
    uint8_t* dst = crops + ((size_t)n * H * W + (size_t)cy * W + cx) * 3;
    const int x = v.x0 + cx, y = v.y0 + cy;
    if (!v.rotate) {
        const uint8_t* s = src + ((size_t)y * W + x) * 3;
        dst[0] = s[0], dst[1] = s[1], dst[2] = s[2];
        return;
    }
    // 10-bit fixed-point source coordinates, rounded to 1/32 pixel
    const long long xf = (round_ll((v.minv[1] * y + v.minv[2]) * 1024.0) + 16 + round_ll(v.minv[0] * x * 1024.0)) >> 5;
    const long long yf = (round_ll((v.minv[4] * y + v.minv[5]) * 1024.0) + 16 + round_ll(v.minv[3] * x * 1024.0)) >> 5;
    const long long sx = xf >> 5, sy = yf >> 5;
    const int fx = (int)(xf & 31), fy = (int)(yf & 31);
    const int w00 = (32 - fx) * (32 - fy) * 32, w01 = fx * (32 - fy) * 32, w10 = (32 - fx) * fy * 32, w11 = fx * fy * 32;
    const bool x0ok = sx >= 0 && sx < W, x1ok = sx + 1 >= 0 && sx + 1 < W;
    const bool y0ok = sy >= 0 && sy < H, y1ok = sy + 1 >= 0 && sy + 1 < H;
    int acc[3] = {0, 0, 0};
    if (y0ok && x0ok) {
        const uint8_t* s = src + ((size_t)sy * W + sx) * 3;
        acc[0] += s[0] * w00, acc[1] += s[1] * w00, acc[2] += s[2] * w00;
    }
    if (y0ok && x1ok) {
        const uint8_t* s = src + ((size_t)sy * W + sx + 1) * 3;
        acc[0] += s[0] * w01, acc[1] += s[1] * w01, acc[2] += s[2] * w01;
    }
    if (y1ok && x0ok) {
        const uint8_t* s = src + ((size_t)(sy + 1) * W + sx) * 3;
        acc[0] += s[0] * w10, acc[1] += s[1] * w10, acc[2] += s[2] * w10;
    }
    if (y1ok && x1ok) {
        const uint8_t* s = src + ((size_t)(sy + 1) * W + sx + 1) * 3;
        acc[0] += s[0] * w11, acc[1] += s[1] * w11, acc[2] += s[2] * w11;
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) dst[c] = (uint8_t)((acc[c] + (1 << 14)) >> 15);
}

// ---- stage 2 helpers: cv::resize(INTER_AREA) paths
enum ResizeMode { kCopy = 0, kAreaFast = 1, kArea = 2, kLinear = 3 };

struct Axis {  // one direction of the resize
    double scale, inv_scale;
    int iscale;
    bool fast, shrink;
};

__device__ __forceinline__ Axis make_axis(int ssize, int dsize) {
    Axis a;
    a.inv_scale = (double)dsize / ssize;
    a.scale = 1.0 / a.inv_scale;
    a.iscale = (int)rint(a.scale);
    a.fast = fabs(a.scale - a.iscale) < 2.220446049250313e-16;
    a.shrink = a.scale >= 1.0;
    return a;
}

// area-mode bilinear coefficients (11-bit fixed point) of destination index d
__device__ __forceinline__ void linear_coef(int d, int ssize, const Axis& ax, int& ofs, int& c0, int& c1) {
    int s = (int)floor(d * ax.scale);
    float f = (float)((d + 1) - (s + 1) * ax.inv_scale);
    f = f <= 0.f ? 0.f : f - floorf(f);
    if (s < 0) f = 0.f, s = 0;
    if (s >= ssize - 1) f = 0.f, s = ssize - 1;
    ofs = s;
    c0 = (int)fminf(fmaxf(rintf((1.f - f) * 2048.f), -32768.f), 32767.f);
    c1 = (int)fminf(fmaxf(rintf(f * 2048.f), -32768.f), 32767.f);
}

// taps of destination index d in the general area path: up to `first + count` weights
struct AreaTaps {
    int sx1, sx2;      // full-weight cells [sx1, sx2)
    float w_lo, w_mid, w_hi;
    bool has_lo, has_hi;
};

__device__ __forceinline__ AreaTaps area_taps(int d, int ssize, double scale) {
    AreaTaps t;
    const double f1 = d * scale, f2 = f1 + scale;
    const double cell = fmin(scale, ssize - f1);
    int sx1 = (int)ceil(f1), sx2 = (int)floor(f2);
    sx2 = min(sx2, ssize - 1);
    sx1 = min(sx1, sx2);
    t.sx1 = sx1, t.sx2 = sx2;
    t.has_lo = sx1 - f1 > 1e-3;
    t.w_lo = (float)((sx1 - f1) / cell);
    t.w_mid = (float)(1.0 / cell);
    t.has_hi = f2 - sx2 > 1e-3;
    t.w_hi = (float)(fmin(fmin(f2 - sx2, 1.0), cell) / cell);
    return t;
}

__device__ __forceinline__ void row_area(const uint8_t* __restrict__ row, const AreaTaps& tx, float buf[3]) {
    buf[0] = buf[1] = buf[2] = 0.f;
    if (tx.has_lo) {
        const uint8_t* s = row + (size_t)(tx.sx1 - 1) * 3;
#pragma unroll
        for (int c = 0; c < 3; ++c) buf[c] = buf[c] + (float)s[c] * tx.w_lo;
    }
    for (int sx = tx.sx1; sx < tx.sx2; ++sx) {
        const uint8_t* s = row + (size_t)sx * 3;
#pragma unroll
        for (int c = 0; c < 3; ++c) buf[c] = buf[c] + (float)s[c] * tx.w_mid;
    }
    if (tx.has_hi) {
        const uint8_t* s = row + (size_t)tx.sx2 * 3;
#pragma unroll
        for (int c = 0; c < 3; ++c) buf[c] = buf[c] + (float)s[c] * tx.w_hi;
    }
}

__device__ __forceinline__ int clamp_u8(float v) { return (int)fminf(fmaxf(rintf(v), 0.f), 255.f); }

// ---- stage 2 helpers: 8-bit BGR <-> HSV (H in [0,180)) and the jitter between them
__device__ __forceinline__ void color_jitter(int px[3], const ViewParam& v) {
    const int b = px[0], g = px[1], r = px[2];
    const int vmax = max(max(b, g), r), vmin = min(min(b, g), r), diff = vmax - vmin;
    const long long sdiv = vmax ? round_ll(1044480.0 / (1.0 * vmax)) : 0;            // (255 << 12) / v
    const long long hdiv = diff ? round_ll(737280.0 / (6.0 * diff)) : 0;             // (180 << 12) / (6 diff)
    const int sat = (int)((diff * sdiv + 2048) >> 12);
    long long hn = vmax == r ? (g - b) : (vmax == g ? (b - r + 2 * diff) : (r - g + 4 * diff));
    int hue = (int)((hn * hdiv + 2048) >> 12);
    if (hue < 0) hue += 180;
    // the reference scales in float64, clips to [0,255] and truncates to 8 bits
    const int h8 = (int)fmin(fmax(hue * v.h, 0.0), 255.0);
    const int s8 = (int)fmin(fmax(sat * v.s, 0.0), 255.0);
    const int v8 = (int)fmin(fmax(vmax * v.a + v.b, 0.0), 255.0);
    // HSV -> BGR in float32
    const float s = (float)s8 * (1.f / 255.f), val = (float)v8 * (1.f / 255.f);
    float bb, gg, rr;
    if (s == 0.f) {
        bb = gg = rr = val;
    } else {
        float hh = (float)h8 * (6.f / 180.f);
        if (hh >= 6.f) hh = hh - 6.f;
        int sector = (int)floorf(hh);
        float f = hh - (float)sector;
        if ((unsigned)sector >= 6u) sector = 0, f = 0.f;
        float tab[4];
        tab[0] = val;
        tab[1] = val * (1.f - s);
        tab[2] = val * (1.f - s * f);
        tab[3] = val * (1.f - s * (1.f - f));
        // (b, g, r) table indices per sector, packed 2 bits each
        constexpr unsigned kB = 1u | (1u << 2) | (3u << 4) | (0u << 6) | (0u << 8) | (2u << 10);
        constexpr unsigned kG = 3u | (0u << 2) | (0u << 4) | (2u << 6) | (1u << 8) | (1u << 10);
        constexpr unsigned kR = 0u | (2u << 2) | (1u << 4) | (1u << 6) | (3u << 8) | (0u << 10);
        bb = tab[(kB >> (2 * sector)) & 3u];
        gg = tab[(kG >> (2 * sector)) & 3u];
        rr = tab[(kR >> (2 * sector)) & 3u];
    }
    px[0] = clamp_u8(bb * 255.f), px[1] = clamp_u8(gg * 255.f), px[2] = clamp_u8(rr * 255.f);
}

struct Norm {
    float mean[3], stdv[3];
};

// ---- stage 2: resize -> colour jitter -> ToTensor/Normalize
template <bool NHWC>
__global__ __launch_bounds__(BX* BY) void resize_color_norm_kernel(const uint8_t* __restrict__ crops, int B, int H, int W,
                                                                    const double* __restrict__ params, int out_h, int out_w,
                                                                    Norm norm, float* __restrict__ out) {
    const int n = blockIdx.z;
    const int dx = blockIdx.x * BX + threadIdx.x, dy = blockIdx.y * BY + threadIdx.y;
    if (dx >= out_w || dy >= out_h) return;
    const ViewParam v = load_param(params, n);
    const uint8_t* img = crops + (size_t)n * H * W * 3;  // window rows have the source stride W
    const size_t stride = (size_t)W * 3;
    const int sw = v.cw, sh = v.ch;
    int px[3];
    const Axis ax = make_axis(sw, out_w), ay = make_axis(sh, out_h);
    int mode;
    if (sw == out_w && sh == out_h)
        mode = kCopy;
    else if (ax.shrink && ay.shrink)
        mode = (ax.fast && ay.fast) ? kAreaFast : kArea;
    else
        mode = kLinear;

    if (mode == kCopy) {
        const uint8_t* s = img + dy * stride + (size_t)dx * 3;
        px[0] = s[0], px[1] = s[1], px[2] = s[2];
    } else if (mode == kAreaFast) {
        int sum[3] = {0, 0, 0};
        for (int j = 0; j < ay.iscale; ++j) {
            const uint8_t* s = img + (size_t)(dy * ay.iscale + j) * stride + (size_t)dx * ax.iscale * 3;
            for (int i = 0; i < ax.iscale; ++i, s += 3) sum[0] += s[0], sum[1] += s[1], sum[2] += s[2];
        }
        if (ax.iscale == 2 && ay.iscale == 2) {
#pragma unroll
            for (int c = 0; c < 3; ++c) px[c] = (sum[c] + 2) >> 2;
        } else {
            const float inv = (float)(1.0 / (ax.iscale * ay.iscale));
#pragma unroll
            for (int c = 0; c < 3; ++c) px[c] = clamp_u8((float)sum[c] * inv);
        }
    } else if (mode == kArea) {
        const AreaTaps tx = area_taps(dx, sw, ax.scale), ty = area_taps(dy, sh, ay.scale);
        float acc[3] = {0.f, 0.f, 0.f}, buf[3];
        if (ty.has_lo) {
            row_area(img + (size_t)(ty.sx1 - 1) * stride, tx, buf);
#pragma unroll
            for (int c = 0; c < 3; ++c) acc[c] = acc[c] + buf[c] * ty.w_lo;
        }
        for (int sy = ty.sx1; sy < ty.sx2; ++sy) {
            row_area(img + (size_t)sy * stride, tx, buf);
#pragma unroll
            for (int c = 0; c < 3; ++c) acc[c] = acc[c] + buf[c] * ty.w_mid;
        }
        if (ty.has_hi) {
            row_area(img + (size_t)ty.sx2 * stride, tx, buf);
#pragma unroll
            for (int c = 0; c < 3; ++c) acc[c] = acc[c] + buf[c] * ty.w_hi;
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) px[c] = clamp_u8(acc[c]);
    } else {
        int xo, xa0, xa1, yo, yb0, yb1;
        linear_coef(dx, sw, ax, xo, xa0, xa1);
        linear_coef(dy, sh, ay, yo, yb0, yb1);
        const int x1 = min(xo + 1, sw - 1), y1 = min(yo + 1, sh - 1);
        const uint8_t *r0 = img + (size_t)yo * stride, *r1 = img + (size_t)y1 * stride;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const int s0 = r0[(size_t)xo * 3 + c] * xa0 + r0[(size_t)x1 * 3 + c] * xa1;
            const int s1 = r1[(size_t)xo * 3 + c] * xa0 + r1[(size_t)x1 * 3 + c] * xa1;
            const int o = (((yb0 * (s0 >> 4)) >> 16) + ((yb1 * (s1 >> 4)) >> 16) + 2) >> 2;
            px[c] = min(max(o, 0), 255);
        }
    }
    if (v.color) color_jitter(px, v);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float t = ((float)px[c] / 255.f - norm.mean[c]) / norm.stdv[c];
        if (NHWC)
            out[(((size_t)n * out_h + dy) * out_w + dx) * 3 + c] = t;
        else
            out[(((size_t)n * 3 + c) * out_h + dy) * out_w + dx] = t;
    }
}

}  // namespace
}  // namespace peclr

using namespace peclr;

extern "C" int peclr_augment_warp_crop_u8(const uint8_t* images, int B, int H, int W, int n_views, const double* params,
                                          uint8_t* crops, peclr_stream_t stream) {
    if (!images || !params || !crops) return PECLR_ERR_NULL;
    if (B <= 0 || H <= 0 || W <= 0 || n_views <= 0 || (long long)B * n_views > 65535) return PECLR_ERR_SHAPE;
    dim3 grid((W + BX - 1) / BX, (H + BY - 1) / BY, B * n_views);
    hipLaunchKernelGGL(warp_crop_kernel, grid, dim3(BX, BY), 0, static_cast<hipStream_t>(stream), images, B, H, W, params, crops);
    return launch_status();
}

extern "C" int peclr_augment_resize_color_norm(const uint8_t* crops, int B, int H, int W, int n_views, const double* params,
                                               int out_h, int out_w, const float* mean, const float* stdv,
                                               int channels_last, float* out, peclr_stream_t stream) {
    if (!crops || !params || !mean || !stdv || !out) return PECLR_ERR_NULL;
    if (B <= 0 || H <= 0 || W <= 0 || n_views <= 0 || out_h <= 0 || out_w <= 0 || (long long)B * n_views > 65535)
        return PECLR_ERR_SHAPE;
    Norm norm;
    for (int c = 0; c < 3; ++c) {
        norm.mean[c] = mean[c];  // host pointers: three floats each, passed by value to the kernel
        norm.stdv[c] = stdv[c];
        if (!(norm.stdv[c] > 0.f)) return PECLR_ERR_SHAPE;
    }
    dim3 grid((out_w + BX - 1) / BX, (out_h + BY - 1) / BY, B * n_views);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (channels_last)
        hipLaunchKernelGGL((resize_color_norm_kernel<true>), grid, dim3(BX, BY), 0, s, crops, B, H, W, params, out_h, out_w, norm, out);
    else
        hipLaunchKernelGGL((resize_color_norm_kernel<false>), grid, dim3(BX, BY), 0, s, crops, B, H, W, params, out_h, out_w, norm, out);
    return launch_status();
}
