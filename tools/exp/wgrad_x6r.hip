// fp32 weight gradient of the 3x3 / padding-1 / stride-1 convolutions on the bf16 matrix cores at fp32 accuracy, third
// generation:  dW[co][tap][ci] = sum over output pixels r of dY[r][co] * X[r + tap][ci]  with every operand element split
// (x == h + m + l, common.hpp) exactly ONCE per workgroup.
//
// gemm_x6w_kernel (gemm_x6t.hip) holds a fragment as 8 consecutive PIXELS of one channel, so a tap's one-pixel shift is a
// 2-byte shift inside a fragment and X has to be loaded and split once per tap -- nine times.  Here the planes lie in LDS
// PIXEL-major ([plane][32-channel block][slot][32 channels x 2 B]) and gfx950's transposing LDS read (ds_read_b64_tr_b16)
// builds the k-contiguous MFMA fragments on the way out, at ANY slot offset: X is split once into a RING of slots that every
// k-step advances by 32, and the nine taps read it at their offsets.
// The contraction runs over the padded linear pixel space of wgrad_h.hip (one shared zero column per image row, one shared
// zero row per image: tap (a, b) is the uniform shift (a - 1)(W + 1) + (b - 1), no per-(pixel, tap) masks; pad slots are
// written as zeros).  Per k-step of 32 slots a wave runs 2 x 9 taps x 6 products = 108 MFMAs against 12 + 108 transposing
// reads and ~25 split instructions: the VALU / LDS-store share of gemm_x6w (its limit) is gone.
// Workgroup = 8 waves, MBLK x NBLK 32-channel blocks of dY / X (4 x 1: 128 output x 32 input channels; 2 x 2 for 64 output
// channels), wave = one 32 x 32 block x a tap group (five or four taps).  96 - 120 KiB of LDS: one workgroup per CU, the global
// loads of step t + 1 are in flight under the MFMAs of step t.
#include "common.hpp"

namespace peclr {
namespace {

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

struct X6RArgs {
    const float* A;                      // dY [images * H * W][lda]
    const float* B;                      // X  [images * H * W][ldb]
    float* slabs;                        // [n_slabs][M][9 * N]
    int M, N, lda, ldb;
    int H, W, images;
    int P;                               // padded slots: images * (H + 1) * (W + 1)
    int pchunk;                          // padded slots per slab (multiple of 32)
};

__device__ __forceinline__ f32x16 rmma(const uint4& a, const uint4& b, f32x16 acc) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
}
__device__ __forceinline__ uint4 rtr(const unsigned char* p0, const unsigned char* p1) {     // slots k .. k + 3 and k + 4 .. k + 7
    typedef __attribute__((address_space(3))) s16x4* lp;
    const uint2 lo = __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(p0)));
    const uint2 hi = __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(p1)));
    return make_uint4(lo.x, lo.y, hi.x, hi.y);
}

constexpr int RING = 256;                // slots of the X ring (W <= 62: lead + lag 2 x 64, the k-step's 32, one group being written)

// Eight waves: wave = (32 x 32 block, tap group): taps 0 - 4 or 5 - 8 -- two waves per SIMD, so that one wave's transposing reads
// and their latency run under the other's MFMAs (four waves of nine taps: 150 TFLOP/s; a wave alone on its SIMD exposes
// every LDS round trip)
// ABL: ablation switches for tools/exp/x6w_ablate.hip (0 in the library): 1 = planes stored without the split (raw halves),
// 2 = no plane stores inside the loop, 4 = no global loads inside the loop
template <int MBLK, int NBLK, int ABL = 0>
__global__ __launch_bounds__(512, 1) void wgrad_x6r_kernel(X6RArgs g) {
    static_assert(MBLK * NBLK == 4, "four 32 x 32 blocks, two tap groups each");
    constexpr int APL = MBLK * 2048;                     // bytes of one dY plane of a stage: [block][32 slots][64 B]
    constexpr int ASTG = 3 * APL;
    constexpr int XPL = NBLK * RING * 64;                // bytes of one X ring plane: [block][RING slots][64 B]
    constexpr int X0 = 2 * ASTG;
    __shared__ __attribute__((aligned(16))) unsigned char lds[X0 + 3 * XPL];
    const int tid = threadIdx.x, lane = tid & 63, wave = (tid >> 6) & 3, tapg = tid >> 8;
    const int tap0 = tapg * 5, ntaps = tapg ? 4 : 5;      // this wave's taps: [tap0, tap0 + ntaps)
    const int W1 = g.W + 1, HW1 = (g.H + 1) * W1;
    const int L = (g.W + 2 + 31) / 32 * 32;              // lead / lag of the ring around the current k-step, in slots
    const int G0 = 2 * L / 32 + 1;                       // ring groups a k-step reads
    const int ntn = g.N / (32 * NBLK);
    const int m0 = (int)(blockIdx.x / ntn) * 32 * MBLK, n0 = (int)(blockIdx.x % ntn) * 32 * NBLK;
    const int p_begin = blockIdx.y * g.pchunk;
    const int p_end = min(g.P, p_begin + g.pchunk);
    const int nk = (p_end - p_begin + 31) / 32;
    const int mblk = wave / NBLK, nblk = wave % NBLK;

    f32x16 acc[5];
#pragma unroll
    for (int t = 0; t < 5; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    // loader roles: thread -> slot (tid >> 4) of a 32-slot group, float4 column c4 = tid & 15 (+ 16 j) of the slot's channels
    // (8 MBLK float4 of dY and 8 NBLK of X per slot: threads past the X columns sit the X loads out)
    const int lslot = tid >> 4, c4 = tid & 15;
    constexpr int NFA = (MBLK + 1) / 2, NFX = (NBLK + 1) / 2;
    auto row_of = [&](int p, bool live, size_t& row) -> bool {      // padded slot -> pixel row; false: a zero slot
        const int img = p / HW1, rem = p - img * HW1, hh = rem / W1, ww = rem - hh * W1;
        row = ((size_t)img * g.H + (hh - 1)) * g.W + (ww - 1);
        return live && p >= 0 && img < g.images && hh >= 1 && ww >= 1;
    };
    f32x4 ra[NFA], rx[NFX];
    auto load_a = [&](int t) {                            // dY slots [p_begin + 32 t, + 32)
        const int p = p_begin + 32 * t + lslot;
        size_t row;
        const bool real = row_of(p, p < p_end, row);
#pragma unroll
        for (int j = 0; j < NFA; ++j) {
            ra[j] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (real && c4 + 16 * j < 8 * MBLK) ra[j] = *reinterpret_cast<const f32x4*>(g.A + row * g.lda + m0 + 4 * (c4 + 16 * j));
        }
    };
    auto load_x = [&](int u) {                            // ring group u: X slots [p_begin - L + 32 u, + 32)
        const int p = p_begin - L + 32 * u + lslot;
        size_t row;
        const bool real = row_of(p, true, row);
#pragma unroll
        for (int j = 0; j < NFX; ++j) {
            rx[j] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (real && c4 + 16 * j < 8 * NBLK) rx[j] = *reinterpret_cast<const f32x4*>(g.B + row * g.ldb + n0 + 4 * (c4 + 16 * j));
        }
    };
    // four channels of one slot -> 8 bytes in each of the three planes: float4 column q = c4 + 16 j lies in channel block q >> 3,
    // at byte (q & 7) * 8 of the slot's 64
    auto split_to = [&](const f32x4& v, unsigned char* d, int plane_stride) {
        unsigned h[2], m[2], l[2];
        if constexpr (ABL & 1) {
            h[0] = __float_as_uint(v[0]); m[0] = __float_as_uint(v[1]); l[0] = h[0] ^ m[0];
            h[1] = __float_as_uint(v[2]); m[1] = __float_as_uint(v[3]); l[1] = h[1] ^ m[1];
        } else {
            split3_pk(v[0], v[1], h[0], m[0], l[0]);
            split3_pk(v[2], v[3], h[1], m[1], l[1]);
        }
        *reinterpret_cast<uint2*>(d) = make_uint2(h[0], h[1]);
        *reinterpret_cast<uint2*>(d + plane_stride) = make_uint2(m[0], m[1]);
        *reinterpret_cast<uint2*>(d + 2 * plane_stride) = make_uint2(l[0], l[1]);
    };
    auto store_a = [&](int t) {
        unsigned char* d = lds + (t & 1) * ASTG + lslot * 64;
#pragma unroll
        for (int j = 0; j < NFA; ++j) {
            const int q = c4 + 16 * j;
            if (q < 8 * MBLK) split_to(ra[j], d + (q >> 3) * 2048 + (q & 7) * 8, APL);
        }
    };
    auto store_x = [&](int u) {
        unsigned char* d = lds + X0 + ((32 * u + lslot) & (RING - 1)) * 64;
#pragma unroll
        for (int j = 0; j < NFX; ++j) {
            const int q = c4 + 16 * j;
            if (q < 8 * NBLK) split_to(rx[j], d + (q >> 3) * RING * 64 + (q & 7) * 8, XPL);
        }
    };

    // prologue: dY(0) and the ring groups step 0 reads
    load_a(0);
    store_a(0);
    for (int u = 0; u < G0; ++u) { load_x(u); store_x(u); }

    const int ll = lane & 15, gq = lane >> 4;
    const int fpix = 8 * (gq >> 1) + (ll >> 2);           // slot of this source lane inside a 16-slot k-extent (second read: + 4)
    const int fch = (16 * (gq & 1) + 4 * (ll & 3)) * 2;
    for (int t = 0; t < nk; ++t) {
        __syncthreads();                                  // planes of step t are in the LDS; every wave is done with step t - 1
        const bool more = t + 1 < nk;
        if constexpr (!(ABL & 4))
            if (more) { load_a(t + 1); load_x(t + G0); }  // in flight under this step's MFMAs
        const unsigned char* sa = lds + (t & 1) * ASTG + mblk * 2048 + fch;
        const unsigned char* sx = lds + X0 + nblk * RING * 64 + fch;
        const int xbase = 32 * t + L + fpix;              // ring slot (before the wrap) of this lane's first pixel at shift 0
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            uint4 af[3];
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                const unsigned char* b = sa + p * APL + (kk * 16 + fpix) * 64;
                af[p] = rtr(b, b + 4 * 64);
            }
            uint4 bf[5][3];
#pragma unroll
            for (int u = 0; u < 5; ++u) {
                const int tap = tap0 + (u < ntaps ? u : 0);           // (the four-tap group reads its first tap twice; that product is dropped)
                const int ta = tap / 3, tb = tap - 3 * ta;
                const int shift = (ta - 1) * W1 + (tb - 1);
                const unsigned s0 = (unsigned)(xbase + kk * 16 + shift) & (RING - 1), s1 = (unsigned)(xbase + kk * 16 + shift + 4) & (RING - 1);
#pragma unroll
                for (int p = 0; p < 3; ++p) bf[u][p] = rtr(sx + p * XPL + s0 * 64, sx + p * XPL + s1 * 64);
            }
            // six of the nine partial products, smallest first (planes: 0 = h, 1 = m, 2 = l); consecutive MFMAs write different accumulators
#define PECLR_X6R(P, Q) _Pragma("unroll") for (int u = 0; u < 5; ++u) if (u < 4 || ntaps == 5) acc[u] = rmma(af[P], bf[u][Q], acc[u]);
            PECLR_X6R(2, 0) PECLR_X6R(0, 2) PECLR_X6R(1, 1) PECLR_X6R(1, 0) PECLR_X6R(0, 1) PECLR_X6R(0, 0)
#undef PECLR_X6R
        }
        if constexpr (!(ABL & 2))
            if (more) { store_a(t + 1); store_x(t + G0); }    // other stage / the ring group no step <= t reads
    }
    float* slab = g.slabs + (size_t)blockIdx.y * g.M * 9 * g.N;
    const int i = lane & 31, kh = lane >> 5;
    const int mb = m0 + 32 * mblk, nb = n0 + 32 * nblk;
#pragma unroll
    for (int u = 0; u < 5; ++u)
        if (u < ntaps) {
#pragma unroll
            for (int r = 0; r < 16; ++r) slab[(size_t)(mb + mfma32_row(r, kh)) * (9 * g.N) + (tap0 + u) * g.N + nb + i] = acc[u][r];
        }
}

}  // namespace
}  // namespace peclr

using namespace peclr;

extern "C" int peclr_wgrad3_x6r_slabs(int M, int N, int images, int H, int W) {
    if (M <= 0 || N <= 0 || images <= 0 || H <= 0 || W <= 0 || M % 64 || N % 64 || W > 62) return 0;
    const long P = (long)images * (H + 1) * (W + 1);
    const long tiles = M % 128 ? (long)(M / 64) * (N / 64) : (long)(M / 128) * (N / 32);
    long s = (512 + tiles - 1) / tiles;                  // two rounds of one workgroup per CU
    const long max_s = (P + 24 * 32 - 1) / (24 * 32);    // at least 24 k-steps per slab (the ring warm-up is 3 - 5 groups)
    if (s > max_s) s = max_s;
    if (s < 1) s = 1;
    const long pchunk = ((P + s - 1) / s + 31) / 32 * 32;
    return (int)((P + pchunk - 1) / pchunk);
}

// dW slabs [n_slabs][Cout][9 * Cin] (fp32) of a 3x3 / padding-1 / stride-1 convolution from fp32 NHWC activations dY [images, H,
// W, M = Cout], X [images, H, W, N = Cin]; M, N multiples of 64, W <= 62.  Sum the slabs with peclr_slab_reduce_f32.
extern "C" int peclr_wgrad3_x6r_f32(int M, int N, int images, int H, int W, const float* A, const float* B, float* slabs, int n_slabs,
                                    peclr_stream_t stream) {
    if (!A || !B || !slabs) return PECLR_ERR_NULL;
    if (M <= 0 || N <= 0 || images <= 0 || H <= 0 || W <= 0 || M % 64 || N % 64 || W > 62) return PECLR_ERR_SHAPE;
    if ((long)images * (H + 1) * (W + 1) > 0x7fffffffL / 2) return PECLR_ERR_SHAPE;
    if (!aligned16(A) || !aligned16(B) || !aligned16(slabs)) return PECLR_ERR_ALIGN;
    if (n_slabs != peclr_wgrad3_x6r_slabs(M, N, images, H, W)) return PECLR_ERR_WORKSPACE;
    X6RArgs g;
    g.A = A; g.B = B; g.slabs = slabs; g.M = M; g.N = N; g.lda = M; g.ldb = N; g.H = H; g.W = W; g.images = images;
    g.P = images * (H + 1) * (W + 1);
    g.pchunk = ((g.P + n_slabs - 1) / n_slabs + 31) / 32 * 32;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (M % 128) hipLaunchKernelGGL((wgrad_x6r_kernel<2, 2>), dim3((M / 64) * (N / 64), n_slabs), dim3(512), 0, s, g);
    else hipLaunchKernelGGL((wgrad_x6r_kernel<4, 1>), dim3((M / 128) * (N / 32), n_slabs), dim3(512), 0, s, g);
    return launch_status();
}
