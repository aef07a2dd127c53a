// Ablation harness for the fp32 3x3 weight-gradient kernels (not part of the library): gemm_x6w_kernel (nine taps, X split once per
// tap) and wgrad_x6r_kernel (ring: every element split once), timed with parts of their loops switched off at the 3x3 shapes of the
// C2 step (ResNet-50, 2 x 128 views @224).  Round 6 question: what would X / dY that arrive ALREADY split (bf16 planes written by
// the BatchNorm pass that produces them) buy the weight gradients?  Upper bounds: "no split" = the planes still travel through
// registers and ds_write; "no plane stores" = they arrive by LDS-DMA (no VALU, no ds_write).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-inline-asm -I peclr_amd/csrc tools/exp/x6w_ablate.hip -o tools/exp/x6w_ablate
#include "gemm_x6t.hip"
#include "wgrad_x6r.hip"

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void fill(float* p, size_t n, unsigned seed, float scale) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned x = (unsigned)i * 2654435761u + seed;
        x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
        p[i] = scale * ((int)(x & 0xFFFFFF) - 0x800000) * (1.f / 0x800000);
    }
}

struct V { const char* name; void (*fn)(hipStream_t); };
static X6TArgs GW;
static X6RArgs GR;
static int SW, SR;        // slabs

template <int ABL> static void lw(hipStream_t st) {
    hipLaunchKernelGGL((gemm_x6w_kernel<4, ABL>), dim3(((GW.M + 127) / 128) * ((GW.N + 63) / 64), SW), dim3(512), 0, st, GW);
}
template <int ABL> static void lr(hipStream_t st) {
    hipLaunchKernelGGL((wgrad_x6r_kernel<4, 1, ABL>), dim3((GR.M / 128) * (GR.N / 32), SR), dim3(512), 0, st, GR);
}
static void lv4(hipStream_t st) {
    hipLaunchKernelGGL((gemm_x6w2_kernel<4>), dim3(((GW.M + 127) / 128) * ((GW.N + 63) / 64), SW), dim3(512), 0, st, GW);
}
static void lv2(hipStream_t st) {
    hipLaunchKernelGGL((gemm_x6w2_kernel<2>), dim3(((GW.M + 63) / 64) * ((GW.N + 63) / 64), SW), dim3(512), 0, st, GW);
}
template <int ABL> static void lw2(hipStream_t st) {
    hipLaunchKernelGGL((gemm_x6w_kernel<2, ABL>), dim3(((GW.M + 63) / 64) * ((GW.N + 63) / 64), SW), dim3(512), 0, st, GW);
}
template <int ABL> static void lr2(hipStream_t st) {
    hipLaunchKernelGGL((wgrad_x6r_kernel<2, 2, ABL>), dim3((GR.M / 64) * (GR.N / 64), SR), dim3(512), 0, st, GR);
}

static int run(const char* what, const V* vs, int nv, double flops, unsigned char* junk) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    printf("%s  (six products at 2.5 PFLOP/s: %.1f us)\n", what, flops * 6 / 2.5e15 * 1e6);
    float base = 0.f;
    for (int i = 0; i < nv; ++i) {
        std::vector<float> ts;
        for (int r = 0; r < 9; ++r) {
            CK(hipMemsetAsync(junk, r, 512u << 20, 0));
            CK(hipEventRecord(e0, 0)); vs[i].fn(0); CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ts.push_back(ms * 1e3f);
        }
        CK(hipGetLastError());
        std::sort(ts.begin() + 1, ts.end());
        if (vs[i].name[0] != ' ') base = ts[5];              // un-indented rows are the reference of the rows below them
        printf("   %-52s %8.1f us (min %8.1f)  %6.1f TF  %+5.1f %%\n", vs[i].name, ts[5], ts[1], flops / ts[5] / 1e6, 100.0 * (ts[5] - base) / base);
    }
    return 0;
}

int main() {
    const int shapes3[][3] = {{256, 56, 64}, {256, 28, 128}, {256, 14, 256}, {256, 7, 512}};       // (images, H = W, C)
    float *A, *B, *slabs, *zeros; unsigned char* junk;
    const size_t nel = (size_t)256 * 56 * 56 * 64;
    CK(hipMalloc(&A, nel * 4)); CK(hipMalloc(&B, nel * 4)); CK(hipMalloc(&slabs, (size_t)512 << 20)); CK(hipMalloc(&junk, 512u << 20));
    CK(hipMalloc(&zeros, 256)); CK(hipMemset(zeros, 0, 256));
    fill<<<4096, 256>>>(A, nel, 1, 1.f);
    fill<<<4096, 256>>>(B, nel, 2, 1.f);
    for (auto& sh : shapes3) {
        const int NB = sh[0], H = sh[1], C = sh[2], K = NB * H * H;
        SW = peclr_gemm_x6t_slabs(C, C, K, 9);
        SR = peclr_wgrad3_x6r_slabs(C, C, NB, H, H);
        if ((size_t)std::max(SW, SR) * C * 9 * C * 4 > ((size_t)512 << 20)) { printf("slabs too large\n"); return 1; }
        GW = X6TArgs{};
        GW.A = A; GW.B = B; GW.slabs = slabs; GW.M = C; GW.N = C; GW.K = K; GW.lda = C; GW.ldb = C; GW.ldc = 9 * C;
        GW.kchunk = ((K + SW - 1) / SW + TK - 1) / TK * TK;
        GW.taps = 9; GW.H = H; GW.W = H; GW.stride = 1; GW.zeros = zeros;
        GR = X6RArgs{};
        GR.A = A; GR.B = B; GR.slabs = slabs; GR.M = C; GR.N = C; GR.lda = C; GR.ldb = C; GR.H = H; GR.W = H; GR.images = NB;
        GR.P = NB * (H + 1) * (H + 1);
        GR.pchunk = ((GR.P + SR - 1) / SR + 31) / 32 * 32;
        char what[160];
        snprintf(what, sizeof what, "3x3 weight gradient: %d images %d x %d, %d channels (slabs: nine-tap %d, ring %d)", NB, H, H, C, SW, SR);
        {   // the re-scheduled kernel must reproduce the first form bit for bit (same products, same order per accumulator)
            const size_t nb = (size_t)SW * C * 9 * C * 4;
            std::vector<unsigned char> h0(nb), h1(nb);
            CK(hipMemset(slabs, 0xFF, nb));
            if (C == 64) lw2<0>(0); else lw<0>(0);
            CK(hipMemcpy(h0.data(), slabs, nb, hipMemcpyDeviceToHost));
            CK(hipMemset(slabs, 0xFF, nb));
            if (C == 64) lv2(0); else lv4(0);
            CK(hipMemcpy(h1.data(), slabs, nb, hipMemcpyDeviceToHost));
            size_t bad = 0;
            for (size_t i = 0; i < nb; ++i) bad += h0[i] != h1[i];
            printf("   gemm_x6w2 vs gemm_x6w: %zu of %zu bytes differ%s\n", bad, nb, bad ? "  <-- MISMATCH" : " (bit-identical)");
        }
        if (C == 64) {
            const V vs[] = {{"nine-tap kernel (gemm_x6w), full", lw2<0>}, {"  X stored without the split", lw2<1>}, {"  X and dY stored without the split", lw2<3>},
                            {"  no plane stores in the loop", lw2<7>}, {"  ... and no global loads: MFMAs + fragment reads", lw2<15>}, {"  re-scheduled loop (gemm_x6w2)", lv2},
                            {"ring kernel (wgrad_x6r), full", lr2<0>}, {"  planes stored without the split", lr2<1>}, {"  no plane stores in the loop", lr2<3>},
                            {"  ... and no global loads: MFMAs + transposing reads", lr2<7>}};
            if (run(what, vs, 10, 2.0 * K * C * 9 * C, junk)) return 1;
        } else {
            const V vs[] = {{"nine-tap kernel (gemm_x6w), full", lw<0>}, {"  X stored without the split", lw<1>}, {"  X and dY stored without the split", lw<3>},
                            {"  no plane stores in the loop", lw<7>}, {"  ... and no global loads: MFMAs + fragment reads", lw<15>}, {"  re-scheduled loop (gemm_x6w2)", lv4},
                            {"ring kernel (wgrad_x6r), full", lr<0>}, {"  planes stored without the split", lr<1>}, {"  no plane stores in the loop", lr<3>},
                            {"  ... and no global loads: MFMAs + transposing reads", lr<7>}};
            if (run(what, vs, 10, 2.0 * K * C * 9 * C, junk)) return 1;
        }
    }
    return 0;
}
