// Ablation harness for the fp32 3x3 weight-gradient kernels (not part of the library): gemm_x6w_kernel (nine taps, X split once per
// tap) and wgrad_x6r_kernel (ring: every element split once), timed with parts of their loops switched off at the 3x3 shapes of the
// C2 step (ResNet-50, 2 x 128 views @224).  Round 6 question: what would X / dY that arrive ALREADY split (bf16 planes written by
// the BatchNorm pass that produces them) buy the weight gradients?  Upper bounds: "no split" = the planes still travel through
// registers and ds_write; "no plane stores" = they arrive by LDS-DMA (no VALU, no ds_write).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-inline-asm -I peclr_amd/csrc -I tools/exp tools/exp/x6w_ablate.hip -o tools/exp/x6w_ablate
// (wgrad_x6r.hip, the ring kernel, left the library at the end of round 6 and lives next to this file)
#include "gemm_x6t.hip"
#include "wgrad_x6r.hip"                                  // (tools/exp: -I tools/exp)

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

// 1x1 weight gradients: the first form (gemm_x6t_kernel) against the re-scheduled one (gemm_x6t2_kernel), tile choice as the library's
static bool L1_PF2;
template <int MT, int NT, int WGM> static void l1(bool two, dim3 grid, hipStream_t st) {
    if (two && L1_PF2) hipLaunchKernelGGL((gemm_x6t2_kernel<MT, NT, WGM, true>), grid, dim3(512), 0, st, GW);
    else if (two) hipLaunchKernelGGL((gemm_x6t2_kernel<MT, NT, WGM>), grid, dim3(512), 0, st, GW);
    else hipLaunchKernelGGL((gemm_x6t_kernel<MT, NT, WGM, false, 1>), grid, dim3(512), 0, st, GW);
}
static bool L1_TWO;
static void launch1(hipStream_t st) {
    const TilePick t = pick_tile(GW.M, GW.N);
    const dim3 grid(((GW.M + tile_m(t) - 1) / tile_m(t)) * ((GW.N + tile_n(t) - 1) / tile_n(t)), SW);
    if (t.wgm == 2) { if (t.nt == 2) l1<1, 2, 2>(L1_TWO, grid, st); else l1<1, 1, 2>(L1_TWO, grid, st); }
    else if (t.nt == 1) { if (t.mt == 2) l1<2, 1, 4>(L1_TWO, grid, st); else l1<1, 1, 4>(L1_TWO, grid, st); }
    else if (t.mt == 2 && t.nt == 4) l1<2, 4, 4>(L1_TWO, grid, st);
    else if (t.mt == 2) l1<2, 2, 4>(L1_TWO, grid, st);
    else if (t.nt == 4) l1<1, 4, 4>(L1_TWO, grid, st);
    else l1<1, 2, 4>(L1_TWO, grid, st);
}
static void l1_old(hipStream_t st) { L1_TWO = false; L1_PF2 = false; launch1(st); }
static void l1_new(hipStream_t st) { L1_TWO = true; L1_PF2 = false; launch1(st); }
static void l1_pf2(hipStream_t st) { L1_TWO = true; L1_PF2 = true; launch1(st); }

static int one_by_one(float* A, float* B, float* slabs, float* zeros, unsigned char* junk) {
    // (rows, Cout = M, Cin = N) of ResNet-50's 1x1 convolutions at 2 x 128 views @224
    const int shapes[][3] = {{802816, 64, 64}, {802816, 256, 64}, {802816, 64, 256}, {200704, 128, 512}, {200704, 512, 128}, {200704, 128, 256},
                             {50176, 256, 1024}, {50176, 1024, 256}, {50176, 256, 512}, {12544, 512, 2048}, {12544, 2048, 512}, {12544, 512, 1024}};
    for (auto& sh : shapes) {
        const int K = sh[0], M = sh[1], N = sh[2];
        SW = peclr_gemm_x6t_slabs(M, N, K, 1);
        GW = X6TArgs{};
        GW.A = A; GW.B = B; GW.slabs = slabs; GW.M = M; GW.N = N; GW.K = K; GW.lda = M; GW.ldb = N; GW.ldc = N;
        GW.kchunk = ((K + SW - 1) / SW + TK - 1) / TK * TK;
        GW.taps = 1; GW.H = 1; GW.W = 1; GW.stride = 1; GW.zeros = zeros;
        const size_t nb = (size_t)SW * M * N * 4;
        if (nb > ((size_t)512 << 20)) { printf("slabs too large\n"); return 1; }
        std::vector<unsigned char> h0(nb), h1(nb);
        CK(hipMemset(slabs, 0xFF, nb)); l1_old(0); CK(hipMemcpy(h0.data(), slabs, nb, hipMemcpyDeviceToHost));
        CK(hipMemset(slabs, 0xFF, nb)); l1_new(0); CK(hipMemcpy(h1.data(), slabs, nb, hipMemcpyDeviceToHost));
        size_t bad = 0;
        for (size_t i = 0; i < nb; ++i) bad += h0[i] != h1[i];
        printf("   gemm_x6t2 vs gemm_x6t: %zu of %zu bytes differ%s\n", bad, nb, bad ? "  <-- MISMATCH" : " (bit-identical)");
        char what[160];
        snprintf(what, sizeof what, "1x1 weight gradient: %d rows, Cout %d x Cin %d (slabs %d)", K, M, N, SW);
        CK(hipMemset(slabs, 0xFF, nb)); l1_pf2(0); CK(hipMemcpy(h1.data(), slabs, nb, hipMemcpyDeviceToHost));
        bad = 0;
        for (size_t i = 0; i < nb; ++i) bad += h0[i] != h1[i];
        printf("   gemm_x6t2 (loads two steps ahead) vs gemm_x6t: %zu of %zu bytes differ%s\n", bad, nb, bad ? "  <-- MISMATCH" : " (bit-identical)");
        const V vs[] = {{"first form (gemm_x6t)", l1_old}, {"  re-scheduled loop (gemm_x6t2)", l1_new}, {"  ... loads two k-steps ahead", l1_pf2}};
        if (run(what, vs, 3, 2.0 * K * M * N, junk)) return 1;
    }
    return 0;
}

int main(int argc, char** argv) {
    const int shapes3[][3] = {{256, 56, 64}, {256, 28, 128}, {256, 14, 256}, {256, 7, 512}};       // (images, H = W, C)
    float *A, *B, *slabs, *zeros; unsigned char* junk;
    const size_t nel = (size_t)802816 * 256;                  // (the widest 1x1 operand)
    CK(hipMalloc(&A, nel * 4)); CK(hipMalloc(&B, nel * 4)); CK(hipMalloc(&slabs, (size_t)512 << 20)); CK(hipMalloc(&junk, 512u << 20));
    CK(hipMalloc(&zeros, 256)); CK(hipMemset(zeros, 0, 256));
    fill<<<4096, 256>>>(A, nel, 1, 1.f);
    fill<<<4096, 256>>>(B, nel, 2, 1.f);
    if (argc > 1 && argv[1][0] == '1') return one_by_one(A, B, slabs, zeros, junk);
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
