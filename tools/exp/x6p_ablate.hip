// Ablation harness for gemm_x6p_kernel (not part of the library): times the kernel with parts of its loop switched off, at the
// shapes of the C2 step (ResNet-50, 2 x 128 views @224).  Round 6 question: what would activations that arrive ALREADY split
// (three bf16 planes written by their producer) buy?  Upper bound = the variant without split and plane stores (the planes would
// still have to be fetched: 6 instead of 4 bytes per element).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-inline-asm -I peclr_amd/csrc tools/exp/x6p_ablate.hip -o tools/exp/x6p_ablate
#include "gemm_x6p.hip"

#include <algorithm>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void fill(float* p, size_t n, unsigned seed, float scale) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned x = (unsigned)i * 2654435761u + seed;
        x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
        p[i] = scale * ((int)(x & 0xFFFFFF) - 0x800000) * (1.f / 0x800000);
    }
}

template <int WM, int ABL, int NTL = 4>
static void launch(const X6PArgs& g, hipStream_t st) {
    const int tm = 128 * WM, nrb = (g.M + tm - 1) / tm;
    hipLaunchKernelGGL((gemm_x6p_kernel<WM, ABL, true, true, 1, NTL>), dim3(8 * ((nrb + 7) / 8) * (g.N / (32 * NTL))), dim3(256), 0, st, g);
}
template <int WM, int ABL, int NTL = 4>
static void launch3(const X6PArgs& g, hipStream_t st) {
    const int tm = 128 * WM, nrb = (g.M + tm - 1) / tm;
    hipLaunchKernelGGL((gemm_x6p_kernel<WM, ABL, true, true, 9, NTL, true>), dim3(8 * ((nrb + 7) / 8) * (g.N / (32 * NTL))), dim3(256), 0, st, g);
}

template <int WM, int ABL, int NTL = 4>
static void launch3p(const X6PArgs& g, hipStream_t st) {      // the fp16-pair arm (NP = 2) of the halo kernel
    const int tm = 128 * WM, nrb = (g.M + tm - 1) / tm;
    hipLaunchKernelGGL((gemm_x6p_kernel<WM, ABL, true, true, 9, NTL, true, 2>), dim3(8 * ((nrb + 7) / 8) * (g.N / (32 * NTL))), dim3(256), 0, st, g);
}

struct V { const char* name; void (*fn)(const X6PArgs&, hipStream_t); };

static int run(const char* what, const X6PArgs& g, const V* vs, int nv, double flops, unsigned char* junk) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    printf("%s  (six products at 2.5 PFLOP/s: %.1f us)\n", what, flops * 6 / 2.5e15 * 1e6);
    float base = 0.f;
    for (int i = 0; i < nv; ++i) {
        std::vector<float> ts;
        for (int r = 0; r < 9; ++r) {
            CK(hipMemsetAsync(junk, r, 512u << 20, 0));
            CK(hipEventRecord(e0, 0)); vs[i].fn(g, 0); CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ts.push_back(ms * 1e3f);
        }
        CK(hipGetLastError());
        std::sort(ts.begin() + 1, ts.end());
        if (i == 0 || !base) base = ts[5];
        if (vs[i].name[0] == 'f') base = ts[5];              // "full ..." rows are the reference of the rows below them
        printf("   %-44s %8.1f us (min %8.1f)  %6.1f TF  %+5.1f %%\n", vs[i].name, ts[5], ts[1], flops / ts[5] / 1e6, 100.0 * (ts[5] - base) / base);
    }
    return 0;
}

int main(int argc, char** argv) {          // no argument: everything; "3": the 3x3 shapes only; "3p": their fp16-pair arm
    const bool only3 = argc > 1 && argv[1][0] == '3';
    // 1x1 forward shapes of ResNet-50 at 2 x 128 views @224: (rows, Cout, Cin)
    const int shapes[][3] = {{802816, 256, 64}, {200704, 512, 128}, {200704, 128, 512}, {50176, 1024, 256}, {50176, 256, 1024},
                             {12544, 2048, 512}, {12544, 512, 2048}, {16384, 2048, 512}};
    // 3x3 / stride-1 shapes: (images, H = W, C)
    const int shapes3[][3] = {{256, 56, 64}, {256, 28, 128}, {256, 14, 256}, {256, 7, 512}};
    float *A, *C, *W, *zeros; unsigned char *Bp, *junk;
    CK(hipMalloc(&A, (size_t)802816 * 512 * 4)); CK(hipMalloc(&C, (size_t)802816 * 512 * 4));
    CK(hipMalloc(&W, (size_t)2048 * 4608 * 4)); CK(hipMalloc(&Bp, (size_t)2048 * 4608 * 6)); CK(hipMalloc(&junk, 512u << 20));
    CK(hipMalloc(&zeros, 256)); CK(hipMemset(zeros, 0, 256));
    fill<<<4096, 256>>>(A, (size_t)802816 * 512, 1, 1.f);
    fill<<<4096, 256>>>(W, (size_t)2048 * 4608, 2, 0.05f);
    for (auto& sh : shapes) {
        if (only3) break;
        const int M = sh[0], N = sh[1], K = sh[2];
        PackDesc d{(int64_t)W, (int64_t)Bp, N, K, K, 0, 0, 0}, *dd;
        CK(hipMalloc(&dd, sizeof(d))); CK(hipMemcpy(dd, &d, sizeof(d), hipMemcpyHostToDevice));
        x6_pack_kernel<<<(N / 128) * (K / 16), 256>>>(dd, 1);
        X6PArgs g{};
        g.A = A; g.Bp = Bp; g.out = C; g.M = M; g.N = N; g.K = K; g.lda = K; g.ldo = N; g.stride = 1; g.H = g.W = g.Hin = g.Win = 1;
        g.stream_out = (size_t)M * N * 4 > ((size_t)64 << 20);
        const V vs[] = {{"full / 256-row tiles", launch<2, 0>}, {"  no split, no plane stores", launch<2, 1>}, {"  ... and no activation loads", launch<2, 3>},
                        {"  MFMAs + fragment reads + barrier", launch<2, 7>}, {"  THREE products of the six (rest as full)", launch<2, 64>},
                        {"full / 128-row tiles", launch<1, 0>}, {"  no split, no plane stores", launch<1, 1>}, {"  ... and no activation loads", launch<1, 3>},
                        {"  MFMAs + fragment reads + barrier", launch<1, 7>}, {"  THREE products of the six (rest as full)", launch<1, 64>}};
        char what[128];
        snprintf(what, sizeof what, "1x1: rows %d, Cin %d -> Cout %d", M, K, N);
        if (run(what, g, vs, 10, 2.0 * M * N * K, junk)) return 1;
        CK(hipFree(dd));
    }
    for (auto& sh : shapes3) {
        const int NB = sh[0], H = sh[1], Cc = sh[2], M = NB * H * H, N = Cc, K = 9 * Cc;
        PackDesc d{(int64_t)W, (int64_t)Bp, N, K, K, 0, 0, 0}, *dd;       // (any filter: timing only)
        CK(hipMalloc(&dd, sizeof(d))); CK(hipMemcpy(dd, &d, sizeof(d), hipMemcpyHostToDevice));
        x6_pack_kernel<<<((N + 127) / 128) * (K / 16), 256>>>(dd, 1);
        X6PArgs g{};
        g.A = A; g.Bp = Bp; g.out = C; g.M = M; g.N = N; g.K = K; g.lda = Cc; g.ldo = N; g.stride = 1; g.H = g.Hin = H; g.W = g.Win = H;
        g.zeros = zeros;
        g.stream_out = (size_t)M * N * 4 > ((size_t)64 << 20);
        char what[128];
        snprintf(what, sizeof what, "3x3 halo: %d images %d x %d, %d channels", NB, H, H, Cc);
        {
            static float* sc = nullptr;
            if (!sc) { const float one[2] = {1.f, 1.f}; CK(hipMalloc(&sc, 8)); CK(hipMemcpy(sc, one, 8, hipMemcpyHostToDevice)); }
            g.a_absmax = sc; g.w_scale = sc + 1;
        }
        if (only3 && argv[1][1] == 'p') {                 // "3p": the pair arm only (any bytes as planes: timing)
            if (Cc == 64) {
                const V vs[] = {{"fpair / 256-row tiles", launch3p<2, 0, 2>}, {"  no B DMA in the loop", launch3p<2, 36, 2>}, {"  ... and no barriers", launch3p<2, 52, 2>},
                                {"  ... and no patch work", launch3p<2, 55, 2>}, {"fpair / 128-row tiles", launch3p<1, 0, 2>}};
                if (run(what, g, vs, 5, 2.0 * M * N * K, junk)) return 1;
            } else {
                const V vs[] = {{"fpair / 256-row tiles", launch3p<2, 0>}, {"  no B DMA in the loop", launch3p<2, 36>}, {"  ... and no barriers", launch3p<2, 52>},
                                {"  ... and no patch work", launch3p<2, 55>}, {"fpair / 128-row tiles", launch3p<1, 0>}};
                if (run(what, g, vs, 5, 2.0 * M * N * K, junk)) return 1;
            }
            CK(hipFree(dd));
            continue;
        }
        if (Cc == 64) {
            const V vs[] = {{"full / 256-row tiles", launch3<2, 0, 2>}, {"  no split, no plane stores", launch3<2, 1, 2>}, {"  ... and no patch loads", launch3<2, 3, 2>}, {"  THREE products of the six (rest as full)", launch3<2, 64, 2>}};
            if (run(what, g, vs, 4, 2.0 * M * N * K, junk)) return 1;
        } else {
            const V vs[] = {{"full / 256-row tiles", launch3<2, 0>}, {"  no split, no plane stores", launch3<2, 1>}, {"  ... and no patch loads", launch3<2, 3>}, {"  THREE products of the six (rest as full)", launch3<2, 64>},
                            {"fTHREE products / 256-row tiles", launch3<2, 64>}, {"  no B DMA in the loop (fragments from buffer 0)", launch3<2, 64 + 4 + 32>},
                            {"  ... and no barriers", launch3<2, 64 + 4 + 32 + 16>}, {"  ... and no patch loads / split / stores", launch3<2, 64 + 4 + 32 + 16 + 3>},
                            {"  barriers and DMA, but no patch work", launch3<2, 64 + 3>},
                            {"fTHREE products / 128-row tiles", launch3<1, 64>}, {"  no B DMA in the loop (fragments from buffer 0)", launch3<1, 64 + 4 + 32>},
                            {"  ... and no barriers", launch3<1, 64 + 4 + 32 + 16>}, {"  ... and no patch loads / split / stores", launch3<1, 64 + 4 + 32 + 16 + 3>}};
            if (run(what, g, vs, 13, 2.0 * M * N * K, junk)) return 1;
        }
        CK(hipFree(dd));
    }
    return 0;
}
