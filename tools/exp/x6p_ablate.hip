// Ablation harness for gemm_x6p_kernel (not part of the library): times the kernel with parts of its loop switched off.
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

template <int WM, int ABL, bool AREG = true, bool ILV = true>
static void launch(const X6PArgs& g, hipStream_t st) {
    const int tm = 128 * WM, nrb = (g.M + tm - 1) / tm;
    hipLaunchKernelGGL((gemm_x6p_kernel<WM, ABL, AREG, ILV>), dim3(8 * ((nrb + 7) / 8) * (g.N / PN)), dim3(256), 0, st, g);
}

int main() {
    const int shapes[][3] = {{16384, 2048, 512}, {16384, 2048, 4096}};
    float *A, *C, *W; unsigned char *Bp, *junk;
    CK(hipMalloc(&A, (size_t)200704 * 2048 * 4)); CK(hipMalloc(&C, (size_t)200704 * 2048 * 4));
    CK(hipMalloc(&W, (size_t)2048 * 4096 * 4)); CK(hipMalloc(&Bp, (size_t)2048 * 4096 * 6)); CK(hipMalloc(&junk, 512u << 20));
    fill<<<4096, 256>>>(A, (size_t)200704 * 2048, 1, 1.f);
    fill<<<4096, 256>>>(W, (size_t)2048 * 4096, 2, 0.05f);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (auto& sh : shapes) {
        const int M = sh[0], N = sh[1], K = sh[2];
        PackDesc d{(int64_t)W, (int64_t)Bp, N, K, K, 0, 0, 0}, *dd;
        CK(hipMalloc(&dd, sizeof(d))); CK(hipMemcpy(dd, &d, sizeof(d), hipMemcpyHostToDevice));
        x6_pack_kernel<<<(N / 128) * (K / 16), 256>>>(dd, 1);
        X6PArgs g{};
        g.A = A; g.Bp = Bp; g.out = C; g.M = M; g.N = N; g.K = K; g.lda = K; g.ldo = N; g.stride = 1;
        g.stream_out = (size_t)M * N * 4 > ((size_t)64 << 20);
        struct V { const char* name; void (*fn)(const X6PArgs&, hipStream_t); };
        const V vs[] = {{"full/256", launch<2, 0>}, {"full/128", launch<1, 0>},
                        {"mfma only/256", launch<2, 7>}, {"mfma only/128", launch<1, 7>},
                        {"mfma only, no barrier/256", launch<2, 23>}, {"mfma only, no barrier/128", launch<1, 23>},
                        {"mfma only, no barrier, one B buffer/256", launch<2, 55>}, {"mfma only, no barrier, one B buffer/128", launch<1, 55>}};
        printf("M=%d N=%d K=%d  (ideal MFMA time %.1f us)\n", M, N, K, 2.0 * M * N * K * 6 / 2.5e15 * 1e6);
        for (auto& v : vs) {
            std::vector<float> ts;
            for (int r = 0; r < 9; ++r) {
                CK(hipMemsetAsync(junk, r, 512u << 20, 0));
                CK(hipEventRecord(e0, 0)); v.fn(g, 0); CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ts.push_back(ms * 1e3f);
            }
            std::sort(ts.begin() + 1, ts.end());
            printf("   %-20s %7.1f us (min %7.1f)  %6.1f TF\n", v.name, ts[5], ts[1], 2.0 * M * N * K / ts[5] / 1e6);
        }
        CK(hipFree(dd));
    }
    return 0;
}
