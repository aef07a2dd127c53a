// Experiment (round 5): does the SEGMENT a store instruction writes per row matter to HBM?  out[m][n] (fp32, N columns) written tile by
// tile as the GEMM epilogues do: a workgroup owns TR rows x 128 columns; a store instruction of a wave covers 64 / LPR rows x (LPR x 16)
// bytes: LPR = 8 lanes per row = 128-byte pieces (what gemm_x6p / conv_h write), 16 = 256 B, 32 = 512 B (the whole 128-column row).
// Optionally reads a [M][K] input first (1 read : N / K writes).   hipcc --offload-arch=gfx950 -O3 seg_write.hip -o seg_write
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float v4f __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s\n", hipGetErrorString(e)); return 1; } } while (0)

template <int LPR, int NT>
__global__ __launch_bounds__(256) void k_seg(float* o, int M, int N) {
    constexpr int TR = 128, TC = 128;
    const int nct = N / TC;
    const int j = blockIdx.x / 8;
    const int rb = 8 * (j / nct) + (int)(blockIdx.x % 8), ct = j % nct;
    if (rb * TR >= M) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int RPI = 64 / LPR;                       // rows per store instruction
    const int er = lane / LPR, ec = (lane % LPR) * 4;
    const v4f v = {1.f, 2.f, 3.f, (float)lane};
    // wave w owns rows w * 32 .. + 31 of the tile; passes over the 128 columns in pieces of LPR * 4
    for (int c0 = 0; c0 < TC; c0 += LPR * 4)
        for (int r0 = 0; r0 < 32; r0 += RPI) {
            const size_t m = (size_t)rb * TR + wave * 32 + r0 + er;
            float* p = o + m * N + ct * TC + c0 + ec;
            if (NT) __builtin_nontemporal_store(v, reinterpret_cast<v4f*>(p)); else *reinterpret_cast<v4f*>(p) = v;
        }
}

int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 802816, N = argc > 2 ? atoi(argv[2]) : 256;
    const size_t n = (size_t)M * N;
    float* o; char* junk;
    CK(hipMalloc(&o, n * 4)); CK(hipMalloc(&junk, (size_t)512 << 20));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto run = [&](const char* name, auto launch) {
        float tot = 0, best = 1e9;
        for (int r = 0; r < 7; ++r) {
            (void)hipMemsetAsync(junk, r, (size_t)512 << 20, 0);
            (void)hipEventRecord(e0); launch(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            if (r >= 2) { tot += ms; best = ms < best ? ms : best; }
        }
        printf("%-26s avg %7.1f us  -> %5.2f TB/s (best %5.2f)\n", name, tot / 5 * 1e3, n * 4.0 / (tot / 5 * 1e-3) / 1e12, n * 4.0 / (best * 1e-3) / 1e12);
    };
    const int nrb = (M + 127) / 128;
    const dim3 grid(8 * ((nrb + 7) / 8) * (N / 128));
    printf("M = %d N = %d: %.0f MB written in 128 x 128 tiles\n", M, N, n * 4 / 1e6);
    run("128 B pieces plain", [&] { hipLaunchKernelGGL((k_seg<8, 0>), grid, dim3(256), 0, 0, o, M, N); });
    run("256 B pieces plain", [&] { hipLaunchKernelGGL((k_seg<16, 0>), grid, dim3(256), 0, 0, o, M, N); });
    run("512 B pieces plain", [&] { hipLaunchKernelGGL((k_seg<32, 0>), grid, dim3(256), 0, 0, o, M, N); });
    run("128 B pieces nt", [&] { hipLaunchKernelGGL((k_seg<8, 1>), grid, dim3(256), 0, 0, o, M, N); });
    run("256 B pieces nt", [&] { hipLaunchKernelGGL((k_seg<16, 1>), grid, dim3(256), 0, 0, o, M, N); });
    run("512 B pieces nt", [&] { hipLaunchKernelGGL((k_seg<32, 1>), grid, dim3(256), 0, 0, o, M, N); });
    return 0;
}
