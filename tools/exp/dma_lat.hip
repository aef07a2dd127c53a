// Latency of ONE LDS-DMA instruction (global_load_lds_dwordx4, 1 KiB per wave) against a plain 16-byte-per-lane load, one wave on
// an otherwise idle chip: from a window that stays in L2, and from lines never touched before (HBM).  Shader clocks.
//   hipcc --offload-arch=gfx950 -O3 -o dma_lat dma_lat.hip && ./dma_lat
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void lat(const unsigned char* src, size_t stride, int iters, int mode, unsigned long long* out) {
    __shared__ __attribute__((aligned(1024))) unsigned char lds[4096];
    typedef __attribute__((address_space(3))) unsigned char* lptr_t;
    const unsigned lds0 = (unsigned)(__UINTPTR_TYPE__)(lptr_t)lds;
    const int lane = threadIdx.x;
    unsigned long long tot = 0;
    unsigned sink = 0;
    for (int it = 0; it < iters; ++it) {
        const unsigned char* p = src + (size_t)it * stride + lane * 16;
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        const unsigned long long t0 = __builtin_readcyclecounter();
        if (mode == 0) {
            asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off\n\ts_waitcnt vmcnt(0)" :: "v"(p), "s"(lds0) : "memory", "m0");
        } else {
            uint4 v;
            asm volatile("global_load_dwordx4 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
            sink += v.x;
        }
        const unsigned long long t1 = __builtin_readcyclecounter();
        tot += t1 - t0;
    }
    if (lane == 0) { out[0] = tot; out[1] = sink + lds[0]; }
}

int main() {
    unsigned char* buf;
    unsigned long long* out;
    const size_t bytes = 1ull << 30;
    hipMalloc(&buf, bytes);
    hipMemset(buf, 1, bytes);
    hipMalloc(&out, 16);
    unsigned long long h[2];
    const int iters = 2000;
    for (int mode = 0; mode < 2; ++mode) {
        for (int pass = 0; pass < 3; ++pass) {     // stride 0: the same 1 KiB (L2 after the first touch); 1 KiB within 256 KiB window via second run
            hipLaunchKernelGGL(lat, dim3(1), dim3(64), 0, 0, buf, (size_t)0, iters, mode, out);
            hipMemcpy(h, out, 16, hipMemcpyDeviceToHost);
        }
        printf("%s same KiB (L2/TCP hit): %6.0f clocks\n", mode == 0 ? "LDS-DMA  " : "register ", (double)h[0] / iters);
        hipLaunchKernelGGL(lat, dim3(1), dim3(64), 0, 0, buf + (size_t)(mode + 1) * (256u << 20), (size_t)65536, iters, mode, out);
        hipMemcpy(h, out, 16, hipMemcpyDeviceToHost);
        printf("%s fresh lines (HBM):     %6.0f clocks\n", mode == 0 ? "LDS-DMA  " : "register ", (double)h[0] / iters);
    }
    // L2 hit but not TCP: walk a 2 MiB window twice
    for (int mode = 0; mode < 2; ++mode) {
        for (int pass = 0; pass < 2; ++pass) {
            hipLaunchKernelGGL(lat, dim3(1), dim3(64), 0, 0, buf + (768u << 20), (size_t)1024, 2048, mode, out);
            hipMemcpy(h, out, 16, hipMemcpyDeviceToHost);
        }
        printf("%s 2 MiB window, second walk (L2): %6.0f clocks\n", mode == 0 ? "LDS-DMA  " : "register ", (double)h[0] / 2048);
    }
    return 0;
}
