// How fast does a CU fill LDS from L2 / from HBM?  LDS-DMA (global_load_lds_dwordx4) against register loads + ds_write_b128.
// Decides the operand path of the 16-bit convolution kernels (conv_h.hip): their tiles move 24 KiB per 512 MFMA cycles.
//   hipcc --offload-arch=gfx950 -O3 -o dma_rate dma_rate.hip && ./dma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__device__ __forceinline__ void dma16(const void* src, unsigned lds_off) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(src), "s"(lds_off) : "memory", "m0");
}

// mode 0: DMA; mode 1: register loads + ds_write_b128.  Every wave moves `pieces` KiB per iteration from its workgroup's
// window of `span` bytes (span small -> L2 / MALL resident; span = whole buffer slice -> streamed from HBM).
template <int MODE, int PIECES>
__global__ __launch_bounds__(256, 2) void fill(const unsigned char* src, size_t wg_stride, size_t span, int iters, unsigned* sink) {
    __shared__ __attribute__((aligned(1024))) unsigned char lds[4 * PIECES * 1024 * 2];
    typedef __attribute__((address_space(3))) unsigned char* lptr_t;
    const unsigned lds0 = (unsigned)(__UINTPTR_TYPE__)(lptr_t)lds;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned char* base = src + (size_t)blockIdx.x * wg_stride;
    size_t off = (size_t)wave * PIECES * 1024;
    unsigned acc = 0;
    uint4 v[PIECES];
    if constexpr (MODE == 1) {
#pragma unroll
        for (int p = 0; p < PIECES; ++p) v[p] = *reinterpret_cast<const uint4*>(base + off + p * 1024 + lane * 16);
    }
    for (int it = 0; it < iters; ++it) {
        const unsigned buf = (it & 1) * 4 * PIECES * 1024 + wave * PIECES * 1024;
        if constexpr (MODE == 0) {
#pragma unroll
            for (int p = 0; p < PIECES; ++p) dma16(base + off + p * 1024 + lane * 16, lds0 + buf + p * 1024);
            asm volatile("s_waitcnt vmcnt(%0)" :: "n"(PIECES) : "memory");    // previous iteration's pieces have landed
        }
        off += 4 * PIECES * 1024;
        if (off + 4 * PIECES * 1024 > span) off = (size_t)wave * PIECES * 1024;
        if constexpr (MODE == 1) {       // software pipeline: next iteration's loads in flight while this one's are stored
            uint4 w[PIECES];
#pragma unroll
            for (int p = 0; p < PIECES; ++p) w[p] = *reinterpret_cast<const uint4*>(base + off + p * 1024 + lane * 16);
#pragma unroll
            for (int p = 0; p < PIECES; ++p) *reinterpret_cast<uint4*>(lds + buf + p * 1024 + lane * 16) = v[p];
#pragma unroll
            for (int p = 0; p < PIECES; ++p) v[p] = w[p];
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    acc = reinterpret_cast<unsigned*>(lds)[threadIdx.x];
    if (acc == 0x12345678u) sink[0] = acc;
}

template <int MODE, int PIECES>
static void run(const char* what, const unsigned char* buf, size_t bytes, size_t span, int wgs, int iters, unsigned* sink) {
    const size_t stride = span >= bytes / wgs ? bytes / wgs : span;   // small span: neighbouring windows; large: disjoint slices
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(a);
        hipLaunchKernelGGL((fill<MODE, PIECES>), dim3(wgs), dim3(256), 0, 0, buf, stride, span, iters, sink);
        hipEventRecord(b);
        hipEventSynchronize(b);
    }
    float ms; hipEventElapsedTime(&ms, a, b);
    const double moved = (double)wgs * iters * 4 * PIECES * 1024;
    printf("%-34s %5d WGs x %5d iters x %2d KiB: %8.1f us  %7.2f TB/s  %6.1f B/clk/CU @2.4GHz\n", what, wgs, iters, 4 * PIECES,
           ms * 1e3, moved / ms / 1e9, moved / (ms * 1e-3) / 256 / 2.4e9);
}

int main() {
    const size_t bytes = (size_t)4 << 30;
    unsigned char* buf; unsigned* sink;
    hipMalloc(&buf, bytes); hipMalloc(&sink, 64);
    hipMemset(buf, 1, bytes);
    for (int wgs : {256, 512}) {
        // L2-resident: every workgroup cycles over a 64 KiB window (512 x 64 KiB = 32 MiB: L2 + MALL)
        run<0, 2>("DMA  2 KiB/wave/iter, 64 KiB window", buf, bytes, 64 << 10, wgs, 4000, sink);
        run<0, 4>("DMA  4 KiB/wave/iter, 64 KiB window", buf, bytes, 64 << 10, wgs, 2000, sink);
        run<0, 6>("DMA  6 KiB/wave/iter, 96 KiB window", buf, bytes, 96 << 10, wgs, 1500, sink);
        run<1, 2>("REG  2 KiB/wave/iter, 64 KiB window", buf, bytes, 64 << 10, wgs, 4000, sink);
        run<1, 4>("REG  4 KiB/wave/iter, 64 KiB window", buf, bytes, 64 << 10, wgs, 2000, sink);
        run<1, 6>("REG  6 KiB/wave/iter, 96 KiB window", buf, bytes, 96 << 10, wgs, 1500, sink);
        // streamed from HBM: disjoint 8 / 4 MiB slices
        run<0, 4>("DMA  4 KiB/wave/iter, HBM stream", buf, bytes, bytes / wgs, wgs, (int)(bytes / wgs / 16384), sink);
        run<0, 6>("DMA  6 KiB/wave/iter, HBM stream", buf, bytes, bytes / wgs, wgs, (int)(bytes / wgs / 24576), sink);
        run<1, 4>("REG  4 KiB/wave/iter, HBM stream", buf, bytes, bytes / wgs, wgs, (int)(bytes / wgs / 16384), sink);
        run<1, 6>("REG  6 KiB/wave/iter, HBM stream", buf, bytes, bytes / wgs, wgs, (int)(bytes / wgs / 24576), sink);
    }
    return 0;
}
