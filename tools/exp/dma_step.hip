// The conv_h main loop without its arithmetic: per step a workgroup of four waves waits for the W chunk requested D steps ago
// (hand-counted vmcnt), meets at a barrier and requests the next chunk (8 KiB: two 1 KiB LDS-DMA instructions per wave) from an
// L2-resident window.  Clocks per step for D = 1, 2, 3 and with / without the barrier: is the LDS-DMA latency of the real
// kernel (2 600 clocks seen) a property of this pattern?
//   hipcc --offload-arch=gfx950 -O3 -o dma_step dma_step.hip && ./dma_step
#include <hip/hip_runtime.h>
#include <cstdio>

__device__ __forceinline__ void dma16(const void* src, unsigned lds_off) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(src), "s"(lds_off) : "memory", "m0");
}

template <int D, int BAR, int SPIN>
__global__ __launch_bounds__(256) void steps(const unsigned char* src, size_t window, int iters, unsigned long long* out) {
    __shared__ __attribute__((aligned(1024))) unsigned char lds[(D + 1) * 8192];
    typedef __attribute__((address_space(3))) unsigned char* lptr_t;
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(__UINTPTR_TYPE__)(lptr_t)lds);
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned char* base = src + (size_t)blockIdx.x * 65536 % window;
    auto issue = [&](int t) {
        const unsigned char* s = base + ((size_t)t * 8192) % window + wave * 2048 + lane * 16;
        const unsigned d = lds0 + (t % (D + 1)) * 8192 + wave * 2048;
        dma16(s, d);
        dma16(s + 1024, d + 1024);
    };
    for (int q = 0; q < D; ++q) issue(q);
    const unsigned long long t0 = __builtin_readcyclecounter();
    float f = (float)lane;
    typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf8;
    typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f16v;
    f16v acc[8];
    for (int q = 0; q < 8; ++q) for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
    bf8 ra, rb;
    int vx[8] = {lane, 1, 2, 3, 4, 5, 6, 7};
    int sx[4] = {1, 2, 3, 4};
    for (int r = 0; r < 8; ++r) { ra[r] = (__bf16)(float)lane; rb[r] = (__bf16)1.0f; }
    for (int t = 0; t < iters; ++t) {
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * (D - 1)) : "memory");
        if (BAR) __builtin_amdgcn_s_barrier();
        issue(t + D);
        if (SPIN == 1) {                                   // LDS reads of the landed chunk + 16 products (as the ring kernel's step)
            const unsigned char* sb = lds + (t % (D + 1)) * 8192 + lane * 16;
            bf8 af0 = *reinterpret_cast<const bf8*>(sb), af1 = *reinterpret_cast<const bf8*>(sb + 1024);
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int y = 0; y < 4; ++y) {
                    const bf8 b = *reinterpret_cast<const bf8*>(sb + (2 * y + kk) * 1024);
                    acc[2 * y] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af0, b, acc[2 * y], 0, 0, 0);
                    acc[2 * y + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af1, b, acc[2 * y + 1], 0, 0, 0);
                }
        } else if (SPIN >= 3) {                            // products + (SPIN - 2) * 32 independent VALU / SALU instructions
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[q & 7] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ra, rb, acc[q & 7], 0, 0, 0);
            if (SPIN < 10) {
#pragma unroll
                for (int q = 0; q < (SPIN - 2) * 32; ++q) asm volatile("v_add_u32 %0, %0, %1" : "+v"(vx[q & 7]) : "v"(lane));
            } else {
#pragma unroll
                for (int q = 0; q < (SPIN - 10) * 32; ++q) asm volatile("s_add_u32 %0, %0, 3" : "+s"(sx[q & 3]) :: "scc");
            }
        } else if (SPIN == 2) {                            // the products alone, operands in registers
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[q & 7] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ra, rb, acc[q & 7], 0, 0, 0);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_readcyclecounter();
    __syncthreads();
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = t1 - t0; out[1] = (unsigned long long)f + lds[5]; }
    if (vx[0] + vx[1] + vx[2] + vx[3] + vx[4] + vx[5] + vx[6] + vx[7] + sx[0] + sx[1] + sx[2] + sx[3] == 77) out[1] = 9;
    if (acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3] + acc[4][4] + acc[5][5] + acc[6][6] + acc[7][7] == 12345.f) out[1] = 7;
}

template <int D, int BAR, int SPIN>
void run(const unsigned char* buf, size_t window, int wgs, unsigned long long* out) {
    unsigned long long h[2];
    const int iters = 2000;
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL((steps<D, BAR, SPIN>), dim3(wgs), dim3(256), 0, 0, buf, window, iters, out);
        (void)hipMemcpy(h, out, 16, hipMemcpyDeviceToHost);
    }
    printf("D %d barrier %d arithmetic mode %d, window %6zu KiB, %4d workgroups: %7.0f clocks per step\n", D, BAR, SPIN, window >> 10, wgs,
           (double)h[0] / iters);
}

int main() {
    unsigned char* buf;
    unsigned long long* out;
    (void)hipMalloc(&buf, 1ull << 30);
    (void)hipMemset(buf, 1, 1ull << 30);
    (void)hipMalloc(&out, 16);
    for (int wgs : {256, 512}) {
        run<2, 1, 2>(buf, 1 << 20, wgs, out);
        run<2, 1, 3>(buf, 1 << 20, wgs, out);       // + 32 VALU
        run<2, 1, 4>(buf, 1 << 20, wgs, out);       // + 64 VALU
        run<2, 1, 6>(buf, 1 << 20, wgs, out);       // + 128 VALU
        run<2, 1, 12>(buf, 1 << 20, wgs, out);      // + 64 SALU
        run<2, 1, 14>(buf, 1 << 20, wgs, out);      // + 128 SALU
    }
    return 0;
}
