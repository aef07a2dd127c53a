// How many workgroups with L KiB of static LDS does a CU hold?  256 threads, a fixed spin; the time of 2 x 256 (3 x, 4 x) workgroups
// against 256 tells whether they shared the CUs.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int LDS_BYTES>
__global__ __launch_bounds__(256) void spin(int iters, float* out) {
    __shared__ unsigned char lds[LDS_BYTES];
    float f = threadIdx.x;
    lds[threadIdx.x] = 1;
    __syncthreads();
#pragma unroll 1
    for (int i = 0; i < iters; ++i) f = __builtin_fmaf(f, 1.0001f, (float)lds[(threadIdx.x + i) & 255]);
    if (f == 12345.f) out[0] = f;
}
template <int L>
void run(float* out) {
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    float ms[5];
    int k = 0;
    for (int wgs : {256, 512, 768, 1024}) {
        for (int rep = 0; rep < 2; ++rep) {
            (void)hipEventRecord(a);
            hipLaunchKernelGGL(spin<L>, dim3(wgs), dim3(256), 0, 0, 20000, out);
            (void)hipEventRecord(b);
            (void)hipEventSynchronize(b);
            (void)hipEventElapsedTime(&ms[k], a, b);
        }
        ++k;
    }
    printf("LDS %6d bytes: 256 workgroups %.3f ms | 512: %.3f | 768: %.3f | 1024: %.3f\n", L, ms[0], ms[1], ms[2], ms[3]);
}
int main() {
    float* out;
    (void)hipMalloc(&out, 4);
    run<16384>(out); run<40960>(out); run<49152>(out); run<53248>(out); run<57344>(out); run<61440>(out); run<65536>(out);
    return 0;
}
