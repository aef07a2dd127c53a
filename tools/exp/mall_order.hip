// Does a consumer that walks a just-written tensor BACKWARDS (most recently written rows first) find them in the memory-side
// cache (256 MB)?  writer: streams S bytes front to back; reader: sums them front to back or back to front.  Times of the reader.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(256) void writer(uint4* p, size_t n16) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) p[i] = make_uint4(i, 1, 2, 3);
}
__global__ __launch_bounds__(256) void writer_blocked(uint4* p, size_t n16) {          // workgroup b owns a contiguous slice (as row blocks)
    const size_t per = (n16 + gridDim.x - 1) / gridDim.x, b0 = (size_t)blockIdx.x * per;
    for (size_t i = b0 + threadIdx.x; i < b0 + per && i < n16; i += 256) p[i] = make_uint4(i, 1, 2, 3);
}
template <int BACK>
__global__ __launch_bounds__(256) void reader(const uint4* p, size_t n16, unsigned* out) {
    const size_t per = (n16 + gridDim.x - 1) / gridDim.x;
    const size_t b = BACK ? gridDim.x - 1 - blockIdx.x : blockIdx.x, b0 = b * per;
    unsigned s = 0;
    for (size_t i = b0 + threadIdx.x; i < b0 + per && i < n16; i += 256) { const uint4 v = p[i]; s += v.x ^ v.y ^ v.z ^ v.w; }
    if (s == 0x12345u) out[0] = s;
}
int main() {
    uint4* buf; unsigned* out;
    (void)hipMalloc(&buf, 1ull << 30); (void)hipMalloc(&out, 4);
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    for (size_t mb : {103, 205, 411, 822}) {
        const size_t n16 = mb * 1000000 / 16;
        const int wgs = (int)(n16 / 256 / 16);          // 16 x 4 KiB per workgroup
        for (int back = 0; back < 2; ++back) {
            float best = 1e9f;
            for (int rep = 0; rep < 5; ++rep) {
                hipLaunchKernelGGL(writer_blocked, dim3(wgs), dim3(256), 0, 0, buf, n16);
                (void)hipEventRecord(a);
                if (back) hipLaunchKernelGGL(reader<1>, dim3(wgs), dim3(256), 0, 0, buf, n16, out);
                else hipLaunchKernelGGL(reader<0>, dim3(wgs), dim3(256), 0, 0, buf, n16, out);
                (void)hipEventRecord(b); (void)hipEventSynchronize(b);
                float ms; (void)hipEventElapsedTime(&ms, a, b);
                best = ms < best ? ms : best;
            }
            printf("%4zu MB written front to back, read %s: %7.1f us  %5.2f TB/s\n", mb, back ? "back to front" : "front to back", best * 1e3, mb / best / 1e3);
        }
    }
    return 0;
}
