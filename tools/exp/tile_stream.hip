// Experiment (round 5): what does the ACCESS PATTERN of the entry-gradient GEMM's epilogue reach, without the GEMM?
// out[m][n] = d[m][n] + x[m][n] over an [M][N] fp32 matrix (2 reads + 1 write per element, like addend + BatchNorm-x + output),
// with the rows handed out as the GEMM kernels hand them out:
//   lin      grid-stride over 16-byte words (what bn2d_* kernels do)
//   tile     workgroup = TR rows x TC columns (lane: 16 bytes of a row, 8 lanes = 128 B; rows er + 8 jj), column tiles of a row block
//            on consecutive workgroups -- exactly the x6p epilogue's geometry -- U rows per lane in flight
//   rows     workgroup = TR rows x ALL columns, walked column tile by column tile by the same workgroup
// each with plain / non-temporal accesses.   hipcc --offload-arch=gfx950 -O3 tile_stream.hip -o tile_stream
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float v4f __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s\n", hipGetErrorString(e)); return 1; } } while (0)

template <int NT> __device__ __forceinline__ v4f ld(const float* a) {
    return NT ? __builtin_nontemporal_load(reinterpret_cast<const v4f*>(a)) : *reinterpret_cast<const v4f*>(a);
}
template <int NT> __device__ __forceinline__ void st(float* a, v4f x) {
    if (NT) __builtin_nontemporal_store(x, reinterpret_cast<v4f*>(a)); else *reinterpret_cast<v4f*>(a) = x;
}

template <int NT, int U>
__global__ __launch_bounds__(256) void k_lin(const float* d, const float* x, float* o, size_t n4) {
    for (size_t base = (size_t)blockIdx.x * U * 256 + threadIdx.x; base < n4; base += (size_t)gridDim.x * U * 256) {
        v4f a[U], b[U];
#pragma unroll
        for (int j = 0; j < U; ++j) { const size_t k = base + (size_t)j * 256; if (k < n4) { a[j] = ld<NT>(d + 4 * k); b[j] = ld<NT>(x + 4 * k); } }
#pragma unroll
        for (int j = 0; j < U; ++j) { const size_t k = base + (size_t)j * 256; if (k < n4) st<NT>(o + 4 * k, a[j] + b[j]); }
    }
}

// workgroup tile TR rows x TC columns (TC = 32 per "column tile pass" of a wave: lane = (er = lane >> 3, ec = 4 (lane & 7)), rows er + 8 jj)
// wave w of 4 owns rows w * TR / 4 ...; passes over TC / 32 column slices one after the other (as the epilogue's y loop)
template <int NT, int TR, int TC>
__global__ __launch_bounds__(256) void k_tile(const float* d, const float* x, float* o, int M, int N) {
    const int nct = N / TC;
    const int j = blockIdx.x / 8;
    const int rb = 8 * (j / nct) + (int)(blockIdx.x % 8), ct = j % nct;
    if (rb * TR >= M) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int er = lane >> 3, ec = (lane & 7) * 4;
    constexpr int RW = TR / 4;                       // rows per wave
    for (int y = 0; y < TC / 32; ++y)
        for (int a = 0; a < RW / 32; ++a) {
            v4f dv[4], xv[4];
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const size_t m = (size_t)rb * TR + wave * RW + a * 32 + er + 8 * jj;
                const size_t off = m * N + ct * TC + y * 32 + ec;
                dv[jj] = ld<NT>(d + off); xv[jj] = ld<0>(x + off);
            }
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const size_t m = (size_t)rb * TR + wave * RW + a * 32 + er + 8 * jj;
                st<NT>(o + m * N + ct * TC + y * 32 + ec, dv[jj] + xv[jj]);
            }
        }
}

// workgroup = TR rows x all N columns: lane = 16 bytes, a wave reads 1 KiB of ONE row per instruction (N = 256: the whole row),
// U rows per lane in flight
template <int NT, int TR, int U>
__global__ __launch_bounds__(256) void k_rows(const float* d, const float* x, float* o, int M, int N) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int per_row = N / 256;                     // 1 KiB pieces per row
    const size_t row0 = (size_t)blockIdx.x * TR + wave * (TR / 4);
    for (int p = 0; p < per_row; ++p)
        for (int r = 0; r < TR / 4; r += U) {
            v4f dv[U], xv[U];
#pragma unroll
            for (int u = 0; u < U; ++u) { const size_t off = (row0 + r + u) * N + p * 256 + lane * 4; dv[u] = ld<NT>(d + off); xv[u] = ld<0>(x + off); }
#pragma unroll
            for (int u = 0; u < U; ++u) { const size_t off = (row0 + r + u) * N + p * 256 + lane * 4; st<NT>(o + off, dv[u] + xv[u]); }
        }
}

int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 802816, N = argc > 2 ? atoi(argv[2]) : 256;
    const size_t n = (size_t)M * N, n4 = n / 4;
    float *d, *x, *o; char* junk;
    CK(hipMalloc(&d, n * 4)); CK(hipMalloc(&x, n * 4)); CK(hipMalloc(&o, n * 4)); CK(hipMalloc(&junk, (size_t)512 << 20));
    CK(hipMemset(d, 0, n * 4)); CK(hipMemset(x, 0, n * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const double bytes = 12.0 * n;
    auto run = [&](const char* name, auto launch) {
        float tot = 0, best = 1e9;
        for (int r = 0; r < 7; ++r) {
            (void)hipMemsetAsync(junk, r, (size_t)512 << 20, 0);            // flush the memory-side cache
            (void)hipEventRecord(e0); launch(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            if (r >= 2) { tot += ms; best = ms < best ? ms : best; }
        }
        printf("%-34s avg %7.1f us  -> %5.2f TB/s (best %5.2f)\n", name, tot / 5 * 1e3, bytes / (tot / 5 * 1e-3) / 1e12, bytes / (best * 1e-3) / 1e12);
    };
    printf("M = %d N = %d: %.0f MB per tensor, 2 reads + 1 write\n", M, N, n * 4 / 1e6);
    run("lin U=8 plain grid 4096", [&] { hipLaunchKernelGGL((k_lin<0, 8>), dim3(4096), dim3(256), 0, 0, d, x, o, n4); });
    run("lin U=8 nt    grid 4096", [&] { hipLaunchKernelGGL((k_lin<1, 8>), dim3(4096), dim3(256), 0, 0, d, x, o, n4); });
    run("lin U=4 nt    grid 2048", [&] { hipLaunchKernelGGL((k_lin<1, 4>), dim3(2048), dim3(256), 0, 0, d, x, o, n4); });
    {
        const int nrb = (M + 127) / 128;
        run("tile 128x64 plain", [&] { hipLaunchKernelGGL((k_tile<0, 128, 64>), dim3(8 * ((nrb + 7) / 8) * (N / 64)), dim3(256), 0, 0, d, x, o, M, N); });
        run("tile 128x64 nt", [&] { hipLaunchKernelGGL((k_tile<1, 128, 64>), dim3(8 * ((nrb + 7) / 8) * (N / 64)), dim3(256), 0, 0, d, x, o, M, N); });
        run("tile 128x128 nt", [&] { hipLaunchKernelGGL((k_tile<1, 128, 128>), dim3(8 * ((nrb + 7) / 8) * (N / 128)), dim3(256), 0, 0, d, x, o, M, N); });
        if (N % 256 == 0) run("tile 128x256 nt", [&] { hipLaunchKernelGGL((k_tile<1, 128, 256>), dim3(8 * ((nrb + 7) / 8) * (N / 256)), dim3(256), 0, 0, d, x, o, M, N); });
        const int nrb2 = (M + 255) / 256;
        run("tile 256x64 nt", [&] { hipLaunchKernelGGL((k_tile<1, 256, 64>), dim3(8 * ((nrb2 + 7) / 8) * (N / 64)), dim3(256), 0, 0, d, x, o, M, N); });
    }
    if (N % 256 == 0 && M % 128 == 0) {
        run("rows 128 x N, U=4 nt", [&] { hipLaunchKernelGGL((k_rows<1, 128, 4>), dim3(M / 128), dim3(256), 0, 0, d, x, o, M, N); });
        run("rows 128 x N, U=8 nt", [&] { hipLaunchKernelGGL((k_rows<1, 128, 8>), dim3(M / 128), dim3(256), 0, 0, d, x, o, M, N); });
        run("rows 64 x N, U=8 nt", [&] { hipLaunchKernelGGL((k_rows<1, 64, 8>), dim3(M / 64), dim3(256), 0, 0, d, x, o, M, N); });
        run("rows 128 x N, U=8 plain", [&] { hipLaunchKernelGGL((k_rows<0, 128, 8>), dim3(M / 128), dim3(256), 0, 0, d, x, o, M, N); });
    }
    return 0;
}
