#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef short v4s __attribute__((ext_vector_type(4)));
__global__ void k(int mode, int pitch_bytes, short* out) {
    __shared__ __attribute__((aligned(16))) unsigned short lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (unsigned short)i;
    __syncthreads();
    const int l = threadIdx.x;
    // lane's byte address
    int addr;
    if (mode == 0) addr = l * 8;                         // lane-linear 8 bytes each
    else if (mode == 1) addr = (l & 15) * pitch_bytes + (l >> 4) * 8;   // lane i of a 16-group -> row i; group g -> 4 columns g*4..
    else addr = (l & 15) * 8 + (l >> 4) * pitch_bytes;
    typedef __attribute__((address_space(3))) v4s* lp;
    v4s r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(lds) + 0 + 0 * 0 + ((addr) / 8));
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = r[j];
}
int main() {
    short* d; hipMalloc(&d, 64 * 4 * 2);
    short h[256];
    for (int mode = 0; mode < 3; ++mode) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, mode, 64, d);
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("mode %d (element indices read; lds[i] = i, 2 bytes each)\n", mode);
        for (int l = 0; l < 64; ++l) { printf("  lane %2d: %4d %4d %4d %4d", l, h[l*4], h[l*4+1], h[l*4+2], h[l*4+3]); if (l % 4 == 3) printf("\n"); }
    }
    return 0;
}
