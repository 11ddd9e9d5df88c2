// Probe of ds_read_b64_tr_b16 (gfx950): which LDS element lands in (lane, j)?
// LDS holds element ids 0..N-1 as uint16; lane l supplies byte address addr[l]; prints src id per (lane, j).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void probe(const int* addr, unsigned short* out) {
    __shared__ __attribute__((aligned(16))) unsigned short lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (unsigned short)i;
    __syncthreads();
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)((__attribute__((address_space(3))) char*)lds + addr[threadIdx.x]));
    for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = (unsigned short)v[j];
}
int main() {
    int h_addr[64]; unsigned short h_out[256];
    // lane l: row = (l&15)>>2 (+ 4 rows per 16-lane group), col quad = l&3 ; row stride 32 B (16 bf16)
    for (int l = 0; l < 64; ++l) h_addr[l] = (((l >> 4) * 4 + ((l & 15) >> 2)) * 32) + (l & 3) * 8;
    int* d_addr; unsigned short* d_out;
    hipMalloc(&d_addr, sizeof(h_addr)); hipMalloc(&d_out, sizeof(h_out));
    hipMemcpy(d_addr, h_addr, sizeof(h_addr), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d_addr, d_out);
    hipMemcpy(h_out, d_out, sizeof(h_out), hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; ++l) {
        printf("lane %2d addr %4d :", l, h_addr[l]);
        for (int j = 0; j < 4; ++j) printf("  elem %4d (row %2d col %2d)", h_out[l * 4 + j], h_out[l * 4 + j] / 16, h_out[l * 4 + j] % 16);
        printf("\n");
    }
    return 0;
}
