// common.h -- shared helpers for the gfx950 kernels of libvnet_hip.so
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include "../../include/vnet_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
// two fp32 -> packed bf16 pair, round-to-nearest-even (the rounding of every bf16 operand in this library)
__device__ __forceinline__ uint32_t pk_bf16(float lo, float hi) {
    uint32_t r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}

#define VNET_LAUNCH_CHECK()                                  \
    do {                                                     \
        hipError_t e__ = hipGetLastError();                  \
        if (e__ != hipSuccess) return (int)e__;              \
    } while (0)

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
static inline int round_up(int a, int b) { return ceil_div(a, b) * b; }
static inline size_t align_up(size_t a, size_t b) { return (a + b - 1) / b * b; }

// 64-lane butterfly sum (wave = 64 on gfx950)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// XCD-aware bijective remap of a linear workgroup id (8 XCDs, block b runs on XCD b%8):
// consecutive remapped ids share an XCD/L2, so neighbouring bricks (which share halo voxels and
// all weights) hit the same L2.  Speed only; any placement is correct.
__device__ __forceinline__ int xcd_remap(int orig, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = orig & 7, k = orig >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
}
