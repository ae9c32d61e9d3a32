// Shared device helpers for the mdctGAN gfx950 kernels.  CDNA4 only: wave = 64 lanes,
// f32-input MFMA (v_mfma_f32_32x32x2_f32), 160 KiB LDS per CU.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define MG_OK 0
#define MG_ERR_ARG (-1)      // bad argument (shape / alignment / null)
#define MG_ERR_UNSUPPORTED (-2)

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// D(32x32) += A(32x2) * B(2x32), exact f32.  Lane l supplies A[i = l & 31][k = l >> 5] and
// B[k = l >> 5][j = l & 31]; result reg r of lane l is D[(r & 3) + 8 * (r >> 2) + 4 * (l >> 5)][l & 31].
__device__ __forceinline__ f32x16 mfma32x32x2(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}
// f16-input MFMA (v_mfma_f32_32x32x16_f16): D(32x32) += A(32x16) * B(16x32), exact f16 products, f32 accumulate.
// Lane l supplies 8 halves of row i = l & 31 of A (and of column j = l & 31 of B) for the k-group l >> 5; A and B use
// the same slot -> k map, so any chunk layout that gives both operands the same 8 k per (lane half, slot) is correct.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x16 mfma32x32x16h(f16x8 a, f16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ uint32_t pack_h2(float lo, float hi) {      // round-to-nearest-even, like Tensor.half()
    f16x2 h = {(_Float16)lo, (_Float16)hi};
    return __builtin_bit_cast(uint32_t, h);
}
__device__ __forceinline__ float round_h(float v) { return (float)(_Float16)v; }
__device__ __forceinline__ int mfma32_row(int reg, int lane) { return (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5); }

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
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
    return v;
}

// Monotone float <-> uint encoding so atomicMin/atomicMax on uint32 order like floats.
__device__ __forceinline__ unsigned f2ord(float f) {
    unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord2f(unsigned u) {
    return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u);
}

// XCD-aware block remap (8 XCDs, block b runs on XCD b % 8): gives each XCD a contiguous
// chunk of the logical tile space so neighbouring tiles share its private L2.  Bijective for
// any grid size.
__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
    const int q = nblk >> 3, r = nblk & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}

#define MG_CHECK_LAUNCH()                              \
    do {                                               \
        hipError_t e__ = hipGetLastError();            \
        if (e__ != hipSuccess) return (int)e__;        \
    } while (0)
