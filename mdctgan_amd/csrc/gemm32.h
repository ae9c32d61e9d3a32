// 32-deep chunk of the exact-f32 MFMA GEMM over row-major LDS images (row pitch 36 floats, k contiguous): shared by the
// forward convolution kernel (conv_igemm.hip) and the MDCT GEMM kernel (mdct.hip).  Included inside each translation
// unit's anonymous namespace.
#pragma once

__device__ __forceinline__ float4 g32_ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }

constexpr int BK2 = 32, LDK2 = BK2 + 4;

template <int MB, int NB, typename F0, typename F1>
__device__ __forceinline__ void mma_chunk32(const float* Ap, const float* Bp, f32x16 (&acc)[MB][NB], int wm0, int wn0,
                                            int lane, F0&& after_s0, F1&& after_s1) {
    const int r = lane & 31, kh = lane >> 5;
    const float* ap = Ap + (wm0 + r) * LDK2 + 4 * kh;
    const float* bp = Bp + (wn0 + r) * LDK2 + 4 * kh;
    constexpr int NS = BK2 / 8;
    float4 a[2][MB], b[2][NB];
#pragma unroll
    for (int mi = 0; mi < MB; ++mi) a[0][mi] = g32_ld4(ap + 32 * mi * LDK2);
#pragma unroll
    for (int ni = 0; ni < NB; ++ni) b[0][ni] = g32_ld4(bp + 32 * ni * LDK2);
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const int cur = s & 1, nxt = cur ^ 1;
        if (s + 1 < NS) {
#pragma unroll
            for (int mi = 0; mi < MB; ++mi) a[nxt][mi] = g32_ld4(ap + 32 * mi * LDK2 + 8 * (s + 1));
#pragma unroll
            for (int ni = 0; ni < NB; ++ni) b[nxt][ni] = g32_ld4(bp + 32 * ni * LDK2 + 8 * (s + 1));
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int mi = 0; mi < MB; ++mi)
#pragma unroll
            for (int ni = 0; ni < NB; ++ni) {
                acc[mi][ni] = mfma32x32x2(a[cur][mi].x, b[cur][ni].x, acc[mi][ni]);
                acc[mi][ni] = mfma32x32x2(a[cur][mi].y, b[cur][ni].y, acc[mi][ni]);
                acc[mi][ni] = mfma32x32x2(a[cur][mi].z, b[cur][ni].z, acc[mi][ni]);
                acc[mi][ni] = mfma32x32x2(a[cur][mi].w, b[cur][ni].w, acc[mi][ni]);
            }
        __builtin_amdgcn_sched_barrier(0);
        if (s == 0) after_s0();
        if (s == 1) after_s1();
        if (s <= 1) __builtin_amdgcn_sched_barrier(0);
    }
}


// f16-compute variant: the same row pitch, but a row holds its 32 k as halves in the first 64 bytes (written as 8-byte
// groups of four by the stage); lane (r, kh) reads the 8 halves k = 16 s + 8 kh .. +7 of step s with one ds_read_b128
// and feeds one v_mfma_f32_32x32x16_f16 per accumulator tile -- two steps per chunk.
__device__ __forceinline__ void g32_st_h4(float* row_base, int q, const float4 v) {       // k = 4q .. 4q+3 of this row
    uint2 pk;
    pk.x = pack_h2(v.x, v.y);
    pk.y = pack_h2(v.z, v.w);
    *reinterpret_cast<uint2*>(reinterpret_cast<char*>(row_base) + 8 * q) = pk;
}
template <int MB, int NB, typename F0, typename F1>
__device__ __forceinline__ void mma_chunk32_h(const float* Ap, const float* Bp, f32x16 (&acc)[MB][NB], int wm0, int wn0,
                                              int lane, F0&& after_s0, F1&& after_s1) {
    const int r = lane & 31, kh = lane >> 5;
    const char* ap = reinterpret_cast<const char*>(Ap + (wm0 + r) * LDK2) + 16 * kh;
    const char* bp = reinterpret_cast<const char*>(Bp + (wn0 + r) * LDK2) + 16 * kh;
    f16x8 a[2][MB], b[2][NB];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
#pragma unroll
        for (int mi = 0; mi < MB; ++mi) a[s][mi] = *reinterpret_cast<const f16x8*>(ap + 32 * mi * LDK2 * 4 + 32 * s);
#pragma unroll
        for (int ni = 0; ni < NB; ++ni) b[s][ni] = *reinterpret_cast<const f16x8*>(bp + 32 * ni * LDK2 * 4 + 32 * s);
    }
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int mi = 0; mi < MB; ++mi)
#pragma unroll
            for (int ni = 0; ni < NB; ++ni) acc[mi][ni] = mfma32x32x16h(a[s][mi], b[s][ni], acc[mi][ni]);
    __builtin_amdgcn_sched_barrier(0);
    after_s0();
    after_s1();
}
