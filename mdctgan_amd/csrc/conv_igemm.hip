// K3 / K4 / K6: implicit-GEMM convolution (forward, data-gradient == transposed convolution,
// weight-gradient) on the gfx950 f32 MFMA pipe (v_mfma_f32_32x32x2_f32, exact float32).
//
// Replaces nn.Conv2d / nn.ConvTranspose2d / nn.ReflectionPad2d forward + backward of the reference
// (models/networks.py:207-210, 308-309, 329, 349-352, 387-392, 406-411, 440, 456, 649-670).
//
// Layout: activations NHWC, weights OHWI ([Co][KH][KW][Ci]).  Every pass is one GEMM
//      FWD    Y[m=(b,oy,ox)][n=co]     = sum_{k=(ky,kx,ci)} Xgather[m][k]  * W[n][k]
//      DGRAD  dX[m=(b,iy,ix)][n=ci]    = sum_{k=(ky,kx,co)} dYgather[m][k] * W[k.co][k.tap][n]
//      WGRAD  dW[co][n=(ky,kx,ci)]     = sum_{k=(b,oy,ox)}  dY[k][co]      * Xgather[k][n]
// The gathers fold zero padding, ReflectionPad2d (forward: index reflection; backward: the up-to-4
// padded positions that alias an interior pixel are summed in the loader), the stride (DGRAD runs
// one launch slice per output-parity class so no MFMA work is spent on structural zeros) and the
// transposed-convolution geometry into address generation: no padded / dilated tensor ever exists.
//
// Tile: BM x BN x 16, 256 threads = 2x2 waves, each wave (BM/2)x(BN/2) as 32x32 MFMA blocks.
// LDS tiles are k-major ([16][rows + 4]): the MFMA fragment read is one conflict-free ds_read_b32 per
// operand per MFMA (lanes 0-31 consecutive rows), k-contiguous global float4s are scattered with
// 2-way (free) ds_write_b32, row-contiguous ones land as ds_write_b128.  Register-staged double
// buffering, one barrier per k-chunk; with 64-cycle MFMAs the loader hides completely.
#include "common.h"
#include "mdctgan_hip.h"
#include <stdio.h>
#include <stdlib.h>
#include <cstring>
#include <type_traits>
#include <map>
#include <mutex>
#include <tuple>
#include <hip/hip_ext.h>

namespace {

constexpr int BK = 16;

struct Geom {
    int B, H, W, Ci, OH, OW, Co, KH, KW, s, p, reflect;
};

// Batched launches (the 16 Winograd positions run as one grid): element strides between consecutive problems.
// wino_perm: the data-gradient uses the 180-degree rotated filter, which in the Winograd domain is the position
// permutation xi -> (3,1,2,0)[xi] on both axes (G * flip == P * G).
struct Batch {
    long long sa, sw, so;
    int wino_perm;
};
__device__ __forceinline__ int wino_flip(int z) {
    const int xi = z >> 2, nu = z & 3;
    const int fx = (xi == 0) ? 3 : (xi == 3 ? 0 : xi), fn = (nu == 0) ? 3 : (nu == 3 ? 0 : nu);
    return fx * 4 + fn;
}

__device__ __forceinline__ int reflect_idx(int i, int n) {
    if (i < 0) i = -i;
    if (i >= n) i = 2 * (n - 1) - i;
    return i;
}

__device__ __forceinline__ float apply_act(float v, int act) {
    if (act == MG_ACT_RELU) return fmaxf(v, 0.0f);
    if (act == MG_ACT_LRELU02) return v > 0.0f ? v : 0.2f * v;
    if (act == MG_ACT_TANH) return tanhf(v);
    return v;
}

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float4 zero4() { return make_float4(0.f, 0.f, 0.f, 0.f); }
__device__ __forceinline__ void add4(float4& a, const float4 b) { a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; }

// k-contiguous float4 (4 consecutive k of one row) -> k-major LDS tile.
// P = 1 (f16 compute, TAG 2 kernels): the tile holds halves, two consecutive k of a row packed in one 32-bit word at
// [k / 2][row] (same pitch in words), so a lane's 8 k of one MFMA step are 4 words of its row.
template <int LD, int P = 0>
__device__ __forceinline__ void st_kcontig(float* S, int row, int q, const float4 v) {
    if (P) {
        uint32_t* W = reinterpret_cast<uint32_t*>(S);
        W[(2 * q + 0) * LD + row] = pack_h2(v.x, v.y);
        W[(2 * q + 1) * LD + row] = pack_h2(v.z, v.w);
    } else {
        S[(4 * q + 0) * LD + row] = v.x;
        S[(4 * q + 1) * LD + row] = v.y;
        S[(4 * q + 2) * LD + row] = v.z;
        S[(4 * q + 3) * LD + row] = v.w;
    }
}
// row-contiguous float4 (4 consecutive rows at one k)
template <int LD, int P = 0>
__device__ __forceinline__ void st_rowcontig(float* S, int k, int row, const float4 v) {
    if (P) {
        _Float16* H = reinterpret_cast<_Float16*>(S) + (((k >> 1) * LD + row) << 1) + (k & 1);
        H[0] = (_Float16)v.x;
        H[2] = (_Float16)v.y;
        H[4] = (_Float16)v.z;
        H[6] = (_Float16)v.w;
    } else {
        *reinterpret_cast<float4*>(S + k * LD + row) = v;
    }
}

// One 16-deep chunk on the f16 MFMA pipe: one 32x32x16 instruction per accumulator tile.
template <int MB, int NB, int LDA, int LDB, typename F0, typename F1>
__device__ __forceinline__ void mma_chunk_h(const float* Ap, const float* Bp, f32x16 (&acc)[MB][NB], int wm0, int wn0,
                                            int lane, F0&& after_kp0, F1&& after_kp1) {
    const int r = lane & 31, kh = lane >> 5;
    const uint32_t* ap = reinterpret_cast<const uint32_t*>(Ap) + (4 * kh) * LDA + wm0 + r;
    const uint32_t* bp = reinterpret_cast<const uint32_t*>(Bp) + (4 * kh) * LDB + wn0 + r;
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    u32x4 a[MB], b[NB];
#pragma unroll
    for (int mi = 0; mi < MB; ++mi)
#pragma unroll
        for (int j = 0; j < 4; ++j) a[mi][j] = ap[j * LDA + 32 * mi];
#pragma unroll
    for (int ni = 0; ni < NB; ++ni)
#pragma unroll
        for (int j = 0; j < 4; ++j) b[ni][j] = bp[j * LDB + 32 * ni];
#pragma unroll
    for (int mi = 0; mi < MB; ++mi)
#pragma unroll
        for (int ni = 0; ni < NB; ++ni)
            acc[mi][ni] = mfma32x32x16h(__builtin_bit_cast(f16x8, a[mi]), __builtin_bit_cast(f16x8, b[ni]), acc[mi][ni]);
    __builtin_amdgcn_sched_barrier(0);
    after_kp0();
    after_kp1();
}

// One 16-deep k-chunk of MFMAs, software-pipelined by hand (the compiler will not do either on its own):
//  * the A/B fragments of k-pair kp+1 are read from LDS into a second register set BEFORE the MFMAs of k-pair kp
//    are issued, so the ds_read latency sits under 256 cycles of queued MFMA work;
//  * the staging work of the NEXT chunks rides inside this chunk's MFMA stream: after k-pair 0 the registers that
//    hold chunk c+1 (loaded one chunk ago) are written to the idle LDS buffer (`after_kp0`), after k-pair 1 the
//    global loads + address arithmetic for chunk c+2 are issued into the same registers (`after_kp1`).  VALU /
//    VMEM / DS instructions issue while the 64-cycle MFMAs execute, instead of in a serial phase between chunks
//    (ablation on the 1024-channel layer: loads 17 %, LDS stores + barrier 5 % of the kernel before this).
template <int MB, int NB, int LDA, int LDB, typename F0, typename F1>
__device__ __forceinline__ void mma_chunk(const float* Ap, const float* Bp, f32x16 (&acc)[MB][NB], int wm0, int wn0,
                                          int lane, F0&& after_kp0, F1&& after_kp1) {
    const int r = lane & 31, kh = lane >> 5;
    const float* ap = Ap + kh * LDA + wm0 + r;
    const float* bp = Bp + kh * LDB + wn0 + r;
    constexpr int NKP = BK / 2, RING = 4, AHEAD = 3;   // fragments are read 3 k-pairs ahead of their MFMAs
    float a[RING][MB], b[RING][NB];
#pragma unroll
    for (int pk = 0; pk < AHEAD; ++pk) {
#pragma unroll
        for (int mi = 0; mi < MB; ++mi) a[pk][mi] = ap[2 * pk * LDA + 32 * mi];
#pragma unroll
        for (int ni = 0; ni < NB; ++ni) b[pk][ni] = bp[2 * pk * LDB + 32 * ni];
    }
#pragma unroll
    for (int kp = 0; kp < NKP; ++kp) {
        const int cur = kp % RING, nxt = (kp + AHEAD) % RING;
        if (kp + AHEAD < NKP) {
#pragma unroll
            for (int mi = 0; mi < MB; ++mi) a[nxt][mi] = ap[2 * (kp + AHEAD) * LDA + 32 * mi];
#pragma unroll
            for (int ni = 0; ni < NB; ++ni) b[nxt][ni] = bp[2 * (kp + AHEAD) * LDB + 32 * ni];
        }
        __builtin_amdgcn_sched_barrier(0);     // keep the prefetch ds_reads ahead of this k-pair's MFMAs
#pragma unroll
        for (int mi = 0; mi < MB; ++mi)
#pragma unroll
            for (int ni = 0; ni < NB; ++ni) acc[mi][ni] = mfma32x32x2(a[cur][mi], b[cur][ni], acc[mi][ni]);
        __builtin_amdgcn_sched_barrier(0);
        if (kp == 0) after_kp0();
        if (kp == 1) after_kp1();
        if (kp <= 1) __builtin_amdgcn_sched_barrier(0);
    }
}

// ================================================================================================
// FWD
// ================================================================================================
template <int BM, int BN, bool VEC, int TAG = 0>   // TAG bit 0: dense batched GEMM operands (1: Winograd F(2x2,3x3), 5: the
                                                   // 25-position families of wino4.h / wino42.h -- distinct symbols for profilers); bit 1: f16 compute
__global__ __launch_bounds__(256) void conv_fwd_kernel(Geom g, const float* __restrict__ x,
                                                       const float* __restrict__ w,
                                                       const float* __restrict__ bias, float* __restrict__ y,
                                                       int act, int chunks_per_split, float* __restrict__ part,
                                                       Batch bt) {
    constexpr int MB = BM / 64, NB = BN / 64, LDA = BM + 4, LDB = BN + 4;
    constexpr int NVA = BM * 4 / 256, NVB = BN * 4 / 256;
    __shared__ __attribute__((aligned(16))) float smem[2 * BK * (LDA + LDB)];
    auto As = [&](int buf) -> float* { return smem + buf * (BK * LDA); };
    auto Bs = [&](int buf) -> float* { return smem + 2 * BK * LDA + buf * (BK * LDB); };

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int M = g.B * g.OH * g.OW, N = g.Co, K = g.KH * g.KW * g.Ci;
    x += (size_t)blockIdx.z * bt.sa;
    w += (size_t)blockIdx.z * bt.sw;
    y += (size_t)blockIdx.z * bt.so;
    if (part) part += ((size_t)blockIdx.y * gridDim.z + blockIdx.z) * ((size_t)M * N);
    const int cpt = g.Ci / BK;   // chunks per tap (VEC)
    const int total_chunks = VEC ? g.KH * g.KW * cpt : (K + BK - 1) / BK;
    const int c_begin = blockIdx.y * chunks_per_split;
    const int nchunks = min(total_chunks, c_begin + chunks_per_split);   // exclusive end of this split
    const int tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN;
    const int t = xcd_remap(blockIdx.x, tiles_m * tiles_n);
    const int m0 = (t % tiles_m) * BM, n0 = (t / tiles_m) * BN;
    const int q = tid & 3, r0 = tid >> 2;

    int iy0[NVA], ix0[NVA], pb[NVA];
    bool va_ok[NVA];
#pragma unroll
    for (int i = 0; i < NVA; ++i) {
        const int m = m0 + r0 + 64 * i;
        va_ok[i] = m < M;
        const int mm = va_ok[i] ? m : 0;
        const int b = mm / (g.OH * g.OW), rem = mm - b * (g.OH * g.OW);
        const int oy = rem / g.OW, ox = rem - oy * g.OW;
        iy0[i] = oy * g.s - g.p;
        ix0[i] = ox * g.s - g.p;
        pb[i] = b * g.H * g.W;
    }

    auto src_pixel = [&](int i, int ky, int kx) -> int {   // -1: zero padding
        int iy = iy0[i] + ky, ix = ix0[i] + kx;
        if (g.reflect) {
            iy = reflect_idx(iy, g.H);
            ix = reflect_idx(ix, g.W);
        } else if (iy < 0 || iy >= g.H || ix < 0 || ix >= g.W) {
            return -1;
        }
        return pb[i] + iy * g.W + ix;
    };

    auto load_a = [&](int c, float4 (&va)[NVA]) {
        if (VEC) {
            const int tap = c / cpt, ci0 = (c - tap * cpt) * BK;
            const int ky = tap / g.KW, kx = tap - ky * g.KW;
#pragma unroll
            for (int i = 0; i < NVA; ++i) {
                va[i] = zero4();
                if (!va_ok[i]) continue;
                const int px = src_pixel(i, ky, kx);
                if (px >= 0) va[i] = ld4(x + (size_t)px * g.Ci + ci0 + 4 * q);
            }
        } else {
#pragma unroll
            for (int i = 0; i < NVA; ++i) {
                float e[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int k = c * BK + 4 * q + j;
                    e[j] = 0.0f;
                    if (va_ok[i] && k < K) {
                        const int tap = k / g.Ci, ci = k - tap * g.Ci;
                        const int ky = tap / g.KW, kx = tap - ky * g.KW;
                        const int px = src_pixel(i, ky, kx);
                        if (px >= 0) e[j] = x[(size_t)px * g.Ci + ci];
                    }
                }
                va[i] = make_float4(e[0], e[1], e[2], e[3]);
            }
        }
    };
    auto load_b = [&](int c, float4 (&vb)[NVB]) {
#pragma unroll
        for (int i = 0; i < NVB; ++i) {
            const int n = n0 + r0 + 64 * i;
            vb[i] = zero4();
            if (n >= N) continue;
            if (VEC) {
                vb[i] = ld4(w + (size_t)n * K + c * BK + 4 * q);
            } else {
                float e[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int k = c * BK + 4 * q + j;
                    e[j] = (k < K) ? w[(size_t)n * K + k] : 0.0f;
                }
                vb[i] = make_float4(e[0], e[1], e[2], e[3]);
            }
        }
    };

    f32x16 acc[MB][NB];
#pragma unroll
    for (int mi = 0; mi < MB; ++mi)
#pragma unroll
        for (int ni = 0; ni < NB; ++ni) acc[mi][ni] = f32x16{0};
    const int wm0 = (wave >> 1) * (BM / 2), wn0 = (wave & 1) * (BN / 2);

    float4 va[NVA], vb[NVB];
    auto stash = [&](int buf) {
#pragma unroll
        for (int i = 0; i < NVA; ++i) st_kcontig<LDA, ((TAG & 2) != 0)>(As(buf), r0 + 64 * i, q, va[i]);
#pragma unroll
        for (int i = 0; i < NVB; ++i) st_kcontig<LDB, ((TAG & 2) != 0)>(Bs(buf), r0 + 64 * i, q, vb[i]);
    };
    if (c_begin < nchunks) {
        load_a(c_begin, va);
        load_b(c_begin, vb);
        stash(0);
    }
    __syncthreads();
    if (c_begin + 1 < nchunks) {
        load_a(c_begin + 1, va);
        load_b(c_begin + 1, vb);
    }
    for (int c = c_begin; c < nchunks; ++c) {
        const int cur = (c - c_begin) & 1;
        auto f0 = [&]() { if (c + 1 < nchunks) stash(cur ^ 1); };
        auto f1 = [&]() { if (c + 2 < nchunks) { load_a(c + 2, va); load_b(c + 2, vb); } };
        if ((TAG & 2) != 0) mma_chunk_h<MB, NB, LDA, LDB>(As(cur), Bs(cur), acc, wm0, wn0, lane, f0, f1);
        else mma_chunk<MB, NB, LDA, LDB>(As(cur), Bs(cur), acc, wm0, wn0, lane, f0, f1);
        __syncthreads();
    }

#pragma unroll
    for (int mi = 0; mi < MB; ++mi)
#pragma unroll
        for (int ni = 0; ni < NB; ++ni) {
            const int col = n0 + wn0 + 32 * ni + (lane & 31);
            const float bv = (bias && !part && col < N) ? bias[col] : 0.0f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm0 + 32 * mi + mfma32_row(r, lane);
                if (row < M && col < N) {
                    if (part) part[(size_t)row * N + col] = acc[mi][ni][r];
                    else {
                        const float v = apply_act(acc[mi][ni][r] + bv, act);
                        y[(size_t)row * N + col] = ((TAG & 2) != 0) ? round_h(v) : v;
                    }
                }
            }
        }
}

// ================================================================================================
// FWD, 32-deep chunks, row-major LDS images ("full-line staging").
// Both operands of the forward GEMM are k-contiguous in HBM (NHWC activations, OHWI weights, and the Winograd
// V/U matrices), so a chunk of 32 k is one full 128-byte line per row: a wave-wide global_load_dwordx4 covers
// 8 rows x 128 B instead of 16 rows x 64 B (half the texture-addresser work for the same bytes), the stage is one
// ds_write_b128 per 16 bytes instead of four ds_write_b32, and a lane fetches 4 consecutive k of its row with one
// ds_read_b128.  The MFMA k index inside a chunk is free as long as A and B agree: lane half `kh` owns
// k = 8*s + 4*kh + j for MFMA (s, j).  Row pitch 36 floats: 16-byte aligned rows, conflict-free for both the
// 8-lanes-per-row writes and the 32-rows-at-one-k reads.
// ================================================================================================
#include "gemm32.h"

template <int BM, int BN, int TAG = 0>
__global__ __launch_bounds__(256) void conv_fwd32_kernel(Geom g, const float* __restrict__ x,
                                                         const float* __restrict__ w,
                                                         const float* __restrict__ bias, float* __restrict__ y,
                                                         int act, int chunks_per_split, float* __restrict__ part,
                                                         Batch bt) {
    constexpr int MB = BM / 64, NB = BN / 64;
    constexpr int NVA = BM / 32, NVB = BN / 32;
    extern __shared__ __attribute__((aligned(16))) float smem32[];
    auto As = [&](int buf) -> float* { return smem32 + buf * (BM * LDK2); };
    auto Bs = [&](int buf) -> float* { return smem32 + 2 * BM * LDK2 + buf * (BN * LDK2); };

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int M = g.B * g.OH * g.OW, N = g.Co, K = g.KH * g.KW * g.Ci;
    x += (size_t)blockIdx.z * bt.sa;
    w += (size_t)blockIdx.z * bt.sw;
    y += (size_t)blockIdx.z * bt.so;
    if (part) part += ((size_t)blockIdx.y * gridDim.z + blockIdx.z) * ((size_t)M * N);
    const int cpt = g.Ci / BK2;
    const int total_chunks = g.KH * g.KW * cpt;
    const int c_begin = blockIdx.y * chunks_per_split;
    const int nchunks = min(total_chunks, c_begin + chunks_per_split);
    const int tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN;
    const int t = xcd_remap(blockIdx.x, tiles_m * tiles_n);
    const int m0 = (t % tiles_m) * BM, n0 = (t / tiles_m) * BN;
    const int q = tid & 7, r0 = tid >> 3;

    int iy0[NVA], ix0[NVA], pb[NVA];
    bool va_ok[NVA];
#pragma unroll
    for (int i = 0; i < NVA; ++i) {
        const int m = m0 + r0 + 32 * i;
        va_ok[i] = m < M;
        const int mm = va_ok[i] ? m : 0;
        const int b = mm / (g.OH * g.OW), rem = mm - b * (g.OH * g.OW);
        const int oy = rem / g.OW, ox = rem - oy * g.OW;
        iy0[i] = oy * g.s - g.p;
        ix0[i] = ox * g.s - g.p;
        pb[i] = b * g.H * g.W;
    }
    const float* wrow[NVB];
#pragma unroll
    for (int i = 0; i < NVB; ++i) {
        const int n = n0 + r0 + 32 * i;
        wrow[i] = (n < N) ? w + (size_t)n * K + 4 * q : nullptr;
    }

    // TAG 1 / 3 (batched plain GEMM of the Winograd path): both operands are dense row-major matrices, so staging is a
    // pointer per row fixed for the whole K loop + one load per chunk -- no tap / padding / reflection arithmetic.  An
    // MFMA-bound loop pays ~2.6 cycles for every staging instruction even in the MFMAs' shadow (scripts/ubench), so the
    // instruction count of the stage, not its placement, is what the remaining efficiency hangs on.  Rows past M / N
    // re-read the last row: their accumulators are never stored.
    const float* pa[NVA];
    const float* pbq[NVB];
#pragma unroll
    for (int i = 0; i < NVA; ++i) pa[i] = x + (size_t)min(m0 + r0 + 32 * i, M - 1) * K + 4 * q;
#pragma unroll
    for (int i = 0; i < NVB; ++i) pbq[i] = w + (size_t)min(n0 + r0 + 32 * i, N - 1) * K + 4 * q;
    int a_ky, a_kx, a_ci0;       // position of the next chunk load_a will stage
    {
        const int tap = c_begin / cpt;
        a_ci0 = (c_begin - tap * cpt) * BK2;
        a_ky = tap / g.KW;
        a_kx = tap - a_ky * g.KW;
    }
    auto load_a = [&](int c, float4 (&va)[NVA]) {
        if (TAG & 1) {
#pragma unroll
            for (int i = 0; i < NVA; ++i) va[i] = ld4(pa[i] + (size_t)c * BK2);
            return;
        }
        // load_a runs once per chunk in increasing c, so (ky, kx, ci0) is a running wave-uniform state, not c / cpt
        const int ky = a_ky, kx = a_kx, ci0 = a_ci0;
        a_ci0 += BK2;
        if (a_ci0 >= g.Ci) {
            a_ci0 = 0;
            if (++a_kx == g.KW) { a_kx = 0; ++a_ky; }
        }
#pragma unroll
        for (int i = 0; i < NVA; ++i) {
            // branch-free: out-of-range taps load pixel (0, 0) of the row's sample and are masked afterwards
            int iy = iy0[i] + ky, ix = ix0[i] + kx;
            bool ok = va_ok[i];
            if (g.reflect) {
                iy = reflect_idx(iy, g.H);
                ix = reflect_idx(ix, g.W);
            } else {
                ok = ok && (unsigned)iy < (unsigned)g.H && (unsigned)ix < (unsigned)g.W;
            }
            iy = ok ? iy : 0;
            ix = ok ? ix : 0;
            // 32-bit element offset (geom_ok bounds every tensor below 2^31 elements): base + voffset addressing
            const float4 v = ld4(x + (unsigned)(pb[i] + iy * g.W + ix) * (unsigned)g.Ci + (unsigned)(ci0 + 4 * q));
            va[i] = ok ? v : zero4();
        }
    };
    auto load_b = [&](int c, float4 (&vb)[NVB]) {
        if (TAG & 1) {
#pragma unroll
            for (int i = 0; i < NVB; ++i) vb[i] = ld4(pbq[i] + (size_t)c * BK2);
            return;
        }
#pragma unroll
        for (int i = 0; i < NVB; ++i) vb[i] = wrow[i] ? ld4(wrow[i] + (size_t)c * BK2) : zero4();
    };

    f32x16 acc[MB][NB];
#pragma unroll
    for (int mi = 0; mi < MB; ++mi)
#pragma unroll
        for (int ni = 0; ni < NB; ++ni) acc[mi][ni] = f32x16{0};
    const int wm0 = (wave >> 1) * (BM / 2), wn0 = (wave & 1) * (BN / 2);

    float4 va[NVA], vb[NVB];
    auto stash = [&](int buf) {
#pragma unroll
        for (int i = 0; i < NVA; ++i) {
            if ((TAG & 2) != 0) g32_st_h4(As(buf) + (r0 + 32 * i) * LDK2, q, va[i]);
            else *reinterpret_cast<float4*>(As(buf) + (r0 + 32 * i) * LDK2 + 4 * q) = va[i];
        }
#pragma unroll
        for (int i = 0; i < NVB; ++i) {
            if ((TAG & 2) != 0) g32_st_h4(Bs(buf) + (r0 + 32 * i) * LDK2, q, vb[i]);
            else *reinterpret_cast<float4*>(Bs(buf) + (r0 + 32 * i) * LDK2 + 4 * q) = vb[i];
        }
    };
    if (c_begin < nchunks) {
        load_a(c_begin, va);
        load_b(c_begin, vb);
        stash(0);
    }
    __syncthreads();
    if (c_begin + 1 < nchunks) {
        load_a(c_begin + 1, va);
        load_b(c_begin + 1, vb);
    }
    for (int c = c_begin; c < nchunks; ++c) {
        const int cur = (c - c_begin) & 1;
        auto f0 = [&]() { if (c + 1 < nchunks) stash(cur ^ 1); };
        auto f1 = [&]() { if (c + 2 < nchunks) { load_a(c + 2, va); load_b(c + 2, vb); } };
        if ((TAG & 2) != 0) mma_chunk32_h<MB, NB>(As(cur), Bs(cur), acc, wm0, wn0, lane, f0, f1);
        else mma_chunk32<MB, NB>(As(cur), Bs(cur), acc, wm0, wn0, lane, f0, f1);
        __syncthreads();
    }

#pragma unroll
    for (int mi = 0; mi < MB; ++mi)
#pragma unroll
        for (int ni = 0; ni < NB; ++ni) {
            const int col = n0 + wn0 + 32 * ni + (lane & 31);
            const float bv = (bias && !part && col < N) ? bias[col] : 0.0f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm0 + 32 * mi + mfma32_row(r, lane);
                if (row < M && col < N) {
                    if (part) part[(size_t)row * N + col] = acc[mi][ni][r];
                    else {
                        const float v = apply_act(acc[mi][ni][r] + bv, act);
                        y[(size_t)row * N + col] = ((TAG & 2) != 0) ? round_h(v) : v;
                    }
                }
            }
        }
}

// ================================================================================================
// DGRAD (== transposed convolution forward).  blockIdx.z = output-parity class (stride^2 of them).
// ================================================================================================
template <int BM, int BN, bool VECA, bool VECB, int TAG = 0>
__global__ __launch_bounds__(256) void conv_dgrad_kernel(Geom g, const float* __restrict__ dy,
                                                         const float* __restrict__ w,
                                                         const float* __restrict__ bias,
                                                         float* __restrict__ dx, int act, int chunks_per_split,
                                                         float* __restrict__ part, Batch bt) {
    constexpr int MB = BM / 64, NB = BN / 64, LDA = BM + 4, LDB = BN + 4;
    constexpr int NVA = BM * 4 / 256, NVB = BN * 4 / 256;
    __shared__ __attribute__((aligned(16))) float smem[2 * BK * (LDA + LDB)];
    auto As = [&](int buf) -> float* { return smem + buf * (BK * LDA); };
    auto Bs = [&](int buf) -> float* { return smem + 2 * BK * LDA + buf * (BK * LDB); };

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int s = g.s;
    const int cls = blockIdx.z % (s * s), batch = blockIdx.z / (s * s);
    // Parity classes differ in tap count (3x3 stride 2: 1, 2, 2, 4 taps).  Workgroups are dispatched in blockIdx.z
    // order, so the class with the most taps goes first (longest-processing-time-first: the short ones fill the tail).
    int py = 0, px = 0;
    if (s > 1) {
        const int zi = cls / s, zj = cls - zi * s;
        const int t0y = (g.KH - (g.p % s) + s - 1) / s, t1y = (g.KH - ((1 + g.p) % s) + s - 1) / s;
        const int t0x = (g.KW - (g.p % s) + s - 1) / s, t1x = (g.KW - ((1 + g.p) % s) + s - 1) / s;
        const int hy = t1y > t0y ? 1 : 0, hx = t1x > t0x ? 1 : 0;      // the heavier parity per axis (s == 2)
        py = zi == 0 ? hy : 1 - hy;
        px = zj == 0 ? hx : 1 - hx;
    }
    const int Hc = (g.H - py + s - 1) / s, Wc = (g.W - px + s - 1) / s;   // pixels of this class
    const int M = g.B * Hc * Wc, N = g.Ci;
    dy += (size_t)batch * bt.sa;
    w += (size_t)(bt.wino_perm ? wino_flip(batch) : batch) * bt.sw;
    dx += (size_t)batch * bt.so;
    // split-K slabs are whole dx images (classes write disjoint pixels of the same slab)
    if (part) part += ((size_t)blockIdx.y * (gridDim.z / (s * s)) + batch) * ((size_t)g.B * g.H * g.W * N);
    const int tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN;
    if ((int)blockIdx.x >= tiles_m * tiles_n) return;
    // taps that reach this class: ky = ky0 + s*i
    const int ky0 = (py + g.p) % s, kx0 = (px + g.p) % s;
    const int nky = (g.KH - ky0 + s - 1) / s, nkx = (g.KW - kx0 + s - 1) / s;
    const int oyb = (py + g.p) / s, oxb = (px + g.p) / s;                 // oy = yy + oyb - tyi
    const int ntap = nky * nkx;
    const int Kc = ntap * g.Co;
    const int cpt = g.Co / BK;
    const int total_chunks = VECA ? ntap * cpt : (Kc + BK - 1) / BK;
    // stride > 1: the classes have different K, so each splits its own chunk range evenly over gridDim.y
    const int cps = (s > 1 && gridDim.y > 1) ? (total_chunks + (int)gridDim.y - 1) / (int)gridDim.y : chunks_per_split;
    const int c_begin = blockIdx.y * cps;
    const int nchunks = min(total_chunks, c_begin + cps);   // exclusive end of this split
    const int t = xcd_remap(blockIdx.x, tiles_m * tiles_n);
    const int m0 = (t % tiles_m) * BM, n0 = (t / tiles_m) * BN;
    const int q = tid & 3, r0 = tid >> 2;

    int yy[NVA], xx[NVA], bb[NVA];
    bool va_ok[NVA];
#pragma unroll
    for (int i = 0; i < NVA; ++i) {
        const int m = m0 + r0 + 64 * i;
        va_ok[i] = m < M;
        const int mm = va_ok[i] ? m : 0;
        bb[i] = mm / (Hc * Wc);
        const int rem = mm - bb[i] * (Hc * Wc);
        yy[i] = rem / Wc;
        xx[i] = rem - yy[i] * Wc;
    }

    // reflect (stride 1): the padded rows / cols that alias this pixel: i+p, p-i, 2(H-1)-i+p (fixed per row)
    int cy[NVA][3], cx[NVA][3];
    if (g.reflect) {
#pragma unroll
        for (int i = 0; i < NVA; ++i) {
            const int iy = yy[i], ix = xx[i], p = g.p;
            cy[i][0] = iy + p;
            cy[i][1] = (iy >= 1 && iy <= p) ? p - iy : -1000000;
            cy[i][2] = (iy >= g.H - 1 - p && iy <= g.H - 2) ? 2 * (g.H - 1) - iy + p : -1000000;
            cx[i][0] = ix + p;
            cx[i][1] = (ix >= 1 && ix <= p) ? p - ix : -1000000;
            cx[i][2] = (ix >= g.W - 1 - p && ix <= g.W - 2) ? 2 * (g.W - 1) - ix + p : -1000000;
        }
    }

    // gather of dY for row i and tap index (tyi, txi): float4 at channel offset co (VEC) or scalar
    auto gather4 = [&](int i, int tyi, int txi, int co) -> float4 {
        float4 v = zero4();
        if (!g.reflect) {
            const int oy = yy[i] + oyb - tyi, ox = xx[i] + oxb - txi;
            if (oy >= 0 && oy < g.OH && ox >= 0 && ox < g.OW)
                v = ld4(dy + (unsigned)((bb[i] * g.OH + oy) * g.OW + ox) * (unsigned)g.Co + (unsigned)co);
        } else {   // stride 1: iy = yy, ky = tyi.  Issue every (usually one) aliasing load first, sum afterwards:
                   // accumulating inside the branches would put a vmcnt wait behind each taken load.
            float4 t[9];
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                const int oy = cy[i][a] - tyi;
#pragma unroll
                for (int c2 = 0; c2 < 3; ++c2) {
                    const int ox = cx[i][c2] - txi;
                    const bool ok = (unsigned)oy < (unsigned)g.OH && (unsigned)ox < (unsigned)g.OW;
                    t[a * 3 + c2] = ok ? ld4(dy + (unsigned)((bb[i] * g.OH + oy) * g.OW + ox) * (unsigned)g.Co + (unsigned)co) : zero4();
                }
            }
#pragma unroll
            for (int j = 0; j < 9; ++j) add4(v, t[j]);
        }
        return v;
    };
    auto gather1 = [&](int i, int tyi, int txi, int co) -> float {
        float v = 0.0f;
        if (!g.reflect) {
            const int oy = yy[i] + oyb - tyi, ox = xx[i] + oxb - txi;
            if (oy >= 0 && oy < g.OH && ox >= 0 && ox < g.OW)
                v = dy[((size_t)(bb[i] * g.OH + oy) * g.OW + ox) * g.Co + co];
        } else {
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                const int oy = cy[i][a] - tyi;
                if (oy < 0 || oy >= g.OH) continue;
#pragma unroll
                for (int c2 = 0; c2 < 3; ++c2) {
                    const int ox = cx[i][c2] - txi;
                    if (ox < 0 || ox >= g.OW) continue;
                    v += dy[((size_t)(bb[i] * g.OH + oy) * g.OW + ox) * g.Co + co];
                }
            }
        }
        return v;
    };

    // TAG 1 / 3 (batched plain GEMM of the Winograd path: 1x1 geometry, one class, dense operands): a pointer per
    // staged row fixed for the whole K loop, one load per chunk (see conv_fwd32_kernel).  Rows / columns past the
    // edge re-read the last valid ones; their accumulators are never stored.
    const float* pa[NVA];
    const float* pbq[NVB];
    if (TAG & 1) {
#pragma unroll
        for (int i = 0; i < NVA; ++i) pa[i] = dy + (size_t)min(m0 + r0 + 64 * i, M - 1) * g.Co + 4 * q;
#pragma unroll
        for (int i = 0; i < NVB; ++i)
            pbq[i] = w + (size_t)(tid / (BN / 4) + i * (1024 / BN)) * g.Ci + min(n0 + 4 * (tid % (BN / 4)), N - 4);
    }
    auto load_a = [&](int c, float4 (&va)[NVA]) {
        if (TAG & 1) {
#pragma unroll
            for (int i = 0; i < NVA; ++i) va[i] = ld4(pa[i] + (size_t)c * BK);
            return;
        }
        if (VECA) {
            const int tap = c / cpt, co0 = (c - tap * cpt) * BK;
            const int tyi = tap / nkx, txi = tap - tyi * nkx;
#pragma unroll
            for (int i = 0; i < NVA; ++i) va[i] = va_ok[i] ? gather4(i, tyi, txi, co0 + 4 * q) : zero4();
        } else {
#pragma unroll
            for (int i = 0; i < NVA; ++i) {
                float e[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int k = c * BK + 4 * q + j;
                    e[j] = 0.0f;
                    if (va_ok[i] && k < Kc) {
                        const int tap = k / g.Co, co = k - tap * g.Co;
                        const int tyi = tap / nkx, txi = tap - tyi * nkx;
                        e[j] = gather1(i, tyi, txi, co);
                    }
                }
                va[i] = make_float4(e[0], e[1], e[2], e[3]);
            }
        }
    };
    // B[k=(tap,co)][n=ci] = W[co][ky][kx][ci]: contiguous along n
    const int bk_l = tid / (BN / 4), bn_q = tid % (BN / 4);       // NVB passes step k by 256/(BN/4)
    auto load_b = [&](int c, float4 (&vb)[NVB]) {
        if (TAG & 1) {
#pragma unroll
            for (int i = 0; i < NVB; ++i) vb[i] = ld4(pbq[i] + (size_t)c * BK * g.Ci);
            return;
        }
#pragma unroll
        for (int i = 0; i < NVB; ++i) {
            const int kl = bk_l + i * (1024 / BN);
            int tap, co;
            if (VECA) {
                tap = c / cpt;
                co = (c - tap * cpt) * BK + kl;
            } else {
                const int k = c * BK + kl;
                tap = k / g.Co;
                co = k - tap * g.Co;
                if (k >= Kc) { vb[i] = zero4(); continue; }
            }
            const int tyi = tap / nkx, txi = tap - tyi * nkx;
            const int ky = ky0 + s * tyi, kx = kx0 + s * txi;
            const float* wp = w + (unsigned)((co * g.KH + ky) * g.KW + kx) * (unsigned)g.Ci;
            const int n = n0 + 4 * bn_q;
            if (VECB) {
                vb[i] = (n < N) ? ld4(wp + n) : zero4();
            } else {
                float e[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) e[j] = (n + j < N) ? wp[n + j] : 0.0f;
                vb[i] = make_float4(e[0], e[1], e[2], e[3]);
            }
        }
    };

    f32x16 acc[MB][NB];
#pragma unroll
    for (int mi = 0; mi < MB; ++mi)
#pragma unroll
        for (int ni = 0; ni < NB; ++ni) acc[mi][ni] = f32x16{0};
    const int wm0 = (wave >> 1) * (BM / 2), wn0 = (wave & 1) * (BN / 2);

    float4 va[NVA], vb[NVB];
    auto stash = [&](int buf) {
#pragma unroll
        for (int i = 0; i < NVA; ++i) st_kcontig<LDA, ((TAG & 2) != 0)>(As(buf), r0 + 64 * i, q, va[i]);
#pragma unroll
        for (int i = 0; i < NVB; ++i) st_rowcontig<LDB, ((TAG & 2) != 0)>(Bs(buf), bk_l + i * (1024 / BN), 4 * bn_q, vb[i]);
    };
    if (c_begin < nchunks) {
        load_a(c_begin, va);
        load_b(c_begin, vb);
        stash(0);
    }
    __syncthreads();
    if (c_begin + 1 < nchunks) {
        load_a(c_begin + 1, va);
        load_b(c_begin + 1, vb);
    }
    for (int c = c_begin; c < nchunks; ++c) {
        const int cur = (c - c_begin) & 1;
        auto f0 = [&]() { if (c + 1 < nchunks) stash(cur ^ 1); };
        auto f1 = [&]() { if (c + 2 < nchunks) { load_a(c + 2, va); load_b(c + 2, vb); } };
        if ((TAG & 2) != 0) mma_chunk_h<MB, NB, LDA, LDB>(As(cur), Bs(cur), acc, wm0, wn0, lane, f0, f1);
        else mma_chunk<MB, NB, LDA, LDB>(As(cur), Bs(cur), acc, wm0, wn0, lane, f0, f1);
        __syncthreads();
    }

#pragma unroll
    for (int mi = 0; mi < MB; ++mi) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + wm0 + 32 * mi + mfma32_row(r, lane);
            if (m >= M) continue;
            const int b = m / (Hc * Wc), rem = m - b * (Hc * Wc);
            const int y2 = rem / Wc, x2 = rem - y2 * Wc;
            const size_t o = ((size_t)(b * g.H + y2 * s + py) * g.W + x2 * s + px) * g.Ci;
#pragma unroll
            for (int ni = 0; ni < NB; ++ni) {
                const int col = n0 + wn0 + 32 * ni + (lane & 31);
                if (col < N) {
                    if (part) {   // split-K: raw partial sums at the pixel's own offset in the slab
                        part[o + col] = acc[mi][ni][r];
                    } else {
                        const float bv = bias ? bias[col] : 0.0f;
                        const float v = apply_act(acc[mi][ni][r] + bv, act);
                        dx[o + col] = ((TAG & 2) != 0) ? round_h(v) : v;
                    }
                }
            }
        }
    }
}

// ================================================================================================
// WGRAD.  rows = co, cols = n = (ky,kx,ci), reduction over pixels m; blockIdx.z = split of m.
// ================================================================================================
template <int BM, int BN, bool VECA, bool VECB, int TAG = 0>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(Geom g, const float* __restrict__ x,
                                                         const float* __restrict__ dy, float* __restrict__ out,
                                                         int chunks_per_split, int accumulate, Batch bt) {
    constexpr int MB = BM / 64, NB = BN / 64, LDA = BM + 4, LDB = BN + 4;
    constexpr int NVA = BM * 4 / 256, NVB = BN * 4 / 256;
    __shared__ __attribute__((aligned(16))) float smem[2 * BK * (LDA + LDB)];
    auto As = [&](int buf) -> float* { return smem + buf * (BK * LDA); };
    auto Bs = [&](int buf) -> float* { return smem + 2 * BK * LDA + buf * (BK * LDB); };

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int Mtot = g.B * g.OH * g.OW;          // reduction length
    const int R = g.Co, N = g.KH * g.KW * g.Ci;  // output rows / cols
    x += (size_t)blockIdx.y * bt.sa;
    dy += (size_t)blockIdx.y * bt.sw;
    const int tiles_m = (R + BM - 1) / BM, tiles_n = (N + BN - 1) / BN;
    const int t = xcd_remap(blockIdx.x, tiles_m * tiles_n);
    const int r_0 = (t % tiles_m) * BM, n0 = (t / tiles_m) * BN;
    const int total_chunks = (Mtot + BK - 1) / BK;
    const int c_begin = blockIdx.z * chunks_per_split;
    const int c_end = min(total_chunks, c_begin + chunks_per_split);

    const int ak_l = tid / (BM / 4), a_q = tid % (BM / 4);
    const int bk_l = tid / (BN / 4), b_q = tid % (BN / 4);

    // this thread's B columns (fixed through the loop)
    int b_tap[4], b_ci[4];
    {
        const int n = n0 + 4 * b_q;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int nn = n + (VECB ? 0 : j);
            b_tap[j] = (nn < N) ? nn / g.Ci : -1;
            b_ci[j] = (nn < N) ? nn - b_tap[j] * g.Ci : 0;
        }
    }

    // TAG 1 / 3 with a reduction length that is a multiple of 16 (the Winograd tile count): dense [t][channel]
    // operands, a pointer per staged row + one load per chunk
    const bool plain = (TAG & 1) && (Mtot % BK == 0);
    const float* pa[NVA];
    const float* pbq[NVB];
    if (TAG & 1) {
#pragma unroll
        for (int i = 0; i < NVA; ++i) pa[i] = dy + (size_t)(ak_l + i * (1024 / BM)) * g.Co + min(r_0 + 4 * a_q, R - 4);
#pragma unroll
        for (int i = 0; i < NVB; ++i) pbq[i] = x + (size_t)(bk_l + i * (1024 / BN)) * g.Ci + min(n0 + 4 * b_q, N - 4);
    }
    auto load_a = [&](int c, float4 (&va)[NVA]) {
        if (plain) {
#pragma unroll
            for (int i = 0; i < NVA; ++i) va[i] = ld4(pa[i] + (size_t)c * BK * g.Co);
            return;
        }
#pragma unroll
        for (int i = 0; i < NVA; ++i) {
            const int m = c * BK + ak_l + i * (1024 / BM);
            const int co = r_0 + 4 * a_q;
            va[i] = zero4();
            if (m >= Mtot) continue;
            if (VECA) {
                if (co < R) va[i] = ld4(dy + (unsigned)m * (unsigned)g.Co + (unsigned)co);
            } else {
                float e[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) e[j] = (co + j < R) ? dy[(size_t)m * g.Co + co + j] : 0.0f;
                va[i] = make_float4(e[0], e[1], e[2], e[3]);
            }
        }
    };
    // This thread's B taps as (ky, kx) (fixed through the loop), and its reduction pixels (b, oy, ox): chunk c stages
    // pixel m = 16 c + row, so consecutive load_b calls (which come in increasing c, each chunk exactly once) advance
    // every staged pixel by 16 -- kept as a running (b, oy, ox) instead of two integer divisions per row and chunk.
    int b_ky[4], b_kx[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        b_ky[j] = b_tap[j] >= 0 ? b_tap[j] / g.KW : 0;
        b_kx[j] = b_tap[j] >= 0 ? b_tap[j] - b_ky[j] * g.KW : 0;
    }
    const int adv_oy = BK / g.OW, adv_ox = BK - adv_oy * g.OW;
    int pm[NVB], pbb[NVB], poy[NVB], pox[NVB];
#pragma unroll
    for (int i = 0; i < NVB; ++i) {
        pm[i] = c_begin * BK + bk_l + i * (1024 / BN);
        pbb[i] = pm[i] / (g.OH * g.OW);
        const int rem = pm[i] - pbb[i] * (g.OH * g.OW);
        poy[i] = rem / g.OW;
        pox[i] = rem - poy[i] * g.OW;
    }
    auto src_pixel = [&](int b, int oy, int ox, int ky, int kx) -> int {
        int iy = oy * g.s - g.p + ky, ix = ox * g.s - g.p + kx;
        if (g.reflect) {
            iy = reflect_idx(iy, g.H);
            ix = reflect_idx(ix, g.W);
        } else if (iy < 0 || iy >= g.H || ix < 0 || ix >= g.W) {
            return -1;
        }
        return (b * g.H + iy) * g.W + ix;
    };
    auto load_b = [&](int c, float4 (&vb)[NVB]) {
        if (plain) {
#pragma unroll
            for (int i = 0; i < NVB; ++i) vb[i] = ld4(pbq[i] + (size_t)c * BK * g.Ci);
            return;
        }
#pragma unroll
        for (int i = 0; i < NVB; ++i) {
            const int m = pm[i], b = pbb[i], oy = poy[i], ox = pox[i];
            // advance this row's pixel to the next chunk
            pm[i] += BK;
            pox[i] += adv_ox;
            poy[i] += adv_oy;
            if (pox[i] >= g.OW) { pox[i] -= g.OW; ++poy[i]; }
            while (poy[i] >= g.OH) { poy[i] -= g.OH; ++pbb[i]; }
            vb[i] = zero4();
            if (m >= Mtot) continue;
            if (VECB) {
                if (b_tap[0] < 0) continue;
                const int px = src_pixel(b, oy, ox, b_ky[0], b_kx[0]);
                if (px >= 0) vb[i] = ld4(x + (unsigned)px * (unsigned)g.Ci + (unsigned)b_ci[0]);
            } else {
                float e[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    e[j] = 0.0f;
                    if (b_tap[j] < 0) continue;
                    const int px = src_pixel(b, oy, ox, b_ky[j], b_kx[j]);
                    if (px >= 0) e[j] = x[(size_t)px * g.Ci + b_ci[j]];
                }
                vb[i] = make_float4(e[0], e[1], e[2], e[3]);
            }
        }
    };

    f32x16 acc[MB][NB];
#pragma unroll
    for (int mi = 0; mi < MB; ++mi)
#pragma unroll
        for (int ni = 0; ni < NB; ++ni) acc[mi][ni] = f32x16{0};
    const int wm0 = (wave >> 1) * (BM / 2), wn0 = (wave & 1) * (BN / 2);

    float4 va[NVA], vb[NVB];
    auto stash = [&](int buf) {
#pragma unroll
        for (int i = 0; i < NVA; ++i) st_rowcontig<LDA, ((TAG & 2) != 0)>(As(buf), ak_l + i * (1024 / BM), 4 * a_q, va[i]);
#pragma unroll
        for (int i = 0; i < NVB; ++i) st_rowcontig<LDB, ((TAG & 2) != 0)>(Bs(buf), bk_l + i * (1024 / BN), 4 * b_q, vb[i]);
    };
    if (c_begin < c_end) {
        load_a(c_begin, va);
        load_b(c_begin, vb);
        stash(0);
    }
    __syncthreads();
    if (c_begin + 1 < c_end) {
        load_a(c_begin + 1, va);
        load_b(c_begin + 1, vb);
    }
    for (int c = c_begin; c < c_end; ++c) {
        const int cur = (c - c_begin) & 1;
        auto f0 = [&]() { if (c + 1 < c_end) stash(cur ^ 1); };
        auto f1 = [&]() { if (c + 2 < c_end) { load_a(c + 2, va); load_b(c + 2, vb); } };
        if ((TAG & 2) != 0) mma_chunk_h<MB, NB, LDA, LDB>(As(cur), Bs(cur), acc, wm0, wn0, lane, f0, f1);
        else mma_chunk<MB, NB, LDA, LDB>(As(cur), Bs(cur), acc, wm0, wn0, lane, f0, f1);
        __syncthreads();
    }

    float* o = out + ((size_t)blockIdx.z * gridDim.y + blockIdx.y) * ((size_t)R * N) ;
    if (gridDim.z == 1) o = out + (size_t)blockIdx.y * bt.so;
#pragma unroll
    for (int mi = 0; mi < MB; ++mi)
#pragma unroll
        for (int ni = 0; ni < NB; ++ni) {
            const int col = n0 + wn0 + 32 * ni + (lane & 31);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = r_0 + wm0 + 32 * mi + mfma32_row(r, lane);
                if (row < R && col < N) {
                    const size_t idx = (size_t)row * N + col;
                    o[idx] = accumulate ? o[idx] + acc[mi][ni][r] : acc[mi][ni][r];
                }
            }
        }
}

// ================================================================================================
// Lean dense C[M][N] = A^T B for the Winograd-domain weight gradients (A = Md [K][M], B = V [K][N], K = tiles): the
// same 64x64 tile, k-major LDS images and hand-pipelined chunk as conv_wgrad_kernel<64, 64, true, true, 1>, without
// the convolution geometry (tap / pixel state, border cases, the generic loaders) that a K = 256 reduction -- 128 MFMAs
// per wave -- cannot amortise, and with NSUB 16-deep sub-chunks per barrier.  M multiple of 64, N of 64 NB, K of 16 * NSUB.
// grid: x = tiles (XCD-remapped), y = batch, z = K splits.
// ================================================================================================
template <int NSUB, int TAG, int NB = 1>      // TAG: 1 = F(2x2,3x3), 5 = the 25-position families; tile 64 x (64 NB)
__global__ __launch_bounds__(256) void dense_tn64_kernel(const float* __restrict__ A, const float* __restrict__ B,
                                                         float* __restrict__ C, int M, int N, int K, long long sa,
                                                         long long sb, long long sc, int cps) {
    constexpr int BKW = BK * NSUB, LDA = 64, BN = 64 * NB, LDB = BN;
    __shared__ __attribute__((aligned(16))) float smem[2 * BKW * (LDA + LDB)];
    auto As = [&](int buf) -> float* { return smem + buf * (BKW * LDA); };
    auto Bs = [&](int buf) -> float* { return smem + 2 * BKW * LDA + buf * (BKW * LDB); };
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tiles_m = M / 64, tiles_n = N / BN;
    const int t = xcd_remap(blockIdx.x, tiles_m * tiles_n);
    const int m0 = (t % tiles_m) * 64, n0 = (t / tiles_m) * BN;
    const int total_chunks = K / BKW;
    const int c_begin = blockIdx.z * cps, c_end = min(total_chunks, c_begin + cps);
    const int kl = tid >> 4, q = tid & 15;            // A: k row inside a 16-deep sub-chunk, float4 column
    const int klb = tid / (BN / 4), qb = tid % (BN / 4);   // B: 256 / (BN/4) k rows per pass
    constexpr int BROWS = 1024 / BN, NVB = BK / BROWS;
    const float* pa = A + (size_t)blockIdx.y * sa + (size_t)(c_begin * BKW + kl) * M + m0 + 4 * q;
    const float* pb = B + (size_t)blockIdx.y * sb + (size_t)(c_begin * BKW + klb) * N + n0 + 4 * qb;
    const size_t stepa = (size_t)BK * M;
    float4 va[NSUB], vb[NSUB][NVB];
    auto load = [&]() {
#pragma unroll
        for (int i = 0; i < NSUB; ++i) {
            va[i] = ld4(pa);
            pa += stepa;
#pragma unroll
            for (int j = 0; j < NVB; ++j) vb[i][j] = ld4(pb + (size_t)(j * BROWS) * N);
            pb += (size_t)BK * N;
        }
    };
    auto stash = [&](int buf) {
#pragma unroll
        for (int i = 0; i < NSUB; ++i) {
            *reinterpret_cast<float4*>(As(buf) + (kl + BK * i) * LDA + 4 * q) = va[i];
#pragma unroll
            for (int j = 0; j < NVB; ++j)
                *reinterpret_cast<float4*>(Bs(buf) + (klb + j * BROWS + BK * i) * LDB + 4 * qb) = vb[i][j];
        }
    };
    f32x16 acc[1][NB];
#pragma unroll
    for (int ni = 0; ni < NB; ++ni) acc[0][ni] = f32x16{0};
    const int wm0 = (wave >> 1) * 32, wn0 = (wave & 1) * (BN / 2);
    if (c_begin < c_end) {
        load();
        stash(0);
    }
    __syncthreads();
    if (c_begin + 1 < c_end) load();
    for (int c = c_begin; c < c_end; ++c) {
        const int cur = (c - c_begin) & 1;
        auto f0 = [&]() { if (c + 1 < c_end) stash(cur ^ 1); };
        auto f1 = [&]() { if (c + 2 < c_end) load(); };
        auto nop = [&]() {};
        mma_chunk<1, NB, LDA, LDB>(As(cur), Bs(cur), acc, wm0, wn0, lane, f0, f1);
#pragma unroll
        for (int sc_ = 1; sc_ < NSUB; ++sc_)
            mma_chunk<1, NB, LDA, LDB>(As(cur) + sc_ * BK * LDA, Bs(cur) + sc_ * BK * LDB, acc, wm0, wn0, lane, nop, nop);
        __syncthreads();
    }
    float* o = C + ((size_t)blockIdx.z * gridDim.y + blockIdx.y) * ((size_t)M * N);
    if (gridDim.z == 1) o = C + (size_t)blockIdx.y * sc;
#pragma unroll
    for (int ni = 0; ni < NB; ++ni) {
        const int col = n0 + wn0 + 32 * ni + (lane & 31);
#pragma unroll
        for (int r = 0; r < 16; ++r) o[(size_t)(m0 + wm0 + mfma32_row(r, lane)) * N + col] = acc[0][ni][r];
    }
}

// Lean dense C[M][N] = A B for the Winograd-domain data gradients (A = Md [M][K] with K = Co contiguous, B = U [K][N]):
// conv_dgrad_kernel<64, 64, true, true, 1>'s tile and chunk without the convolution geometry.  N, K multiples of 64 / 16;
// rows past M re-read the last row and are not stored.  grid: x = tiles (XCD-remapped), y = K splits, z = batch.
template <int TAG>
__global__ __launch_bounds__(256) void dense_nn64_kernel(const float* __restrict__ A, const float* __restrict__ B,
                                                         float* __restrict__ C, float* __restrict__ part, int M, int N,
                                                         int K, long long sa, long long sb, long long sc, int cps) {
    constexpr int LDA = 64 + 4, LDB = 64;           // A is scattered in with ds_write_b32 (2-way at pitch 68), B row-wise
    __shared__ __attribute__((aligned(16))) float smem[2 * BK * (LDA + LDB)];
    auto As = [&](int buf) -> float* { return smem + buf * (BK * LDA); };
    auto Bs = [&](int buf) -> float* { return smem + 2 * BK * LDA + buf * (BK * LDB); };
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tiles_m = (M + 63) / 64, tiles_n = N / 64;
    const int t = xcd_remap(blockIdx.x, tiles_m * tiles_n);
    const int m0 = (t % tiles_m) * 64, n0 = (t / tiles_m) * 64;
    const int total_chunks = K / BK;
    const int c_begin = blockIdx.y * cps, c_end = min(total_chunks, c_begin + cps);
    const int qa = tid & 3, ra = tid >> 2;            // A: 4 consecutive k of row ra
    const int kb = tid >> 4, qb = tid & 15;           // B: row kb of the chunk, float4 column qb
    const float* pa = A + (size_t)blockIdx.z * sa + (size_t)min(m0 + ra, M - 1) * K + c_begin * BK + 4 * qa;
    const float* pb = B + (size_t)blockIdx.z * sb + (size_t)(c_begin * BK + kb) * N + n0 + 4 * qb;
    const size_t stepb = (size_t)BK * N;
    float4 va, vb;
    auto load = [&]() {
        va = ld4(pa);
        vb = ld4(pb);
        pa += BK;
        pb += stepb;
    };
    auto stash = [&](int buf) {
        st_kcontig<LDA>(As(buf), ra, qa, va);
        *reinterpret_cast<float4*>(Bs(buf) + kb * LDB + 4 * qb) = vb;
    };
    f32x16 acc[1][1];
    acc[0][0] = f32x16{0};
    const int wm0 = (wave >> 1) * 32, wn0 = (wave & 1) * 32;
    if (c_begin < c_end) {
        load();
        stash(0);
    }
    __syncthreads();
    if (c_begin + 1 < c_end) load();
    for (int c = c_begin; c < c_end; ++c) {
        const int cur = (c - c_begin) & 1;
        auto f0 = [&]() { if (c + 1 < c_end) stash(cur ^ 1); };
        auto f1 = [&]() { if (c + 2 < c_end) load(); };
        mma_chunk<1, 1, LDA, LDB>(As(cur), Bs(cur), acc, wm0, wn0, lane, f0, f1);
        __syncthreads();
    }
    float* o = part ? part + ((size_t)blockIdx.y * gridDim.z + blockIdx.z) * ((size_t)M * N) : C + (size_t)blockIdx.z * sc;
    const int col = n0 + wn0 + (lane & 31);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm0 + mfma32_row(r, lane);
        if (row < M) o[(size_t)row * N + col] = acc[0][0][r];
    }
}

__global__ void splitk_reduce_kernel(const float* __restrict__ part, int S, size_t n, float* __restrict__ out,
                                     int accumulate) {
    const size_t n4 = n / 4;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        float4 s = zero4();
        int z = 0;
        for (; z + 3 < S; z += 4) {          // four slabs per trip in flight; added in slab order (the same bits)
            float4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = ld4(part + (size_t)(z + u) * n + 4 * i);
#pragma unroll
            for (int u = 0; u < 4; ++u) add4(s, v[u]);
        }
        for (; z < S; ++z) add4(s, ld4(part + (size_t)z * n + 4 * i));
        if (accumulate) add4(s, ld4(out + 4 * i));
        *reinterpret_cast<float4*>(out + 4 * i) = s;
    }
    if (blockIdx.x == 0)
        for (size_t i = n4 * 4 + threadIdx.x; i < n; i += blockDim.x) {
            float s = 0.0f;
            for (int z = 0; z < S; ++z) s += part[(size_t)z * n + i];
            out[i] = accumulate ? out[i] + s : s;
        }
}

// split-K epilogue of the forward / data-gradient passes: out = act(sum_s part[s] + bias[col])
// addend (optional): out = result + addend -- a skip connection's gradient joining a data gradient (mg_wino_tiles.add), added in
// float32 AFTER the float16 rounding of an autocast result, exactly where the separate add launch put it
__global__ void splitk_epilogue_kernel(const float* __restrict__ part, int S, size_t n, int N,
                                       const float* __restrict__ bias, int act, float* __restrict__ out,
                                       int round_f16 = 0, const float* __restrict__ addend = nullptr) {
    const size_t n4 = n / 4;   // launcher guarantees N % 4 == 0
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        float4 s = zero4();
        int z = 0;
        for (; z + 3 < S; z += 4) {          // four slabs per trip in flight; added in slab order (the same bits)
            float4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = ld4(part + (size_t)(z + u) * n + 4 * i);
#pragma unroll
            for (int u = 0; u < 4; ++u) add4(s, v[u]);
        }
        for (; z < S; ++z) add4(s, ld4(part + (size_t)z * n + 4 * i));
        if (bias) add4(s, ld4(bias + (4 * i) % N));
        s.x = apply_act(s.x, act); s.y = apply_act(s.y, act); s.z = apply_act(s.z, act); s.w = apply_act(s.w, act);
        if (round_f16) { s.x = round_h(s.x); s.y = round_h(s.y); s.z = round_h(s.z); s.w = round_h(s.w); }
        if (addend) add4(s, ld4(addend + 4 * i));
        *reinterpret_cast<float4*>(out + 4 * i) = s;
    }
}

// column sums of a [M, C] matrix.  part[blockIdx.y][c] = sum over this block's row slice; with final != 0 (single
// slice) the result goes straight to out[c] (+= when accumulate).
__global__ void colsum_partial_kernel(const float* __restrict__ a, long long M, int C, long long rows_per_split,
                                      float* __restrict__ part, int final, int accumulate) {
    __shared__ float red[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), rg = threadIdx.x >> 6;
    const long long m_begin = (long long)blockIdx.y * rows_per_split;
    const long long m_end = min(M, m_begin + rows_per_split);
    float s = 0.0f;
    if (c < C) {
        // eight rows per trip: their loads are in flight together (one 4-byte load per thread and trip is pure latency); the
        // additions keep the row order -- the same bits
        long long m = m_begin + rg;
        for (; m + 28 < m_end; m += 32) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = a[(m + 4 * u) * C + c];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; m < m_end; m += 4) s += a[m * C + c];
    }
    red[rg][threadIdx.x & 63] = s;
    __syncthreads();
    if (rg == 0 && c < C) {
        const float v = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
        if (final) part[c] = accumulate ? part[c] + v : v;
        else part[(size_t)blockIdx.y * C + c] = v;
    }
}

}  // namespace

extern "C" __attribute__((visibility("hidden"))) int mg_conv_rowdot_kq(const mg_conv_geom* g);       // conv_rowdot.hip (library-internal)
extern "C" __attribute__((visibility("hidden"))) int mg_instnorm_slab_ok(int HW, int C);              // norm_act.hip (library-internal)
extern "C" __attribute__((visibility("hidden"))) int mg_instnorm_fwd_slabs(const float* part, int S, const float* bias, int round_f16,
                                                                          float* x_out, int B, int HW, int C, float eps, int act,
                                                                          const float* residual, float* y, float* mean, float* rstd,
                                                                          void* stream, void* y16);
#include "wino.h"
#include "wino4.h"
#include "wino42.h"
#include "conv_smallc.h"
namespace {
// One-shot event probe: bench.py arms it right before a call whose main GEMM kernel it wants timed on the launch
// stream.  The LDS-DMA launchers (mg_launch below) hand the two events to hipExtLaunchKernelGGL, which stamps them with
// the dispatch's own begin / end times -- the duration rocprofv3 reports for that kernel; the older launch sites record the
// events immediately around their launch (a few microseconds of queue latency included).
hipEvent_t g_probe_e0 = nullptr, g_probe_e1 = nullptr;
bool g_probe_exact = false;
inline void probe_begin(hipStream_t st) {
    g_probe_exact = false;
    if (g_probe_e0) hipEventRecord(g_probe_e0, st);
}
inline void probe_end(hipStream_t st) {
    if (g_probe_e1 && !g_probe_exact) hipEventRecord(g_probe_e1, st);
    g_probe_e0 = g_probe_e1 = nullptr;
    g_probe_exact = false;
}
template <typename KernelT, typename ArgT>
inline void mg_launch(KernelT kern, dim3 grid, dim3 block, size_t lds, hipStream_t st, const ArgT& a) {
    if (g_probe_e0 && g_probe_e1 && !g_probe_exact) {
        hipExtLaunchKernelGGL(kern, grid, block, (uint32_t)lds, st, g_probe_e0, g_probe_e1, 0, a);
        g_probe_exact = true;
    } else {
        hipLaunchKernelGGL(kern, grid, block, lds, st, a);
    }
}
#include "dense_gemm.h"

// ---------------------------------------------------------------------------------------------------------
// Batched dense GEMMs of the Winograd families on dense_gemm.h (float32 precision; the f16 path keeps the convolution
// kernels' dense instances).  pass 0: C[T][Co] = V[T][Kc] U[Co][Kc]^T; pass 1: dV[T][Kc] = Md[T][Co] U[Co][Kc];
// pass 2: dU[Co][Kc] = Md[T][Co]^T V[T][Kc].
// ---------------------------------------------------------------------------------------------------------
struct DensePlan { int bm, bn, splits, cps; bool dma, ok; };
struct DenseDims { long long M; int N, K; };
inline DenseDims dense_dims(int pass, long long T, int Co, int Kc) {
    if (pass == 0) return {T, Co, Kc};
    if (pass == 1) return {T, Kc, Co};
    return {(long long)Co, Kc, (int)T};
}
DensePlan dense_plan(int pass, int P, long long T, int Co, int Kc, bool hp) {
    constexpr bool off = false, no_dma = false;      // (rounds 1-4 had ablation switches here; the register-staged path below stays for shapes the DMA path does not take)
    const DenseDims dd = dense_dims(pass, T, Co, Kc);
    DensePlan p{64, 64, 1, 1 << 28, false, false};
    if (off || hp || dd.N % 64 != 0 || dd.K % 4 != 0 || dd.K < 4 || dd.M < 1 || T >= (1LL << 31)) return p;
    if (pass == 2 && dd.M % 4 != 0) return p;
    // every operand of one position must stay below 2 GiB (32-bit byte offsets of the LDS-DMA path)
    if ((double)dd.M * dd.K * 4.0 >= 2e9 || (double)dd.N * dd.K * 4.0 >= 2e9) return p;
    p.ok = true;
    // LDS-DMA staging: whole 32-deep chunks, or the weight-gradient GEMM (both operands [K][rows]: the K tail is zero-filled
    // by the buffer range check)
    p.dma = (dd.K % DG_BK == 0 || pass == 2) && !no_dma;
    // Cost model calibrated on scripts/ubench/gemm_bench.hip (MI355X): the workgroups of one CU share its MFMA pipe, so a
    // launch takes ceil(workgroups / 256) tile-times; efficiencies are the measured large-grid figures per tile; a grid with
    // fewer than two workgroups per CU cannot cover its barrier drains (x 0.85); split-K pays one pass over the slabs.
    struct Cand { int bm, bn; double eff_dma, eff_reg; };
    static const Cand cands[4] = {{64, 64, 0.83, 0.71}, {64, 128, 0.885, 0.74}, {128, 64, 0.855, 0.74}, {128, 128, 0.91, 0.80}};
    static const int split_opts[8] = {1, 2, 3, 4, 6, 8, 12, 16};
    const int chunks = (dd.K + DG_BK - 1) / DG_BK;
    int f_bm = 0, f_bn = 0, f_sp = 0;
    double best = 1e300;
    for (const Cand& c : cands) {
        if (dd.N % c.bn != 0) continue;
        if (!p.dma && !((c.bm == 64 && c.bn == 64) || (c.bm == 128 && c.bn == 128))) continue;   // register-path instances
        if (f_bm && (c.bm != f_bm || c.bn != f_bn)) continue;
        const long long w = ((dd.M + c.bm - 1) / c.bm) * (long long)(dd.N / c.bn) * P;
        const double tile_us = 2.0 * c.bm * c.bn * DG_BK / (157.3e12 / 256.0) * 1e6 / (p.dma ? c.eff_dma : c.eff_reg);
        for (int sp : split_opts) {
            if (sp > 1 && chunks / sp < 8) break;
            if (f_sp && sp != f_sp) continue;
            const int cps = (chunks + sp - 1) / sp;
            const int spl = (chunks + cps - 1) / cps;
            const long long wg = w * spl;
            double t = (double)((wg + 255) / 256) * tile_us * (cps + 1.2);
            // forward / data gradient: 500 workgroups already count as two rounds (17x33 maps on 128x128 tiles: 110 against
            // 125 us with the penalty); the weight gradient's short K loops keep the wider band (measured both ways)
            if (wg < (pass == 2 ? 512 : 384)) t /= 0.85;
            if (spl > 1) t += (double)(spl + 1) * P * (double)dd.M * dd.N * 4.0 / 4e12 * 1e6 + 3.0;
            if (t < best) { best = t; p.bm = c.bm; p.bn = c.bn; p.splits = spl; p.cps = cps; }
        }
    }
    if (best == 1e300) { p.ok = false; return p; }
    if (p.splits == 1) p.cps = 1 << 28;
    return p;
}
inline int dense_splits(int pass, int P, long long T, int Co, int Kc, bool hp) {
    const DensePlan p = dense_plan(pass, P, T, Co, Kc, hp);
    return p.ok ? p.splits : 0;
}
// MG_F32_SPLIT=1 (opt-in, read per call so a test can flip it): the Winograd-domain GEMMs multiply float32 operands as three
// exact bf16 pieces each on v_mfma_f32_32x32x16_bf16 (dense_gemm.h: dg_chunk_b8; 8 of the 9 piece products, float32
// accumulation) -- float32-accurate results (error vs float64 1.3e-6 of the output scale against 1.5e-6 for
// v_mfma_f32_32x32x2_f32) at half the MFMA cycles.  Off by default: the bench's float32 line is the float32 pipe.
inline bool f32_split() {
    const char* e = getenv("MG_F32_SPLIT");
    return e && e[0] == '1';
}
inline bool dense_split(const DensePlan& p, int P, int N) {
    return p.dma && (P == 16 || P == 25) && N % 128 == 0 && f32_split();
}
template <int AL, int BL>
void dense_launch(const DensePlan& p, const DgArgs& a, hipStream_t st) {
    if (dense_split(p, a.P, a.N)) {
        if (a.P == 16) dgemm32g_launch<128, 128, 2, 2, AL, BL, 2, 16, 1>(a, st);
        else dgemm32g_launch<128, 128, 2, 2, AL, BL, 2, 25, 1>(a, st);
    } else if (p.dma && a.P == 16) {
        if (p.bm == 128 && p.bn == 128) dgemm32g_launch<128, 128, 2, 2, AL, BL, 2, 16>(a, st);
        else if (p.bm == 64 && p.bn == 128) dgemm32g_launch<64, 128, 2, 2, AL, BL, 2, 16>(a, st);
        else if (p.bm == 128 && p.bn == 64) dgemm32g_launch<128, 64, 2, 2, AL, BL, 2, 16>(a, st);
        else dgemm32g_launch<64, 64, 2, 2, AL, BL, 2, 16>(a, st);
    } else if (p.dma) {
        if (p.bm == 128 && p.bn == 128) dgemm32g_launch<128, 128, 2, 2, AL, BL, 2, 25>(a, st);
        else if (p.bm == 64 && p.bn == 128) dgemm32g_launch<64, 128, 2, 2, AL, BL, 2, 25>(a, st);
        else if (p.bm == 128 && p.bn == 64) dgemm32g_launch<128, 64, 2, 2, AL, BL, 2, 25>(a, st);
        else dgemm32g_launch<64, 64, 2, 2, AL, BL, 2, 25>(a, st);
    } else {
        if (p.bm == 128) dgemm32_launch<128, 128, 4, 2, AL, BL>(a, st);
        else dgemm32_launch<64, 64, 2, 2, AL, BL>(a, st);
    }
}
void dense_name(int pass, int P, const DensePlan& p, int N, char* out, int out_len) {
    const int al = pass == 2 ? DG_RC : DG_KC, bl = pass == 0 ? DG_KC : DG_RC;
    if (dense_split(p, P, N)) snprintf(out, out_len, "dgemm32g_kernel<128, 128, 2, 2, %d, %d, 2, %d, 1>", al, bl, P);
    else if (p.dma) snprintf(out, out_len, "dgemm32g_kernel<%d, %d, 2, 2, %d, %d, 2, %d, 0>", p.bm, p.bn, al, bl, P);
    else snprintf(out, out_len, "dgemm32_kernel<%d, %d, %d, 2, %d, %d, 0>", p.bm, p.bn, p.bm == 128 ? 4 : 2, al, bl);
}

Geom to_geom(const mg_conv_geom* g) {
    return Geom{g->B, g->H, g->W, g->Ci, g->OH, g->OW, g->Co, g->KH, g->KW, g->stride, g->pad, g->reflect};
}

inline bool prec_h(const mg_conv_geom* g) { return g->precision == MG_PRECISION_F16; }

bool geom_ok(const mg_conv_geom* g) {
    if (g && g->precision != MG_PRECISION_F32 && g->precision != MG_PRECISION_F16) return false;
    if (!g || g->B <= 0 || g->H <= 0 || g->W <= 0 || g->Ci <= 0 || g->Co <= 0 || g->KH <= 0 || g->KW <= 0) return false;
    if (g->stride < 1 || g->stride > 2 || g->pad < 0) return false;
    if (g->OH != (g->H + 2 * g->pad - g->KH) / g->stride + 1) return false;
    if (g->OW != (g->W + 2 * g->pad - g->KW) / g->stride + 1) return false;
    if (g->reflect && (g->pad >= g->H || g->pad >= g->W)) return false;
    const long long lim = 1LL << 31;      // kernels index activations / weights with 32-bit element offsets
    if ((long long)g->B * g->H * g->W * g->Ci >= lim || (long long)g->B * g->OH * g->OW * g->Co >= lim ||
        (long long)g->Co * g->KH * g->KW * g->Ci >= lim)
        return false;
    return true;
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }


inline unsigned dense_grid(size_t n) {
    size_t b = (n + 255) / 256;
    return (unsigned)(b > 8192 ? 8192 : (b < 1 ? 1 : b));
}
// Runs the GEMM (and its split-K combine) when the shape is eligible; false: the caller takes the convolution kernels.
bool dense_wino_gemm(int pass, int P, long long T, int Co, int Kc, const float* A, const float* B, float* C, float* part,
                     bool hp, hipStream_t st) {
    const DensePlan p = dense_plan(pass, P, T, Co, Kc, hp);
    if (!p.ok) return false;
    const DenseDims dd = dense_dims(pass, T, Co, Kc);
    DgArgs a{};
    a.A = A; a.B = B; a.C = C; a.part = p.splits > 1 ? part : nullptr;
    a.M = (int)dd.M; a.N = dd.N; a.K = dd.K;
    a.P = P; a.splits = p.splits; a.cps = p.cps;
    a.sc = dd.M * dd.N;
    probe_begin(st);
    if (pass == 0) {            // V [T][Kc] x U [Co][Kc]^T
        a.lda = Kc; a.ldb = Kc; a.sa = T * Kc; a.sb = (long long)Co * Kc;
        dense_launch<DG_KC, DG_KC>(p, a, st);
    } else if (pass == 1) {     // Md [T][Co] x U [Co][Kc]
        a.lda = Co; a.ldb = Kc; a.sa = T * Co; a.sb = (long long)Co * Kc;
        dense_launch<DG_KC, DG_RC>(p, a, st);
    } else {                    // Md [T][Co]^T x V [T][Kc]
        a.lda = Co; a.ldb = Kc; a.sa = T * Co; a.sb = T * Kc;
        dense_launch<DG_RC, DG_RC>(p, a, st);
    }
    probe_end(st);
    if (p.splits > 1) {
        const size_t n = (size_t)P * dd.M * dd.N;
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3(dense_grid(n / 4)), dim3(256), 0, st, (const float*)part, p.splits, n, C,
                           0);
    }
    return true;
}
inline size_t slab_count(int old_splits, int dense) {      // split-K slabs a workspace must hold (either path may run)
    const int m = old_splits > dense ? old_splits : dense;
    return m > 1 ? (size_t)m : 0;
}

// A forward convolution whose caller normalises the result right away (mg_conv_fwd_instnorm_*) may leave its split-K slabs
// unreduced: the InstanceNorm slab kernel sums them itself (norm_act.hip: mg_instnorm_fwd_slabs) -- one launch fewer per layer.
// The caller arms g_fwd_defer; a path that can oblige fills it INSTEAD of launching splitk_epilogue_kernel.
struct FwdDefer { const float* part; int splits; const float* bias; int round_f16; bool filled; };
static thread_local FwdDefer* g_fwd_defer = nullptr;
#include "dense_gemm_h.h"
#include "conv_h16.h"
#include "conv_dma.h"
#include "conv_co1.h"
// float32 -> float16 staging copy of n elements (n % 8 == 0 for every eligible layer: Ci, Co % 64 == 0)
inline void cd_cast16(const float* src, void* dst, size_t n, hipStream_t st) {
    hipLaunchKernelGGL(h16_cast_kernel, dim3(h16_grid(n / 8)), dim3(256), 0, st, src, (_Float16*)dst, n / 8);
}
// ReflectionPad2d(1) + 3x3 stride-1 data gradient on the DMA kernel: the zero-padded "full" data gradient over the padded
// (H+2) x (W+2) domain (geometry gp: pad 0, same OH x OW), folded back onto H x W by wino_fold_reflect_kernel
inline bool cd_reflect_dgrad_geom(const mg_conv_geom* g, mg_conv_geom* gp) {
    if (!g->reflect || g->stride != 1 || g->pad != 1 || g->KH != 3 || g->KW != 3 || g->Ci % 4) return false;
    *gp = *g;
    gp->H += 2; gp->W += 2; gp->pad = 0; gp->reflect = 0;
    return conv_dma_dgrad_ok(gp);
}
// staging + split-K slabs of the DMA data gradient
inline size_t cd_dgrad_ws(const mg_conv_geom* g) {
    const CdPlan cp = conv_dma_dgrad_plan(g);
    const size_t stage = conv_dma_half(g) ? conv_dma_h_dy_bytes(g) + conv_dma_h_w_bytes(g) : 0;
    return stage + (cp.splits > 1 ? (size_t)cp.splits * g->B * g->H * g->W * g->Ci * sizeof(float) : 0) + 256;
}
inline size_t cd_reflect_dxp_bytes(const mg_conv_geom* gp) { return cd_al((size_t)gp->B * gp->H * gp->W * gp->Ci * sizeof(float)); }
// the DMA data gradient proper.  wsp: cd_dgrad_ws(g) bytes (may be null when nothing is needed); round_f16: autocast output
// rounding of the direct result (the reflect wrapper rounds after its fold instead)
int cd_dgrad_run(const mg_conv_geom* g, const float* dy, const float* w, const float* bias, float* dx, int act, char* wsp,
                 hipStream_t st, const float* u, float* md, int round_f16, int dy16_filled = 0);
// a layer any of whose passes runs on the float16 implicit GEMMs keeps a cached float16 copy of its weights
// (mutually exclusive with the Winograd path, which the pass entry points try FIRST: a layer that is wino_ok() keeps
// float32 U / V / Md images under MG_PRECISION_F16, so it must never be handed the float16-sized buffers of this path)
bool wino_ok(const mg_conv_geom* g);
inline bool conv_dma_h_any(const mg_conv_geom* g) {
    mg_conv_geom gp;
    return conv_dma_half(g) && !h16_ok(g) && !mg_conv_rowdot_kq(g) && !wino_ok(g) &&
           (conv_dma_fwd_ok(g) || conv_dma_dgrad_ok(g) || cd_reflect_dgrad_geom(g, &gp));
}
inline bool cd_dgrad_any(const mg_conv_geom* g) {
    mg_conv_geom gp;
    return conv_dma_dgrad_ok(g) || cd_reflect_dgrad_geom(g, &gp);
}
int cd_dgrad_run(const mg_conv_geom* g, const float* dy, const float* w, const float* bias, float* dx, int act, char* wsp,
                 hipStream_t st, const float* u, float* md, int round_f16, int dy16_filled) {
    CdPlan cp = conv_dma_dgrad_plan(g);
    const bool half = conv_dma_half(g);
    const void* dyin = dy;
    const void* win = w;
    if (half) {
        void* dy16 = md ? (void*)md : (void*)wsp;      // shared with the weight gradient
        if (!(md && dy16_filled))                      // MG_TILES_MD_FILLED: the producer of dy wrote the copy (mg_instnorm_bwd_h)
            cd_cast16(dy, dy16, (size_t)g->B * g->OH * g->OW * g->Co, st);
        dyin = dy16;
        wsp += conv_dma_h_dy_bytes(g);
        if (u) {
            win = u;
        } else {
            cd_cast16(w, wsp, (size_t)g->Co * g->KH * g->KW * g->Ci, st);
            win = wsp;
        }
        wsp += conv_dma_h_w_bytes(g);
        MG_CHECK_LAUNCH();
    }
    probe_begin(st);
    conv_dma_dgrad_launch(g, cp, dyin, win, bias, dx, act, (float*)wsp, st, round_f16);
    probe_end(st);
    MG_CHECK_LAUNCH();
    if (cp.splits > 1) {
        const size_t n = (size_t)g->B * g->H * g->W * g->Ci;
        const unsigned blocks = (unsigned)((n / 4 + 255) / 256 > 4096 ? 4096 : (n / 4 + 255) / 256);
        hipLaunchKernelGGL(splitk_epilogue_kernel, dim3(blocks), dim3(256), 0, st, (const float*)wsp, cp.splits, n, g->Ci,
                           bias, act, dx, round_f16);
        MG_CHECK_LAUNCH();
    }
    return MG_OK;
}


// The 32-deep forward kernel needs dynamic LDS above the 64 KB static limit for its 128x128 tile.
inline bool fwd32_enabled() { static const bool off = getenv("MG_NO_BK32") != nullptr; return !off; }
template <int BM_, int BN_, int TAG_>
void launch_fwd32(dim3 grid, hipStream_t st, const Geom& gg, const float* x, const float* w, const float* bias, float* y,
                  int act, int cps32, float* part, const Batch& bt) {
    constexpr size_t lds = (size_t)2 * (BM_ + BN_) * LDK2 * sizeof(float);
    static bool once = false;
    if (!once) {
        hipFuncSetAttribute((const void*)conv_fwd32_kernel<BM_, BN_, TAG_>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds);
        once = true;
    }
    hipLaunchKernelGGL((conv_fwd32_kernel<BM_, BN_, TAG_>), grid, dim3(256), lds, st, gg, x, w, bias, y, act, cps32, part,
                       bt);
}
struct ColsumPlan { int splits; long long rows_per_split; };
ColsumPlan colsum_plan(long long M, int C) {
    if (M <= 256) return {1, M};                  // one launch: each block walks all rows of its 64 columns
    const int cb = (C + 63) / 64;
    long long splits = (1024 + cb - 1) / cb;
    if (splits > 256) splits = 256;
    if (splits > (M + 63) / 64) splits = (M + 63) / 64;
    if (splits < 1) splits = 1;
    long long rps = (M + splits - 1) / splits;
    splits = (M + rps - 1) / rps;
    return {(int)splits, rps};
}

// tile / split selection shared by the launchers and mg_conv_plan_name().  A pass needs >= ~2 workgroups per CU
// (256 CUs) to hide its own global-load latency behind other workgroups' MFMAs; deep-K, small-M*N layers (the
// 1024-channel 8x16 bottleneck: 64 tiles of 128x128) get there by splitting K across blockIdx.y.
struct TilePlan { int bm, bn, splits, cps; bool k32; };
// pass: 0 = forward-kernel GEMM (k-contiguous operands, may use the 32-deep kernel), 1 = data-gradient kernel
TilePlan gemm_plan(long long M, int N, int chunks, int classes, bool can_split, int pass) {
    // Cost model in units of one 32x32x2 MFMA (64 cycles): a workgroup's 4 waves own the CU's 4 SIMDs, so the
    // workgroups mapped to one CU serialise on the MFMA pipe; a lone workgroup per CU cannot hide its own staging;
    // split-K pays one extra pass over (splits + 1) output-sized slabs at ~4 TB/s.
    if (const char* f = getenv("MG_FORCE_PLAN")) {      // tuning harness: "bm,bn,splits"
        int bm = 0, bn = 0, sp = 1;
        if (sscanf(f, "%d,%d,%d", &bm, &bn, &sp) == 3 && sp >= 1) {
            if (!can_split || chunks / sp < 1) sp = 1;
            const int cps = (chunks + sp - 1) / sp;
            return {bm, bn, (chunks + cps - 1) / cps, cps, pass == 0};
        }
    }
    // Plans measured on MI355X for the layer shapes of BASELINE configs[1] at batch 8 (scripts/tune_conv.py sweeps
    // tile x split with MG_FORCE_PLAN and keeps the fastest); anything else falls through to the cost model.
    struct Tuned { int pass; long long M; int N, chunks, classes, bm, bn, sp, k32; };
    static const Tuned tuned[] = {
        {0, 256, 1024, 64, 16, 64, 64, 1, 1},      // Winograd forward GEMMs, 1024-channel 8x16 ResNet blocks
        {1, 256, 1024, 64, 16, 64, 64, 1, 0},      // Winograd data-gradient GEMMs (transposed pipeline: same 256 tiles)
        {1, 360, 1024, 64, 16, 128, 128, 2, 0},    // ... and over the 10x18 padded domain (MG_WINO_DGRAD=padded)
        {0, 1024, 1024, 288, 1, 128, 128, 8, 1},   // 512->1024 stride-2 forward (and the 1024->512 ConvTranspose backward)
        {0, 4096, 512, 144, 1, 128, 128, 4, 1},    // 256->512
        {0, 16384, 256, 72, 1, 64, 64, 1, 1},      // 128->256
        {0, 65536, 128, 36, 1, 128, 128, 1, 1},    // 64->128
        {0, 4896, 512, 256, 1, 128, 128, 3, 1},    // D 256->512 4x4 s1 forward
        {0, 4488, 256, 128, 1, 64, 64, 6, 0},      // D 128->256 4x4 s2 forward
        {0, 17160, 128, 64, 1, 64, 64, 3, 1},      // D 64->128 4x4 s2 forward
        {1, 4488, 256, 512, 1, 64, 64, 8, 0},      // D 256->512 4x4 s1 data gradient
        {1, 1024, 512, 576, 4, 128, 64, 4, 0},     // stride-2 data gradients / ConvTranspose forwards (4 parity classes,
        {1, 4096, 256, 288, 4, 64, 64, 1, 0},      //  each splitting its own K range)
        {1, 16384, 128, 144, 4, 128, 64, 1, 0},
        {1, 65536, 64, 72, 4, 128, 64, 1, 0},
        {1, 4488, 128, 256, 4, 64, 64, 2, 0},
        {1, 17160, 64, 128, 4, 64, 64, 1, 0},
        {1, 1224, 128, 256, 4, 64, 64, 3, 0},      // second discriminator scale
        {1, 4488, 64, 128, 4, 64, 64, 3, 0},
        {1, 8976, 128, 256, 4, 64, 64, 1, 0},      // batch 16 (stacked discriminator-loss pass)
        {1, 34320, 64, 128, 4, 64, 64, 1, 0},
        {1, 2448, 128, 256, 4, 64, 64, 4, 0},
        {1, 8976, 64, 128, 4, 64, 64, 2, 0},
        // batch-64 inference (BASELINE configs[4]): forward only
        {0, 2048, 1024, 64, 16, 128, 128, 1, 1},   // F(2x2,3x3) forward GEMMs, 2048 tiles
        {0, 8192, 1024, 288, 1, 128, 128, 1, 1},   // 512->1024 stride-2 forward
        {0, 32768, 512, 144, 1, 128, 128, 1, 1},   // 256->512
        {0, 2448, 512, 16, 25, 64, 64, 1, 1},      // F(2x2,4x4) forward GEMMs of the 256->512 discriminator layer, batch 16
        // F(4x4,2x2) GEMMs of the stride-2 discriminator layers (K = 4 Ci): 128->256 at 17x33 out (720 tiles at batch 16,
        // 360 at batch 8), 64->128 at 33x65 out (2448 / 1224 tiles)
        {0, 720, 256, 32, 25, 64, 64, 1, 1},
        {0, 2448, 128, 16, 25, 128, 128, 1, 1},
        {1, 720, 512, 16, 25, 64, 64, 1, 0},
        {1, 360, 512, 16, 25, 64, 64, 1, 0},
        {1, 2448, 256, 8, 25, 64, 64, 1, 0},
        {1, 1224, 256, 8, 25, 128, 128, 1, 0},
        // configs[2] (LocalEnhancer) Winograd GEMMs: 2048-channel 4x8 trunk blocks (64 tiles), 128-channel 64x128 local blocks
        {0, 64, 2048, 128, 16, 64, 64, 4, 1},
        {1, 64, 2048, 128, 16, 64, 64, 1, 0},
        {0, 16384, 128, 8, 16, 64, 64, 1, 0},
        {1, 16384, 128, 8, 16, 128, 128, 1, 0},
        // second discriminator scale (64x128 input), batch 8
        {0, 1440, 512, 256, 1, 64, 64, 4, 1},
        {1, 1224, 256, 512, 1, 64, 64, 12, 0},
        {0, 1224, 256, 128, 1, 64, 64, 6, 1},
        {0, 4488, 128, 64, 1, 64, 64, 3, 1},
        // both scales at batch 16 (the stacked fake + real pass of the discriminator loss)
        {0, 9792, 512, 256, 1, 128, 128, 3, 1},
        {1, 8976, 256, 512, 1, 128, 128, 12, 0},
        {0, 8976, 256, 128, 1, 64, 64, 4, 1},
        {0, 34320, 128, 64, 1, 64, 64, 2, 1},
        {0, 2880, 512, 256, 1, 64, 64, 4, 1},
        {1, 2448, 256, 512, 1, 64, 64, 8, 0},
        {0, 2448, 256, 128, 1, 64, 64, 6, 1},
        {0, 8976, 128, 64, 1, 64, 64, 3, 1},
    };
    for (const Tuned& t : tuned)
        if (t.pass == pass && t.M == M && t.N == N && t.chunks == chunks && t.classes == classes &&
            (t.sp == 1 || can_split)) {
            const int cps = (chunks + t.sp - 1) / t.sp;
            return {t.bm, t.bn, (chunks + cps - 1) / cps, cps, t.k32 != 0};
        }
    struct Cand { int bm, bn; };
    const Cand cands[3] = {{128, 128}, {128, 64}, {64, 64}};
    const int split_opts[9] = {1, 2, 3, 4, 6, 8, 12, 16, 24};
    double best = 1e300;
    TilePlan out{64, 64, 1, chunks, false};
    for (const Cand& c : cands) {
        if (c.bm == 128 && c.bn == 128 && (N < 96 || M < 96)) continue;
        if (c.bm == 128 && c.bn == 64 && M < 96) continue;
        const long long tiles = ((M + c.bm - 1) / c.bm) * ((N + c.bn - 1) / c.bn) * classes;
        const double mfma = 8.0 * (c.bm / 64) * (c.bn / 64);
        for (int sp : split_opts) {
            if (sp > 1 && (!can_split || chunks / sp < 8)) break;
            const int cps = (chunks + sp - 1) / sp;
            const int spl = (chunks + cps - 1) / cps;
            const long long wgs = tiles * spl;
            const double work = cps * (mfma + 3.0) + 14.0;
            double t = (double)((wgs + 255) / 256) * work;
            if (wgs < 2 * 256) t *= 1.25;
            if (spl > 1) t += (double)(spl + 1) * (double)M * N * classes * 4.0 / 4e12 / 29e-9 * 256.0 / 256.0;
            // deep reductions measured faster on the 32-deep forward kernel wherever it was tried (>= 144 chunks)
            if (t < best) { best = t; out = {c.bm, c.bn, spl, cps, pass == 0 && chunks >= 128}; }
        }
    }
    return out;
}
inline bool use_k32(const TilePlan& tp, int Ci) {
    return tp.k32 && fwd32_enabled() && Ci % BK2 == 0 && (tp.splits == 1 || tp.cps % 2 == 0);
}
TilePlan fwd_plan(const mg_conv_geom* g) {
    const long long M = (long long)g->B * g->OH * g->OW;
    const bool vec = g->Ci % BK == 0;
    const int chunks = vec ? g->KH * g->KW * (g->Ci / BK) : (g->KH * g->KW * g->Ci + BK - 1) / BK;
    return gemm_plan(M, g->Co, chunks, 1, g->Co % 4 == 0, 0);
}
TilePlan dgrad_plan(const mg_conv_geom* g) {
    const int s = g->stride;
    const long long Mc = (long long)g->B * ((g->H + s - 1) / s) * ((g->W + s - 1) / s);
    const bool vec = g->Co % BK == 0;
    const int chunks = vec ? g->KH * g->KW * (g->Co / BK) : (g->KH * g->KW * g->Co + BK - 1) / BK;
    return gemm_plan(Mc, g->Ci, chunks, s * s, g->Ci % 4 == 0, 1);
}

struct WgradPlan { bool big; int tiles; int splits; int cps; };
WgradPlan wgrad_plan(const mg_conv_geom* g) {
    const int R = g->Co, N = g->KH * g->KW * g->Ci;
    const long long Mtot = (long long)g->B * g->OH * g->OW;
    const int chunks = (int)((Mtot + BK - 1) / BK);
    const int t128 = ((R + 127) / 128) * ((N + 127) / 128);
    bool big = t128 >= 96 && R >= 128;
    int want = -1;
    if (const char* f = getenv("MG_FORCE_WGRAD")) {      // tuning harness: "big(0|1),splits"
        int b = 0, sp = 1;
        if (sscanf(f, "%d,%d", &b, &sp) == 2 && sp >= 1) { big = b != 0 && R >= 64; want = sp; }
    } else {
        // measured on MI355X (scripts/tune_conv.py --wgrad) for the configs[1] layers; the heuristic below covers the rest
        struct Tuned { int R, N; long long M; int big, sp; };
        static const Tuned tuned[] = {
#include "wgrad_plans.inc"
        };
        for (const Tuned& t : tuned)
            if (t.R == R && t.N == N && t.M == Mtot) { big = t.big != 0; want = t.sp; break; }
    }
    const int tiles = big ? t128 : ((R + 63) / 64) * ((N + 63) / 64);
    int splits = want > 0 ? want : (tiles >= 512 ? 1 : (768 + tiles - 1) / tiles);
    const int max_splits = chunks / 8 > 0 ? chunks / 8 : 1;
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
    int cps = (chunks + splits - 1) / splits;
    splits = (chunks + cps - 1) / cps;
    return {big, tiles, splits, cps};
}

// ---------------------------------------------------------------------------------------------------------
// Winograd F(2x2,3x3) orchestration (transforms in wino.h, the 16 GEMMs as one batched implicit-GEMM launch)
// ---------------------------------------------------------------------------------------------------------
bool h16_ok(const mg_conv_geom* g);
bool wino_ok(const mg_conv_geom* g) {
    static const bool off = getenv("MG_NO_WINOGRAD") != nullptr;
    if (h16_ok(g)) return false;             // weight-dominated autocast layers take the float16 GEMM path (conv_h16.h)
    constexpr bool off_h = false;
    // f16 GEMMs are fast enough that Winograd only pays where the 16 transformed-weight matrices are amortised over
    // many tiles: wide layers (>= 256 channels) with >= 256 tiles (the 1024-channel 8x16 blocks of configs[1]; not the
    // 2048-channel 4x8 trunk of configs[2], where reading 16 * Co * Ci transformed weights would dominate)
    constexpr int min_tiles_h = 256;       // (64 measured slower on configs[2])
    if (prec_h(g) && (off_h || g->Ci < 256 || g->Co < 256 || (long long)g->B * (g->H / 2) * (g->W / 2) < min_tiles_h)) return false;
    return !off && g->KH == 3 && g->KW == 3 && g->stride == 1 && g->pad == 1 && g->Ci % 16 == 0 && g->Co % 16 == 0 &&
           g->Ci >= 32 && g->Co >= 32 && g->H % 2 == 0 && g->W % 2 == 0 && g->H >= 2 && g->W >= 2;
}
struct WinoDims { long long T, Tp; int TH, TW, THp, TWp; };
WinoDims wino_dims(const mg_conv_geom* g) {
    WinoDims d;
    d.TH = g->H / 2; d.TW = g->W / 2; d.T = (long long)g->B * d.TH * d.TW;
    d.THp = (g->H + 2) / 2; d.TWp = (g->W + 2) / 2; d.Tp = (long long)g->B * d.THp * d.TWp;
    return d;
}
inline size_t al256(size_t n) { return (n + 63) / 64 * 64; }   // in floats

size_t wino_fwd_ws(const mg_conv_geom* g) {
    const WinoDims d = wino_dims(g);
    const TilePlan tp = gemm_plan(d.T, g->Co, g->Ci / BK, 16, true, 0);
    return (al256((size_t)16 * g->Co * g->Ci) + al256((size_t)16 * d.T * g->Ci) + al256((size_t)16 * d.T * g->Co) +
            al256(slab_count(tp.splits, dense_splits(0, 16, d.T, g->Co, g->Ci, prec_h(g))) * 16 * d.T * g->Co)) * sizeof(float) + 256;
}
size_t wino_dgrad_ws(const mg_conv_geom* g) {      // U | A dy A^T | dV | dd | split-K slabs
    const WinoDims d = wino_dims(g);
    const TilePlan tp = gemm_plan(d.T, g->Ci, g->Co / BK, 16, true, 1);
    return (al256((size_t)16 * g->Co * g->Ci) + al256((size_t)16 * d.T * g->Co) + 2 * al256((size_t)16 * d.T * g->Ci) +
            al256(slab_count(tp.splits, dense_splits(1, 16, d.T, g->Co, g->Ci, prec_h(g))) * 16 * d.T * g->Ci)) * sizeof(float) + 256;
}
struct WinoWgradPlan { bool big; int tiles, splits, cps; };
WinoWgradPlan wino_wgrad_plan(const mg_conv_geom* g) {
    const WinoDims d = wino_dims(g);
    const int chunks = (int)((d.T + BK - 1) / BK);
    const int t128 = ((g->Co + 127) / 128) * ((g->Ci + 127) / 128);
    // short reductions (<= 1024 tiles per position, the 8x16 blocks of configs[1]): 64x64 tiles measured 125 vs 148 us
    bool big = g->Co >= 128 && g->Ci >= 128 && d.T > 1024;
    const int want = -1;
    const int tiles = big ? t128 : ((g->Co + 63) / 64) * ((g->Ci + 63) / 64);
    int splits = want > 0 ? want : (tiles * 16 >= 512 ? 1 : (768 + tiles * 16 - 1) / (tiles * 16));
    const int max_splits = chunks / 8 > 0 ? chunks / 8 : 1;
    if (splits > max_splits) splits = max_splits;
    int cps = (chunks + splits - 1) / splits;
    splits = (chunks + cps - 1) / cps;
    return {big, tiles, splits, cps};
}
inline size_t wino_wgrad_slabs(const mg_conv_geom* g) {
    return slab_count(wino_wgrad_plan(g).splits, dense_splits(2, 16, wino_dims(g).T, g->Co, g->Ci, prec_h(g)));
}
size_t wino_wgrad_ws(const mg_conv_geom* g) {
    const WinoDims d = wino_dims(g);
    const WinoWgradPlan p = wino_wgrad_plan(g);
    const size_t cs = (mg_colsum_workspace((long long)g->B * g->OH * g->OW, g->Co) + 255) / 4;
    return (al256((size_t)16 * d.T * g->Ci) + al256((size_t)16 * d.T * g->Co) + al256((size_t)16 * g->Co * g->Ci) +
            al256(wino_wgrad_slabs(g) * 16 * g->Co * g->Ci) + al256(cs)) * sizeof(float) + 256;
}

template <typename Launch>
int wino_launch_tiles(const TilePlan& tp, Launch&& launch) {
    if (tp.bm == 128 && tp.bn == 128) launch(std::integral_constant<int, 128>{}, std::integral_constant<int, 128>{});
    else if (tp.bm == 64) launch(std::integral_constant<int, 64>{}, std::integral_constant<int, 64>{});
    else launch(std::integral_constant<int, 128>{}, std::integral_constant<int, 64>{});
    return 0;
}

// The Winograd-domain weight-gradient GEMM on dense_tn64_kernel (64x64 tiles, float32): Kc = Ci, or 4 Ci for wino42
inline bool lean_wgrad_ok(const WinoWgradPlan& p, long long T, int Co, int Kc, bool hp) {
    constexpr bool off = false;
    return !off && !hp && !p.big && Co % 64 == 0 && Kc % 64 == 0 && T % BK == 0;
}
// 64 x 128 tiles (two MFMAs per A fragment) where they still give >= 4 workgroups per CU: 83 -> 79.5 us on the
// 1024-channel layer
inline bool lean_wgrad_wide(const WinoWgradPlan& p, int P, int Co, int Kc) {
    constexpr bool on = true;
    return on && P == 16 && Kc % 128 == 0 && (long long)(Co / 64) * (Kc / 128) * P * p.splits >= 1024;
}
inline void launch_lean_wgrad(const WinoWgradPlan& p, int P, long long T, int Co, int Kc, const float* Md, const float* V,
                              float* target, hipStream_t st) {
    dim3 grid((unsigned)((Co / 64) * (Kc / 64)), P, p.splits);
    const int cps = p.splits == 1 ? (1 << 29) : p.cps;
    if (lean_wgrad_wide(p, P, Co, Kc)) {
        dim3 g2((unsigned)((Co / 64) * (Kc / 128)), P, p.splits);
        hipLaunchKernelGGL((dense_tn64_kernel<1, 1, 2>), g2, dim3(256), 0, st, Md, V, target, Co, Kc, (int)T, T * Co, T * Kc,
                           (long long)Co * Kc, cps);
        return;
    }
    if (P == 16)
        hipLaunchKernelGGL((dense_tn64_kernel<1, 1>), grid, dim3(256), 0, st, Md, V, target, Co, Kc, (int)T, T * Co, T * Kc,
                           (long long)Co * Kc, cps);
    else
        hipLaunchKernelGGL((dense_tn64_kernel<1, 5>), grid, dim3(256), 0, st, Md, V, target, Co, Kc, (int)T, T * Co, T * Kc,
                           (long long)Co * Kc, cps);
}

// ... and the data-gradient GEMM on dense_nn64_kernel (64x64 plans only)
inline bool lean_dgrad_ok(const TilePlan& tp, long long T, int Nc, int Kc, bool hp) {
    constexpr bool off = false;
    return !off && !hp && tp.bm == 64 && tp.bn == 64 && Nc % 64 == 0 && Kc % BK == 0 && T >= 1;
}
inline void launch_lean_dgrad(const TilePlan& tp, int P, long long T, int Nc, int Kc, const float* Md, const float* U,
                              float* dV, float* part, hipStream_t st) {
    dim3 grid((unsigned)(((T + 63) / 64) * (Nc / 64)), tp.splits, P);
    const int cps = tp.splits == 1 ? (1 << 29) : tp.cps;
    if (P == 16)
        hipLaunchKernelGGL(dense_nn64_kernel<1>, grid, dim3(256), 0, st, Md, U, dV, part, (int)T, Nc, Kc, T * Kc,
                           (long long)Kc * Nc, T * Nc, cps);
    else
        hipLaunchKernelGGL(dense_nn64_kernel<5>, grid, dim3(256), 0, st, Md, U, dV, part, (int)T, Nc, Kc, T * Kc,
                           (long long)Kc * Nc, T * Nc, cps);
}

int wino_fwd(const mg_conv_geom* g, const float* x, const float* w, const float* bias, float* y, int act, float* ws,
             hipStream_t st, const float* u_pre, float* v_keep, const WinoNorm* nrm = nullptr, bool v_filled = false) {
    const WinoDims d = wino_dims(g);
    float* U = ws;
    float* V = U + al256((size_t)16 * g->Co * g->Ci);
    float* Mx = V + al256((size_t)16 * d.T * g->Ci);
    float* part = Mx + al256((size_t)16 * d.T * g->Co);
    if (u_pre) U = const_cast<float*>(u_pre);
    else hipLaunchKernelGGL(wino_weight_xform_kernel, dim3(wino_grid((size_t)g->Co * g->Ci / 4)), dim3(256), 0, st, w,
                            g->Co, g->Ci, U);
    if (v_keep) V = v_keep;          // the caller keeps B^T x B for the weight gradient
    // (v_filled: x's producer wrote B^T x B already -- mg_conv_fwd_instnorm_next of the layer in front)
    if (!(v_filled && v_keep))
        hipLaunchKernelGGL(wino_input_xform_kernel, dim3(wino_grid((size_t)d.T * g->Ci / 4)), dim3(256), 0, st, x, g->B, g->H,
                           g->W, g->Ci, d.TH, d.TW, 1, g->reflect, V);
    if (!dense_wino_gemm(0, 16, d.T, g->Co, g->Ci, V, U, Mx, part, prec_h(g), st)) {
    const Geom gg{1, 1, (int)d.T, g->Ci, 1, (int)d.T, g->Co, 1, 1, 1, 0, 0};
    const TilePlan tp = gemm_plan(d.T, g->Co, g->Ci / BK, 16, true, 0);
    float* pp = tp.splits > 1 ? part : nullptr;
    const Batch bt{d.T * g->Ci, (long long)g->Co * g->Ci, d.T * g->Co, 0};
    probe_begin(st);
    const bool hp = prec_h(g);
    const bool k32 = hp ? (fwd32_enabled() && g->Ci % BK2 == 0 && (tp.splits == 1 || tp.cps % 2 == 0)) : use_k32(tp, g->Ci);
    wino_launch_tiles(tp, [&](auto bm, auto bn) {
        constexpr int BM_ = decltype(bm)::value, BN_ = decltype(bn)::value;
        dim3 grid((unsigned)(((d.T + BM_ - 1) / BM_) * ((g->Co + BN_ - 1) / BN_)), tp.splits, 16);
        if (hp && k32)
            launch_fwd32<BM_, BN_, 3>(grid, st, gg, V, U, nullptr, Mx, MG_ACT_NONE, tp.splits == 1 ? (1 << 29) : tp.cps / 2,
                                      pp, bt);
        else if (hp)
            hipLaunchKernelGGL((conv_fwd_kernel<BM_, BN_, true, 3>), grid, dim3(256), 0, st, gg, (const float*)V,
                               (const float*)U, (const float*)nullptr, Mx, MG_ACT_NONE, tp.cps, pp, bt);
        else if (k32)
            launch_fwd32<BM_, BN_, 1>(grid, st, gg, V, U, nullptr, Mx, MG_ACT_NONE, tp.splits == 1 ? (1 << 29) : tp.cps / 2,
                                      pp, bt);
        else
            hipLaunchKernelGGL((conv_fwd_kernel<BM_, BN_, true, 1>), grid, dim3(256), 0, st, gg, (const float*)V,
                               (const float*)U, (const float*)nullptr, Mx, MG_ACT_NONE, tp.cps, pp, bt);
    });
    probe_end(st);
    if (pp) {
        const size_t n = (size_t)16 * d.T * g->Co;
        hipLaunchKernelGGL(splitk_epilogue_kernel, dim3(wino_grid(n / 4)), dim3(256), 0, st, (const float*)pp, tp.splits,
                           n, g->Co, (const float*)nullptr, MG_ACT_NONE, Mx);
    }
    }
    if (nrm) {           // output transform + InstanceNorm in one kernel (y = the raw convolution output, nrm->y the normalised one)
        const dim3 grid(g->Co / 32, g->B);
        const int nt = (d.TH * d.TW + 31) / 32;
#define MG_OUT_NORM(NT_) hipLaunchKernelGGL(wino_out_norm_kernel<NT_>, grid, dim3(256), 0, st, (const float*)Mx, g->B, d.TH, d.TW, g->Co, bias, *nrm, y)
#define MG_OUT_NORM_NEXT(NT_) hipLaunchKernelGGL((wino_out_norm_kernel<NT_, true>), grid, dim3(256), 0, st, (const float*)Mx, g->B, d.TH, d.TW, g->Co, bias, *nrm, y)
        if (nrm->v_next && nt == 1) MG_OUT_NORM_NEXT(1);
        else if (nrm->v_next && nt == 2) MG_OUT_NORM_NEXT(2);
        else if (nrm->v_next) return MG_ERR_ARG;       // (callers ask mg_conv_wino_vnext_ok first)
        else if (nt == 1) MG_OUT_NORM(1); else if (nt == 2) MG_OUT_NORM(2); else if (nt == 3) MG_OUT_NORM(3); else if (nt == 4) MG_OUT_NORM(4); else MG_OUT_NORM(5);
#undef MG_OUT_NORM_NEXT
#undef MG_OUT_NORM
        MG_CHECK_LAUNCH();
        return MG_OK;
    }
    hipLaunchKernelGGL(wino_output_xform_kernel, dim3(wino_grid((size_t)d.T * g->Co / 4)), dim3(256), 0, st,
                       (const float*)Mx, g->B, d.TH, d.TW, g->Co, bias, act, y, (int)prec_h(g));
    MG_CHECK_LAUNCH();
    return MG_OK;
}

// mg_wino_tiles.add of the mg_conv_dgrad_w call in flight on this thread: a path that can fold "dx += add" into its last
// kernel takes it and clears it; whatever is left is added by a separate pass at the end of the call.
static thread_local const float* g_dgrad_add = nullptr;

// Data gradient as the transpose of the forward pipeline (wino.h): T tiles, no padded domain.
int wino_dgrad_t(const mg_conv_geom* g, const float* dy, const float* w, const float* bias, float* dx, int act, float* ws,
                 hipStream_t st, const float* u_pre, float* md_keep) {
    const WinoDims d = wino_dims(g);
    const long long T = d.T;
    float* U = ws;
    float* Md = U + al256((size_t)16 * g->Co * g->Ci);
    float* dV = Md + al256((size_t)16 * T * g->Co);
    float* dd = dV + al256((size_t)16 * T * g->Ci);
    float* part = dd + al256((size_t)16 * T * g->Ci);
    if (u_pre) U = const_cast<float*>(u_pre);
    else hipLaunchKernelGGL(wino_weight_xform_kernel, dim3(wino_grid((size_t)g->Co * g->Ci / 4)), dim3(256), 0, st, w,
                            g->Co, g->Ci, U);
    if (md_keep) Md = md_keep;       // the caller keeps A dy A^T for the weight gradient
    if (dy)                          // dy == nullptr: md_keep already holds it (mg_instnorm_bwd_wino_md)
        hipLaunchKernelGGL(wino_dy_xform_kernel, dim3(wino_grid((size_t)T * g->Co / 4)), dim3(256), 0, st, dy, g->B, d.TH,
                           d.TW, g->Co, Md);
    if (!dense_wino_gemm(1, 16, T, g->Co, g->Ci, Md, U, dV, part, prec_h(g), st)) {
    const Geom gg{1, 1, (int)T, g->Ci, 1, (int)T, g->Co, 1, 1, 1, 0, 0};
    const TilePlan tp = gemm_plan(T, g->Ci, g->Co / BK, 16, true, 1);
    float* pp = tp.splits > 1 ? part : nullptr;
    const Batch bt{T * g->Co, (long long)g->Co * g->Ci, T * g->Ci, 0};      // dV_z = dM_z U_z: same position, no flip
    probe_begin(st);
    if (lean_dgrad_ok(tp, T, g->Ci, g->Co, prec_h(g)))
        launch_lean_dgrad(tp, 16, T, g->Ci, g->Co, Md, U, dV, pp, st);
    else
    wino_launch_tiles(tp, [&](auto bm, auto bn) {
        constexpr int BM_ = decltype(bm)::value, BN_ = decltype(bn)::value;
        dim3 grid((unsigned)(((T + BM_ - 1) / BM_) * ((g->Ci + BN_ - 1) / BN_)), tp.splits, 16);
        if (prec_h(g))
            hipLaunchKernelGGL((conv_dgrad_kernel<BM_, BN_, true, true, 3>), grid, dim3(256), 0, st, gg, (const float*)Md,
                               (const float*)U, (const float*)nullptr, dV, MG_ACT_NONE, tp.cps, pp, bt);
        else
            hipLaunchKernelGGL((conv_dgrad_kernel<BM_, BN_, true, true, 1>), grid, dim3(256), 0, st, gg, (const float*)Md,
                               (const float*)U, (const float*)nullptr, dV, MG_ACT_NONE, tp.cps, pp, bt);
    });
    probe_end(st);
    if (pp) {
        const size_t n = (size_t)16 * T * g->Ci;
        hipLaunchKernelGGL(splitk_epilogue_kernel, dim3(wino_grid(n / 4)), dim3(256), 0, st, (const float*)pp, tp.splits,
                           n, g->Ci, (const float*)nullptr, MG_ACT_NONE, dV);
    }
    }
    if (!prec_h(g) && wino_dd_gather_ok(g->H, g->W, g->Ci)) {      // small maps: patches through LDS, one kernel
        const int ts = (g->H / 2) * (g->W / 2);
        const size_t lds = (size_t)ts * 16 * 8 * sizeof(float4);
        const dim3 grid(g->Ci / 32, g->B);
        auto go = [&](auto kern) {
            hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, (const float*)dV, g->B, g->H, g->W, g->Ci, g->reflect, bias, act, dx,
                               g_dgrad_add);
            g_dgrad_add = nullptr;                 // consumed: mg_conv_dgrad_w has nothing left to add
        };
        if (ts <= 32) go(wino_dd_gather_kernel<1>); else go(wino_dd_gather_kernel<2>);
        MG_CHECK_LAUNCH();
        return MG_OK;
    }
    hipLaunchKernelGGL(wino_dd_xform_kernel, dim3(wino_grid((size_t)T * g->Ci / 4)), dim3(256), 0, st, (const float*)dV, T,
                       g->Ci, dd);
    hipLaunchKernelGGL(wino_dx_gather_kernel, dim3(wino_grid((size_t)g->B * g->H * g->W * g->Ci / 4)), dim3(256), 0, st,
                       (const float*)dd, g->B, g->H, g->W, g->Ci, g->reflect, bias, act, dx, (int)prec_h(g));
    MG_CHECK_LAUNCH();
    return MG_OK;
}

int wino_dgrad(const mg_conv_geom* g, const float* dy, const float* w, const float* bias, float* dx, int act, float* ws,
               hipStream_t st, const float* u_pre, float* md_keep) {
    return wino_dgrad_t(g, dy, w, bias, dx, act, ws, st, u_pre, md_keep);
}

int wino_wgrad(const mg_conv_geom* g, const float* x, const float* dy, float* dw, int accumulate, float* ws,
               hipStream_t st, const float* v_in, const float* md_in, const mg_wino_adam* ad = nullptr) {
    const WinoDims d = wino_dims(g);
    const WinoWgradPlan p = wino_wgrad_plan(g);
    float* V = ws;
    float* Md = V + al256((size_t)16 * d.T * g->Ci);
    float* dU = Md + al256((size_t)16 * d.T * g->Co);
    float* part = dU + al256((size_t)16 * g->Co * g->Ci);
    if (v_in) V = const_cast<float*>(v_in);
    else hipLaunchKernelGGL(wino_input_xform_kernel, dim3(wino_grid((size_t)d.T * g->Ci / 4)), dim3(256), 0, st, x, g->B,
                            g->H, g->W, g->Ci, d.TH, d.TW, 1, g->reflect, V);
    if (md_in) Md = const_cast<float*>(md_in);
    else hipLaunchKernelGGL(wino_dy_xform_kernel, dim3(wino_grid((size_t)d.T * g->Co / 4)), dim3(256), 0, st, dy, g->B,
                            d.TH, d.TW, g->Co, Md);
    if (!dense_wino_gemm(2, 16, d.T, g->Co, g->Ci, Md, V, dU, part, prec_h(g), st)) {
    const Geom gg{1, 1, (int)d.T, g->Ci, 1, (int)d.T, g->Co, 1, 1, 1, 0, 0};
    float* target = p.splits > 1 ? part : dU;
    const Batch bt{d.T * g->Ci, d.T * g->Co, (long long)g->Co * g->Ci, 0};
    dim3 grid((unsigned)p.tiles, 16, p.splits);
    probe_begin(st);
    if (lean_wgrad_ok(p, d.T, g->Co, g->Ci, prec_h(g)))
        launch_lean_wgrad(p, 16, d.T, g->Co, g->Ci, Md, V, target, st);
    else
    if (p.big && prec_h(g))
        hipLaunchKernelGGL((conv_wgrad_kernel<128, 128, true, true, 3>), grid, dim3(256), 0, st, gg, (const float*)V,
                           (const float*)Md, target, p.cps, 0, bt);
    else if (p.big)
        hipLaunchKernelGGL((conv_wgrad_kernel<128, 128, true, true, 1>), grid, dim3(256), 0, st, gg, (const float*)V,
                           (const float*)Md, target, p.cps, 0, bt);
    else if (prec_h(g))
        hipLaunchKernelGGL((conv_wgrad_kernel<64, 64, true, true, 3>), grid, dim3(256), 0, st, gg, (const float*)V,
                           (const float*)Md, target, p.cps, 0, bt);
    else
        hipLaunchKernelGGL((conv_wgrad_kernel<64, 64, true, true, 1>), grid, dim3(256), 0, st, gg, (const float*)V,
                           (const float*)Md, target, p.cps, 0, bt);
    probe_end(st);
    if (p.splits > 1) {
        const size_t n = (size_t)16 * g->Co * g->Ci;
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3(wino_grid(n / 4)), dim3(256), 0, st, (const float*)part, p.splits, n,
                           dU, 0);
    }
    }
    if (ad)     // dw IS the weight tensor here: inverse transform + Adam + next iteration's forward transform in one pass
        hipLaunchKernelGGL(wino_adam_kernel, dim3(wino_grid((size_t)g->Co * g->Ci / 4)), dim3(256), 0, st, (const float*)dU, g->Co,
                           g->Ci, dw, ad->m, ad->v, ad->u, ad->state, ad->beta1, ad->beta2, ad->eps, ad->grad_scale);
    else
    hipLaunchKernelGGL(wino_dweight_xform_kernel, dim3(wino_grid((size_t)g->Co * g->Ci / 4)), dim3(256), 0, st,
                       (const float*)dU, g->Co, g->Ci, dw, accumulate);
    MG_CHECK_LAUNCH();
    return MG_OK;
}

// ---------------------------------------------------------------------------------------------------------
// Winograd F(2x2,4x4) orchestration (transforms in wino4.h): the stride-1 4x4 PatchGAN layers.  25 GEMMs as one
// batched launch of the same kernels the F(2x2,3x3) path uses.
// ---------------------------------------------------------------------------------------------------------
bool wino4_ok(const mg_conv_geom* g) {
    static const bool off = getenv("MG_NO_WINOGRAD4") != nullptr;
    constexpr int min_c = 32;
    return !off && !prec_h(g) && g->KH == 4 && g->KW == 4 && g->stride == 1 && g->pad == 2 && !g->reflect &&
           g->Ci % 16 == 0 && g->Co % 16 == 0 && g->Ci >= min_c && g->Co >= min_c && (g->H & 1) && (g->W & 1);
}
struct Wino4Dims { long long T; int TH, TW; };
Wino4Dims wino4_dims(const mg_conv_geom* g) {
    Wino4Dims d;
    d.TH = (g->H + 1) / 2; d.TW = (g->W + 1) / 2; d.T = (long long)g->B * d.TH * d.TW;
    return d;
}
WinoWgradPlan wino4_wgrad_plan(const mg_conv_geom* g) {
    const Wino4Dims d = wino4_dims(g);
    const int chunks = (int)((d.T + BK - 1) / BK);
    bool big = g->Co >= 128 && g->Ci >= 128 && d.T > 4096;      // measured at 2448 tiles: 64x64 232 us, 128x128 252 us
    int want = -1;
    const int tiles = big ? ((g->Co + 127) / 128) * ((g->Ci + 127) / 128) : ((g->Co + 63) / 64) * ((g->Ci + 63) / 64);
    int splits = want > 0 ? want : (tiles * 25 >= 512 ? (chunks >= 128 ? 3 : 1) : (768 + tiles * 25 - 1) / (tiles * 25));
    const int max_splits = chunks / 8 > 0 ? chunks / 8 : 1;
    if (splits > max_splits) splits = max_splits;
    int cps = (chunks + splits - 1) / splits;
    splits = (chunks + cps - 1) / cps;
    return {big, tiles, splits, cps};
}
inline size_t wino4_wgrad_slabs(const mg_conv_geom* g) {
    return slab_count(wino4_wgrad_plan(g).splits, dense_splits(2, 25, wino4_dims(g).T, g->Co, g->Ci, false));
}
size_t wino4_fwd_ws(const mg_conv_geom* g) {
    const Wino4Dims d = wino4_dims(g);
    const TilePlan tp = gemm_plan(d.T, g->Co, g->Ci / BK, 25, true, 0);
    return (al256((size_t)25 * g->Co * g->Ci) + al256((size_t)25 * d.T * g->Ci) + al256((size_t)25 * d.T * g->Co) +
            al256(slab_count(tp.splits, dense_splits(0, 25, d.T, g->Co, g->Ci, false)) * 25 * d.T * g->Co)) * sizeof(float) + 256;
}
size_t wino4_dgrad_ws(const mg_conv_geom* g) {      // U | A dy A^T | dV | dd | split-K slabs
    const Wino4Dims d = wino4_dims(g);
    const TilePlan tp = gemm_plan(d.T, g->Ci, g->Co / BK, 25, true, 1);
    return (al256((size_t)25 * g->Co * g->Ci) + al256((size_t)25 * d.T * g->Co) + 2 * al256((size_t)25 * d.T * g->Ci) +
            al256(slab_count(tp.splits, dense_splits(1, 25, d.T, g->Co, g->Ci, false)) * 25 * d.T * g->Ci)) * sizeof(float) + 256;
}
size_t wino4_wgrad_ws(const mg_conv_geom* g) {
    const Wino4Dims d = wino4_dims(g);
    const WinoWgradPlan p = wino4_wgrad_plan(g);
    const size_t cs = (mg_colsum_workspace((long long)g->B * g->OH * g->OW, g->Co) + 255) / 4;
    return (al256((size_t)25 * d.T * g->Ci) + al256((size_t)25 * d.T * g->Co) + al256((size_t)25 * g->Co * g->Ci) +
            al256(wino4_wgrad_slabs(g) * 25 * g->Co * g->Ci) + al256(cs)) * sizeof(float) + 256;
}

int wino4_fwd(const mg_conv_geom* g, const float* x, const float* w, const float* bias, float* y, int act, float* ws,
              hipStream_t st, const float* u_pre, float* v_keep) {
    const Wino4Dims d = wino4_dims(g);
    float* U = ws;
    float* V = U + al256((size_t)25 * g->Co * g->Ci);
    float* Mx = V + al256((size_t)25 * d.T * g->Ci);
    float* part = Mx + al256((size_t)25 * d.T * g->Co);
    if (u_pre) U = const_cast<float*>(u_pre);
    else hipLaunchKernelGGL(wino4_weight_xform_kernel, dim3(wino_grid((size_t)g->Co * g->Ci / 2)), dim3(256), 0, st, w,
                            g->Co, g->Ci, U);
    if (v_keep) V = v_keep;
    hipLaunchKernelGGL(wino4_input_xform_kernel, dim3(wino_grid((size_t)d.T * g->Ci / 2)), dim3(256), 0, st, x, g->B, g->H,
                       g->W, g->Ci, d.TH, d.TW, V);
    if (!dense_wino_gemm(0, 25, d.T, g->Co, g->Ci, V, U, Mx, part, false, st)) {
    const Geom gg{1, 1, (int)d.T, g->Ci, 1, (int)d.T, g->Co, 1, 1, 1, 0, 0};
    const TilePlan tp = gemm_plan(d.T, g->Co, g->Ci / BK, 25, true, 0);
    float* pp = tp.splits > 1 ? part : nullptr;
    const Batch bt{d.T * g->Ci, (long long)g->Co * g->Ci, d.T * g->Co, 0};
    probe_begin(st);
    const bool k32 = use_k32(tp, g->Ci);
    wino_launch_tiles(tp, [&](auto bm, auto bn) {
        constexpr int BM_ = decltype(bm)::value, BN_ = decltype(bn)::value;
        dim3 grid((unsigned)(((d.T + BM_ - 1) / BM_) * ((g->Co + BN_ - 1) / BN_)), tp.splits, 25);
        if (k32)
            launch_fwd32<BM_, BN_, 5>(grid, st, gg, V, U, nullptr, Mx, MG_ACT_NONE, tp.splits == 1 ? (1 << 29) : tp.cps / 2,
                                      pp, bt);
        else
            hipLaunchKernelGGL((conv_fwd_kernel<BM_, BN_, true, 5>), grid, dim3(256), 0, st, gg, (const float*)V,
                               (const float*)U, (const float*)nullptr, Mx, MG_ACT_NONE, tp.cps, pp, bt);
    });
    probe_end(st);
    if (pp) {
        const size_t n = (size_t)25 * d.T * g->Co;
        hipLaunchKernelGGL(splitk_epilogue_kernel, dim3(wino_grid(n / 4)), dim3(256), 0, st, (const float*)pp, tp.splits,
                           n, g->Co, (const float*)nullptr, MG_ACT_NONE, Mx);
    }
    }
    hipLaunchKernelGGL(wino4_output_xform_kernel, dim3(wino_grid((size_t)d.T * g->Co / 2)), dim3(256), 0, st,
                       (const float*)Mx, g->B, d.TH, d.TW, g->Co, bias, act, y);
    MG_CHECK_LAUNCH();
    return MG_OK;
}

int wino4_dgrad(const mg_conv_geom* g, const float* dy, const float* w, float* dx, float* ws, hipStream_t st,
                const float* u_pre, float* md_keep) {
    const Wino4Dims d = wino4_dims(g);
    const long long T = d.T;
    float* U = ws;
    float* Md = U + al256((size_t)25 * g->Co * g->Ci);
    float* dV = Md + al256((size_t)25 * T * g->Co);
    float* dd = dV + al256((size_t)25 * T * g->Ci);
    float* part = dd + al256((size_t)25 * T * g->Ci);
    if (u_pre) U = const_cast<float*>(u_pre);
    else hipLaunchKernelGGL(wino4_weight_xform_kernel, dim3(wino_grid((size_t)g->Co * g->Ci / 2)), dim3(256), 0, st, w,
                            g->Co, g->Ci, U);
    if (md_keep) Md = md_keep;
    hipLaunchKernelGGL(wino4_dy_xform_kernel, dim3(wino_grid((size_t)T * g->Co / 2)), dim3(256), 0, st, dy, g->B, d.TH,
                       d.TW, g->Co, Md);
    if (!dense_wino_gemm(1, 25, T, g->Co, g->Ci, Md, U, dV, part, false, st)) {
    const Geom gg{1, 1, (int)T, g->Ci, 1, (int)T, g->Co, 1, 1, 1, 0, 0};
    const TilePlan tp = gemm_plan(T, g->Ci, g->Co / BK, 25, true, 1);
    float* pp = tp.splits > 1 ? part : nullptr;
    const Batch bt{T * g->Co, (long long)g->Co * g->Ci, T * g->Ci, 0};
    probe_begin(st);
    if (lean_dgrad_ok(tp, T, g->Ci, g->Co, false))
        launch_lean_dgrad(tp, 25, T, g->Ci, g->Co, Md, U, dV, pp, st);
    else
    wino_launch_tiles(tp, [&](auto bm, auto bn) {
        constexpr int BM_ = decltype(bm)::value, BN_ = decltype(bn)::value;
        dim3 grid((unsigned)(((T + BM_ - 1) / BM_) * ((g->Ci + BN_ - 1) / BN_)), tp.splits, 25);
        hipLaunchKernelGGL((conv_dgrad_kernel<BM_, BN_, true, true, 5>), grid, dim3(256), 0, st, gg, (const float*)Md,
                           (const float*)U, (const float*)nullptr, dV, MG_ACT_NONE, tp.cps, pp, bt);
    });
    probe_end(st);
    if (pp) {
        const size_t n = (size_t)25 * T * g->Ci;
        hipLaunchKernelGGL(splitk_epilogue_kernel, dim3(wino_grid(n / 4)), dim3(256), 0, st, (const float*)pp, tp.splits,
                           n, g->Ci, (const float*)nullptr, MG_ACT_NONE, dV);
    }
    }
    hipLaunchKernelGGL(wino4_dd_xform_kernel, dim3(wino_grid((size_t)T * g->Ci / 2)), dim3(256), 0, st, (const float*)dV, T,
                       g->Ci, dd);
    hipLaunchKernelGGL(wino4_dx_gather_kernel, dim3(wino_grid((size_t)g->B * g->H * g->W * g->Ci / 4)), dim3(256), 0, st,
                       (const float*)dd, g->B, g->H, g->W, g->Ci, d.TH, d.TW, dx);
    MG_CHECK_LAUNCH();
    return MG_OK;
}

int wino4_wgrad(const mg_conv_geom* g, const float* x, const float* dy, float* dw, int accumulate, float* ws,
                hipStream_t st, const float* v_in, const float* md_in) {
    const Wino4Dims d = wino4_dims(g);
    const WinoWgradPlan p = wino4_wgrad_plan(g);
    float* V = ws;
    float* Md = V + al256((size_t)25 * d.T * g->Ci);
    float* dU = Md + al256((size_t)25 * d.T * g->Co);
    float* part = dU + al256((size_t)25 * g->Co * g->Ci);
    if (v_in) V = const_cast<float*>(v_in);
    else hipLaunchKernelGGL(wino4_input_xform_kernel, dim3(wino_grid((size_t)d.T * g->Ci / 2)), dim3(256), 0, st, x, g->B,
                            g->H, g->W, g->Ci, d.TH, d.TW, V);
    if (md_in) Md = const_cast<float*>(md_in);
    else hipLaunchKernelGGL(wino4_dy_xform_kernel, dim3(wino_grid((size_t)d.T * g->Co / 2)), dim3(256), 0, st, dy, g->B,
                            d.TH, d.TW, g->Co, Md);
    if (!dense_wino_gemm(2, 25, d.T, g->Co, g->Ci, Md, V, dU, part, false, st)) {
    const Geom gg{1, 1, (int)d.T, g->Ci, 1, (int)d.T, g->Co, 1, 1, 1, 0, 0};
    float* target = p.splits > 1 ? part : dU;
    const Batch bt{d.T * g->Ci, d.T * g->Co, (long long)g->Co * g->Ci, 0};
    dim3 grid((unsigned)p.tiles, 25, p.splits);
    probe_begin(st);
    if (lean_wgrad_ok(p, d.T, g->Co, g->Ci, false))
        launch_lean_wgrad(p, 25, d.T, g->Co, g->Ci, Md, V, target, st);
    else if (p.big)
        hipLaunchKernelGGL((conv_wgrad_kernel<128, 128, true, true, 5>), grid, dim3(256), 0, st, gg, (const float*)V,
                           (const float*)Md, target, p.cps, 0, bt);
    else
        hipLaunchKernelGGL((conv_wgrad_kernel<64, 64, true, true, 5>), grid, dim3(256), 0, st, gg, (const float*)V,
                           (const float*)Md, target, p.cps, 0, bt);
    probe_end(st);
    if (p.splits > 1) {
        const size_t n = (size_t)25 * g->Co * g->Ci;
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3(wino_grid(n / 4)), dim3(256), 0, st, (const float*)part, p.splits, n,
                           dU, 0);
    }
    }
    hipLaunchKernelGGL(wino4_dweight_xform_kernel, dim3(wino_grid((size_t)g->Co * g->Ci / 2)), dim3(256), 0, st,
                       (const float*)dU, g->Co, g->Ci, dw, accumulate);
    MG_CHECK_LAUNCH();
    return MG_OK;
}

// ---------------------------------------------------------------------------------------------------------
// Winograd F(4x4,2x2) over the space-to-depth view (wino42.h): the stride-2 4x4 PatchGAN layers.  K = 4 Ci.
// ---------------------------------------------------------------------------------------------------------
bool wino42_ok(const mg_conv_geom* g) {
    static const bool off = getenv("MG_NO_WINOGRAD42") != nullptr;
    constexpr int min_c = 16;
    if (off || prec_h(g) || g->KH != 4 || g->KW != 4 || g->stride != 2 || g->pad != 2 || g->reflect || g->Ci % 16 ||
        g->Co % 16 || g->Ci < min_c || g->Co < min_c || g->H < 2 || g->W < 2)
        return false;
    // the transforms move 25/16 of the phase-channel data per pass: measured faster than the direct kernels from
    // T * 4Ci * Co ~ 4e7 up (batch 16: 64->128 at 33x65 out 124 -> 91 us, 128->256 at 17x33 out 136 -> 97 us forward),
    // slower below (64->128 at 17x33 out: 39 -> 48 us).  Read per call so the parity tests can exercise small shapes.
    const char* mw = getenv("MG_WINO42_MIN_WORK");
    const double min_work = mw ? atof(mw) : 4e7;
    const double T = (double)g->B * ((g->OH + 3) / 4) * ((g->OW + 3) / 4);
    return T * 4.0 * g->Ci * g->Co >= min_work;
}
struct Wino42Dims { long long T; int TH, TW, K4; };
Wino42Dims wino42_dims(const mg_conv_geom* g) {
    Wino42Dims d;
    d.TH = (g->OH + 3) / 4; d.TW = (g->OW + 3) / 4; d.T = (long long)g->B * d.TH * d.TW; d.K4 = 4 * g->Ci;
    return d;
}
WinoWgradPlan wino42_wgrad_plan(const mg_conv_geom* g) {
    const Wino42Dims d = wino42_dims(g);
    const int chunks = (int)((d.T + BK - 1) / BK);
    bool big = g->Co >= 128 && d.K4 >= 128 && d.T > 4096;
    int want = -1;
    const int tiles = big ? ((g->Co + 127) / 128) * ((d.K4 + 127) / 128) : ((g->Co + 63) / 64) * ((d.K4 + 63) / 64);
    int splits = want > 0 ? want : (tiles * 25 >= 512 ? (chunks >= 128 ? 3 : 1) : (768 + tiles * 25 - 1) / (tiles * 25));
    const int max_splits = chunks / 8 > 0 ? chunks / 8 : 1;
    if (splits > max_splits) splits = max_splits;
    int cps = (chunks + splits - 1) / splits;
    splits = (chunks + cps - 1) / cps;
    return {big, tiles, splits, cps};
}
inline size_t wino42_wgrad_slabs(const mg_conv_geom* g) {
    return slab_count(wino42_wgrad_plan(g).splits, dense_splits(2, 25, wino42_dims(g).T, g->Co, wino42_dims(g).K4, false));
}
size_t wino42_fwd_ws(const mg_conv_geom* g) {
    const Wino42Dims d = wino42_dims(g);
    const TilePlan tp = gemm_plan(d.T, g->Co, d.K4 / BK, 25, true, 0);
    return (al256((size_t)25 * g->Co * d.K4) + al256((size_t)25 * d.T * d.K4) + al256((size_t)25 * d.T * g->Co) +
            al256(slab_count(tp.splits, dense_splits(0, 25, d.T, g->Co, d.K4, false)) * 25 * d.T * g->Co)) * sizeof(float) + 256;
}
size_t wino42_dgrad_ws(const mg_conv_geom* g) {      // U | A dy A^T | dV | dd | split-K slabs
    const Wino42Dims d = wino42_dims(g);
    const TilePlan tp = gemm_plan(d.T, d.K4, g->Co / BK, 25, true, 1);
    return (al256((size_t)25 * g->Co * d.K4) + al256((size_t)25 * d.T * g->Co) + 2 * al256((size_t)25 * d.T * d.K4) +
            al256(slab_count(tp.splits, dense_splits(1, 25, d.T, g->Co, d.K4, false)) * 25 * d.T * d.K4)) * sizeof(float) + 256;
}
size_t wino42_wgrad_ws(const mg_conv_geom* g) {
    const Wino42Dims d = wino42_dims(g);
    const WinoWgradPlan p = wino42_wgrad_plan(g);
    const size_t cs = (mg_colsum_workspace((long long)g->B * g->OH * g->OW, g->Co) + 255) / 4;
    return (al256((size_t)25 * d.T * d.K4) + al256((size_t)25 * d.T * g->Co) + al256((size_t)25 * g->Co * d.K4) +
            al256(wino42_wgrad_slabs(g) * 25 * g->Co * d.K4) + al256(cs)) * sizeof(float) + 256;
}

int wino42_fwd(const mg_conv_geom* g, const float* x, const float* w, const float* bias, float* y, int act, float* ws,
               hipStream_t st, const float* u_pre, float* v_keep) {
    const Wino42Dims d = wino42_dims(g);
    float* U = ws;
    float* V = U + al256((size_t)25 * g->Co * d.K4);
    float* Mx = V + al256((size_t)25 * d.T * d.K4);
    float* part = Mx + al256((size_t)25 * d.T * g->Co);
    if (u_pre) U = const_cast<float*>(u_pre);
    else hipLaunchKernelGGL(wino42_weight_xform_kernel, dim3(wino_grid((size_t)g->Co * d.K4 / 2)), dim3(256), 0, st, w,
                            g->Co, g->Ci, U);
    if (v_keep) V = v_keep;
    hipLaunchKernelGGL(wino42_input_xform_kernel, dim3(wino_grid((size_t)d.T * d.K4 / 2)), dim3(256), 0, st, x, g->B, g->H,
                       g->W, g->Ci, d.TH, d.TW, V);
    if (!dense_wino_gemm(0, 25, d.T, g->Co, d.K4, V, U, Mx, part, false, st)) {
    const Geom gg{1, 1, (int)d.T, d.K4, 1, (int)d.T, g->Co, 1, 1, 1, 0, 0};
    const TilePlan tp = gemm_plan(d.T, g->Co, d.K4 / BK, 25, true, 0);
    float* pp = tp.splits > 1 ? part : nullptr;
    const Batch bt{d.T * d.K4, (long long)g->Co * d.K4, d.T * g->Co, 0};
    probe_begin(st);
    const bool k32 = use_k32(tp, d.K4);
    wino_launch_tiles(tp, [&](auto bm, auto bn) {
        constexpr int BM_ = decltype(bm)::value, BN_ = decltype(bn)::value;
        dim3 grid((unsigned)(((d.T + BM_ - 1) / BM_) * ((g->Co + BN_ - 1) / BN_)), tp.splits, 25);
        if (k32)
            launch_fwd32<BM_, BN_, 5>(grid, st, gg, V, U, nullptr, Mx, MG_ACT_NONE, tp.splits == 1 ? (1 << 29) : tp.cps / 2,
                                      pp, bt);
        else
            hipLaunchKernelGGL((conv_fwd_kernel<BM_, BN_, true, 5>), grid, dim3(256), 0, st, gg, (const float*)V,
                               (const float*)U, (const float*)nullptr, Mx, MG_ACT_NONE, tp.cps, pp, bt);
    });
    probe_end(st);
    if (pp) {
        const size_t n = (size_t)25 * d.T * g->Co;
        hipLaunchKernelGGL(splitk_epilogue_kernel, dim3(wino_grid(n / 4)), dim3(256), 0, st, (const float*)pp, tp.splits,
                           n, g->Co, (const float*)nullptr, MG_ACT_NONE, Mx);
    }
    }
    hipLaunchKernelGGL(wino42_output_xform_kernel, dim3(wino_grid((size_t)d.T * g->Co / 2)), dim3(256), 0, st,
                       (const float*)Mx, g->B, g->OH, g->OW, d.TH, d.TW, g->Co, bias, act, y);
    MG_CHECK_LAUNCH();
    return MG_OK;
}

int wino42_dgrad(const mg_conv_geom* g, const float* dy, const float* w, float* dx, float* ws, hipStream_t st,
                 const float* u_pre, float* md_keep) {
    const Wino42Dims d = wino42_dims(g);
    const long long T = d.T;
    float* U = ws;
    float* Md = U + al256((size_t)25 * g->Co * d.K4);
    float* dV = Md + al256((size_t)25 * T * g->Co);
    float* dd = dV + al256((size_t)25 * T * d.K4);
    float* part = dd + al256((size_t)25 * T * d.K4);
    if (u_pre) U = const_cast<float*>(u_pre);
    else hipLaunchKernelGGL(wino42_weight_xform_kernel, dim3(wino_grid((size_t)g->Co * d.K4 / 2)), dim3(256), 0, st, w,
                            g->Co, g->Ci, U);
    if (md_keep) Md = md_keep;
    hipLaunchKernelGGL(wino42_dy_xform_kernel, dim3(wino_grid((size_t)T * g->Co / 2)), dim3(256), 0, st, dy, g->B, g->OH,
                       g->OW, d.TH, d.TW, g->Co, Md);
    if (!dense_wino_gemm(1, 25, T, g->Co, d.K4, Md, U, dV, part, false, st)) {
    const Geom gg{1, 1, (int)T, d.K4, 1, (int)T, g->Co, 1, 1, 1, 0, 0};
    const TilePlan tp = gemm_plan(T, d.K4, g->Co / BK, 25, true, 1);
    float* pp = tp.splits > 1 ? part : nullptr;
    const Batch bt{T * g->Co, (long long)g->Co * d.K4, T * d.K4, 0};
    probe_begin(st);
    if (lean_dgrad_ok(tp, T, d.K4, g->Co, false))
        launch_lean_dgrad(tp, 25, T, d.K4, g->Co, Md, U, dV, pp, st);
    else
    wino_launch_tiles(tp, [&](auto bm, auto bn) {
        constexpr int BM_ = decltype(bm)::value, BN_ = decltype(bn)::value;
        dim3 grid((unsigned)(((T + BM_ - 1) / BM_) * ((d.K4 + BN_ - 1) / BN_)), tp.splits, 25);
        hipLaunchKernelGGL((conv_dgrad_kernel<BM_, BN_, true, true, 5>), grid, dim3(256), 0, st, gg, (const float*)Md,
                           (const float*)U, (const float*)nullptr, dV, MG_ACT_NONE, tp.cps, pp, bt);
    });
    probe_end(st);
    if (pp) {
        const size_t n = (size_t)25 * T * d.K4;
        hipLaunchKernelGGL(splitk_epilogue_kernel, dim3(wino_grid(n / 4)), dim3(256), 0, st, (const float*)pp, tp.splits,
                           n, d.K4, (const float*)nullptr, MG_ACT_NONE, dV);
    }
    }
    hipLaunchKernelGGL(wino4_dd_xform_kernel, dim3(wino_grid((size_t)T * d.K4 / 2)), dim3(256), 0, st, (const float*)dV, T,
                       d.K4, dd);
    hipLaunchKernelGGL(wino42_dx_gather_kernel, dim3(wino_grid((size_t)g->B * g->H * g->W * g->Ci / 4)), dim3(256), 0, st,
                       (const float*)dd, g->B, g->H, g->W, g->Ci, d.TH, d.TW, dx);
    MG_CHECK_LAUNCH();
    return MG_OK;
}

int wino42_wgrad(const mg_conv_geom* g, const float* x, const float* dy, float* dw, int accumulate, float* ws,
                 hipStream_t st, const float* v_in, const float* md_in) {
    const Wino42Dims d = wino42_dims(g);
    const WinoWgradPlan p = wino42_wgrad_plan(g);
    float* V = ws;
    float* Md = V + al256((size_t)25 * d.T * d.K4);
    float* dU = Md + al256((size_t)25 * d.T * g->Co);
    float* part = dU + al256((size_t)25 * g->Co * d.K4);
    if (v_in) V = const_cast<float*>(v_in);
    else hipLaunchKernelGGL(wino42_input_xform_kernel, dim3(wino_grid((size_t)d.T * d.K4 / 2)), dim3(256), 0, st, x, g->B,
                            g->H, g->W, g->Ci, d.TH, d.TW, V);
    if (md_in) Md = const_cast<float*>(md_in);
    else hipLaunchKernelGGL(wino42_dy_xform_kernel, dim3(wino_grid((size_t)d.T * g->Co / 2)), dim3(256), 0, st, dy, g->B,
                            g->OH, g->OW, d.TH, d.TW, g->Co, Md);
    if (!dense_wino_gemm(2, 25, d.T, g->Co, d.K4, Md, V, dU, part, false, st)) {
    const Geom gg{1, 1, (int)d.T, d.K4, 1, (int)d.T, g->Co, 1, 1, 1, 0, 0};
    float* target = p.splits > 1 ? part : dU;
    const Batch bt{d.T * d.K4, d.T * g->Co, (long long)g->Co * d.K4, 0};
    dim3 grid((unsigned)p.tiles, 25, p.splits);
    probe_begin(st);
    if (lean_wgrad_ok(p, d.T, g->Co, d.K4, false))
        launch_lean_wgrad(p, 25, d.T, g->Co, d.K4, Md, V, target, st);
    else if (p.big)
        hipLaunchKernelGGL((conv_wgrad_kernel<128, 128, true, true, 5>), grid, dim3(256), 0, st, gg, (const float*)V,
                           (const float*)Md, target, p.cps, 0, bt);
    else
        hipLaunchKernelGGL((conv_wgrad_kernel<64, 64, true, true, 5>), grid, dim3(256), 0, st, gg, (const float*)V,
                           (const float*)Md, target, p.cps, 0, bt);
    probe_end(st);
    if (p.splits > 1) {
        const size_t n = (size_t)25 * g->Co * d.K4;
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3(wino_grid(n / 4)), dim3(256), 0, st, (const float*)part, p.splits, n,
                           dU, 0);
    }
    }
    hipLaunchKernelGGL(wino42_dweight_xform_kernel, dim3(wino_grid((size_t)g->Co * d.K4 / 2)), dim3(256), 0, st,
                       (const float*)dU, g->Co, g->Ci, dw, accumulate);
    MG_CHECK_LAUNCH();
    return MG_OK;
}

// ---------------------------------------------------------------------------------------------------------
// Ci <= 4 layers (conv_smallc.h): VALU kernels for the data gradient and the weight gradient
// ---------------------------------------------------------------------------------------------------------
bool smallc_enabled() {
    constexpr bool off = false;
    return !off;
}
bool smallc_dgrad_ok(const mg_conv_geom* g) {
    const int s = g->stride;       // the class's weights sit in LDS: taps x Co float4s
    return smallc_enabled() && g->Ci >= 1 && g->Ci <= 4 && !g->reflect && g->Co % 4 == 0 && g->Co >= 16 &&
           (size_t)((g->KH + s - 1) / s) * ((g->KW + s - 1) / s) * g->Co * 16 <= 60 * 1024;
}
// 0: not eligible; otherwise the template instance id
int smallc_wgrad_kind(const mg_conv_geom* g) {
    if (!smallc_enabled() || g->Co < 16 || ((g->stride * g->Ci) & 1)) return 0;
    if ((size_t)g->KH * (((g->OW - 1) * g->stride + g->KW) * g->Ci + 4) * sizeof(float) > 60000) return 0;
    if ((long long)g->B * g->OH > 65535LL * 32) return 0;
    if (g->KH == 4 && g->KW == 4 && g->Ci == 3) return 1;
    if (g->KH == 7 && g->KW == 7 && g->Ci == 2) return 2;
    if (g->KH == 3 && g->KW == 3 && g->Ci == 4) return 3;
    return 0;
}
// even stride * Ci keeps every pixel's patch segment 8-byte aligned in the staged rows
struct SmallcWgradPlan { int wgs, rowlen; size_t lds; int mfma_g; };     // wgs: partial rows; mfma_g: 0 or the MFMA kernel's pixel groups
// the MFMA form (conv_smallc_wgrad_mfma_kernel): pixel groups per output row (each writes a partial row), 0 = VALU kernel
int smallc_wgrad_mfma_groups(const mg_conv_geom* g, int rowlen) {
    constexpr bool off = false;
    if (off || (size_t)g->KH * rowlen > 256 * 16 || g->Co % 64 != 0) return 0;
    if (g->KH == 7 && g->KW == 7 && g->Ci == 2 && g->stride == 1) return 1;     // K = 98: four k blocks, one per wave
    if (g->KH == 4 && g->KW == 4 && g->Ci == 3 && g->stride == 2) return 2;     // K = 48: two k blocks x two pixel groups
    return 0;
}
SmallcWgradPlan smallc_wgrad_plan(const mg_conv_geom* g) {
    const int ncols = (g->OW - 1) * g->stride + g->KW;
    const int rowlen = (ncols * g->Ci + 3) / 4 * 4;
    const int mg = smallc_wgrad_mfma_groups(g, rowlen);
    return {g->B * g->OH * (mg ? mg : 1), rowlen, (size_t)g->KH * rowlen * sizeof(float), mg};
}
// workspace: per-row partials | column-sum scratch of their reduction | column-sum scratch of the bias gradient
struct SmallcWs { size_t part, red, cs, total; };      // float offsets / counts
SmallcWs smallc_wgrad_layout(const mg_conv_geom* g) {
    const SmallcWgradPlan p = smallc_wgrad_plan(g);
    const size_t n = (size_t)g->Co * g->KH * g->KW * g->Ci;
    SmallcWs w;
    w.part = 0;
    w.red = al256((size_t)p.wgs * n);
    w.cs = w.red + al256((mg_colsum_workspace(p.wgs, (int)n) + 255) / 4);
    w.total = w.cs + al256((mg_colsum_workspace((long long)g->B * g->OH * g->OW, g->Co) + 255) / 4);
    return w;
}
size_t smallc_wgrad_ws(const mg_conv_geom* g) { return smallc_wgrad_layout(g).total * sizeof(float) + 256; }
int smallc_wgrad(const mg_conv_geom* g, const float* x, const float* dy, float* dw, int accumulate, float* ws,
                 hipStream_t st) {
    const SmallcWgradPlan p = smallc_wgrad_plan(g);
    const SmallcWs lay = smallc_wgrad_layout(g);
    const Geom gg = to_geom(g);
    const dim3 grid((unsigned)(g->B * g->OH), (unsigned)((g->Co + 63) / 64));
    if (p.mfma_g) {
        const size_t lds = p.lds + 16;               // + the zeroed slot the k >= K lanes read
        if (g->KH == 7) hipLaunchKernelGGL((conv_smallc_wgrad_mfma_kernel<7, 7, 2, 1>), grid, dim3(256), lds, st, gg, x, dy, ws, p.rowlen, (int)prec_h(g));
        else hipLaunchKernelGGL((conv_smallc_wgrad_mfma_kernel<4, 4, 3, 2>), grid, dim3(256), lds, st, gg, x, dy, ws, p.rowlen, (int)prec_h(g));
    } else
    switch (smallc_wgrad_kind(g)) {
    case 1: hipLaunchKernelGGL((conv_smallc_wgrad_kernel<4, 4, 3>), grid, dim3(256), p.lds, st, gg, x, dy, ws, p.rowlen, (int)prec_h(g)); break;
    case 2: hipLaunchKernelGGL((conv_smallc_wgrad_kernel<7, 7, 2>), grid, dim3(448), p.lds, st, gg, x, dy, ws, p.rowlen, (int)prec_h(g)); break;
    case 3: hipLaunchKernelGGL((conv_smallc_wgrad_kernel<3, 3, 4>), grid, dim3(192), p.lds, st, gg, x, dy, ws, p.rowlen, (int)prec_h(g)); break;
    default: return MG_ERR_UNSUPPORTED;
    }
    MG_CHECK_LAUNCH();
    const size_t n = (size_t)g->Co * g->KH * g->KW * g->Ci;
    return mg_colsum(ws, p.wgs, (int)n, dw, accumulate, ws + lay.red, mg_colsum_workspace(p.wgs, (int)n), st);
}
// forward on the MFMA pipe (conv_smallc_fwd_kernel): 0 = not eligible, else the template instance
int smallc_fwd_kind(const mg_conv_geom* g) {
    constexpr bool off = false;
    if (!smallc_enabled() || off || g->Co % 64 != 0 || (long long)g->B * g->OH > 0x7fffffffLL) return 0;
    if ((size_t)g->KH * (((g->OW - 1) * g->stride + g->KW) * g->Ci + 4) > 256 * 16) return 0;      // staged through 16 registers per thread
    if (g->KH == 7 && g->KW == 7 && g->Ci == 2 && g->stride == 1) return 1;      // the generator stem: 66 against 93 us
    // the 3 -> 64 4x4 stride-2 first discriminator layer (K = 48, 129-pixel rows = 5 pixel blocks for 4 waves) measured
    // SLOWER here than on the generic kernel (44 against 37 us at batch 16): instance kept for MG_SMALLC_FWD_D=1 only
    constexpr bool with_d = false;
    if (with_d && g->KH == 4 && g->KW == 4 && g->Ci == 3 && g->stride == 2) return 2;
    return 0;
}
int smallc_fwd(const mg_conv_geom* g, const float* x, const float* w, const float* bias, float* y, int act, hipStream_t st) {
    const SmallcWgradPlan p = smallc_wgrad_plan(g);     // the same staged rows
    const Geom gg = to_geom(g);
    const int rows = g->B * g->OH, cblocks = (g->Co + 63) / 64;
    int rpw = 1;       // rows per workgroup (the weights' LDS image is paid once per workgroup): ~4 resident workgroups per CU (81 VGPRs, 40 KB)
    while ((long long)((rows + rpw - 1) / rpw) * cblocks > 1024 && rpw < 8) ++rpw;
    const dim3 grid((unsigned)((rows + rpw - 1) / rpw), (unsigned)cblocks);
    const size_t lds = p.lds + (size_t)64 * ((g->KH * g->KW * g->Ci) | 1) * sizeof(float);
    switch (smallc_fwd_kind(g)) {
    case 1: hipLaunchKernelGGL((conv_smallc_fwd_kernel<7, 7, 2, 1>), grid, dim3(256), lds, st, gg, x, w, bias, y, act, p.rowlen, rpw, (int)prec_h(g)); break;
    case 2: hipLaunchKernelGGL((conv_smallc_fwd_kernel<4, 4, 3, 2>), grid, dim3(256), lds, st, gg, x, w, bias, y, act, p.rowlen, rpw, (int)prec_h(g)); break;
    default: return MG_ERR_UNSUPPORTED;
    }
    MG_CHECK_LAUNCH();
    return MG_OK;
}
int smallc_dgrad(const mg_conv_geom* g, const float* dy, const float* w, float* dx, hipStream_t st) {
    const Geom gg = to_geom(g);
    const int s = g->stride;
    const long long Mc = (long long)g->B * ((g->H + s - 1) / s) * ((g->W + s - 1) / s);
    const dim3 grid((unsigned)((Mc + 255) / 256), (unsigned)(s * s));
    // the heaviest class's taps x Co float4s of weights in LDS
    const size_t wl_bytes = (size_t)((g->KH + s - 1) / s) * ((g->KW + s - 1) / s) * g->Co * 16;
#define MG_SMALLC_DG(CI_) do { \
        if (prec_h(g)) hipLaunchKernelGGL((conv_smallc_dgrad_kernel<CI_, true>), grid, dim3(256), wl_bytes, st, gg, dy, w, dx); \
        else hipLaunchKernelGGL((conv_smallc_dgrad_kernel<CI_, false>), grid, dim3(256), wl_bytes, st, gg, dy, w, dx); } while (0)
    switch (g->Ci) {
    case 1: MG_SMALLC_DG(1); break;
    case 2: MG_SMALLC_DG(2); break;
    case 3: MG_SMALLC_DG(3); break;
    default: MG_SMALLC_DG(4); break;
    }
#undef MG_SMALLC_DG
    MG_CHECK_LAUNCH();
    return MG_OK;
}

}  // namespace

extern "C" {

// (library-internal: conv_rowdot.hip; hidden from the .so's exports)
__attribute__((visibility("hidden"))) int mg_conv_rowdot_kq(const mg_conv_geom* g);
__attribute__((visibility("hidden"))) int mg_conv_rowdot_fwd(const mg_conv_geom* g, const float* x, const float* w, const float* bias, float* y, int act,
                       void* stream);
__attribute__((visibility("hidden"))) size_t mg_conv_rowdot_wgrad_workspace(const mg_conv_geom* g);
__attribute__((visibility("hidden"))) int mg_conv_rowdot_wgrad(const mg_conv_geom* g, const float* x, const float* dy, float* dw, float* dbias,
                         int accumulate, void* workspace, size_t workspace_bytes, void* stream);

int mg_abi_version(void) { return 4; }
int mg_conv_geom_size(void) { return (int)sizeof(mg_conv_geom); }

void mg_probe_arm(void* e0, void* e1) {
    g_probe_e0 = (hipEvent_t)e0;
    g_probe_e1 = (hipEvent_t)e1;
}

// FLOPs the main GEMM kernel of a pass issues for this geometry (2*M*N*K of the GEMM it actually runs: the direct
// convolution's 2*MACs, or 16 Winograd-domain GEMMs = 1/2.25 of that on the unpadded tile grid).
double mg_conv_plan_flops(int pass, const mg_conv_geom* g) {
    if (!geom_ok(g)) return 0.0;
    const double direct = 2.0 * g->B * g->OH * g->OW * (double)g->Co * g->KH * g->KW * g->Ci;
    if (co1_gemm_ok(g)) return 2.0 * CO1_TAPS * (double)g->B * g->H * g->W * g->Ci;      // the 64-tap GEMM (conv_co1.h)
    if (wino_ok(g) && !mg_conv_rowdot_kq(g)) {
        const WinoDims d = wino_dims(g);
        const double T = (double)d.T;
        return 2.0 * 16.0 * T * (double)g->Co * g->Ci;
    }
    if (wino4_ok(g) && !mg_conv_rowdot_kq(g)) return 2.0 * 25.0 * (double)wino4_dims(g).T * (double)g->Co * g->Ci;
    if (wino42_ok(g)) return 2.0 * 25.0 * (double)wino42_dims(g).T * (double)g->Co * 4.0 * g->Ci;
    return direct;
}

// Name of the kernel instance a pass would launch for this geometry (matches the symbol rocprofv3 reports,
// minus the anonymous-namespace prefix).  pass: 0 fwd, 1 dgrad, 2 wgrad.
int mg_conv_plan_splits(int pass, const mg_conv_geom* g) {
    char name[96];
    if (!geom_ok(g) || mg_conv_plan_name(pass, g, name, (int)sizeof name) != MG_OK) return 0;
    if (pass == 0 && strncmp(name, "conv_fwd_dma_kernel", 19) == 0) return conv_dma_fwd_plan(g).splits;
    if (pass == 2 && strncmp(name, "conv_wgrad_dma_kernel", 21) == 0) return conv_dma_wgrad_plan(g).splits;
    if (pass == 1 && strncmp(name, "conv_dgrad_dma_kernel", 21) == 0) {
        mg_conv_geom gp;
        return (conv_dma_dgrad_ok(g) ? conv_dma_dgrad_plan(g) : (cd_reflect_dgrad_geom(g, &gp), conv_dma_dgrad_plan(&gp))).splits;
    }
    return 0;
}
int mg_conv_plan_name(int pass, const mg_conv_geom* g, char* out, int out_len) {
    if (!geom_ok(g) || !out || out_len < 64) return MG_ERR_ARG;
    const int kq = mg_conv_rowdot_kq(g);
    if (co1_gemm_ok(g)) {          // single-output-channel layers as tap GEMMs (conv_co1.h)
        snprintf(out, out_len, pass == 0 ? "dgemm32g_kernel<64, 128, 2, 2, 0, 0, 2, 0, 0>"
                               : pass == 1 ? "dgemm32g_kernel<128, 64, 2, 2, 1, 1, 2, 0, 0>" : "dgemm32g_kernel<64, 64, 2, 2, 0, 1, 2, 0, 0>");
        return MG_OK;
    }
    if (h16_ok(g) && !kq) {
        const long long px = (long long)g->B * g->OH * g->OW;
        const int Kw = g->KH * g->KW * g->Ci;
        bool deep;
        if (pass == 0) deep = h16_deep(px, g->Co, h16_plan(px, g->Co, Kw, false).splits, false);
        else if (pass == 1) deep = h16_deep((long long)g->B * g->H * g->W, g->Ci, h16_plan((long long)g->B * g->H * g->W, g->Ci, g->KH * g->KW * g->Co, true).splits, true);
        else deep = h16_deep(g->Co, Kw, h16_plan(g->Co, Kw, h16_mp(px), false).splits, false);
        HgArgs probe{};
        probe.N = pass == 1 ? g->Ci : g->Co; probe.K = pass == 1 ? g->KH * g->KW * g->Co : Kw;
        if (pass == 2 && h16_wgrad_as(g)) snprintf(out, out_len, "hgemm_as_kernel<%d, false, false>", h16_mp(px) / 64);
        else if (pass != 2 && h16_sa_on() && hgemm_sa_ok(probe)) snprintf(out, out_len, pass == 1 ? "hgemm_sa_kernel<true, false>" : "hgemm_sa_kernel<false, true>");
        else
        snprintf(out, out_len, pass == 1 ? "hgemm_kernel<128, 128, 4, 2, true, %d>" : "hgemm_kernel<128, 128, 4, 2, false, %d>", deep ? 3 : 2);
    } else if (wino_ok(g) && !kq && 
        dense_plan(pass, 16, wino_dims(g).T, g->Co, g->Ci, prec_h(g)).ok) {
        dense_name(pass, 16, dense_plan(pass, 16, wino_dims(g).T, g->Co, g->Ci, prec_h(g)), dense_dims(pass, wino_dims(g).T, g->Co, g->Ci).N, out, out_len);
    } else if (wino4_ok(g) && !kq && dense_plan(pass, 25, wino4_dims(g).T, g->Co, g->Ci, false).ok) {
        dense_name(pass, 25, dense_plan(pass, 25, wino4_dims(g).T, g->Co, g->Ci, false), dense_dims(pass, wino4_dims(g).T, g->Co, g->Ci).N, out, out_len);
    } else if (wino42_ok(g) && !kq && dense_plan(pass, 25, wino42_dims(g).T, g->Co, wino42_dims(g).K4, false).ok) {
        dense_name(pass, 25, dense_plan(pass, 25, wino42_dims(g).T, g->Co, wino42_dims(g).K4, false), dense_dims(pass, wino42_dims(g).T, g->Co, wino42_dims(g).K4).N, out, out_len);
    } else if (wino_ok(g) && !kq) {
        const WinoDims d = wino_dims(g);
        if (pass == 0) {
            const TilePlan tp = gemm_plan(d.T, g->Co, g->Ci / BK, 16, true, 0);
            const bool k32h = fwd32_enabled() && g->Ci % BK2 == 0 && (tp.splits == 1 || tp.cps % 2 == 0);
            if (prec_h(g) && k32h)
                snprintf(out, out_len, "conv_fwd32_kernel<%d, %d, 3>", tp.bm, tp.bn);
            else if (prec_h(g))
                snprintf(out, out_len, "conv_fwd_kernel<%d, %d, true, 3>", tp.bm, tp.bn);
            else if (use_k32(tp, g->Ci))
                snprintf(out, out_len, "conv_fwd32_kernel<%d, %d, 1>", tp.bm, tp.bn);
            else
                snprintf(out, out_len, "conv_fwd_kernel<%d, %d, true, 1>", tp.bm, tp.bn);
        } else if (pass == 1) {
            const TilePlan tp = gemm_plan(d.T, g->Ci, g->Co / BK, 16, true, 1);
            if (lean_dgrad_ok(tp, d.T, g->Ci, g->Co, prec_h(g))) snprintf(out, out_len, "dense_nn64_kernel<1>");
            else
            snprintf(out, out_len, "conv_dgrad_kernel<%d, %d, true, true, %d>", tp.bm, tp.bn, prec_h(g) ? 3 : 1);
        } else {
            const WinoWgradPlan p = wino_wgrad_plan(g);
            if (lean_wgrad_ok(p, d.T, g->Co, g->Ci, prec_h(g)))
                snprintf(out, out_len, lean_wgrad_wide(p, 16, g->Co, g->Ci) ? "dense_tn64_kernel<1, 1, 2>" : "dense_tn64_kernel<1, 1, 1>");
            else
            snprintf(out, out_len, "conv_wgrad_kernel<%d, %d, true, true, %d>", p.big ? 128 : 64, p.big ? 128 : 64,
                     prec_h(g) ? 3 : 1);
        }
    } else if (wino4_ok(g) && !kq) {
        const Wino4Dims d = wino4_dims(g);
        if (pass == 0) {
            const TilePlan tp = gemm_plan(d.T, g->Co, g->Ci / BK, 25, true, 0);
            if (use_k32(tp, g->Ci)) snprintf(out, out_len, "conv_fwd32_kernel<%d, %d, 5>", tp.bm, tp.bn);
            else snprintf(out, out_len, "conv_fwd_kernel<%d, %d, true, 5>", tp.bm, tp.bn);
        } else if (pass == 1) {
            const TilePlan tp = gemm_plan(d.T, g->Ci, g->Co / BK, 25, true, 1);
            if (lean_dgrad_ok(tp, d.T, g->Ci, g->Co, false)) snprintf(out, out_len, "dense_nn64_kernel<5>");
            else
            snprintf(out, out_len, "conv_dgrad_kernel<%d, %d, true, true, 5>", tp.bm, tp.bn);
        } else {
            const WinoWgradPlan p = wino4_wgrad_plan(g);
            if (lean_wgrad_ok(p, d.T, g->Co, g->Ci, false)) snprintf(out, out_len, "dense_tn64_kernel<1, 5, 1>");
            else
            snprintf(out, out_len, "conv_wgrad_kernel<%d, %d, true, true, 5>", p.big ? 128 : 64, p.big ? 128 : 64);
        }
    } else if (wino42_ok(g) && !kq) {
        const Wino42Dims d = wino42_dims(g);
        if (pass == 0) {
            const TilePlan tp = gemm_plan(d.T, g->Co, d.K4 / BK, 25, true, 0);
            if (use_k32(tp, d.K4)) snprintf(out, out_len, "conv_fwd32_kernel<%d, %d, 5>", tp.bm, tp.bn);
            else snprintf(out, out_len, "conv_fwd_kernel<%d, %d, true, 5>", tp.bm, tp.bn);
        } else if (pass == 1) {
            const TilePlan tp = gemm_plan(d.T, d.K4, g->Co / BK, 25, true, 1);
            if (lean_dgrad_ok(tp, d.T, d.K4, g->Co, false)) snprintf(out, out_len, "dense_nn64_kernel<5>");
            else
            snprintf(out, out_len, "conv_dgrad_kernel<%d, %d, true, true, 5>", tp.bm, tp.bn);
        } else {
            const WinoWgradPlan p = wino42_wgrad_plan(g);
            if (lean_wgrad_ok(p, d.T, g->Co, d.K4, false)) snprintf(out, out_len, "dense_tn64_kernel<1, 5, 1>");
            else
            snprintf(out, out_len, "conv_wgrad_kernel<%d, %d, true, true, 5>", p.big ? 128 : 64, p.big ? 128 : 64);
        }
    } else if (pass == 0 && !kq && smallc_fwd_kind(g)) {
        snprintf(out, out_len, "conv_smallc_fwd_kernel<%d, %d, %d, %d>", g->KH, g->KW, g->Ci, g->stride);
    } else if (pass == 1 && smallc_dgrad_ok(g)) {
        snprintf(out, out_len, "conv_smallc_dgrad_kernel<%d, %s>", g->Ci, prec_h(g) ? "true" : "false");
    } else if (pass == 2 && !kq && smallc_wgrad_kind(g)) {
        if (smallc_wgrad_plan(g).mfma_g) snprintf(out, out_len, "conv_smallc_wgrad_mfma_kernel<%d, %d, %d, %d>", g->KH, g->KW, g->Ci, g->stride);
        else snprintf(out, out_len, "conv_smallc_wgrad_kernel<%d, %d, %d>", g->KH, g->KW, g->Ci);
    } else if (kq && pass == 0) {
        snprintf(out, out_len, "conv_rowdot_fwd_kernel<%d>", kq);
    } else if (kq && pass == 2) {
        snprintf(out, out_len, "conv_rowdot_wgrad_kernel<%d>", kq);
    } else if (pass == 1 && cd_dgrad_any(g) && !smallc_dgrad_ok(g)) {
        mg_conv_geom gp;
        const CdPlan cp = conv_dma_dgrad_ok(g) ? conv_dma_dgrad_plan(g) : (cd_reflect_dgrad_geom(g, &gp), conv_dma_dgrad_plan(&gp));
        snprintf(out, out_len, "conv_dgrad_dma_kernel<%d, %d, %s, %d>", cp.bm, cp.bn, prec_h(g) ? "true" : "false", prec_h(g) ? cd_half_nbuf() : 2);
    } else if (pass == 0 && conv_dma_fwd_ok(g)) {
        const CdPlan cp = conv_dma_fwd_plan(g);
        snprintf(out, out_len, "conv_fwd_dma_kernel<%d, %d, %s, %d>", cp.bm, cp.bn, prec_h(g) ? "true" : "false", prec_h(g) ? cd_half_nbuf() : 2);
    } else if (pass == 2 && conv_dma_wgrad_ok(g)) {
        const CdPlan cp = conv_dma_wgrad_plan(g);
        const bool rr = conv_dma_wgrad_rowreg(g) && !getenv("MG_NO_WGRAD_RR") && (!prec_h(g) || cd_half_nbuf() == 2);
        snprintf(out, out_len, "conv_wgrad_dma_kernel<%d, %d, %s, %d, %s>", cp.bm, cp.bn, prec_h(g) ? "true" : "false", prec_h(g) ? cd_half_nbuf() : 2,
                 rr ? "true" : "false");
    } else if (pass == 0) {
        const TilePlan tp = fwd_plan(g);
        const bool vec16 = g->Ci % BK == 0;
        const bool k32h = vec16 && fwd32_enabled() && g->Ci % BK2 == 0 && (tp.splits == 1 || tp.cps % 2 == 0);
        if (prec_h(g) && k32h)
            snprintf(out, out_len, "conv_fwd32_kernel<%d, %d, 2>", tp.bm, tp.bn);
        else if (!prec_h(g) && use_k32(tp, g->Ci))
            snprintf(out, out_len, "conv_fwd32_kernel<%d, %d, 0>", tp.bm, tp.bn);
        else
            snprintf(out, out_len, "conv_fwd_kernel<%d, %d, %s, %d>", tp.bm, tp.bn, (g->Ci % BK == 0) ? "true" : "false",
                     prec_h(g) ? 2 : 0);
    } else if (pass == 1) {
        const TilePlan tp = dgrad_plan(g);
        snprintf(out, out_len, "conv_dgrad_kernel<%d, %d, %s, %s, %d>", tp.bm, tp.bn, (g->Co % BK == 0) ? "true" : "false",
                 (g->Ci % 4 == 0) ? "true" : "false", prec_h(g) ? 2 : 0);
    } else if (pass == 2) {
        const WgradPlan p = wgrad_plan(g);
        snprintf(out, out_len, "conv_wgrad_kernel<%d, %d, %s, %s, %d>", p.big ? 128 : 64, p.big ? 128 : 64,
                 (g->Co % 4 == 0) ? "true" : "false", (g->Ci % 4 == 0) ? "true" : "false", prec_h(g) ? 2 : 0);
    } else {
        return MG_ERR_ARG;
    }
    return MG_OK;
}

size_t mg_conv_fwd_workspace(const mg_conv_geom* g) {
    if (!geom_ok(g)) return 0;
    if (co1_gemm_ok(g)) return co1_fwd_ws(g);
    if (h16_ok(g)) return h16_fwd_ws(g);
    if (wino_ok(g)) return wino_fwd_ws(g);
    if (wino4_ok(g) && !mg_conv_rowdot_kq(g)) return wino4_fwd_ws(g);
    if (wino42_ok(g)) return wino42_fwd_ws(g);
    const TilePlan tp = fwd_plan(g);
    int sp = tp.splits;
    if (conv_dma_fwd_ok(g) && conv_dma_fwd_plan(g).splits > sp) sp = conv_dma_fwd_plan(g).splits;
    const size_t stage = (conv_dma_fwd_ok(g) && conv_dma_half(g)) ? conv_dma_h_x_bytes(g) + conv_dma_h_w_bytes(g) : 0;
    return stage + (sp > 1 ? (size_t)sp * g->B * g->OH * g->OW * g->Co * sizeof(float) + 256 : 256);
}
size_t mg_conv_dgrad_workspace(const mg_conv_geom* g) {
    if (!geom_ok(g)) return 0;
    if (co1_gemm_ok(g)) return co1_dgrad_ws(g);
    if (h16_ok(g)) return h16_dgrad_ws(g);
    if (wino_ok(g)) return wino_dgrad_ws(g);
    if (wino4_ok(g)) return wino4_dgrad_ws(g);
    if (wino42_ok(g)) return wino42_dgrad_ws(g);
    const TilePlan tp = dgrad_plan(g);
    int sp = tp.splits;
    size_t base = sp > 1 ? (size_t)sp * g->B * g->H * g->W * g->Ci * sizeof(float) + 256 : 256;
    mg_conv_geom gp;
    if (conv_dma_dgrad_ok(g)) {
        if (cd_dgrad_ws(g) > base) base = cd_dgrad_ws(g);
    } else if (cd_reflect_dgrad_geom(g, &gp)) {
        const size_t need = cd_reflect_dxp_bytes(&gp) + cd_dgrad_ws(&gp);
        if (need > base) base = need;
    }
    return base;
}

size_t mg_conv_wino_weights_bytes(const mg_conv_geom* g) {
    if (!geom_ok(g)) return 0;
    if (co1_gemm_ok(g)) return (size_t)CO1_TAPS * g->Ci * sizeof(float);      // tap GEMM: the weights zero-padded to 64 tap rows
    if (mg_conv_rowdot_kq(g)) return 0;
    if (h16_ok(g)) return h16_weights_bytes(g);         // the float16 weight copy of the autocast GEMM path (conv_h16.h)
    if (conv_dma_h_any(g)) return h16_weights_bytes(g); // ... and of the float16 implicit GEMMs (conv_dma.h)
    if (wino4_ok(g)) return (size_t)25 * g->Co * g->Ci * sizeof(float);
    if (wino42_ok(g)) return (size_t)25 * g->Co * 4 * g->Ci * sizeof(float);
    if (!wino_ok(g)) return 0;
    return (size_t)16 * g->Co * g->Ci * sizeof(float);
}
int mg_conv_wino_prepare(const mg_conv_geom* g, const float* w, float* u, void* stream) {
    if (!mg_conv_wino_weights_bytes(g) || !w || !u || !aligned16(w) || !aligned16(u)) return MG_ERR_ARG;
    if (co1_gemm_ok(g)) {
        co1_pad_w(g, w, u, (hipStream_t)stream);
        MG_CHECK_LAUNCH();
        return MG_OK;
    }
    if (h16_ok(g) || conv_dma_h_any(g)) return h16_prepare(g, w, u, (hipStream_t)stream);
    if (wino4_ok(g)) {
        hipLaunchKernelGGL(wino4_weight_xform_kernel, dim3(wino_grid((size_t)g->Co * g->Ci / 2)), dim3(256), 0,
                           (hipStream_t)stream, w, g->Co, g->Ci, u);
        MG_CHECK_LAUNCH();
        return MG_OK;
    }
    if (wino42_ok(g)) {
        hipLaunchKernelGGL(wino42_weight_xform_kernel, dim3(wino_grid((size_t)g->Co * 4 * g->Ci / 2)), dim3(256), 0,
                           (hipStream_t)stream, w, g->Co, g->Ci, u);
        MG_CHECK_LAUNCH();
        return MG_OK;
    }
    hipLaunchKernelGGL(wino_weight_xform_kernel, dim3(wino_grid((size_t)g->Co * g->Ci / 4)), dim3(256), 0,
                       (hipStream_t)stream, w, g->Co, g->Ci, u);
    MG_CHECK_LAUNCH();
    return MG_OK;
}

size_t mg_conv_wino_tiles_bytes(const mg_conv_geom* g, int which) {
    if (!mg_conv_wino_weights_bytes(g) || h16_ok(g)) return 0;
    if (co1_gemm_ok(g)) return which == 1 ? co1_zt_bytes(g) : (which == 0 ? co1_xr_bytes(g) : 0);   // Gt (data gradient -> weight gradient); autocast: the rounded x
    if (conv_dma_h_any(g)) {
        // float16 implicit GEMMs: the "tiles" are the float16 copies of x (forward -> weight gradient) and of dy (data
        // gradient -> weight gradient), so that each tensor is cast once per step
        if (!conv_dma_wgrad_ok(g)) return 0;
        if (which == 0) return conv_dma_fwd_ok(g) ? conv_dma_h_x_bytes(g) : 0;
        if (which == 1) return cd_dgrad_any(g) ? conv_dma_h_dy_bytes(g) : 0;
        return 0;
    }
    if (wino4_ok(g)) {
        const Wino4Dims d4 = wino4_dims(g);
        return which == 0 ? (size_t)25 * d4.T * g->Ci * sizeof(float) : which == 1 ? (size_t)25 * d4.T * g->Co * sizeof(float) : 0;
    }
    if (wino42_ok(g)) {
        const Wino42Dims d2 = wino42_dims(g);
        return which == 0 ? (size_t)25 * d2.T * d2.K4 * sizeof(float) : which == 1 ? (size_t)25 * d2.T * g->Co * sizeof(float) : 0;
    }
    const WinoDims d = wino_dims(g);
    if (which == 0) return (size_t)16 * d.T * g->Ci * sizeof(float);
    if (which == 1) return (size_t)16 * d.T * g->Co * sizeof(float);
    return 0;
}
static bool wino_tiles_ok(const mg_conv_geom* g, const mg_wino_tiles* t) {
    if (!t || (!t->u && !t->v && !t->md)) return true;
    if (!mg_conv_wino_weights_bytes(g)) return false;
    return (!t->u || aligned16(t->u)) && (!t->v || aligned16(t->v)) && (!t->md || aligned16(t->md));
}

int mg_conv_fwd(const mg_conv_geom* g, const float* x, const float* w, const float* bias, float* y, int act,
                void* workspace, size_t workspace_bytes, void* stream) {
    return mg_conv_fwd_w(g, x, w, bias, y, act, workspace, workspace_bytes, stream, nullptr);
}
int mg_conv_dgrad(const mg_conv_geom* g, const float* dy, const float* w, const float* bias, float* dx, int act,
                  void* workspace, size_t workspace_bytes, void* stream) {
    return mg_conv_dgrad_w(g, dy, w, bias, dx, act, workspace, workspace_bytes, stream, nullptr);
}
int mg_conv_wgrad(const mg_conv_geom* g, const float* x, const float* dy, float* dw, float* dbias, int accumulate,
                  void* workspace, size_t workspace_bytes, void* stream) {
    return mg_conv_wgrad_w(g, x, dy, dw, dbias, accumulate, workspace, workspace_bytes, stream, nullptr);
}

int mg_conv_wino_md_from_norm_ok(const mg_conv_geom* g) {
    return (geom_ok(g) && wino_ok(g) && !prec_h(g) && !mg_conv_rowdot_kq(g) &&
            wino_out_norm_ok(g->OH / 2, g->OW / 2, g->Co)) ? 1 : 0;
}
int mg_instnorm_bwd_wino_md(const mg_conv_geom* g, const float* gy, const float* y_raw, const float* mean, const float* rstd,
                            int act, float* md, void* stream) {
    if (!gy || !y_raw || !mean || !rstd || !md) return MG_ERR_ARG;
    if (!mg_conv_wino_md_from_norm_ok(g)) return MG_ERR_UNSUPPORTED;
    if (act != MG_ACT_NONE && act != MG_ACT_RELU && act != MG_ACT_LRELU02) return MG_ERR_UNSUPPORTED;   // the only derivatives wino_norm_bwd_dy_kernel has
    if (!aligned16(gy) || !aligned16(y_raw) || !aligned16(mean) || !aligned16(rstd) || !aligned16(md)) return MG_ERR_ARG;
    const WinoDims d = wino_dims(g);
    const dim3 grid(g->Co / 32, g->B);
    const int nt = (d.TH * d.TW + 31) / 32;
    hipStream_t st = (hipStream_t)stream;
#define MG_NORM_BWD_DY(NT_) hipLaunchKernelGGL(wino_norm_bwd_dy_kernel<NT_>, grid, dim3(256), 0, st, gy, y_raw, mean, rstd, g->B, d.TH, d.TW, g->Co, act, md)
    if (nt == 1) MG_NORM_BWD_DY(1); else if (nt == 2) MG_NORM_BWD_DY(2); else if (nt == 3) MG_NORM_BWD_DY(3); else if (nt == 4) MG_NORM_BWD_DY(4); else MG_NORM_BWD_DY(5);
#undef MG_NORM_BWD_DY
    MG_CHECK_LAUNCH();
    return MG_OK;
}
size_t mg_conv_fwd_instnorm_workspace(const mg_conv_geom* g) {
    if (!geom_ok(g)) return 0;
    const size_t a = mg_conv_fwd_workspace(g), b = mg_instnorm_workspace(g->B, g->OH * g->OW, g->Co);
    return a > b ? a : b;
}
int mg_conv_fwd_instnorm_w(const mg_conv_geom* g, const float* x, const float* w, const float* bias, float* y_raw, float eps,
                           int act, const float* residual, float* y, float* mean, float* rstd, void* workspace,
                           size_t workspace_bytes, void* stream, const mg_wino_tiles* wt) {
    return mg_conv_fwd_instnorm_h(g, x, w, bias, y_raw, eps, act, residual, y, mean, rstd, workspace, workspace_bytes, stream, wt, nullptr);
}
// The fused F(2x2,3x3) output transform + InstanceNorm kernel can also write the NEXT 3x3 stride-1 pad-1 layer's input image
// (wino.h: wino_out_norm_kernel<NT, true>): float32, maps of <= 64 tiles per sample.
static bool wino_vnext_ok(const mg_conv_geom* g) {
    return geom_ok(g) && wino_ok(g) && !prec_h(g) && !mg_conv_rowdot_kq(g) && wino_out_norm_ok(g->OH / 2, g->OW / 2, g->Co) &&
           wino_out_norm_next_ok(g->OH / 2, g->OW / 2, g->Co) && g->Co % 16 == 0;
}
int mg_conv_wino_vnext_ok(const mg_conv_geom* g) { return g && wino_vnext_ok(g) ? 1 : 0; }

static int conv_fwd_instnorm_impl(const mg_conv_geom* g, const float* x, const float* w, const float* bias, float* y_raw, float eps,
                                  int act, const float* residual, float* y, float* mean, float* rstd, void* workspace,
                                  size_t workspace_bytes, void* stream, const mg_wino_tiles* wt, void* y16, float* v_next,
                                  int next_reflect);

int mg_conv_fwd_instnorm_next(const mg_conv_geom* g, const float* x, const float* w, const float* bias, float* y_raw, float eps,
                              int act, const float* residual, float* y, float* mean, float* rstd, void* workspace,
                              size_t workspace_bytes, void* stream, const mg_wino_tiles* wt, float* v_next, int next_reflect) {
    if (!g || !v_next || !aligned16(v_next) || !wino_vnext_ok(g)) return MG_ERR_ARG;
    return conv_fwd_instnorm_impl(g, x, w, bias, y_raw, eps, act, residual, y, mean, rstd, workspace, workspace_bytes, stream, wt, nullptr,
                                  v_next, next_reflect);
}

int mg_conv_fwd_instnorm_h(const mg_conv_geom* g, const float* x, const float* w, const float* bias, float* y_raw, float eps,
                           int act, const float* residual, float* y, float* mean, float* rstd, void* workspace,
                           size_t workspace_bytes, void* stream, const mg_wino_tiles* wt, void* y16) {
    return conv_fwd_instnorm_impl(g, x, w, bias, y_raw, eps, act, residual, y, mean, rstd, workspace, workspace_bytes, stream, wt, y16,
                                  nullptr, 0);
}

static int conv_fwd_instnorm_impl(const mg_conv_geom* g, const float* x, const float* w, const float* bias, float* y_raw, float eps,
                                  int act, const float* residual, float* y, float* mean, float* rstd, void* workspace,
                                  size_t workspace_bytes, void* stream, const mg_wino_tiles* wt, void* y16, float* v_next,
                                  int next_reflect) {
    if (y16 && !prec_h(g)) return MG_ERR_UNSUPPORTED;          // the float32 layers' fused inverse transform + norm has no float16 output
    if (!geom_ok(g) || !x || !w || !y || !mean || !rstd) return MG_ERR_ARG;
    if (!wino_tiles_ok(g, wt)) return MG_ERR_ARG;
    if (!workspace || workspace_bytes < mg_conv_fwd_instnorm_workspace(g)) return MG_ERR_ARG;
    // y_raw == NULL (inference: no backward pass will read the raw convolution output): the fused kernel skips that store -- a
    // fifth of its HBM bytes at batch 64 -- and the two-call path normalises in place
    if (wino_ok(g) && !prec_h(g) && !mg_conv_rowdot_kq(g) && wino_out_norm_ok(g->OH / 2, g->OW / 2, g->Co) && aligned16(x) &&
        aligned16(w) && (!y_raw || aligned16(y_raw)) && aligned16(y) && aligned16(mean) && aligned16(rstd) && aligned16(workspace) &&
        (!bias || aligned16(bias)) && (!residual || aligned16(residual))) {
        const WinoNorm nrm{eps, act, residual, y, mean, rstd, v_next, next_reflect};
        return wino_fwd(g, x, w, bias, y_raw, MG_ACT_NONE, (float*)workspace, (hipStream_t)stream, wt ? wt->u : nullptr,
                        wt ? wt->v : nullptr, &nrm, wt && wt->v && (wt->flags & MG_TILES_V_FILLED));
    }
    if (v_next) return MG_ERR_ARG;           // (unreachable after wino_vnext_ok: the fused path above is the only writer of v_next)
    float* const raw_out = y_raw;            // NULL under no_grad: the slab kernel then skips that store, the two-launch path normalises in place
    if (!y_raw) y_raw = y;
    // small maps: the one-launch InstanceNorm kernel also finishes a split-K convolution (sums the slabs, bias, autocast rounding)
    FwdDefer defer{nullptr, 0, nullptr, 0, false};
    const bool no_defer = getenv("MG_NO_FWD_DEFER") != nullptr;      // (read per call: the bit-identity test flips it)
    const bool can_defer = !no_defer && mg_instnorm_slab_ok(g->OH * g->OW, g->Co) && g->Co % 4 == 0 && (!bias || aligned16(bias));
    g_fwd_defer = can_defer ? &defer : nullptr;
    const int rc = mg_conv_fwd_w(g, x, w, bias, y_raw, MG_ACT_NONE, workspace, workspace_bytes, stream, wt);
    g_fwd_defer = nullptr;
    if (rc != MG_OK) return rc;
    if (defer.filled)
        return mg_instnorm_fwd_slabs(defer.part, defer.splits, defer.bias, defer.round_f16, raw_out, g->B, g->OH * g->OW, g->Co, eps, act,
                                     residual, y, mean, rstd, stream, y16);
    return mg_instnorm_fwd_h(y_raw, g->B, g->OH * g->OW, g->Co, eps, act, residual, y, mean, rstd, workspace, workspace_bytes, stream, y16);
}

int mg_conv_fwd_w(const mg_conv_geom* g, const float* x, const float* w, const float* bias, float* y, int act,
                  void* workspace, size_t workspace_bytes, void* stream, const mg_wino_tiles* wt) {
    if (!geom_ok(g) || !x || !w || !y) return MG_ERR_ARG;
    if (!wino_tiles_ok(g, wt)) return MG_ERR_ARG;
    const float* u = wt ? wt->u : nullptr;
    if (h16_ok(g) && workspace && workspace_bytes >= h16_fwd_ws(g) && aligned16(x) && aligned16(w) && aligned16(y) &&
        aligned16(workspace) && (!bias || aligned16(bias)))
        return h16_fwd(g, x, w, bias, y, act, (char*)workspace, (hipStream_t)stream, u,
                       (wt && wt->v && (wt->flags & MG_TILES_V_FILLED)) ? (const void*)wt->v : nullptr);
    if (co1_gemm_ok(g) && workspace && workspace_bytes >= co1_fwd_ws(g) && aligned16(x) && aligned16(w) && aligned16(workspace))
        return co1_fwd(g, x, w, bias, y, act, (char*)workspace, (hipStream_t)stream, u, wt ? wt->v : nullptr);
    if (mg_conv_rowdot_kq(g) && aligned16(x) && aligned16(w)) {
        probe_begin((hipStream_t)stream);
        const int rc = mg_conv_rowdot_fwd(g, x, w, bias, y, act, stream);
        probe_end((hipStream_t)stream);
        return rc;
    }
    if (wino_ok(g) && workspace && workspace_bytes >= wino_fwd_ws(g) && aligned16(x) && aligned16(w) && aligned16(y) &&
        aligned16(workspace) && (!bias || aligned16(bias)))
        return wino_fwd(g, x, w, bias, y, act, (float*)workspace, (hipStream_t)stream, u, wt ? wt->v : nullptr, nullptr,
                        wt && wt->v && !prec_h(g) && (wt->flags & MG_TILES_V_FILLED));
    if (wino4_ok(g) && workspace && workspace_bytes >= wino4_fwd_ws(g) && aligned16(x) && aligned16(w) && aligned16(y) &&
        aligned16(workspace) && (!bias || aligned16(bias)))
        return wino4_fwd(g, x, w, bias, y, act, (float*)workspace, (hipStream_t)stream, u, wt ? wt->v : nullptr);
    if (wino42_ok(g) && workspace && workspace_bytes >= wino42_fwd_ws(g) && aligned16(x) && aligned16(w) && aligned16(y) &&
        aligned16(workspace) && (!bias || aligned16(bias)))
        return wino42_fwd(g, x, w, bias, y, act, (float*)workspace, (hipStream_t)stream, u, wt ? wt->v : nullptr);
    if (smallc_fwd_kind(g)) {
        probe_begin((hipStream_t)stream);
        const int rc = smallc_fwd(g, x, w, bias, y, act, (hipStream_t)stream);
        probe_end((hipStream_t)stream);
        return rc;
    }
    const Geom gg = to_geom(g);
    hipStream_t st = (hipStream_t)stream;
    const long long M = (long long)g->B * g->OH * g->OW;
    const int N = g->Co;
    const bool cd_half = conv_dma_half(g);
    if (conv_dma_fwd_ok(g) && aligned16(x) && aligned16(w) && aligned16(y) && (!bias || aligned16(bias)) &&
        (!cd_half || (workspace && aligned16(workspace) && workspace_bytes >= mg_conv_fwd_workspace(g)))) {
        CdPlan cp = conv_dma_fwd_plan(g);
        if (cp.splits > 1 && (!workspace || workspace_bytes < mg_conv_fwd_workspace(g) || !aligned16(workspace))) {
            cp.splits = 1;
            cp.cps = 1 << 28;
        }
        const void* xin = x;
        const void* win = w;
        char* wsp = (char*)workspace;
        if (cd_half) {          // float16 copies: the activation by a cast pass, the weights from the cache when there is one
            void* x16 = (wt && wt->v) ? (void*)wt->v : (void*)wsp;         // kept by the caller for the weight gradient
            if (!(wt && wt->v && (wt->flags & MG_TILES_V_FILLED)))         // ... unless x's producer already wrote it (mg_instnorm_fwd_h)
                cd_cast16(x, x16, (size_t)g->B * g->H * g->W * g->Ci, st);
            xin = x16;
            wsp += conv_dma_h_x_bytes(g);
            if (u) {
                win = u;
            } else {
                cd_cast16(w, wsp, (size_t)g->Co * g->KH * g->KW * g->Ci, st);
                win = wsp;
            }
            wsp += conv_dma_h_w_bytes(g);
            MG_CHECK_LAUNCH();
        }
        probe_begin(st);
        conv_dma_fwd_launch(g, cp, xin, win, bias, y, act, (float*)wsp, st);
        probe_end(st);
        MG_CHECK_LAUNCH();
        if (cp.splits > 1) {
            const size_t n = (size_t)M * N;
            if (g_fwd_defer && act == MG_ACT_NONE) {
                *g_fwd_defer = FwdDefer{(const float*)wsp, cp.splits, bias, cd_half ? 1 : 0, true};
                return MG_OK;
            }
            const unsigned blocks = (unsigned)((n / 4 + 255) / 256 > 4096 ? 4096 : (n / 4 + 255) / 256);
            hipLaunchKernelGGL(splitk_epilogue_kernel, dim3(blocks), dim3(256), 0, st, (const float*)wsp, cp.splits, n, N,
                               bias, act, y, cd_half ? 1 : 0);
            MG_CHECK_LAUNCH();
        }
        return MG_OK;
    }
    const bool vec = (g->Ci % BK == 0) && aligned16(x) && aligned16(w);
    TilePlan tp = fwd_plan(g);
    if (tp.splits > 1 && (!workspace || workspace_bytes < mg_conv_fwd_workspace(g) || !aligned16(y) ||
                          (bias && !aligned16(bias)))) {
        tp.splits = 1;
        tp.cps = 1 << 30;
    }
    float* part = tp.splits > 1 ? (float*)workspace : nullptr;
    const bool hp = prec_h(g);
    // f16: the 32-deep kernel whenever the shape allows it (its stage / fragment traffic is what the f16 loop is made of)
    const bool k32 = vec && (hp ? (fwd32_enabled() && g->Ci % BK2 == 0 && (tp.splits == 1 || tp.cps % 2 == 0))
                                : use_k32(tp, g->Ci));
#define MG_LAUNCH_FWD(BM_, BN_)                                                                                    \
    do {                                                                                                           \
        dim3 grid((unsigned)(((M + BM_ - 1) / BM_) * ((N + BN_ - 1) / BN_)), tp.splits);                           \
        if (k32 && hp) launch_fwd32<BM_, BN_, 2>(grid, st, gg, x, w, bias, y, act, tp.splits == 1 ? (1 << 29) : tp.cps / 2, part, Batch{0, 0, 0, 0}); \
        else if (k32) launch_fwd32<BM_, BN_, 0>(grid, st, gg, x, w, bias, y, act, tp.splits == 1 ? (1 << 29) : tp.cps / 2, part, Batch{0, 0, 0, 0}); \
        else if (hp && vec) hipLaunchKernelGGL((conv_fwd_kernel<BM_, BN_, true, 2>), grid, dim3(256), 0, st, gg, x, w, bias, y, act, tp.cps, part, Batch{0, 0, 0, 0});  \
        else if (hp) hipLaunchKernelGGL((conv_fwd_kernel<BM_, BN_, false, 2>), grid, dim3(256), 0, st, gg, x, w, bias, y, act, tp.cps, part, Batch{0, 0, 0, 0});    \
        else if (vec) hipLaunchKernelGGL((conv_fwd_kernel<BM_, BN_, true>), grid, dim3(256), 0, st, gg, x, w, bias, y, act, tp.cps, part, Batch{0, 0, 0, 0});  \
        else hipLaunchKernelGGL((conv_fwd_kernel<BM_, BN_, false>), grid, dim3(256), 0, st, gg, x, w, bias, y, act, tp.cps, part, Batch{0, 0, 0, 0});    \
    } while (0)
    probe_begin(st);
    if (tp.bm == 128 && tp.bn == 128) MG_LAUNCH_FWD(128, 128);
    else if (tp.bm == 64) MG_LAUNCH_FWD(64, 64);
    else MG_LAUNCH_FWD(128, 64);
    probe_end(st);
#undef MG_LAUNCH_FWD
    MG_CHECK_LAUNCH();
    if (part) {
        const size_t n = (size_t)M * N;
        const unsigned blocks = (unsigned)((n / 4 + 255) / 256 > 4096 ? 4096 : (n / 4 + 255) / 256);
        hipLaunchKernelGGL(splitk_epilogue_kernel, dim3(blocks), dim3(256), 0, st, (const float*)part, tp.splits, n, N,
                           bias, act, y, (int)hp);
        MG_CHECK_LAUNCH();
    }
    return MG_OK;
}

static int dgrad_dispatch(const mg_conv_geom* g, const float* dy, const float* w, const float* bias, float* dx, int act,
                          void* workspace, size_t workspace_bytes, void* stream, const mg_wino_tiles* wt);
int mg_conv_dgrad_w(const mg_conv_geom* g, const float* dy, const float* w, const float* bias, float* dx, int act,
                    void* workspace, size_t workspace_bytes, void* stream, const mg_wino_tiles* wt) {
    if (wt && wt->add && !aligned16(wt->add)) return MG_ERR_ARG;
    g_dgrad_add = wt ? wt->add : nullptr;
    int rc = dgrad_dispatch(g, dy, w, bias, dx, act, workspace, workspace_bytes, stream, wt);
    const float* left = g_dgrad_add;
    g_dgrad_add = nullptr;
    if (rc == MG_OK && left) rc = mg_add(dx, left, dx, (long long)g->B * g->H * g->W * g->Ci, stream);
    return rc;
}
static int dgrad_dispatch(const mg_conv_geom* g, const float* dy, const float* w, const float* bias, float* dx, int act,
                          void* workspace, size_t workspace_bytes, void* stream, const mg_wino_tiles* wt) {
    if (!geom_ok(g) || !w || !dx) return MG_ERR_ARG;
    if (!wino_tiles_ok(g, wt)) return MG_ERR_ARG;
    // dy == NULL: wt->md already holds A dy A^T (mg_instnorm_bwd_wino_md) -- F(2x2,3x3) layers in the transposed formulation
    if (!dy && !(wt && wt->md && mg_conv_wino_md_from_norm_ok(g) && !bias && act == MG_ACT_NONE && workspace &&
                 workspace_bytes >= wino_dgrad_ws(g) && aligned16(w) && aligned16(dx) && aligned16(workspace)))
        return MG_ERR_ARG;
    const float* u = wt ? wt->u : nullptr;
    if (g->reflect && g->stride != 1) return MG_ERR_UNSUPPORTED;
    if (h16_ok(g) && !bias && act == MG_ACT_NONE && workspace && workspace_bytes >= h16_dgrad_ws(g) && aligned16(dy) &&
        aligned16(w) && aligned16(dx) && aligned16(workspace))
        return h16_dgrad(g, dy, w, dx, (char*)workspace, (hipStream_t)stream, u, &g_dgrad_add);
    if (co1_gemm_ok(g) && !bias && act == MG_ACT_NONE && workspace && workspace_bytes >= co1_dgrad_ws(g) && aligned16(w) &&
        aligned16(dx) && aligned16(workspace))
        return co1_dgrad(g, dy, w, dx, (char*)workspace, (hipStream_t)stream, u, wt ? wt->md : nullptr);
    if (smallc_dgrad_ok(g) && !bias && act == MG_ACT_NONE && aligned16(dy)) {
        probe_begin((hipStream_t)stream);
        const int rc = smallc_dgrad(g, dy, w, dx, (hipStream_t)stream);
        probe_end((hipStream_t)stream);
        return rc;
    }
    if (wino_ok(g) && workspace &&
        workspace_bytes >= wino_dgrad_ws(g) &&
        aligned16(dy) && aligned16(w) && aligned16(dx) && aligned16(workspace) && (!bias || aligned16(bias)))
        return wino_dgrad(g, dy, w, bias, dx, act, (float*)workspace, (hipStream_t)stream, u, wt ? wt->md : nullptr);
    if (wino4_ok(g) && !bias && act == MG_ACT_NONE && workspace && workspace_bytes >= wino4_dgrad_ws(g) && aligned16(dy) &&
        aligned16(w) && aligned16(dx) && aligned16(workspace))
        return wino4_dgrad(g, dy, w, dx, (float*)workspace, (hipStream_t)stream, u, wt ? wt->md : nullptr);
    if (wino42_ok(g) && !bias && act == MG_ACT_NONE && workspace && workspace_bytes >= wino42_dgrad_ws(g) && aligned16(dy) &&
        aligned16(w) && aligned16(dx) && aligned16(workspace))
        return wino42_dgrad(g, dy, w, dx, (float*)workspace, (hipStream_t)stream, u, wt ? wt->md : nullptr);
    const Geom gg = to_geom(g);
    hipStream_t st = (hipStream_t)stream;
    const int s = g->stride;
    if (aligned16(dy) && aligned16(w) && aligned16(dx) && (!bias || aligned16(bias)) && workspace && aligned16(workspace) &&
        workspace_bytes >= mg_conv_dgrad_workspace(g)) {
        float* md = wt ? wt->md : nullptr;
        if (conv_dma_dgrad_ok(g))
            return cd_dgrad_run(g, dy, w, bias, dx, act, (char*)workspace, st, u, md, conv_dma_half(g) ? 1 : 0,
                                (wt && (wt->flags & MG_TILES_MD_FILLED)) ? 1 : 0);
        mg_conv_geom gp;
        if (!bias && act == MG_ACT_NONE && cd_reflect_dgrad_geom(g, &gp)) {
            float* dxp = (float*)workspace;
            const int rc = cd_dgrad_run(&gp, dy, w, nullptr, dxp, MG_ACT_NONE, (char*)workspace + cd_reflect_dxp_bytes(&gp), st, u,
                                        md, 0, (wt && (wt->flags & MG_TILES_MD_FILLED)) ? 1 : 0);
            if (rc != MG_OK) return rc;
            const float* addend = g_dgrad_add;      // the skip connection's gradient rides in the fold (mg_wino_tiles.add)
            g_dgrad_add = nullptr;
            hipLaunchKernelGGL(wino_fold_reflect_kernel, dim3(wino_grid((size_t)g->B * g->H * g->W * g->Ci / 4)), dim3(256), 0,
                               st, (const float*)dxp, g->B, g->H, g->W, g->Ci, dx, (int)prec_h(g), addend);
            MG_CHECK_LAUNCH();
            return MG_OK;
        }
    }
    // every input pixel is produced by exactly one class launch; classes cover all of [0,H)x[0,W)
    const long long Mc = (long long)g->B * ((g->H + s - 1) / s) * ((g->W + s - 1) / s);   // largest class
    const int N = g->Ci;
    const bool veca = (g->Co % BK == 0) && aligned16(dy);
    const bool vecb = (g->Ci % 4 == 0) && aligned16(w);
    TilePlan tp = dgrad_plan(g);
    if (tp.splits > 1 && (!workspace || workspace_bytes < mg_conv_dgrad_workspace(g) || !aligned16(dx) ||
                          (bias && !aligned16(bias)))) {
        tp.splits = 1;
        tp.cps = 1 << 30;
    }
    float* part = tp.splits > 1 ? (float*)workspace : nullptr;
    const bool hp = prec_h(g);
#define MG_LAUNCH_DGRAD_T(BM_, BN_, TAG_)                                                                          \
    do {                                                                                                           \
        dim3 grid((unsigned)(((Mc + BM_ - 1) / BM_) * ((N + BN_ - 1) / BN_)), tp.splits, s * s);                   \
        if (veca && vecb)                                                                                          \
            hipLaunchKernelGGL((conv_dgrad_kernel<BM_, BN_, true, true, TAG_>), grid, dim3(256), 0, st, gg, dy, w, bias, dx, act, tp.cps, part, Batch{0, 0, 0, 0});  \
        else if (veca)                                                                                             \
            hipLaunchKernelGGL((conv_dgrad_kernel<BM_, BN_, true, false, TAG_>), grid, dim3(256), 0, st, gg, dy, w, bias, dx, act, tp.cps, part, Batch{0, 0, 0, 0}); \
        else if (vecb)                                                                                             \
            hipLaunchKernelGGL((conv_dgrad_kernel<BM_, BN_, false, true, TAG_>), grid, dim3(256), 0, st, gg, dy, w, bias, dx, act, tp.cps, part, Batch{0, 0, 0, 0}); \
        else                                                                                                       \
            hipLaunchKernelGGL((conv_dgrad_kernel<BM_, BN_, false, false, TAG_>), grid, dim3(256), 0, st, gg, dy, w, bias, dx, act, tp.cps, part, Batch{0, 0, 0, 0});\
    } while (0)
#define MG_LAUNCH_DGRAD(BM_, BN_) do { if (hp) MG_LAUNCH_DGRAD_T(BM_, BN_, 2); else MG_LAUNCH_DGRAD_T(BM_, BN_, 0); } while (0)
    probe_begin(st);
    if (tp.bm == 128 && tp.bn == 128) MG_LAUNCH_DGRAD(128, 128);
    else if (tp.bm == 64) MG_LAUNCH_DGRAD(64, 64);
    else MG_LAUNCH_DGRAD(128, 64);
    probe_end(st);
#undef MG_LAUNCH_DGRAD_T
#undef MG_LAUNCH_DGRAD
    MG_CHECK_LAUNCH();
    if (part) {
        const size_t n = (size_t)g->B * g->H * g->W * N;
        const unsigned blocks = (unsigned)((n / 4 + 255) / 256 > 4096 ? 4096 : (n / 4 + 255) / 256);
        hipLaunchKernelGGL(splitk_epilogue_kernel, dim3(blocks), dim3(256), 0, st, (const float*)part, tp.splits, n, N,
                           bias, act, dx, (int)hp);
        MG_CHECK_LAUNCH();
    }
    return MG_OK;
}

size_t mg_colsum_workspace(long long M, int C) {
    const ColsumPlan p = colsum_plan(M, C);
    return ((size_t)p.splits + 1) * C * sizeof(float);
}

int mg_colsum(const float* a, long long M, int C, float* out, int accumulate, void* workspace,
              size_t workspace_bytes, void* stream) {
    if (!a || !out || M <= 0 || C <= 0 || !workspace) return MG_ERR_ARG;
    if (workspace_bytes < mg_colsum_workspace(M, C)) return MG_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    const ColsumPlan p = colsum_plan(M, C);
    float* part = (float*)workspace;                 // [splits][C]
    if (p.splits == 1) {
        hipLaunchKernelGGL(colsum_partial_kernel, dim3((C + 63) / 64, 1), dim3(256), 0, st, a, M, C, M, out, 1, accumulate);
    } else {
        hipLaunchKernelGGL(colsum_partial_kernel, dim3((C + 63) / 64, p.splits), dim3(256), 0, st, a, M, C,
                           p.rows_per_split, part, 0, 0);
        hipLaunchKernelGGL(colsum_partial_kernel, dim3((C + 63) / 64, 1), dim3(256), 0, st, (const float*)part,
                           (long long)p.splits, C, (long long)p.splits, out, 1, accumulate);
    }
    MG_CHECK_LAUNCH();
    return MG_OK;
}

size_t mg_conv_wgrad_workspace(const mg_conv_geom* g) {
    if (!geom_ok(g)) return 0;
    if (co1_gemm_ok(g)) return co1_wgrad_ws(g);
    if (mg_conv_rowdot_kq(g)) return mg_conv_rowdot_wgrad_workspace(g);
    if (h16_ok(g)) return h16_wgrad_ws(g);
    if (wino_ok(g)) return wino_wgrad_ws(g);
    if (wino4_ok(g)) return wino4_wgrad_ws(g);
    if (wino42_ok(g)) return wino42_wgrad_ws(g);
    if (smallc_wgrad_kind(g)) return smallc_wgrad_ws(g);
    const WgradPlan p = wgrad_plan(g);
    int sp = p.splits;
    if (conv_dma_wgrad_ok(g) && conv_dma_wgrad_plan(g).splits > sp) sp = conv_dma_wgrad_plan(g).splits;
    size_t wg = sp > 1 ? (size_t)sp * g->Co * g->KH * g->KW * g->Ci * sizeof(float) : 0;
    if (conv_dma_wgrad_ok(g) && conv_dma_half(g)) wg += conv_dma_h_x_bytes(g) + conv_dma_h_dy_bytes(g);
    const size_t cs = mg_colsum_workspace((long long)g->B * g->OH * g->OW, g->Co);
    return (wg > cs ? wg : cs) + 256;
}

int mg_conv_wgrad_adam_ok(const mg_conv_geom* g) {
    return (geom_ok(g) && !co1_gemm_ok(g) && !mg_conv_rowdot_kq(g) && !h16_ok(g) && wino_ok(g) && !prec_h(g) && g->Ci % 4 == 0) ? 1 : 0;
}
int mg_conv_wgrad_adam_w(const mg_conv_geom* g, const float* x, const float* dy, float* w, const mg_wino_adam* ad,
                         void* workspace, size_t workspace_bytes, void* stream, const mg_wino_tiles* wt) {
    if (!mg_conv_wgrad_adam_ok(g)) return MG_ERR_UNSUPPORTED;
    if (!w || !ad || !ad->m || !ad->v || !ad->u || !ad->state || !wino_tiles_ok(g, wt)) return MG_ERR_ARG;
    if ((!x || !dy) && !(wt && wt->v && wt->md && mg_conv_wino_md_from_norm_ok(g))) return MG_ERR_ARG;
    if (!workspace || workspace_bytes < mg_conv_wgrad_workspace(g)) return MG_ERR_ARG;
    if (!aligned16(w) || !aligned16(ad->m) || !aligned16(ad->v) || !aligned16(ad->u) || !aligned16(workspace) ||
        (x && !aligned16(x)) || (dy && !aligned16(dy)))
        return MG_ERR_ARG;
    return wino_wgrad(g, x, dy, w, 0, (float*)workspace, (hipStream_t)stream, wt ? wt->v : nullptr, wt ? wt->md : nullptr, ad);
}

int mg_conv_wgrad_w(const mg_conv_geom* g, const float* x, const float* dy, float* dw, float* dbias, int accumulate,
                    void* workspace, size_t workspace_bytes, void* stream, const mg_wino_tiles* wt) {
    return mg_conv_wgrad_chk(g, x, dy, dw, dbias, accumulate, workspace, workspace_bytes, stream, wt, nullptr);
}
int mg_conv_wgrad_checks_finite(const mg_conv_geom* g) {
    return (geom_ok(g) && !co1_gemm_ok(g) && !mg_conv_rowdot_kq(g) && h16_ok(g) && h16_wgrad_as(g)) ? 1 : 0;
}
int mg_conv_wgrad_h16_ok(const mg_conv_geom* g) { return mg_conv_wgrad_checks_finite(g); }
int mg_conv_wgrad_h16(const mg_conv_geom* g, const float* x, const float* dy, void* dw16, int accumulate, void* workspace,
                      size_t workspace_bytes, void* stream, float* found_inf) {
    if (!geom_ok(g) || !dw16 || !x || !dy) return MG_ERR_ARG;
    if (!mg_conv_wgrad_h16_ok(g)) return MG_ERR_UNSUPPORTED;
    if (!workspace || workspace_bytes < mg_conv_wgrad_workspace(g)) return MG_ERR_ARG;
    if (!aligned16(x) || !aligned16(dy) || !aligned16(dw16) || !aligned16(workspace)) return MG_ERR_ARG;
    return h16_wgrad(g, x, dy, nullptr, accumulate, (char*)workspace, (hipStream_t)stream, found_inf, dw16);
}
int mg_conv_wgrad_chk(const mg_conv_geom* g, const float* x, const float* dy, float* dw, float* dbias, int accumulate,
                      void* workspace, size_t workspace_bytes, void* stream, const mg_wino_tiles* wt, float* found_inf) {
    if (!geom_ok(g) || !dw) return MG_ERR_ARG;
    if (found_inf && !mg_conv_wgrad_checks_finite(g)) return MG_ERR_UNSUPPORTED;   // nobody would look at dw: refuse instead of skipping silently
    if (found_inf && !(x && dy && aligned16(x) && aligned16(dy) && aligned16(dw) && aligned16(workspace))) return MG_ERR_ARG;
    if (!wino_tiles_ok(g, wt)) return MG_ERR_ARG;
    // x / dy may be NULL when the caller hands over both Winograd images of an F(2x2,3x3) layer (no bias gradient then)
    if ((!x || !dy) && !(wt && wt->v && wt->md && mg_conv_wino_md_from_norm_ok(g) && !dbias && aligned16(dw) && aligned16(workspace)))
        return MG_ERR_ARG;
    if (workspace_bytes < mg_conv_wgrad_workspace(g) || !workspace) return MG_ERR_ARG;
    if (co1_gemm_ok(g) && aligned16(x) && aligned16(workspace))
        return co1_wgrad(g, x, dy, dw, dbias, accumulate, (char*)workspace, (hipStream_t)stream, wt ? wt->v : nullptr, wt ? wt->md : nullptr);
    if (mg_conv_rowdot_kq(g) && aligned16(x) && aligned16(workspace)) {
        probe_begin((hipStream_t)stream);
        const int rc = mg_conv_rowdot_wgrad(g, x, dy, dw, dbias, accumulate, workspace, workspace_bytes, stream);
        probe_end((hipStream_t)stream);
        return rc;
    }
    if (h16_ok(g) && aligned16(x) && aligned16(dy) && aligned16(dw) && aligned16(workspace)) {
        const int rc = h16_wgrad(g, x, dy, dw, accumulate, (char*)workspace, (hipStream_t)stream, found_inf);
        if (rc != MG_OK) return rc;
        if (dbias)
            return mg_colsum(dy, (long long)g->B * g->OH * g->OW, g->Co, dbias, accumulate, (char*)workspace + h16_wgrad_cs_offset(g),
                             mg_colsum_workspace((long long)g->B * g->OH * g->OW, g->Co), stream);
        return MG_OK;
    }
    if (wino_ok(g) && aligned16(x) && aligned16(dy) && aligned16(dw) && aligned16(workspace)) {
        const int rc = wino_wgrad(g, x, dy, dw, accumulate, (float*)workspace, (hipStream_t)stream, wt ? wt->v : nullptr,
                                  wt ? wt->md : nullptr);
        if (rc != MG_OK) return rc;
        if (dbias) {
            const WinoDims d = wino_dims(g);
            float* cs = (float*)workspace + al256((size_t)16 * d.T * g->Ci) + al256((size_t)16 * d.T * g->Co) +
                        al256((size_t)16 * g->Co * g->Ci) +
                        al256(wino_wgrad_slabs(g) * 16 * g->Co * g->Ci);
            return mg_colsum(dy, (long long)g->B * g->OH * g->OW, g->Co, dbias, accumulate, cs,
                             mg_colsum_workspace((long long)g->B * g->OH * g->OW, g->Co), stream);
        }
        return MG_OK;
    }
    if (smallc_wgrad_kind(g) && aligned16(dw) && aligned16(workspace)) {
        probe_begin((hipStream_t)stream);
        const int rc = smallc_wgrad(g, x, dy, dw, accumulate, (float*)workspace, (hipStream_t)stream);
        probe_end((hipStream_t)stream);
        if (rc != MG_OK) return rc;
        if (dbias) {
            float* cs = (float*)workspace + smallc_wgrad_layout(g).cs;
            return mg_colsum(dy, (long long)g->B * g->OH * g->OW, g->Co, dbias, accumulate, cs,
                             mg_colsum_workspace((long long)g->B * g->OH * g->OW, g->Co), stream);
        }
        return MG_OK;
    }
    if (wino4_ok(g) && aligned16(x) && aligned16(dy) && aligned16(dw) && aligned16(workspace)) {
        const int rc = wino4_wgrad(g, x, dy, dw, accumulate, (float*)workspace, (hipStream_t)stream, wt ? wt->v : nullptr,
                                   wt ? wt->md : nullptr);
        if (rc != MG_OK) return rc;
        if (dbias) {
            const Wino4Dims d = wino4_dims(g);
            const WinoWgradPlan p4 = wino4_wgrad_plan(g);
            float* cs = (float*)workspace + al256((size_t)25 * d.T * g->Ci) + al256((size_t)25 * d.T * g->Co) +
                        al256((size_t)25 * g->Co * g->Ci) + al256(wino4_wgrad_slabs(g) * 25 * g->Co * g->Ci);
            return mg_colsum(dy, (long long)g->B * g->OH * g->OW, g->Co, dbias, accumulate, cs,
                             mg_colsum_workspace((long long)g->B * g->OH * g->OW, g->Co), stream);
        }
        return MG_OK;
    }
    if (wino42_ok(g) && aligned16(x) && aligned16(dy) && aligned16(dw) && aligned16(workspace)) {
        const int rc = wino42_wgrad(g, x, dy, dw, accumulate, (float*)workspace, (hipStream_t)stream, wt ? wt->v : nullptr,
                                    wt ? wt->md : nullptr);
        if (rc != MG_OK) return rc;
        if (dbias) {
            const Wino42Dims d = wino42_dims(g);
            const WinoWgradPlan p4 = wino42_wgrad_plan(g);
            float* cs = (float*)workspace + al256((size_t)25 * d.T * d.K4) + al256((size_t)25 * d.T * g->Co) +
                        al256((size_t)25 * g->Co * d.K4) + al256(wino42_wgrad_slabs(g) * 25 * g->Co * d.K4);
            return mg_colsum(dy, (long long)g->B * g->OH * g->OW, g->Co, dbias, accumulate, cs,
                             mg_colsum_workspace((long long)g->B * g->OH * g->OW, g->Co), stream);
        }
        return MG_OK;
    }
    const Geom gg = to_geom(g);
    hipStream_t st = (hipStream_t)stream;
    if (conv_dma_wgrad_ok(g) && aligned16(x) && aligned16(dy) && aligned16(dw) && aligned16(workspace)) {
        const CdPlan cp = conv_dma_wgrad_plan(g);
        const size_t n_out = (size_t)g->Co * g->KH * g->KW * g->Ci;
        const void* xin = x;
        const void* dyin = dy;
        char* wsp = (char*)workspace;
        if (conv_dma_half(g)) {          // float16 copies: the caller's (made by the forward / data-gradient call) or cast here
            if (wt && wt->v) {
                xin = wt->v;
            } else {
                cd_cast16(x, wsp, (size_t)g->B * g->H * g->W * g->Ci, st);
                xin = wsp;
            }
            wsp += conv_dma_h_x_bytes(g);
            if (wt && wt->md) {
                dyin = wt->md;
            } else {
                cd_cast16(dy, wsp, (size_t)g->B * g->OH * g->OW * g->Co, st);
                dyin = wsp;
            }
            wsp += conv_dma_h_dy_bytes(g);
            MG_CHECK_LAUNCH();
        }
        probe_begin(st);
        conv_dma_wgrad_launch(g, cp, xin, dyin, dw, accumulate, (float*)wsp, st);
        probe_end(st);
        MG_CHECK_LAUNCH();
        if (cp.splits > 1) {
            const unsigned blocks = (unsigned)((n_out / 4 + 255) / 256 > 4096 ? 4096 : (n_out / 4 + 255) / 256 + 1);
            hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, st, (const float*)wsp, cp.splits, n_out, dw,
                               accumulate);
            MG_CHECK_LAUNCH();
        }
        if (dbias)
            return mg_colsum(dy, (long long)g->B * g->OH * g->OW, g->Co, dbias, accumulate, workspace, workspace_bytes, stream);
        return MG_OK;
    }
    const WgradPlan p = wgrad_plan(g);
    const bool veca = (g->Co % 4 == 0) && aligned16(dy);
    const bool vecb = (g->Ci % 4 == 0) && aligned16(x);
    const size_t n_out = (size_t)g->Co * g->KH * g->KW * g->Ci;
    float* target = p.splits > 1 ? (float*)workspace : dw;
    const int acc_direct = (p.splits > 1) ? 0 : accumulate;
    const bool hp = prec_h(g);
#define MG_LAUNCH_WGRAD_T(BM_, BN_, TAG_)                                                                           \
    do {                                                                                                            \
        dim3 grid((unsigned)p.tiles, 1, p.splits);                                                                  \
        if (veca && vecb)                                                                                           \
            hipLaunchKernelGGL((conv_wgrad_kernel<BM_, BN_, true, true, TAG_>), grid, dim3(256), 0, st, gg, x, dy, target, p.cps, acc_direct, Batch{0, 0, 0, 0});  \
        else if (veca)                                                                                              \
            hipLaunchKernelGGL((conv_wgrad_kernel<BM_, BN_, true, false, TAG_>), grid, dim3(256), 0, st, gg, x, dy, target, p.cps, acc_direct, Batch{0, 0, 0, 0}); \
        else if (vecb)                                                                                              \
            hipLaunchKernelGGL((conv_wgrad_kernel<BM_, BN_, false, true, TAG_>), grid, dim3(256), 0, st, gg, x, dy, target, p.cps, acc_direct, Batch{0, 0, 0, 0}); \
        else                                                                                                        \
            hipLaunchKernelGGL((conv_wgrad_kernel<BM_, BN_, false, false, TAG_>), grid, dim3(256), 0, st, gg, x, dy, target, p.cps, acc_direct, Batch{0, 0, 0, 0});\
    } while (0)
#define MG_LAUNCH_WGRAD(BM_, BN_) do { if (hp) MG_LAUNCH_WGRAD_T(BM_, BN_, 2); else MG_LAUNCH_WGRAD_T(BM_, BN_, 0); } while (0)
    probe_begin(st);
    if (p.big) MG_LAUNCH_WGRAD(128, 128);
    else MG_LAUNCH_WGRAD(64, 64);
    probe_end(st);
#undef MG_LAUNCH_WGRAD_T
#undef MG_LAUNCH_WGRAD
    MG_CHECK_LAUNCH();
    if (p.splits > 1) {
        const unsigned blocks = (unsigned)((n_out / 4 + 255) / 256 > 4096 ? 4096 : (n_out / 4 + 255) / 256 + 1);
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, st, (const float*)workspace, p.splits,
                           n_out, dw, accumulate);
        MG_CHECK_LAUNCH();
    }
    if (dbias) {
        const int rc = mg_colsum(dy, (long long)g->B * g->OH * g->OW, g->Co, dbias, accumulate, workspace,
                                 workspace_bytes, stream);
        if (rc != MG_OK) return rc;
    }
    return MG_OK;
}

}  // extern "C"
