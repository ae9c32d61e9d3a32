// Convolutions with a single output channel (the generator's 7x7 tanh head, models/networks.py:243-244, 351-352,
// and the PatchGAN output layer, :669-670): y[m] = act(b + sum_k A[m][k] * w[k]).  An N = 1 GEMM wastes 63/64 of
// an MFMA tile, so these run as wave-per-pixel dot products on the VALU instead: the 64 lanes split K = KH*KW*Ci
// into float4 slices (channels of 4 consecutive taps are contiguous in NHWC, so every wave load is 1 KiB of
// 256-B runs), weights stay in registers for the whole kernel, activations come from L1/L2 (each input pixel is
// re-used by KH*KW outputs), lanes combine with a DPP butterfly.  HBM-bound in principle (input read once);
// in practice bounded by the L1 request rate: ~KH*KW*Ci*4 B per output pixel.
//
//   rowdot_fwd   : forward (bias + activation fused)
//   rowdot_wgrad : dw[k] = sum_m dy[m] * A[m][k], dbias = sum_m dy[m]; per-wave partial rows reduced by mg_colsum
// The data gradient of these layers has N = Ci >= 64 and stays on the implicit-GEMM kernel.
#include <cstdlib>
#include "common.h"
#include "mdctgan_hip.h"

namespace {

struct Geom {
    int B, H, W, Ci, OH, OW, Co, KH, KW, s, p, reflect;
};

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

// lane-private slice descriptors: slice j of this lane covers k = 4 * (j * 64 + lane) .. +3, i.e. tap (ky, kx) and
// channels ci..ci+3.  meta = ky | kx << 8 | ci << 16, or -1 past the end of K.
template <int KQ>
__device__ __forceinline__ void make_meta(const Geom& g, int lane, int (&meta)[KQ]) {
    const int K = g.KH * g.KW * g.Ci;
#pragma unroll
    for (int j = 0; j < KQ; ++j) {
        const int k = 4 * (j * 64 + lane);
        if (k < K) {
            const int tap = k / g.Ci, ci = k - tap * g.Ci;
            const int ky = tap / g.KW, kx = tap - ky * g.KW;
            meta[j] = ky | (kx << 8) | (ci << 16);
        } else {
            meta[j] = -1;
        }
    }
}

__device__ __forceinline__ float4 round_h4(float4 v) { return make_float4(round_h(v.x), round_h(v.y), round_h(v.z), round_h(v.w)); }

// hp != 0 (MG_PRECISION_F16): operands are rounded to float16 as they are loaded, products / sums stay float32
template <int KQ>
__device__ __forceinline__ void gather_row(const Geom& g, const float* __restrict__ x, int b, int oy, int ox,
                                           const int (&meta)[KQ], float4 (&a)[KQ], int hp) {
#pragma unroll
    for (int j = 0; j < KQ; ++j) {
        a[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (meta[j] < 0) continue;
        int iy = oy * g.s - g.p + (meta[j] & 0xff), ix = ox * g.s - g.p + ((meta[j] >> 8) & 0xff);
        if (g.reflect) {
            iy = reflect_idx(iy, g.H);
            ix = reflect_idx(ix, g.W);
        } else if (iy < 0 || iy >= g.H || ix < 0 || ix >= g.W) {
            continue;
        }
        a[j] = *reinterpret_cast<const float4*>(x + ((size_t)(b * g.H + iy) * g.W + ix) * g.Ci + (meta[j] >> 16));
        if (hp) a[j] = round_h4(a[j]);
    }
}

template <int KQ>
__global__ __launch_bounds__(256) void conv_rowdot_fwd_kernel(Geom g, const float* __restrict__ x,
                                                              const float* __restrict__ w,
                                                              const float* __restrict__ bias, float* __restrict__ y,
                                                              int act, int hp) {
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = gridDim.x * 4;
    const int M = g.B * g.OH * g.OW;
    int meta[KQ];
    make_meta<KQ>(g, lane, meta);
    float4 wv[KQ];
#pragma unroll
    for (int j = 0; j < KQ; ++j)
        wv[j] = (meta[j] >= 0) ? *reinterpret_cast<const float4*>(w + 4 * (j * 64 + lane)) : make_float4(0.f, 0.f, 0.f, 0.f);
    if (hp) {
#pragma unroll
        for (int j = 0; j < KQ; ++j) wv[j] = round_h4(wv[j]);
    }
    const float bv = bias ? bias[0] : 0.0f;
    for (int m = wave; m < M; m += nwaves) {
        const int b = m / (g.OH * g.OW), rem = m - b * (g.OH * g.OW);
        const int oy = rem / g.OW, ox = rem - oy * g.OW;
        float4 a[KQ];
        gather_row<KQ>(g, x, b, oy, ox, meta, a, hp);
        float s = 0.0f;
#pragma unroll
        for (int j = 0; j < KQ; ++j) s += a[j].x * wv[j].x + a[j].y * wv[j].y + a[j].z * wv[j].z + a[j].w * wv[j].w;
        s = wave_sum(s);
        if (lane == 0) {
            const float v = apply_act(s + bv, act);
            y[m] = hp ? round_h(v) : v;
        }
    }
}

// per-wave partial rows: part[wave][0..K) = sum_m dy[m] * A[m][k], part[wave][K] = sum_m dy[m]; row pitch KP
template <int KQ>
__global__ __launch_bounds__(256) void conv_rowdot_wgrad_kernel(Geom g, const float* __restrict__ x,
                                                                const float* __restrict__ dy, float* __restrict__ part,
                                                                int KP, int hp) {
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = gridDim.x * 4;
    const int M = g.B * g.OH * g.OW, K = g.KH * g.KW * g.Ci;
    int meta[KQ];
    make_meta<KQ>(g, lane, meta);
    float4 acc[KQ];
#pragma unroll
    for (int j = 0; j < KQ; ++j) acc[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    float gsum = 0.0f;
    for (int m = wave; m < M; m += nwaves) {
        const int b = m / (g.OH * g.OW), rem = m - b * (g.OH * g.OW);
        const int oy = rem / g.OW, ox = rem - oy * g.OW;
        const float g0 = dy[m];
        const float gv = hp ? round_h(g0) : g0;
        float4 a[KQ];
        gather_row<KQ>(g, x, b, oy, ox, meta, a, hp);
#pragma unroll
        for (int j = 0; j < KQ; ++j) {
            acc[j].x += gv * a[j].x; acc[j].y += gv * a[j].y; acc[j].z += gv * a[j].z; acc[j].w += gv * a[j].w;
        }
        gsum += g0;
    }
    float* row = part + (size_t)wave * KP;
#pragma unroll
    for (int j = 0; j < KQ; ++j)
        if (meta[j] >= 0) *reinterpret_cast<float4*>(row + 4 * (j * 64 + lane)) = acc[j];
    if (lane == 0) *reinterpret_cast<float4*>(row + K) = make_float4(gsum, 0.f, 0.f, 0.f);
}

// dw (+)= colsum[0..K), dbias (+)= colsum[K]
__global__ void rowdot_finish_kernel(const float* __restrict__ sums, int K, float* __restrict__ dw,
                                     float* __restrict__ dbias, int accumulate) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < K) dw[i] = accumulate ? dw[i] + sums[i] : sums[i];
    if (i == K && dbias) dbias[0] = accumulate ? dbias[0] + sums[K] : sums[K];
}

// ------------------------------------------------------------------------------------------------------------
// LDS-tiled variants for stride-1 single-output-channel layers (the 7x7 tanh head: 64 channels at 128x256; the 4x4
// PatchGAN outputs).  The wave-per-pixel kernels above re-read every input pixel KH*KW times through L1/L2; here a
// workgroup owns an 8x32 output tile, stages the (8+KH-1) x (32+KW-1) input patch of 16 channels at a time in LDS
// (80-byte pixel pitch: conflict-free ds_read_b128 across a row of pixels), and every thread owns one output pixel
// (forward) or a (tap, 4-channel) slice of the weight gradient (backward).  Weights are wave-uniform scalar loads.
// ------------------------------------------------------------------------------------------------------------
constexpr int CT_TY = 8, CT_TX = 32, CT_CC = 16, CT_PITCH = 20;

template <int KH, int KW>
__device__ __forceinline__ void ct_stage(const Geom& g, const float* __restrict__ x, int b, int oy0, int ox0, int c0,
                                         float* __restrict__ patch, int hp) {
    constexpr int PH = CT_TY + KH - 1, PW = CT_TX + KW - 1;
    for (int i = threadIdx.x; i < PH * PW * 4; i += 256) {
        const int pp = i >> 2, q = i & 3;
        const int py = pp / PW, px = pp - py * PW;
        int iy = oy0 - g.p + py, ix = ox0 - g.p + px;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        bool ok = true;
        if (g.reflect) {
            iy = reflect_idx(iy, g.H);
            ix = reflect_idx(ix, g.W);
            ok = iy >= 0 && iy < g.H && ix >= 0 && ix < g.W;      // tiles hanging over the edge of the map
        } else {
            ok = iy >= 0 && iy < g.H && ix >= 0 && ix < g.W;
        }
        if (ok) v = *reinterpret_cast<const float4*>(x + ((size_t)(b * g.H + iy) * g.W + ix) * g.Ci + c0 + 4 * q);
        if (hp) v = round_h4(v);
        *reinterpret_cast<float4*>(patch + pp * CT_PITCH + 4 * q) = v;
    }
}

template <int KH, int KW>
__global__ __launch_bounds__(256) void co1_tile_fwd_kernel(Geom g, const float* __restrict__ x,
                                                           const float* __restrict__ w, const float* __restrict__ bias,
                                                           float* __restrict__ y, int act, int hp) {
    constexpr int PH = CT_TY + KH - 1, PW = CT_TX + KW - 1;
    __shared__ __attribute__((aligned(16))) float patch[PH * PW * CT_PITCH];
    const int b = blockIdx.z, oy0 = blockIdx.y * CT_TY, ox0 = blockIdx.x * CT_TX;
    const int ty = threadIdx.x >> 5, tx = threadIdx.x & 31;
    float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;     // four independent FMA chains
    for (int c0 = 0; c0 < g.Ci; c0 += CT_CC) {
        __syncthreads();
        ct_stage<KH, KW>(g, x, b, oy0, ox0, c0, patch, hp);
        __syncthreads();
#pragma unroll 1
        for (int ky = 0; ky < KH; ++ky)         // rolled: the fully unrolled 7x7 body does not fit the instruction cache
#pragma unroll
            for (int kx = 0; kx < KW; ++kx) {
                const float* pp = patch + ((ty + ky) * PW + tx + kx) * CT_PITCH;
                const float* wp = w + (size_t)(ky * KW + kx) * g.Ci + c0;        // wave-uniform
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 a = *reinterpret_cast<const float4*>(pp + 4 * q);
                    float4 wv = *reinterpret_cast<const float4*>(wp + 4 * q);
                    if (hp) wv = round_h4(wv);
                    a0 = fmaf(a.x, wv.x, a0);
                    a1 = fmaf(a.y, wv.y, a1);
                    a2 = fmaf(a.z, wv.z, a2);
                    a3 = fmaf(a.w, wv.w, a3);
                }
            }
    }
    const float acc = (a0 + a1) + (a2 + a3);
    const int oy = oy0 + ty, ox = ox0 + tx;
    if (oy < g.OH && ox < g.OW) {
        const float v = apply_act(acc + (bias ? bias[0] : 0.0f), act);
        y[(size_t)(b * g.OH + oy) * g.OW + ox] = hp ? round_h(v) : v;
    }
}

// part[tile][0..K) = sum over the tile's pixels of dy * x(tap, channel); part[tile][K] = sum dy.  Row pitch KP.
template <int KH, int KW>
__global__ __launch_bounds__(256) void co1_tile_wgrad_kernel(Geom g, const float* __restrict__ x,
                                                             const float* __restrict__ dy, float* __restrict__ part,
                                                             int KP, int hp) {
    constexpr int PH = CT_TY + KH - 1, PW = CT_TX + KW - 1, NT = KH * KW;
    __shared__ __attribute__((aligned(16))) float patch[PH * PW * CT_PITCH];
    __shared__ float dyt[CT_TY * CT_TX];
    __shared__ float red[4];
    const int b = blockIdx.z, oy0 = blockIdx.y * CT_TY, ox0 = blockIdx.x * CT_TX;
    const int tile = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    float* row = part + (size_t)tile * KP;
    {   // dy tile (zero outside the map) and its plain sum for the bias gradient
        const int ty = threadIdx.x >> 5, tx = threadIdx.x & 31;
        const int oy = oy0 + ty, ox = ox0 + tx;
        const float g0 = (oy < g.OH && ox < g.OW) ? dy[(size_t)(b * g.OH + oy) * g.OW + ox] : 0.0f;
        dyt[threadIdx.x] = hp ? round_h(g0) : g0;
        const float s = wave_sum(g0);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    }
    // this thread's weight-gradient slice: tap = tid / 4 (taps beyond NT idle), channels 4*(tid % 4) of each chunk
    const int tap = threadIdx.x >> 2, cq = threadIdx.x & 3;
    const int ky = tap / KW, kx = tap - ky * KW;
    const int K = NT * g.Ci;
    for (int c0 = 0; c0 < g.Ci; c0 += CT_CC) {
        __syncthreads();
        ct_stage<KH, KW>(g, x, b, oy0, ox0, c0, patch, hp);
        __syncthreads();
        if (tap < NT) {
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
            const float* pp = patch + (ky * PW + kx) * CT_PITCH + 4 * cq;
#pragma unroll 1
            for (int py = 0; py < CT_TY; ++py)
#pragma unroll 8
                for (int px = 0; px < CT_TX; ++px) {
                    const float gv = dyt[py * CT_TX + px];
                    const float4 a = *reinterpret_cast<const float4*>(pp + (py * PW + px) * CT_PITCH);
                    acc.x = fmaf(gv, a.x, acc.x); acc.y = fmaf(gv, a.y, acc.y);
                    acc.z = fmaf(gv, a.z, acc.z); acc.w = fmaf(gv, a.w, acc.w);
                }
            *reinterpret_cast<float4*>(row + (size_t)tap * g.Ci + c0 + 4 * cq) = acc;
        }
    }
    if (threadIdx.x == 0) *reinterpret_cast<float4*>(row + K) = make_float4(red[0] + red[1] + red[2] + red[3], 0.f, 0.f, 0.f);
}

inline int co1_tile_kind(const mg_conv_geom* g) {      // 0: not eligible, 1: 7x7, 2: 4x4
    constexpr bool off = false;
    if (off || g->Co != 1 || g->stride != 1 || g->Ci % CT_CC != 0) return 0;
    if (g->reflect && (g->pad >= g->H || g->pad >= g->W)) return 0;
    if (g->KH == 7 && g->KW == 7) return 1;
    // 4x4 PatchGAN outputs (512 channels on 19x35 maps: few tiles, 32 channel chunks each) measured 4x slower than
    // the wave-per-pixel kernels -- instantiated for tests (MG_CO1_TILE_4X4=1), not used by default
    constexpr bool k4 = false;           // (the tiled 4x4 form measured 4x slower than wave-per-pixel on the PatchGAN outputs)
    if (k4 && g->KH == 4 && g->KW == 4) return 2;
    return 0;
}
inline dim3 co1_grid(const mg_conv_geom* g) {
    return dim3((g->OW + CT_TX - 1) / CT_TX, (g->OH + CT_TY - 1) / CT_TY, g->B);
}

inline int rowdot_waves(long long M) {
    long long w = (M + 3) / 4;          // >= 4 pixels per wave
    if (w > 4096) w = 4096;
    if (w < 4) w = 4;
    return (int)((w + 3) / 4 * 4);
}

}  // namespace

// internal to the library (called from conv_igemm.hip, another translation unit): not part of the C ABI, hidden from the .so's exports
#define MG_INTERNAL __attribute__((visibility("hidden")))
extern "C" {

// 0: not applicable, otherwise the per-lane slice count of the kernel instance that would run
MG_INTERNAL int mg_conv_rowdot_kq(const mg_conv_geom* g) {
    if (!g || g->Co != 1 || g->Ci % 4 != 0 || g->KH > 255 || g->KW > 255 || g->Ci >= 32768) return 0;
    const int K = g->KH * g->KW * g->Ci;
    if (K <= 16 * 256) return 16;
    if (K <= 32 * 256) return 32;
    return 0;
}

MG_INTERNAL int mg_conv_rowdot_fwd(const mg_conv_geom* g, const float* x, const float* w, const float* bias, float* y, int act,
                       void* stream) {
    const int kq = mg_conv_rowdot_kq(g);
    if (!kq || !x || !w || !y) return MG_ERR_ARG;
    const Geom gg{g->B, g->H, g->W, g->Ci, g->OH, g->OW, g->Co, g->KH, g->KW, g->stride, g->pad, g->reflect};
    if (const int kind = co1_tile_kind(g)) {
        if (kind == 1)
            hipLaunchKernelGGL((co1_tile_fwd_kernel<7, 7>), co1_grid(g), dim3(256), 0, (hipStream_t)stream, gg, x, w, bias, y,
                               act, g->precision);
        else
            hipLaunchKernelGGL((co1_tile_fwd_kernel<4, 4>), co1_grid(g), dim3(256), 0, (hipStream_t)stream, gg, x, w, bias, y,
                               act, g->precision);
        MG_CHECK_LAUNCH();
        return MG_OK;
    }
    const int nw = rowdot_waves((long long)g->B * g->OH * g->OW);
    if (kq == 16)
        hipLaunchKernelGGL(conv_rowdot_fwd_kernel<16>, dim3(nw / 4), dim3(256), 0, (hipStream_t)stream, gg, x, w, bias, y, act,
                           g->precision);
    else
        hipLaunchKernelGGL(conv_rowdot_fwd_kernel<32>, dim3(nw / 4), dim3(256), 0, (hipStream_t)stream, gg, x, w, bias, y, act,
                           g->precision);
    MG_CHECK_LAUNCH();
    return MG_OK;
}

MG_INTERNAL size_t mg_conv_rowdot_wgrad_workspace(const mg_conv_geom* g) {
    if (!mg_conv_rowdot_kq(g)) return 0;
    const int KP = g->KH * g->KW * g->Ci + 4;
    const dim3 tg = co1_grid(g);
    const int nw = co1_tile_kind(g) ? (int)(tg.x * tg.y * tg.z) : rowdot_waves((long long)g->B * g->OH * g->OW);
    return ((size_t)nw * KP + KP) * sizeof(float) + mg_colsum_workspace(nw, KP) + 512;
}

MG_INTERNAL int mg_conv_rowdot_wgrad(const mg_conv_geom* g, const float* x, const float* dy, float* dw, float* dbias,
                         int accumulate, void* workspace, size_t workspace_bytes, void* stream) {
    const int kq = mg_conv_rowdot_kq(g);
    if (!kq || !x || !dy || !dw || !workspace || workspace_bytes < mg_conv_rowdot_wgrad_workspace(g)) return MG_ERR_ARG;
    const Geom gg{g->B, g->H, g->W, g->Ci, g->OH, g->OW, g->Co, g->KH, g->KW, g->stride, g->pad, g->reflect};
    hipStream_t st = (hipStream_t)stream;
    const int K = g->KH * g->KW * g->Ci, KP = K + 4;
    const int kind = co1_tile_kind(g);
    const dim3 tg = co1_grid(g);
    const int nw = kind ? (int)(tg.x * tg.y * tg.z) : rowdot_waves((long long)g->B * g->OH * g->OW);
    float* part = (float*)workspace;                  // [nw][KP]
    float* sums = part + (size_t)nw * KP;             // [KP]
    char* cs_ws = (char*)(sums + KP);
    cs_ws += (16 - (reinterpret_cast<uintptr_t>(cs_ws) & 15)) & 15;
    if (kind == 1)
        hipLaunchKernelGGL((co1_tile_wgrad_kernel<7, 7>), tg, dim3(256), 0, st, gg, x, dy, part, KP, g->precision);
    else if (kind == 2)
        hipLaunchKernelGGL((co1_tile_wgrad_kernel<4, 4>), tg, dim3(256), 0, st, gg, x, dy, part, KP, g->precision);
    else if (kq == 16)
        hipLaunchKernelGGL(conv_rowdot_wgrad_kernel<16>, dim3(nw / 4), dim3(256), 0, st, gg, x, dy, part, KP, g->precision);
    else
        hipLaunchKernelGGL(conv_rowdot_wgrad_kernel<32>, dim3(nw / 4), dim3(256), 0, st, gg, x, dy, part, KP, g->precision);
    MG_CHECK_LAUNCH();
    const int rc = mg_colsum(part, nw, KP, sums, 0, cs_ws, mg_colsum_workspace(nw, KP), stream);
    if (rc != MG_OK) return rc;
    hipLaunchKernelGGL(rowdot_finish_kernel, dim3((K + 256) / 256), dim3(256), 0, st, (const float*)sums, K, dw, dbias,
                       accumulate);
    MG_CHECK_LAUNCH();
    return MG_OK;
}

}  // extern "C"
