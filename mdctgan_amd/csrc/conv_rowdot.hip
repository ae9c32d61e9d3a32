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

inline int rowdot_waves(long long M) {
    long long w = (M + 3) / 4;          // >= 4 pixels per wave
    if (w > 4096) w = 4096;
    if (w < 4) w = 4;
    return (int)((w + 3) / 4 * 4);
}

}  // namespace

extern "C" {

// 0: not applicable, otherwise the per-lane slice count of the kernel instance that would run
int mg_conv_rowdot_kq(const mg_conv_geom* g) {
    if (!g || g->Co != 1 || g->Ci % 4 != 0 || g->KH > 255 || g->KW > 255 || g->Ci >= 32768) return 0;
    const int K = g->KH * g->KW * g->Ci;
    if (K <= 16 * 256) return 16;
    if (K <= 32 * 256) return 32;
    return 0;
}

int mg_conv_rowdot_fwd(const mg_conv_geom* g, const float* x, const float* w, const float* bias, float* y, int act,
                       void* stream) {
    const int kq = mg_conv_rowdot_kq(g);
    if (!kq || !x || !w || !y) return MG_ERR_ARG;
    const Geom gg{g->B, g->H, g->W, g->Ci, g->OH, g->OW, g->Co, g->KH, g->KW, g->stride, g->pad, g->reflect};
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

size_t mg_conv_rowdot_wgrad_workspace(const mg_conv_geom* g) {
    if (!mg_conv_rowdot_kq(g)) return 0;
    const int KP = g->KH * g->KW * g->Ci + 4;
    const int nw = rowdot_waves((long long)g->B * g->OH * g->OW);
    return ((size_t)nw * KP + KP) * sizeof(float) + mg_colsum_workspace(nw, KP) + 512;
}

int mg_conv_rowdot_wgrad(const mg_conv_geom* g, const float* x, const float* dy, float* dw, float* dbias,
                         int accumulate, void* workspace, size_t workspace_bytes, void* stream) {
    const int kq = mg_conv_rowdot_kq(g);
    if (!kq || !x || !dy || !dw || !workspace || workspace_bytes < mg_conv_rowdot_wgrad_workspace(g)) return MG_ERR_ARG;
    const Geom gg{g->B, g->H, g->W, g->Ci, g->OH, g->OW, g->Co, g->KH, g->KW, g->stride, g->pad, g->reflect};
    hipStream_t st = (hipStream_t)stream;
    const int K = g->KH * g->KW * g->Ci, KP = K + 4;
    const int nw = rowdot_waves((long long)g->B * g->OH * g->OW);
    float* part = (float*)workspace;                  // [nw][KP]
    float* sums = part + (size_t)nw * KP;             // [KP]
    char* cs_ws = (char*)(sums + KP);
    cs_ws += (16 - (reinterpret_cast<uintptr_t>(cs_ws) & 15)) & 15;
    if (kq == 16)
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
