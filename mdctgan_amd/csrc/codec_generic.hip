// Generic-geometry transform + codec (SURVEY A2 / A5-A8 beyond the hot path's n_fft = 512 arcsinh configuration):
//   * MDCT4 / IMDCT4 for any win_length <= n_fft and hop_length <= win_length (models/mdct.py:365-489; the reference's
//     class default is n_fft = 2048): framing + window, the cosine contraction as a dense exact-f32 GEMM (the 1x1 case of
//     mg_conv_fwd), window + overlap-add + centre crop;
//   * every branch of Audio2MDCT.normalize / denormalize (models/pix2pixHD_model.py:83-137): arcsinh, raw, the dB codec
//     (torchaudio amplitude_to_DB / DB_to_amplitude, restated from their published formulas: torchaudio is not installed,
//     parity of those two functions is UNPINNED) and --explicit_encoding (two dB channels mixing the positive and negative
//     parts with alpha), with the fixed (--abs_norm) or per-sample min/max range normalisation, the (v, 2|v| + nr0) pair
//     and the mean / variance sums.
// HBM-bound float4 / scalar streams; the n_fft = 512 arcsinh path keeps its fused kernels (mdct.hip).
#include "common.h"
#include "mdctgan_hip.h"
#include <math.h>

namespace {

constexpr float LN10F = 2.3025851249694824f;   // float32(log(10)), as torch.log(torch.tensor(10.0))
enum { C_RAW = 0, C_ARCSINH = 1, C_RANGE = 2, C_DB = 3, C_EXPLICIT = 4 };

struct CodecG { int mode; float gain, alpha, min_value, nr0, nr1, mn, mx; };

__device__ __forceinline__ float amp_to_db(float a, float amin) {       // aF.amplitude_to_DB(a, 20.0, amin, 1.0)
    return 20.0f * log10f(fmaxf(a, amin)) - 20.0f;
}
__device__ __forceinline__ float db_to_amp(float l) {                   // aF.DB_to_amplitude(l, 10.0, 0.5)
    return 10.0f * powf(powf(10.0f, 0.1f * l), 0.5f);
}
__device__ __forceinline__ void encode(float x, const CodecG& c, float& l0, float& l1) {
    l1 = 0.0f;
    if (c.mode == C_ARCSINH) l0 = asinhf(c.gain * x) / LN10F;
    else if (c.mode == C_DB) l0 = amp_to_db(fabsf(x) + c.min_value, c.min_value);
    else if (c.mode == C_EXPLICIT) {
        const float neg = 0.5f * (fabsf(x) - x), pos = x + neg;
        l0 = amp_to_db(c.alpha * pos + (1.0f - c.alpha) * neg, c.min_value);
        l1 = amp_to_db((1.0f - c.alpha) * pos + c.alpha * neg, c.min_value);
    } else l0 = x;
}

__global__ void frames_window_kernel(const float* __restrict__ x, int T, int win, int hop, int start_pad, int F,
                                     const float* __restrict__ w, float* __restrict__ frames) {
    const int b = blockIdx.y;
    const size_t n = (size_t)F * win;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int f = (int)(i / win), k = (int)(i - (size_t)f * win);
        const long long t = (long long)f * hop + k - start_pad;
        frames[(size_t)b * n + i] = (t >= 0 && t < T) ? __fmul_rn(x[(size_t)b * T + t], w[k]) : 0.0f;     // mdct.py:410
    }
}

// out [B][C][n]; defer: write the un-normalised value and collect per-(b, channel) min / max
__global__ void codec_forward_kernel(const float* __restrict__ X, int n, CodecG c, int C, int defer, float* __restrict__ out,
                                     float* __restrict__ pair, unsigned* __restrict__ minmax_ord, double* __restrict__ stats) {
    const int b = blockIdx.y, lane = threadIdx.x & 63;
    float mn0 = INFINITY, mx0 = -INFINITY, mn1 = INFINITY, mx1 = -INFINITY;
    double s1 = 0.0, s2 = 0.0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        float l0, l1;
        encode(X[(size_t)b * n + i], c, l0, l1);
        if (stats && c.mode != C_RAW) {
            s1 += (double)l0; s2 += (double)l0 * l0;
            if (C == 2) { s1 += (double)l1; s2 += (double)l1 * l1; }
        }
        float v0 = l0, v1 = l1;
        if (c.mode != C_RAW) {
            if (defer) {
                mn0 = fminf(mn0, l0); mx0 = fmaxf(mx0, l0);
                mn1 = fminf(mn1, l1); mx1 = fmaxf(mx1, l1);
            } else {
                v0 = (l0 - c.mn) / (c.mx - c.mn) * (c.nr1 - c.nr0) + c.nr0;
                v1 = (l1 - c.mn) / (c.mx - c.mn) * (c.nr1 - c.nr0) + c.nr0;
            }
        }
        out[((size_t)b * C) * n + i] = v0;
        if (C == 2) out[((size_t)b * C + 1) * n + i] = v1;
        if (pair && !defer) *reinterpret_cast<float2*>(pair + 2 * ((size_t)b * n + i)) = make_float2(v0, fabsf(v0) * 2.0f + c.nr0);
    }
    if (defer && minmax_ord) {
        mn0 = wave_min(mn0); mx0 = wave_max(mx0);
        if (lane == 0) { atomicMin(minmax_ord + 2 * (b * C), f2ord(mn0)); atomicMax(minmax_ord + 2 * (b * C) + 1, f2ord(mx0)); }
        if (C == 2) {
            mn1 = wave_min(mn1); mx1 = wave_max(mx1);
            if (lane == 0) { atomicMin(minmax_ord + 2 * (b * C + 1), f2ord(mn1)); atomicMax(minmax_ord + 2 * (b * C + 1) + 1, f2ord(mx1)); }
        }
    }
    if (stats && c.mode != C_RAW) {
        s1 = wave_sum_d(s1); s2 = wave_sum_d(s2);
        if (lane == 0) { atomicAdd(stats, s1); atomicAdd(stats + 1, s2); }
    }
}
__global__ void init_minmax_kernel(unsigned* p, int n_pairs) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_pairs) { p[2 * i] = 0xffffffffu; p[2 * i + 1] = 0u; }
}
// second pass of the per-sample normalisation: rows = B * C
__global__ void codec_range_kernel(float* __restrict__ out, float* __restrict__ pair, int n, const unsigned* __restrict__ minmax_ord,
                                   float nr0, float nr1, float* __restrict__ mn_out, float* __restrict__ mx_out) {
    const int row = blockIdx.y;
    const float mn = ord2f(minmax_ord[2 * row]), mx = ord2f(minmax_ord[2 * row + 1]);
    if (blockIdx.x == 0 && threadIdx.x == 0) { mn_out[row] = mn; mx_out[row] = mx; }
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const size_t o = (size_t)row * n + i;
        const float v = (out[o] - mn) / (mx - mn) * (nr1 - nr0) + nr0;
        out[o] = v;
        if (pair) *reinterpret_cast<float2*>(pair + 2 * o) = make_float2(v, fabsf(v) * 2.0f + nr0);
    }
}

// spec [B][C][n] -> X [B][n]
__global__ void codec_inverse_kernel(const float* __restrict__ spec, int n, CodecG c, int C, const float* __restrict__ mn_b,
                                     const float* __restrict__ mx_b, float* __restrict__ X) {
    const int b = blockIdx.y;
    float mn0 = c.mn, mx0 = c.mx, mn1 = c.mn, mx1 = c.mx;
    if (mn_b) { mn0 = mn_b[b * C]; mx0 = mx_b[b * C]; if (C == 2) { mn1 = mn_b[b * C + 1]; mx1 = mx_b[b * C + 1]; } }
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const float v0 = spec[((size_t)b * C) * n + i];
        float r;
        if (c.mode == C_RAW) r = v0;
        else {
            const float l0 = (v0 - c.nr0) / (c.nr1 - c.nr0) * (mx0 - mn0) + mn0;
            if (c.mode == C_ARCSINH) r = sinhf(l0 * LN10F) / c.gain;
            else if (c.mode == C_DB) r = db_to_amp(l0) - c.min_value;
            else if (c.mode == C_EXPLICIT) {
                const float v1 = spec[((size_t)b * C + 1) * n + i];
                const float l1 = (v1 - c.nr0) / (c.nr1 - c.nr0) * (mx1 - mn1) + mn1;
                r = ((db_to_amp(l0) - c.min_value) - (db_to_amp(l1) - c.min_value)) / (2.0f * c.alpha - 1.0f);
            } else r = l0;
        }
        X[(size_t)b * n + i] = r;
    }
}

// out[b][t] = 4/N * sum_f w[n] * Y[b][f][n], n = t + crop - f * hop in [0, win)   (mdct.py:473-486: window, fold, crop)
template <typename OutT>
__global__ void overlap_add_kernel(const float* __restrict__ Y, int F, int win, int hop, int n_fft, const float* __restrict__ w,
                                   int crop, OutT* __restrict__ out, int out_len) {
    const int b = blockIdx.y;
    const float scale = 4.0f / (float)n_fft;
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < out_len; t += gridDim.x * blockDim.x) {
        const int tp = t + crop;
        int f_lo = (tp - win + hop) / hop;          // ceil((tp - win + 1) / hop) for tp - win + 1 > 0
        if (tp - win + 1 <= 0) f_lo = 0;
        int f_hi = tp / hop;
        if (f_hi > F - 1) f_hi = F - 1;
        float acc = 0.0f;
        for (int f = f_lo; f <= f_hi; ++f) {
            const int k = tp - f * hop;
            acc += w[k] * Y[((size_t)b * F + f) * win + k];
        }
        out[(size_t)b * out_len + t] = (OutT)(acc * scale);
    }
}

inline unsigned cg_grid(size_t n) {
    size_t b = (n + 255) / 256;
    return (unsigned)(b > 4096 ? 4096 : (b < 1 ? 1 : b));
}

}  // namespace

extern "C" {

int mg_frames_window(const float* x, int B, int T, int win, int hop, int start_pad, int F, const float* window,
                     float* frames, void* stream) {
    if (!x || !window || !frames || B <= 0 || T <= 0 || win <= 0 || hop <= 0 || F <= 0 || start_pad < 0) return MG_ERR_ARG;
    hipLaunchKernelGGL(frames_window_kernel, dim3(cg_grid((size_t)F * win), B), dim3(256), 0, (hipStream_t)stream, x, T, win, hop,
                       start_pad, F, window, frames);
    MG_CHECK_LAUNCH();
    return MG_OK;
}

int mg_codec_forward(const float* X, int B, int n, int mode, float gain, float alpha, float min_value, float nr0, float nr1,
                     float src_min, float src_max, int per_sample, float* out, float* pair, float* min_out, float* max_out,
                     void* scratch_u32, double* stats, void* stream) {
    if (!X || !out || B <= 0 || n <= 0 || mode < C_RAW || mode > C_EXPLICIT) return MG_ERR_ARG;
    const int C = mode == C_EXPLICIT ? 2 : 1;
    if (pair && C != 1) return MG_ERR_ARG;
    if (per_sample && (mode == C_RAW || !min_out || !max_out || !scratch_u32)) return MG_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    const CodecG c{mode, gain, alpha, min_value, nr0, nr1, src_min, src_max};
    if (stats) hipMemsetAsync(stats, 0, 2 * sizeof(double), st);
    if (per_sample)
        hipLaunchKernelGGL(init_minmax_kernel, dim3((B * C + 255) / 256), dim3(256), 0, st, (unsigned*)scratch_u32, B * C);
    hipLaunchKernelGGL(codec_forward_kernel, dim3(cg_grid(n) > 256 ? 256 : cg_grid(n), B), dim3(256), 0, st, X, n, c, C, per_sample,
                       out, pair, (unsigned*)scratch_u32, stats);
    if (per_sample)
        hipLaunchKernelGGL(codec_range_kernel, dim3(cg_grid(n) > 256 ? 256 : cg_grid(n), B * C), dim3(256), 0, st, out, pair, n,
                           (const unsigned*)scratch_u32, nr0, nr1, min_out, max_out);
    MG_CHECK_LAUNCH();
    return MG_OK;
}

int mg_codec_inverse(const float* spec, int B, int n, int mode, float gain, float alpha, float min_value, float nr0, float nr1,
                     float src_min, float src_max, const float* min_b, const float* max_b, float* X, void* stream) {
    if (!spec || !X || B <= 0 || n <= 0 || mode < C_RAW || mode > C_EXPLICIT || ((min_b == nullptr) != (max_b == nullptr)))
        return MG_ERR_ARG;
    const CodecG c{mode, gain, alpha, min_value, nr0, nr1, src_min, src_max};
    hipLaunchKernelGGL(codec_inverse_kernel, dim3(cg_grid(n) > 256 ? 256 : cg_grid(n), B), dim3(256), 0, (hipStream_t)stream, spec, n,
                       c, mode == C_EXPLICIT ? 2 : 1, min_b, max_b, X);
    MG_CHECK_LAUNCH();
    return MG_OK;
}

int mg_overlap_add(const float* Y, int B, int F, int win, int hop, int n_fft, const float* window, int crop, void* out,
                   int out_len, int is_f64, void* stream) {
    if (!Y || !window || !out || B <= 0 || F <= 0 || win <= 0 || hop <= 0 || hop > win || out_len <= 0 || crop < 0) return MG_ERR_ARG;
    if (crop + out_len > (F - 1) * hop + win) return MG_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (is_f64)
        hipLaunchKernelGGL(overlap_add_kernel<double>, dim3(cg_grid(out_len), B), dim3(256), 0, st, Y, F, win, hop, n_fft, window,
                           crop, (double*)out, out_len);
    else
        hipLaunchKernelGGL(overlap_add_kernel<float>, dim3(cg_grid(out_len), B), dim3(256), 0, st, Y, F, win, hop, n_fft, window,
                           crop, (float*)out, out_len);
    MG_CHECK_LAUNCH();
    return MG_OK;
}

}  // extern "C"
