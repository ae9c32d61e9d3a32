// F2: the evaluation metrics of util/util.py:132-177 (compute_matrics) on the device, so that
// train.py:104-134 (eval_model) and generate_audio.py:60-61 never copy waveforms to the host.
//   row sums          sum hr^2, sum (sr - hr)^2, sum (lr - hr)^2 per clip, double precision  -> MSE, SNR_sr, SNR_lr
//   STFT frames       reflect-padded (center=True), windowed frames of n_fft samples -- the A operand of the DFT, which
//                     runs as a dense [frames, n_fft] x [n_fft, 2 * (n_fft / 2 + 1)] GEMM on the f32 MFMA pipe (the
//                     1x1 case of mg_conv_fwd; the host passes the cos / -sin table as the "weights")
//   LSD per frame     sqrt(mean_k (log10(|X_hr|^2 + 1e-6) - log10(|X_sr|^2 + 1e-6))^2)
// All HBM-bound streams next to a small GEMM.
#include "common.h"
#include "mdctgan_hip.h"

namespace {

__global__ __launch_bounds__(256) void metrics_rows_kernel(const float* __restrict__ hr, const float* __restrict__ lr,
                                                           const float* __restrict__ sr, int T, double* __restrict__ out) {
    __shared__ double red[3][4];
    const int b = blockIdx.x;
    const float* h = hr + (size_t)b * T;
    const float* l = lr + (size_t)b * T;
    const float* s = sr + (size_t)b * T;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0;
    for (int i = threadIdx.x; i < T; i += 256) {
        const double hv = h[i], ds = (double)s[i] - hv, dl = (double)l[i] - hv;
        a0 += hv * hv;
        a1 += ds * ds;
        a2 += dl * dl;
    }
    a0 = wave_sum_d(a0); a1 = wave_sum_d(a1); a2 = wave_sum_d(a2);
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = a0; red[1][threadIdx.x >> 6] = a1; red[2][threadIdx.x >> 6] = a2; }
    __syncthreads();
    if (threadIdx.x < 3) out[b * 3 + threadIdx.x] = red[threadIdx.x][0] + red[threadIdx.x][1] + red[threadIdx.x][2] + red[threadIdx.x][3];
}

__device__ __forceinline__ int reflect_t(int t, int T) {
    if (t < 0) t = -t;
    if (t >= T) t = 2 * (T - 1) - t;
    return t;
}

// frames[(b * F + f) * N + n] = x[b][reflect(f * hop + n - (center ? N / 2 : 0))] * window[n]
__global__ void stft_frames_kernel(const float* __restrict__ x, int B, int T, const float* __restrict__ window, int N,
                                   int hop, int center, int F, float* __restrict__ frames) {
    const size_t total = (size_t)B * F * N;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int n = (int)(i % N);
        const size_t bf = i / N;
        const int f = (int)(bf % F), b = (int)(bf / F);
        int t = f * hop + n - (center ? N / 2 : 0);
        float v = 0.0f;
        if (center) v = x[(size_t)b * T + reflect_t(t, T)];
        else if (t < T) v = x[(size_t)b * T + t];
        frames[i] = v * window[n];
    }
}

// spectra: [M][2 * NB] interleaved (re, im).  out[m] = sqrt(mean_k (log10(p_a + 1e-6) - log10(p_b + 1e-6))^2)
__global__ __launch_bounds__(64) void lsd_frames_kernel(const float* __restrict__ sa, const float* __restrict__ sb, int NB,
                                                        float* __restrict__ out) {
    const int m = blockIdx.x;
    const float2* a = reinterpret_cast<const float2*>(sa + (size_t)m * 2 * NB);
    const float2* b = reinterpret_cast<const float2*>(sb + (size_t)m * 2 * NB);
    double acc = 0.0;
    for (int k = threadIdx.x; k < NB; k += 64) {
        const float2 va = a[k], vb = b[k];
        const double pa = (double)va.x * va.x + (double)va.y * va.y, pb = (double)vb.x * vb.x + (double)vb.y * vb.y;
        const double d = log10(pa + 1e-6) - log10(pb + 1e-6);
        acc += d * d;
    }
    acc = wave_sum_d(acc);
    if (threadIdx.x == 0) out[m] = (float)sqrt(acc / NB);
}

}  // namespace

extern "C" {

int mg_metrics_rows(const float* hr, const float* lr, const float* sr, int B, int T, double* out, void* stream) {
    if (!hr || !lr || !sr || !out || B <= 0 || T <= 0) return MG_ERR_ARG;
    hipLaunchKernelGGL(metrics_rows_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, hr, lr, sr, T, out);
    MG_CHECK_LAUNCH();
    return MG_OK;
}

int mg_stft_num_frames(int T, int n_fft, int hop, int center) {
    if (T <= 0 || n_fft <= 0 || hop <= 0) return -1;
    const int padded = center ? T + 2 * (n_fft / 2) : T;
    return padded < n_fft ? -1 : 1 + (padded - n_fft) / hop;
}

int mg_stft_frames(const float* x, int B, int T, const float* window, int n_fft, int hop, int center, float* frames,
                   void* stream) {
    const int F = mg_stft_num_frames(T, n_fft, hop, center);
    if (!x || !window || !frames || B <= 0 || F <= 0 || (center && n_fft / 2 >= T)) return MG_ERR_ARG;
    const size_t total = (size_t)B * F * n_fft;
    const unsigned blocks = (unsigned)((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256);
    hipLaunchKernelGGL(stft_frames_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, B, T, window, n_fft, hop,
                       center, F, frames);
    MG_CHECK_LAUNCH();
    return MG_OK;
}

int mg_lsd_frames(const float* spec_a, const float* spec_b, long long n_frames, int n_bins, float* out, void* stream) {
    if (!spec_a || !spec_b || !out || n_frames <= 0 || n_bins <= 0) return MG_ERR_ARG;
    hipLaunchKernelGGL(lsd_frames_kernel, dim3((unsigned)n_frames), dim3(64), 0, (hipStream_t)stream, spec_a, spec_b, n_bins,
                       out);
    MG_CHECK_LAUNCH();
    return MG_OK;
}

}  // extern "C"
