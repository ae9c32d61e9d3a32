// F1: torchaudio.functional.resample (sinc_interp_hann) as a polyphase FIR on the device -- the
// reference's data path builds the low-rate input by resampling every clip down and back up on the CPU
// (data/audio_dataset.py:66-71, 171-177).
//   out[b][n * new + p] = sum_k xpad[b][n * orig + k] * kern[p][k],   xpad = x with `width` zeros in front,
//   k in [0, 2 * width + orig), p in [0, new), cropped to out_len = ceil(new * L / orig).
// HBM-bound: one thread per output sample, the (<= few KB) filter bank in LDS, input reads coalesced along n for the
// down-sampling case and broadcast within a phase group for the up-sampling case.
#include "common.h"
#include "mdctgan_hip.h"

namespace {

template <bool BANK_IN_LDS>     // false: filter banks above 64 KB (e.g. 44.1 kHz <-> 48 kHz = 147:160) stay in L2
__global__ __launch_bounds__(256) void resample_kernel(const float* __restrict__ x, int L, const float* __restrict__ kern,
                                                       int orig, int new_, int width, float* __restrict__ out,
                                                       int out_len) {
    extern __shared__ float kl_s[];                // [new][K]
    const int K = 2 * width + orig;
    if (BANK_IN_LDS) {
        for (int i = threadIdx.x; i < new_ * K; i += blockDim.x) kl_s[i] = kern[i];
        __syncthreads();
    }
    const float* kl = BANK_IN_LDS ? kl_s : kern;
    const int b = blockIdx.y;
    const float* xb = x + (size_t)b * L;
    for (int o = blockIdx.x * blockDim.x + threadIdx.x; o < out_len; o += gridDim.x * blockDim.x) {
        const int n = o / new_, p = o - n * new_;
        const int t0 = n * orig - width;           // sample index of tap 0
        const float* kp = kl + p * K;
        float acc = 0.0f;
        const int k_lo = t0 < 0 ? -t0 : 0;
        const int k_hi = (t0 + K > L) ? L - t0 : K;
        for (int k = k_lo; k < k_hi; ++k) acc = fmaf(xb[t0 + k], kp[k], acc);
        out[(size_t)b * out_len + o] = acc;
    }
}

}  // namespace

extern "C" {

long long mg_resample_length(long long L, int orig, int new_) {
    if (L <= 0 || orig <= 0 || new_ <= 0) return -1;
    return ((long long)new_ * L + orig - 1) / orig;
}

int mg_resample(const float* x, int B, int L, const float* kern, int orig, int new_, int width, float* out, int out_len,
                void* stream) {
    if (!x || !kern || !out || B <= 0 || L <= 0 || orig <= 0 || new_ <= 0 || width < 0) return MG_ERR_ARG;
    if ((long long)out_len != mg_resample_length(L, orig, new_)) return MG_ERR_ARG;
    const size_t lds = (size_t)new_ * (2 * width + orig) * sizeof(float);
    const unsigned bx = (unsigned)((out_len + 255) / 256 > 1024 ? 1024 : (out_len + 255) / 256);
    if (lds <= 48 * 1024)
        hipLaunchKernelGGL(resample_kernel<true>, dim3(bx, B), dim3(256), lds, (hipStream_t)stream, x, L, kern, orig, new_,
                           width, out, out_len);
    else
        hipLaunchKernelGGL(resample_kernel<false>, dim3(bx, B), dim3(256), 0, (hipStream_t)stream, x, L, kern, orig, new_,
                           width, out, out_len);
    MG_CHECK_LAUNCH();
    return MG_OK;
}

}  // extern "C"
