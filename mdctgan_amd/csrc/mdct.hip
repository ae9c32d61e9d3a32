// K1 / K2: framed KBD-windowed MDCT / IMDCT (TDAC fold + 256-point DCT-IV as an f32 MFMA
// contraction) with the arcsinh / range-norm codec and the overlap-add fused in.
//
// Replaces the ATen chains of the reference:
//   K1  models/mdct.py:392-425 (MDCT4.forward)  + models/pix2pixHD_model.py:83-125 (normalize)
//       + :400-402 (the |x|*2+nr0 second channel)
//   K2  models/pix2pixHD_model.py:127-137 (denormalize) + models/mdct.py:457-489 (IMDCT4.forward)
//
// Geometry: n_fft = win = 2*M, hop = M (M = 256 on the hot path).
//
// Two kernels per direction:
//   * mdct_ct.h (mdct4_ct_kernel / imdct4_ct_kernel): the DCT-IV factored into 8- and 16-point DFT stages (12 288 multiply-adds
//     per frame instead of 65 536) -- the default at every size; also writes the NHWC pair alone and the stitched waveform;
//   * this file (mdct4_kernel / imdct4_kernel): the generic dense-table kernels -- per-sample min / max normalisation, returned
//     frames (return_frames=True), float64 output, unaligned or > 4 GiB operands, no stage-matrix image.  One workgroup = FT frames
//     of one clip = one FT x 256 x 256 GEMM tile; 4 waves, each owning 64 output columns of every frame; A (frames x k) is folded
//     straight from HBM into LDS (257-float row pitch), B = the DCT-IV cosine table streams from L2 into VGPRs.
// (Rounds 2-4 also had a tiled-GEMM K1, a table-stationary f32 pair and a bf16 x 3 pair; the factored kernels beat them at every
// size and they were retired from the library in round 5 -- scripts/ubench/mdct_bs.h, mdct_b3.h keep the latter two for the harnesses.)
#include <cstdlib>
#include "common.h"

namespace {

constexpr int M = 256;          // bins per frame = hop
constexpr int LDA = M + 1;      // LDS row pitch (floats)

enum Codec { CODEC_RAW = 0, CODEC_ARCSINH = 1, CODEC_RANGE = 2 };

struct CodecParams {
    int mode;            // Codec
    float gain;          // arcsinh gain
    float nr0, nr1;      // norm_range
    float mn, mx;        // src_range when per_sample == 0
    const float* mn_b;   // per-sample min / max when per_sample == 1 (IMDCT side)
    const float* mx_b;
    int per_sample;
};

// K2 writing into ONE stitched waveform (generate_audio.py:40-53): segment index of clip 0 times pitch minus overlap, samples of
// the waveform, pitch = segment length - overlap, overlap.  pitch == 0: the plain [B, out_len] output.
struct StitchArgs { long long base, total; int pitch, overlap; };

constexpr float LN10F = 2.3025851249694824f;   // float32(log(10)), as torch.log(torch.tensor(10.0))

__device__ __forceinline__ float decode(float v, const CodecParams& c, float mn, float mx) {
    if (c.mode == CODEC_RAW) return v;
    float l = (v - c.nr0) / (c.nr1 - c.nr0) * (mx - mn) + mn;
    if (c.mode == CODEC_ARCSINH) return sinhf(l * LN10F) / c.gain;
    return l;
}

// ---------------------------------------------------------------------------------------------
// K1.  grid = (ceil(F / 32), B), block = 256.
//   audio [B, T] -> spec [B, F, 256] (optionally NHWC pair in2 [B, F, 256, 2] = (v, 2|v| + nr0),
//   optionally windowed frames [B, F, 512], optionally pre-normalisation L plus per-clip min/max
//   and global sum / sum-of-squares for the returned statistics).
// ---------------------------------------------------------------------------------------------
template <int FT>
__global__ __launch_bounds__(256) void mdct4_kernel(
    const float* __restrict__ audio, int T, int F, const float* __restrict__ window,
    const float* __restrict__ dct4, CodecParams cp, float* __restrict__ spec, float* __restrict__ in2,
    float* __restrict__ frames_out, int defer_norm, unsigned* __restrict__ minmax_ord,
    double* __restrict__ stats) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* us = smem;                    // [FT][LDA] folded, windowed frames (A operand)
    float* ws = us + FT * LDA;           // [2 * M] window

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.y, f0 = blockIdx.x * FT;
    const float* x = audio + (size_t)b * T;
    for (int i = tid; i < 2 * M; i += 256) ws[i] = window[i];
    __syncthreads();

    // sample n of frame f sits at padded position f*M + n, i.e. audio index f*M + n - M (centre padding, zeros outside)
    auto sample = [&](int f, int n) -> float {
        const int t = f * M + n - M;
        return (t >= 0 && t < T) ? x[t] : 0.0f;
    };
    // windowed frame z_f[n] = fl32(x[f*M + n] * w[n]) (mdct.py:410, float32 like the reference), then the
    // TDAC fold  u = [-c_r - d, a - b_r]  with quarters a, b, c, d of z.
    constexpr int Q = M / 2;
#pragma unroll 8
    for (int i = tid; i < FT * M; i += 256) {
        const int j = i / M, n = i % M, f = f0 + j;
        float u = 0.0f;
        if (f < F) {
            if (n < Q) {
                const int n1 = 3 * Q - 1 - n, n2 = 3 * Q + n;
                u = -__fmul_rn(sample(f, n1), ws[n1]) - __fmul_rn(sample(f, n2), ws[n2]);
            } else {
                const int m = n - Q, n1 = m, n2 = 2 * Q - 1 - m;
                u = __fmul_rn(sample(f, n1), ws[n1]) - __fmul_rn(sample(f, n2), ws[n2]);
            }
        }
        us[j * LDA + n] = u;
    }
    if (frames_out) {
        for (int i = tid; i < FT * 2 * M; i += 256) {
            const int j = i / (2 * M), n = i % (2 * M);
            if (f0 + j < F) frames_out[((size_t)b * F + f0 + j) * (2 * M) + n] = sample(f0 + j, n) * ws[n];
        }
    }
    __syncthreads();

    // FT x 64 (per wave) x 256 contraction on the f32 MFMA pipe
    constexpr int RB = FT / 32;
    f32x16 acc[RB][2];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) { acc[rb][0] = f32x16{0}; acc[rb][1] = f32x16{0}; }
    const int arow = lane & 31, khalf = lane >> 5;
    const float* ap = us + arow * LDA + khalf;
    // the wave's 64 columns as two interleaved MFMA blocks (block nb = columns 2j + nb): lane j fetches both with
    // one 8-byte load per k
    const float* bp = dct4 + (size_t)khalf * M + wave * 64 + 2 * (lane & 31);
#pragma unroll 8
    for (int kp = 0; kp < M / 2; ++kp) {
        const float2 bb = *reinterpret_cast<const float2*>(bp + (size_t)(2 * kp) * M);
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
            const float a = ap[rb * 32 * LDA + 2 * kp];
            acc[rb][0] = mfma32x32x2(a, bb.x, acc[rb][0]);
            acc[rb][1] = mfma32x32x2(a, bb.y, acc[rb][1]);
        }
    }

    // epilogue: codec + stores
    float mn = cp.mn, mx = cp.mx;
    double s1 = 0.0, s2 = 0.0;
    float vmin = INFINITY, vmax = -INFINITY;
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int f = f0 + rb * 32 + mfma32_row(r, lane);
            const int col = wave * 64 + 2 * (lane & 31) + nb;
            const float xv = acc[rb][nb][r];
            if (f >= F) continue;
            float l = xv;
            if (cp.mode == CODEC_ARCSINH) l = asinhf(cp.gain * xv) / LN10F;
            if (stats && cp.mode != CODEC_RAW) { s1 += (double)l; s2 += (double)l * (double)l; }
            float v;
            if (cp.mode == CODEC_RAW) v = xv;
            else if (defer_norm) { v = l; vmin = fminf(vmin, l); vmax = fmaxf(vmax, l); }
            else v = (l - mn) / (mx - mn) * (cp.nr1 - cp.nr0) + cp.nr0;
            const size_t o = ((size_t)b * F + f) * M + col;
            spec[o] = v;
            if (in2 && !defer_norm) {
                float2 pr = make_float2(v, fabsf(v) * 2.0f + cp.nr0);
                *reinterpret_cast<float2*>(in2 + 2 * o) = pr;
            }
        }
    }
    if (defer_norm && minmax_ord) {
        vmin = wave_min(vmin); vmax = wave_max(vmax);
        if (lane == 0) {
            atomicMin(minmax_ord + 2 * b, f2ord(vmin));
            atomicMax(minmax_ord + 2 * b + 1, f2ord(vmax));
        }
    }
    if (stats && cp.mode != CODEC_RAW) {
        s1 = wave_sum_d(s1); s2 = wave_sum_d(s2);
        if (lane == 0) { atomicAdd(stats, s1); atomicAdd(stats + 1, s2); }
    }
}

// second pass of the per-sample min/max normalisation (no --abs_norm): in place over spec [B, n]
__global__ void range_norm_kernel(float* __restrict__ spec, float* __restrict__ in2, int n,
                                  const unsigned* __restrict__ minmax_ord, float nr0, float nr1,
                                  float* __restrict__ mn_out, float* __restrict__ mx_out) {
    const int b = blockIdx.y;
    const float mn = ord2f(minmax_ord[2 * b]), mx = ord2f(minmax_ord[2 * b + 1]);
    if (blockIdx.x == 0 && threadIdx.x == 0) { mn_out[b] = mn; mx_out[b] = mx; }
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const size_t o = (size_t)b * n + i;
        const float v = (spec[o] - mn) / (mx - mn) * (nr1 - nr0) + nr0;
        spec[o] = v;
        if (in2) *reinterpret_cast<float2*>(in2 + 2 * o) = make_float2(v, fabsf(v) * 2.0f + nr0);
    }
}

__global__ void fill_u32_pairs(unsigned* p, int n_pairs) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_pairs) { p[2 * i] = 0xffffffffu; p[2 * i + 1] = 0u; }
}

// ---------------------------------------------------------------------------------------------
// K2.  grid = (ceil(F / 32), B), block = 256.
//   spec [B, F, 256] (normalised) -> audio [B, (F-1)*256]   (mdct.py:484-486 centre crop)
//   Workgroup (f0) decodes frames f0 .. f0+31 on the MFMA pipe and frame f0-1 (the halo whose second
//   half overlaps hop-block f0) as a VALU dot product riding in the MFMA shadow, then emits hop-blocks
//   h = f0 .. f0+31:  out[(h-1)*M + n] = 4/N * ( w[n] * y_h[n] + w[n+M] * y_{h-1}[n+M] ).
// ---------------------------------------------------------------------------------------------
template <typename OutT, int FT>
__global__ __launch_bounds__(256) void imdct4_kernel(
    const float* __restrict__ spec, int F, const float* __restrict__ window, const float* __restrict__ dct4,
    CodecParams cp, OutT* __restrict__ audio, int out_len, float* __restrict__ frames_out, StitchArgs sa) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* as = smem;                     // [(FT + 1)][LDA]: rows 0..FT-1 = frames f0.., row FT = halo frame f0-1
    float* ws = as + (FT + 1) * LDA;      // [2 * M]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.y, f0 = blockIdx.x * FT;
    float mn = cp.mn, mx = cp.mx;
    if (cp.per_sample) { mn = cp.mn_b[b]; mx = cp.mx_b[b]; }

#pragma unroll 8
    for (int i = tid; i < (FT + 1) * M; i += 256) {
        const int j = i / M, k = i % M;
        const int f = (j == FT) ? f0 - 1 : f0 + j;
        float v = 0.0f;
        if (f >= 0 && f < F) v = decode(spec[((size_t)b * F + f) * M + k], cp, mn, mx);
        as[j * LDA + k] = v;
    }
    for (int i = tid; i < 2 * M; i += 256) ws[i] = window[i];
    __syncthreads();

    constexpr int RB = FT / 32;
    f32x16 acc[RB][2];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) { acc[rb][0] = f32x16{0}; acc[rb][1] = f32x16{0}; }
    float h0 = 0.0f, h1 = 0.0f;
    const int arow = lane & 31, khalf = lane >> 5;
    const float* ap = as + arow * LDA + khalf;
    const float* hp = as + FT * LDA + khalf;
    const float* bp = dct4 + (size_t)khalf * M + wave * 64 + 2 * (lane & 31);     // interleaved column blocks, as in K1
#pragma unroll 8
    for (int kp = 0; kp < M / 2; ++kp) {
        const float xh = hp[2 * kp];
        const float2 bb = *reinterpret_cast<const float2*>(bp + (size_t)(2 * kp) * M);
        const float b0 = bb.x, b1 = bb.y;
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
            const float a = ap[rb * 32 * LDA + 2 * kp];
            acc[rb][0] = mfma32x32x2(a, b0, acc[rb][0]);
            acc[rb][1] = mfma32x32x2(a, b1, acc[rb][1]);
        }
        h0 = fmaf(xh, b0, h0);
        h1 = fmaf(xh, b1, h1);
    }
    h0 += __shfl_xor(h0, 32, 64);
    h1 += __shfl_xor(h1, 32, 64);
    __syncthreads();                       // everyone is done reading `as` as the A operand

    // v = DCT-IV(X) back into LDS (same buffer): rows 0..FT-1 frames, row FT halo
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int j = rb * 32 + mfma32_row(r, lane);
            as[j * LDA + wave * 64 + 2 * (lane & 31) + nb] = acc[rb][nb][r];
        }
    }
    if (lane < 32) {
        as[FT * LDA + wave * 64 + 2 * lane] = h0;
        as[FT * LDA + wave * 64 + 2 * lane + 1] = h1;
    }
    __syncthreads();

    // unfold y = [v2, -v2_r, -v1_r, -v1], window, overlap-add, scale, centre crop
    constexpr int Q = M / 2;
    const float scale = 4.0f / (2 * M);
    for (int i = tid; i < FT * M; i += 256) {
        const int j = i / M, n = i % M;
        const int h = f0 + j;
        if (h < 1 || h > F - 1) continue;
        const float* vc = as + j * LDA;                               // frame h
        const float* vp = as + ((j == 0) ? FT : j - 1) * LDA;         // frame h-1
        const float yc = (n < Q) ? vc[Q + n] : -vc[3 * Q - 1 - n];     // y_h[n]
        const float yp = (n < Q) ? -vp[Q - 1 - n] : -vp[n - Q];        // y_{h-1}[n + M]
        const int t = (h - 1) * M + n;
        if (t >= out_len) continue;
        const OutT o = (OutT)(scale * (ws[n] * yc + ws[n + M] * yp));
        if (sa.pitch == 0) { audio[(size_t)b * out_len + t] = o; continue; }
        // the stitched waveform (see imdct4_ct_kernel): halved and added inside the cross-fade zones, stored elsewhere
        const long long gi = sa.base + (long long)b * sa.pitch + t;
        if (gi < 0 || gi >= sa.total) continue;
        if (t < sa.overlap || t >= out_len - sa.overlap) unsafeAtomicAdd(audio + gi, (OutT)0.5 * o);
        else audio[gi] = o;
    }
    if (frames_out) {   // windowed synthesis frames [B, F, 2M] (return_frames=True), mdct.py:473-475
        for (int i = tid; i < FT * 2 * M; i += 256) {
            const int j = i / (2 * M), n = i % (2 * M);
            if (f0 + j >= F) continue;
            const float* v = as + j * LDA;
            float y;
            if (n < Q) y = v[Q + n];
            else if (n < 2 * Q) y = -v[3 * Q - 1 - n];
            else if (n < 3 * Q) y = -v[3 * Q - 1 - n];
            else y = -v[n - 3 * Q];
            frames_out[((size_t)b * F + f0 + j) * (2 * M) + n] = y * ws[n];
        }
    }
}

// generate_audio.py:40-53: segments [n_seg, L] -> one waveform.  overlap == 0: concatenation.  overlap > 0: the first
// and last `overlap` samples of EVERY segment are halved, segments are overlap-added at stride L - overlap
// (F.fold), and `overlap` samples are cropped from both ends.  Gather form: each output sample sums the (<= ceil(L /
// stride)) segments that cover it in ascending segment order -- deterministic, no atomics.
template <typename T>
__global__ void stitch_kernel(const T* __restrict__ seg, int n_seg, int L, int overlap, T* __restrict__ out,
                              long long out_len) {
    const int stride = L - overlap;
    for (long long o = (long long)blockIdx.x * blockDim.x + threadIdx.x; o < out_len;
         o += (long long)gridDim.x * blockDim.x) {
        if (overlap == 0) {
            out[o] = seg[o];
            continue;
        }
        const long long t = o + overlap;
        long long i0 = (t - L + stride) / stride;      // ceil((t - L + 1) / stride)
        if (t - L + 1 <= 0) i0 = 0;
        long long i1 = t / stride;
        if (i1 > n_seg - 1) i1 = n_seg - 1;
        T acc = (T)0;
        for (long long i = i0; i <= i1; ++i) {
            const int k = (int)(t - i * stride);
            const T v = seg[i * L + k];
            acc += (k < overlap || k >= L - overlap) ? v * (T)0.5 : v;
        }
        out[o] = acc;
    }
}

}  // namespace

#include "mdct_codec.h"
#include "mdct_ct.h"

// frames per workgroup: a whole-clip tile (4 MFMAs per B fetch) once the launch fills the chip, 32 otherwise
static int frames_per_wg(int B, int F) {
    if (const char* f = getenv("MG_MDCT_FT")) { const int v = atoi(f); return (v == 128 || v == 64) ? v : 32; }
    (void)B; (void)F;
    return 32;      // measured on MI355X at 4096 clips: K1 1.99 / 1.87 / 2.95 ms and K2 1.19 / 1.54 / 2.49 ms for 32 / 64 / 128
}
// which kernel the last mg_mdct4_forward [0] / mg_imdct4_* [1] call launched (mg_mdct_last_kernel: bench.py names it in `roofline`)
static const char* g_last_kernel[2] = {"", ""};
template <typename K>
static void allow_lds(K kernel, size_t lds) {
    hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
}
// ---------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------
extern "C" {

// See include/mdctgan_hip.h for the contract.
int mg_mdct4_forward(const float* audio, int B, int T, int n_fft, const float* window, const float* dct4,
                     const float* dct4_image, int codec, float gain, float nr0, float nr1, float src_min, float src_max, int per_sample,
                     float* spec, float* in2, float* frames_out, float* min_out, float* max_out,
                     double* stats, unsigned* scratch_u32, void* stream) {
    if (!audio || !window || !dct4 || (!spec && !in2) || B <= 0 || T <= 0) return MG_ERR_ARG;
    if (n_fft != 2 * M) return MG_ERR_UNSUPPORTED;
    if (per_sample && (!scratch_u32 || !min_out || !max_out || codec == CODEC_RAW)) return MG_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    const int pad_tail = (T % M) ? (M - T % M) : 0;
    const int F = (T + 2 * M + pad_tail - 2 * M) / M + 1;
    CodecParams cp{codec, gain, nr0, nr1, src_min, src_max, nullptr, nullptr, per_sample};
    static bool attr_done = false;
    if (!attr_done) {
        allow_lds(mdct4_kernel<128>, (128 * LDA + 2 * M) * sizeof(float));
        allow_lds(mdct4_kernel<64>, (64 * LDA + 2 * M) * sizeof(float));
        attr_done = true;
    }
    if (stats) hipMemsetAsync(stats, 0, 2 * sizeof(double), st);
    if (per_sample) hipLaunchKernelGGL(fill_u32_pairs, dim3((B + 255) / 256), dim3(256), 0, st, scratch_u32, B);
    // The factored transform (mdct_ct.h: two small dense stages on the f32 pipe instead of the 256 x 256 table, two 8-wave workgroups
    // per CU) wherever its fast path applies -- measured faster than every earlier kernel at every size (6.5 vs 12.2 us at 8 clips,
    // 7.9 vs 16.3 at 64, 260 vs 446 at 4096; the table-stationary f32 / bf16 x 3 kernels of rounds 3-4 live on under scripts/ubench/).
    // It reads the stage-matrix image (mg_dct4_image): without one, for per-sample ranges, returned frames, unaligned or > 4 GiB
    // operands the generic kernel below runs.  MG_MDCT_CT=0 forces the generic kernel.
    const bool legacy_forced = !dct4_image || getenv("MG_MDCT_FT");
    auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    const long long n_rows = (long long)B * F;
    {
        const long long n_tiles = (n_rows + CT_ROWS - 1) / CT_ROWS;
        bool ct = true;
        if (const char* e = getenv("MG_MDCT_CT")) ct = atoi(e) != 0;
        if (ct && !legacy_forced && T % 4 == 0 && !per_sample && !frames_out && (codec == CODEC_RAW || codec == CODEC_ARCSINH) &&
            !(codec == CODEC_RAW && in2) && n_rows * M * 8 < (1ll << 32) - (1ll << 18) && (long long)B * T * 4 < (1ll << 32) &&
            al16(audio) && al16(window) && al16(dct4_image) && (!spec || al16(spec)) && (!in2 || al16(in2)) && (spec || in2)) {
            const float* img = dct4_image;
            const dim3 grid((unsigned)(n_tiles < 512 ? n_tiles : 512)), block(CT_NT);
#define MG_K1_CT(MODE_, SPEC_, PAIR_, STATS_)                                                                             \
    do {                                                                                                                   \
        static bool attr = false;                                                                                          \
        if (!attr) { allow_lds(mdct4_ct_kernel<MODE_, SPEC_, PAIR_, STATS_>, CT_K1_LDS); attr = true; }                    \
        hipLaunchKernelGGL((mdct4_ct_kernel<MODE_, SPEC_, PAIR_, STATS_>), grid, block, CT_K1_LDS, st, audio, B, T, F, window, img, cp, \
                           spec, in2, stats);                                                                              \
    } while (0)
#define MG_K1_CT_S(MODE_, SPEC_, PAIR_) do { if (stats) MG_K1_CT(MODE_, SPEC_, PAIR_, true); else MG_K1_CT(MODE_, SPEC_, PAIR_, false); } while (0)
            if (codec == CODEC_RAW) MG_K1_CT(CODEC_RAW, true, false, false);
            else if (in2 && spec) MG_K1_CT_S(CODEC_ARCSINH, true, true);
            else if (in2) MG_K1_CT_S(CODEC_ARCSINH, false, true);
            else MG_K1_CT_S(CODEC_ARCSINH, true, false);
#undef MG_K1_CT_S
#undef MG_K1_CT
            MG_CHECK_LAUNCH();
            g_last_kernel[0] = "mdct4_ct_kernel (csrc/mdct_ct.h)";
            return MG_OK;
        }
    }
    if (!spec) return MG_ERR_ARG;          // (only the factored kernel writes the pair alone)
    const int ft = frames_per_wg(B, F);
    if (ft == 128)
        hipLaunchKernelGGL(mdct4_kernel<128>, dim3((F + 127) / 128, B), dim3(256), (128 * LDA + 2 * M) * sizeof(float), st,
                           audio, T, F, window, dct4, cp, spec, in2, frames_out, per_sample, scratch_u32, stats);
    else if (ft == 64)
        hipLaunchKernelGGL(mdct4_kernel<64>, dim3((F + 63) / 64, B), dim3(256), (64 * LDA + 2 * M) * sizeof(float), st,
                           audio, T, F, window, dct4, cp, spec, in2, frames_out, per_sample, scratch_u32, stats);
    else
        hipLaunchKernelGGL(mdct4_kernel<32>, dim3((F + 31) / 32, B), dim3(256), (32 * LDA + 2 * M) * sizeof(float), st,
                           audio, T, F, window, dct4, cp, spec, in2, frames_out, per_sample, scratch_u32, stats);
    MG_CHECK_LAUNCH();
    g_last_kernel[0] = "mdct4_kernel (csrc/mdct.hip)";
    if (per_sample) {
        hipLaunchKernelGGL(range_norm_kernel, dim3(64, B), dim3(256), 0, st, spec, in2, F * M, scratch_u32, nr0,
                           nr1, min_out, max_out);
        MG_CHECK_LAUNCH();
    }
    return MG_OK;
}

long long mg_dct4_image_floats(int n_fft) {
    if (n_fft != 2 * M) return 0;
    return CT_IMG;      // the stage matrices of the factored transform, in the lane order the kernels load them
}

int mg_dct4_image(const float* dct4, float* image, void* stream) {
    if (!dct4 || !image || (reinterpret_cast<uintptr_t>(dct4) & 15) || (reinterpret_cast<uintptr_t>(image) & 15)) return MG_ERR_ARG;
    hipLaunchKernelGGL(dct4_ct_image_kernel, dim3(CT_IMG / 256), dim3(256), 0, (hipStream_t)stream, image);
    MG_CHECK_LAUNCH();
    return MG_OK;
}

int mg_mdct4_num_frames(int T, int n_fft) {
    const int m = n_fft / 2;
    const int pad_tail = (T % m) ? (m - T % m) : 0;
    return (T + pad_tail) / m + 1;
}

static int imdct4_dispatch(const float* spec, int B, int F, int n_fft, const float* window, const float* dct4,
                           const float* dct4_image, int codec, float gain, float nr0, float nr1, float src_min, float src_max,
                           const float* min_b, const float* max_b, void* audio, int out_len, int out_f64,
                           float* frames_out, StitchArgs sa, void* stream) {
    if (!spec || !window || !dct4 || !audio || B <= 0 || F <= 0) return MG_ERR_ARG;
    if (n_fft != 2 * M) return MG_ERR_UNSUPPORTED;
    if (out_len <= 0 || out_len > (F - 1) * M) return MG_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    CodecParams cp{codec, gain, nr0, nr1, src_min, src_max, min_b, max_b, (min_b && max_b) ? 1 : 0};
    static bool attr_done = false;
    if (!attr_done) {
        allow_lds(imdct4_kernel<float, 128>, (129 * LDA + 2 * M) * sizeof(float));
        allow_lds(imdct4_kernel<double, 128>, (129 * LDA + 2 * M) * sizeof(float));
        allow_lds(imdct4_kernel<float, 64>, (65 * LDA + 2 * M) * sizeof(float));
        allow_lds(imdct4_kernel<double, 64>, (65 * LDA + 2 * M) * sizeof(float));
        attr_done = true;
    }
    {
        auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
        // round 4 (second half): the factored transform (mdct_ct.h); a workgroup walks whole clips: >= 512 clips fill two per CU
        // Measured (scripts/ubench/mdct_b3_bench, profiles/r04_mdct_ct_ubench.log): 249 vs 442 us at 4096 clips, 18.3 vs 27.9 at 128,
        // 17.6 vs 17.0 at 64, 16.7 vs 15.5 at 8 against the f32-pipe / bf16 x 3 kernels (a workgroup's tiles of a clip are sequential:
        // latency, not work, below ~96 clips -- a microsecond there buys one kernel for every size, so that the stitched store,
        // which exists in this kernel and the generic one only, gives the bits of the plain store followed by mg_stitch_segments).
        const bool st4 = sa.pitch != 0 && sa.pitch % 4 == 0 && sa.overlap % 4 == 0 && sa.total * 4 < (1ll << 32) - (1ll << 16);
        bool ct = true;
        if (const char* e = getenv("MG_MDCT_CT")) ct = atoi(e) != 0 || st4;
        if (ct && (sa.pitch == 0 || st4) && dct4_image && !getenv("MG_MDCT_FT") && !frames_out && !out_f64 && out_len % 4 == 0 && al16(spec) && al16(window) &&
            al16(dct4_image) && al16(audio) && (long long)B * F * M * 4 < (1ll << 32) && (long long)B * out_len * 4 < (1ll << 32) - (1ll << 16) &&
            codec >= CODEC_RAW && codec <= CODEC_RANGE) {
            const float* img = dct4_image;
            const dim3 grid((unsigned)(B < 512 ? B : 512));
#define MG_K2_CT(MODE_, ST_)                                                                                              \
    do {                                                                                                                   \
        static bool attr = false;                                                                                          \
        if (!attr) { allow_lds(imdct4_ct_kernel<MODE_, ST_>, CT_K2_LDS); attr = true; }                                    \
        hipLaunchKernelGGL((imdct4_ct_kernel<MODE_, ST_>), grid, dim3(CT_NT), CT_K2_LDS, st, spec, B, F, window, img, cp, (float*)audio, out_len, sa); \
    } while (0)
#define MG_K2_CT_S(MODE_) do { if (st4) MG_K2_CT(MODE_, true); else MG_K2_CT(MODE_, false); } while (0)
            if (codec == CODEC_RAW) MG_K2_CT_S(CODEC_RAW); else if (codec == CODEC_ARCSINH) MG_K2_CT_S(CODEC_ARCSINH); else MG_K2_CT_S(CODEC_RANGE);
#undef MG_K2_CT_S
#undef MG_K2_CT
            MG_CHECK_LAUNCH();
            g_last_kernel[1] = st4 ? "imdct4_ct_kernel<stitched> (csrc/mdct_ct.h)" : "imdct4_ct_kernel (csrc/mdct_ct.h)";
            return MG_OK;
        }
    }
#define MG_IMDCT(T_, FT_)                                                                                            \
    hipLaunchKernelGGL((imdct4_kernel<T_, FT_>), dim3((F + FT_ - 1) / FT_, B), dim3(256),                            \
                       ((FT_ + 1) * LDA + 2 * M) * sizeof(float), st, spec, F, window, dct4, cp, (T_*)audio, out_len, \
                       frames_out, sa)
    const int ft = frames_per_wg(B, F);
    if (ft == 128) {
        if (out_f64) MG_IMDCT(double, 128); else MG_IMDCT(float, 128);
    } else if (ft == 64) {
        if (out_f64) MG_IMDCT(double, 64); else MG_IMDCT(float, 64);
    } else {
        if (out_f64) MG_IMDCT(double, 32); else MG_IMDCT(float, 32);
    }
#undef MG_IMDCT
    MG_CHECK_LAUNCH();
    g_last_kernel[1] = "imdct4_kernel (csrc/mdct.hip)";
    return MG_OK;
}

int mg_imdct4_forward(const float* spec, int B, int F, int n_fft, const float* window, const float* dct4,
                      const float* dct4_image, int codec, float gain, float nr0, float nr1, float src_min, float src_max,
                      const float* min_b, const float* max_b, void* audio, int out_len, int out_f64,
                      float* frames_out, void* stream) {
    return imdct4_dispatch(spec, B, F, n_fft, window, dct4, dct4_image, codec, gain, nr0, nr1, src_min, src_max, min_b, max_b, audio,
                           out_len, out_f64, frames_out, StitchArgs{0, 0, 0, 0}, stream);
}

// See include/mdctgan_hip.h: K2 with generate_audio.py:40-53 folded into its overlap-add store.
int mg_imdct4_stitched(const float* spec, int B, int F, int n_fft, const float* window, const float* dct4,
                       const float* dct4_image, int codec, float gain, float nr0, float nr1, float src_min, float src_max,
                       const float* min_b, const float* max_b, void* out, long long out_total, int seg_len, int overlap,
                       long long first_seg, int zero_out, int out_f64, void* stream) {
    if (!out || seg_len <= 0 || overlap < 0 || 2 * overlap >= seg_len || first_seg < 0 || out_total <= 0) return MG_ERR_ARG;
    if (zero_out && overlap > 0) hipMemsetAsync(out, 0, (size_t)out_total * (out_f64 ? 8 : 4), (hipStream_t)stream);
    const int pitch = seg_len - overlap;
    return imdct4_dispatch(spec, B, F, n_fft, window, dct4, dct4_image, codec, gain, nr0, nr1, src_min, src_max, min_b, max_b, out,
                           seg_len, out_f64, nullptr, StitchArgs{first_seg * pitch - overlap, out_total, pitch, overlap}, stream);
}

const char* mg_mdct_last_kernel(int which) { return g_last_kernel[which == 1 ? 1 : 0]; }

long long mg_stitch_length(int n_seg, int seg_len, int overlap) {
    if (n_seg <= 0 || seg_len <= 0 || overlap < 0 || 2 * overlap >= seg_len) return -1;
    if (overlap == 0) return (long long)n_seg * seg_len;
    return (long long)(n_seg - 1) * (seg_len - overlap) + seg_len - 2LL * overlap;
}

int mg_stitch_segments(const void* seg, int n_seg, int seg_len, int overlap, void* out, int is_f64, void* stream) {
    const long long n = mg_stitch_length(n_seg, seg_len, overlap);
    if (!seg || !out || n <= 0) return MG_ERR_ARG;
    const unsigned blocks = (unsigned)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256);
    if (is_f64)
        hipLaunchKernelGGL(stitch_kernel<double>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const double*)seg,
                           n_seg, seg_len, overlap, (double*)out, n);
    else
        hipLaunchKernelGGL(stitch_kernel<float>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const float*)seg,
                           n_seg, seg_len, overlap, (float*)out, n);
    MG_CHECK_LAUNCH();
    return MG_OK;
}

}  // extern "C"
