// K1 / K2 as table-stationary contractions (round 3).  Included by mdct.hip (needs M, CodecParams, Codec, LN10F).
//
// The 256-point DCT-IV is X = U * D4 with ONE 256 x 256 table for every frame of every clip.  256 KB of float32 do not fit
// the LDS, but they fit the register file of a CU: a workgroup of 8 waves holds the whole table in VGPRs -- wave w keeps
// the 32 output bins [32w, 32w + 32) for all 256 k as 32 float4 (128 registers) -- and streams 32-frame row tiles past it.
// Per tile a wave issues 128 v_mfma_f32_32x32x2_f32 whose B operand is already in registers and whose A operand is one
// conflict-free ds_read_b128 per four MFMAs; nothing of the table is re-read from L2 after the prologue, there is no
// per-chunk barrier (one barrier per 32-frame tile), and the MFMA stream is as lean as a stream can be.
//
// k order: lane (row / bin = lane & 31, kh = lane >> 5) of MFMA step (jj, t), jj = 0..31, t = 0..3, contracts
// k = 8 jj + 4 kh + t -- any permutation of k is a valid MFMA schedule as long as A and B agree -- so both operands are
// 16-byte vectors of 4 consecutive k: the table row of a bin (D4 is symmetric, row n == column n) and the folded frame row.
//
// K1 (mdct4_bs_kernel<NW>): grid (workers, 8 / NW).  A workgroup of NW waves covers NW * 32 bins of every tile it visits
// (NW = 8: all 256 bins, one workgroup per CU, the throughput shape; NW = 2: 64 bins, four workgroups share a row tile and
// each rebuilds the folded frames -- the latency shape for a handful of clips).  Tiles are 32 consecutive rows of the
// [B * F, 256] frame matrix, double-buffered in LDS: the signal loads of tile i + 1 are issued before the MFMA loop of
// tile i and folded (window, TDAC fold, float32 like mdct.py:410) into the other buffer after its epilogue.
// K2 (imdct4_bs_kernel): 8 waves, tile = 32 frames of one clip + the halo frame f0 - 1 as a VALU dot product against the
// same registers (each lane holds half of the k of its bin; the halves meet in one DPP exchange).
#pragma once

namespace {

constexpr int BS_ROWS = 32;             // frames per tile (one 32 x 32 MFMA block per wave)
constexpr int BS_LDA = M + 4;           // LDS row pitch in floats: 65 16-byte slots -> ds_read_b128 of 16 rows hits 16 bank groups
constexpr float LOG10_2F = 0.30102999566398120f;
constexpr float INV_LN10F = 0.43429448190325176f;

// asinh(y) / ln(10) in ~14 issue slots (the libm asinhf + two IEEE divisions of the round-2 epilogue were ~190 VALU
// instructions per value: at 64 values per lane they cost more than the tile's MFMAs).  |y| >= 1/8: log2(|y| + sqrt(y^2 + 1))
// on v_sqrt_f32 / v_log_f32 (1 ulp each; <= 5e-7 relative in the result); |y| < 1/8: the odd series through y^7
// (truncation 2e-9 relative).  Against the reference's float32 torch.arcsinh: a few ulp, 1000x inside the 5e-4 bar of the
// normalised spectrogram (tests/test_mdct_gpu.py::test_fast_codec_math).
__device__ __forceinline__ float asinh_over_ln10(float y) {
    const float a = fabsf(y), a2 = a * a;
    const float s = a + __builtin_amdgcn_sqrtf(a2 + 1.0f);
    const float big = __builtin_amdgcn_logf(s) * LOG10_2F;
    const float p = fmaf(a2, fmaf(a2, fmaf(a2, -0.044642857142857144f, 0.075f), -0.16666666666666666f), 1.0f);
    const float small = a * p * INV_LN10F;
    return copysignf(a < 0.125f ? small : big, y);
}
// sinh(x) for the decoder: (e^x - e^-x) / 2 on v_exp_f32 above 1/4 (relative 3e-7), odd series through x^7 below.
__device__ __forceinline__ float sinh_fast(float x) {
    const float a = fabsf(x), a2 = a * a;
    const float e = __builtin_amdgcn_exp2f(a * 1.4426950408889634f);
    const float big = 0.5f * (e - __builtin_amdgcn_rcpf(e));
    const float p = fmaf(a2, fmaf(a2, fmaf(a2, 1.984126984126984e-4f, 8.333333333333333e-3f), 0.16666666666666666f), 1.0f);
    return copysignf(a < 0.25f ? a * p : big, x);
}
// x / d for a constant d, rd = fl(1 / d): one Newton step on the residual -- the correctly rounded quotient except in rare
// double-rounding cases, three instructions instead of v_div_scale / v_div_fmas / v_div_fixup.
__device__ __forceinline__ float div_const(float x, float d, float rd) {
    const float q = x * rd;
    return fmaf(fmaf(-q, d, x), rd, q);
}

struct BsCodec {      // CodecParams with the constants of the fixed-range normalisation folded once per kernel
    int mode;
    float gain, nr0, span, mn, d, rd;       // span = nr1 - nr0, d = mx - mn
};
__device__ __forceinline__ BsCodec bs_codec(const CodecParams& cp) {
    BsCodec c;
    c.mode = cp.mode; c.gain = cp.gain; c.nr0 = cp.nr0; c.span = cp.nr1 - cp.nr0; c.mn = cp.mn; c.d = cp.mx - cp.mn;
    c.rd = 1.0f / c.d;
    return c;
}
// l = log-domain value (for the statistics), returns the normalised value
__device__ __forceinline__ float bs_encode(float xv, const BsCodec& c, float& l) {
    if (c.mode == CODEC_RAW) { l = xv; return xv; }
    l = (c.mode == CODEC_ARCSINH) ? asinh_over_ln10(c.gain * xv) : xv;
    return div_const(l - c.mn, c.d, c.rd) * c.span + c.nr0;
}

__device__ __forceinline__ float4 bs_ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float4 bs_rev4(const float4 v) { return make_float4(v.w, v.z, v.y, v.x); }

// ------------------------------------------------------------------------------------------------------------------
// K1
// ------------------------------------------------------------------------------------------------------------------
template <int NW>
__global__ __launch_bounds__(NW * 64) void mdct4_bs_kernel(const float* __restrict__ audio, int B, int T, int F,
                                                           const float* __restrict__ window, const float* __restrict__ dct4,
                                                           CodecParams cp, float* __restrict__ spec, float* __restrict__ in2,
                                                           double* __restrict__ stats) {
    constexpr int NT = NW * 64, Q = M / 2;
    constexpr int GROUPS = BS_ROWS * (M / 4) / NT;        // float4 groups of the folded tile per thread
    extern __shared__ __attribute__((aligned(16))) float bs_smem[];
    float* ws = bs_smem;                                  // [2 M] window
    float* abuf = bs_smem + 2 * M;                        // [2][BS_ROWS][BS_LDA]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int rows = B * F, n_tiles = (rows + BS_ROWS - 1) / BS_ROWS;
    const int col = (blockIdx.y * NW + wave) * 32 + (lane & 31), kh = lane >> 5;
    const BsCodec cd = bs_codec(cp);

    // the wave's slab of the table: 32 float4 = bins [col] x k = 8 jj + 4 kh + (0..3)
    float4 bt[32];
#pragma unroll
    for (int jj = 0; jj < 32; ++jj) bt[jj] = bs_ld4(dct4 + (size_t)col * M + 8 * jj + 4 * kh);
    for (int i = tid; i < 2 * M / 4; i += NT) reinterpret_cast<float4*>(ws)[i] = bs_ld4(window + 4 * i);

    // group gi of a thread: row r = (tid + gi * NT) / 64 of the tile, u[n .. n + 3] with n = 4 * ((tid + gi * NT) % 64)
    float4 x1[GROUPS], x2[GROUPS];
    auto offsets = [&](int n, int& o1, int& o2) {
        if (n < Q) { o1 = 3 * Q - 4 - n; o2 = 3 * Q + n; }        // u = -rev(z[o1..]) - z[o2..]
        else { o1 = n - Q; o2 = 3 * Q - 4 - n; }                   // u =  z[o1..] - rev(z[o2..])
    };
    auto load_tile = [&](int tile) {
#pragma unroll
        for (int gi = 0; gi < GROUPS; ++gi) {
            const int i = tid + gi * NT, r = i >> 6, n = 4 * (i & 63);
            const int m = tile * BS_ROWS + r;
            int o1, o2;
            offsets(n, o1, o2);
            x1[gi] = x2[gi] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (m < rows) {
                const int b = m / F, f = m - b * F;
                const float* x = audio + (size_t)b * T;
                const int t1 = f * M - M + o1, t2 = f * M - M + o2;     // T % 4 == 0: a float4 is inside or outside as a whole
                if (t1 >= 0 && t1 + 3 < T) x1[gi] = bs_ld4(x + t1);
                if (t2 >= 0 && t2 + 3 < T) x2[gi] = bs_ld4(x + t2);
            }
        }
    };
    auto fold_tile = [&](int buf) {
#pragma unroll
        for (int gi = 0; gi < GROUPS; ++gi) {
            const int i = tid + gi * NT, r = i >> 6, n = 4 * (i & 63);
            int o1, o2;
            offsets(n, o1, o2);
            const float4 w1 = bs_ld4(ws + o1), w2 = bs_ld4(ws + o2);
            const float4 a = x1[gi], c = x2[gi];
            // z = fl32(x * w) (mdct.py:410), then the TDAC fold
            const float4 z1 = make_float4(__fmul_rn(a.x, w1.x), __fmul_rn(a.y, w1.y), __fmul_rn(a.z, w1.z), __fmul_rn(a.w, w1.w));
            const float4 z2 = make_float4(__fmul_rn(c.x, w2.x), __fmul_rn(c.y, w2.y), __fmul_rn(c.z, w2.z), __fmul_rn(c.w, w2.w));
            float4 u;
            if (n < Q) {
                const float4 r1 = bs_rev4(z1);
                u = make_float4(-r1.x - z2.x, -r1.y - z2.y, -r1.z - z2.z, -r1.w - z2.w);
            } else {
                const float4 r2 = bs_rev4(z2);
                u = make_float4(z1.x - r2.x, z1.y - r2.y, z1.z - r2.z, z1.w - r2.w);
            }
            *reinterpret_cast<float4*>(abuf + (size_t)buf * BS_ROWS * BS_LDA + r * BS_LDA + n) = u;
        }
    };

    double s1 = 0.0, s2 = 0.0;
    int tile = blockIdx.x;
    if (tile < n_tiles) load_tile(tile);
    __syncthreads();                              // window in LDS
    if (tile < n_tiles) fold_tile(0);
    __syncthreads();
    int buf = 0;
    for (; tile < n_tiles; tile += gridDim.x, buf ^= 1) {
        const int next = tile + gridDim.x;
        if (next < n_tiles) load_tile(next);      // global loads in flight under the MFMA stream
        f32x16 acc = f32x16{0};
        const float* ap = abuf + (size_t)buf * BS_ROWS * BS_LDA + (lane & 31) * BS_LDA + 4 * kh;
#pragma unroll
        for (int jj = 0; jj < 32; ++jj) {
            const float4 a = bs_ld4(ap + 8 * jj);
            acc = mfma32x32x2(a.x, bt[jj].x, acc);
            acc = mfma32x32x2(a.y, bt[jj].y, acc);
            acc = mfma32x32x2(a.z, bt[jj].z, acc);
            acc = mfma32x32x2(a.w, bt[jj].w, acc);
        }
        // epilogue: codec + stores (lane: bin col, 16 frames)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = tile * BS_ROWS + mfma32_row(r, lane);
            if (m >= rows) continue;
            float l;
            const float v = bs_encode(acc[r], cd, l);
            if (stats && cd.mode != CODEC_RAW) { s1 += (double)l; s2 += (double)l * (double)l; }
            const size_t o = (size_t)m * M + col;
            spec[o] = v;
            if (in2) *reinterpret_cast<float2*>(in2 + 2 * o) = make_float2(v, fabsf(v) * 2.0f + cd.nr0);
        }
        if (next < n_tiles) fold_tile(buf ^ 1);
        __syncthreads();
    }
    if (stats && cd.mode != CODEC_RAW) {
        s1 = wave_sum_d(s1); s2 = wave_sum_d(s2);
        if (lane == 0) { atomicAdd(stats, s1); atomicAdd(stats + 1, s2); }
    }
}

constexpr size_t BS_K1_LDS = (size_t)(2 * M + 2 * BS_ROWS * BS_LDA) * sizeof(float);

// ------------------------------------------------------------------------------------------------------------------
// K2.  grid = workers, block = 512.  Tile = (clip b, frames f0 .. f0 + 31); emits hop blocks h = f0 .. f0 + 31:
//   out[(h - 1) M + n] = 4 / N * (w[n] y_h[n] + w[n + M] y_{h-1}[n + M]),  y = [v2, -v2_r, -v1_r, -v1] of v = DCT-IV(X).
// ------------------------------------------------------------------------------------------------------------------
template <typename OutT>
__global__ __launch_bounds__(512) void imdct4_bs_kernel(const float* __restrict__ spec, int B, int F,
                                                        const float* __restrict__ window, const float* __restrict__ dct4,
                                                        CodecParams cp, OutT* __restrict__ audio, int out_len) {
    constexpr int NT = 512, Q = M / 2;
    constexpr int GROUPS = (BS_ROWS + 1) * (M / 4) / NT + 1;     // 33 rows x 64 float4 groups over 512 threads: 4 full + 1 partial
    extern __shared__ __attribute__((aligned(16))) float bs_smem[];
    float* ws = bs_smem;                                  // [2 M]
    float* abuf = bs_smem + 2 * M;                        // [2][BS_ROWS + 1][BS_LDA]: row 32 = halo frame f0 - 1
    constexpr int TILE_F = (BS_ROWS + 1) * BS_LDA;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tiles_per_clip = (F + BS_ROWS - 1) / BS_ROWS, n_tiles = B * tiles_per_clip;
    const int col = wave * 32 + (lane & 31), kh = lane >> 5;

    float4 bt[32];
#pragma unroll
    for (int jj = 0; jj < 32; ++jj) bt[jj] = bs_ld4(dct4 + (size_t)col * M + 8 * jj + 4 * kh);
    for (int i = tid; i < 2 * M / 4; i += NT) reinterpret_cast<float4*>(ws)[i] = bs_ld4(window + 4 * i);

    float4 xr[GROUPS];
    auto load_tile = [&](int tile) {
        const int b = tile / tiles_per_clip, f0 = (tile - b * tiles_per_clip) * BS_ROWS;
#pragma unroll
        for (int gi = 0; gi < GROUPS; ++gi) {
            const int i = tid + gi * NT, j = i >> 6, k = 4 * (i & 63);
            xr[gi] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (j <= BS_ROWS) {
                const int f = (j == BS_ROWS) ? f0 - 1 : f0 + j;
                if (f >= 0 && f < F) xr[gi] = bs_ld4(spec + ((size_t)b * F + f) * M + k);
            }
        }
    };
    const float span = cp.nr1 - cp.nr0, rspan = 1.0f / span, rgain = 1.0f / cp.gain;
    auto dec1 = [&](float v, float mn, float d) -> float {      // Audio2MDCT.denormalize, pix2pixHD_model.py:127-137
        if (cp.mode == CODEC_RAW) return v;
        const float l = div_const(v - cp.nr0, span, rspan) * d + mn;
        if (cp.mode == CODEC_ARCSINH) return div_const(sinh_fast(l * LN10F), cp.gain, rgain);
        return l;
    };
    auto decode_tile = [&](int tile, int buf) {
        const int b = tile / tiles_per_clip, f0 = (tile - b * tiles_per_clip) * BS_ROWS;
        float mn = cp.mn, mx = cp.mx;
        if (cp.per_sample) { mn = cp.mn_b[b]; mx = cp.mx_b[b]; }
        const float d = mx - mn;
#pragma unroll
        for (int gi = 0; gi < GROUPS; ++gi) {
            const int i = tid + gi * NT, j = i >> 6, k = 4 * (i & 63);
            if (j > BS_ROWS) continue;
            const int f = (j == BS_ROWS) ? f0 - 1 : f0 + j;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);       // frames outside the clip contribute nothing (decode(0) != 0)
            if (f >= 0 && f < F) v = make_float4(dec1(xr[gi].x, mn, d), dec1(xr[gi].y, mn, d), dec1(xr[gi].z, mn, d), dec1(xr[gi].w, mn, d));
            *reinterpret_cast<float4*>(abuf + (size_t)buf * TILE_F + j * BS_LDA + k) = v;
        }
    };

    int tile = blockIdx.x;
    if (tile < n_tiles) load_tile(tile);
    __syncthreads();
    if (tile < n_tiles) decode_tile(tile, 0);
    __syncthreads();
    int buf = 0;
    const float scale = 4.0f / (2 * M);
    for (; tile < n_tiles; tile += gridDim.x, buf ^= 1) {
        const int next = tile + gridDim.x;
        if (next < n_tiles) load_tile(next);
        float* at = abuf + (size_t)buf * TILE_F;
        f32x16 acc = f32x16{0};
        float4 h4 = make_float4(0.f, 0.f, 0.f, 0.f);       // halo frame: this lane's half of the k of its bin, four partial sums
        const float* ap = at + (lane & 31) * BS_LDA + 4 * kh;
        const float* hp = at + BS_ROWS * BS_LDA + 4 * kh;
#pragma unroll
        for (int jj = 0; jj < 32; ++jj) {
            const float4 a = bs_ld4(ap + 8 * jj);
            const float4 xh = bs_ld4(hp + 8 * jj);
            acc = mfma32x32x2(a.x, bt[jj].x, acc);
            acc = mfma32x32x2(a.y, bt[jj].y, acc);
            acc = mfma32x32x2(a.z, bt[jj].z, acc);
            acc = mfma32x32x2(a.w, bt[jj].w, acc);
            h4.x = fmaf(xh.x, bt[jj].x, h4.x); h4.y = fmaf(xh.y, bt[jj].y, h4.y);
            h4.z = fmaf(xh.z, bt[jj].z, h4.z); h4.w = fmaf(xh.w, bt[jj].w, h4.w);
        }
        float h = (h4.x + h4.y) + (h4.z + h4.w);
        h += __shfl_xor(h, 32, 64);
        __syncthreads();                           // every wave is done reading the tile as the A operand
        // v = DCT-IV(X) back into the same buffer: rows 0..31 frames, row 32 halo
#pragma unroll
        for (int r = 0; r < 16; ++r) at[mfma32_row(r, lane) * BS_LDA + col] = acc[r];
        if (lane < 32) at[BS_ROWS * BS_LDA + col] = h;
        __syncthreads();
        // unfold, window, overlap-add, scale, centre crop: 32 hop blocks x 256 samples, float4 per thread
        {
            const int b = tile / tiles_per_clip, f0 = (tile - b * tiles_per_clip) * BS_ROWS;
#pragma unroll
            for (int gi = 0; gi < BS_ROWS * (M / 4) / NT; ++gi) {
                const int i = tid + gi * NT, j = i >> 6, n = 4 * (i & 63);
                const int hh = f0 + j;
                if (hh < 1 || hh > F - 1) continue;
                const float* vc = at + j * BS_LDA;                                   // frame h
                const float* vp = at + ((j == 0) ? BS_ROWS : j - 1) * BS_LDA;        // frame h - 1
                float4 yc, yp;
                if (n < Q) {      // y_h[n] = v_h[Q + n];  y_{h-1}[n + M] = -v_{h-1}[Q - 1 - n]
                    yc = bs_ld4(vc + Q + n);
                    const float4 t = bs_rev4(bs_ld4(vp + Q - 4 - n));
                    yp = make_float4(-t.x, -t.y, -t.z, -t.w);
                } else {          // y_h[n] = -v_h[3Q - 1 - n];  y_{h-1}[n + M] = -v_{h-1}[n - Q]
                    const float4 t = bs_rev4(bs_ld4(vc + 3 * Q - 4 - n));
                    yc = make_float4(-t.x, -t.y, -t.z, -t.w);
                    const float4 u = bs_ld4(vp + n - Q);
                    yp = make_float4(-u.x, -u.y, -u.z, -u.w);
                }
                const float4 w0 = bs_ld4(ws + n), w1 = bs_ld4(ws + n + M);
                const float o0 = scale * (w0.x * yc.x + w1.x * yp.x), o1 = scale * (w0.y * yc.y + w1.y * yp.y);
                const float o2 = scale * (w0.z * yc.z + w1.z * yp.z), o3 = scale * (w0.w * yc.w + w1.w * yp.w);
                const int t0 = (hh - 1) * M + n;
                OutT* dst = audio + (size_t)b * out_len + t0;
                if (t0 + 3 < out_len && (((size_t)b * out_len + t0) & 3) == 0) {
                    if constexpr (sizeof(OutT) == 4) {
                        *reinterpret_cast<float4*>(dst) = make_float4(o0, o1, o2, o3);
                    } else {
                        dst[0] = (OutT)o0; dst[1] = (OutT)o1; dst[2] = (OutT)o2; dst[3] = (OutT)o3;
                    }
                } else {
                    if (t0 < out_len) dst[0] = (OutT)o0;
                    if (t0 + 1 < out_len) dst[1] = (OutT)o1;
                    if (t0 + 2 < out_len) dst[2] = (OutT)o2;
                    if (t0 + 3 < out_len) dst[3] = (OutT)o3;
                }
            }
        }
        if (next < n_tiles) decode_tile(next, buf ^ 1);
        __syncthreads();
    }
}

constexpr size_t BS_K2_LDS = (size_t)(2 * M + 2 * (BS_ROWS + 1) * BS_LDA) * sizeof(float);

}  // namespace
