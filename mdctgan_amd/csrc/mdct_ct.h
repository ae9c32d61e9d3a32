// K1 / K2 with the 256-point DCT-IV FACTORED into two small dense stages on the f32 MFMA pipe (round 4).  Included by mdct.hip
// after mdct_codec.h (needs M, CodecParams, the bs_* helpers).
//
// The dense table costs 65 536 multiply-adds per frame whatever pipe runs them.  The DCT-IV of size N = 256 is an N/2-point complex
// DFT between two twiddles (models/mdct.py:596-628 FastMDCT4 uses the same identity with torch.fft):
//     c[n] = (u[2n] + i u[N-1-2n]) e^{-i pi n / N},   Y = e^{-i pi (k + 1/4) / N} DFT_128(c),   X[2k] = Re Y[k],  X[N-1-2k] = -Im Y[k]
// and the 128-point DFT factors (Cooley-Tukey, n = 16 n1 + n2, k = k1 + 8 k2) into 8-point DFTs over n1, a twiddle, and
// 16-point DFTs over n2.  With every twiddle folded into the stage matrices, a frame is
//     stage A: for each n2 (16 of them)  [a_0..a_7, b_0..b_7] (a_n1 = u[32 n1 + 2 n2], b_n1 = u[255 - 32 n1 - 2 n2])  x  MA_n2 [16 x 16]
//     stage B: for each k1 ( 8 of them)  [Re Y'(n2 = 0..15), Im Y'(n2)]                                              x  MB_k1 [32 x 32]
// = 12 288 multiply-adds per frame (5.3x fewer), in plain float32 MFMAs (v_mfma_f32_16x16x4_f32 / v_mfma_f32_32x32x2_f32) with
// float64-built matrices: against the float64 oracle the result is CLOSER than the dense float32 contraction (two short sums
// instead of one 256-term sum: 2.6e-7 vs 6.8e-7 of max |X| in a numpy model).  48 KB of matrices live in registers (24 per lane).
//
// Workgroup = 8 waves, tile = 32 frames, 79 KB of LDS -> two workgroups per CU (4 waves per SIMD: while one workgroup is in a
// VALU phase the other's MFMAs run; the bf16 x 3 kernels of mdct_b3.h have ONE wave per SIMD and hide ~5 instructions per MFMA).
// Phases of a tile (barrier between them):
//   P1  K1: window + TDAC fold (float32, bit-exact frames as mdct.py:410) / K2: decode (denormalise, sinh) -- scattered into the
//       stage-A operand records  U1[frame][n2][16]   (record position of input kk: 4 (kk % 4) + kk / 4 -> one ds_read_b128 per lane)
//   P2  stage A: wave w owns n2 = 2 w, 2 w + 1; 16 MFMAs 16x16x4; results scattered into  Yb[k1][frame][32]
//   P3  stage B: wave w owns k1 = w; 16 MFMAs 32x32x2; the accumulator column of a lane is bin 2 k1 + 16 j (j < 16) or 255 - 2 k1 - 16 (j - 16)
//   P4  K1: codec in registers; values into the output tile (LDS, aliases U1, float4-swizzled)     K2: v into that tile (+ row 31 into the halo ring)
//   P5  K1: 16-byte stores of whole 1 KiB rows (one contiguous row per wave instruction)           K2: unfold / window / overlap-add / 16-byte stores
#pragma once

namespace {

constexpr int CT_ROWS = 32;
constexpr int CT_NT = 512;
constexpr int CT_REC = 20;                       // floats between the n2 records of a frame (16 used: conflict-free scatter)
constexpr int CT_FS1 = 16 * CT_REC + 8;          // floats between frames in U1 (328 = 8 mod 64: conflict-free 16-lane b128 reads)
constexpr int CT_FS2 = 36;                       // floats between frames in Yb
constexpr int CT_KS2 = CT_ROWS * CT_FS2;         // floats between the k1 blocks of Yb (0 mod 32: the rotation by k1 below spreads the writes)
constexpr int CT_OS = M;                         // floats between rows of the output tile
// LDS layouts: every access of a tile is conflict-free under CDNA4's per-instruction banking (ds_write_b32: two 32-lane groups on
// (a / 4) mod 32; ds_read_b128: four 16-lane groups on (a / 4) mod 64) -- checked with a bank simulator that reproduced the first
// version's SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE exactly (240 conflict cycles of 384 per wave and tile; now 0 of 144).
constexpr int CT_U1 = CT_ROWS * CT_FS1;          // 10 368 floats
constexpr int CT_YB = 8 * CT_KS2;                // 9 248 floats
constexpr int CT_IMG = 512 * 24;                 // floats of the matrix image: 24 per (wave, lane)
static_assert(CT_U1 >= CT_ROWS * CT_OS, "the output tile aliases U1");

typedef float ct_f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ ct_f4 ct_mfma16(float a, float b, ct_f4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// Matrix image (mg_dct4_image): img[(wave * 64 + lane) * 24 + r]
//   r = 4 q + s  (q = 0, 1; s = 0..3):  MA_{n2 = 2 wave + q}[kk = 4 (lane >> 4) + s][j = lane & 15]      (B operand of 16x16x4 step s)
//   r = 8 + s    (s = 0..15):           MB_{k1 = wave}[kk = 2 ((s - wave) & 15) + (lane >> 5)][j = lane & 31]   (B operand of 32x32x2 step s)
// (which input an MFMA step contracts is free as long as both operands agree: the orders follow the LDS layouts below)
// evaluated in double from integer phase numerators (exact argument reduction), rounded once to float32.
__global__ void dct4_ct_image_kernel(float* __restrict__ img) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= CT_IMG) return;
    const int r = i % 24, lane = (i / 24) & 63, wave = i / (24 * 64);
    double v;
    if (r < 8) {
        const int q = r >> 2, s = r & 3, n2 = 2 * wave + q, kk = 4 * (lane >> 4) + s, j = lane & 15;
        const int n1 = kk & 7, k1 = j & 7;
        const bool b_in = kk >= 8, im_out = j >= 8;
        const int t = (4 * n2 * k1 + 64 * n1 * k1 + 16 * n1 + n2) & 511;          // c = exp(-i pi t / 256)
        const double re = cospi(t / 256.0), im = -sinpi(t / 256.0);
        v = !im_out ? (b_in ? -im : re) : (b_in ? re : im);       // Re Y' = Re c a - Im c b;  Im Y' = Im c a + Re c b
    } else {
        const int s = r - 8, k1 = wave, kk = 2 * ((s - wave) & 15) + (lane >> 5), j = lane & 31;
        const int n2 = kk & 15, k2 = j & 15, k = k1 + 8 * k2;
        const bool im_in = kk >= 16, odd_out = j >= 16;
        const int t = (4 * k + 1 + 128 * n2 * k2) & 2047;                          // d = exp(-i pi t / 1024)
        const double re = cospi(t / 1024.0), im = -sinpi(t / 1024.0);
        v = !odd_out ? (im_in ? -im : re) : (im_in ? -re : -im);  // X[2k] = Re d ReY - Im d ImY;  X[255-2k] = -(Im d ReY + Re d ImY)
    }
    img[i] = (float)v;
}

// word offset of u[m] (m = 0..255) inside a frame's U1 block: even m = 32 n1 + 2 n2 is input a_n1 of record n2, odd m is
// input b_n1 of the record of 255 - m.  Input kk of a record sits in float4 (kk >> 2) ^ (n2 >> 3), component kk & 3: the 32 lanes of
// a fold / decode store (8 records x 4 inputs) then cover the 32 banks once.
__device__ __forceinline__ int ct_u1_word(int m) {
    const int e = (m & 1) ? 255 - m : m;
    const int n2 = (e & 31) >> 1, kk = (e >> 5) + ((m & 1) ? 8 : 0);
    return n2 * CT_REC + 4 * ((kk >> 2) ^ (n2 >> 3)) + (kk & 3);
}
// word offset of bin b inside a row of the output tile: float4 index swizzled by its bits 4-5, component by bit 3 (the 16 even bins
// a wave writes per instruction are 16 words apart: unswizzled they share two banks).  A reader of float4 q4 fetches slot
// ct_out_slot(q4) and swaps its halves when q4 & 8 (ct_out_fix).
__device__ __forceinline__ int ct_out_slot(int q4) { return q4 ^ ((q4 >> 4) & 3); }
__device__ __forceinline__ int ct_out_word(int b) {
    const int q4 = b >> 2;
    return 4 * ct_out_slot(q4) + ((b & 3) ^ ((q4 >> 2) & 2));
}
__device__ __forceinline__ float4 ct_out_fix(float4 v, int q4) { return (q4 & 8) ? make_float4(v.z, v.w, v.x, v.y) : v; }

// Stages A and B of one tile.  U1 holds the 32 frames' operand records; on return lane (j = lane & 31, kh = lane >> 5) of wave k1
// holds, in acc[r], output column j of frame mfma32_row(r, lane): bin ct_bin(k1, j).  Contains the barrier between the stages;
// the caller puts one before (U1 complete) and must not touch Yb.
__device__ __forceinline__ int ct_bin(int k1, int j) { return j < 16 ? 2 * k1 + 16 * j : 255 - 2 * k1 - 16 * (j - 16); }

__device__ __forceinline__ f32x16 ct_stages(const float* __restrict__ U1, float* __restrict__ Yb, const float (&ma)[2][4],
                                            const float (&mb)[16], int wave, int lane) {
    // ---- stage A: rows = 16 frames (two blocks), K = 16 inputs of record n2, columns = (k1, re / im)
    {
        const int i = lane & 15, kq = lane >> 4;
        const int j = lane & 15, k1o = j & 7, reim = j >> 3;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int n2 = 2 * wave + q;
            const int kk2 = n2 + 16 * reim;                                     // input index of stage B
            float* ydst = Yb + k1o * CT_KS2 + 16 * (kk2 & 1) + (((kk2 >> 1) + k1o) & 15);     // rotated by k1: MB's row order follows
#pragma unroll
            for (int rb = 0; rb < 2; ++rb) {
                const float4 a = bs_ld4(U1 + (16 * rb + i) * CT_FS1 + n2 * CT_REC + 4 * (kq ^ (n2 >> 3)));      // inputs 4 kq .. 4 kq + 3
                ct_f4 d = ct_f4{0.f, 0.f, 0.f, 0.f};
                d = ct_mfma16(a.x, ma[q][0], d);
                d = ct_mfma16(a.y, ma[q][1], d);
                d = ct_mfma16(a.z, ma[q][2], d);
                d = ct_mfma16(a.w, ma[q][3], d);
#pragma unroll
                for (int r = 0; r < 4; ++r) ydst[(16 * rb + 4 * kq + r) * CT_FS2] = d[r];      // frame 16 rb + 4 (lane >> 4) + r
            }
        }
    }
    __syncthreads();
    // ---- stage B: rows = 32 frames, K = 32 (Re / Im Y' over n2), columns = 32 outputs of k1 = wave
    f32x16 acc = f32x16{0};
    {
        const float* src = Yb + wave * CT_KS2 + (lane & 31) * CT_FS2 + 16 * (lane >> 5);
        const float4 a0 = bs_ld4(src), a1 = bs_ld4(src + 4), a2 = bs_ld4(src + 8), a3 = bs_ld4(src + 12);
        acc = mfma32x32x2(a0.x, mb[0], acc);  acc = mfma32x32x2(a0.y, mb[1], acc);  acc = mfma32x32x2(a0.z, mb[2], acc);  acc = mfma32x32x2(a0.w, mb[3], acc);
        acc = mfma32x32x2(a1.x, mb[4], acc);  acc = mfma32x32x2(a1.y, mb[5], acc);  acc = mfma32x32x2(a1.z, mb[6], acc);  acc = mfma32x32x2(a1.w, mb[7], acc);
        acc = mfma32x32x2(a2.x, mb[8], acc);  acc = mfma32x32x2(a2.y, mb[9], acc);  acc = mfma32x32x2(a2.z, mb[10], acc); acc = mfma32x32x2(a2.w, mb[11], acc);
        acc = mfma32x32x2(a3.x, mb[12], acc); acc = mfma32x32x2(a3.y, mb[13], acc); acc = mfma32x32x2(a3.z, mb[14], acc); acc = mfma32x32x2(a3.w, mb[15], acc);
    }
    return acc;
}

__device__ __forceinline__ void ct_load_matrices(const float* __restrict__ img, float (&ma)[2][4], float (&mb)[16], int wave, int lane) {
    const float4* p = reinterpret_cast<const float4*>(img + (size_t)(wave * 64 + lane) * 24);
    const float4 v0 = p[0], v1 = p[1], v2 = p[2], v3 = p[3], v4 = p[4], v5 = p[5];
    ma[0][0] = v0.x; ma[0][1] = v0.y; ma[0][2] = v0.z; ma[0][3] = v0.w;
    ma[1][0] = v1.x; ma[1][1] = v1.y; ma[1][2] = v1.z; ma[1][3] = v1.w;
    mb[0] = v2.x; mb[1] = v2.y; mb[2] = v2.z; mb[3] = v2.w; mb[4] = v3.x; mb[5] = v3.y; mb[6] = v3.z; mb[7] = v3.w;
    mb[8] = v4.x; mb[9] = v4.y; mb[10] = v4.z; mb[11] = v4.w; mb[12] = v5.x; mb[13] = v5.y; mb[14] = v5.z; mb[15] = v5.w;
}

// ------------------------------------------------------------------------------------------------------------------
// K1.  grid = workers (<= 512: two per CU), block = 512.  SPEC / PAIR / STATS as in mdct_b3.h.
// ------------------------------------------------------------------------------------------------------------------
template <int MODE, bool SPEC, bool PAIR, bool STATS>
__global__ __launch_bounds__(CT_NT, 4) void mdct4_ct_kernel(const float* __restrict__ audio, int B, int T, int F,
                                                            const float* __restrict__ window, const float* __restrict__ img,
                                                            CodecParams cp, float* __restrict__ spec, float* __restrict__ in2,
                                                            double* __restrict__ stats) {
    constexpr int Q = M / 2;
    extern __shared__ __attribute__((aligned(16))) float ct_smem[];
    float* U1 = ct_smem;                   // [32][CT_FS1] operand records; later the output tile [32][CT_OS]
    float* Yb = ct_smem + CT_U1;           // [8][CT_KS2]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rows = B * F, n_tiles = (rows + CT_ROWS - 1) / CT_ROWS, G = gridDim.x;
    const float k1c = (float)(((double)cp.nr1 - (double)cp.nr0) / ((double)cp.mx - (double)cp.mn));
    const float k0c = (float)((double)cp.nr0 - (double)cp.mn * (((double)cp.nr1 - (double)cp.nr0) / ((double)cp.mx - (double)cp.mn)));
    const float gain = cp.gain, nr0 = cp.nr0;
    const __amdgpu_buffer_rsrc_t r_audio = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(audio), 0, (unsigned)B * (unsigned)T * 4u, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_spec = __builtin_amdgcn_make_buffer_rsrc(spec, 0, SPEC ? (unsigned)rows * M * 4u : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_in2 = __builtin_amdgcn_make_buffer_rsrc(in2, 0, PAIR ? (unsigned)rows * M * 8u : 0u, 0x00020000);

    float ma[2][4], mb[16];
    ct_load_matrices(img, ma, mb, wave, lane);

    // fold group gi: row r = wave + 8 gi, u[n .. n + 3], n = 4 lane (mdct_bs.h):  u = s z[straight] - rev(z[reversed]), z = fl32(x w)
    const int n = 4 * lane;
    const bool lo = n < Q;
    const int o1 = 3 * Q - 4 - n, o2 = lo ? 3 * Q + n : n - Q;
    const float4 fw1 = bs_ld4(window + o1);
    float4 fw2 = bs_ld4(window + o2);
    if (lo) fw2 = make_float4(-fw2.x, -fw2.y, -fw2.z, -fw2.w);          // the sign rides in the window: fl32(x * -w) = -fl32(x * w)
    const int w0 = ct_u1_word(n), w1 = ct_u1_word(n + 1), w2 = ct_u1_word(n + 2), w3 = ct_u1_word(n + 3);
    bs_v4u x1[4], x2[4];
    auto load_tile = [&](int tile) {
#pragma unroll
        for (int gi = 0; gi < 4; ++gi) {
            const int m = tile * CT_ROWS + wave + 8 * gi;
            const int b = m / F, f = m - b * F;
            const int t1 = f * M - M + o1, t2 = f * M - M + o2;             // T % 4 == 0: a float4 is inside or outside the clip as a whole
            const unsigned base = (unsigned)b * (unsigned)T;
            const unsigned a1 = (m < rows && t1 >= 0 && t1 + 3 < T) ? (base + (unsigned)t1) * 4u : BS_OOB;
            const unsigned a2 = (m < rows && t2 >= 0 && t2 + 3 < T) ? (base + (unsigned)t2) * 4u : BS_OOB;
            x1[gi] = __builtin_amdgcn_raw_buffer_load_b128(r_audio, a1, 0, 0);      // out of range reads 0: the zero padding
            x2[gi] = __builtin_amdgcn_raw_buffer_load_b128(r_audio, a2, 0, 0);
        }
    };
    auto fold_tile = [&]() {
#pragma unroll
        for (int gi = 0; gi < 4; ++gi) {
            const float4 a = __builtin_bit_cast(float4, x1[gi]), c = __builtin_bit_cast(float4, x2[gi]);
            const float4 z1 = make_float4(__fmul_rn(a.x, fw1.x), __fmul_rn(a.y, fw1.y), __fmul_rn(a.z, fw1.z), __fmul_rn(a.w, fw1.w));
            const float4 z2 = make_float4(__fmul_rn(c.x, fw2.x), __fmul_rn(c.y, fw2.y), __fmul_rn(c.z, fw2.z), __fmul_rn(c.w, fw2.w));
            float* dst = U1 + (wave + 8 * gi) * CT_FS1;
            dst[w0] = z2.x - z1.w; dst[w1] = z2.y - z1.z; dst[w2] = z2.z - z1.y; dst[w3] = z2.w - z1.x;
        }
    };
    double sd1 = 0.0, sd2 = 0.0;
    const int j = lane & 31, kh = lane >> 5;
    const int obin = ct_out_word(ct_bin(wave, j));               // this lane's column in the output tile
    const int q4 = tid & 63, fr0 = tid >> 6;                      // P5: float4 q4 of rows fr0 + 8 g

    int tile = blockIdx.x;
    load_tile(tile < n_tiles ? tile : n_tiles);
    for (; tile < n_tiles; tile += G) {
        fold_tile();                                              // P1
        __syncthreads();
        f32x16 acc = ct_stages(U1, Yb, ma, mb, wave, lane);       // P2, barrier, P3
        load_tile(tile + G < n_tiles ? tile + G : n_tiles);       // the next tile's audio: in flight under P4 / P5
        // P4: codec, statistics, output tile (every wave finished reading U1 at the barrier inside ct_stages)
        float f1 = 0.0f, f2 = 0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int fr = mfma32_row(r, lane);
            float l = acc[r], v = acc[r];
            if (MODE != CODEC_RAW) {
                l = asinh_over_ln10(gain * acc[r]);
                v = fmaf(l, k1c, k0c);
                if (STATS) {
                    const float lm = (tile * CT_ROWS + fr < rows) ? l : 0.0f;
                    f1 += lm; f2 = fmaf(lm, lm, f2);
                }
            }
            U1[fr * CT_OS + obin] = v;
        }
        if (STATS && MODE != CODEC_RAW) { sd1 += (double)f1; sd2 += (double)f2; }
        __syncthreads();
        // P5: whole rows, 16 bytes per lane, one contiguous KiB per wave instruction
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int fr = fr0 + 8 * g;
            const float4 v = ct_out_fix(bs_ld4(U1 + fr * CT_OS + 4 * ct_out_slot(q4)), q4);
            const unsigned row = (unsigned)(tile * CT_ROWS + fr);
            if (SPEC) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(bs_v4u, v), r_spec, (row * M + 4u * q4) * 4u, 0, 0);
            if (PAIR) {
                const bs_v4u p0 = {__float_as_uint(v.x), __float_as_uint(fmaf(fabsf(v.x), 2.0f, nr0)), __float_as_uint(v.y), __float_as_uint(fmaf(fabsf(v.y), 2.0f, nr0))};
                const bs_v4u p1 = {__float_as_uint(v.z), __float_as_uint(fmaf(fabsf(v.z), 2.0f, nr0)), __float_as_uint(v.w), __float_as_uint(fmaf(fabsf(v.w), 2.0f, nr0))};
                __builtin_amdgcn_raw_buffer_store_b128(p0, r_in2, (row * M + 4u * q4) * 8u, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b128(p1, r_in2, (row * M + 4u * q4) * 8u + 16u, 0, 0);
            }
        }
        __syncthreads();                                          // the tile has been read: the next fold may overwrite it
    }
    if (STATS && MODE != CODEC_RAW) {
        // one pair of double atomics per WORKGROUP (the grid ends together: 8 192 atomics on two addresses were a serial tail)
        sd1 = wave_sum_d(sd1); sd2 = wave_sum_d(sd2);
        double* red = reinterpret_cast<double*>(Yb);               // (the loop ended on a barrier: the tile buffers are free)
        if (lane == 0) { red[2 * wave] = sd1; red[2 * wave + 1] = sd2; }
        __syncthreads();
        if (tid < 2) {
            double a = 0.0;
#pragma unroll
            for (int w = 0; w < CT_NT / 64; ++w) a += red[2 * w + tid];
            atomicAdd(stats + tid, a);
        }
    }
}

constexpr size_t CT_K1_LDS = (size_t)(CT_U1 + CT_YB) * sizeof(float);

// ------------------------------------------------------------------------------------------------------------------
// K2.  grid = workers (<= 512, <= B), block = 512.  A workgroup walks whole clips, tile by tile in frame order: the frame in front
// of a tile is row 31 of the tile before (2-slot LDS ring).
// ------------------------------------------------------------------------------------------------------------------
//
// ST (generate_audio.py:40-53 folded into the overlap-add store): clip c is segment sa.first + c of ONE stitched waveform of
// sa.total samples; its sample t lands at (sa.first + c) * sa.pitch - sa.overlap + t (pitch = segment length - overlap; positions
// outside [0, total) are the reference's final crop).  The first and last `overlap` samples of every segment are halved and ADDED
// (global_atomic_add_f32 into a waveform the caller zeroed: at most two segments meet in a sample and a two-term float sum does
// not depend on the order, so the result equals F.fold's bit for bit); everything else is a plain 16-byte store.  (StitchArgs: mdct.hip)
template <int MODE, bool ST = false>
__global__ __launch_bounds__(CT_NT, 4) void imdct4_ct_kernel(const float* __restrict__ spec, int B, int F,
                                                             const float* __restrict__ window, const float* __restrict__ img,
                                                             CodecParams cp, float* __restrict__ audio, int out_len, StitchArgs sa) {
    constexpr int Q = M / 2;
    extern __shared__ __attribute__((aligned(16))) float ct_smem[];
    float* U1 = ct_smem;
    float* Yb = ct_smem + CT_U1;
    float* halo = Yb + CT_YB;              // [2][CT_OS] row 31 of the last two tiles (swizzled like the tile's rows)
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tpc = (F + CT_ROWS - 1) / CT_ROWS, G = gridDim.x;
    const int my_clips = ((int)blockIdx.x < B) ? (B - 1 - (int)blockIdx.x) / G + 1 : 0;
    const int n_seq = my_clips * tpc;
    const __amdgpu_buffer_rsrc_t r_spec = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(spec), 0, (unsigned)B * (unsigned)F * M * 4u, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_out = __builtin_amdgcn_make_buffer_rsrc(audio, 0, ST ? (unsigned)sa.total * 4u : (unsigned)B * (unsigned)out_len * 4u, 0x00020000);

    float ma[2][4], mb[16];
    ct_load_matrices(img, ma, mb, wave, lane);
    const int n = 4 * lane;
    const bool lo = n < Q;
    const int w0 = ct_u1_word(n), w1 = ct_u1_word(n + 1), w2 = ct_u1_word(n + 2), w3 = ct_u1_word(n + 3);
    const float4 uw0 = bs_ld4(window + n), uw1 = bs_ld4(window + n + M);
    const float scale = 4.0f / (2 * M), rgain = 1.0f / cp.gain;
    const int j = lane & 31, kh = lane >> 5;
    const int obin = ct_out_word(ct_bin(wave, j));
    // unfold reads of this lane: y_h[n] = v_h[Q + n] | -v_h[3Q - 1 - n];   y_{h-1}[n + M] = -v_{h-1}[Q - 1 - n] | -v_{h-1}[n - Q]
    const int uc_q4 = (lo ? Q + n : 3 * Q - 4 - n) >> 2, up_q4 = (lo ? Q - 4 - n : n - Q) >> 2;
    const int uc_w = 4 * ct_out_slot(uc_q4), up_w = 4 * ct_out_slot(up_q4);

    int nx_clip = blockIdx.x, nx_t = 0;                            // the tile being loaded
    bs_v4u xr[4];
    auto load_tile = [&]() {
#pragma unroll
        for (int gi = 0; gi < 4; ++gi) {
            const int f = nx_t * CT_ROWS + wave + 8 * gi;
            const bool ok = nx_clip < B && f < F;
            xr[gi] = __builtin_amdgcn_raw_buffer_load_b128(r_spec, ok ? (((unsigned)nx_clip * F + (unsigned)f) * M + 4u * lane) * 4u : BS_OOB, 0, 0);
        }
    };
    for (int t = tid; t < 2 * CT_OS; t += CT_NT) halo[t] = 0.0f;
    load_tile();
    for (int i = 0; i < n_seq; ++i) {
        const int clip = nx_clip, f0 = nx_t * CT_ROWS;
        // P1: decode into the operand records.  x = v c1 + c0 (= ln10 ((v - nr0) / (nr1 - nr0) (max - min) + min)), X = sinh(x) / gain
        {
            float mn = cp.mn, mx = cp.mx;
            if (cp.per_sample) { mn = cp.mn_b[clip]; mx = cp.mx_b[clip]; }
            const double k = ((double)mx - (double)mn) / ((double)cp.nr1 - (double)cp.nr0);
            const double sc = (MODE == CODEC_ARCSINH) ? (double)LN10F : 1.0;
            const float c1 = (float)(k * sc), c0 = (float)(((double)mn - (double)cp.nr0 * k) * sc);
            auto dec1 = [&](float v) -> float {
                if (MODE == CODEC_RAW) return v;
                const float x = fmaf(v, c1, c0);
                return MODE == CODEC_ARCSINH ? sinh_fast(x) * rgain : x;
            };
#pragma unroll
            for (int gi = 0; gi < 4; ++gi) {
                const float4 x = __builtin_bit_cast(float4, xr[gi]);
                const bool ok = f0 + wave + 8 * gi < F;                       // frames outside the clip contribute nothing (decode(0) != 0)
                float* dst = U1 + (wave + 8 * gi) * CT_FS1;
                dst[w0] = ok ? dec1(x.x) : 0.0f; dst[w1] = ok ? dec1(x.y) : 0.0f; dst[w2] = ok ? dec1(x.z) : 0.0f; dst[w3] = ok ? dec1(x.w) : 0.0f;
            }
        }
        if (++nx_t == tpc) { nx_t = 0; nx_clip += G; }
        __syncthreads();
        f32x16 acc = ct_stages(U1, Yb, ma, mb, wave, lane);
        load_tile();
        // P4: v = DCT-IV(X) into the output tile; row 31 also into this tile's halo slot
#pragma unroll
        for (int r = 0; r < 16; ++r) U1[mfma32_row(r, lane) * CT_OS + obin] = acc[r];
        if (kh == 1) halo[(i & 1) * CT_OS + obin] = acc[15];                 // row 31 = register 15 of the upper half
        __syncthreads();
        // P5: hop block h = f0 + fr:  out[(h - 1) M + n] = 4 / N (w[n] y_h[n] + w[n + M] y_{h-1}[n + M])
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int fr = wave + 8 * g;
            const float* vc = U1 + fr * CT_OS;
            const float* vp = fr == 0 ? halo + ((i + 1) & 1) * CT_OS : U1 + (fr - 1) * CT_OS;     // (the tile before wrote slot (i - 1) & 1)
            const float4 c = ct_out_fix(bs_ld4(vc + uc_w), uc_q4), q = ct_out_fix(bs_ld4(vp + up_w), up_q4);
            const float4 yc = lo ? c : make_float4(-c.w, -c.z, -c.y, -c.x);
            const float4 yp = lo ? make_float4(-q.w, -q.z, -q.y, -q.x) : make_float4(-q.x, -q.y, -q.z, -q.w);
            const float4 o = make_float4(scale * (uw0.x * yc.x + uw1.x * yp.x), scale * (uw0.y * yc.y + uw1.y * yp.y),
                                         scale * (uw0.z * yc.z + uw1.z * yp.z), scale * (uw0.w * yc.w + uw1.w * yp.w));
            const int hh = f0 + fr, t0 = (hh - 1) * M + n;                   // out_len % 4 == 0: a float4 is inside or outside the crop as a whole
            const bool ok = hh >= 1 && hh <= F - 1 && t0 + 3 < out_len;
            if (!ST) {
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(bs_v4u, o), r_out,
                                                       ok ? ((unsigned)clip * (unsigned)out_len + (unsigned)t0) * 4u : BS_OOB, 0, 0);
            } else {
                // pitch, overlap, out_len % 4 == 0: a float4 is inside or outside a cross-fade zone / the final crop as a whole
                const long long gi = sa.base + (long long)clip * sa.pitch + t0;
                const bool in = ok && gi >= 0 && gi + 3 < sa.total;
                const bool zone = t0 < sa.overlap || t0 >= out_len - sa.overlap;
                if (zone) {
                    if (in) {
                        float* dst = audio + gi;
                        unsafeAtomicAdd(dst, 0.5f * o.x); unsafeAtomicAdd(dst + 1, 0.5f * o.y);
                        unsafeAtomicAdd(dst + 2, 0.5f * o.z); unsafeAtomicAdd(dst + 3, 0.5f * o.w);
                    }
                } else {
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(bs_v4u, o), r_out, in ? (unsigned)gi * 4u : BS_OOB, 0, 0);
                }
            }
        }
        __syncthreads();
    }
}

constexpr size_t CT_K2_LDS = (size_t)(CT_U1 + CT_YB + 2 * CT_OS) * sizeof(float);

}  // namespace
