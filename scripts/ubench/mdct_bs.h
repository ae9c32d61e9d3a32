// RETIRED from the product library in round 5 (mdct_ct.h is selected at every size): kept for scripts/ubench/mdct_bs_bench.hip.
// K1 / K2 as table-stationary contractions (round 3).  Include after mdctgan_amd/csrc/mdct.hip (needs M, CodecParams, mdct_codec.h).
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
constexpr int BS_K2_AROWS = 40;         // K2 operand tile rows in LDS: 32 frames + the halo + padding (every wave stores its 5th group)

// ------------------------------------------------------------------------------------------------------------------
// K1
// ------------------------------------------------------------------------------------------------------------------
// Register image of the table (mg_dct4_image; the kernels' `dct4` parameter IS this image): float4 img[slab = bin / 32][jj][lane] =
// D4[32 slab + (lane & 31)][8 jj + 4 (lane >> 5) + (0..3)] -- a wave's load of step jj is one coalesced 1 KiB read (the
// same values straight from the [M][M] table are 32 lines x 32 bytes per instruction).
__global__ void dct4_image_kernel(const float* __restrict__ d4, float* __restrict__ img) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;            // one float4 each: M * M / 4 of them
    if (i >= M * M / 4) return;
    const int lane = i & 63, jj = (i >> 6) & 31, slab = i >> 11;
    reinterpret_cast<float4*>(img)[i] = *reinterpret_cast<const float4*>(d4 + (size_t)(32 * slab + (lane & 31)) * M + 8 * jj + 4 * (lane >> 5));
}

// One tile of a wave is 128 MFMAs on ONE accumulator (64-cycle dependent issue, 128 with the SIMD's second wave beside it):
// every gap takes a handful of independent instructions for free, but the wave issues IN ORDER -- work placed behind a run
// of MFMAs only sees the last gap.  The kernel therefore runs three tiles at once per wave, interleaved per MFMA: the MFMA
// stream of tile i, the codec + stores of tile i - 1 out of a second accumulator (one frame row per two table steps) and
// the fold of tile i + 1 into the other LDS buffer (signal loads issued at the top of the iteration, folded in the second
// half of the stream).  The iteration body is ONE basic block -- masked lanes are out-of-range buffer offsets, the codec and
// the pair are template parameters, selects instead of branches -- cut by hand into pieces of a few instructions, one piece
// behind each MFMA, fenced with sched_barrier so that the compiler keeps them there (phases measured before, scripts/ubench/mdct_bs_bench at 4096 clips: 485 us of
// MFMA + 160 fold + 105 codec + 100..200 stores, nothing overlapping).
// MODE: CODEC_RAW / CODEC_ARCSINH.  DBG (ubench only): bit 0 no global stores, bit 1 no codec arithmetic, bit 2 no fold.

template <int NW, int MODE, bool PAIR, bool STATS = false, int DBG = 0>
__global__ __launch_bounds__(NW * 64) void mdct4_bs_kernel(const float* __restrict__ audio, int B, int T, int F,
                                                           const float* __restrict__ window, const float* __restrict__ dct4,
                                                           CodecParams cp, float* __restrict__ spec, float* __restrict__ in2,
                                                           double* __restrict__ stats) {
    constexpr int NT = NW * 64, Q = M / 2;
    constexpr int GROUPS = BS_ROWS * (M / 4) / NT;        // float4 groups of the folded tile per thread (4 / 16)
    constexpr int FSTEP = 16 / GROUPS;                    // fold slice g rides behind table step 16 + g * FSTEP
    extern __shared__ __attribute__((aligned(16))) float bs_smem[];
    float* ws = bs_smem;                                  // [2 M] window, then [2 M] its negative
    float* abuf = bs_smem + 4 * M;                        // [2][BS_ROWS][BS_LDA]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int rows = B * F, n_tiles = (rows + BS_ROWS - 1) / BS_ROWS, G = gridDim.x;
    const int slab = blockIdx.y * NW + wave, col = slab * 32 + (lane & 31), kh = lane >> 5;
    BsCodec cd = bs_codec(cp);
    cd.mode = MODE;
    const float k1 = (float)(((double)cp.nr1 - (double)cp.nr0) / ((double)cp.mx - (double)cp.mn));
    const float k0 = (float)((double)cp.nr0 - (double)cp.mn * (((double)cp.nr1 - (double)cp.nr0) / ((double)cp.mx - (double)cp.mn)));
    const __amdgpu_buffer_rsrc_t r_audio = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(audio), 0, (unsigned)B * (unsigned)T * 4u, 0x00020000);
    // DBG bit 0: an empty range drops every store (the arithmetic stays alive)
    const __amdgpu_buffer_rsrc_t r_spec = __builtin_amdgcn_make_buffer_rsrc(spec, 0, ((DBG & 1) || !spec) ? 0u : (unsigned)rows * M * 4u, 0x00020000);      // (spec == NULL with the pair: an empty range drops the stores)
    const __amdgpu_buffer_rsrc_t r_in2 = __builtin_amdgcn_make_buffer_rsrc(in2, 0, (PAIR && !(DBG & 1)) ? (unsigned)rows * M * 8u : 0u, 0x00020000);

    // group gi of a thread: row r = (tid + gi * NT) / 64 of the tile, u[n .. n + 3] with n = 4 * lane.  With z = fl32(x * w):
    //   n <  Q:  u = -rev(z[3Q-4-n ..]) - z[3Q+n ..]        n >= Q:  u = z[n-Q ..] - rev(z[3Q-4-n ..])
    // i.e. for every lane  u = s * z[oS ..] - rev(z[oR ..])  with oR = 3Q-4-n, (oS, s) = (3Q+n, -1) | (n-Q, +1); the sign rides in
    // the window (the second LDS copy is -w: fl32(x * -w) = -fl32(x * w)), so a half wave needs no selects
    const int n = 4 * lane;
    const bool lo = n < Q;
    const int o1 = 3 * Q - 4 - n, o2 = lo ? 3 * Q + n : n - Q;         // (o1: the reversed run, o2: the straight one)
    const float* wS = ws + (lo ? 2 * M : 0) + o2;
    bs_v4u x1[GROUPS], x2[GROUPS];
    auto load_tile = [&](int tile) {
#pragma unroll
        for (int gi = 0; gi < GROUPS; ++gi) {
            const int r = wave + gi * NW;                 // (tid + gi * NT) >> 6
            const int m = tile * BS_ROWS + r;
            const int b = m / F, f = m - b * F;
            const int t1 = f * M - M + o1, t2 = f * M - M + o2;     // T % 4 == 0: a float4 is inside or outside the clip as a whole
            const unsigned base = (unsigned)b * (unsigned)T;
            const unsigned a1 = (m < rows && t1 >= 0 && t1 + 3 < T) ? (base + (unsigned)t1) * 4u : BS_OOB;
            const unsigned a2 = (m < rows && t2 >= 0 && t2 + 3 < T) ? (base + (unsigned)t2) * 4u : BS_OOB;
            x1[gi] = __builtin_amdgcn_raw_buffer_load_b128(r_audio, a1, 0, 0);      // out of range reads 0: the zero padding
            x2[gi] = __builtin_amdgcn_raw_buffer_load_b128(r_audio, a2, 0, 0);
        }
    };
    // The fold of one group and the codec + store of one frame row, cut into pieces that ride behind single MFMAs
    // (fold_piece 0..3, epi_piece 0..7); state between the pieces lives in these registers.
    float4 fw1, fw2, fz1, fz2;
    auto fold_piece = [&](int pc, int gi, int buf) {
        const int r = wave + gi * NW;
        if (pc == 0) {
            fw1 = bs_ld4(ws + o1); fw2 = bs_ld4(wS);
        } else if (pc == 1) {       // z = fl32(x * w) (mdct.py:410)
            const float4 a = __builtin_bit_cast(float4, x1[gi]), c = __builtin_bit_cast(float4, x2[gi]);
            fz1 = make_float4(__fmul_rn(a.x, fw1.x), __fmul_rn(a.y, fw1.y), __fmul_rn(a.z, fw1.z), __fmul_rn(a.w, fw1.w));
            fz2 = make_float4(__fmul_rn(c.x, fw2.x), __fmul_rn(c.y, fw2.y), __fmul_rn(c.z, fw2.z), __fmul_rn(c.w, fw2.w));
        } else if (pc == 2) {       // the TDAC fold:  u = (+-z)[straight] - rev(z[reversed])
            fz1 = make_float4(fz2.x - fz1.w, fz2.y - fz1.z, fz2.z - fz1.y, fz2.w - fz1.x);
        } else {
            *reinterpret_cast<float4*>(abuf + (size_t)buf * BS_ROWS * BS_LDA + r * BS_LDA + n) = fz1;
        }
    };
    auto fold_slice = [&](int gi, int buf) {
#pragma unroll
        for (int pc = 0; pc < 4; ++pc) fold_piece(pc, gi, buf);
    };
    float ea, ea2, et, ep, esq, esm, el, ev, elog = 0.0f;
    double s1 = 0.0, s2 = 0.0;      // STATS: sum / sum of squares of the log-domain values (the returned mean / std)
    float f1 = 0.0f, f2 = 0.0f;     // ... of this lane's 16 values of the tile in flight: float32 inside a tile (three VALU operations per
                                    // value in the MFMA's shadow instead of four double-rate ones), double across tiles
    int em0 = 0;                    // STATS: first frame row of this lane in the previous tile (rows: none)
    unsigned eob = BS_OOB, eob2 = BS_OOB;   // byte offset of (first frame row of this lane's 16, bin col) in spec / in the pair
    auto epi_piece = [&](int pc, float xv, int r) {      // frame row emrow + (r & 3) + 8 (r >> 2), bin col
        const int dm = (r & 3) + 8 * (r >> 2);
        if (MODE == CODEC_RAW || (DBG & 2)) {
            if (pc == 0) ev = xv;
        } else if (pc == 0) {
            const float y = cd.gain * xv;
            ea = fabsf(y); ea2 = ea * ea; et = ea2 + 1.0f; el = y;
        } else if (pc == 1) {
            esq = __builtin_amdgcn_sqrtf(et);
            ep = fmaf(ea2, fmaf(ea2, fmaf(ea2, -0.044642857142857144f, 0.075f), -0.16666666666666666f), 1.0f);
        } else if (pc == 2) {
            et = __builtin_amdgcn_logf(ea + esq);
            esm = ea * ep * INV_LN10F;
        } else if (pc == 3) {
            const float big = et * LOG10_2F;
            elog = copysignf(ea < 0.125f ? esm : big, el);
        } else if (pc == 4) {
            // (l - min) / (max - min) * (nr1 - nr0) + nr0 as ONE fma with the constants folded in double: within an ulp of the
            // exact value (the reference's four float32 operations: within two)
            ev = fmaf(elog, k1, k0);
        }
        if (STATS && pc == 5 && MODE != CODEC_RAW) {
            const float l = (em0 + dm < rows) ? elog : 0.0f;
            f1 += l; f2 = fmaf(l, l, f2);
        }
        // addressing costs no VALU: the row inside the tile is the instruction's scalar offset, and a row behind the last one
        // lies behind num_records (the range check covers voffset + soffset), so the hardware drops it
        if (pc == 6) {
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(ev), r_spec, eob, dm * (M * 4), 0);
        } else if (pc == 7 && PAIR) {
            const bs_v2u pr = {__float_as_uint(ev), __float_as_uint(fmaf(fabsf(ev), 2.0f, cd.nr0))};     // (x 2 is exact: == |v| * 2 + nr0)
            __builtin_amdgcn_raw_buffer_store_b64(pr, r_in2, eob2, dm * (M * 8), 0);
        }
    };
    auto epi_row = [&](float xv, int r) {
#pragma unroll
        for (int pc = 0; pc < 8; ++pc) epi_piece(pc, xv, r);
    };
    auto epi_begin = [&](int tile) {         // tile == n_tiles: nothing to store
        if (STATS && MODE != CODEC_RAW) { s1 += (double)f1; s2 += (double)f2; f1 = f2 = 0.0f; }     // the finished tile's sums
        const unsigned m0 = (unsigned)tile * BS_ROWS + 4u * kh;
        eob = (tile < n_tiles) ? (m0 * M + (unsigned)col) * 4u : BS_OOB - 31u * M * 8u;      // (+ soffset stays out of range, no wrap)
        eob2 = (tile < n_tiles) ? 2u * eob : eob;
        em0 = (tile < n_tiles) ? (int)m0 : rows;
    };

    // prologue: the first tile's signal first (HBM latency), then window and table; the fold runs while the table arrives
    int tile = blockIdx.x;
    load_tile(tile < n_tiles ? tile : n_tiles);           // (tile n_tiles: every row out of range)
    float4 wv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (tid < 2 * M / 4) wv = bs_ld4(window + 4 * tid);
    float4 bt[32];          // the wave's slab of the table: bins [col] x k = 8 jj + 4 kh + (0..3)
    {
        const float4* img = reinterpret_cast<const float4*>(dct4) + (size_t)slab * 32 * 64 + lane;
#pragma unroll
        for (int jj = 0; jj < 32; ++jj) bt[jj] = img[jj * 64];
    }
    if (tid < 2 * M / 4) {
        reinterpret_cast<float4*>(ws)[tid] = wv;
        reinterpret_cast<float4*>(ws + 2 * M)[tid] = make_float4(-wv.x, -wv.y, -wv.z, -wv.w);
    }
    __syncthreads();                              // window in LDS
#pragma unroll
    for (int gi = 0; gi < GROUPS; ++gi) fold_slice(gi, 0);
    __syncthreads();

    f32x16 accp = f32x16{0};
    int buf = 0;
    epi_begin(n_tiles);                           // no previous tile yet
    constexpr int PPJ = 4 / FSTEP;                // fold pieces per table step in the second half (1: NW = 8, 4: NW = 2)
    for (; tile < n_tiles; tile += G, buf ^= 1) {
        const int next = tile + G;
        if (!(DBG & 4)) load_tile(next < n_tiles ? next : n_tiles);      // global loads in flight under the first half of the stream
        // (two alternating accumulator chains measured no better: 550 vs 512 us for the bare MFMA stream at 4096 clips)
        f32x16 acc = f32x16{0};
        const float* ap = abuf + (size_t)buf * BS_ROWS * BS_LDA + (lane & 31) * BS_LDA + 4 * kh;
        float4 a = bs_ld4(ap), an = a;
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int jj = 0; jj < 32; ++jj) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const float av = t == 0 ? a.x : t == 1 ? a.y : t == 2 ? a.z : a.w;
                const float bv = t == 0 ? bt[jj].x : t == 1 ? bt[jj].y : t == 2 ? bt[jj].z : bt[jj].w;
                acc = mfma32x32x2(av, bv, acc);
                if (t == 0 && jj + 1 < 32) an = bs_ld4(ap + 8 * (jj + 1));       // next step's A operand, one step ahead
                epi_piece(4 * (jj & 1) + t, accp[jj >> 1], jj >> 1);
                if (!(DBG & 4) && jj >= 16 && t >= 4 - PPJ)
                    fold_piece(((jj - 16) % FSTEP) * PPJ + (t - (4 - PPJ)), (jj - 16) / FSTEP, buf ^ 1);
                __builtin_amdgcn_sched_barrier(0);      // nothing moves across: each MFMA keeps its piece in its shadow
            }
            a = an;
        }
        accp = acc;
        epi_begin(tile);
        __syncthreads();
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) epi_row(accp[r], r);
    if (STATS && MODE != CODEC_RAW) {
        s1 += (double)f1; s2 += (double)f2;
        s1 = wave_sum_d(s1); s2 = wave_sum_d(s2);
        if (lane == 0) { atomicAdd(stats, s1); atomicAdd(stats + 1, s2); }
    }
}

constexpr size_t BS_K1_LDS = (size_t)(4 * M + 2 * BS_ROWS * BS_LDA) * sizeof(float);

// ------------------------------------------------------------------------------------------------------------------
// K2.  grid = workers, block = 512.  Tile = (clip b, frames f0 .. f0 + 31); emits hop blocks h = f0 .. f0 + 31:
//   out[(h - 1) M + n] = 4 / N * (w[n] y_h[n] + w[n + M] y_{h-1}[n + M]),  y = [v2, -v2_r, -v1_r, -v1] of v = DCT-IV(X).
// Same structure as K1: per wave three tiles in flight, one piece of the side work behind each MFMA -- the DCT stream of
// tile i (+ the halo frame f0 - 1 as four FMAs per table step), the unfold / window / overlap-add / store of tile i - 1 out
// of the other v buffer, the decode (denormalise, sinh) of tile i + 1 into the other operand buffer.  One barrier per tile.
// ------------------------------------------------------------------------------------------------------------------
template <int MODE>
__global__ __launch_bounds__(512) void imdct4_bs_kernel(const float* __restrict__ spec, int B, int F,
                                                        const float* __restrict__ window, const float* __restrict__ dct4,
                                                        CodecParams cp, float* __restrict__ audio, int out_len) {
    constexpr int NT = 512, Q = M / 2;
    constexpr int DG = 5;                                     // decode groups per thread: rows wave + 8 gi (row 32 = halo; 33..39 padding)
    constexpr int UG = BS_ROWS * (M / 4) / NT;                // unfold groups per thread: 4
    constexpr int A_F = BS_K2_AROWS * BS_LDA;                 // operand tile: rows 0..31 = frames f0.., row 32 = halo frame f0 - 1
    constexpr int V_F = (BS_ROWS + 1) * BS_LDA;               // v tile: the same rows
    extern __shared__ __attribute__((aligned(16))) float bs_smem[];
    float* abuf = bs_smem;                                // [2][A_F] decoded coefficients (the A operand)
    float* vbuf = abuf + 2 * A_F;                         // [2][V_F] v = DCT-IV(X) of the tile before
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tiles_per_clip = (F + BS_ROWS - 1) / BS_ROWS, n_tiles = B * tiles_per_clip, G = gridDim.x;
    const int col = wave * 32 + (lane & 31), kh = lane >> 5;
    const __amdgpu_buffer_rsrc_t r_spec = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(spec), 0, (unsigned)B * (unsigned)F * M * 4u, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_out = __builtin_amdgcn_make_buffer_rsrc(audio, 0, (unsigned)B * (unsigned)out_len * 4u, 0x00020000);

    // denormalise + decode folded into  x = v * c1 + c0  (= ln10 * ((v - nr0) / (nr1 - nr0) * (max - min) + min)),
    // X = sinh(x) / gain -- constants in double once per tile (per-sample ranges change them per clip)
    float c1 = 1.0f, c0 = 0.0f;
    const float rgain = 1.0f / cp.gain;
    auto tile_consts = [&](int tile) {
        const int b = (tile < n_tiles ? tile : 0) / tiles_per_clip;
        float mn = cp.mn, mx = cp.mx;
        if (cp.per_sample) { mn = cp.mn_b[b]; mx = cp.mx_b[b]; }
        const double k = ((double)mx - (double)mn) / ((double)cp.nr1 - (double)cp.nr0);
        const double sc = (MODE == CODEC_ARCSINH) ? (double)LN10F : 1.0;
        c1 = (float)(k * sc);
        c0 = (float)(((double)mn - (double)cp.nr0 * k) * sc);
    };
    bs_v4u xr[DG];
    float xok[DG];          // 1 / 0: frames outside the clip contribute nothing (decode(0) != 0); a factor, not a branch
    auto load_tile = [&](int tile, int g0, int g1) {        // groups [g0, g1): issued in two batches to keep fewer registers live
        const int tl = tile < n_tiles ? tile : 0;
        const int b = tl / tiles_per_clip, f0 = (tl - b * tiles_per_clip) * BS_ROWS;
#pragma unroll
        for (int gi = g0; gi < g1; ++gi) {
            const int j = wave + 8 * gi;                 // (tid + gi * NT) >> 6
            const int f = (j == BS_ROWS) ? f0 - 1 : f0 + j;
            const bool ok = tile < n_tiles && j <= BS_ROWS && f >= 0 && f < F;
            xok[gi] = ok ? 1.0f : 0.0f;
            const unsigned o = ok ? (((unsigned)b * F + (unsigned)f) * M + 4u * lane) * 4u : BS_OOB;
            xr[gi] = __builtin_amdgcn_raw_buffer_load_b128(r_spec, o, 0, 0);
        }
    };
    auto dec1 = [&](float v) -> float {
        if (MODE == CODEC_RAW) return v;
        const float x = fmaf(v, c1, c0);
        if (MODE == CODEC_ARCSINH) return sinh_fast(x) * rgain;
        return x;
    };
    auto decode_piece = [&](int pc, int gi, int buf) {      // pieces 0..3: one coefficient each, straight to LDS (no register tile)
        const float4 x = __builtin_bit_cast(float4, xr[gi]);
        float* dst = abuf + (size_t)buf * A_F + (wave + 8 * gi) * BS_LDA + 4 * lane;
        const float xv = pc == 0 ? x.x : pc == 1 ? x.y : pc == 2 ? x.z : x.w;
        dst[pc] = dec1(xv) * xok[gi];                       // (an out-of-range load returned 0: dec1 of it is finite)
    };
    // unfold of the previous tile, group gi: hop block j = wave + 8 gi, samples n = 4 lane ..+3
    const float scale = 4.0f / (2 * M);
    const int n = 4 * lane;
    const bool lo = n < Q;
    unsigned pbase = 0;         // element index of (clip b, sample (f0 - 1) M + n) of the previous tile in the output
    int pf0 = 0;
    bool pvalid = false;
    float4 uc, up;
    const float4 uw0 = bs_ld4(window + n), uw1 = bs_ld4(window + n + M);      // this lane's window values: the same for every hop block
    auto unfold_piece = [&](int pc, int gi, int vb) {
        const int j = wave + 8 * gi;
        const float* vc = vbuf + (size_t)vb * V_F + j * BS_LDA;                                   // frame h
        const float* vp = vbuf + (size_t)vb * V_F + ((j == 0) ? BS_ROWS : j - 1) * BS_LDA;        // frame h - 1
        if (pc == 0) {
            // y_h[n] = v_h[Q + n] | -v_h[3Q - 1 - n];   y_{h-1}[n + M] = -v_{h-1}[Q - 1 - n] | -v_{h-1}[n - Q]
            uc = bs_ld4(vc + (lo ? Q + n : 3 * Q - 4 - n));
            up = bs_ld4(vp + (lo ? Q - 4 - n : n - Q));
        } else if (pc == 1) {
            const float4 c = uc, q = up;
            uc = lo ? c : make_float4(-c.w, -c.z, -c.y, -c.x);
            up = lo ? make_float4(-q.w, -q.z, -q.y, -q.x) : make_float4(-q.x, -q.y, -q.z, -q.w);
        } else if (pc == 2) {
            uc = make_float4(scale * (uw0.x * uc.x + uw1.x * up.x), scale * (uw0.y * uc.y + uw1.y * up.y),
                             scale * (uw0.z * uc.z + uw1.z * up.z), scale * (uw0.w * uc.w + uw1.w * up.w));
        } else {
            const int hh = pf0 + j, t0 = (hh - 1) * M + n;       // out_len % 4 == 0: a float4 is inside or outside the crop as a whole
            const bool ok = pvalid && hh >= 1 && hh <= F - 1 && t0 + 3 < out_len;
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(bs_v4u, uc), r_out, ok ? (pbase + (unsigned)(j * M)) * 4u : BS_OOB, 0, 0);
        }
    };

    // prologue
    int tile = blockIdx.x;
    tile_consts(tile);
    load_tile(tile, 0, DG);
    // (decode first, one value at a time, THEN fetch the table: 20 interleaved sinh evaluations beside 128 live table registers
    // spill; the microsecond of table latency this exposes is paid once per workgroup)
#pragma unroll
    for (int gi = 0; gi < DG; ++gi)
#pragma unroll
        for (int pc = 0; pc < 4; ++pc) {
            decode_piece(pc, gi, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    float4 bt[32];
    {
        const float4* img = reinterpret_cast<const float4*>(dct4) + (size_t)wave * 32 * 64 + lane;
#pragma unroll
        for (int jj = 0; jj < 32; ++jj) bt[jj] = img[jj * 64];
    }
    __syncthreads();

    int buf = 0, vb = 0;
    for (; tile < n_tiles; tile += G, buf ^= 1, vb ^= 1) {
        tile_consts(tile + G);                      // (scalar work, outside the pinned stream)
        const float* at = abuf + (size_t)buf * A_F;
        f32x16 acc = f32x16{0};
        float4 h4 = make_float4(0.f, 0.f, 0.f, 0.f);       // halo frame: this lane's half of the k of its bin, four partial sums
        const float* ap = at + (lane & 31) * BS_LDA + 4 * kh;
        const float* hp = at + BS_ROWS * BS_LDA + 4 * kh;
        float4 a = bs_ld4(ap), an = a, xh = a;
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int jj = 0; jj < 32; ++jj) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const float av = t == 0 ? a.x : t == 1 ? a.y : t == 2 ? a.z : a.w;
                const float bv = t == 0 ? bt[jj].x : t == 1 ? bt[jj].y : t == 2 ? bt[jj].z : bt[jj].w;
                acc = mfma32x32x2(av, bv, acc);
                if (t == 0) { xh = bs_ld4(hp + 8 * jj); if (jj + 1 < 32) an = bs_ld4(ap + 8 * (jj + 1)); }
                if (t == 0 && jj == 7) load_tile(tile + G, 0, 3);      // next tile's coefficients: issued behind the unfold pieces,
                if (t == 0 && jj == 14) load_tile(tile + G, 3, DG);    // decoded from step 12 / 18 on
                if (t == 3) {
                    h4.x = fmaf(xh.x, bt[jj].x, h4.x); h4.y = fmaf(xh.y, bt[jj].y, h4.y);
                    h4.z = fmaf(xh.z, bt[jj].z, h4.z); h4.w = fmaf(xh.w, bt[jj].w, h4.w);
                }
                if (t == 1 || t == 2) {
                    const int slot = 2 * jj + (t - 1);          // 64 slots: 16 unfold pieces (their data is in LDS), later 20 decode pieces
                    if (slot < 4 * UG) unfold_piece(slot & 3, slot >> 2, vb ^ 1);
                    else if (slot >= 24 && slot < 24 + 4 * DG) decode_piece((slot - 24) & 3, (slot - 24) >> 2, buf ^ 1);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            a = an;
        }
        float h = (h4.x + h4.y) + (h4.z + h4.w);
        h += __shfl_xor(h, 32, 64);
        // v = DCT-IV(X) into this tile's v buffer: rows 0..31 frames, row 32 halo (the unfold of the NEXT iteration reads it)
        float* vt = vbuf + (size_t)vb * V_F;
#pragma unroll
        for (int r = 0; r < 16; ++r) vt[mfma32_row(r, lane) * BS_LDA + col] = acc[r];
        if (lane < 32) vt[BS_ROWS * BS_LDA + col] = h;
        {
            const int pb = tile / tiles_per_clip;
            pf0 = (tile - pb * tiles_per_clip) * BS_ROWS;
            pbase = (unsigned)pb * (unsigned)out_len + (unsigned)((pf0 - 1) * M + n);      // (wraps for pf0 == 0: those blocks are masked)
            pvalid = true;
        }
        __syncthreads();
    }
    // drain: the last tile's unfold
#pragma unroll
    for (int gi = 0; gi < UG; ++gi)
#pragma unroll
        for (int pc = 0; pc < 4; ++pc) unfold_piece(pc, gi, vb ^ 1);
}

constexpr size_t BS_K2_LDS = (size_t)(2 * BS_K2_AROWS * BS_LDA + 2 * (BS_ROWS + 1) * BS_LDA) * sizeof(float);

}  // namespace
