// RETIRED from the product library in round 5 (mdct_ct.h is selected at every size): kept for scripts/ubench/mdct_b3_bench.hip.
// Include after mdctgan_amd/csrc/mdct.hip and scripts/ubench/mdct_bs.h.
// K1 / K2 with the 256-point DCT-IV on the bf16 MFMA pipe, float32-accurate (round 4).  Included by mdct.hip after
// mdct_bs.h (needs M, CodecParams, bs_* helpers).
//
// Why: the contraction is MFMA-bound (68.7 GFLOP per 4096 clips: 437 us at the f32 MFMA peak against 134..201 us of HBM
// time), and v_mfma_f32_32x32x2_f32 runs at 1/16 of the rate of v_mfma_f32_32x32x16_bf16.  A float32 value is the EXACT sum
// of three bf16 pieces (x = h + m + l: h = bf16(x), m = bf16(x - h), l = bf16(x - h - m); both subtractions are exact), a
// bf16 x bf16 product is exact in the MFMA's float32 accumulator, so
//     a * b = ah bh + (ah bm + am bh) + (am bm + ah bl + al bh) + O(2^-24 |a b|)
// -- six bf16 MFMAs per 16 k instead of eight f32 MFMAs of twice the length: 192 cycles instead of 512.  The three dropped
// terms (am bl, al bm, al bl) are below one float32 ulp of the product; measured against the float64 oracle the spectra are
// as close as the f32-pipe kernels' (tests/test_mdct_gpu.py, same 2e-6 * max bar; frames stay bit-exact: the window
// multiply and the TDAC fold are the float32 operations of mdct_bs.h).
//
// Table-stationary like mdct_bs.h, but the table is now three bf16 images (3 x 128 KB): a workgroup of FOUR waves, one per
// SIMD with the whole 512-entry register file (256 VGPR + 256 AGPR) -- wave w keeps bins [64 w, 64 w + 64) for all 256 k
// and all three pieces in 384 registers (the MFMA reads its B operand straight from AGPRs) and streams 32-frame row tiles
// past them.  Per tile and wave: 16 k-blocks x (3 ds_read_b128 of the A pieces + 12 MFMAs) = 192 MFMAs = 6144 cycles.
//
// K1 (mdct4_b3_kernel) per tile, interleaved behind single MFMAs (one basic block, pieces pinned with sched_barrier):
//   * the codec + stores of tile i - 1 out of the previous accumulators (32 values per lane);
//   * 33 LDS-DMA pieces (buffer_load ... lds, 1 KiB each) that bring tile i + 1's hop blocks of raw audio into LDS --
//     no register holds a load in flight (the budget has none left);
//   * after a mid-stream barrier the fold of tile i + 1: window (float32 product, mdct.py:410), TDAC fold, split into the
//     three bf16 pieces, written as MFMA A operands into the other LDS buffer.
// K2 (imdct4_b3_kernel): a workgroup walks whole clips, tile by tile in frame order, so the frame in front of a tile is the
// last row of the tile before (kept in a 3-slot LDS ring) -- no halo recomputation.
#pragma once

namespace {

typedef unsigned b3_u4 __attribute__((ext_vector_type(4)));
typedef unsigned b3_u2 __attribute__((ext_vector_type(2)));
typedef __bf16 b3_bf8 __attribute__((ext_vector_type(8)));
typedef int b3_v4i __attribute__((ext_vector_type(4)));

constexpr int B3_ROWS = 32;                     // frames per tile
constexpr int B3_NT = 256;                      // 4 waves
constexpr int B3_BLK = 33 * 16;                 // bytes of one (k-block, k-half) group of the A image: 32 rows x 16 B + 16 B pad
constexpr int B3_PIECE = 32 * B3_BLK;           // one bf16 piece of a tile: 16 k-blocks x 2 halves
constexpr int B3_ABUF = 3 * B3_PIECE;           // 50 688 B
constexpr int B3_RAW = 37 * 1024;               // 33 hop blocks of raw audio (+ 3 slots that only keep the DMA count uniform) + 1 KiB of zeros
constexpr int B3_IMG_U4 = 3 * 4 * 2 * 16 * 64;  // table image: [piece][wave][bin block][k-block][lane] x 16 B = 384 KB

__device__ __forceinline__ unsigned b3_pk(float a, float b) {          // (bf16(a), bf16(b)) round-to-nearest-even, a in the low half
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float b3_lo(unsigned p) { return __uint_as_float(p << 16); }
__device__ __forceinline__ float b3_hi(unsigned p) { return __uint_as_float(p & 0xffff0000u); }
// four float32 -> their three bf16 pieces (4 x bf16 = 8 bytes per piece)
__device__ __forceinline__ void b3_split4(const float4 u, b3_u2& h, b3_u2& m, b3_u2& l) {
    h.x = b3_pk(u.x, u.y); h.y = b3_pk(u.z, u.w);
    const float r0 = u.x - b3_lo(h.x), r1 = u.y - b3_hi(h.x), r2 = u.z - b3_lo(h.y), r3 = u.w - b3_hi(h.y);
    m.x = b3_pk(r0, r1); m.y = b3_pk(r2, r3);
    l.x = b3_pk(r0 - b3_lo(m.x), r1 - b3_hi(m.x)); l.y = b3_pk(r2 - b3_lo(m.y), r3 - b3_hi(m.y));
}
__device__ __forceinline__ f32x16 b3_mfma(b3_u4 a, b3_u4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(b3_bf8, a), __builtin_bit_cast(b3_bf8, b), c, 0, 0, 0);
}
// One LDS-DMA piece: 64 lanes x 16 bytes from the buffer at byte offset voff (per lane) + soff (scalar) (out of range: zeros) to LDS byte address
// lds_dst + 16 * lane.  Inline asm like dense_gemm.h::dg_dma16 (the builtin makes hipcc wait vmcnt(0) before the next ds_read).
__device__ __forceinline__ void b3_dma16(unsigned voff, b3_v4i rsrc, unsigned lds_dst, unsigned soff) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %4 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(rsrc), "s"(lds_dst), "s"(soff)
                 : "memory");
}
__device__ __forceinline__ void b3_wait_vm0() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
template <int N>
__device__ __forceinline__ void b3_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// Table image: img[(((p * 4 + wave) * 2 + blk) * 16 + kb) * 64 + lane] = piece p of D4[bin][16 kb + 8 (lane >> 5) + (0..7)],
// bin = 64 wave + 32 blk + (lane & 31)  (D4 is symmetric: row bin == column bin): the B operand of MFMA step kb, coalesced.
__global__ void dct4_b3_image_kernel(const float* __restrict__ d4, b3_u4* __restrict__ img) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 4 * 2 * 16 * 64) return;
    const int lane = i & 63, kb = (i >> 6) & 15, blk = (i >> 10) & 1, wave = i >> 11;
    const float* src = d4 + (size_t)(64 * wave + 32 * blk + (lane & 31)) * M + 16 * kb + 8 * (lane >> 5);
    b3_u2 h0, m0, l0, h1, m1, l1;
    b3_split4(*reinterpret_cast<const float4*>(src), h0, m0, l0);
    b3_split4(*reinterpret_cast<const float4*>(src + 4), h1, m1, l1);
    constexpr int P = 4 * 2 * 16 * 64;
    img[i] = b3_u4{h0.x, h0.y, h1.x, h1.y};
    img[P + i] = b3_u4{m0.x, m0.y, m1.x, m1.y};
    img[2 * P + i] = b3_u4{l0.x, l0.y, l1.x, l1.y};
}

// The 12 MFMAs of one k-block, smallest terms first; SLOT(s) runs the side work of MFMA slot 12 * kb + s right behind it.
// a3 is used first and a1 last, so that the A pieces of the next k-block can be fetched into the same registers piece by
// piece (a3 after slot 1, a2 after slot 5, a1 after slot 11): 12 registers of A operand, every fetch >= 6 MFMAs ahead.
#define B3_KBLOCK(KB, SLOT)                                                                        \
    acc0 = b3_mfma(a3, bt[0][0][KB], acc0); SLOT(0);  acc1 = b3_mfma(a3, bt[0][1][KB], acc1); SLOT(1);  \
    if (KB + 1 < 16) a3 = *reinterpret_cast<const b3_u4*>(ap + 2 * B3_PIECE + (KB + 1) * 2 * B3_BLK);   \
    acc0 = b3_mfma(a2, bt[1][0][KB], acc0); SLOT(2);  acc1 = b3_mfma(a2, bt[1][1][KB], acc1); SLOT(3);  \
    acc0 = b3_mfma(a2, bt[0][0][KB], acc0); SLOT(4);  acc1 = b3_mfma(a2, bt[0][1][KB], acc1); SLOT(5);  \
    if (KB + 1 < 16) a2 = *reinterpret_cast<const b3_u4*>(ap + B3_PIECE + (KB + 1) * 2 * B3_BLK);       \
    acc0 = b3_mfma(a1, bt[2][0][KB], acc0); SLOT(6);  acc1 = b3_mfma(a1, bt[2][1][KB], acc1); SLOT(7);  \
    acc0 = b3_mfma(a1, bt[1][0][KB], acc0); SLOT(8);  acc1 = b3_mfma(a1, bt[1][1][KB], acc1); SLOT(9);  \
    acc0 = b3_mfma(a1, bt[0][0][KB], acc0); SLOT(10); acc1 = b3_mfma(a1, bt[0][1][KB], acc1); SLOT(11); \
    if (KB + 1 < 16) a1 = *reinterpret_cast<const b3_u4*>(ap + (KB + 1) * 2 * B3_BLK);

// ------------------------------------------------------------------------------------------------------------------
// K1.  grid = workers (<= 256), block = 256; F >= 32.  DBG (ubench only): bit 0 no global stores, 1 no codec arithmetic, 2 neither
// DMA nor fold, 3 no DMA, 4 no fold.  SPEC: write the 1-channel spectrogram; PAIR: write the 2-channel network input
// (v, 2|v| + nr0) -- with PAIR alone the spectrogram is channel 0 of the pair (393 216 B per clip instead of 526 848).
// ------------------------------------------------------------------------------------------------------------------
template <int MODE, bool SPEC, bool PAIR, bool STATS, int DBG = 0, int DS = 6>      // DS: one DMA piece every DS-th slot (from slot 1 on)
__global__ __launch_bounds__(B3_NT) void mdct4_b3_kernel(const float* __restrict__ audio, int B, int T, int F,
                                                         const float* __restrict__ window, const b3_u4* __restrict__ img,
                                                         CodecParams cp, float* __restrict__ spec, float* __restrict__ in2,
                                                         double* __restrict__ stats) {
    constexpr int Q = M / 2;
    extern __shared__ __attribute__((aligned(16))) unsigned char b3_smem[];
    float* ws = reinterpret_cast<float*>(b3_smem);                 // [2 M] window, then [2 M] its negative
    unsigned char* abuf = b3_smem + 4 * M * sizeof(float);         // [2][B3_ABUF] A operand pieces
    float* raw = reinterpret_cast<float*>(abuf + 2 * B3_ABUF);     // [33][256] hop blocks of the tile being folded
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rows = B * F, n_tiles = (rows + B3_ROWS - 1) / B3_ROWS, G = gridDim.x;
    const int col = wave * 64 + (lane & 31), kh = lane >> 5;       // bin of accumulator block 0 (block 1: col + 32)
    const float k1 = (float)(((double)cp.nr1 - (double)cp.nr0) / ((double)cp.mx - (double)cp.mn));
    const float k0 = (float)((double)cp.nr0 - (double)cp.mn * (((double)cp.nr1 - (double)cp.nr0) / ((double)cp.mx - (double)cp.mn)));
    const float gain = cp.gain, nr0 = cp.nr0;
    b3_v4i r_audio;                                                // (a plain V# for the inline-asm DMA)
    {
        const unsigned long long a = (unsigned long long)audio;
        r_audio[0] = (int)(unsigned)a; r_audio[1] = (int)((unsigned)(a >> 32) & 0xffffu);
        r_audio[2] = (int)((unsigned)B * (unsigned)T * 4u); r_audio[3] = 0x00020000;
    }
    const __amdgpu_buffer_rsrc_t r_spec = __builtin_amdgcn_make_buffer_rsrc(spec, 0, (SPEC && !(DBG & 1)) ? (unsigned)rows * M * 4u : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_in2 = __builtin_amdgcn_make_buffer_rsrc(in2, 0, (PAIR && !(DBG & 1)) ? (unsigned)rows * M * 8u : 0u, 0x00020000);
    const unsigned raw_lds = (unsigned)(uintptr_t)raw;             // LDS byte address (the low 32 bits of a __shared__ pointer)

    // Row 0 of the tile in flight as (clip, frame): advanced by 32 G rows per iteration without a division (F >= 32: a tile
    // touches at most two clips, row r is (nb + 1, nf + r - F) when nf + r >= F).  `next` describes the tile being prepared.
    const int adv = B3_ROWS * G, adv_b = adv / F, adv_f = adv - adv_b * F;
    int nb = 0, nf = 0;                                            // set in the prologue
    bool nvalid = false;
    auto advance = [&]() {
        nb += adv_b; nf += adv_f;
        if (nf >= F) { nf -= F; ++nb; }
    };
    // hop block `slot` (0: the first half of row 0; r + 1: the second half of row r; 33..35: nothing) of the tile being
    // prepared, fetched by wave slot % 4.  No branch: an unused slot / a row outside the batch / samples behind the clip's end
    // are out-of-range offsets, which the DMA fills with zeros.
    auto dma_slot = [&](int slot) {
        // everything but the last compare is scalar arithmetic, and the selects are masks -- written with `?:` on these
        // wave-uniform conditions hipcc branches around each piece (18 branches in the MFMA stream)
        const int r = slot == 0 ? 0 : slot - 1;
        const int fr = nf + r;
        const int wrap = (int)(fr >= F);
        const int b = nb + wrap;
        const int f = fr - (wrap ? F : 0) - (slot == 0 ? 1 : 0);
        const unsigned row_ok = (unsigned)nvalid & (unsigned)(slot < 33) & (unsigned)(b < B) & (unsigned)(f >= 0);
        const int lim = row_ok ? (T - f * M) >> 2 : 0;           // lanes [0, lim) hold samples of the clip (T % 4 == 0)
        const unsigned soff = row_ok ? ((unsigned)b * (unsigned)T + (unsigned)(f * M)) * 4u : 0u;
        const unsigned dead = 0u - (unsigned)(lane >= lim);      // all ones for a lane behind the clip's end / an unused piece
        b3_dma16((16u * lane) | dead, r_audio, raw_lds + 1024u * slot, soff);
    };
    // Fold group gi: row r = wave + 4 gi, u[n .. n + 3], n = 4 lane (mdct_bs.h).  Lanes n < Q read the second half of the frame
    // (hop block r + 1) only, lanes n >= Q the first half (hop block r: zero padding when the row is frame 0 of its clip):
    //   n <  Q:  u = -rev(z[3Q-4-n ..]) - z[3Q+n ..]        n >= Q:  u = z[n-Q ..] - rev(z[3Q-4-n ..]),   z = fl32(x * w)
    const int n = 4 * lane;
    const bool lo = n < Q;
    const int o1 = 3 * Q - 4 - n, o2 = lo ? 3 * Q + n : n - Q;     // frame sample index of the reversed / the straight run
    const float* wS = ws + (lo ? 2 * M : 0) + o2;                  // (the straight run's sign rides in the window copy)
    const int s1 = lo ? o1 - M : o1, s2 = lo ? o2 - M : o2;        // the same runs as offsets inside the lane's hop block
    float4 fz1, fz2, fw1, fw2;
    b3_u2 fh, fm, fl;
    auto fold_piece = [&](int pc, int gi, int buf) {
        const int r = wave + 4 * gi;
        if (pc == 0) {
            // frame 0 of a clip: its first half is the zero padding -> the upper lanes read the block of zeros instead (a
            // scalar select of the block index: no branch, no per-lane select)
            const int fr = nf + r;
            const bool first = fr == 0 || fr == F;
            const int slot = lo ? r + 1 : (first ? 36 : r);
            const float* blk = raw + slot * M;
            fz1 = bs_ld4(blk + s1); fz2 = bs_ld4(blk + s2);
        } else if (pc == 1) {
            fw1 = bs_ld4(ws + o1); fw2 = bs_ld4(wS);
        } else if (pc == 2) {       // z = fl32(x * w) (mdct.py:410)
            fz1 = make_float4(__fmul_rn(fz1.x, fw1.x), __fmul_rn(fz1.y, fw1.y), __fmul_rn(fz1.z, fw1.z), __fmul_rn(fz1.w, fw1.w));
            fz2 = make_float4(__fmul_rn(fz2.x, fw2.x), __fmul_rn(fz2.y, fw2.y), __fmul_rn(fz2.z, fw2.z), __fmul_rn(fz2.w, fw2.w));
        } else if (pc == 3) {       // the TDAC fold:  u = (+-z)[straight] - rev(z[reversed])
            fz1 = make_float4(fz2.x - fz1.w, fz2.y - fz1.z, fz2.z - fz1.y, fz2.w - fz1.x);
        } else if (pc == 4) {
            b3_split4(fz1, fh, fm, fl);
        } else {
            // A image: k-block n >> 4, k-half (n >> 3) & 1 -> group n >> 3 = lane >> 1; 8-byte half (n >> 2) & 1 = lane & 1
            unsigned char* dst = abuf + (size_t)buf * B3_ABUF + (lane >> 1) * B3_BLK + r * 16 + (lane & 1) * 8;
            *reinterpret_cast<b3_u2*>(dst) = fh;
            *reinterpret_cast<b3_u2*>(dst + B3_PIECE) = fm;
            *reinterpret_cast<b3_u2*>(dst + 2 * B3_PIECE) = fl;
        }
    };
    // epilogue of the previous tile: value v = 16 blk + reg of this lane: frame row em0 + (r & 3) + 8 (r >> 2), bin col + 32 blk
    float ea, ea2, et, ep, esq, esm, el, ev, elog = 0.0f;
    double sd1 = 0.0, sd2 = 0.0;
    float f1 = 0.0f, f2 = 0.0f;
    int em0 = 0;
    unsigned eob = BS_OOB, eob2 = BS_OOB;
    auto epi_piece = [&](int pc, float xv, int v) {
        const int r = v & 15, blk = v >> 4;
        const int dm = (r & 3) + 8 * (r >> 2);
        if (MODE == CODEC_RAW || (DBG & 2)) {
            if (pc == 0) ev = xv;
        } else if (pc == 0) {
            const float y = gain * xv;
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
            ev = fmaf(elog, k1, k0);          // (l - min) / (max - min) * (nr1 - nr0) + nr0, constants folded in double (mdct_bs.h)
        }
        if (STATS && pc == 5 && MODE != CODEC_RAW) {
            const float l = (em0 + dm < rows) ? elog : 0.0f;
            f1 += l; f2 = fmaf(l, l, f2);
        }
        if (pc == 6 && SPEC) {
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(ev), r_spec, eob + 128u * blk, dm * (M * 4), 0);
        } else if (pc == 7 && PAIR) {
            const bs_v2u pr = {__float_as_uint(ev), __float_as_uint(fmaf(fabsf(ev), 2.0f, nr0))};
            __builtin_amdgcn_raw_buffer_store_b64(pr, r_in2, eob2 + 256u * blk, dm * (M * 8), 0);
        }
    };
    auto epi_begin = [&](int tile) {         // tile == n_tiles: nothing to store
        if (STATS && MODE != CODEC_RAW) { sd1 += (double)f1; sd2 += (double)f2; f1 = f2 = 0.0f; }
        const unsigned m0 = (unsigned)tile * B3_ROWS + 4u * kh;
        eob = (tile < n_tiles) ? (m0 * M + (unsigned)col) * 4u : BS_OOB - 31u * M * 8u - 256u;
        eob2 = (tile < n_tiles) ? 2u * eob : eob;
        em0 = (tile < n_tiles) ? (int)m0 : rows;
    };

    // prologue: raw audio of the first tile by DMA, the window, the table (384 registers), then the first fold
    int tile = blockIdx.x;
    {
        const int m0 = tile * B3_ROWS;
        nb = m0 / F; nf = m0 - nb * F; nvalid = tile < n_tiles;
    }
#pragma unroll
    for (int i = 0; i < 9; ++i) dma_slot(wave + 4 * i);
    if (tid < 64) reinterpret_cast<float4*>(raw + 36 * M)[tid] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (tid < 2 * M / 4) {
        const float4 wv = bs_ld4(window + 4 * tid);
        reinterpret_cast<float4*>(ws)[tid] = wv;
        reinterpret_cast<float4*>(ws + 2 * M)[tid] = make_float4(-wv.x, -wv.y, -wv.z, -wv.w);
    }
    b3_u4 bt[3][2][16];
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
        for (int bk = 0; bk < 2; ++bk)
#pragma unroll
            for (int kb = 0; kb < 16; ++kb) bt[p][bk][kb] = img[(((p * 4 + wave) * 2 + bk) * 16 + kb) * 64 + lane];
    b3_wait_vm0();
    __syncthreads();
#pragma unroll
    for (int gi = 0; gi < 8; ++gi)
#pragma unroll
        for (int pc = 0; pc < 6; ++pc) fold_piece(pc, gi, 0);
    __syncthreads();

    f32x16 accp0 = f32x16{0}, accp1 = f32x16{0};
    int buf = 0;
    epi_begin(n_tiles);
    for (; tile < n_tiles; tile += G, buf ^= 1) {
        advance();                                                           // (nb, nf): row 0 of tile + G, the one being prepared
        nvalid = tile + G < n_tiles;                                         // (behind the last tile: every row out of range -> zeros)
        f32x16 acc0 = f32x16{0}, acc1 = f32x16{0};
        const unsigned char* ap = abuf + (size_t)buf * B3_ABUF + kh * B3_BLK + (lane & 31) * 16;
        b3_u4 a1 = *reinterpret_cast<const b3_u4*>(ap), a2 = *reinterpret_cast<const b3_u4*>(ap + B3_PIECE),
              a3 = *reinterpret_cast<const b3_u4*>(ap + 2 * B3_PIECE);
        __builtin_amdgcn_sched_barrier(0);
        // first half of the stream (k-blocks 0..9, 120 slots): per slot two codec / store pieces of the previous tile's 32 values
        // (value v in slots 4 v' ..: 32 values x 8 pieces = 256 pieces over 128 slots -> k-blocks 0..10 carry them), and one
        // DMA piece every third slot from slot 1 on
#define B3_SLOT_A(KB, s)                                                                                         \
        do {                                                                                                       \
            constexpr int slot = 12 * (KB) + (s);                                                                  \
            if (slot < 128) {                                                                                      \
                constexpr int v = slot >> 2, p0 = 2 * (slot & 3);                                                  \
                const float xv = v < 16 ? accp0[v & 15] : accp1[v & 15];                                           \
                epi_piece(p0, xv, v); epi_piece(p0 + 1, xv, v);                                                    \
            }                                                                                                      \
            if (!(DBG & 4) && !(DBG & 8) && slot % DS == 1 % DS && slot / DS < 9) dma_slot(wave + 4 * (slot / DS));     \
            __builtin_amdgcn_sched_barrier(0);                                                                     \
        } while (0)
#define SLOT0(s) B3_SLOT_A(0, s)
#define SLOT1(s) B3_SLOT_A(1, s)
#define SLOT2(s) B3_SLOT_A(2, s)
#define SLOT3(s) B3_SLOT_A(3, s)
#define SLOT4(s) B3_SLOT_A(4, s)
#define SLOT5(s) B3_SLOT_A(5, s)
#define SLOT6(s) B3_SLOT_A(6, s)
#define SLOT7(s) B3_SLOT_A(7, s)
#define SLOT8(s) B3_SLOT_A(8, s)
#define SLOT9(s) B3_SLOT_A(9, s)
        B3_KBLOCK(0, SLOT0) B3_KBLOCK(1, SLOT1) B3_KBLOCK(2, SLOT2) B3_KBLOCK(3, SLOT3) B3_KBLOCK(4, SLOT4)
        B3_KBLOCK(5, SLOT5) B3_KBLOCK(6, SLOT6) B3_KBLOCK(7, SLOT7) B3_KBLOCK(8, SLOT8) B3_KBLOCK(9, SLOT9)
        // k-block 10: the last 8 epilogue slots (120..127); then the raw audio of the next tile must have landed everywhere
#define SLOT10(s) B3_SLOT_A(10, s)
        B3_KBLOCK(10, SLOT10)
        // The DMA pieces were issued in slots 1 .. 25; vmcnt retires in order on gfx9 (loads and stores alike), so "at most as many
        // operations outstanding as were issued after the last piece" means every piece has landed -- WITHOUT draining the
        // codec's stores of slots 27 .. 127 (values 6 .. 31: 26 per output tensor), which vmcnt(0) would wait for (measured:
        // +250 us per 4096 clips).  The counter saturates at 63.
        {
            constexpr int last_dma = 1 % DS + 8 * DS;                      // slot of the last piece
            constexpr int first_after = (last_dma - 3 + 4) / 4 + ((last_dma - 3) % 4 == 0 ? 1 : 0);   // first value whose store slot 4 v + 3 lies behind it
            constexpr int after = (32 - (first_after < 0 ? 0 : first_after)) * ((SPEC ? 1 : 0) + (PAIR ? 1 : 0));
            static_assert(DS != 6 || after == 20 * ((SPEC ? 1 : 0) + (PAIR ? 1 : 0)), "count");
            b3_wait_vm<(after < 63 ? after : 63)>();
        }
        __syncthreads();
        // second half (k-blocks 11..15, 60 slots): the fold of the next tile, 8 groups x 6 pieces = 48 pieces
#define B3_SLOT_B(KB, s)                                                                                         \
        do {                                                                                                       \
            constexpr int q = 12 * ((KB) - 11) + (s);                                                              \
            if (!(DBG & 4) && !(DBG & 16) && q < 48) fold_piece(q % 6, q / 6, buf ^ 1);                            \
            __builtin_amdgcn_sched_barrier(0);                                                                     \
        } while (0)
#define SLOT11(s) B3_SLOT_B(11, s)
#define SLOT12(s) B3_SLOT_B(12, s)
#define SLOT13(s) B3_SLOT_B(13, s)
#define SLOT14(s) B3_SLOT_B(14, s)
#define SLOT15(s) B3_SLOT_B(15, s)
        B3_KBLOCK(11, SLOT11) B3_KBLOCK(12, SLOT12) B3_KBLOCK(13, SLOT13) B3_KBLOCK(14, SLOT14) B3_KBLOCK(15, SLOT15)
#undef SLOT0
#undef SLOT1
#undef SLOT2
#undef SLOT3
#undef SLOT4
#undef SLOT5
#undef SLOT6
#undef SLOT7
#undef SLOT8
#undef SLOT9
#undef SLOT10
#undef SLOT11
#undef SLOT12
#undef SLOT13
#undef SLOT14
#undef SLOT15
#undef B3_SLOT_A
#undef B3_SLOT_B
        accp0 = acc0; accp1 = acc1;
        epi_begin(tile);
        __syncthreads();
    }
#pragma unroll
    for (int v = 0; v < 32; ++v)
#pragma unroll
        for (int pc = 0; pc < 8; ++pc) epi_piece(pc, v < 16 ? accp0[v & 15] : accp1[v & 15], v);
    if (STATS && MODE != CODEC_RAW) {
        sd1 += (double)f1; sd2 += (double)f2;
        sd1 = wave_sum_d(sd1); sd2 = wave_sum_d(sd2);
        if (lane == 0) { atomicAdd(stats, sd1); atomicAdd(stats + 1, sd2); }
    }
}

constexpr size_t B3_K1_LDS = (size_t)4 * M * sizeof(float) + 2 * B3_ABUF + B3_RAW;

// ------------------------------------------------------------------------------------------------------------------
// K2.  grid = workers (<= 256, <= B), block = 256.  A workgroup walks whole clips b = blockIdx.x, + G, ..., each tile by tile in
// frame order: tile = frames f0 .. f0 + 31, emits hop blocks h = f0 .. f0 + 31:
//   out[(h - 1) M + n] = 4 / N * (w[n] y_h[n] + w[n + M] y_{h-1}[n + M]),  y = [v2, -v2_r, -v1_r, -v1] of v = DCT-IV(X);
// frame f0 - 1 is row 31 of the tile before (same clip, kept in a 2-slot LDS ring; hop block 0 of a clip is cropped anyway).
// Per tile behind the 192 MFMAs: the unfold / window / overlap-add / store of tile i - 1 out of the v tile in LDS, the loads
// (two batches of four float4 per thread) and the decode (denormalise, sinh) + bf16 split of tile i + 1 into the other
// operand buffer.  Two barriers per tile: every wave has finished reading v before the new v tile is written.
// ------------------------------------------------------------------------------------------------------------------
template <int MODE, int DBG = 0>
__global__ __launch_bounds__(B3_NT) void imdct4_b3_kernel(const float* __restrict__ spec, int B, int F,
                                                          const float* __restrict__ window, const b3_u4* __restrict__ img,
                                                          CodecParams cp, float* __restrict__ audio, int out_len) {
    constexpr int Q = M / 2;
    constexpr int V_F = B3_ROWS * BS_LDA;                          // v tile: [32][M + 4] floats
    extern __shared__ __attribute__((aligned(16))) unsigned char b3_smem[];
    unsigned char* abuf = b3_smem;                                 // [2][B3_ABUF] decoded coefficients as bf16 pieces
    float* vbuf = reinterpret_cast<float*>(abuf + 2 * B3_ABUF);    // [V_F] v = DCT-IV(X) of the tile before
    float* halo = vbuf + V_F;                                      // [2][M] row 31 of the last two tiles
    float* wl = halo + 2 * M;                                      // [2 M] window
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tpc = (F + B3_ROWS - 1) / B3_ROWS, G = gridDim.x;
    const int my_clips = ((int)blockIdx.x < B) ? (B - 1 - (int)blockIdx.x) / G + 1 : 0;
    const int n_seq = my_clips * tpc;                              // this workgroup's tiles, clip by clip, frames ascending
    const int col = wave * 64 + (lane & 31), kh = lane >> 5;
    const __amdgpu_buffer_rsrc_t r_spec = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(spec), 0, (unsigned)B * (unsigned)F * M * 4u, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_out = __builtin_amdgcn_make_buffer_rsrc(audio, 0, (DBG & 1) ? 0u : (unsigned)B * (unsigned)out_len * 4u, 0x00020000);
    // the tile being prepared (decoded) as (clip, tile of the clip): advanced without divisions; behind the last clip: nx_clip >= B
    int nx_clip = blockIdx.x, nx_t = 0;
    auto nx_advance = [&]() {
        if (++nx_t == tpc) { nx_t = 0; nx_clip += G; }
    };

    // x = v * c1 + c0 (= ln10 * ((v - nr0) / (nr1 - nr0) * (max - min) + min)), X = sinh(x) / gain; constants in double per tile
    float c1 = 1.0f, c0 = 0.0f;
    const float rgain = 1.0f / cp.gain;
    auto tile_consts = [&]() {
        const int b = nx_clip < B ? nx_clip : 0;
        float mn = cp.mn, mx = cp.mx;
        if (cp.per_sample) { mn = cp.mn_b[b]; mx = cp.mx_b[b]; }
        const double k = ((double)mx - (double)mn) / ((double)cp.nr1 - (double)cp.nr0);
        const double sc = (MODE == CODEC_ARCSINH) ? (double)LN10F : 1.0;
        c1 = (float)(k * sc);
        c0 = (float)(((double)mn - (double)cp.nr0 * k) * sc);
    };
    bs_v4u xr[8];
    int nf0 = 0;                                                 // first frame of the tile being prepared, F when there is none
    auto load_tile = [&]() {                                     // decode group g: row wave + 4 g, coefficients 4 lane ..+3
        const int b = nx_clip < B ? nx_clip : 0;
        nf0 = nx_clip < B ? nx_t * B3_ROWS : F;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int f = nf0 + wave + 4 * q;                    // (wave-uniform: the validity test is scalar; a mask, not a branch)
            const unsigned dead = 0u - (unsigned)(f >= F);
            xr[q] = __builtin_amdgcn_raw_buffer_load_b128(r_spec, ((((unsigned)b * F + (unsigned)f) * M + 4u * lane) * 4u) | dead, 0, 0);
        }
    };
    auto dec1 = [&](float v) -> float {
        if (MODE == CODEC_RAW) return v;
        const float x = fmaf(v, c1, c0);
        if (MODE == CODEC_ARCSINH) return sinh_fast(x) * rgain;
        return x;
    };
    float4 dz;
    b3_u2 dh, dm_, dl;
    auto decode_piece = [&](int pc, int g, int buf) {            // group g (0..7) of the tile being prepared
        const float4 x = __builtin_bit_cast(float4, xr[g]);
        // frames outside the clip contribute nothing (decode(0) != 0): an AND with a scalar mask (`ok ? dec1(x) : 0` on this
        // wave-uniform condition compiles to a branch around the decode)
        const unsigned live = 0u - (unsigned)(nf0 + wave + 4 * g < F);
        auto keep = [&](float v) { return __uint_as_float(__float_as_uint(v) & live); };
        if (pc == 0) dz.x = keep(dec1(x.x));
        else if (pc == 1) dz.y = keep(dec1(x.y));
        else if (pc == 2) dz.z = keep(dec1(x.z));
        else if (pc == 3) dz.w = keep(dec1(x.w));
        else if (pc == 4) b3_split4(dz, dh, dm_, dl);
        else {
            unsigned char* dst = abuf + (size_t)buf * B3_ABUF + (lane >> 1) * B3_BLK + (wave + 4 * g) * 16 + (lane & 1) * 8;
            *reinterpret_cast<b3_u2*>(dst) = dh;
            *reinterpret_cast<b3_u2*>(dst + B3_PIECE) = dm_;
            *reinterpret_cast<b3_u2*>(dst + 2 * B3_PIECE) = dl;
        }
    };
    // unfold of the previous tile, group gi: hop block j = wave + 4 gi, samples n = 4 lane ..+3
    const float scale = 4.0f / (2 * M);
    const int n = 4 * lane;
    const bool lo = n < Q;
    unsigned pbase = 0;         // element index of (clip, sample (f0 - 1) M + n) of the previous tile in the output
    int pf0 = 0, phalo = 0;
    bool pvalid = false;
    float4 uc, up;
    auto unfold_piece = [&](int pc, int gi) {
        const int j = wave + 4 * gi;
        const float* vc = vbuf + j * BS_LDA;                                       // frame h
        const float* vp = (j == 0) ? halo + phalo * M : vbuf + (j - 1) * BS_LDA;   // frame h - 1
        if (pc == 0) {
            // y_h[n] = v_h[Q + n] | -v_h[3Q - 1 - n];   y_{h-1}[n + M] = -v_{h-1}[Q - 1 - n] | -v_{h-1}[n - Q]
            uc = bs_ld4(vc + (lo ? Q + n : 3 * Q - 4 - n));
            up = bs_ld4(vp + (lo ? Q - 4 - n : n - Q));
        } else if (pc == 1) {
            const float4 c = uc, q = up;
            uc = lo ? c : make_float4(-c.w, -c.z, -c.y, -c.x);
            up = lo ? make_float4(-q.w, -q.z, -q.y, -q.x) : make_float4(-q.x, -q.y, -q.z, -q.w);
        } else if (pc == 2) {
            const float4 uw0 = bs_ld4(wl + n), uw1 = bs_ld4(wl + n + M);      // (the window from LDS: the register file has no room for it)
            uc = make_float4(scale * (uw0.x * uc.x + uw1.x * up.x), scale * (uw0.y * uc.y + uw1.y * up.y),
                             scale * (uw0.z * uc.z + uw1.z * up.z), scale * (uw0.w * uc.w + uw1.w * up.w));
        } else {
            const int hh = pf0 + j, t0 = (hh - 1) * M + n;       // out_len % 4 == 0: a float4 is inside or outside the crop as a whole
            const bool ok = pvalid && hh >= 1 && hh <= F - 1 && t0 + 3 < out_len;
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(bs_v4u, uc), r_out, ok ? (pbase + (unsigned)(j * M)) * 4u : BS_OOB, 0, 0);
        }
    };

    // prologue: decode the first tile, then fetch the table
    tile_consts();
    load_tile();
#pragma unroll
    for (int g = 0; g < 8; ++g)
#pragma unroll
        for (int pc = 0; pc < 6; ++pc) { decode_piece(pc, g, 0); __builtin_amdgcn_sched_barrier(0); }
    b3_u4 bt[3][2][16];
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
        for (int bk = 0; bk < 2; ++bk)
#pragma unroll
            for (int kb = 0; kb < 16; ++kb) bt[p][bk][kb] = img[(((p * 4 + wave) * 2 + bk) * 16 + kb) * 64 + lane];
    if (tid < M) { halo[tid] = 0.0f; halo[M + tid] = 0.0f; }
    if (tid < 2 * M / 4) reinterpret_cast<float4*>(wl)[tid] = bs_ld4(window + 4 * tid);
    __syncthreads();

    int buf = 0;
    for (int i = 0; i < n_seq; ++i, buf ^= 1) {
        const int cur_clip = nx_clip, cur_f0 = nx_t * B3_ROWS;
        nx_advance();
        tile_consts();
        f32x16 acc0 = f32x16{0}, acc1 = f32x16{0};
        const unsigned char* ap = abuf + (size_t)buf * B3_ABUF + kh * B3_BLK + (lane & 31) * 16;
        b3_u4 a1 = *reinterpret_cast<const b3_u4*>(ap), a2 = *reinterpret_cast<const b3_u4*>(ap + B3_PIECE),
              a3 = *reinterpret_cast<const b3_u4*>(ap + 2 * B3_PIECE);
        __builtin_amdgcn_sched_barrier(0);
        // slot 1: the 8 loads of the next tile (1.2 us ahead of their first use); even slots 0..62: the 32 unfold pieces of the
        // previous tile; even slots 96..190: the 48 decode pieces of the next tile
#define B3_SLOT_K2(KB, s)                                                                                        \
        do {                                                                                                       \
            constexpr int slot = 12 * (KB) + (s);                                                                  \
            if (slot < 64 && (slot & 1) == 0) unfold_piece((slot >> 1) & 3, slot >> 3);                            \
            if (slot == 1) load_tile();                                                                            \
            if (slot >= 96 && (slot & 1) == 0) decode_piece(((slot - 96) >> 1) % 6, ((slot - 96) >> 1) / 6, buf ^ 1); \
            __builtin_amdgcn_sched_barrier(0);                                                                     \
        } while (0)
#define SLOT0(s) B3_SLOT_K2(0, s)
#define SLOT1(s) B3_SLOT_K2(1, s)
#define SLOT2(s) B3_SLOT_K2(2, s)
#define SLOT3(s) B3_SLOT_K2(3, s)
#define SLOT4(s) B3_SLOT_K2(4, s)
#define SLOT5(s) B3_SLOT_K2(5, s)
#define SLOT6(s) B3_SLOT_K2(6, s)
#define SLOT7(s) B3_SLOT_K2(7, s)
#define SLOT8(s) B3_SLOT_K2(8, s)
#define SLOT9(s) B3_SLOT_K2(9, s)
#define SLOT10(s) B3_SLOT_K2(10, s)
#define SLOT11(s) B3_SLOT_K2(11, s)
#define SLOT12(s) B3_SLOT_K2(12, s)
#define SLOT13(s) B3_SLOT_K2(13, s)
#define SLOT14(s) B3_SLOT_K2(14, s)
#define SLOT15(s) B3_SLOT_K2(15, s)
        B3_KBLOCK(0, SLOT0) B3_KBLOCK(1, SLOT1) B3_KBLOCK(2, SLOT2) B3_KBLOCK(3, SLOT3) B3_KBLOCK(4, SLOT4) B3_KBLOCK(5, SLOT5)
        B3_KBLOCK(6, SLOT6) B3_KBLOCK(7, SLOT7) B3_KBLOCK(8, SLOT8) B3_KBLOCK(9, SLOT9) B3_KBLOCK(10, SLOT10) B3_KBLOCK(11, SLOT11)
        B3_KBLOCK(12, SLOT12) B3_KBLOCK(13, SLOT13) B3_KBLOCK(14, SLOT14) B3_KBLOCK(15, SLOT15)
#undef SLOT0
#undef SLOT1
#undef SLOT2
#undef SLOT3
#undef SLOT4
#undef SLOT5
#undef SLOT6
#undef SLOT7
#undef SLOT8
#undef SLOT9
#undef SLOT10
#undef SLOT11
#undef SLOT12
#undef SLOT13
#undef SLOT14
#undef SLOT15
#undef B3_SLOT_K2
        __syncthreads();                           // every wave is done with the previous v tile (and with abuf[buf])
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = mfma32_row(r, lane);
            vbuf[row * BS_LDA + col] = acc0[r];
            vbuf[row * BS_LDA + col + 32] = acc1[r];
        }
        if (kh == 1) { halo[(i & 1) * M + col] = acc0[15]; halo[(i & 1) * M + col + 32] = acc1[15]; }      // row 31 = reg 15 of the upper half
        {
            const int pb = cur_clip;
            pf0 = cur_f0;
            pbase = (unsigned)pb * (unsigned)out_len + (unsigned)((pf0 - 1) * M + n);      // (wraps for pf0 == 0: that block is masked)
            phalo = (i + 1) & 1;                   // the tile before this one wrote slot (i - 1) & 1
            pvalid = true;
        }
        __syncthreads();
    }
    // drain: the last tile's unfold
#pragma unroll
    for (int gi = 0; gi < 8; ++gi)
#pragma unroll
        for (int pc = 0; pc < 4; ++pc) unfold_piece(pc, gi);
}

constexpr size_t B3_K2_LDS = (size_t)2 * B3_ABUF + (size_t)(B3_ROWS * BS_LDA + 4 * M) * sizeof(float);

}  // namespace
