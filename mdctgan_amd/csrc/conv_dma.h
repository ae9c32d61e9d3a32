// Implicit-GEMM convolution passes with LDS-DMA staging (float32 precision): the direct (non-Winograd) layers -- the
// stride-2 3x3 ladder of the generator and its transposed twins, the 4x4 PatchGAN layers below the Winograd thresholds --
// on the structure of dgemm32g_kernel (dense_gemm.h): 32-deep chunks, unpadded XOR-swizzled LDS images, buffer_load ... lds
// from inline asm with counted vmcnt, two buffers, one barrier per chunk.
//   The gather is folded into the DMA's per-lane byte offset: a row of the A tile is the 128-byte run of 32 channels of ONE
// source pixel (NHWC), so a lane's offset is  pixel(row, tap) * C * 4 + 16 * quad  and only changes when the tap changes
// (the chunk's channel offset is the instruction's scalar offset).  A tap that falls into the zero padding gets an offset
// behind the buffer's num_records: the hardware range check returns zeros -- no branches, no masking instructions.
// Reflection padding reflects the pixel index instead.
//   FWD    Y[m=(b,oy,ox)][co]  = sum_{tap,ci} X[pixel(m, tap)][ci] * W[co][tap][ci]      A: gathered, B: dense k-contiguous
//   WGRAD  dW[co][(tap,ci)]    = sum_{m}      dY[m][co]          * X[pixel(m, tap)][ci]   A: dense rows, B: gathered rows
//          (a 64-column tile lies inside one tap when Ci % 64 == 0: the tap is a per-workgroup constant)
// Eligibility (conv_dma_ok): MG_PRECISION_F32, Ci % 32 == 0 (WGRAD: Ci % 64 == 0), Co % 64 == 0, tensors < 2 GiB.
#pragma once

constexpr unsigned CD_OOB = 0x80000000u;      // >= any num_records: the DMA lane reads zeros

// HALF instances (autocast, MG_PRECISION_F16): the same kernels over float16 copies of the activations and weights (cast
// pre-passes, conv_igemm.hip) on v_mfma_f32_32x32x16_f16.  A chunk row is still 128 bytes -- 64 channels -- so the gather,
// the LDS images of the k-contiguous operands and the pipeline are unchanged; ds_read_b128 yields the 8-half MFMA operand.
// Row-contiguous operands (the weight gradient's two sides, the data gradient's weights) are k-major LDS images [64][R]
// halves read with ds_read_b64_tr_b16 (16 lanes fetch a [4 k][16 columns] block, every lane receives the 4 k of its
// column): two reads are one MFMA operand.  The 16-byte slots of a k row are XOR-swizzled on the DMA source side so that
// the 8 row segments a 32-lane half reads fall into 8 different 32-byte bank groups.
template <bool HALF>
struct CdElem {
    static constexpr unsigned ES = HALF ? 2u : 4u;      // element bytes
    static constexpr int CK = HALF ? 64 : 32;           // K depth of one chunk (128-byte rows)
};
template <int MB, int NB, int ALAY, int BLAY, int BM, int BN>
__device__ __forceinline__ void cd_chunk_h(const float* As, const float* Bs, f32x16 (&acc)[MB][NB], int wm0, int wn0, int lane) {
    const int r = lane & 31, kh = lane >> 5;
    f16x8 a[2][MB], b[2][NB];
    auto fetch = [&](int s, int buf) {
#pragma unroll
        for (int mi = 0; mi < MB; ++mi)
            a[buf][mi] = ALAY == DG_KC ? cd_frag_kc(As, wm0 + 32 * mi + r, s, kh) : cd_frag_rc<BM>(As, wm0 + 32 * mi, s, lane);
#pragma unroll
        for (int ni = 0; ni < NB; ++ni)
            b[buf][ni] = BLAY == DG_KC ? cd_frag_kc(Bs, wn0 + 32 * ni + r, s, kh) : cd_frag_rc<BN>(Bs, wn0 + 32 * ni, s, lane);
    };
    fetch(0, 0);
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        if (s + 1 < 4) fetch(s + 1, (s + 1) & 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int mi = 0; mi < MB; ++mi)
#pragma unroll
            for (int ni = 0; ni < NB; ++ni) acc[mi][ni] = mfma32x32x16h(a[s & 1][mi], b[s & 1][ni], acc[mi][ni]);
        __builtin_amdgcn_sched_barrier(0);
    }
}
template <bool HALF, int MB, int NB, int ALAY, int BLAY, int BM, int BN>
__device__ __forceinline__ void cd_chunk(const float* As, const float* Bs, f32x16 (&acc)[MB][NB], int wm0, int wn0, int lane) {
    if constexpr (HALF) cd_chunk_h<MB, NB, ALAY, BLAY, BM, BN>(As, Bs, acc, wm0, wn0, lane);
    else dg_chunk_g<MB, NB, ALAY, BLAY, BM, BN>(As, Bs, acc, wm0, wn0, lane);
}
// byte offset of the 16 bytes lane `lane` fetches for piece `piece` of a row-contiguous operand tile ([K][R], ld elements
// between k rows): k row inside the chunk and source column (swizzled for the float16 transpose reads)
template <bool HALF, int R>
__device__ __forceinline__ void cd_rc_lane(int piece, int lane, int& k, int& col) {
    constexpr int EPL = HALF ? 8 : 4, LPR = R / EPL;        // elements per lane, lanes per k row
    k = piece * (64 / LPR) + lane / LPR;
    const int j = lane % LPR;
    col = EPL * (HALF ? (j ^ cd_rc_swz<R>(k)) : j);
}

struct CdArgs {
    const void* x;           // FWD: input [B,H,W,Ci];  WGRAD: input x;  DGRAD: dy           (float32, HALF: float16)
    const void* w;           // FWD: weights [Co][KH*KW*Ci];  WGRAD: dy [B,OH,OW,Co];  DGRAD: weights
    const float* bias;
    float* y;                // FWD: output [M][Co];  WGRAD: dW [Co][KH*KW*Ci]
    float* part;             // split-K slabs
    int B, H, W, Ci, OH, OW, Co, KH, KW, s, p, reflect;
    int act, tiles_m, tiles_n, splits, cps, accumulate;
    int round_f16;           // HALF: round the direct result through float16 (autocast output)
    int cls_order;           // DGRAD, stride 2: 1 = classes in the order (all taps, half, one tap, half), see the kernel
    int gm;                  // FWD: tile order inside a K split: groups of gm row tiles x all column tiles, row tiles fastest
                             // (0: row tiles fastest over the whole split).  Chosen on the host by cd_pick_gm.
};

// The tile order decides what an XCD's private 4 MB L2 can share: an XCD is handed a contiguous run of the linear tile index
// (xcd_remap) and runs ~64 of its tiles at a time, which walk K together -- an operand panel is fetched once per WINDOW of
// concurrent tiles that use it.  With row tiles fastest over the whole launch, an XCD of the batch-64 512 -> 1024 ladder rung
// (64 x 8 tiles) runs one column of 64 row tiles: the weight panel is shared, but every XCD pulls the WHOLE input through its
// L2 -- measured 1.20 GB per launch against 119 MB algorithmic.  cd_pick_gm evaluates that window model for a few group sizes and
// returns the cheapest (0 = keep the plain order).  a_bytes / b_bytes: what one tile reads of either operand per K split.
inline int cd_pick_gm(int tiles_m, int tiles_n, int splits, double a_bytes, double b_bytes) {
    const long long tiles = (long long)tiles_m * tiles_n, total = tiles * splits;
    if (tiles_m < 2 || tiles_n < 2 || total > (1 << 16)) return 0;
    // (memoised: the model walks every tile of the launch -- once per shape, not once per call)
    static std::mutex mu;
    static std::map<std::tuple<int, int, int, long long, long long>, int> memo;
    const auto key = std::make_tuple(tiles_m, tiles_n, splits, (long long)a_bytes, (long long)b_bytes);
    {
        std::lock_guard<std::mutex> lock(mu);
        auto it = memo.find(key);
        if (it != memo.end()) return it->second;
    }
    constexpr int WINDOW = 64;
    auto cost = [&](int gm) {
        double c = 0.0;
        const long long q = total / 8, r = total % 8;
        long long base = 0;
        for (int x = 0; x < 8; ++x) {
            const long long cnt = q + (x < r ? 1 : 0);
            for (long long w0 = 0; w0 < cnt; w0 += WINDOW) {
                unsigned long long seen_m[64] = {0}, seen_n[64] = {0};     // (split, tile) bitmaps, up to 4096 each
                int um = 0, un = 0;
                const long long w1 = w0 + WINDOW < cnt ? w0 + WINDOW : cnt;
                for (long long i = w0; i < w1; ++i) {
                    const long long L = base + i;
                    const int sp = (int)(L / tiles);
                    const int rem = (int)(L - (long long)sp * tiles);
                    int tn, tm;
                    if (gm > 0) {
                        const int per = gm * tiles_n, grp = rem / per, r2 = rem - grp * per;
                        const int sz = tiles_m - grp * gm < gm ? tiles_m - grp * gm : gm;
                        tn = r2 / sz; tm = grp * gm + (r2 - tn * sz);
                    } else {
                        tn = rem / tiles_m; tm = rem - tn * tiles_m;
                    }
                    const unsigned km = (unsigned)(sp * tiles_m + tm) & 4095u, kn = (unsigned)(sp * tiles_n + tn) & 4095u;
                    if (!(seen_m[km >> 6] >> (km & 63) & 1ull)) { seen_m[km >> 6] |= 1ull << (km & 63); ++um; }
                    if (!(seen_n[kn >> 6] >> (kn & 63) & 1ull)) { seen_n[kn >> 6] |= 1ull << (kn & 63); ++un; }
                }
                c += um * a_bytes + un * b_bytes;
            }
            base += cnt;
        }
        return c;
    };
    int best = 0;
    double best_c = cost(0);
    for (int gm : {1, 2, 4, 8, 16, 32}) {
        if (gm >= tiles_m) break;
        const double c = cost(gm);
        if (c < 0.9 * best_c) { best_c = c; best = gm; }       // (a tenth better or the plain order stays)
    }
    std::lock_guard<std::mutex> lock(mu);
    memo[key] = best;
    return best;
}

// K-split boundaries: split j of S covers the chunks [total * j / S, total * (j + 1) / S) -- any S <= total works and the splits
// differ by at most one chunk (a ceil(total / S) stride leaves a short or an empty last split).
__device__ __forceinline__ int cd_split_bound(int total, int S, int j) {
    return j <= 0 ? 0 : (j >= S ? total : (int)(((long long)total * j) / S));
}
__device__ __forceinline__ unsigned cd_pixel_off(const CdArgs& g, int pb, int iy, int ix, unsigned cbytes) {
    if (g.reflect) {
        iy = reflect_idx(iy, g.H);
        ix = reflect_idx(ix, g.W);
    } else if ((unsigned)iy >= (unsigned)g.H || (unsigned)ix >= (unsigned)g.W) {
        return CD_OOB;
    }
    return (unsigned)(pb + iy * g.W + ix) * cbytes;
}

// ------------------------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------------------------
template <int BM, int BN, bool HALF = false, int NBUF = 2>
__global__ __launch_bounds__(256) void conv_fwd_dma_kernel(CdArgs g) {
#if defined(__HIP_DEVICE_COMPILE__)
    using Cfg = DgCfgG<BM, BN, 2, 2, DG_KC, DG_KC, NBUF>;
    constexpr unsigned ES = CdElem<HALF>::ES;
    constexpr int CK = CdElem<HALF>::CK;
    constexpr int MB = Cfg::MB, NB = Cfg::NB, PA = Cfg::PA, PB = Cfg::PB;
    extern __shared__ __attribute__((aligned(1024))) float cd_smem[];
    float* As0 = cd_smem;
    float* Bs0 = cd_smem + NBUF * Cfg::ASZ;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int M = g.B * g.OH * g.OW, N = g.Co, KT = g.KH * g.KW, K = KT * g.Ci;
    const int tiles = g.tiles_m * g.tiles_n;
    const int L = xcd_remap(blockIdx.x, tiles * g.splits);
    const int sp = L / tiles, rem = L - sp * tiles;
    int tn, tm;
    if (g.gm > 0) {                                                 // groups of gm row tiles x all column tiles (cd_pick_gm)
        const int per = g.gm * g.tiles_n, grp = rem / per, r2 = rem - grp * per;
        const int sz = min(g.gm, g.tiles_m - grp * g.gm);
        tn = r2 / sz;
        tm = grp * g.gm + (r2 - tn * sz);
    } else {
        tn = rem / g.tiles_m;                                       // consecutive tiles share the weight panel
        tm = rem - tn * g.tiles_m;
    }
    const int m0 = tm * BM, n0 = tn * BN;
    const int cpt = g.Ci / CK, total_chunks = KT * cpt;
    const int c_begin = cd_split_bound(total_chunks, g.splits, sp), c_end = cd_split_bound(total_chunks, g.splits, sp + 1);

    auto make_rsrc = [](const void* p, unsigned bytes) -> dg_v4i {
        const unsigned long long a = (unsigned long long)p;
        dg_v4i r;
        r[0] = (int)(unsigned)a; r[1] = (int)((unsigned)(a >> 32) & 0xffffu); r[2] = (int)bytes; r[3] = 0x00020000;
        return r;
    };
    const unsigned cbytes = (unsigned)g.Ci * ES;
    const dg_v4i ra = make_rsrc(g.x, (unsigned)g.B * (unsigned)g.H * (unsigned)g.W * cbytes);
    const dg_v4i rb = make_rsrc(g.w, (unsigned)N * (unsigned)K * ES);
    // this lane's PA rows of the A tile: output pixel -> top-left input coordinate and sample base
    int iy0[PA], ix0[PA], pb[PA];
    unsigned qa[PA];
#pragma unroll
    for (int i = 0; i < PA; ++i) {
        const int row = 8 * (wave * PA + i) + (lane >> 3);
        qa[i] = 16u * (unsigned)((lane & 7) ^ ((row >> 1) & 7));
        const int m = min(m0 + row, M - 1);
        const int b = m / (g.OH * g.OW), r2 = m - b * (g.OH * g.OW);
        const int oy = r2 / g.OW, ox = r2 - oy * g.OW;
        iy0[i] = oy * g.s - g.p;
        ix0[i] = ox * g.s - g.p;
        pb[i] = b * g.H * g.W;
    }
    unsigned va[PA], vb[PB];
#pragma unroll
    for (int i = 0; i < PB; ++i) {
        const int row = 8 * (wave * PB + i) + (lane >> 3), q = (lane & 7) ^ ((row >> 1) & 7);
        vb[i] = (unsigned)min(n0 + row, N - 1) * (unsigned)K * ES + 16u * q;
    }
    int tap = c_begin / cpt, cc = c_begin - tap * cpt;       // running position of the NEXT chunk to issue
    auto set_tap = [&]() {
        const int ky = tap / g.KW, kx = tap - ky * g.KW;
#pragma unroll
        for (int i = 0; i < PA; ++i) {
            const unsigned o = cd_pixel_off(g, pb[i], iy0[i] + ky, ix0[i] + kx, cbytes);
            va[i] = o == CD_OOB ? CD_OOB : o + qa[i];
        }
    };
    set_tap();
    const unsigned lds_a0 = (unsigned)(size_t)(dg_lds_ptr)As0 + (unsigned)(wave * PA) * 1024u;
    const unsigned lds_b0 = (unsigned)(size_t)(dg_lds_ptr)Bs0 + (unsigned)(wave * PB) * 1024u;
    auto issue = [&](int c, int buf) {        // chunks are issued in increasing c: (tap, cc) is a running state
        const unsigned la = lds_a0 + (unsigned)buf * (unsigned)(Cfg::ASZ * 4), lb = lds_b0 + (unsigned)buf * (unsigned)(Cfg::BSZ * 4);
        const unsigned sa_off = (unsigned)cc * 128u, sb_off = (unsigned)c * 128u;
#pragma unroll
        for (int i = 0; i < PA; ++i) dg_dma16(va[i], ra, la + 1024u * i, sa_off);
#pragma unroll
        for (int i = 0; i < PB; ++i) dg_dma16(vb[i], rb, lb + 1024u * i, sb_off);
        if (++cc == cpt) { cc = 0; ++tap; if (c + 1 < c_end) set_tap(); }
    };

    f32x16 acc[MB][NB];
#pragma unroll
    for (int mi = 0; mi < MB; ++mi)
#pragma unroll
        for (int ni = 0; ni < NB; ++ni) acc[mi][ni] = f32x16{0};
    const int wm0 = (wave >> 1) * (BM / 2), wn0 = (wave & 1) * (BN / 2);
    // NBUF LDS buffers keep NBUF - 1 chunks in flight (counted vmcnt); see cd_half_nbuf() for why everything runs two
    constexpr int AHEAD = NBUF - 1, PER = PA + PB;
#pragma unroll
    for (int i = 0; i < AHEAD; ++i)
        if (c_begin + i < c_end) issue(c_begin + i, i);
    int cur = 0, nxt = AHEAD % NBUF;
    for (int c = c_begin; c < c_end; ++c) {
        if (NBUF == 2 || c + AHEAD > c_end) dg_wait_vmcnt<0>();
        else dg_wait_vmcnt<(AHEAD - 1) * PER>();
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if (c + AHEAD < c_end) issue(c + AHEAD, nxt);
        cd_chunk<HALF, MB, NB, DG_KC, DG_KC, BM, BN>(As0 + cur * Cfg::ASZ, Bs0 + cur * Cfg::BSZ, acc, wm0, wn0, lane);
        cur = cur + 1 == NBUF ? 0 : cur + 1;
        nxt = nxt + 1 == NBUF ? 0 : nxt + 1;
    }

    float* o = g.part ? g.part + (size_t)sp * ((size_t)M * N) : g.y;
    const bool direct = g.part == nullptr;
#pragma unroll
    for (int mi = 0; mi < MB; ++mi)
#pragma unroll
        for (int ni = 0; ni < NB; ++ni) {
            const int col = n0 + wn0 + 32 * ni + (lane & 31);
            const float bv = (direct && g.bias && col < N) ? g.bias[col] : 0.0f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm0 + 32 * mi + mfma32_row(r, lane);
                if (row < M && col < N) {
                    const float v = acc[mi][ni][r];
                    const float va2 = apply_act(v + bv, g.act);
                    o[(size_t)row * N + col] = direct ? ((HALF && g.round_f16) ? round_h(va2) : va2) : v;
                }
            }
        }
#endif
}

// ------------------------------------------------------------------------------------------------------------------
// weight gradient: rows co, columns (tap, ci), reduction over output pixels (split over workgroups)
// ------------------------------------------------------------------------------------------------------------------
// RR ("row-regular" gather, zero padding): when a chunk of 32 pixels (float16 instances: 64) is a whole number of output rows or an aligned piece of one
// (OW % CK == 0 or CK % OW == 0 -- every ladder rung), the source pixel of chunk row k is  S(chunk) + const(k):  S is wave-uniform
// and advanced on the scalar unit, the per-lane part never changes, and the padding turns into three precomputed lane offsets
// (first / interior / last chunk of a row or image; out-of-range lanes hold CD_OOB).  The general path's per-lane coordinate
// walk is ~50 VALU instructions per chunk, and a VALU instruction issued while the other resident waves keep the matrix pipe
// full waits for a gap between their MFMAs: s_memtime stamps put the gather at 2300-2600 of a chunk's ~4800 cycles per wave.
template <int BM, int BN, bool HALF = false, int NBUF = 2, bool RR = false>
__global__ __launch_bounds__(256) void conv_wgrad_dma_kernel(CdArgs g) {
#if defined(__HIP_DEVICE_COMPILE__)
    using Cfg = DgCfgG<BM, BN, 2, 2, DG_RC, DG_RC, NBUF>;
    constexpr unsigned ES = CdElem<HALF>::ES;
    constexpr int CK = CdElem<HALF>::CK;
    constexpr int MB = Cfg::MB, NB = Cfg::NB, PA = Cfg::PA, PB = Cfg::PB;
    static_assert(BN == 64 || BN == 128, "a column tile must lie inside one tap (Ci % BN == 0; the planner checks it)");
    extern __shared__ __attribute__((aligned(1024))) float cd_smem[];
    float* As0 = cd_smem;
    float* Bs0 = cd_smem + NBUF * Cfg::ASZ;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int Mpx = g.B * g.OH * g.OW, R = g.Co, N = g.KH * g.KW * g.Ci;
    const int tiles = g.tiles_m * g.tiles_n;
    const int L = xcd_remap(blockIdx.x, tiles * g.splits);
    const int sp = L / tiles, rem = L - sp * tiles;
    const int tn = rem / g.tiles_m, tm = rem - tn * g.tiles_m;
    const int r_0 = tm * BM, n0 = tn * BN;
    const int tap = n0 / g.Ci, ci0 = n0 - tap * g.Ci;
    const int ky = tap / g.KW, kx = tap - ky * g.KW;
    const int total_chunks = (Mpx + CK - 1) / CK;
    const int c_begin = cd_split_bound(total_chunks, g.splits, sp), c_end = cd_split_bound(total_chunks, g.splits, sp + 1);

    auto make_rsrc = [](const void* p, unsigned bytes) -> dg_v4i {
        const unsigned long long a = (unsigned long long)p;
        dg_v4i r;
        r[0] = (int)(unsigned)a; r[1] = (int)((unsigned)(a >> 32) & 0xffffu); r[2] = (int)bytes; r[3] = 0x00020000;
        return r;
    };
    const unsigned cbytes = (unsigned)g.Ci * ES;
    // A = dy [Mpx][Co]: rows behind Mpx are behind num_records (zeros) -- the K tail is free
    const dg_v4i ra = make_rsrc(g.w, (unsigned)Mpx * (unsigned)g.Co * ES);
    const dg_v4i rb = make_rsrc(g.x, (unsigned)g.B * (unsigned)g.H * (unsigned)g.W * cbytes);
    constexpr int EPL = HALF ? 8 : 4;                    // elements per lane (16 bytes)
    unsigned va[PA];
#pragma unroll
    for (int i = 0; i < PA; ++i) {
        int k, col;
        cd_rc_lane<HALF, BM>(wave * PA + i, lane, k, col);
        va[i] = ((unsigned)k * (unsigned)g.Co + (unsigned)min(r_0 + col, R - EPL)) * ES;
    }
    // B rows: pixel m = 32 c + krow of this lane, as running (b, oy, ox)
    // Zero padding: the source coordinates (iy, ix) and the linear source pixel are carried along with (oy, ox), so a
    // chunk's gather address is additions and two range checks, not the multiplications of cd_pixel_off (reflection keeps
    // those: the reflected index is not linear in the step).
    int pm[PB], pbb[PB], poy[PB], pox[PB], siy[PB], six[PB], src[PB];
    unsigned qb[PB];
#pragma unroll
    for (int i = 0; i < PB; ++i) {
        int k, col;
        cd_rc_lane<HALF, BN>(wave * PB + i, lane, k, col);
        qb[i] = (unsigned)(ci0 + col) * ES;
        pm[i] = c_begin * CK + k;
        pbb[i] = pm[i] / (g.OH * g.OW);
        const int r2 = pm[i] - pbb[i] * (g.OH * g.OW);
        poy[i] = r2 / g.OW;
        pox[i] = r2 - poy[i] * g.OW;
        siy[i] = poy[i] * g.s - g.p + ky;
        six[i] = pox[i] * g.s - g.p + kx;
        src[i] = (pbb[i] * g.H + siy[i]) * g.W + six[i];
    }
    const int adv_oy = CK / g.OW, adv_ox = CK - adv_oy * g.OW;
    const int d_x = adv_ox * g.s, d_y = adv_oy * g.s, d_src = d_y * g.W + d_x;              // per chunk
    const int w_x = g.OW * g.s, w_src = g.s * g.W - w_x;                                  // ox wrapped: next output row
    const int w_y = g.OH * g.s, wb_src = (g.H - w_y) * g.W;                               // oy wrapped: next sample
    const unsigned lds_a0 = (unsigned)(size_t)(dg_lds_ptr)As0 + (unsigned)(wave * PA) * 1024u;
    const unsigned lds_b0 = (unsigned)(size_t)(dg_lds_ptr)Bs0 + (unsigned)(wave * PB) * 1024u;
    // row-regular gather state (RR): lane constants and the scalar walk
    unsigned vfirst[PB], vmid[PB], vlast[PB];
    const bool rr_a = g.OW >= CK;                                     // a chunk is a piece of one output row / a chunk is CK / OW whole rows
    const int rr_rows = rr_a ? 1 : CK / g.OW;
    const int rr_cpp = rr_a ? g.OW / CK : (g.OH * g.OW) / CK;         // chunks per period (row / image)
    const int rr_n2 = rr_a ? g.OH : 1;                                // periods per second-level wrap
    const int rr_step = (rr_a ? CK * g.s : rr_rows * g.s * g.W) * (int)cbytes;
    const int rr_w1 = (rr_a ? (g.s * g.W - g.OW * g.s) : (g.H - g.OH * g.s) * g.W) * (int)cbytes;
    const int rr_w2 = rr_a ? (g.H - g.OH * g.s) * g.W * (int)cbytes : 0;
    int rr_pos = 0, rr_r2 = 0, rr_S = 0;
    dg_v4i rbr = rb;
    if constexpr (RR) {
        const long long shift = ((long long)(ky - g.p) * g.W + (kx - g.p)) * (long long)cbytes;
        rbr = make_rsrc((const char*)g.x + shift, 0x80000000u);
#pragma unroll
        for (int i = 0; i < PB; ++i) {
            int k, col;
            cd_rc_lane<HALF, BN>(wave * PB + i, lane, k, col);
            const int j = rr_a ? 0 : k / g.OW, xk = rr_a ? k : k - (k / g.OW) * g.OW;
            const unsigned lc = (unsigned)((j * g.s * g.W + xk * g.s) * (int)cbytes) + (unsigned)(ci0 + col) * ES;
            // first / last chunk of the period: the lanes whose pixel falls into the padding
            const int ox_f = xk, ox_l = rr_a ? g.OW - CK + xk : xk;
            const int oy_f = j, oy_l = rr_a ? 0 : g.OH - rr_rows + j;             // (rr_a: the row check is scalar)
            const bool okf = (unsigned)(ox_f * g.s + kx - g.p) < (unsigned)g.W && (rr_a || (unsigned)(oy_f * g.s + ky - g.p) < (unsigned)g.H);
            const bool okl = (unsigned)(ox_l * g.s + kx - g.p) < (unsigned)g.W && (rr_a || (unsigned)(oy_l * g.s + ky - g.p) < (unsigned)g.H);
            vfirst[i] = (okf && (rr_cpp > 1 || okl)) ? lc : CD_OOB;
            vlast[i] = okl ? lc : CD_OOB;
            vmid[i] = (rr_a || (unsigned)(xk * g.s + kx - g.p) < (unsigned)g.W) ? lc : CD_OOB;      // (rr_a: interior chunks never touch the edge, checked on the host)
        }
        const int per = c_begin / rr_cpp;                               // c = (per, pos);  per = (b, r2)
        rr_pos = c_begin - per * rr_cpp;
        const int bb = per / rr_n2;
        rr_r2 = per - bb * rr_n2;
        rr_S = rr_a ? ((bb * g.H + rr_r2 * g.s) * g.W + rr_pos * CK * g.s) * (int)cbytes
                    : ((bb * g.H + rr_pos * rr_rows * g.s) * g.W) * (int)cbytes;
    }
    auto issue_rr = [&](int c, int buf) {
        const unsigned la = (unsigned)__builtin_amdgcn_readfirstlane((int)(lds_a0 + (unsigned)buf * (unsigned)(Cfg::ASZ * 4)));
        const unsigned lb = (unsigned)__builtin_amdgcn_readfirstlane((int)(lds_b0 + (unsigned)buf * (unsigned)(Cfg::BSZ * 4)));
        const unsigned sa_off = (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)c * (unsigned)CK * (unsigned)g.Co * ES));
        const unsigned sb_off = (unsigned)__builtin_amdgcn_readfirstlane(rr_S);
        const int pos = __builtin_amdgcn_readfirstlane(rr_pos);
        const bool row_ok = !rr_a || (unsigned)(__builtin_amdgcn_readfirstlane(rr_r2) * g.s + ky - g.p) < (unsigned)g.H;
#pragma unroll
        for (int i = 0; i < PA; ++i) dg_dma16(va[i], ra, la + 1024u * i, sa_off);
        if (!row_ok) {
            const unsigned voob = CD_OOB;
#pragma unroll
            for (int i = 0; i < PB; ++i) dg_dma16(voob, rbr, lb + 1024u * i, 0u);
        } else if (pos == 0) {
#pragma unroll
            for (int i = 0; i < PB; ++i) dg_dma16(vfirst[i], rbr, lb + 1024u * i, sb_off);
        } else if (pos == rr_cpp - 1) {
#pragma unroll
            for (int i = 0; i < PB; ++i) dg_dma16(vlast[i], rbr, lb + 1024u * i, sb_off);
        } else {
#pragma unroll
            for (int i = 0; i < PB; ++i) dg_dma16(vmid[i], rbr, lb + 1024u * i, sb_off);
        }
        rr_S += rr_step;
        if (++rr_pos == rr_cpp) {
            rr_pos = 0; rr_S += rr_w1;
            if (++rr_r2 == rr_n2) { rr_r2 = 0; rr_S += rr_w2; }
        }
    };
    auto issue = [&](int c, int buf) {
        if constexpr (RR) { issue_rr(c, buf); return; }
        // wave-uniform values; readfirstlane keeps them in SGPRs under the register pressure of the 128-column instances
        const unsigned la = (unsigned)__builtin_amdgcn_readfirstlane((int)(lds_a0 + (unsigned)buf * (unsigned)(Cfg::ASZ * 4)));
        const unsigned lb = (unsigned)__builtin_amdgcn_readfirstlane((int)(lds_b0 + (unsigned)buf * (unsigned)(Cfg::BSZ * 4)));
        const unsigned sa_off = (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)c * (unsigned)CK * (unsigned)g.Co * ES));
#pragma unroll
        for (int i = 0; i < PA; ++i) dg_dma16(va[i], ra, la + 1024u * i, sa_off);
#pragma unroll
        for (int i = 0; i < PB; ++i) {
            unsigned o = CD_OOB;
            if (g.reflect) {
                if (pm[i] < Mpx) o = cd_pixel_off(g, pbb[i] * g.H * g.W, siy[i], six[i], cbytes) + qb[i];
            } else if (pm[i] < Mpx && (unsigned)siy[i] < (unsigned)g.H && (unsigned)six[i] < (unsigned)g.W) {
                o = (unsigned)src[i] * cbytes + qb[i];
            }
            dg_dma16(o, rb, lb + 1024u * i, 0u);
            pm[i] += CK;
            pox[i] += adv_ox; six[i] += d_x;
            poy[i] += adv_oy; siy[i] += d_y;
            src[i] += d_src;
            if (pox[i] >= g.OW) { pox[i] -= g.OW; six[i] -= w_x; ++poy[i]; siy[i] += g.s; src[i] += w_src; }
            while (poy[i] >= g.OH) { poy[i] -= g.OH; siy[i] -= w_y; ++pbb[i]; src[i] += wb_src; }
        }
    };

    f32x16 acc[MB][NB];
#pragma unroll
    for (int mi = 0; mi < MB; ++mi)
#pragma unroll
        for (int ni = 0; ni < NB; ++ni) acc[mi][ni] = f32x16{0};
    const int wm0 = (wave >> 1) * (BM / 2), wn0 = (wave & 1) * (BN / 2);
    // NBUF LDS buffers keep NBUF - 1 chunks in flight (counted vmcnt); see cd_half_nbuf() for why everything runs two
    constexpr int AHEAD = NBUF - 1, PER = PA + PB;
#pragma unroll
    for (int i = 0; i < AHEAD; ++i)
        if (c_begin + i < c_end) issue(c_begin + i, i);
    int cur = 0, nxt = AHEAD % NBUF;
    for (int c = c_begin; c < c_end; ++c) {
        if (NBUF == 2 || c + AHEAD > c_end) dg_wait_vmcnt<0>();
        else dg_wait_vmcnt<(AHEAD - 1) * PER>();
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if (c + AHEAD < c_end) issue(c + AHEAD, nxt);
        cd_chunk<HALF, MB, NB, DG_RC, DG_RC, BM, BN>(As0 + cur * Cfg::ASZ, Bs0 + cur * Cfg::BSZ, acc, wm0, wn0, lane);
        cur = cur + 1 == NBUF ? 0 : cur + 1;
        nxt = nxt + 1 == NBUF ? 0 : nxt + 1;
    }

    float* o = g.part ? g.part + (size_t)sp * ((size_t)R * N) : g.y;
    const bool direct = g.part == nullptr;
#pragma unroll
    for (int mi = 0; mi < MB; ++mi)
#pragma unroll
        for (int ni = 0; ni < NB; ++ni) {
            const int col = n0 + wn0 + 32 * ni + (lane & 31);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = r_0 + wm0 + 32 * mi + mfma32_row(r, lane);
                if (row < R && col < N) {
                    const size_t idx = (size_t)row * N + col;
                    o[idx] = (direct && g.accumulate) ? o[idx] + acc[mi][ni][r] : acc[mi][ni][r];
                }
            }
        }
#endif
}

// ------------------------------------------------------------------------------------------------------------------
// data gradient == transposed convolution forward (zero padding; stride 1 or 2).  One grid slice per output-parity class
// (stride^2 of them) so no MFMA work is spent on structural zeros:
//   dX[m=(b,iy,ix)][ci] = sum_{taps of the class, co} dY[b, (iy + p - ky)/s, (ix + p - kx)/s][co] * W[co][ky][kx][ci]
// A: gathered rows of dY (k-contiguous, range-checked like the forward's), B: the weights viewed [(tap, co)][ci]
// (row-contiguous: k-major LDS image, ds_read_b32 fragments).  grid = (tiles, classes, splits).
// ------------------------------------------------------------------------------------------------------------------
template <int BM, int BN, bool HALF = false, int NBUF = 2>
__global__ __launch_bounds__(256) void conv_dgrad_dma_kernel(CdArgs g) {
#if defined(__HIP_DEVICE_COMPILE__)
    using Cfg = DgCfgG<BM, BN, 2, 2, DG_KC, DG_RC, NBUF>;
    constexpr unsigned ES = CdElem<HALF>::ES;
    constexpr int CK = CdElem<HALF>::CK;
    constexpr int MB = Cfg::MB, NB = Cfg::NB, PA = Cfg::PA, PB = Cfg::PB;
    extern __shared__ __attribute__((aligned(1024))) float cd_smem[];
    float* As0 = cd_smem;
    float* Bs0 = cd_smem + NBUF * Cfg::ASZ;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int s = g.s, KT = g.KH * g.KW;
    // heaviest class first (most taps): workgroups are dispatched in blockIdx order
    int py = 0, px = 0;
    if (s > 1) {
        // with fewer workgroups per class than CUs the dispatcher's round robin puts class y and class y + 2 on the same
        // CUs: (4 taps, 2, 1, 2) pairs the heaviest with the lightest, (4, 2, 2, 1) would pair 4 + 2 against 2 + 1
        // ... and a K split (blockIdx.z) of a small grid lands on the same CUs as split 0: rotate the classes by two per
        // split so that a CU's second workgroup is the light partner again, not another slice of the same class
        const int cy = (g.cls_order && s == 2) ? (int)((blockIdx.y + 2 * blockIdx.z) & 3) : (int)blockIdx.y;
        const int zi = cy / s;
        const int zj = (cy - zi * s) ^ ((g.cls_order && s == 2) ? zi : 0);
        const int t0y = (g.KH - (g.p % s) + s - 1) / s, t1y = (g.KH - ((1 + g.p) % s) + s - 1) / s;
        const int t0x = (g.KW - (g.p % s) + s - 1) / s, t1x = (g.KW - ((1 + g.p) % s) + s - 1) / s;
        const int hy = t1y > t0y ? 1 : 0, hx = t1x > t0x ? 1 : 0;
        py = zi == 0 ? hy : 1 - hy;
        px = zj == 0 ? hx : 1 - hx;
    }
    const int Hc = (g.H - py + s - 1) / s, Wc = (g.W - px + s - 1) / s;
    const int M = g.B * Hc * Wc, N = g.Ci;
    const int tiles_m = (M + BM - 1) / BM, tiles_n = N / BN;
    if ((int)blockIdx.x >= tiles_m * tiles_n) return;
    const int t = xcd_remap(blockIdx.x, tiles_m * tiles_n);
    int tn, tm;
    if (g.gm > 0) {                                                 // groups of gm row tiles x all column tiles (cd_pick_gm)
        const int per = g.gm * tiles_n, grp = t / per, r2 = t - grp * per;
        const int sz = min(g.gm, tiles_m - grp * g.gm);
        tn = r2 / sz;
        tm = grp * g.gm + (r2 - tn * sz);
    } else {
        tn = t / tiles_m;
        tm = t - tn * tiles_m;
    }
    const int m0 = tm * BM, n0 = tn * BN;
    const int ky0 = (py + g.p) % s, kx0 = (px + g.p) % s;
    const int nky = (g.KH - ky0 + s - 1) / s, nkx = (g.KW - kx0 + s - 1) / s;
    const int oyb = (py + g.p) / s, oxb = (px + g.p) / s;
    const int cpt = g.Co / CK, total_chunks = nky * nkx * cpt;
    // every class splits its own K range evenly
    const int c_begin = cd_split_bound(total_chunks, (int)gridDim.z, (int)blockIdx.z), c_end = cd_split_bound(total_chunks, (int)gridDim.z, (int)blockIdx.z + 1);

    auto make_rsrc = [](const void* p, unsigned bytes) -> dg_v4i {
        const unsigned long long a = (unsigned long long)p;
        dg_v4i r;
        r[0] = (int)(unsigned)a; r[1] = (int)((unsigned)(a >> 32) & 0xffffu); r[2] = (int)bytes; r[3] = 0x00020000;
        return r;
    };
    const unsigned cbytes = (unsigned)g.Co * ES;
    const dg_v4i ra = make_rsrc(g.x, (unsigned)g.B * (unsigned)g.OH * (unsigned)g.OW * cbytes);          // dy
    const dg_v4i rb = make_rsrc(g.w, (unsigned)g.Co * (unsigned)KT * (unsigned)g.Ci * ES);
    int yy[PA], xx[PA], pb[PA];
    unsigned qa[PA];
#pragma unroll
    for (int i = 0; i < PA; ++i) {
        const int row = 8 * (wave * PA + i) + (lane >> 3);
        qa[i] = 16u * (unsigned)((lane & 7) ^ ((row >> 1) & 7));
        const int m = min(m0 + row, M - 1);
        const int b = m / (Hc * Wc), r2 = m - b * (Hc * Wc);
        yy[i] = r2 / Wc + oyb;
        xx[i] = r2 - (r2 / Wc) * Wc + oxb;
        pb[i] = b * g.OH * g.OW;
    }
    unsigned va[PA], vb[PB];
#pragma unroll
    for (int i = 0; i < PB; ++i) {
        int k, col;
        cd_rc_lane<HALF, BN>(wave * PB + i, lane, k, col);
        vb[i] = ((unsigned)k * (unsigned)(KT * g.Ci) + (unsigned)(n0 + col)) * ES;
    }
    int tapi = c_begin / cpt, cc = c_begin - tapi * cpt;
    unsigned tap_off_b = 0;
    auto set_tap = [&]() {
        const int tyi = tapi / nkx, txi = tapi - tyi * nkx;
#pragma unroll
        for (int i = 0; i < PA; ++i) {
            const int oy = yy[i] - tyi, ox = xx[i] - txi;
            const bool ok = (unsigned)oy < (unsigned)g.OH && (unsigned)ox < (unsigned)g.OW;
            va[i] = ok ? (unsigned)(pb[i] + oy * g.OW + ox) * cbytes + qa[i] : CD_OOB;
        }
        tap_off_b = (unsigned)(((ky0 + s * tyi) * g.KW + kx0 + s * txi) * g.Ci) * ES;
    };
    set_tap();
    const unsigned lds_a0 = (unsigned)(size_t)(dg_lds_ptr)As0 + (unsigned)(wave * PA) * 1024u;
    const unsigned lds_b0 = (unsigned)(size_t)(dg_lds_ptr)Bs0 + (unsigned)(wave * PB) * 1024u;
    const unsigned co_stride = (unsigned)(KT * g.Ci) * ES * (unsigned)CK;      // one chunk of output channels further
    auto issue = [&](int c, int buf) {
        const unsigned la = lds_a0 + (unsigned)buf * (unsigned)(Cfg::ASZ * 4), lb = lds_b0 + (unsigned)buf * (unsigned)(Cfg::BSZ * 4);
        const unsigned sa_off = (unsigned)cc * 128u, sb_off = tap_off_b + (unsigned)cc * co_stride;
#pragma unroll
        for (int i = 0; i < PA; ++i) dg_dma16(va[i], ra, la + 1024u * i, sa_off);
#pragma unroll
        for (int i = 0; i < PB; ++i) dg_dma16(vb[i], rb, lb + 1024u * i, sb_off);
        if (++cc == cpt) { cc = 0; ++tapi; if (c + 1 < c_end) set_tap(); }
    };

    f32x16 acc[MB][NB];
#pragma unroll
    for (int mi = 0; mi < MB; ++mi)
#pragma unroll
        for (int ni = 0; ni < NB; ++ni) acc[mi][ni] = f32x16{0};
    const int wm0 = (wave >> 1) * (BM / 2), wn0 = (wave & 1) * (BN / 2);
    // NBUF LDS buffers keep NBUF - 1 chunks in flight (counted vmcnt); see cd_half_nbuf() for why everything runs two
    constexpr int AHEAD = NBUF - 1, PER = PA + PB;
#pragma unroll
    for (int i = 0; i < AHEAD; ++i)
        if (c_begin + i < c_end) issue(c_begin + i, i);
    int cur = 0, nxt = AHEAD % NBUF;
    for (int c = c_begin; c < c_end; ++c) {
        if (NBUF == 2 || c + AHEAD > c_end) dg_wait_vmcnt<0>();
        else dg_wait_vmcnt<(AHEAD - 1) * PER>();
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if (c + AHEAD < c_end) issue(c + AHEAD, nxt);
        cd_chunk<HALF, MB, NB, DG_KC, DG_RC, BM, BN>(As0 + cur * Cfg::ASZ, Bs0 + cur * Cfg::BSZ, acc, wm0, wn0, lane);
        cur = cur + 1 == NBUF ? 0 : cur + 1;
        nxt = nxt + 1 == NBUF ? 0 : nxt + 1;
    }

    // split-K slabs are whole dx images (classes write disjoint pixels of the same slab)
    float* o = g.part ? g.part + (size_t)blockIdx.z * ((size_t)g.B * g.H * g.W * N) : g.y;
    const bool direct = g.part == nullptr;
    // a lane's 16 rows per block are m_base + {0..3, 8..11, 16..19, 24..27}: one division for the first, then carries
    // (three integer divisions per row cost more VALU work than the rows' MFMAs on the light parity classes)
#pragma unroll
    for (int mi = 0; mi < MB; ++mi) {
        const int m_base = m0 + wm0 + 32 * mi + 4 * (lane >> 5);
        int bq = m_base / (Hc * Wc);
        int y2 = (m_base - bq * (Hc * Wc)) / Wc, x2 = m_base - bq * (Hc * Wc) - y2 * Wc;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            if (r) {
                x2 += (r & 3) ? 1 : 5;
                while (x2 >= Wc) { x2 -= Wc; ++y2; }
                while (y2 >= Hc) { y2 -= Hc; ++bq; }
            }
            const int m = m_base + (r & 3) + 8 * (r >> 2);
            if (m >= M) continue;
            const size_t off = ((size_t)(bq * g.H + y2 * s + py) * g.W + x2 * s + px) * N;
#pragma unroll
            for (int ni = 0; ni < NB; ++ni) {
                const int col = n0 + wn0 + 32 * ni + (lane & 31);
                const float v = acc[mi][ni][r];
                const float vb2 = apply_act(v + (g.bias ? g.bias[col] : 0.0f), g.act);
                o[off + col] = direct ? ((HALF && g.round_f16) ? round_h(vb2) : vb2) : v;
            }
        }
    }
#endif
}

inline bool conv_dma_enabled() { return true; }
inline bool conv_dma_half(const mg_conv_geom* g) { return g->precision == MG_PRECISION_F16; }
inline bool conv_dma_prec_ok(const mg_conv_geom* g) {
    constexpr bool off_h = false;
    if (!conv_dma_enabled()) return false;
    return g->precision == MG_PRECISION_F32 || (g->precision == MG_PRECISION_F16 && !off_h);
}
inline int conv_dma_ck(const mg_conv_geom* g) { return conv_dma_half(g) ? 64 : 32; }
inline double conv_dma_es(const mg_conv_geom* g) { return conv_dma_half(g) ? 2.0 : 4.0; }
inline bool conv_dma_fwd_ok(const mg_conv_geom* g) {
    return conv_dma_prec_ok(g) && g->Ci % conv_dma_ck(g) == 0 && g->Co % 64 == 0 &&
           (double)g->B * g->H * g->W * g->Ci * conv_dma_es(g) < 2e9 && (double)g->Co * g->KH * g->KW * g->Ci * conv_dma_es(g) < 2e9;
}
inline bool conv_dma_wgrad_ok(const mg_conv_geom* g) {
    return conv_dma_prec_ok(g) && g->Ci % 64 == 0 && g->Co % 64 == 0 &&
           (double)g->B * g->H * g->W * g->Ci * conv_dma_es(g) < 2e9 && (double)g->B * g->OH * g->OW * g->Co * conv_dma_es(g) < 2e9;
}
inline bool conv_dma_dgrad_ok(const mg_conv_geom* g) {
    return conv_dma_prec_ok(g) && !g->reflect && g->Co % conv_dma_ck(g) == 0 && g->Ci % 64 == 0 &&
           (double)g->B * g->OH * g->OW * g->Co * conv_dma_es(g) < 2e9 && (double)g->Co * g->KH * g->KW * g->Ci * conv_dma_es(g) < 2e9;
}
// float16 staging of the HALF instances (bytes, 256-aligned): activations cast by a pre-pass, weights unless a cached copy
// (mg_wino_tiles.u, mg_conv_wino_prepare) is handed in
inline size_t cd_al(size_t b) { return (b + 255) & ~(size_t)255; }
inline size_t conv_dma_h_x_bytes(const mg_conv_geom* g) { return cd_al((size_t)g->B * g->H * g->W * g->Ci * 2); }
inline size_t conv_dma_h_dy_bytes(const mg_conv_geom* g) { return cd_al((size_t)g->B * g->OH * g->OW * g->Co * 2); }
inline size_t conv_dma_h_w_bytes(const mg_conv_geom* g) { return cd_al((size_t)g->Co * g->KH * g->KW * g->Ci * 2); }
struct CdPlan { int bm, bn, splits, cps; };
// cost-model constants of the float16 data gradient (conv_dma_dgrad_plan), fitted in round 3 (scripts/sweep_half_tiles.sh)
inline double cd_half_rate() { return 4.0; }
inline double cd_half_fixed() { return 4.0; }
inline long long cd_half_fill() { return 512; }
// Plans of the forward pass and the weight gradient on the convolution's GEMM view.  ck = K depth of a chunk (32 float32 / 64 float16).
// Three models: rounds of resident workgroups for the float32 weight gradient and for both float16 passes (round 5), the dense
// plan's cost model (dense_plan above) for the float32 forward pass.
CdPlan conv_dma_plan(long long M, int N, int chunks, bool wgrad, int ck, int tap_cols = 0) {   // tap_cols: Ci of a weight gradient
    struct Cand { int bm, bn; double eff; };
    static const Cand cands[4] = {{64, 64, 0.83}, {64, 128, 0.885}, {128, 64, 0.855}, {128, 128, 0.91}};
    static const int split_opts[12] = {1, 2, 3, 4, 6, 8, 12, 16, 24, 32, 48, 64};
    const bool half = ck == 64;
    const double rate = 157.3e12 / 256.0, fixed = 1.2;
    const long long fill = 512;
    CdPlan p{64, 64, 1, 1 << 28};
    int f_bm = 0, f_bn = 0, f_sp = 0;
    if (const char* f = getenv("MG_FORCE_CONV_DMA")) {      // tuning harness: "bm,bn,splits"
        if (sscanf(f, "%d,%d,%d", &f_bm, &f_bn, &f_sp) != 3) f_bm = f_bn = f_sp = 0;
    }
    double best = 1e300;
    if (!half && wgrad) {
        // float32 weight gradients: rounds of resident workgroups (round 5; s_memtime stamps in the kernels -- scripts/ubench/
        // conv_dma_stamps.diff -- and scripts/tune_ladder.py sweeps at batch 2 / 8 / 64).  A CU holds `slots` workgroups of a tile
        // shape (measured: 4 of the 32 KiB ones, not the 5 the LDS size suggests); r of them sharing the matrix pipe take
        // max(r x MFMA cycles / eff, ~3000) shader cycles per 32-deep chunk each -- a lone 64 x 64 workgroup is bound by its own DMA
        // round trip, not by the pipe -- and the workgroups of a round start and finish together, so the rounds add up and a last
        // round of one workgroup per CU costs almost a full one.  Hence: any split count up to 64 (not a fixed ladder of them), and
        // the count that fills the slots exactly once wins (64 -> 128 rung: 18 tiles x 42 = 756 workgroups, 118 -> 104 us; 128 -> 256:
        // 72 x 14 = 1008, 102 -> 95).  Constants fitted to the sweeps (pick within 3 % of the measured best on the four rungs);
        // the forward pass keeps the model below -- re-planning it bought nothing and re-associates the forward sums.
        struct CandW { int bm, bn, slots; double pen; };
        static const CandW cw[4] = {{64, 64, 4, 1.06}, {64, 128, 3, 0.91}, {128, 64, 3, 0.93}, {128, 128, 2, 0.90}};
        for (const CandW& c : cw) {
            if (N % c.bn != 0 || tap_cols % c.bn != 0) continue;                 // a column tile lies inside one tap
            if (f_bm && (c.bm != f_bm || c.bn != f_bn)) continue;
            const long long w = ((M + c.bm - 1) / c.bm) * (long long)(N / c.bn);
            const double mfma = 1024.0 * (c.bm / 64) * (c.bn / 64), eff = 0.903 * c.pen;
            const int sp_max = f_sp ? f_sp : (chunks / 2 < 64 ? (chunks / 2 < 1 ? 1 : chunks / 2) : 64);
            for (int sp = f_sp ? f_sp : 1; sp <= sp_max && sp <= chunks; ++sp) {
                const long long wg = w * sp;
                if (sp > 1 && wg > (1 << 16)) break;      // (an unsplit tiling is always a candidate, however large the GEMM)
                const int cps = (chunks + sp - 1) / sp;
                long long left = (wg + 255) / 256;
                double cycles = 0.0;
                while (left > 0) {
                    const int r = left < c.slots ? (int)left : c.slots;
                    left -= r;
                    const double per_chunk = r * mfma / eff;
                    cycles += 1000.0 + cps * (per_chunk > 2971.0 ? per_chunk : 2971.0) + (double)r * c.bm * c.bn * 4.0 * 256.0 / 4000.0;
                }
                double t = 8.0 + cycles / 2200.0;
                if (sp > 1) t += (double)(sp + 1) * (double)M * N * 4.0 / 6e12 * 1e6 + 6.74;      // slabs + the reduce launch
                if (t < best) { best = t; p = {c.bm, c.bn, sp, cps}; }
            }
        }
        if (p.splits == 1) p.cps = 1 << 28;
        return p;
    }
    if (half) {
        // float16 instances, forward and weight gradient: the same rounds model with its own constants.  A chunk holds only 4 / 8 / 16
        // MFMAs per wave here, so a workgroup's chunk is bound by its DMA round trip (~1500 cycles) up to four resident ones and
        // the fixed costs of a 30-50 us launch weigh more.  Fitted to scripts/tune_ladder.py --f16 on the configs[2] rungs
        // (profiles/r05_ladder_tile_split_sweep_f16.log: the pick is within 3 % of the measured best on 12 of 12 layer passes where
        // the round-3 model was 8 % off on average, 30 % on the 512 -> 1024 forward); configs[2] --fp16: six interleaved pairs of
        // bench runs, 13.86 -> 13.67 ms in the mean, every pair <= 0.
        struct CandH { int bm, bn, slots; double pen_f, pen_w; };
        static const CandH ch[4] = {{64, 64, 4, 1.064, 1.3}, {64, 128, 3, 0.74, 0.966}, {128, 64, 3, 0.716, 1.3}, {128, 128, 2, 1.057, 1.3}};
        for (const CandH& c : ch) {
            if (N % c.bn != 0 || (wgrad && tap_cols % c.bn != 0)) continue;
            if (f_bm && (c.bm != f_bm || c.bn != f_bn)) continue;
            const long long w = ((M + c.bm - 1) / c.bm) * (long long)(N / c.bn);
            const double mfma = 495.0 * (c.bm / 64) * (c.bn / 64) / (wgrad ? c.pen_w : c.pen_f);
            const int sp_max = f_sp ? f_sp : (chunks / 2 < 32 ? (chunks / 2 < 1 ? 1 : chunks / 2) : 32);
            for (int sp = f_sp ? f_sp : 1; sp <= sp_max && sp <= chunks; ++sp) {
                const long long wg = w * sp;
                if (sp > 1 && wg > (1 << 16)) break;      // (an unsplit tiling is always a candidate, however large the GEMM)
                const int cps = (chunks + sp - 1) / sp;
                long long left = (wg + 255) / 256;
                double cycles = 0.0;
                while (left > 0) {
                    const int r = left < c.slots ? (int)left : c.slots;
                    left -= r;
                    const double per_chunk = r * mfma;
                    cycles += 500.0 + cps * (per_chunk > 1477.0 ? per_chunk : 1477.0) + (double)r * c.bm * c.bn * 4.0 * 256.0 / 5000.0;
                }
                double t = 12.0 + cycles / 2200.0;
                if (sp > 1) t += (double)(sp + 1) * (double)M * N * 4.0 / 8e12 * 1e6 + 2.13;
                if (t < best) { best = t; p = {c.bm, c.bn, sp, cps}; }
            }
        }
        if (p.splits == 1) p.cps = 1 << 28;
        return p;
    }
    for (int ci = 0; ci < 4; ++ci) {
        const Cand& c = cands[ci];
        if (N % c.bn != 0) continue;                              // (float32 forward pass)
        if (f_bm && (c.bm != f_bm || c.bn != f_bn)) continue;
        const long long w = ((M + c.bm - 1) / c.bm) * (long long)(N / c.bn);
        const double tile_us = 2.0 * c.bm * c.bn * ck / rate * 1e6 / c.eff;
        for (int sp : split_opts) {
            if (sp > 1 && chunks / sp < 8) break;
            if (f_sp && sp != f_sp) continue;
            const int cps = (chunks + sp - 1) / sp;
            const int spl = sp;                                  // (cd_split_bound: every split count up to `chunks` is exact)
            const long long wg = w * spl;
            double t = (double)((wg + 255) / 256) * tile_us * (cps + fixed);
            if (wg < fill) t /= 0.85;
            if (spl > 1) t += (double)(spl + 1) * (double)M * N * 4.0 / 4e12 * 1e6 + 3.0;
            if (t < best) { best = t; p = {c.bm, c.bn, spl, cps}; }
        }
    }
    if (p.splits == 1) p.cps = 1 << 28;
    return p;
}
inline CdPlan conv_dma_fwd_plan(const mg_conv_geom* g) {
    const int ck = conv_dma_ck(g);
    return conv_dma_plan((long long)g->B * g->OH * g->OW, g->Co, g->KH * g->KW * (g->Ci / ck), false, ck);
}
inline CdPlan conv_dma_wgrad_plan(const mg_conv_geom* g) {
    const long long Mpx = (long long)g->B * g->OH * g->OW;
    const int ck = conv_dma_ck(g);
    return conv_dma_plan(g->Co, g->KH * g->KW * g->Ci, (int)((Mpx + ck - 1) / ck), true, ck, g->Ci);
}
template <typename KernelT>
inline void cd_launch(KernelT kern, size_t lds, dim3 grid, const CdArgs& a, hipStream_t st) {
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    mg_launch(kern, grid, dim3(256), lds, st, a);
}
inline CdArgs cd_args(const mg_conv_geom* g) {
    CdArgs a{};
    a.B = g->B; a.H = g->H; a.W = g->W; a.Ci = g->Ci; a.OH = g->OH; a.OW = g->OW; a.Co = g->Co; a.KH = g->KH; a.KW = g->KW;
    a.s = g->stride; a.p = g->pad; a.reflect = g->reflect;
    a.round_f16 = conv_dma_half(g) ? 1 : 0;
    return a;
}
// x, w: float32 tensors, or (conv_dma_half(g)) their float16 copies
// LDS buffers of the float16 instances (MG_HALF_NBUF=2|3|4, default 2).  Deeper rings were measured and lose: the float16 MFMAs of
// a chunk are 8x shorter than the DMA round trip, but three chunks in flight cost a resident workgroup per CU (128 x 128: 38.7 ->
// 57 us on the 128-channel 64x128 layers; 64 x 64 tiles: 47 -> 48 us); the kernels' time is the same for every tile shape
// (scripts/sweep_half_tiles.sh), i.e. neither operand bytes nor the ring depth set it
inline int cd_half_nbuf() {
    static const int v = [] { const char* e = getenv("MG_HALF_NBUF"); const int n = e ? atoi(e) : 2; return n < 2 ? 2 : (n > 4 ? 4 : n); }();
    return v;
}
#define CD_TILE_DISPATCH(KERNEL, ALAY, BLAY, HALF_, NB_)                                                                                    \
    do {                                                                                                                                    \
        if (p.bm == 128 && p.bn == 128) cd_launch(KERNEL<128, 128, HALF_, NB_>, DgCfgG<128, 128, 2, 2, ALAY, BLAY, NB_>::LDS_BYTES, grid, a, st); \
        else if (p.bm == 64 && p.bn == 128) cd_launch(KERNEL<64, 128, HALF_, NB_>, DgCfgG<64, 128, 2, 2, ALAY, BLAY, NB_>::LDS_BYTES, grid, a, st); \
        else if (p.bm == 128 && p.bn == 64) cd_launch(KERNEL<128, 64, HALF_, NB_>, DgCfgG<128, 64, 2, 2, ALAY, BLAY, NB_>::LDS_BYTES, grid, a, st); \
        else cd_launch(KERNEL<64, 64, HALF_, NB_>, DgCfgG<64, 64, 2, 2, ALAY, BLAY, NB_>::LDS_BYTES, grid, a, st);                            \
    } while (0)
#define CD_DISPATCH(KERNEL, ALAY, BLAY, half)                                                  \
    do {                                                                                       \
        if (!(half)) CD_TILE_DISPATCH(KERNEL, ALAY, BLAY, false, 2);                           \
        else if (cd_half_nbuf() == 4) CD_TILE_DISPATCH(KERNEL, ALAY, BLAY, true, 4);           \
        else if (cd_half_nbuf() == 3) CD_TILE_DISPATCH(KERNEL, ALAY, BLAY, true, 3);           \
        else CD_TILE_DISPATCH(KERNEL, ALAY, BLAY, true, 2);                                    \
    } while (0)
void conv_dma_fwd_launch(const mg_conv_geom* g, const CdPlan& p, const void* x, const void* w, const float* bias, float* y,
                         int act, float* part, hipStream_t st) {
    CdArgs a = cd_args(g);
    a.x = x; a.w = w; a.bias = bias; a.y = y; a.part = p.splits > 1 ? part : nullptr; a.act = act;
    const long long M = (long long)g->B * g->OH * g->OW;
    a.tiles_m = (int)((M + p.bm - 1) / p.bm); a.tiles_n = g->Co / p.bn; a.splits = p.splits; a.cps = p.cps;
    {
        const double es = conv_dma_half(g) ? 2.0 : 4.0;
        // a row tile's share of the input (every input pixel belongs to ~one row tile; taps re-read it from L2), a column tile's weights
        const double a_bytes = (double)g->B * g->H * g->W * g->Ci * es / a.tiles_m / a.splits;
        const double b_bytes = (double)p.bn * g->KH * g->KW * g->Ci * es / a.splits;
        a.gm = cd_pick_gm(a.tiles_m, a.tiles_n, a.splits, a_bytes, b_bytes);
    }
    const dim3 grid((unsigned)((long long)a.tiles_m * a.tiles_n * a.splits));
    CD_DISPATCH(conv_fwd_dma_kernel, 0, 0, conv_dma_half(g));
}
// eligibility of the row-regular gather (conv_wgrad_dma_kernel<..., RR = true>): zero padding, whole 32- / 64-pixel chunks that
// are an aligned piece of one output row or a whole number of rows of one image, and only the first / last chunk of a row (image)
// may touch the padding -- the kernel masks those two with precomputed lane offsets and takes the interior ones unmasked
inline bool conv_dma_wgrad_rowreg(const mg_conv_geom* g) {
    if (g->reflect) return false;
    const int CK = conv_dma_ck(g);                       // 32 pixels (float32) / 64 (float16 instances)
    const long long Mpx = (long long)g->B * g->OH * g->OW;
    if (Mpx % CK != 0) return false;
    const int s = g->stride, p = g->pad;
    if (g->OW >= CK) {
        if (g->OW % CK != 0) return false;
        const int cpp = g->OW / CK;
        for (int pos = 1; pos + 1 < cpp; ++pos)
            for (int k = 0; k < CK; ++k) {
                const int lo = (pos * CK + k) * s - p, hi = lo + g->KW - 1;
                if (lo < 0 || hi >= g->W) return false;
            }
    } else {
        if (CK % g->OW != 0 || (g->OH * g->OW) % CK != 0) return false;
        const int rows = CK / g->OW, cpp = g->OH * g->OW / CK;
        for (int pos = 1; pos + 1 < cpp; ++pos)
            for (int j = 0; j < rows; ++j) {
                const int lo = (pos * rows + j) * s - p, hi = lo + g->KH - 1;
                if (lo < 0 || hi >= g->H) return false;
            }
    }
    return true;
}
void conv_dma_wgrad_launch(const mg_conv_geom* g, const CdPlan& p, const void* x, const void* dy, float* dw, int accumulate,
                           float* part, hipStream_t st) {
    CdArgs a = cd_args(g);
    a.x = x; a.w = dy; a.y = dw; a.part = p.splits > 1 ? part : nullptr; a.accumulate = accumulate;
    a.tiles_m = (g->Co + p.bm - 1) / p.bm; a.tiles_n = g->KH * g->KW * g->Ci / p.bn; a.splits = p.splits; a.cps = p.cps;
    const dim3 grid((unsigned)((long long)a.tiles_m * a.tiles_n * a.splits));
    if (conv_dma_wgrad_rowreg(g) && !getenv("MG_NO_WGRAD_RR") && (!conv_dma_half(g) || cd_half_nbuf() == 2)) {
        if (conv_dma_half(g)) {
            if (p.bm == 128 && p.bn == 128) cd_launch(conv_wgrad_dma_kernel<128, 128, true, 2, true>, DgCfgG<128, 128, 2, 2, 1, 1, 2>::LDS_BYTES, grid, a, st);
            else if (p.bm == 64 && p.bn == 128) cd_launch(conv_wgrad_dma_kernel<64, 128, true, 2, true>, DgCfgG<64, 128, 2, 2, 1, 1, 2>::LDS_BYTES, grid, a, st);
            else if (p.bm == 128 && p.bn == 64) cd_launch(conv_wgrad_dma_kernel<128, 64, true, 2, true>, DgCfgG<128, 64, 2, 2, 1, 1, 2>::LDS_BYTES, grid, a, st);
            else cd_launch(conv_wgrad_dma_kernel<64, 64, true, 2, true>, DgCfgG<64, 64, 2, 2, 1, 1, 2>::LDS_BYTES, grid, a, st);
            return;
        }
        if (p.bm == 128 && p.bn == 128) cd_launch(conv_wgrad_dma_kernel<128, 128, false, 2, true>, DgCfgG<128, 128, 2, 2, 1, 1, 2>::LDS_BYTES, grid, a, st);
        else if (p.bm == 64 && p.bn == 128) cd_launch(conv_wgrad_dma_kernel<64, 128, false, 2, true>, DgCfgG<64, 128, 2, 2, 1, 1, 2>::LDS_BYTES, grid, a, st);
        else if (p.bm == 128 && p.bn == 64) cd_launch(conv_wgrad_dma_kernel<128, 64, false, 2, true>, DgCfgG<128, 64, 2, 2, 1, 1, 2>::LDS_BYTES, grid, a, st);
        else cd_launch(conv_wgrad_dma_kernel<64, 64, false, 2, true>, DgCfgG<64, 64, 2, 2, 1, 1, 2>::LDS_BYTES, grid, a, st);
        return;
    }
    CD_DISPATCH(conv_wgrad_dma_kernel, 1, 1, conv_dma_half(g));
}

// data gradient: the plan is made for the heaviest parity class (M = pixels of one class, all taps / stride^2 of the chunks)
inline CdPlan conv_dma_dgrad_plan(const mg_conv_geom* g) {
    const int s = g->stride, ck = conv_dma_ck(g);
    const bool half = conv_dma_half(g);
    const double rate = (half ? cd_half_rate() : 1.0) * 157.3e12 / 256.0, fixed = half ? cd_half_fixed() : 1.2;
    const long long fill = half ? cd_half_fill() : 512;
    const long long Mc = (long long)g->B * ((g->H + s - 1) / s) * ((g->W + s - 1) / s);
    const int chunks = ((g->KH + s - 1) / s) * ((g->KW + s - 1) / s) * (g->Co / ck);
    // the classes run side by side: count their workgroups when judging how full the chip is
    CdPlan best{64, 64, 1, 1 << 28};
    double best_t = 1e300;
    struct Cand { int bm, bn; double eff; };
    static const Cand cands_f[4] = {{64, 64, 0.83}, {64, 128, 0.885}, {128, 64, 0.855}, {128, 128, 0.91}};
    // float16 instances: 64 x 128 is the fastest tile on 7 of the 10 configs[2] shapes that admit it (same sweep as the weight
    // gradient's: 512 -> 1024 channels 41.6 vs 46.9 us, 1024 -> 2048 50.9 vs 58.9) and within 2 us on the rest
    static const Cand cands_h[4] = {{64, 64, 0.83}, {64, 128, 1.0}, {128, 64, 0.855}, {128, 128, 0.91}};
    const Cand* cands = half ? cands_h : cands_f;
    static const int split_opts[8] = {1, 2, 3, 4, 6, 8, 12, 16};
    int f_bm = 0, f_bn = 0, f_sp = 0;
    if (const char* f = getenv("MG_FORCE_CONV_DMA")) {
        if (sscanf(f, "%d,%d,%d", &f_bm, &f_bn, &f_sp) != 3) f_bm = f_bn = f_sp = 0;
    }
    const int total_k = g->KH * g->KW * (g->Co / ck);          // chunks summed over the classes
    for (int ci = 0; ci < 4; ++ci) {
        const Cand& c = cands[ci];
        if (g->Ci % c.bn != 0) continue;
        if (f_bm && (c.bm != f_bm || c.bn != f_bn)) continue;
        const long long w = ((Mc + c.bm - 1) / c.bm) * (long long)(g->Ci / c.bn);
        const double tile_us = 2.0 * c.bm * c.bn * ck / rate * 1e6 / c.eff;
        for (int sp : split_opts) {
            if (sp > 1 && chunks / sp < 8) break;
            if (f_sp && sp != f_sp) continue;
            const long long wg = w * s * s * sp;
            // fewer workgroups per class than CUs: a CU holds workgroups of two classes only (y and y + 2 of the launch
            // order, see the kernel), the pair (all taps, one tap) sets the time: 2.5 taps against the average 2.25
            // one workgroup per CU or fewer: nothing to pair with, the all-taps class alone sets the time
            const double k_wg = wg <= 256 ? (double)chunks
                                          : (double)total_k / (s * s) * ((w * sp >= 256 || total_k == chunks * s * s) ? 1.0 : 1.12);
            double t = (double)((wg + 255) / 256) * tile_us * (k_wg / sp + fixed);
            if (wg < fill) t /= 0.85;
            if (sp > 1) t += (double)(sp + 1) * (double)g->B * g->H * g->W * g->Ci * 4.0 / 4e12 * 1e6 + 3.0;
            if (t < best_t) { best_t = t; best = {c.bm, c.bn, sp, 0}; }
        }
    }
    return best;
}
void conv_dma_dgrad_launch(const mg_conv_geom* g, const CdPlan& p, const void* dy, const void* w, const float* bias, float* dx,
                           int act, float* part, hipStream_t st, int round_f16) {
    CdArgs a = cd_args(g);
    a.round_f16 = round_f16;
    a.x = dy; a.w = w; a.bias = bias; a.y = dx; a.part = p.splits > 1 ? part : nullptr; a.act = act;
    const int s = g->stride;
    const long long Mc = (long long)g->B * ((g->H + s - 1) / s) * ((g->W + s - 1) / s);
    const dim3 grid((unsigned)(((Mc + p.bm - 1) / p.bm) * (g->Ci / p.bn)), (unsigned)(s * s), (unsigned)p.splits);
    // class order (stride 2, odd kernels: 4 / 2 / 2 / 1 taps).  All workgroups resident at once (LDS: 5 / 3 / 2 per CU for
    // 64x64 / 64x128, 128x64 / 128x128 tiles): the round robin pairs class y with y + 2 on a CU, so (4, 2, 1, 2) balances
    // (512 -> 1024 channels at 8x16: 107 against 119 us).  More rounds than that: heaviest first is the better list order
    // (128 -> 256 channels at 64x128: 87 against 100 us).  MG_DGRAD_CLASS_ORDER=0|1 forces one.
    const long long slots = 256LL * (p.bm == 64 && p.bn == 64 ? 5 : (p.bm == 128 && p.bn == 128 ? 2 : 3));
    a.cls_order = (long long)grid.x * grid.y * grid.z <= slots ? 1 : 0;
    {
        // per parity class (grid.y) and K split (grid.z) the tiles of grid.x are handed to the XCDs in runs: same window model as
        // the forward pass, with dy as the row operand (a class reads its taps' share of dy) and the weights as the column operand
        const int tiles_m = (int)((Mc + p.bm - 1) / p.bm), tiles_n = g->Ci / p.bn;
        const double es = conv_dma_half(g) ? 2.0 : 4.0;
        const double a_bytes = (double)g->B * g->OH * g->OW * g->Co * es / tiles_m / p.splits;
        const double b_bytes = (double)p.bn * g->KH * g->KW * g->Co * es / (s * s) / p.splits;
        a.gm = cd_pick_gm(tiles_m, tiles_n, 1, a_bytes, b_bytes);
    }
    CD_DISPATCH(conv_dgrad_dma_kernel, 0, 1, conv_dma_half(g));
}
