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

struct CdArgs {
    const float* x;          // FWD: input [B,H,W,Ci];  WGRAD: input x
    const float* w;          // FWD: weights [Co][KH*KW*Ci];  WGRAD: dy [B,OH,OW,Co]
    const float* bias;
    float* y;                // FWD: output [M][Co];  WGRAD: dW [Co][KH*KW*Ci]
    float* part;             // split-K slabs
    int B, H, W, Ci, OH, OW, Co, KH, KW, s, p, reflect;
    int act, tiles_m, tiles_n, splits, cps, accumulate;
};

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
template <int BM, int BN>
__global__ __launch_bounds__(256) void conv_fwd_dma_kernel(CdArgs g) {
#if defined(__HIP_DEVICE_COMPILE__)
    using Cfg = DgCfgG<BM, BN, 2, 2, DG_KC, DG_KC, 2>;
    constexpr int MB = Cfg::MB, NB = Cfg::NB, PA = Cfg::PA, PB = Cfg::PB;
    extern __shared__ __attribute__((aligned(1024))) float cd_smem[];
    float* As0 = cd_smem;
    float* Bs0 = cd_smem + 2 * Cfg::ASZ;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int M = g.B * g.OH * g.OW, N = g.Co, KT = g.KH * g.KW, K = KT * g.Ci;
    const int tiles = g.tiles_m * g.tiles_n;
    const int L = xcd_remap(blockIdx.x, tiles * g.splits);
    const int sp = L / tiles, rem = L - sp * tiles;
    const int tn = rem / g.tiles_m, tm = rem - tn * g.tiles_m;      // consecutive tiles share the weight panel
    const int m0 = tm * BM, n0 = tn * BN;
    const int cpt = g.Ci / DG_BK, total_chunks = KT * cpt;
    const int c_begin = sp * g.cps, c_end = min(total_chunks, c_begin + g.cps);

    auto make_rsrc = [](const float* p, unsigned bytes) -> dg_v4i {
        const unsigned long long a = (unsigned long long)p;
        dg_v4i r;
        r[0] = (int)(unsigned)a; r[1] = (int)((unsigned)(a >> 32) & 0xffffu); r[2] = (int)bytes; r[3] = 0x00020000;
        return r;
    };
    const unsigned cbytes = (unsigned)g.Ci * 4u;
    const dg_v4i ra = make_rsrc(g.x, (unsigned)g.B * (unsigned)g.H * (unsigned)g.W * cbytes);
    const dg_v4i rb = make_rsrc(g.w, (unsigned)N * (unsigned)K * 4u);
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
        vb[i] = ((unsigned)min(n0 + row, N - 1) * (unsigned)K + 4u * q) * 4u;
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
        const unsigned sa_off = (unsigned)cc * (DG_BK * 4u), sb_off = (unsigned)c * (DG_BK * 4u);
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
    if (c_begin < c_end) issue(c_begin, 0);
    for (int c = c_begin; c < c_end; ++c) {
        const int cur = (c - c_begin) & 1;
        dg_wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if (c + 1 < c_end) issue(c + 1, cur ^ 1);
        dg_chunk_g<MB, NB, DG_KC, DG_KC, BM, BN>(As0 + cur * Cfg::ASZ, Bs0 + cur * Cfg::BSZ, acc, wm0, wn0, lane);
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
                    o[(size_t)row * N + col] = direct ? apply_act(v + bv, g.act) : v;
                }
            }
        }
#endif
}

// ------------------------------------------------------------------------------------------------------------------
// weight gradient: rows co, columns (tap, ci), reduction over output pixels (split over workgroups)
// ------------------------------------------------------------------------------------------------------------------
template <int BM, int BN>
__global__ __launch_bounds__(256) void conv_wgrad_dma_kernel(CdArgs g) {
#if defined(__HIP_DEVICE_COMPILE__)
    using Cfg = DgCfgG<BM, BN, 2, 2, DG_RC, DG_RC, 2>;
    constexpr int MB = Cfg::MB, NB = Cfg::NB, PA = Cfg::PA, PB = Cfg::PB;
    static_assert(BN == 64, "a column tile must lie inside one tap (Ci % 64 == 0)");
    extern __shared__ __attribute__((aligned(1024))) float cd_smem[];
    float* As0 = cd_smem;
    float* Bs0 = cd_smem + 2 * Cfg::ASZ;
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
    const int total_chunks = (Mpx + DG_BK - 1) / DG_BK;
    const int c_begin = sp * g.cps, c_end = min(total_chunks, c_begin + g.cps);

    auto make_rsrc = [](const float* p, unsigned bytes) -> dg_v4i {
        const unsigned long long a = (unsigned long long)p;
        dg_v4i r;
        r[0] = (int)(unsigned)a; r[1] = (int)((unsigned)(a >> 32) & 0xffffu); r[2] = (int)bytes; r[3] = 0x00020000;
        return r;
    };
    const unsigned cbytes = (unsigned)g.Ci * 4u;
    // A = dy [Mpx][Co]: rows behind Mpx are behind num_records (zeros) -- the K tail is free
    const dg_v4i ra = make_rsrc(g.w, (unsigned)Mpx * (unsigned)g.Co * 4u);
    const dg_v4i rb = make_rsrc(g.x, (unsigned)g.B * (unsigned)g.H * (unsigned)g.W * cbytes);
    constexpr int LPRA = BM / 4, LPRB = BN / 4;          // lanes per k row
    unsigned va[PA];
#pragma unroll
    for (int i = 0; i < PA; ++i) {
        const int k = (wave * PA + i) * (64 / LPRA) + lane / LPRA, c4 = lane % LPRA;
        va[i] = ((unsigned)k * (unsigned)g.Co + (unsigned)min(r_0 + 4 * c4, R - 4)) * 4u;
    }
    // B rows: pixel m = 32 c + krow of this lane, as running (b, oy, ox)
    int pm[PB], pbb[PB], poy[PB], pox[PB];
    const unsigned qb = (unsigned)(ci0 + 4 * (lane % LPRB)) * 4u;
#pragma unroll
    for (int i = 0; i < PB; ++i) {
        pm[i] = c_begin * DG_BK + (wave * PB + i) * (64 / LPRB) + lane / LPRB;
        pbb[i] = pm[i] / (g.OH * g.OW);
        const int r2 = pm[i] - pbb[i] * (g.OH * g.OW);
        poy[i] = r2 / g.OW;
        pox[i] = r2 - poy[i] * g.OW;
    }
    const int adv_oy = DG_BK / g.OW, adv_ox = DG_BK - adv_oy * g.OW;
    const unsigned lds_a0 = (unsigned)(size_t)(dg_lds_ptr)As0 + (unsigned)(wave * PA) * 1024u;
    const unsigned lds_b0 = (unsigned)(size_t)(dg_lds_ptr)Bs0 + (unsigned)(wave * PB) * 1024u;
    auto issue = [&](int c, int buf) {
        const unsigned la = lds_a0 + (unsigned)buf * (unsigned)(Cfg::ASZ * 4), lb = lds_b0 + (unsigned)buf * (unsigned)(Cfg::BSZ * 4);
        const unsigned sa_off = (unsigned)c * (unsigned)DG_BK * (unsigned)g.Co * 4u;
#pragma unroll
        for (int i = 0; i < PA; ++i) dg_dma16(va[i], ra, la + 1024u * i, sa_off);
#pragma unroll
        for (int i = 0; i < PB; ++i) {
            unsigned o = CD_OOB;
            if (pm[i] < Mpx) {
                o = cd_pixel_off(g, pbb[i] * g.H * g.W, poy[i] * g.s - g.p + ky, pox[i] * g.s - g.p + kx, cbytes);
                if (o != CD_OOB) o += qb;
            }
            dg_dma16(o, rb, lb + 1024u * i, 0u);
            pm[i] += DG_BK;
            pox[i] += adv_ox;
            poy[i] += adv_oy;
            if (pox[i] >= g.OW) { pox[i] -= g.OW; ++poy[i]; }
            while (poy[i] >= g.OH) { poy[i] -= g.OH; ++pbb[i]; }
        }
    };

    f32x16 acc[MB][NB];
#pragma unroll
    for (int mi = 0; mi < MB; ++mi)
#pragma unroll
        for (int ni = 0; ni < NB; ++ni) acc[mi][ni] = f32x16{0};
    const int wm0 = (wave >> 1) * (BM / 2), wn0 = (wave & 1) * (BN / 2);
    if (c_begin < c_end) issue(c_begin, 0);
    for (int c = c_begin; c < c_end; ++c) {
        const int cur = (c - c_begin) & 1;
        dg_wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if (c + 1 < c_end) issue(c + 1, cur ^ 1);
        dg_chunk_g<MB, NB, DG_RC, DG_RC, BM, BN>(As0 + cur * Cfg::ASZ, Bs0 + cur * Cfg::BSZ, acc, wm0, wn0, lane);
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
template <int BM, int BN>
__global__ __launch_bounds__(256) void conv_dgrad_dma_kernel(CdArgs g) {
#if defined(__HIP_DEVICE_COMPILE__)
    using Cfg = DgCfgG<BM, BN, 2, 2, DG_KC, DG_RC, 2>;
    constexpr int MB = Cfg::MB, NB = Cfg::NB, PA = Cfg::PA, PB = Cfg::PB;
    extern __shared__ __attribute__((aligned(1024))) float cd_smem[];
    float* As0 = cd_smem;
    float* Bs0 = cd_smem + 2 * Cfg::ASZ;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int s = g.s, KT = g.KH * g.KW;
    // heaviest class first (most taps): workgroups are dispatched in blockIdx order
    int py = 0, px = 0;
    if (s > 1) {
        const int zi = blockIdx.y / s, zj = blockIdx.y - zi * s;
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
    const int tn = t / tiles_m, tm = t - tn * tiles_m;
    const int m0 = tm * BM, n0 = tn * BN;
    const int ky0 = (py + g.p) % s, kx0 = (px + g.p) % s;
    const int nky = (g.KH - ky0 + s - 1) / s, nkx = (g.KW - kx0 + s - 1) / s;
    const int oyb = (py + g.p) / s, oxb = (px + g.p) / s;
    const int cpt = g.Co / DG_BK, total_chunks = nky * nkx * cpt;
    const int cps = (total_chunks + (int)gridDim.z - 1) / (int)gridDim.z;     // every class splits its own K range evenly
    const int c_begin = blockIdx.z * cps, c_end = min(total_chunks, c_begin + cps);

    auto make_rsrc = [](const float* p, unsigned bytes) -> dg_v4i {
        const unsigned long long a = (unsigned long long)p;
        dg_v4i r;
        r[0] = (int)(unsigned)a; r[1] = (int)((unsigned)(a >> 32) & 0xffffu); r[2] = (int)bytes; r[3] = 0x00020000;
        return r;
    };
    const unsigned cbytes = (unsigned)g.Co * 4u;
    const dg_v4i ra = make_rsrc(g.x, (unsigned)g.B * (unsigned)g.OH * (unsigned)g.OW * cbytes);          // dy
    const dg_v4i rb = make_rsrc(g.w, (unsigned)g.Co * (unsigned)KT * (unsigned)g.Ci * 4u);
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
    constexpr int LPRB = BN / 4;
    unsigned va[PA], vb[PB];
#pragma unroll
    for (int i = 0; i < PB; ++i) {
        const int k = (wave * PB + i) * (64 / LPRB) + lane / LPRB, c4 = lane % LPRB;
        vb[i] = ((unsigned)k * (unsigned)(KT * g.Ci) + (unsigned)(n0 + 4 * c4)) * 4u;
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
        tap_off_b = (unsigned)(((ky0 + s * tyi) * g.KW + kx0 + s * txi) * g.Ci) * 4u;
    };
    set_tap();
    const unsigned lds_a0 = (unsigned)(size_t)(dg_lds_ptr)As0 + (unsigned)(wave * PA) * 1024u;
    const unsigned lds_b0 = (unsigned)(size_t)(dg_lds_ptr)Bs0 + (unsigned)(wave * PB) * 1024u;
    const unsigned co_stride = (unsigned)(KT * g.Ci) * 4u * (unsigned)DG_BK;      // 32 output channels further
    auto issue = [&](int c, int buf) {
        const unsigned la = lds_a0 + (unsigned)buf * (unsigned)(Cfg::ASZ * 4), lb = lds_b0 + (unsigned)buf * (unsigned)(Cfg::BSZ * 4);
        const unsigned sa_off = (unsigned)cc * (DG_BK * 4u), sb_off = tap_off_b + (unsigned)cc * co_stride;
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
    if (c_begin < c_end) issue(c_begin, 0);
    for (int c = c_begin; c < c_end; ++c) {
        const int cur = (c - c_begin) & 1;
        dg_wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if (c + 1 < c_end) issue(c + 1, cur ^ 1);
        dg_chunk_g<MB, NB, DG_KC, DG_RC, BM, BN>(As0 + cur * Cfg::ASZ, Bs0 + cur * Cfg::BSZ, acc, wm0, wn0, lane);
    }

    // split-K slabs are whole dx images (classes write disjoint pixels of the same slab)
    float* o = g.part ? g.part + (size_t)blockIdx.z * ((size_t)g.B * g.H * g.W * N) : g.y;
    const bool direct = g.part == nullptr;
#pragma unroll
    for (int mi = 0; mi < MB; ++mi)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + wm0 + 32 * mi + mfma32_row(r, lane);
            if (m >= M) continue;
            const int b = m / (Hc * Wc), r2 = m - b * (Hc * Wc);
            const int y2 = r2 / Wc, x2 = r2 - y2 * Wc;
            const size_t off = ((size_t)(b * g.H + y2 * s + py) * g.W + x2 * s + px) * N;
#pragma unroll
            for (int ni = 0; ni < NB; ++ni) {
                const int col = n0 + wn0 + 32 * ni + (lane & 31);
                const float v = acc[mi][ni][r];
                o[off + col] = direct ? apply_act(v + (g.bias ? g.bias[col] : 0.0f), g.act) : v;
            }
        }
#endif
}

inline bool conv_dma_enabled() { static const bool off = getenv("MG_NO_CONV_DMA") != nullptr; return !off; }
inline bool conv_dma_fwd_ok(const mg_conv_geom* g) {
    return conv_dma_enabled() && g->precision == MG_PRECISION_F32 && g->Ci % DG_BK == 0 && g->Co % 64 == 0 &&
           (double)g->B * g->H * g->W * g->Ci * 4.0 < 2e9 && (double)g->Co * g->KH * g->KW * g->Ci * 4.0 < 2e9;
}
inline bool conv_dma_wgrad_ok(const mg_conv_geom* g) {
    return conv_dma_enabled() && g->precision == MG_PRECISION_F32 && g->Ci % 64 == 0 && g->Co % 64 == 0 &&
           (double)g->B * g->H * g->W * g->Ci * 4.0 < 2e9 && (double)g->B * g->OH * g->OW * g->Co * 4.0 < 2e9;
}
inline bool conv_dma_dgrad_ok(const mg_conv_geom* g) {
    return conv_dma_enabled() && g->precision == MG_PRECISION_F32 && !g->reflect && g->Co % DG_BK == 0 && g->Ci % 64 == 0 &&
           (double)g->B * g->OH * g->OW * g->Co * 4.0 < 2e9 && (double)g->Co * g->KH * g->KW * g->Ci * 4.0 < 2e9;
}
struct CdPlan { int bm, bn, splits, cps; };
// the dense plan's cost model (dense_plan above) on the convolution's GEMM view, DMA instances only
CdPlan conv_dma_plan(long long M, int N, int chunks, bool wgrad) {
    struct Cand { int bm, bn; double eff; };
    static const Cand cands[4] = {{64, 64, 0.83}, {64, 128, 0.885}, {128, 64, 0.855}, {128, 128, 0.91}};
    static const int split_opts[12] = {1, 2, 3, 4, 6, 8, 12, 16, 24, 32, 48, 64};
    CdPlan p{64, 64, 1, 1 << 28};
    int f_bm = 0, f_bn = 0, f_sp = 0;
    if (const char* f = getenv("MG_FORCE_CONV_DMA")) {      // tuning harness: "bm,bn,splits"
        if (sscanf(f, "%d,%d,%d", &f_bm, &f_bn, &f_sp) != 3) f_bm = f_bn = f_sp = 0;
    }
    double best = 1e300;
    for (const Cand& c : cands) {
        if (N % c.bn != 0 || (wgrad && c.bn != 64)) continue;
        if (f_bm && (c.bm != f_bm || c.bn != f_bn)) continue;
        const long long w = ((M + c.bm - 1) / c.bm) * (long long)(N / c.bn);
        const double tile_us = 2.0 * c.bm * c.bn * DG_BK / (157.3e12 / 256.0) * 1e6 / c.eff;
        for (int sp : split_opts) {
            if (sp > 1 && chunks / sp < 8) break;
            if (f_sp && sp != f_sp) continue;
            const int cps = (chunks + sp - 1) / sp;
            const int spl = (chunks + cps - 1) / cps;
            const long long wg = w * spl;
            double t = (double)((wg + 255) / 256) * tile_us * (cps + 1.2);
            if (wg < 512) t /= 0.85;
            if (spl > 1) t += (double)(spl + 1) * (double)M * N * 4.0 / 4e12 * 1e6 + 3.0;
            if (t < best) { best = t; p = {c.bm, c.bn, spl, cps}; }
        }
    }
    if (p.splits == 1) p.cps = 1 << 28;
    return p;
}
inline CdPlan conv_dma_fwd_plan(const mg_conv_geom* g) {
    return conv_dma_plan((long long)g->B * g->OH * g->OW, g->Co, g->KH * g->KW * (g->Ci / DG_BK), false);
}
inline CdPlan conv_dma_wgrad_plan(const mg_conv_geom* g) {
    const long long Mpx = (long long)g->B * g->OH * g->OW;
    return conv_dma_plan(g->Co, g->KH * g->KW * g->Ci, (int)((Mpx + DG_BK - 1) / DG_BK), true);
}
template <typename KernelT>
inline void cd_launch(KernelT kern, size_t lds, unsigned grid, const CdArgs& a, hipStream_t st) {
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, st, a);
}
inline CdArgs cd_args(const mg_conv_geom* g) {
    CdArgs a{};
    a.B = g->B; a.H = g->H; a.W = g->W; a.Ci = g->Ci; a.OH = g->OH; a.OW = g->OW; a.Co = g->Co; a.KH = g->KH; a.KW = g->KW;
    a.s = g->stride; a.p = g->pad; a.reflect = g->reflect;
    return a;
}
void conv_dma_fwd_launch(const mg_conv_geom* g, const CdPlan& p, const float* x, const float* w, const float* bias, float* y,
                         int act, float* part, hipStream_t st) {
    CdArgs a = cd_args(g);
    a.x = x; a.w = w; a.bias = bias; a.y = y; a.part = p.splits > 1 ? part : nullptr; a.act = act;
    const long long M = (long long)g->B * g->OH * g->OW;
    a.tiles_m = (int)((M + p.bm - 1) / p.bm); a.tiles_n = g->Co / p.bn; a.splits = p.splits; a.cps = p.cps;
    const unsigned grid = (unsigned)((long long)a.tiles_m * a.tiles_n * a.splits);
    if (p.bm == 128 && p.bn == 128) cd_launch(conv_fwd_dma_kernel<128, 128>, DgCfgG<128, 128, 2, 2, 0, 0, 2>::LDS_BYTES, grid, a, st);
    else if (p.bm == 64 && p.bn == 128) cd_launch(conv_fwd_dma_kernel<64, 128>, DgCfgG<64, 128, 2, 2, 0, 0, 2>::LDS_BYTES, grid, a, st);
    else if (p.bm == 128 && p.bn == 64) cd_launch(conv_fwd_dma_kernel<128, 64>, DgCfgG<128, 64, 2, 2, 0, 0, 2>::LDS_BYTES, grid, a, st);
    else cd_launch(conv_fwd_dma_kernel<64, 64>, DgCfgG<64, 64, 2, 2, 0, 0, 2>::LDS_BYTES, grid, a, st);
}
void conv_dma_wgrad_launch(const mg_conv_geom* g, const CdPlan& p, const float* x, const float* dy, float* dw, int accumulate,
                           float* part, hipStream_t st) {
    CdArgs a = cd_args(g);
    a.x = x; a.w = dy; a.y = dw; a.part = p.splits > 1 ? part : nullptr; a.accumulate = accumulate;
    a.tiles_m = (g->Co + p.bm - 1) / p.bm; a.tiles_n = g->KH * g->KW * g->Ci / p.bn; a.splits = p.splits; a.cps = p.cps;
    const unsigned grid = (unsigned)((long long)a.tiles_m * a.tiles_n * a.splits);
    if (p.bm == 128) cd_launch(conv_wgrad_dma_kernel<128, 64>, DgCfgG<128, 64, 2, 2, 1, 1, 2>::LDS_BYTES, grid, a, st);
    else cd_launch(conv_wgrad_dma_kernel<64, 64>, DgCfgG<64, 64, 2, 2, 1, 1, 2>::LDS_BYTES, grid, a, st);
}

// data gradient: the plan is made for the heaviest parity class (M = pixels of one class, all taps / stride^2 of the chunks)
inline CdPlan conv_dma_dgrad_plan(const mg_conv_geom* g) {
    const int s = g->stride;
    const long long Mc = (long long)g->B * ((g->H + s - 1) / s) * ((g->W + s - 1) / s);
    const int chunks = ((g->KH + s - 1) / s) * ((g->KW + s - 1) / s) * (g->Co / DG_BK);
    // the classes run side by side: count their workgroups when judging how full the chip is
    CdPlan best{64, 64, 1, 1 << 28};
    double best_t = 1e300;
    struct Cand { int bm, bn; double eff; };
    static const Cand cands[4] = {{64, 64, 0.83}, {64, 128, 0.885}, {128, 64, 0.855}, {128, 128, 0.91}};
    static const int split_opts[8] = {1, 2, 3, 4, 6, 8, 12, 16};
    int f_bm = 0, f_bn = 0, f_sp = 0;
    if (const char* f = getenv("MG_FORCE_CONV_DMA")) {
        if (sscanf(f, "%d,%d,%d", &f_bm, &f_bn, &f_sp) != 3) f_bm = f_bn = f_sp = 0;
    }
    const int total_k = g->KH * g->KW * (g->Co / DG_BK);          // chunks summed over the classes
    for (const Cand& c : cands) {
        if (g->Ci % c.bn != 0) continue;
        if (f_bm && (c.bm != f_bm || c.bn != f_bn)) continue;
        const long long w = ((Mc + c.bm - 1) / c.bm) * (long long)(g->Ci / c.bn);
        const double tile_us = 2.0 * c.bm * c.bn * DG_BK / (157.3e12 / 256.0) * 1e6 / c.eff;
        for (int sp : split_opts) {
            if (sp > 1 && chunks / sp < 8) break;
            if (f_sp && sp != f_sp) continue;
            const long long wg = w * s * s * sp;
            double t = (double)((wg + 255) / 256) * tile_us * ((double)total_k / (s * s) / sp + 1.2);
            if (wg < 512) t /= 0.85;
            if (sp > 1) t += (double)(sp + 1) * (double)g->B * g->H * g->W * g->Ci * 4.0 / 4e12 * 1e6 + 3.0;
            if (t < best_t) { best_t = t; best = {c.bm, c.bn, sp, 0}; }
        }
    }
    return best;
}
void conv_dma_dgrad_launch(const mg_conv_geom* g, const CdPlan& p, const float* dy, const float* w, const float* bias, float* dx,
                           int act, float* part, hipStream_t st) {
    CdArgs a = cd_args(g);
    a.x = dy; a.w = w; a.bias = bias; a.y = dx; a.part = p.splits > 1 ? part : nullptr; a.act = act;
    const int s = g->stride;
    const long long Mc = (long long)g->B * ((g->H + s - 1) / s) * ((g->W + s - 1) / s);
    const dim3 grid((unsigned)(((Mc + p.bm - 1) / p.bm) * (g->Ci / p.bn)), (unsigned)(s * s), (unsigned)p.splits);
    auto go = [&](auto kern, size_t lds) {
        hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, a);
    };
    if (p.bm == 128 && p.bn == 128) go(conv_dgrad_dma_kernel<128, 128>, DgCfgG<128, 128, 2, 2, 0, 1, 2>::LDS_BYTES);
    else if (p.bm == 64 && p.bn == 128) go(conv_dgrad_dma_kernel<64, 128>, DgCfgG<64, 128, 2, 2, 0, 1, 2>::LDS_BYTES);
    else if (p.bm == 128 && p.bn == 64) go(conv_dgrad_dma_kernel<128, 64>, DgCfgG<128, 64, 2, 2, 0, 1, 2>::LDS_BYTES);
    else go(conv_dgrad_dma_kernel<64, 64>, DgCfgG<64, 64, 2, 2, 0, 1, 2>::LDS_BYTES);
}
