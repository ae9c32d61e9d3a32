// Data gradient and weight gradient of convolutions with a handful of input channels (Ci <= 4): the first
// discriminator layer (3 -> 64, 4x4 stride 2, models/networks.py:649) and the generator stem (2 -> 64, 7x7 reflect,
// models/networks.py:207).  As implicit GEMMs these have N = Ci = 3 (data gradient) or a reduction over 10^5 pixels
// into a 64 x 48 result (weight gradient): the MFMA tiles run nearly empty and the loaders gather single floats
// (3-4 TFLOP/s measured).  Both are a few hundred FMAs per pixel on data that streams once, so they run on the VALU
// with the weights / patch values as wave-uniform scalar operands and the activations as coalesced vector loads.
// Included in conv_igemm.hip's anonymous namespace (Geom, reflect_idx, ld4, splitk_reduce_kernel).
#pragma once

namespace {

// dx[b][iy][ix][ci] = sum_{ky,kx,co} dy[b][(iy+p-ky)/s][(ix+p-kx)/s][co] * w[co][ky][kx][ci]     (zero padding)
// One thread per input pixel of one stride-parity class (blockIdx.y), so the tap set is uniform across the wave and
// the weights come in through scalar loads; dy rows are read as float4 per lane (each dy pixel serves (K/s)^2 lanes
// out of L1 / L2).
template <int CI, bool hp>      // hp compile-time: a run-time flag costs two conversions and a select per weight in the FMA loop
__global__ __launch_bounds__(256) void conv_smallc_dgrad_kernel(Geom g, const float* __restrict__ dy,
                                                                const float* __restrict__ w, float* __restrict__ dx) {
    // hp (MG_PRECISION_F16): operands rounded to float16 before the multiply, float32 accumulation, float16-valued output
    // The class's weights go through LDS as [tap][co] float4s (ci padded to 4): one broadcast ds_read_b128 per (tap, co)
    // instead of two scalar loads whose out-of-order return forces s_waitcnt lgkmcnt(0) in front of every FMA group.
    extern __shared__ __attribute__((aligned(16))) float4 wl_smallc[];
    const int s = g.s, cls = blockIdx.y, py = cls / s, px = cls - py * s;
    const int Hc = (g.H - py + s - 1) / s, Wc = (g.W - px + s - 1) / s;
    const int Mc = g.B * Hc * Wc;
    const int m = blockIdx.x * 256 + threadIdx.x;
    if (Hc <= 0 || Wc <= 0) return;
    const int ky0 = (py + g.p) % s, kx0 = (px + g.p) % s;
    const int nky = (g.KH - ky0 + s - 1) / s, nkx = (g.KW - kx0 + s - 1) / s;
    const int wstride = g.KH * g.KW * CI;
    for (int e = threadIdx.x; e < nky * nkx * g.Co; e += 256) {
        const int t = e / g.Co, co = e - t * g.Co;
        const int ky = ky0 + (t / nkx) * s, kx = kx0 + (t % nkx) * s;
        const float* wp = w + (size_t)co * wstride + (size_t)(ky * g.KW + kx) * CI;
        float v[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int c = 0; c < CI; ++c) v[c] = hp ? round_h(wp[c]) : wp[c];
        wl_smallc[e] = make_float4(v[0], v[1], v[2], v[3]);
    }
    __syncthreads();
    const bool live = m < Mc;
    const int mm = live ? m : 0;
    const int b = mm / (Hc * Wc), rem = mm - b * (Hc * Wc);
    const int yy = rem / Wc, xx = rem - yy * Wc;
    const int iy = s * yy + py, ix = s * xx + px;
    float acc[CI];
#pragma unroll
    for (int c = 0; c < CI; ++c) acc[c] = 0.0f;
    int t = 0;
    for (int ky = ky0; ky < g.KH; ky += s) {
        const int oy = (iy + g.p - ky) / s;          // exact: (iy + p - ky) is a multiple of s in this class
        const bool oky = (iy + g.p - ky) >= 0 && oy < g.OH;
        for (int kx = kx0; kx < g.KW; kx += s, ++t) {
            const int ox = (ix + g.p - kx) / s;
            const bool ok = live && oky && (ix + g.p - kx) >= 0 && ox < g.OW;
            const float* dp = dy + (size_t)((b * g.OH + (ok ? oy : 0)) * g.OW + (ok ? ox : 0)) * g.Co;
            const float4* wt = wl_smallc + (size_t)t * g.Co;
#pragma unroll 2
            for (int co = 0; co < g.Co; co += 4) {
                float4 d = ld4(dp + co);
                if (!ok) d = zero4();
                float dv[4] = {d.x, d.y, d.z, d.w};
                if (hp) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) dv[j] = round_h(dv[j]);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float4 wv4 = wt[co + j];
                    const float wv[4] = {wv4.x, wv4.y, wv4.z, wv4.w};
#pragma unroll
                    for (int c = 0; c < CI; ++c) acc[c] = fmaf(dv[j], wv[c], acc[c]);
                }
            }
        }
    }
    if (live) {
        float* o = dx + (size_t)((b * g.H + iy) * g.W + ix) * CI;
#pragma unroll
        for (int c = 0; c < CI; ++c) o[c] = hp ? round_h(acc[c]) : acc[c];
    }
}

// part[row][co][ky][kx*CI + ci] = sum over the pixels of output row (b, oy) of dy[p][co] * x[b][oy*s-p+ky][ox*s-p+kx][ci]
// One workgroup per output row, KH waves: wave ky owns filter row ky, lane <-> output channel (64 per blockIdx.y),
// KW * CI accumulators per thread.  The KH input rows the output row touches are staged once into LDS with the padding
// (zero / reflection) materialised, so a pixel's KW * CI patch values are one contiguous, wave-uniform LDS segment
// (broadcast ds_read_b64) and the loop has no border cases; dy comes in as one coalesced 256-byte load per pixel and
// wave, eight pixels in flight.
template <int KH, int KW, int CI>
__global__ __launch_bounds__(KH * 64) void conv_smallc_wgrad_kernel(Geom g, const float* __restrict__ x,
                                                                    const float* __restrict__ dy,
                                                                    float* __restrict__ part, int rowlen, int hp) {
    constexpr int KWC = KW * CI, U = 8;
    static_assert(KWC % 2 == 0, "patch segments are read as float2");
    extern __shared__ __attribute__((aligned(16))) float xs_smallc[];
    const int ky = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int co = blockIdx.y * 64 + lane;
    const bool co_ok = co < g.Co;
    const int b = blockIdx.x / g.OH, oy = blockIdx.x - b * g.OH;
    // stage: xs[r][jx * CI + c] = x[b][iy(r)][jx - p][c] with padding applied, jx in [0, (OW-1)*s + KW)
    const int ncols = (g.OW - 1) * g.s + KW;
    for (int r = 0; r < KH; ++r) {
        int iy = oy * g.s - g.p + r;
        bool rok = true;
        if (g.reflect) iy = reflect_idx(iy, g.H); else rok = iy >= 0 && iy < g.H;
        const float* xr = x + (size_t)(b * g.H + (rok ? iy : 0)) * g.W * CI;
        for (int e = threadIdx.x; e < rowlen; e += KH * 64) {
            const int jx = e / CI, c = e - jx * CI;
            int ix = jx - g.p;
            bool ok = rok && jx < ncols;
            if (g.reflect) ix = reflect_idx(ix, g.W); else ok = ok && ix >= 0 && ix < g.W;
            const float xv = ok ? xr[(size_t)ix * CI + c] : 0.0f;
            xs_smallc[r * rowlen + e] = hp ? round_h(xv) : xv;
        }
    }
    __syncthreads();
    float acc[KWC];
#pragma unroll
    for (int k = 0; k < KWC; ++k) acc[k] = 0.0f;
    const float* row = xs_smallc + ky * rowlen;
    const float* dyp = dy + (size_t)(blockIdx.x * g.OW) * g.Co + (co_ok ? co : 0);
    const int step = g.s * CI;                         // floats between consecutive pixels' segments (even)
    for (int ox = 0; ox < g.OW; ox += U) {
        float d[U];
#pragma unroll
        for (int j = 0; j < U; ++j) d[j] = (co_ok && ox + j < g.OW) ? dyp[(size_t)(ox + j) * g.Co] : 0.0f;
        if (hp) {
#pragma unroll
            for (int j = 0; j < U; ++j) d[j] = round_h(d[j]);
        }
#pragma unroll
        for (int j = 0; j < U; ++j) {
            const int oxx = min(ox + j, g.OW - 1);      // past-the-end pixels carry d == 0
            const float2* seg = reinterpret_cast<const float2*>(row + oxx * step);
#pragma unroll
            for (int k = 0; k < KWC / 2; ++k) {
                const float2 v = seg[k];
                acc[2 * k] = fmaf(d[j], v.x, acc[2 * k]);
                acc[2 * k + 1] = fmaf(d[j], v.y, acc[2 * k + 1]);
            }
        }
    }
    if (co_ok) {
        float* o = part + ((size_t)blockIdx.x * g.Co + co) * (KH * KWC) + ky * KWC;
#pragma unroll
        for (int k = 0; k < KWC; ++k) o[k] = acc[k];
    }
}

// ---- forward of the same layers on the MFMA pipe ----------------------------------------------------------------------------
// y[b][oy][ox][co] = act(bias[co] + sum_k patch(b, oy, ox)[k] * w[co][k]),  k = (ky, kx, ci) in OHWI order, K = KH*KW*CI even.
// As a GEMM this is [pixels] x [K <= 98] x [Co]: the implicit-GEMM loaders of conv_igemm.hip gather single floats for it
// (2-3 channels per pixel: 29 TFLOP/s on the 2 -> 64 stem).  Here one workgroup owns an output row: the KH input rows it touches
// are staged once into LDS with the padding materialised (as in the weight gradient above), so the A fragment of
// v_mfma_f32_32x32x2_f32 -- lane (pixel r, k parity h) needs patch(r)[2j + h] -- is ONE ds_read_b32 at
// ky(k) * rowlen + ox * S * CI + kxci(k), consecutive pixels -> consecutive banks; the B fragments (w[co][2j + h], K / 2 per
// 32-channel block) sit in LDS for all the rows a workgroup does (rows_per_wg consecutive ones, so that staging them is spread)
// and are read beside the A fragment.  A wave takes 32-pixel blocks of the row, 64 output channels each.
// The B fragments are read from LDS inside the loop (round 6).  Holding them in 2 x K/2 registers (rounds 4-5) needed ~260 VGPRs:
// ONE workgroup -- one wave per SIMD -- fitted a CU, and the matrix pipe idled through every staging barrier and store burst
// (0.37 of the f32 peak at the inference batch).  From LDS (row pitch K | 1: conflict-free) a wave needs 81, four workgroups are
// resident and cover each other's gaps: 460 -> 358 us at batch 64, 65 -> 51 us at batch 8; the same MFMA sequence on the same
// operands, so the same bits.
template <int KH, int KW, int CI, int S>
__global__ __launch_bounds__(256, 3) void conv_smallc_fwd_kernel(Geom g, const float* __restrict__ x,
                                                                 const float* __restrict__ w, const float* __restrict__ bias,
                                                                 float* __restrict__ y, int act, int rowlen, int rows_per_wg,
                                                                 int hp) {
    constexpr int KWC = KW * CI, K = KH * KWC, KS = K | 1, KWC2 = KWC / 2;
    static_assert(K % 2 == 0 && KWC % 2 == 0, "k pairs never straddle a filter row");
    extern __shared__ __attribute__((aligned(16))) float xs_smallc[];
    float* wl = xs_smallc + KH * rowlen;            // [64][KS] weights of this channel block
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int co0 = blockIdx.y * 64;
    const int ncols = (g.OW - 1) * S + KW;
    const int nrows = g.B * g.OH;
    {   // 64 x K weights of this block (Co % 64 == 0, K even: whole float2s), all loads of a thread in flight
        const float2* wsrc = reinterpret_cast<const float2*>(w + (size_t)co0 * K);
        constexpr int W2 = 64 * K / 2;
#pragma unroll
        for (int it = 0; it < (W2 + 255) / 256; ++it) {
            const int e = threadIdx.x + 256 * it;
            if (e < W2) {
                float2 v = wsrc[e];
                if (hp) { v.x = round_h(v.x); v.y = round_h(v.y); }
                const int co = (2 * e) / K, k = 2 * e - co * K;
                wl[co * KS + k] = v.x;
                wl[co * KS + k + 1] = v.y;
            }
        }
    }
    __syncthreads();
    // B fragments of this lane: w[co0 + 32 nb + r][2 j + h]
    const float* wb0 = wl + r * KS + h;
    const float* wb1 = wl + (32 + r) * KS + h;
    const float bv0 = (bias && co0 + r < g.Co) ? bias[co0 + r] : 0.0f;
    const float bv1 = (bias && co0 + 32 + r < g.Co) ? bias[co0 + 32 + r] : 0.0f;
    const int nblk = (g.OW + 31) / 32;
    const int row_end = min(nrows, (int)(blockIdx.x + 1) * rows_per_wg);
    // the staged rows of the NEXT output row travel through registers while this one is multiplied (NIT loads in flight per
    // thread instead of one dependent load per loop trip)
    constexpr int NIT = 16;
    const int nstage = KH * rowlen;                  // <= 256 * NIT (checked by the launcher)
    float pre[NIT];
    auto fetch = [&](int row) {
        const int b = row / g.OH, oy = row - b * g.OH;
        int rr = threadIdx.x / rowlen, ee = threadIdx.x - rr * rowlen;       // (staged row, element) of e, carried along
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int e = threadIdx.x + 256 * it;
            float xv = 0.0f;
            if (e < nstage) {
                const int jx = ee / CI, c = ee - jx * CI;
                int iy = oy * S - g.p + rr, ix = jx - g.p;
                bool ok = jx < ncols;
                if (g.reflect) { iy = reflect_idx(iy, g.H); ix = reflect_idx(ix, g.W); }
                else ok = ok && iy >= 0 && iy < g.H && ix >= 0 && ix < g.W;
                if (ok) xv = x[((size_t)(b * g.H + iy) * g.W + ix) * CI + c];
            }
            pre[it] = hp ? round_h(xv) : xv;
            ee += 256;
            while (ee >= rowlen) { ee -= rowlen; ++rr; }
        }
    };
    if ((int)(blockIdx.x * rows_per_wg) < row_end) fetch(blockIdx.x * rows_per_wg);
    for (int row = blockIdx.x * rows_per_wg; row < row_end; ++row) {
        __syncthreads();                             // the previous row's fragments are read
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int e = threadIdx.x + 256 * it;
            if (e < nstage) xs_smallc[e] = pre[it];
        }
        __syncthreads();
        if (row + 1 < row_end) fetch(row + 1);
        float* yr = y + ((size_t)row * g.OW) * g.Co;
        for (int mb = wave; mb < nblk; mb += 4) {
            const int ox = min(32 * mb + r, g.OW - 1);
            const float* ap = xs_smallc + ox * (S * CI);
            f32x16 acc0 = f32x16{0}, acc1 = f32x16{0};
            {
                // A fragment: lane (pixel r, k parity h) needs patch(r)[2 j + h] = filter row j / (KWC / 2), position 2 (j % (KWC / 2)) + h.
                // One filter row (KWC / 2 k pairs) per trip: its A and B values are in flight together and the next row's are not
                // hoisted above this row's MFMAs (fully unrolled, the compiler loads all 3 K / 2 values first and spills).
                const float* arow = ap + h;
                const float* b0 = wb0;
                const float* b1 = wb1;
#pragma unroll 1
                for (int ky = 0; ky < KH; ++ky) {
                    float av[KWC2], bv[2][KWC2];
#pragma unroll
                    for (int jj = 0; jj < KWC2; ++jj) {
                        av[jj] = arow[2 * jj];
                        bv[0][jj] = b0[2 * jj];
                        bv[1][jj] = b1[2 * jj];
                    }
#pragma unroll
                    for (int jj = 0; jj < KWC2; ++jj) {
                        acc0 = mfma32x32x2(av[jj], bv[0][jj], acc0);
                        acc1 = mfma32x32x2(av[jj], bv[1][jj], acc1);
                    }
                    arow += rowlen;
                    b0 += KWC;
                    b1 += KWC;
                }
            }
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int px = 32 * mb + mfma32_row(q, lane);
                if (px >= g.OW) continue;
                const int c0 = co0 + r, c1 = co0 + 32 + r;
                if (c0 < g.Co) {
                    const float v = apply_act(acc0[q] + bv0, act);
                    yr[(size_t)px * g.Co + c0] = hp ? round_h(v) : v;
                }
                if (c1 < g.Co) {
                    const float v = apply_act(acc1[q] + bv1, act);
                    yr[(size_t)px * g.Co + c1] = hp ? round_h(v) : v;
                }
            }
        }
    }
}

// ---- weight gradient of the same layers on the MFMA pipe ---------------------------------------------------------------------
// part[prow][co][k] = sum over the pixels of an output row of dy[px][co] * patch(px)[k]: per row a [64 co] x [K] x [OW pixels]
// GEMM.  The VALU kernel above spends its time on the broadcast LDS reads of the patch (KW * CI / 2 ds_read_b64 per pixel and
// wave: 103 us on the stem).  Here the A fragment of v_mfma_f32_32x32x2_f32 is dy read straight from HBM (lane (co r, pixel
// parity h): 128 contiguous bytes per pixel), the B fragment one ds_read_b32 of the staged rows (lane (k r, h) at
// ky(k) * rowlen + kxci(k) + pixel * S * CI; k >= K reads a zeroed slot with stride 0).  A wave owns one 32-wide block of k
// and both 32-channel blocks; with fewer k blocks than waves the spare waves take every G-th pixel pair and write their own
// partial row (the reduction over rows follows anyway).
template <int KH, int KW, int CI, int S>
__global__ __launch_bounds__(256) void conv_smallc_wgrad_mfma_kernel(Geom g, const float* __restrict__ x,
                                                                     const float* __restrict__ dy, float* __restrict__ part,
                                                                     int rowlen, int hp) {
    constexpr int KWC = KW * CI, K = KH * KWC, NBLK = (K + 31) / 32, G = 4 / NBLK, U = 8;
    static_assert(NBLK == 1 || NBLK == 2 || NBLK == 4, "k blocks per wave layout");
    extern __shared__ __attribute__((aligned(16))) float xs_smallc[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int b = blockIdx.x / g.OH, oy = blockIdx.x - b * g.OH;
    const int co0 = blockIdx.y * 64;
    const int ncols = (g.OW - 1) * S + KW;
    const int nstage = KH * rowlen;                   // + 4 zeroed floats behind it
    {
        constexpr int NIT = 16;                       // nstage <= 256 * NIT (checked by the launcher)
        float pre[NIT];
        int rr = threadIdx.x / rowlen, ee = threadIdx.x - rr * rowlen;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int e = threadIdx.x + 256 * it;
            float xv = 0.0f;
            if (e < nstage) {
                const int jx = ee / CI, c = ee - jx * CI;
                int iy = oy * S - g.p + rr, ix = jx - g.p;
                bool ok = jx < ncols;
                if (g.reflect) { iy = reflect_idx(iy, g.H); ix = reflect_idx(ix, g.W); }
                else ok = ok && iy >= 0 && iy < g.H && ix >= 0 && ix < g.W;
                if (ok) xv = x[((size_t)(b * g.H + iy) * g.W + ix) * CI + c];
            }
            pre[it] = hp ? round_h(xv) : xv;
            ee += 256;
            while (ee >= rowlen) { ee -= rowlen; ++rr; }
        }
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int e = threadIdx.x + 256 * it;
            if (e < nstage) xs_smallc[e] = pre[it];
        }
        if (threadIdx.x < 4) xs_smallc[nstage + threadIdx.x] = 0.0f;
    }
    __syncthreads();
    const int nb = wave % NBLK, grp = wave / NBLK;
    const int k = 32 * nb + r;
    const bool kok = k < K;
    const int ky = kok ? k / KWC : 0;
    const int bbase = kok ? ky * rowlen + (k - ky * KWC) : nstage;
    const int bstep = kok ? S * CI : 0;
    // the two 32-channel blocks are the even and the odd channels of the 64: a lane's A values for both are ONE 8-byte load
    const float* dyr = dy + (size_t)blockIdx.x * g.OW * g.Co + co0 + 2 * r;
    const bool c0ok = co0 + 2 * r + 1 < g.Co;       // Co % 64 == 0 at every call site; a ragged block loads nothing
    f32x16 acc0 = f32x16{0}, acc1 = f32x16{0};
    const int npairs = (g.OW + 1) / 2;
    for (int pp0 = grp; pp0 < npairs; pp0 += G * U) {
        float a0[U], a1[U], bv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int px = 2 * (pp0 + u * G) + h;
            const bool ok = px < g.OW;
            const int pc = ok ? px : g.OW - 1;
            const float2 av = (ok && c0ok) ? *reinterpret_cast<const float2*>(dyr + (size_t)pc * g.Co) : make_float2(0.0f, 0.0f);
            a0[u] = av.x;
            a1[u] = av.y;
            bv[u] = xs_smallc[bbase + pc * bstep];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (hp) { a0[u] = round_h(a0[u]); a1[u] = round_h(a1[u]); }
            acc0 = mfma32x32x2(a0[u], bv[u], acc0);
            acc1 = mfma32x32x2(a1[u], bv[u], acc1);
        }
    }
    if (kok) {
        float* o = part + ((size_t)(blockIdx.x * G + grp) * g.Co + co0) * K + k;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int cr = mfma32_row(q, lane);
            if (co0 + 2 * cr + 1 < g.Co) {
                o[(size_t)(2 * cr) * K] = acc0[q];
                o[(size_t)(2 * cr + 1) * K] = acc1[q];
            }
        }
    }
}

}  // namespace
