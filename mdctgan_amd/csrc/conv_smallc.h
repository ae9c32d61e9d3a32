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
template <int CI>
__global__ __launch_bounds__(256) void conv_smallc_dgrad_kernel(Geom g, const float* __restrict__ dy,
                                                                const float* __restrict__ w, float* __restrict__ dx,
                                                                int hp) {
    // hp (MG_PRECISION_F16): operands rounded to float16 before the multiply, float32 accumulation, float16-valued output
    const int s = g.s, cls = blockIdx.y, py = cls / s, px = cls - py * s;
    const int Hc = (g.H - py + s - 1) / s, Wc = (g.W - px + s - 1) / s;
    const int Mc = g.B * Hc * Wc;
    const int m = blockIdx.x * 256 + threadIdx.x;
    if (Hc <= 0 || Wc <= 0) return;
    const bool live = m < Mc;
    const int mm = live ? m : 0;
    const int b = mm / (Hc * Wc), rem = mm - b * (Hc * Wc);
    const int yy = rem / Wc, xx = rem - yy * Wc;
    const int iy = s * yy + py, ix = s * xx + px;
    const int ky0 = (py + g.p) % s, kx0 = (px + g.p) % s;
    float acc[CI];
#pragma unroll
    for (int c = 0; c < CI; ++c) acc[c] = 0.0f;
    for (int ky = ky0; ky < g.KH; ky += s) {
        const int oy = (iy + g.p - ky) / s;          // exact: (iy + p - ky) is a multiple of s in this class
        const bool oky = (iy + g.p - ky) >= 0 && oy < g.OH;
        for (int kx = kx0; kx < g.KW; kx += s) {
            const int ox = (ix + g.p - kx) / s;
            const bool ok = live && oky && (ix + g.p - kx) >= 0 && ox < g.OW;
            const float* dp = dy + (size_t)((b * g.OH + (ok ? oy : 0)) * g.OW + (ok ? ox : 0)) * g.Co;
            const float* wp = w + (size_t)(ky * g.KW + kx) * CI;
            const int wstride = g.KH * g.KW * CI;
            for (int co = 0; co < g.Co; co += 4) {
                float4 d = ld4(dp + co);
                if (!ok) d = zero4();
                float dv[4] = {d.x, d.y, d.z, d.w};
                if (hp) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) dv[j] = round_h(dv[j]);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int c = 0; c < CI; ++c) {
                        float wv = wp[(size_t)(co + j) * wstride + c];
                        if (hp) wv = round_h(wv);
                        acc[c] = fmaf(dv[j], wv, acc[c]);
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

}  // namespace
