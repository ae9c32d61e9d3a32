// Single-output-channel convolutions (the generator's 7x7 tanh head, networks.py:244,352; the PatchGAN output layer,
// networks.py:668-670) as a TAP GEMM on the LDS-DMA kernels (dense_gemm.h) -- Ci % 64 == 0, KH * KW <= 64.
// MG_PRECISION_F16 (autocast) runs the SAME float32 GEMMs on operands rounded to float16 values first (x by a rounding
// pass, the weights while they are padded, dy while it is scattered): a product of two float16 values is exact in float32,
// so this is the arithmetic of the float16 MFMA with float32 accumulation; y and dx are rounded through float16.
// As a convolution GEMM such a layer has N = 1: 63/64 of an MFMA tile idle, which is why it ran on VALU dot products
// (conv_rowdot.hip: every input element re-read once per tap).  Contract over the channels FIRST instead:
//     Zt[tap][p]  = sum_ci W[tap][ci] * X[p][ci]            p = every INPUT pixel   (GEMM 64 x pixels x Ci, W zero-padded to 64 taps)
//     y[o]        = act(bias + sum_tap Zt[tap][src(o, tap)])                        (gather: padding / reflection live here)
// and backwards, with Gt[tap][p] = sum of dy[o] over the outputs o whose tap reads input pixel p (<= 1 with zero padding,
// <= 9 with reflection):
//     dW[tap][ci] = sum_p Gt[tap][p] * X[p][ci]              (GEMM 64 x Ci x pixels, split over the pixels)
//     dX[p][ci]   = sum_tap Gt[tap][p] * W[tap][ci]          (GEMM pixels x Ci x 64)
// Every x element is read once per pass, the FLOPs (64/taps more than necessary) run on the MFMA pipe, and the tap-major
// Zt / Gt make both the GEMM epilogue and the gather / scatter kernels fully coalesced.
// Included inside the anonymous namespace of conv_igemm.hip, after dense_gemm.h.
#pragma once

constexpr int CO1_TAPS = 64;       // tap rows of the padded weight matrix / Zt / Gt

inline bool co1_gemm_ok(const mg_conv_geom* g) {
    constexpr bool off = false;
    const long long Mp = (long long)g->B * g->H * g->W;
    return !off && (g->precision == MG_PRECISION_F32 || g->precision == MG_PRECISION_F16) && g->Co == 1 && g->KH * g->KW <= CO1_TAPS && g->Ci % 64 == 0 && Mp % 4 == 0 &&
           Mp * CO1_TAPS < (1LL << 29) && Mp * g->Ci < (1LL << 29);
}
inline long long co1_ldz(const mg_conv_geom* g) { return ((long long)g->B * g->H * g->W + 31) / 32 * 32; }
inline size_t co1_al(size_t b) { return (b + 255) & ~(size_t)255; }
inline size_t co1_wp_bytes(const mg_conv_geom* g) { return co1_al((size_t)CO1_TAPS * g->Ci * 4); }
inline size_t co1_zt_bytes(const mg_conv_geom* g) { return co1_al((size_t)CO1_TAPS * co1_ldz(g) * 4); }
inline int co1_wgrad_splits(const mg_conv_geom* g) {
    const int chunks = (int)(co1_ldz(g) / DG_BK), tiles = g->Ci / 64;
    int s = (256 + tiles - 1) / tiles;                    // one workgroup per CU: the slabs are re-read by one reduction pass
    if (s > chunks / 4) s = chunks / 4;
    return s < 1 ? 1 : s;
}
inline bool co1_half(const mg_conv_geom* g) { return g->precision == MG_PRECISION_F16; }
// autocast: x rounded to float16 values (float32 storage), made by the forward call and kept for the weight gradient
inline size_t co1_xr_bytes(const mg_conv_geom* g) { return co1_half(g) ? co1_al((size_t)g->B * g->H * g->W * g->Ci * 4) : 0; }
inline size_t co1_fwd_ws(const mg_conv_geom* g) { return co1_wp_bytes(g) + co1_zt_bytes(g) + co1_xr_bytes(g) + 256; }
inline size_t co1_dgrad_ws(const mg_conv_geom* g) { return co1_wp_bytes(g) + co1_zt_bytes(g) + 256; }
inline size_t co1_wgrad_ws(const mg_conv_geom* g) {
    return co1_zt_bytes(g) + co1_al((size_t)co1_wgrad_splits(g) * CO1_TAPS * g->Ci * 4) + 1024 + co1_xr_bytes(g) + 256;   // + dbias partials
}

// wp [64][Ci]: the OHWI weights of the single output channel are already [tap][ci]; rows >= taps are zero
__global__ void co1_pad_w_kernel(const float* __restrict__ w, int n_real, int n_all, float* __restrict__ wp, int half) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n_all; i += gridDim.x * blockDim.x)
        wp[i] = i < n_real ? (half ? round_h(w[i]) : w[i]) : 0.0f;
}
// xr = x rounded through float16 (n4 float4s)
__global__ void co1_round_kernel(const float* __restrict__ x, size_t n4, float* __restrict__ xr) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        float4 v = ld4(x + 4 * i);
        v.x = round_h(v.x); v.y = round_h(v.y); v.z = round_h(v.z); v.w = round_h(v.w);
        *reinterpret_cast<float4*>(xr + 4 * i) = v;
    }
}

// y[o] = act(bias + sum over taps of Zt[tap][source pixel]): one thread per output pixel, consecutive threads = consecutive
// ox, so every tap's read is a coalesced row segment of Zt
__global__ void co1_gather_kernel(Geom g, const float* __restrict__ zt, long long ldz, const float* __restrict__ bias, int act,
                                  float* __restrict__ y, int half) {
    const long long n = (long long)g.B * g.OH * g.OW;
    for (long long o = (long long)blockIdx.x * blockDim.x + threadIdx.x; o < n; o += (long long)gridDim.x * blockDim.x) {
        const int ox = (int)(o % g.OW);
        const long long r = o / g.OW;
        const int oy = (int)(r % g.OH), b = (int)(r / g.OH);
        float s = bias ? bias[0] : 0.0f;
        for (int ky = 0; ky < g.KH; ++ky) {
            int iy = oy * g.s - g.p + ky;
            if (g.reflect) iy = reflect_idx(iy, g.H);
            else if ((unsigned)iy >= (unsigned)g.H) continue;
            const float* zrow = zt + ((long long)b * g.H + iy) * g.W;
            for (int kx = 0; kx < g.KW; ++kx) {
                int ix = ox * g.s - g.p + kx;
                if (g.reflect) ix = reflect_idx(ix, g.W);
                else if ((unsigned)ix >= (unsigned)g.W) continue;
                s += zrow[(long long)(ky * g.KW + kx) * ldz + ix];
            }
        }
        s = apply_act(s, act);
        y[o] = half ? round_h(s) : s;
    }
}

// output coordinates whose tap k reads input coordinate i (one axis): u = o * s - p + k runs over the padded axis and
// reflects onto i for u in {i, -i, 2 (n - 1) - i}
__device__ __forceinline__ int co1_sources(int i, int k, int n, int on, int s, int p, int reflect, int (&out)[3]) {
    int cand[3] = {i, -i, 2 * (n - 1) - i};
    const int nc = reflect ? 3 : 1;
    int cnt = 0;
    for (int c = 0; c < nc; ++c) {
        const int u = cand[c];
        if (c == 1 && i == 0) continue;               // -0 == 0: the same position
        if (c == 2 && i == n - 1) continue;
        if (u < -p || u > n - 1 + p) continue;
        const int t = u + p - k;
        if (t < 0 || t % s != 0) continue;
        const int o = t / s;
        if (o < on) out[cnt++] = o;
    }
    return cnt;
}
// Gt[tap][p] (gather form, deterministic).  grid = (input rows / 4, taps); a workgroup = 4 rows x 64 threads of ONE tap,
// lanes on consecutive columns (coalesced dy reads and Gt writes): tap and row quantities are wave-uniform, a thread only
// resolves its columns.  With stride 1 a row of Gt is a shifted copy of a row of dy (one source per element away from the
// border); reflection adds the aliased rows / columns near the border.  Rows [taps, 64) and columns [Mp, ldz) are zeroed by
// co1_zero_pad_kernel (they meet zero weights / out-of-range x rows in the GEMMs, and 0 * garbage must not be NaN).
template <int S>
__global__ __launch_bounds__(256) void co1_scatter_kernel(Geom g, const float* __restrict__ dy, long long ldz, float* __restrict__ gt,
                                                          int half) {
    const int tap = blockIdx.y, row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;     // row = b * H + iy
    if (row >= g.B * g.H) return;
    const int b = row / g.H, iy = row - b * g.H;
    const int ky = tap / g.KW, kx = tap - ky * g.KW;
    int oys[3];
    const int ny = co1_sources(iy, ky, g.H, g.OH, S, g.p, g.reflect, oys);
    const float* dyb = dy + (long long)b * g.OH * g.OW;
    const float* drow = dyb + (long long)(ny ? oys[0] : 0) * g.OW;
    float* out = gt + (long long)tap * ldz + (long long)row * g.W;
    for (int ix = lane; ix < g.W; ix += 64) {
        float v = 0.0f;
        if (ny) {
            const bool interior = !g.reflect || (ix > g.p && ix < g.W - 1 - g.p);
            if (ny == 1 && interior) {          // the common case: exactly one source
                const int t = ix + g.p - kx;
                if (t >= 0 && t % S == 0 && t / S < g.OW) v = half ? round_h(drow[t / S]) : drow[t / S];
            } else {
                int oxs[3];
                const int nx = co1_sources(ix, kx, g.W, g.OW, S, g.p, g.reflect, oxs);
                for (int a = 0; a < ny; ++a)
                    for (int c = 0; c < nx; ++c) {
                        const float d = dyb[(long long)oys[a] * g.OW + oxs[c]];
                        v += half ? round_h(d) : d;
                    }
            }
        }
        out[ix] = v;
    }
}
// zero rows [taps, 64) and the columns [Mp, ldz) of the real tap rows
__global__ void co1_zero_pad_kernel(float* __restrict__ gt, int taps, long long Mp, long long ldz) {
    const long long tail = ldz - Mp, n_tail = (long long)taps * tail, n_rows = (long long)(CO1_TAPS - taps) * ldz;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n_tail + n_rows; i += (long long)gridDim.x * blockDim.x) {
        if (i < n_tail) gt[(i / tail) * ldz + Mp + i % tail] = 0.0f;
        else gt[(long long)taps * ldz + (i - n_tail)] = 0.0f;
    }
}

// dw[i] (+)= sum over slabs of part[s][i] for the first n elements (= the real tap rows) of each [64][Ci] slab.  A workgroup
// owns 64 consecutive outputs; its sixteen 64-thread groups each sum every sixteenth slab, combined in fixed order
// (deterministic).
__global__ __launch_bounds__(1024) void co1_reduce_kernel(const float* __restrict__ part, int S, long long slab, int n,
                                                          float* __restrict__ out, int accumulate) {
    __shared__ float red[16][64];
    const int i = blockIdx.x * 64 + (threadIdx.x & 63), zg = threadIdx.x >> 6;
    float s = 0.0f;
    if (i < n)
        for (int z = zg; z < S; z += 16) s += part[(long long)z * slab + i];
    red[zg][threadIdx.x & 63] = s;
    __syncthreads();
    if (zg == 0 && i < n) {
        float t = 0.0f;
#pragma unroll
        for (int q = 0; q < 16; ++q) t += red[q][threadIdx.x];
        out[i] = accumulate ? out[i] + t : t;
    }
}

// dbias = sum of dy (one output channel): two fixed-order stages (deterministic)
__global__ void co1_sum_partial_kernel(const float* __restrict__ dy, long long n, float* __restrict__ part) {
    __shared__ float red[256];
    float s = 0.0f;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) s += dy[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) part[blockIdx.x] = red[0];
}
__global__ void co1_sum_final_kernel(const float* __restrict__ part, int n, float* __restrict__ out, int accumulate) {
    __shared__ float red[256];
    red[threadIdx.x] = (int)threadIdx.x < n ? part[threadIdx.x] : 0.0f;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = accumulate ? out[0] + red[0] : red[0];
}

inline unsigned co1_grid(long long n) {
    long long b = (n + 255) / 256;
    return (unsigned)(b > 16384 ? 16384 : (b < 1 ? 1 : b));
}

// Zt[64][ldz] = Wp[64][Ci] x X[Mp][Ci]^T   (columns [Mp, ldz) are never written and never gathered)
inline void co1_gemm_z(const mg_conv_geom* g, const float* wp, const float* x, float* zt, hipStream_t st) {
    DgArgs a{};
    a.A = wp; a.B = x; a.C = zt; a.part = nullptr;
    a.M = CO1_TAPS; a.N = (int)((long long)g->B * g->H * g->W); a.K = g->Ci; a.lda = g->Ci; a.ldb = g->Ci;
    a.ldc = (int)co1_ldz(g);
    a.P = 1; a.splits = 1; a.cps = 1 << 28;
    dgemm32g_launch<64, 128, 2, 2, DG_KC, DG_KC>(a, st);
}
// dWp[64][Ci] = Gt[64][ldz] x X[Mp][Ci], reduction over the pixels split into `splits` slabs
inline void co1_gemm_dw(const mg_conv_geom* g, const float* gt, const float* x, float* slabs, int splits, hipStream_t st) {
    DgArgs a{};
    a.A = gt; a.B = x; a.C = slabs; a.part = splits > 1 ? slabs : nullptr;
    a.M = CO1_TAPS; a.N = g->Ci; a.K = (int)co1_ldz(g); a.lda = (int)co1_ldz(g); a.ldb = g->Ci;
    a.kb = (int)((long long)g->B * g->H * g->W);          // x has Mp rows: the zero-padded tail of Gt meets zeros
    const int chunks = a.K / DG_BK;
    a.cps = (chunks + splits - 1) / splits;
    a.splits = (chunks + a.cps - 1) / a.cps;
    a.P = 1;
    dgemm32g_launch<64, 64, 2, 2, DG_KC, DG_RC>(a, st);
}
// dX[Mp][Ci] = Gt[64][Mp]^T x Wp[64][Ci]
inline void co1_gemm_dx(const mg_conv_geom* g, const float* gt, const float* wp, float* dx, hipStream_t st) {
    DgArgs a{};
    a.A = gt; a.B = wp; a.C = dx; a.part = nullptr;
    a.M = (int)((long long)g->B * g->H * g->W); a.N = g->Ci; a.K = CO1_TAPS; a.lda = (int)co1_ldz(g); a.ldb = g->Ci;
    a.P = 1; a.splits = 1; a.cps = 1 << 28;
    a.round_f16 = co1_half(g) ? 1 : 0;
    dgemm32g_launch<128, 64, 2, 2, DG_RC, DG_RC>(a, st);
}

inline void co1_pad_w(const mg_conv_geom* g, const float* w, float* wp, hipStream_t st) {
    hipLaunchKernelGGL(co1_pad_w_kernel, dim3(co1_grid(CO1_TAPS * g->Ci)), dim3(256), 0, st, w, g->KH * g->KW * g->Ci, CO1_TAPS * g->Ci, wp,
                       (int)co1_half(g));
}
// u: the caller's padded weights (mg_conv_wino_prepare) or null; md: the caller's Gt buffer (data gradient -> weight gradient)
inline void co1_round_x(const mg_conv_geom* g, const float* x, float* xr, hipStream_t st) {
    const size_t n4 = (size_t)g->B * g->H * g->W * g->Ci / 4;
    hipLaunchKernelGGL(co1_round_kernel, dim3(co1_grid((long long)n4)), dim3(256), 0, st, x, n4, xr);
}
int co1_fwd(const mg_conv_geom* g, const float* x, const float* w, const float* bias, float* y, int act, char* ws, hipStream_t st,
            const float* u, float* v) {
    float* wp = (float*)ws;
    float* zt = (float*)(ws + co1_wp_bytes(g));
    if (u) wp = const_cast<float*>(u);
    else co1_pad_w(g, w, wp, st);
    if (co1_half(g)) {          // the rounded x: into the caller's buffer (kept for the weight gradient) or the workspace
        float* xr = v ? v : (float*)(ws + co1_wp_bytes(g) + co1_zt_bytes(g));
        co1_round_x(g, x, xr, st);
        x = xr;
    }
    probe_begin(st);
    co1_gemm_z(g, wp, x, zt, st);
    probe_end(st);
    const Geom gg{g->B, g->H, g->W, g->Ci, g->OH, g->OW, g->Co, g->KH, g->KW, g->stride, g->pad, g->reflect};
    hipLaunchKernelGGL(co1_gather_kernel, dim3(co1_grid((long long)g->B * g->OH * g->OW)), dim3(256), 0, st, gg, (const float*)zt,
                       co1_ldz(g), bias, act, y, (int)co1_half(g));
    MG_CHECK_LAUNCH();
    return MG_OK;
}
inline void co1_scatter(const mg_conv_geom* g, const float* dy, float* gt, hipStream_t st) {
    const Geom gg{g->B, g->H, g->W, g->Ci, g->OH, g->OW, g->Co, g->KH, g->KW, g->stride, g->pad, g->reflect};
    const dim3 grid((unsigned)((g->B * g->H + 3) / 4), (unsigned)(g->KH * g->KW));
    const long long Mp = (long long)g->B * g->H * g->W, ldz = co1_ldz(g);
    const int taps = g->KH * g->KW;
    if (taps < CO1_TAPS || ldz > Mp)
        hipLaunchKernelGGL(co1_zero_pad_kernel, dim3(co1_grid((long long)taps * (ldz - Mp) + (long long)(CO1_TAPS - taps) * ldz)), dim3(256),
                           0, st, gt, taps, Mp, ldz);
    if (g->stride == 1) hipLaunchKernelGGL(co1_scatter_kernel<1>, grid, dim3(256), 0, st, gg, dy, ldz, gt, (int)co1_half(g));
    else hipLaunchKernelGGL(co1_scatter_kernel<2>, grid, dim3(256), 0, st, gg, dy, ldz, gt, (int)co1_half(g));
}
int co1_dgrad(const mg_conv_geom* g, const float* dy, const float* w, float* dx, char* ws, hipStream_t st, const float* u, float* md) {
    float* wp = (float*)ws;
    float* gt = md ? md : (float*)(ws + co1_wp_bytes(g));
    if (u) wp = const_cast<float*>(u);
    else co1_pad_w(g, w, wp, st);
    co1_scatter(g, dy, gt, st);
    probe_begin(st);
    co1_gemm_dx(g, gt, wp, dx, st);
    probe_end(st);
    MG_CHECK_LAUNCH();
    return MG_OK;
}
int co1_wgrad(const mg_conv_geom* g, const float* x, const float* dy, float* dw, float* dbias, int accumulate, char* ws,
              hipStream_t st, const float* v, const float* md) {
    float* gt = md ? const_cast<float*>(md) : (float*)ws;
    if (co1_half(g)) {
        if (v) {
            x = v;
        } else {
            float* xr = (float*)(ws + co1_zt_bytes(g) + co1_al((size_t)co1_wgrad_splits(g) * CO1_TAPS * g->Ci * 4) + 1024);
            co1_round_x(g, x, xr, st);
            x = xr;
        }
    }
    float* slabs = (float*)(ws + co1_zt_bytes(g));
    if (dbias) {
        float* part = (float*)(ws + co1_zt_bytes(g) + co1_al((size_t)co1_wgrad_splits(g) * CO1_TAPS * g->Ci * 4));
        hipLaunchKernelGGL(co1_sum_partial_kernel, dim3(256), dim3(256), 0, st, dy, (long long)g->B * g->OH * g->OW, part);
        hipLaunchKernelGGL(co1_sum_final_kernel, dim3(1), dim3(256), 0, st, (const float*)part, 256, dbias, accumulate);
    }
    const int splits = co1_wgrad_splits(g), taps = g->KH * g->KW;
    if (!md) co1_scatter(g, dy, gt, st);
    probe_begin(st);
    co1_gemm_dw(g, gt, x, slabs, splits, st);
    probe_end(st);
    const int chunks = (int)(co1_ldz(g) / DG_BK), cps = (chunks + splits - 1) / splits, real = (chunks + cps - 1) / cps;
    hipLaunchKernelGGL(co1_reduce_kernel, dim3((unsigned)((taps * g->Ci + 63) / 64)), dim3(1024), 0, st, (const float*)slabs, real,
                       (long long)CO1_TAPS * g->Ci, taps * g->Ci, dw, accumulate);
    MG_CHECK_LAUNCH();
    return MG_OK;
}
