// Autocast (--fp16, MG_PRECISION_F16) convolutions of the weight-dominated, small-spatial layers -- the 2048-channel 4x8
// trunk of BASELINE configs[2]/[3] (75 MB of float16 weights against 1 MB of activations per layer), the 1x1 projections
// of its bottleneck-transformer blocks, the 1024-channel 8x16 blocks of configs[1] -- as dense float16 GEMMs
// (dense_gemm_h.h: LDS-DMA staging, v_mfma_f32_32x32x16_f16, float32 accumulation) over operands a pre-pass materialises:
//   forward        Y  [px][Co]        = Xcol  [px][(tap, ci)] * W16 [Co][(tap, ci)]^T       Xcol : float16 im2col of x
//   data gradient  dX [px][Ci]        = dYcol [px][(tap, co)] * W16 viewed [(tap, co)][ci]    dYcol: float16 gather of dy, the
//                                                                                            reflection aliases summed first
//   weight grad.   dW [Co][(tap, ci)] = dYt   [Co][px]        * XcolT [(tap, ci)][px]^T      both float16, px padded to 64
// W16 is a float16 copy of the OHWI weights (same layout): mg_conv_wino_prepare builds it once per weight version and the
// forward and data-gradient passes of a step share it through mg_wino_tiles.u -- the weights are read from HBM as 2 bytes
// per element, which is what bounds these layers.  The arithmetic is that of the TAG-2 implicit-GEMM kernels (operands
// rounded to float16, exact products, float32 accumulation, forward / data-gradient outputs rounded through float16).
// Eligibility: stride 1, Ci % 64 == 0, Co % 64 == 0, output pixels <= MG_H16_MAX_RATIO x Co (default 2, measured: configs[1] --fp16 90.5 / 92.4 / 92.5 steps/s at 1 / 2 / 4
// ).  Included inside conv_igemm.hip's second anonymous namespace.
#pragma once

__device__ __forceinline__ uint4 h16_pack8(const float4 a, const float4 b) {
    uint4 r;
    r.x = pack_h2(a.x, a.y); r.y = pack_h2(a.z, a.w); r.z = pack_h2(b.x, b.y); r.w = pack_h2(b.z, b.w);
    return r;
}

__global__ void h16_cast_kernel(const float* __restrict__ w, _Float16* __restrict__ o, size_t n8) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (size_t)gridDim.x * blockDim.x)
        *reinterpret_cast<uint4*>(o + 8 * i) = h16_pack8(ld4(w + 8 * i), ld4(w + 8 * i + 4));
}

// source pixel (b*H + iy)*W + ix of output pixel m and tap (ky, kx); -1: zero padding
__device__ __forceinline__ int h16_src_pixel(const Geom& g, int m, int ky, int kx) {
    const int b = m / (g.OH * g.OW), rem = m - b * (g.OH * g.OW);
    const int oy = rem / g.OW, ox = rem - oy * g.OW;
    int iy = oy * g.s - g.p + ky, ix = ox * g.s - g.p + kx;
    if (g.reflect) {
        iy = reflect_idx(iy, g.H);
        ix = reflect_idx(ix, g.W);
    } else if (iy < 0 || iy >= g.H || ix < 0 || ix >= g.W) {
        return -1;
    }
    return (b * g.H + iy) * g.W + ix;
}

// Xcol [M][KT * Ci] float16, one thread per 8 channels
__global__ void h16_im2col_kernel(Geom g, const float* __restrict__ x, _Float16* __restrict__ xcol) {
    const int C8 = g.Ci / 8, KT = g.KH * g.KW;
    const size_t M = (size_t)g.B * g.OH * g.OW, total = M * KT * C8;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c8 = (int)(i % C8);
        const int tap = (int)((i / C8) % KT);
        const int m = (int)(i / ((size_t)C8 * KT));
        const int px = h16_src_pixel(g, m, tap / g.KW, tap % g.KW);
        uint4 v = make_uint4(0, 0, 0, 0);
        if (px >= 0) {
            const float* s = x + (size_t)px * g.Ci + 8 * c8;
            v = h16_pack8(ld4(s), ld4(s + 4));
        }
        *reinterpret_cast<uint4*>(xcol + ((size_t)m * KT + tap) * g.Ci + 8 * c8) = v;
    }
}

// dYcol [Min][KT * Co] float16 for the stride-1 data gradient: row m = input pixel (b, iy, ix), column (tap, co) holds
// dy[b, iy + p - ky, ix + p - kx, co]; with reflection padding every padded position that aliases the pixel contributes
// (the transpose of the forward gather), summed in float32 before the float16 rounding like the implicit-GEMM kernel.
__global__ void h16_dycol_kernel(Geom g, const float* __restrict__ dy, _Float16* __restrict__ dycol) {
    const int C8 = g.Co / 8, KT = g.KH * g.KW, p = g.p;
    const size_t M = (size_t)g.B * g.H * g.W, total = M * KT * C8;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c8 = (int)(i % C8);
        const int tap = (int)((i / C8) % KT);
        const int m = (int)(i / ((size_t)C8 * KT));
        const int ky = tap / g.KW, kx = tap - ky * g.KW;
        const int b = m / (g.H * g.W), rem = m - b * (g.H * g.W);
        const int iy = rem / g.W, ix = rem - iy * g.W;
        int cy[3], cx[3];
        cy[0] = iy + p; cx[0] = ix + p;
        cy[1] = cy[2] = cx[1] = cx[2] = -1000000;
        if (g.reflect) {
            if (iy >= 1 && iy <= p) cy[1] = p - iy;
            if (iy >= g.H - 1 - p && iy <= g.H - 2) cy[2] = 2 * (g.H - 1) - iy + p;
            if (ix >= 1 && ix <= p) cx[1] = p - ix;
            if (ix >= g.W - 1 - p && ix <= g.W - 2) cx[2] = 2 * (g.W - 1) - ix + p;
        }
        float4 a0 = zero4(), a1 = zero4();
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const int oy = cy[a] - ky;
            if ((unsigned)oy >= (unsigned)g.OH) continue;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const int ox = cx[c] - kx;
                if ((unsigned)ox >= (unsigned)g.OW) continue;
                const float* s = dy + ((size_t)(b * g.OH + oy) * g.OW + ox) * g.Co + 8 * c8;
                add4(a0, ld4(s));
                add4(a1, ld4(s + 4));
            }
        }
        *reinterpret_cast<uint4*>(dycol + ((size_t)m * KT + tap) * g.Co + 8 * c8) = h16_pack8(a0, a1);
    }
}

// Transposed float16 gather: out[(tap * C + c)][Mp] = src[pixel(m, tap)][c] (0 past M / in the padding), 64 x 64 tiles
// through LDS so both the float32 reads (along c) and the float16 writes (along m) are coalesced.  KT == 1 with an
// identity geometry transposes a plain [M][C] matrix (dy -> dYt).
__device__ __forceinline__ void h16_colT_body(const Geom& g, const float* __restrict__ src, int C, int M, int Mp, int identity,
                                              _Float16* __restrict__ out, int m0, int c0, int tap) {
    __shared__ _Float16 tile[64][72];                  // [c][m], pitch 72 halves: 16-byte aligned rows
    const int KT = identity ? 1 : g.KH * g.KW;
    const int t = threadIdx.x;
    const int ky = tap / (identity ? 1 : g.KW), kx = tap - ky * (identity ? 1 : g.KW);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int mi = (t >> 4) + 16 * i, c4 = t & 15;
        const int m = m0 + mi;
        float4 v = zero4();
        if (m < M) {
            const int px = identity ? m : h16_src_pixel(g, m, ky, kx);
            if (px >= 0) v = ld4(src + (size_t)px * C + c0 + 4 * c4);
        }
        tile[4 * c4 + 0][mi] = (_Float16)v.x;
        tile[4 * c4 + 1][mi] = (_Float16)v.y;
        tile[4 * c4 + 2][mi] = (_Float16)v.z;
        tile[4 * c4 + 3][mi] = (_Float16)v.w;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int c = (t >> 3) + 32 * j, m8 = t & 7;
        const uint4 v = *reinterpret_cast<const uint4*>(&tile[c][8 * m8]);
        *reinterpret_cast<uint4*>(out + ((size_t)tap * C + c0 + c) * Mp + m0 + 8 * m8) = v;
    }
    (void)KT;
}
__global__ __launch_bounds__(256) void h16_colT_kernel(Geom g, const float* __restrict__ src, int C, int M, int Mp,
                                                       int identity, _Float16* __restrict__ out) {
    h16_colT_body(g, src, C, M, Mp, identity, out, blockIdx.x * 64, blockIdx.y * 64, blockIdx.z);
}
// Both operands of the weight gradient in ONE launch: z == 0 transposes dy -> dYt [Co][Mp], z = 1 + tap gathers x -> XcolT rows of
// that tap (grid.y covers the wider of Co / 64 and Ci / 64: blocks past an operand's channels leave).  The two launches it replaces
// were 5-6 us each on the 2048-channel trunk (25 pairs per configs[2] --fp16 step); same stores, same values.
__global__ __launch_bounds__(256) void h16_colT2_kernel(Geom g, const float* __restrict__ dy, int Co, _Float16* __restrict__ dyt,
                                                        const float* __restrict__ x, int Ci, _Float16* __restrict__ xct, int M,
                                                        int Mp) {
    const int c0 = blockIdx.y * 64;
    if (blockIdx.z == 0) {
        if (c0 < Co) h16_colT_body(g, dy, Co, M, Mp, 1, dyt, blockIdx.x * 64, c0, 0);
    } else {
        if (c0 < Ci) h16_colT_body(g, x, Ci, M, Mp, 0, xct, blockIdx.x * 64, c0, (int)blockIdx.z - 1);
    }
}

inline double h16_max_ratio() {
    constexpr double r = 2.0;
    return r;
}
bool h16_ok(const mg_conv_geom* g) {
    constexpr bool off = false;
    if (off || g->precision != MG_PRECISION_F16 || g->stride != 1) return false;
    if (g->Ci % 64 != 0 || g->Co % 64 != 0) return false;
    const long long M = (long long)g->B * g->OH * g->OW, Min = (long long)g->B * g->H * g->W;
    const long long Mx = M > Min ? M : Min;
    if ((double)Mx > h16_max_ratio() * g->Co || Mx > 8192) return false;
    if ((double)g->Co * g->KH * g->KW * g->Ci * 2.0 >= 2e9) return false;       // 32-bit byte offsets of the DMA descriptors
    return true;
}
struct H16Plan { int bm, bn, splits, cps; };
// tile 128 x 128 (8 waves) -- 128 x 64 (4 waves) for the row-contiguous-B data gradient -- and the smallest power-of-two
// K split that gives every CU a workgroup (scripts/ubench/hgemm_bench.hip: the weight stream needs >= 256 workgroups in
// flight; more splits only add slab traffic)
H16Plan h16_plan(long long M, int N, int K, bool brc) {
    H16Plan p{128, 128, 1, 1 << 28};
    (void)brc;            // both operand forms run 128 x 128 tiles (the row-contiguous one since it reads with ds_read_b64_tr_b16)
    const long long tiles = ((M + p.bm - 1) / p.bm) * ((N + p.bn - 1) / p.bn);
    const int chunks = K / HG_BK;
    int s = 1;
    while (tiles * s < 256 && chunks / (2 * s) >= 4 && s < 64) s *= 2;
    if (s > 1) { p.cps = (chunks + s - 1) / s; p.splits = (chunks + p.cps - 1) / p.cps; }
    if (p.splits == 1) p.cps = 1 << 28;
    return p;
}
inline size_t h16_al(size_t bytes) { return (bytes + 255) / 256 * 256; }
inline int h16_mp(long long M) { return (int)((M + 63) / 64 * 64); }

size_t h16_weights_bytes(const mg_conv_geom* g) { return (size_t)g->Co * g->KH * g->KW * g->Ci * 2; }
size_t h16_fwd_ws(const mg_conv_geom* g) {
    const long long M = (long long)g->B * g->OH * g->OW;
    const int K = g->KH * g->KW * g->Ci;
    const H16Plan p = h16_plan(M, g->Co, K, false);
    return h16_al(h16_weights_bytes(g)) + h16_al((size_t)M * K * 2) + h16_al((size_t)p.splits * M * g->Co * 4) + 256;
}
size_t h16_dgrad_ws(const mg_conv_geom* g) {
    const long long M = (long long)g->B * g->H * g->W;
    const int K = g->KH * g->KW * g->Co;
    const H16Plan p = h16_plan(M, g->Ci, K, true);
    return h16_al(h16_weights_bytes(g)) + h16_al((size_t)M * K * 2) + h16_al((size_t)p.splits * M * g->Ci * 4) + 256;
}
size_t h16_wgrad_cs_offset(const mg_conv_geom* g) {      // byte offset of the bias-gradient column-sum scratch
    const long long M = (long long)g->B * g->OH * g->OW;
    const int Mp = h16_mp(M), N = g->KH * g->KW * g->Ci;
    const H16Plan p = h16_plan(g->Co, N, Mp, false);
    return h16_al((size_t)g->Co * Mp * 2) + h16_al((size_t)N * Mp * 2) + (p.splits > 1 ? h16_al((size_t)p.splits * g->Co * N * 4) : 0);
}
size_t h16_wgrad_ws(const mg_conv_geom* g) {
    const long long M = (long long)g->B * g->OH * g->OW;
    return h16_wgrad_cs_offset(g) + (mg_colsum_workspace(M, g->Co) + 255) / 256 * 256 + 256;
}
inline unsigned h16_grid(size_t n) {
    size_t b = (n + 255) / 256;
    return (unsigned)(b > 16384 ? 16384 : (b < 1 ? 1 : b));
}
int h16_prepare(const mg_conv_geom* g, const float* w, void* w16, hipStream_t st) {
    const size_t n8 = (size_t)g->Co * g->KH * g->KW * g->Ci / 8;
    hipLaunchKernelGGL(h16_cast_kernel, dim3(h16_grid(n8)), dim3(256), 0, st, w, (_Float16*)w16, n8);
    MG_CHECK_LAUNCH();
    return MG_OK;
}
inline bool h16_deep(long long M, int N, int splits, bool brc) {
    (void)brc;
    return ((M + 127) / 128) * ((N + 127) / 128) * splits <= 512;
}
inline void h16_launch(const H16Plan& p, bool brc, const HgArgs& a, hipStream_t st) {
    // three LDS buffers: two chunks in flight per workgroup -- the weights come cold from HBM every step (1.4 GB of float16
    // copies per configs[2] iteration, nothing stays in the 256 MiB Infinity Cache) and one chunk per CU does not cover that
    // latency (scripts/ubench/hgemm_bench with HG_ROT=6: 43.5 -> 36.7 us on the 2048-channel trunk shape)
    // Only when the grid is a single round (<= 2 workgroups per CU would fit anyway): the weight gradient's thousands of
    // short workgroups want two 64 KiB workgroups per CU instead (same harness: 56 vs 78 us).
    const bool deep = h16_deep(a.M, a.N, p.splits, brc);
    if (brc) {
        if (deep) hgemm_launch<128, 128, 4, 2, true, 3>(a, st);
        else hgemm_launch<128, 128, 4, 2, true, 2>(a, st);
    } else {
        if (deep) hgemm_launch<128, 128, 4, 2, false, 3>(a, st);
        else hgemm_launch<128, 128, 4, 2, false, 2>(a, st);
    }
}

// MG_NO_HGEMM_SA=1: the round-2..5 kernels (im2col matrix + hgemm_kernel) instead of the weight-streaming form (read per call: the
// bit-identity test flips it)
inline bool h16_sa_on() { return getenv("MG_NO_HGEMM_SA") == nullptr; }
// x16_pre (nullable): float16(x) already written by x's producer (MG_TILES_V_FILLED)
int h16_fwd(const mg_conv_geom* g, const float* x, const float* w, const float* bias, float* y, int act, char* ws,
            hipStream_t st, const void* w16_pre, const void* x16_pre = nullptr) {
    const long long M = (long long)g->B * g->OH * g->OW;
    const int K = g->KH * g->KW * g->Ci;
    const H16Plan p = h16_plan(M, g->Co, K, false);
    _Float16* w16 = (_Float16*)ws;
    _Float16* xcol = (_Float16*)(ws + h16_al(h16_weights_bytes(g)));
    float* part = (float*)((char*)xcol + h16_al((size_t)M * K * 2));
    if (w16_pre) w16 = (_Float16*)const_cast<void*>(w16_pre);
    else if (h16_prepare(g, w, w16, st) != MG_OK) return MG_ERR_ARG;
    const Geom gg = to_geom(g);
    HgArgs a{};
    a.B = w16; a.C = y; a.part = p.splits > 1 ? part : nullptr;
    a.M = (int)M; a.N = g->Co; a.K = K; a.ldb = K; a.splits = p.splits; a.cps = p.cps;
    a.bias = (act == MG_ACT_NONE) ? bias : nullptr; a.round_f16 = 1; a.accumulate = 0; a.b_cpt = 1 << 30; a.b_tap_stride = 0;
    const bool sa = h16_sa_on() && hgemm_sa_ok(a) && (size_t)g->B * g->H * g->W * g->Ci * 2 < (1ull << 31);
    if (sa) {
        // no im2col matrix: the GEMM gathers its A rows from float16(x) [B*H*W][Ci] -- the producer's copy, or one cast pass
        const void* x16 = x16_pre;
        if (!x16) {
            const size_t n8 = (size_t)g->B * g->H * g->W * g->Ci / 8;
            hipLaunchKernelGGL(h16_cast_kernel, dim3(h16_grid(n8)), dim3(256), 0, st, x, xcol, n8);
            x16 = xcol;
        }
        a.A = x16; a.lda = g->Ci;
        a.a_cpt = g->Ci / HG_BK; a.gH = g->H; a.gW = g->W; a.gOH = g->OH; a.gOW = g->OW; a.gKW = g->KW; a.gs = g->stride; a.gp = g->pad;
        a.greflect = g->reflect;
    } else {
        hipLaunchKernelGGL(h16_im2col_kernel, dim3(h16_grid((size_t)M * K / 8)), dim3(256), 0, st, gg, x, xcol);
        a.A = xcol; a.lda = K;
    }
    const bool fused_epilogue = p.splits == 1 && act == MG_ACT_NONE;
    if (!fused_epilogue) { a.bias = nullptr; a.round_f16 = 0; if (p.splits == 1) { a.part = part; } }
    probe_begin(st);
    if (sa) hgemm_sa_launch<false, true>(a, st);
    else h16_launch(p, false, a, st);
    probe_end(st);
    if (!fused_epilogue) {
        const size_t n = (size_t)M * g->Co;
        if (g_fwd_defer && act == MG_ACT_NONE && p.splits > 1) {
            *g_fwd_defer = FwdDefer{part, p.splits, bias, 1, true};        // the caller's InstanceNorm kernel finishes the sum
        } else {
            hipLaunchKernelGGL(splitk_epilogue_kernel, dim3(h16_grid(n / 4)), dim3(256), 0, st, (const float*)part, p.splits, n, g->Co,
                               bias, act, y, 1);
        }
    }
    MG_CHECK_LAUNCH();
    return MG_OK;
}

// addend (nullable in / out): a tensor to add to dx (mg_wino_tiles.add); consumed -- *addend set to NULL -- when the split-K
// epilogue launch can take it
int h16_dgrad(const mg_conv_geom* g, const float* dy, const float* w, float* dx, char* ws, hipStream_t st, const void* w16_pre,
              const float** addend = nullptr) {
    const long long M = (long long)g->B * g->H * g->W;
    const int KT = g->KH * g->KW, K = KT * g->Co;
    const H16Plan p = h16_plan(M, g->Ci, K, true);
    _Float16* w16 = (_Float16*)ws;
    _Float16* dycol = (_Float16*)(ws + h16_al(h16_weights_bytes(g)));
    float* part = (float*)((char*)dycol + h16_al((size_t)M * K * 2));
    if (w16_pre) w16 = (_Float16*)const_cast<void*>(w16_pre);
    else if (h16_prepare(g, w, w16, st) != MG_OK) return MG_ERR_ARG;
    const Geom gg = to_geom(g);
    hipLaunchKernelGGL(h16_dycol_kernel, dim3(h16_grid((size_t)M * K / 8)), dim3(256), 0, st, gg, dy, dycol);
    HgArgs a{};
    a.A = dycol; a.B = w16; a.C = dx; a.part = p.splits > 1 ? part : nullptr;
    a.M = (int)M; a.N = g->Ci; a.K = K; a.lda = K; a.ldb = KT * g->Ci; a.splits = p.splits; a.cps = p.cps;
    a.bias = nullptr; a.round_f16 = 1; a.accumulate = 0; a.b_cpt = g->Co / HG_BK; a.b_tap_stride = g->Ci;
    probe_begin(st);
    if (h16_sa_on() && hgemm_sa_ok(a)) hgemm_sa_launch<true, false>(a, st);
    else h16_launch(p, true, a, st);
    probe_end(st);
    if (p.splits > 1) {
        const size_t n = (size_t)M * g->Ci;
        const float* ad = addend ? *addend : nullptr;
        if (addend) *addend = nullptr;
        hipLaunchKernelGGL(splitk_epilogue_kernel, dim3(h16_grid(n / 4)), dim3(256), 0, st, (const float*)part, p.splits, n, g->Ci,
                           (const float*)nullptr, MG_ACT_NONE, dx, 1, ad);
    }
    MG_CHECK_LAUNCH();
    return MG_OK;
}

// Short reductions (the padded pixel count is the K of this GEMM: 256 on the 2048-channel 4x8 trunk) run A-stationary
// (hgemm_as_kernel); MG_HGEMM_AS=0 keeps them on hgemm_kernel.
inline bool h16_wgrad_as(const mg_conv_geom* g) {
    constexpr bool on = true;
    const long long M = (long long)g->B * g->OH * g->OW;
    return on && hgemm_as_ok(g->Co, g->KH * g->KW * g->Ci, h16_mp(M));
}
// dw16 != NULL (only with h16_wgrad_as(g)): the gradient is stored as float16 there and dw is not touched
int h16_wgrad(const mg_conv_geom* g, const float* x, const float* dy, float* dw, int accumulate, char* ws, hipStream_t st,
              float* found_inf, void* dw16 = nullptr) {
    const long long M = (long long)g->B * g->OH * g->OW;
    const int Mp = h16_mp(M), N = g->KH * g->KW * g->Ci;
    const H16Plan p = h16_plan(g->Co, N, Mp, false);
    _Float16* dyt = (_Float16*)ws;
    _Float16* xct = (_Float16*)(ws + h16_al((size_t)g->Co * Mp * 2));
    float* part = (float*)((char*)xct + h16_al((size_t)N * Mp * 2));
    const Geom gg = to_geom(g);
    constexpr bool two = false;
    if (two) {
        hipLaunchKernelGGL(h16_colT_kernel, dim3(Mp / 64, g->Co / 64, 1), dim3(256), 0, st, gg, dy, g->Co, (int)M, Mp, 1, dyt);
        hipLaunchKernelGGL(h16_colT_kernel, dim3(Mp / 64, g->Ci / 64, g->KH * g->KW), dim3(256), 0, st, gg, x, g->Ci, (int)M, Mp, 0, xct);
    } else {
        const int cy = (g->Co > g->Ci ? g->Co : g->Ci) / 64;
        hipLaunchKernelGGL(h16_colT2_kernel, dim3(Mp / 64, cy, 1 + g->KH * g->KW), dim3(256), 0, st, gg, dy, g->Co, dyt, x, g->Ci, xct,
                           (int)M, Mp);
    }
    if (h16_wgrad_as(g)) {
        probe_begin(st);
        if (dw16) hgemm_as_launch(dyt, xct, dw16, g->Co, N, Mp, accumulate, found_inf, st, true);
        else hgemm_as_launch(dyt, xct, dw, g->Co, N, Mp, accumulate, found_inf, st);
        probe_end(st);
        MG_CHECK_LAUNCH();
        return MG_OK;
    }
    if (dw16) return MG_ERR_UNSUPPORTED;
    HgArgs a{};
    a.A = dyt; a.B = xct; a.C = dw; a.part = p.splits > 1 ? part : nullptr;
    a.M = g->Co; a.N = N; a.K = Mp; a.lda = Mp; a.ldb = Mp; a.splits = p.splits; a.cps = p.cps;
    a.bias = nullptr; a.round_f16 = 0; a.accumulate = accumulate; a.b_cpt = 1 << 30; a.b_tap_stride = 0;
    probe_begin(st);
    h16_launch(p, false, a, st);
    probe_end(st);
    if (p.splits > 1) {
        const size_t n = (size_t)g->Co * N;
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3(h16_grid(n / 4)), dim3(256), 0, st, (const float*)part, p.splits, n, dw,
                           accumulate);
    }
    MG_CHECK_LAUNCH();
    return MG_OK;
}
