// Dense GEMM on the f16 MFMA pipe (v_mfma_f32_32x32x16_f16, float32 accumulation) for the autocast (--fp16) path:
//   C[M][N] (float32) = A[M][K] (float16) * B[N][K]^T (float16), both operands k-contiguous, staged by LDS-DMA.
// Used for the small-spatial, weight-dominated convolutions of BASELINE configs[2]/[3] (the 2048-channel 4x8 trunk: 75 MB of
// float16 weights against 1 MB of activations per layer): the activation side is materialised as a float16 im2col matrix
// by a pre-pass (conv_h16.h), so forward, data gradient and weight gradient are all this one kernel --
//   forward        Y    [px][Co]       = Xcol  [px][(tap, ci)]  * W16  [Co][(tap, ci)]^T
//   data gradient  dX   [px][Ci]       = dYcol [px][(tap, co)]  * W16t [Ci][(tap, co)]^T
//   weight grad.   dW   [Co][(tap,ci)] = dYt   [Co][px]         * XcolT[(tap, ci)][px]^T
// Same structure as dgemm32g_kernel (dense_gemm.h): 64-deep chunks = 128-byte rows, XOR-swizzled unpadded LDS images,
// buffer_load ... lds issued from inline asm with counted vmcnt, two buffers, one barrier per chunk; one ds_read_b128 is the
// 8-half operand of one MFMA.  K % 64 == 0.  Included inside the anonymous namespace, after dense_gemm.h.
#pragma once

constexpr int HG_BK = 64;          // halves per chunk row (128 bytes)

// ---- float16 fragment reads shared with the implicit GEMMs (conv_dma.h) ----
// k-contiguous image: one ds_read_b128 = the 8 halves of an MFMA operand.  Row-contiguous image [64][R] halves: two
// ds_read_b64_tr_b16 (16 lanes fetch a [4 k][16 columns] block, every lane receives the 4 k of its column); the 16-byte
// slots of a k row are XOR-swizzled on the DMA source side (cd_rc_swz) so that the 8 row segments a 32-lane half reads fall
// into 8 different 32-byte bank groups.
template <int R>
__device__ __forceinline__ int cd_rc_swz(int k) { return R == 128 ? ((k & 3) << 2) : (((k >> 1) & 1) << 2); }

typedef __fp16 cd_h4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
typedef cd_h4 __attribute__((address_space(3))) * cd_h4_lds;

__device__ __forceinline__ f16x8 cd_frag_kc(const float* S, int row, int s, int kh) {
    return *reinterpret_cast<const f16x8*>(S + row * 32 + 4 * ((2 * s + kh) ^ ((row >> 1) & 7)));
}
// RC image [64][R] halves; col0 = tile-relative first column of the wave's 32-column block
template <int R>
__device__ __forceinline__ f16x8 cd_frag_rc(const float* S, int col0, int s, int lane) {
    static_assert(R == 64 || R == 128, "row-contiguous operand tiles are 64 or 128 columns wide");
    const int i = lane & 15;
    const int col = col0 + 16 * ((lane >> 4) & 1) + 4 * (i & 3);
    const int k = 16 * s + 8 * (lane >> 5) + (i >> 2);
    const char* p = reinterpret_cast<const char*>(S) + k * (2 * R) + 16 * ((col >> 3) ^ cd_rc_swz<R>(k)) + 2 * (col & 7);
    const cd_h4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4f16((cd_h4_lds)(p));
    const cd_h4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4f16((cd_h4_lds)(p + 8 * R));       // k + 4: same swizzle
    f16x8 v;
    v[0] = (_Float16)lo[0]; v[1] = (_Float16)lo[1]; v[2] = (_Float16)lo[2]; v[3] = (_Float16)lo[3];
    v[4] = (_Float16)hi[0]; v[5] = (_Float16)hi[1]; v[6] = (_Float16)hi[2]; v[7] = (_Float16)hi[3];
    return v;
}


struct HgArgs {
    const void* A;           // float16 [M][lda]
    const void* B;           // float16 [N][ldb]
    float* C;                // float32 [M][N]
    float* part;             // split-K slabs [splits][M][N] (nullptr: direct)
    const float* bias;       // [N] or nullptr (direct path only)
    int M, N, K, lda, ldb;
    int tiles_m, tiles_n, splits, cps;
    int round_f16;           // round the result through float16 (autocast output), direct path only
    int accumulate;          // C += result (direct path only)
    // BRC instances (data gradient): B is [K][N] with N contiguous -- the OHWI weights W[co][tap][n = ci] read with
    // k = tap * Co + co, i.e. k row (tap, co) starts at element (co * ldb + tap * b_tap_stride); ldb = KH*KW*Ci is the
    // distance between consecutive co, b_cpt = Co / 64 the chunks per tap.  A plain [K][N] matrix: ldb = N, b_cpt = 1 << 30.
    int b_cpt, b_tap_stride;
    // A-gather (hgemm_sa_kernel<.., AG = true>, the forward pass without an im2col matrix): A is the float16 activation
    // x16 [B*H*W][Ci] (lda = Ci) and row m of the GEMM at chunk c reads pixel src(m, tap = c / a_cpt) -- zero padding is an
    // out-of-range buffer offset, reflection a reflected index.  K = KH*KW*Ci, a_cpt = Ci / 64.
    int a_cpt, gH, gW, gOH, gOW, gKW, gs, gp, greflect;
};

template <int BM, int BN, int WGM, int WGN, bool BRC = false, int NBUF = 2>
struct HgCfg {
    static constexpr int NT = 64 * WGM * WGN, NW = WGM * WGN;
    static constexpr int MB = BM / WGM / 32, NB = BN / WGN / 32;
    static constexpr int ASZ = BM * 32, BSZ = BN * 32;                // 32-bit words per buffer (64 halves per row)
    static constexpr int PA = BM / 8 / NW, PB = BN / 8 / NW;          // 1 KiB pieces per wave and chunk
    static constexpr size_t LDS_BYTES = (size_t)NBUF * (ASZ + BSZ) * 4;
    static_assert(NBUF >= 2 && NBUF <= 4, "pipeline depth");
    static_assert((BM / 8) % NW == 0 && (BN / 8) % NW == 0, "pieces must divide evenly over the waves");
    static_assert(!BRC || (BN == 64 || BN == 128), "row-contiguous B: a 1 KiB piece is 8 or 4 whole k rows");
};

// BRC: B is row-contiguous ([K][N], see HgArgs).  Its LDS image is k-major ([64][BN] halves, filled by the same DMA in
// pieces of 1024 / (2 BN) whole k rows, slots swizzled); a lane gets the 8 k of its column with two transpose reads
// (cd_frag_rc) -- no second, transposed float16 copy of every weight tensor.
// NBUF LDS buffers keep NBUF - 1 chunks in flight: with one workgroup per CU (grid == 256 on the weight-streaming trunk
// shapes) a two-buffer loop has one 32 KiB chunk outstanding per CU -- a quarter of what the HBM latency needs.
template <int BM, int BN, int WGM, int WGN, bool BRC = false, int NBUF = 2>
__global__ __launch_bounds__(64 * WGM * WGN) void hgemm_kernel(HgArgs g) {
#if defined(__HIP_DEVICE_COMPILE__)
    using Cfg = HgCfg<BM, BN, WGM, WGN, BRC, NBUF>;
    constexpr int MB = Cfg::MB, NB = Cfg::NB, PA = Cfg::PA, PB = Cfg::PB;
    extern __shared__ __attribute__((aligned(1024))) float hg_smem[];
    float* As0 = hg_smem;
    float* Bs0 = hg_smem + NBUF * Cfg::ASZ;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tiles = g.tiles_m * g.tiles_n;
    const int L = xcd_remap(blockIdx.x, tiles * g.splits);
    // consecutive L share the K split and the n tile: an XCD streams one [BN][K / splits] weight panel through its L2 for
    // all m tiles before moving on
    const int sp = L / tiles;
    const int rem = L - sp * tiles;
    const int tn = rem / g.tiles_m, tm = rem - tn * g.tiles_m;
    const int m0 = tm * BM, n0 = tn * BN;
    const int total_chunks = g.K / HG_BK;
    const int c_begin = sp * g.cps, c_end = min(total_chunks, c_begin + g.cps);

    auto make_rsrc = [](const void* p, unsigned bytes) -> dg_v4i {
        const unsigned long long a = (unsigned long long)p;
        dg_v4i r;
        r[0] = (int)(unsigned)a;
        r[1] = (int)((unsigned)(a >> 32) & 0xffffu);
        r[2] = (int)bytes;
        r[3] = 0x00020000;
        return r;
    };
    const dg_v4i ra = make_rsrc(g.A, (unsigned)g.M * (unsigned)g.lda * 2u);
    const dg_v4i rb = make_rsrc(g.B, BRC ? 0x7fffffffu : (unsigned)g.N * (unsigned)g.ldb * 2u);
    unsigned va[PA], vb[PB];
#pragma unroll
    for (int i = 0; i < PA; ++i) {
        const int row = 8 * (wave * PA + i) + (lane >> 3), q = (lane & 7) ^ ((row >> 1) & 7);
        va[i] = (unsigned)min(m0 + row, g.M - 1) * (unsigned)g.lda * 2u + 16u * q;
    }
#pragma unroll
    for (int i = 0; i < PB; ++i) {
        if constexpr (BRC) {
            constexpr int LPR = BN / 8;                   // lanes per k row (16 bytes = 8 halves each)
            const int k = (wave * PB + i) * (64 / LPR) + lane / LPR, c8 = (lane % LPR) ^ cd_rc_swz<BN>(k);
            vb[i] = ((unsigned)k * (unsigned)g.ldb + (unsigned)min(n0 + 8 * c8, g.N - 8)) * 2u;
        } else {
            const int row = 8 * (wave * PB + i) + (lane >> 3), q = (lane & 7) ^ ((row >> 1) & 7);
            vb[i] = (unsigned)min(n0 + row, g.N - 1) * (unsigned)g.ldb * 2u + 16u * q;
        }
    }
    const unsigned lds_a0 = (unsigned)(size_t)(dg_lds_ptr)As0 + (unsigned)(wave * PA) * 1024u;
    const unsigned lds_b0 = (unsigned)(size_t)(dg_lds_ptr)Bs0 + (unsigned)(wave * PB) * 1024u;
    auto issue = [&](int c, int buf) {
        const unsigned off = (unsigned)c * (HG_BK * 2u);
        unsigned off_b = off;
        if (BRC) {
            const int tap = c / g.b_cpt, cc = c - tap * g.b_cpt;
            off_b = ((unsigned)tap * (unsigned)g.b_tap_stride + (unsigned)cc * (unsigned)HG_BK * (unsigned)g.ldb) * 2u;
        }
        const unsigned la = lds_a0 + (unsigned)buf * (unsigned)(Cfg::ASZ * 4), lb = lds_b0 + (unsigned)buf * (unsigned)(Cfg::BSZ * 4);
#pragma unroll
        for (int i = 0; i < PA; ++i) dg_dma16(va[i], ra, la + 1024u * i, off);
#pragma unroll
        for (int i = 0; i < PB; ++i) dg_dma16(vb[i], rb, lb + 1024u * i, off_b);
    };

    f32x16 acc[MB][NB];
#pragma unroll
    for (int mi = 0; mi < MB; ++mi)
#pragma unroll
        for (int ni = 0; ni < NB; ++ni) acc[mi][ni] = f32x16{0};
    const int wm0 = (wave / WGN) * (BM / WGM), wn0 = (wave % WGN) * (BN / WGN);
    const int r = lane & 31, kh = lane >> 5;
#if defined(HG_DEBUG_NO_LDS)
    f16x8 a[2][MB], b[2][NB];
#endif

    constexpr int AHEAD = NBUF - 1, PER = PA + PB;         // chunks in flight, DMA instructions per chunk and lane
#pragma unroll
    for (int i = 0; i < AHEAD; ++i)
        if (c_begin + i < c_end) issue(c_begin + i, i);
    int cur = 0, nxt = AHEAD % NBUF;
    for (int c = c_begin; c < c_end; ++c) {
        // chunk c has landed when at most the younger chunks' instructions are outstanding (fewer near the tail: wait for all)
        if (NBUF == 2 || c + AHEAD > c_end) dg_wait_vmcnt<0>();
        else dg_wait_vmcnt<(AHEAD - 1) * PER>();
#if !defined(HG_DEBUG_NO_BARRIER)
        __builtin_amdgcn_s_barrier();
#endif
        __builtin_amdgcn_sched_barrier(0);
#if !defined(HG_DEBUG_NO_DMA)
        if (c + AHEAD < c_end) issue(c + AHEAD, nxt);
#endif
        const float* As = As0 + cur * Cfg::ASZ;
        const float* Bs = Bs0 + cur * Cfg::BSZ;
#if !defined(HG_DEBUG_NO_LDS)
        f16x8 a[2][MB], b[2][NB];
#endif
        auto fetch = [&](int s, int buf) {
#if defined(HG_DEBUG_NO_LDS)
            if (c > c_begin) return;
#endif
#pragma unroll
            for (int mi = 0; mi < MB; ++mi) {
                const int row = wm0 + 32 * mi + r;
                a[buf][mi] = *reinterpret_cast<const f16x8*>(As + row * 32 + 4 * ((2 * s + kh) ^ ((row >> 1) & 7)));
            }
#pragma unroll
            for (int ni = 0; ni < NB; ++ni) {
                const int row = wn0 + 32 * ni + r;
                if constexpr (BRC) {
                    b[buf][ni] = cd_frag_rc<BN>(Bs, wn0 + 32 * ni, s, lane);
                } else {
                    b[buf][ni] = *reinterpret_cast<const f16x8*>(Bs + row * 32 + 4 * ((2 * s + kh) ^ ((row >> 1) & 7)));
                }
            }
        };
#if defined(HG_DEBUG_NO_COMPUTE)
        if (g.K < 0)
#endif
        {
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
        cur = cur + 1 == NBUF ? 0 : cur + 1;
        nxt = nxt + 1 == NBUF ? 0 : nxt + 1;
    }

    float* o = g.part ? g.part + (size_t)sp * ((size_t)g.M * g.N) : g.C;
    const bool direct = g.part == nullptr;
#pragma unroll
    for (int mi = 0; mi < MB; ++mi)
#pragma unroll
        for (int ni = 0; ni < NB; ++ni) {
            const int col = n0 + wn0 + 32 * ni + (lane & 31);
            const float bv = (direct && g.bias && col < g.N) ? g.bias[col] : 0.0f;
#pragma unroll
            for (int rr = 0; rr < 16; ++rr) {
                const int row = m0 + wm0 + 32 * mi + mfma32_row(rr, lane);
                if (row < g.M && col < g.N) {
                    float v = acc[mi][ni][rr];
                    if (direct) {
                        v += bv;
                        if (g.round_f16) v = round_h(v);
                        if (g.accumulate) v += o[(size_t)row * g.N + col];
                    }
                    o[(size_t)row * g.N + col] = v;
                }
            }
        }
#endif
}

template <int BM, int BN, int WGM, int WGN, bool BRC = false, int NBUF = 2>
inline void hgemm_launch(const HgArgs& a0, hipStream_t st) {
    using Cfg = HgCfg<BM, BN, WGM, WGN, BRC, NBUF>;
    HgArgs a = a0;
    a.tiles_m = (a.M + BM - 1) / BM;
    a.tiles_n = (a.N + BN - 1) / BN;
    static bool once = false;
    if (!once) {
        hipFuncSetAttribute((const void*)hgemm_kernel<BM, BN, WGM, WGN, BRC, NBUF>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)Cfg::LDS_BYTES);
        once = true;
    }
    const unsigned grid = (unsigned)((long long)a.tiles_m * a.tiles_n * a.splits);
    mg_launch(hgemm_kernel<BM, BN, WGM, WGN, BRC, NBUF>, dim3(grid), dim3(Cfg::NT), Cfg::LDS_BYTES, st, a);
}

// ---- A-stationary form for a SHORT reduction (the weight gradient of the small-spatial trunk: K = padded pixel count = 256 on
// the 2048-channel 4x8 layers, M = Co = 2048, N = 9 Ci = 18432) ----
// hgemm_kernel spends that shape waiting: a workgroup's four 64-deep chunks arrive one latency after the other with 32 KiB in
// flight, and every 128 x 128 tile re-fetches its A rows (measured 57 us for 151 MB of output, 2.6 TB/s).  Here a workgroup owns
// one 128-row A panel for a GROUP of column tiles: the panel lives in VGPRs for the whole kernel (a wave's 32 rows x K halves =
// 16 K/64 registers), only B is streamed, as one continuous ring of [128][64] chunks that runs across tile boundaries, filled by
// LW dedicated loader waves.  The loaders are the only waves that count vmcnt -- the eight MFMA waves' global stores (32 per tile
// and lane) would otherwise sit in the same counter and every "chunk landed" wait would also drain the stores of the tile before.
// NBUF buffers of 16 KiB: NBUF - 1 chunks (112 KiB at 8) in flight per CU, one workgroup per CU.
// `flag` (optional): set to 1.0f when a result is not a finite float16 (|v| >= 65520 rounds to inf) -- the GradScaler's inf / nan
// check done on the accumulators, which saves a second pass over the largest gradients of the model.  The results are an autocast
// layer's weight gradient: float16 values in the reference (the cast's backward widens them into the float32 .grad), so float16
// is both the overflow criterion and -- H16 instances -- the storage format (round 6: half the output stream, which is what
// bounds this kernel, and 2 bytes fewer per parameter in the Adam pass).
struct HgAsArgs {
    const void* A;           // float16 [M][K]   (lda = K)
    const void* B;           // float16 [N][K]   (ldb = K)
    void* C;                 // [M][N] float32, or float16 (H16 instances)
    float* flag;
    int M, N, groups, accumulate;
};
constexpr int HG_AS_NBUF = 8, HG_AS_LW = 2;

__device__ __forceinline__ float hg_swap_adjacent(float v) {      // the value of lane ^ 1 (DPP quad_perm [1, 0, 3, 2])
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0xB1, 0xF, 0xF, true));
}
template <int KC, bool ACC, bool H16 = false>  // K = 64 KC; ACC: C += result; H16: C is float16
__global__ __launch_bounds__(64 * (8 + HG_AS_LW)) void hgemm_as_kernel(HgAsArgs g) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int NBUF = HG_AS_NBUF, LW = HG_AS_LW, AHEAD = NBUF - 1, BSZ = 128 * 32, K = 64 * KC;
    constexpr int PPW = 16 / LW;                            // 1 KiB pieces per loader wave and chunk
    static_assert((AHEAD - 1) * PPW <= 63, "vmcnt is a 6-bit counter");
    extern __shared__ __attribute__((aligned(1024))) float hg_as_smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tiles_m = g.M / 128, tiles_n = g.N / 128;
    const int L = xcd_remap(blockIdx.x, tiles_m * g.groups);
    // consecutive L = the m panels of one column group: an XCD streams that group's B rows through its L2 once for all of them
    const int grp = L / tiles_m, tm = L - grp * tiles_m;
    const int t0 = (int)((long long)grp * tiles_n / g.groups), t1 = (int)((long long)(grp + 1) * tiles_n / g.groups);
    const int Q = (t1 - t0) * KC;                           // chunks of this workgroup's stream
    const int m0 = tm * 128;

    if (wave >= 8) {
        // ---- loader waves ----
        const int lw = wave - 8;
        const unsigned long long a = (unsigned long long)g.B;
        dg_v4i rb;
        rb[0] = (int)(unsigned)a;
        rb[1] = (int)((unsigned)(a >> 32) & 0xffffu);
        rb[2] = (int)((unsigned)g.N * (unsigned)K * 2u);
        rb[3] = 0x00020000;
        unsigned vb[PPW];
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            const int row = 8 * (lw * PPW + i) + (lane >> 3), q = (lane & 7) ^ ((row >> 1) & 7);
            vb[i] = (unsigned)row * (unsigned)(K * 2) + 16u * q;
        }
        const unsigned lds0 = (unsigned)(size_t)(dg_lds_ptr)hg_as_smem + (unsigned)(lw * PPW) * 1024u;
        auto issue = [&](int q, int buf) {
            const int t = q / KC, c = q - t * KC;
            const unsigned soff = ((unsigned)(t0 + t) * 128u * (unsigned)K + (unsigned)c * 64u) * 2u;
            const unsigned lb = lds0 + (unsigned)buf * (unsigned)(BSZ * 4);
#pragma unroll
            for (int i = 0; i < PPW; ++i) dg_dma16(vb[i], rb, lb + 1024u * i, soff);
        };
#pragma unroll
        for (int i = 0; i < AHEAD; ++i)
            if (i < Q) issue(i, i);
        int nxt = AHEAD % NBUF;
        for (int q = 0; q < Q; ++q) {
            if (q + AHEAD > Q) dg_wait_vmcnt<0>();
            else dg_wait_vmcnt<(AHEAD - 1) * PPW>();
            __builtin_amdgcn_s_barrier();                   // chunk q is in LDS; everybody is done with chunk q - 1
            __builtin_amdgcn_sched_barrier(0);
            if (q + AHEAD < Q) issue(q + AHEAD, nxt);
            nxt = nxt + 1 == NBUF ? 0 : nxt + 1;
        }
        return;
    }

    // ---- MFMA waves: 4 (rows) x 2 (columns), 32 x 64 results each ----
    const int wm = wave >> 1, wn0 = (wave & 1) * 64;
    const int r = lane & 31, kh = lane >> 5;
    f16x8 a[4 * KC];
    {
        const _Float16* ap = (const _Float16*)g.A + (size_t)(m0 + 32 * wm + r) * K + 8 * kh;
#pragma unroll
        for (int s = 0; s < 4 * KC; ++s) a[s] = *reinterpret_cast<const f16x8*>(ap + 16 * s);
        // the panel has landed before the loop starts: otherwise the compiler waits for it at its first use INSIDE the loop, with a
        // count that also drains the stores of the tile before on every later trip
        __builtin_amdgcn_s_waitcnt(0x0F70);                 // vmcnt(0)
    }
    f32x16 acc[2] = {f32x16{0}, f32x16{0}};
    bool bad = false;
    int cur = 0, c = 0, t = t0;
    for (int q = 0; q < Q; ++q) {
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        const float* Bs = hg_as_smem + cur * BSZ;
        f16x8 b[2][2];
        auto fetch = [&](int s, int buf) {
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) b[buf][ni] = cd_frag_kc(Bs, wn0 + 32 * ni + r, s, kh);
        };
        fetch(0, 0);
        // the A fragment index depends on the chunk within the tile: a switch over c keeps `a` in registers
#pragma unroll
        for (int cc = 0; cc < KC; ++cc) {
            if (c == cc) {
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    if (s + 1 < 4) fetch(s + 1, (s + 1) & 1);
#pragma unroll
                    for (int ni = 0; ni < 2; ++ni) acc[ni] = mfma32x32x16h(a[4 * cc + s], b[s & 1][ni], acc[ni]);
                }
            }
        }
        cur = cur + 1 == NBUF ? 0 : cur + 1;
        if (++c == KC) {
            // tile finished: store it (plain dword stores, 2 x 128 bytes per instruction) and start the next one
            c = 0;
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) {
                if constexpr (H16) {
                    // float16 store: a lane holds one column and 16 rows, so adjacent lanes trade one value per row pair and every
                    // lane stores 4 bytes -- even lanes row rr of columns (r, r + 1), odd lanes row rr + 1 of (r - 1, r): 64-byte
                    // runs, half the store instructions of the float32 form
                    _Float16* o = (_Float16*)g.C + (size_t)(m0 + 32 * wm) * g.N + (size_t)t * 128 + wn0 + 32 * ni + (r & ~1);
                    const bool odd = lane & 1;
#pragma unroll
                    for (int rr = 0; rr < 16; rr += 2) {
                        const float v0 = acc[ni][rr], v1 = acc[ni][rr + 1];
                        const float got = hg_swap_adjacent(odd ? v0 : v1);
                        float lo = odd ? got : v0, hi = odd ? v1 : got;
                        unsigned* p = reinterpret_cast<unsigned*>(o + (size_t)mfma32_row(odd ? rr + 1 : rr, lane) * g.N);
                        if constexpr (ACC) {
                            const unsigned old = *p;
                            lo += (float)__builtin_bit_cast(_Float16, (unsigned short)(old & 0xffffu));
                            hi += (float)__builtin_bit_cast(_Float16, (unsigned short)(old >> 16));
                        }
                        bad |= !(fabsf(lo) < 65520.0f) | !(fabsf(hi) < 65520.0f);
                        *p = pack_h2(lo, hi);
                    }
                } else {
                    float* o = (float*)g.C + (size_t)(m0 + 32 * wm) * g.N + (size_t)t * 128 + wn0 + 32 * ni + r;
#pragma unroll
                    for (int rr = 0; rr < 16; ++rr) {
                        float v = acc[ni][rr];
                        float* p = o + (size_t)mfma32_row(rr, lane) * g.N;
                        if constexpr (ACC) v += *p;
                        bad |= !(fabsf(v) < 65520.0f);          // the float16 criterion (see HgAsArgs): the consumer rounds through float16
                        *p = v;
                    }
                }
                acc[ni] = f32x16{0};
            }
            ++t;
        }
    }
    if (g.flag && __any(bad) && lane == 0) *g.flag = 1.0f;
#endif
}

inline bool hgemm_as_ok(long long M, int N, int K) {
    return M % 128 == 0 && N % 128 == 0 && (K == 64 || K == 128 || K == 192 || K == 256) && M / 128 <= 256 &&
           (long long)N * K * 2 < (1ll << 32);
}
template <int KC, bool ACC, bool H16>
inline void hgemm_as_go2(const HgAsArgs& a, int grid, hipStream_t st) {
    constexpr size_t lds = (size_t)HG_AS_NBUF * 128 * 32 * 4;
    static bool once = false;
    if (!once) {
        hipFuncSetAttribute((const void*)hgemm_as_kernel<KC, ACC, H16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        once = true;
    }
    mg_launch(hgemm_as_kernel<KC, ACC, H16>, dim3((unsigned)grid), dim3(64 * (8 + HG_AS_LW)), lds, st, a);
}
template <int KC>
inline void hgemm_as_go(const HgAsArgs& a, int grid, hipStream_t st, bool h16) {
    if (h16) {
        if (a.accumulate) hgemm_as_go2<KC, true, true>(a, grid, st);
        else hgemm_as_go2<KC, false, true>(a, grid, st);
    } else {
        if (a.accumulate) hgemm_as_go2<KC, true, false>(a, grid, st);
        else hgemm_as_go2<KC, false, false>(a, grid, st);
    }
}
// C: float32 [M][N], or float16 when c_half
inline void hgemm_as_launch(const void* A, const void* B, void* C, long long M, int N, int K, int accumulate, float* flag, hipStream_t st,
                            bool c_half = false) {
    HgAsArgs a{A, B, C, flag, (int)M, N, 1, accumulate};
    const int tiles_m = (int)(M / 128), tiles_n = N / 128;
    int groups = 256 / tiles_m;                              // one workgroup per CU ...
    if (groups > tiles_n) groups = tiles_n;
    while (groups > 1 && tiles_n / groups < 3) --groups;     // ... of at least three tiles: the panel load and the ring's fill are paid once per workgroup
    if (const char* f = getenv("MG_HGEMM_AS_GROUPS")) { const int v = atoi(f); if (v >= 1 && v <= tiles_n) groups = v; }
    a.groups = groups;
    switch (K / 64) {
        case 1: hgemm_as_go<1>(a, tiles_m * groups, st, c_half); break;
        case 2: hgemm_as_go<2>(a, tiles_m * groups, st, c_half); break;
        case 3: hgemm_as_go<3>(a, tiles_m * groups, st, c_half); break;
        default: hgemm_as_go<4>(a, tiles_m * groups, st, c_half); break;
    }
}

// ---- Loader waves and separate rings for A and B: the weight-STREAMING form (round 6) ----
// Forward / data gradient of the small-spatial trunk: M = 256 pixels, N = 2048, K = 18432 -- 75 MB of cold float16 weights against
// an activation matrix that lives in L2.  hgemm_as_kernel's scheme with both operands streamed: two loader waves fill a B ring of
// NBB = 6 buffers (5 chunks = 80 KiB of weights in flight per CU), two more an A ring of NA = 3, each loader counts only its own
// DMA's vmcnt, and the MFMA waves issue no memory instruction inside the loop.  AG: the A loaders GATHER the rows from the float16
// activation itself (source pixel per row and tap; zero padding = an out-of-range offset): no im2col matrix exists -- that launch
// and its 9x traffic are what this kernel removes from the step.
// What it does NOT change is the GEMM's own time (scripts/ubench/hgemm_bench, profiles/r06_hgemm_sa_ubench.log: 34.4 vs 34.9 us
// with the harness's reduce launch included, cold weights): ring depth 3 -> 6, MFMA wave tiles 32x64 / 64x64 / 64x32 / 32x128 and
// even weights stored as contiguous 16 KiB blocks all land within 3 % -- the shape is bound by neither HBM latency, LDS reads nor
// DRAM locality but by the matrix pipe at the clock this part holds under dense float16 MFMAs (HISTORY.md section 3 "Round 5":
// 1.1-1.25 PF = 15.5-17.6 us for these 19.3 GFLOP) plus the fill / drain of 256 single-round workgroups.
// Same operand values and the same chunk order per accumulator as hgemm_kernel: bit-identical results.
constexpr int HG_SA_NA = 3, HG_SA_NBB = 6, HG_SA_LW = 2;       // buffers of the A / B rings; loader waves PER ring

// WGM x WGN MFMA waves over the 128 x 128 tile: 4 x 2 (32 x 64 results per wave: 12 ds_read_b128 per 8 MFMAs) or 2 x 2 (64 x 64: 16
// reads per 16 MFMAs -- a third fewer LDS bytes per MFMA, which is what bounds the 8-wave form on this part: 768 LDS cycles against 512
// MFMA cycles per chunk).
template <bool BRC, bool AG, int WGM = 4, int WGN = 2>
__global__ __launch_bounds__(64 * (WGM * WGN + 2 * HG_SA_LW)) void hgemm_sa_kernel(HgArgs g) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int NA = HG_SA_NA, NBB = HG_SA_NBB, LW = HG_SA_LW, SZ = 128 * 32;      // 32-bit words per 16 KiB buffer
    constexpr int NMW = WGM * WGN, MB = 128 / WGM / 32, NB = 128 / WGN / 32;
    constexpr int PPW = 16 / LW;                            // 1 KiB pieces per loader wave and chunk
    static_assert((NBB - 2) * PPW <= 63 && (NA - 2) * PPW <= 63, "vmcnt is a 6-bit counter");
    extern __shared__ __attribute__((aligned(1024))) float hg_sa_smem[];
    float* const As0 = hg_sa_smem;
    float* const Bs0 = hg_sa_smem + NA * SZ;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tiles = g.tiles_m * g.tiles_n;
    const int L = xcd_remap(blockIdx.x, tiles * g.splits);
    const int sp = L / tiles, rem = L - sp * tiles;
    const int tn = rem / g.tiles_m, tm = rem - tn * g.tiles_m;
    const int m0 = tm * 128, n0 = tn * 128;
    const int total_chunks = g.K / HG_BK;
    const int c_begin = sp * g.cps, c_end = min(total_chunks, c_begin + g.cps);
    const int Q = c_end - c_begin;
    constexpr unsigned OOB = 0x80000000u;

    if (wave >= NMW + LW) {
        // ---- A loader waves ----
        const int lw = wave - NMW - LW;
        const unsigned long long a = (unsigned long long)g.A;
        const unsigned a_bytes = AG ? (unsigned)(g.M / (g.gOH * g.gOW)) * (unsigned)(g.gH * g.gW) * (unsigned)g.lda * 2u      // x16 [B*H*W][Ci]
                                    : (unsigned)g.M * (unsigned)g.lda * 2u;
        dg_v4i ra;
        ra[0] = (int)(unsigned)a;
        ra[1] = (int)((unsigned)(a >> 32) & 0xffffu);
        ra[2] = (int)a_bytes;
        ra[3] = 0x00020000;
        unsigned va[PPW];
        int gb[PPW], goy[PPW], gox[PPW];
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            const int row = 8 * (lw * PPW + i) + (lane >> 3), q = (lane & 7) ^ ((row >> 1) & 7);
            const int m = m0 + row;
            va[i] = m < g.M ? (unsigned)m * (unsigned)g.lda * 2u + 16u * q : OOB;      // rows past M read zeros (never stored)
            if (AG) {
                const int hw = g.gOH * g.gOW, mm = min(m, g.M - 1);
                gb[i] = mm / hw;
                const int rm = mm - gb[i] * hw;
                goy[i] = rm / g.gOW;
                gox[i] = rm - goy[i] * g.gOW;
            }
        }
        int cur_tap = -1;
        const unsigned lds0 = (unsigned)(size_t)(dg_lds_ptr)As0 + (unsigned)(lw * PPW) * 1024u;
        auto issue = [&](int q, int buf) {
            const int c = c_begin + q;
            unsigned soff = (unsigned)c * (HG_BK * 2u);
            if constexpr (AG) {
                const int tap = c / g.a_cpt, cc = c - tap * g.a_cpt;
                soff = (unsigned)cc * (HG_BK * 2u);
                if (tap != cur_tap) {                           // the source pixel of every row of this lane for the new tap
                    cur_tap = tap;
                    const int ky = tap / g.gKW, kx = tap - ky * g.gKW;
#pragma unroll
                    for (int i = 0; i < PPW; ++i) {
                        const int row = 8 * (lw * PPW + i) + (lane >> 3), qs = (lane & 7) ^ ((row >> 1) & 7);
                        int iy = goy[i] * g.gs - g.gp + ky, ix = gox[i] * g.gs - g.gp + kx;
                        bool ok = m0 + row < g.M;
                        if (g.greflect) {
                            iy = iy < 0 ? -iy : (iy >= g.gH ? 2 * (g.gH - 1) - iy : iy);
                            ix = ix < 0 ? -ix : (ix >= g.gW ? 2 * (g.gW - 1) - ix : ix);
                        } else {
                            ok = ok && iy >= 0 && iy < g.gH && ix >= 0 && ix < g.gW;
                        }
                        va[i] = ok ? (unsigned)((gb[i] * g.gH + iy) * g.gW + ix) * (unsigned)g.lda * 2u + 16u * qs : OOB;
                    }
                }
            }
            const unsigned la = lds0 + (unsigned)buf * (unsigned)(SZ * 4);
#pragma unroll
            for (int i = 0; i < PPW; ++i) dg_dma16(va[i], ra, la + 1024u * i, soff);
        };
#pragma unroll
        for (int i = 0; i < NA - 1; ++i)
            if (i < Q) issue(i, i);
        int nxt = (NA - 1) % NA;
        for (int q = 0; q < Q; ++q) {
            if (q + (NA - 1) > Q) dg_wait_vmcnt<0>();
            else dg_wait_vmcnt<(NA - 2) * PPW>();
            __builtin_amdgcn_s_barrier();                   // chunk q is in LDS; everybody is done with chunk q - 1
            __builtin_amdgcn_sched_barrier(0);
            if (q + (NA - 1) < Q) issue(q + (NA - 1), nxt);
            nxt = nxt + 1 == NA ? 0 : nxt + 1;
        }
        return;
    }
    if (wave >= NMW) {
        // ---- B loader waves ----
        const int lw = wave - NMW;
        const unsigned long long a = (unsigned long long)g.B;
        dg_v4i rb;
        rb[0] = (int)(unsigned)a;
        rb[1] = (int)((unsigned)(a >> 32) & 0xffffu);
        rb[2] = BRC ? 0x7fffffff : (int)((unsigned)g.N * (unsigned)g.ldb * 2u);
        rb[3] = 0x00020000;
        unsigned vb[PPW];
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            const int piece = lw * PPW + i;
            if constexpr (BRC) {
                const int k = piece * 4 + (lane >> 4), c8 = (lane & 15) ^ cd_rc_swz<128>(k);      // a piece = 4 whole k rows of 128 columns
                vb[i] = ((unsigned)k * (unsigned)g.ldb + (unsigned)min(n0 + 8 * c8, g.N - 8)) * 2u;
            } else {
                const int row = 8 * piece + (lane >> 3), q = (lane & 7) ^ ((row >> 1) & 7);
                vb[i] = (unsigned)min(n0 + row, g.N - 1) * (unsigned)g.ldb * 2u + 16u * q;
            }
        }
        const unsigned lds0 = (unsigned)(size_t)(dg_lds_ptr)Bs0 + (unsigned)(lw * PPW) * 1024u;
        auto issue = [&](int q, int buf) {
            const int c = c_begin + q;
            unsigned soff = (unsigned)c * (HG_BK * 2u);
            if (BRC) {
                const int tap = c / g.b_cpt, cc = c - tap * g.b_cpt;
                soff = ((unsigned)tap * (unsigned)g.b_tap_stride + (unsigned)cc * (unsigned)HG_BK * (unsigned)g.ldb) * 2u;
            }
            const unsigned lb = lds0 + (unsigned)buf * (unsigned)(SZ * 4);
#pragma unroll
            for (int i = 0; i < PPW; ++i) dg_dma16(vb[i], rb, lb + 1024u * i, soff);
        };
#pragma unroll
        for (int i = 0; i < NBB - 1; ++i)
            if (i < Q) issue(i, i);
        int nxt = (NBB - 1) % NBB;
        for (int q = 0; q < Q; ++q) {
            if (q + (NBB - 1) > Q) dg_wait_vmcnt<0>();
            else dg_wait_vmcnt<(NBB - 2) * PPW>();
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            if (q + (NBB - 1) < Q) issue(q + (NBB - 1), nxt);
            nxt = nxt + 1 == NBB ? 0 : nxt + 1;
        }
        return;
    }

    // ---- MFMA waves: WGM (rows) x WGN (columns); no memory instruction inside the loop ----
    const int wm0 = (wave / WGN) * (128 / WGM), wn0 = (wave % WGN) * (128 / WGN);
    const int r = lane & 31, kh = lane >> 5;
    f32x16 acc[MB][NB];
#pragma unroll
    for (int mi = 0; mi < MB; ++mi)
#pragma unroll
        for (int ni = 0; ni < NB; ++ni) acc[mi][ni] = f32x16{0};
    int ca = 0, cb = 0;
    for (int q = 0; q < Q; ++q) {
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        const float* As = As0 + ca * SZ;
        const float* Bs = Bs0 + cb * SZ;
        f16x8 a[2][MB], b[2][NB];
        auto fetch = [&](int s, int buf) {
#pragma unroll
            for (int mi = 0; mi < MB; ++mi) a[buf][mi] = cd_frag_kc(As, wm0 + 32 * mi + r, s, kh);
#pragma unroll
            for (int ni = 0; ni < NB; ++ni) {
                if constexpr (BRC) b[buf][ni] = cd_frag_rc<128>(Bs, wn0 + 32 * ni, s, lane);
                else b[buf][ni] = cd_frag_kc(Bs, wn0 + 32 * ni + r, s, kh);
            }
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
        ca = ca + 1 == NA ? 0 : ca + 1;
        cb = cb + 1 == NBB ? 0 : cb + 1;
    }
    float* o = g.part ? g.part + (size_t)sp * ((size_t)g.M * g.N) : g.C;
    const bool direct = g.part == nullptr;
#pragma unroll
    for (int mi = 0; mi < MB; ++mi)
#pragma unroll
        for (int ni = 0; ni < NB; ++ni) {
            const int col = n0 + wn0 + 32 * ni + (lane & 31);
            const float bv = (direct && g.bias && col < g.N) ? g.bias[col] : 0.0f;
#pragma unroll
            for (int rr = 0; rr < 16; ++rr) {
                const int row = m0 + wm0 + 32 * mi + mfma32_row(rr, lane);
                if (row < g.M && col < g.N) {
                    float v = acc[mi][ni][rr];
                    if (direct) {
                        v += bv;
                        if (g.round_f16) v = round_h(v);
                        if (g.accumulate) v += o[(size_t)row * g.N + col];
                    }
                    o[(size_t)row * g.N + col] = v;
                }
            }
        }
#endif
}

inline bool hgemm_sa_ok(const HgArgs& a) { return a.N % 128 == 0 && a.K % HG_BK == 0; }
template <bool BRC, bool AG, int WGM = 4, int WGN = 2>
inline void hgemm_sa_launch(const HgArgs& a0, hipStream_t st) {
    HgArgs a = a0;
    a.tiles_m = (a.M + 127) / 128;
    a.tiles_n = a.N / 128;
    constexpr size_t lds = (size_t)(HG_SA_NA + HG_SA_NBB) * 128 * 32 * 4;
    static bool once = false;
    if (!once) {
        hipFuncSetAttribute((const void*)hgemm_sa_kernel<BRC, AG, WGM, WGN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        once = true;
    }
    const unsigned grid = (unsigned)((long long)a.tiles_m * a.tiles_n * a.splits);
    mg_launch(hgemm_sa_kernel<BRC, AG, WGM, WGN>, dim3(grid), dim3(64 * (WGM * WGN + 2 * HG_SA_LW)), lds, st, a);
}
