// The f16 MFMA main loop, second generation: a 256 x BN x 64 workgroup tile run as a ping-pong between two groups of four waves.
//
// dense_gemm_h.h's loop (one barrier per 64-deep chunk, all waves in lock step) leaves the matrix pipe idle whenever the waves of a
// SIMD read fragments or wait for a chunk at the same time: 0.63-0.99 PFLOP/s on 4096^3.  Here the eight waves of a workgroup are two
// groups (wave >> 2) that run ONE BARRIER APART: while group 0 multiplies (8 MFMAs = 256 cycles on its SIMD), group 1 -- the other
// wave of every SIMD -- reads its next fragments from LDS and issues the next LDS-DMA pieces, and at the next barrier they swap.
// A K-tile (64 halves) is four such phases, one 64 x 32 quadrant of the wave's 128 x 64 (BN = 256) results each:
//     q0  read A-top (8 x ds_read_b128) + B-left (4)   -> (top, left)
//     q1  read B-right (4)                             -> (top, right)
//     q2  read A-bottom (8)                            -> (bottom, right)
//     q3  --                                           -> (bottom, left)       (B-left stays in registers)
// Operands are staged by LDS-DMA in HALF-tiles of 16 KiB (A-top / A-bottom = the 64-row halves of both groups' 128 rows, B-left /
// B-right = the 32-column halves of the four column blocks), 2 pieces of 1 KiB per wave, one half-tile per phase, into two K-tile
// buffers of 64 KiB.  A half-tile buffer is re-filled two phases after its last fragment read and read five phases after its issue:
//     phase 4u + 0 issues B-right(u + 1), + 1: A-bottom(u + 1), + 2: A-top(u + 2), + 3: B-left(u + 2)
// so four half-tiles (64 KiB) are always in flight per CU and the only wait is a counted `s_waitcnt vmcnt(8)` per phase (the
// half-tile read in phase p + 1 is retired in phase p, one barrier before the first reader: the other group's pieces are waited for
// one barrier later than ours).  LDS rows are 128 bytes, 16-byte slots XOR-swizzled on the DMA source side (slot ^ ((row >> 1) & 7)):
// every 16-lane group of a ds_read_b128 covers all 64 banks once.
//
// C[M][N] (float32) = A[M][K] (float16) * B[N][K]^T (float16); K % 64 == 0, K >= 128.  Included inside the anonymous namespace after
// dense_gemm.h / dense_gemm_h.h.
#pragma once
// EXPERIMENT (ubench only): the ping-pong form of the 256 x 256 x 64 loop, kept to measure slot balance (DBG 128: six fragment reads in every
// phase instead of 12 / 4 / 8 / 0 -- timing only, results are wrong)

struct HppArgs {
    const void* A;           // float16 [M][lda]
    const void* B;           // float16 [N][ldb]
    float* C;                // float32 [M][ldc]
    const float* bias;       // [N] or nullptr
    int M, N, K, lda, ldb, ldc;
    int tiles_m, tiles_n;
    int round_f16;           // round the result through float16 (autocast output)
    int accumulate;          // C += result
};

// LDS byte offsets inside a K-tile buffer
constexpr unsigned HPP_AT = 0, HPP_AB = 16384, HPP_BL = 32768, HPP_BR = 49152, HPP_BUF = 65536;

template <int N>
__device__ __forceinline__ void hpp_wait_raw() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int DBG>
struct hpp_wait_sel {
    template <int N>
    static __device__ __forceinline__ void go() { if (!(DBG & 8)) hpp_wait_raw<N>(); }
};
// DBG (ubench ablations only; results are wrong): 1 no DMA in the main loop, 2 no fragment reads, 4 no MFMAs, 8 no counted waits,
// 16 no priority flips
template <int DBG = 0>
__global__ __launch_bounds__(512) void hpp_kernel(HppArgs g) {
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(1024))) float hpp_smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;

    // tile order: an XCD's consecutive tiles form a 4 (m) x 8 (n) block -- 12 operand panels per K-tile for 32 workgroups
    const int tiles = g.tiles_m * g.tiles_n;
    const int L = xcd_remap(blockIdx.x, tiles);
    int tm, tn;
    {
        constexpr int GM = 4;
        if (g.tiles_m % GM == 0) {
            const int grp = L / (GM * g.tiles_n), r2 = L - grp * GM * g.tiles_n;
            tn = r2 / GM;
            tm = grp * GM + (r2 - tn * GM);
        } else {
            tn = L / g.tiles_m;
            tm = L - tn * g.tiles_m;
        }
    }
    const int m0 = tm * 256, n0 = tn * 256;
    const int T = g.K / 64;

    auto make_rsrc = [](const void* p, unsigned bytes) -> dg_v4i {
        const unsigned long long a = (unsigned long long)p;
        dg_v4i r;
        r[0] = (int)(unsigned)a;
        r[1] = (int)((unsigned)(a >> 32) & 0xffffu);
        r[2] = (int)bytes;
        r[3] = 0x00020000;
        return r;
    };
    // rows / columns past M / N read as zeros (the range check covers voffset + soffset)
    const dg_v4i ra = make_rsrc(g.A, (unsigned)g.M * (unsigned)g.lda * 2u);
    const dg_v4i rb = make_rsrc(g.B, (unsigned)g.N * (unsigned)g.ldb * 2u);

    // This wave's two pieces of a half-tile: LDS rows 16 wave .. + 15 of its 128.  A half-tile row rho = 64 wr' + i is tile row
    // 128 wr' + 64 half + i; B: rho = 32 wc' + j is tile column 64 wc' + 32 half + j.  The half goes into the scalar offset.
    unsigned va[2], vb[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int rho = 16 * wave + 8 * j + (lane >> 3);
        const unsigned q = (unsigned)((lane & 7) ^ ((rho >> 1) & 7));
        va[j] = (unsigned)(m0 + 128 * (rho >> 6) + (rho & 63)) * (unsigned)g.lda * 2u + 16u * q;
        vb[j] = (unsigned)(n0 + 64 * (rho >> 5) + (rho & 31)) * (unsigned)g.ldb * 2u + 16u * q;
    }
    const unsigned half_a = 64u * (unsigned)g.lda * 2u, half_b = 32u * (unsigned)g.ldb * 2u;
    const unsigned lds0 = (unsigned)(size_t)(dg_lds_ptr)hpp_smem + (unsigned)wave * 2048u;
    // kind: 0 A-top, 1 A-bottom, 2 B-left, 3 B-right of K-tile t
    auto issue = [&](int kind, int t) {
        if ((DBG & 1) && t >= 2) return;
        const unsigned dst = lds0 + (unsigned)(t & 1) * HPP_BUF + (unsigned)kind * 16384u;
        const unsigned koff = (unsigned)t * 128u;
        if (kind < 2) {
            const unsigned so = koff + (kind == 1 ? half_a : 0u);
            dg_dma16(va[0], ra, dst, so);
            dg_dma16(va[1], ra, dst + 1024u, so);
        } else {
            const unsigned so = koff + (kind == 3 ? half_b : 0u);
            dg_dma16(vb[0], rb, dst, so);
            dg_dma16(vb[1], rb, dst + 1024u, so);
        }
    };

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f32x16{0};
#define hpp_wait hpp_wait_sel<DBG>::template go

    const int r = lane & 31, kh = lane >> 5, x = (r >> 1) & 7;
    // fragment addresses: lane part per k-step s (the swizzled slot), the rest is wave-uniform
    unsigned fo[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) fo[s] = (unsigned)r * 128u + 16u * (unsigned)((2 * s + kh) ^ x);
    const unsigned a_w = (unsigned)wr * 8192u, b_w = (unsigned)wc * 4096u;
    const char* smem = reinterpret_cast<const char*>(hpp_smem);

    f16x8 a[2][4], bl[4], br[4];
    auto read_a = [&](unsigned base, int n = 8) {      // base: buffer + HPP_AT / HPP_AB
        if ((DBG & 2) && g.K > 0) return;
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int s = 0; s < 4; ++s) if (mi * 4 + s < n) a[mi][s] = *reinterpret_cast<const f16x8*>(smem + base + a_w + 4096u * mi + fo[s]);
    };
    auto read_b = [&](f16x8 (&b)[4], unsigned base, int n = 4) {
        if ((DBG & 2) && g.K > 0) return;
#pragma unroll
        for (int s = 0; s < 4; ++s) if (s < n) b[s] = *reinterpret_cast<const f16x8*>(smem + base + b_w + fo[s]);
    };
    auto mma = [&](int row2, int col, const f16x8 (&b)[4]) {
        if ((DBG & 4) && g.K > 0) return;
        if (!(DBG & 16)) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) {
                if constexpr (DBG & 32) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc[row2 + mi][col]) : "v"(a[mi][s]), "v"(b[s]));
                else acc[row2 + mi][col] = mfma32x32x16h(a[mi][s], b[s], acc[row2 + mi][col]);
            }
        if (!(DBG & 16)) __builtin_amdgcn_s_setprio(0);
    };

    // ---- prologue: the six half-tiles phases -6 .. -1 would have issued
    issue(0, 0); issue(2, 0); issue(3, 0); issue(1, 0); issue(0, 1); issue(2, 1);
    hpp_wait<8>();                                   // A-top(0), B-left(0) have landed
    __builtin_amdgcn_s_barrier();
    if (wr == 1) __builtin_amdgcn_s_barrier();      // group 1 runs one barrier behind
    __builtin_amdgcn_sched_barrier(0);

    // One K-tile = four phases.  ISS: how many of this K-tile's four issues exist (4: all, 2: only q0 / q1, 0: none);
    // W0..W3: the vmcnt each phase leaves outstanding (-1: no wait).
    unsigned ts[28];
#pragma unroll
    for (int i = 0; i < 28; ++i) ts[i] = 0;
#define HPP_TS(i) do { if constexpr (DBG & 64) { if (u == T / 2) ts[i] = (unsigned)__builtin_amdgcn_s_memtime(); } } while (0)
#define HPP_PHASE_END()                     \
    __builtin_amdgcn_sched_barrier(0);     \
    __builtin_amdgcn_s_barrier();          \
    __builtin_amdgcn_sched_barrier(0)
    auto ktile = [&](int u, auto iss_c, auto w0_c, auto w1_c, auto w3_c) {
        constexpr int ISS = decltype(iss_c)::value, W0 = decltype(w0_c)::value, W1 = decltype(w1_c)::value, W3 = decltype(w3_c)::value;
        const unsigned buf = (unsigned)(u & 1) * HPP_BUF;
        // q0
        HPP_TS(0);
        if (DBG & 128) { read_b(bl, buf + HPP_BL, 2); read_a(buf + HPP_AT, 4); } else { read_b(bl, buf + HPP_BL); read_a(buf + HPP_AT); }
        __builtin_amdgcn_sched_barrier(0);
        HPP_TS(1);
        if (ISS >= 2) issue(3, u + 1);
        HPP_TS(2);
        hpp_wait<W0>();
        HPP_TS(3);
        HPP_PHASE_END();
        HPP_TS(4);
        mma(0, 0, bl);
        HPP_TS(5);
        HPP_PHASE_END();
        HPP_TS(6);
        // q1
        if (DBG & 128) { read_b(br, buf + HPP_BR, 2); read_a(buf + HPP_AT, 4); } else read_b(br, buf + HPP_BR);
        __builtin_amdgcn_sched_barrier(0);
        HPP_TS(7);
        if (ISS >= 2) issue(1, u + 1);
        HPP_TS(8);
        hpp_wait<W1>();
        HPP_TS(9);
        HPP_PHASE_END();
        HPP_TS(10);
        mma(0, 1, br);
        HPP_TS(11);
        HPP_PHASE_END();
        HPP_TS(12);
        // q2 (nothing is read in q3, so nothing to retire here)
        if (DBG & 128) { read_b(bl, buf + HPP_BL, 2); read_a(buf + HPP_AB, 4); } else read_a(buf + HPP_AB);
        __builtin_amdgcn_sched_barrier(0);
        HPP_TS(13);
        if (ISS >= 4) issue(0, u + 2);
        HPP_TS(14);
        HPP_PHASE_END();
        HPP_TS(15);
        mma(2, 1, br);
        HPP_TS(16);
        HPP_PHASE_END();
        HPP_TS(17);
        // q3
        if (DBG & 128) { read_b(br, buf + HPP_BR, 2); read_a(buf + HPP_AB, 4); }
        if (ISS >= 4) issue(2, u + 2);
        HPP_TS(18);
        if (W3 >= 0) hpp_wait<(W3 >= 0 ? W3 : 0)>();
        HPP_TS(19);
        HPP_PHASE_END();
        HPP_TS(20);
        mma(2, 0, bl);
        HPP_TS(21);
        HPP_PHASE_END();
        HPP_TS(22);
    };
    using I0 = std::integral_constant<int, 0>;
    using I2 = std::integral_constant<int, 2>;
    using I4 = std::integral_constant<int, 4>;
    using I8 = std::integral_constant<int, 8>;
    for (int u = 0; u < T - 2; ++u) ktile(u, I4{}, I8{}, I8{}, I8{});
    // K-tile T - 2: B-right / A-bottom of the last tile are still to issue; at q3 A-top / B-left of the last tile must have
    // landed with only those two behind them
    ktile(T - 2, I2{}, I8{}, I8{}, I4{});
    // K-tile T - 1: q0 retires B-right (A-bottom behind it), q1 retires A-bottom
    ktile(T - 1, I0{}, I2{}, I0{}, std::integral_constant<int, -1>{});
#undef HPP_PHASE_END
#undef HPP_TS
    if constexpr (DBG & 64) {
        if (blockIdx.x == 8 && lane == 0 && g.bias) {
            unsigned* o = (unsigned*)g.bias + wave * 32;
#pragma unroll
            for (int i = 0; i < 28; ++i) o[i] = ts[i];
        }
    }
#undef hpp_wait
    if (wr == 0) __builtin_amdgcn_s_barrier();      // pairs with group 1's last barrier

    // ---- epilogue
    if constexpr (DBG & 32) asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" ::: "memory");     // asm MFMAs: the compiler does not know their latency
    const bool rnd = g.round_f16 != 0;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = n0 + 64 * wc + 32 * j + r;
            const float bv = (!(DBG & 64) && g.bias && col < g.N) ? g.bias[col] : 0.0f;
#pragma unroll
            for (int rr = 0; rr < 16; ++rr) {
                const int row = m0 + 128 * wr + 32 * i + mfma32_row(rr, lane);
                if (row < g.M && col < g.N) {
                    float v = acc[i][j][rr] + bv;
                    if (rnd) v = round_h(v);
                    float* p = g.C + (size_t)row * g.ldc + col;
                    if (g.accumulate) v += *p;
                    *p = v;
                }
            }
        }
#endif
}

inline bool hpp_ok(long long M, int N, int K) { return K % 64 == 0 && K >= 128 && M > 0 && N > 0; }

template <int DBG = 0>
inline void hpp_launch(const HppArgs& a0, hipStream_t st) {
    HppArgs a = a0;
    a.tiles_m = (a.M + 255) / 256;
    a.tiles_n = (a.N + 255) / 256;
    static bool once = false;
    if (!once) {
        hipFuncSetAttribute((const void*)hpp_kernel<DBG>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(2 * HPP_BUF));
        once = true;
    }
    mg_launch(hpp_kernel<DBG>, dim3((unsigned)(a.tiles_m * a.tiles_n)), dim3(512), (size_t)(2 * HPP_BUF), st, a);
}
