// The f16 MFMA main loop, second generation: a 256 x 256 x 64 workgroup tile run as a BALANCED ping-pong of two groups of four waves.
//
// dense_gemm_h.h's loop (one barrier per 64-deep chunk: wait for the whole chunk, then every wave reads fragments, then multiplies)
// tops out at 0.63-0.99 PFLOP/s on 4096^3.  Here the eight waves (2 x 4, 128 x 64 results each) are two groups (wave >> 2) that run
// ONE BARRIER APART: while group 0 multiplies (8 MFMAs back to back = 256 cycles on its SIMD), group 1 -- the other wave of every
// SIMD -- reads fragments and issues LDS-DMA, and at the next barrier they swap.  What makes it work is that the two kinds of slot
// take the same time.  Measured (s_memtime stamps, scripts/ubench/h8pp.h): a wave issues one ds_read_b128 per ~20 cycles and one
// LDS-DMA piece per ~65, so a load slot has room for SIX reads and TWO pieces against the partner's 256 MFMA cycles:
//   * a K-tile (64 halves) is four phases, one 64 x 32 quadrant of the wave's results each: (top, left), (top, right), (bottom, right),
//     (bottom, left).  Their fragments -- A-top 8 reads, B-left 4, B-right 4, A-bottom 8 -- are read six per load slot:
//         slot 0: A-top[k-steps 1..3]            slot 1: B-right + A-bottom[k-step 0]
//         slot 2: A-bottom[k-steps 1..3]         slot 3: B-left and A-top[k-step 0] of the NEXT K-tile
//     (the first version read 12 / 4 / 8 / 0 and ran at 1.04-1.10 PF: the 12-read slot took 370+ cycles);
//   * operands are staged in HALF-tiles of 16 KiB (A-top / A-bottom = the 64-row halves of both groups' 128 rows, B-left / B-right
//     = the 32-column halves of the four wave columns), 2 pieces of 1 KiB per wave and phase, into 8 slots (two per kind):
//     phase 4t + 0 issues A-bottom(t + 1), + 1: B-left(t + 2), + 2: A-top(t + 2), + 3: B-right(t + 2) -- every slot is re-filled two
//     phases after its last fragment read and first read four to five phases after its issue, four half-tiles (64 KiB) are in flight
//     per CU, and the only wait is a counted `s_waitcnt vmcnt(8)` (a half-tile is retired one phase before its first reader: the
//     other group's pieces are waited for one barrier later than ours);
//   * LDS rows are 128 bytes, 16-byte slots XOR-swizzled on the DMA source side (slot ^ ((row >> 1) & 7)): every 16-lane group of
//     a ds_read_b128 covers all 64 banks once.
//
// C[M][N] (float32) = A[M][K] (float16) * B[N][K]^T (float16); K % 64 == 0 (an odd K / 64 runs one zero K-tile more).  A ubench-only
// kernel (scripts/ubench/hgemm_bench.hip): no layer of this model has the >= 256 tiles of 256 x 256 it
// needs.  Included inside the anonymous namespace after
// dense_gemm.h / dense_gemm_h.h.
#pragma once
#include <type_traits>

struct Hg8Args {
    const void* A;           // float16 [M][lda]
    const void* B;           // float16 [N][ldb]
    float* C;                // float32 [M][ldc]
    const float* bias;       // [N] or nullptr
    int M, N, K, lda, ldb, ldc;
    int tiles_m, tiles_n;
    int round_f16;           // round the result through float16 (autocast output)
    int accumulate;          // C += result
};

// half-tile kinds: slot of (K-tile t, kind) = 4 (t & 1) + kind
enum { H8_BL = 0, H8_AT = 1, H8_BR = 2, H8_AB = 3 };
constexpr unsigned H8_SLOT = 16384, H8_LDS = 8 * H8_SLOT;

template <int N>
__device__ __forceinline__ void h8_wait_raw() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// DBG (ubench ablations only; results are wrong): 1 no DMA in the main loop, 2 no fragment reads, 4 no MFMAs, 8 no counted waits
template <int DBG = 0>
__global__ __launch_bounds__(512) void hgemm8_kernel(Hg8Args g) {
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(1024))) float h8_smem[];
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
    const unsigned lds0 = (unsigned)(size_t)(dg_lds_ptr)h8_smem + (unsigned)wave * 2048u;
    // piece i (0 / 1) of half-tile (K-tile t, kind); par = t & 1
    auto issue = [&](int t, int par, int kind, int i) {
        if ((DBG & 1) && t >= 2) return;
        const unsigned dst = lds0 + (unsigned)(4 * par + kind) * H8_SLOT + 1024u * (unsigned)i;
        const unsigned koff = t < T ? (unsigned)t * 128u : 0x80000000u;      // the padding K-tile of an odd K / 64 reads zeros
        if (kind == H8_AT || kind == H8_AB) dg_dma16(va[i], ra, dst, (unsigned)__builtin_amdgcn_readfirstlane((int)(koff + (kind == H8_AB ? half_a : 0u))));
        else dg_dma16(vb[i], rb, dst, (unsigned)__builtin_amdgcn_readfirstlane((int)(koff + (kind == H8_BR ? half_b : 0u))));
    };

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f32x16{0};

    const int r = lane & 31, kh = lane >> 5, x = (r >> 1) & 7;
    // fragment addresses: per k-step s one register for A and one for B (lane part = the swizzled slot, plus the wave's rows); the
    // K-tile parity (64 KiB) and the half-tile kind are immediate offsets
    unsigned fa[4], fb[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const unsigned fo = (unsigned)r * 128u + 16u * (unsigned)((2 * s + kh) ^ x);
        fa[s] = fo + (unsigned)wr * 8192u;
        fb[s] = fo + (unsigned)wc * 4096u;
    }
    // (dynamic LDS starts at address 0 of the workgroup's allocation -- the kernel has no static __shared__ -- so the addresses are
    // used as they are: going through the h8_smem pointer costs a v_add of zero per read, in the slot that has no time to spare)
    typedef f16x8 __attribute__((address_space(3))) * h8_frag_ptr;
    auto frag_a = [&](int par, int kind, int mi, int s) -> f16x8 {
        return *(h8_frag_ptr)(size_t)(fa[s] + (unsigned)(4 * par + kind) * H8_SLOT + 4096u * mi);
    };
    auto frag_b = [&](int par, int kind, int s) -> f16x8 { return *(h8_frag_ptr)(size_t)(fb[s] + (unsigned)(4 * par + kind) * H8_SLOT); };
    // (the second K-tile parity is 64 KiB away, beyond a ds_read's 16-bit offset: the compiler keeps a second set of eight address
    // registers for it.  Toggling one set instead costs eight VALU operations per K-tile, and a VALU operation between two MFMAs of a
    // back-to-back stream costs ~40 cycles of the matrix pipe: measured 1.25 -> 1.13 PF)

    // Registers: a[mi][s] holds A-top, then (k-steps 1..3) A-bottom; y0[mi] the k-step 0 fragments of A-bottom, which are read while
    // A-top is still in use.  B-left of K-tile t lives in b[t & 1], B-right in b[(t & 1) ^ 1]: B-left(t + 1) is read into the registers
    // B-right(t) has just left.
    f16x8 a[2][4], y0[2], b[2][4];

#define H8_PHASE_END()                     \
    __builtin_amdgcn_sched_barrier(0);     \
    __builtin_amdgcn_s_barrier();          \
    __builtin_amdgcn_sched_barrier(0)
    const bool no_lds = (DBG & 2) && g.K > 0, no_mma = (DBG & 4) && g.K > 0;
    // One K-tile of parity PAR.  MODE 0: steady state; 1: K-tile T - 2 (only A-bottom(T - 1) is still to issue); 2: K-tile T - 1.
    auto ktile = [&](int t, auto par_c, auto mode_c) {
        constexpr int PAR = decltype(par_c)::value, MODE = decltype(mode_c)::value;
        // ---- phase 0: (top, left)
        if (!no_lds) {
#pragma unroll
            for (int s = 1; s < 4; ++s)
#pragma unroll
                for (int mi = 0; mi < 2; ++mi) a[mi][s] = frag_a(PAR, H8_AT, mi, s);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (MODE <= 1) { issue(t + 1, PAR ^ 1, H8_AB, 0); issue(t + 1, PAR ^ 1, H8_AB, 1); }
        if (!(DBG & 8)) h8_wait_raw<(MODE == 2 ? 0 : 8)>();           // B-right(t), A-bottom(t)
        H8_PHASE_END();
        if (!no_mma) {
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int mi = 0; mi < 2; ++mi) acc[mi][0] = mfma32x32x16h(a[mi][s], b[PAR][s], acc[mi][0]);
        }
        H8_PHASE_END();
        // ---- phase 1: (top, right)
        if (!no_lds) {
#pragma unroll
            for (int s = 0; s < 4; ++s) b[PAR ^ 1][s] = frag_b(PAR, H8_BR, s);
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) y0[mi] = frag_a(PAR, H8_AB, mi, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (MODE == 0) { issue(t + 2, PAR, H8_BL, 0); issue(t + 2, PAR, H8_BL, 1); }
        H8_PHASE_END();
        if (!no_mma) {
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int mi = 0; mi < 2; ++mi) acc[mi][1] = mfma32x32x16h(a[mi][s], b[PAR ^ 1][s], acc[mi][1]);
        }
        H8_PHASE_END();
        // ---- phase 2: (bottom, right)
        if (!no_lds) {
#pragma unroll
            for (int s = 1; s < 4; ++s)
#pragma unroll
                for (int mi = 0; mi < 2; ++mi) a[mi][s] = frag_a(PAR, H8_AB, mi, s);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (MODE == 0) { issue(t + 2, PAR, H8_AT, 0); issue(t + 2, PAR, H8_AT, 1); }
        if (MODE <= 1 && !(DBG & 8)) h8_wait_raw<(MODE == 0 ? 8 : 4)>();   // B-left(t + 1), A-top(t + 1)
        H8_PHASE_END();
if (!no_mma) {
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int mi = 0; mi < 2; ++mi) acc[2 + mi][1] = mfma32x32x16h(s == 0 ? y0[mi] : a[mi][s], b[PAR ^ 1][s], acc[2 + mi][1]);
        }
        H8_PHASE_END();
        // ---- phase 3: (bottom, left); the load slot fetches the first fragments of K-tile t + 1
        if (MODE <= 1 && !no_lds) {
#pragma unroll
            for (int s = 0; s < 4; ++s) b[PAR ^ 1][s] = frag_b(PAR ^ 1, H8_BL, s);
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) a[mi][0] = frag_a(PAR ^ 1, H8_AT, mi, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (MODE == 0) { issue(t + 2, PAR, H8_BR, 0); issue(t + 2, PAR, H8_BR, 1); }
        H8_PHASE_END();
        if (!no_mma) {
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int mi = 0; mi < 2; ++mi) acc[2 + mi][0] = mfma32x32x16h(s == 0 ? y0[mi] : a[mi][s], b[PAR][s], acc[2 + mi][0]);
        }
        H8_PHASE_END();
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;

    // ---- prologue: the seven half-tiles phases -7 .. -1 would have issued, then the fragments phase -1 would have read
#pragma unroll
    for (int j = 0; j < 7; ++j) { issue(j >> 2, j >> 2, j & 3, 0); issue(j >> 2, j >> 2, j & 3, 1); }
    h8_wait_raw<10>();                              // B-left(0), A-top(0) have landed
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < 4; ++s) b[0][s] = frag_b(0, H8_BL, s);
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) a[mi][0] = frag_a(0, H8_AT, mi, 0);
    __builtin_amdgcn_sched_barrier(0);
    if (wr == 1) __builtin_amdgcn_s_barrier();      // group 1 runs one barrier behind
    __builtin_amdgcn_sched_barrier(0);

    // an odd number of K-tiles runs one more, which reads zeros (see issue): the loop body is two K-tiles (the B register sets swap roles)
    const int Te = T + (T & 1);
    unsigned long long tc0 = 0, tc1 = 0;
    if constexpr (DBG & 64) tc0 = __builtin_amdgcn_s_memtime();
    int t = 0;
    for (; t < Te - 2; t += 2) {
        ktile(t, I0{}, I0{});
        ktile(t + 1, I1{}, I0{});
    }
    ktile(t, I0{}, I1{});
    ktile(t + 1, I1{}, I2{});
#undef H8_PHASE_END
    if (wr == 0) __builtin_amdgcn_s_barrier();      // pairs with group 1's last barrier
    if constexpr (DBG & 64) tc1 = __builtin_amdgcn_s_memtime();

    // ---- epilogue
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
    if constexpr (DBG & 64) {
        if (lane == 0 && wave == 0 && g.bias) {
            unsigned long long* o = (unsigned long long*)g.bias + 4 * blockIdx.x;
            o[0] = tc0; o[1] = tc1; o[2] = __builtin_amdgcn_s_memtime();
        }
    }
#endif
}

inline bool hgemm8_ok(long long M, int N, int K) { return K % 64 == 0 && K >= 128 && M > 0 && N > 0; }

template <int DBG = 0>
inline void hgemm8_launch(const Hg8Args& a0, hipStream_t st) {
    Hg8Args a = a0;
    a.tiles_m = (a.M + 255) / 256;
    a.tiles_n = (a.N + 255) / 256;
    static bool once = false;
    if (!once) {
        (void)hipFuncSetAttribute((const void*)hgemm8_kernel<DBG>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)H8_LDS);
        once = true;
    }
    mg_launch(hgemm8_kernel<DBG>, dim3((unsigned)(a.tiles_m * a.tiles_n)), dim3(512), (size_t)H8_LDS, st, a);
}
