// Batched dense GEMM on the exact-f32 MFMA pipe for the Winograd-domain products (forward, data gradient, weight
// gradient of the F(2x2,3x3) / F(2x2,4x4) / F(4x4,2x2) families): C_z[M][N] = A_z[M][K] * B_z[K][N], z = 0..P-1.
//
// What differs from the convolution kernels' dense paths (conv_igemm.hip) -- all measured on MI355X with
// scripts/ubench/gemm_bench.hip:
//  * the workgroup tile and the per-wave tile are template parameters: a wave owns MB x NB accumulator blocks of 32x32
//    (two or four independent MFMA chains per wave, every LDS fragment feeds NB / MB MFMAs), 4 or 8 waves per workgroup;
//  * 32-deep chunks for every operand layout: k-contiguous operands (DG_KC: [rows][K]) are staged as full 128-byte
//    lines into a row-major LDS image (pitch 36) and read back with one ds_read_b128 per four k; row-contiguous operands
//    (DG_RC: [K][rows]) land k-major with ds_write_b128 and are read with ds_read_b32 -- no transposition anywhere;
//  * ONE 1-D grid over (position z, K split, tile): the XCD remap hands every XCD whole positions, so A_z and B_z are
//    pulled through exactly one private L2 (the 2-D grids of the convolution kernels gave every XCD a slice of every
//    position: A_z was fetched by all eight L2s, 218 MB of HBM traffic per launch against 96 MB algorithmic);
//  * K tails (K % 32 != 0, K % 4 == 0) are zero-filled in the stage, rows past M re-read the last row and are not stored.
// Included inside the including translation unit's anonymous namespace, after common.h.
#pragma once

constexpr int DG_BK = 32, DG_LDK = DG_BK + 4;
enum { DG_KC = 0, DG_RC = 1 };

struct DgArgs {
    const float* A;
    const float* B;
    float* C;
    float* part;            // split-K slabs [splits][P][M][N] (nullptr: splits == 1, results go to C)
    int M, N, K;
    int lda, ldb;           // DG_KC: elements between consecutive rows; DG_RC: elements between consecutive k
    long long sa, sb, sc;   // elements between consecutive positions
    int P, tiles_m, tiles_n, splits, cps;   // cps: 32-deep chunks per split
    int ldc;                // dgemm32g: elements between consecutive rows of C (0: N); split-K slabs stay [M][N]
    int kb;                 // dgemm32g: rows a row-contiguous B really has (0: K) -- k rows behind it read as zeros
    int round_f16;          // dgemm32g: round the direct result through float16 (autocast output of the tap GEMMs)
};

__device__ __forceinline__ float4 dg_ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }

// Tile order inside one position: tm fastest, in groups of 8 row tiles.  An XCD runs ~64 workgroups at a time (two per CU) and they
// walk K together, so of the operands only what the CONCURRENT tiles share is fetched once: with tm simply fastest and 16 row tiles
// (batch 64: M = 2048 pixel tiles) a wave of workgroups is 16 x 4 tiles and the 8 MB A_z slab is pulled through the 4 MB L2 twice
// (measured 470 MB per launch against 336 MB algorithmic); 8 x 8 tiles fetch A_z once and the 4 MB B_z twice.
__device__ __forceinline__ void dg_tile_order(int rem, int tiles_m, int tiles_n, int& tn, int& tm) {
    constexpr int GM = 8;
    if (tiles_m > GM && tiles_m % GM == 0) {
        const int grp = rem / (GM * tiles_n), r2 = rem - grp * GM * tiles_n;
        tn = r2 / GM;
        tm = grp * GM + (r2 - tn * GM);
    } else {
        tn = rem / tiles_m;
        tm = rem - tn * tiles_m;
    }
}

template <int BM, int BN, int WGM, int WGN, int ALAY, int BLAY>
struct DgCfg {
    static constexpr int NT = 64 * WGM * WGN;
    static constexpr int MB = BM / WGM / 32, NB = BN / WGN / 32;
    static constexpr int ASZ = (ALAY == DG_KC) ? BM * DG_LDK : DG_BK * BM;
    static constexpr int BSZ = (BLAY == DG_KC) ? BN * DG_LDK : DG_BK * BN;
    static constexpr size_t LDS_BYTES = (size_t)2 * (ASZ + BSZ) * sizeof(float);
    static_assert(BM % (32 * WGM) == 0 && BN % (32 * WGN) == 0, "wave tile must be whole 32x32 blocks");
    static_assert((BM * 8) % NT == 0 && (BN * 8) % NT == 0, "stage must divide evenly over the threads");
};

// One operand's stage: R rows of the tile, layout LAY.  NV float4 per thread and chunk.
template <int R, int LAY, int NT>
struct DgStage {
    static constexpr int NV = R * 8 / NT;
    static constexpr int QR = R / 4;                 // DG_RC: float4 per k row
    static constexpr int KSTEP = (LAY == DG_RC) ? NT / QR : 0;
    static_assert(LAY == DG_KC || (NT % QR == 0 && KSTEP * NV == DG_BK), "row-contiguous stage shape");
    const float* p[NV];
    const float* safe;                               // a valid address for the masked loads of a K tail
    int kq;                                          // DG_KC: 4 * (tid & 7); DG_RC: tid / QR (k row of slot 0)
    int lds_off[NV];
    float4 v[NV];

    __device__ __forceinline__ void init(const float* base, int ld, int r0, int rows_total, int tid, int k_begin) {
        safe = base;
        if (LAY == DG_KC) {
            kq = 4 * (tid & 7);
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int row = (tid >> 3) + i * (NT / 8);
                p[i] = base + (size_t)min(r0 + row, rows_total - 1) * ld + k_begin + kq;
                lds_off[i] = row * DG_LDK + kq;
            }
        } else {
            const int cq = tid % QR;
            kq = tid / QR;
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int k = kq + i * KSTEP;
                p[i] = base + (size_t)(k_begin + k) * ld + min(r0 + 4 * cq, rows_total - 4);
                lds_off[i] = k * R + 4 * cq;
            }
        }
    }
    // k0: first k of the chunk being loaded; K: reduction length.  Only the last chunk can be partial (K % 32 != 0): it
    // takes the masked path (wave-uniform branch).  The common path must not touch the loaded registers -- a select on
    // them would put an s_waitcnt vmcnt right behind the loads and stall the wave for the full memory latency in the
    // middle of every chunk (measured: 17 us of an 89 us launch).
    __device__ __forceinline__ void load(int k0, int K, int ld) {
        if (k0 + DG_BK <= K) {
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                v[i] = dg_ld4(p[i]);
                p[i] += (LAY == DG_KC) ? DG_BK : (size_t)DG_BK * ld;
            }
        } else {
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int k = k0 + ((LAY == DG_KC) ? kq : kq + i * KSTEP);
                const bool ok = k < K;
                const float4 t = dg_ld4(ok ? p[i] : safe);
                v[i] = ok ? t : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
    }
    __device__ __forceinline__ void stash(float* S) const {
#pragma unroll
        for (int i = 0; i < NV; ++i) *reinterpret_cast<float4*>(S + lds_off[i]) = v[i];
    }
};

// ABL (ablation bits, tuning harness only; 0 in the product): 1 no global loads, 2 no LDS stores, 4 no barriers,
// 8 fragments read once per chunk, 16 no result stores
template <int MB, int NB, int ALAY, int BLAY, int BM, int BN, int ABL = 0, typename F0, typename F1>
__device__ __forceinline__ void dg_chunk(const float* As, const float* Bs, f32x16 (&acc)[MB][NB], int wm0, int wn0,
                                         int lane, F0&& after_s0, F1&& after_s1) {
    const int r = lane & 31, kh = lane >> 5;
    float a[2][MB][4], b[2][NB][4];
    auto fetch = [&](int s, int buf) {
#pragma unroll
        for (int mi = 0; mi < MB; ++mi) {
            if (ALAY == DG_KC) {
                const float4 t = dg_ld4(As + (wm0 + 32 * mi + r) * DG_LDK + 8 * s + 4 * kh);
                a[buf][mi][0] = t.x; a[buf][mi][1] = t.y; a[buf][mi][2] = t.z; a[buf][mi][3] = t.w;
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) a[buf][mi][j] = As[(8 * s + 4 * kh + j) * BM + wm0 + 32 * mi + r];
            }
        }
#pragma unroll
        for (int ni = 0; ni < NB; ++ni) {
            if (BLAY == DG_KC) {
                const float4 t = dg_ld4(Bs + (wn0 + 32 * ni + r) * DG_LDK + 8 * s + 4 * kh);
                b[buf][ni][0] = t.x; b[buf][ni][1] = t.y; b[buf][ni][2] = t.z; b[buf][ni][3] = t.w;
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) b[buf][ni][j] = Bs[(8 * s + 4 * kh + j) * BN + wn0 + 32 * ni + r];
            }
        }
    };
    fetch(0, 0);
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        if (s + 1 < 4) {
            if (ABL & 8) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
#pragma unroll
                    for (int mi = 0; mi < MB; ++mi) a[(s + 1) & 1][mi][j] = a[s & 1][mi][j];
#pragma unroll
                    for (int ni = 0; ni < NB; ++ni) b[(s + 1) & 1][ni][j] = b[s & 1][ni][j];
                }
            } else {
                fetch(s + 1, (s + 1) & 1);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int mi = 0; mi < MB; ++mi)
#pragma unroll
                for (int ni = 0; ni < NB; ++ni) acc[mi][ni] = mfma32x32x2(a[s & 1][mi][j], b[s & 1][ni][j], acc[mi][ni]);
        __builtin_amdgcn_sched_barrier(0);
        if (s == 0) after_s0();
        if (s == 1) after_s1();
        if (s <= 1) __builtin_amdgcn_sched_barrier(0);
    }
}

template <int BM, int BN, int WGM, int WGN, int ALAY, int BLAY, int ABL = 0>
__global__ __launch_bounds__(64 * WGM * WGN) void dgemm32_kernel(DgArgs g) {
    using Cfg = DgCfg<BM, BN, WGM, WGN, ALAY, BLAY>;
    constexpr int NT = Cfg::NT, MB = Cfg::MB, NB = Cfg::NB;
    extern __shared__ __attribute__((aligned(16))) float dg_smem[];
    float* As0 = dg_smem;
    float* Bs0 = dg_smem + 2 * Cfg::ASZ;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tiles = g.tiles_m * g.tiles_n, per_z = tiles * g.splits;
    const int L = xcd_remap(blockIdx.x, per_z * g.P);
    const int z = L / per_z;
    int rem = L - z * per_z;
    const int sp = rem / tiles;
    rem -= sp * tiles;
    const int tn = rem / g.tiles_m, tm = rem - tn * g.tiles_m;
    const int m0 = tm * BM, n0 = tn * BN;
    const int total_chunks = (g.K + DG_BK - 1) / DG_BK;
    const int c_begin = sp * g.cps, c_end = min(total_chunks, c_begin + g.cps);

    DgStage<BM, ALAY, NT> sa;
    DgStage<BN, BLAY, NT> sb;
    sa.init(g.A + (size_t)z * g.sa, g.lda, m0, g.M, tid, c_begin * DG_BK);
    sb.init(g.B + (size_t)z * g.sb, g.ldb, n0, g.N, tid, c_begin * DG_BK);

    f32x16 acc[MB][NB];
#pragma unroll
    for (int mi = 0; mi < MB; ++mi)
#pragma unroll
        for (int ni = 0; ni < NB; ++ni) acc[mi][ni] = f32x16{0};
    const int wm0 = (wave / WGN) * (BM / WGM), wn0 = (wave % WGN) * (BN / WGN);

    if (c_begin < c_end) {
        sa.load(c_begin * DG_BK, g.K, g.lda);
        sb.load(c_begin * DG_BK, g.K, g.ldb);
        sa.stash(As0);
        sb.stash(Bs0);
    }
    __syncthreads();
    if (c_begin + 1 < c_end) {
        sa.load((c_begin + 1) * DG_BK, g.K, g.lda);
        sb.load((c_begin + 1) * DG_BK, g.K, g.ldb);
    }
    for (int c = c_begin; c < c_end; ++c) {
        const int cur = (c - c_begin) & 1;
        auto f0 = [&]() {
            if (!(ABL & 2) && c + 1 < c_end) {
                sa.stash(As0 + (cur ^ 1) * Cfg::ASZ);
                sb.stash(Bs0 + (cur ^ 1) * Cfg::BSZ);
            }
        };
        auto f1 = [&]() {
            if (!(ABL & 1) && c + 2 < c_end) {
                sa.load((c + 2) * DG_BK, g.K, g.lda);
                sb.load((c + 2) * DG_BK, g.K, g.ldb);
            }
        };
        dg_chunk<MB, NB, ALAY, BLAY, BM, BN, ABL>(As0 + cur * Cfg::ASZ, Bs0 + cur * Cfg::BSZ, acc, wm0, wn0, lane, f0, f1);
        if (!(ABL & 4)) __syncthreads();
    }

    float* o = g.part ? g.part + ((size_t)sp * g.P + z) * ((size_t)g.M * g.N) : g.C + (size_t)z * g.sc;
#pragma unroll
    for (int mi = 0; mi < MB; ++mi)
#pragma unroll
        for (int ni = 0; ni < NB; ++ni) {
            const int col = n0 + wn0 + 32 * ni + (lane & 31);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm0 + 32 * mi + mfma32_row(r, lane);
                if ((ABL & 16) ? (acc[mi][ni][r] == 1.2345e-30f) : (row < g.M && col < g.N))
                    o[(size_t)row * g.N + col] = acc[mi][ni][r];
            }
        }
}

template <int BM, int BN, int WGM, int WGN, int ALAY, int BLAY, int ABL = 0>
inline void dgemm32_launch(const DgArgs& a0, hipStream_t st) {
    using Cfg = DgCfg<BM, BN, WGM, WGN, ALAY, BLAY>;
    DgArgs a = a0;
    a.tiles_m = (a.M + BM - 1) / BM;
    a.tiles_n = (a.N + BN - 1) / BN;
    static bool once = false;
    if (!once) {
        hipFuncSetAttribute((const void*)dgemm32_kernel<BM, BN, WGM, WGN, ALAY, BLAY, ABL>,
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)Cfg::LDS_BYTES);
        once = true;
    }
    const unsigned grid = (unsigned)((long long)a.tiles_m * a.tiles_n * a.splits * a.P);
    mg_launch(dgemm32_kernel<BM, BN, WGM, WGN, ALAY, BLAY, ABL>, dim3(grid), dim3(Cfg::NT), Cfg::LDS_BYTES, st, a);
}


// =====================================================================================================================
// The same GEMM with LDS-DMA staging (buffer_load_dwordx4 ... lds): global memory -> LDS without a VGPR round trip, no
// ds_write pass, no per-chunk address arithmetic (the chunk offset is the instruction's scalar offset).  Measured on the
// register-staged kernel above (scripts/ubench/gemm_bench.hip, ablation table): of a 75.8 us launch 5.8 us are the
// global loads and 5.7 us the LDS stores of the stage; the MFMA-only floor of the launch shape is 61 us.
//  * a wave instruction moves 64 lanes x 16 B = 1 KiB to a wave-uniform LDS address + 16 * lane, so the LDS images carry no
//    padding.  DG_KC: 8 rows x 128 B per piece, the 16-byte quad q of row r is stored at quad q ^ ((r >> 1) & 7) (the
//    source address is permuted, the 128-byte line per row is still one coalesced request) and ds_read_b128 applies the same
//    involution: conflict-free for its 16-lane groups.  DG_RC: pieces are runs of whole k rows, read with ds_read_b32.
//  * two LDS buffers: chunk c + 1 streams in while chunk c is multiplied; one barrier per chunk.
//  * K % 32 == 0, or both operands row-contiguous (their K tail reads as zeros through the buffer range check); the
//    launcher falls back to the register-staged kernel otherwise.
// =====================================================================================================================
typedef __attribute__((address_space(3))) void* dg_lds_ptr;

template <int BM, int BN, int WGM, int WGN, int ALAY, int BLAY, int NBUF = 2>
struct DgCfgG {
    static constexpr int NT = 64 * WGM * WGN, NW = WGM * WGN;
    static constexpr int MB = BM / WGM / 32, NB = BN / WGN / 32;
    static constexpr int ASZ = BM * DG_BK, BSZ = BN * DG_BK;          // floats per buffer, unpadded
    static constexpr int PA = BM / 8 / NW, PB = BN / 8 / NW;          // 1 KiB pieces per wave and chunk
    static constexpr size_t LDS_BYTES = (size_t)NBUF * (ASZ + BSZ) * sizeof(float);
    static_assert(NBUF >= 2 && NBUF <= 4, "two to four LDS buffers");
    static_assert((BM / 8) % NW == 0 && (BN / 8) % NW == 0, "pieces must divide evenly over the waves");
};

// byte offset of this lane's 16 bytes for piece `piece` of an operand tile with R rows
template <int R, int LAY>
__device__ __forceinline__ unsigned dg_piece_voffset(int piece, int lane, int r0, int rows_total, int ld) {
    if (LAY == DG_KC) {
        const int row = 8 * piece + (lane >> 3), pq = lane & 7;
        const int q = pq ^ ((row >> 1) & 7);
        return ((unsigned)min(r0 + row, rows_total - 1) * (unsigned)ld + 4u * q) * 4u;
    } else {
        constexpr int LPR = R / 4;                 // lanes per k row
        const int k = piece * (64 / LPR) + lane / LPR, c4 = lane % LPR;
        return ((unsigned)k * (unsigned)ld + (unsigned)min(r0 + 4 * c4, rows_total - 4)) * 4u;
    }
}

// Row-contiguous operands ([K][rows]) with TWO 32-row blocks per wave are read interleaved (round 4): lane r of the MFMA stands for
// tile rows 2 r and 2 r + 1 (block mi = the row's parity) instead of r and 32 + r, so the two blocks' values of one k are
// 8 adjacent bytes -- one ds_read_b64 per k instead of two ds_read_b32 (the weight-gradient GEMM, both operands row-contiguous,
// issued 16 LDS reads per 8 MFMAs, the forward GEMM 4).  Which lane computes which element of C is the kernel's own business:
// dg_tile_row / dg_tile_col give the epilogue the same mapping.
template <int LAY, int BLOCKS>
struct DgInterleave { static constexpr bool on = (LAY == DG_RC && BLOCKS == 2); };
template <bool INT, int BLOCKS>
__device__ __forceinline__ int dg_tile_index(int blk, int r) { return INT ? BLOCKS * r + blk : 32 * blk + r; }

// INTL: opt-in (dgemm32g_kernel, whose epilogue knows the mapping); the direct-convolution kernels of conv_dma.h share this
// chunk with their own block-mapped epilogues.
template <int MB, int NB, int ALAY, int BLAY, int BM, int BN, bool INTL = false>
__device__ __forceinline__ void dg_chunk_g(const float* As, const float* Bs, f32x16 (&acc)[MB][NB], int wm0, int wn0,
                                           int lane) {
    const int r = lane & 31, kh = lane >> 5;
    constexpr bool AINT = INTL && DgInterleave<ALAY, MB>::on, BINT = INTL && DgInterleave<BLAY, NB>::on;
    float a[2][MB][4], b[2][NB][4];
    auto fetch = [&](int s, int buf) {
        if (AINT) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float2 t = *reinterpret_cast<const float2*>(As + (8 * s + 4 * kh + j) * BM + wm0 + 2 * r);
                a[buf][0][j] = t.x; a[buf][MB - 1][j] = t.y;
            }
        } else {
#pragma unroll
            for (int mi = 0; mi < MB; ++mi) {
                const int row = wm0 + 32 * mi + r;
                if (ALAY == DG_KC) {
                    const float4 t = dg_ld4(As + row * DG_BK + 4 * ((2 * s + kh) ^ ((row >> 1) & 7)));
                    a[buf][mi][0] = t.x; a[buf][mi][1] = t.y; a[buf][mi][2] = t.z; a[buf][mi][3] = t.w;
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) a[buf][mi][j] = As[(8 * s + 4 * kh + j) * BM + row];
                }
            }
        }
        if (BINT) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float2 t = *reinterpret_cast<const float2*>(Bs + (8 * s + 4 * kh + j) * BN + wn0 + 2 * r);
                b[buf][0][j] = t.x; b[buf][NB - 1][j] = t.y;
            }
        } else {
#pragma unroll
            for (int ni = 0; ni < NB; ++ni) {
                const int row = wn0 + 32 * ni + r;
                if (BLAY == DG_KC) {
                    const float4 t = dg_ld4(Bs + row * DG_BK + 4 * ((2 * s + kh) ^ ((row >> 1) & 7)));
                    b[buf][ni][0] = t.x; b[buf][ni][1] = t.y; b[buf][ni][2] = t.z; b[buf][ni][3] = t.w;
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) b[buf][ni][j] = Bs[(8 * s + 4 * kh + j) * BN + row];
                }
            }
        }
    };
    fetch(0, 0);
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        if (s + 1 < 4) fetch(s + 1, (s + 1) & 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int mi = 0; mi < MB; ++mi)
#pragma unroll
                for (int ni = 0; ni < NB; ++ni) acc[mi][ni] = mfma32x32x2(a[s & 1][mi][j], b[s & 1][ni][j], acc[mi][ni]);
        __builtin_amdgcn_sched_barrier(0);
    }
}


// ------------------------------------------------------------------------------------------------------------------
// float32 GEMM chunk on the bf16 MFMA pipe ("3 x bf16"): v_mfma_f32_32x32x2_f32 runs at 1/16 of the rate of
// v_mfma_f32_32x32x16_bf16, so a float32 product is formed from bf16 pieces instead.  x = hi + mid + lo with hi =
// bf16(x), mid = bf16(x - hi), lo = bf16(x - hi - mid) (round-to-nearest-even; both subtractions are exact in float32) is an
// EXACT decomposition of a normal float32: |x - hi| <= 2^-8 ulp_exp, |x - hi - mid| <= 2^-16, and what is left is a
// multiple of the float32 ulp below 2^7 ulps, i.e. fits bf16's 8-bit significand.  Every bf16 x bf16 product is exact in
// float32 (16 bits), so a.b = sum of the 9 piece products exactly; the kernel issues 8 of them and drops lo.lo (<= 2^-32
// of |a||b|, far below the 2^-24 of the float32 accumulation both forms share).  Accumulation is the MFMA's float32
// accumulator, rounded once per 16 k and product group (9 roundings per 16 k against 8 for the float32 instruction).
// 8 x 32-cycle MFMAs per 32x32x16 block instead of 8 x 64-cycle ones: twice the float32 pipe's rate at float32 accuracy;
// the split costs ~5.5 VALU operations per loaded operand element, issued in the MFMAs' shadow.
// ------------------------------------------------------------------------------------------------------------------
typedef __bf16 dg_bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned dg_u32x4 __attribute__((ext_vector_type(4)));
struct DgSplit { dg_u32x4 hi, mid, lo; };
__device__ __forceinline__ unsigned dg_cvt_pk_bf16(float a, float b) {
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ void dg_split8(const float (&x)[8], DgSplit& o) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float x0 = x[2 * j], x1 = x[2 * j + 1];
        const unsigned h = dg_cvt_pk_bf16(x0, x1);
        const float r0 = x0 - __uint_as_float(h << 16), r1 = x1 - __uint_as_float(h & 0xffff0000u);
        const unsigned m = dg_cvt_pk_bf16(r0, r1);
        const float q0 = r0 - __uint_as_float(m << 16), q1 = r1 - __uint_as_float(m & 0xffff0000u);
        o.hi[j] = h;
        o.mid[j] = m;
        o.lo[j] = dg_cvt_pk_bf16(q0, q1);
    }
}
__device__ __forceinline__ f32x16 dg_mfma_bf16(dg_u32x4 a, dg_u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(dg_bf16x8, a), __builtin_bit_cast(dg_bf16x8, b), c, 0, 0, 0);
}
template <int MB, int NB, int ALAY, int BLAY, int BM, int BN>
__device__ __forceinline__ void dg_chunk_b8(const float* As, const float* Bs, f32x16 (&acc)[MB][NB], int wm0, int wn0, int lane) {
    const int r = lane & 31, kh = lane >> 5;
    auto load8 = [&](const float* S, int lay, int R, int row, int s2, float (&x)[8]) {
        if (lay == DG_KC) {
            const int sw = (row >> 1) & 7;
            const float4 t0 = dg_ld4(S + row * DG_BK + 4 * ((4 * s2 + 2 * kh) ^ sw));
            const float4 t1 = dg_ld4(S + row * DG_BK + 4 * ((4 * s2 + 2 * kh + 1) ^ sw));
            x[0] = t0.x; x[1] = t0.y; x[2] = t0.z; x[3] = t0.w; x[4] = t1.x; x[5] = t1.y; x[6] = t1.z; x[7] = t1.w;
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) x[j] = S[(16 * s2 + 8 * kh + j) * R + row];
        }
    };
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
        DgSplit a[MB], b[NB];
#pragma unroll
        for (int mi = 0; mi < MB; ++mi) {
            float x[8];
            load8(As, ALAY, BM, wm0 + 32 * mi + r, s2, x);
            dg_split8(x, a[mi]);
        }
#pragma unroll
        for (int ni = 0; ni < NB; ++ni) {
            float x[8];
            load8(Bs, BLAY, BN, wn0 + 32 * ni + r, s2, x);
            dg_split8(x, b[ni]);
        }
#pragma unroll
        for (int mi = 0; mi < MB; ++mi)
#pragma unroll
            for (int ni = 0; ni < NB; ++ni) {
                f32x16 c = acc[mi][ni];
                c = dg_mfma_bf16(a[mi].lo, b[ni].mid, c);       // small terms first
                c = dg_mfma_bf16(a[mi].mid, b[ni].lo, c);
                c = dg_mfma_bf16(a[mi].lo, b[ni].hi, c);
                c = dg_mfma_bf16(a[mi].hi, b[ni].lo, c);
                c = dg_mfma_bf16(a[mi].mid, b[ni].mid, c);
                c = dg_mfma_bf16(a[mi].mid, b[ni].hi, c);
                c = dg_mfma_bf16(a[mi].hi, b[ni].mid, c);
                c = dg_mfma_bf16(a[mi].hi, b[ni].hi, c);
                acc[mi][ni] = c;
            }
    }
}

typedef int dg_v4i __attribute__((ext_vector_type(4)));

// One LDS-DMA piece: 64 lanes x 16 bytes from buffer `rsrc` at byte offset voff (per lane) + soff (scalar) to LDS byte
// address lds_dst + 16 * lane.  Inline asm on purpose: hipcc tracks the builtin form as an LDS write that may alias every
// later ds_read and puts s_waitcnt vmcnt(0) in front of the first fragment read after each issue -- the DMA of chunk
// c + 1 would be waited for before chunk c is multiplied.  An asm load is outside hipcc's counters: the kernel counts it
// itself (dg_wait_vmcnt) and orders it with a barrier.  M0 is compiler-reserved: saved and restored in the statement.
__device__ __forceinline__ void dg_dma16(unsigned voff, dg_v4i rsrc, unsigned lds_dst, unsigned soff) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %4 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(rsrc), "s"(lds_dst), "s"(soff)
                 : "memory");
}

template <int N>
__device__ __forceinline__ void dg_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// PZ: the batch count as a compile-time constant (16: F(2x2,3x3); 25: F(2x2,4x4) / F(4x4,2x2); 0: read g.P) -- one symbol per
// transform family, so a profile separates the 1024-channel residual trunk from the ladder's 5x5-tile layers.
template <int BM, int BN, int WGM, int WGN, int ALAY, int BLAY, int NBUF = 2, int PZ = 0, int EMU = 0>
__global__ __launch_bounds__(64 * WGM * WGN) void dgemm32g_kernel(DgArgs g) {
#if defined(__HIP_DEVICE_COMPILE__)     // the buffer-resource type does not exist in the host pass (which only needs the stub)
    using Cfg = DgCfgG<BM, BN, WGM, WGN, ALAY, BLAY, NBUF>;
    constexpr int MB = Cfg::MB, NB = Cfg::NB, PA = Cfg::PA, PB = Cfg::PB;
    extern __shared__ __attribute__((aligned(1024))) float dg_smem_g[];
    float* As0 = dg_smem_g;
    float* Bs0 = dg_smem_g + NBUF * Cfg::ASZ;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nz = PZ ? PZ : g.P;
    const int tiles = g.tiles_m * g.tiles_n, per_z = tiles * g.splits;
    const int L = xcd_remap(blockIdx.x, per_z * nz);
    const int z = L / per_z;
    int rem = L - z * per_z;
    const int sp = rem / tiles;
    rem -= sp * tiles;
    int tn, tm;
    dg_tile_order(rem, g.tiles_m, g.tiles_n, tn, tm);
    const int m0 = tm * BM, n0 = tn * BN;
    const int total_chunks = (g.K + DG_BK - 1) / DG_BK;    // a partial last chunk needs row-contiguous operands (see below)
    const int c_begin = sp * g.cps, c_end = min(total_chunks, c_begin + g.cps);

    // buffer descriptors over one position's operand (raw buffer, 32-bit byte offsets: the launcher bounds an operand
    // below 2 GiB): {base[31:0], base[47:32], num_records, DST_SEL/format word of a raw gfx9 buffer}
    // num_records = the operand's size in bytes: a lane whose offset falls behind it reads 0.  For a row-contiguous
    // operand ([K][rows]) that is exactly the k >= K part of a partial last chunk, so K tails cost nothing there.
    auto make_rsrc = [](const float* p, unsigned bytes) -> dg_v4i {
        const unsigned long long a = (unsigned long long)p;
        dg_v4i r;
        r[0] = (int)(unsigned)a;
        r[1] = (int)((unsigned)(a >> 32) & 0xffffu);
        r[2] = (int)bytes;
        r[3] = 0x00020000;
        return r;
    };
    const dg_v4i ra = make_rsrc(g.A + (size_t)z * g.sa, (unsigned)((ALAY == DG_KC ? g.M : g.K) * g.lda) * 4u);
    const dg_v4i rb = make_rsrc(g.B + (size_t)z * g.sb, (unsigned)((BLAY == DG_KC ? g.N : (g.kb ? g.kb : g.K)) * g.ldb) * 4u);
    unsigned va[PA], vb[PB];
#pragma unroll
    for (int i = 0; i < PA; ++i) va[i] = dg_piece_voffset<BM, ALAY>(wave * PA + i, lane, m0, g.M, g.lda);
#pragma unroll
    for (int i = 0; i < PB; ++i) vb[i] = dg_piece_voffset<BN, BLAY>(wave * PB + i, lane, n0, g.N, g.ldb);
    const unsigned step_a = (ALAY == DG_KC) ? DG_BK * 4u : (unsigned)DG_BK * (unsigned)g.lda * 4u;
    const unsigned step_b = (BLAY == DG_KC) ? DG_BK * 4u : (unsigned)DG_BK * (unsigned)g.ldb * 4u;

    // LDS byte addresses of this wave's first piece in buffer 0 (dynamic LDS starts at offset 0 of the workgroup's
    // allocation: the kernel declares no static __shared__)
    const unsigned lds_a0 = (unsigned)(size_t)(dg_lds_ptr)As0 + (unsigned)(wave * PA) * 1024u;
    const unsigned lds_b0 = (unsigned)(size_t)(dg_lds_ptr)Bs0 + (unsigned)(wave * PB) * 1024u;
    auto issue = [&](int c, int buf) {
        const unsigned sa_off = (unsigned)c * step_a, sb_off = (unsigned)c * step_b;
        const unsigned la = lds_a0 + (unsigned)buf * (unsigned)(Cfg::ASZ * 4), lb = lds_b0 + (unsigned)buf * (unsigned)(Cfg::BSZ * 4);
#pragma unroll
        for (int i = 0; i < PA; ++i) dg_dma16(va[i], ra, la + 1024u * i, sa_off);
#pragma unroll
        for (int i = 0; i < PB; ++i) dg_dma16(vb[i], rb, lb + 1024u * i, sb_off);
    };

    f32x16 acc[MB][NB];
#pragma unroll
    for (int mi = 0; mi < MB; ++mi)
#pragma unroll
        for (int ni = 0; ni < NB; ++ni) acc[mi][ni] = f32x16{0};
    const int wm0 = (wave / WGN) * (BM / WGM), wn0 = (wave % WGN) * (BN / WGN);

    if (NBUF == 2) {
        if (c_begin < c_end) issue(c_begin, 0);
        for (int c = c_begin; c < c_end; ++c) {
            const int cur = (c - c_begin) & 1;
            dg_wait_vmcnt<0>();                // this wave's pieces of chunk c have landed ...
            __builtin_amdgcn_s_barrier();      // ... everybody's have, and buffer cur ^ 1 is no longer being read
            __builtin_amdgcn_sched_barrier(0);
            if (c + 1 < c_end) issue(c + 1, cur ^ 1);
            if constexpr (EMU) dg_chunk_b8<MB, NB, ALAY, BLAY, BM, BN>(As0 + cur * Cfg::ASZ, Bs0 + cur * Cfg::BSZ, acc, wm0, wn0, lane);
            else dg_chunk_g<MB, NB, ALAY, BLAY, BM, BN, true>(As0 + cur * Cfg::ASZ, Bs0 + cur * Cfg::BSZ, acc, wm0, wn0, lane);
        }
    } else {
        // three buffers: two chunks in flight, the DMA of chunk c + 1 stays outstanding across the barrier of chunk c
        // (counted vmcnt + raw s_barrier: __syncthreads() would drain it)
        if (c_begin < c_end) issue(c_begin, 0);
        if (c_begin + 1 < c_end) issue(c_begin + 1, 1);
        int cur = 0;
        for (int c = c_begin; c < c_end; ++c) {
            if (c + 1 < c_end) dg_wait_vmcnt<PA + PB>(); else dg_wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();      // every wave's pieces of chunk c are in LDS; buffer (cur + 2) % 3 is free
            __builtin_amdgcn_sched_barrier(0);
            const int nxt2 = cur == 0 ? 2 : cur - 1;
            if (c + 2 < c_end) issue(c + 2, nxt2);
            if constexpr (EMU) dg_chunk_b8<MB, NB, ALAY, BLAY, BM, BN>(As0 + cur * Cfg::ASZ, Bs0 + cur * Cfg::BSZ, acc, wm0, wn0, lane);
            else dg_chunk_g<MB, NB, ALAY, BLAY, BM, BN, true>(As0 + cur * Cfg::ASZ, Bs0 + cur * Cfg::BSZ, acc, wm0, wn0, lane);
            cur = cur == 2 ? 0 : cur + 1;
        }
    }

    float* o = g.part ? g.part + ((size_t)sp * nz + z) * ((size_t)g.M * g.N) : g.C + (size_t)z * g.sc;
    const size_t ldc = (g.part || !g.ldc) ? (size_t)g.N : (size_t)g.ldc;
    const bool rnd = g.round_f16 != 0 && g.part == nullptr;        // wave-uniform
    // (the float32 chunk reads row-contiguous operands interleaved: see dg_chunk_g; the bf16 x 3 chunk keeps the block mapping)
    constexpr bool AINT = !EMU && DgInterleave<ALAY, MB>::on, BINT = !EMU && DgInterleave<BLAY, NB>::on;
    if (m0 + BM <= g.M && n0 + BN <= g.N) {        // interior tile (wave-uniform): no per-element predicates
        float* ow = o + (size_t)(m0 + wm0) * ldc + n0 + wn0;
#pragma unroll
        for (int mi = 0; mi < MB; ++mi)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float* orow = ow + (size_t)dg_tile_index<AINT, MB>(mi, mfma32_row(r, lane)) * ldc;
                if (BINT) {        // the two column blocks of a lane are adjacent columns: one 8-byte store
                    const float v0 = rnd ? round_h(acc[mi][0][r]) : acc[mi][0][r], v1 = rnd ? round_h(acc[mi][NB - 1][r]) : acc[mi][NB - 1][r];
                    *reinterpret_cast<float2*>(orow + 2 * (lane & 31)) = make_float2(v0, v1);
                } else {
#pragma unroll
                    for (int ni = 0; ni < NB; ++ni) orow[32 * ni + (lane & 31)] = rnd ? round_h(acc[mi][ni][r]) : acc[mi][ni][r];
                }
            }
    } else {
#pragma unroll
        for (int mi = 0; mi < MB; ++mi)
#pragma unroll
            for (int ni = 0; ni < NB; ++ni) {
                const int col = n0 + wn0 + dg_tile_index<BINT, NB>(ni, lane & 31);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = m0 + wm0 + dg_tile_index<AINT, MB>(mi, mfma32_row(r, lane));
                    if (row < g.M && col < g.N) o[(size_t)row * ldc + col] = rnd ? round_h(acc[mi][ni][r]) : acc[mi][ni][r];
                }
            }
    }
#endif
}

template <int BM, int BN, int WGM, int WGN, int ALAY, int BLAY, int NBUF = 2, int PZ = 0, int EMU = 0>
inline void dgemm32g_launch(const DgArgs& a0, hipStream_t st) {
    using Cfg = DgCfgG<BM, BN, WGM, WGN, ALAY, BLAY, NBUF>;
    DgArgs a = a0;
    a.tiles_m = (a.M + BM - 1) / BM;
    a.tiles_n = (a.N + BN - 1) / BN;
    static bool once = false;
    if (!once) {
        hipFuncSetAttribute((const void*)dgemm32g_kernel<BM, BN, WGM, WGN, ALAY, BLAY, NBUF, PZ, EMU>,
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)Cfg::LDS_BYTES);
        once = true;
    }
    const unsigned grid = (unsigned)((long long)a.tiles_m * a.tiles_n * a.splits * a.P);
    mg_launch(dgemm32g_kernel<BM, BN, WGM, WGN, ALAY, BLAY, NBUF, PZ, EMU>, dim3(grid), dim3(Cfg::NT), Cfg::LDS_BYTES, st, a);
}
