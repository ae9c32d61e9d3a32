// Stand-alone tuning harness for mdctgan_amd/csrc/dense_gemm.h: every (tile, wave grid) variant on the batched GEMM shapes
// of the Winograd families, checked against a naive float64-accumulating kernel, timed with HIP events (interleaved rounds).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I mdctgan_amd/csrc -I include scripts/ubench/gemm_bench.hip -o /tmp/gemm_bench
#include "common.h"
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
namespace {
template <typename KernelT, typename ArgT>
inline void mg_launch(KernelT kern, dim3 grid, dim3 block, size_t lds, hipStream_t st, const ArgT& a) {
    hipLaunchKernelGGL(kern, grid, block, lds, st, a);
}
#include "dense_gemm.h"

__global__ void ref_gemm_kernel(DgArgs g, int alay, int blay, float* out) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long per = (long long)g.M * g.N;
    if (idx >= per * g.P) return;
    const int z = (int)(idx / per);
    const int m = (int)((idx % per) / g.N), n = (int)(idx % g.N);
    const float* A = g.A + (size_t)z * g.sa;
    const float* B = g.B + (size_t)z * g.sb;
    double s = 0.0;
    for (int k = 0; k < g.K; ++k) {
        const float av = alay == DG_KC ? A[(size_t)m * g.lda + k] : A[(size_t)k * g.lda + m];
        const float bv = blay == DG_KC ? B[(size_t)n * g.ldb + k] : B[(size_t)k * g.ldb + n];
        s += (double)av * bv;
    }
    out[(size_t)z * g.sc + (size_t)m * g.N + n] = (float)s;
}
__global__ void fill_kernel(float* p, size_t n, unsigned seed) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned x = (unsigned)i * 2654435761u + seed;
        x ^= x >> 15; x *= 2246822519u; x ^= x >> 13; x *= 3266489917u; x ^= x >> 16;
        p[i] = ((float)(x & 0xffffff) / 8388608.0f) - 1.0f;
    }
}
__global__ void maxdiff_kernel(const float* a, const float* b, size_t n, float* out) {
    float m = 0.f, mx = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        m = fmaxf(m, fabsf(a[i] - b[i]));
        mx = fmaxf(mx, fabsf(b[i]));
    }
    atomicMax((unsigned*)out, __float_as_uint(m));
    atomicMax((unsigned*)out + 1, __float_as_uint(mx));
}

struct Variant { const char* name; int bm, bn; void (*launch)(const DgArgs&, hipStream_t); };
#define V(BM, BN, WM, WN, AL, BL) {#BM "x" #BN "/" #WM "x" #WN, BM, BN, dgemm32_launch<BM, BN, WM, WN, AL, BL>}
#define G(BM, BN, WM, WN, AL, BL) {#BM "x" #BN "/" #WM "x" #WN " glds", BM, BN, dgemm32g_launch<BM, BN, WM, WN, AL, BL>}
#define G3(BM, BN, WM, WN, AL, BL) {#BM "x" #BN "/" #WM "x" #WN " glds3", BM, BN, dgemm32g_launch<BM, BN, WM, WN, AL, BL, 3>}
#define GE(BM, BN, WM, WN, AL, BL) {#BM "x" #BN "/" #WM "x" #WN " glds b8", BM, BN, dgemm32g_launch<BM, BN, WM, WN, AL, BL, 2, 0, 1>}
#define VSET(AL, BL) { V(64, 64, 2, 2, AL, BL), V(128, 64, 2, 2, AL, BL), V(64, 128, 2, 2, AL, BL), V(128, 128, 2, 2, AL, BL), \
                       V(128, 128, 4, 2, AL, BL), V(128, 128, 2, 4, AL, BL), V(128, 64, 4, 2, AL, BL), V(64, 128, 2, 4, AL, BL), \
                       V(256, 64, 4, 2, AL, BL), V(64, 256, 2, 4, AL, BL), V(256, 128, 4, 2, AL, BL), V(128, 256, 2, 4, AL, BL), \
                       G(64, 64, 2, 2, AL, BL), G(128, 64, 2, 2, AL, BL), G(64, 128, 2, 2, AL, BL), G(128, 128, 2, 2, AL, BL), \
                       G(128, 128, 4, 2, AL, BL), G(128, 128, 2, 4, AL, BL), G(256, 128, 4, 2, AL, BL), G(128, 256, 2, 4, AL, BL), \
                       G(64, 256, 2, 4, AL, BL), G(256, 64, 4, 2, AL, BL), \
                       G3(64, 64, 2, 2, AL, BL), G3(128, 64, 2, 2, AL, BL), G3(64, 128, 2, 2, AL, BL), G3(128, 128, 2, 2, AL, BL), \
                       G3(128, 128, 4, 2, AL, BL), G3(128, 128, 2, 4, AL, BL), G3(64, 256, 2, 4, AL, BL), G3(256, 64, 4, 2, AL, BL), \
                       GE(64, 64, 2, 2, AL, BL), GE(128, 64, 2, 2, AL, BL), GE(64, 128, 2, 2, AL, BL), GE(128, 128, 2, 2, AL, BL), GE(128, 128, 4, 2, AL, BL), GE(256, 128, 4, 2, AL, BL) }
const Variant v_kk[] = VSET(DG_KC, DG_KC);
const Variant v_kr[] = VSET(DG_KC, DG_RC);
const Variant v_rr[] = VSET(DG_RC, DG_RC);

#define VA(BM, BN, WM, WN, ABL) {#BM "x" #BN "/" #WM "x" #WN " abl" #ABL, BM, BN, dgemm32_launch<BM, BN, WM, WN, DG_KC, DG_KC, ABL>}
const Variant v_abl[] = {
    VA(64, 64, 2, 2, 0), VA(64, 64, 2, 2, 1), VA(64, 64, 2, 2, 3), VA(64, 64, 2, 2, 7), VA(64, 64, 2, 2, 15), VA(64, 64, 2, 2, 31),
    VA(64, 64, 2, 2, 16), VA(64, 64, 2, 2, 8), VA(64, 64, 2, 2, 4),
    VA(128, 128, 2, 2, 0), VA(128, 128, 2, 2, 1), VA(128, 128, 2, 2, 3), VA(128, 128, 2, 2, 7), VA(128, 128, 2, 2, 15), VA(128, 128, 2, 2, 31),
    VA(128, 128, 2, 2, 16), VA(128, 128, 2, 2, 8), VA(128, 128, 2, 2, 4),
    VA(128, 128, 4, 2, 0), VA(128, 128, 4, 2, 7), VA(128, 128, 4, 2, 15), VA(128, 128, 4, 2, 31),
    VA(64, 128, 2, 2, 0), VA(64, 128, 2, 2, 7), VA(64, 128, 2, 2, 15), VA(64, 128, 2, 2, 31),
};
struct Problem { const char* name; int P, M, N, K, alay, blay; };
}  // namespace

int main(int argc, char** argv) {
    const char* only = argc > 1 ? argv[1] : "";
    const Problem probs[] = {
        {"f23_fwd   T256  1024x1024", 16, 256, 1024, 1024, DG_KC, DG_KC},
        {"f23_dgrad T256  1024x1024", 16, 256, 1024, 1024, DG_KC, DG_RC},
        {"f23_wgrad T256  1024x1024", 16, 1024, 1024, 256, DG_RC, DG_RC},
        {"f23_fwd   T2048 1024x1024 (batch 64)", 16, 2048, 1024, 1024, DG_KC, DG_KC},
        {"f44_fwd   T2448 256->512", 25, 2448, 512, 256, DG_KC, DG_KC},
        {"f44_dgrad T2448 512->256", 25, 2448, 256, 512, DG_KC, DG_RC},
        {"f44_wgrad T2448 512x256", 25, 512, 256, 2448, DG_RC, DG_RC},
        {"f42_fwd   T720 K512->256", 25, 720, 256, 512, DG_KC, DG_KC},
        {"f42_wgrad T720 256x512", 25, 256, 512, 720, DG_RC, DG_RC},
        {"f23_fwd   T64 2048x2048 (cfg2 trunk)", 16, 64, 2048, 2048, DG_KC, DG_KC},
        {"f23_dgrad T64 2048x2048", 16, 64, 2048, 2048, DG_KC, DG_RC},
        {"f23_wgrad T64 2048x2048", 16, 2048, 2048, 64, DG_RC, DG_RC},
        {"f23_fwd   T16384 128x128 (cfg2 local)", 16, 16384, 128, 128, DG_KC, DG_KC},
        {"f23_wgrad T16384 128x128", 16, 128, 128, 16384, DG_RC, DG_RC},
    };
    hipStream_t st;
    hipStreamCreate(&st);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float* dstat;
    hipMalloc(&dstat, 8);
    for (const Problem& pr : probs) {
        if (only[0] && !strstr(pr.name, only)) continue;
        const size_t na = (size_t)pr.P * pr.M * pr.K, nb = (size_t)pr.P * pr.K * pr.N, nc = (size_t)pr.P * pr.M * pr.N;
        float *A, *B, *C, *R, *part;
        hipMalloc(&A, na * 4); hipMalloc(&B, nb * 4); hipMalloc(&C, nc * 4); hipMalloc(&R, nc * 4);
        const int max_splits = 8;
        hipMalloc(&part, nc * 4 * max_splits);
        fill_kernel<<<1024, 256, 0, st>>>(A, na, 1u);
        fill_kernel<<<1024, 256, 0, st>>>(B, nb, 7u);
        DgArgs g{};
        g.A = A; g.B = B; g.C = C; g.part = nullptr; g.M = pr.M; g.N = pr.N; g.K = pr.K;
        g.lda = pr.alay == DG_KC ? pr.K : pr.M;
        g.ldb = pr.blay == DG_KC ? pr.K : pr.N;
        g.sa = (long long)pr.M * pr.K; g.sb = (long long)pr.K * pr.N; g.sc = (long long)pr.M * pr.N;
        g.P = pr.P; g.splits = 1; g.cps = 1 << 28;
        DgArgs gr = g; gr.C = R;
        ref_gemm_kernel<<<(unsigned)((nc + 255) / 256), 256, 0, st>>>(gr, pr.alay, pr.blay, R);
        hipStreamSynchronize(st);
        const Variant* vs = pr.alay == DG_RC ? v_rr : (pr.blay == DG_RC ? v_kr : v_kk);
        const int nv = 36;
        const double flops = 2.0 * pr.P * pr.M * (double)pr.N * pr.K;
        printf("== %s  P=%d M=%d N=%d K=%d  %.2f GFLOP (peak-time %.1f us)\n", pr.name, pr.P, pr.M, pr.N, pr.K, flops / 1e9,
               flops / 157.3e12 * 1e6);
        std::vector<std::vector<float>> times(nv);
        std::vector<float> err(nv, -1.f);
        std::vector<char> skip(nv, 0);
        for (int vi = 0; vi < nv; ++vi) {
            const Variant& v = vs[vi];
            if (pr.N % v.bn != 0 || (pr.alay == DG_RC && pr.M % 4) ) { skip[vi] = 1; continue; }
            if (v.bm > pr.M * 2 || v.bn > pr.N) { skip[vi] = 1; continue; }
            if (strstr(v.name, "glds") && pr.K % 32 != 0 && !(pr.alay == DG_RC && pr.blay == DG_RC)) { skip[vi] = 1; continue; }
            hipMemsetAsync(C, 0, nc * 4, st);
            hipMemsetAsync(dstat, 0, 8, st);
            v.launch(g, st);
            maxdiff_kernel<<<512, 256, 0, st>>>(C, R, nc, dstat);
            float h[2];
            hipMemcpyAsync(h, dstat, 8, hipMemcpyDeviceToHost, st);
            hipStreamSynchronize(st);
            if (hipGetLastError() != hipSuccess) { skip[vi] = 1; continue; }
            err[vi] = h[0] / (h[1] > 0 ? h[1] : 1.f);
        }
        const int rounds = 5, iters = 10;
        for (int rd = 0; rd < rounds; ++rd)
            for (int vi = 0; vi < nv; ++vi) {
                if (skip[vi]) continue;
                vs[vi].launch(g, st);
                hipEventRecord(e0, st);
                for (int it = 0; it < iters; ++it) vs[vi].launch(g, st);
                hipEventRecord(e1, st);
                hipEventSynchronize(e1);
                float ms;
                hipEventElapsedTime(&ms, e0, e1);
                times[vi].push_back(ms * 1e3f / iters);
            }
        for (int vi = 0; vi < nv; ++vi) {
            if (skip[vi]) continue;
            std::sort(times[vi].begin(), times[vi].end());
            const float mn = times[vi][0], med = times[vi][rounds / 2];
            const int tiles = ((pr.M + vs[vi].bm - 1) / vs[vi].bm) * (pr.N / vs[vi].bn) * pr.P;
            printf("   %-19s wgs %5d  min %8.1f us  med %8.1f us  %6.1f TF (%4.1f%%)  relerr %.1e%s\n", vs[vi].name, tiles, mn, med,
                   flops / med / 1e6, flops / med / 1e6 / 157.3 * 100.0, err[vi], err[vi] > 2e-5f ? "  <-- WRONG" : "");
        }
        // split-K on the best small-grid shapes (weight-gradient-like problems with few tiles)
        if (pr.K >= 2048) {
            for (int sp : {2, 4, 8}) {
                for (int vi : {0, 1, 3}) {
                    if (skip[vi]) continue;
                    DgArgs gs = g;
                    gs.part = part; gs.splits = sp;
                    const int chunks = (pr.K + 31) / 32;
                    gs.cps = (chunks + sp - 1) / sp;
                    gs.splits = (chunks + gs.cps - 1) / gs.cps;
                    vs[vi].launch(gs, st);
                    hipEventRecord(e0, st);
                    for (int it = 0; it < iters; ++it) vs[vi].launch(gs, st);
                    hipEventRecord(e1, st);
                    hipEventSynchronize(e1);
                    float ms;
                    hipEventElapsedTime(&ms, e0, e1);
                    printf("   %-14s splitK %d: %8.1f us (without the reduction pass)\n", vs[vi].name, gs.splits, ms * 1e3f / iters);
                }
            }
        }
        fflush(stdout);
        hipFree(A); hipFree(B); hipFree(C); hipFree(R); hipFree(part);
    }
    // ---- ablation on the forward problem (results are wrong by construction: timing only)
    if (!only[0] || strstr("ablation", only)) {
        for (int M : {256, 2048}) {
            const int P = 16, N = 1024, K = 1024;
            const size_t na = (size_t)P * M * K, nb = (size_t)P * K * N, nc = (size_t)P * M * N;
            float *A, *B, *C;
            hipMalloc(&A, na * 4); hipMalloc(&B, nb * 4); hipMalloc(&C, nc * 4);
            fill_kernel<<<1024, 256, 0, st>>>(A, na, 1u);
            fill_kernel<<<1024, 256, 0, st>>>(B, nb, 7u);
            DgArgs g{};
            g.A = A; g.B = B; g.C = C; g.part = nullptr; g.M = M; g.N = N; g.K = K; g.lda = K; g.ldb = K;
            g.sa = (long long)M * K; g.sb = (long long)K * N; g.sc = (long long)M * N; g.P = P; g.splits = 1; g.cps = 1 << 28;
            const double flops = 2.0 * P * M * (double)N * K;
            printf("== ablation fwd M=%d (peak-time %.1f us): abl bits 1 no global loads, 2 no LDS stores, 4 no barriers, 8 no fragment reads, 16 no stores\n",
                   M, flops / 157.3e12 * 1e6);
            for (const Variant& v : v_abl) {
                std::vector<float> ts;
                for (int rd = 0; rd < 3; ++rd) {
                    v.launch(g, st);
                    hipEventRecord(e0, st);
                    for (int it = 0; it < 10; ++it) v.launch(g, st);
                    hipEventRecord(e1, st);
                    hipEventSynchronize(e1);
                    float ms;
                    hipEventElapsedTime(&ms, e0, e1);
                    ts.push_back(ms * 1e2f);
                }
                std::sort(ts.begin(), ts.end());
                printf("   %-22s min %8.1f us  med %8.1f us  (%4.1f%% of peak)\n", v.name, ts[0], ts[1], flops / ts[1] / 1e6 / 157.3 * 100.0);
            }
            fflush(stdout);
            hipFree(A); hipFree(B); hipFree(C);
        }
    }
    return 0;
}
