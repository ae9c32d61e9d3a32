// Tuning harness for mdctgan_amd/csrc/dense_gemm_h.h (f16 GEMM, LDS-DMA): the im2col GEMM shapes of the --fp16 trunk layers.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I mdctgan_amd/csrc -I include -I scripts/ubench scripts/ubench/hgemm_bench.hip -o scripts/ubench/hgemm_bench
// (round 6: the round-5 research kernels -- dense_gemm_h8.h, the 256 x 256 x 64 balanced ping-pong at 98.7 % of the MFMA issue bound /
// 1.07-1.25 PF wall clock under DVFS, and h8pp.h, its unbalanced first version -- left the tree: no layer of this model has the >= 256
// tiles of 256 x 256 they need; findings and numbers are in HISTORY.md section 3 "Round 5", logs in profiles/r05_hgemm8_ubench.log)
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
#include "dense_gemm_h.h"

__global__ void ref_kernel(const _Float16* A, const _Float16* B, float* out, int M, int N, int K, int lda, int ldb) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)M * N) return;
    const int m = (int)(idx / N), n = (int)(idx % N);
    double s = 0.0;
    for (int k = 0; k < K; ++k) s += (double)(float)A[(size_t)m * lda + k] * (double)(float)B[(size_t)n * ldb + k];
    out[idx] = (float)s;
}
__global__ void fill_h(_Float16* p, size_t n, unsigned seed) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned x = (unsigned)i * 2654435761u + seed;
        x ^= x >> 15; x *= 2246822519u; x ^= x >> 13; x *= 3266489917u; x ^= x >> 16;
        p[i] = (_Float16)(((float)(x & 0xffffff) / 8388608.0f) - 1.0f);
    }
}
// HG_DATA=zero | relu (half the values zero, the rest uniform in [0, 1)) | small (uniform * 2^-6): what the DVFS governor does with the data
__global__ void refill_h(_Float16* p, size_t n, int mode) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float v = (float)p[i];
        p[i] = (_Float16)(mode == 0 ? 0.0f : mode == 1 ? (v > 0.0f ? v : 0.0f) : v * 0.015625f);
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
__global__ void reduce_kernel(const float* part, int S, size_t n, float* out) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n / 4; i += (size_t)gridDim.x * blockDim.x) {
        float4 s = make_float4(0, 0, 0, 0);
        for (int z = 0; z < S; ++z) { const float4 t = *(const float4*)(part + (size_t)z * n + 4 * i); s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w; }
        *(float4*)(out + 4 * i) = s;
    }
}
struct Variant { const char* name; int bm, bn; void (*launch)(const HgArgs&, hipStream_t); };
__global__ void ref_rc_kernel(const _Float16* A, const _Float16* B, float* out, int M, int N, int K) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)M * N) return;
    const int m = (int)(idx / N), n = (int)(idx % N);
    double s = 0.0;
    for (int k = 0; k < K; ++k) s += (double)(float)A[(size_t)m * K + k] * (double)(float)B[(size_t)k * N + n];
    out[idx] = (float)s;
}
#define VR(BM, BN, WM, WN) {#BM "x" #BN "/" #WM "x" #WN " rc", BM, BN, hgemm_launch<BM, BN, WM, WN, true>}
#define VRN(BM, BN, WM, WN, NB) {#BM "x" #BN "/" #WM "x" #WN " rc n" #NB, BM, BN, hgemm_launch<BM, BN, WM, WN, true, NB>}
const Variant vrc[] = {{"SA 4x2 rc", 128, 128, hgemm_sa_launch<true, false>}, {"SA 2x2 rc", 128, 128, hgemm_sa_launch<true, false, 2, 2>}, {"SA 2x4 rc", 128, 128, hgemm_sa_launch<true, false, 2, 4>}, VR(128, 64, 2, 2), VR(128, 128, 4, 2), VR(256, 64, 4, 2), VR(256, 128, 4, 2),
                       VRN(128, 64, 2, 2, 3), VRN(128, 64, 2, 2, 4), VRN(128, 128, 4, 2, 3), VRN(128, 128, 4, 2, 4), VRN(256, 64, 4, 2, 3), VRN(256, 64, 4, 2, 4),
                       VRN(256, 128, 4, 2, 3)};
#define V(BM, BN, WM, WN) {#BM "x" #BN "/" #WM "x" #WN, BM, BN, hgemm_launch<BM, BN, WM, WN>}
#define VN(BM, BN, WM, WN, NB) {#BM "x" #BN "/" #WM "x" #WN " n" #NB, BM, BN, hgemm_launch<BM, BN, WM, WN, false, NB>}
const Variant vs[] = {{"SA 4x2", 128, 128, hgemm_sa_launch<false, false>}, {"SA 2x2", 128, 128, hgemm_sa_launch<false, false, 2, 2>}, {"SA 2x4", 128, 128, hgemm_sa_launch<false, false, 2, 4>}, {"SA 4x1", 128, 128, hgemm_sa_launch<false, false, 4, 1>}, V(128, 64, 2, 2), V(128, 128, 2, 2), V(128, 128, 4, 2), V(256, 64, 4, 2), V(256, 128, 4, 2), V(256, 256, 4, 2),
                      V(256, 128, 2, 2), V(256, 256, 2, 2), V(512, 128, 4, 2), V(256, 128, 4, 1), V(512, 128, 4, 1),
                      VN(128, 64, 2, 2, 3), VN(128, 64, 2, 2, 4), VN(128, 128, 2, 2, 3), VN(128, 128, 2, 2, 4), VN(128, 128, 4, 2, 3), VN(128, 128, 4, 2, 4),
                      VN(256, 64, 4, 2, 3), VN(256, 64, 4, 2, 4), VN(256, 128, 4, 2, 3), VN(256, 256, 4, 2, 3) };
struct Problem { const char* name; int M, N, K; };
}  // namespace

int main(int argc, char** argv) {
    const char* only = argc > 1 ? argv[1] : "";
    const Problem probs[] = {
        {"trunk2048 fwd/dgrad  px256", 256, 2048, 18432},
        {"trunk2048 wgrad      px256", 2048, 18432, 256},
        {"cfg1 1024ch fwd      px1024", 1024, 1024, 9216},
        {"cfg1 1024ch wgrad    px1024", 1024, 9216, 1024},
        {"ladder 1024->2048    px256", 256, 2048, 9216},
        {"square 4096", 4096, 4096, 4096},
        {"square 8192", 8192, 8192, 8192},
        {"local128 as a dense GEMM (65536 px x 128 co, K = 9 x 128)", 65536, 128, 1152},
    };
    hipStream_t st; hipStreamCreate(&st);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float* dstat; hipMalloc(&dstat, 8);
    for (const Problem& pr : probs) {
        if (only[0] && !strstr(pr.name, only)) continue;
        const size_t na = (size_t)pr.M * pr.K, nb = (size_t)pr.N * pr.K, nc = (size_t)pr.M * pr.N;
        _Float16 *A, *B; float *C, *R, *part;
        const int nrot = getenv("HG_ROT") ? atoi(getenv("HG_ROT")) : 1;     // > 1: cycle through copies of B (defeats the 256 MiB Infinity Cache)
        hipMalloc(&A, na * 2); hipMalloc(&B, nb * 2 * nrot); hipMalloc(&C, nc * 4); hipMalloc(&R, nc * 4);
        const int max_splits = 32;
        hipMalloc(&part, nc * 4 * (pr.M * (size_t)pr.N > (64u << 20) ? 1 : max_splits));
        fill_h<<<1024, 256, 0, st>>>(A, na, 1u);
        for (int i = 0; i < nrot; ++i) fill_h<<<1024, 256, 0, st>>>(B + (size_t)i * nb, nb, 7u);
        if (const char* dm = getenv("HG_DATA")) {
            const int mode = !strcmp(dm, "zero") ? 0 : !strcmp(dm, "relu") ? 1 : 2;
            refill_h<<<1024, 256, 0, st>>>(A, na, mode);
            if (mode != 1) refill_h<<<1024, 256, 0, st>>>(B, nb * nrot, mode);      // relu: activations only, weights stay dense
        }
        ref_kernel<<<(unsigned)((nc + 255) / 256), 256, 0, st>>>(A, B, R, pr.M, pr.N, pr.K, pr.K, pr.K);
        hipStreamSynchronize(st);
        const double flops = 2.0 * pr.M * (double)pr.N * pr.K;
        const double bytes = (na + nb) * 2.0 + nc * 4.0;
        printf("== %s  M=%d N=%d K=%d  %.2f GFLOP, %.1f MB (%.1f us at 5 TB/s, %.1f us at 2.5 PF)\n", pr.name, pr.M, pr.N, pr.K,
               flops / 1e9, bytes / 1e6, bytes / 5e12 * 1e6, flops / 2.5e15 * 1e6);
        for (const Variant& v : vs) {
            if (pr.N % v.bn != 0 || v.bm > 2 * pr.M) continue;
            const int chunks = pr.K / 64;
            for (int sp : {1, 2, 4, 8, 16, 32}) {
                if (sp > 1 && (chunks / sp < 4 || pr.M * (size_t)pr.N > (64u << 20))) continue;
                const long long wgs = (long long)((pr.M + v.bm - 1) / v.bm) * (pr.N / v.bn) * sp;
                if (wgs > 16384 || (sp > 1 && wgs > 2048) || (wgs < 64)) continue;
                if (v.name[0] == '8' && sp > 1) continue;
                if (getenv("HG_ONLY8") && !strstr(v.name, "8p") && !strstr(v.name, "8-phase")) continue;
                HgArgs g{};
                g.A = A; g.B = B; g.C = C; g.part = sp > 1 ? part : nullptr; g.bias = nullptr;
                g.M = pr.M; g.N = pr.N; g.K = pr.K; g.lda = pr.K; g.ldb = pr.K;
                g.cps = (chunks + sp - 1) / sp; g.splits = (chunks + g.cps - 1) / g.cps; g.round_f16 = 0; g.accumulate = 0; g.b_cpt = 1 << 30; g.b_tap_stride = 0;
                int rot = 0;
                auto run = [&]() {
                    g.B = B + (size_t)rot * nb;
                    rot = rot + 1 == nrot ? 0 : rot + 1;
                    v.launch(g, st);
                    if (g.splits > 1) reduce_kernel<<<1024, 256, 0, st>>>(part, g.splits, nc, C);
                };
                hipMemsetAsync(C, 0, nc * 4, st);
                hipMemsetAsync(dstat, 0, 8, st);
                run();
                maxdiff_kernel<<<512, 256, 0, st>>>(C, R, nc, dstat);
                float h[2];
                hipMemcpyAsync(h, dstat, 8, hipMemcpyDeviceToHost, st);
                hipStreamSynchronize(st);
                if (hipGetLastError() != hipSuccess) { printf("   %-12s launch failed\n", v.name); continue; }
                std::vector<float> ts;
                for (int rd = 0; rd < 3; ++rd) {
                    run();
                    hipEventRecord(e0, st);
                    for (int it = 0; it < 10; ++it) run();
                    hipEventRecord(e1, st);
                    hipEventSynchronize(e1);
                    float ms; hipEventElapsedTime(&ms, e0, e1);
                    ts.push_back(ms * 1e2f);
                }
                std::sort(ts.begin(), ts.end());
                const float err = h[0] / (h[1] > 0 ? h[1] : 1.f);
                printf("   %-18s splits %2d wgs %5lld  med %8.1f us  %7.1f TF  %6.2f TB/s  relerr %.1e%s\n", v.name, g.splits, wgs, ts[1],
                       flops / ts[1] / 1e6, bytes / ts[1] / 1e6, err, err > 1e-4f ? "  <-- WRONG" : "");
            }
        }
        if (hgemm_as_ok(pr.M, pr.N, pr.K)) {
            const int tn = pr.N / 128, tmm = pr.M / 128;
            for (int groups : {0, 256 / tmm / 2, 2 * 256 / tmm, 3 * 256 / tmm}) {
                if (groups > tn) continue;
                char buf[16]; snprintf(buf, sizeof buf, "%d", groups);
                if (groups) setenv("MG_HGEMM_AS_GROUPS", buf, 1); else unsetenv("MG_HGEMM_AS_GROUPS");
                float* flag = dstat + 1;
                auto run = [&]() { hgemm_as_launch(A, B, C, pr.M, pr.N, pr.K, 0, nullptr, st); };
                hipMemsetAsync(C, 0, nc * 4, st);
                hipMemsetAsync(dstat, 0, 8, st);
                run();
                maxdiff_kernel<<<512, 256, 0, st>>>(C, R, nc, dstat);
                float h[2];
                hipMemcpyAsync(h, dstat, 8, hipMemcpyDeviceToHost, st);
                hipStreamSynchronize(st);
                if (hipGetLastError() != hipSuccess) { printf("   A-stationary launch failed\n"); continue; }
                std::vector<float> ts;
                for (int rd = 0; rd < 3; ++rd) {
                    run();
                    hipEventRecord(e0, st);
                    for (int it = 0; it < 10; ++it) run();
                    hipEventRecord(e1, st);
                    hipEventSynchronize(e1);
                    float ms; hipEventElapsedTime(&ms, e0, e1);
                    ts.push_back(ms * 1e2f);
                }
                std::sort(ts.begin(), ts.end());
                const float err = h[0] / (h[1] > 0 ? h[1] : 1.f);
                printf("   A-stationary       groups %3d           med %8.1f us  %7.1f TF  %6.2f TB/s  relerr %.1e%s\n", groups, ts[1],
                       flops / ts[1] / 1e6, bytes / ts[1] / 1e6, err, err > 1e-4f ? "  <-- WRONG" : "");
                (void)flag;
            }
            unsetenv("MG_HGEMM_AS_GROUPS");
        }
        fflush(stdout);
        hipFree(A); hipFree(B); hipFree(C); hipFree(R); hipFree(part);
    }
    // ---- data-gradient form: B row-contiguous [K][N]
    for (const Problem& pr : {Problem{"trunk2048 dgrad (B [K][N])", 256, 2048, 18432}, Problem{"cfg1 1024ch dgrad (B [K][N])", 1024, 1024, 9216}}) {
        if (only[0] && !strstr(pr.name, only)) continue;
        const size_t na = (size_t)pr.M * pr.K, nb = (size_t)pr.N * pr.K, nc = (size_t)pr.M * pr.N;
        _Float16 *A, *B; float *C, *R, *part;
        hipMalloc(&A, na * 2); hipMalloc(&B, nb * 2); hipMalloc(&C, nc * 4); hipMalloc(&R, nc * 4); hipMalloc(&part, nc * 4 * 32);
        fill_h<<<1024, 256, 0, st>>>(A, na, 1u);
        fill_h<<<1024, 256, 0, st>>>(B, nb, 7u);
        ref_rc_kernel<<<(unsigned)((nc + 255) / 256), 256, 0, st>>>(A, B, R, pr.M, pr.N, pr.K);
        hipStreamSynchronize(st);
        const double flops = 2.0 * pr.M * (double)pr.N * pr.K;
        printf("== %s  M=%d N=%d K=%d\n", pr.name, pr.M, pr.N, pr.K);
        for (const Variant& v : vrc) {
            if (pr.N % v.bn != 0) continue;
            const int chunks = pr.K / 64;
            for (int sp : {4, 8, 16, 32}) {
                const long long wgs = (long long)((pr.M + v.bm - 1) / v.bm) * (pr.N / v.bn) * sp;
                if (wgs > 2048 || wgs < 128) continue;
                HgArgs g{};
                g.A = A; g.B = B; g.C = C; g.part = part; g.M = pr.M; g.N = pr.N; g.K = pr.K; g.lda = pr.K; g.ldb = pr.N;
                g.b_cpt = 1 << 30; g.b_tap_stride = 0;
                g.cps = (chunks + sp - 1) / sp; g.splits = (chunks + g.cps - 1) / g.cps;
                auto run = [&]() { v.launch(g, st); reduce_kernel<<<1024, 256, 0, st>>>(part, g.splits, nc, C); };
                hipMemsetAsync(dstat, 0, 8, st);
                run();
                maxdiff_kernel<<<512, 256, 0, st>>>(C, R, nc, dstat);
                float h[2];
                hipMemcpyAsync(h, dstat, 8, hipMemcpyDeviceToHost, st);
                hipStreamSynchronize(st);
                std::vector<float> ts;
                for (int rd = 0; rd < 3; ++rd) {
                    run();
                    hipEventRecord(e0, st);
                    for (int it = 0; it < 10; ++it) run();
                    hipEventRecord(e1, st);
                    hipEventSynchronize(e1);
                    float ms; hipEventElapsedTime(&ms, e0, e1);
                    ts.push_back(ms * 1e2f);
                }
                std::sort(ts.begin(), ts.end());
                const float err = h[0] / (h[1] > 0 ? h[1] : 1.f);
                printf("   %-22s splits %2d wgs %5lld  med %8.1f us  %7.1f TF  relerr %.1e%s\n", v.name, g.splits, wgs, ts[1], flops / ts[1] / 1e6, err,
                       err > 1e-4f ? "  <-- WRONG" : "");
            }
        }
        fflush(stdout);
        hipFree(A); hipFree(B); hipFree(C); hipFree(R); hipFree(part);
    }
    return 0;
}
