// Streaming-rate harness for the fused Adam update (mdctgan_amd/csrc/norm_act.hip::adam_dev_kernel): 16 B read + 12 B written per parameter.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/ubench/adam_bench.hip -o scripts/ubench/adam_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef float f4 __attribute__((ext_vector_type(4)));
template <int MODE, int UNROLL>
__global__ __launch_bounds__(256) void adam_k(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, size_t n4,
                                             float b1, float b2, float eps, float step_size, float bc2_sqrt, _Float16* __restrict__ w16) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i0 < n4; i0 += stride * UNROLL) {
        f4 pv[UNROLL], mv[UNROLL], vv[UNROLL], gv[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const size_t i = i0 + u * stride;
            if (i < n4) {
                if (MODE & 1) {
                    pv[u] = __builtin_nontemporal_load((const f4*)p + i); mv[u] = __builtin_nontemporal_load((const f4*)m + i);
                    vv[u] = __builtin_nontemporal_load((const f4*)v + i); gv[u] = __builtin_nontemporal_load((const f4*)g + i);
                } else {
                    pv[u] = ((const f4*)p)[i]; mv[u] = ((const f4*)m)[i]; vv[u] = ((const f4*)v)[i]; gv[u] = ((const f4*)g)[i];
                }
            }
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const size_t i = i0 + u * stride;
            if (i < n4) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float gi = gv[u][j];
                    const float mi = mv[u][j] + (gi - mv[u][j]) * (1.0f - b1);
                    const float vi = vv[u][j] * b2 + (1.0f - b2) * gi * gi;
                    mv[u][j] = mi; vv[u][j] = vi;
                    pv[u][j] = pv[u][j] - step_size * (mi / (sqrtf(vi) / bc2_sqrt + eps));
                }
                if (MODE & 2) {
                    __builtin_nontemporal_store(pv[u], (f4*)p + i); __builtin_nontemporal_store(mv[u], (f4*)m + i); __builtin_nontemporal_store(vv[u], (f4*)v + i);
                } else {
                    ((f4*)p)[i] = pv[u]; ((f4*)m)[i] = mv[u]; ((f4*)v)[i] = vv[u];
                }
                if (MODE & 4) {
                    typedef _Float16 h4 __attribute__((ext_vector_type(4)));
                    h4 hv = {(_Float16)pv[u][0], (_Float16)pv[u][1], (_Float16)pv[u][2], (_Float16)pv[u][3]};
                    ((h4*)w16)[i] = hv;
                }
            }
        }
    }
}
__global__ void check_k(const float* __restrict__ g, size_t n4, float* out) {
    float bad = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const f4 t = ((const f4*)g)[i];
        const float s = t[0] * 0.f + t[1] * 0.f + t[2] * 0.f + t[3] * 0.f;
        if (!(s == 0.f)) bad = 1.f;
    }
    if (bad != 0.f) *out = 1.f;
}
int main(int argc, char** argv) {
    const size_t n = (argc > 1 ? (size_t)atoll(argv[1]) : 700) * 1000000ull, n4 = n / 4;
    float *p, *g, *m, *v, *flag; _Float16* w16;
    hipMalloc(&p, n * 4); hipMalloc(&g, n * 4); hipMalloc(&m, n * 4); hipMalloc(&v, n * 4); hipMalloc(&w16, n * 2); hipMalloc(&flag, 4);
    hipMemset(p, 0, n * 4); hipMemset(g, 0, n * 4); hipMemset(m, 0, n * 4); hipMemset(v, 0, n * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto timeit = [&](const char* name, auto launch, double bytes) {
        std::vector<float> ts;
        for (int r = 0; r < 5; ++r) {
            launch();
            hipEventRecord(e0, 0);
            for (int i = 0; i < 3; ++i) launch();
            hipEventRecord(e1, 0); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); ts.push_back(ms / 3);
        }
        std::sort(ts.begin(), ts.end());
        printf("%-44s %8.1f us  %6.2f TB/s\n", name, ts[2] * 1e3, bytes / ts[2] / 1e9);
    };
#define RUN(MODE, UNR, BLOCKS) timeit("adam mode " #MODE " unroll " #UNR " blocks " #BLOCKS, [&]() { adam_k<MODE, UNR><<<BLOCKS, 256>>>(p, g, m, v, n4, 0.5f, 0.999f, 1e-8f, 1e-4f, 1.f, w16); }, n * ((MODE & 4) ? 30.0 : 28.0))
    RUN(0, 1, 4096); RUN(0, 1, 2048); RUN(0, 1, 8192); RUN(0, 1, 16384); RUN(0, 2, 4096); RUN(0, 2, 2048); RUN(0, 4, 2048); RUN(0, 4, 1024);
    RUN(1, 1, 4096); RUN(2, 1, 4096); RUN(3, 1, 4096); RUN(3, 2, 4096); RUN(3, 2, 2048); RUN(3, 4, 2048);
    RUN(4, 1, 4096); RUN(7, 2, 4096);
    timeit("check 4096", [&]() { check_k<<<4096, 256>>>(g, n4, flag); }, n * 4.0);
    timeit("check 16384", [&]() { check_k<<<16384, 256>>>(g, n4, flag); }, n * 4.0);
    return 0;
}
