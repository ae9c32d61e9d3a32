// Sustained MFMA rate of this device under the clocks it actually holds: N waves per SIMD, 4 independent accumulator chains,
// nothing but MFMAs.  The dense peaks in MI355X_MICROARCH.md (157.3 TFLOP/s f32, 2.5 PFLOP/s f16) assume 2.4 GHz.
//   hipcc --offload-arch=gfx950 -O3 scripts/ubench/mfma_peak.hip -o scripts/ubench/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
template <int KIND>
__global__ __launch_bounds__(256) void kern(float* out, int iters) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = f32x16{0};
    const float a = threadIdx.x * 0.001f, b = 1.0f + threadIdx.x * 0.002f;
    f16x8 ha, hb;
    for (int i = 0; i < 8; ++i) { ha[i] = (_Float16)(a + i); hb[i] = (_Float16)(b - i); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                if (KIND == 0) acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[m], 0, 0, 0);
                else acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, acc[m], 0, 0, 0);
            }
    }
    float s = 0;
    for (int m = 0; m < 4; ++m) for (int i = 0; i < 16; ++i) s += acc[m][i];
    if (s == 123.456f) out[0] = s;
}
template <int KIND>
void run(const char* name, int wgs_per_cu, double flop_per_mfma) {
    float* out; hipMalloc(&out, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int iters : {2000, 20000, 100000}) {
        const int grid = 256 * wgs_per_cu;
        kern<KIND><<<grid, 256>>>(out, 100);
        hipEventRecord(e0, 0);
        kern<KIND><<<grid, 256>>>(out, iters);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double flops = (double)grid * 4 * iters * 16 * flop_per_mfma;
        printf("%-28s waves/SIMD %d  iters %6d  %9.1f us  %8.1f TFLOP/s\n", name, wgs_per_cu, iters, ms * 1e3, flops / ms / 1e9);
    }
}
int main() {
    run<0>("v_mfma_f32_32x32x2_f32", 1, 32.0 * 32 * 2 * 2);
    run<0>("v_mfma_f32_32x32x2_f32", 2, 32.0 * 32 * 2 * 2);
    run<1>("v_mfma_f32_32x32x16_f16", 1, 32.0 * 32 * 16 * 2);
    run<1>("v_mfma_f32_32x32x16_f16", 2, 32.0 * 32 * 16 * 2);
    return 0;
}
