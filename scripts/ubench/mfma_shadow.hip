// How many independent VALU / LDS / SALU instructions issue in the shadow of one v_mfma_f32_32x32x2_f32?
// One wave per SIMD (256 blocks x 256 threads), 4 independent accumulators, K filler instructions after each MFMA.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int K, int KIND>
__global__ __launch_bounds__(256) void kern(float* out, int iters) {
    __shared__ float lds[4096];
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = f32x16{0};
    float a = threadIdx.x * 0.001f, b = 1.0f + threadIdx.x * 0.002f;
    float f[16];
    for (int i = 0; i < 16; ++i) f[i] = a + i;
    int si = iters;
    lds[threadIdx.x] = a;
    __syncthreads();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            __builtin_amdgcn_sched_barrier(0);
            acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[m], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < K; ++k) {
                if (KIND == 0) f[k & 15] = __builtin_fmaf(f[k & 15], 1.0001f, 0.5f);                 // VALU
                else if (KIND == 1) f[k & 15] += lds[(threadIdx.x + 64 * k + it) & 4095];             // DS read (+ VALU add)
                else asm volatile("s_add_u32 %0, %0, 1" : "+s"(si));                                   // SALU
            }
        }
    }
    float s = 0;
    for (int i = 0; i < 16; ++i) s += f[i];
    for (int m = 0; m < 4; ++m) s += acc[m][0];
    out[blockIdx.x * 256 + threadIdx.x] = s + si;
}

template <int K, int KIND>
void run(const char* name) {
    float* out; hipMalloc(&out, 256 * 256 * 4);
    const int iters = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((kern<K, KIND>), dim3(256), dim3(256), 0, 0, out, 100);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((kern<K, KIND>), dim3(256), dim3(256), 0, 0, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double ns_per_mfma = ms * 1e6 / (iters * 4.0);
    printf("%-6s K=%2d : %.1f ns per MFMA (%.0f clk @2.4GHz)\n", name, K, ns_per_mfma, ns_per_mfma * 2.4);
    hipFree(out);
}
int main() {
    run<0, 0>("valu"); run<2, 0>("valu"); run<4, 0>("valu"); run<6, 0>("valu"); run<8, 0>("valu"); run<10, 0>("valu");
    run<12, 0>("valu"); run<14, 0>("valu"); run<16, 0>("valu"); run<20, 0>("valu"); run<24, 0>("valu"); run<32, 0>("valu");
    run<1, 1>("ds"); run<2, 1>("ds"); run<4, 1>("ds"); run<8, 1>("ds");
    run<4, 2>("salu"); run<8, 2>("salu"); run<16, 2>("salu"); run<32, 2>("salu");
    return 0;
}
