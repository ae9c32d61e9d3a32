// Semantics probe for ds_read_b64_tr_b16 (__builtin_amdgcn_ds_read_tr16_b64_v4f16): prints which elements every lane receives.
//   hipcc --offload-arch=gfx950 -O3 scripts/ubench/tr_probe.hip -o scripts/ubench/tr_probe
#include <hip/hip_runtime.h>
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void k(const _Float16* in, float* out) {
    __shared__ __attribute__((aligned(16))) _Float16 s[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) s[i] = in[i];
    __syncthreads();
    const int lane = threadIdx.x;
    typedef __fp16 h4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
    typedef h4 __attribute__((address_space(3))) * lp;
    h4 a = __builtin_amdgcn_ds_read_tr16_b64_v4f16((lp)(s + 4 * lane));
    h4 b = __builtin_amdgcn_ds_read_tr16_b64_v4f16((lp)(s + 256 + 4 * lane));
    f16x8 v = {(_Float16)a[0], (_Float16)a[1], (_Float16)a[2], (_Float16)a[3], (_Float16)b[0], (_Float16)b[1], (_Float16)b[2], (_Float16)b[3]};
    f32x16 acc = {0};
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(v, v, acc, 0, 0, 0);
    for (int i = 0; i < 16; ++i) out[lane * 16 + i] = acc[i];
    out[2048 + lane * 4 + 0] = a[0]; out[2048 + lane * 4 + 1] = a[1]; out[2048 + lane * 4 + 2] = a[2]; out[2048 + lane * 4 + 3] = a[3];
}
int main() {
    _Float16* in; float* out;
    hipMalloc(&in, 8192); hipMalloc(&out, 16384);
    _Float16 h[4096]; for (int i = 0; i < 4096; ++i) h[i] = (_Float16)(float)(i % 2048);
    hipMemcpy(in, h, 8192, hipMemcpyHostToDevice);
    k<<<1, 64>>>(in, out);
    float o[4096]; hipMemcpy(o, out, 16384, hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; ++l) printf("lane %2d: %g %g %g %g\n", l, o[2048 + 4 * l], o[2048 + 4 * l + 1], o[2048 + 4 * l + 2], o[2048 + 4 * l + 3]);
    return 0;
}
