// What a tile of the table-stationary K1 / K2 kernels (mdctgan_amd/csrc/mdct_bs.h) spends where: the product kernels and
// variants with parts switched off, timed with HIP events.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I include -I scripts/ubench scripts/ubench/mdct_bs_bench.hip -o scripts/ubench/mdct_bs_bench
//   scripts/ubench/mdct_bs_bench [clips=4096]
#include "../../mdctgan_amd/csrc/mdct.hip"
#include "mdct_bs.h"       // the retired table-stationary kernels (round 5: no longer in the library)
#include <cmath>
#include <cstdio>
#include <vector>

template <typename F>
static float time_ms(F launch, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) launch();
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    for (int i = 0; i < iters; ++i) launch();
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    return ms / iters;
}

int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 4096, T = 32512, F = 128;
    std::vector<float> hx((size_t)B * T), hw(512), hd(256 * 256);
    unsigned s = 12345u;
    for (auto& v : hx) { s = s * 1664525u + 1013904223u; v = 0.05f * ((float)(s >> 8) / 8388608.0f - 1.0f); }
    for (int i = 0; i < 512; ++i) hw[i] = (float)sin(M_PI * (i + 0.5) / 512.0);
    for (int n = 0; n < 256; ++n) for (int k = 0; k < 256; ++k) hd[n * 256 + k] = (float)cos(M_PI / 256.0 * (n + 0.5) * (k + 0.5));
    float *x, *w, *d, *spec, *in2, *y;
    hipMalloc(&x, hx.size() * 4); hipMalloc(&w, 2048); hipMalloc(&d, (256 * 256 + 256 * 256 + mg_dct4_image_floats(512)) * 4);
    float* dimg = d + 256 * 256;
    hipMalloc(&spec, (size_t)B * F * 256 * 4); hipMalloc(&in2, (size_t)B * F * 512 * 4); hipMalloc(&y, (size_t)B * T * 4);
    hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(w, hw.data(), 2048, hipMemcpyHostToDevice);
    hipMemcpy(d, hd.data(), 256 * 256 * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(dct4_image_kernel, dim3(256 * 256 / 4 / 256), dim3(256), 0, 0, d, dimg);      // the f32 register image of the table
    float* dimg_ct = dimg + 256 * 256;                                                                   // the library's image (factored kernels)
    mg_dct4_image(d, dimg_ct, nullptr);
    CodecParams cp{CODEC_ARCSINH, 1000.f, -1.f, 1.f, -5.f, 5.f, nullptr, nullptr, 0};
    const int n_tiles = B * F / 32, iters = B >= 1024 ? 10 : 200;
    const double gflop = 2.0 * B * F * 256.0 * 256.0 * 1e-9;
    auto report = [&](const char* name, float ms) {
        printf("%-44s %9.2f us  %6.1f TFLOP/s (%.2f of 157.3)  %6.0f GB/s\n", name, ms * 1e3, gflop / ms, gflop / ms / 157.3,
               B * 393216.0 / ms * 1e-6);
    };
#define K1(NW_, MODE_, PAIR_, DBG_, name)                                                                                  \
    {                                                                                                                       \
        auto k = mdct4_bs_kernel<NW_, MODE_, PAIR_, false, DBG_>;                                                                  \
        hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)BS_K1_LDS);                    \
        const dim3 grid(n_tiles < 256 ? n_tiles : 256, 8 / NW_);                                                            \
        report(name, time_ms([&] { hipLaunchKernelGGL(k, grid, dim3(NW_ * 64), BS_K1_LDS, 0, x, B, T, F, w, dimg, cp, spec, in2, (double*)nullptr); }, iters)); \
    }
    printf("== K1, %d clips (%d row tiles)\n", B, n_tiles);
    K1(8, CODEC_ARCSINH, true, 0, "K1 nw8 (product)");
    K1(8, CODEC_ARCSINH, true, 1, "K1 nw8, no global stores");
    K1(8, CODEC_ARCSINH, true, 2, "K1 nw8, no codec math");
    K1(8, CODEC_ARCSINH, true, 3, "K1 nw8, no stores, no codec");
    K1(8, CODEC_ARCSINH, true, 4, "K1 nw8, no fold");
    K1(8, CODEC_ARCSINH, true, 7, "K1 nw8 MFMA + LDS only");
    K1(2, CODEC_ARCSINH, true, 0, "K1 nw2 (product, small batches)");
    K1(2, CODEC_ARCSINH, true, 7, "K1 nw2 MFMA + LDS only");
    K1(8, CODEC_ARCSINH, false, 0, "K1 nw8, no pair");
    K1(8, CODEC_RAW, false, 0, "K1 nw8, RAW codec, no pair");
    printf("== through the C ABI (mg_mdct4_forward / mg_imdct4_forward)\n");
    report("mg_mdct4_forward arcsinh + pair", time_ms([&] { mg_mdct4_forward(x, B, T, 512, w, d, dimg_ct, CODEC_ARCSINH, 1000.f, -1.f, 1.f, -5.f, 5.f, 0, spec, in2, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr); }, iters));
    report("mg_imdct4_forward arcsinh", time_ms([&] { mg_imdct4_forward(spec, B, F, 512, w, d, dimg_ct, CODEC_ARCSINH, 1000.f, -1.f, 1.f, -5.f, 5.f, nullptr, nullptr, y, T, 0, nullptr, nullptr); }, iters));
    report("mg_imdct4_forward raw", time_ms([&] { mg_imdct4_forward(spec, B, F, 512, w, d, dimg_ct, CODEC_RAW, 1000.f, -1.f, 1.f, -5.f, 5.f, nullptr, nullptr, y, T, 0, nullptr, nullptr); }, iters));
    return 0;
}
