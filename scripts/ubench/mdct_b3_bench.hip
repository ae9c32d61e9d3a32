// The bf16 x 3 table-stationary K1 / K2 (mdctgan_amd/csrc/mdct_b3.h) beside the f32-pipe kernels (mdct_bs.h): time per launch
// with parts switched off, and the largest difference between the two kernels' spectra / waveforms on the same input.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I include -I scripts/ubench scripts/ubench/mdct_b3_bench.hip -o scripts/ubench/mdct_b3_bench
//   scripts/ubench/mdct_b3_bench [clips=4096]
#include "../../mdctgan_amd/csrc/mdct.hip"
#include "mdct_bs.h"       // the retired table-stationary kernels (round 5: no longer in the library)
#include "mdct_b3.h"
#include <cmath>
#include <cstdio>
#include <vector>

// Minimum over three rounds of `iters` back-to-back launches, after a warm-up of the same length: the first launches of a
// process run at a lower clock (the "product" line, measured first, read 595 us where the same kernel read 475 us at the
// end of the run), so single-shot numbers taken early are not comparable with ones taken late.
template <typename F>
static float time_ms(F launch, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < iters; ++i) launch();
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int rd = 0; rd < 3; ++rd) {
        hipEventRecord(e0, 0);
        for (int i = 0; i < iters; ++i) launch();
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms = 0.f;
        hipEventElapsedTime(&ms, e0, e1);
        best = ms / iters < best ? ms / iters : best;
    }
    return best;
}

static double max_diff(const float* a, const float* b, size_t n, double* amax) {
    std::vector<float> ha(n), hb(n);
    hipMemcpy(ha.data(), a, n * 4, hipMemcpyDeviceToHost);
    hipMemcpy(hb.data(), b, n * 4, hipMemcpyDeviceToHost);
    double d = 0, m = 0;
    for (size_t i = 0; i < n; ++i) { d = fmax(d, fabs((double)ha[i] - hb[i])); m = fmax(m, fabs((double)hb[i])); }
    *amax = m;
    return d;
}

int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 4096, T = 32512, F = 128;
    std::vector<float> hx((size_t)B * T), hw(512), hd(256 * 256);
    unsigned s = 12345u;
    for (auto& v : hx) { s = s * 1664525u + 1013904223u; v = 0.05f * ((float)(s >> 8) / 8388608.0f - 1.0f); }
    for (int i = 0; i < 512; ++i) hw[i] = (float)sin(M_PI * (i + 0.5) / 512.0);
    for (int n = 0; n < 256; ++n) for (int k = 0; k < 256; ++k) hd[n * 256 + k] = (float)cos(M_PI / 256.0 * (n + 0.5) * (k + 0.5));
    float *x, *w, *d, *spec, *spec2, *in2, *y, *y2;
    hipMalloc(&x, hx.size() * 4); hipMalloc(&w, 2048); hipMalloc(&d, (256 * 256 + 256 * 256 + B3_IMG_U4 * 4 + mg_dct4_image_floats(512)) * 4);
    hipMalloc(&spec, (size_t)B * F * 256 * 4); hipMalloc(&spec2, (size_t)B * F * 256 * 4); hipMalloc(&in2, (size_t)B * F * 512 * 4);
    hipMalloc(&y, (size_t)B * T * 4); hipMalloc(&y2, (size_t)B * T * 4);
    hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(w, hw.data(), 2048, hipMemcpyHostToDevice);
    hipMemcpy(d, hd.data(), 256 * 256 * 4, hipMemcpyHostToDevice);
    float* dimg = d + 256 * 256;
    hipLaunchKernelGGL(dct4_image_kernel, dim3(256 * 256 / 4 / 256), dim3(256), 0, 0, d, dimg);
    hipLaunchKernelGGL(dct4_b3_image_kernel, dim3(4 * 2 * 16 * 64 / 256), dim3(256), 0, 0, d, reinterpret_cast<b3_u4*>(dimg + 256 * 256));
    float* dimg_ct = dimg + 256 * 256 + B3_IMG_U4 * 4;          // the library's image (factored kernels)
    mg_dct4_image(d, dimg_ct, nullptr);
    const b3_u4* img3 = reinterpret_cast<const b3_u4*>(dimg + 256 * 256);
    CodecParams cp{CODEC_ARCSINH, 1000.f, -1.f, 1.f, -5.f, 5.f, nullptr, nullptr, 0};
    CodecParams cpr{CODEC_RAW, 1000.f, -1.f, 1.f, -5.f, 5.f, nullptr, nullptr, 0};
    const int n_tiles = B * F / 32, iters = B >= 1024 ? 20 : 200;
    const double gflop = 2.0 * B * F * 256.0 * 256.0 * 1e-9;
    auto report = [&](const char* name, float ms) {
        printf("%-52s %9.2f us  %6.1f TFLOP/s f32-equivalent  %6.0f GB/s (261 120 B per clip)\n", name, ms * 1e3, gflop / ms, B * 261120.0 / ms * 1e-6);
    };
    const dim3 grid(n_tiles < 256 ? n_tiles : 256);
#define K1B3(MODE_, SPEC_, PAIR_, DBG_, cp_, out_, name)                                                                  \
    {                                                                                                                       \
        auto k = mdct4_b3_kernel<MODE_, SPEC_, PAIR_, false, DBG_>;                                                         \
        hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)B3_K1_LDS);                    \
        report(name, time_ms([&] { hipLaunchKernelGGL(k, grid, dim3(256), B3_K1_LDS, 0, x, B, T, F, w, img3, cp_, out_, in2, (double*)nullptr); }, iters)); \
    }
    printf("== K1 bf16 x 3, %d clips (%d row tiles)\n", B, n_tiles);
    K1B3(CODEC_RAW, true, false, 0, cpr, spec2, "K1 b3 RAW");
    {
        auto k = mdct4_bs_kernel<8, CODEC_RAW, false, false, 0>;
        hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)BS_K1_LDS);
        report("K1 f32 pipe (mdct_bs) RAW", time_ms([&] { hipLaunchKernelGGL(k, dim3(grid.x, 1), dim3(512), BS_K1_LDS, 0, x, B, T, F, w, dimg, cpr, spec, (float*)nullptr, (double*)nullptr); }, iters));
        double amax;
        const double dd = max_diff(spec2, spec, (size_t)B * F * 256, &amax);
        printf("   RAW spectra: max |b3 - f32 pipe| = %.3e, max |X| = %.3e -> %.3e relative to the maximum\n", dd, amax, dd / amax);
    }
    K1B3(CODEC_ARCSINH, true, false, 0, cp, spec2, "K1 b3 arcsinh, spectrogram only (product)");
    K1B3(CODEC_ARCSINH, false, true, 0, cp, spec2, "K1 b3 arcsinh, pair only");
    K1B3(CODEC_ARCSINH, true, true, 0, cp, spec2, "K1 b3 arcsinh, spectrogram + pair");
    K1B3(CODEC_ARCSINH, true, false, 1, cp, spec2, "K1 b3 arcsinh, no global stores");
    K1B3(CODEC_ARCSINH, true, false, 2, cp, spec2, "K1 b3 arcsinh, no codec math");
    K1B3(CODEC_ARCSINH, true, false, 4, cp, spec2, "K1 b3 arcsinh, no DMA, no fold");
    K1B3(CODEC_ARCSINH, true, false, 8, cp, spec2, "K1 b3 arcsinh, no DMA (fold of stale data)");
    K1B3(CODEC_ARCSINH, true, false, 16, cp, spec2, "K1 b3 arcsinh, no fold (DMA + wait + barrier)");
    K1B3(CODEC_ARCSINH, true, false, 7, cp, spec2, "K1 b3 MFMA + A reads only");
#define K1B3D(DS_, name)                                                                                                  \
    {                                                                                                                       \
        auto k = mdct4_b3_kernel<CODEC_ARCSINH, true, false, false, 0, DS_>;                                                \
        hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)B3_K1_LDS);                    \
        report(name, time_ms([&] { hipLaunchKernelGGL(k, grid, dim3(256), B3_K1_LDS, 0, x, B, T, F, w, img3, cp, spec2, in2, (double*)nullptr); }, iters)); \
    }
    K1B3D(1, "K1 b3 product, DMA pieces in slots 0..8");
    K1B3D(2, "K1 b3 product, DMA piece every 2nd slot");
    K1B3D(6, "K1 b3 product, DMA piece every 6th slot");
    K1B3D(12, "K1 b3 product, DMA piece every 12th slot");
    printf("== K2 bf16 x 3\n");
    {
        hipMemsetAsync(y, 0, (size_t)B * T * 4, 0); hipMemsetAsync(y2, 0, (size_t)B * T * 4, 0);
        mg_mdct4_forward(x, B, T, 512, w, d, dimg_ct, CODEC_ARCSINH, 1000.f, -1.f, 1.f, -5.f, 5.f, 0, spec, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr);
        auto k3 = imdct4_b3_kernel<CODEC_ARCSINH, 0>;
        hipFuncSetAttribute((const void*)k3, hipFuncAttributeMaxDynamicSharedMemorySize, (int)B3_K2_LDS);
        const dim3 g2(B < 256 ? B : 256);
        report("K2 b3 arcsinh (product)", time_ms([&] { hipLaunchKernelGGL(k3, g2, dim3(256), B3_K2_LDS, 0, spec, B, F, w, img3, cp, y, T); }, iters));
        auto k3n = imdct4_b3_kernel<CODEC_ARCSINH, 1>;
        hipFuncSetAttribute((const void*)k3n, hipFuncAttributeMaxDynamicSharedMemorySize, (int)B3_K2_LDS);
        report("K2 b3 arcsinh, no global stores", time_ms([&] { hipLaunchKernelGGL(k3n, g2, dim3(256), B3_K2_LDS, 0, spec, B, F, w, img3, cp, y2, T); }, iters));
        auto k3r = imdct4_b3_kernel<CODEC_RAW, 0>;
        hipFuncSetAttribute((const void*)k3r, hipFuncAttributeMaxDynamicSharedMemorySize, (int)B3_K2_LDS);
        report("K2 b3 RAW", time_ms([&] { hipLaunchKernelGGL(k3r, g2, dim3(256), B3_K2_LDS, 0, spec, B, F, w, img3, cpr, y2, T); }, iters));
        auto kb = imdct4_bs_kernel<CODEC_ARCSINH>;
        hipFuncSetAttribute((const void*)kb, hipFuncAttributeMaxDynamicSharedMemorySize, (int)BS_K2_LDS);
        const int nt2 = B * ((F + 31) / 32);
        report("K2 f32 pipe (mdct_bs) arcsinh", time_ms([&] { hipLaunchKernelGGL(kb, dim3(nt2 < 256 ? nt2 : 256), dim3(512), BS_K2_LDS, 0, spec, B, F, w, dimg, cp, y2, T); }, iters));
        double amax;
        const double dd = max_diff(y, y2, (size_t)B * T, &amax);
        printf("   waveforms: max |b3 - f32 pipe| = %.3e, max |y| = %.3e -> %.3e relative to the maximum\n", dd, amax, dd / amax);
    }
    printf("== factored transform (mdct_ct.h), two 8-wave workgroups per CU\n");
    {
        const float* imgc = dimg_ct;
        const dim3 gc(n_tiles < 512 ? n_tiles : 512);
#define K1CT(MODE_, SPEC_, PAIR_, cp_, out_, name)                                                                        \
    {                                                                                                                       \
        auto k = mdct4_ct_kernel<MODE_, SPEC_, PAIR_, false>;                                                               \
        hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)CT_K1_LDS);                    \
        report(name, time_ms([&] { hipLaunchKernelGGL(k, gc, dim3(512), CT_K1_LDS, 0, x, B, T, F, w, imgc, cp_, out_, in2, (double*)nullptr); }, iters)); \
    }
        {
            auto kb = mdct4_bs_kernel<8, CODEC_RAW, false, false, 0>;
            hipFuncSetAttribute((const void*)kb, hipFuncAttributeMaxDynamicSharedMemorySize, (int)BS_K1_LDS);
            hipLaunchKernelGGL(kb, dim3(grid.x, 1), dim3(512), BS_K1_LDS, 0, x, B, T, F, w, dimg, cpr, spec, (float*)nullptr, (double*)nullptr);
        }
        K1CT(CODEC_RAW, true, false, cpr, spec2, "K1 ct RAW");
        double amax;
        double dd = max_diff(spec2, spec, (size_t)B * F * 256, &amax);
        printf("   RAW spectra: max |ct - f32 pipe| = %.3e, max |X| = %.3e -> %.3e relative to the maximum\n", dd, amax, dd / amax);
        K1CT(CODEC_ARCSINH, true, false, cp, spec2, "K1 ct arcsinh, spectrogram only (product)");
        {
            double* st2; hipMalloc(&st2, 16); hipMemset(st2, 0, 16);
            auto k = mdct4_ct_kernel<CODEC_ARCSINH, true, false, true>;
            hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)CT_K1_LDS);
            report("K1 ct arcsinh, spectrogram + statistics", time_ms([&] { hipLaunchKernelGGL(k, gc, dim3(512), CT_K1_LDS, 0, x, B, T, F, w, imgc, cp, spec2, in2, st2); }, iters));
            hipFree(st2);
        }
        K1CT(CODEC_ARCSINH, false, true, cp, spec2, "K1 ct arcsinh, pair only");
        K1CT(CODEC_ARCSINH, true, true, cp, spec2, "K1 ct arcsinh, spectrogram + pair");
        // K2
        mg_mdct4_forward(x, B, T, 512, w, d, dimg_ct, CODEC_ARCSINH, 1000.f, -1.f, 1.f, -5.f, 5.f, 0, spec, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr);
        auto kc = imdct4_ct_kernel<CODEC_ARCSINH>;
        hipFuncSetAttribute((const void*)kc, hipFuncAttributeMaxDynamicSharedMemorySize, (int)CT_K2_LDS);
        const dim3 g2c(B < 512 ? B : 512);
        hipMemsetAsync(y, 0, (size_t)B * T * 4, 0);
        report("K2 ct arcsinh (product)", time_ms([&] { hipLaunchKernelGGL(kc, g2c, dim3(512), CT_K2_LDS, 0, spec, B, F, w, imgc, cp, y, T, StitchArgs{0, 0, 0, 0}); }, iters));
        auto kcr = imdct4_ct_kernel<CODEC_RAW>;
        hipFuncSetAttribute((const void*)kcr, hipFuncAttributeMaxDynamicSharedMemorySize, (int)CT_K2_LDS);
        report("K2 ct RAW", time_ms([&] { hipLaunchKernelGGL(kcr, g2c, dim3(512), CT_K2_LDS, 0, spec, B, F, w, imgc, cpr, y2, T, StitchArgs{0, 0, 0, 0}); }, iters));
        auto kb2 = imdct4_bs_kernel<CODEC_ARCSINH>;
        hipFuncSetAttribute((const void*)kb2, hipFuncAttributeMaxDynamicSharedMemorySize, (int)BS_K2_LDS);
        const int nt2 = B * ((F + 31) / 32);
        hipLaunchKernelGGL(kb2, dim3(nt2 < 256 ? nt2 : 256), dim3(512), BS_K2_LDS, 0, spec, B, F, w, dimg, cp, y2, T);
        dd = max_diff(y, y2, (size_t)B * T, &amax);
        printf("   waveforms: max |ct - f32 pipe| = %.3e, max |y| = %.3e -> %.3e relative to the maximum\n", dd, amax, dd / amax);
        dd = max_diff(y, x, (size_t)B * T, &amax);
        printf("   round trip ct K1 (default dispatch) -> ct K2 vs input: %.3e (max |x| %.3e)\n", dd, amax);
    }
    printf("== through the C ABI\n");
    report("mg_mdct4_forward arcsinh", time_ms([&] { mg_mdct4_forward(x, B, T, 512, w, d, dimg_ct, CODEC_ARCSINH, 1000.f, -1.f, 1.f, -5.f, 5.f, 0, spec, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr); }, iters));
    report("mg_imdct4_forward arcsinh", time_ms([&] { mg_imdct4_forward(spec, B, F, 512, w, d, dimg_ct, CODEC_ARCSINH, 1000.f, -1.f, 1.f, -5.f, 5.f, nullptr, nullptr, y, T, 0, nullptr, nullptr); }, iters));
    {
        setenv("MG_MDCT_CT", "0", 1);
        mg_imdct4_forward(spec, B, F, 512, w, d, dimg_ct, CODEC_ARCSINH, 1000.f, -1.f, 1.f, -5.f, 5.f, nullptr, nullptr, y2, T, 0, nullptr, nullptr);
        unsetenv("MG_MDCT_CT");
        double amax;
        const double dd = max_diff(y, y2, (size_t)B * T, &amax);
        printf("   waveforms: max |default - generic kernel| = %.3e, max |y| = %.3e; round trip vs input:", dd, amax);
        const double rt = max_diff(y, x, (size_t)B * T, &amax);
        printf(" %.3e (max |x| %.3e)\n", rt, amax);
    }
    return 0;
}
