// Codec arithmetic and small helpers shared by the transform kernels (mdct_ct.h; the retired table-stationary kernels under
// scripts/ubench/ use them too).  Included by mdct.hip after CodecParams / M are defined.
#pragma once

namespace {


constexpr float LOG10_2F = 0.30102999566398120f;
constexpr float INV_LN10F = 0.43429448190325176f;

// asinh(y) / ln(10) in ~14 issue slots (the libm asinhf + two IEEE divisions of the round-2 epilogue were ~190 VALU
// instructions per value: at 64 values per lane they cost more than the tile's MFMAs).  |y| >= 1/8: log2(|y| + sqrt(y^2 + 1))
// on v_sqrt_f32 / v_log_f32 (1 ulp each; <= 5e-7 relative in the result); |y| < 1/8: the odd series through y^7
// (truncation 2e-9 relative).  Against the reference's float32 torch.arcsinh: a few ulp, 1000x inside the 5e-4 bar of the
// normalised spectrogram (tests/test_mdct_gpu.py::test_fast_codec_math).
__device__ __forceinline__ float asinh_over_ln10(float y) {
    const float a = fabsf(y), a2 = a * a;
    const float s = a + __builtin_amdgcn_sqrtf(a2 + 1.0f);
    const float big = __builtin_amdgcn_logf(s) * LOG10_2F;
    const float p = fmaf(a2, fmaf(a2, fmaf(a2, -0.044642857142857144f, 0.075f), -0.16666666666666666f), 1.0f);
    const float small = a * p * INV_LN10F;
    return copysignf(a < 0.125f ? small : big, y);
}
// sinh(x) for the decoder: (e^x - e^-x) / 2 on v_exp_f32 above 1/4 (relative 3e-7), odd series through x^7 below.
__device__ __forceinline__ float sinh_fast(float x) {
    const float a = fabsf(x), a2 = a * a;
    const float e = __builtin_amdgcn_exp2f(a * 1.4426950408889634f);
    const float big = 0.5f * (e - __builtin_amdgcn_rcpf(e));
    const float p = fmaf(a2, fmaf(a2, fmaf(a2, 1.984126984126984e-4f, 8.333333333333333e-3f), 0.16666666666666666f), 1.0f);
    return copysignf(a < 0.25f ? a * p : big, x);
}
// x / d for a constant d, rd = fl(1 / d): one Newton step on the residual -- the correctly rounded quotient except in rare
// double-rounding cases, three instructions instead of v_div_scale / v_div_fmas / v_div_fixup.
__device__ __forceinline__ float div_const(float x, float d, float rd) {
    const float q = x * rd;
    return fmaf(fmaf(-q, d, x), rd, q);
}

struct BsCodec {      // CodecParams with the constants of the fixed-range normalisation folded once per kernel
    int mode;
    float gain, nr0, span, mn, d, rd;       // span = nr1 - nr0, d = mx - mn
};
__device__ __forceinline__ BsCodec bs_codec(const CodecParams& cp) {
    BsCodec c;
    c.mode = cp.mode; c.gain = cp.gain; c.nr0 = cp.nr0; c.span = cp.nr1 - cp.nr0; c.mn = cp.mn; c.d = cp.mx - cp.mn;
    c.rd = 1.0f / c.d;
    return c;
}
// l = log-domain value (for the statistics), returns the normalised value
__device__ __forceinline__ float bs_encode(float xv, const BsCodec& c, float& l) {
    if (c.mode == CODEC_RAW) { l = xv; return xv; }
    l = (c.mode == CODEC_ARCSINH) ? asinh_over_ln10(c.gain * xv) : xv;
    return div_const(l - c.mn, c.d, c.rd) * c.span + c.nr0;
}

__device__ __forceinline__ float4 bs_ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float4 bs_rev4(const float4 v) { return make_float4(v.w, v.z, v.y, v.x); }

typedef unsigned bs_v2u __attribute__((ext_vector_type(2)));
typedef unsigned bs_v4u __attribute__((ext_vector_type(4)));
constexpr unsigned BS_OOB = 0xffffffffu;          // buffer offset behind every num_records: the lane's access is dropped

}  // namespace
