// Winograd F(2x2, 4x4) for the stride-1 4x4 convolutions of the PatchGAN discriminators (kw 4, padw 2:
// models/networks.py:649-670 -- the 256 -> 512 layer before the output layer is two thirds of a discriminator's
// FLOPs): 25 multiplies per 2x2 output tile and channel pair instead of 64, i.e. the MFMA work drops 2.56x.
// Interpolation points (0, 1, -1, 1/2, inf): float32 error ~2e-6 of the output scale at K = 16 * 256 (the direct
// kernel: ~2e-7), inside the 3e-5 parity bound of the convolution tests.
//
//   forward   Y  = A^T [ sum_ci (G g G^T) .* (B^T d B) ] A      d: 5x5 input patch at (2 ty - 2, 2 tx - 2), zero padded
//   dgrad     the transpose: dd_t = B [ (A dy A^T) x U ] B^T per tile, overlapping tiles (stride 2, size 5) summed in dx
//   wgrad     dg = G^T [ sum_tiles (B^T d B) .* (A dy A^T) ] G
// Geometry: H, W odd (OH = H + 1 = 2 TH, OW = W + 1 = 2 TW).  Layouts: V / M: [25][T][C], U: [25][Co][Ci],
// dd: [T][25][C].  The 25 GEMMs run as one batched launch of the implicit-GEMM kernels, as for F(2x2, 3x3).
#pragma once

namespace {

struct W4_AT { static constexpr float m[2][5] = {{1, 1, 1, 1, 0}, {0, 1, -1, 0.5f, 1}}; };
struct W4_A { static constexpr float m[5][2] = {{1, 0}, {1, 1}, {1, -1}, {1, 0.5f}, {0, 1}}; };
struct W4_G {
    static constexpr float m[5][4] = {{2, 0, 0, 0},
                                      {1, 1, 1, 1},
                                      {-1.0f / 3, 1.0f / 3, -1.0f / 3, 1.0f / 3},
                                      {-8.0f / 3, -4.0f / 3, -2.0f / 3, -1.0f / 3},
                                      {0, 0, 0, 1}};
};
struct W4_GT {
    static constexpr float m[4][5] = {{2, 1, -1.0f / 3, -8.0f / 3, 0},
                                      {0, 1, 1.0f / 3, -4.0f / 3, 0},
                                      {0, 1, -1.0f / 3, -2.0f / 3, 0},
                                      {0, 1, 1.0f / 3, -1.0f / 3, 1}};
};
struct W4_BT {
    static constexpr float m[5][5] = {{0.5f, -1, -0.5f, 1, 0},
                                      {0, -0.5f, 0.5f, 1, 0},
                                      {0, 0.5f, -1.5f, 1, 0},
                                      {0, -1, 0, 1, 0},
                                      {0, 0.5f, -1, -0.5f, 1}};
};
struct W4_B {
    static constexpr float m[5][5] = {{0.5f, 0, 0, 0, 0},
                                      {-1, -0.5f, 0.5f, -1, 0.5f},
                                      {-0.5f, 0.5f, -1.5f, 0, -1},
                                      {1, 1, 1, 1, -0.5f},
                                      {0, 0, 0, 0, 1}};
};

typedef float2 w4v;     // channel pair per thread: 25-element tiles stay in ~100 VGPRs
__device__ __forceinline__ w4v w4ld(const float* p) { return *reinterpret_cast<const float2*>(p); }
__device__ __forceinline__ void w4st(float* p, const w4v v) { *reinterpret_cast<float2*>(p) = v; }
__device__ __forceinline__ w4v w4zero() { return make_float2(0.f, 0.f); }

// acc (+)= l * v with the multiplications by 0 / +-1 resolved at compile time (loops around this are fully unrolled)
__device__ __forceinline__ void w4acc(w4v& acc, bool& first, const float l, const w4v v) {
    if (l == 0.0f) return;
    w4v t;
    if (l == 1.0f) t = v;
    else if (l == -1.0f) t = make_float2(-v.x, -v.y);
    else t = make_float2(l * v.x, l * v.y);
    acc = first ? t : make_float2(acc.x + t.x, acc.y + t.y);
    first = false;
}
// out[RO][NC] = L in[RI][NC]
template <class L, int RO, int RI, int NC>
__device__ __forceinline__ void w4rows(const w4v (&in)[RI][NC], w4v (&out)[RO][NC]) {
#pragma unroll
    for (int i = 0; i < RO; ++i)
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            w4v acc = w4zero();
            bool first = true;
#pragma unroll
            for (int k = 0; k < RI; ++k) w4acc(acc, first, L::m[i][k], in[k][c]);
            out[i][c] = acc;
        }
}
// out[NR][RO] = in[NR][RI] L^T
template <class L, int RO, int RI, int NR>
__device__ __forceinline__ void w4cols(const w4v (&in)[NR][RI], w4v (&out)[NR][RO]) {
#pragma unroll
    for (int r = 0; r < NR; ++r)
#pragma unroll
        for (int j = 0; j < RO; ++j) {
            w4v acc = w4zero();
            bool first = true;
#pragma unroll
            for (int k = 0; k < RI; ++k) w4acc(acc, first, L::m[j][k], in[r][k]);
            out[r][j] = acc;
        }
}

// U[25][Co][Ci] = G g G^T from OHWI weights [Co][4][4][Ci]
__global__ void wino4_weight_xform_kernel(const float* __restrict__ w, int Co, int Ci, float* __restrict__ U) {
    const int C2 = Ci / 2;
    const size_t total = (size_t)Co * C2;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c2 = (int)(i % C2), co = (int)(i / C2);
        w4v g[4][4], tmp[5][4], o[5][5];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int c = 0; c < 4; ++c) g[r][c] = w4ld(w + ((size_t)(co * 4 + r) * 4 + c) * Ci + 2 * c2);
        w4rows<W4_G, 5, 4, 4>(g, tmp);
        w4cols<W4_G, 5, 4, 5>(tmp, o);
#pragma unroll
        for (int r = 0; r < 5; ++r)
#pragma unroll
            for (int c = 0; c < 5; ++c) w4st(U + ((size_t)(r * 5 + c) * Co + co) * Ci + 2 * c2, o[r][c]);
    }
}

// V[25][T][C] = B^T d B, d = x[b][2 ty - 2 .. +4][2 tx - 2 .. +4] with zeros outside the image
__global__ void wino4_input_xform_kernel(const float* __restrict__ x, int B, int H, int W, int C, int TH, int TW,
                                         float* __restrict__ V) {
    const int C2 = C / 2;
    const size_t T = (size_t)B * TH * TW, total = T * C2;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c2 = (int)(i % C2);
        const size_t t = i / C2;
        const int tx = (int)(t % TW), ty = (int)((t / TW) % TH), b = (int)(t / ((size_t)TW * TH));
        w4v d[5][5], tmp[5][5];
#pragma unroll
        for (int r = 0; r < 5; ++r) {
            const int iy = 2 * ty - 2 + r;
            const bool oky = iy >= 0 && iy < H;
#pragma unroll
            for (int c = 0; c < 5; ++c) {
                const int ix = 2 * tx - 2 + c;
                const bool ok = oky && ix >= 0 && ix < W;
                d[r][c] = ok ? w4ld(x + ((size_t)(b * H + iy) * W + ix) * C + 2 * c2) : w4zero();
            }
        }
        w4rows<W4_BT, 5, 5, 5>(d, tmp);
#pragma unroll
        for (int r = 0; r < 5; ++r) {           // one output row at a time: keeps the live set at tmp + 5
            w4v in1[1][5], o1[1][5];
#pragma unroll
            for (int c = 0; c < 5; ++c) in1[0][c] = tmp[r][c];
            w4cols<W4_BT, 5, 5, 1>(in1, o1);
#pragma unroll
            for (int c = 0; c < 5; ++c) w4st(V + ((size_t)(r * 5 + c) * T + t) * C + 2 * c2, o1[0][c]);
        }
    }
}

// out[B][2TH][2TW][C] = act(A^T M A + bias)
__global__ void wino4_output_xform_kernel(const float* __restrict__ Mx, int B, int TH, int TW, int C,
                                          const float* __restrict__ bias, int act, float* __restrict__ out) {
    const int C2 = C / 2;
    const size_t T = (size_t)B * TH * TW, total = T * C2;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c2 = (int)(i % C2);
        const size_t t = i / C2;
        const int tx = (int)(t % TW), ty = (int)((t / TW) % TH), b = (int)(t / ((size_t)TW * TH));
        w4v m[5][5], tmp[2][5], y[2][2];
#pragma unroll
        for (int r = 0; r < 5; ++r)
#pragma unroll
            for (int c = 0; c < 5; ++c) m[r][c] = w4ld(Mx + ((size_t)(r * 5 + c) * T + t) * C + 2 * c2);
        w4rows<W4_AT, 2, 5, 5>(m, tmp);
        w4cols<W4_AT, 2, 5, 2>(tmp, y);
        const w4v bv = bias ? w4ld(bias + 2 * c2) : w4zero();
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                w4v v = make_float2(apply_act(y[r][c].x + bv.x, act), apply_act(y[r][c].y + bv.y, act));
                w4st(out + ((size_t)(b * 2 * TH + 2 * ty + r) * (2 * TW) + 2 * tx + c) * C + 2 * c2, v);
            }
    }
}

// Md[25][T][C] = A dy A^T, dy: [B][2TH][2TW][C]
__global__ void wino4_dy_xform_kernel(const float* __restrict__ dy, int B, int TH, int TW, int C, float* __restrict__ Md) {
    const int C2 = C / 2;
    const size_t T = (size_t)B * TH * TW, total = T * C2;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c2 = (int)(i % C2);
        const size_t t = i / C2;
        const int tx = (int)(t % TW), ty = (int)((t / TW) % TH), b = (int)(t / ((size_t)TW * TH));
        w4v d[2][2], tmp[5][2], o[5][5];
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int c = 0; c < 2; ++c)
                d[r][c] = w4ld(dy + ((size_t)(b * 2 * TH + 2 * ty + r) * (2 * TW) + 2 * tx + c) * C + 2 * c2);
        w4rows<W4_A, 5, 2, 2>(d, tmp);
        w4cols<W4_A, 5, 2, 5>(tmp, o);
#pragma unroll
        for (int r = 0; r < 5; ++r)
#pragma unroll
            for (int c = 0; c < 5; ++c) w4st(Md + ((size_t)(r * 5 + c) * T + t) * C + 2 * c2, o[r][c]);
    }
}

// dd[T][25][C] = B dV B^T, dV: [25][T][C]
__global__ void wino4_dd_xform_kernel(const float* __restrict__ dV, long long T, int C, float* __restrict__ dd) {
    const int C2 = C / 2;
    const size_t total = (size_t)T * C2;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c2 = (int)(i % C2);
        const size_t t = i / C2;
        w4v v[5][5], tmp[5][5];
#pragma unroll
        for (int r = 0; r < 5; ++r)
#pragma unroll
            for (int c = 0; c < 5; ++c) v[r][c] = w4ld(dV + ((size_t)(r * 5 + c) * T + t) * C + 2 * c2);
        w4rows<W4_B, 5, 5, 5>(v, tmp);
#pragma unroll
        for (int r = 0; r < 5; ++r) {
            w4v in1[1][5], o1[1][5];
#pragma unroll
            for (int c = 0; c < 5; ++c) in1[0][c] = tmp[r][c];
            w4cols<W4_B, 5, 5, 1>(in1, o1);
#pragma unroll
            for (int c = 0; c < 5; ++c) w4st(dd + ((size_t)t * 25 + r * 5 + c) * C + 2 * c2, o1[0][c]);
        }
    }
}

// dx[b][iy][ix] = sum over the tiles whose 5x5 patch covers (iy, ix): patch row r = iy + 2 - 2 ty in [0, 4]
__global__ void wino4_dx_gather_kernel(const float* __restrict__ dd, int B, int H, int W, int C, int TH, int TW,
                                       float* __restrict__ dx) {
    const int C4 = C / 4;
    const size_t total = (size_t)B * H * W * C4;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % C4);
        const size_t p = i / C4;
        const int ix = (int)(p % W), iy = (int)((p / W) % H), b = (int)(p / ((size_t)W * H));
        float4 acc = zero4();
        const int ty_hi = (iy + 2) >> 1, tx_hi = (ix + 2) >> 1;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const int ty = ty_hi - a, r = iy + 2 - 2 * ty;
            if (ty < 0 || ty >= TH || r > 4) continue;
#pragma unroll
            for (int e = 0; e < 3; ++e) {
                const int tx = tx_hi - e, c = ix + 2 - 2 * tx;
                if (tx < 0 || tx >= TW || c > 4) continue;
                const size_t t = ((size_t)b * TH + ty) * TW + tx;
                add4(acc, ld4(dd + (t * 25 + r * 5 + c) * C + 4 * c4));
            }
        }
        *reinterpret_cast<float4*>(dx + p * C + 4 * c4) = acc;
    }
}

// dw[Co][4][4][Ci] (+)= G^T dU G, dU: [25][Co][Ci]
__global__ void wino4_dweight_xform_kernel(const float* __restrict__ dU, int Co, int Ci, float* __restrict__ dw,
                                           int accumulate) {
    const int C2 = Ci / 2;
    const size_t total = (size_t)Co * C2;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c2 = (int)(i % C2), co = (int)(i / C2);
        w4v u[5][5], tmp[4][5], o[4][4];
#pragma unroll
        for (int r = 0; r < 5; ++r)
#pragma unroll
            for (int c = 0; c < 5; ++c) u[r][c] = w4ld(dU + ((size_t)(r * 5 + c) * Co + co) * Ci + 2 * c2);
        w4rows<W4_GT, 4, 5, 5>(u, tmp);
        w4cols<W4_GT, 4, 5, 4>(tmp, o);
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                float* q = dw + ((size_t)(co * 4 + r) * 4 + c) * Ci + 2 * c2;
                w4v v = o[r][c];
                if (accumulate) {
                    const w4v old = w4ld(q);
                    v = make_float2(old.x + v.x, old.y + v.y);
                }
                w4st(q, v);
            }
    }
}

}  // namespace
