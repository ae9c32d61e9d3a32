// Winograd F(4x4, 2x2) for the stride-2 4x4 convolutions of the PatchGAN discriminators (kw 4, stride 2, padw 2:
// models/networks.py:649-660).  A stride-2 4x4 convolution is a stride-1 2x2 convolution over the space-to-depth view
// of its input: with k = 2 j + a,  y[o] = sum_{a, j} X_a[o + j - 1] W_a[j],  X_a[i] = x[2 i + a],  W_a[j] = w[2 j + a],
// i.e. 4 Ci phase-channels (ay, ax, ci) and taps (jy, jx).  F(4x4, 2x2) then spends 25 multiplies per 4x4 output tile
// and phase-channel pair instead of 64 (2.56x fewer, less the ragged last tile row / column).  Same interpolation points
// as F(2x2, 4x4) (wino4.h: 0, 1, -1, 1/2, inf), so B^T / B are shared; float32 error ~5e-6 of the output scale.
//
//   forward   Y  = A^T [ sum_c' (G W G^T) .* (B^T X B) ] A     X: 5x5 phase patch at (4 ty - 1, 4 tx - 1), zero outside
//   dgrad     dd_t = B [ (A dy A^T) x U ] B^T per tile and phase-channel, overlapping tiles summed into dx
//   wgrad     dW = G^T [ sum_tiles (B^T X B) .* (A dy A^T) ] G, scattered back to the 4x4 taps
// The space-to-depth tensor never exists: the transforms address x / dx / w directly.  Layouts: V: [25][T][4 Ci],
// U: [25][Co][4 Ci], M / Md: [25][T][Co], dd: [T][25][4 Ci]; T = B * ceil(OH / 4) * ceil(OW / 4).
#pragma once

namespace {

struct W42_AT {
    static constexpr float m[4][5] = {{1, 1, 1, 1, 0}, {0, 1, -1, 0.5f, 0}, {0, 1, 1, 0.25f, 0}, {0, 1, -1, 0.125f, 1}};
};
struct W42_A {
    static constexpr float m[5][4] = {{1, 0, 0, 0}, {1, 1, 1, 1}, {1, -1, 1, -1}, {1, 0.5f, 0.25f, 0.125f}, {0, 0, 0, 1}};
};
struct W42_G {
    static constexpr float m[5][2] = {{2, 0}, {1, 1}, {-1.0f / 3, 1.0f / 3}, {-8.0f / 3, -4.0f / 3}, {0, 1}};
};
struct W42_GT {
    static constexpr float m[2][5] = {{2, 1, -1.0f / 3, -8.0f / 3, 0}, {0, 1, 1.0f / 3, -4.0f / 3, 1}};
};

// phase-channel c' = (ay * 2 + ax) * Ci + ci
struct Phase { int ay, ax, ci; };
__device__ __forceinline__ Phase w42_phase(int cp, int Ci) {
    const int a = cp / Ci;
    return {a >> 1, a & 1, cp - a * Ci};
}

// U[25][Co][4 Ci] = G W G^T, W[jy][jx] = w[co][2 jy + ay][2 jx + ax][ci]
__global__ void wino42_weight_xform_kernel(const float* __restrict__ w, int Co, int Ci, float* __restrict__ U) {
    const int K4 = 4 * Ci, C2 = K4 / 2;
    const size_t total = (size_t)Co * C2;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int cp = 2 * (int)(i % C2), co = (int)(i / C2);
        const Phase ph = w42_phase(cp, Ci);
        w4v g[2][2], tmp[5][2], o[5][5];
#pragma unroll
        for (int jy = 0; jy < 2; ++jy)
#pragma unroll
            for (int jx = 0; jx < 2; ++jx)
                g[jy][jx] = w4ld(w + ((size_t)(co * 4 + 2 * jy + ph.ay) * 4 + 2 * jx + ph.ax) * Ci + ph.ci);
        w4rows<W42_G, 5, 2, 2>(g, tmp);
        w4cols<W42_G, 5, 2, 5>(tmp, o);
#pragma unroll
        for (int r = 0; r < 5; ++r)
#pragma unroll
            for (int c = 0; c < 5; ++c) w4st(U + ((size_t)(r * 5 + c) * Co + co) * K4 + cp, o[r][c]);
    }
}

// V[25][T][4 Ci] = B^T X B, X[r][c] = x[b][2 (4 ty - 1 + r) + ay][2 (4 tx - 1 + c) + ax][ci] (0 outside the image)
__global__ void wino42_input_xform_kernel(const float* __restrict__ x, int B, int H, int W, int Ci, int TH, int TW,
                                          float* __restrict__ V) {
    const int K4 = 4 * Ci, C2 = K4 / 2;
    const size_t T = (size_t)B * TH * TW, total = T * C2;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int cp = 2 * (int)(i % C2);
        const size_t t = i / C2;
        const int tx = (int)(t % TW), ty = (int)((t / TW) % TH), b = (int)(t / ((size_t)TW * TH));
        const Phase ph = w42_phase(cp, Ci);
        w4v d[5][5], tmp[5][5];
#pragma unroll
        for (int r = 0; r < 5; ++r) {
            const int sy = 4 * ty - 1 + r, iy = 2 * sy + ph.ay;
            const bool oky = sy >= 0 && iy < H;
#pragma unroll
            for (int c = 0; c < 5; ++c) {
                const int sx = 4 * tx - 1 + c, ix = 2 * sx + ph.ax;
                const bool ok = oky && sx >= 0 && ix < W;
                d[r][c] = ok ? w4ld(x + ((size_t)(b * H + iy) * W + ix) * Ci + ph.ci) : w4zero();
            }
        }
        w4rows<W4_BT, 5, 5, 5>(d, tmp);
#pragma unroll
        for (int r = 0; r < 5; ++r) {
            w4v in1[1][5], o1[1][5];
#pragma unroll
            for (int c = 0; c < 5; ++c) in1[0][c] = tmp[r][c];
            w4cols<W4_BT, 5, 5, 1>(in1, o1);
#pragma unroll
            for (int c = 0; c < 5; ++c) w4st(V + ((size_t)(r * 5 + c) * T + t) * K4 + cp, o1[0][c]);
        }
    }
}

// out[B][OH][OW][C] = act(A^T M A + bias) on the 4x4 tile at (4 ty, 4 tx), clipped to the image
__global__ void wino42_output_xform_kernel(const float* __restrict__ Mx, int B, int OH, int OW, int TH, int TW, int C,
                                           const float* __restrict__ bias, int act, float* __restrict__ out) {
    const int C2 = C / 2;
    const size_t T = (size_t)B * TH * TW, total = T * C2;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c2 = (int)(i % C2);
        const size_t t = i / C2;
        const int tx = (int)(t % TW), ty = (int)((t / TW) % TH), b = (int)(t / ((size_t)TW * TH));
        w4v m[5][5], tmp[4][5], y[4][4];
#pragma unroll
        for (int r = 0; r < 5; ++r)
#pragma unroll
            for (int c = 0; c < 5; ++c) m[r][c] = w4ld(Mx + ((size_t)(r * 5 + c) * T + t) * C + 2 * c2);
        w4rows<W42_AT, 4, 5, 5>(m, tmp);
        w4cols<W42_AT, 4, 5, 4>(tmp, y);
        const w4v bv = bias ? w4ld(bias + 2 * c2) : w4zero();
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int oy = 4 * ty + r;
            if (oy >= OH) continue;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int ox = 4 * tx + c;
                if (ox >= OW) continue;
                const w4v v = make_float2(apply_act(y[r][c].x + bv.x, act), apply_act(y[r][c].y + bv.y, act));
                w4st(out + ((size_t)(b * OH + oy) * OW + ox) * C + 2 * c2, v);
            }
        }
    }
}

// Md[25][T][C] = A dy A^T, dy tile 4x4 at (4 ty, 4 tx), zeros past the image
__global__ void wino42_dy_xform_kernel(const float* __restrict__ dy, int B, int OH, int OW, int TH, int TW, int C,
                                       float* __restrict__ Md) {
    const int C2 = C / 2;
    const size_t T = (size_t)B * TH * TW, total = T * C2;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c2 = (int)(i % C2);
        const size_t t = i / C2;
        const int tx = (int)(t % TW), ty = (int)((t / TW) % TH), b = (int)(t / ((size_t)TW * TH));
        w4v d[4][4], tmp[5][4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int oy = 4 * ty + r;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int ox = 4 * tx + c;
                d[r][c] = (oy < OH && ox < OW) ? w4ld(dy + ((size_t)(b * OH + oy) * OW + ox) * C + 2 * c2) : w4zero();
            }
        }
        w4rows<W42_A, 5, 4, 4>(d, tmp);
#pragma unroll
        for (int r = 0; r < 5; ++r) {
            w4v in1[1][4], o1[1][5];
#pragma unroll
            for (int c = 0; c < 4; ++c) in1[0][c] = tmp[r][c];
            w4cols<W42_A, 5, 4, 1>(in1, o1);
#pragma unroll
            for (int c = 0; c < 5; ++c) w4st(Md + ((size_t)(r * 5 + c) * T + t) * C + 2 * c2, o1[0][c]);
        }
    }
}

// dx[b][iy][ix][ci] = sum of dd over the tiles whose phase patch holds (sy, sx) = (iy >> 1, ix >> 1): patch row
// r = sy + 1 - 4 ty in [0, 4] -- tile (sy + 1) >> 2, and the tile before it when r == 0 there (its row 4).
// dd: [T][25][4 Ci] as written by wino4_dd_xform_kernel with C = 4 Ci.
__global__ void wino42_dx_gather_kernel(const float* __restrict__ dd, int B, int H, int W, int Ci, int TH, int TW,
                                        float* __restrict__ dx) {
    const int C4 = Ci / 4, K4 = 4 * Ci;
    const size_t total = (size_t)B * H * W * C4;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % C4);
        const size_t p = i / C4;
        const int ix = (int)(p % W), iy = (int)((p / W) % H), b = (int)(p / ((size_t)W * H));
        const int sy = iy >> 1, sx = ix >> 1;
        const int cp = ((iy & 1) * 2 + (ix & 1)) * Ci + 4 * c4;
        float4 acc = zero4();
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const int ty = ((sy + 1) >> 2) - a, r = sy + 1 - 4 * ty;
            if (ty < 0 || ty >= TH || r > 4) continue;
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int tx = ((sx + 1) >> 2) - e, c = sx + 1 - 4 * tx;
                if (tx < 0 || tx >= TW || c > 4) continue;
                const size_t t = ((size_t)b * TH + ty) * TW + tx;
                add4(acc, ld4(dd + (t * 25 + r * 5 + c) * K4 + cp));
            }
        }
        *reinterpret_cast<float4*>(dx + p * Ci + 4 * c4) = acc;
    }
}

// dw[co][2 jy + ay][2 jx + ax][ci] (+)= (G^T dU G)[jy][jx], dU: [25][Co][4 Ci]
__global__ void wino42_dweight_xform_kernel(const float* __restrict__ dU, int Co, int Ci, float* __restrict__ dw,
                                            int accumulate) {
    const int K4 = 4 * Ci, C2 = K4 / 2;
    const size_t total = (size_t)Co * C2;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int cp = 2 * (int)(i % C2), co = (int)(i / C2);
        const Phase ph = w42_phase(cp, Ci);
        w4v u[5][5], tmp[2][5], o[2][2];
#pragma unroll
        for (int r = 0; r < 5; ++r)
#pragma unroll
            for (int c = 0; c < 5; ++c) u[r][c] = w4ld(dU + ((size_t)(r * 5 + c) * Co + co) * K4 + cp);
        w4rows<W42_GT, 2, 5, 5>(u, tmp);
        w4cols<W42_GT, 2, 5, 2>(tmp, o);
#pragma unroll
        for (int jy = 0; jy < 2; ++jy)
#pragma unroll
            for (int jx = 0; jx < 2; ++jx) {
                float* q = dw + ((size_t)(co * 4 + 2 * jy + ph.ay) * 4 + 2 * jx + ph.ax) * Ci + ph.ci;
                w4v v = o[jy][jx];
                if (accumulate) {
                    const w4v old = w4ld(q);
                    v = make_float2(old.x + v.x, old.y + v.y);
                }
                w4st(q, v);
            }
    }
}

}  // namespace
