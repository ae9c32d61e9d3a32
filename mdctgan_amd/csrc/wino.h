// Winograd F(2x2, 3x3) for the stride-1 3x3 convolutions (the 18 ResnetBlock convs are 81 % of the generator's
// FLOPs, models/networks.py:440, 456): 16 multiplies per 2x2 output tile and channel pair instead of 36, i.e. the
// MFMA work drops 2.25x while the result stays float32-exact up to the usual F(2,3) rounding (the reference's own
// cuDNN path uses the same algorithm family under cudnn.benchmark, train.py:25).
//
//   forward   Y  = A^T [ sum_ci (G g G^T) .* (B^T d B) ] A            d: 4x4 input patch (reflect / zero pad 1)
//   dgrad     the exact transpose of forward: dd_t = B [ (A dy A^T) x U ] B^T per tile, gathered back into dx (the
//             reflection / zero padding is the transpose of the forward gather); MG_WINO_DGRAD=padded selects the older
//             formulation dXp = A^T [ sum_co (G g' G^T) .* (B^T dy B) ] A over the (H+2)x(W+2) padded domain
//   wgrad     dg = G^T [ sum_tiles (B^T d B) .* (A dy A^T) ] G
// The element-wise products summed over channels / tiles are 16 independent GEMMs, run as ONE batched launch of
// the implicit-GEMM kernels with a 1x1 geometry (conv_igemm.hip); everything here is the HBM-bound transforms.
// Layouts: V / M: [16][T][C] (T = B * TH * TW tiles), U: [16][Co][Ci].
#pragma once

namespace {

// ---- 1-D building blocks (applied to rows then columns) -------------------------------------------------------
__device__ __forceinline__ void bt4(const float4 d0, const float4 d1, const float4 d2, const float4 d3, float4 (&o)[4]) {
    o[0] = make_float4(d0.x - d2.x, d0.y - d2.y, d0.z - d2.z, d0.w - d2.w);
    o[1] = make_float4(d1.x + d2.x, d1.y + d2.y, d1.z + d2.z, d1.w + d2.w);
    o[2] = make_float4(d2.x - d1.x, d2.y - d1.y, d2.z - d1.z, d2.w - d1.w);
    o[3] = make_float4(d1.x - d3.x, d1.y - d3.y, d1.z - d3.z, d1.w - d3.w);
}
__device__ __forceinline__ float4 f4add(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 f4sub(float4 a, float4 b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }
__device__ __forceinline__ float4 f4scale(float4 a, float s) { return make_float4(a.x * s, a.y * s, a.z * s, a.w * s); }
__device__ __forceinline__ float4 f4neg(float4 a) { return make_float4(-a.x, -a.y, -a.z, -a.w); }

// V[16][T][C] = B^T d B.  Patch origin (2*ty - org, 2*tx - org); outside [0,H)x[0,W): reflect (org must be 1) or 0.
__global__ void wino_input_xform_kernel(const float* __restrict__ x, int B, int H, int W, int C, int TH, int TW,
                                        int org, int reflect, float* __restrict__ V) {
    const int C4 = C / 4;
    const size_t T = (size_t)B * TH * TW, total = T * C4;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % C4);
        const size_t t = i / C4;
        const int tx = (int)(t % TW), ty = (int)((t / TW) % TH), b = (int)(t / ((size_t)TW * TH));
        float4 d[4][4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            int iy = 2 * ty - org + r;
            bool oky = true;
            if (reflect) iy = reflect_idx(iy, H); else oky = (iy >= 0 && iy < H);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                int ix = 2 * tx - org + c;
                bool ok = oky;
                if (reflect) ix = reflect_idx(ix, W); else ok = ok && (ix >= 0 && ix < W);
                d[r][c] = ok ? ld4(x + ((size_t)(b * H + iy) * W + ix) * C + 4 * c4) : zero4();
            }
        }
        float4 tmp[4][4];   // tmp[:, c] = B^T d[:, c]
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float4 o[4];
            bt4(d[0][c], d[1][c], d[2][c], d[3][c], o);
            tmp[0][c] = o[0]; tmp[1][c] = o[1]; tmp[2][c] = o[2]; tmp[3][c] = o[3];
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float4 o[4];
            bt4(tmp[r][0], tmp[r][1], tmp[r][2], tmp[r][3], o);
#pragma unroll
            for (int c = 0; c < 4; ++c)
                *reinterpret_cast<float4*>(V + ((size_t)(r * 4 + c) * T + t) * C + 4 * c4) = o[c];
        }
    }
}

// U[16][Co][Ci] = G g G^T from OHWI weights [Co][3][3][Ci]
__global__ void wino_weight_xform_kernel(const float* __restrict__ w, int Co, int Ci, float* __restrict__ U) {
    const int C4 = Ci / 4;
    const size_t total = (size_t)Co * C4;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % C4), co = (int)(i / C4);
        float4 g[3][3];
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c) g[r][c] = ld4(w + ((size_t)(co * 3 + r) * 3 + c) * Ci + 4 * c4);
        float4 tmp[4][3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            tmp[0][c] = g[0][c];
            tmp[1][c] = f4scale(f4add(f4add(g[0][c], g[1][c]), g[2][c]), 0.5f);
            tmp[2][c] = f4scale(f4add(f4sub(g[0][c], g[1][c]), g[2][c]), 0.5f);
            tmp[3][c] = g[2][c];
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float4 o[4];
            o[0] = tmp[r][0];
            o[1] = f4scale(f4add(f4add(tmp[r][0], tmp[r][1]), tmp[r][2]), 0.5f);
            o[2] = f4scale(f4add(f4sub(tmp[r][0], tmp[r][1]), tmp[r][2]), 0.5f);
            o[3] = tmp[r][2];
#pragma unroll
            for (int c = 0; c < 4; ++c)
                *reinterpret_cast<float4*>(U + ((size_t)(r * 4 + c) * Co + co) * Ci + 4 * c4) = o[c];
        }
    }
}

// out[B][2TH][2TW][C] = act(A^T M A + bias)
__global__ void wino_output_xform_kernel(const float* __restrict__ Mx, int B, int TH, int TW, int C,
                                         const float* __restrict__ bias, int act, float* __restrict__ out,
                                         int round_f16 = 0) {
    const int C4 = C / 4;
    const size_t T = (size_t)B * TH * TW, total = T * C4;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % C4);
        const size_t t = i / C4;
        const int tx = (int)(t % TW), ty = (int)((t / TW) % TH), b = (int)(t / ((size_t)TW * TH));
        float4 m[4][4];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int c = 0; c < 4; ++c) m[r][c] = ld4(Mx + ((size_t)(r * 4 + c) * T + t) * C + 4 * c4);
        float4 tmp[2][4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            tmp[0][c] = f4add(f4add(m[0][c], m[1][c]), m[2][c]);
            tmp[1][c] = f4sub(f4sub(m[1][c], m[2][c]), m[3][c]);
        }
        const float4 bv = bias ? ld4(bias + 4 * c4) : zero4();
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            float4 y0 = f4add(f4add(f4add(tmp[r][0], tmp[r][1]), tmp[r][2]), bv);
            float4 y1 = f4add(f4sub(f4sub(tmp[r][1], tmp[r][2]), tmp[r][3]), bv);
            y0.x = apply_act(y0.x, act); y0.y = apply_act(y0.y, act); y0.z = apply_act(y0.z, act); y0.w = apply_act(y0.w, act);
            y1.x = apply_act(y1.x, act); y1.y = apply_act(y1.y, act); y1.z = apply_act(y1.z, act); y1.w = apply_act(y1.w, act);
            if (round_f16) {     // MG_PRECISION_F16: the pass output is a float16 value
                y0.x = round_h(y0.x); y0.y = round_h(y0.y); y0.z = round_h(y0.z); y0.w = round_h(y0.w);
                y1.x = round_h(y1.x); y1.y = round_h(y1.y); y1.z = round_h(y1.z); y1.w = round_h(y1.w);
            }
            float* o = out + ((size_t)(b * 2 * TH + 2 * ty + r) * (2 * TW) + 2 * tx) * C + 4 * c4;
            *reinterpret_cast<float4*>(o) = y0;
            *reinterpret_cast<float4*>(o + C) = y1;
        }
    }
}


// ---- output transform fused with InstanceNorm2d(affine=False) (+ activation, + residual) ---------------------------------
// The trunk's maps are small (8x16 pixels at configs[1]): a sample's 32-channel slab is 32..160 tiles, so ONE workgroup can
// inverse-transform it, take the statistics over all its pixels and normalise without a second kernel or a second read --
// the InstanceNorm statistics live in the epilogue of the kernel that produces the convolution output.  Thread = (channel
// quad, tile lane): NT tiles of 2x2 pixels each, kept in registers between the statistics and the apply.  Sums in double,
// fixed-order LDS reduction (deterministic), same arithmetic as wino_output_xform_kernel followed by norm_slab_fwd_kernel.
struct WinoNorm {
    float eps;
    int act;
    const float* residual;     // [B][2TH][2TW][C] or nullptr
    float* y;                  // act((y_raw - mean) * rstd) + residual
    float* mean;               // [B][C]
    float* rstd;
    float* v_next;             // NEXT instances: V' = B^T y B of the OUTPUT y, [16][T][C] -- the input image of a following 3x3 stride-1
    int next_reflect;          //   pad-1 convolution over the same map (reflect / zero padding): ResnetBlock's next layer skips its transform
};
// NEXT: the slab's normalised plane (the 2TH x 2TW pixels x 32 channels the workgroup has just produced) is also staged in LDS and every
// thread builds the next convolution's Winograd input tiles from it -- the arithmetic of wino_input_xform_kernel on the same float32
// values, so V' is bit for bit what that kernel would compute from y: one launch and one read of y less per trunk layer.
template <int NT, bool NEXT = false>
__global__ __launch_bounds__(256) void wino_out_norm_kernel(const float* __restrict__ Mx, int B, int TH, int TW, int C,
                                                            const float* __restrict__ bias, WinoNorm nrm,
                                                            float* __restrict__ y_raw) {
    __shared__ double red[2][32][32];
    __shared__ __attribute__((aligned(16))) float plane[NEXT ? NT * 32 * 4 * 32 : 4];      // [pixel][32 channels]
    const int cq = threadIdx.x & 7, tl = threadIdx.x >> 3;
    const int b = blockIdx.y, c0 = blockIdx.x * 32 + 4 * cq;
    const int Ts = TH * TW;
    const size_t T = (size_t)B * Ts;
    float4 v[NT][4];
    double s1[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0};
    const float4 bv = bias ? ld4(bias + c0) : zero4();
#pragma unroll
    for (int i = 0; i < NT; ++i) {
        const int tile = tl + 32 * i;
        if (tile < Ts) {
            const size_t t = (size_t)b * Ts + tile;
            float4 m[4][4];
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int c = 0; c < 4; ++c) m[r][c] = ld4(Mx + ((size_t)(r * 4 + c) * T + t) * C + c0);
            float4 tmp[2][4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                tmp[0][c] = f4add(f4add(m[0][c], m[1][c]), m[2][c]);
                tmp[1][c] = f4sub(f4sub(m[1][c], m[2][c]), m[3][c]);
            }
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                v[i][2 * r] = f4add(f4add(f4add(tmp[r][0], tmp[r][1]), tmp[r][2]), bv);
                v[i][2 * r + 1] = f4add(f4sub(f4sub(tmp[r][1], tmp[r][2]), tmp[r][3]), bv);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                s1[0] += (double)v[i][q].x; s2[0] += (double)v[i][q].x * (double)v[i][q].x;
                s1[1] += (double)v[i][q].y; s2[1] += (double)v[i][q].y * (double)v[i][q].y;
                s1[2] += (double)v[i][q].z; s2[2] += (double)v[i][q].z * (double)v[i][q].z;
                s1[3] += (double)v[i][q].w; s2[3] += (double)v[i][q].w * (double)v[i][q].w;
            }
        }
    }
    // red[0/1][tile lane][channel] -> 8 partials per channel -> totals in every thread of the channel quad
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        red[0][tl][4 * cq + j] = s1[j];
        red[1][tl][4 * cq + j] = s2[j];
    }
    __syncthreads();
    {
        const int c = threadIdx.x & 31, part = threadIdx.x >> 5;
        double a = 0.0, bsum = 0.0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            a += red[0][4 * part + i][c];
            bsum += red[1][4 * part + i][c];
        }
        __syncthreads();
        red[0][part][c] = a;
        red[1][part][c] = bsum;
    }
    __syncthreads();
    const int HW = 4 * Ts;
    float mu[4], rs[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        double x1 = 0.0, x2 = 0.0;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            x1 += red[0][i][4 * cq + j];
            x2 += red[1][i][4 * cq + j];
        }
        const double mval = x1 / HW;
        double var = x2 / HW - mval * mval;
        if (var < 0.0) var = 0.0;
        mu[j] = (float)mval;
        rs[j] = (float)(1.0 / sqrt(var + (double)nrm.eps));
    }
    if (tl == 0) {
        *reinterpret_cast<float4*>(nrm.mean + (size_t)b * C + c0) = make_float4(mu[0], mu[1], mu[2], mu[3]);
        *reinterpret_cast<float4*>(nrm.rstd + (size_t)b * C + c0) = make_float4(rs[0], rs[1], rs[2], rs[3]);
    }
#pragma unroll
    for (int i = 0; i < NT; ++i) {
        const int tile = tl + 32 * i;
        if (tile < Ts) {
            const int ty = tile / TW, tx = tile - ty * TW;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const size_t off = ((size_t)(b * 2 * TH + 2 * ty + (q >> 1)) * (2 * TW) + 2 * tx + (q & 1)) * C + c0;
                if (y_raw) *reinterpret_cast<float4*>(y_raw + off) = v[i][q];      // (inference: nobody reads the raw output again)
                float4 o;
                o.x = apply_act((v[i][q].x - mu[0]) * rs[0], nrm.act);
                o.y = apply_act((v[i][q].y - mu[1]) * rs[1], nrm.act);
                o.z = apply_act((v[i][q].z - mu[2]) * rs[2], nrm.act);
                o.w = apply_act((v[i][q].w - mu[3]) * rs[3], nrm.act);
                if (nrm.residual) {
                    const float4 rr = ld4(nrm.residual + off);
                    o.x += rr.x; o.y += rr.y; o.z += rr.z; o.w += rr.w;
                }
                *reinterpret_cast<float4*>(nrm.y + off) = o;
                if (NEXT) *reinterpret_cast<float4*>(plane + ((2 * ty + (q >> 1)) * (2 * TW) + 2 * tx + (q & 1)) * 32 + 4 * cq) = o;
            }
        }
    }
    if (NEXT) {
        __syncthreads();
        const int H = 2 * TH, W = 2 * TW;
#pragma unroll
        for (int i = 0; i < NT; ++i) {
            const int tile = tl + 32 * i;
            if (tile < Ts) {
                const int ty = tile / TW, tx = tile - ty * TW;
                const size_t t = (size_t)b * Ts + tile;
                float4 d[4][4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    int iy = 2 * ty - 1 + r;
                    bool oky = true;
                    if (nrm.next_reflect) iy = reflect_idx(iy, H); else oky = (iy >= 0 && iy < H);
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        int ix = 2 * tx - 1 + c;
                        bool ok = oky;
                        if (nrm.next_reflect) ix = reflect_idx(ix, W); else ok = ok && (ix >= 0 && ix < W);
                        d[r][c] = ok ? *reinterpret_cast<const float4*>(plane + (iy * W + ix) * 32 + 4 * cq) : zero4();
                    }
                }
                float4 tmp[4][4];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    float4 o[4];
                    bt4(d[0][c], d[1][c], d[2][c], d[3][c], o);
                    tmp[0][c] = o[0]; tmp[1][c] = o[1]; tmp[2][c] = o[2]; tmp[3][c] = o[3];
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float4 o[4];
                    bt4(tmp[r][0], tmp[r][1], tmp[r][2], tmp[r][3], o);
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        *reinterpret_cast<float4*>(nrm.v_next + ((size_t)(r * 4 + c) * T + t) * C + c0) = o[c];
                }
            }
        }
    }
}
// the NEXT instances keep a slab's plane in LDS: up to 64 tiles (NT <= 2: 32 KiB beside the 16 KiB reduction buffer)
inline bool wino_out_norm_next_ok(int TH, int TW, int C) { return C % 32 == 0 && TH * TW <= 64; }

// ---- InstanceNorm backward fused with the data gradient's A dy A^T transform ------------------------------------------------
// The mirror of wino_out_norm_kernel: a workgroup owns a (sample, 32-channel) slab, a thread NT tiles of 2x2 pixels.  It
// reads the gradient at the norm's output and the raw convolution output, reduces the two InstanceNorm backward sums over
// the slab, forms the gradient at the convolution output in registers and writes its Winograd image Md = A dy A^T
// directly -- the convolution-output gradient itself never goes to HBM (the data and weight gradients both start from Md).
// Same arithmetic as norm_slab_bwd_kernel followed by wino_dy_xform_kernel.
template <int NT>
__global__ __launch_bounds__(256) void wino_norm_bwd_dy_kernel(const float* __restrict__ gy, const float* __restrict__ x,
                                                               const float* __restrict__ mean, const float* __restrict__ rstd,
                                                               int B, int TH, int TW, int C, int act, float* __restrict__ Md) {
    __shared__ double red[2][32][32];
    const int cq = threadIdx.x & 7, tl = threadIdx.x >> 3;
    const int b = blockIdx.y, c0 = blockIdx.x * 32 + 4 * cq;
    const int Ts = TH * TW, HW = 4 * Ts;
    const size_t T = (size_t)B * Ts;
    const float4 mu4 = ld4(mean + (size_t)b * C + c0), rs4 = ld4(rstd + (size_t)b * C + c0);
    const float mu[4] = {mu4.x, mu4.y, mu4.z, mu4.w}, rs[4] = {rs4.x, rs4.y, rs4.z, rs4.w};
    float xh[NT][4][4], gq[NT][4][4];          // [tile][pixel q = 2r + c][channel]
    double s1[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < NT; ++i) {
        const int tile = tl + 32 * i;
        if (tile < Ts) {
            const int ty = tile / TW, tx = tile - ty * TW;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const size_t off = ((size_t)(b * 2 * TH + 2 * ty + (q >> 1)) * (2 * TW) + 2 * tx + (q & 1)) * C + c0;
                float xv[4], gv[4];
                *reinterpret_cast<float4*>(xv) = ld4(x + off);
                *reinterpret_cast<float4*>(gv) = ld4(gy + off);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    xh[i][q][j] = (xv[j] - mu[j]) * rs[j];
                    const float d = act == MG_ACT_RELU ? (xh[i][q][j] > 0.0f ? 1.0f : 0.0f)
                                  : act == MG_ACT_LRELU02 ? (xh[i][q][j] > 0.0f ? 1.0f : 0.2f) : 1.0f;
                    gq[i][q][j] = gv[j] * d;
                    s1[j] += (double)gq[i][q][j];
                    s2[j] += (double)gq[i][q][j] * (double)xh[i][q][j];
                }
            }
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int j = 0; j < 4; ++j) xh[i][q][j] = gq[i][q][j] = 0.f;
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        red[0][tl][4 * cq + j] = s1[j];
        red[1][tl][4 * cq + j] = s2[j];
    }
    __syncthreads();
    {
        const int c = threadIdx.x & 31, part = threadIdx.x >> 5;
        double a = 0.0, bsum = 0.0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            a += red[0][4 * part + i][c];
            bsum += red[1][4 * part + i][c];
        }
        __syncthreads();
        red[0][part][c] = a;
        red[1][part][c] = bsum;
    }
    __syncthreads();
    float m1[4], m2[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        double x1 = 0.0, x2 = 0.0;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            x1 += red[0][i][4 * cq + j];
            x2 += red[1][i][4 * cq + j];
        }
        m1[j] = (float)(x1 / HW);
        m2[j] = (float)(x2 / HW);
    }
#pragma unroll
    for (int i = 0; i < NT; ++i) {
        const int tile = tl + 32 * i;
        if (tile < Ts) {
            float4 d[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float o[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = rs[j] * (gq[i][q][j] - m1[j] - xh[i][q][j] * m2[j]);
                d[q] = make_float4(o[0], o[1], o[2], o[3]);
            }
            const float4 y00 = d[0], y01 = d[1], y10 = d[2], y11 = d[3];
            float4 tmp[4][2] = {{y00, y01}, {f4add(y00, y10), f4add(y01, y11)}, {f4sub(y00, y10), f4sub(y01, y11)},
                                {f4neg(y10), f4neg(y11)}};
            const size_t t = (size_t)b * Ts + tile;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float4 o[4] = {tmp[r][0], f4add(tmp[r][0], tmp[r][1]), f4sub(tmp[r][0], tmp[r][1]), f4neg(tmp[r][1])};
#pragma unroll
                for (int c = 0; c < 4; ++c) *reinterpret_cast<float4*>(Md + ((size_t)(r * 4 + c) * T + t) * C + c0) = o[c];
            }
        }
    }
}
inline bool wino_out_norm_ok(int TH, int TW, int C) {
    constexpr bool off = false;
    return !off && C % 32 == 0 && TH * TW <= 160;
}

// ReflectionPad2d(1) backward: dX[i][j] = sum of the padded positions aliasing (i, j).  dXp: [B][H+2][W+2][C]
// addend (optional): dx = fold + addend (float32 add after the autocast rounding: a skip connection's gradient, mg_wino_tiles.add)
__global__ void wino_fold_reflect_kernel(const float* __restrict__ dxp, int B, int H, int W, int C,
                                         float* __restrict__ dx, int round_f16 = 0, const float* __restrict__ addend = nullptr) {
    const int C4 = C / 4;
    const size_t total = (size_t)B * H * W * C4;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % C4);
        size_t r = i / C4;
        const int ix = (int)(r % W); r /= W;
        const int iy = (int)(r % H);
        const int b = (int)(r / H);
        const int cy[3] = {iy + 1, (iy == 1) ? 0 : -1, (iy == H - 2) ? H + 1 : -1};
        const int cx[3] = {ix + 1, (ix == 1) ? 0 : -1, (ix == W - 2) ? W + 1 : -1};
        float4 s = zero4();
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            if (cy[a] < 0) continue;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                if (cx[c] < 0) continue;
                add4(s, ld4(dxp + ((size_t)(b * (H + 2) + cy[a]) * (W + 2) + cx[c]) * C + 4 * c4));
            }
        }
        if (round_f16) { s.x = round_h(s.x); s.y = round_h(s.y); s.z = round_h(s.z); s.w = round_h(s.w); }
        if (addend) add4(s, ld4(addend + i * 4));
        *reinterpret_cast<float4*>(dx + i * 4) = s;
    }
}

// ---- data gradient as the exact transpose of the forward pipeline -----------------------------------------------
// forward:  x --gather (reflect / zero pad)--> d_t --B^T . B--> V --GEMM--> M --A^T . A--> y
// backward: dy --A . A^T (wino_dy_xform)--> dM --GEMM with U--> dV --B . B^T (below)--> dd_t --scatter^T--> dx
// so the GEMMs run over the same T = B*H/2*W/2 tiles as the forward pass (no padded (H+2)x(W+2) domain).
// dd[T][16][C] = B dV B^T,  B = (B^T)^T: rows [1,0,0,0], [0,1,-1,1], [-1,1,1,0], [0,0,0,-1].
__device__ __forceinline__ void b4(const float4 v0, const float4 v1, const float4 v2, const float4 v3, float4 (&o)[4]) {
    o[0] = v0;
    o[1] = f4add(f4sub(v1, v2), v3);
    o[2] = f4sub(f4add(v1, v2), v0);
    o[3] = f4neg(v3);
}
__global__ void wino_dd_xform_kernel(const float* __restrict__ dV, long long T, int C, float* __restrict__ dd) {
    const int C4 = C / 4;
    const size_t total = (size_t)T * C4;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % C4);
        const size_t t = i / C4;
        float4 v[4][4];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int c = 0; c < 4; ++c) v[r][c] = ld4(dV + ((size_t)(r * 4 + c) * T + t) * C + 4 * c4);
        float4 tmp[4][4];   // tmp[:, c] = B v[:, c]
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float4 o[4];
            b4(v[0][c], v[1][c], v[2][c], v[3][c], o);
            tmp[0][c] = o[0]; tmp[1][c] = o[1]; tmp[2][c] = o[2]; tmp[3][c] = o[3];
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float4 o[4];
            b4(tmp[r][0], tmp[r][1], tmp[r][2], tmp[r][3], o);
#pragma unroll
            for (int c = 0; c < 4; ++c) *reinterpret_cast<float4*>(dd + ((size_t)t * 16 + r * 4 + c) * C + 4 * c4) = o[c];
        }
    }
}

// Transpose of the forward gather: pixel (iy, ix) collects dd_t[r][c] from every (tile, patch position) that read it --
// patch rows 2*ty - 1 + r == iy, plus (reflect) the padded rows -1 and H that alias rows 1 and H - 2; same for columns.
// At most 3 x 3 terms, summed in a fixed order.
__device__ __forceinline__ int wino_sources(int i, int n, int tiles, int reflect, int (&tt)[3], int (&rr)[3]) {
    int cnt = 0;
    const int cand[3] = {i, (reflect && i == 1) ? -1 : -1000, (reflect && i == n - 2) ? n : -1000};
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const int p = cand[a];
        if (p < -1) continue;
        const int t0 = (p + 1) >> 1;            // floor((p + 1) / 2), p >= -1
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int t = t0 - k, r = p + 1 - 2 * t;
            if (t >= 0 && t < tiles && r >= 0 && r < 4 && cnt < 3) { tt[cnt] = t; rr[cnt] = r; ++cnt; }
        }
    }
    return cnt;
}
__global__ void wino_dx_gather_kernel(const float* __restrict__ dd, int B, int H, int W, int C, int reflect,
                                      const float* __restrict__ bias, int act, float* __restrict__ dx, int round_f16) {
    const int C4 = C / 4, TH = H / 2, TW = W / 2;
    const size_t total = (size_t)B * H * W * C4;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % C4);
        size_t q = i / C4;
        const int ix = (int)(q % W); q /= W;
        const int iy = (int)(q % H);
        const int b = (int)(q / H);
        int ty[3], ry[3], tx[3], rx[3];
        const int ny = wino_sources(iy, H, TH, reflect, ty, ry), nx = wino_sources(ix, W, TW, reflect, tx, rx);
        float4 s = bias ? ld4(bias + 4 * c4) : zero4();
        float4 acc = zero4();
        for (int a = 0; a < ny; ++a)
            for (int c = 0; c < nx; ++c)
                add4(acc, ld4(dd + (((size_t)(b * TH + ty[a]) * TW + tx[c]) * 16 + ry[a] * 4 + rx[c]) * C + 4 * c4));
        add4(s, acc);
        s.x = apply_act(s.x, act); s.y = apply_act(s.y, act); s.z = apply_act(s.z, act); s.w = apply_act(s.w, act);
        if (round_f16) { s.x = round_h(s.x); s.y = round_h(s.y); s.z = round_h(s.z); s.w = round_h(s.w); }
        *reinterpret_cast<float4*>(dx + i * 4) = s;
    }
}


// dd transform + gather in one kernel for small maps: a workgroup owns a (sample, 32-channel) slab, computes the B dV B^T
// patches of all its tiles into LDS ([tile][16][32 channels], 2 KiB per tile) and gathers dx from there -- the same terms
// in the same order as wino_dd_xform_kernel + wino_dx_gather_kernel (bit-identical), without the dd round trip through HBM.
template <int NT>
__global__ __launch_bounds__(256) void wino_dd_gather_kernel(const float* __restrict__ dV, int B, int H, int W, int C, int reflect,
                                                             const float* __restrict__ bias, int act, float* __restrict__ dx,
                                                             const float* __restrict__ addend) {
    extern __shared__ __attribute__((aligned(16))) float4 wdg_lds[];     // [Ts][16][8]
    const int cq = threadIdx.x & 7, tl = threadIdx.x >> 3;
    const int b = blockIdx.y, c0 = blockIdx.x * 32 + 4 * cq;
    const int TH = H / 2, TW = W / 2, Ts = TH * TW;
    const size_t T = (size_t)B * Ts;
#pragma unroll
    for (int i = 0; i < NT; ++i) {
        const int tile = tl + 32 * i;
        if (tile < Ts) {
            const size_t t = (size_t)b * Ts + tile;
            float4 v[4][4];
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int c = 0; c < 4; ++c) v[r][c] = ld4(dV + ((size_t)(r * 4 + c) * T + t) * C + c0);
            float4 tmp[4][4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                float4 o[4];
                b4(v[0][c], v[1][c], v[2][c], v[3][c], o);
                tmp[0][c] = o[0]; tmp[1][c] = o[1]; tmp[2][c] = o[2]; tmp[3][c] = o[3];
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float4 o[4];
                b4(tmp[r][0], tmp[r][1], tmp[r][2], tmp[r][3], o);
#pragma unroll
                for (int c = 0; c < 4; ++c) wdg_lds[(tile * 16 + r * 4 + c) * 8 + cq] = o[c];
            }
        }
    }
    __syncthreads();
    const int HW = H * W;
    const float4 bv = bias ? ld4(bias + c0) : zero4();
    for (int p = tl; p < HW; p += 32) {
        const int iy = p / W, ix = p - iy * W;
        int ty[3], ry[3], tx[3], rx[3];
        const int ny = wino_sources(iy, H, TH, reflect, ty, ry), nx = wino_sources(ix, W, TW, reflect, tx, rx);
        float4 s = bv;
        float4 acc = zero4();
        for (int a = 0; a < ny; ++a)
            for (int c = 0; c < nx; ++c) add4(acc, wdg_lds[((ty[a] * TW + tx[c]) * 16 + ry[a] * 4 + rx[c]) * 8 + cq]);
        add4(s, acc);
        s.x = apply_act(s.x, act); s.y = apply_act(s.y, act); s.z = apply_act(s.z, act); s.w = apply_act(s.w, act);
        if (addend) add4(s, ld4(addend + ((size_t)b * HW + p) * C + c0));      // the skip connection's gradient (mg_wino_tiles.add)
        *reinterpret_cast<float4*>(dx + ((size_t)b * HW + p) * C + c0) = s;
    }
}
inline bool wino_dd_gather_ok(int H, int W, int C) {
    constexpr bool off = false;
    return !off && C % 32 == 0 && (H / 2) * (W / 2) <= 64;
}

// Mdy[16][T][C] = A dy A^T for the 2x2 tiles of dy [B][2TH][2TW][C]
__global__ void wino_dy_xform_kernel(const float* __restrict__ dy, int B, int TH, int TW, int C, float* __restrict__ Md) {
    const int C4 = C / 4;
    const size_t T = (size_t)B * TH * TW, total = T * C4;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % C4);
        const size_t t = i / C4;
        const int tx = (int)(t % TW), ty = (int)((t / TW) % TH), b = (int)(t / ((size_t)TW * TH));
        const float* p = dy + ((size_t)(b * 2 * TH + 2 * ty) * (2 * TW) + 2 * tx) * C + 4 * c4;
        const float4 y00 = ld4(p), y01 = ld4(p + C), y10 = ld4(p + (size_t)2 * TW * C), y11 = ld4(p + (size_t)2 * TW * C + C);
        // tmp = A dy (4x2), A = [[1,0],[1,1],[1,-1],[0,-1]]
        float4 tmp[4][2] = {{y00, y01}, {f4add(y00, y10), f4add(y01, y11)}, {f4sub(y00, y10), f4sub(y01, y11)},
                            {f4neg(y10), f4neg(y11)}};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float4 o[4] = {tmp[r][0], f4add(tmp[r][0], tmp[r][1]), f4sub(tmp[r][0], tmp[r][1]), f4neg(tmp[r][1])};
#pragma unroll
            for (int c = 0; c < 4; ++c)
                *reinterpret_cast<float4*>(Md + ((size_t)(r * 4 + c) * T + t) * C + 4 * c4) = o[c];
        }
    }
}

// dW [Co][3][3][Ci] (+)= G^T dU G,  dU: [16][Co][Ci]
__global__ void wino_dweight_xform_kernel(const float* __restrict__ dU, int Co, int Ci, float* __restrict__ dw,
                                          int accumulate) {
    const int C4 = Ci / 4;
    const size_t total = (size_t)Co * C4;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % C4), co = (int)(i / C4);
        float4 u[4][4];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int c = 0; c < 4; ++c) u[r][c] = ld4(dU + ((size_t)(r * 4 + c) * Co + co) * Ci + 4 * c4);
        float4 tmp[3][4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float4 h1 = f4scale(u[1][c], 0.5f), h2 = f4scale(u[2][c], 0.5f);
            tmp[0][c] = f4add(f4add(u[0][c], h1), h2);
            tmp[1][c] = f4sub(h1, h2);
            tmp[2][c] = f4add(f4add(h1, h2), u[3][c]);
        }
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const float4 h1 = f4scale(tmp[r][1], 0.5f), h2 = f4scale(tmp[r][2], 0.5f);
            float4 o[3] = {f4add(f4add(tmp[r][0], h1), h2), f4sub(h1, h2), f4add(f4add(h1, h2), tmp[r][3])};
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                float* p = dw + ((size_t)(co * 3 + r) * 3 + c) * Ci + 4 * c4;
                if (accumulate) add4(o[c], ld4(p));
                *reinterpret_cast<float4*>(p) = o[c];
            }
        }
    }
}

// Weight side of a trunk layer in ONE pass (round 3; single process, float32): the weight gradient's inverse transform
// dw = G^T dU G, the Adam update of (w, m, v) and the forward transform U = G w G^T of the UPDATED weights for the next
// iteration -- what wino_dweight_xform_kernel, adam_dev_kernel and wino_weight_xform_kernel do in three passes with the
// gradient and the weights making an HBM round trip each (467 MB -> 354 MB per 1024-channel layer).  The arithmetic is theirs,
// operation for operation (bit-identical: tests/test_conv_gpu.py::test_wgrad_adam_fusion_is_bit_identical).  The Adam clock
// has not ticked yet when this runs (backward precedes optimizer.step()): the bias corrections come from state[4], state[5],
// which the last tick left for exactly this purpose.
__global__ void wino_adam_kernel(const float* __restrict__ dU, int Co, int Ci, float* __restrict__ w, float* __restrict__ m,
                                 float* __restrict__ v, float* __restrict__ U, const double* __restrict__ state, float b1,
                                 float b2, float eps, float gscale) {
    const float step_size = (float)(state[1] / state[4]), bc2_sqrt = (float)state[5];
    const int C4 = Ci / 4;
    const size_t total = (size_t)Co * C4;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % C4), co = (int)(i / C4);
        float4 g[3][3];
        {   // wino_dweight_xform_kernel
            float4 u[4][4];
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int c = 0; c < 4; ++c) u[r][c] = ld4(dU + ((size_t)(r * 4 + c) * Co + co) * Ci + 4 * c4);
            float4 tmp[3][4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float4 h1 = f4scale(u[1][c], 0.5f), h2 = f4scale(u[2][c], 0.5f);
                tmp[0][c] = f4add(f4add(u[0][c], h1), h2);
                tmp[1][c] = f4sub(h1, h2);
                tmp[2][c] = f4add(f4add(h1, h2), u[3][c]);
            }
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const float4 h1 = f4scale(tmp[r][1], 0.5f), h2 = f4scale(tmp[r][2], 0.5f);
                g[r][0] = f4add(f4add(tmp[r][0], h1), h2);
                g[r][1] = f4sub(h1, h2);
                g[r][2] = f4add(f4add(h1, h2), tmp[r][3]);
            }
        }
        // adam_dev_kernel on the nine taps
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const size_t o = ((size_t)(co * 3 + r) * 3 + c) * Ci + 4 * c4;
                float4 pv = ld4(w + o), mv = ld4(m + o), vv = ld4(v + o);
                float* pp = reinterpret_cast<float*>(&pv); float* mm = reinterpret_cast<float*>(&mv);
                float* vp = reinterpret_cast<float*>(&vv); const float* gg = reinterpret_cast<const float*>(&g[r][c]);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float gi = gg[j] * gscale;
                    const float mi = mm[j] + (gi - mm[j]) * (1.0f - b1);
                    const float vi = vp[j] * b2 + (1.0f - b2) * gi * gi;
                    mm[j] = mi; vp[j] = vi;
                    pp[j] = pp[j] - step_size * (mi / (sqrtf(vi) / bc2_sqrt + eps));
                }
                *reinterpret_cast<float4*>(w + o) = pv;
                *reinterpret_cast<float4*>(m + o) = mv;
                *reinterpret_cast<float4*>(v + o) = vv;
                g[r][c] = pv;         // the updated weights
            }
        // wino_weight_xform_kernel
        float4 tmp[4][3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            tmp[0][c] = g[0][c];
            tmp[1][c] = f4scale(f4add(f4add(g[0][c], g[1][c]), g[2][c]), 0.5f);
            tmp[2][c] = f4scale(f4add(f4sub(g[0][c], g[1][c]), g[2][c]), 0.5f);
            tmp[3][c] = g[2][c];
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float4 o[4];
            o[0] = tmp[r][0];
            o[1] = f4scale(f4add(f4add(tmp[r][0], tmp[r][1]), tmp[r][2]), 0.5f);
            o[2] = f4scale(f4add(f4sub(tmp[r][0], tmp[r][1]), tmp[r][2]), 0.5f);
            o[3] = tmp[r][2];
#pragma unroll
            for (int c = 0; c < 4; ++c)
                *reinterpret_cast<float4*>(U + ((size_t)(r * 4 + c) * Co + co) * Ci + 4 * c4) = o[c];
        }
    }
}

inline unsigned wino_grid(size_t n) {
    size_t b = (n + 255) / 256;
    return (unsigned)(b > 8192 ? 8192 : (b < 1 ? 1 : b));
}

}  // namespace
