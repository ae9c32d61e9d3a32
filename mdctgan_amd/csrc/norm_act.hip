// K5 / K7 / K8 / K9 / K11 / K12: InstanceNorm (+activation, +residual) forward/backward, activation
// backward, pooling / upsampling, discriminator-input assembly, LSGAN / L1 losses, fused Adam.
// All HBM-bound: float4 (16 B / lane) coalesced NHWC streams, double-precision statistics.
//
// Replaces (reference): nn.InstanceNorm2d(affine=False) networks.py:26; ReLU / LeakyReLU(0.2) :306,
// :650-666; the residual add :462; nn.AvgPool2d(3,2,1,count_include_pad=False) :249-250, :525-526;
// interpolate(nearest, x2) :396; cat(lr, s, 2|s|+nr0) pix2pixHD_model.py:420-424; MSELoss / L1Loss
// networks.py:127-137, pix2pixHD_model.py:443-451; torch.optim.Adam pix2pixHD_model.py:350-364.
#include <cstdlib>
#include "common.h"
#include "mdctgan_hip.h"

namespace {

__device__ __forceinline__ float act_fwd(float v, int act) {
    if (act == MG_ACT_RELU) return fmaxf(v, 0.0f);
    if (act == MG_ACT_LRELU02) return v > 0.0f ? v : 0.2f * v;
    if (act == MG_ACT_TANH) return tanhf(v);
    return v;
}
// derivative given the PRE-activation value
__device__ __forceinline__ float act_grad_pre(float pre, int act) {
    if (act == MG_ACT_RELU) return pre > 0.0f ? 1.0f : 0.0f;
    if (act == MG_ACT_LRELU02) return pre > 0.0f ? 1.0f : 0.2f;
    return 1.0f;
}
// derivative given the POST-activation value
__device__ __forceinline__ float act_grad_post(float y, int act) {
    if (act == MG_ACT_RELU) return y > 0.0f ? 1.0f : 0.0f;
    if (act == MG_ACT_LRELU02) return y > 0.0f ? 1.0f : 0.2f;
    if (act == MG_ACT_TANH) return 1.0f - y * y;
    return 1.0f;
}

struct NormPlan { int cblocks, splits, rows_per_split; };
NormPlan norm_plan(int B, int HW, int C) {
    const int cb = (C + 63) / 64;
    int splits = (1024 + B * cb - 1) / (B * cb);
    if (splits > 32) splits = 32;          // the finalize kernel walks the splits serially: keep that loop short
    if (splits > (HW + 31) / 32) splits = (HW + 31) / 32;
    if (splits < 1) splits = 1;
    int rps = (HW + splits - 1) / splits;
    splits = (HW + rps - 1) / rps;
    return {cb, splits, rps};
}

// partial per-(b, c) sums over a slice of pixels.  MODE 0: (sum x, sum x^2).
// MODE 1 (backward): g = dy * act'(xhat); (sum g, sum g * xhat).
// VEC: 16 lanes x float4 cover the block's 64 channels (256 B per pixel), 16 pixel rows in flight per iteration.
// dy2 (optional, MODE 1): a second gradient of the same tensor (dy + dy2, one float32 addition -- what autograd's accumulation add
// computes when the tensor has two consumers: the discriminator features that also feed the feature-matching loss)
template <int MODE, bool VEC>
__global__ __launch_bounds__(256) void norm_partial_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                           const float* __restrict__ mean,
                                                           const float* __restrict__ rstd, int HW, int C,
                                                           int rows_per_split, int act, double* __restrict__ part,
                                                           const float* __restrict__ dy2 = nullptr) {
    constexpr int V = VEC ? 4 : 1, CL = 64 / V, RG = 256 / CL;
    __shared__ double red[2][RG][64];
    const int cl = threadIdx.x % CL, rg = threadIdx.x / CL;
    const int c = blockIdx.x * 64 + cl * V, b = blockIdx.y, sp = blockIdx.z;
    const int p0 = sp * rows_per_split, p1 = min(HW, p0 + rows_per_split);
    double s1[V], s2[V];
#pragma unroll
    for (int j = 0; j < V; ++j) s1[j] = s2[j] = 0.0;
    if (c < C) {
        const size_t base = (size_t)b * HW * C + c;
        float mu[V], rs[V];
#pragma unroll
        for (int j = 0; j < V; ++j) { mu[j] = 0.f; rs[j] = 0.f; }
        if (MODE == 1) {
#pragma unroll
            for (int j = 0; j < V; ++j) { mu[j] = mean[b * C + c + j]; rs[j] = rstd[b * C + c + j]; }
        }
        auto fetch = [&](int p, float* xv, float* gv) {
            if (VEC) {
                *reinterpret_cast<float4*>(xv) = *reinterpret_cast<const float4*>(x + base + (size_t)p * C);
                if (MODE == 1) *reinterpret_cast<float4*>(gv) = *reinterpret_cast<const float4*>(dy + base + (size_t)p * C);
                if (MODE == 1 && dy2) {
                    const float4 e = *reinterpret_cast<const float4*>(dy2 + base + (size_t)p * C);
                    gv[0] += e.x; gv[1] += e.y; gv[2] += e.z; gv[3] += e.w;
                }
            } else {
                xv[0] = x[base + (size_t)p * C];
                if (MODE == 1) { gv[0] = dy[base + (size_t)p * C]; if (dy2) gv[0] += dy2[base + (size_t)p * C]; }
            }
        };
        auto add = [&](const float* xv, const float* gv) {
#pragma unroll
            for (int j = 0; j < V; ++j) {
                if (MODE == 0) {
                    s1[j] += (double)xv[j];
                    s2[j] += (double)xv[j] * (double)xv[j];
                } else {
                    const float xh = (xv[j] - mu[j]) * rs[j];
                    const float gq = gv[j] * act_grad_pre(xh, act);
                    s1[j] += (double)gq;
                    s2[j] += (double)gq * (double)xh;
                }
            }
        };
        // four pixel rows per trip: their loads are in flight together (one load per thread and trip is latency-bound at ~4.5 TB/s
        // on the batch-64 maps); the sums take the rows in the same order as a row-per-trip loop -- the same bits
        constexpr int U = 4;
        int p = p0 + rg;
        for (; p + (U - 1) * RG < p1; p += U * RG) {
            float xv[U][V], gv[U][V];
#pragma unroll
            for (int u = 0; u < U; ++u) fetch(p + u * RG, xv[u], gv[u]);
#pragma unroll
            for (int u = 0; u < U; ++u) add(xv[u], gv[u]);
        }
        for (; p < p1; p += RG) {
            float xv[V], gv[V];
            fetch(p, xv, gv);
            add(xv, gv);
        }
    }
#pragma unroll
    for (int j = 0; j < V; ++j) {
        red[0][rg][cl * V + j] = s1[j];
        red[1][rg][cl * V + j] = s2[j];
    }
    __syncthreads();
    if (threadIdx.x < 64 && blockIdx.x * 64 + threadIdx.x < C) {
        double a = 0.0, q = 0.0;
#pragma unroll
        for (int i = 0; i < RG; ++i) { a += red[0][i][threadIdx.x]; q += red[1][i][threadIdx.x]; }
        const size_t o = (((size_t)b * gridDim.z + sp) * C + blockIdx.x * 64 + threadIdx.x) * 2;
        part[o] = a;
        part[o + 1] = q;
    }
}

// MODE 0: mean, rstd = 1/sqrt(var_biased + eps).  MODE 1: (sum g)/HW, (sum g*xhat)/HW.
template <int MODE>
__global__ void norm_finalize_kernel(const double* __restrict__ part, int B, int S, int C, int HW, float eps,
                                     float* __restrict__ o1, float* __restrict__ o2) {
    // eight lanes per (b, c): the S partials are a serial chain of dependent-latency loads otherwise (9 us for 512
    // outputs); fixed summation order -> deterministic
    constexpr int G = 8;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = t / G, sl = t % G;
    const bool live = i < B * C;
    const int b = live ? i / C : 0, c = live ? i - b * C : 0;
    double s1 = 0.0, s2 = 0.0;
    if (live)
        for (int s = sl; s < S; s += G) {
            const size_t o = (((size_t)b * S + s) * C + c) * 2;
            s1 += part[o];
            s2 += part[o + 1];
        }
#pragma unroll
    for (int d = 1; d < G; d <<= 1) {
        s1 += __shfl_xor(s1, d, G);
        s2 += __shfl_xor(s2, d, G);
    }
    if (!live || sl != 0) return;
    if (MODE == 0) {
        const double mu = s1 / HW;
        double var = s2 / HW - mu * mu;
        if (var < 0.0) var = 0.0;
        o1[i] = (float)mu;
        o2[i] = (float)(1.0 / sqrt(var + (double)eps));
    } else {
        o1[i] = (float)(s1 / HW);
        o2[i] = (float)(s2 / HW);
    }
}

typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));

template <bool VEC>
__global__ void norm_apply_fwd_kernel(const float* __restrict__ x, const float* __restrict__ mean,
                                      const float* __restrict__ rstd, const float* __restrict__ residual, int HW,
                                      int C, int act, float* __restrict__ y, size_t total, _Float16* __restrict__ y16 = nullptr) {
    // y16 (optional, autocast): the float16 copy the next convolution's operand staging would otherwise make in a pass of its own
    constexpr int V = VEC ? 4 : 1;
    const size_t per_b = (size_t)HW * C;
    for (size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * V; i < total;
         i += (size_t)gridDim.x * blockDim.x * V) {
        const int b = (int)(i / per_b);
        const int c = (int)(i % C);
        if (VEC) {
            const float4 v = *reinterpret_cast<const float4*>(x + i);
            const float4 mu = *reinterpret_cast<const float4*>(mean + b * C + c);
            const float4 rs = *reinterpret_cast<const float4*>(rstd + b * C + c);
            float4 o;
            o.x = act_fwd((v.x - mu.x) * rs.x, act);
            o.y = act_fwd((v.y - mu.y) * rs.y, act);
            o.z = act_fwd((v.z - mu.z) * rs.z, act);
            o.w = act_fwd((v.w - mu.w) * rs.w, act);
            if (residual) {
                const float4 r = *reinterpret_cast<const float4*>(residual + i);
                o.x += r.x; o.y += r.y; o.z += r.z; o.w += r.w;
            }
            *reinterpret_cast<float4*>(y + i) = o;
            if (y16) *reinterpret_cast<h16x4*>(y16 + i) = h16x4{(_Float16)o.x, (_Float16)o.y, (_Float16)o.z, (_Float16)o.w};
        } else {
            float o = act_fwd((x[i] - mean[b * C + c]) * rstd[b * C + c], act);
            if (residual) o += residual[i];
            y[i] = o;
            if (y16) y16[i] = (_Float16)o;
        }
    }
}

template <bool VEC>
__global__ void norm_apply_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                      const float* __restrict__ mean, const float* __restrict__ rstd,
                                      const float* __restrict__ m1, const float* __restrict__ m2, int HW, int C,
                                      int act, float* __restrict__ dx, size_t total, _Float16* __restrict__ dx16 = nullptr,
                                      const float* __restrict__ dy2 = nullptr) {
    constexpr int V = VEC ? 4 : 1;
    const size_t per_b = (size_t)HW * C;
    for (size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * V; i < total;
         i += (size_t)gridDim.x * blockDim.x * V) {
        const int b = (int)(i / per_b);
        const int c = (int)(i % C);
        float xv[4], gv[4], o[4];
        if (VEC) {
            *reinterpret_cast<float4*>(xv) = *reinterpret_cast<const float4*>(x + i);
            *reinterpret_cast<float4*>(gv) = *reinterpret_cast<const float4*>(dy + i);
            if (dy2) {
                const float4 e = *reinterpret_cast<const float4*>(dy2 + i);
                gv[0] += e.x; gv[1] += e.y; gv[2] += e.z; gv[3] += e.w;
            }
        } else {
            xv[0] = x[i];
            gv[0] = dy[i];
            if (dy2) gv[0] += dy2[i];
        }
#pragma unroll
        for (int j = 0; j < V; ++j) {
            const int sc = b * C + c + j;
            const float rs = rstd[sc];
            const float xh = (xv[j] - mean[sc]) * rs;
            const float gq = gv[j] * act_grad_pre(xh, act);
            o[j] = rs * (gq - m1[sc] - xh * m2[sc]);
        }
        if (VEC) {
            *reinterpret_cast<float4*>(dx + i) = *reinterpret_cast<float4*>(o);
            if (dx16) *reinterpret_cast<h16x4*>(dx16 + i) = h16x4{(_Float16)o[0], (_Float16)o[1], (_Float16)o[2], (_Float16)o[3]};
        } else {
            dx[i] = o[0];
            if (dx16) dx16[i] = (_Float16)o[0];
        }
    }
}

// The apply passes for C % 4 == 0 with the (sample, channel) bookkeeping hoisted out of the element loop: a block is 16 channel
// quads (64 channels, 256 B of a pixel) x 16 pixel rows, a thread keeps its four channels' statistics in registers and walks pixel
// rows -- norm_apply_{fwd,bwd}_kernel pay two 64-bit integer divisions and up to 16 scalar statistic loads per float4
// (measured 2.8 TB/s on the 64-channel full-resolution maps).  Same arithmetic per element: same bits.
// grid (ceil(C / 64), B, row splits), block 256.
template <bool BWD>
__global__ __launch_bounds__(256) void norm_apply_rows_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                              const float* __restrict__ mean, const float* __restrict__ rstd,
                                                              const float* __restrict__ m1, const float* __restrict__ m2,
                                                              const float* __restrict__ residual, int HW, int C, int rows_per_split,
                                                              int act, float* __restrict__ out, _Float16* __restrict__ out16,
                                                              const float* __restrict__ dy2 = nullptr) {
    const int cl = threadIdx.x & 15, rg = threadIdx.x >> 4;
    const int c = blockIdx.x * 64 + 4 * cl, b = blockIdx.y;
    if (c >= C) return;
    const float4 mu = *reinterpret_cast<const float4*>(mean + b * C + c);
    const float4 rs = *reinterpret_cast<const float4*>(rstd + b * C + c);
    float4 a1 = make_float4(0.f, 0.f, 0.f, 0.f), a2 = a1;
    if (BWD) {
        a1 = *reinterpret_cast<const float4*>(m1 + b * C + c);
        a2 = *reinterpret_cast<const float4*>(m2 + b * C + c);
    }
    const int p0 = blockIdx.z * rows_per_split, p1 = min(HW, p0 + rows_per_split);
    const size_t base = (size_t)b * HW * C + c;
    auto one = [&](size_t i, const float4 v, float4 g, const float4 r) {
        float4 o;
        if (!BWD) {
            o.x = act_fwd((v.x - mu.x) * rs.x, act);
            o.y = act_fwd((v.y - mu.y) * rs.y, act);
            o.z = act_fwd((v.z - mu.z) * rs.z, act);
            o.w = act_fwd((v.w - mu.w) * rs.w, act);
            if (residual) { o.x += r.x; o.y += r.y; o.z += r.z; o.w += r.w; }
        } else {
            if (dy2) { g.x += r.x; g.y += r.y; g.z += r.z; g.w += r.w; }
            const float xh0 = (v.x - mu.x) * rs.x, xh1 = (v.y - mu.y) * rs.y, xh2 = (v.z - mu.z) * rs.z, xh3 = (v.w - mu.w) * rs.w;
            const float g0 = g.x * act_grad_pre(xh0, act), g1 = g.y * act_grad_pre(xh1, act);
            const float g2 = g.z * act_grad_pre(xh2, act), g3 = g.w * act_grad_pre(xh3, act);
            o.x = rs.x * (g0 - a1.x - xh0 * a2.x);
            o.y = rs.y * (g1 - a1.y - xh1 * a2.y);
            o.z = rs.z * (g2 - a1.z - xh2 * a2.z);
            o.w = rs.w * (g3 - a1.w - xh3 * a2.w);
        }
        *reinterpret_cast<float4*>(out + i) = o;
        if (out16) *reinterpret_cast<h16x4*>(out16 + i) = h16x4{(_Float16)o.x, (_Float16)o.y, (_Float16)o.z, (_Float16)o.w};
    };
    // r: the residual (forward) / the second gradient dy2 (backward)
    const float* third = BWD ? dy2 : residual;
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    // four pixel rows per trip: their loads are in flight together (elementwise: the same bits as a row per trip)
    constexpr int U = 4;
    int p = p0 + rg;
    for (; p + (U - 1) * 16 < p1; p += U * 16) {
        float4 v[U], g[U], r[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t i = base + (size_t)(p + 16 * u) * C;
            v[u] = *reinterpret_cast<const float4*>(x + i);
            g[u] = BWD ? *reinterpret_cast<const float4*>(dy + i) : z4;
            r[u] = third ? *reinterpret_cast<const float4*>(third + i) : z4;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) one(base + (size_t)(p + 16 * u) * C, v[u], g[u], r[u]);
    }
    for (; p < p1; p += 16) {
        const size_t i = base + (size_t)p * C;
        one(i, *reinterpret_cast<const float4*>(x + i), BWD ? *reinterpret_cast<const float4*>(dy + i) : z4,
            third ? *reinterpret_cast<const float4*>(third + i) : z4);
    }
}
// row splits so that the launch has ~4096 workgroups (16 rows per pass and block: at least 16 rows per split)
inline dim3 norm_rows_grid(int B, int HW, int C, int* rows_per_split) {
    const int cb = (C + 63) / 64;
    int splits = (4096 + B * cb - 1) / (B * cb);
    if (splits > (HW + 15) / 16) splits = (HW + 15) / 16;
    if (splits < 1) splits = 1;
    if (splits > 65535) splits = 65535;
    *rows_per_split = (HW + splits - 1) / splits;
    return dim3((unsigned)cb, (unsigned)B, (unsigned)((HW + *rows_per_split - 1) / *rows_per_split));
}
inline bool norm_rows_on() { return getenv("MG_NO_NORM_ROWS") == nullptr; }      // (read per call: the parity test flips it)

// ------------------------------------------------------------------------------------------------------------
// Single-launch InstanceNorm for small maps (HW <= 32 * NP): one workgroup owns a (sample, 32-channel) slab
// (HW x 128 B, coalesced as 8 pixels x 128 B per wave load), keeps it in registers, reduces the statistics
// through LDS in double precision, and applies the normalisation without re-reading HBM.  Replaces the
// partial / finalize / apply sequence (3 launches, 2 reads) for the 8x16 ... 18x34 maps where those launches
// are pure latency.
// ------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void slab_reduce(double s1[4], double s2[4], double (*red)[32][32], int cq, int pl,
                                            double* o1, double* o2) {
    // red[0/1][pixel-lane][channel]; then 32 threads... every thread ends with the totals of its 4 channels
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        red[0][pl][4 * cq + j] = s1[j];
        red[1][pl][4 * cq + j] = s2[j];
    }
    __syncthreads();
    const int c = threadIdx.x & 31, part = threadIdx.x >> 5;   // 8 parts x 4 pixel-lanes
    double a = 0.0, b = 0.0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        a += red[0][4 * part + i][c];
        b += red[1][4 * part + i][c];
    }
    __syncthreads();
    red[0][part][c] = a;
    red[1][part][c] = b;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        double x = 0.0, y = 0.0;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            x += red[0][i][4 * cq + j];
            y += red[1][i][4 * cq + j];
        }
        o1[j] = x;
        o2[j] = y;
    }
}

// SLABS: x is not a tensor yet but the S split-K slabs of the convolution in front (part[z][B*HW][C], summed in slab order, + bias,
// rounded through float16 for an autocast layer: exactly splitk_epilogue_kernel's arithmetic) -- the kernel finishes the
// convolution, writes its raw output to x_out (nullable: nobody reads it under no_grad) and normalises, one launch instead of two.
struct NormSlabSrc { const float* part; const float* bias; float* x_out; size_t n; int S, round_f16; };
__device__ __forceinline__ float norm_round_h(float v) { return (float)(_Float16)v; }
template <int NP, bool SLABS = false>
__global__ __launch_bounds__(256) void norm_slab_fwd_kernel(const float* __restrict__ x, int HW, int C, float eps,
                                                            int act, const float* __restrict__ residual,
                                                            float* __restrict__ y, float* __restrict__ mean,
                                                            float* __restrict__ rstd, _Float16* __restrict__ y16 = nullptr,
                                                            NormSlabSrc src = NormSlabSrc{}) {
    __shared__ double red[2][32][32];
    const int cq = threadIdx.x & 7, pl = threadIdx.x >> 3;
    const int b = blockIdx.y, c0 = blockIdx.x * 32 + 4 * cq;
    const size_t base = (size_t)b * HW * C + c0;
    float4 v[NP];
    double s1[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0};
    float4 bias4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (SLABS && src.bias) bias4 = *reinterpret_cast<const float4*>(src.bias + c0);
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const int p = pl + 32 * i;
        v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (p < HW) {
            if (SLABS) {
                float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
                for (int z = 0; z < src.S; ++z) {
                    const float4 q = *reinterpret_cast<const float4*>(src.part + (size_t)z * src.n + base + (size_t)p * C);
                    a.x += q.x; a.y += q.y; a.z += q.z; a.w += q.w;
                }
                a.x += bias4.x; a.y += bias4.y; a.z += bias4.z; a.w += bias4.w;
                if (src.round_f16) { a.x = norm_round_h(a.x); a.y = norm_round_h(a.y); a.z = norm_round_h(a.z); a.w = norm_round_h(a.w); }
                if (src.x_out) *reinterpret_cast<float4*>(src.x_out + base + (size_t)p * C) = a;
                v[i] = a;
            } else {
                v[i] = *reinterpret_cast<const float4*>(x + base + (size_t)p * C);
            }
            s1[0] += (double)v[i].x; s2[0] += (double)v[i].x * (double)v[i].x;
            s1[1] += (double)v[i].y; s2[1] += (double)v[i].y * (double)v[i].y;
            s1[2] += (double)v[i].z; s2[2] += (double)v[i].z * (double)v[i].z;
            s1[3] += (double)v[i].w; s2[3] += (double)v[i].w * (double)v[i].w;
        }
    }
    double t1[4], t2[4];
    slab_reduce(s1, s2, red, cq, pl, t1, t2);
    float mu[4], rs[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const double m = t1[j] / HW;
        double var = t2[j] / HW - m * m;
        if (var < 0.0) var = 0.0;
        mu[j] = (float)m;
        rs[j] = (float)(1.0 / sqrt(var + (double)eps));
    }
    if (pl == 0) {
        *reinterpret_cast<float4*>(mean + b * C + c0) = make_float4(mu[0], mu[1], mu[2], mu[3]);
        *reinterpret_cast<float4*>(rstd + b * C + c0) = make_float4(rs[0], rs[1], rs[2], rs[3]);
    }
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const int p = pl + 32 * i;
        if (p < HW) {
            float4 o;
            o.x = act_fwd((v[i].x - mu[0]) * rs[0], act);
            o.y = act_fwd((v[i].y - mu[1]) * rs[1], act);
            o.z = act_fwd((v[i].z - mu[2]) * rs[2], act);
            o.w = act_fwd((v[i].w - mu[3]) * rs[3], act);
            if (residual) {
                const float4 r = *reinterpret_cast<const float4*>(residual + base + (size_t)p * C);
                o.x += r.x; o.y += r.y; o.z += r.z; o.w += r.w;
            }
            *reinterpret_cast<float4*>(y + base + (size_t)p * C) = o;
            if (y16) *reinterpret_cast<h16x4*>(y16 + base + (size_t)p * C) = h16x4{(_Float16)o.x, (_Float16)o.y, (_Float16)o.z, (_Float16)o.w};
        }
    }
}

template <int NP>
__global__ __launch_bounds__(256) void norm_slab_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                            const float* __restrict__ mean,
                                                            const float* __restrict__ rstd, int HW, int C, int act,
                                                            float* __restrict__ dx, _Float16* __restrict__ dx16 = nullptr,
                                                            const float* __restrict__ dy2 = nullptr) {
    __shared__ double red[2][32][32];
    const int cq = threadIdx.x & 7, pl = threadIdx.x >> 3;
    const int b = blockIdx.y, c0 = blockIdx.x * 32 + 4 * cq;
    const size_t base = (size_t)b * HW * C + c0;
    const float4 mu4 = *reinterpret_cast<const float4*>(mean + b * C + c0);
    const float4 rs4 = *reinterpret_cast<const float4*>(rstd + b * C + c0);
    const float mu[4] = {mu4.x, mu4.y, mu4.z, mu4.w}, rs[4] = {rs4.x, rs4.y, rs4.z, rs4.w};
    float xh[NP][4], gq[NP][4];
    double s1[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const int p = pl + 32 * i;
        if (p < HW) {
            float xv[4], gv[4];
            *reinterpret_cast<float4*>(xv) = *reinterpret_cast<const float4*>(x + base + (size_t)p * C);
            *reinterpret_cast<float4*>(gv) = *reinterpret_cast<const float4*>(dy + base + (size_t)p * C);
            if (dy2) {
                const float4 e = *reinterpret_cast<const float4*>(dy2 + base + (size_t)p * C);
                gv[0] += e.x; gv[1] += e.y; gv[2] += e.z; gv[3] += e.w;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                xh[i][j] = (xv[j] - mu[j]) * rs[j];
                gq[i][j] = gv[j] * act_grad_pre(xh[i][j], act);
                s1[j] += (double)gq[i][j];
                s2[j] += (double)gq[i][j] * (double)xh[i][j];
            }
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) xh[i][j] = gq[i][j] = 0.f;
        }
    }
    double t1[4], t2[4];
    slab_reduce(s1, s2, red, cq, pl, t1, t2);
    float m1[4], m2[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        m1[j] = (float)(t1[j] / HW);
        m2[j] = (float)(t2[j] / HW);
    }
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const int p = pl + 32 * i;
        if (p < HW) {
            float o[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = rs[j] * (gq[i][j] - m1[j] - xh[i][j] * m2[j]);
            *reinterpret_cast<float4*>(dx + base + (size_t)p * C) = *reinterpret_cast<float4*>(o);
            if (dx16) *reinterpret_cast<h16x4*>(dx16 + base + (size_t)p * C) = h16x4{(_Float16)o[0], (_Float16)o[1], (_Float16)o[2], (_Float16)o[3]};
        }
    }
}

inline int slab_np(int HW, int C) {     // 0: not eligible
    constexpr bool off = false;
    if (off || C % 32 != 0 || HW > 640) return 0;
    return HW <= 128 ? 4 : HW <= 256 ? 8 : HW <= 512 ? 16 : 20;
}

__global__ void act_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y, float* __restrict__ dx,
                               size_t n, int act, const float* __restrict__ dy2 = nullptr) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        dx[i] = (dy2 ? dy[i] + dy2[i] : dy[i]) * act_grad_post(y[i], act);
}
// nn.Sigmoid (the PatchGAN output under --no_lsgan, networks.py:676-677): y = 1 / (1 + exp(-x)); backward dx = dy y (1 - y)
__global__ void sigmoid_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        y[i] = 1.0f / (1.0f + expf(-x[i]));
}
__global__ void sigmoid_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y, float* __restrict__ dx, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        dx[i] = dy[i] * (y[i] * (1.0f - y[i]));
}
__global__ void add_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ o, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        o[i] = a[i] + b[i];
}

// AvgPool2d(3, stride=2, padding=1, count_include_pad=False)
__global__ void avgpool_fwd_kernel(const float* __restrict__ x, int B, int H, int W, int C, int OH, int OW,
                                   float* __restrict__ y) {
    const size_t total = (size_t)B * OH * OW * C;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        size_t r = i / C;
        const int ox = (int)(r % OW); r /= OW;
        const int oy = (int)(r % OH);
        const int b = (int)(r / OH);
        const int y0 = max(0, 2 * oy - 1), y1 = min(H - 1, 2 * oy + 1);
        const int x0 = max(0, 2 * ox - 1), x1 = min(W - 1, 2 * ox + 1);
        float s = 0.0f;
        for (int yy = y0; yy <= y1; ++yy)
            for (int xx = x0; xx <= x1; ++xx) s += x[((size_t)(b * H + yy) * W + xx) * C + c];
        y[i] = s / (float)((y1 - y0 + 1) * (x1 - x0 + 1));
    }
}
__global__ void avgpool_bwd_kernel(const float* __restrict__ dy, int B, int H, int W, int C, int OH, int OW,
                                   float* __restrict__ dx) {
    const size_t total = (size_t)B * H * W * C;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        size_t r = i / C;
        const int ix = (int)(r % W); r /= W;
        const int iy = (int)(r % H);
        const int b = (int)(r / H);
        float s = 0.0f;
        for (int oy = iy / 2; oy <= (iy + 1) / 2; ++oy) {
            if (oy >= OH) continue;
            const int ny = min(H - 1, 2 * oy + 1) - max(0, 2 * oy - 1) + 1;
            for (int ox = ix / 2; ox <= (ix + 1) / 2; ++ox) {
                if (ox >= OW) continue;
                const int nx = min(W - 1, 2 * ox + 1) - max(0, 2 * ox - 1) + 1;
                s += dy[((size_t)(b * OH + oy) * OW + ox) * C + c] / (float)(ny * nx);
            }
        }
        dx[i] = s;
    }
}

__global__ void upsample_fwd_kernel(const float* __restrict__ x, int B, int H, int W, int C, float* __restrict__ y) {
    const size_t total = (size_t)B * 2 * H * 2 * W * C;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        size_t r = i / C;
        const int ox = (int)(r % (2 * W)); r /= (2 * W);
        const int oy = (int)(r % (2 * H));
        const int b = (int)(r / (2 * H));
        y[i] = x[((size_t)(b * H + oy / 2) * W + ox / 2) * C + c];
    }
}
__global__ void upsample_bwd_kernel(const float* __restrict__ dy, int B, int H, int W, int C, float* __restrict__ dx) {
    const size_t total = (size_t)B * H * W * C;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        size_t r = i / C;
        const int ix = (int)(r % W); r /= W;
        const int iy = (int)(r % H);
        const int b = (int)(r / H);
        const size_t o = ((size_t)(b * 2 * H + 2 * iy) * 2 * W + 2 * ix) * C + c;
        dx[i] = dy[o] + dy[o + C] + dy[o + (size_t)2 * W * C] + dy[o + (size_t)2 * W * C + C];
    }
}

__global__ void dinput_fwd_kernel(const float* __restrict__ lr, const float* __restrict__ s, size_t n, float nr0,
                                  float* __restrict__ out) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float v = s[i];
        out[3 * i] = lr[i];
        out[3 * i + 1] = v;
        out[3 * i + 2] = fabsf(v) * 2.0f + nr0;
    }
}
__global__ void dinput_bwd_kernel(const float* __restrict__ g, const float* __restrict__ s, size_t n,
                                  float* __restrict__ ds) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float v = s[i];
        const float sg = (v > 0.0f) ? 1.0f : (v < 0.0f ? -1.0f : 0.0f);
        ds[i] = g[3 * i + 1] + 2.0f * sg * g[3 * i + 2];
    }
}
// torch.cat((a, b), dim=1) of NHWC tensors: out[p][0..Ca) = a[p], out[p][Ca..Ca+Cb) = b[p]; and its backward (a split)
__global__ void cat2_fwd_kernel(const float* __restrict__ a, int Ca, const float* __restrict__ b, int Cb, size_t n,
                                float* __restrict__ out) {
    const int Cc = Ca + Cb;
    const size_t total = n * Cc;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t p = i / Cc;
        const int c = (int)(i - p * Cc);
        out[i] = c < Ca ? a[p * Ca + c] : b[p * Cb + (c - Ca)];
    }
}
__global__ void cat2_bwd_kernel(const float* __restrict__ g, int Ca, int Cb, size_t n, float* __restrict__ ga,
                                float* __restrict__ gb) {
    const int Cc = Ca + Cb;
    const size_t total = n * Cc;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t p = i / Cc;
        const int c = (int)(i - p * Cc);
        if (c < Ca) { if (ga) ga[p * Ca + c] = g[i]; }
        else if (gb) gb[p * Cb + (c - Ca)] = g[i];
    }
}
__global__ void pair_fwd_kernel(const float* __restrict__ s, size_t n, float nr0, float* __restrict__ out) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float v = s[i];
        *reinterpret_cast<float2*>(out + 2 * i) = make_float2(v, fabsf(v) * 2.0f + nr0);
    }
}

// mean / unbiased std of a tensor from its {sum, sum of squares} (K1's statistics output): the float64 operations of
// (s0 / n).float(), ((s1 - s0 * s0 / n) / (n - 1)).clamp_min(0).sqrt().float() in one launch
__global__ void stats_finalize_kernel(const double* __restrict__ st, double n, float* __restrict__ out) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        const double s0 = st[0], s1 = st[1];
        out[0] = (float)(s0 / n);
        double var = (s1 - s0 * s0 / n) / (n - 1.0);
        if (!(var > 0.0)) var = var != var ? var : 0.0;       // clamp_min(0) keeps nan
        out[1] = (float)sqrt(var);
    }
}
// ---- losses: stage 1 block partials (double), stage 2 fixed-order sum -> deterministic ----
template <int KIND>   // 0: (p - t)^2   1: |a - b|
__global__ __launch_bounds__(256) void loss_partial_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                           float target, size_t n, double* __restrict__ part) {
    __shared__ double red[4];
    double s = 0.0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        if (KIND == 2) {      // nn.BCELoss against a constant label: log terms clamped at -100 like torch's
            const float pv = a[i];
            s += (double)((target - 1.0f) * fmaxf(logf(1.0f - pv), -100.0f) - target * fmaxf(logf(pv), -100.0f));
            continue;
        }
        const float d = a[i] - (KIND == 0 ? target : b[i]);
        s += (KIND == 0) ? (double)d * (double)d : (double)fabsf(d);
    }
    s = wave_sum_d(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
__global__ void loss_final_kernel(const double* __restrict__ part, int nb, double inv_n, float scale,
                                  float* __restrict__ loss, int accumulate) {
    // one wave, fixed lane-strided order then a fixed butterfly: deterministic
    double s = 0.0;
    for (int i = threadIdx.x; i < nb; i += 64) s += part[i];
    s = wave_sum_d(s);
    if (threadIdx.x == 0) {
        const float v = (float)(s * inv_n) * scale;
        loss[0] = accumulate ? loss[0] + v : v;
    }
}
template <int KIND>
__global__ void loss_grad_kernel(const float* __restrict__ a, const float* __restrict__ b, float target, size_t n,
                                 float coef, const float* __restrict__ go, float* __restrict__ grad) {
    const float gsc = coef * (go ? go[0] : 1.0f);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        if (KIND == 2) {      // binary_cross_entropy_backward: (p - t) / max((1 - p) p, 1e-12)
            const float pv = a[i];
            grad[i] = (pv - target) / fmaxf((1.0f - pv) * pv, 1e-12f) * gsc;
            continue;
        }
        const float d = a[i] - (KIND == 0 ? target : b[i]);
        if (KIND == 0) grad[i] = 2.0f * d * gsc;
        else grad[i] = (d > 0.0f ? 1.0f : (d < 0.0f ? -1.0f : 0.0f)) * gsc;
    }
}

// ---- the same losses over a LIST of tensors in one launch per stage (the feature-matching loss sums an L1 term per
// discriminator layer: 8-12 tensors -> 3 launches instead of 3 per tensor).  Every tensor keeps the block count and the
// grid-stride pattern of its stand-alone launch and the terms are added in list order, so the result is bit-identical to the
// accumulate-in-place sequence of single-tensor calls.
struct LossList {
    const float* a[MG_LOSS_MAX_ITEMS];
    const float* b[MG_LOSS_MAX_ITEMS];
    float* grad[MG_LOSS_MAX_ITEMS];
    unsigned long long n[MG_LOSS_MAX_ITEMS];
    unsigned long long zero_tail[MG_LOSS_MAX_ITEMS];   // backward: elements after grad[i][n) to clear (the stacked batch's other half)
    unsigned first_block[MG_LOSS_MAX_ITEMS + 1];       // forward: partial blocks of item i are [first_block[i], first_block[i + 1])
    float coef[MG_LOSS_MAX_ITEMS];                     // backward: scale / n, divided on the host like the single-tensor entry points do
    int count;
};
template <int KIND>
__global__ __launch_bounds__(256) void loss_partial_multi_kernel(LossList L, float target, double* __restrict__ part) {
    __shared__ double red[4];
    int it = 0;
    while (it + 1 < L.count && blockIdx.x >= L.first_block[it + 1]) ++it;
    const unsigned blk = blockIdx.x - L.first_block[it], nblk = L.first_block[it + 1] - L.first_block[it];
    const float* __restrict__ a = L.a[it];
    const float* __restrict__ b = L.b[it];
    const size_t n = (size_t)L.n[it];
    double s = 0.0;
    for (size_t i = (size_t)blk * blockDim.x + threadIdx.x; i < n; i += (size_t)nblk * blockDim.x) {
        if (KIND == 2) {
            const float pv = a[i];
            s += (double)((target - 1.0f) * fmaxf(logf(1.0f - pv), -100.0f) - target * fmaxf(logf(pv), -100.0f));
            continue;
        }
        const float d = a[i] - (KIND == 0 ? target : b[i]);
        s += (KIND == 0) ? (double)d * (double)d : (double)fabsf(d);
    }
    s = wave_sum_d(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
// one wave per item (same lane-strided order and butterfly as loss_final_kernel), then thread 0 adds the terms in list order
__global__ __launch_bounds__(64 * MG_LOSS_MAX_ITEMS) void loss_final_multi_kernel(LossList L, const double* __restrict__ part,
                                                                                  float scale, float* __restrict__ loss,
                                                                                  int accumulate) {
    __shared__ float term[MG_LOSS_MAX_ITEMS];
    const int it = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (it < L.count) {
        const int nb = (int)(L.first_block[it + 1] - L.first_block[it]);
        const double* p = part + L.first_block[it];
        double s = 0.0;
        for (int i = lane; i < nb; i += 64) s += p[i];
        s = wave_sum_d(s);
        if (lane == 0) term[it] = (float)(s * (1.0 / (double)L.n[it])) * scale;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float acc = accumulate ? loss[0] : 0.0f;
        for (int i = 0; i < L.count; ++i) acc = (i == 0 && !accumulate) ? term[0] : acc + term[i];
        loss[0] = acc;
    }
}
template <int KIND>
__global__ void loss_grad_multi_kernel(LossList L, float target, const float* __restrict__ go) {
    const int it = blockIdx.y;
    const float* __restrict__ a = L.a[it];
    const float* __restrict__ b = L.b[it];
    float* __restrict__ grad = L.grad[it];
    const size_t n = (size_t)L.n[it], nz = (size_t)L.zero_tail[it];
    const float gsc = L.coef[it] * (go ? go[0] : 1.0f);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n + nz; i += (size_t)gridDim.x * blockDim.x) {
        if (i >= n) { grad[i] = 0.0f; continue; }
        if (KIND == 2) {
            const float pv = a[i];
            grad[i] = (pv - target) / fmaxf((1.0f - pv) * pv, 1e-12f) * gsc;
            continue;
        }
        const float d = a[i] - (KIND == 0 ? target : b[i]);
        if (KIND == 0) grad[i] = 2.0f * d * gsc;
        else grad[i] = (d > 0.0f ? 1.0f : (d < 0.0f ? -1.0f : 0.0f)) * gsc;
    }
}

__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                            float* __restrict__ v, size_t n, float step_size, float b1, float b2, float eps,
                            float bc2_sqrt, float gscale) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float gi = g[i] * gscale;
        const float mi = m[i] + (gi - m[i]) * (1.0f - b1);           // lerp, as torch's exp_avg.lerp_(grad, 1-beta1)
        const float vi = v[i] * b2 + (1.0f - b2) * gi * gi;           // mul_(beta2).addcmul_(g, g, 1-beta2)
        m[i] = mi;
        v[i] = vi;
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        p[i] = p[i] - step_size * (mi / denom);
    }
}

// Device-resident optimiser clock (hipGraph-replayable: nothing step-dependent is baked into kernel arguments).
// state = {step, lr, step_size = lr / (1 - beta1^step), sqrt(1 - beta2^step)} as doubles.
__global__ void adam_tick_kernel(double* __restrict__ state, double b1, double b2) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        const double step = state[0] + 1.0;
        state[0] = step;
        state[2] = state[1] / (1.0 - pow(b1, step));
        state[3] = sqrt(1.0 - pow(b2, step));
        state[4] = 1.0 - pow(b1, step + 1.0);       // the NEXT step's bias-correction terms: kernels that update weights
        state[5] = sqrt(1.0 - pow(b2, step + 1.0)); // during backward (before this clock ticks) read these (mg_conv_wgrad_adam_w)
    }
}
__global__ void adam_prime_kernel(double* __restrict__ state, double b1, double b2) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        state[4] = 1.0 - pow(b1, state[0] + 1.0);
        state[5] = sqrt(1.0 - pow(b2, state[0] + 1.0));
    }
}
// ---- GradScaler on the device (train.py:65-70, 183-199).  scaler = {scale, growth_tracker, found_inf[slot]...} ----
__global__ __launch_bounds__(256) void scaler_check_kernel(const float* __restrict__ g, size_t n, float* __restrict__ flag) {
    bool bad = false;
    const size_t n4 = n / 4;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const float4 v = *reinterpret_cast<const float4*>(g + 4 * i);
        bad |= !(fabsf(v.x) <= 3.4028234663852886e38f) | !(fabsf(v.y) <= 3.4028234663852886e38f) |
               !(fabsf(v.z) <= 3.4028234663852886e38f) | !(fabsf(v.w) <= 3.4028234663852886e38f);
    }
    if (blockIdx.x == 0)
        for (size_t i = n4 * 4 + threadIdx.x; i < n; i += blockDim.x) bad |= !(fabsf(g[i]) <= 3.4028234663852886e38f);
    if (__any(bad) && (threadIdx.x & 63) == 0) *flag = 1.0f;     // every writer stores the same value
}
__global__ void scaler_update_kernel(float* __restrict__ sc, int slots, float growth, float backoff, int interval) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    float found = 0.0f;
    for (int i = 0; i < slots; ++i) { found += sc[2 + i]; sc[2 + i] = 0.0f; }
    if (found != 0.0f) {
        sc[0] *= backoff;
        sc[1] = 0.0f;
    } else {
        const float t = sc[1] + 1.0f;
        if (t >= (float)interval) { sc[0] *= growth; sc[1] = 0.0f; }
        else sc[1] = t;
    }
}
__global__ void adam_tick_amp_kernel(double* __restrict__ state, double b1, double b2, const float* __restrict__ flag) {
    if (threadIdx.x == 0 && blockIdx.x == 0 && *flag == 0.0f) {
        const double step = state[0] + 1.0;
        state[0] = step;
        state[2] = state[1] / (1.0 - pow(b1, step));
        state[3] = sqrt(1.0 - pow(b2, step));
        state[4] = 1.0 - pow(b1, step + 1.0);
        state[5] = sqrt(1.0 - pow(b2, step + 1.0));
    }
}
__global__ void adam_dev_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                float* __restrict__ v, size_t n, const double* __restrict__ state, float b1, float b2,
                                float eps, float gscale, const float* __restrict__ loss_scale,
                                const float* __restrict__ found_inf, _Float16* __restrict__ p16 = nullptr) {
    if (found_inf && *found_inf != 0.0f) return;          // GradScaler.step: skip the update on inf / nan gradients
    if (loss_scale) gscale = gscale / *loss_scale;        // unscale
    const float step_size = (float)state[2], bc2_sqrt = (float)state[3];
    const size_t n4 = n / 4;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        float4 pv = *reinterpret_cast<float4*>(p + 4 * i), mv = *reinterpret_cast<float4*>(m + 4 * i),
               vv = *reinterpret_cast<float4*>(v + 4 * i);
        const float4 gv = *reinterpret_cast<const float4*>(g + 4 * i);
        float* pp = reinterpret_cast<float*>(&pv); float* mm = reinterpret_cast<float*>(&mv);
        float* vp = reinterpret_cast<float*>(&vv); const float* gg = reinterpret_cast<const float*>(&gv);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float gi = gg[j] * gscale;
            const float mi = mm[j] + (gi - mm[j]) * (1.0f - b1);
            const float vi = vp[j] * b2 + (1.0f - b2) * gi * gi;
            mm[j] = mi; vp[j] = vi;
            pp[j] = pp[j] - step_size * (mi / (sqrtf(vi) / bc2_sqrt + eps));
        }
        *reinterpret_cast<float4*>(p + 4 * i) = pv;
        *reinterpret_cast<float4*>(m + 4 * i) = mv;
        *reinterpret_cast<float4*>(v + 4 * i) = vv;
        if (p16) {          // float16 shadow of the parameters (the autocast convolutions' weight operand): 2 more bytes written
            typedef _Float16 h4 __attribute__((ext_vector_type(4)));
            const h4 hv = {(_Float16)pp[0], (_Float16)pp[1], (_Float16)pp[2], (_Float16)pp[3]};
            *reinterpret_cast<h4*>(p16 + 4 * i) = hv;
        }
    }
    if (blockIdx.x == 0)
        for (size_t i = n4 * 4 + threadIdx.x; i < n; i += blockDim.x) {
            const float gi = g[i] * gscale;
            const float mi = m[i] + (gi - m[i]) * (1.0f - b1);
            const float vi = v[i] * b2 + (1.0f - b2) * gi * gi;
            m[i] = mi; v[i] = vi;
            p[i] = p[i] - step_size * (mi / (sqrtf(vi) / bc2_sqrt + eps));
            if (p16) p16[i] = (_Float16)p[i];
        }
}

// ---- --fp16 optimiser passes over a SEGMENTED arena (round 6) ----------------------------------------------------------
// Under torch.autocast a convolution's weight / bias gradient is a float16 tensor (the cast's backward widens it into the
// float32 .grad, train.py:161-164, 183-199): its values are float16-rounded and it overflows at 65504.  Segments say how a run of
// the arena carries that: MG_GRAD_F32 (BatchNorm / position-embedding parameters: float32 in the reference too), MG_GRAD_AUTOCAST
// (float32 storage, rounded through float16 where it is consumed -- here), MG_GRAD_F16 (stored as float16 in g16 by its own
// weight-gradient kernel: mg_conv_wgrad_h16).  Offsets / lengths are multiples of 8 elements.
typedef _Float16 seg_h4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float seg_round_h(float v) { return (float)(_Float16)v; }
template <int MODE>
__device__ __forceinline__ void adam_seg_span(float* __restrict__ p, const float* __restrict__ g, const _Float16* __restrict__ g16,
                                              float* __restrict__ m, float* __restrict__ v, _Float16* __restrict__ p16, size_t n4,
                                              float step_size, float bc2_sqrt, float b1, float b2, float eps, float gscale) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        float4 pv = *reinterpret_cast<float4*>(p + 4 * i), mv = *reinterpret_cast<float4*>(m + 4 * i),
               vv = *reinterpret_cast<float4*>(v + 4 * i);
        float gg[4];
        if (MODE == MG_GRAD_F16) {
            const seg_h4 h = *reinterpret_cast<const seg_h4*>(g16 + 4 * i);
            gg[0] = (float)h[0]; gg[1] = (float)h[1]; gg[2] = (float)h[2]; gg[3] = (float)h[3];
        } else {
            const float4 gv = *reinterpret_cast<const float4*>(g + 4 * i);
            gg[0] = gv.x; gg[1] = gv.y; gg[2] = gv.z; gg[3] = gv.w;
            if (MODE == MG_GRAD_AUTOCAST) {
#pragma unroll
                for (int j = 0; j < 4; ++j) gg[j] = seg_round_h(gg[j]);
            }
        }
        float* pp = reinterpret_cast<float*>(&pv); float* mm = reinterpret_cast<float*>(&mv); float* vp = reinterpret_cast<float*>(&vv);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float gi = gg[j] * gscale;
            const float mi = mm[j] + (gi - mm[j]) * (1.0f - b1);
            const float vi = vp[j] * b2 + (1.0f - b2) * gi * gi;
            mm[j] = mi; vp[j] = vi;
            pp[j] = pp[j] - step_size * (mi / (sqrtf(vi) / bc2_sqrt + eps));
        }
        *reinterpret_cast<float4*>(p + 4 * i) = pv;
        *reinterpret_cast<float4*>(m + 4 * i) = mv;
        *reinterpret_cast<float4*>(v + 4 * i) = vv;
        if (p16) {
            const seg_h4 hv = {(_Float16)pp[0], (_Float16)pp[1], (_Float16)pp[2], (_Float16)pp[3]};
            *reinterpret_cast<seg_h4*>(p16 + 4 * i) = hv;
        }
    }
}
__global__ void adam_seg_kernel(float* __restrict__ p, const float* __restrict__ g, const _Float16* __restrict__ g16,
                                float* __restrict__ m, float* __restrict__ v, _Float16* __restrict__ p16,
                                const mg_grad_seg* __restrict__ segs, int nsegs, const double* __restrict__ state, float b1,
                                float b2, float eps, float gscale, const float* __restrict__ loss_scale,
                                const float* __restrict__ found_inf) {
    if (found_inf && *found_inf != 0.0f) return;          // GradScaler.step: skip the update on inf / nan gradients
    if (loss_scale) gscale = gscale / *loss_scale;        // unscale
    const float step_size = (float)state[2], bc2_sqrt = (float)state[3];
    for (int s = 0; s < nsegs; ++s) {
        const size_t off = (size_t)segs[s].off, n4 = (size_t)segs[s].n / 4;
        const int mode = segs[s].mode;
        _Float16* h = p16 ? p16 + off : nullptr;
        if (mode == MG_GRAD_F16)
            adam_seg_span<MG_GRAD_F16>(p + off, g + off, g16 + off, m + off, v + off, h, n4, step_size, bc2_sqrt, b1, b2, eps, gscale);
        else if (mode == MG_GRAD_AUTOCAST)
            adam_seg_span<MG_GRAD_AUTOCAST>(p + off, g + off, g16, m + off, v + off, h, n4, step_size, bc2_sqrt, b1, b2, eps, gscale);
        else
            adam_seg_span<MG_GRAD_F32>(p + off, g + off, g16, m + off, v + off, h, n4, step_size, bc2_sqrt, b1, b2, eps, gscale);
    }
}
// GradScaler's inf / nan test per segment: float32 -> finite; autocast -> a finite float16 after rounding (|v| < 65520); float16
// storage -> exponent bits.  Segments with skip_check != 0 were checked by the kernel that produced them.
__global__ __launch_bounds__(256) void scaler_check_seg_kernel(const float* __restrict__ g, const _Float16* __restrict__ g16,
                                                               const mg_grad_seg* __restrict__ segs, int nsegs,
                                                               float* __restrict__ flag) {
    bool bad = false;
    for (int s = 0; s < nsegs; ++s) {
        if (segs[s].skip_check) continue;
        const size_t off = (size_t)segs[s].off, n = (size_t)segs[s].n;
        const int mode = segs[s].mode;
        if (mode == MG_GRAD_F16) {
            const size_t n8 = n / 8;
            for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (size_t)gridDim.x * blockDim.x) {
                const uint4 q = *reinterpret_cast<const uint4*>(g16 + off + 8 * i);
                const unsigned w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
                for (int j = 0; j < 4; ++j) bad |= ((w[j] & 0x7c00u) == 0x7c00u) | ((w[j] & 0x7c000000u) == 0x7c000000u);
            }
        } else {
            const float lim = mode == MG_GRAD_AUTOCAST ? 65520.0f : 3.4028234663852886e38f;
            const size_t n4 = n / 4;
            for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
                const float4 q = *reinterpret_cast<const float4*>(g + off + 4 * i);
                if (mode == MG_GRAD_AUTOCAST)
                    bad |= !(fabsf(q.x) < lim) | !(fabsf(q.y) < lim) | !(fabsf(q.z) < lim) | !(fabsf(q.w) < lim);
                else
                    bad |= !(fabsf(q.x) <= lim) | !(fabsf(q.y) <= lim) | !(fabsf(q.z) <= lim) | !(fabsf(q.w) <= lim);
            }
        }
    }
    if (__any(bad) && (threadIdx.x & 63) == 0) *flag = 1.0f;     // every writer stores the same value
}

inline unsigned grid_for(size_t n, int per_thread = 1) {
    size_t b = (n + 256 * (size_t)per_thread - 1) / (256 * (size_t)per_thread);
    if (b > 4096) b = 4096;
    if (b < 1) b = 1;
    return (unsigned)b;
}
inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

extern "C" {

size_t mg_instnorm_workspace(int B, int HW, int C) {
    const NormPlan p = norm_plan(B, HW, C);
    return ((size_t)B * p.splits * C * 2) * sizeof(double) + (size_t)2 * B * C * sizeof(float) + 256;
}

int mg_instnorm_fwd(const float* x, int B, int HW, int C, float eps, int act, const float* residual, float* y,
                    float* mean, float* rstd, void* workspace, size_t workspace_bytes, void* stream) {
    return mg_instnorm_fwd_h(x, B, HW, C, eps, act, residual, y, mean, rstd, workspace, workspace_bytes, stream, nullptr);
}
int mg_instnorm_fwd_h(const float* x, int B, int HW, int C, float eps, int act, const float* residual, float* y,
                      float* mean, float* rstd, void* workspace, size_t workspace_bytes, void* stream, void* y16v) {
    _Float16* y16 = (_Float16*)y16v;
    if (y16 && ((reinterpret_cast<uintptr_t>(y16) & 7) != 0 || C % 4 != 0)) return MG_ERR_ARG;
    if (!x || !y || !mean || !rstd || !workspace || B <= 0 || HW <= 0 || C <= 0) return MG_ERR_ARG;
    if (workspace_bytes < mg_instnorm_workspace(B, HW, C)) return MG_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (const int np = slab_np(HW, C); np && al16(x) && al16(y) && al16(mean) && al16(rstd) && (!residual || al16(residual))) {
        const dim3 grid(C / 32, B);
#define MG_SLAB_FWD(NP_) hipLaunchKernelGGL((norm_slab_fwd_kernel<NP_, false>), grid, dim3(256), 0, st, x, HW, C, eps, act, residual, y, mean, rstd, y16, NormSlabSrc{})
        if (np == 4) MG_SLAB_FWD(4); else if (np == 8) MG_SLAB_FWD(8); else if (np == 16) MG_SLAB_FWD(16); else MG_SLAB_FWD(20);
#undef MG_SLAB_FWD
        MG_CHECK_LAUNCH();
        return MG_OK;
    }
    const NormPlan p = norm_plan(B, HW, C);
    double* part = (double*)workspace;
    if (C % 4 == 0 && al16(x))
        hipLaunchKernelGGL((norm_partial_kernel<0, true>), dim3(p.cblocks, B, p.splits), dim3(256), 0, st, x, nullptr,
                           nullptr, nullptr, HW, C, p.rows_per_split, act, part);
    else
        hipLaunchKernelGGL((norm_partial_kernel<0, false>), dim3(p.cblocks, B, p.splits), dim3(256), 0, st, x, nullptr,
                           nullptr, nullptr, HW, C, p.rows_per_split, act, part);
    hipLaunchKernelGGL(norm_finalize_kernel<0>, dim3((B * C * 8 + 255) / 256), dim3(256), 0, st, part, B, p.splits, C, HW,
                       eps, mean, rstd);
    const size_t total = (size_t)B * HW * C;
    const bool vec = (C % 4 == 0) && al16(x) && al16(y) && (!residual || al16(residual));
    if (vec && norm_rows_on() && B <= 65535) {
        int rps = 0;
        const dim3 grid = norm_rows_grid(B, HW, C, &rps);
        hipLaunchKernelGGL(norm_apply_rows_kernel<false>, grid, dim3(256), 0, st, x, (const float*)nullptr, mean, rstd,
                           (const float*)nullptr, (const float*)nullptr, residual, HW, C, rps, act, y, y16);
    } else if (vec)
        hipLaunchKernelGGL(norm_apply_fwd_kernel<true>, dim3(grid_for(total, 4)), dim3(256), 0, st, x, mean, rstd,
                           residual, HW, C, act, y, total, y16);
    else
        hipLaunchKernelGGL(norm_apply_fwd_kernel<false>, dim3(grid_for(total)), dim3(256), 0, st, x, mean, rstd,
                           residual, HW, C, act, y, total, y16);
    MG_CHECK_LAUNCH();
    return MG_OK;
}

// library-internal (conv_igemm.hip: mg_conv_fwd_instnorm_*): InstanceNorm straight from a convolution's split-K slabs
__attribute__((visibility("hidden"))) int mg_instnorm_slab_ok(int HW, int C) { return slab_np(HW, C) ? 1 : 0; }
__attribute__((visibility("hidden"))) int mg_instnorm_fwd_slabs(const float* part, int S, const float* bias, int round_f16, float* x_out,
                                                               int B, int HW, int C, float eps, int act, const float* residual,
                                                               float* y, float* mean, float* rstd, void* stream, void* y16v) {
    const int np = slab_np(HW, C);
    _Float16* y16 = (_Float16*)y16v;
    if (!np || !part || S < 1 || !y || !mean || !rstd) return MG_ERR_ARG;
    if (!al16(part) || !al16(y) || !al16(mean) || !al16(rstd) || (residual && !al16(residual)) || (bias && !al16(bias)) ||
        (x_out && !al16(x_out)) || (y16 && (reinterpret_cast<uintptr_t>(y16) & 7)))
        return MG_ERR_ARG;
    const NormSlabSrc src{part, bias, x_out, (size_t)B * HW * C, S, round_f16};
    const dim3 grid(C / 32, B);
    hipStream_t st = (hipStream_t)stream;
#define MG_SLAB_FWD_S(NP_) hipLaunchKernelGGL((norm_slab_fwd_kernel<NP_, true>), grid, dim3(256), 0, st, (const float*)nullptr, HW, C, eps, act, residual, y, mean, rstd, y16, src)
    if (np == 4) MG_SLAB_FWD_S(4); else if (np == 8) MG_SLAB_FWD_S(8); else if (np == 16) MG_SLAB_FWD_S(16); else MG_SLAB_FWD_S(20);
#undef MG_SLAB_FWD_S
    MG_CHECK_LAUNCH();
    return MG_OK;
}

int mg_instnorm_bwd(const float* dy, const float* x, const float* mean, const float* rstd, int B, int HW, int C,
                    int act, float* dx, void* workspace, size_t workspace_bytes, void* stream) {
    return mg_instnorm_bwd_h(dy, x, mean, rstd, B, HW, C, act, dx, workspace, workspace_bytes, stream, nullptr);
}
int mg_instnorm_bwd_h(const float* dy, const float* x, const float* mean, const float* rstd, int B, int HW, int C,
                      int act, float* dx, void* workspace, size_t workspace_bytes, void* stream, void* dx16v) {
    return mg_instnorm_bwd_add(dy, nullptr, x, mean, rstd, B, HW, C, act, dx, workspace, workspace_bytes, stream, dx16v);
}
int mg_instnorm_bwd_add(const float* dy, const float* dy2, const float* x, const float* mean, const float* rstd, int B, int HW, int C,
                        int act, float* dx, void* workspace, size_t workspace_bytes, void* stream, void* dx16v) {
    _Float16* dx16 = (_Float16*)dx16v;
    if (dy2 && !al16(dy2)) return MG_ERR_ARG;
    if (dx16 && ((reinterpret_cast<uintptr_t>(dx16) & 7) != 0 || C % 4 != 0)) return MG_ERR_ARG;
    if (!dy || !x || !mean || !rstd || !dx || !workspace || B <= 0 || HW <= 0 || C <= 0) return MG_ERR_ARG;
    if (workspace_bytes < mg_instnorm_workspace(B, HW, C)) return MG_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (const int np = slab_np(HW, C); np && al16(x) && al16(dy) && al16(dx) && al16(mean) && al16(rstd)) {
        const dim3 grid(C / 32, B);
#define MG_SLAB_BWD(NP_) hipLaunchKernelGGL(norm_slab_bwd_kernel<NP_>, grid, dim3(256), 0, st, dy, x, mean, rstd, HW, C, act, dx, dx16, dy2)
        if (np == 4) MG_SLAB_BWD(4); else if (np == 8) MG_SLAB_BWD(8); else if (np == 16) MG_SLAB_BWD(16); else MG_SLAB_BWD(20);
#undef MG_SLAB_BWD
        MG_CHECK_LAUNCH();
        return MG_OK;
    }
    const NormPlan p = norm_plan(B, HW, C);
    double* part = (double*)workspace;
    float* m1 = (float*)((char*)workspace + (size_t)B * p.splits * C * 2 * sizeof(double));
    float* m2 = m1 + (size_t)B * C;
    if (C % 4 == 0 && al16(x) && al16(dy))
        hipLaunchKernelGGL((norm_partial_kernel<1, true>), dim3(p.cblocks, B, p.splits), dim3(256), 0, st, x, dy, mean,
                           rstd, HW, C, p.rows_per_split, act, part, dy2);
    else
        hipLaunchKernelGGL((norm_partial_kernel<1, false>), dim3(p.cblocks, B, p.splits), dim3(256), 0, st, x, dy, mean,
                           rstd, HW, C, p.rows_per_split, act, part, dy2);
    hipLaunchKernelGGL(norm_finalize_kernel<1>, dim3((B * C * 8 + 255) / 256), dim3(256), 0, st, part, B, p.splits, C, HW,
                       0.0f, m1, m2);
    const size_t total = (size_t)B * HW * C;
    const bool vec = (C % 4 == 0) && al16(x) && al16(dy) && al16(dx);
    if (vec && norm_rows_on() && B <= 65535) {
        int rps = 0;
        const dim3 grid = norm_rows_grid(B, HW, C, &rps);
        hipLaunchKernelGGL(norm_apply_rows_kernel<true>, grid, dim3(256), 0, st, x, dy, mean, rstd, (const float*)m1,
                           (const float*)m2, (const float*)nullptr, HW, C, rps, act, dx, dx16, dy2);
    } else if (vec)
        hipLaunchKernelGGL(norm_apply_bwd_kernel<true>, dim3(grid_for(total, 4)), dim3(256), 0, st, dy, x, mean, rstd,
                           m1, m2, HW, C, act, dx, total, dx16, dy2);
    else
        hipLaunchKernelGGL(norm_apply_bwd_kernel<false>, dim3(grid_for(total)), dim3(256), 0, st, dy, x, mean, rstd,
                           m1, m2, HW, C, act, dx, total, dx16, dy2);
    MG_CHECK_LAUNCH();
    return MG_OK;
}

int mg_act_bwd(const float* dy, const float* y, float* dx, long long n, int act, void* stream) {
    return mg_act_bwd_add(dy, nullptr, y, dx, n, act, stream);
}
int mg_act_bwd_add(const float* dy, const float* dy2, const float* y, float* dx, long long n, int act, void* stream) {
    if (!dy || !y || !dx || n <= 0) return MG_ERR_ARG;
    hipLaunchKernelGGL(act_bwd_kernel, dim3(grid_for((size_t)n)), dim3(256), 0, (hipStream_t)stream, dy, y, dx,
                       (size_t)n, act, dy2);
    MG_CHECK_LAUNCH();
    return MG_OK;
}

int mg_add(const float* a, const float* b, float* out, long long n, void* stream) {
    if (!a || !b || !out || n <= 0) return MG_ERR_ARG;
    hipLaunchKernelGGL(add_kernel, dim3(grid_for((size_t)n)), dim3(256), 0, (hipStream_t)stream, a, b, out, (size_t)n);
    MG_CHECK_LAUNCH();
    return MG_OK;
}

int mg_avgpool3s2_fwd(const float* x, int B, int H, int W, int C, float* y, void* stream) {
    if (!x || !y || B <= 0 || H <= 0 || W <= 0 || C <= 0) return MG_ERR_ARG;
    const int OH = (H - 1) / 2 + 1, OW = (W - 1) / 2 + 1;
    hipLaunchKernelGGL(avgpool_fwd_kernel, dim3(grid_for((size_t)B * OH * OW * C)), dim3(256), 0, (hipStream_t)stream,
                       x, B, H, W, C, OH, OW, y);
    MG_CHECK_LAUNCH();
    return MG_OK;
}
int mg_avgpool3s2_bwd(const float* dy, int B, int H, int W, int C, float* dx, void* stream) {
    if (!dy || !dx || B <= 0 || H <= 0 || W <= 0 || C <= 0) return MG_ERR_ARG;
    const int OH = (H - 1) / 2 + 1, OW = (W - 1) / 2 + 1;
    hipLaunchKernelGGL(avgpool_bwd_kernel, dim3(grid_for((size_t)B * H * W * C)), dim3(256), 0, (hipStream_t)stream, dy,
                       B, H, W, C, OH, OW, dx);
    MG_CHECK_LAUNCH();
    return MG_OK;
}
int mg_upsample2x_fwd(const float* x, int B, int H, int W, int C, float* y, void* stream) {
    if (!x || !y || B <= 0 || H <= 0 || W <= 0 || C <= 0) return MG_ERR_ARG;
    hipLaunchKernelGGL(upsample_fwd_kernel, dim3(grid_for((size_t)B * 4 * H * W * C)), dim3(256), 0,
                       (hipStream_t)stream, x, B, H, W, C, y);
    MG_CHECK_LAUNCH();
    return MG_OK;
}
int mg_upsample2x_bwd(const float* dy, int B, int H, int W, int C, float* dx, void* stream) {
    if (!dy || !dx || B <= 0 || H <= 0 || W <= 0 || C <= 0) return MG_ERR_ARG;
    hipLaunchKernelGGL(upsample_bwd_kernel, dim3(grid_for((size_t)B * H * W * C)), dim3(256), 0, (hipStream_t)stream,
                       dy, B, H, W, C, dx);
    MG_CHECK_LAUNCH();
    return MG_OK;
}

int mg_dinput_fwd(const float* lr, const float* s, long long n, float nr0, float* out, void* stream) {
    if (!lr || !s || !out || n <= 0) return MG_ERR_ARG;
    hipLaunchKernelGGL(dinput_fwd_kernel, dim3(grid_for((size_t)n)), dim3(256), 0, (hipStream_t)stream, lr, s,
                       (size_t)n, nr0, out);
    MG_CHECK_LAUNCH();
    return MG_OK;
}
int mg_dinput_bwd(const float* dout, const float* s, long long n, float* ds, void* stream) {
    if (!dout || !s || !ds || n <= 0) return MG_ERR_ARG;
    hipLaunchKernelGGL(dinput_bwd_kernel, dim3(grid_for((size_t)n)), dim3(256), 0, (hipStream_t)stream, dout, s,
                       (size_t)n, ds);
    MG_CHECK_LAUNCH();
    return MG_OK;
}
int mg_cat2_fwd(const float* a, int Ca, const float* b, int Cb, long long n, float* out, void* stream) {
    if (!a || !b || !out || n <= 0 || Ca <= 0 || Cb <= 0) return MG_ERR_ARG;
    hipLaunchKernelGGL(cat2_fwd_kernel, dim3(grid_for((size_t)n * (Ca + Cb))), dim3(256), 0, (hipStream_t)stream, a, Ca, b, Cb,
                       (size_t)n, out);
    MG_CHECK_LAUNCH();
    return MG_OK;
}
int mg_cat2_bwd(const float* g, int Ca, int Cb, long long n, float* ga, float* gb, void* stream) {
    if (!g || (!ga && !gb) || n <= 0 || Ca <= 0 || Cb <= 0) return MG_ERR_ARG;
    hipLaunchKernelGGL(cat2_bwd_kernel, dim3(grid_for((size_t)n * (Ca + Cb))), dim3(256), 0, (hipStream_t)stream, g, Ca, Cb,
                       (size_t)n, ga, gb);
    MG_CHECK_LAUNCH();
    return MG_OK;
}
int mg_stats_finalize(const double* stats, long long n, float* mean_std, void* stream) {
    if (!stats || !mean_std || n < 2) return MG_ERR_ARG;
    hipLaunchKernelGGL(stats_finalize_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, stats, (double)n, mean_std);
    MG_CHECK_LAUNCH();
    return MG_OK;
}
int mg_pair_fwd(const float* s, long long n, float nr0, float* out, void* stream) {
    if (!s || !out || n <= 0) return MG_ERR_ARG;
    hipLaunchKernelGGL(pair_fwd_kernel, dim3(grid_for((size_t)n)), dim3(256), 0, (hipStream_t)stream, s, (size_t)n,
                       nr0, out);
    MG_CHECK_LAUNCH();
    return MG_OK;
}

size_t mg_loss_workspace(void) { return 1024 * sizeof(double); }

int mg_mse_const_fwd(const float* pred, long long n, float target, float scale, float* loss, int accumulate,
                     void* workspace, void* stream) {
    if (!pred || !loss || !workspace || n <= 0) return MG_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    size_t nb = ((size_t)n + 1023) / 1024;
    if (nb > 1024) nb = 1024;
    hipLaunchKernelGGL(loss_partial_kernel<0>, dim3((unsigned)nb), dim3(256), 0, st, pred, nullptr, target, (size_t)n,
                       (double*)workspace);
    hipLaunchKernelGGL(loss_final_kernel, dim3(1), dim3(64), 0, st, (const double*)workspace, (int)nb, 1.0 / (double)n,
                       scale, loss, accumulate);
    MG_CHECK_LAUNCH();
    return MG_OK;
}
int mg_mse_const_bwd(const float* pred, long long n, float target, float scale, const float* grad_out, float* grad,
                     void* stream) {
    if (!pred || !grad || n <= 0) return MG_ERR_ARG;
    hipLaunchKernelGGL(loss_grad_kernel<0>, dim3(grid_for((size_t)n)), dim3(256), 0, (hipStream_t)stream, pred, nullptr,
                       target, (size_t)n, scale / (float)n, grad_out, grad);
    MG_CHECK_LAUNCH();
    return MG_OK;
}
int mg_bce_const_fwd(const float* pred, long long n, float target, float scale, float* loss, int accumulate,
                     void* workspace, void* stream) {
    if (!pred || !loss || !workspace || n <= 0) return MG_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    size_t nb = ((size_t)n + 1023) / 1024;
    if (nb > 1024) nb = 1024;
    hipLaunchKernelGGL(loss_partial_kernel<2>, dim3((unsigned)nb), dim3(256), 0, st, pred, nullptr, target, (size_t)n,
                       (double*)workspace);
    hipLaunchKernelGGL(loss_final_kernel, dim3(1), dim3(64), 0, st, (const double*)workspace, (int)nb, 1.0 / (double)n,
                       scale, loss, accumulate);
    MG_CHECK_LAUNCH();
    return MG_OK;
}
int mg_bce_const_bwd(const float* pred, long long n, float target, float scale, const float* grad_out, float* grad,
                     void* stream) {
    if (!pred || !grad || n <= 0) return MG_ERR_ARG;
    hipLaunchKernelGGL(loss_grad_kernel<2>, dim3(grid_for((size_t)n)), dim3(256), 0, (hipStream_t)stream, pred, nullptr,
                       target, (size_t)n, scale / (float)n, grad_out, grad);
    MG_CHECK_LAUNCH();
    return MG_OK;
}
int mg_sigmoid_fwd(const float* x, float* y, long long n, void* stream) {
    if (!x || !y || n <= 0) return MG_ERR_ARG;
    hipLaunchKernelGGL(sigmoid_fwd_kernel, dim3(grid_for((size_t)n)), dim3(256), 0, (hipStream_t)stream, x, y, (size_t)n);
    MG_CHECK_LAUNCH();
    return MG_OK;
}
int mg_sigmoid_bwd(const float* dy, const float* y, float* dx, long long n, void* stream) {
    if (!dy || !y || !dx || n <= 0) return MG_ERR_ARG;
    hipLaunchKernelGGL(sigmoid_bwd_kernel, dim3(grid_for((size_t)n)), dim3(256), 0, (hipStream_t)stream, dy, y, dx, (size_t)n);
    MG_CHECK_LAUNCH();
    return MG_OK;
}
int mg_l1_fwd(const float* a, const float* b, long long n, float scale, float* loss, int accumulate, void* workspace,
              void* stream) {
    if (!a || !b || !loss || !workspace || n <= 0) return MG_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    size_t nb = ((size_t)n + 1023) / 1024;
    if (nb > 1024) nb = 1024;
    hipLaunchKernelGGL(loss_partial_kernel<1>, dim3((unsigned)nb), dim3(256), 0, st, a, b, 0.0f, (size_t)n,
                       (double*)workspace);
    hipLaunchKernelGGL(loss_final_kernel, dim3(1), dim3(64), 0, st, (const double*)workspace, (int)nb, 1.0 / (double)n,
                       scale, loss, accumulate);
    MG_CHECK_LAUNCH();
    return MG_OK;
}
int mg_l1_bwd(const float* a, const float* b, long long n, float scale, const float* grad_out, float* grad_a,
              void* stream) {
    if (!a || !b || !grad_a || n <= 0) return MG_ERR_ARG;
    hipLaunchKernelGGL(loss_grad_kernel<1>, dim3(grid_for((size_t)n)), dim3(256), 0, (hipStream_t)stream, a, b, 0.0f,
                       (size_t)n, scale / (float)n, grad_out, grad_a);
    MG_CHECK_LAUNCH();
    return MG_OK;
}

size_t mg_loss_multi_workspace(void) { return (size_t)MG_LOSS_MAX_ITEMS * 1024 * sizeof(double); }
static bool loss_list(const mg_loss_item* items, int count, bool need_b, bool need_grad, LossList* L) {
    if (!items || count < 1 || count > MG_LOSS_MAX_ITEMS) return false;
    L->count = count;
    unsigned fb = 0;
    for (int i = 0; i < count; ++i) {
        if (!items[i].a || items[i].n <= 0 || (need_b && !items[i].b) || (need_grad && !items[i].grad) || items[i].zero_tail < 0)
            return false;
        L->a[i] = items[i].a; L->b[i] = items[i].b; L->grad[i] = items[i].grad;
        L->n[i] = (unsigned long long)items[i].n; L->zero_tail[i] = (unsigned long long)items[i].zero_tail;
        L->first_block[i] = fb;
        size_t nb = ((size_t)items[i].n + 1023) / 1024;
        if (nb > 1024) nb = 1024;
        fb += (unsigned)nb;
    }
    L->first_block[count] = fb;
    return true;
}
int mg_loss_multi_fwd(int kind, const mg_loss_item* items, int count, float target, float scale, float* loss, int accumulate,
                      void* workspace, size_t workspace_bytes, void* stream) {
    LossList L;
    if (kind < 0 || kind > 2 || !loss || !workspace || workspace_bytes < mg_loss_multi_workspace()) return MG_ERR_ARG;
    if (!loss_list(items, count, kind == 1, false, &L)) return MG_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid(L.first_block[count]);
    double* part = (double*)workspace;
    if (kind == 0) hipLaunchKernelGGL(loss_partial_multi_kernel<0>, grid, dim3(256), 0, st, L, target, part);
    else if (kind == 1) hipLaunchKernelGGL(loss_partial_multi_kernel<1>, grid, dim3(256), 0, st, L, target, part);
    else hipLaunchKernelGGL(loss_partial_multi_kernel<2>, grid, dim3(256), 0, st, L, target, part);
    hipLaunchKernelGGL(loss_final_multi_kernel, dim3(1), dim3(64 * MG_LOSS_MAX_ITEMS), 0, st, L, (const double*)part, scale, loss,
                       accumulate);
    MG_CHECK_LAUNCH();
    return MG_OK;
}
int mg_loss_multi_bwd(int kind, const mg_loss_item* items, int count, float target, float scale, const float* grad_out,
                      void* stream) {
    LossList L;
    if (kind < 0 || kind > 2) return MG_ERR_ARG;
    if (!loss_list(items, count, kind == 1, true, &L)) return MG_ERR_ARG;
    size_t nmax = 0;
    for (int i = 0; i < count; ++i) {
        const size_t ni = (size_t)(items[i].n + items[i].zero_tail);
        if (ni > nmax) nmax = ni;
        L.coef[i] = scale / (float)items[i].n;
    }
    const dim3 grid(grid_for(nmax), count);
    hipStream_t st = (hipStream_t)stream;
    if (kind == 0) hipLaunchKernelGGL(loss_grad_multi_kernel<0>, grid, dim3(256), 0, st, L, target, grad_out);
    else if (kind == 1) hipLaunchKernelGGL(loss_grad_multi_kernel<1>, grid, dim3(256), 0, st, L, target, grad_out);
    else hipLaunchKernelGGL(loss_grad_multi_kernel<2>, grid, dim3(256), 0, st, L, target, grad_out);
    MG_CHECK_LAUNCH();
    return MG_OK;
}

int mg_adam_step(float* p, const float* g, float* m, float* v, long long n, float lr, float beta1, float beta2,
                 float eps, int step, float grad_scale, void* stream) {
    if (!p || !g || !m || !v || n <= 0 || step < 1) return MG_ERR_ARG;
    const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
    hipLaunchKernelGGL(adam_kernel, dim3(grid_for((size_t)n, 4)), dim3(256), 0, (hipStream_t)stream, p, g, m, v,
                       (size_t)n, (float)((double)lr / bc1), beta1, beta2, eps, (float)sqrt(bc2), grad_scale);
    MG_CHECK_LAUNCH();
    return MG_OK;
}


int mg_adam_tick(double* state, float beta1, float beta2, void* stream) {
    if (!state) return MG_ERR_ARG;
    hipLaunchKernelGGL(adam_tick_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, state, (double)beta1, (double)beta2);
    MG_CHECK_LAUNCH();
    return MG_OK;
}

int mg_adam_prime(double* state, float beta1, float beta2, void* stream) {
    if (!state) return MG_ERR_ARG;
    hipLaunchKernelGGL(adam_prime_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, state, (double)beta1, (double)beta2);
    MG_CHECK_LAUNCH();
    return MG_OK;
}

int mg_adam_step_dev(float* p, const float* g, float* m, float* v, long long n, const double* state, float beta1,
                     float beta2, float eps, float grad_scale, void* stream) {
    if (!p || !g || !m || !v || !state || n <= 0) return MG_ERR_ARG;
    if (!al16(p) || !al16(g) || !al16(m) || !al16(v)) return MG_ERR_ARG;
    hipLaunchKernelGGL(adam_dev_kernel, dim3(grid_for((size_t)n, 8)), dim3(256), 0, (hipStream_t)stream, p, g, m, v,
                       (size_t)n, state, beta1, beta2, eps, grad_scale, (const float*)nullptr, (const float*)nullptr);
    MG_CHECK_LAUNCH();
    return MG_OK;
}

int mg_scaler_check(const float* g, long long n, float* scaler, int slot, void* stream) {
    if (!g || !scaler || n <= 0 || slot < 0 || slot >= MG_SCALER_SLOTS || !al16(g)) return MG_ERR_ARG;
    hipLaunchKernelGGL(scaler_check_kernel, dim3(grid_for((size_t)n, 16)), dim3(256), 0, (hipStream_t)stream, g, (size_t)n,
                       scaler + 2 + slot);
    MG_CHECK_LAUNCH();
    return MG_OK;
}

int mg_scaler_update(float* scaler, float growth_factor, float backoff_factor, int growth_interval, void* stream) {
    if (!scaler || growth_interval < 1) return MG_ERR_ARG;
    hipLaunchKernelGGL(scaler_update_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, scaler, MG_SCALER_SLOTS,
                       growth_factor, backoff_factor, growth_interval);
    MG_CHECK_LAUNCH();
    return MG_OK;
}

int mg_adam_tick_amp(double* state, float beta1, float beta2, const float* scaler, int slot, void* stream) {
    if (!state || !scaler || slot < 0 || slot >= MG_SCALER_SLOTS) return MG_ERR_ARG;
    hipLaunchKernelGGL(adam_tick_amp_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, state, (double)beta1,
                       (double)beta2, scaler + 2 + slot);
    MG_CHECK_LAUNCH();
    return MG_OK;
}

int mg_adam_step_amp(float* p, const float* g, float* m, float* v, long long n, const double* state, float beta1,
                     float beta2, float eps, float grad_scale, const float* scaler, int slot, void* stream) {
    if (!p || !g || !m || !v || !state || !scaler || n <= 0 || slot < 0 || slot >= MG_SCALER_SLOTS) return MG_ERR_ARG;
    if (!al16(p) || !al16(g) || !al16(m) || !al16(v)) return MG_ERR_ARG;
    hipLaunchKernelGGL(adam_dev_kernel, dim3(grid_for((size_t)n, 8)), dim3(256), 0, (hipStream_t)stream, p, g, m, v,
                       (size_t)n, state, beta1, beta2, eps, grad_scale, scaler, scaler + 2 + slot);
    MG_CHECK_LAUNCH();
    return MG_OK;
}

int mg_adam_step_h(float* p, const float* g, float* m, float* v, void* p16, long long n, const double* state, float beta1,
                   float beta2, float eps, float grad_scale, const float* scaler, int slot, void* stream) {
    if (!p || !g || !m || !v || !p16 || !state || n <= 0) return MG_ERR_ARG;
    if (scaler && (slot < 0 || slot >= MG_SCALER_SLOTS)) return MG_ERR_ARG;
    if (!al16(p) || !al16(g) || !al16(m) || !al16(v) || (reinterpret_cast<uintptr_t>(p16) & 7)) return MG_ERR_ARG;
    hipLaunchKernelGGL(adam_dev_kernel, dim3(grid_for((size_t)n, 8)), dim3(256), 0, (hipStream_t)stream, p, g, m, v,
                       (size_t)n, state, beta1, beta2, eps, grad_scale, scaler, scaler ? scaler + 2 + slot : (const float*)nullptr,
                       (_Float16*)p16);
    MG_CHECK_LAUNCH();
    return MG_OK;
}

static bool segs_ok(const mg_grad_seg* segs, int nsegs) { return segs && nsegs > 0 && nsegs <= 4096; }
int mg_scaler_check_segs(const float* g, const void* g16, const mg_grad_seg* segs, int nsegs, long long n_total, float* scaler,
                         int slot, void* stream) {
    if (!g || !scaler || !segs_ok(segs, nsegs) || n_total <= 0 || slot < 0 || slot >= MG_SCALER_SLOTS || !al16(g) || (g16 && !al16(g16)))
        return MG_ERR_ARG;
    hipLaunchKernelGGL(scaler_check_seg_kernel, dim3(grid_for((size_t)n_total, 16)), dim3(256), 0, (hipStream_t)stream, g,
                       (const _Float16*)g16, segs, nsegs, scaler + 2 + slot);
    MG_CHECK_LAUNCH();
    return MG_OK;
}
int mg_adam_step_segs(float* p, const float* g, const void* g16, float* m, float* v, void* p16, const mg_grad_seg* segs, int nsegs,
                      long long n_total, const double* state, float beta1, float beta2, float eps, float grad_scale,
                      const float* scaler, int slot, void* stream) {
    if (!p || !g || !m || !v || !state || !segs_ok(segs, nsegs) || n_total <= 0) return MG_ERR_ARG;
    if (scaler && (slot < 0 || slot >= MG_SCALER_SLOTS)) return MG_ERR_ARG;
    if (!al16(p) || !al16(g) || !al16(m) || !al16(v) || (p16 && !al16(p16)) || (g16 && !al16(g16))) return MG_ERR_ARG;
    hipLaunchKernelGGL(adam_seg_kernel, dim3(grid_for((size_t)n_total, 8)), dim3(256), 0, (hipStream_t)stream, p, g,
                       (const _Float16*)g16, m, v, (_Float16*)p16, segs, nsegs, state, beta1, beta2, eps, grad_scale, scaler,
                       scaler ? scaler + 2 + slot : (const float*)nullptr);
    MG_CHECK_LAUNCH();
    return MG_OK;
}

}  // extern "C"
