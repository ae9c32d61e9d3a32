/* libmdctgan_hip.so -- C ABI of the MI355X (gfx950) mdctGAN hot path.
 *
 * The reference (neoncloud/mdctGAN) is pure Python/PyTorch and has no FFI layer of its own: the
 * seam is the module API of models/mdct.py, models/pix2pixHD_model.py and models/networks.py.
 * Each entry point below replaces the chain of ATen ops one of those methods dispatches; the
 * reference site it replaces is cited on every declaration (paths relative to the reference
 * repository).  The Python package mdctgan_amd binds these with ctypes and re-exposes the reference's own
 * class / function names (see INTEGRATION.md for the binding a maintainer would add).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (HBM) unless it says "host"; no torch types anywhere;
 *   - activations are float32 NHWC ([B, H, W, C], C fastest); spectrograms [B, F, 256] are the
 *     C == 1 case of both NHWC and the reference's NCHW;
 *   - convolution weights are float32 "OHWI": [Cout, KH, KW, Cin] for nn.Conv2d and
 *     [Cin_T, KH, KW, Cout_T] for nn.ConvTranspose2d.  A reference-shaped state_dict tensor
 *     ([Cout, Cin, KH, KW] / [Cin_T, Cout_T, KH, KW]) is the SAME memory viewed through
 *     .permute(0, 3, 1, 2), so checkpoints load unchanged;
 *   - `stream` is a hipStream_t (0 = the null stream); all work is enqueued asynchronously;
 *   - return value: 0 on success, a positive hipError_t, or MG_ERR_* (negative) for a rejected
 *     argument.  Nothing is ever computed on the host.
 */
#ifndef MDCTGAN_HIP_H
#define MDCTGAN_HIP_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MG_OK 0
#define MG_ERR_ARG (-1)
#define MG_ERR_UNSUPPORTED (-2)

/* codec selector: raw MDCT coefficients, arcsinh + range-norm, range-norm only (--raw_mdct) */
#define MG_CODEC_RAW 0
#define MG_CODEC_ARCSINH 1
#define MG_CODEC_RANGE 2

/* activation selector for fused epilogues */
#define MG_ACT_NONE 0
#define MG_ACT_RELU 1
#define MG_ACT_LRELU02 2
#define MG_ACT_TANH 3

/* 3 since mg_mdct4_forward / mg_imdct4_forward take the factored DCT-IV image (dct4_image) in the middle of their argument lists: a
   caller built against an older header must see a different number here before it passes `codec` where a pointer is expected.
   4 (round 6): mg_grad_seg / mg_scaler_check_segs / mg_adam_step_segs / mg_conv_wgrad_h16 added; mg_conv_wgrad_chk's found_inf
   test uses float16's overflow criterion (|v| >= 65520), as the --fp16 optimiser passes do. */
int mg_abi_version(void);
/* sizeof(mg_conv_geom) as the library was built (13 ints = 52 bytes): a binding checks its own struct against it */
int mg_conv_geom_size(void);

/* ------------------------------------------------------------------------------------------
 * K1  MDCT4.forward (models/mdct.py:392-425) fused with Audio2MDCT.normalize
 *     (models/pix2pixHD_model.py:83-125) and the 2-channel input build (:400-402).
 *
 *   audio [B, T] -> spec [B, F, n_fft/2], F = mg_mdct4_num_frames(T, n_fft); n_fft == 512 (hop 256).
 *   codec RAW: spec = X.  ARCSINH: L = asinh(gain*X)/ln10.  RANGE: L = X.  Then
 *   spec = (L - min)/(max - min)*(nr1 - nr0) + nr0, with (min, max) = (src_min, src_max)
 *   (--abs_norm) or, when per_sample != 0, the per-clip min/max of L (written to min_out/max_out [B]).
 *   in2 (nullable): NHWC pair [B, F, 256, 2] = (spec, 2*|spec| + nr0) -- the generator input.
 *   frames_out (nullable): windowed frames [B, F, n_fft] (return_frames=True).
 *   stats (nullable): double[2] = sum(L), sum(L^2) over the batch (for the returned mean / std).
 *   scratch_u32: >= 2*B uint32, required when per_sample != 0.
 *   window [n_fft] and dct4 [n_fft/2][n_fft/2] = cos(pi/M (n+1/2)(k+1/2)) are device tables.
 *   dct4_image (nullable): mg_dct4_image_floats(n_fft) floats that mg_dct4_image(dct4, dct4_image, stream) fills once per
 *   table -- the operand images the table-stationary kernels keep in registers (csrc/mdct_bs.h: the float32 register image;
 *   csrc/mdct_b3.h: three bf16 piece images, the float32 value being their exact sum).  NULL selects the kernels that read
 *   the plain table only (a caller written against the round-1 ABI passes its m*m table and NULL).  16-byte aligned.
 *   spec may be NULL when in2 is given and the table-stationary kernels apply (dct4_image present, T % 4 == 0, 16-byte aligned
 *   pointers, no per_sample, no frames_out; MG_ERR_ARG otherwise): the spectrogram is then channel 0 of the pair only
 *   (393 216 B per clip moved instead of 526 848).
 */
long long mg_dct4_image_floats(int n_fft);
int mg_dct4_image(const float* dct4, float* dct4_image, void* stream);
int mg_mdct4_forward(const float* audio, int B, int T, int n_fft, const float* window, const float* dct4,
                     const float* dct4_image, int codec, float gain, float nr0, float nr1, float src_min, float src_max,
                     int per_sample, float* spec, float* in2, float* frames_out, float* min_out, float* max_out,
                     double* stats, unsigned* scratch_u32, void* stream);
int mg_mdct4_num_frames(int T, int n_fft);

/* K2  Audio2MDCT.denormalize (models/pix2pixHD_model.py:127-137) fused with IMDCT4.forward
 *     (models/mdct.py:457-489: inverse DCT, window, fold() overlap-add, 4/N scale, centre crop).
 *   spec [B, F, 256] -> audio [B, out_len], out_len <= (F-1)*256; float32, or float64 when out_f64.
 *   min_b / max_b (nullable, [B]): per-clip range; otherwise (src_min, src_max).
 *   frames_out (nullable): windowed synthesis frames [B, F, n_fft].
 */
int mg_imdct4_forward(const float* spec, int B, int F, int n_fft, const float* window, const float* dct4,
                      const float* dct4_image, int codec, float gain, float nr0, float nr1, float src_min, float src_max,
                      const float* min_b, const float* max_b, void* audio, int out_len, int out_f64,
                      float* frames_out, void* stream);

/* K2 + generate_audio.py:40-53 in ONE kernel: IMDCT4.forward whose overlap-add store writes straight into the stitched waveform,
 *     replacing `audio.append(sr_audio)` (generate_audio.py:37), the two `*= 0.5`, F.fold and the final crop (generate_audio.py:
 *     43-50) -- and torch.cat(...).view(1, -1) (generate_audio.py:52) when gen_overlap == 0.
 *   spec [B, F, 256] are segments first_seg .. first_seg + B - 1 of a signal cut into segments of seg_len samples at stride
 *   seg_len - overlap (data/audio_dataset.py:153-167 seg_pad_audio); out is the WHOLE stitched waveform, out_total =
 *   mg_stitch_length(n_seg, seg_len, overlap) samples.  Sample t of segment s lands at s * (seg_len - overlap) - overlap + t
 *   (positions outside [0, out_total) are the reference's crop); the first / last `overlap` samples of every segment are halved
 *   and added to the neighbour's (two-term float sums: order-free, equal to F.fold's bit for bit), the rest is stored.  So with
 *   overlap > 0 the waveform must be zero before the first batch writes: zero_out != 0 makes this call clear it first
 *   (a memset node in front of the kernel: pass it for the batch with first_seg == 0).  Batches may arrive in any order.
 *   float32 output (float64 when out_f64 -- generic kernel); seg_len <= (F-1)*256.  Same codec arguments as mg_imdct4_forward.
 */
int mg_imdct4_stitched(const float* spec, int B, int F, int n_fft, const float* window, const float* dct4,
                       const float* dct4_image, int codec, float gain, float nr0, float nr1, float src_min, float src_max,
                       const float* min_b, const float* max_b, void* out, long long out_total, int seg_len, int overlap,
                       long long first_seg, int zero_out, int out_f64, void* stream);

/* Which kernel the last mg_mdct4_forward (which == 0) / mg_imdct4_forward / mg_imdct4_stitched (which == 1) call of this process
 * launched -- a static string ("mdct4_ct_kernel (csrc/mdct_ct.h)", ...).  Diagnostic: bench.py names the measured kernel with it. */
const char* mg_mdct_last_kernel(int which);

/* F1  torchaudio.functional.resample(waveform, orig_freq, new_freq) with its defaults (sinc_interp_hann,
 * lowpass_filter_width 6, rolloff 0.99) as the reference's data path calls it (data/audio_dataset.py:66-71, 171-177):
 * x [B, L] -> out [B, mg_resample_length(L, orig, new)], orig / new the gcd-reduced rates.  kern [new, 2*width + orig]
 * is the polyphase filter bank torchaudio's _get_sinc_resample_kernel builds (the host computes it once per rate pair
 * in float64 and passes it in float32, as torchaudio does). */
long long mg_resample_length(long long L, int orig, int new_);
int mg_resample(const float* x, int B, int L, const float* kern, int orig, int new_, int width, float* out, int out_len,
                void* stream);

/* F2  util/util.py:132-177 compute_matrics on the device (train.py:104-134 eval_model, generate_audio.py:60).
 *   mg_metrics_rows   out[b] = {sum hr^2, sum (sr - hr)^2, sum (lr - hr)^2} (double) for clips [B, T]  -> MSE / SNR
 *   mg_stft_frames    reflect-padded (center != 0), windowed frames [B * F, n_fft], F = mg_stft_num_frames(): the A
 *                     operand of the DFT, which the caller runs as the 1x1 case of mg_conv_fwd against a
 *                     [2 * (n_fft/2 + 1), n_fft] table of (cos, -sin) rows  (aF.spectrogram, util.py:170-171)
 *   mg_lsd_frames     per frame, from two interleaved (re, im) spectra [n_frames, 2 * n_bins]:
 *                     sqrt(mean_k (log10(|a_k|^2 + 1e-6) - log10(|b_k|^2 + 1e-6))^2)        (util.py:172-174) */
int mg_metrics_rows(const float* hr, const float* lr, const float* sr, int B, int T, double* out, void* stream);
int mg_stft_num_frames(int T, int n_fft, int hop, int center);
int mg_stft_frames(const float* x, int B, int T, const float* window, int n_fft, int hop, int center, float* frames,
                   void* stream);
int mg_lsd_frames(const float* spec_a, const float* spec_b, long long n_frames, int n_bins, float* out, void* stream);

/* Segment stitching of generate_audio.py:40-53: seg [n_seg, seg_len] (the [n_seg,1,1,T] inference outputs) -> one
 * waveform of mg_stitch_length() samples (-1: invalid arguments; 2*overlap must be < seg_len).  overlap == 0
 * concatenates; overlap > 0 halves the first/last `overlap` samples of every segment, overlap-adds at stride
 * seg_len - overlap (F.fold) and crops `overlap` samples from both ends.  float32, or float64 when is_f64. */
long long mg_stitch_length(int n_seg, int seg_len, int overlap);
int mg_stitch_segments(const void* seg, int n_seg, int seg_len, int overlap, void* out, int is_f64, void* stream);

/* ------------------------------------------------------------------------------------------
 * Generic-geometry transform and the remaining codec branches (csrc/codec_generic.hip): MDCT4 / IMDCT4 for any
 * win_length <= n_fft, hop_length <= win_length (models/mdct.py:365-489; the dense cosine contraction itself runs as the 1x1
 * case of mg_conv_fwd), and Audio2MDCT.normalize / denormalize (models/pix2pixHD_model.py:83-137) in every mode.
 * ------------------------------------------------------------------------------------------ */
#define MG_CODEC_DB 3        /* 20 log10(max(|X| + min_value, min_value)) - 20  (torchaudio amplitude_to_DB, unpinned) */
#define MG_CODEC_EXPLICIT 4  /* --explicit_encoding: two dB channels of alpha-mixed positive / negative parts */
/* frames[b, f, k] = fl32(x[b, f*hop + k - start_pad] * window[k]), zero outside [0, T)   (mdct.py:393-410) */
int mg_frames_window(const float* x, int B, int T, int win, int hop, int start_pad, int F, const float* window,
                     float* frames, void* stream);
/* X [B][n] raw coefficients -> out [B][C][n] (C = 2 for MG_CODEC_EXPLICIT, else 1), optional pair [B][n][2] =
 * (v, 2|v| + nr0) (C == 1), per_sample: min/max per (b, channel) into min_out / max_out [B*C] (scratch_u32: 2*B*C words),
 * stats: double[2] = sum, sum of squares of the pre-normalisation values. */
int mg_codec_forward(const float* X, int B, int n, int mode, float gain, float alpha, float min_value, float nr0, float nr1,
                     float src_min, float src_max, int per_sample, float* out, float* pair, float* min_out, float* max_out,
                     void* scratch_u32, double* stats, void* stream);
/* spec [B][C][n] -> X [B][n]; min_b / max_b [B*C] or both null (src_min / src_max) */
int mg_codec_inverse(const float* spec, int B, int n, int mode, float gain, float alpha, float min_value, float nr0, float nr1,
                     float src_min, float src_max, const float* min_b, const float* max_b, float* X, void* stream);
/* out[b, t] = 4/n_fft * sum_f window[k] * Y[b, f, k], k = t + crop - f*hop   (mdct.py:469-486), float32 or float64 out */
int mg_overlap_add(const float* Y, int B, int F, int win, int hop, int n_fft, const float* window, int crop, void* out,
                   int out_len, int is_f64, void* stream);

/* ------------------------------------------------------------------------------------------
 * K3/K4/K6  Implicit-GEMM convolution on the f32 MFMA pipe (exact float32).
 *   Replaces nn.Conv2d / nn.ConvTranspose2d / nn.ReflectionPad2d forward and backward at
 *   models/networks.py:207-210, 308-309, 329, 349-352, 387-392, 406-411, 440, 456, 649-670.
 *
 *   The geometry always describes the *convolution* (for ConvTranspose2d: the convolution whose
 *   data-gradient it is): x [B, H, W, Ci] --(KHxKW, stride, pad, zero|reflect)--> y [B, OH, OW, Co].
 */
typedef struct {
    int B, H, W, Ci;
    int OH, OW, Co;
    int KH, KW, stride, pad;
    int reflect; /* 0: zero padding, 1: reflection padding (ReflectionPad2d(pad) folded into the gather) */
    int precision; /* MG_PRECISION_F32: exact float32 on the f32 MFMA pipe.  MG_PRECISION_F16: the arithmetic of
                    * torch.autocast(float16) for convolutions (train.py:161-164 with --fp16): operands rounded to
                    * float16 as they are staged, products on the f16 MFMA pipe, float32 accumulation, outputs of the
                    * forward / data-gradient passes rounded through float16 (overflow -> inf, as a float16 tensor
                    * would), weight gradients kept in float32.  Tensors stay float32 in memory. */
} mg_conv_geom;
#define MG_PRECISION_F32 0
#define MG_PRECISION_F16 1

/* y = act(conv(x, w) + bias)            (bias nullable).  workspace (nullable): mg_conv_fwd_workspace() bytes of
 * scratch that lets deep-K / small-M*N layers split K across workgroups (without it they run unsplit). */
int mg_conv_fwd(const mg_conv_geom* g, const float* x, const float* w, const float* bias, float* y, int act,
                void* workspace, size_t workspace_bytes, void* stream);
size_t mg_conv_fwd_workspace(const mg_conv_geom* g);
/* dx = conv^T(dy, w) (+ bias, act)      data gradient of the convolution == ConvTranspose2d forward */
int mg_conv_dgrad(const mg_conv_geom* g, const float* dy, const float* w, const float* bias, float* dx, int act,
                  void* workspace, size_t workspace_bytes, void* stream);
size_t mg_conv_dgrad_workspace(const mg_conv_geom* g);
/* Layers with channel counts that are multiples of 16 run as batched GEMMs over Winograd-transformed operands:
 *   3x3 stride-1 pad-1 (ResnetBlock)            F(2x2,3x3), P = 16 positions, K = Ci,   T = B*(H/2)*(W/2) tiles (H, W even)
 *   4x4 stride-1 pad-2 (PatchGAN, H and W odd)  F(2x2,4x4), P = 25,           K = Ci,   T = B*((H+1)/2)*((W+1)/2)
 *   4x4 stride-2 pad-2 (PatchGAN, large only)   F(4x4,2x2) over the space-to-depth view, P = 25, K = 4*Ci,
 *                                               T = B*ceil(OH/4)*ceil(OW/4)
 * A training step runs forward, data gradient and weight gradient of a layer with the
 * same weights / activations / output gradient, so the caller may hold the transformed images and hand them to the _w
 * entry points (every pointer optional; NULL = transform internally, which is what the plain entry points do):
 *   u   U = G w G^T            P*Co*K floats   mg_conv_wino_prepare() fills it; read by fwd and dgrad
 *   v   V = B^T x B            P*T*K floats    WRITTEN by mg_conv_fwd_w, read by mg_conv_wgrad_w
 *   md  Md = A dy A^T          P*T*Co floats   WRITTEN by mg_conv_dgrad_w, read by mg_conv_wgrad_w
 * Sizes (never computed by the caller): mg_conv_wino_weights_bytes() and mg_conv_wino_tiles_bytes(g, 0 = v | 1 = md); 0 means
 * "this geometry / configuration does not use that image" and the pointer must stay NULL.  Every image must come
 * from the tensors passed in the same step. */
typedef struct {
    const float* u;
    float* v;
    float* md;
    const float* add;   /* mg_conv_dgrad_w only, any geometry: dx += add ([B,H,W,Ci]) -- the gradient a skip connection brings to
                         * the same tensor (ResnetBlock, models/networks.py:462), folded into the call's last kernel where the path
                         * allows and added by a separate pass otherwise; NULL for every other call */
    unsigned flags;     /* MG_TILES_*_FILLED (MG_PRECISION_F16 layers whose tiles are plain float16 copies): the copy is already in
                         * the buffer -- written by the kernel that produced x / dy (mg_instnorm_fwd_h, mg_instnorm_bwd_h,
                         * mg_conv_fwd_instnorm_h) -- so the call skips its own cast pass.  Paths that do not use the copy ignore it. */
} mg_wino_tiles;
#define MG_TILES_V_FILLED 1u    /* mg_conv_fwd_w: v holds float16(x) (MG_PRECISION_F16 implicit GEMMs) / B^T x B written by
                                   mg_conv_fwd_instnorm_next of the layer in front (float32 F(2x2,3x3) layers) */
#define MG_TILES_MD_FILLED 2u   /* mg_conv_dgrad_w: md holds float16(dy) */
size_t mg_conv_wino_weights_bytes(const mg_conv_geom* g);
size_t mg_conv_wino_tiles_bytes(const mg_conv_geom* g, int which);
int mg_conv_wino_prepare(const mg_conv_geom* g, const float* w, float* u, void* stream);
int mg_conv_fwd_w(const mg_conv_geom* g, const float* x, const float* w, const float* bias, float* y, int act,
                  void* workspace, size_t workspace_bytes, void* stream, const mg_wino_tiles* tiles);
/* y_raw = conv(x, w) + bias (mg_conv_fwd_w with no activation) followed by mg_instnorm_fwd(y_raw) -> y = act((y_raw - mean)
 * * rstd) + residual, mean / rstd [B][Co]: the [conv3x3, InstanceNorm2d(affine=False), ReLU] / [conv3x3, InstanceNorm2d, + x]
 * pairs of ResnetBlock (models/networks.py:440-462).  When the layer runs as Winograd F(2x2,3x3) and a sample's map is
 * <= 640 pixels, the inverse transform, the statistics and the normalisation are ONE kernel; otherwise the two calls are
 * made back to back -- same results either way (statistics in double, fixed reduction order).  y_raw is kept because the
 * InstanceNorm backward recomputes the normalised value from it; y_raw = NULL (inference) skips that store (the two-call path
 * then normalises in place).  workspace >= mg_conv_fwd_instnorm_workspace(g). */
size_t mg_conv_fwd_instnorm_workspace(const mg_conv_geom* g);
int mg_conv_fwd_instnorm_w(const mg_conv_geom* g, const float* x, const float* w, const float* bias, float* y_raw, float eps,
                           int act, const float* residual, float* y, float* mean, float* rstd, void* workspace,
                           size_t workspace_bytes, void* stream, const mg_wino_tiles* wt);
/* The backward mirror: for a layer with mg_conv_wino_md_from_norm_ok(g) != 0, mg_instnorm_bwd_wino_md turns the gradient gy at
 * the InstanceNorm output (y_raw, mean, rstd, act as saved by mg_conv_fwd_instnorm_w) straight into md = A dy A^T, the
 * Winograd image of the gradient at the convolution output (mg_conv_wino_tiles_bytes(g, 1) bytes) -- InstanceNorm backward
 * and the data gradient's first transform in one kernel; dy itself is never written.  mg_conv_dgrad_w(g, dy = NULL, ...,
 * wt->md = md) and mg_conv_wgrad_w(g, x = NULL, dy = NULL, ..., dbias = NULL, wt->v, wt->md) then start from the images. */
int mg_conv_wino_md_from_norm_ok(const mg_conv_geom* g);
int mg_instnorm_bwd_wino_md(const mg_conv_geom* g, const float* gy, const float* y_raw, const float* mean, const float* rstd,
                            int act, float* md, void* stream);
/* mg_conv_fwd_instnorm_w that also writes float16(y) (y16: B*OH*OW*Co halves, 8-byte aligned; NULL = none) for the next
 * MG_PRECISION_F16 layer's MG_TILES_V_FILLED.  MG_PRECISION_F16 geometries only (MG_ERR_UNSUPPORTED otherwise). */
int mg_conv_fwd_instnorm_h(const mg_conv_geom* g, const float* x, const float* w, const float* bias, float* y_raw, float eps,
                           int act, const float* residual, float* y, float* mean, float* rstd, void* workspace,
                           size_t workspace_bytes, void* stream, const mg_wino_tiles* tiles, void* y16);
/* mg_conv_fwd_instnorm_w that ALSO writes v_next = B^T y B, the Winograd input image of its own output y for a following 3x3
 * stride-1 pad-1 convolution over the same map (next_reflect: that layer's padding mode): ResnetBlock chains conv -> InstanceNorm ->
 * conv (models/networks.py:440-462), and the workgroup that normalises a slab of y holds exactly the pixels the next layer's
 * tiles need.  v_next: mg_conv_wino_tiles_bytes(g_next, 0) bytes (16 * T * Co floats); the next layer's call takes it as tiles->v
 * with MG_TILES_V_FILLED and skips its own input transform -- bit-identical results (the same float32 arithmetic on the same
 * values).  Only where mg_conv_wino_vnext_ok(g) != 0 (float32 F(2x2,3x3) layers on maps of <= 64 tiles per sample); MG_ERR_ARG
 * elsewhere.  Replaces nothing in the reference: it removes one launch and one read of y per trunk layer. */
int mg_conv_wino_vnext_ok(const mg_conv_geom* g);
int mg_conv_fwd_instnorm_next(const mg_conv_geom* g, const float* x, const float* w, const float* bias, float* y_raw, float eps,
                              int act, const float* residual, float* y, float* mean, float* rstd, void* workspace,
                              size_t workspace_bytes, void* stream, const mg_wino_tiles* tiles, float* v_next, int next_reflect);
int mg_conv_dgrad_w(const mg_conv_geom* g, const float* dy, const float* w, const float* bias, float* dx, int act,
                    void* workspace, size_t workspace_bytes, void* stream, const mg_wino_tiles* tiles);
/* dw [Co, KH, KW, Ci] = sum over pixels; dbias [Co] (nullable) = column sums of dy.
 * accumulate != 0 adds into dw / dbias instead of overwriting.  workspace: mg_conv_wgrad_workspace() bytes. */
int mg_conv_wgrad(const mg_conv_geom* g, const float* x, const float* dy, float* dw, float* dbias, int accumulate,
                  void* workspace, size_t workspace_bytes, void* stream);
size_t mg_conv_wgrad_workspace(const mg_conv_geom* g);
int mg_conv_wgrad_w(const mg_conv_geom* g, const float* x, const float* dy, float* dw, float* dbias, int accumulate,
                    void* workspace, size_t workspace_bytes, void* stream, const mg_wino_tiles* tiles);
/* --fp16 (train.py:183-199, torch.cuda.amp.GradScaler.unscale_'s inf / nan check): where mg_conv_wgrad_checks_finite(g) != 0 the
 * weight-gradient kernel inspects its own results -- mg_conv_wgrad_chk stores 1.0f to *found_inf (device memory; left alone
 * otherwise) when an element of dw, after accumulation, is inf or nan, so the caller can leave dw out of its mg_scaler_check
 * pass (the 2048-channel trunk layers of configs[2] / [3]: 151 MB each).  found_inf = NULL: same as mg_conv_wgrad_w.
 * found_inf != NULL on a geometry without the check: MG_ERR_UNSUPPORTED. */
int mg_conv_wgrad_checks_finite(const mg_conv_geom* g);
int mg_conv_wgrad_chk(const mg_conv_geom* g, const float* x, const float* dy, float* dw, float* dbias, int accumulate,
                      void* workspace, size_t workspace_bytes, void* stream, const mg_wino_tiles* tiles, float* found_inf);
/* Round 6: the same weight gradient STORED as float16 (dw16: Co * KH * KW * Ci halves, OHWI) -- the dtype an autocast layer's
 * weight gradient has in the reference (train.py:161-164).  Only where mg_conv_wgrad_h16_ok(g) != 0 (the short-reduction GEMM
 * of the weight-streaming trunk layers, whose 151 MB float32 output stream is what bounds it); found_inf (nullable) as above,
 * with float16's overflow criterion; accumulate adds into the float16 values.  No bias gradient (those layers' biases feed an
 * InstanceNorm: identically zero). */
int mg_conv_wgrad_h16_ok(const mg_conv_geom* g);
int mg_conv_wgrad_h16(const mg_conv_geom* g, const float* x, const float* dy, void* dw16, int accumulate, void* workspace,
                      size_t workspace_bytes, void* stream, float* found_inf);
/* Round 3 (single process, float32): the weight side of a Winograd F(2x2,3x3) layer in one pass.  Instead of writing dw,
 * the call applies torch.optim.Adam's update (pix2pixHD_model.py:350-351) to (w, m, v) straight from the Winograd-domain
 * gradient and leaves u = G w G^T of the UPDATED weights (mg_conv_wino_weights_bytes(g) bytes) for the next forward:
 * the gradient never exists in HBM, the weights make one round trip instead of three.  state: the optimiser's device
 * clock double[6] = {step, lr, lr / (1 - b1^step), sqrt(1 - b2^step), 1 - b1^(step+1), sqrt(1 - b2^(step+1))} BEFORE this
 * iteration's mg_adam_tick (the call reads [1], [4], [5]; mg_adam_tick / mg_adam_tick_amp maintain them, mg_adam_prime
 * initialises [4], [5] for a fresh or restored clock).  Same results as mg_conv_wgrad_w + mg_adam_step_dev +
 * mg_conv_wino_prepare, bit for bit.  The caller guarantees what the fusion assumes: one optimiser step per backward
 * pass, no other contribution to this weight's gradient, no gradient reduction across processes, no loss scaling. */
typedef struct {
    float* m;
    float* v;
    float* u;
    const double* state;
    float beta1, beta2, eps, grad_scale;
} mg_wino_adam;
int mg_conv_wgrad_adam_ok(const mg_conv_geom* g);
int mg_conv_wgrad_adam_w(const mg_conv_geom* g, const float* x, const float* dy, float* w, const mg_wino_adam* adam,
                         void* workspace, size_t workspace_bytes, void* stream, const mg_wino_tiles* tiles);
/* Name of the kernel instance a pass (0 fwd, 1 dgrad, 2 wgrad) launches for this geometry -- the symbol
 * rocprofv3 reports -- so bench.py can attribute event-timed launches per kernel.  out: host buffer >= 64 B. */
int mg_conv_plan_name(int pass, const mg_conv_geom* g, char* out, int out_len);
/* K splits of that launch where the pass takes the LDS-DMA implicit-GEMM kernels (conv_{fwd,dgrad,wgrad}_dma_kernel; 1 = unsplit),
 * 0 for every other kernel family: lets a host-only test pin the planner (workgroups = tiles x splits). */
int mg_conv_plan_splits(int pass, const mg_conv_geom* g);
/* FLOPs issued by that kernel for this geometry (direct: 2*MACs; Winograd layers: the P batched GEMMs, 2*P*T*Co*K). */
double mg_conv_plan_flops(int pass, const mg_conv_geom* g);
/* One-shot timing probe for bench.py: the next mg_conv_{fwd,dgrad,wgrad} call records hipEvent e0 / e1 on its launch
 * stream immediately around its main GEMM kernel (not around transforms / split-K epilogues), then disarms. */
void mg_probe_arm(void* e0, void* e1);
/* column sums: out[c] (+)= sum_m a[m, c]  -- bias gradients of ConvTranspose2d layers */
int mg_colsum(const float* a, long long M, int C, float* out, int accumulate, void* workspace,
              size_t workspace_bytes, void* stream);
size_t mg_colsum_workspace(long long M, int C);

/* ------------------------------------------------------------------------------------------
 * K5  InstanceNorm2d(affine=False, eps) with fused activation / residual (networks.py:26, 306,
 *     462, 650-666).  x, y: [B, HW, C].  mean / rstd: [B, C] (saved for backward).
 *     y = act((x - mean) * rstd) (+ residual).
 */
int mg_instnorm_fwd(const float* x, int B, int HW, int C, float eps, int act, const float* residual, float* y,
                    float* mean, float* rstd, void* workspace, size_t workspace_bytes, void* stream);
/* dx from dy (gradient wrt y, excluding the residual branch), the saved x, mean, rstd. */
int mg_instnorm_bwd(const float* dy, const float* x, const float* mean, const float* rstd, int B, int HW, int C,
                    int act, float* dx, void* workspace, size_t workspace_bytes, void* stream);
size_t mg_instnorm_workspace(int B, int HW, int C);
/* --fp16 (train.py:161-164): the same two calls, additionally writing the float16 copy of their output (y16 / dx16: B*HW*C
 * halves, 8-byte aligned, C % 4 == 0; NULL = none) -- the operand staging the next autocast convolution would otherwise do in
 * a cast pass of its own (hand it over with MG_TILES_V_FILLED / MG_TILES_MD_FILLED).  The float32 results are unchanged. */
int mg_instnorm_fwd_h(const float* x, int B, int HW, int C, float eps, int act, const float* residual, float* y,
                      float* mean, float* rstd, void* workspace, size_t workspace_bytes, void* stream, void* y16);
int mg_instnorm_bwd_h(const float* dy, const float* x, const float* mean, const float* rstd, int B, int HW, int C,
                      int act, float* dx, void* workspace, size_t workspace_bytes, void* stream, void* dx16);
/* Round 6: the same passes with a SECOND gradient of the tensor (dy2, nullable): the kernels read dy + dy2 (one float32 addition per
 * element, exactly autograd's accumulation add) -- a discriminator feature map has two consumers, the next layer and the
 * feature-matching loss (pix2pixHD_model.py:443-451), and the loss's gradient joins the layer's here instead of in an add launch. */
int mg_instnorm_bwd_add(const float* dy, const float* dy2, const float* x, const float* mean, const float* rstd, int B, int HW, int C,
                        int act, float* dx, void* workspace, size_t workspace_bytes, void* stream, void* dx16);

/* ------------------------------------------------------------------------------------------
 * K10  bottleneck-transformer block pieces (bottleneck_transformer_pytorch==0.1.4 BottleStack, call sites
 *      models/networks.py:232-235, 341-344; third-party, parity unpinned).  The block's 1x1 convolutions use
 *      mg_conv_*.
 *  BatchNorm2d over R = B*H*W rows of an NHWC tensor [R, C]:
 *      training: batch statistics (biased var for normalisation, unbiased into running_var, momentum m);
 *      eval: running statistics.  y = act(gamma * xhat + beta + residual)  (act in {NONE, RELU}; residual nullable).
 */
/*  Round 3: statistics and application are separate launches with per-slice double-precision partial sums
 *  [mg_batchnorm_slices()][2][C] between them (mg_batchnorm_workspace(C) bytes): sums -> (a data-parallel run may
 *  all-reduce the partials here: SyncBN, then count = rows of the whole batch) -> apply.
 *      mg_batchnorm_sums      partials of (sum x, sum x^2)
 *      mg_batchnorm_fwd       training: statistics from `sums` / `count`;  eval: running statistics (sums may be NULL)
 *      mg_batchnorm_bwd_sums  partials of (sum g, sum g * xhat), g = dy masked by the ReLU of y
 *      mg_batchnorm_bwd       dgamma / dbeta from local_sums (this rank's samples), dx from batch_sums / count */
size_t mg_batchnorm_workspace(int C);
int mg_batchnorm_slices(void);
int mg_batchnorm_sums(const float* x, int R, int C, void* sums, void* stream);
int mg_batchnorm_fwd(const float* x, int R, int C, float eps, float momentum, int training, const float* gamma,
                     const float* beta, float* running_mean, float* running_var, const float* residual, int act,
                     float* y, float* save_mean, float* save_rstd, const void* sums, double count, void* stream);
int mg_batchnorm_bwd_sums(const float* dy, const float* x, const float* y, int R, int C, const float* mean,
                          const float* rstd, int act, void* sums, void* stream);
int mg_batchnorm_bwd(const float* dy, const float* x, const float* y, int R, int C, const float* gamma,
                     const float* mean, const float* rstd, int act, int training, float* dx, float* dresidual,
                     float* dgamma, float* dbeta, int accumulate, const void* local_sums, const void* batch_sums,
                     double count, void* stream);
/*  Multi-head self attention with absolute position embeddings (rel_pos_emb=False):
 *      qkv [B, fh*fw, 3*heads*d] (channel = which*heads*d + head*d + dd), emb_h [fh, d], emb_w [fw, d];
 *      sim = (q * d^-0.5) (k + emb_h[y] + emb_w[x])^T, out [B, fh*fw, heads*d] = softmax(sim) v.
 *      P [B, heads, n, n] receives the probabilities (saved for backward).  fh*fw <= 128, d <= 128. */
int mg_attention_fwd(const float* qkv, const float* emb_h, const float* emb_w, int B, int fh, int fw, int heads, int d,
                     float* out, float* P, void* stream);
int mg_attention_bwd(const float* qkv, const float* emb_h, const float* emb_w, const float* dout, const float* P, int B,
                     int fh, int fw, int heads, int d, float* dqkv, float* demb_h, float* demb_w, int accumulate,
                     void* workspace, size_t workspace_bytes, void* stream);
size_t mg_attention_bwd_workspace(int B, int fh, int fw, int heads, int d);

/* K7  elementwise activation backward for conv epilogues: dx = dy * act'(y)  (in place allowed) */
int mg_act_bwd(const float* dy, const float* y, float* dx, long long n, int act, void* stream);
int mg_act_bwd_add(const float* dy, const float* dy2, const float* y, float* dx, long long n, int act, void* stream);
/* out = a + b (residual joins outside a norm), in place allowed */
int mg_add(const float* a, const float* b, float* out, long long n, void* stream);

/* K8  AvgPool2d(3, stride 2, padding 1, count_include_pad=False) (networks.py:249-250, 525-526) */
int mg_avgpool3s2_fwd(const float* x, int B, int H, int W, int C, float* y, void* stream);
int mg_avgpool3s2_bwd(const float* dy, int B, int H, int W, int C, float* dx, void* stream);
/* K9  nearest x2 upsample (networks.py:396) */
int mg_upsample2x_fwd(const float* x, int B, int H, int W, int C, float* y, void* stream);
int mg_upsample2x_bwd(const float* dy, int B, int H, int W, int C, float* dx, void* stream);

/* discriminator input assembly (pix2pixHD_model.py:420-424, 439-440):
 *   out [B, HW, 3] = (lr, s, 2|s| + nr0);  backward: ds = g1 + 2*sign(s)*g2 */
int mg_dinput_fwd(const float* lr, const float* s, long long n, float nr0, float* out, void* stream);
/* ... and without --abs_spectro --arcsinh_transform (pix2pixHD_model.py:425-427, 440 else branch): the plain
 *   torch.cat((a, b), dim=1) of NHWC tensors, n pixels of Ca / Cb channels, and its backward (ga / gb nullable) */
int mg_cat2_fwd(const float* a, int Ca, const float* b, int Cb, long long n, float* out, void* stream);
int mg_cat2_bwd(const float* g, int Ca, int Cb, long long n, float* ga, float* gb, void* stream);
int mg_dinput_bwd(const float* dout, const float* s, long long n, float* ds, void* stream);
/* generator input pair [n, 2] = (s, 2|s| + nr0) from a spectrogram (pix2pixHD_model.py:400-402) */
int mg_pair_fwd(const float* s, long long n, float nr0, float* out, void* stream);
/* mean_std[0] = stats[0] / n, mean_std[1] = sqrt(max(0, (stats[1] - stats[0]^2 / n) / (n - 1))) -- the `mean` / `std` entries of
 * to_spectro's norm_param (models/pix2pixHD_model.py:108-109: log_spectro.mean(), log_spectro.var().sqrt()) from the {sum, sum of
 * squares} K1 accumulates, in float64 like the torch expression it replaces. */
int mg_stats_finalize(const double* stats, long long n, float* mean_std, void* stream);

/* ------------------------------------------------------------------------------------------
 * K11  losses (networks.py:127-137 GANLoss/LSGAN, pix2pixHD_model.py:443-451 feature matching).
 *   fwd: loss[0] (+)= scale * mean((pred - target)^2)   |   scale * mean(|a - b|)
 *        (two-stage fixed-order reduction in double: deterministic; workspace >= mg_loss_workspace()).
 *   bwd: grad = d loss / d pred (resp. a) * grad_out[0]  (grad_out: device scalar, nullable == 1).
 */
size_t mg_loss_workspace(void);
int mg_mse_const_fwd(const float* pred, long long n, float target, float scale, float* loss, int accumulate,
                     void* workspace, void* stream);
int mg_mse_const_bwd(const float* pred, long long n, float target, float scale, const float* grad_out, float* grad,
                     void* stream);
/* --no_lsgan (networks.py:106-109, 676-677): nn.BCELoss against a constant label on probabilities (log terms clamped at -100,
 * gradient (p - t) / max((1 - p) p, 1e-12) -- torch's rules), and the nn.Sigmoid that produces them. */
int mg_bce_const_fwd(const float* pred, long long n, float target, float scale, float* loss, int accumulate,
                     void* workspace, void* stream);
int mg_bce_const_bwd(const float* pred, long long n, float target, float scale, const float* grad_out, float* grad,
                     void* stream);
int mg_sigmoid_fwd(const float* x, float* y, long long n, void* stream);
int mg_sigmoid_bwd(const float* dy, const float* y, float* dx, long long n, void* stream);
int mg_l1_fwd(const float* a, const float* b, long long n, float scale, float* loss, int accumulate, void* workspace,
              void* stream);
int mg_l1_bwd(const float* a, const float* b, long long n, float scale, const float* grad_out, float* grad_a,
              void* stream);
/* The same three losses over a LIST of tensors, one launch per stage (kind 0: mean((a - target)^2), 1: mean(|a - b|), 2: BCE against
 * the constant label `target`): loss (+)= sum_i scale * loss_i -- the feature-matching sum over the discriminators' intermediate
 * layers (models/pix2pixHD_model.py:440-449) and the per-scale GAN terms.  Bit-identical to the accumulate-in-place sequence of
 * the single-tensor calls in list order.  Backward: grad[i][0..n) = d(scale * loss_i)/da_i * grad_out[0]; additionally
 * grad[i][n .. n + zero_tail) = 0 (the other half of a stacked [fake; real] batch).  items: HOST array, count <= MG_LOSS_MAX_ITEMS. */
#define MG_LOSS_MAX_ITEMS 16
typedef struct {
    const float* a;
    const float* b;        /* kind 1 only */
    float* grad;           /* backward only */
    long long n;
    long long zero_tail;   /* backward only */
} mg_loss_item;
size_t mg_loss_multi_workspace(void);
int mg_loss_multi_fwd(int kind, const mg_loss_item* items, int count, float target, float scale, float* loss, int accumulate,
                      void* workspace, size_t workspace_bytes, void* stream);
int mg_loss_multi_bwd(int kind, const mg_loss_item* items, int count, float target, float scale, const float* grad_out,
                      void* stream);

/* K12  fused Adam over one flat float32 buffer (torch.optim.Adam semantics, no weight decay / amsgrad):
 *      pix2pixHD_model.py:350-351, 363-364; train.py:186-202.  step is 1-based.
 *      grad_scale multiplies g first (1/world_size for DDP, 1/loss_scale for AMP). */
int mg_adam_step(float* p, const float* g, float* m, float* v, long long n, float lr, float beta1, float beta2,
                 float eps, int step, float grad_scale, void* stream);

/* hipGraph-replayable variant: the optimiser clock lives in HBM.  state = double[6] {step, lr, lr/(1-beta1^step),
 * sqrt(1-beta2^step), 1-beta1^(step+1), sqrt(1-beta2^(step+1))}; mg_adam_tick advances it on the device (host writes state[1] = lr when the schedule
 * changes), mg_adam_step_dev reads it -- no step-dependent value is baked into a kernel argument. */
int mg_adam_tick(double* state, float beta1, float beta2, void* stream);
/* state[4] = 1 - beta1^(step+1), state[5] = sqrt(1 - beta2^(step+1)) for the clock as it stands (a fresh or restored clock; every
 * tick maintains them afterwards): the terms mg_conv_wgrad_adam_w reads.  The clock is double[6]. */
int mg_adam_prime(double* state, float beta1, float beta2, void* stream);
int mg_adam_step_dev(float* p, const float* g, float* m, float* v, long long n, const double* state, float beta1,
                     float beta2, float eps, float grad_scale, void* stream);

/* torch.cuda.amp.GradScaler on the device (train.py:65-70: one scaler; :183-199: scale(loss).backward(),
 * scaler.step(optimizer) per optimiser, scaler.update() once per iteration) -- no host synchronisation, so the AMP
 * step stays hipGraph-capturable.  scaler: float[2 + MG_SCALER_SLOTS] = {loss scale S, growth tracker,
 * found_inf[slot]...} (one slot per optimiser; the caller initialises {init_scale, 0, 0...}).
 *   mg_scaler_check   found_inf[slot] = 1 if any gradient in g[0..n) is inf / nan  (GradScaler.unscale_'s check)
 *   mg_adam_tick_amp / mg_adam_step_amp   mg_adam_tick / mg_adam_step_dev with the gradients divided by S, and a
 *                     no-op -- parameters, moments and step counter untouched -- when found_inf[slot] != 0
 *   mg_scaler_update  any found_inf: S *= backoff_factor, tracker = 0; else tracker += 1 and, at growth_interval,
 *                     S *= growth_factor, tracker = 0.  Clears every found_inf slot. */
#define MG_SCALER_SLOTS 2
int mg_scaler_check(const float* g, long long n, float* scaler, int slot, void* stream);
int mg_scaler_update(float* scaler, float growth_factor, float backoff_factor, int growth_interval, void* stream);
int mg_adam_tick_amp(double* state, float beta1, float beta2, const float* scaler, int slot, void* stream);
int mg_adam_step_amp(float* p, const float* g, float* m, float* v, long long n, const double* state, float beta1,
                     float beta2, float eps, float grad_scale, const float* scaler, int slot, void* stream);
/* mg_adam_step_amp (scaler != NULL) or mg_adam_step_dev (scaler == NULL) that also writes p16[i] = (float16) p[i] for the
 * updated parameters: the float16 weight operand of the autocast convolutions (train.py --fp16; autocast casts every
 * weight per call, pix2pixHD_model.py:285-301) is refreshed by the optimiser pass that already streams the parameters
 * instead of by one cast launch per layer.  A skipped step (found_inf) leaves p and p16 untouched. */
int mg_adam_step_h(float* p, const float* g, float* m, float* v, void* p16, long long n, const double* state, float beta1,
                   float beta2, float eps, float grad_scale, const float* scaler, int slot, void* stream);

/* Round 6 -- the --fp16 optimiser passes over a SEGMENTED gradient arena (train.py:161-164, 183-199).  Under torch.autocast a
 * convolution's weight / bias gradient is a float16 tensor (the cast's backward widens it into the float32 .grad): its values are
 * float16-rounded and overflow at 65504 -- which is what the reference's GradScaler backs off on.  A segment = a run of the arena
 * [off, off + n) (multiples of 8 elements; device array `segs`) and how its gradient is carried:
 *   MG_GRAD_F32       float32 in g, used as stored (BatchNorm and position-embedding parameters: float32 in the reference too)
 *   MG_GRAD_AUTOCAST  float32 in g, rounded through float16 where it is consumed: finite iff |v| < 65520, Adam reads half(v)
 *   MG_GRAD_F16       float16 in g16 at the same element index, written by mg_conv_wgrad_h16 (half the bytes in both passes)
 * skip_check != 0: the producing kernel already ran the inf / nan test on this segment (mg_conv_wgrad_h16 / mg_conv_wgrad_chk).
 * mg_scaler_check_segs == mg_scaler_check, mg_adam_step_segs == mg_adam_step_h (p16 nullable), one launch each over all
 * segments; n_total (sum of the lengths) sizes the grid. */
#define MG_GRAD_F32 0
#define MG_GRAD_AUTOCAST 1
#define MG_GRAD_F16 2
typedef struct {
    long long off;
    long long n;
    int mode;
    int skip_check;
} mg_grad_seg;
int mg_scaler_check_segs(const float* g, const void* g16, const mg_grad_seg* segs, int nsegs, long long n_total, float* scaler,
                         int slot, void* stream);
int mg_adam_step_segs(float* p, const float* g, const void* g16, float* m, float* v, void* p16, const mg_grad_seg* segs, int nsegs,
                      long long n_total, const double* state, float beta1, float beta2, float eps, float grad_scale,
                      const float* scaler, int slot, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MDCTGAN_HIP_H */
