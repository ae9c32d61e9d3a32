"""MDCT4 / IMDCT4 with the reference's call signatures (models/mdct.py:359-489), executed by the
gfx950 kernels K1 / K2 in csrc/mdct.hip through the C ABI (include/mdctgan_hip.h).

Differences from the reference, all deliberate and documented in DESIGN.md:

* arithmetic is float32 end to end (the DCT-IV runs on the f32 MFMA pipe); the reference promotes to
  complex128 after its float32 window multiply.  Tolerances: tests/test_mdct_gpu.py.
* the returned spectrogram / waveform dtype is float32 unless ``dtype=torch.float64`` is requested
  (then the float32 result is widened, for code that relies on the reference's float64 outputs).
* frame padding follows the signal length T (the reference uses ``len(signal)`` == batch size for
  2-D input, SURVEY A2'); identical for every legal segment length (T % hop == 0).
* geometry: win_length == n_fft == 2 * hop_length == 512 (the hot path) runs the fused kernels K1 / K2 (csrc/mdct.hip:
  TDAC fold + 256-point DCT-IV); every other legal geometry of the reference (win_length <= n_fft, hop_length <=
  win_length, e.g. the class default n_fft = 2048) runs the generic path of csrc/codec_generic.hip: framing + window, the
  [frames, win] x [win, n_fft/2] cosine contraction as a dense exact-float32 MFMA GEMM (the 1x1 case of mg_conv_fwd),
  window + overlap-add.  There is no eager / CPU fallback.
"""
from __future__ import annotations

import math

import torch

from . import _lib


def kbdwin(N: int, beta: float = 12.0, device="cpu") -> torch.Tensor:
    """Kaiser-Bessel-derived window (reference: util/util.py:179-186).  Host-side table: built with the
    same float32 op chain on the CPU so it is bit-identical to the reference's, then moved."""
    assert N % 2 == 0, "N must be even"
    w = torch.kaiser_window(window_length=N // 2 + 1, beta=beta * torch.pi, periodic=False, dtype=torch.float32)
    half = torch.sqrt(torch.cumsum(w, dim=0) / w.sum())[:-1]
    return torch.cat((half, half.flip(dims=(0,))), dim=0).to(device)


_dct4_cache = {}


def dct4_table(m: int, device) -> torch.Tensor:
    """D4[n, k] = cos(pi/M (n + 1/2)(k + 1/2)) evaluated in float64 on the host, rounded once to float32 -- followed, in the
    same buffer, by the stage-matrix image the factored kernels load (mg_dct4_image, mg_dct4_image_floats(2 m) floats; include/mdctgan_hip.h: the `dct4_image` argument of
    mg_mdct4_forward / mg_imdct4_forward).  Returns the flat tensor; ``dct4_table(m, dev)[:m * m].view(m, m)`` is the table
    itself, ``dct4_image(t, m)`` the image part (None for geometries without fused kernels)."""
    key = (m, str(device))
    t = _dct4_cache.get(key)
    if t is None:
        n = torch.arange(m, dtype=torch.float64) + 0.5
        tab = torch.cos((math.pi / m) * torch.outer(n, n)).to(torch.float32)
        lib = _lib.load() if m == 256 else None          # the fused n_fft = 512 kernels are the only users of the images
        extra = int(lib.mg_dct4_image_floats(2 * m)) if lib is not None else 0
        t = torch.zeros(m * m + extra, dtype=torch.float32, device=device)
        t[:m * m] = tab.reshape(-1).to(device)
        if extra:
            _lib.check(lib.mg_dct4_image(_lib.ptr(t), t.data_ptr() + 4 * m * m, _lib.stream()), "mg_dct4_image")
        _dct4_cache[key] = t
    return t


def dct4_image(table: torch.Tensor, m: int):
    """The image part of a dct4_table() buffer as a raw pointer argument (None: no image behind the table)."""
    return (table.data_ptr() + 4 * m * m) if table.numel() > m * m else None


def _make_window(window, win_length, device):
    if window is None:
        window = torch.ones
    if callable(window):
        win_length = int(win_length)
        w = window(win_length)
    else:
        w = window
        win_length = len(window)
    return w.to(device=device, dtype=torch.float32).contiguous(), win_length


def _check_geometry(n_fft, hop_length, win_length):
    """True: the fused n_fft = 512 kernels apply; False: the generic path."""
    assert win_length <= n_fft, "Window lenth %d should be no more than fft length %d" % (win_length, n_fft)
    assert hop_length <= win_length, "You hopped more than one frame"
    if n_fft % 2 or n_fft < 4:
        raise NotImplementedError("n_fft must be even")
    return n_fft == 512 and win_length == n_fft and hop_length * 2 == n_fft


_mdct_tab_cache = {}


def mdct_table(n_fft: int, win_length: int, device, transposed: bool) -> torch.Tensor:
    """C[n, k] = cos(2 pi / N (n + 1/2 + N/4)(k + 1/2)), n < win_length, k < N/2 (float64 on the host, rounded once).
    transposed: [N/2][win] (rows are the k-contiguous weight rows of the forward GEMM); else [win][N/2] (inverse)."""
    key = (n_fft, win_length, str(device), transposed)
    t = _mdct_tab_cache.get(key)
    if t is None:
        n = torch.arange(win_length, dtype=torch.float64)[:, None] + 0.5 + n_fft / 4.0
        k = torch.arange(n_fft // 2, dtype=torch.float64)[None, :] + 0.5
        c = torch.cos((2.0 * math.pi / n_fft) * n * k)
        t = (c.t() if transposed else c).contiguous().to(torch.float32).to(device)
        _mdct_tab_cache[key] = t
    return t


def num_frames(T: int, win_length: int, hop_length: int, center: bool = True) -> int:
    """Frames MDCT4.forward produces for a T-sample signal (mdct.py:393-407 with the T-based tail padding, SURVEY A2')."""
    start = hop_length if center else 0
    end = start + (hop_length - T % hop_length if T % hop_length else 0)
    return (T + start + end - win_length) // hop_length + 1


def _dense(a2d: torch.Tensor, w2d: torch.Tensor) -> torch.Tensor:
    """[R, K] x [N, K]^T on the exact-float32 MFMA GEMM (a 1x1 convolution over R 'pixels')."""
    from . import ops
    R, K = a2d.shape
    g = ops.conv_geom(1, 1, R, K, w2d.shape[0], 1, 1, 1, 0, False)
    return ops.conv_fwd(g, a2d.view(1, 1, R, K), w2d.view(w2d.shape[0], 1, 1, K)).view(R, w2d.shape[0])


def mdct4_generic(audio, window, n_fft, hop_length, center=True, want_frames=False):
    """MDCT4.forward for any legal geometry: audio [B, T] -> (raw coefficients [B, F, n_fft/2], frames [B, F, win] | None)."""
    lib = _lib.load()
    audio = _lib.f32c(audio)
    B, T = audio.shape
    win = window.numel()
    F = num_frames(T, win, hop_length, center)
    if F <= 0:
        raise ValueError("signal of %d samples is too short for win_length=%d" % (T, win))
    frames = torch.empty(B, F, win, dtype=torch.float32, device=audio.device)
    _lib.check(lib.mg_frames_window(_lib.ptr(audio), B, T, win, hop_length, hop_length if center else 0, F,
                                    _lib.ptr(window), _lib.ptr(frames), _lib.stream()), "mg_frames_window")
    spec = _dense(frames.view(B * F, win), mdct_table(n_fft, win, audio.device, True)).view(B, F, n_fft // 2)
    return spec, (frames if want_frames else None)


def imdct4_generic(spec, window, n_fft, hop_length, center=True, out_length=None, out_dtype=torch.float32,
                   want_frames=False):
    """IMDCT4.forward for any legal geometry: raw coefficients [B, F, n_fft/2] -> (audio [B, T_out], frames | None)."""
    lib = _lib.load()
    spec = _lib.f32c(spec)
    B, F, M = spec.shape
    win = window.numel()
    y = _dense(spec.view(B * F, M), mdct_table(n_fft, win, spec.device, False)).view(B, F, win)
    full = (F - 1) * hop_length + win
    crop = win // 2 if center else 0
    t_out = full - crop - ((win + 1) // 2 if center else 0)          # signal[win//2 : -win//2]  (mdct.py:484-486)
    if out_length is not None:
        t_out = min(t_out, int(out_length))
    audio = torch.empty(B, t_out, dtype=out_dtype, device=spec.device)
    _lib.check(lib.mg_overlap_add(_lib.ptr(y), B, F, win, hop_length, n_fft, _lib.ptr(window), crop, _lib.ptr(audio), t_out,
                                  int(out_dtype == torch.float64), _lib.stream()), "mg_overlap_add")
    frames = None
    if want_frames:
        frames = y * window
    return audio, frames


def codec_forward(raw, *, codec, gain=1.0, alpha=0.6, min_value=1e-7, norm_range=(0.0, 1.0), src_range=(0.0, 1.0),
                  per_sample=False, want_pair=False, want_stats=False):
    """Audio2MDCT.normalize on raw coefficients [B, F, M] (any codec, csrc/codec_generic.hip) -> dict like mdct4_codec,
    spec [B, C, F, M] with C = 2 for the explicit encoding."""
    lib = _lib.load()
    raw = _lib.f32c(raw)
    B, F, M = raw.shape
    C = 2 if codec == _lib.MG_CODEC_EXPLICIT else 1
    dev = raw.device
    spec = torch.empty(B, C, F, M, dtype=torch.float32, device=dev)
    pair = torch.empty(B, F, M, 2, dtype=torch.float32, device=dev) if (want_pair and C == 1) else None
    stats = torch.empty(2, dtype=torch.float64, device=dev) if want_stats else None
    mn = mx = scratch = None
    if per_sample:
        mn = torch.empty(B * C, dtype=torch.float32, device=dev)
        mx = torch.empty(B * C, dtype=torch.float32, device=dev)
        scratch = torch.empty(2 * B * C, dtype=torch.int32, device=dev)
    _lib.check(lib.mg_codec_forward(_lib.ptr(raw), B, F * M, codec, gain, alpha, min_value, norm_range[0], norm_range[1],
                                    src_range[0], src_range[1], int(per_sample), _lib.ptr(spec), _lib.ptr(pair), _lib.ptr(mn),
                                    _lib.ptr(mx), _lib.ptr(scratch), _lib.ptr(stats), _lib.stream()), "mg_codec_forward")
    return {"spec4": spec, "spec": spec[:, 0] if C == 1 else None, "pair": pair, "frames": None,
            "min": mn.view(B, C) if per_sample else None, "max": mx.view(B, C) if per_sample else None, "stats": stats}


def codec_inverse(spec4, *, codec, gain=1.0, alpha=0.6, min_value=1e-7, norm_range=(0.0, 1.0), src_range=(0.0, 1.0),
                  min_b=None, max_b=None):
    """Audio2MDCT.denormalize (+ the explicit-encoding channel combination of to_audio): [B, C, F, M] -> raw [B, F, M]."""
    lib = _lib.load()
    spec4 = _lib.f32c(spec4)
    B, C, F, M = spec4.shape
    assert C == (2 if codec == _lib.MG_CODEC_EXPLICIT else 1)
    raw = torch.empty(B, F, M, dtype=torch.float32, device=spec4.device)
    if min_b is not None:
        min_b, max_b = _lib.f32c(min_b.reshape(-1)), _lib.f32c(max_b.reshape(-1))
        assert min_b.numel() == B * C and max_b.numel() == B * C
    _lib.check(lib.mg_codec_inverse(_lib.ptr(spec4), B, F * M, codec, gain, alpha, min_value, norm_range[0], norm_range[1],
                                    src_range[0], src_range[1], _lib.ptr(min_b), _lib.ptr(max_b), _lib.ptr(raw),
                                    _lib.stream()), "mg_codec_inverse")
    return raw


def mdct4_codec(audio, window, dct4, n_fft, *, codec=_lib.MG_CODEC_RAW, gain=1.0, norm_range=(0.0, 1.0),
                src_range=(0.0, 1.0), per_sample=False, want_pair=False, want_frames=False, want_stats=False):
    """K1 launcher.  audio [B, T] (device, float32) -> dict(spec [B,F,M], pair [B,F,M,2]|None, frames|None,
    min/max [B]|None, stats double[2]|None)."""
    lib = _lib.load()
    audio = _lib.f32c(audio)
    B, T = audio.shape
    M = n_fft // 2
    F = lib.mg_mdct4_num_frames(T, n_fft)
    dev = audio.device
    pair = torch.empty(B, F, M, 2, dtype=torch.float32, device=dev) if want_pair else None
    image = dct4_image(dct4, M)
    # with the pair the spectrogram is its channel 0 (a strided view): K1 then writes 393 216 B per clip instead of 526 848
    import os
    legacy = os.environ.get("MG_MDCT_CT") == "0" or "MG_MDCT_FT" in os.environ      # (the generic kernels, forced)
    pair_only = (want_pair and image is not None and not per_sample and not want_frames and codec == _lib.MG_CODEC_ARCSINH
                 and T % 4 == 0 and audio.data_ptr() % 16 == 0 and window.data_ptr() % 16 == 0 and not legacy
                 # the factored kernels address through 32-bit buffer offsets (csrc/mdct.hip: the same guards decide there);
                 # beyond them the dense-table kernel runs and needs a real spectrogram buffer
                 and B * F * M * 8 < (1 << 32) - (1 << 18) and B * T * 4 < (1 << 32))
    spec = pair[..., 0] if pair_only else torch.empty(B, F, M, dtype=torch.float32, device=dev)
    frames = torch.empty(B, F, n_fft, dtype=torch.float32, device=dev) if want_frames else None
    stats = torch.empty(2, dtype=torch.float64, device=dev) if want_stats else None
    mn = mx = scratch = None
    if per_sample:
        mn = torch.empty(B, dtype=torch.float32, device=dev)
        mx = torch.empty(B, dtype=torch.float32, device=dev)
        scratch = torch.empty(2 * B, dtype=torch.int32, device=dev)
    rc = lib.mg_mdct4_forward(_lib.ptr(audio), B, T, n_fft, _lib.ptr(window), _lib.ptr(dct4), image, codec, gain,
                              norm_range[0], norm_range[1], src_range[0], src_range[1], int(per_sample),
                              None if pair_only else _lib.ptr(spec), _lib.ptr(pair), _lib.ptr(frames), _lib.ptr(mn), _lib.ptr(mx),
                              _lib.ptr(stats), _lib.ptr(scratch), _lib.stream())
    _lib.check(rc, "mg_mdct4_forward")
    return {"spec": spec, "pair": pair, "frames": frames, "min": mn, "max": mx, "stats": stats}


def imdct4_codec(spec, window, dct4, n_fft, *, codec=_lib.MG_CODEC_RAW, gain=1.0, norm_range=(0.0, 1.0),
                 src_range=(0.0, 1.0), min_b=None, max_b=None, out_length=None, out_dtype=torch.float32,
                 want_frames=False, stitch=None):
    """K2 launcher.  spec [B, F, M] (device) -> (audio [B, T_out], frames|None).

    stitch = (out, gen_overlap, first_seg[, segment_length]): the clips are segments first_seg.. of ONE waveform and K2's overlap-add store writes them
    straight into `out` [mg_stitch_length(n_seg, T_out, gen_overlap)] with generate_audio.py:40-53's cross-fade (mg_imdct4_stitched);
    returns (out, None).  The batch with first_seg == 0 clears `out` when gen_overlap > 0."""
    lib = _lib.load()
    spec = _lib.f32c(spec)
    B, F, M = spec.shape
    t_out = (F - 1) * M
    if out_length is not None:
        t_out = min(t_out, int(out_length))
    if stitch is not None:
        out, overlap, first = stitch[:3]
        if len(stitch) > 3 and int(stitch[3]) != t_out:
            # the caller sized `out` for segments of stitch[3] samples; the spectrogram handed in decodes to t_out: the store's
            # bounds check would drop samples or leave part of `out` unwritten without a word
            raise ValueError("stitched K2: the output was sized for %d-sample segments, the spectrogram decodes to %d"
                             % (int(stitch[3]), t_out))
        if want_frames or out.dtype != out_dtype or not out.is_contiguous():
            raise ValueError("stitched K2: contiguous output of the requested dtype, no synthesis frames")
        if min_b is not None:
            min_b, max_b = _lib.f32c(min_b.reshape(-1)), _lib.f32c(max_b.reshape(-1))
            assert min_b.numel() == B and max_b.numel() == B
        rc = lib.mg_imdct4_stitched(_lib.ptr(spec), B, F, n_fft, _lib.ptr(window), _lib.ptr(dct4), dct4_image(dct4, M), codec, gain,
                                    norm_range[0], norm_range[1], src_range[0], src_range[1], _lib.ptr(min_b), _lib.ptr(max_b),
                                    _lib.ptr(out), out.numel(), t_out, int(overlap), int(first), int(first == 0),
                                    int(out_dtype == torch.float64), _lib.stream())
        _lib.check(rc, "mg_imdct4_stitched")
        return out, None
    audio = torch.empty(B, t_out, dtype=out_dtype, device=spec.device)
    frames = torch.empty(B, F, n_fft, dtype=torch.float32, device=spec.device) if want_frames else None
    if min_b is not None:
        min_b, max_b = _lib.f32c(min_b.reshape(-1)), _lib.f32c(max_b.reshape(-1))
        assert min_b.numel() == B and max_b.numel() == B
    rc = lib.mg_imdct4_forward(_lib.ptr(spec), B, F, n_fft, _lib.ptr(window), _lib.ptr(dct4), dct4_image(dct4, M), codec, gain,
                               norm_range[0], norm_range[1], src_range[0], src_range[1], _lib.ptr(min_b),
                               _lib.ptr(max_b), _lib.ptr(audio), t_out, int(out_dtype == torch.float64),
                               _lib.ptr(frames), _lib.stream())
    _lib.check(rc, "mg_imdct4_forward")
    return audio, frames


class MDCT4(torch.nn.Module):
    """models/mdct.py:359-425.  forward(signal, return_frames=False) -> (spec [..., F, n_fft/2], frames)."""

    def __init__(self, n_fft=2048, hop_length=None, win_length=None, window=None, center=True,
                 pad_mode="constant", device="cuda", dtype=torch.float32) -> None:
        super().__init__()
        self.n_fft, self.pad_mode, self.device, self.hop_length, self.center = n_fft, pad_mode, device, hop_length, center
        self.window, self.win_length = _make_window(window, win_length, device)
        self.fused = _check_geometry(self.n_fft, self.hop_length, self.win_length) and center
        if pad_mode != "constant":
            raise NotImplementedError("HIP MDCT4 implements zero ('constant') padding")
        self.out_dtype = dtype

    def forward(self, signal, return_frames: bool = False):
        lead = signal.shape[:-1]
        x = signal.reshape(-1, signal.shape[-1])
        if self.window.device != x.device:
            self.window = self.window.to(x.device)
        if self.fused:
            r = mdct4_codec(x, self.window, dct4_table(self.n_fft // 2, x.device), self.n_fft, want_frames=return_frames)
            sp, fr = r["spec"], r["frames"]
        else:
            sp, fr = mdct4_generic(x, self.window, self.n_fft, self.hop_length, self.center, return_frames)
        spec = sp.reshape(*lead, *sp.shape[1:]).to(self.out_dtype)
        frames = fr.reshape(*lead, *fr.shape[1:]) if return_frames else torch.empty(1)
        return spec, frames


class IMDCT4(torch.nn.Module):
    """models/mdct.py:428-489.  forward(spec [B, F, n_fft/2], return_frames=False) -> (audio [B,1,1,T], frames)."""

    def __init__(self, n_fft=2048, hop_length=None, win_length=None, window=None, center=True,
                 pad_mode="constant", out_length=None, device="cuda", dtype=torch.float32) -> None:
        super().__init__()
        self.n_fft, self.pad_mode, self.device, self.hop_length = n_fft, pad_mode, device, hop_length
        self.center, self.out_length = center, out_length
        self.window, self.win_length = _make_window(window, win_length, device)
        self.fused = _check_geometry(self.n_fft, self.hop_length, self.win_length) and center
        self.out_dtype = dtype

    def forward(self, signal, return_frames: bool = False):
        assert signal.dim() == 3, "Only tensors shaped in BHW are supported, got tensor of shape %s" % (
            str(signal.size()))
        assert signal.size()[-1] == self.n_fft // 2, \
            "The last dim of input tensor should match the n_fft. Expected %d ,got %d" % (self.n_fft, signal.size()[-1])
        if self.window.device != signal.device:
            self.window = self.window.to(signal.device)
        if self.fused:
            audio, frames = imdct4_codec(signal, self.window, dct4_table(self.n_fft // 2, signal.device), self.n_fft,
                                         out_length=self.out_length, out_dtype=self.out_dtype, want_frames=return_frames)
        else:
            audio, frames = imdct4_generic(signal, self.window, self.n_fft, self.hop_length, self.center, self.out_length,
                                           self.out_dtype, return_frames)
        return audio[:, None, None, :], (frames if return_frames else torch.zeros(1))
