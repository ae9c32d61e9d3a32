"""Oracle (test infrastructure): KBD window, MDCT4 / IMDCT4 and the arcsinh /
range-norm codec, restated in numpy float64.

Follows the reference's arithmetic, not its FFT mechanics:

* ``kbd_window``      <- util/util.py:179-186 (kbdwin)
* ``mdct4``           <- models/mdct.py:392-425 (MDCT4.forward)
* ``imdct4``          <- models/mdct.py:457-489 (IMDCT4.forward)
* ``normalize``       <- models/pix2pixHD_model.py:83-125 (Audio2MDCT.normalize; the dB / explicit-encoding branches call
                         torchaudio's amplitude_to_DB / DB_to_amplitude, which are not installed here: restated from
                         their published formulas, parity of those two functions UNPINNED)
* ``denormalize``     <- models/pix2pixHD_model.py:127-137
* ``to_spectro``      <- models/pix2pixHD_model.py:32-81 (arcsinh / raw branches)
* ``to_audio``        <- models/pix2pixHD_model.py:139-163

The reference evaluates the transform as twiddle * FFT * twiddle in complex128
(mdct.py:387-390, 421-423 and 452-455, 464-466).  Taking the real part of that
product is algebraically the real cosine contraction

    X[f, k] = sum_n z_f[n] * cos((2*pi/N) * (n + 1/2 + N/4) * (k + 1/2))

with z_f the float32 windowed frame, and the inverse uses the same matrix
transposed.  The oracle evaluates exactly that in float64; agreement with the
reference's FFT path is pinned by tests/golden (<= 1e-9 relative).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.
"""
from __future__ import annotations

import functools
import math

import numpy as np

LN10 = math.log(10.0)


# --------------------------------------------------------------------------
# window
# --------------------------------------------------------------------------
def kbd_window_f64(n: int, beta: float = 12.0) -> np.ndarray:
    """Kaiser-Bessel-derived window evaluated in float64 (analytic cross-check only)."""
    if n % 2:
        raise AssertionError("N must be even")
    length = n // 2 + 1
    k = np.arange(length, dtype=np.float64)
    ratio = 2.0 * k / (length - 1) - 1.0
    w = np.i0(beta * np.pi * np.sqrt(np.clip(1.0 - ratio * ratio, 0.0, None))) / np.i0(beta * np.pi)
    half = np.sqrt(np.cumsum(w) / w.sum())[:-1]
    return np.concatenate([half, half[::-1]])


def kbd_window(n: int, beta: float = 12.0) -> np.ndarray:
    """Kaiser-Bessel-derived window, float32 [n].  util/util.py:179-186.

    w = kaiser(n/2 + 1, beta*pi) (symmetric); half = sqrt(cumsum(w)/sum(w))[:-1];
    result = concat(half, reversed(half)).  The reference builds it from float32
    torch ops on the CPU; a band-limited LR signal is sensitive to the last bit
    of the window (a 3e-7 window change moves empty-band bins by 5e-6 in the
    normalised spectrogram), so the oracle performs the same float32 op chain
    instead of rounding a float64 evaluation (``kbd_window_f64``, used as a
    cross-check to 3e-7).
    """
    import torch
    if n % 2:
        raise AssertionError("N must be even")
    w = torch.kaiser_window(window_length=n // 2 + 1, beta=beta * torch.pi, periodic=False, dtype=torch.float32)
    half = torch.sqrt(torch.cumsum(w, dim=0) / w.sum())[:-1]
    return torch.cat((half, half.flip(0))).numpy()


# --------------------------------------------------------------------------
# cosine matrices
# --------------------------------------------------------------------------
@functools.lru_cache(maxsize=8)
def mdct_matrix(n_fft: int) -> np.ndarray:
    """C[n, k] = cos((2*pi/N)(n + 1/2 + N/4)(k + 1/2)), float64 [N, N/2]."""
    n = np.arange(n_fft, dtype=np.float64)[:, None]
    k = np.arange(n_fft // 2, dtype=np.float64)[None, :]
    return np.cos((2.0 * np.pi / n_fft) * (n + 0.5 + n_fft / 4.0) * (k + 0.5))


@functools.lru_cache(maxsize=8)
def dct4_matrix(m: int) -> np.ndarray:
    """D4[n, k] = cos((pi/M)(n + 1/2)(k + 1/2)), float64 [M, M] (symmetric)."""
    n = np.arange(m, dtype=np.float64)
    return np.cos((np.pi / m) * np.outer(n + 0.5, n + 0.5))


# --------------------------------------------------------------------------
# framing
# --------------------------------------------------------------------------
def frame_signal(x: np.ndarray, win_length: int, hop: int, center: bool = True) -> np.ndarray:
    """Zero-pad and slice into overlapping frames.  mdct.py:393-407.

    NOTE (SURVEY A2'): the reference derives the tail padding from
    ``len(signal)`` which for a [B, T] input is B, not T.  For every legal
    segment length (T % hop == 0) both rules give no tail padding.  The oracle
    (and the HIP path) use the T-based rule.
    """
    x = np.asarray(x)
    t = x.shape[-1]
    start = hop if center else 0
    end = start
    if t % hop:
        end = start + hop - t % hop
    xp = np.pad(x, [(0, 0)] * (x.ndim - 1) + [(start, end)])
    n_frames = (xp.shape[-1] - win_length) // hop + 1
    idx = np.arange(win_length)[None, :] + hop * np.arange(n_frames)[:, None]
    return xp[..., idx]


def mdct4(x: np.ndarray, window: np.ndarray, n_fft: int, hop: int, center: bool = True):
    """MDCT4.forward: returns (spec float64 [..., F, N/2], frames float32 [..., F, win])."""
    win = np.asarray(window, dtype=np.float32)
    frames = (frame_signal(np.asarray(x, dtype=np.float32), len(win), hop, center) * win).astype(np.float32)
    z = frames.astype(np.float64)
    if n_fft > len(win):  # mdct.py:417-419
        z = np.pad(z, [(0, 0)] * (z.ndim - 1) + [(0, n_fft - len(win))])
    return z @ mdct_matrix(n_fft), frames


def tdac_fold(z: np.ndarray) -> np.ndarray:
    """[a, b, c, d] quarters -> u = [-c_r - d, a - b_r] (length N/2); MDCT = DCT-IV(u)."""
    q = z.shape[-1] // 4
    a, b, c, d = z[..., :q], z[..., q:2 * q], z[..., 2 * q:3 * q], z[..., 3 * q:]
    return np.concatenate([-c[..., ::-1] - d, a - b[..., ::-1]], axis=-1)


def tdac_unfold(v: np.ndarray) -> np.ndarray:
    """v = DCT-IV(X) (length M) -> y (length 2M) = [v2, -v2_r, -v1_r, -v1]."""
    h = v.shape[-1] // 2
    v1, v2 = v[..., :h], v[..., h:]
    return np.concatenate([v2, -v2[..., ::-1], -v1[..., ::-1], -v1], axis=-1)


def mdct4_folded(x, window, n_fft, hop, center=True):
    """Same transform through the TDAC fold + 256-point DCT-IV (the HIP kernel's form)."""
    win = np.asarray(window, dtype=np.float32)
    assert len(win) == n_fft
    frames = (frame_signal(np.asarray(x, dtype=np.float32), n_fft, hop, center) * win).astype(np.float32)
    return tdac_fold(frames.astype(np.float64)) @ dct4_matrix(n_fft // 2)


def imdct4(spec: np.ndarray, window: np.ndarray, n_fft: int, hop: int, center: bool = True,
           out_length=None):
    """IMDCT4.forward: spec [B, F, N/2] -> (audio float64 [B, 1, 1, T], frames [B, F, win])."""
    spec = np.asarray(spec, dtype=np.float64)
    if spec.ndim != 3:
        raise AssertionError("Only tensors shaped in BHW are supported")
    if spec.shape[-1] != n_fft // 2:
        raise AssertionError("The last dim of input tensor should match the n_fft")
    win = np.asarray(window, dtype=np.float32).astype(np.float64)
    y = spec @ mdct_matrix(n_fft).T                      # mdct.py:464-466
    y = y[..., :len(win)] * win                          # mdct.py:469-473
    b, f, wl = y.shape
    out_len = (f - 1) * hop + wl
    out = np.zeros((b, out_len), dtype=np.float64)
    for i in range(f):                                   # fold == overlap-add, mdct.py:480-482
        out[:, i * hop:i * hop + wl] += y[:, i]
    out *= 4.0 / n_fft
    if center:
        out = out[:, wl // 2:out_len - wl // 2]          # mdct.py:484-486
    if out_length is not None:
        out = out[:, :out_length]
    return out[:, None, None, :], y


# --------------------------------------------------------------------------
# codec
# --------------------------------------------------------------------------
def amplitude_to_db(x, multiplier, amin, db_multiplier):
    """torchaudio.functional.amplitude_to_DB (top_db=None): multiplier * log10(clamp(x, min=amin)) - multiplier * db_multiplier."""
    return multiplier * np.log10(np.clip(x, amin, None)) - multiplier * db_multiplier


def db_to_amplitude(x, ref, power):
    """torchaudio.functional.DB_to_amplitude: ref * (10 ** (0.1 x)) ** power."""
    return ref * np.power(np.power(10.0, 0.1 * x), power)


def normalize(spec, *, arcsinh_transform=True, raw_mdct=False, arcsinh_gain=1000.0,
              abs_norm=True, src_range=(-5.0, 5.0), norm_range=(-1.0, 1.0), explicit_encoding=False, alpha=0.6,
              min_value=1e-7):
    """Audio2MDCT.normalize, every branch (pix2pixHD_model.py:84-106).  spec float64 [B, 1, F, W].

    Returns (normalised float64 [B, C, F, W], max, min, mean, std) like pix2pixHD_model.py:125.
    """
    spec = np.asarray(spec, dtype=np.float64)
    if explicit_encoding:
        neg = 0.5 * (np.abs(spec) - spec)
        pos = spec + neg
        log_spec = np.concatenate((amplitude_to_db(alpha * pos + (1 - alpha) * neg, 20.0, min_value, 1.0),
                                   amplitude_to_db((1 - alpha) * pos + alpha * neg, 20.0, min_value, 1.0)), axis=1)
    elif arcsinh_transform:
        log_spec = np.arcsinh(arcsinh_gain * spec) / np.float64(np.float32(LN10))  # torch.log(tensor(10.)) is fp32
    elif raw_mdct:
        log_spec = spec
    else:
        log_spec = amplitude_to_db(np.abs(spec) + min_value, 20.0, min_value, 1.0)
    mean = np.float32(log_spec.mean())
    std = np.float32(np.sqrt(log_spec.var(ddof=1)))
    if not abs_norm:
        flat = log_spec.reshape(log_spec.shape[0], log_spec.shape[1], -1)
        a_max = flat.max(-1)[:, :, None, None].astype(np.float32)
        a_min = flat.min(-1)[:, :, None, None].astype(np.float32)
    else:
        a_min = np.array([src_range[0]], dtype=np.float32)[None, None, None, :]
        a_max = np.array([src_range[1]], dtype=np.float32)[None, None, None, :]
    out = (log_spec - a_min) / (a_max - a_min)
    out = out * (norm_range[1] - norm_range[0]) + norm_range[0]
    return out, a_max, a_min, mean, std


def denormalize(log_spec, a_min, a_max, *, arcsinh_transform=True, raw_mdct=False,
                arcsinh_gain=1000.0, norm_range=(-1.0, 1.0), explicit_encoding=False, min_value=1e-7):
    """Audio2MDCT.denormalize.  pix2pixHD_model.py:127-137 (note the branch order there: arcsinh, raw, else dB -- the
    explicit encoding reaches the dB branch only with arcsinh / raw off, as in normalize's precedence)."""
    x = (np.asarray(log_spec).astype(np.float64) - norm_range[0]) / (norm_range[1] - norm_range[0])
    x = x * (np.asarray(a_max, dtype=np.float64) - np.asarray(a_min, dtype=np.float64)) + np.asarray(a_min, dtype=np.float64)
    if arcsinh_transform and not explicit_encoding:
        return np.sinh(x * np.float64(np.float32(LN10))) / arcsinh_gain
    if raw_mdct and not explicit_encoding:
        return x
    return db_to_amplitude(x, 10.0, 0.5) - min_value


def to_spectro(audio, window, n_fft, hop, **codec):
    """Audio2MDCT.to_spectro without the (dead on this path) phase / mask extras.

    Returns (log_spectro float32 [B,1,F,W], norm_param dict).  pix2pixHD_model.py:32-81.
    """
    spec, frames = mdct4(audio, window, n_fft, hop, center=True)
    spec = spec[:, None]
    out, a_max, a_min, mean, std = normalize(spec, **codec)
    return out.astype(np.float32), {"max": a_max, "min": a_min, "mean": mean, "std": std, "frames": frames}


def to_audio(log_spec, norm_param, window, n_fft, hop, pha=None, **codec):
    """Audio2MDCT.to_audio.  pix2pixHD_model.py:139-163 (the dB branch multiplies the magnitudes by ``pha``; its random
    pseudo-phase for up_ratio > 1 is the caller's business)."""
    alpha = codec.get("alpha", 0.6)
    explicit = codec.get("explicit_encoding", False)
    db = not explicit and not codec.get("arcsinh_transform", True) and not codec.get("raw_mdct", False)
    keys = ("arcsinh_transform", "raw_mdct", "arcsinh_gain", "norm_range", "explicit_encoding", "min_value")
    spec = denormalize(log_spec, norm_param["min"], norm_param["max"], **{k: v for k, v in codec.items() if k in keys})
    if explicit:
        spec = ((spec[:, 0] - spec[:, 1]) / (2 * alpha - 1))[:, None]
    elif db and pha is not None:
        spec = spec * np.asarray(pha, dtype=np.float64)
    audio, _ = imdct4(spec[:, 0], window, n_fft, hop, center=True)
    return audio


def stitch_segments(audio, segment_length: int, gen_overlap: int = 0):
    """generate_audio.py:40-53: concat, or halve the overlapping edges and overlap-add.

    audio: [n_seg, 1, 1, T] -> [1, total].
    """
    a = np.array(audio, dtype=np.float64, copy=True)
    n_seg = a.shape[0]
    if gen_overlap > 0:
        stride = segment_length - gen_overlap
        out_len = (n_seg - 1) * stride + segment_length
        a[..., :gen_overlap] *= 0.5
        a[..., -gen_overlap:] *= 0.5
        a = a.reshape(n_seg, segment_length)
        out = np.zeros(out_len, dtype=np.float64)
        for i in range(n_seg):
            out[i * stride:i * stride + segment_length] += a[i]
        return out[None, gen_overlap:out_len - gen_overlap]
    return a.reshape(1, -1)
