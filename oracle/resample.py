"""Oracle (test infrastructure): torchaudio.functional.resample restated in numpy float64.

The reference's data path (data/audio_dataset.py:66-71, 171-177) calls ``aF.resample(waveform, orig_freq, new_freq)``
with torchaudio's defaults: resampling_method="sinc_interp_hann", lowpass_filter_width=6, rolloff=0.99.  torchaudio is a
third-party dependency that is NOT installed in this image (requirements.txt lists it unpinned), so this file restates
its published algorithm (torchaudio/functional/functional.py, ``_get_sinc_resample_kernel`` +
``_apply_sinc_resample_kernel``, BSD-2): **parity unpinned** -- there is no torchaudio here to generate vectors from and
the reference holds no fixture for this path; the tests pin the restatement only through properties (unit DC gain,
pass-band sines, identity at equal rates) and the HIP kernel against this restatement.
"""
from __future__ import annotations

import math

import numpy as np


def sinc_resample_kernel(orig_freq: int, new_freq: int, lowpass_filter_width: int = 6, rolloff: float = 0.99):
    """(kernels [new, 2*width + orig] float32, width, orig, new) for the gcd-reduced rate pair."""
    g = math.gcd(int(orig_freq), int(new_freq))
    orig, new = int(orig_freq) // g, int(new_freq) // g
    base_freq = min(orig, new) * rolloff
    width = math.ceil(lowpass_filter_width * orig / base_freq)
    idx = np.arange(-width, width + orig, dtype=np.float64)[None, :] / orig
    # torchaudio divides an int64 arange by new_freq: the quotient is float32 before it meets the float64 idx
    phase = (np.arange(0, -new, -1, dtype=np.int64) / np.float32(new)).astype(np.float32).astype(np.float64)[:, None]
    t = (phase + idx) * base_freq
    t = np.clip(t, -lowpass_filter_width, lowpass_filter_width)
    window = np.cos(t * math.pi / lowpass_filter_width / 2) ** 2
    t = t * math.pi
    scale = base_freq / orig
    with np.errstate(invalid="ignore", divide="ignore"):
        kern = np.where(t == 0, 1.0, np.sin(t) / np.where(t == 0, 1.0, t))
    kern = kern * window * scale
    return kern.astype(np.float32), width, orig, new


def resample(waveform, orig_freq: int, new_freq: int, lowpass_filter_width: int = 6, rolloff: float = 0.99):
    """waveform [..., L] -> [..., ceil(new * L / orig)] (float64 accumulation over the float32 kernel)."""
    x = np.asarray(waveform, dtype=np.float64)
    if int(orig_freq) == int(new_freq):
        return x.copy()
    kern, width, orig, new = sinc_resample_kernel(orig_freq, new_freq, lowpass_filter_width, rolloff)
    shape = x.shape
    x2 = x.reshape(-1, shape[-1])
    L = x2.shape[1]
    xp = np.pad(x2, ((0, 0), (width, width + orig)))
    K = kern.shape[1]
    n_frames = (xp.shape[1] - K) // orig + 1
    frames = np.lib.stride_tricks.sliding_window_view(xp, K, axis=1)[:, ::orig][:, :n_frames]     # [W, n_frames, K]
    out = np.einsum("wnk,pk->wnp", frames, kern.astype(np.float64)).reshape(x2.shape[0], -1)
    target = int(math.ceil(new * L / orig))
    return out[:, :target].reshape(shape[:-1] + (target,))


def lr_from_hr(hr, hr_rate: int, lr_rate: int):
    """data/audio_dataset.py:68-71: down to lr_rate and back up to hr_rate (the low-rate input the model sees)."""
    return resample(resample(hr, hr_rate, lr_rate), lr_rate, hr_rate)
