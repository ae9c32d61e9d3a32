"""Oracle (test infrastructure): torchaudio.functional.resample restated in numpy float64.

The reference's data path (data/audio_dataset.py:66-71, 171-177) calls ``aF.resample(waveform, orig_freq, new_freq)``
with torchaudio's defaults: resampling_method="sinc_interp_hann", lowpass_filter_width=6, rolloff=0.99.  torchaudio is a
third-party dependency that is NOT installed in this image (requirements.txt lists it unpinned), so this file restates
its published algorithm (torchaudio/functional/functional.py, ``_get_sinc_resample_kernel`` +
``_apply_sinc_resample_kernel``, BSD-2): **parity unpinned** -- there is no torchaudio here to generate vectors from and
the reference holds no fixture for this path; the tests pin the restatement only through properties (unit DC gain,
pass-band sines, identity at equal rates) and the HIP kernel against this restatement.

The dataset chain AROUND the resampler is pinned (round 4): tests/golden/g13_dataset_chain.npz was captured from the
reference's own AudioDataset / AudioAppDataset classes with aF.resample stood in for by ``resample`` below (cast to float32,
the dtype torchaudio returns for float32 input); ``crop_window`` / ``training_item`` / ``inference_segments`` restate that chain
and tests/test_oracle_golden.py holds them to the fixture bit for bit.
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


def _resample32(x, orig, new):
    """aF.resample on a float32 tensor returns float32: the chain below rounds between calls as the reference's tensors do."""
    return resample(np.asarray(x, dtype=np.float32), orig, new).astype(np.float32)


def crop_window(audio_length: int, fs: int, segment_length: int, hr_rate: int) -> int:
    """data/audio_dataset.py:44: the exclusive upper bound of the random start frame.  The segment is scaled to the FILE's
    rate here, but the load that follows takes ``segment_length`` frames at the file's rate regardless (:48-49): a 16 kHz file
    yields 3x the duration, which the crop after resampling cuts back.  <= 0: the whole (short) file is loaded."""
    return int(audio_length - segment_length * fs / hr_rate)


def seg_pad_train(waveform, segment_length: int):
    """AudioDataset.seg_pad_audio (:102-110) followed by __getitem__'s squeeze(0): [1, L] -> [segment_length]."""
    w = np.asarray(waveform)
    if w.shape[1] >= segment_length:
        return w[0][:segment_length]
    return np.pad(w, ((0, 0), (0, segment_length - w.shape[1])))[0]


def training_item(waveform, fs: int, hr_rate: int, lr_rate: int, segment_length: int):
    """AudioDataset.__getitem__ (:66-82, add_noise off): loaded [1, L] float32 at rate fs -> (HR, LR) [segment_length]."""
    hr = _resample32(waveform, fs, hr_rate)
    lr = _resample32(_resample32(waveform, fs, lr_rate), lr_rate, hr_rate)
    return seg_pad_train(hr, segment_length), seg_pad_train(lr, segment_length)


def seg_pad_test(audio, segment_length: int, overlap: int):
    """AudioTestDataset.seg_pad_audio (:153-167): [1, L] or [L] -> [n_seg, segment_length]."""
    a = np.asarray(audio).reshape(-1)
    length = len(a)
    if length >= segment_length:
        n = int(math.ceil(length / segment_length))
        a = np.pad(a, (overlap, segment_length * n - length + overlap))
        step = segment_length - overlap
        count = (len(a) - segment_length) // step + 1
        return np.stack([a[i * step:i * step + segment_length] for i in range(count)])
    return np.pad(a, (0, segment_length - length))[None]


def inference_segments(raw_audio, fs: int, hr_rate: int, lr_rate: int, segment_length: int, overlap: int = 0, is_lr_input: bool = False):
    """AudioTestDataset.read_audio's DC shift (:147: raw += 1e-4 - mean(raw), float32) + post_processing (:169-186, add_noise
    off) + seg_pad_audio: raw [1, L] -> (lr_audio [1, L'], segments [n_seg, segment_length])."""
    raw = np.asarray(raw_audio, dtype=np.float32)
    import torch
    t = torch.from_numpy(raw.copy())
    t += 1e-4 - torch.mean(t)                      # torch's float32 mean (pairwise), as the reference computes it
    raw = t.numpy()
    if is_lr_input:
        lr_audio = _resample32(raw, fs, hr_rate)
    else:
        lr_audio = _resample32(_resample32(raw, fs, lr_rate), lr_rate, hr_rate)
    return lr_audio, seg_pad_test(lr_audio, segment_length, overlap)


