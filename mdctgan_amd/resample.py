"""``torchaudio.functional.resample`` for the data path of the reference (data/audio_dataset.py:66-71, 171-177) on the
device: same signature and defaults (sinc_interp_hann, lowpass_filter_width=6, rolloff=0.99), the polyphase filter bank
built once per rate pair exactly as torchaudio's ``_get_sinc_resample_kernel`` does (float64, stored as float32), the
convolution as one HIP launch (``mg_resample``).  torchaudio itself is not a dependency.
"""
from __future__ import annotations

import math

import torch

from . import _lib

_kernels = {}


def _sinc_kernel(orig_freq: int, new_freq: int, lowpass_filter_width: int, rolloff: float, device):
    key = (int(orig_freq), int(new_freq), int(lowpass_filter_width), float(rolloff), str(device))
    hit = _kernels.get(key)
    if hit is not None:
        return hit
    g = math.gcd(int(orig_freq), int(new_freq))
    orig, new = int(orig_freq) // g, int(new_freq) // g
    base_freq = min(orig, new) * rolloff
    width = math.ceil(lowpass_filter_width * orig / base_freq)
    idx = torch.arange(-width, width + orig, dtype=torch.float64)[None, None] / orig
    t = torch.arange(0, -new, -1)[:, None, None] / new + idx          # int64 / int -> float32, then float64
    t = t * base_freq
    t = t.clamp_(-lowpass_filter_width, lowpass_filter_width)
    window = torch.cos(t * math.pi / lowpass_filter_width / 2) ** 2
    t = t * math.pi
    scale = base_freq / orig
    kern = torch.where(t == 0, torch.tensor(1.0, dtype=t.dtype), t.sin() / t)
    kern = (kern * window * scale).to(torch.float32).reshape(new, 2 * width + orig).contiguous().to(device)
    _kernels[key] = (kern, width, orig, new)
    return _kernels[key]


def resample(waveform: torch.Tensor, orig_freq: int, new_freq: int, lowpass_filter_width: int = 6,
             rolloff: float = 0.99, resampling_method: str = "sinc_interp_hann") -> torch.Tensor:
    """waveform [..., L] (device) -> [..., ceil(new_freq * L / orig_freq)]."""
    if resampling_method != "sinc_interp_hann":
        raise NotImplementedError("resampling_method %r (the reference uses the default sinc_interp_hann)" % resampling_method)
    if orig_freq <= 0 or new_freq <= 0:
        raise ValueError("Original frequency and desired frequency should be positive")
    if int(orig_freq) == int(new_freq):
        return waveform
    lib = _lib.load()
    kern, width, orig, new = _sinc_kernel(orig_freq, new_freq, lowpass_filter_width, rolloff, waveform.device)
    shape = waveform.shape
    x = _lib.f32c(waveform.reshape(-1, shape[-1]))
    n_out = lib.mg_resample_length(shape[-1], orig, new)
    out = torch.empty(x.shape[0], n_out, dtype=torch.float32, device=x.device)
    _lib.check(lib.mg_resample(_lib.ptr(x), x.shape[0], x.shape[1], _lib.ptr(kern), orig, new, width, _lib.ptr(out), n_out,
                               _lib.stream()), "mg_resample")
    return out.reshape(shape[:-1] + (n_out,))


def seg_pad_audio(waveform: torch.Tensor, segment_length: int) -> torch.Tensor:
    """AudioDataset.seg_pad_audio (data/audio_dataset.py:102-110) for a batch [B, L]: crop or zero pad to segment_length."""
    L = waveform.shape[-1]
    if L >= segment_length:
        return waveform[..., :segment_length]
    return torch.nn.functional.pad(waveform, (0, segment_length - L))


def make_training_pair(waveform: torch.Tensor, orig_sample_rate: int, hr_sampling_rate: int, lr_sampling_rate: int,
                       segment_length: int):
    """AudioDataset.__getitem__ (data/audio_dataset.py:66-82) without the optional noise: HR = resample to hr_rate; LR =
    resample to lr_rate and back up to hr_rate; both cropped / padded to segment_length.  [B, L] -> (lr, hr)."""
    hr = resample(waveform, orig_sample_rate, hr_sampling_rate)
    lr = resample(resample(waveform, orig_sample_rate, lr_sampling_rate), lr_sampling_rate, hr_sampling_rate)
    return seg_pad_audio(lr, segment_length), seg_pad_audio(hr, segment_length)


def make_test_segments(raw_audio: torch.Tensor, in_sampling_rate: int, hr_sampling_rate: int, lr_sampling_rate: int,
                       segment_length: int, gen_overlap: int = 0, is_lr_input: bool = False):
    """AudioTestDataset (data/audio_dataset.py:141-186; add_noise off) for one waveform [1, L] in HBM: the DC shift of
    read_audio (``raw += 1e-4 - mean(raw)``), then the low-rate input of the model -- a file that already IS low-rate
    (``--is_lr_input``) is only brought up to hr_rate, anything else goes down to lr_rate and back up -- cut into segments
    by seg_pad_audio (generate_audio.segment_audio).  -> (lr_audio [1, L'], segments [n_seg, segment_length])."""
    from .generate_audio import segment_audio
    raw = raw_audio.to(torch.float32)
    raw = raw + (1e-4 - raw.mean())
    if is_lr_input:
        lr_audio = resample(raw, in_sampling_rate, hr_sampling_rate)
    else:
        lr_audio = resample(resample(raw, in_sampling_rate, lr_sampling_rate), lr_sampling_rate, hr_sampling_rate)
    return lr_audio, segment_audio(lr_audio, segment_length, gen_overlap)
