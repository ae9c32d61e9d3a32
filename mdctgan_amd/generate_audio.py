"""Whole-utterance inference: the device-side part of generate_audio.py:22-53.

The reference script walks the dataset loader, calls ``model.inference`` per batch, copies every ``sr_audio`` to the
host and concatenates (or cross-fades with ``F.fold`` when ``--gen_overlap > 0``).  Here the segments stay in HBM:
batches of low-rate segments go through ``model.inference`` and one gather kernel (``mg_stitch_segments``) writes the
waveform.  Dataset loading, resampling and the metrics of ``util.compute_matrics`` are outside the hot path.
"""
from __future__ import annotations

import torch

from . import ops


def segment_audio(lr_audio: torch.Tensor, segment_length: int, gen_overlap: int = 0) -> torch.Tensor:
    """[T] or [1, T] waveform -> [n_seg, segment_length] at stride ``segment_length - gen_overlap``, zero padded at the
    end (what the reference's dataset does with ``--gen_overlap`` when it unfolds the utterance)."""
    x = lr_audio.reshape(-1)
    stride = segment_length - gen_overlap
    if stride <= 0:
        raise ValueError("gen_overlap must be smaller than segment_length")
    n_seg = max(1, -(-(x.numel() - gen_overlap) // stride))
    total = (n_seg - 1) * stride + segment_length
    if total > x.numel():
        x = torch.cat([x, x.new_zeros(total - x.numel())])
    return x.unfold(0, segment_length, stride).contiguous()


def generate(model, lr_segments: torch.Tensor, batch_size: int = 64, gen_overlap: int = 0) -> torch.Tensor:
    """lr_segments [n_seg, T] (device) -> stitched super-resolved waveform [1, total] (generate_audio.py:28-53)."""
    if lr_segments.dim() != 2:
        raise ValueError("lr_segments must be [n_seg, T]")
    outs = []
    with torch.no_grad():
        for i in range(0, lr_segments.shape[0], batch_size):
            _, sr_audio, _, _, _ = model.inference(lr_segments[i:i + batch_size])
            outs.append(sr_audio)
    audio = torch.cat(outs, dim=0)                      # [n_seg, 1, 1, T]
    return ops.stitch_segments(audio, audio.shape[-1], gen_overlap)
