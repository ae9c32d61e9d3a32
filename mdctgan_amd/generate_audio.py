"""Whole-utterance inference: the device-side part of generate_audio.py:22-53.

The reference script walks the dataset loader, calls ``model.inference`` per batch, copies every ``sr_audio`` to the
host and concatenates (or cross-fades with ``F.fold`` when ``--gen_overlap > 0``).  Here the waveform is built in HBM by
the decoder itself: batches of low-rate segments go through ``model.inference`` and K2's overlap-add store writes each
segment at its place in the stitched waveform, cross-fade included (``mg_imdct4_stitched``); codec geometries K2 does not
cover decode per batch and one gather kernel (``mg_stitch_segments``) stitches.  Dataset loading, resampling and the
metrics of ``util.compute_matrics`` are outside the hot path.
"""
from __future__ import annotations

import os

import torch

from . import _lib
from . import functional as Fh
from . import ops


def segment_audio(audio: torch.Tensor, segment_length: int, gen_overlap: int = 0) -> torch.Tensor:
    """AudioTestDataset.seg_pad_audio (data/audio_dataset.py:153-167): [T] or [1, T] waveform -> [n_seg, segment_length].
    A waveform of at least one segment is zero padded by ``gen_overlap`` in front and up to ``ceil(T / L) * L +
    gen_overlap`` behind, then unfolded at stride ``L - gen_overlap``; a shorter one is padded to a single segment."""
    x = audio.reshape(-1)
    length = x.numel()
    if gen_overlap < 0 or gen_overlap >= segment_length:
        raise ValueError("gen_overlap must be in [0, segment_length)")
    if length >= segment_length:
        num_segments = -(-length // segment_length)
        x = torch.nn.functional.pad(x, (gen_overlap, segment_length * num_segments - length + gen_overlap))
        return x.unfold(0, segment_length, segment_length - gen_overlap).contiguous()
    return torch.nn.functional.pad(x, (0, segment_length - length)).unsqueeze(0)


def generate(model, lr_segments: torch.Tensor, batch_size: int = 64, gen_overlap: int = 0) -> torch.Tensor:
    """lr_segments [n_seg, T] (device) -> stitched super-resolved waveform [1, total] (generate_audio.py:28-53)."""
    if lr_segments.dim() != 2:
        raise ValueError("lr_segments must be [n_seg, T]")
    outs = []
    # The fused codec geometry: K2's overlap-add store writes every batch's segments straight into the stitched waveform
    # (mg_imdct4_stitched: the halving, F.fold and the crop of generate_audio.py:43-50 -- or the torch.cat of :52 -- happen in
    # the store; no list of segments, no concatenation, no stitching launch).  Other geometries decode per batch and stitch after.
    pre = model.preprocess
    fused = bool(getattr(pre, "fused", False)) and os.environ.get("MG_NO_STITCHED_K2") != "1"
    out = None
    if fused:
        n_seg, T = lr_segments.shape
        seg_len = (_lib.load().mg_mdct4_num_frames(T, pre.n_fft) - 1) * (pre.n_fft // 2)
        total = _lib.load().mg_stitch_length(n_seg, seg_len, gen_overlap)
        if total <= 0:
            raise ValueError("invalid stitching geometry: n_seg=%d segment_length=%d gen_overlap=%d" % (n_seg, seg_len, gen_overlap))
        out = torch.empty(total, dtype=torch.float32, device=lr_segments.device)
    # generate_audio.py:27 calls model.eval() first: the BatchNorm2d layers of the bottleneck-attention blocks must use
    # their running statistics (and must not update them) during inference
    was_training = model.training
    model.eval()
    try:
        with torch.no_grad():
            for i in range(0, lr_segments.shape[0], batch_size):
                if fused:
                    model.inference(lr_segments[i:i + batch_size], stitch=(out, gen_overlap, i, seg_len))
                else:
                    _, sr_audio, _, _, _ = model.inference(lr_segments[i:i + batch_size])
                    outs.append(sr_audio)
    finally:
        model.train(was_training)
    if fused:
        return out.view(1, -1)
    audio = torch.cat(outs, dim=0)                      # [n_seg, 1, 1, T]
    return ops.stitch_segments(audio, audio.shape[-1], gen_overlap)


def make_graphed_generate(model, lr_segments: torch.Tensor, batch_size: int = 64, gen_overlap: int = 0, warmup: int = 2):
    """Capture generate() for a fixed segment count into one hipGraph (K1, ~120 generator launches and the stitching K2
    per batch) and return run(lr_segments) -> stitched waveform (a buffer that the next replay overwrites).  The
    generator weights must not change between capture and replay without re-capturing: the Winograd layers read the
    transformed-weight images cached by the warm-up calls."""
    static_in = lr_segments.clone()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(max(warmup, 1)):
            generate(model, static_in, batch_size, gen_overlap)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        out = generate(model, static_in, batch_size, gen_overlap)

    # The captured launches hold raw pointers to the transformed-weight images the warm-up cached on the parameters
    # (weight._mg_u_cache).  Pin those tensors for the graph's lifetime -- an eager inference after an optimiser step
    # would replace the cache entries and free them -- and refuse to replay once the weights have moved.
    params = list(model.netG.parameters())
    pinned = [getattr(p, "_mg_u_cache", None) for p in params]
    stamp = (Fh.WEIGHT_EPOCH[0], tuple(p._version for p in params), tuple(p.data_ptr() for p in params))

    def run(lr=None):
        now = (Fh.WEIGHT_EPOCH[0], tuple(p._version for p in params), tuple(p.data_ptr() for p in params))
        if now != stamp:
            raise RuntimeError("the generator's weights changed after make_graphed_generate() captured them "
                               "(optimizer step / load_state_dict): capture again")
        if lr is not None:
            static_in.copy_(lr, non_blocking=True)
        graph.replay()
        return out
    run.graph, run.pinned = graph, pinned
    return run
