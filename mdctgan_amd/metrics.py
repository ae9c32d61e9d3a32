"""``compute_matrics`` (util/util.py:132-177) on the device: MSE, SNR of the super-resolved and of the low-rate waveform,
log-spectral distance.  Same call and return tuple as the reference -- ``(mse, snr_sr, snr_lr, 0, 0, 0, lsd)``, Python
floats -- but the waveforms stay in HBM: three row reductions in double precision, and for the LSD the power
spectrograms of ``aF.spectrogram(n_fft=2*opt.n_fft, hop=2*opt.hop_length, window=kbdwin(2*opt.win_length), center,
power=2)`` as windowed reflect-padded frames times a dense DFT table on the f32 MFMA pipe.
"""
from __future__ import annotations

import math

import torch

from . import _lib, ops
from .mdct import kbdwin

_tables = {}


def _dft_table(n_fft: int, device):
    """[2 * (n_fft/2 + 1), n_fft]: rows 2k = cos(2 pi k n / N), 2k+1 = -sin(2 pi k n / N) (float64 on the host, once)."""
    key = (n_fft, str(device))
    t = _tables.get(key)
    if t is None:
        k = torch.arange(n_fft // 2 + 1, dtype=torch.float64)[:, None]
        n = torch.arange(n_fft, dtype=torch.float64)[None, :]
        ang = 2.0 * math.pi * k * n / n_fft
        t = torch.stack((torch.cos(ang), -torch.sin(ang)), dim=1).reshape(-1, n_fft).to(torch.float32).to(device).contiguous()
        _tables[key] = t
    return t


def power_spectra(audio: torch.Tensor, n_fft: int, hop_length: int, window: torch.Tensor, center: bool = True):
    """[B, T] -> interleaved (re, im) STFT coefficients [B * F, 2 * (n_fft/2 + 1)] and F."""
    lib = _lib.load()
    x = _lib.f32c(audio)
    B, T = x.shape
    F = lib.mg_stft_num_frames(T, n_fft, hop_length, int(center))
    if F <= 0:
        raise ValueError("signal of %d samples is too short for n_fft=%d" % (T, n_fft))
    frames = torch.empty(B * F, n_fft, dtype=torch.float32, device=x.device)
    w = _lib.f32c(window.to(x.device))
    _lib.check(lib.mg_stft_frames(_lib.ptr(x), B, T, _lib.ptr(w), n_fft, hop_length, int(center), _lib.ptr(frames),
                                  _lib.stream()), "mg_stft_frames")
    table = _dft_table(n_fft, x.device)
    g = ops.conv_geom(1, 1, B * F, n_fft, table.shape[0], 1, 1, 1, 0, False)
    spec = ops.conv_fwd(g, frames.view(1, 1, B * F, n_fft), table.view(table.shape[0], 1, 1, n_fft))
    return spec.view(B * F, table.shape[0]), F


def compute_matrics(hr_audio, lr_audio, sr_audio, opt):
    lib = _lib.load()
    dev = sr_audio.device
    hr = _lib.f32c(hr_audio.to(dev).reshape(-1, hr_audio.shape[-1]))
    lr = _lib.f32c(lr_audio.to(dev).reshape(-1, lr_audio.shape[-1]))
    sr = _lib.f32c(sr_audio.reshape(-1, sr_audio.shape[-1]))
    B, T = sr.shape
    sums = torch.empty(B, 3, dtype=torch.float64, device=dev)
    _lib.check(lib.mg_metrics_rows(_lib.ptr(hr), _lib.ptr(lr), _lib.ptr(sr), B, T, _lib.ptr(sums), _lib.stream()),
               "mg_metrics_rows")
    mse = (sums[:, 1].sum() / (B * T)).item()
    snr_sr = (10 * torch.log10(sums[:, 0] / sums[:, 1])).mean().item()
    snr_lr = (10 * torch.log10(sums[:, 0] / sums[:, 2])).mean().item()
    n_fft, hop, win = 2 * opt.n_fft, 2 * opt.hop_length, 2 * opt.win_length
    if win != n_fft:
        raise NotImplementedError("win_length != n_fft")
    window = kbdwin(win)
    sa, _ = power_spectra(hr, n_fft, hop, window, bool(opt.center))
    sb, _ = power_spectra(sr, n_fft, hop, window, bool(opt.center))
    per_frame = torch.empty(sa.shape[0], dtype=torch.float32, device=dev)
    _lib.check(lib.mg_lsd_frames(_lib.ptr(sa), _lib.ptr(sb), sa.shape[0], n_fft // 2 + 1, _lib.ptr(per_frame),
                                 _lib.stream()), "mg_lsd_frames")
    lsd = per_frame.double().mean().item()
    return mse, snr_sr, snr_lr, 0, 0, 0, lsd


def eval_model(model, eval_batches, opt, eval_path=None):
    """train.py:104-134: run ``model.inference`` over the evaluation batches (dicts with ``LR_audio`` / ``HR_audio`` like the
    reference's dataloader items, or ``(lr, hr)`` pairs), score every batch with ``compute_matrics`` and average -- the
    same five columns, appended to ``eval_path`` as a CSV row when given.  Everything up to the per-batch Python floats
    stays on the device; the model is put in eval mode for the loop (BatchNorm running statistics of the attention
    blocks) and back in train mode afterwards, like the reference."""
    import csv
    import numpy as np
    err, snr, snr_seg, pesq, lsd = [], [], [], [], []
    was_training = model.training
    dev = getattr(model, "device", "cuda")
    try:
        for j, item in enumerate(eval_batches):
            model.eval()
            lr_audio, hr_audio = (item["LR_audio"], item["HR_audio"]) if isinstance(item, dict) else item
            lr_audio, hr_audio = lr_audio.to(dev), hr_audio.to(dev)
            with torch.no_grad():
                _, sr_audio, _, _, _ = model.inference(lr_audio)
                _mse, _snr_sr, _snr_lr, _ssnr_sr, _ssnr_lr, _pesq, _lsd = compute_matrics(
                    hr_audio.squeeze(), lr_audio.squeeze(), sr_audio.squeeze(), opt)
            err.append(_mse)
            snr.append((_snr_lr, _snr_sr))
            snr_seg.append((_ssnr_lr, _ssnr_sr))
            pesq.append(_pesq)
            lsd.append(_lsd)
            if j >= opt.eval_size:
                break
    finally:
        model.train(was_training)
    result = {"err": float(np.mean(err)), "snr": float(np.mean(snr)), "snr_seg": float(np.mean(snr_seg)),
              "pesq": float(np.mean(pesq)), "lsd": float(np.mean(lsd))}
    if eval_path:
        with open(eval_path, "a") as f:
            writer = csv.DictWriter(f, fieldnames=result.keys())
            if f.tell() == 0:
                writer.writeheader()
            writer.writerow(result)
    return result
