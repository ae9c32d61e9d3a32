"""Measured decision for SURVEY 8(f) F4 / VERDICT r1 item 9: the N/4-point complex-FFT form of the MDCT
(models/mdct.py:596-628, FastMDCT4; README.md:99-110) against K1's dense 256-point DCT-IV contraction on the f32 MFMA pipe,
at B = 8, 64 and 4096 clips of 32512 samples.  The FFT leg is built from library parts the way the reference builds it:
torch elementwise ops (fold, twiddles) + torch.fft (rocFFT), complex64.  Prints per-batch times and the max deviation of
the FFT leg from K1 (both against float64 are ~1e-6 of max|X|).  Not part of the product path.
    python scripts/fft_vs_dense_mdct.py
"""
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from mdctgan_amd import _lib  # noqa: E402
from mdctgan_amd.mdct import dct4_table, kbdwin, mdct4_codec  # noqa: E402

N, M, T = 512, 256, 32512


def fft_mdct(x, window, tw, tw2):
    """[B, T] -> [B, F, 256]: frame, window, TDAC fold, DCT-IV through one 128-point complex FFT per frame."""
    xp = torch.nn.functional.pad(x, (M, M))
    z = xp.unfold(-1, N, M) * window                              # [B, F, 512]
    q = M // 2
    a, b, c, d = z[..., :q], z[..., q:2 * q], z[..., 2 * q:3 * q], z[..., 3 * q:]
    u = torch.cat((-c.flip(-1) - d, a - b.flip(-1)), dim=-1)      # [B, F, 256]
    v = torch.complex(u[..., 0::2], u[..., 1::2].flip(-1)) * tw   # (u[2n] + i u[M-1-2n]) e^{-i pi (4n+1)/(4M)}
    y = torch.fft.fft(v, dim=-1) * tw2                            # e^{-i pi k / M}
    out = torch.empty_like(u)
    out[..., 0::2] = y.real
    out[..., 1::2] = (-y.imag).flip(-1)
    return out


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    dev = "cuda"
    w = kbdwin(N).to(dev)
    d4 = dct4_table(M, dev)
    n = torch.arange(M // 2, dtype=torch.float64)
    tw = torch.exp(-1j * math.pi * (4 * n + 1) / (4 * M)).to(torch.complex64).to(dev)
    tw2 = torch.exp(-1j * math.pi * n / M).to(torch.complex64).to(dev)
    for B in (8, 64, 4096):
        x = 0.05 * torch.randn(B, T, device=dev)
        k1 = lambda: mdct4_codec(x, w, d4, N, codec=_lib.MG_CODEC_RAW)["spec"]     # noqa: E731
        ff = lambda: fft_mdct(x, w, tw, tw2)                                            # noqa: E731
        a, b = k1(), ff()
        dev_max = (a - b).abs().max().item() / a.abs().max().item()
        t1, t2 = timeit(k1), timeit(ff)
        gb = B * 261120 / 1e9
        print("B=%5d  K1 dense-MFMA %8.1f us (%6.1f GB/s algorithmic)   FFT leg (torch ops + rocFFT) %8.1f us (%6.1f GB/s)   "
              "max |K1 - FFT| / max|X| = %.1e" % (B, t1, gb / t1 * 1e6, t2, gb / t2 * 1e6, dev_max), flush=True)


if __name__ == "__main__":
    main()
