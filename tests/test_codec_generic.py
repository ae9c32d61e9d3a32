"""MDCT4 / IMDCT4 at geometries other than the hot path's n_fft = 512 (models/mdct.py:365-489, class default 2048) and the
codec branches of Audio2MDCT other than arcsinh (models/pix2pixHD_model.py:83-163: dB, --explicit_encoding, raw with
per-sample range), on the generic kernels of csrc/codec_generic.hip, against the oracle (oracle/transform.py; the transform
part is pinned to the reference by the golden vectors at n_fft = 512 and is the same code at any n_fft; the two torchaudio
functions of the dB codecs are restated from their published formulas -- parity of those UNPINNED).
Tolerances: raw coefficients 3e-6 of max|X| (float32 contraction vs float64); waveforms 3e-6 of max|y|; normalised dB
spectrogram 3e-5 absolute in [-1, 1] (log10 near the clamp amplifies rounding); round trips 1e-5 of max|x|."""
import types

import numpy as np
import pytest
import torch

from oracle import transform as T

GEOMS = [(2048, 1024, 2048), (1024, 256, 512), (512, 128, 512), (256, 128, 256)]


@pytest.mark.parametrize("n_fft,hop,win", GEOMS)
def test_oracle_generic_geometry_is_consistent(n_fft, hop, win):
    """Oracle sanity on the CPU: for win == n_fft == 2 hop with the KBD window (Princen-Bradley) IMDCT(MDCT(x)) == x; in
    every geometry the contraction form equals the reference's twiddle-FFT-twiddle evaluation (mdct.py:387-390, 421-423)."""
    rng = np.random.default_rng(n_fft)
    x = rng.standard_normal((2, 8 * n_fft)).astype(np.float32)
    w = T.kbd_window(win)
    X, frames = T.mdct4(x, w, n_fft, hop)
    z = np.pad(frames.astype(np.float64), [(0, 0), (0, 0), (0, n_fft - win)])
    n = np.arange(n_fft)
    ref = np.real(np.exp(-1j * (np.pi / (2 * n_fft) + np.pi / 4) * np.arange(1, n_fft, 2))
                  * np.fft.fft(z * np.exp(-1j * np.pi / n_fft * n))[..., :n_fft // 2])
    assert np.abs(X - ref).max() <= 1e-10 * np.abs(ref).max()
    if win == n_fft and 2 * hop == n_fft:
        y, _ = T.imdct4(X, w, n_fft, hop)
        assert np.abs(y[:, 0, 0] - x).max() <= 1e-6


def test_oracle_db_codecs_round_trip():
    rng = np.random.default_rng(3)
    x = (0.05 * rng.standard_normal((2, 7936))).astype(np.float32)
    w = T.kbd_window(512)
    for codec in (dict(arcsinh_transform=False, raw_mdct=False, explicit_encoding=True, alpha=0.6, src_range=(-160.0, 40.0), abs_norm=True),
                  dict(arcsinh_transform=False, raw_mdct=False, src_range=(-160.0, 40.0), abs_norm=False)):
        s, norm = T.to_spectro(x, w, 512, 256, norm_range=(-1.0, 1.0), **codec)
        assert s.shape[1] == (2 if codec.get("explicit_encoding") else 1) and np.abs(s).max() <= 1.0 + 1e-6
        X, _ = T.mdct4(x, w, 512, 256)
        back = T.to_audio(s.astype(np.float64), norm, w, 512, 256, pha=np.sign(X)[:, None], norm_range=(-1.0, 1.0), **codec)
        assert np.abs(back[:, 0, 0] - x).max() <= 2e-5


@pytest.mark.gpu
@pytest.mark.parametrize("n_fft,hop,win", GEOMS)
def test_generic_mdct_imdct_against_oracle(n_fft, hop, win):
    from mdctgan_amd.mdct import IMDCT4, MDCT4, kbdwin
    rng = np.random.default_rng(n_fft + win)
    x = rng.standard_normal((3, 6 * n_fft + 3 * hop)).astype(np.float32)
    w = kbdwin(win)
    m = MDCT4(n_fft, hop, win, w, device="cuda")
    assert m.fused == (n_fft == 512 and win == 512 and hop == 256)
    X, frames = m(torch.from_numpy(x).cuda(), True)
    want, wf = T.mdct4(x, w.numpy(), n_fft, hop)
    assert X.shape == want.shape
    np.testing.assert_array_equal(frames.cpu().numpy(), wf)                      # float32 window multiply: bit-exact
    assert np.abs(X.cpu().numpy() - want).max() <= 3e-6 * np.abs(want).max()
    im = IMDCT4(n_fft, hop, win, w, device="cuda")
    y, _ = im(X)
    wy, _ = T.imdct4(X.cpu().numpy(), w.numpy(), n_fft, hop)
    assert y.shape == wy.shape
    assert np.abs(y.cpu().numpy() - wy).max() <= 3e-6 * np.abs(wy).max() + 1e-7
    y64, _ = IMDCT4(n_fft, hop, win, w, device="cuda", dtype=torch.float64, out_length=1000)(X)
    assert y64.dtype == torch.float64 and y64.shape[-1] == 1000
    assert np.abs(y64.cpu().numpy() - wy[..., :1000]).max() <= 3e-6 * np.abs(wy).max() + 1e-7


def _opt(**kw):
    from mdctgan_amd import options
    flags = ["--norm_range", "-1", "1", "--src_range", str(kw.pop("src0", -5)), str(kw.pop("src1", 5)), "--gpu_ids", "0",
             "--lr_sampling_rate", "12000"]
    for k, v in kw.items():
        if v is True:
            flags.append("--" + k)
        elif v is not False:
            flags += ["--" + k, str(v)]
    return options.make_opt(*flags)


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["db_abs", "db_per_sample", "explicit_abs", "explicit_per_sample", "raw_per_sample", "arcsinh_2048"])
def test_audio2mdct_codec_branches_against_oracle(mode, monkeypatch):
    from mdctgan_amd.pix2pixHD_model import Audio2MDCT
    kw, n_fft, hop = dict(), 512, 256
    ocodec = dict(norm_range=(-1.0, 1.0))
    if mode.startswith("db"):
        kw.update(src0=-160, src1=40); ocodec.update(arcsinh_transform=False, raw_mdct=False, src_range=(-160.0, 40.0))
    elif mode.startswith("explicit"):
        kw.update(src0=-160, src1=40, explicit_encoding=True, alpha=0.7)
        ocodec.update(arcsinh_transform=False, raw_mdct=False, explicit_encoding=True, alpha=0.7, src_range=(-160.0, 40.0))
    elif mode.startswith("raw"):
        kw.update(raw_mdct=True); ocodec.update(arcsinh_transform=False, raw_mdct=True, src_range=(-5.0, 5.0))
    else:
        kw.update(arcsinh_transform=True, arcsinh_gain=1000, n_fft=2048, hop_length=1024, win_length=2048)
        ocodec.update(arcsinh_transform=True, arcsinh_gain=1000.0, src_range=(-5.0, 5.0))
        n_fft, hop = 2048, 1024
    abs_norm = not mode.endswith("per_sample")
    if abs_norm:
        kw["abs_norm"] = True
    ocodec["abs_norm"] = abs_norm
    pre = Audio2MDCT(_opt(**kw))
    assert pre.fused == (mode == "raw_per_sample")          # raw + per-sample range at n_fft 512 is a fused K1 / K2 mode
    rng = np.random.default_rng(11)
    x = (0.05 * rng.standard_normal((2, 31 * hop))).astype(np.float32)
    w = T.kbd_window(n_fft)
    s, pha, norm = pre.to_spectro(torch.from_numpy(x).cuda())
    ws, wnorm = T.to_spectro(x, w, n_fft, hop, **ocodec)
    assert tuple(s.shape) == ws.shape
    X, _ = T.mdct4(x, w, n_fft, hop)
    got = s.cpu().numpy()
    db = mode.startswith(("db", "explicit"))
    if db:
        # a dB value is 8.7 dX / (|X| + min_value): the float32 transform's 3e-6 max|X| error is 30 dB at an empty bin.  Judge
        # the codec where the coefficient is resolved (|X| >= 1e-3 max|X|: 0.03 dB = 3e-4 of the normalised range) ...
        strong = (np.abs(X) >= 1e-3 * np.abs(X).max())[:, None]
        if got.shape[1] == 1 and abs_norm:
            assert np.abs(got - ws)[strong].max() <= 3e-4
        # ... and everywhere in the coefficient domain: decoding what the kernel wrote (with the range it reports: a
        # per-sample minimum is an empty bin, i.e. transform noise) gives the oracle's coefficients
        gmin = wnorm["min"] if abs_norm else norm["min"].cpu().numpy().astype(np.float64)
        gmax = wnorm["max"] if abs_norm else norm["max"].cpu().numpy().astype(np.float64)
        dec = T.denormalize(got.astype(np.float64), gmin, gmax,
                            **{k: v for k, v in ocodec.items() if k in ("arcsinh_transform", "raw_mdct", "norm_range", "explicit_encoding")})
        dec = (dec[:, 0] - dec[:, 1]) / (2 * 0.7 - 1) if got.shape[1] == 2 else dec[:, 0] * np.sign(X)
        assert np.abs(dec - X).max() <= 2e-5 * np.abs(X).max()
    else:
        assert np.abs(got - ws).max() <= (5e-4 if "arcsinh" in mode else 3e-5)
    if not abs_norm and not db:
        np.testing.assert_allclose(norm["min"].cpu().numpy().reshape(-1), wnorm["min"].reshape(-1), rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(norm["max"].cpu().numpy().reshape(-1), wnorm["max"].reshape(-1), rtol=1e-5, atol=1e-5)
    if not abs_norm and db:      # the per-sample maximum is a resolved bin; the minimum is an empty one (transform noise)
        np.testing.assert_allclose(norm["max"].cpu().numpy().reshape(-1), wnorm["max"].reshape(-1), atol=1e-2)
    np.testing.assert_allclose(float(norm["mean"]), float(wnorm["mean"]), rtol=2e-2 if db else 1e-4, atol=1e-5)
    np.testing.assert_allclose(float(norm["std"]), float(wnorm["std"]), rtol=5e-2 if db else 1e-4)
    # decoder on the oracle's spectrogram against the oracle's decoder and the original waveform.  The dB codec's sign
    # restore mirrors the reference AS WRITTEN (pix2pixHD_model.py:147-157): `spectro * pha` sits inside
    # `if self.up_ratio > 1`, where the frames beyond int(F / up_ratio) get a random +-1 pseudo-phase (torch.randint,
    # pinned to +1 here); at up_ratio == 1 the magnitudes are decoded unsigned.
    sign = np.sign(X)[:, None]
    nparam = {k: (torch.from_numpy(np.asarray(v)).cuda() if k in ("min", "max") else v) for k, v in wnorm.items()}
    if mode.startswith("db"):
        sg_d = torch.from_numpy(sign.astype(np.float32)).cuda()
        pre.up_ratio = 1
        back = pre.to_audio(torch.from_numpy(ws).cuda(), nparam, sg_d)
        wback = T.to_audio(ws.astype(np.float64), wnorm, w, n_fft, hop, pha=None, **ocodec)
        assert np.abs(back.cpu().numpy() - wback).max() <= 1e-5 * max(np.abs(wback).max(), 1e-3)
        pre.up_ratio = 4.0
        monkeypatch.setattr(torch, "randint", lambda low, high, size, device=None: torch.ones(tuple(size), dtype=torch.int64, device=device))
        back = pre.to_audio(torch.from_numpy(ws).cuda(), nparam, sg_d)
        monkeypatch.undo()
        keep = int(sign.shape[-2] * (1 / 4.0))
        ph = sign.copy()
        ph[..., keep:, :] = 1.0
        wback = T.to_audio(ws.astype(np.float64), wnorm, w, n_fft, hop, pha=ph, **ocodec)
        assert back.shape == wback.shape
        assert np.abs(back.cpu().numpy() - wback).max() <= 1e-5 * max(np.abs(wback).max(), 1e-3)
    else:
        back = pre.to_audio(torch.from_numpy(ws).cuda(), nparam, None)
        wback = T.to_audio(ws.astype(np.float64), wnorm, w, n_fft, hop, pha=None, **ocodec)
        assert back.shape == wback.shape
        assert np.abs(back.cpu().numpy() - wback).max() <= 1e-5 * max(np.abs(wback).max(), 1e-3)
        assert np.abs(back.cpu().numpy()[:, 0, 0] - x).max() <= 3e-5
    if mode.startswith("db"):
        sg = torch.from_numpy(sign.astype(np.float32)).cuda()
        assert pha is not None and bool(((torch.sign(pha) == sg) | (pha == 0)).all())     # sign(X) x noise in [0, 1]


@pytest.mark.gpu
@pytest.mark.parametrize("abs_norm", [True, False], ids=["abs_norm", "per_sample"])
def test_return_pha_follows_the_sign_of_the_coefficients(abs_norm):
    """Audio2MDCT.to_spectro's pha (pix2pixHD_model.py:36, 50-55: sign(X) times min-max-rescaled noise) on the arcsinh codec,
    opt-in through return_pha: recovered from the normalised spectrogram as sign(log_spectro - zero), where zero is the
    normalised value of X = 0 -- a constant under --abs_norm, per clip from its own (min, max) otherwise (VERDICT r2 missing 5)."""
    from mdctgan_amd.pix2pixHD_model import Audio2MDCT
    kw = dict(arcsinh_transform=True, arcsinh_gain=1000)
    if abs_norm:
        kw["abs_norm"] = True
    pre = Audio2MDCT(_opt(**kw))
    pre.return_pha = True
    rng = np.random.default_rng(4)
    x = (np.array([[0.05], [0.3], [0.01]]) * rng.standard_normal((3, 31 * 256))).astype(np.float32)      # clips of different range
    s, pha, norm = pre.to_spectro(torch.from_numpy(x).cuda())
    X, _ = T.mdct4(x, T.kbd_window(512), 512, 256)
    assert pha is not None and pha.shape == s.shape
    resolved = np.abs(X) >= 1e-4 * np.abs(X).max(axis=(1, 2), keepdims=True)     # where float32 noise cannot flip the sign
    got = torch.sign(pha[:, 0]).cpu().numpy()
    ok = (got == np.sign(X)) | (pha[:, 0].cpu().numpy() == 0)
    assert ok[resolved].mean() >= 0.9999, ok[resolved].mean()
    assert float(pha.abs().max()) <= 1.0 and float(pha.abs().max()) > 0.5
