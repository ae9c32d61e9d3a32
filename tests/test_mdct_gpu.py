"""K1 / K2 (csrc/mdct.hip) through the C ABI against the CPU oracle and the golden vectors captured
from the reference.  Stated tolerances (float32 MFMA contraction vs the reference's complex128 FFT):
  MDCT coefficients   <= 2e-6 * max|X|          (3e-5 abs at |X|max ~ 56)
  normalised spectro  <= 5e-4 abs in [-1, 1]    (arcsinh gain 1000 amplifies near-zero bins)
  IMDCT waveform      <= 2e-6 * max|y| (+1e-7)
  round trip          <= 5e-6 at sigma = 1 (max over 2M samples)
"""
import os

import numpy as np
import pytest
import torch

from oracle import transform

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module")
def mods(golden):
    """The window tensor is the golden one (captured from the reference in the build container): torch's float32
    kaiser_window / cumsum differ in the last bit between CPU micro-architectures, for the reference just as for
    us, and the golden spectra were produced with that exact window."""
    from mdctgan_amd.mdct import IMDCT4, MDCT4
    w = torch.from_numpy(golden("g1_kbdwin")["w512"])
    return MDCT4(512, 256, 512, w, device=DEV), IMDCT4(512, 256, 512, w, device=DEV), w.numpy()


def test_mdct4_golden(mods, golden):
    mdct, _, w = mods
    g = golden("g2_mdct4")
    X, frames = mdct(torch.from_numpy(g["x"]).to(DEV), True)
    assert X.shape == (2, 32, 256) and frames.shape == (2, 32, 512)
    np.testing.assert_array_equal(frames.cpu().numpy(), g["frames"])
    Xc = X.cpu().numpy().astype(np.float64)
    for b in range(2):
        assert np.abs(Xc[b] - g["X"][b]).max() <= 2e-6 * np.abs(g["X"][b]).max()
    # without frames the factored-transform kernel runs (csrc/mdct_ct.h; returned frames take the generic dense-table kernel of
    # csrc/mdct.hip): same float32 window products and fold, the DCT-IV as two short stage sums -> the same 2e-6 bar, not the same bits
    X2, fr2 = mdct(torch.from_numpy(g["x"]).to(DEV))
    assert fr2.numel() == 1
    for b in range(2):
        assert np.abs(X2[b].cpu().numpy().astype(np.float64) - g["X"][b]).max() <= 2e-6 * np.abs(g["X"][b]).max()


def test_mdct4_ragged_and_1d(mods):
    mdct, _, w = mods
    rng = np.random.default_rng(3)
    for T in (7936 + 100, 300, 256, 1):
        x = rng.standard_normal((3, T)).astype(np.float32)
        want, _ = transform.mdct4(x, w, 512, 256)
        got, _ = mdct(torch.from_numpy(x).to(DEV))
        assert got.shape == want.shape
        assert np.abs(got.cpu().numpy() - want).max() <= 2e-6 * max(1.0, np.abs(want).max())
    x1 = rng.standard_normal(7936).astype(np.float32)
    got, _ = mdct(torch.from_numpy(x1).to(DEV))
    want, _ = transform.mdct4(x1[None], w, 512, 256)
    assert got.shape == (32, 256)
    assert np.abs(got.cpu().numpy() - want[0]).max() <= 2e-6 * np.abs(want).max()


def test_imdct4_golden(mods, golden):
    _, imdct, _ = mods
    g = golden("g3_imdct4")
    for kin, kout in (("X", "y"), ("Xr", "yr")):
        y, fr = imdct(torch.from_numpy(g[kin]).to(DEV).float())
        assert y.shape == g[kout].shape and fr.numel() == 1
        err = np.abs(y.cpu().numpy() - g[kout]).max()
        assert err <= 2e-6 * np.abs(g[kout]).max() + 1e-7, err
    y, fr = imdct(torch.from_numpy(g["X"]).to(DEV).float(), True)
    assert np.abs(fr.cpu().numpy() - g["yframes"]).max() <= 2e-6 * np.abs(g["yframes"]).max()


def test_round_trip_full_size(mods):
    mdct, imdct, _ = mods
    torch.manual_seed(0)
    x = torch.randn(64, 32512, device=DEV)
    X, _ = mdct(x)
    assert X.shape == (64, 128, 256)
    y, _ = imdct(X)
    assert y.shape == (64, 1, 1, 32512)
    assert (y[:, 0, 0] - x).abs().max().item() < 5e-6     # max over 2.08M samples, |x| up to ~5.3
    # linearity at full size
    x2 = torch.randn(64, 32512, device=DEV)
    X2, _ = mdct(x2)
    X12, _ = mdct(x + 2 * x2)
    assert (X12 - (X + 2 * X2)).abs().max().item() < 2e-4
    # float64 output option widens the float32 result
    from mdctgan_amd.mdct import IMDCT4
    im64 = IMDCT4(512, 256, 512, imdct.window, device=DEV, dtype=torch.float64)
    y64, _ = im64(X)
    # (the float64 store is the generic dense-table kernel's (csrc/mdct.hip), the float32 one the factored kernel's (csrc/mdct_ct.h):
    # other summation orders, same bar)
    assert y64.dtype == torch.float64 and (y64.float() - y).abs().max().item() <= 2e-6 * y.abs().max().item()


def _codec_kw(abs_norm):
    return dict(arcsinh_transform=True, raw_mdct=False, arcsinh_gain=1000.0, abs_norm=abs_norm,
                src_range=(-5.0, 5.0), norm_range=(-1.0, 1.0))


def test_codec_kernels_golden(mods, golden):
    from mdctgan_amd import _lib
    from mdctgan_amd.mdct import dct4_table, imdct4_codec, mdct4_codec
    w = mods[0].window
    d4 = dct4_table(256, DEV)
    for tag, per_sample in (("abs", False), ("minmax", True)):
        g = golden("g4_codec_" + tag)
        r = mdct4_codec(torch.from_numpy(g["x"]).to(DEV), w, d4, 512, codec=_lib.MG_CODEC_ARCSINH, gain=1000.0,
                        norm_range=(-1.0, 1.0), src_range=(-5.0, 5.0), per_sample=per_sample, want_pair=True,
                        want_stats=True)
        s = r["spec"].cpu().numpy()
        assert np.abs(s - g["log_spectro"][:, 0]).max() <= 5e-4
        pair = r["pair"].cpu().numpy()
        np.testing.assert_array_equal(pair[..., 0], s)
        np.testing.assert_allclose(pair[..., 1], np.abs(s) * 2 - 1, atol=1e-7)
        n = s.size
        st = r["stats"].cpu().numpy()
        mean, var = st[0] / n, (st[1] - st[0] ** 2 / n) / (n - 1)
        np.testing.assert_allclose(mean, g["mean"], atol=2e-5)
        np.testing.assert_allclose(np.sqrt(var), g["std"], rtol=1e-4)
        if per_sample:
            np.testing.assert_allclose(r["min"].cpu().numpy(), g["min"].reshape(-1), atol=2e-4)
            np.testing.assert_allclose(r["max"].cpu().numpy(), g["max"].reshape(-1), atol=2e-4)
        mn = torch.from_numpy(g["min"].reshape(-1)).to(DEV) if per_sample else None
        mx = torch.from_numpy(g["max"].reshape(-1)).to(DEV) if per_sample else None
        audio, _ = imdct4_codec(torch.from_numpy(g["log_spectro"][:, 0]).to(DEV), w, d4, 512,
                                codec=_lib.MG_CODEC_ARCSINH, gain=1000.0, norm_range=(-1.0, 1.0),
                                src_range=(-5.0, 5.0), min_b=mn, max_b=mx)
        ref = g["audio"][:, 0, 0]
        assert np.abs(audio.cpu().numpy() - ref).max() <= 2e-6 * np.abs(ref).max() + 1e-7


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_stitch_segments_golden_and_oracle(dtype, golden):
    """generate_audio.py:40-53 (fixture G8 captured from the reference's fold-based cross-fade) and the oracle on
    ragged overlaps; float64 is bit-exact (two addends at most), float32 to 1 ulp of the sum."""
    from mdctgan_amd import ops
    from oracle import transform as T
    g = golden("g8_stitch")
    seg = torch.from_numpy(g["seg"]).to(dtype).to(DEV)
    ov = int(g["overlap"])
    tol = 0.0 if dtype == torch.float64 else 1e-6
    got = ops.stitch_segments(seg, seg.shape[-1], ov).cpu().double().numpy()
    assert got.shape == g["stitched"].shape and np.abs(got - g["stitched"]).max() <= tol
    got = ops.stitch_segments(seg, seg.shape[-1], 0).cpu().double().numpy()
    assert got.shape == g["concat"].shape and np.abs(got - g["concat"]).max() <= tol
    rng = np.random.default_rng(3)
    for n_seg, L, ov in ((1, 64, 8), (5, 96, 47), (4, 128, 1), (7, 50, 20)):
        a = rng.standard_normal((n_seg, 1, 1, L))
        want = T.stitch_segments(a, L, ov)
        got = ops.stitch_segments(torch.from_numpy(a).to(dtype).to(DEV), L, ov).cpu().double().numpy()
        assert got.shape == want.shape and np.abs(got - want).max() <= (1e-12 if dtype == torch.float64 else 1e-6)
    with pytest.raises(ValueError):
        ops.stitch_segments(seg, seg.shape[-1], seg.shape[-1] // 2)


def test_frame_tile_sizes_agree(mods, monkeypatch):
    """The 32 / 64 / 128-frame workgroup tiles give bit-identical spectra and frames (every output element sees the same
    k-ordered MFMA accumulation); the waveform agrees to rounding (the hop block at a tile boundary takes its
    predecessor frame from the VALU halo path instead of the MFMA tile)."""
    mdct, imdct, _ = mods
    gen = torch.Generator().manual_seed(11)
    x = torch.randn(3, 32512 + 77, generator=gen).to(DEV)        # ragged: 129 frames
    outs = {}
    for ft in ("32", "64", "128"):
        monkeypatch.setenv("MG_MDCT_FT", ft)
        X, fr = mdct(x, True)
        y, yfr = imdct(X, True)
        outs[ft] = (X.clone(), fr.clone(), y.clone(), yfr.clone())
    for ft in ("64", "128"):
        for i in (0, 1, 3):
            assert torch.equal(outs["32"][i], outs[ft][i])
        assert (outs["32"][2] - outs[ft][2]).abs().max().item() <= 2e-6 * outs["32"][2].abs().max().item()


@pytest.mark.parametrize("shape", [(2, 7936), (5, 32512), (3, 32512 + 76), (64, 32512), (1, 260)],
                         ids=["2x32fr", "5x128fr", "3x129fr_ragged", "64x128fr", "1x3fr"])
def test_factored_kernels_match_generic_kernels_and_oracle(mods, shape, monkeypatch):
    """csrc/mdct_ct.h (K1 / K2 with the DCT-IV factored into 8- and 16-point DFT stages: the default)
    against the generic dense-table kernels of csrc/mdct.hip (MG_MDCT_CT=0) and the float64 oracle: raw coefficients 2e-6 * max|X|, the arcsinh /
    fixed-range codec 5e-4 against the oracle (its own bar) and 2e-6 against the older kernel (same coefficients to 2e-6 of
    the largest, different libm: the fast asinh is a few ulp from asinhf), K2 on one and the same spectrogram 2e-6 * max|y|.
    129 frames per clip: row tiles straddle clips in K1 and the last tile of a clip is ragged in K2."""
    from mdctgan_amd import _lib
    from mdctgan_amd.mdct import dct4_table, imdct4_codec, mdct4_codec
    mdct, imdct, w = mods
    B, T = shape
    gen = torch.Generator().manual_seed(B * 1000 + T)
    x = (0.05 * torch.randn(B, T, generator=gen)).to(DEV)
    win, d4 = torch.from_numpy(w).to(DEV), dct4_table(256, DEV)
    kw = dict(codec=_lib.MG_CODEC_ARCSINH, gain=1000.0, norm_range=(-1.0, 1.0), src_range=(-5.0, 5.0), per_sample=False,
              want_pair=True, want_stats=True)
    out = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("MG_MDCT_CT", mode)
        raw = mdct4_codec(x, win, d4, 512)["spec"]
        r = mdct4_codec(x, win, d4, 512, **kw)
        y, _ = imdct4_codec(r["spec"] if mode == "1" else out["1"][1], win, d4, 512, codec=_lib.MG_CODEC_ARCSINH, gain=1000.0,
                            norm_range=(-1.0, 1.0), src_range=(-5.0, 5.0))
        yraw, _ = imdct4_codec(raw if mode == "1" else out["1"][0], win, d4, 512)
        out[mode] = (raw.clone(), r["spec"].clone(), r["pair"].clone(), r["stats"].clone(), y.clone(), yraw.clone())
    want_raw, _ = transform.mdct4(x.cpu().numpy(), w, 512, 256)
    scale = np.abs(want_raw).max()
    for mode in ("1", "0"):
        assert np.abs(out[mode][0].cpu().numpy() - want_raw).max() <= 2e-6 * scale, mode
    want, _ = transform.to_spectro(x.cpu().numpy(), w, 512, 256, arcsinh_transform=True, raw_mdct=False, arcsinh_gain=1000.0,
                                   abs_norm=True, src_range=(-5.0, 5.0), norm_range=(-1.0, 1.0))
    s_new, s_old = out["1"][1].cpu().numpy(), out["0"][1].cpu().numpy()
    assert np.abs(s_new - want[:, 0]).max() <= 5e-4
    # the two kernels differ by the coefficient rounding (2e-6 * max|X| amplified by d asinh(1000 X) / dX <= 1000 / ln10 / 5
    # per normalised unit at X = 0) and a few ulp of the codec
    assert np.abs(s_new - s_old).max() <= 2e-6 * scale * 1000.0 / np.log(10.0) / 5.0 + 2e-6
    pair = out["1"][2].cpu().numpy()
    np.testing.assert_array_equal(pair[..., 0], s_new)
    np.testing.assert_allclose(pair[..., 1], np.abs(s_new) * 2 - 1, atol=1e-7)
    assert torch.allclose(out["1"][3], out["0"][3], rtol=1e-5)
    # K2: both kernels decode the SAME spectrogram (the new K1's)
    for i in (4, 5):
        a, b_ = out["1"][i].cpu().numpy(), out["0"][i].cpu().numpy()
        assert a.shape == b_.shape
        assert np.abs(a - b_).max() <= 2e-6 * max(np.abs(b_).max(), 1e-3) + 1e-7, i
    # ... and K2(K1(x)) == x
    assert (out["1"][5].reshape(B, -1)[:, :T] - x[:, :out["1"][5].reshape(B, -1).shape[1]]).abs().max().item() <= 5e-6 * max(1.0, x.abs().max().item())


def test_fast_codec_math(mods):
    """The codec arithmetic of csrc/mdct_codec.h on its own: K1's normalised output against a float64 evaluation of
    (asinh(gain X) / ln 10 - min) / (max - min) * (nr1 - nr0) + nr0 on K1's own raw coefficients X (so the contraction's
    rounding drops out): <= 1e-6 in the [-1, 1] range (a float32 ulp of |l| <= 5 is 4.8e-7), over amplitudes from 1e-9
    (series branch) to 50 (log branch); K2's decode (sinh) against float64 on the same spectrogram: 1e-6 relative."""
    from mdctgan_amd import _lib
    from mdctgan_amd.mdct import dct4_table, imdct4_codec, mdct4_codec
    _, _, w = mods
    win, d4 = torch.from_numpy(w).to(DEV), dct4_table(256, DEV)
    gen = torch.Generator().manual_seed(5)
    amp = 10.0 ** torch.linspace(-9, 0.5, 16)[:, None]
    x = (amp * torch.randn(16, 7936, generator=gen)).to(DEV)
    X = mdct4_codec(x, win, d4, 512)["spec"].double().cpu().numpy()
    s = mdct4_codec(x, win, d4, 512, codec=_lib.MG_CODEC_ARCSINH, gain=1000.0, norm_range=(-1.0, 1.0), src_range=(-5.0, 5.0))["spec"]
    want = (np.arcsinh(1000.0 * X) / np.log(10.0) + 5.0) / 10.0 * 2.0 - 1.0
    assert np.abs(s.cpu().numpy() - want).max() <= 1e-6
    small = np.abs(1000.0 * X) < 0.125                 # the series branch: relative accuracy of the log-domain value
    l_got = (s.double().cpu().numpy() + 1.0) / 2.0 * 10.0 - 5.0
    l_want = np.arcsinh(1000.0 * X) / np.log(10.0)
    assert small.sum() > 1000 and np.abs(l_got - l_want)[small].max() <= 1.5e-6    # the float32 roundings of (l + 5) / 10 * 2 - 1 themselves
    # decoder: spectrogram values across the whole range -> raw coefficients (K2's A operand) checked through a RAW K2 of the
    # float64-decoded spectrogram
    sg = (torch.rand(4, 32, 256, generator=gen) * 2 - 1).to(DEV)
    y, _ = imdct4_codec(sg, win, d4, 512, codec=_lib.MG_CODEC_ARCSINH, gain=1000.0, norm_range=(-1.0, 1.0), src_range=(-5.0, 5.0))
    dec = np.sinh(((sg.double().cpu().numpy() + 1.0) / 2.0 * 10.0 - 5.0) * np.log(10.0)) / 1000.0
    y2, _ = imdct4_codec(torch.from_numpy(dec).float().to(DEV), win, d4, 512)
    assert (y - y2).abs().max().item() <= 3e-6 * y2.abs().max().item()


def test_k1_writes_nothing_outside_its_outputs(mods):
    """The last row tile of a ragged batch (3 clips x 129 frames = 387 rows) is masked by the buffer
    descriptor's range check (voffset + scalar row offset against num_records), not by branches: outputs embedded in
    guard-filled arenas must come back with the guards intact and the same values as the stand-alone call."""
    from mdctgan_amd import _lib
    from mdctgan_amd.mdct import dct4_image, dct4_table, mdct4_codec
    _, _, w = mods
    lib = _lib.load()
    B, T = 3, 32512 + 76
    F = lib.mg_mdct4_num_frames(T, 512)
    assert F == 129
    x = (0.05 * torch.randn(B, T, generator=torch.Generator().manual_seed(1))).to(DEV)
    win, d4 = torch.from_numpy(w).to(DEV), dct4_table(256, DEV)
    n = B * F * 256
    G = 1 << 16
    arena_s = torch.full((G + n + G,), 7.5, device=DEV)
    arena_p = torch.full((G + 2 * n + G,), 7.5, device=DEV)
    spec, pair = arena_s[G:G + n], arena_p[G:G + 2 * n]
    arena_s.fill_(7.5)
    arena_p.fill_(7.5)
    rc = lib.mg_mdct4_forward(_lib.ptr(x), B, T, 512, _lib.ptr(win), _lib.ptr(d4), dct4_image(d4, 256), _lib.MG_CODEC_ARCSINH, 1000.0,
                              -1.0, 1.0, -5.0, 5.0, 0, spec.data_ptr(), pair.data_ptr(), None, None, None, None, None, _lib.stream())
    assert rc == 0
    torch.cuda.synchronize()
    for arena, m in ((arena_s, n), (arena_p, 2 * n)):
        assert bool((arena[:G] == 7.5).all()) and bool((arena[G + m:] == 7.5).all())
    r = mdct4_codec(x, win, d4, 512, codec=_lib.MG_CODEC_ARCSINH, gain=1000.0, norm_range=(-1.0, 1.0), src_range=(-5.0, 5.0), want_pair=True)
    # (with the pair the launcher hands out the spectrogram as channel 0 of the pair)
    assert torch.equal(r["spec"].reshape(-1), spec) and torch.equal(r["pair"].reshape(-1), pair)
    assert r["spec"].data_ptr() == r["pair"].data_ptr() and bool((spec != 7.5).all())
    # a caller written against the round-1 ABI: the plain m x m table and no image -> the kernels that need none, same values
    plain = d4[:256 * 256].clone()
    spec1 = torch.empty_like(spec)
    rc = lib.mg_mdct4_forward(_lib.ptr(x), B, T, 512, _lib.ptr(win), _lib.ptr(plain), None, _lib.MG_CODEC_ARCSINH, 1000.0,
                              -1.0, 1.0, -5.0, 5.0, 0, spec1.data_ptr(), None, None, None, None, None, None, _lib.stream())
    # (two float32 evaluations of arcsinh(1000 X): bins next to zero differ by a few 1e-5 of the [-1, 1] range; the bar vs the
    # reference is 5e-4)
    assert rc == 0 and (spec1 - spec).abs().max().item() <= 2e-4


@pytest.mark.parametrize("shape", [(5, 32512), (3, 32512 + 76), (64, 32512), (2, 7936)],
                         ids=["5x128fr", "3x129fr_ragged", "64x128fr", "2x32fr"])
def test_large_batch_kernels_against_oracle(mods, golden, shape, monkeypatch):
    """csrc/mdct_ct.h (the DCT-IV factored into 8- and 16-point DFT stages on the f32 pipe; the default K1 / K2 at every size)
    held to the same bars as every other K1 / K2 against the float64
    oracle: raw coefficients 2e-6 * max|X|, normalised spectrogram 5e-4, pair channel 0 == spectrogram and channel 1 = 2|v| - 1,
    statistics, waveform 2e-6 * max|y|, K2(K1(x)) == x; clips of 129 frames: row tiles straddle clips in K1 and the last
    tile of a clip is ragged in K2.  Plus the golden fixture G2 / G3 through the same kernels."""
    from mdctgan_amd import _lib
    from mdctgan_amd.mdct import dct4_table, imdct4_codec, mdct4_codec
    _, _, w = mods
    monkeypatch.setenv("MG_MDCT_CT", "1")
    B, T = shape
    gen = torch.Generator().manual_seed(B * 1000 + T + 7)
    x = (0.05 * torch.randn(B, T, generator=gen)).to(DEV)
    win, d4 = torch.from_numpy(w).to(DEV), dct4_table(256, DEV)
    kw = dict(codec=_lib.MG_CODEC_ARCSINH, gain=1000.0, norm_range=(-1.0, 1.0), src_range=(-5.0, 5.0))
    raw = mdct4_codec(x, win, d4, 512)["spec"]
    r = mdct4_codec(x, win, d4, 512, want_pair=True, want_stats=True, **kw)
    r1 = mdct4_codec(x, win, d4, 512, want_stats=True, **kw)                 # spectrogram only
    want_raw, _ = transform.mdct4(x.cpu().numpy(), w, 512, 256)
    scale = np.abs(want_raw).max()
    assert np.abs(raw.cpu().numpy() - want_raw).max() <= 2e-6 * scale
    want, _ = transform.to_spectro(x.cpu().numpy(), w, 512, 256, arcsinh_transform=True, raw_mdct=False, arcsinh_gain=1000.0,
                                   abs_norm=True, src_range=(-5.0, 5.0), norm_range=(-1.0, 1.0))
    s_pair, s_only = r["spec"].cpu().numpy(), r1["spec"].cpu().numpy()
    assert np.abs(s_only - want[:, 0]).max() <= 5e-4 and np.array_equal(s_pair, s_only)
    pair = r["pair"].cpu().numpy()
    np.testing.assert_array_equal(pair[..., 0], s_only)
    np.testing.assert_allclose(pair[..., 1], np.abs(s_only) * 2 - 1, atol=1e-7)
    l64 = np.arcsinh(1000.0 * want_raw) / np.log(10.0)
    st = r1["stats"].cpu().numpy()
    assert abs(st[0] - l64.sum()) <= 1e-5 * np.abs(l64).sum() and abs(st[1] - (l64 ** 2).sum()) <= 1e-5 * (l64 ** 2).sum()
    assert torch.equal(r["stats"], r1["stats"])
    # K2 on the float64-exact coefficients / spectrogram
    yraw, _ = imdct4_codec(raw, win, d4, 512)
    want_y, _ = transform.imdct4(raw.double().cpu().numpy(), w, 512, 256)
    want_y = want_y.reshape(B, -1)
    got_y = yraw.reshape(B, -1).cpu().numpy()
    assert got_y.shape == want_y.shape and np.abs(got_y - want_y).max() <= 2e-6 * np.abs(want_y).max() + 1e-7
    assert (yraw.reshape(B, -1)[:, :T] - x[:, :got_y.shape[1]]).abs().max().item() <= 5e-6 * max(1.0, x.abs().max().item())
    y, _ = imdct4_codec(r1["spec"], win, d4, 512, **kw)
    want_a = transform.to_audio(r1["spec"][:, None].cpu().numpy(), {"min": np.float64(-5.0), "max": np.float64(5.0)}, w, 512, 256,
                                arcsinh_transform=True, raw_mdct=False, arcsinh_gain=1000.0, abs_norm=True, src_range=(-5.0, 5.0),
                                norm_range=(-1.0, 1.0))
    want_a = np.asarray(want_a).reshape(B, -1)
    # (the float32 decode -- one fma + sinh on v_exp_f32, 3e-7 relative per coefficient -- in front of the transform)
    assert np.abs(y.reshape(B, -1).cpu().numpy() - want_a).max() <= 2e-5 * np.abs(want_a).max() + 1e-7
    # the golden fixtures of the reference through the forced kernels
    g2, g3 = golden("g2_mdct4"), golden("g3_imdct4")
    X = mdct4_codec(torch.from_numpy(g2["x"]).to(DEV), win, d4, 512)["spec"]
    assert np.abs(X.cpu().numpy() - g2["X"]).max() <= 2e-6 * np.abs(g2["X"]).max()
    yg, _ = imdct4_codec(torch.from_numpy(g3["Xr"]).float().to(DEV), win, d4, 512)
    assert np.abs(yg.cpu().numpy() - g3["yr"].reshape(yg.shape)).max() <= 2e-6 * np.abs(g3["yr"]).max() + 1e-7


@pytest.mark.parametrize("overlap", [0, 1024, 256, 100], ids=["cat", "ov1024", "ov256", "ov100_generic"])
def test_stitched_k2_equals_decode_then_stitch(mods, golden, overlap):
    """mg_imdct4_stitched (generate_audio.py:40-53 inside K2's overlap-add store): 7 segments decoded in batches of 3 / 3 / 1
    -- handed over out of order -- land in ONE waveform that is bit for bit mg_stitch_segments(mg_imdct4_forward(...)), whose
    stitching is pinned to the reference's F.fold by fixture G8; overlap 100 is not a multiple of 4 and takes the generic
    kernel (scalar stores), float64 output included.  Then G8 itself through K1 -> stitching K2: the reference's stitched
    waveform to the transform's round-trip accuracy."""
    from mdctgan_amd import _lib, ops
    from mdctgan_amd.mdct import dct4_table, imdct4_codec, mdct4_codec
    _, _, w = mods
    win, d4 = torch.from_numpy(w).to(DEV), dct4_table(256, DEV)
    lib = _lib.load()
    gen = torch.Generator().manual_seed(31 + overlap)
    n_seg, F = 7, 128
    spec = (2 * torch.rand(n_seg, F, 256, generator=gen) - 1).to(DEV)
    kw = dict(codec=_lib.MG_CODEC_ARCSINH, gain=1000.0, norm_range=(-1.0, 1.0), src_range=(-5.0, 5.0))
    for dtype in (torch.float32, torch.float64):
        plain, _ = imdct4_codec(spec, win, d4, 512, out_dtype=dtype, **kw)
        L = plain.shape[-1]
        want = ops.stitch_segments(plain, L, overlap)
        total = lib.mg_stitch_length(n_seg, L, overlap)
        out = torch.full((total,), float("nan"), dtype=dtype, device=DEV)      # (the first batch clears it when overlap > 0)
        for first, n in ((0, 3), (6, 1), (3, 3)):
            imdct4_codec(spec[first:first + n], win, d4, 512, out_dtype=dtype, stitch=(out, overlap, first), **kw)
        assert want.shape == (1, total) and torch.equal(out, want[0]), (overlap, dtype, (out - want[0]).abs().max().item())
        name = lib.mg_mdct_last_kernel(1).decode()
        fast = dtype == torch.float32 and overlap % 4 == 0
        assert ("imdct4_ct_kernel<stitched>" in name) == fast and (fast or "imdct4_kernel" in name), name
    # G8: the reference's segments -> K1 (raw coefficients) -> stitching K2 == the reference's fold-based stitching
    g = golden("g8_stitch")
    seg = torch.from_numpy(g["seg"]).float().reshape(3, -1).to(DEV)
    ov = int(g["overlap"])
    X = mdct4_codec(seg, win, d4, 512)["spec"]
    for o, ref in ((ov, g["stitched"]), (0, g["concat"])):
        out = torch.empty(ref.shape[-1], device=DEV)
        imdct4_codec(X, win, d4, 512, stitch=(out, o, 0))
        assert np.abs(out.cpu().double().numpy() - ref[0]).max() <= 5e-6 * np.abs(ref).max()
    with pytest.raises(ValueError):
        imdct4_codec(X, win, d4, 512, stitch=(torch.empty(100, device=DEV), seg.shape[-1] // 2, 0))
    # edges: ONE segment (both cross-fade zones are cropped away: the waveform is the segment's middle, untouched by the halving),
    # and a single-clip batch landing in the middle of a longer waveform
    one = spec[:1]
    plain1, _ = imdct4_codec(one, win, d4, 512, **kw)
    if overlap:
        out1 = torch.full((lib.mg_stitch_length(1, L, overlap),), float("nan"), device=DEV)
        imdct4_codec(one, win, d4, 512, stitch=(out1, overlap, 0), **kw)
        assert out1.numel() == L - 2 * overlap and torch.equal(out1, plain1[0, overlap:L - overlap])
    out = torch.zeros(lib.mg_stitch_length(n_seg, L, overlap), device=DEV)
    imdct4_codec(spec[3:4], win, d4, 512, stitch=(out, overlap, 3), **kw)
    lo = 3 * (L - overlap) - overlap
    ref = imdct4_codec(spec[3:4], win, d4, 512, **kw)[0][0].clone()
    if overlap:
        ref[:overlap] *= 0.5
        ref[L - overlap:] *= 0.5
    assert torch.equal(out[lo:lo + L], ref) and out[:lo].abs().max().item() == 0 and out[lo + L:].abs().max().item() == 0
