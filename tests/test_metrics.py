"""F2: compute_matrics (util/util.py:132-177) -- oracle sanity on the CPU, the HIP path against the oracle on
the GPU.  The LSD's spectrogram is torchaudio's in the reference; torchaudio is not installed, the oracle restates it
through torch.stft (oracle/metrics.py header), so that term is pinned to torch.stft only."""
import types

import numpy as np
import pytest
import torch

from oracle import metrics as M


def _signals(B=3, T=32512, seed=0):
    rng = np.random.default_rng(seed)
    hr = 0.1 * rng.standard_normal((B, T))
    sr = hr + 0.01 * rng.standard_normal((B, T))
    lr = hr + 0.03 * rng.standard_normal((B, T))
    return hr.astype(np.float32), lr.astype(np.float32), sr.astype(np.float32)


def test_oracle_metrics_sanity():
    hr, lr, sr = _signals()
    mse, snr_sr, snr_lr, a, b, c, lsd = M.compute_matrics(hr, lr, sr)
    assert (a, b, c) == (0, 0, 0)
    assert abs(mse - 1e-4) < 1e-5                                  # noise variance 0.01^2
    assert abs(snr_sr - 20.0) < 0.2 and abs(snr_lr - 10.46) < 0.2     # 10 log10(0.1^2 / sigma^2)
    assert 0.05 < lsd < 0.5
    assert M.compute_matrics(hr, lr, hr + 0.0)[6] == 0.0
    p = M.spectrogram_power(hr, 1024, 512, 1024, np.ones(1024))
    assert p.shape == (3, 513, 64)
    # Parseval on one interior frame (rectangular window, onesided): sum_k c_k |X_k|^2 = N sum_n x_n^2
    x = hr[0, 512 * 4 - 512: 512 * 4 + 512].astype(np.float64)
    c = np.ones(513); c[1:-1] = 2.0
    assert abs((c * p[0, :, 4]).sum() - 1024 * (x ** 2).sum()) < 1e-6 * 1024 * (x ** 2).sum()


@pytest.mark.gpu
@pytest.mark.parametrize("center", [True, False])
def test_compute_matrics_on_device(center):
    from mdctgan_amd.metrics import compute_matrics
    hr, lr, sr = _signals(seed=3)
    opt = types.SimpleNamespace(n_fft=512, hop_length=256, win_length=512, center=center)
    want = M.compute_matrics(hr, lr, sr, center=center)
    got = compute_matrics(torch.from_numpy(hr), torch.from_numpy(lr), torch.from_numpy(sr).to("cuda"), opt)
    assert len(got) == 7 and got[3:6] == (0, 0, 0)
    assert abs(got[0] - want[0]) <= 1e-5 * want[0]
    assert abs(got[1] - want[1]) <= 1e-4 and abs(got[2] - want[2]) <= 1e-4          # dB
    assert abs(got[6] - want[6]) <= 2e-4 * want[6]
    # 1-D waveforms (generate_audio.py:60 passes [1, n] / squeezed tensors)
    got1 = compute_matrics(torch.from_numpy(hr[0]), torch.from_numpy(lr[0]), torch.from_numpy(sr[0]).to("cuda"), opt)
    want1 = M.compute_matrics(hr[0], lr[0], sr[0], center=center)
    assert abs(got1[6] - want1[6]) <= 2e-4 * want1[6] and abs(got1[1] - want1[1]) <= 1e-4


@pytest.mark.gpu
def test_eval_model_loop(tmp_path):
    """train.py:104-134 on the device: inference + compute_matrics per batch, the five averaged columns, a CSV row per
    call, eval mode inside the loop and the previous mode restored -- checked against the per-batch metrics computed here
    from the model's own inference outputs."""
    from mdctgan_amd import options
    from mdctgan_amd.metrics import compute_matrics, eval_model
    from mdctgan_amd.pix2pixHD_model import create_model
    from oracle import nets as onets
    opt = options.make_opt(*options.SPECTRAL_FLAGS, "--lr_sampling_rate", "12000", "--netG", "global", "--ngf", "4",
                           "--n_blocks_global", "2", "--n_blocks_attn_g", "1", "--heads_g", "2", "--dim_head_g", "8",
                           "--num_D", "2", "--ndf", "8", "--batchSize", "2", "--bins", "32", "--segment_length", "7936",
                           "--gpu_ids", "0", "--eval_size", "1")
    model = create_model(opt)
    onets.fill_deterministic(model.netG)
    rng = np.random.default_rng(1)
    batches = [{"LR_audio": torch.from_numpy(0.05 * rng.standard_normal((2, 7936)).astype(np.float32)),
                "HR_audio": torch.from_numpy(0.05 * rng.standard_normal((2, 7936)).astype(np.float32))} for _ in range(4)]
    path = str(tmp_path / "eval.csv")
    assert model.training
    res = eval_model(model, batches, opt, path)
    assert model.training                                   # restored
    assert set(res) == {"err", "snr", "snr_seg", "pesq", "lsd"} and res["pesq"] == 0 and res["snr_seg"] == 0
    # eval_size = 1 -> batches 0 and 1 are scored (the reference breaks after j >= eval_size)
    model.eval()
    want = []
    for b in batches[:2]:
        with torch.no_grad():
            sr = model.inference(b["LR_audio"].cuda())[1]
        want.append(compute_matrics(b["HR_audio"], b["LR_audio"], sr.squeeze(), opt))
    assert abs(res["err"] - np.mean([w[0] for w in want])) <= 1e-6 * res["err"]
    assert abs(res["lsd"] - np.mean([w[6] for w in want])) <= 1e-5 * res["lsd"]
    assert abs(res["snr"] - np.mean([(w[2], w[1]) for w in want])) <= 1e-4
    eval_model(model, batches, opt, path)
    rows = open(path).read().strip().splitlines()
    assert rows[0] == "err,snr,snr_seg,pesq,lsd" and len(rows) == 3


@pytest.mark.gpu
def test_compute_matrics_against_the_reference_fixture(golden):
    """G12: the reference's own compute_matrics (float32 on the host).  The device path accumulates its sums in double and
    runs the STFT as an exact-float32 GEMM, so it sits closer to the float64 yardstick than the reference does: MSE / SNR to
    float32 rounding of the REFERENCE's sums, the LSD within the reference's own float32 error (1.5e-3 on the quiet case)."""
    from mdctgan_amd.metrics import compute_matrics
    g = golden("g12_metrics")
    opt = types.SimpleNamespace(n_fft=int(g["n_fft"]), hop_length=int(g["hop_length"]), win_length=int(g["win_length"]),
                                center=bool(g["center"]))
    cases = [(g["hr0"], g["lr0"], g["sr0"], g["metrics0"]), (g["hr1"], g["lr1"], g["sr1"], g["metrics1"]),
             (g["hr2"], 0.5 * g["hr2"], 0.9 * g["hr2"], g["metrics2"])]
    for hr, lr, sr, want in cases:
        got = compute_matrics(torch.from_numpy(hr), torch.from_numpy(lr), torch.from_numpy(sr).to("cuda"), opt)
        exact = M.compute_matrics(hr, lr, sr, center=opt.center)
        assert got[3:6] == (0, 0, 0)
        assert abs(got[0] - want[0]) <= 2e-6 * want[0]
        assert abs(got[1] - want[1]) <= 2e-5 and abs(got[2] - want[2]) <= 2e-5                # dB
        assert abs(got[6] - want[6]) <= 1.5e-3 * want[6]
        assert abs(got[6] - exact[6]) <= 2e-4 * exact[6]
