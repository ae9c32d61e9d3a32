#!/usr/bin/env python
"""Generate tests/golden/*.npz by running the REFERENCE itself (read-only import
from /root/reference) in the build container.  Test infrastructure; never runs
on the GPU box (the reference does not travel) -- only the vectors do.

    python oracle/gen_golden.py            # rewrites tests/golden/

The reference has no tests / golden vectors of its own; these
captured outputs are what pins the oracle (tests/test_oracle_golden.py) and, through
the oracle and directly, the HIP path (tests/test_*_gpu.py).

Stub modules: torch_scatter (FastMDCT4 only), torchvision.models (dead Vgg19),
torchaudio.functional (dB branches only) are absent from this image and unused on
the hot path; empty stand-ins let `models.*` import.  `bottleneck_transformer_pytorch`
is absent too: the G7 fixture (keys/shapes only, parity unpinned) uses the oracle's
restatement as the stand-in.
"""
import argparse
import os
import sys
import tempfile
import types

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
OUT = os.path.join(REPO, "tests", "golden")
sys.path.insert(0, REPO)


def import_reference():
    for name in ["torch_scatter", "torchvision", "torchvision.models", "torchaudio", "torchaudio.functional"]:
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["torch_scatter"].scatter = None
    sys.modules["torchvision"].models = sys.modules["torchvision.models"]
    sys.modules["torchaudio"].functional = sys.modules["torchaudio.functional"]
    from oracle import nets as onets
    bot = types.ModuleType("bottleneck_transformer_pytorch")

    def BottleStack(*, dim, fmap_size, dim_out, num_layers, proj_factor, downsample, heads, dim_head, activation,
                    rel_pos_emb):
        assert not downsample and not rel_pos_emb
        return onets.BotStackRef(dim, fmap_size, dim_out, num_layers, proj_factor, heads, dim_head)
    bot.BottleStack = BottleStack
    sys.modules["bottleneck_transformer_pytorch"] = bot
    sys.path.insert(0, REF)
    import models.mdct as rmdct
    import models.networks as rnet
    import models.pix2pixHD_model as rmodel
    import util.util as rutil
    return rmdct, rnet, rmodel, rutil


def ref_options(extra):
    """TrainOptions().parse() of the reference, in a scratch cwd (it writes opt.txt)."""
    from options.train_options import TrainOptions
    argv = sys.argv
    cwd = os.getcwd()
    tmp = tempfile.mkdtemp()
    try:
        os.chdir(tmp)
        sys.argv = ["train.py", "--gpu_ids", "-1", "--name", "golden"] + extra
        import contextlib
        import io
        with contextlib.redirect_stdout(io.StringIO()):
            opt = TrainOptions().parse()
    finally:
        sys.argv = argv
        os.chdir(cwd)
    return opt


SPECTRAL = ["--arcsinh_transform", "--abs_spectro", "--arcsinh_gain", "1000", "--norm_range", "-1", "1",
            "--src_range", "-5", "5", "--lr_sampling_rate", "12000"]


def quiet(fn, *a, **k):
    import contextlib
    import io
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=OUT)
    args = ap.parse_args()
    os.makedirs(args.out, exist_ok=True)
    rmdct, rnet, rmodel, rutil = import_reference()
    from oracle import nets as onets
    g = torch.Generator().manual_seed(1234)

    def save(name, **arrs):
        np.savez_compressed(os.path.join(args.out, name + ".npz"), **arrs)
        print("wrote", name, {k: (v.shape, str(v.dtype)) for k, v in arrs.items() if hasattr(v, "shape")})

    # ---- G1 window ----------------------------------------------------
    save("g1_kbdwin", w512=rutil.kbdwin(512).numpy(), w1024=rutil.kbdwin(1024).numpy())

    # ---- G2/G3 transform ----------------------------------------------
    win = rutil.kbdwin(512)
    mdct = rmdct.MDCT4(n_fft=512, hop_length=256, win_length=512, window=win, device="cpu")
    imdct = rmdct.IMDCT4(n_fft=512, hop_length=256, win_length=512, window=win, device="cpu")
    x = torch.randn(2, 7936, generator=g)
    x[1] *= 0.05
    X, frames = mdct(x, True)
    y, yframes = imdct(X.clone(), True)
    Xr = torch.randn(2, 32, 256, generator=g, dtype=torch.float64)
    yr, _ = imdct(Xr.clone())
    save("g2_mdct4", x=x.numpy(), X=X.numpy(), frames=frames.numpy())
    save("g3_imdct4", X=X.numpy(), y=y.numpy(), yframes=yframes.numpy(), Xr=Xr.numpy(), yr=yr.numpy())

    # ---- G4 codec -----------------------------------------------------
    xa = 0.05 * torch.randn(2, 7936, generator=g)
    for tag, extra in (("abs", ["--abs_norm"]), ("minmax", [])):
        opt = ref_options(SPECTRAL + extra)
        pre = rmodel.Audio2MDCT(opt)
        torch.manual_seed(0)
        s, pha, norm = pre.to_spectro(xa)
        audio = pre.to_audio(s, norm, pha)
        save("g4_codec_" + tag, x=xa.numpy(), log_spectro=s.numpy(), max=norm["max"].numpy(),
             min=norm["min"].numpy(), mean=norm["mean"].numpy(), std=norm["std"].numpy(), audio=audio.numpy())

    # ---- G5 generators ------------------------------------------------
    gin = torch.rand(1, 2, 32, 256, generator=g) * 2 - 1
    variants = {
        "global": dict(netG="global", ngf=8, n_downsample_global=4, n_blocks_global=2, n_blocks_local=1),
        "local": dict(netG="local", ngf=4, n_downsample_global=3, n_blocks_global=2, n_blocks_local=1),
        "global_resconv_interp": dict(netG="global", ngf=4, n_downsample_global=3, n_blocks_global=1,
                                      n_blocks_local=1, upsample_type="interpolate", downsample_type="resconv"),
    }
    for tag, v in variants.items():
        net = quiet(rnet.define_G, 2, 1, v["ngf"], v["netG"], v["n_downsample_global"], v["n_blocks_global"], 1,
                    v["n_blocks_local"], "instance", gpu_ids=[], upsample_type=v.get("upsample_type", "transconv"),
                    downsample_type=v.get("downsample_type", "conv"), input_size=(32, 256), n_attn_g=0)
        onets.fill_deterministic(net)
        with torch.no_grad():
            out = net(gin)
        save("g5_netG_" + tag, x=gin.numpy(), y=out.numpy(),
             keys=np.array(list(net.state_dict().keys())),
             shapes=np.array([str(tuple(p.shape)) for p in net.state_dict().values()]))

    # ---- G6 discriminator + losses + one optimisation step ---------------
    rnet.GlobalGenerator.set_freeze = lambda self, *a, **k: None  # reference bug, harness patch
    opt = ref_options(SPECTRAL + ["--abs_norm", "--netG", "global", "--ngf", "4", "--n_blocks_global", "2",
                                  "--n_blocks_attn_g", "0", "--num_D", "2", "--ndf", "8", "--batchSize", "2",
                                  "--bins", "32", "--segment_length", "7936"])
    model = quiet(lambda: rmodel.Pix2PixHDModel())
    quiet(model.initialize, opt)
    onets.fill_deterministic(model.netG)
    onets.fill_deterministic(model.netD)
    hr = 0.05 * torch.randn(2, 7936, generator=g)
    spec = torch.fft.rfft(hr)
    spec[:, spec.shape[-1] // 4:] = 0           # ideal low-pass to 6 kHz (12k -> 48k)
    lr = torch.fft.irfft(spec, n=hr.shape[-1])
    torch.manual_seed(0)
    losses, sr = model._forward(lr, hr, infer=True)
    ld = dict(zip(model.loss_names, [torch.mean(v) for v in losses]))
    loss_D = (ld["D_fake"] + ld["D_real"]) * 0.5            # train.py:175-178
    loss_G = ld["G_GAN"] + ld["G_GAN_Feat"]
    model.optimizer_G.zero_grad(); loss_G.backward()
    gG = {k: p.grad.clone() for k, p in model.netG.named_parameters()}
    gD_from_G = {k: p.grad.clone() for k, p in model.netD.named_parameters()}
    model.optimizer_G.step()
    model.optimizer_D.zero_grad(); loss_D.backward()
    gD = {k: p.grad.clone() for k, p in model.netD.named_parameters()}
    model.optimizer_D.step()
    torch.manual_seed(0)
    losses2, _ = model._forward(lr, hr, infer=False)
    ld2 = dict(zip(model.loss_names, [float(torch.mean(v)) for v in losses2]))
    arrs = dict(lr=lr.numpy(), hr=hr.numpy(), sr_spectro=sr.detach().numpy(),
                loss_names=np.array(model.loss_names),
                losses=np.array([float(ld[k]) for k in model.loss_names]),
                losses_after=np.array([ld2[k] for k in model.loss_names]))
    for k, v in gG.items():
        arrs["gG/" + k] = v.numpy()
    for k, v in gD.items():
        arrs["gD/" + k] = v.numpy()
    for k, v in model.netG.state_dict().items():
        arrs["pG_after/" + k] = v.numpy()
    for k, v in model.netD.state_dict().items():
        arrs["pD_after/" + k] = v.numpy()
    arrs["keysD"] = np.array(list(model.netD.state_dict().keys()))
    # inference through the facade (pix2pixHD_model.py:618-638), after the step
    torch.manual_seed(0)
    sr_s, sr_audio, _, _, lr_s = model.inference(lr)
    arrs["inf_sr_spectro"] = sr_s.numpy()
    arrs["inf_sr_audio"] = sr_audio.numpy()
    arrs["inf_lr_spectro"] = lr_s.numpy()
    save("g6_step_global", **arrs)

    # D features alone (deterministic weights, fresh net)
    netD = quiet(rnet.define_D, 3, 8, 3, "instance", False, 2, True, gpu_ids=[])
    onets.fill_deterministic(netD)
    din = torch.rand(2, 3, 32, 256, generator=g) * 2 - 1
    with torch.no_grad():
        feats = netD(din)
    arrs = {"x": din.numpy()}
    for i, sc in enumerate(feats):
        for j, f in enumerate(sc):
            arrs["f%d_%d" % (i, j)] = f.numpy()
    save("g6_netD", **arrs)

    # ---- G7 local + BoT keys/shapes (UNPINNED numerics) ---------------
    net = quiet(rnet.define_G, 2, 1, 8, "local", 3, 2, 1, 1, "instance", gpu_ids=[], input_size=(64, 256),
                n_attn_g=2, heads_g=2, dim_head_g=16, proj_factor_g=4)
    save("g7_local_bot_keys", keys=np.array(list(net.state_dict().keys())),
         shapes=np.array([str(tuple(p.shape)) for p in net.state_dict().values()]))

    # ---- G8 segment stitching (generate_audio.py:40-53 restated verbatim-in-behaviour) ----
    seg = torch.randn(3, 1, 1, 7936, generator=g, dtype=torch.float64)
    from torch.nn.functional import fold
    ov = 512
    stride = 7936 - ov
    out_len = (3 - 1) * stride + 7936
    a = seg.clone()
    a[..., :ov] *= 0.5
    a[..., -ov:] *= 0.5
    a = a.squeeze().transpose(-1, -2)
    a = fold(a, kernel_size=(1, 7936), stride=(1, stride), output_size=(1, out_len)).squeeze(0)
    a = a[..., ov:-ov]
    save("g8_stitch", seg=seg.numpy(), overlap=np.array(ov), stitched=a.numpy(),
         concat=seg.reshape(1, -1).numpy())

    # ---- G9 the --fp16 branch of train.py:160-202: autocast forward + one GradScaler ----------------------
    # torch.autocast("cpu", float16) stands in for torch.cuda.amp.autocast (same cast policy for conv / conv_transpose
    # -> float16, instance_norm / relu / tanh follow their input, mse_loss / l1_loss -> float32); torch.amp.GradScaler
    # ("cpu") for torch.cuda.amp.GradScaler (same defaults: 65536, x2 / x0.5, interval 2000).
    model = quiet(lambda: rmodel.Pix2PixHDModel())
    quiet(model.initialize, opt)
    onets.fill_deterministic(model.netG)
    onets.fill_deterministic(model.netD)
    hr = 0.05 * torch.randn(2, 7936, generator=g)
    spec = torch.fft.rfft(hr)
    spec[:, spec.shape[-1] // 4:] = 0
    lr = torch.fft.irfft(spec, n=hr.shape[-1])
    scaler = torch.amp.GradScaler("cpu")
    hist = []
    for it in range(2):
        torch.manual_seed(0)
        with torch.autocast("cpu", dtype=torch.float16):
            losses, _ = model._forward(lr, hr, infer=False)
        ld = dict(zip(model.loss_names, [torch.mean(v) for v in losses]))
        loss_D = (ld["D_fake"] + ld["D_real"]) * 0.5
        loss_G = ld["G_GAN"] + ld["G_GAN_Feat"]
        model.optimizer_G.zero_grad()
        scaler.scale(loss_G).backward()
        if it == 0:
            gG = {k: p.grad.clone() / scaler.get_scale() for k, p in model.netG.named_parameters()}
        scaler.step(model.optimizer_G)
        model.optimizer_D.zero_grad()
        scaler.scale(loss_D).backward()
        if it == 0:
            gD = {k: p.grad.clone() / scaler.get_scale() for k, p in model.netD.named_parameters()}
        scaler.step(model.optimizer_D)
        scaler.update()
        hist.append([float(ld[k]) for k in model.loss_names] + [scaler.get_scale()])
    arrs = dict(lr=lr.numpy(), hr=hr.numpy(), loss_names=np.array(model.loss_names),
                losses=np.array(hist[0][:-1]), losses_step2=np.array(hist[1][:-1]),
                scale_after=np.array([h[-1] for h in hist]))
    for k, v in gG.items():
        arrs["gG/" + k] = v.numpy()
    for k, v in gD.items():
        arrs["gD/" + k] = v.numpy()
    save("g9_step_global_fp16", **arrs)

    # ---- G10 LocalEnhancer with the local attention sandwich (networks.py:218-237: the shared-module lists) ----------
    # The reference's own module tree and forward; the bottleneck-transformer arithmetic inside is the stand-in above
    # (third party, unpinned) -- what this pins is the wiring: which modules are shared, applied how often, on which maps.
    net = quiet(rnet.define_G, 2, 1, 4, "local", 2, 1, 1, 3, "instance", gpu_ids=[], input_size=(64, 256),
                n_attn_g=0, n_attn_l=1, proj_factor_l=4, heads_l=2, dim_head_l=8)
    onets.fill_deterministic(net)
    net.eval()
    gin = torch.rand(1, 2, 64, 256, generator=g) * 2 - 1
    with torch.no_grad():
        out = net(gin)
    save("g10_netG_local_attn_l", x=gin.numpy(), y=out.numpy(), keys=np.array(list(net.state_dict().keys())),
         shapes=np.array([str(tuple(p.shape)) for p in net.state_dict().values()]))

    # ---- G11 ImagePool (util/image_pool.py:4-31, --pool_size > 0): which image a query returns -------------------------
    # Images are labelled by a running id (their constant value); the fixture holds the ids of the returned batch for 8
    # queries of 2 images through a pool of 3 under random.seed(7): fill-up, then 50 % swap with a random slot.
    import random
    from util.image_pool import ImagePool
    random.seed(7)
    pool = ImagePool(3)
    returned = []
    for q in range(8):
        batch = torch.stack([torch.full((2, 3, 4), float(2 * q + i)) for i in range(2)])
        returned.append(pool.query(batch)[:, 0, 0, 0].numpy().copy())
    save("g11_image_pool", returned=np.stack(returned), pool_size=np.array(3), seed=np.array(7))

    # ---- G12 compute_matrics (util/util.py:132-177; train.py:104-134 eval_model, generate_audio.py:60-61) ---------------
    # The reference's OWN function on two [2, 32512] triples.  Its one third-party call, torchaudio.functional.spectrogram
    # (util.py:170-171; torchaudio is not installed), is stood in for by the 6-line torch.stft wrapper below -- torchaudio's
    # published definition of spectrogram(pad=0, power=2, normalized=False): stft(center, pad_mode="reflect", onesided) then
    # |.|^power.  Everything else -- MSE, the two SNRs, kbdwin(2 * win_length), the 2x STFT geometry, log10(. + 1e-6), the
    # mean over dim=-2 (FREQUENCY bins) before the sqrt -- is the reference's arithmetic, in float32 as it runs there.
    def spectrogram_stub(waveform, pad, window, n_fft, hop_length, win_length, power, normalized, center=True,
                         pad_mode="reflect", onesided=True):
        assert pad == 0 and power == 2 and normalized is False
        s = torch.stft(waveform.reshape(-1, waveform.shape[-1]), n_fft=n_fft, hop_length=hop_length, win_length=win_length,
                       window=window, center=center, pad_mode=pad_mode, normalized=False, onesided=onesided, return_complex=True)
        return (s.abs() ** power).reshape(waveform.shape[:-1] + s.shape[-2:])
    sys.modules["torchaudio.functional"].spectrogram = spectrogram_stub
    mopt = ref_options(SPECTRAL)
    arrs = dict(n_fft=np.array(mopt.n_fft), hop_length=np.array(mopt.hop_length), win_length=np.array(mopt.win_length),
                center=np.array(bool(mopt.center)))
    for case, (sigma, err) in enumerate([(0.05, 0.3), (1.0, 0.02)]):
        hr_a = sigma * torch.randn(2, 32512, generator=g)
        spec_ = torch.fft.rfft(hr_a)
        spec_[:, spec_.shape[-1] // 4:] = 0
        lr_a = torch.fft.irfft(spec_, n=32512)
        sr_a = lr_a + err * (hr_a - lr_a) + 0.01 * sigma * torch.randn(2, 32512, generator=g)
        got = rutil.compute_matrics(hr_a, lr_a, sr_a, mopt)
        assert got[3:6] == (0, 0, 0)
        arrs.update({"hr%d" % case: hr_a.numpy(), "lr%d" % case: lr_a.numpy(), "sr%d" % case: sr_a.numpy(),
                     "metrics%d" % case: np.array(got, dtype=np.float64)})
    hr_1d = 0.1 * torch.randn(32512, generator=g)                       # generate_audio.py:60 passes [1, T]; 1-D works too
    arrs.update(hr2=hr_1d.numpy(), metrics2=np.array(rutil.compute_matrics(hr_1d, 0.5 * hr_1d, 0.9 * hr_1d, mopt), dtype=np.float64))
    save("g12_metrics", **arrs)

    # ---- G13 dataset chain (data/audio_dataset.py:34-110 AudioDataset, :153-186 AudioTestDataset / AudioAppDataset) ------
    # The reference's dataset classes run as they are; the three torchaudio entry points they touch are stood in for:
    # aF.resample by oracle/resample.py (the restatement of torchaudio's algorithm: that part stays UNPINNED), torchaudio.info /
    # torchaudio.load by an in-memory "file" table.  What this pins is everything around the resampler: the random crop window
    # (max_audio_start scales the segment by fs / hr_rate, the load takes segment_length frames AT THE FILE RATE), the
    # HR / LR = down-then-up order, crop-or-pad to segment_length, the test set's DC shift (+1e-4 - mean), is_lr_input, and
    # seg_pad_audio with --gen_overlap (front pad = overlap, unfold stride = segment_length - overlap).
    from oracle import resample as oresample
    ta, taf = sys.modules["torchaudio"], sys.modules["torchaudio.functional"]
    sys.modules.setdefault("torchvision.transforms", types.ModuleType("torchvision.transforms"))
    sys.modules["torchvision"].transforms = sys.modules["torchvision.transforms"]
    ta.set_audio_backend = lambda name: None
    files = {"a44k.wav": (44100, 0.1 * torch.randn(1, 50000, generator=g)),
             "b48k_short.wav": (48000, 0.1 * torch.randn(1, 5000, generator=g)),
             "c16k.wav": (16000, 0.1 * torch.randn(1, 20000, generator=g))}
    loads = []

    class Meta:
        pass

    def info(path):
        m = Meta()
        m.sample_rate, m.num_frames = files[path][0], files[path][1].shape[-1]
        return m

    def load(path, frame_offset=0, num_frames=-1):
        fs_, w_ = files[path]
        w_ = w_[:, frame_offset:] if num_frames < 0 else w_[:, frame_offset:frame_offset + num_frames]
        loads.append((path, frame_offset, w_.clone()))
        return w_.clone(), fs_

    def resample_stub(waveform, orig_freq, new_freq):
        return torch.from_numpy(oresample.resample(waveform.numpy(), int(orig_freq), int(new_freq))).to(torch.float32)
    ta.info, ta.load, taf.resample = info, load, resample_stub
    import data.audio_dataset as rdata
    seg = 8192
    ds = rdata.AudioDataset.__new__(rdata.AudioDataset)          # __init__ walks opt.dataroot on disk; set its fields directly
    ds.lr_sampling_rate, ds.hr_sampling_rate, ds.segment_length = 12000, 48000, seg
    ds.n_fft, ds.hop_length, ds.win_length, ds.center, ds.add_noise, ds.snr = 512, 256, 512, True, False, 55
    ds.audio_file = list(files)
    ds.audio_len = [(0, 0)] * len(files)
    torch.manual_seed(1234)                                      # AudioDataset.__init__: torch.manual_seed(opt.seed)
    arrs = dict(segment_length=np.array(seg), hr_rate=np.array(48000), lr_rate=np.array(12000), names=np.array(list(files)))
    for i, name in enumerate(files):
        item = ds[i]
        path, off, loaded = loads[-1]
        assert path == name
        arrs.update({"file%d" % i: files[name][1].numpy(), "fs%d" % i: np.array(files[name][0]), "offset%d" % i: np.array(off),
                     "loaded%d" % i: loaded.numpy(), "HR%d" % i: item["HR_audio"].numpy(), "LR%d" % i: item["LR_audio"].numpy()})

    class TOpt:
        pass
    for j, (overlap, is_lr, fs_in, n) in enumerate([(0, False, 48000, 20000), (256, False, 44100, 30000), (128, True, 12000, 6000),
                                                    (64, False, 48000, 3000)]):
        to = TOpt()
        to.lr_sampling_rate, to.hr_sampling_rate, to.segment_length = 12000, 48000, seg
        to.n_fft, to.hop_length, to.win_length, to.center = 512, 256, 512, True
        to.is_lr_input, to.gen_overlap, to.add_noise, to.snr = is_lr, overlap, False, 55
        raw = 0.1 * torch.randn(1, n, generator=g) + 0.01
        # read_audio() (audio_dataset.py:141-151) is torchaudio.load + the in-place DC shift; AudioAppDataset (:186-204, the
        # reference's own in-memory variant) skips it, so the shift is applied here with the reference's statement
        shifted = raw.clone()
        shifted += 1e-4 - torch.mean(shifted)
        app = rdata.AudioAppDataset(to, shifted.clone(), fs_in)
        arrs.update({"t_raw%d" % j: raw.numpy(), "t_fs%d" % j: np.array(fs_in), "t_overlap%d" % j: np.array(overlap),
                     "t_is_lr%d" % j: np.array(is_lr), "t_lr_audio%d" % j: app.lr_audio.numpy(),
                     "t_segments%d" % j: app.seg_audio.numpy(), "t_len%d" % j: np.array(len(app))})
    save("g13_dataset_chain", **arrs)


if __name__ == "__main__":
    main()
