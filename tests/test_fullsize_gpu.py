"""The convolution passes at the FULL layer sizes of BASELINE configs[1], configs[2] (netG=local: the 2048-channel 4x8
trunk, the 128-channel 64x128 local blocks, the ngf-128 stride-2 ladder, the third discriminator scale) and configs[4]
(batch-64 inference plans) -- the shapes bench.py runs: tuned tile / split plans, Winograd, 32-deep kernels, per-class
split-K, tiled single-channel kernels -- where a float64 CPU reference of the whole tensor would take minutes.
Size-independent properties instead:

* exactness on a sample: 48 random output pixels (all output channels) of the forward pass against a float64
  evaluation of those pixels' receptive fields;
* adjointness: <conv(x), dy> == <x, dgrad(dy)> -- the data gradient is the transpose of the forward map;
* bilinearity: <conv(x; w), dy> == <w, wgrad(x, dy)> -- the weight gradient is the transpose in w.

Inner products are accumulated in float64 on the device.  Float32 passes: 2e-5 relative (3e-5 per element on the
sample); MG_PRECISION_F16 passes: 2e-3 (operands rounded to float16)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"

# (name, B, H, W, Ci, Co, k, stride, pad, reflect) -- scripts/bench_conv.py's table of configs[1] layers
SHAPES = [
    ("bottleneck", 8, 8, 16, 1024, 1024, 3, 1, 1, True),
    ("down512", 8, 16, 32, 512, 1024, 3, 2, 1, False),
    ("down256", 8, 32, 64, 256, 512, 3, 2, 1, False),
    ("down128", 8, 64, 128, 128, 256, 3, 2, 1, False),
    ("down64", 8, 128, 256, 64, 128, 3, 2, 1, False),
    ("stem", 8, 128, 256, 2, 64, 7, 1, 3, True),
    ("head", 8, 128, 256, 64, 1, 7, 1, 3, True),
    ("d3_64_b16", 16, 128, 256, 3, 64, 4, 2, 2, False),
    ("d64_128_b16", 16, 65, 129, 64, 128, 4, 2, 2, False),
    ("d128_256_b16", 16, 33, 65, 128, 256, 4, 2, 2, False),
    ("d256_512_b16", 16, 17, 33, 256, 512, 4, 1, 2, False),
    ("dlast_b16", 16, 18, 34, 512, 1, 4, 1, 2, False),
    ("d1_256_512", 8, 9, 17, 256, 512, 4, 1, 2, False),
]

# configs[2] (netG=local ngf 64 -> global branch ngf 128 on the 64x128 average-pooled input, 4 stride-2 stages, 2048-channel
# 4x8 trunk; local branch: 128-channel blocks at 64x128; num_D 3: a third discriminator scale on the 32x64 input).  The
# discriminator runs the batch-stacked [fake, real] pass, i.e. batch 16.
SHAPES_CFG2 = [
    ("c2_trunk2048", 8, 4, 8, 2048, 2048, 3, 1, 1, True),
    ("c2_local128", 8, 64, 128, 128, 128, 3, 1, 1, True),
    ("c2_gstem128", 8, 64, 128, 2, 128, 7, 1, 3, True),
    ("c2_gdown128", 8, 64, 128, 128, 256, 3, 2, 1, False),
    ("c2_gdown256", 8, 32, 64, 256, 512, 3, 2, 1, False),
    ("c2_gdown512", 8, 16, 32, 512, 1024, 3, 2, 1, False),
    ("c2_gdown1024", 8, 8, 16, 1024, 2048, 3, 2, 1, False),
    ("c2_bot_in", 8, 4, 8, 2048, 512, 1, 1, 0, False),          # bottleneck-transformer 1x1 projections
    ("c2_bot_qkv", 8, 4, 8, 512, 1536, 1, 1, 0, False),
    ("c2_bot_out", 8, 4, 8, 512, 2048, 1, 1, 0, False),
    ("c2_d2_3_64_b16", 16, 64, 128, 3, 64, 4, 2, 2, False),     # second scale, stacked batch
    ("c2_d2_64_128_b16", 16, 33, 65, 64, 128, 4, 2, 2, False),
    ("c2_d2_128_256_b16", 16, 17, 33, 128, 256, 4, 2, 2, False),
    ("c2_d2_256_512_b16", 16, 9, 17, 256, 512, 4, 1, 2, False),
    ("c2_d2_last_b16", 16, 10, 18, 512, 1, 4, 1, 2, False),
    ("c2_d3_3_64_b16", 16, 32, 64, 3, 64, 4, 2, 2, False),      # third scale
    ("c2_d3_64_128_b16", 16, 17, 33, 64, 128, 4, 2, 2, False),
    ("c2_d3_128_256_b16", 16, 9, 17, 128, 256, 4, 2, 2, False),
    ("c2_d3_256_512_b16", 16, 5, 9, 256, 512, 4, 1, 2, False),
    ("c2_d3_last_b16", 16, 6, 10, 512, 1, 4, 1, 2, False),
]

# configs[4]: generate_audio at batch 64 (forward only; tile plans added for these M).  (name, B, H, W, Ci, Co, k, s, p, reflect)
SHAPES_CFG4 = [
    ("c4_bottleneck", 64, 8, 16, 1024, 1024, 3, 1, 1, True),
    ("c4_down512", 64, 16, 32, 512, 1024, 3, 2, 1, False),
    ("c4_down256", 64, 32, 64, 256, 512, 3, 2, 1, False),
    ("c4_down128", 64, 64, 128, 128, 256, 3, 2, 1, False),
    ("c4_down64", 64, 128, 256, 64, 128, 3, 2, 1, False),
    ("c4_stem", 64, 128, 256, 2, 64, 7, 1, 3, True),
    ("c4_head", 64, 128, 256, 64, 1, 7, 1, 3, True),
]


def dot64(a, b):
    return (a.double() * b.double()).sum().item()


def sample_reference(x, w, bias, B, H, W, Ci, Co, k, s, p, reflect, OH, OW, n=48, seed=0):
    """float64 outputs at n random (b, oy, ox): returns indices and [n, Co] values.  x [B,H,W,Ci], w [Co,k,k,Ci] on CPU."""
    rng = np.random.default_rng(seed)
    idx = np.stack([rng.integers(0, B, n), rng.integers(0, OH, n), rng.integers(0, OW, n)], 1)
    idx[0] = (0, 0, 0)
    idx[1] = (B - 1, OH - 1, OW - 1)                     # corners: padding / reflection on both sides
    xd, wd = x.double(), w.double()
    out = torch.zeros(n, Co, dtype=torch.float64)
    for j, (b, oy, ox) in enumerate(idx):
        acc = bias.double().clone() if bias is not None else torch.zeros(Co, dtype=torch.float64)
        for ky in range(k):
            iy = oy * s - p + ky
            if reflect:
                iy = -iy if iy < 0 else (2 * (H - 1) - iy if iy >= H else iy)
            elif iy < 0 or iy >= H:
                continue
            for kx in range(k):
                ix = ox * s - p + kx
                if reflect:
                    ix = -ix if ix < 0 else (2 * (W - 1) - ix if ix >= W else ix)
                elif ix < 0 or ix >= W:
                    continue
                acc += wd[:, ky, kx, :] @ xd[b, iy, ix, :]
        out[j] = acc
    return idx, out


@pytest.mark.parametrize("prec", ["f32", "f16"])
@pytest.mark.parametrize("shape", SHAPES + SHAPES_CFG2, ids=[s[0] for s in SHAPES + SHAPES_CFG2])
def test_full_size_layer_properties(shape, prec):
    from mdctgan_amd import _lib, ops
    name, B, H, W, Ci, Co, k, s, p, reflect = shape
    hp = prec == "f16"
    g = ops.conv_geom(B, H, W, Ci, Co, k, k, s, p, reflect, _lib.PRECISION_F16 if hp else _lib.PRECISION_F32)
    gen = torch.Generator().manual_seed(len(name) + Ci)
    x = torch.randn(B, H, W, Ci, generator=gen)
    w = torch.randn(Co * k * k * Ci, generator=gen).reshape(Co, k, k, Ci) / np.sqrt(Ci * k * k)
    bias = torch.randn(Co, generator=gen)
    dy = torch.randn(B, g.OH, g.OW, Co, generator=gen)
    if hp:   # the properties are stated for the values the kernels actually multiply
        x, w, dy = x.half().float(), w.half().float(), dy.half().float()
    xd, wd, bd, dyd = x.to(DEV), w.to(DEV), bias.to(DEV), dy.to(DEV)
    rtol = 2e-3 if hp else 2e-5

    y = ops.conv_fwd(g, xd, wd, bd)
    idx, want = sample_reference(x, w, bias, B, H, W, Ci, Co, k, s, p, reflect, g.OH, g.OW)
    got = y[idx[:, 0], idx[:, 1], idx[:, 2]].double().cpu()
    tol = (4e-3 if hp else 3e-5) * want.abs().max().item()
    assert (got - want).abs().max().item() <= tol, (name, (got - want).abs().max().item(), want.abs().max().item())

    y0 = ops.conv_fwd(g, xd, wd, None)                           # linear part only for the transpose identities
    lhs = dot64(y0, dyd)
    scale = np.sqrt(dot64(y0, y0) * dot64(dyd, dyd))              # Cauchy-Schwarz scale of the inner product
    dx = ops.conv_dgrad(g, dyd, wd)
    assert abs(lhs - dot64(xd, dx)) <= rtol * scale, (name, "adjoint", lhs, dot64(xd, dx), scale)
    dw = torch.empty(Co, k, k, Ci, device=DEV)
    db = torch.empty(Co, device=DEV)
    ops.conv_wgrad(g, xd, dyd, dw, db)
    assert abs(lhs - dot64(wd, dw)) <= rtol * scale, (name, "wgrad", lhs, dot64(wd, dw), scale)
    assert (db.double().cpu() - dy.double().sum((0, 1, 2))).abs().max().item() <= 1e-4 * np.sqrt(B * g.OH * g.OW) * 4


@pytest.mark.parametrize("shape", SHAPES_CFG4, ids=[s[0] for s in SHAPES_CFG4])
def test_batch64_inference_plans(shape):
    """configs[4] runs the generator forward at batch 64: every layer takes a tile plan (and, for the 3x3 blocks, a
    Winograd GEMM shape) that no training shape exercises.  Forward exactness on 48 sampled output pixels against float64,
    plus the no-grad path's cached transformed weights giving the same bits as the uncached call."""
    from mdctgan_amd import _lib, ops
    name, B, H, W, Ci, Co, k, s, p, reflect = shape
    g = ops.conv_geom(B, H, W, Ci, Co, k, k, s, p, reflect, _lib.PRECISION_F32)
    gen = torch.Generator(device=DEV).manual_seed(len(name) + Ci)
    xd = torch.randn(B, H, W, Ci, generator=gen, device=DEV)          # up to 0.5 GB: drawn on the device
    wd = torch.randn(Co * k * k * Ci, generator=gen, device=DEV).reshape(Co, k, k, Ci) / np.sqrt(Ci * k * k)
    bd = torch.randn(Co, generator=gen, device=DEV)
    y = ops.conv_fwd(g, xd, wd, bd)
    idx, want = sample_reference(xd.cpu(), wd.cpu(), bd.cpu(), B, H, W, Ci, Co, k, s, p, reflect, g.OH, g.OW)
    got = y[idx[:, 0], idx[:, 1], idx[:, 2]].double().cpu()
    tol = 3e-5 * want.abs().max().item()
    assert (got - want).abs().max().item() <= tol, (name, (got - want).abs().max().item(), want.abs().max().item())
    u = ops.wino_weights(g, wd)
    if u is not None:
        y2 = ops.conv_fwd(g, xd, wd, bd, u=u)
        assert torch.equal(y, y2), name


def sample_reference_dgrad(dy, w, bias, B, H, W, Ci, Co, k, s, p, OH, OW, n=48, seed=1):
    """float64 data gradient (= ConvTranspose2d forward, models/networks.py:349-352) at n random input pixels (b, iy, ix),
    all Ci channels: dx[b, iy, ix, :] = bias + sum over (oy, ox, ky, kx) with oy*s - p + ky == iy, ox*s - p + kx == ix of
    w[:, ky, kx, :]^T dy[b, oy, ox, :].  dy [B,OH,OW,Co], w [Co,k,k,Ci] on the CPU; zero padding."""
    rng = np.random.default_rng(seed)
    idx = np.stack([rng.integers(0, B, n), rng.integers(0, H, n), rng.integers(0, W, n)], 1)
    idx[0] = (0, 0, 0)
    idx[1] = (B - 1, H - 1, W - 1)
    idx[2] = (B - 1, H - 2, W - 2)                      # both parities next to the border
    dyd, wd = dy.double(), w.double()
    out = torch.zeros(n, Ci, dtype=torch.float64)
    for j, (b, iy, ix) in enumerate(idx):
        acc = bias.double().clone() if bias is not None else torch.zeros(Ci, dtype=torch.float64)
        for ky in range(k):
            ty = iy + p - ky
            if ty % s or not (0 <= ty // s < OH):
                continue
            for kx in range(k):
                tx = ix + p - kx
                if tx % s or not (0 <= tx // s < OW):
                    continue
                acc += dyd[b, ty // s, tx // s, :] @ wd[:, ky, kx, :]
        out[j] = acc
    return idx, out


@pytest.mark.parametrize("shape", [s_ for s_ in SHAPES_CFG4 if s_[7] == 2], ids=[s_[0] + "_up" for s_ in SHAPES_CFG4 if s_[7] == 2])
def test_batch64_inference_up_ladder(shape):
    """configs[4]'s other half: the generator's up-ladder at batch 64 is the DATA-GRADIENT kernel run
    forward (nn.ConvTranspose2d(3, stride 2, pad 1, output_padding 1) + bias, networks.py:349-352) -- 15 % of the
    inference step on plans (conv_dgrad_dma_kernel<128,128> / <128,64> by parity class) that no training shape uses.
    48 sampled output pixels (all channels) against float64, and the adjoint identity against the forward kernel."""
    from mdctgan_amd import _lib, ops
    name, B, H, W, Ci, Co, k, s, p, reflect = shape
    g = ops.conv_geom(B, H, W, Ci, Co, k, k, s, p, reflect, _lib.PRECISION_F32)
    gen = torch.Generator(device=DEV).manual_seed(len(name) + Co)
    dyd = torch.randn(B, g.OH, g.OW, Co, generator=gen, device=DEV)           # the low-resolution input of the ConvTranspose2d
    wd = torch.randn(Co * k * k * Ci, generator=gen, device=DEV).reshape(Co, k, k, Ci) / np.sqrt(Co * k * k / 4.0)
    bd = torch.randn(Ci, generator=gen, device=DEV)
    dx = ops.conv_dgrad(g, dyd, wd, bd)
    assert ops.plan_name(1, g).startswith("conv_dgrad_dma_kernel"), ops.plan_name(1, g)
    idx, want = sample_reference_dgrad(dyd.cpu(), wd.cpu(), bd.cpu(), B, H, W, Ci, Co, k, s, p, g.OH, g.OW)
    got = dx[idx[:, 0], idx[:, 1], idx[:, 2]].double().cpu()
    assert (got - want).abs().max().item() <= 3e-5 * want.abs().max().item(), (name, (got - want).abs().max().item())
    # <conv(x), dy> == <x, dgrad(dy)> with the forward kernel of the same geometry (float64 inner products on the device)
    xd = torch.randn(B, H, W, Ci, generator=gen, device=DEV)
    y0 = ops.conv_fwd(g, xd, wd, None)
    dx0 = ops.conv_dgrad(g, dyd, wd)
    lhs, rhs = dot64(y0, dyd), dot64(xd, dx0)
    assert abs(lhs - rhs) <= 2e-5 * np.sqrt(dot64(y0, y0) * dot64(dyd, dyd)), (name, lhs, rhs)
    # the no-grad path's cached weight image (if this layer keeps one) gives the same bits
    u = ops.wino_weights(g, wd)
    if u is not None:
        assert torch.equal(ops.conv_dgrad(g, dyd, wd, bd, u=u), dx), name


def test_batch64_generate_against_oracle():
    """generate_audio.py:28-53 at BASELINE configs[4]'s size: 64 segments of 32512 samples, 8 kHz content, through
    generate() (model.inference at batch 64 + segment stitching, the bench's step).  Two of the 64 segments against the CPU
    oracle (HotPathRef.inference, float32 generator + float64 transform) on the same weights: the generator's spectrogram at
    1e-3 of its range (float32 yardstick), the decoded waveform against the oracle's decoder ON THE SAME spectrogram at
    3e-6 (K2 alone), and against the oracle's end-to-end waveform at 1e-3 of its peak."""
    from mdctgan_amd import options
    from mdctgan_amd.generate_audio import generate
    from mdctgan_amd.pix2pixHD_model import create_model
    from oracle import nets as onets
    from oracle import step as ostep
    from oracle import transform
    T, NSEG = 32512, 64
    gen = torch.Generator().manual_seed(77)
    netG = onets.init_weights(onets.build_generator("global", 2, 1, 64, 4, 9, input_size=(128, 256)), gen)
    hr = 0.05 * torch.randn(NSEG, T, generator=gen)
    spec = torch.fft.rfft(hr)
    spec[:, spec.shape[-1] // 6:] = 0                    # 8 kHz content of a 48 kHz clip
    lr = torch.fft.irfft(spec, n=T)
    opt = options.make_opt(*options.SPECTRAL_FLAGS, "--lr_sampling_rate", "8000", "--netG", "global", "--ngf", "64",
                           "--n_downsample_global", "4", "--n_blocks_global", "9", "--n_blocks_attn_g", "0", "--num_D", "2",
                           "--batchSize", str(NSEG), "--gpu_ids", "0")
    model = create_model(opt)
    model.netG.load_state_dict(netG.state_dict())
    lr_d = lr.to(DEV)
    wave = generate(model, lr_d, batch_size=NSEG, gen_overlap=0)
    assert wave.shape == (1, NSEG * T)
    model.eval()
    sr_spectro, sr_audio, _, _, _ = model.inference(lr_d)
    model.train()
    assert torch.equal(wave[0], sr_audio.reshape(-1)), "gen_overlap 0: stitching is concatenation (generate_audio.py:53)"
    pick = [3, 63]
    threads = torch.get_num_threads()
    torch.set_num_threads(min(16, threads))
    try:
        ref = ostep.HotPathRef(netG, None, ostep.CodecCfg(lr_rate=8000), num_D=2)
        o_spec, o_audio, o_norm, _ = ref.inference(lr[pick].numpy())
    finally:
        torch.set_num_threads(threads)
    got_spec = sr_spectro[pick].cpu().numpy()
    o_spec = o_spec.numpy()
    assert np.abs(got_spec - o_spec).max() <= 1e-3 * max(np.abs(o_spec).max(), 1e-3), np.abs(got_spec - o_spec).max()
    c = ostep.CodecCfg(lr_rate=8000)
    same = transform.to_audio(got_spec, o_norm, c.window, c.n_fft, c.hop, **c.codec)        # K2 alone
    got_wave = wave[0].reshape(NSEG, T)[pick].cpu().numpy()
    same = np.asarray(same).reshape(len(pick), T)
    assert np.abs(got_wave - same).max() <= 3e-6 * max(np.abs(same).max(), 1e-3) + 1e-7, np.abs(got_wave - same).max()
    o_audio = np.asarray(o_audio).reshape(len(pick), T)
    assert np.abs(got_wave - o_audio).max() <= 1e-3 * np.abs(o_audio).max(), np.abs(got_wave - o_audio).max()
