"""--fp16 (train.py:65-70, 161-164, 183-199) on the HIP path: MG_PRECISION_F16 convolutions against a float64
evaluation of the same autocast arithmetic (operands rounded to float16, exact products, wide accumulation, outputs
rounded through float16), the device-side GradScaler against torch's rules, and the whole AMP step against the
reference's captured losses (fixture G9) and the oracle's CPU-autocast step.
Tolerances: a float16-rounded output may land one float16 ulp (2^-10 relative) from the float64 evaluation's rounding;
weight gradients stay float32 (3e-5 of max, as in test_conv_gpu.py); losses rtol 2e-2."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import nets as onets
from oracle import step as ostep

pytestmark = pytest.mark.gpu
DEV = "cuda"

# (name, B, Ci, H, W, Co, k, stride, pad, reflect)
CASES = [
    ("res3x3_reflect", 2, 32, 8, 16, 32, 3, 1, 1, True),
    ("stem7x7_reflect_ci2", 2, 2, 32, 64, 16, 7, 1, 3, True),
    ("head7x7_reflect_co1", 2, 16, 32, 64, 1, 7, 1, 3, True),
    ("down3x3_s2", 2, 16, 32, 64, 32, 3, 2, 1, False),
    ("d4x4_s2_ci3", 2, 3, 32, 64, 16, 4, 2, 2, False),
    ("d4x4_s1", 2, 32, 5, 9, 64, 4, 1, 2, False),
    ("d4x4_s1_co1", 2, 64, 6, 10, 1, 4, 1, 2, False),
    ("wide_128tile", 2, 64, 16, 32, 256, 3, 1, 1, True),
    ("bottleneck_like", 8, 128, 8, 16, 128, 3, 1, 1, True),
    ("bottleneck_wino_f16", 8, 256, 8, 16, 256, 3, 1, 1, True),      # wide + 256 tiles: Winograd with f16 GEMMs
    ("wino_f16_zero_pad", 4, 256, 16, 16, 256, 3, 1, 1, False),
    ("conv1x1", 2, 64, 4, 8, 48, 1, 1, 0, False),
    ("smallc_4x4_s2", 2, 3, 33, 65, 64, 4, 2, 2, False),         # csrc/conv_smallc.h with float16-rounded operands
    ("smallc_7x7_reflect", 2, 2, 20, 36, 32, 7, 1, 3, True),
    # weight-dominated layers (pixels <= Co, channels % 64 == 0): float16 im2col GEMMs with LDS-DMA staging (csrc/conv_h16.h)
    ("h16_trunk_like", 2, 128, 4, 8, 128, 3, 1, 1, True),
    ("h16_1x1", 4, 128, 4, 8, 192, 1, 1, 0, False),
    ("h16_5x5_zero_pad_ragged", 1, 64, 6, 10, 64, 5, 1, 2, False),       # 60 pixels: the K padding of the weight gradient
    ("h16_4x4_pad2", 1, 64, 5, 9, 128, 4, 1, 2, False),                   # output larger than the input (6 x 10)
    # activation-dominated layers with channels % 64 == 0: float16 implicit GEMMs with LDS-DMA staging and transpose reads
    # (csrc/conv_dma.h, HALF instances) over float16 copies of the activations
    ("cdh_down3x3_s2", 2, 64, 32, 64, 128, 3, 2, 1, False),
    ("cdh_d4x4_s2_odd", 2, 64, 33, 65, 128, 4, 2, 2, False),              # 17 x 33 outputs: pixel tails everywhere
    ("cdh_3x3_zero_pad", 4, 128, 12, 20, 64, 3, 1, 1, False),
    ("cdh_5x5_ragged", 3, 64, 9, 13, 64, 5, 1, 2, False),                 # 351 pixels: the weight gradient's K tail
    ("cdh_res3x3_reflect_64", 2, 64, 24, 40, 64, 3, 1, 1, True),         # local-enhancer residual block
    ("cdh_up_twin_256_128", 2, 256, 16, 24, 128, 3, 2, 1, False),        # the ladder's 256 -> 128 pair
    # single-output-channel layers as tap GEMMs on float16-rounded operands (csrc/conv_co1.h)
    ("co1_head7x7_reflect_64", 2, 64, 20, 36, 1, 7, 1, 3, True),
    ("co1_dlast_4x4_512", 2, 512, 6, 10, 1, 4, 1, 2, False),
]


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


def h(t):
    """Tensor.half() rounding, kept in float64."""
    return t.float().half().double()


def close_f16(got, want, ulps=1.5, floor=2.0 ** -11):
    """got: float32 values that went through float16; want: float64 before rounding.  1.5 float16 ulps of the element
    plus half an ulp of the largest element: the float32 accumulation error scales with the terms, not with a
    cancelling result, and a reflection-padded data gradient sums its aliased dy taps BEFORE the float16 rounding
    (autocast rounds the padded-domain gradient first and folds afterwards -- one rounding either way)."""
    got = got.double().cpu()
    tol = ulps * 2.0 ** -10 * want.abs() + floor * want.abs().max()
    bad = (got - want).abs() > tol
    assert not bad.any(), ((got - want).abs().max().item(), want.abs().max().item(), int(bad.sum()))
    assert torch.equal(got.float().half().float().double(), got), "output is not float16-representable"


@pytest.mark.parametrize("case", [("row_pieces_OW128", 2, 64, 8, 256, 128, 3), ("one_row_OW64_4x4", 1, 128, 8, 128, 64, 4),
                                  ("two_rows_OW32", 2, 64, 32, 64, 128, 3), ("whole_image_OW8", 4, 64, 16, 16, 64, 3),
                                  ("stride1_4x4_pad2_OW64", 2, 64, 15, 63, 64, 4, 1, 2), ("stride1_3x3_OW64", 2, 64, 5, 64, 64, 3, 1, 1),
                                  ("stride1_1x1_OW128", 1, 128, 4, 128, 64, 1, 1, 0)],
                         ids=lambda c: c[0])
def test_f16_wgrad_row_regular_gather_is_the_general_gather(case, monkeypatch):
    """The float16 instances of the LDS-DMA weight gradient (64-pixel chunks) with the row-regular gather (csrc/conv_dma.h,
    RR = true) against the per-lane coordinate walk (MG_NO_WGRAD_RR=1): the same bits; accuracy is test_conv_f16_precision's
    (its cdh_down3x3_s2 case takes the row-regular kernel)."""
    from mdctgan_amd import _lib, ops
    name, B, Ci, H, W, Co, k = case[:7]
    stride, pad = case[7:] if len(case) > 7 else (2, 1)           # the discriminator's last layers are 4x4 stride 1 pad 2
    g = ops.conv_geom(B, H, W, Ci, Co, k, k, stride, pad, False, _lib.PRECISION_F16)
    assert "conv_wgrad_dma_kernel" in ops.plan_name(2, g) and ops.plan_name(2, g).endswith(", true>")
    assert (B * g.OH * g.OW) % 64 == 0 and (g.OW % 64 == 0 or (64 % g.OW == 0 and (g.OH * g.OW) % 64 == 0))
    gen = torch.Generator().manual_seed(len(name))
    x = torch.randn(B, H, W, Ci, generator=gen).to(DEV)
    gy = torch.randn(B, g.OH, g.OW, Co, generator=gen).to(DEV)
    out = []
    for env in (None, "1"):
        if env:
            monkeypatch.setenv("MG_NO_WGRAD_RR", env)
        dw = torch.full((Co, k, k, Ci), 3.0, dtype=torch.float32, device=DEV)
        ops.conv_wgrad(g, x, gy, dw, None)
        out.append(dw)
    assert torch.equal(out[0], out[1])
    ref = torch.zeros(Co, Ci, k, k, dtype=torch.float64, requires_grad=True)
    F.conv2d(h(x.cpu().permute(0, 3, 1, 2)), ref, None, stride=stride, padding=pad).backward(h(gy.cpu().permute(0, 3, 1, 2)))
    want = nhwc(ref.grad)
    assert (out[0].double().cpu() - want).abs().max().item() <= 3e-5 * want.abs().max().item()


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_conv_f16_precision(case):
    from mdctgan_amd import _lib, ops
    name, B, Ci, H, W, Co, k, s, p, reflect = case
    gen = torch.Generator().manual_seed(len(name) * 7 + Ci)
    x = torch.randn(B, Ci, H, W, generator=gen, dtype=torch.float64)
    w = torch.randn(Co, Ci, k, k, generator=gen, dtype=torch.float64) / np.sqrt(Ci * k * k)
    b = torch.randn(Co, generator=gen, dtype=torch.float64)
    xh, wh = h(x).requires_grad_(), h(w).requires_grad_()
    xp = F.pad(xh, (p, p, p, p), mode="reflect") if (reflect and p) else xh
    y = F.conv2d(xp, wh, b, stride=s, padding=0 if (reflect and p) else p)
    gy = torch.randn(y.shape, generator=gen, dtype=torch.float64)
    y.backward(h(gy))

    g = ops.conv_geom(B, H, W, Ci, Co, k, k, s, p, reflect, _lib.PRECISION_F16)
    xd, wd, bd = nhwc(x).float().to(DEV), nhwc(w).float().to(DEV), b.float().to(DEV)
    gyd = nhwc(gy).float().to(DEV)
    # 3x3 stride-1 layers run as Winograd F(2x2,3x3) with float16 GEMM operands (the transformed tiles V = B^T d B and
    # U = G g G^T are what gets rounded, as in cuDNN's float16 Winograd algorithms that cudnn.benchmark may pick for the
    # reference): the rounding error of 16 transformed products folds into each output -> a few float16 ulps.
    h16 = ops.plan_name(0, g).startswith("hgemm")
    assert h16 == name.startswith("h16_"), ops.plan_name(0, g)
    cdh = "dma_kernel" in ops.plan_name(0, g)
    if name.startswith("cdh_"):
        assert ops.plan_name(0, g).startswith("conv_fwd_dma_kernel") and ", true" in ops.plan_name(0, g)
        assert ops.plan_name(2, g).startswith("conv_wgrad_dma_kernel") and ", true" in ops.plan_name(2, g)
        assert ops.plan_name(1, g).startswith("conv_dgrad_dma_kernel"), ops.plan_name(1, g)      # reflect: padded domain + fold
    co1 = Co == 1 and ops.plan_name(0, g).startswith("dgemm32g_kernel<64, 128")
    assert co1 == (name.startswith("co1_") or name == "d4x4_s1_co1"), ops.plan_name(0, g)
    wino = bool(_lib.load().mg_conv_wino_weights_bytes(g)) and not h16 and not cdh and not co1
    tol = dict(ulps=4.0, floor=2.0 ** -9) if wino else {}
    close_f16(ops.conv_fwd(g, xd, wd, bd), nhwc(y.detach()), **tol)
    close_f16(ops.conv_dgrad(g, gyd, wd), nhwc(xh.grad), **tol)
    if h16 or cdh:      # the caller-held float16 weight copy (one cast per step, shared by forward and data gradient): same bits
        u = ops.wino_weights(g, wd)
        assert u is not None and u.numel() * 4 == Co * k * k * Ci * 2
        assert torch.equal(ops.conv_fwd(g, xd, wd, bd, u=u), ops.conv_fwd(g, xd, wd, bd))
        assert torch.equal(ops.conv_dgrad(g, gyd, wd, u=u), ops.conv_dgrad(g, gyd, wd))
        close_f16(ops.conv_fwd(g, xd, wd, bd, act=ops.ACT_RELU), torch.relu(nhwc(y.detach())))
    dw = torch.full((Co, k, k, Ci), 7.0, dtype=torch.float32, device=DEV)
    ops.conv_wgrad(g, xd, gyd, dw, None)
    want = nhwc(wh.grad)
    wtol = 2e-3 if wino else 3e-5       # Winograd: float16-rounded transformed operands in the weight gradient too
    assert (dw.double().cpu() - want).abs().max().item() <= wtol * want.abs().max().item()
    ops.conv_wgrad(g, xd, gyd, dw, None, accumulate=True)
    assert (dw.double().cpu() - 2 * want).abs().max().item() <= 2 * wtol * want.abs().max().item()
    if wino:
        # a float16 Winograd layer that ALSO satisfies the float16 implicit-GEMM predicates must be sized
        # as a Winograd layer (float32 U / V / Md images) by the buffer-size queries, because the pass entry points take
        # the Winograd branch first.  Caller-held u / v / md through every pass: same bits as the self-contained calls.
        T = B * (H // 2) * (W // 2)
        assert ops.wino_weights_bytes(g) == 16 * Co * Ci * 4, ops.wino_weights_bytes(g)
        u = ops.wino_weights(g, wd)
        v, md = ops.wino_tile_buffers(g, xd.device)
        assert v is not None and v.numel() == 16 * T * Ci and md is not None and md.numel() == 16 * T * Co
        guard = torch.full((4096,), 3.25, device=DEV)      # allocated right behind the images: an overrun would land here
        assert torch.equal(ops.conv_fwd(g, xd, wd, bd, u=u, v_out=v), ops.conv_fwd(g, xd, wd, bd))
        assert torch.equal(ops.conv_dgrad(g, gyd, wd, u=u, md_out=md), ops.conv_dgrad(g, gyd, wd))
        dw1, dw2 = torch.empty_like(dw), torch.empty_like(dw)
        ops.conv_wgrad(g, xd, gyd, dw1, None)
        ops.conv_wgrad(g, xd, gyd, dw2, None, v=v, md=md)
        assert torch.equal(dw1, dw2)
        assert bool((guard == 3.25).all())
    if cdh or co1:   # staged copies of x / dy handed from the forward / data-gradient call to the weight gradient: same bits
        u = ops.wino_weights(g, wd)
        v, md = ops.wino_tile_buffers(g, xd.device)
        assert v is not None and md is not None
        assert torch.equal(ops.conv_fwd(g, xd, wd, bd, u=u, v_out=v), ops.conv_fwd(g, xd, wd, bd))
        if md is not None:
            assert torch.equal(ops.conv_dgrad(g, gyd, wd, u=u, md_out=md), ops.conv_dgrad(g, gyd, wd))
        dw1, dw2 = torch.empty_like(dw), torch.empty_like(dw)
        ops.conv_wgrad(g, xd, gyd, dw1, None)
        ops.conv_wgrad(g, xd, gyd, dw2, None, v=v, md=md)
        assert torch.equal(dw1, dw2)


def test_conv_f16_overflow_becomes_inf():
    """A forward / data-gradient output beyond 65504 rounds to inf, as a float16 tensor would."""
    from mdctgan_amd import _lib, ops
    g = ops.conv_geom(1, 4, 4, 16, 16, 1, 1, 1, 0, False, _lib.PRECISION_F16)
    x = torch.full((1, 4, 4, 16), 300.0, device=DEV)
    w = torch.full((16, 1, 1, 16), 20.0, device=DEV)
    assert torch.isinf(ops.conv_fwd(g, x, w, None)).all()           # 16 * 300 * 20 = 96000
    assert torch.isinf(ops.conv_dgrad(g, x, w)).all()
    dw = torch.empty(16, 1, 1, 16, device=DEV)
    ops.conv_wgrad(g, x, x, dw, None)
    assert torch.isfinite(dw).all() and dw[0, 0, 0, 0].item() == 16 * 300.0 * 300.0   # weight gradients stay float32


def test_grad_scaler_kernels():
    """mg_scaler_check / mg_adam_*_amp / mg_scaler_update == GradScaler.step / update semantics."""
    from mdctgan_amd import amp
    from mdctgan_amd.optim import FusedAdam
    torch.manual_seed(0)
    p = torch.nn.Parameter(torch.randn(1000, device=DEV))
    q = torch.nn.Parameter(torch.randn(1000, device=DEV))
    opt_p, opt_q = FusedAdam([p], lr=1e-2, betas=(0.5, 0.999)), FusedAdam([q], lr=1e-2, betas=(0.5, 0.999))
    ref_p = p.detach().clone().cpu().requires_grad_()
    ref_opt = torch.optim.Adam([ref_p], lr=1e-2, betas=(0.5, 0.999))
    sc = amp.GradScaler(init_scale=1024.0, growth_interval=2)
    for it in range(5):
        gp, gq = torch.randn(1000), torch.randn(1000)
        overflow = it == 1
        opt_p.zero_grad(); opt_q.zero_grad()
        p.grad.copy_((gp * sc.get_scale()).to(DEV)); p._mg_fresh = False
        q.grad.copy_((gq * sc.get_scale()).to(DEV)); q._mg_fresh = False
        if overflow:
            q.grad[17] = float("inf")
        scale_before = sc.get_scale()
        q_before = q.detach().clone()
        sc.step(opt_p)
        sc.step(opt_q)
        sc.update()
        ref_opt.zero_grad(); ref_p.grad = gp.clone(); ref_opt.step()      # p never overflows: steps every iteration
        assert (p.detach().cpu() - ref_p.detach()).abs().max().item() < 1e-6
        if overflow:
            assert torch.equal(q.detach(), q_before), "step must be skipped on inf gradients"
            assert sc.get_scale() == scale_before * 0.5
        else:
            assert not torch.equal(q.detach(), q_before)
    # iterations 0 (ok) 1 (inf: x0.5, tracker 0) 2, 3 (ok, ok: x2 at interval 2) 4 (ok)
    assert sc.get_scale() == 1024.0 and int(sc.state[1].item()) == 1
    assert sc.state_dict()["scale"] == 1024.0


def make_fp16_model():
    from mdctgan_amd import options
    from mdctgan_amd.pix2pixHD_model import create_model
    opt = options.make_opt(*options.SPECTRAL_FLAGS, "--lr_sampling_rate", "12000", "--netG", "global", "--ngf", "4",
                           "--n_blocks_global", "2", "--n_blocks_attn_g", "0", "--num_D", "2", "--ndf", "8",
                           "--batchSize", "2", "--bins", "32", "--segment_length", "7936", "--gpu_ids", "0", "--fp16")
    model = create_model(opt)
    onets.fill_deterministic(model.netG)
    onets.fill_deterministic(model.netD)
    return model


def oracle_model():
    netG = onets.fill_deterministic(onets.build_generator("global", 2, 1, 4, 4, 2, input_size=(32, 256)))
    netD = onets.fill_deterministic(onets.MultiscaleDRef(3, ndf=8, n_layers=3, num_D=2))
    return ostep.HotPathRef(netG, netD, ostep.CodecCfg(), num_D=2)


def test_fp16_step_against_reference_and_oracle(golden):
    g = golden("g9_step_global_fp16")
    lr, hr = torch.from_numpy(g["lr"]).to(DEV), torch.from_numpy(g["hr"]).to(DEV)
    model = make_fp16_model()
    assert model.fp16 and model.scaler is not None and model.scaler.get_scale() == 65536.0
    # 1) forward losses under autocast vs the reference's (captured under torch.autocast(float16))
    want = dict(zip(g["loss_names"], g["losses"]))
    ld = model.optimize_parameters(lr, hr)
    for k in model.loss_names:
        assert abs(ld[k].item() - want[k]) <= 2e-2 * abs(want[k]) + 1e-3, (k, ld[k].item(), want[k])
    # 2) an overflowing scale: every update is skipped, the scale halves once per iteration (train.py:199)
    model2 = make_fp16_model()
    model2.scaler.state[0] = 2.0 ** 60
    before = {k: v.detach().clone() for k, v in model2.netG.state_dict().items()}
    model2.optimize_parameters(lr, hr)
    assert model2.scaler.get_scale() == 2.0 ** 59
    for k, v in model2.netG.state_dict().items():
        assert torch.equal(v, before[k]), k
    assert model2.optimizer_G.state[0].item() == 0        # Adam's step counter did not advance
    # 3) a scale at which the reference's float16 gradients stay finite (256: the oracle's scale stops halving there):
    #    three AMP iterations of the HIP path beside the oracle's CPU-autocast iterations.  The first Adam step moves
    #    every weight by +-lr whatever the gradient's size, so float16-noise sign flips on this 1000x-amplifying toy
    #    net separate the trajectories by a few percent per update: 2e-2 before any update, 1e-1 after one, 2e-1 after two.
    model3, ref = make_fp16_model(), oracle_model()
    model3.scaler.state[0] = 256.0
    rs = torch.amp.GradScaler("cpu", init_scale=256.0)
    for it in range(3):
        lh = model3.optimize_parameters(lr, hr)
        lo = ref.train_step(g["lr"], g["hr"], amp=True, scaler=rs)
        rtol = (2e-2, 1e-1, 2e-1)[it]         # (measured 1.1e-1 after two updates once K1's codec changed by a few ulp)
        for k in model3.loss_names:
            assert abs(lh[k].item() - lo[k]) <= rtol * abs(lo[k]) + 2e-3, (it, k, lh[k].item(), lo[k])
    assert model3.scaler.get_scale() == rs.get_scale() == 256.0
    assert model3.optimizer_G.state[0].item() == 3 and model3.optimizer_D.state[0].item() == 3
    # the float16 shadow of the parameters is what the Adam kernel wrote: bit-equal to a fresh cast after real updates,
    # and untouched by the skipped ones of model2
    for m_ in (model3, model2):
        for opt_ in (m_.optimizer_G, m_.optimizer_D):
            assert opt_.flat_h is not None and torch.equal(opt_.flat_h, opt_.flat_p.to(torch.float16))
    # a parameter rewritten by a torch op (checkpoint load, re-initialisation): the next step re-syncs its slice
    w0 = next(p_ for p_ in model3.netG.parameters() if p_.dim() == 4)
    with torch.no_grad():
        w0.mul_(0.5)
    assert not torch.equal(w0._mg_h, w0._mg_flat.to(torch.float16))
    model3.optimize_parameters(lr, hr)
    assert torch.equal(model3.optimizer_G.flat_h, model3.optimizer_G.flat_p.to(torch.float16))
    # inference is float32 (generate_audio.py has no autocast)
    sr_spectro, sr_audio, *_ = model3.inference(lr)
    assert torch.isfinite(sr_audio).all() and sr_spectro.dtype == torch.float32


def test_short_reduction_wgrad_checks_its_own_results():
    """The weight gradient of a small-spatial layer (256 pixels = the K of its GEMM) runs A-stationary (hgemm_as_kernel,
    csrc/dense_gemm_h.h) and does the GradScaler's inf / nan check on its accumulators (mg_conv_wgrad_chk)."""
    from mdctgan_amd import _lib, ops
    B, H, W, C = 8, 4, 8, 256
    g = ops.conv_geom(B, H, W, C, C, 3, 3, 1, 1, True, _lib.PRECISION_F16)
    assert ops.wgrad_checks_finite(g)
    assert not ops.wgrad_checks_finite(ops.conv_geom(B, H, W, C, C, 3, 3, 1, 1, True, _lib.PRECISION_F32))
    gen = torch.Generator().manual_seed(3)
    x = torch.randn(B, C, H, W, generator=gen)
    dy = torch.randn(B, C, H, W, generator=gen)
    xp = F.pad(h(x), (1, 1, 1, 1), mode="reflect")
    want = torch.nn.grad.conv2d_weight(xp, (C, C, 3, 3), h(dy))                 # float64, float16-rounded operands
    want = want.permute(0, 2, 3, 1).contiguous()                                # OHWI
    xd, dyd = nhwc(x).to(DEV), nhwc(dy).to(DEV)
    dw = torch.full((C, 3, 3, C), float("nan"), device=DEV)
    flag = torch.zeros(1, device=DEV)
    ops.conv_wgrad(g, xd, dyd, dw, None, found_inf=flag)
    assert flag.item() == 0.0
    err = (dw.double().cpu() - want).abs().max().item()
    assert err <= 3e-5 * want.abs().max().item(), err
    first = dw.clone()
    ops.conv_wgrad(g, xd, dyd, dw, None, accumulate=True, found_inf=flag)       # C += result, checked after the addition
    assert torch.equal(dw, first + first) and flag.item() == 0.0
    bad = dyd.clone()
    bad[3, 2, 5, 77] = float("inf")
    ops.conv_wgrad(g, xd, bad, dw, None, found_inf=flag)
    assert flag.item() == 1.0 and not torch.isfinite(dw).all()
    flag.zero_()
    dw.copy_(first)
    dw[200, 1, 1, 9] = float("nan")                                            # a non-finite value already in the buffer
    ops.conv_wgrad(g, xd, dyd, dw, None, accumulate=True, found_inf=flag)
    assert flag.item() == 1.0
    # no flag: plain mg_conv_wgrad_w
    ops.conv_wgrad(g, xd, dyd, dw, None)
    assert torch.equal(dw, first)
    # a geometry whose kernel has no check refuses a flag instead of ignoring it
    g2 = ops.conv_geom(2, 32, 64, 16, 32, 3, 3, 2, 1, False, _lib.PRECISION_F16)
    with pytest.raises(RuntimeError):
        ops.conv_wgrad(g2, torch.zeros(2, 32, 64, 16, device=DEV), torch.zeros(2, 16, 32, 32, device=DEV),
                       torch.zeros(32, 3, 3, 16, device=DEV), None, found_inf=flag)


def test_producer_side_inf_check_changes_nothing(monkeypatch):
    """FusedAdam leaves gradients whose own kernel checked them out of its mg_scaler_check pass: same weights, same scale and
    same skipped steps as with MG_NO_PRODUCER_INF_CHECK=1 (everything through the arena pass), overflow iteration included."""
    from mdctgan_amd import options
    from mdctgan_amd.pix2pixHD_model import create_model

    def run(off):
        monkeypatch.setenv("MG_NO_PRODUCER_INF_CHECK", "1" if off else "0")
        torch.manual_seed(5)
        opt = options.make_opt(*options.SPECTRAL_FLAGS, "--lr_sampling_rate", "12000", "--netG", "global", "--ngf", "16",
                               "--n_blocks_global", "2", "--n_blocks_attn_g", "0", "--num_D", "2", "--ndf", "8",
                               "--batchSize", "2", "--bins", "32", "--segment_length", "7936", "--gpu_ids", "0", "--fp16")
        model = create_model(opt)
        onets.fill_deterministic(model.netG)
        onets.fill_deterministic(model.netD)
        gen = torch.Generator().manual_seed(11)
        hr = (0.1 * torch.randn(2, 7936, generator=gen)).to(DEV)
        lr = (0.1 * torch.randn(2, 7936, generator=gen)).to(DEV)
        model.scaler.state[0] = 64.0
        checked, scales = 0, []
        for it in range(4):
            if it == 2:
                model.scaler.state[0] = 2.0 ** 40          # this iteration overflows: both optimisers skip, the scale halves
            model.optimize_parameters(lr, hr)
            checked = max(checked, sum(bool(getattr(p, "_mg_inf_checked", False)) for p in model.netG.parameters()))
            scales.append(model.scaler.get_scale())
            if it == 2:
                model.scaler.state[0] = 64.0
        return ({k: v.detach().clone() for k, v in model.netG.state_dict().items()}, scales, checked,
                model.optimizer_G.state[0].item())
    w_on, s_on, n_on, steps_on = run(False)
    w_off, s_off, n_off, steps_off = run(True)
    assert n_on > 0 and n_off == 0, (n_on, n_off)           # the trunk's weight gradients took the producer-side check
    assert s_on == s_off and s_on[2] == 2.0 ** 39 and steps_on == steps_off == 3
    for k in w_on:
        assert torch.equal(w_on[k], w_off[k]), k


def test_norm_kernels_write_the_float16_copy():
    """mg_instnorm_fwd_h / mg_instnorm_bwd_h: the float32 results are those of the plain calls, the float16 output is their
    round-to-nearest cast -- on the slab kernels (small maps) and on the partial / finalize / apply sequence (large maps)."""
    from mdctgan_amd import _lib, ops
    gen = torch.Generator().manual_seed(9)
    for (B, H, W, C), act, with_res in (((2, 8, 16, 64), _lib.ACT_RELU, True), ((2, 40, 72, 128), _lib.ACT_LRELU02, False),
                                        ((3, 17, 33, 64), _lib.ACT_NONE, True)):
        x = torch.randn(B, H, W, C, generator=gen).to(DEV)
        res = torch.randn(B, H, W, C, generator=gen).to(DEV) if with_res else None
        gy = torch.randn(B, H, W, C, generator=gen).to(DEV)
        y0, m0, r0 = ops.instnorm_fwd(x, act, res)
        y16 = torch.full((x.numel(),), float("nan"), dtype=torch.float16, device=DEV)
        y1, m1, r1 = ops.instnorm_fwd(x, act, res, y16=y16)
        assert torch.equal(y0, y1) and torch.equal(m0, m1) and torch.equal(r0, r1)
        assert torch.equal(y16.view_as(y1), y1.half())
        d0 = ops.instnorm_bwd(gy, x, m0, r0, act)
        d16 = torch.full((x.numel(),), float("nan"), dtype=torch.float16, device=DEV)
        d1 = ops.instnorm_bwd(gy, x, m0, r0, act, dx16=d16)
        assert torch.equal(d0, d1) and torch.equal(d16.view_as(d1), d1.half())


def test_filled_tiles_skip_the_cast_and_change_nothing():
    """MG_TILES_V_FILLED / MG_TILES_MD_FILLED on the float16 implicit GEMMs: forward / data gradient / weight gradient from a
    caller-made float16 copy == the calls that cast for themselves; a wrong copy shows (i.e. the flag really skips the cast)."""
    from mdctgan_amd import _lib, ops
    gen = torch.Generator().manual_seed(4)
    for (B, Ci, H, W, Co, k, s, p, refl) in ((2, 64, 32, 64, 128, 3, 2, 1, False), (2, 64, 24, 40, 64, 3, 1, 1, True)):
        g = ops.conv_geom(B, H, W, Ci, Co, k, k, s, p, refl, _lib.PRECISION_F16)
        assert ops.precast_ok(0, g) and ops.precast_ok(1, g), (ops.plan_name(0, g), ops.plan_name(1, g))
        x = torch.randn(B, H, W, Ci, generator=gen).to(DEV)
        w = (0.05 * torch.randn(Co, k, k, Ci, generator=gen)).to(DEV)
        dy = torch.randn(B, g.OH, g.OW, Co, generator=gen).to(DEV)
        x16, dy16 = x.half().reshape(-1), dy.half().reshape(-1)
        assert torch.equal(ops.conv_fwd(g, x, w, None, v_out=x16, v_filled=True), ops.conv_fwd(g, x, w, None))
        assert torch.equal(ops.conv_dgrad(g, dy, w, md_out=dy16, md_filled=True), ops.conv_dgrad(g, dy, w))
        dw0, dw1 = torch.empty_like(w), torch.empty_like(w)
        ops.conv_wgrad(g, x, dy, dw0, None)
        ops.conv_wgrad(g, x, dy, dw1, None, v=x16, md=dy16)
        assert torch.equal(dw0, dw1)
        assert not torch.equal(ops.conv_fwd(g, x, w, None, v_out=torch.zeros_like(x16), v_filled=True), ops.conv_fwd(g, x, w, None))
    # a float32 geometry has no float16 output on the fused convolution + norm call
    g32 = ops.conv_geom(2, 8, 16, 64, 64, 3, 3, 1, 1, True, _lib.PRECISION_F32)
    with pytest.raises(RuntimeError):
        ops.conv_fwd_instnorm(g32, torch.zeros(2, 8, 16, 64, device=DEV), torch.zeros(64, 3, 3, 64, device=DEV),
                              y16=torch.zeros(2 * 8 * 16 * 64, dtype=torch.float16, device=DEV))


def test_producer_written_float16_copies_change_nothing(monkeypatch):
    """The AMP step with the norm kernels writing the next convolution's float16 operand (functional._attach_h16 / _h16_of) ==
    the step with MG_NO_H16_PRODUCER=1, bit for bit, and the hand-over really happens in both directions."""
    from mdctgan_amd import functional as Fh
    from mdctgan_amd import options
    from mdctgan_amd.pix2pixHD_model import create_model

    def run(off):
        monkeypatch.setenv("MG_NO_H16_PRODUCER", "1" if off else "0")
        for k in Fh.H16_STATS:
            Fh.H16_STATS[k] = 0
        torch.manual_seed(5)
        opt = options.make_opt(*options.SPECTRAL_FLAGS, "--lr_sampling_rate", "12000", "--netG", "local", "--ngf", "32",
                               "--n_downsample_global", "2", "--n_blocks_global", "2", "--n_blocks_local", "2",
                               "--n_blocks_attn_g", "0", "--num_D", "2", "--ndf", "64", "--batchSize", "2", "--bins", "64",
                               "--segment_length", "16128", "--gpu_ids", "0", "--fp16")
        model = create_model(opt)
        onets.fill_deterministic(model.netG)
        onets.fill_deterministic(model.netD)
        gen = torch.Generator().manual_seed(11)
        hr = (0.1 * torch.randn(2, 16128, generator=gen)).to(DEV)
        lr = (0.1 * torch.randn(2, 16128, generator=gen)).to(DEV)
        model.scaler.state[0] = 64.0
        losses = []
        for it in range(3):
            ld = model.optimize_parameters(lr, hr)
            losses.append({k: v.item() for k, v in ld.items()})
        return ({k: v.detach().clone() for k, v in model.netG.state_dict().items()},
                {k: v.detach().clone() for k, v in model.netD.state_dict().items()}, losses, dict(Fh.H16_STATS))
    g_on, d_on, l_on, st_on = run(False)
    g_off, d_off, l_off, st_off = run(True)
    assert st_on["made"] > 0 and st_on["used_fwd"] > 0 and st_on["used_bwd"] > 0, st_on
    assert st_off == {"made": 0, "used_fwd": 0, "used_bwd": 0}, st_off
    assert l_on == l_off
    for a, b in ((g_on, g_off), (d_on, d_off)):
        for k in a:
            assert torch.equal(a[k], b[k]), k


def test_weight_gradient_stored_as_float16_is_the_rounded_float32_one():
    """mg_conv_wgrad_h16 (round 6): the trunk's short-reduction weight-gradient GEMM storing float16 -- the dtype an autocast
    layer's weight gradient has in the reference (train.py:161-164) -- writes exactly half(what mg_conv_wgrad_chk writes), accumulates
    in float16 storage with a float32 add, and flags a result that is not a finite float16 (|v| >= 65520), which is also the
    criterion of the float32-storing instance now."""
    from mdctgan_amd import _lib, ops
    B, H, W, C = 8, 4, 8, 256
    g = ops.conv_geom(B, H, W, C, C, 3, 3, 1, 1, True, _lib.PRECISION_F16)
    assert ops.wgrad_h16_ok(g) and not ops.wgrad_h16_ok(ops.conv_geom(B, H, W, C, C, 3, 3, 1, 1, True, _lib.PRECISION_F32))
    gen = torch.Generator().manual_seed(3)
    xd = nhwc(torch.randn(B, C, H, W, generator=gen)).to(DEV)
    dyd = nhwc(torch.randn(B, C, H, W, generator=gen)).to(DEV)
    dw = torch.empty(C, 3, 3, C, device=DEV)
    flag = torch.zeros(1, device=DEV)
    ops.conv_wgrad(g, xd, dyd, dw, None, found_inf=flag)
    dw16 = torch.full((C, 3, 3, C), float("nan"), dtype=torch.float16, device=DEV)
    ops.conv_wgrad_h16(g, xd, dyd, dw16, False, found_inf=flag)
    assert flag.item() == 0.0 and torch.equal(dw16, dw.half())
    ops.conv_wgrad_h16(g, xd, dyd, dw16, True, found_inf=flag)                  # += in float32, rounded once
    assert flag.item() == 0.0 and torch.equal(dw16, (dw.half().float() + dw).half())
    # a result beyond float16's range: inf in the float16 store, flagged by BOTH instances (the float32 one keeps the finite value)
    big = dyd * 4096.0
    ops.conv_wgrad(g, xd, big, dw, None, found_inf=flag)
    assert torch.isfinite(dw).all() and dw.abs().max().item() > 65520.0 and flag.item() == 1.0
    flag.zero_()
    ops.conv_wgrad_h16(g, xd, big, dw16, False, found_inf=flag)
    assert flag.item() == 1.0 and torch.equal(dw16, dw.half()) and torch.isinf(dw16).any()
    # a geometry the float16-storing kernel does not take is refused
    g2 = ops.conv_geom(2, 32, 64, 64, 64, 3, 3, 2, 1, False, _lib.PRECISION_F16)
    with pytest.raises(RuntimeError):
        ops.conv_wgrad_h16(g2, torch.zeros(2, 32, 64, 64, device=DEV), torch.zeros(2, 16, 32, 64, device=DEV),
                           torch.zeros(64, 3, 3, 64, dtype=torch.float16, device=DEV), False)


def test_segmented_fp16_optimizer_passes():
    """FusedAdam(half_shadow=True) (--fp16, round 6): one segmented launch per pass (mg_scaler_check_segs / mg_adam_step_segs) in
    which every run of the arena carries its gradient the way the reference's autocast does -- convolution parameters as float16
    VALUES (float32 storage rounded through float16 where consumed, or float16 storage written by the weight-gradient kernel),
    BatchNorm / position-embedding parameters as float32 -- against torch.optim.Adam on those very gradients; the overflow
    criterion follows: 70 000 is inf for an autocast gradient and a number for a float32 one; a skipped step touches nothing."""
    from mdctgan_amd import _lib, amp
    from mdctgan_amd.optim import FusedAdam
    torch.manual_seed(0)
    shapes = [(64, 3, 3, 32), (64,), (40,), (128, 1, 1, 64), (24,), (16, 8)]
    ps = [torch.nn.Parameter(torch.randn(*sh, device=DEV)) for sh in shapes]
    ps[2]._mg_grad_f32 = True          # a BatchNorm weight
    ps[5]._mg_grad_f32 = True          # a position embedding
    ps[3]._mg_g16_ok = True            # a weight whose gradient kernel stores float16
    opt = FusedAdam(ps, lr=1e-2, betas=(0.5, 0.999), half_shadow=True)
    refs = [p.detach().clone().cpu().requires_grad_() for p in ps]
    ref_opt = torch.optim.Adam(refs, lr=1e-2, betas=(0.5, 0.999))
    sc = amp.GradScaler(init_scale=256.0, growth_interval=1000)
    opt.zero_grad()
    assert opt._modes == [_lib.GRAD_AUTOCAST, _lib.GRAD_AUTOCAST, _lib.GRAD_F32, _lib.GRAD_F16, _lib.GRAD_AUTOCAST, _lib.GRAD_F32]
    assert ps[3]._mg_g16 is not None and ps[3]._mg_g16.dtype == torch.float16 and ps[0]._mg_g16 is None
    gen = torch.Generator().manual_seed(1)
    for it in range(5):
        opt.zero_grad()
        scale = sc.get_scale()
        overflow_autocast, big_f32 = it == 1, it == 3
        grads = [torch.randn(*sh, generator=gen) * 3.0 for sh in shapes]
        if big_f32:
            grads[2][7] = 70000.0 / scale * 1.0            # x scale = 70 000: finite as a float32 gradient
        scaled = [g * scale for g in grads]
        if overflow_autocast:
            scaled[0].view(-1)[11] = 70000.0                # ... and inf as a float16 one
        before = [p.detach().clone() for p in ps]
        for p, g_ in zip(ps, scaled):
            if getattr(p, "_mg_g16", None) is not None:
                p._mg_g16.copy_(g_.reshape(-1).half().to(DEV))
            else:
                p.grad.copy_(g_.to(DEV))
            p._mg_fresh = False
        sc.step(opt)
        sc.update()
        if overflow_autocast:
            assert sc.get_scale() == scale * 0.5 and all(torch.equal(p.detach(), b) for p, b in zip(ps, before))
            assert opt.state[0].item() == it               # the clock did not tick
            continue
        assert sc.get_scale() == scale
        ref_opt.zero_grad()
        for r, g_, p in zip(refs, scaled, ps):
            seen = g_ if getattr(p, "_mg_grad_f32", False) else g_.half().float()      # what the reference's .grad holds
            r.grad = seen / scale
        ref_opt.step()
        for p, r in zip(ps, refs):
            assert (p.detach().cpu() - r.detach()).abs().max().item() < 2e-6
        assert torch.equal(opt.flat_h, opt.flat_p.half())
        got = opt.grad_of(ps[3])
        assert got.shape == ps[3].shape and torch.equal(got.reshape(-1), ps[3]._mg_g16.float())
    # MG_NO_G16-style fallback: the same parameters with float32 storage give the same update bit for bit (rounded where consumed)
    a = [torch.nn.Parameter(p.detach().clone()) for p in ps]
    b = [torch.nn.Parameter(p.detach().clone()) for p in ps]
    a[3]._mg_g16_ok = True
    oa, ob = FusedAdam(a, lr=1e-2, betas=(0.5, 0.999), half_shadow=True), FusedAdam(b, lr=1e-2, betas=(0.5, 0.999), half_shadow=True)
    oa.zero_grad(); ob.zero_grad()
    g3 = torch.randn(*shapes[3], generator=gen) * 100.0
    a[3]._mg_g16.copy_(g3.reshape(-1).half().to(DEV)); a[3]._mg_fresh = False
    b[3].grad.copy_(g3.to(DEV)); b[3]._mg_fresh = False
    oa.step(); ob.step()
    assert torch.equal(a[3].detach(), b[3].detach()) and torch.equal(oa.flat_m, ob.flat_m) and torch.equal(oa.flat_v, ob.flat_v)


def test_dead_biases_ride_in_the_float16_arena():
    """A bias whose gradient is identically zero (it feeds an InstanceNorm) behind a weight with float16-stored gradient is carried
    in the float16 arena as well -- whether the arenas are laid out after the first forward pass (the tags are known) or before it
    (ddp.attach: FusedAdam.adopt_g16 moves weight AND bias) -- so segments / data-parallel pieces are not cut at every bias; other
    biases stay float32.  The update equals the float32-storage optimiser's bit for bit (the bias and its moments never move)."""
    from mdctgan_amd import _lib
    from mdctgan_amd import functional as Fh
    from mdctgan_amd.optim import FusedAdam
    torch.manual_seed(0)
    shapes = [(128, 3, 3, 64), (128,), (128, 3, 3, 128), (128,), (64, 3, 3, 128), (64,)]

    def params(tag_before):
        ps = [torch.nn.Parameter(torch.randn(*sh, device=DEV)) for sh in shapes]
        for i in (1, 3, 5):
            Fh.mark_bias_feeds_norm(ps[i]) if i != 5 else None          # the last bias feeds no norm
        if tag_before:
            ps[0]._mg_g16_ok = ps[2]._mg_g16_ok = True
        return ps
    want = [_lib.GRAD_F16, _lib.GRAD_F16, _lib.GRAD_F16, _lib.GRAD_F16, _lib.GRAD_AUTOCAST, _lib.GRAD_AUTOCAST]
    early = params(True)
    o1 = FusedAdam(early, lr=1e-2, betas=(0.5, 0.999), half_shadow=True)
    o1.zero_grad()
    assert o1._modes == want
    late = params(False)
    o2 = FusedAdam(late, lr=1e-2, betas=(0.5, 0.999), half_shadow=True)
    o2.zero_grad()                                                   # arenas first (ddp.attach) ...
    assert o2._modes == [_lib.GRAD_AUTOCAST] * 6 and o2.flat_g16 is None
    for i in (0, 2):                                                 # ... the tags arrive with the first forward pass
        late[i]._mg_g16_ok = True
        o2.adopt_g16(late[i])
    assert o2._modes == want and all((getattr(late[i], "_mg_g16", None) is not None) == (want[i] == _lib.GRAD_F16) for i in range(6))
    plain = params(False)
    o3 = FusedAdam(plain, lr=1e-2, betas=(0.5, 0.999), half_shadow=True)
    o3.zero_grad()
    with torch.no_grad():
        for ps in (late, plain):
            for p, q in zip(ps, early):
                p.copy_(q)
        for o in (o2, o3):
            o.resync_shadow()
    gen = torch.Generator().manual_seed(4)
    for it in range(3):
        grads = [torch.randn(*sh, generator=gen) for sh in shapes]
        for opt, ps in ((o1, early), (o2, late), (o3, plain)):
            opt.zero_grad()
            for i, (p, g_) in enumerate(zip(ps, grads)):
                if i in (1, 3):
                    Fh._zero_grad_bias(p)                            # what the convolution's backward does for a dead bias
                elif getattr(p, "_mg_g16", None) is not None:
                    p._mg_g16.copy_(g_.reshape(-1).half().to(DEV)); p._mg_fresh = False
                else:
                    p.grad.copy_(g_.to(DEV)); p._mg_fresh = False
            opt.step()
        for a, b, c in zip(early, late, plain):
            assert torch.equal(a.detach(), b.detach()) and torch.equal(a.detach(), c.detach()), it
        assert torch.equal(o1.flat_m, o3.flat_m) and torch.equal(o1.flat_v, o3.flat_v) and torch.equal(o2.flat_v, o3.flat_v)
    assert float(o1.flat_m[o1.offsets[1]:o1.offsets[2]].abs().max()) == 0.0       # the dead bias: zero moments, never moved


@pytest.mark.parametrize("case", [("trunk_256_4x8_reflect", 8, 256, 256, 4, 8, 3, 1, True), ("trunk_2048_4x8_reflect", 8, 2048, 2048, 4, 8, 3, 1, True),
                                  ("zero_pad_512_4x8", 8, 512, 512, 4, 8, 3, 1, False), ("ragged_rows_3x4x8", 3, 256, 256, 4, 8, 3, 1, True),
                                  ("bot_1x1_2048_512", 8, 2048, 512, 4, 8, 1, 0, False), ("bot_1x1_512_2048", 8, 512, 2048, 4, 8, 1, 0, False)],
                         ids=lambda c: c[0])
def test_weight_streaming_gemm_equals_the_im2col_gemm(case, monkeypatch):
    """Round 6: forward and data gradient of the weight-streaming --fp16 layers run hgemm_sa_kernel (loader waves, a 6-deep ring for
    the weights and a 3-deep one for the activations; the forward GATHERS its rows from float16(x), no im2col matrix) -- the same
    float16 operands in the same chunk order per accumulator as the im2col + hgemm_kernel pair (MG_NO_HGEMM_SA=1): bit for bit, also
    with float16(x) handed over by the producer (MG_TILES_V_FILLED) and with the skip gradient riding in the epilogue."""
    from mdctgan_amd import _lib, ops
    name, B, Ci, Co, H, W, k, pad, reflect = case
    g = ops.conv_geom(B, H, W, Ci, Co, k, k, 1, pad, reflect, _lib.PRECISION_F16)
    assert ops.plan_name(0, g).startswith("hgemm_sa_kernel<false, true>") and ops.plan_name(1, g).startswith("hgemm_sa_kernel<true, false>")
    assert ops.precast_ok(0, g)
    gen = torch.Generator().manual_seed(len(name))
    x = torch.randn(B, H, W, Ci, generator=gen).to(DEV)
    w = (torch.randn(Co, k, k, Ci, generator=gen) / np.sqrt(k * k * Ci)).to(DEV)
    b = torch.randn(Co, generator=gen).to(DEV)
    dy = torch.randn(B, g.OH, g.OW, Co, generator=gen).to(DEV)
    skip = torch.randn(B, H, W, Ci, generator=gen).to(DEV)
    u = ops.wino_weights(g, w)                                   # the float16 weight copy both passes share
    x16 = x.half().reshape(-1)
    got = (ops.conv_fwd(g, x, w, b), ops.conv_fwd(g, x, w, b, u=u, v_out=x16, v_filled=True), ops.conv_dgrad(g, dy, w, u=u),
           ops.conv_dgrad(g, dy, w, add=skip))
    monkeypatch.setenv("MG_NO_HGEMM_SA", "1")
    assert ops.plan_name(0, g).startswith("hgemm_kernel<")
    ref = (ops.conv_fwd(g, x, w, b), ops.conv_dgrad(g, dy, w, u=u))
    monkeypatch.delenv("MG_NO_HGEMM_SA")
    assert torch.equal(got[0], ref[0]) and torch.equal(got[1], ref[0])
    assert torch.equal(got[2], ref[1]) and torch.equal(got[3], ref[1] + skip)
