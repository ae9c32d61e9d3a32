"""K3/K4/K6 implicit-GEMM convolution kernels (csrc/conv_igemm.hip) through the C ABI against a plain
PyTorch float64 CPU evaluation of the same nn.Conv2d / nn.ConvTranspose2d / nn.ReflectionPad2d op.
Tolerance: float32 accumulation over K terms -> 3e-5 * max|ref| (exact-f32 MFMA, K <= ~10^4)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def batched_wino_gemm(name):
    """The kernel a pass reports is a batched Winograd-domain GEMM (dense_gemm.h, or the convolution kernels' dense 25-position
    instances), not a direct implicit-GEMM convolution."""
    return name.startswith("dgemm32") or "5>" in name

DEV = "cuda"
ACT_RELU_ = 1          # MG_ACT_RELU (include/mdctgan_hip.h)

# (name, B, Ci, H, W, Co, k, stride, pad, reflect)
CONV_CASES = [
    ("res3x3_reflect", 2, 32, 8, 16, 32, 3, 1, 1, True),
    ("res3x3_reflect_tiny", 3, 16, 2, 16, 16, 3, 1, 1, True),
    ("stem7x7_reflect_ci2", 2, 2, 32, 64, 16, 7, 1, 3, True),
    ("head7x7_reflect_co1", 2, 16, 32, 64, 1, 7, 1, 3, True),
    ("down3x3_s2", 2, 16, 32, 64, 32, 3, 2, 1, False),
    ("down3x3_s2_odd", 1, 16, 17, 33, 32, 3, 2, 1, False),
    ("d4x4_s2_ci3", 2, 3, 32, 64, 16, 4, 2, 2, False),
    ("d4x4_s2", 2, 16, 17, 33, 32, 4, 2, 2, False),
    ("d4x4_s1", 2, 32, 5, 9, 64, 4, 1, 2, False),
    ("d4x4_s1_co1", 2, 64, 6, 10, 1, 4, 1, 2, False),
    # single-output-channel layers with Ci % 64 == 0: tap GEMM + gather / scatter (csrc/conv_co1.h)
    ("co1_head7x7_reflect_64", 2, 64, 20, 36, 1, 7, 1, 3, True),
    ("co1_4x4_s2_zero", 2, 64, 10, 14, 1, 4, 2, 2, False),
    ("co1_3x3_reflect_128", 1, 128, 6, 10, 1, 3, 1, 1, True),
    ("conv5x5_p2", 1, 16, 12, 20, 24, 5, 1, 2, False),
    ("conv5x5_p1", 1, 16, 12, 20, 24, 5, 1, 1, False),
    ("conv3x3_p2", 1, 24, 12, 20, 16, 3, 1, 2, False),
    ("conv1x1", 2, 64, 4, 8, 48, 1, 1, 0, False),
    ("wide_128tile", 2, 64, 16, 32, 256, 3, 1, 1, True),
    ("bottleneck_like", 8, 128, 8, 16, 128, 3, 1, 1, True),
    ("down3x3_s2_ci8", 2, 8, 32, 256, 16, 3, 2, 1, False),
    ("down3x3_s2_ci4", 2, 4, 32, 64, 8, 3, 2, 1, False),
    ("conv3x3_ci24_reflect", 2, 24, 16, 32, 40, 3, 1, 1, True),
    ("conv3x3_ci8_co8", 2, 8, 32, 256, 8, 3, 1, 1, True),
    ("wino_zero_pad", 2, 32, 12, 20, 48, 3, 1, 1, False),
    ("wino_reflect_2rows", 3, 64, 2, 16, 64, 3, 1, 1, True),
    ("wino_reflect_big", 2, 128, 16, 32, 128, 3, 1, 1, True),
    ("odd_size_no_wino", 2, 32, 7, 9, 32, 3, 1, 1, True),
    # 4x4 stride-1 PatchGAN layers: Winograd F(2x2,4x4) when H, W are odd (csrc/wino4.h), direct otherwise
    ("wino4_17x33", 2, 64, 17, 33, 128, 4, 1, 2, False),
    ("wino4_9x17_b3", 3, 128, 9, 17, 64, 4, 1, 2, False),
    ("wino4_1x3", 3, 32, 1, 3, 32, 4, 1, 2, False),
    ("d4x4_s1_even_direct", 2, 32, 6, 10, 64, 4, 1, 2, False),
    # 4x4 stride-2 PatchGAN layers: Winograd F(4x4,2x2) over the space-to-depth view (csrc/wino42.h)
    ("wino42_33x65", 2, 64, 33, 65, 128, 4, 2, 2, False),
    ("wino42_even_16x24", 2, 32, 16, 24, 48, 4, 2, 2, False),
    ("wino42_3x5", 3, 16, 3, 5, 16, 4, 2, 2, False),
    ("wino42_2x2", 1, 16, 2, 2, 32, 4, 2, 2, False),
    # Ci <= 4: VALU data- / weight-gradient kernels (csrc/conv_smallc.h)
    ("smallc_4x4_s2_co80", 1, 3, 17, 33, 80, 4, 2, 2, False),
    ("smallc_4x4_s2_b5", 5, 3, 64, 128, 64, 4, 2, 2, False),
    ("smallc_3x3_ci4", 2, 4, 16, 24, 32, 3, 1, 1, False),
    ("smallc_7x7_reflect_b3", 3, 2, 20, 36, 64, 7, 1, 3, True),
    ("smallc_dgrad_ci1", 2, 1, 9, 14, 16, 3, 1, 1, False),
]


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


@pytest.fixture
def small_wino42(monkeypatch):
    """F(4x4,2x2) is only selected for large problems (where it is faster); the parity cases are small."""
    monkeypatch.setenv("MG_WINO42_MIN_WORK", "0")


def ref_conv(x, w, b, stride, pad, reflect):
    if reflect and pad > 0:
        x = F.pad(x, (pad, pad, pad, pad), mode="reflect")
        pad = 0
    return F.conv2d(x, w, b, stride=stride, padding=pad)


def rel_err(got, want):
    return (got.double().cpu() - want).abs().max().item() / max(want.abs().max().item(), 1e-30)


@pytest.mark.parametrize("case", CONV_CASES, ids=[c[0] for c in CONV_CASES])
def test_conv_fwd_dgrad_wgrad(case, small_wino42):
    from mdctgan_amd import ops
    name, B, Ci, H, W, Co, k, s, p, reflect = case
    gen = torch.Generator().manual_seed(hash(name) % 1000)
    x = torch.randn(B, Ci, H, W, generator=gen, dtype=torch.float64, requires_grad=True)
    w = (torch.randn(Co, Ci, k, k, generator=gen, dtype=torch.float64) / np.sqrt(Ci * k * k)).requires_grad_()
    b = torch.randn(Co, generator=gen, dtype=torch.float64, requires_grad=True)
    y = ref_conv(x, w, b, s, p, reflect)
    gy = torch.randn(y.shape, generator=gen, dtype=torch.float64)
    y.backward(gy)

    g = ops.conv_geom(B, H, W, Ci, Co, k, k, s, p, reflect)
    assert (g.OH, g.OW) == tuple(y.shape[2:])
    xd, wd, bd = nhwc(x.detach()).float().to(DEV), nhwc(w.detach()).float().to(DEV), b.detach().float().to(DEV)
    gyd = nhwc(gy).float().to(DEV)
    if name.startswith("wino42"):
        assert batched_wino_gemm(ops.plan_name(1, g))    # the batched 25-position Winograd-domain GEMM, not the direct kernel
    yd = ops.conv_fwd(g, xd, wd, bd)
    assert rel_err(yd, nhwc(y.detach())) < 3e-5
    dxd = ops.conv_dgrad(g, gyd, wd)
    assert rel_err(dxd, nhwc(x.grad)) < 3e-5
    dw = torch.full((Co, k, k, Ci), 7.0, dtype=torch.float32, device=DEV)
    db = torch.full((Co,), 7.0, dtype=torch.float32, device=DEV)
    ops.conv_wgrad(g, xd, gyd, dw, db)
    assert rel_err(dw, nhwc(w.grad)) < 3e-5
    assert rel_err(db, b.grad) < 3e-5
    ops.conv_wgrad(g, xd, gyd, dw, db, accumulate=True)
    assert rel_err(dw, 2 * nhwc(w.grad)) < 3e-5
    assert rel_err(db, 2 * b.grad) < 3e-5
    if name.startswith("co1_") or name == "d4x4_s1_co1":
        # tap GEMM (csrc/conv_co1.h): the padded weights (u) and the scattered dy (md) may be held by the caller -- same bits
        assert ops.plan_name(0, g).startswith("dgemm32g_kernel<64, 128") and ops.tiles_are_casts(g)
        u = ops.wino_weights(g, wd)
        v, md = ops.wino_tile_buffers(g, xd.device)
        assert u is not None and u.numel() == 64 * Ci and v is None and md is not None
        assert torch.equal(ops.conv_fwd(g, xd, wd, bd, u=u), yd)
        assert torch.equal(ops.conv_dgrad(g, gyd, wd, u=u, md_out=md), dxd)
        dw1, dw2 = torch.empty_like(dw), torch.empty_like(dw)
        ops.conv_wgrad(g, xd, gyd, dw1, None)
        ops.conv_wgrad(g, xd, gyd, dw2, None, md=md)
        assert torch.equal(dw1, dw2)
        assert torch.equal(ops.conv_fwd(g, xd, wd, bd, act=ops.ACT_TANH), torch.tanh(yd))


# (name, B, Ci, H, W, Co, k, stride, pad): Ci % 64 == 0 -> the LDS-DMA weight gradient (csrc/conv_dma.h); the first five are
# eligible for its row-regular gather (a 32-pixel chunk = an aligned piece of one output row, one row, or whole rows of one image)
RR_CASES = [
    ("row_pieces_OW128", 2, 64, 8, 256, 64, 3, 2, 1, True),
    ("row_pieces_OW64_4x4", 1, 64, 16, 128, 128, 4, 2, 1, True),
    ("one_row_OW32", 2, 64, 16, 64, 64, 3, 2, 1, True),
    ("two_rows_OW16", 3, 128, 16, 32, 64, 3, 2, 1, True),
    ("whole_image_OW8", 2, 64, 8, 16, 64, 3, 2, 1, True),
    ("stride1_3x3_odd_height_OW64", 2, 64, 5, 64, 64, 3, 1, 1, True),        # (an odd height keeps the layer off the Winograd path)
    ("stride1_3x3_OW32", 2, 64, 7, 32, 128, 3, 1, 1, True),
    ("stride1_1x1_OW64", 2, 64, 8, 64, 64, 1, 1, 0, True),
    ("odd_width_general_path", 2, 64, 17, 33, 64, 3, 2, 1, False),
    ("pixels_not_a_multiple_of_32", 1, 64, 6, 10, 64, 3, 2, 1, False),
]


@pytest.mark.parametrize("case", RR_CASES, ids=[c[0] for c in RR_CASES])
def test_wgrad_row_regular_gather_is_the_general_gather(case, monkeypatch):
    """conv_wgrad_dma_kernel<..., RR = true> (scalar source walk + three precomputed lane offsets) against the per-lane coordinate
    walk of the same kernel (MG_NO_WGRAD_RR=1): the same bits, and both against float64."""
    from mdctgan_amd import ops
    name, B, Ci, H, W, Co, k, s, p, eligible = case
    gen = torch.Generator().manual_seed(len(name))
    x = torch.randn(B, Ci, H, W, generator=gen, dtype=torch.float64)
    w = torch.zeros(Co, Ci, k, k, dtype=torch.float64, requires_grad=True)
    y = F.conv2d(x, w, None, stride=s, padding=p)
    gy = torch.randn(y.shape, generator=gen, dtype=torch.float64)
    y.backward(gy)
    g = ops.conv_geom(B, H, W, Ci, Co, k, k, s, p, False)
    assert "conv_wgrad_dma_kernel" in ops.plan_name(2, g)
    pixels = B * g.OH * g.OW
    assert eligible == (pixels % 32 == 0 and (g.OW % 32 == 0 or 32 % g.OW == 0))
    xd, gyd = nhwc(x).float().to(DEV), nhwc(gy).float().to(DEV)
    out = {}
    for tag, env in (("rr", None), ("general", "1")):
        if env:
            monkeypatch.setenv("MG_NO_WGRAD_RR", env)
        else:
            monkeypatch.delenv("MG_NO_WGRAD_RR", raising=False)
        dw = torch.full((Co, k, k, Ci), 3.0, dtype=torch.float32, device=DEV)
        ops.conv_wgrad(g, xd, gyd, dw, None)
        assert rel_err(dw, nhwc(w.grad)) < 3e-5, tag
        ops.conv_wgrad(g, xd, gyd, dw, None, accumulate=True)
        assert rel_err(dw, 2 * nhwc(w.grad)) < 3e-5, tag
        out[tag] = dw
    assert torch.equal(out["rr"], out["general"])
    for splits in (3, 7, 13):            # any split count: the K range is cut at floor(chunks * j / splits)
        if (pixels + 31) // 32 >= splits:
            monkeypatch.setenv("MG_FORCE_CONV_DMA", "64,64,%d" % splits)
            monkeypatch.delenv("MG_NO_WGRAD_RR", raising=False)
            dw = torch.empty(Co, k, k, Ci, dtype=torch.float32, device=DEV)
            ops.conv_wgrad(g, xd, gyd, dw, None)
            assert rel_err(dw, nhwc(w.grad)) < 3e-5, splits
            monkeypatch.delenv("MG_FORCE_CONV_DMA")


def test_conv_fwd_fused_activations():
    from mdctgan_amd import ops
    gen = torch.Generator().manual_seed(5)
    x = torch.randn(2, 16, 9, 13, generator=gen, dtype=torch.float64)
    w = torch.randn(8, 16, 3, 3, generator=gen, dtype=torch.float64) / 12
    b = torch.randn(8, generator=gen, dtype=torch.float64)
    y = F.conv2d(x, w, b, padding=1)
    g = ops.conv_geom(2, 9, 13, 16, 8, 3, 3, 1, 1, False)
    xd, wd, bd = nhwc(x).float().to(DEV), nhwc(w).float().to(DEV), b.float().to(DEV)
    for act, fn in ((ops.ACT_LRELU02, lambda t: F.leaky_relu(t, 0.2)), (ops.ACT_TANH, torch.tanh),
                    (ops.ACT_RELU, torch.relu)):
        assert rel_err(ops.conv_fwd(g, xd, wd, bd, act), nhwc(fn(y))) < 3e-5


CONVT_CASES = [("up3x3", 2, 32, 8, 16, 16), ("up3x3_odd", 1, 16, 5, 7, 32), ("up_big", 2, 128, 16, 32, 64)]


@pytest.mark.parametrize("case", CONVT_CASES, ids=[c[0] for c in CONVT_CASES])
def test_conv_transpose(case):
    """nn.ConvTranspose2d(k3, s2, p1, output_padding=1) == the data-gradient kernel of the stride-2 conv."""
    from mdctgan_amd import ops
    name, B, Cin, h, w_, Cout = case
    gen = torch.Generator().manual_seed(11)
    x = torch.randn(B, Cin, h, w_, generator=gen, dtype=torch.float64, requires_grad=True)
    w = (torch.randn(Cin, Cout, 3, 3, generator=gen, dtype=torch.float64) / np.sqrt(Cin * 9)).requires_grad_()
    b = torch.randn(Cout, generator=gen, dtype=torch.float64, requires_grad=True)
    y = F.conv_transpose2d(x, w, b, stride=2, padding=1, output_padding=1)
    gy = torch.randn(y.shape, generator=gen, dtype=torch.float64)
    y.backward(gy)
    # equivalent convolution: high-res [B, 2h, 2w, Cout] --3x3 s2 p1--> low-res [B, h, w, Cin]
    g = ops.conv_geom(B, 2 * h, 2 * w_, Cout, Cin, 3, 3, 2, 1, False)
    assert (g.OH, g.OW) == (h, w_)
    xd, gyd = nhwc(x.detach()).float().to(DEV), nhwc(gy).float().to(DEV)
    wd = nhwc(w.detach()).float().to(DEV)          # [Cin_T, kh, kw, Cout_T] == OHWI of the equivalent conv
    bd = b.detach().float().to(DEV)
    yd = ops.conv_dgrad(g, xd, wd, bd)
    assert rel_err(yd, nhwc(y.detach())) < 3e-5
    dxd = ops.conv_fwd(g, gyd, wd)                 # input gradient of the transposed conv == forward conv
    assert rel_err(dxd, nhwc(x.grad)) < 3e-5
    dw = torch.empty(Cin, 3, 3, Cout, dtype=torch.float32, device=DEV)
    ops.conv_wgrad(g, gyd, xd, dw)                 # roles swapped: "x" = high-res grad, "dy" = low-res input
    assert rel_err(dw, nhwc(w.grad)) < 3e-5
    db = torch.empty(Cout, dtype=torch.float32, device=DEV)
    ops.colsum(gyd.reshape(-1, Cout), db)
    assert rel_err(db, b.grad) < 3e-5


@pytest.mark.parametrize("reflect", [True, False])
def test_winograd_shared_weight_transform(reflect):
    """mg_conv_wino_prepare + the _u entry points: forward and data gradient with a caller-held transformed-weight
    image give bit-identical results to the calls that transform internally; non-Winograd geometries report 0 bytes."""
    from mdctgan_amd import ops
    gen = torch.Generator().manual_seed(5)
    B, C, H, W = 2, 64, 8, 16
    g = ops.conv_geom(B, H, W, C, C, 3, 3, 1, 1, reflect)
    x = torch.randn(B, H, W, C, generator=gen).to(DEV)
    dy = torch.randn(B, H, W, C, generator=gen).to(DEV)
    w = (torch.randn(C, 3, 3, C, generator=gen) * 0.05).to(DEV)
    u = ops.wino_weights(g, w)
    assert u is not None and u.numel() == 16 * C * C
    assert torch.equal(ops.conv_fwd(g, x, w, None, ops.ACT_NONE, u), ops.conv_fwd(g, x, w))
    assert torch.equal(ops.conv_dgrad(g, dy, w, u=u), ops.conv_dgrad(g, dy, w))
    # the tile images a step shares: V = B^T x B kept by forward, Md = A dy A^T kept by the data gradient, both consumed
    # by the weight gradient -- same bits as the calls that transform internally
    v, md = ops.wino_tile_buffers(g, x.device)
    assert v is not None and v.numel() == 16 * (B * H * W // 4) * C and md is not None and md.numel() == v.numel()
    assert torch.equal(ops.conv_fwd(g, x, w, None, ops.ACT_NONE, u, v), ops.conv_fwd(g, x, w))
    assert torch.equal(ops.conv_dgrad(g, dy, w, u=u, md_out=md), ops.conv_dgrad(g, dy, w))
    dw0, dw1 = torch.empty(C, 3, 3, C, device=DEV), torch.empty(C, 3, 3, C, device=DEV)
    ops.conv_wgrad(g, x, dy, dw0, None)
    ops.conv_wgrad(g, x, dy, dw1, None, v=v, md=md)
    assert torch.equal(dw0, dw1)
    g2 = ops.conv_geom(B, H, W, C, C, 3, 3, 2, 1, False)
    assert ops.wino_weights(g2, w) is None


def test_winograd42_shared_images(small_wino42):
    """F(4x4,2x2) over the space-to-depth view (stride-2 4x4 layers) behind the caller-held images."""
    from mdctgan_amd import ops
    gen = torch.Generator().manual_seed(7)
    B, Ci, Co, H, W = 2, 32, 64, 17, 33
    g = ops.conv_geom(B, H, W, Ci, Co, 4, 4, 2, 2, False)
    assert (g.OH, g.OW) == (H // 2 + 1, W // 2 + 1)
    x = torch.randn(B, H, W, Ci, generator=gen).to(DEV)
    dy = torch.randn(B, g.OH, g.OW, Co, generator=gen).to(DEV)
    w = (torch.randn(Co, 4, 4, Ci, generator=gen) * 0.05).to(DEV)
    u = ops.wino_weights(g, w)
    assert u is not None and u.numel() == 25 * Co * 4 * Ci
    T = B * ((g.OH + 3) // 4) * ((g.OW + 3) // 4)
    v, md = ops.wino_tile_buffers(g, x.device)
    assert v.numel() == 25 * T * 4 * Ci and md.numel() == 25 * T * Co
    assert torch.equal(ops.conv_fwd(g, x, w, None, ops.ACT_NONE, u, v), ops.conv_fwd(g, x, w))
    assert torch.equal(ops.conv_dgrad(g, dy, w, u=u, md_out=md), ops.conv_dgrad(g, dy, w))
    dw0, dw1 = torch.empty(Co, 4, 4, Ci, device=DEV), torch.empty(Co, 4, 4, Ci, device=DEV)
    ops.conv_wgrad(g, x, dy, dw0, None)
    ops.conv_wgrad(g, x, dy, dw1, None, v=v, md=md)
    assert torch.equal(dw0, dw1)


def test_winograd42_selected_by_problem_size():
    """Default gating: the batch-16 layers of the first discriminator scale take F(4x4,2x2), the small second-scale
    layers keep the direct kernels (measured slower there)."""
    from mdctgan_amd import ops
    big = ops.conv_geom(16, 65, 129, 64, 128, 4, 4, 2, 2, False)
    small = ops.conv_geom(16, 33, 65, 64, 128, 4, 4, 2, 2, False)
    assert batched_wino_gemm(ops.plan_name(0, big)) and ops.wino_weights_bytes(big) == 25 * 128 * 256 * 4
    assert not batched_wino_gemm(ops.plan_name(0, small)) and ops.wino_weights_bytes(small) == 0      # a direct kernel


def test_winograd4_shared_images():
    """The F(2x2,4x4) path behind the same caller-held images: U (25 x Co x Ci), V and Md (25 x tiles x C) shared
    between forward, data gradient and weight gradient give the bits of the self-contained calls."""
    from mdctgan_amd import ops
    gen = torch.Generator().manual_seed(6)
    B, Ci, Co, H, W = 2, 64, 96, 9, 17
    g = ops.conv_geom(B, H, W, Ci, Co, 4, 4, 1, 2, False)
    assert (g.OH, g.OW) == (H + 1, W + 1)
    x = torch.randn(B, H, W, Ci, generator=gen).to(DEV)
    dy = torch.randn(B, H + 1, W + 1, Co, generator=gen).to(DEV)
    w = (torch.randn(Co, 4, 4, Ci, generator=gen) * 0.05).to(DEV)
    u = ops.wino_weights(g, w)
    assert u is not None and u.numel() == 25 * Co * Ci
    T = B * ((H + 1) // 2) * ((W + 1) // 2)
    v, md = ops.wino_tile_buffers(g, x.device)
    assert v.numel() == 25 * T * Ci and md.numel() == 25 * T * Co
    assert torch.equal(ops.conv_fwd(g, x, w, None, ops.ACT_NONE, u, v), ops.conv_fwd(g, x, w))
    assert torch.equal(ops.conv_dgrad(g, dy, w, u=u, md_out=md), ops.conv_dgrad(g, dy, w))
    dw0, dw1 = torch.empty(Co, 4, 4, Ci, device=DEV), torch.empty(Co, 4, 4, Ci, device=DEV)
    ops.conv_wgrad(g, x, dy, dw0, None)
    ops.conv_wgrad(g, x, dy, dw1, None, v=v, md=md)
    assert torch.equal(dw0, dw1)
    assert batched_wino_gemm(ops.plan_name(1, g))   # the batched 25-position Winograd-domain GEMM


@pytest.mark.parametrize("case", [("trunk_3x3", 8, 8, 16, 256, 256, 3, 1, 1, True, 16), ("d_4x4_s1", 4, 17, 33, 256, 512, 4, 1, 2, False, 25),
                                  ("d_4x4_s2", 16, 33, 65, 128, 256, 4, 2, 2, False, 25)], ids=lambda c: c[0])
def test_f32_split_gemms_are_float32_accurate(case, monkeypatch):
    """MG_F32_SPLIT=1 (opt-in): the Winograd-domain GEMMs take each float32 operand as three exact bf16 pieces on the bf16
    MFMA pipe (csrc/dense_gemm.h::dg_chunk_b8).  All three passes of a layer against a float64 convolution: the error of the
    split path is not larger than 1.5x the native float32 path's, and both are at float32 rounding level."""
    from mdctgan_amd import ops
    _, B, H, W, Ci, Co, k, stride, pad, reflect, P = case
    gen = torch.Generator().manual_seed(5)
    g = ops.conv_geom(B, H, W, Ci, Co, k, k, stride, pad, reflect)
    x = torch.randn(B, H, W, Ci, generator=gen)
    w = torch.randn(Co, k, k, Ci, generator=gen) / np.sqrt(k * k * Ci)
    dy = torch.randn(B, g.OH, g.OW, Co, generator=gen)
    x64 = x.double().permute(0, 3, 1, 2).requires_grad_(True)
    w64 = w.double().permute(0, 3, 1, 2).requires_grad_(True)
    if reflect:
        y64 = torch.nn.functional.conv2d(torch.nn.functional.pad(x64, (pad,) * 4, mode="reflect"), w64, stride=stride)
    else:
        y64 = torch.nn.functional.conv2d(x64, w64, stride=stride, padding=pad)
    y64.backward(dy.double().permute(0, 3, 1, 2))
    ref = (y64.detach().permute(0, 2, 3, 1), x64.grad.permute(0, 2, 3, 1), w64.grad.permute(0, 2, 3, 1))

    def run():
        xd, wd, dyd = x.to(DEV), w.to(DEV), dy.to(DEV)
        dw = torch.empty_like(wd)
        ops.conv_wgrad(g, xd, dyd, dw, None)
        outs = (ops.conv_fwd(g, xd, wd), ops.conv_dgrad(g, dyd, wd), dw)
        return [float((o.double().cpu() - r).abs().max() / r.abs().max()) for o, r in zip(outs, ref)]
    monkeypatch.delenv("MG_F32_SPLIT", raising=False)
    assert ops.plan_name(2, g).endswith(", %d, 0>" % P)
    native = run()
    monkeypatch.setenv("MG_F32_SPLIT", "1")
    assert ops.plan_name(2, g) == "dgemm32g_kernel<128, 128, 2, 2, 1, 1, 2, %d, 1>" % P
    split = run()
    for e_n, e_s in zip(native, split):
        assert e_n < 5e-5 and e_s < 5e-5
        assert e_s <= 1.5 * e_n + 1e-7


@pytest.mark.parametrize("case", [("trunk_like_wino", 8, 128, 8, 16, ACT_RELU_, True), ("trunk_tiny_4x8", 4, 256, 4, 8, 0, True),
                                  ("bigger_map_16x32", 2, 128, 16, 32, 0, False), ("fallback_direct", 2, 16, 8, 16, ACT_RELU_, True),
                                  # the bench's own trunk shapes: configs[1] 1024 ch @ 8x16 and configs[2]
                                  # 2048 ch @ 4x8 at batch 8, configs[4] 1024 ch @ 8x16 at batch 64 -- other tiles / splits
                                  ("configs1_trunk_1024_8x16", 8, 1024, 8, 16, ACT_RELU_, True),
                                  ("configs2_trunk_2048_4x8", 8, 2048, 4, 8, ACT_RELU_, True),
                                  ("configs4_trunk_1024_8x16_b64", 64, 1024, 8, 16, ACT_RELU_, True)],
                         ids=lambda c: c[0])
def test_conv_fwd_instnorm_matches_separate_calls(case):
    """mg_conv_fwd_instnorm_w (the Winograd inverse transform, the InstanceNorm statistics and the apply in one kernel on small
    maps; csrc/wino.h::wino_out_norm_kernel) == mg_conv_fwd_w followed by mg_instnorm_fwd."""
    from mdctgan_amd import ops
    name, B, C, H, W, act, with_res = case
    gen = torch.Generator().manual_seed(len(name))
    x = torch.randn(B, H, W, C, generator=gen).to(DEV)
    w = (torch.randn(C, 3, 3, C, generator=gen) / np.sqrt(9 * C)).to(DEV)
    b = torch.randn(C, generator=gen).to(DEV)
    res = torch.randn(B, H, W, C, generator=gen).to(DEV) if with_res else None
    g = ops.conv_geom(B, H, W, C, C, 3, 3, 1, 1, True)
    y0 = ops.conv_fwd(g, x, w, b)
    n0, m0, r0 = ops.instnorm_fwd(y0, act, res, 1e-5)
    y, y_raw, mean, rstd = ops.conv_fwd_instnorm(g, x, w, b, act, res, 1e-5)
    assert torch.equal(y_raw, y0)                                   # the same inverse transform arithmetic
    assert (mean - m0).abs().max().item() <= 1e-6 * (m0.abs().max().item() + 1.0)
    assert (rstd - r0).abs().max().item() <= 2e-6 * r0.abs().max().item()
    assert (y - n0).abs().max().item() <= 1e-5 * (n0.abs().max().item() + 1.0)
    # with the caller-held Winograd images (u, v) as the training step passes them
    u = ops.wino_weights(g, w)
    if u is not None:
        v, _ = ops.wino_tile_buffers(g, x.device, want_md=False)
        y2, yr2, _, _ = ops.conv_fwd_instnorm(g, x, w, b, act, res, 1e-5, u=u, v_out=v)
        assert torch.equal(yr2, y_raw) and torch.equal(y2, y)
    # inference (torch.no_grad(): no backward pass will read the raw convolution output): y_raw = NULL -- the fused kernel skips
    # that store, the two-call path normalises in place; same y and statistics bit for bit
    y3, yr3, m3, r3 = ops.conv_fwd_instnorm(g, x, w, b, act, res, 1e-5, need_raw=False)
    assert yr3 is None and torch.equal(y3, y) and torch.equal(m3, mean) and torch.equal(r3, rstd)


def test_conv_instnorm_skips_the_raw_output_under_no_grad(monkeypatch):
    """functional.conv_instnorm asks for y_raw only when a backward pass can follow: torch.no_grad() (generate / inference) -> NULL,
    same output bit for bit; with gradients enabled the raw output is saved as before."""
    from mdctgan_amd import functional as Fh
    from mdctgan_amd import ops
    seen = []
    real = ops.conv_fwd_instnorm

    def spy(*a, **kw):
        seen.append(kw.get("need_raw", True))
        return real(*a, **kw)
    monkeypatch.setattr(ops, "conv_fwd_instnorm", spy)
    gen = torch.Generator().manual_seed(3)
    x = torch.randn(2, 64, 8, 16, generator=gen).to(DEV).contiguous(memory_format=torch.channels_last)
    w = torch.nn.Parameter((torch.randn(64, 64, 3, 3, generator=gen) / 24).to(DEV).contiguous(memory_format=torch.channels_last))
    with torch.no_grad():
        y0 = Fh.conv_instnorm(x, w, None, padding=1, reflect=True, act=Fh.ACT_RELU)
    y1 = Fh.conv_instnorm(x, w, None, padding=1, reflect=True, act=Fh.ACT_RELU)
    assert seen == [False, True] and torch.equal(y0, y1)
    y1.sum().backward()
    assert w.grad is not None and bool(torch.isfinite(w.grad).all())


@pytest.mark.parametrize("case", [("small_map_fused", 4, 128, 8, 16, 3, 1, 1, True), ("big_map_two_kernels", 2, 64, 32, 64, 3, 1, 1, True),
                                  ("stride2_direct", 2, 64, 16, 32, 3, 2, 1, False),
                                  ("configs1_trunk_1024_8x16", 8, 1024, 8, 16, 3, 1, 1, True),
                                  ("configs2_trunk_2048_4x8", 8, 2048, 4, 8, 3, 1, 1, True)], ids=lambda c: c[0])
def test_conv_dgrad_add_is_dgrad_plus_tensor(case):
    """mg_wino_tiles.add: dx += add, folded into the gather kernel on small Winograd maps, a separate pass elsewhere --
    bit for bit dgrad(dy) + add either way (one float32 addition per element)."""
    from mdctgan_amd import ops
    name, B, C, H, W, k, s, p, reflect = case
    gen = torch.Generator().manual_seed(len(name))
    g = ops.conv_geom(B, H, W, C, C, k, k, s, p, reflect)
    dy = torch.randn(B, g.OH, g.OW, C, generator=gen).to(DEV)
    w = (torch.randn(C, k, k, C, generator=gen) / np.sqrt(k * k * C)).to(DEV)
    skip = torch.randn(B, H, W, C, generator=gen).to(DEV)
    dx0 = ops.conv_dgrad(g, dy, w)
    dx1 = ops.conv_dgrad(g, dy, w, add=skip)
    assert torch.equal(dx1, dx0 + skip)


@pytest.mark.parametrize("case", [("trunk_like", 8, 128, 8, 16, ACT_RELU_), ("trunk_tiny_4x8", 4, 256, 4, 8, 0), ("map_16x32", 2, 128, 16, 32, 0),
                                  ("configs1_trunk_1024_8x16", 8, 1024, 8, 16, ACT_RELU_), ("configs2_trunk_2048_4x8", 8, 2048, 4, 8, ACT_RELU_)],
                         ids=lambda c: c[0])
def test_instnorm_bwd_wino_md_matches_separate_calls(case):
    """mg_instnorm_bwd_wino_md (InstanceNorm backward + A dy A^T in one kernel, no dy in HBM) followed by the data / weight
    gradient from the images == mg_instnorm_bwd, then the two gradients from dy."""
    from mdctgan_amd import ops
    name, B, C, H, W, act = case
    gen = torch.Generator().manual_seed(3 + len(name))
    x = torch.randn(B, H, W, C, generator=gen).to(DEV)
    w = (torch.randn(C, 3, 3, C, generator=gen) / np.sqrt(9 * C)).to(DEV)
    gy = torch.randn(B, H, W, C, generator=gen).to(DEV)
    g = ops.conv_geom(B, H, W, C, C, 3, 3, 1, 1, True)
    assert ops.wino_md_from_norm_ok(g)
    u = ops.wino_weights(g, w)
    v, md1 = ops.wino_tile_buffers(g, x.device)
    _, md2 = ops.wino_tile_buffers(g, x.device, want_v=False)
    y, y_raw, mean, rstd = ops.conv_fwd_instnorm(g, x, w, None, act, None, 1e-5, u=u, v_out=v)
    d_raw = ops.instnorm_bwd(gy, y_raw, mean, rstd, act)
    dx1 = ops.conv_dgrad(g, d_raw, w, u=u, md_out=md1)
    dw1 = torch.empty(C, 3, 3, C, device=DEV)
    ops.conv_wgrad(g, x, d_raw, dw1, None, v=v, md=md1)
    ops.instnorm_bwd_wino_md(g, gy, y_raw, mean, rstd, act, md2)
    dx2 = ops.conv_dgrad(g, None, w, u=u, md_out=md2)
    dw2 = torch.empty_like(dw1)
    ops.conv_wgrad(g, None, None, dw2, None, v=v, md=md2)
    scale = md1.abs().max().item()
    assert (md1 - md2).abs().max().item() <= 2e-6 * scale
    assert (dx1 - dx2).abs().max().item() <= 1e-5 * dx1.abs().max().item()
    assert (dw1 - dw2).abs().max().item() <= 1e-5 * dw1.abs().max().item()


@pytest.mark.parametrize("case", [("local_block_128_64x128_f16", 2, 128, 64, 128, True, True), ("local_block_128_16x32_f32", 2, 128, 16, 32, True, False),
                                  ("trunk_2048_4x8_f16", 8, 2048, 4, 8, True, True), ("trunk_256_4x8_f16", 8, 256, 4, 8, True, True),
                                  ("zero_pad_128_16x32_f16", 2, 128, 16, 32, False, True)], ids=lambda c: c[0])
def test_dgrad_add_rides_in_the_fold_and_the_split_k_epilogue(case):
    """Round 6: mg_wino_tiles.add is also taken by the reflection fold of the LDS-DMA data gradient (wino_fold_reflect_kernel) and by
    the split-K epilogue of the weight-streaming float16 GEMM -- the ResnetBlock skip gradient of the local enhancer's 128-channel
    blocks and of the --fp16 trunk no longer costs an add launch.  dx == dgrad(dy) + add bit for bit (the addition is float32,
    AFTER the autocast result's float16 rounding, where the separate launch had it)."""
    from mdctgan_amd import _lib, ops
    name, B, C, H, W, reflect, half = case
    gen = torch.Generator().manual_seed(len(name))
    g = ops.conv_geom(B, H, W, C, C, 3, 3, 1, 1, reflect, _lib.PRECISION_F16 if half else _lib.PRECISION_F32)
    dy = torch.randn(B, H, W, C, generator=gen).to(DEV)
    w = (torch.randn(C, 3, 3, C, generator=gen) / np.sqrt(9 * C)).to(DEV)
    skip = torch.randn(B, H, W, C, generator=gen).to(DEV)
    dx0 = ops.conv_dgrad(g, dy, w)
    dx1 = ops.conv_dgrad(g, dy, w, add=skip)
    assert torch.equal(dx1, dx0 + skip)


@pytest.mark.parametrize("case", [("trunk_2048_4x8_f16", 8, 2048, 2048, 4, 8, 1, True, True, True), ("trunk_256_4x8_f16_nores", 8, 256, 256, 4, 8, 1, True, True, False),
                                  ("ladder_512_1024_s2_f32", 8, 512, 1024, 16, 32, 2, False, False, False),
                                  ("ladder_512_1024_s2_f16", 8, 512, 1024, 16, 32, 2, False, True, False),
                                  ("ladder_256_512_s2_f16", 8, 256, 512, 32, 64, 2, False, True, False)], ids=lambda c: c[0])
def test_instnorm_finishes_the_split_k_convolution(case, monkeypatch):
    """Round 6: where a forward convolution leaves split-K slabs and the map is small enough for the one-launch InstanceNorm
    kernel, that kernel sums the slabs itself (+ bias, + the autocast rounding: splitk_epilogue_kernel's arithmetic in its order)
    instead of reading a tensor an epilogue launch wrote: y, the raw convolution output, mean and rstd are bit for bit what the
    two launches give (MG_NO_FWD_DEFER=1), also without the raw output (no_grad) and with the float16 copy for the next layer."""
    from mdctgan_amd import _lib, ops
    name, B, Ci, Co, H, W, stride, reflect, half, with_res = case
    gen = torch.Generator().manual_seed(len(name))
    g = ops.conv_geom(B, H, W, Ci, Co, 3, 3, stride, 1, reflect, _lib.PRECISION_F16 if half else _lib.PRECISION_F32)
    x = torch.randn(B, H, W, Ci, generator=gen).to(DEV)
    w = (torch.randn(Co, 3, 3, Ci, generator=gen) / np.sqrt(9 * Ci)).to(DEV)
    b = torch.randn(Co, generator=gen).to(DEV)
    res = torch.randn(B, g.OH, g.OW, Co, generator=gen).to(DEV) if with_res else None

    def run(need_raw, want16):
        y16 = torch.empty(B * g.OH * g.OW * Co, dtype=torch.float16, device=DEV) if want16 else None
        y, y_raw, mean, rstd = ops.conv_fwd_instnorm(g, x, w, b, ACT_RELU_, res, 1e-5, y16=y16, need_raw=need_raw)
        return y, y_raw, mean, rstd, y16
    monkeypatch.setenv("MG_NO_FWD_DEFER", "1")
    ref = run(True, half)
    monkeypatch.delenv("MG_NO_FWD_DEFER")
    got = run(True, half)
    for a, b_ in zip(ref, got):
        assert (a is None and b_ is None) or torch.equal(a, b_)
    lean = run(False, False)
    assert lean[1] is None and torch.equal(lean[0], ref[0]) and torch.equal(lean[2], ref[2]) and torch.equal(lean[3], ref[3])
    # (whether a case really leaves slabs is the planner's business: mg_conv_plan_splits says so for the LDS-DMA layers)
