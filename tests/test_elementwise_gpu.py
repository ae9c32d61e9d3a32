"""K5/K7/K8/K9/K11/K12 kernels (csrc/norm_act.hip) against plain PyTorch float64 CPU references."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda"


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


def rel_err(got, want):
    return (got.double().cpu() - want).abs().max().item() / max(want.abs().max().item(), 1e-30)


@pytest.mark.parametrize("shape", [(2, 16, 8, 16), (3, 5, 7, 9), (2, 64, 32, 64), (8, 256, 8, 16), (2, 1, 19, 35),
                                   # single-launch slab kernels (C % 32 == 0, HW <= 640): every register depth, ragged HW
                                   (2, 32, 9, 17), (1, 96, 16, 32), (2, 64, 17, 33), (2, 32, 18, 34), (1, 32, 20, 32),
                                   (1, 32, 2, 3), (2, 32, 21, 31)])
@pytest.mark.parametrize("act", ["none", "relu", "lrelu"])
def test_instnorm_fwd_bwd(shape, act):
    from mdctgan_amd import ops
    code = {"none": ops.ACT_NONE, "relu": ops.ACT_RELU, "lrelu": ops.ACT_LRELU02}[act]
    fn = {"none": lambda t: t, "relu": torch.relu, "lrelu": lambda t: F.leaky_relu(t, 0.2)}[act]
    gen = torch.Generator().manual_seed(1)
    x = (torch.randn(*shape, generator=gen, dtype=torch.float64) * 3 + 1.5).requires_grad_()
    res = torch.randn(*shape, generator=gen, dtype=torch.float64)
    y = fn(F.instance_norm(x, eps=1e-5)) + res
    gy = torch.randn(*shape, generator=gen, dtype=torch.float64)
    y.backward(gy)
    xd, rd, gyd = nhwc(x.detach()).float().to(DEV), nhwc(res).float().to(DEV), nhwc(gy).float().to(DEV)
    yd, mean, rstd = ops.instnorm_fwd(xd, code, rd)
    assert rel_err(yd, nhwc(y.detach())) < 1e-5
    dxd = ops.instnorm_bwd(gyd, xd, mean, rstd, code)
    assert rel_err(dxd, nhwc(x.grad)) < 2e-5
    y2, _, _ = ops.instnorm_fwd(xd, code, None)
    assert rel_err(y2, nhwc((y - res).detach())) < 1e-5


@pytest.mark.parametrize("shape", [(8, 64, 128, 256), (2, 64, 32, 64), (8, 128, 64, 128), (3, 12, 7, 9), (2, 4, 19, 35), (8, 256, 33, 65)])
@pytest.mark.parametrize("act", ["none", "relu", "lrelu"])
def test_instnorm_row_kernels_equal_the_flat_ones(shape, act, monkeypatch):
    """norm_apply_rows_kernel (statistics in registers, threads walk pixel rows) against norm_apply_{fwd,bwd}_kernel
    (MG_NO_NORM_ROWS=1): the same arithmetic per element, so the same bits -- outputs, the float16 copies, with and without the
    residual, ragged pixel counts and channel counts that do not fill a 64-channel block."""
    from mdctgan_amd import ops
    code = {"none": ops.ACT_NONE, "relu": ops.ACT_RELU, "lrelu": ops.ACT_LRELU02}[act]
    B, C, H, W = shape
    gen = torch.Generator().manual_seed(11)
    x = (torch.randn(B, H, W, C, generator=gen) * 2 + 0.5).to(DEV)
    res = torch.randn(B, H, W, C, generator=gen).to(DEV)
    gy = torch.randn(B, H, W, C, generator=gen).to(DEV)
    outs = []
    for rows in (False, True):
        if rows:
            monkeypatch.delenv("MG_NO_NORM_ROWS", raising=False)
        else:
            monkeypatch.setenv("MG_NO_NORM_ROWS", "1")
        y16 = torch.zeros(x.numel(), dtype=torch.float16, device=DEV)
        d16 = torch.zeros(x.numel(), dtype=torch.float16, device=DEV)
        y, mean, rstd = ops.instnorm_fwd(x, code, res, y16=y16)
        y2, _, _ = ops.instnorm_fwd(x, code, None)
        dx = ops.instnorm_bwd(gy, x, mean, rstd, code, dx16=d16)
        outs.append((y, y2, dx, y16, d16, mean, rstd))
    for a, b in zip(*outs):
        assert torch.equal(a, b), (shape, act)


def test_act_bwd_add_pool_upsample():
    from mdctgan_amd import ops
    gen = torch.Generator().manual_seed(2)
    for shape in [(2, 3, 32, 64), (1, 8, 17, 33), (2, 4, 5, 6)]:
        x = torch.randn(*shape, generator=gen, dtype=torch.float64, requires_grad=True)
        y = F.avg_pool2d(x, 3, stride=2, padding=1, count_include_pad=False)
        gy = torch.randn(y.shape, generator=gen, dtype=torch.float64)
        y.backward(gy)
        xd = nhwc(x.detach()).float().to(DEV)
        yd = ops.avgpool_fwd(xd)
        assert yd.shape == nhwc(y).shape and rel_err(yd, nhwc(y.detach())) < 1e-6
        dxd = ops.avgpool_bwd(nhwc(gy).float().to(DEV), xd.shape)
        assert rel_err(dxd, nhwc(x.grad)) < 1e-6
        x.grad = None
        u = F.interpolate(x, scale_factor=2.0, mode="nearest")
        gu = torch.randn(u.shape, generator=gen, dtype=torch.float64)
        u.backward(gu)
        assert rel_err(ops.upsample_fwd(xd), nhwc(u.detach())) < 1e-6
        assert rel_err(ops.upsample_bwd(nhwc(gu).float().to(DEV)), nhwc(x.grad)) < 1e-6
    a = torch.randn(1000, generator=gen).to(DEV)
    b = torch.randn(1000, generator=gen).to(DEV)
    assert torch.equal(ops.add(a, b), a + b)
    for act, f in ((ops.ACT_TANH, torch.tanh), (ops.ACT_LRELU02, lambda t: F.leaky_relu(t, 0.2))):
        pre = torch.randn(1000, generator=gen, dtype=torch.float64, requires_grad=True)
        yy = f(pre)
        gg = torch.randn(1000, generator=gen, dtype=torch.float64)
        yy.backward(gg)
        got = ops.act_bwd(gg.float().to(DEV), yy.detach().float().to(DEV), act)
        assert rel_err(got, pre.grad) < 1e-5


def test_dinput_pair_losses_adam():
    from mdctgan_amd import ops
    gen = torch.Generator().manual_seed(3)
    lr = torch.randn(2, 8, 16, generator=gen, dtype=torch.float64)
    s = torch.randn(2, 8, 16, generator=gen, dtype=torch.float64, requires_grad=True)
    out = torch.stack((lr, s, s.abs() * 2 - 1), dim=-1)
    go = torch.randn(out.shape, generator=gen, dtype=torch.float64)
    out.backward(go)
    od = ops.dinput_fwd(lr.float().to(DEV), s.detach().float().to(DEV), -1.0)
    assert rel_err(od, out.detach()) < 1e-6
    assert rel_err(ops.dinput_bwd(go.float().to(DEV), s.detach().float().to(DEV)), s.grad) < 1e-6
    pr = ops.pair_fwd(s.detach().float().to(DEV), -1.0)
    assert rel_err(pr, out.detach()[..., 1:]) < 1e-6
    # losses
    for n in (37, 5320, 300000):
        a = torch.randn(n, generator=gen, dtype=torch.float64, requires_grad=True)
        b = torch.randn(n, generator=gen, dtype=torch.float64)
        la = 0.7 * F.mse_loss(a, torch.ones_like(a)) + 1.3 * F.l1_loss(a, b)
        la.backward()
        ad, bd = a.detach().float().to(DEV), b.float().to(DEV)
        loss = torch.zeros(1, device=DEV)
        ops.mse_const_fwd(ad, 1.0, 0.7, loss, False)
        ops.l1_fwd(ad, bd, 1.3, loss, True)
        assert abs(loss.item() - la.item()) < 1e-5 * abs(la.item())
        go1 = torch.full((1,), 2.0, device=DEV)
        gsum = ops.mse_const_bwd(ad, 1.0, 0.7, go1) + ops.l1_bwd(ad, bd, 1.3, go1)
        assert rel_err(gsum, 2.0 * a.grad) < 1e-5
    # Adam == torch.optim.Adam (betas 0.5/0.999, lr 2e-4), three steps
    p0 = torch.randn(4097, generator=gen)
    pt = p0.clone().requires_grad_()
    opt = torch.optim.Adam([pt], lr=2e-4, betas=(0.5, 0.999))
    pd = p0.clone().to(DEV)
    m = torch.zeros_like(pd)
    v = torch.zeros_like(pd)
    for step in range(1, 4):
        g = torch.randn(4097, generator=gen) * 10 ** float(torch.randint(-6, 1, (1,), generator=gen))
        pt.grad = g.clone()
        opt.step()
        ops.adam_step(pd, g.to(DEV), m, v, 2e-4, 0.5, 0.999, 1e-8, step)
        assert (pd.cpu() - pt.detach()).abs().max().item() <= 2.4e-7   # 1 ulp at |p| < 4: lr * m/denom rounds once differently


def test_multi_tensor_losses_equal_the_single_tensor_calls():
    """mg_loss_multi_fwd / mg_loss_multi_bwd (the feature-matching sum over the discriminator layers as one launch per stage) ==
    the accumulate-in-place sequence of single-tensor calls, bit for bit; the backward also clears the requested tail."""
    from mdctgan_amd import _lib, ops
    lib = _lib.load()

    def bce_fwd(x, t, sc, loss, acc):
        ws = _lib.workspace(lib.mg_loss_workspace(), x.device)
        _lib.check(lib.mg_bce_const_fwd(_lib.ptr(x), x.numel(), t, sc, _lib.ptr(loss), int(acc), _lib.ptr(ws), _lib.stream()), "bce")

    def bce_bwd(x, t, sc, go, out):
        _lib.check(lib.mg_bce_const_bwd(_lib.ptr(x), x.numel(), t, sc, _lib.ptr(go), _lib.ptr(out), _lib.stream()), "bce")
    gen = torch.Generator().manual_seed(13)
    shapes = [(2, 65, 129, 64), (2, 33, 65, 128), (2, 17, 33, 256), (2, 18, 34, 512), (2, 3, 5, 1), (2, 700, 1100, 3)]
    for kind in (ops.LOSS_MSE_CONST, ops.LOSS_L1, ops.LOSS_BCE_CONST):
        a = [torch.rand(s, generator=gen).to("cuda") * 0.98 + 0.01 for s in shapes]
        b = [torch.rand(s, generator=gen).to("cuda") for s in shapes]
        target, scale = (0.0, 1.0) if kind == ops.LOSS_L1 else (1.0, 0.7)
        single_fwd = {ops.LOSS_MSE_CONST: lambda x, y, l, acc: ops.mse_const_fwd(x, target, scale, l, acc),
                      ops.LOSS_L1: lambda x, y, l, acc: ops.l1_fwd(x, y, scale, l, acc),
                      ops.LOSS_BCE_CONST: lambda x, y, l, acc: bce_fwd(x, target, scale, l, acc)}[kind]
        single_bwd = {ops.LOSS_MSE_CONST: lambda x, y, go, out: ops.mse_const_bwd(x, target, scale, go, out=out),
                      ops.LOSS_L1: lambda x, y, go, out: ops.l1_bwd(x, y, scale, go, out=out),
                      ops.LOSS_BCE_CONST: lambda x, y, go, out: bce_bwd(x, target, scale, go, out)}[kind]
        want = torch.full((1,), 3.25, device="cuda")
        got = want.clone()
        for i, (x, y) in enumerate(zip(a, b)):
            single_fwd(x, y, want, True)
        ops.loss_multi_fwd(kind, [(x, y if kind == ops.LOSS_L1 else None, None, 0) for x, y in zip(a, b)], target, scale, got, True)
        assert torch.equal(got, want), (kind, got.item(), want.item())
        fresh = torch.full((1,), float("nan"), device="cuda")
        ops.loss_multi_fwd(kind, [(x, y if kind == ops.LOSS_L1 else None, None, 0) for x, y in zip(a, b)], target, scale, fresh)
        assert torch.equal(fresh, want - 3.25) or abs(fresh.item() - (want.item() - 3.25)) <= 1e-6 * abs(want.item())
        go = torch.full((1,), 0.37, device="cuda")
        rows, wants, bufs = [], [], []
        for x, y in zip(a, b):
            tail = x.numel() // 2
            buf = torch.full((x.numel() + tail,), float("nan"), device="cuda")
            w_ = torch.empty_like(x)
            single_bwd(x, y, go, w_)
            rows.append((x, y if kind == ops.LOSS_L1 else None, buf, tail))
            wants.append(w_)
            bufs.append(buf)
        ops.loss_multi_bwd(kind, rows, target, scale, go)
        for buf, w_ in zip(bufs, wants):
            n = w_.numel()
            assert torch.equal(buf[:n], w_.reshape(-1)) and bool((buf[n:] == 0).all())
