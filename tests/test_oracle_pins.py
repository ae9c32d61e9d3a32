"""The oracle's mask-pinned evaluation (oracle/step.py::MaskPins, pin_activations) -- the checker of
tests/test_fullsize_step_gpu.py::test_full_size_step_gradients_mask_pinned -- held to two statements on the CPU:

* pinned to its OWN decisions (recorded by the same slots) a float64 evaluation reproduces its losses and every gradient to
  1e-12: the pinned slots, the signed L1 and the signed |s| are the same function at that point;
* a float32 evaluation against the float64 evaluation pinned to the float32 run's decisions agrees to float32 rounding on every
  gradient -- the property the GPU test relies on -- where the two unpinned evaluations may differ by whole masks.

Covers netG=global and netG=local with bottleneck-transformer blocks (BatchNorm inside), num_D 2 / 3 (train.py:160-202,
models/pix2pixHD_model.py:416-451)."""
import copy

import numpy as np
import pytest
import torch

from oracle import nets as onets
from oracle import step as ostep


def _build(kind):
    torch.manual_seed(7)
    gen = torch.Generator().manual_seed(11)
    if kind == "global":
        netG = onets.build_generator("global", 2, 1, 8, 2, 2, input_size=(32, 256))
        num_D = 2
    else:
        netG = onets.build_generator("local", 2, 1, 4, 2, 2, 1, input_size=(32, 256), n_attn_g=2, heads_g=2, dim_head_g=8)
        num_D = 3
    netD = onets.MultiscaleDRef(3, 8, 3, num_D)
    onets.init_weights(netG, gen)
    onets.init_weights(netD, gen)
    return netG, netD, num_D


def _audio(batch=2, T=7936):
    g = torch.Generator().manual_seed(5)
    hr = 0.05 * torch.randn(batch, T, generator=g)
    spec = torch.fft.rfft(hr)
    spec[:, spec.shape[-1] // 4:] = 0
    return torch.fft.irfft(spec, n=T).numpy(), hr.numpy()


def _grads(ref, lr, hr, pins=None):
    losses, _ = ref.forward_losses(lr, hr, pins=pins)
    ref.netG.zero_grad(); ref.netD.zero_grad()
    (losses["G_GAN"] + losses["G_GAN_Feat"]).backward(retain_graph=True)
    gG = {k: p.grad.detach().double().numpy().copy() for k, p in ref.netG.named_parameters()}
    ref.netD.zero_grad()
    ((losses["D_fake"] + losses["D_real"]) * 0.5).backward()
    gD = {k: p.grad.detach().double().numpy().copy() for k, p in ref.netD.named_parameters()}
    return {k: float(v.detach()) for k, v in losses.items()}, gG, gD


def _rel(a, b):
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))


@pytest.mark.parametrize("kind", ["global", "local_attn"])
def test_pinned_evaluation_reproduces_itself_and_removes_the_mask_lottery(kind):
    netG, netD, num_D = _build(kind)
    lr, hr = _audio()
    # the float32 evaluation, its decisions recorded by the pinned slots themselves
    g32, d32 = copy.deepcopy(netG), copy.deepcopy(netD)
    ref32 = ostep.HotPathRef(g32, d32, ostep.CodecCfg(), num_D=num_D)
    pins32 = ostep.MaskPins(record=True)
    used_G = ostep.pin_activations(ref32.netG, pins32, pins32.act_G)
    used_D = ostep.pin_activations(ref32.netD, pins32, pins32.act_D)
    assert len(used_G) >= 7 and len(used_D) == 4 * num_D
    l32, gG32, gD32 = _grads(ref32, lr, hr, pins=pins32)
    pins32.finish_recording()
    assert sorted(pins32.act_G) == sorted(set(used_G)) and len(pins32.l1_sign) == 4 * num_D
    # recording does not change the function: the same float32 nets without pinned slots
    plain = ostep.HotPathRef(copy.deepcopy(netG), copy.deepcopy(netD), ostep.CodecCfg(), num_D=num_D)
    lp, gGp, gDp = _grads(plain, lr, hr)
    assert lp == l32
    assert all(np.array_equal(gGp[k], gG32[k]) for k in gGp) and all(np.array_equal(gDp[k], gD32[k]) for k in gDp)

    # float64, unpinned and pinned to its own decisions
    ref64 = ostep.HotPathRef(copy.deepcopy(netG), copy.deepcopy(netD), ostep.CodecCfg(), num_D=num_D, dtype=torch.float64)
    own = ostep.MaskPins(record=True)
    ostep.pin_activations(ref64.netG, own, own.act_G)
    ostep.pin_activations(ref64.netD, own, own.act_D)
    l64, gG64, gD64 = _grads(ref64, lr, hr, pins=own)
    own.finish_recording()
    l64p, gG64p, gD64p = _grads(ref64, lr, hr, pins=own)
    for k in l64:
        assert abs(l64p[k] - l64[k]) <= 1e-12 * abs(l64[k]), k
    for a, b in ((gG64p, gG64), (gD64p, gD64)):
        for k in a:
            assert _rel(a[k], b[k]) <= 1e-12, k

    # float64 pinned to the FLOAT32 run's decisions vs the float32 run: rounding, on every gradient
    ref64b = ostep.HotPathRef(copy.deepcopy(netG), copy.deepcopy(netD), ostep.CodecCfg(), num_D=num_D, dtype=torch.float64)
    ostep.pin_activations(ref64b.netG, pins32, pins32.act_G)
    ostep.pin_activations(ref64b.netD, pins32, pins32.act_D)
    lq, gGq, gDq = _grads(ref64b, lr, hr, pins=pins32)
    worst_pinned = max([_rel(gG32[k], gGq[k]) for k in gGq if np.linalg.norm(gGq[k]) > 1e-12]
                       + [_rel(gD32[k], gDq[k]) for k in gDq if np.linalg.norm(gDq[k]) > 1e-12])
    worst_free = max([_rel(gG32[k], gG64[k]) for k in gG64 if np.linalg.norm(gG64[k]) > 1e-12]
                     + [_rel(gD32[k], gD64[k]) for k in gD64 if np.linalg.norm(gD64[k]) > 1e-12])
    print("%s: float32 vs float64, worst gradient rel-L2 unpinned %.2e, pinned %.2e" % (kind, worst_free, worst_pinned))
    # (biases in front of an InstanceNorm have an exactly-zero gradient: float32 leaves rounding residue there, skipped by the norm test)
    assert worst_pinned <= 2e-4, worst_pinned
    for k in l32:
        assert abs(l32[k] - lq[k]) <= 1e-5 * abs(lq[k]), k
