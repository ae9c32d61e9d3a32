"""Oracle (test infrastructure): the model facade and the G/D optimisation step.

* ``HotPathRef.forward_losses`` <- models/pix2pixHD_model.py:394-451, 616 (forward + _forward)
* ``HotPathRef.train_step``     <- train.py:160-202 (fp32 branch)
* ``HotPathRef.inference``      <- models/pix2pixHD_model.py:618-638
* ``MaskPins`` / ``pin_activations`` -- mask-pinned backward (tests only): the same step with every ReLU / LeakyReLU
  mask, the sign of the L1 feature loss and the sign of |s| in the discriminator input taken from the implementation
  under test instead of from this evaluation's own values.  With the masks equal the two backward passes are the same
  linear map, so float32-vs-float64 gradients agree to rounding (1e-5) instead of to "whose masks flipped" (1e-2).

Torch CPU, float32 nets (float64 optional) and float64 transform, like the
reference.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg may import this module.
"""
from __future__ import annotations

import numpy as np
import torch

from . import nets, transform


class CodecCfg:
    """The spectral flags every BASELINE config carries (train.sh:9-10)."""

    def __init__(self, n_fft=512, hop=256, win=512, arcsinh_gain=1000.0, norm_range=(-1.0, 1.0),
                 src_range=(-5.0, 5.0), abs_norm=True, hr_rate=48000, lr_rate=12000, fit_residual=False,
                 arcsinh_transform=True, raw_mdct=False, abs_spectro=True):
        self.n_fft, self.hop, self.win = n_fft, hop, win
        self.codec = dict(arcsinh_transform=arcsinh_transform, raw_mdct=raw_mdct, arcsinh_gain=arcsinh_gain, abs_norm=abs_norm,
                          src_range=src_range, norm_range=norm_range)
        # pix2pixHD_model.py:400-405, 420-427: the (s, 2|s| + nr0) pair only with --abs_spectro AND --arcsinh_transform
        self.pair = bool(abs_spectro and arcsinh_transform)
        self.norm_range = norm_range
        self.up_ratio = hr_rate / lr_rate
        self.fit_residual = fit_residual
        self.window = transform.kbd_window(win)


class MaskPins:
    """Sign decisions captured from another evaluation of the same step (tests/test_fullsize_step_gpu.py).
      act_G / act_D : {name of the module in front of the activation (its last parameterised predecessor) -> bool tensor,
                       True where the activation's output is positive}.  act_D masks may be batch-stacked [fake, real]
                       (the implementation runs one discriminator pass over 2B samples): ``half`` picks the rows.
      l1_sign       : sign(fake_ij - real_ij) per feature-matching term, in pix2pixHD_model.py:443-451's loop order
      abs_sign      : sign(sr_spectro) (the |s| of the discriminator's second channel, pix2pixHD_model.py:420-424)"""

    def __init__(self, act_G=None, act_D=None, l1_sign=None, abs_sign=None, record=False):
        self.act_G, self.act_D = ({} if act_G is None else act_G), ({} if act_D is None else act_D)
        self.l1_sign, self.abs_sign = l1_sign, abs_sign
        self.half = None          # None | "fake" | "real": which rows of a stacked discriminator mask the running pass sees
        # record=True: the pinned slots evaluate the true activation and note THIS evaluation's decisions (one list entry per
        # call); finish_recording() turns them into the tables above (tests/test_oracle_pins.py: an evaluation pinned to its own
        # decisions must reproduce itself exactly)
        self.record = record

    def finish_recording(self):
        self.record = False
        for k, calls in self.act_G.items():
            assert len(calls) == 1, k
            self.act_G[k] = calls[0]
        for k, calls in self.act_D.items():       # forward_losses: D(fake.detach()), D(real), D(fake)
            assert len(calls) == 3 and torch.equal(calls[0], calls[2]), k
            self.act_D[k] = torch.cat((calls[0], calls[1]), dim=0)
        return self


class _PinnedAct(torch.nn.Module):
    """ReLU (slope 0) / LeakyReLU(slope) with the mask given: y = x * (m + slope * (1 - m)), a linear map of x."""

    def __init__(self, pins, table, key, slope):
        super().__init__()
        self.pins, self.table, self.key, self.slope = pins, table, key, float(slope)

    def forward(self, x):
        if self.pins.record:
            self.table.setdefault(self.key, []).append(x.detach() > 0)
            return torch.nn.functional.leaky_relu(x, self.slope) if self.slope else torch.relu(x)
        m = self.table[self.key]
        if m.shape[0] != x.shape[0]:
            assert m.shape[0] == 2 * x.shape[0] and self.pins.half in ("fake", "real"), (self.key, m.shape, x.shape)
            m = m[:x.shape[0]] if self.pins.half == "fake" else m[x.shape[0]:]
        assert m.shape == x.shape, (self.key, m.shape, x.shape)
        m = m.to(x.dtype)
        return x * (m + self.slope * (1.0 - m)) if self.slope else x * m


def pin_activations(net, pins, table):
    """Replace every ReLU / LeakyReLU slot of every nn.Sequential in ``net`` (and the ``out_act`` of the bottleneck-transformer
    blocks) by a _PinnedAct keyed by the name of the last module with parameters in front of it.  Returns the keys used."""
    used = []
    for prefix, seq in list(net.named_modules()):
        if isinstance(seq, nets._BotBlock):
            key = (prefix + "." if prefix else "") + "net.8"
            seq.out_act = _PinnedAct(pins, table, key, 0.0)
            used.append(key)
        if not isinstance(seq, torch.nn.Sequential):
            continue
        last = None
        for idx, child in enumerate(list(seq)):
            name = (prefix + "." if prefix else "") + str(idx)
            if isinstance(child, (torch.nn.ReLU, torch.nn.LeakyReLU, _PinnedAct)):
                if isinstance(child, _PinnedAct):
                    continue
                assert last is not None, "activation without a producer in front: " + name
                slope = child.negative_slope if isinstance(child, torch.nn.LeakyReLU) else 0.0
                seq[idx] = _PinnedAct(pins, table, last, slope)
                used.append(last)
            elif any(True for _ in child.parameters()):
                last = name
    missing = [k for k in used if k not in table]
    assert pins.record or not missing, "no captured mask for %r" % missing[:8]
    return used


class HotPathRef:
    def __init__(self, netG, netD, cfg: CodecCfg, n_layers_D=3, num_D=2, lambda_feat=10.0, lr=2e-4, beta1=0.5,
                 dtype=torch.float32, use_lsgan=True, feat_loss=True):
        self.gan_loss = nets.lsgan_loss if use_lsgan else nets.bce_gan_loss       # --no_lsgan
        self.feat_loss = feat_loss                                                # --no_ganFeat_loss
        self.netG, self.netD, self.cfg = netG.to(dtype), (netD.to(dtype) if netD is not None else None), cfg
        self.n_layers_D, self.num_D, self.lambda_feat, self.dtype = n_layers_D, num_D, lambda_feat, dtype
        if netD is not None:
            self.opt_G = torch.optim.Adam(self.netG.parameters(), lr=lr, betas=(beta1, 0.999))
            self.opt_D = torch.optim.Adam(self.netD.parameters(), lr=lr, betas=(beta1, 0.999))

    # -- codec ---------------------------------------------------------
    def spectro(self, audio):
        c = self.cfg
        s, norm = transform.to_spectro(np.asarray(audio), c.window, c.n_fft, c.hop, **c.codec)
        return torch.from_numpy(s).to(self.dtype), norm

    def two_channel(self, s, sign=None):
        """sign (mask-pinned evaluation): |s| as s * sign with the sign given."""
        if not getattr(self.cfg, "pair", True):
            return s                                                        # pix2pixHD_model.py:404, 426-427
        mag = s.abs() if sign is None else s * sign.to(s.dtype)
        return torch.cat((s, mag * 2 + self.cfg.norm_range[0]), dim=1)      # pix2pixHD_model.py:400-402

    # -- forward + losses ----------------------------------------------
    def forward_losses(self, lr_audio, hr_audio, pins=None):
        """pins (MaskPins, after pin_activations on both nets): the mask-pinned evaluation."""
        lr_s, _ = self.spectro(lr_audio)
        hr_s, _ = self.spectro(hr_audio)
        sr_s = self.netG(self.two_channel(lr_s))
        if self.cfg.fit_residual:
            sr_s = sr_s + lr_s
        recording = pins is not None and pins.record
        if recording:
            pins.abs_sign = torch.sign(sr_s.detach())
        sr_in = self.two_channel(sr_s, pins.abs_sign if (pins is not None and not recording) else None)
        hr_in = self.two_channel(hr_s)

        def half(which):
            if pins is not None:
                pins.half = which
        half("fake")
        pred_fake_pool = self.netD(torch.cat((lr_s, sr_in.detach()), dim=1))
        loss_D_fake = self.gan_loss(pred_fake_pool, False)
        half("real")
        pred_real = self.netD(torch.cat((lr_s, hr_in), dim=1))
        loss_D_real = self.gan_loss(pred_real, True)
        half("fake")
        pred_fake = self.netD(torch.cat((lr_s, sr_in), dim=1))
        half(None)
        loss_G_GAN = self.gan_loss(pred_fake, True)
        losses = {"G_GAN": loss_G_GAN, "D_real": loss_D_real, "D_fake": loss_D_fake}
        if self.feat_loss:
            if recording:
                pins.l1_sign = [torch.sign(f.detach() - r.detach()) for fs, rs in zip(pred_fake, pred_real) for f, r in zip(fs[:-1], rs[:-1])]
            losses["G_GAN_Feat"] = nets.feature_matching_loss(pred_fake, pred_real, self.n_layers_D, self.num_D, self.lambda_feat,
                                                              signs=pins.l1_sign if (pins is not None and not recording) else None)
        return losses, sr_s

    def train_step(self, lr_audio, hr_audio, amp=False, scaler=None, grads_out=None):
        """One train.py:160-202 iteration.  amp=False: the fp32 branch.  amp=True: the --fp16 branch -- the forward
        under torch.autocast(float16) (train.py:161-164; on the CPU here, same op lists for conv / norm / losses) and
        one GradScaler (torch.amp.GradScaler("cpu")): scale(loss).backward(), scaler.step(opt) for G then D,
        scaler.update() once (train.py:183-199).  Returns the loss dict (floats).  grads_out (a dict, tests only): receives
        "G" / "D" -> {name: gradient} as each optimiser saw them (after the GradScaler's unscale)."""
        if amp:
            with torch.autocast("cpu", dtype=torch.float16):
                losses, _ = self.forward_losses(lr_audio, hr_audio)
        else:
            losses, _ = self.forward_losses(lr_audio, hr_audio)
        loss_D = (losses["D_fake"] + losses["D_real"]) * 0.5
        loss_G = losses["G_GAN"] + losses.get("G_GAN_Feat", 0)
        self.opt_G.zero_grad()
        if amp:
            scaler.scale(loss_G).backward()
            scaler.step(self.opt_G)
        else:
            loss_G.backward()
            self.opt_G.step()
        if grads_out is not None:
            grads_out["G"] = {k: p.grad.detach().numpy().copy() for k, p in self.netG.named_parameters()}
        self.opt_D.zero_grad()
        if amp:
            scaler.scale(loss_D).backward()
            scaler.step(self.opt_D)
            scaler.update()
        else:
            loss_D.backward()
            self.opt_D.step()
        if grads_out is not None:
            grads_out["D"] = {k: p.grad.detach().numpy().copy() for k, p in self.netD.named_parameters()}
        return {k: float(v.detach()) for k, v in losses.items()}

    # -- inference -----------------------------------------------------
    @torch.no_grad()
    def inference(self, lr_audio):
        c = self.cfg
        lr_s, norm = self.spectro(lr_audio)
        sr_s = self.netG(self.two_channel(lr_s))
        if c.fit_residual:  # pix2pixHD_model.py:631-635
            lr_part = int(sr_s.size(-1) / c.up_ratio)
            sr_s[..., :lr_part] *= 1e-3
            sr_s = sr_s + lr_s
        audio = transform.to_audio(sr_s.numpy(), norm, c.window, c.n_fft, c.hop, **c.codec)
        return sr_s, audio, norm, lr_s
