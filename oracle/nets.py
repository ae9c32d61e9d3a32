"""Oracle (test infrastructure): generator / discriminator stacks, LSGAN and
feature-matching losses, restated with stock torch CPU modules (float32 or
float64).  State-dict keys and shapes are identical to the reference so the
golden fixtures (and reference checkpoints) load unchanged.

* ``build_global_generator`` <- models/networks.py:301-357 (GlobalGenerator)
* ``LocalEnhancerRef``       <- models/networks.py:173-267 (LocalEnhancer)
* ``ResBlockRef``            <- models/networks.py:421-463 (ResnetBlock)
* ``ResConvDownRef``         <- models/networks.py:403-417 (ConvResBlock)
* ``NearestUpConvRef``       <- models/networks.py:375-400 (InterpolateUpsample)
* ``MultiscaleDRef``         <- models/networks.py:507-550, 641-692
* ``lsgan_loss``             <- models/networks.py:97-137 (GANLoss, use_lsgan)
* ``BotStackRef``            <- bottleneck_transformer_pytorch==0.1.4 (third party, not in
                                /root/reference; restated from the published MIT source,
                                parity UNPINNED), call sites networks.py:232-235, 341-344
* ``init_weights``           <- models/networks.py:13-19 (weights_init)

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn
import torch.nn.functional as F


def _inorm(c):
    return nn.InstanceNorm2d(c, affine=False)


class ResBlockRef(nn.Module):
    """x + [reflpad1, conv3, IN, ReLU, reflpad1, conv3, IN](x)."""

    def __init__(self, dim):
        super().__init__()
        self.conv_block = nn.Sequential(
            nn.ReflectionPad2d(1), nn.Conv2d(dim, dim, 3), _inorm(dim), nn.ReLU(),
            nn.ReflectionPad2d(1), nn.Conv2d(dim, dim, 3), _inorm(dim))

    def forward(self, x):
        return x + self.conv_block(x)


class ResConvDownRef(nn.Module):
    """--downsample_type resconv: conv1 (k, s, p; C->C), then conv2 5x5 + conv_res 3x3."""

    def __init__(self, cin, cout, kernel_size, stride, padding):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, cin, kernel_size, stride, padding)
        self.conv2 = nn.Conv2d(cin, cout, 5, padding=2)
        self.conv_res = nn.Conv2d(cin, cout, 3, 1, 1)

    def forward(self, x):
        x = self.conv1(x)
        return self.conv2(x) + self.conv_res(x)


class NearestUpConvRef(nn.Module):
    """--upsample_type interpolate: nearest x2; conv2(3x3,p2)(conv1(5x5,p1)(x)) + conv_res(x)."""

    def __init__(self, in_channels, out_channels, **_ignored):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.conv1 = nn.Conv2d(in_channels, out_channels, 5, padding=1)
        self.conv2 = nn.Conv2d(out_channels, out_channels, 3, padding=2)
        self.conv_res = nn.Conv2d(in_channels, out_channels, 3, padding=1)

    def forward(self, x):
        assert x.shape[1] == self.in_channels
        x = F.interpolate(x, scale_factor=2.0, mode="nearest")
        return self.conv2(self.conv1(x)) + self.conv_res(x)


# ---------------------------------------------------------------- BoT (unpinned)
class _AbsPosEmb(nn.Module):
    def __init__(self, fmap, dim_head):
        super().__init__()
        h, w = fmap
        s = dim_head ** -0.5
        self.height = nn.Parameter(torch.randn(h, dim_head) * s)
        self.width = nn.Parameter(torch.randn(w, dim_head) * s)

    def forward(self, q):  # q [b, heads, tokens, d]
        emb = (self.height[:, None, :] + self.width[None, :, :]).reshape(-1, q.shape[-1])
        return torch.einsum("bhid,jd->bhij", q, emb)


class _BotAttention(nn.Module):
    def __init__(self, dim, fmap, heads, dim_head):
        super().__init__()
        self.heads, self.scale = heads, dim_head ** -0.5
        self.to_qkv = nn.Conv2d(dim, heads * dim_head * 3, 1, bias=False)
        self.pos_emb = _AbsPosEmb(fmap, dim_head)

    def forward(self, x):
        b, _, h, w = x.shape
        q, k, v = self.to_qkv(x).chunk(3, dim=1)
        q, k, v = (t.reshape(b, self.heads, -1, h * w).transpose(-1, -2) for t in (q, k, v))
        q = q * self.scale
        sim = torch.einsum("bhid,bhjd->bhij", q, k) + self.pos_emb(q)
        out = torch.einsum("bhij,bhjd->bhid", sim.softmax(dim=-1), v)
        return out.transpose(-1, -2).reshape(b, -1, h, w)


class _BotBlock(nn.Module):
    def __init__(self, dim, fmap, dim_out, proj_factor, heads, dim_head):
        super().__init__()
        act = nn.ReLU()
        if dim != dim_out:
            self.shortcut = nn.Sequential(nn.Conv2d(dim, dim_out, 1, bias=False), nn.BatchNorm2d(dim_out), act)
        else:
            self.shortcut = nn.Identity()
        inner, attn_out = dim_out // proj_factor, heads * dim_head
        self.net = nn.Sequential(
            nn.Conv2d(dim, inner, 1, bias=False), nn.BatchNorm2d(inner), act,
            _BotAttention(inner, fmap, heads, dim_head), nn.Identity(),
            nn.BatchNorm2d(attn_out), act,
            nn.Conv2d(attn_out, dim_out, 1, bias=False), nn.BatchNorm2d(dim_out))
        nn.init.zeros_(self.net[-1].weight)
        self.out_act = nn.ReLU()      # a slot (no parameters, no state-dict key) so that oracle/step.py::pin_activations can pin it

    def forward(self, x):
        return self.out_act(self.net(x) + self.shortcut(x))


class BotStackRef(nn.Module):
    """BottleStack(downsample=False, rel_pos_emb=False) as the reference instantiates it."""

    def __init__(self, dim, fmap_size, dim_out, num_layers, proj_factor, heads, dim_head):
        super().__init__()
        self.dim, self.fmap_size = dim, tuple(fmap_size)
        self.net = nn.Sequential(*[
            _BotBlock(dim if i == 0 else dim_out, self.fmap_size, dim_out, proj_factor, heads, dim_head)
            for i in range(num_layers)])

    def forward(self, x):
        assert x.shape[1] == self.dim and tuple(x.shape[2:]) == self.fmap_size
        return self.net(x)


# ---------------------------------------------------------------- generators
def _down(kind, cin, cout):
    if kind == "conv":
        return nn.Conv2d(cin, cout, kernel_size=3, stride=2, padding=1)
    if kind == "resconv":
        return ResConvDownRef(cin, cout, kernel_size=3, stride=2, padding=1)
    raise NotImplementedError("downsample layer [%s] is not found" % kind)


def _up(kind, cin, cout):
    if kind == "transconv":
        return nn.ConvTranspose2d(cin, cout, kernel_size=3, stride=2, padding=1, output_padding=1)
    if kind == "interpolate":
        return NearestUpConvRef(in_channels=cin, out_channels=cout)
    raise NotImplementedError("upsample layer [%s] is not found" % kind)


def global_generator_layers(input_nc, output_nc, ngf=64, n_down=3, n_blocks=9, up="transconv", down="conv",
                            n_attn=0, input_size=(128, 256), proj_factor=4, heads=4, dim_head=128):
    """The flat layer list of GlobalGenerator.model (networks.py:308-352)."""
    layers = [nn.ReflectionPad2d(3), nn.Conv2d(input_nc, ngf, 7), _inorm(ngf), nn.ReLU()]
    for i in range(n_down):
        c = ngf * 2 ** i
        layers += [_down(down, c, 2 * c), _inorm(2 * c), nn.ReLU()]
    c = ngf * 2 ** n_down
    trunk = [ResBlockRef(c) for _ in range(n_blocks)]
    if n_attn > 0:
        fmap = tuple(s // 2 ** n_down for s in input_size)
        trunk.insert(n_blocks // 2, BotStackRef(c, fmap, c, n_attn, proj_factor, heads, dim_head))
    layers += trunk
    for i in range(n_down):
        c = ngf * 2 ** (n_down - i)
        layers += [_up(up, c, c // 2), _inorm(c // 2), nn.ReLU()]
    layers += [nn.ReflectionPad2d(3), nn.Conv2d(ngf, output_nc, 7), nn.Tanh()]
    return layers


class GlobalGeneratorRef(nn.Module):
    def __init__(self, *a, **kw):
        super().__init__()
        self.model = nn.Sequential(*global_generator_layers(*a, **kw))

    def forward(self, x):
        return self.model(x)


class LocalEnhancerRef(nn.Module):
    """networks.py:173-267 with n_local_enhancers == 1.  n_attn_l > 0 (networks.py:218-237): after the first
    n_blocks_local // 2 local blocks an 8x down-sampling Sequential whose second [conv, norm, ReLU] triple is the SAME
    three modules applied twice (a Python list multiplied by 2), a BottleStack on the /16 map, and -- appended after the
    remaining blocks -- ONE transposed convolution (2 ngf -> 2 ngf) with its norm and ReLU applied three times."""

    def __init__(self, input_nc, output_nc, ngf=32, n_down_global=3, n_blocks_global=9, n_blocks_local=3,
                 up="transconv", down="conv", n_attn_g=0, input_size=(128, 256), proj_factor_g=4, heads_g=4,
                 dim_head_g=128, n_attn_l=0, proj_factor_l=4, heads_l=4, dim_head_l=128):
        super().__init__()
        g = global_generator_layers(input_nc, output_nc, ngf * 2, n_down_global, n_blocks_global, up, down,
                                    n_attn_g, tuple(s // 2 for s in input_size), proj_factor_g, heads_g, dim_head_g)
        self.model = nn.Sequential(*g[:-3])
        self.model1_1 = nn.Sequential(nn.ReflectionPad2d(3), nn.Conv2d(input_nc, ngf, 7), _inorm(ngf), nn.ReLU(),
                                      _down(down, ngf, 2 * ngf), _inorm(2 * ngf), nn.ReLU())
        tail = [ResBlockRef(2 * ngf) for _ in range(n_blocks_local)]
        if n_attn_l > 0:
            middle = n_blocks_local // 2
            d = [_down(down, 2 * ngf, ngf), _inorm(ngf), nn.ReLU()]
            d += [_down(down, ngf, ngf), _inorm(ngf), nn.ReLU()] * 2
            tail.insert(middle, nn.Sequential(*d))
            fmap = tuple(x // 16 for x in input_size)
            tail.insert(middle + 1, BotStackRef(ngf, fmap, 2 * ngf, n_attn_l, proj_factor_l, heads_l, dim_head_l))
            tail += [_up(up, 2 * ngf, 2 * ngf), _inorm(ngf), nn.ReLU()] * 3
        tail += [_up(up, 2 * ngf, ngf), _inorm(ngf), nn.ReLU(),
                 nn.ReflectionPad2d(3), nn.Conv2d(ngf, output_nc, 7), nn.Tanh()]
        self.model1_2 = nn.Sequential(*tail)
        self.downsample = nn.AvgPool2d(3, stride=2, padding=[1, 1], count_include_pad=False)

    def forward(self, x):
        coarse = self.model(self.downsample(x))
        return self.model1_2(self.model1_1(x) + coarse)


class MultiscaleDRef(nn.Module):
    """num_D PatchGANs with getIntermFeat=True (networks.py:507-550, 641-692)."""

    def __init__(self, input_nc, ndf=64, n_layers=3, num_D=3, use_sigmoid=False, interm=True):
        """interm=False (--no_ganFeat_loss): one nn.Sequential per scale, `layer{i}`, Sigmoid included when use_sigmoid
        (--no_lsgan); with interm=True the reference never applies the Sigmoid (networks.py:684-689) -- neither does this."""
        super().__init__()
        self.num_D, self.n_layers, self.interm = num_D, n_layers, interm
        for i in range(num_D):
            stages = [nn.Sequential(nn.Conv2d(input_nc, ndf, 4, 2, 2), nn.LeakyReLU(0.2))]
            nf = ndf
            for _ in range(1, n_layers):
                prev, nf = nf, min(nf * 2, 512)
                stages.append(nn.Sequential(nn.Conv2d(prev, nf, 4, 2, 2), _inorm(nf), nn.LeakyReLU(0.2)))
            prev, nf = nf, min(nf * 2, 512)
            stages.append(nn.Sequential(nn.Conv2d(prev, nf, 4, 1, 2), _inorm(nf), nn.LeakyReLU(0.2)))
            stages.append(nn.Sequential(nn.Conv2d(nf, 1, 4, 1, 2)))
            if not interm:
                flat = [m for st in stages for m in st] + ([nn.Sigmoid()] if use_sigmoid else [])
                setattr(self, "layer%d" % i, nn.Sequential(*flat))
                continue
            for j, s in enumerate(stages):
                setattr(self, "scale%d_layer%d" % (i, j), s)
        self.downsample = nn.AvgPool2d(3, stride=2, padding=[1, 1], count_include_pad=False)

    def forward(self, x):
        result = []
        for i in range(self.num_D):
            if not self.interm:
                result.append([getattr(self, "layer%d" % (self.num_D - 1 - i))(x)])
                if i != self.num_D - 1:
                    x = self.downsample(x)
                continue
            feats, h = [], x
            for j in range(self.n_layers + 2):
                h = getattr(self, "scale%d_layer%d" % (self.num_D - 1 - i, j))(h)
                feats.append(h)
            result.append(feats)
            if i != self.num_D - 1:
                x = self.downsample(x)
        return result


def build_generator(netG, input_nc, output_nc, ngf, n_down_global, n_blocks_global, n_blocks_local=3,
                    up="transconv", down="conv", input_size=(128, 256), n_attn_g=0, proj_factor_g=4, heads_g=4,
                    dim_head_g=128, n_attn_l=0, proj_factor_l=4, heads_l=4, dim_head_l=128):
    """define_G for netG in {global, local} (networks.py:33-56), without weights_init."""
    if netG == "global":
        return GlobalGeneratorRef(input_nc, output_nc, ngf, n_down_global, n_blocks_global, up, down, n_attn_g,
                                  input_size, proj_factor_g, heads_g, dim_head_g)
    if netG == "local":
        return LocalEnhancerRef(input_nc, output_nc, ngf, n_down_global, n_blocks_global, n_blocks_local, up, down,
                                n_attn_g, input_size, proj_factor_g, heads_g, dim_head_g, n_attn_l, proj_factor_l, heads_l,
                                dim_head_l)
    raise NotImplementedError("generator not implemented!")


def init_weights(net, generator=None):
    """weights_init: every *Conv2d* weight ~ N(0, 0.02); BatchNorm2d weight ~ N(1, 0.02), bias 0."""
    for m in net.modules():
        name = m.__class__.__name__
        if "Conv2d" in name or isinstance(m, (nn.Conv2d, nn.ConvTranspose2d)):
            m.weight.data.normal_(0.0, 0.02, generator=generator)
        elif isinstance(m, nn.BatchNorm2d):
            m.weight.data.normal_(1.0, 0.02, generator=generator)
            m.bias.data.fill_(0)
    return net


def fill_deterministic(net, scale=0.02):
    """Closed-form parameter fill used by the golden fixtures (no weight files):
    p.flat[i] = scale_p * sin(0.37 * i + phase(key)), phase = crc32(key) mod 1000 / 100,
    biases included.  Depends only on the key string and the shape.
    """
    import zlib
    with torch.no_grad():
        for key, p in net.state_dict().items():
            if not p.dtype.is_floating_point:
                continue
            idx = torch.arange(p.numel(), dtype=torch.float64)
            val = torch.sin(0.37 * idx + (zlib.crc32(key.encode()) % 1000) / 100.0)
            if key.endswith("running_var"):
                val = 1.0 + 0.1 * val
            elif "running_mean" in key:
                val = 0.05 * val
            elif key.endswith(".weight") and p.dim() == 1:      # BN gamma
                val = 1.0 + scale * val
            elif "pos_emb" in key:
                val = 0.1 * val
            else:
                fan = p[0].numel() if p.dim() > 1 else 1
                val = val * (scale if p.dim() == 1 else min(0.25, 1.5 / math.sqrt(fan)))
            p.copy_(val.reshape(p.shape).to(p.dtype))
    return net


# ---------------------------------------------------------------- losses
def bce_gan_loss(preds, target_is_real: bool):
    """GANLoss(use_lsgan=False): sum over scales of BCELoss(pred[-1], const) (networks.py:106-109)."""
    t = 1.0 if target_is_real else 0.0
    return sum(F.binary_cross_entropy(p[-1], torch.full_like(p[-1], t)) for p in preds)


def lsgan_loss(preds, target_is_real: bool):
    """GANLoss(use_lsgan=True): sum over scales of mse(pred[-1], const)."""
    t = 1.0 if target_is_real else 0.0
    return sum(F.mse_loss(p[-1], torch.full_like(p[-1], t)) for p in preds)


def feature_matching_loss(pred_fake, pred_real, n_layers_D=3, num_D=2, lambda_feat=10.0, signs=None):
    """pix2pixHD_model.py:443-451.  signs (mask-pinned evaluation, oracle/step.py::MaskPins): sign(fake - real) per term, in this
    loop's order -- |d| is then evaluated as d * sign, the linear map the other evaluation's backward applied."""
    loss = 0
    fw, dw = 4.0 / (n_layers_D + 1), 1.0 / num_D
    n = 0
    for i in range(num_D):
        for j in range(len(pred_fake[i]) - 1):
            if signs is None:
                term = F.l1_loss(pred_fake[i][j], pred_real[i][j].detach())
            else:
                term = ((pred_fake[i][j] - pred_real[i][j].detach()) * signs[n].to(pred_fake[i][j].dtype)).mean()
            loss = loss + dw * fw * term * lambda_feat
            n += 1
    return loss
