"""Autograd plumbing over the HIP launchers (ops.py).  Every tensor that crosses these functions is a
logical NCHW tensor in torch.channels_last memory format -- i.e. NHWC in HBM, which is the layout the
kernels consume -- so chaining them never copies.  Convolution weights ([Co, Ci, KH, KW] /
[Cin_T, Cout_T, KH, KW]) are channels_last as well, which is exactly the OHWI layout of the C ABI.

Weight / bias gradients are accumulated by the kernels straight into ``param.grad`` (no AccumulateGrad
add pass, no per-step zero fill): the Function returns None for those inputs.  ``GradSlot`` tracks whether
a parameter's gradient buffer already holds this step's first contribution.
"""
from __future__ import annotations

import os

import torch

from . import _lib, amp, ops
from ._lib import ACT_LRELU02, ACT_NONE, ACT_RELU, ACT_TANH  # noqa: F401

CL = torch.channels_last


def to_cl(x: torch.Tensor) -> torch.Tensor:
    """float32 channels_last (NHWC memory) without copying when it already is."""
    if x.dtype != torch.float32:
        x = x.float()
    return x.contiguous(memory_format=CL)


def nhwc_view(x):
    return x.permute(0, 2, 3, 1)


def nchw_view(y):
    return y.permute(0, 3, 1, 2)


# ------------------------------------------------------------------------------------------------
# sign-decision capture (mask-pinned gradient parity, tests/test_fullsize_step_gpu.py)
# ------------------------------------------------------------------------------------------------
# Every non-smooth point of the step -- ReLU / LeakyReLU masks, sign(fake - real) of the L1 feature loss, sign(s) of the
# discriminator input's |s| channel -- as the forward pass decided it.  TAP = [] switches the capture on: the networks append
# (module in front of the activation, output > 0), the two Functions below their signs.  The oracle then evaluates its float64
# backward on these very decisions (oracle/step.py::MaskPins).  None (always, outside that test): no launch, no copy.
TAP = None


def tap(key, y):
    if TAP is not None:
        TAP.append((key, y.detach() > 0))


# ------------------------------------------------------------------------------------------------
# fused gradient accumulation
# ------------------------------------------------------------------------------------------------
# --fp16: a kernel that produces an activation (InstanceNorm forward) or an activation gradient (InstanceNorm backward) can
# write the float16 copy the next autocast convolution stages its operand from, instead of that convolution running a cast
# pass of its own (68 launches / 0.47 ms per configs[2] step).  The copy travels as an attribute of the tensor it mirrors and
# is only trusted while that tensor is the very one it was made from (storage address, version counter, size).
# MG_NO_H16_PRODUCER=1 turns the mechanism off (same results: the cast rounds the same float32 values).
H16_STATS = {"made": 0, "used_fwd": 0, "used_bwd": 0}


def _want_h16(B, HW, C, half=None):
    """Worth writing a float16 copy beside a [B, HW, C] float32 tensor?  Autocast only (half: the forward's precision -- the
    backward pass runs outside the autocast context); channel counts the float16 implicit GEMMs take (C % 64 == 0);
    activation-dominated sizes (a tensor whose consumer streams weights, pixels <= 2 C, is im2col'ed from float32 by
    conv_h16.h instead)."""
    if half is None:
        half = amp.current_precision() == _lib.PRECISION_F16
    # (round 6: also every SMALL tensor, <= 2 Mi elements -- 2 more bytes per element in a kernel that sits at the launch floor cost
    # nothing, and the stride-2 / transposed rungs next to the trunk then skip their cast launch)
    return (bool(half) and C % 64 == 0 and (B * HW > 2 * C or B * HW * C <= (1 << 21))
            and os.environ.get("MG_NO_H16_PRODUCER", "0") != "1")


def _attach_h16(t, buf):
    if buf is not None:
        t._mg_h16 = (buf, t.data_ptr(), t._version, t.numel())
        H16_STATS["made"] += 1
    return t


def _h16_of(t):
    rec = getattr(t, "_mg_h16", None)
    if rec is None:
        return None
    buf, ptr, ver, n = rec
    if t.data_ptr() != ptr or t._version != ver or t.numel() != n or not t.is_contiguous(memory_format=CL):
        return None
    return buf


# ResnetBlock chains conv3x3 -> InstanceNorm -> (ReLU) -> conv3x3 on the trunk's small maps: the fused output-transform + norm kernel
# of one layer can write the NEXT layer's Winograd input image B^T y B (ops.conv_fwd_instnorm(v_next=...)).  Like the float16 copies
# above, the image travels as an attribute of the tensor it was made from and is only trusted while that tensor is the very one it
# was made from and the consumer's geometry is the one the producer was told (3x3, stride 1, pad 1, same padding mode).
WINO_NEXT_STATS = {"made": 0, "used": 0}


def _attach_wino_v(t, v, key):
    t._mg_wino_v = (v, t.data_ptr(), t._version, key)
    WINO_NEXT_STATS["made"] += 1
    return t


def _wino_v_of(t, key):
    rec = getattr(t, "_mg_wino_v", None)
    if rec is None:
        return None
    v, ptr, ver, k = rec
    if t.data_ptr() != ptr or t._version != ver or k != key or not t.is_contiguous(memory_format=CL):
        return None
    return v


def grad_buffer(p: torch.nn.Parameter):
    """(buffer, accumulate?) for a parameter: allocates p.grad with p's own strides on first use; a buffer that
    was handed out fresh (after zero_grad) is overwritten by the first kernel and accumulated into afterwards."""
    fresh = getattr(p, "_mg_fresh", True)
    if p.grad is None:
        p.grad = torch.empty_like(p)          # preserve_format: same (channels_last) strides as the parameter
        fresh = True
    p._mg_fresh = False
    p._mg_inf_checked = False             # whoever writes next has to say so again (see _producer_flag)
    return p.grad, (not fresh)


def grad_buffer16(p: torch.nn.Parameter):
    """(float16 buffer, accumulate?) of a parameter whose gradient is STORED as float16 (FusedAdam GRAD_F16, --fp16): the same
    fresh / accumulate bookkeeping as grad_buffer; p.grad (the float32 arena view) is not written -- read gradients through grad_of."""
    fresh = getattr(p, "_mg_fresh", True)
    p._mg_fresh = False
    p._mg_inf_checked = False
    return p._mg_g16, (not fresh)


def grad_of(p: torch.nn.Parameter):
    """The gradient the last backward pass left for p, as a float32 tensor shaped like p: p.grad, or -- where the weight-gradient
    kernel stores float16 (the reference's dtype for an autocast layer's gradient) -- that buffer widened."""
    opt = getattr(p, "_mg_opt", None)
    if opt is not None and getattr(p, "_mg_g16", None) is not None:
        return opt.grad_of(p)
    return p.grad


def _tag_g16(g, weight, bias, weight_grad):
    """Remember (before the optimiser lays out its arenas) that this weight's gradient kernel can store float16."""
    if (g.precision == _lib.PRECISION_F16 and weight_grad is True and weight.requires_grad
            and not hasattr(weight, "_mg_g16_ok")):
        weight._mg_g16_ok = bool(ops.wgrad_h16_ok(g))
        opt = getattr(weight, "_mg_opt", None)
        if weight._mg_g16_ok and opt is not None:
            opt.adopt_g16(weight)         # the arenas were laid out before this first forward pass (ddp.attach)


def mark_fresh(params):
    """zero_grad without a memset: the next wgrad kernel overwrites instead of accumulating."""
    for p in params:
        p._mg_fresh = True
        p._mg_inf_checked = False
        p._mg_fused_stamp = None        # a backward pass that was never followed by optimizer.step() must not poison the next one


# A bias added right before InstanceNorm2d(affine=False) is removed again by the mean subtraction: its gradient is
# identically zero (the reference's autograd produces float32 rounding residue there, ~1e-8 of the layer's gradient
# scale, which no two implementations reproduce).  Such parameters get an exactly-zero gradient without a launch.
# MDCTGAN_DEAD_BIAS_GRADS=1 computes the column sums anyway.
COMPUTE_DEAD_BIAS_GRADS = os.environ.get("MDCTGAN_DEAD_BIAS_GRADS", "0") == "1"


def mark_bias_feeds_norm(bias):
    if bias is not None:
        bias._mg_zero_grad = True


def _zero_grad_bias(bias):
    if bias.grad is None:
        bias.grad = torch.zeros_like(bias)
    elif getattr(bias, "_mg_fresh", True) and not getattr(bias, "_mg_known_zero", False):
        bias.grad.zero_()                 # a buffer someone else may have written: clear it once
    bias._mg_known_zero = True
    bias._mg_fresh = False


def _producer_flag(weight, g):
    """--fp16: the GradScaler's found_inf slot of the optimiser that owns `weight`, when this layer's weight-gradient kernel
    checks its own results (ops.wgrad_checks_finite) -- FusedAdam.step then leaves the gradient out of its check pass.  The
    kernel looks at the gradient AFTER accumulation, so the last writer's check covers the whole buffer; every writer of a
    weight gradient passes through here and restates `_mg_inf_checked`."""
    opt = getattr(weight, "_mg_opt", None)
    if opt is None:
        return None
    flag = opt.producer_flag()
    if flag is None or not ops.wgrad_checks_finite(g):
        return None
    return flag


_grad_hooks = []


def register_grad_ready_hook(fn):
    """fn(param) is called right after a kernel finished writing param.grad (used by the DDP reducer)."""
    _grad_hooks.append(fn)
    return fn


def remove_grad_ready_hook(fn):
    if fn in _grad_hooks:
        _grad_hooks.remove(fn)


def _notify(p):
    for fn in _grad_hooks:
        fn(p)


# ------------------------------------------------------------------------------------------------
# transformed-weight cache for inference
# ------------------------------------------------------------------------------------------------
# Under torch.no_grad() the weights are frozen, so a Winograd layer's U = G w G^T is computed once and kept on the
# parameter instead of once per call (18 x 22 us per batch on configs[4]).  The kernels update parameters through raw
# pointers, which torch's version counter does not see: everything that writes weights that way (FusedAdam.step, a
# replay of the captured training step) bumps WEIGHT_EPOCH, and a cached image is only valid for the (version, epoch,
# storage) it was made from.
WEIGHT_EPOCH = [0]


def bump_weight_epoch():
    WEIGHT_EPOCH[0] += 1


def _cached_wino_weights(g, weight, fill=True):
    """fill=False (graph capture): a valid cached image is used (it was made by an eager warm-up call and outlives the
    graph), a miss returns None -- a tensor allocated from the capturing graph's pool must not be cached."""
    w = weight.detach()
    key = (weight._version, WEIGHT_EPOCH[0], w.data_ptr(), g.Ci, g.Co, g.KH, g.stride, g.precision,
           ops.wino_weights_bytes(g))
    hit = getattr(weight, "_mg_u_cache", None)
    if hit is not None and hit[0] == key:
        return hit[1]
    if not fill:
        return None
    u = ops.wino_weights(g, w)
    weight._mg_u_cache = (key, u)
    return u


# ------------------------------------------------------------------------------------------------
# weight-side fusion: gradient inverse transform + Adam + next forward transform in the weight-gradient call
# ------------------------------------------------------------------------------------------------
# For the Winograd F(2x2,3x3) trunk layers (18 x 9.4 M weights on configs[1]) the weight gradient never goes to HBM: the
# call that would write it applies Adam to (w, m, v) and refreshes the layer's transformed weights U for the next iteration
# (include/mdctgan_hip.h: mg_conv_wgrad_adam_w; 467 -> 354 MB of traffic and 3 -> 1 launches per layer).  Valid only where
# FusedAdam.can_fuse() says so (one process, float32), only inside fused_adam_scope -- Pix2PixHDModel.optimize_parameters
# opens it around loss_G.backward(), where exactly one optimiser step follows the backward pass -- and only for weights with
# a single gradient contribution per step.  Everywhere else the three separate kernels run, with the same bits.
class _FusedAdamScope:
    opt = None


class fused_adam_scope:
    def __init__(self, opt):
        self.opt = opt if (opt is not None and opt.can_fuse()) else None

    def __enter__(self):
        self.prev, _FusedAdamScope.opt = _FusedAdamScope.opt, self.opt
        return self

    def __exit__(self, *exc):
        _FusedAdamScope.opt = self.prev
        return False


def _fusable(g, weight, weight_grad):
    """The layer keeps a persistent transformed-weight image that mg_conv_wgrad_adam_w refreshes."""
    opt = getattr(weight, "_mg_opt", None)
    return (opt is not None and weight_grad is True and weight.requires_grad and int(getattr(weight, "_mg_writes", 1)) == 1
            and opt.can_fuse() and ops.wgrad_adam_ok(g))


def _persistent_u(g, weight):
    """U = G w G^T in a buffer that lives with the parameter: recomputed only when the weights changed behind its back."""
    u = getattr(weight, "_mg_u_persist", None)
    nbytes = ops.wino_weights_bytes(g)
    if u is None or u.numel() * 4 != nbytes or u.device != weight.device:
        if torch.cuda.is_current_stream_capturing():
            return None
        u = weight._mg_u_persist = torch.empty(nbytes // 4, dtype=torch.float32, device=weight.device)
        weight._mg_u_ok = None
    if weight._mg_u_ok != weight._version:
        from . import _lib
        _lib.check(_lib.load().mg_conv_wino_prepare(g, _lib.ptr(weight.detach()), _lib.ptr(u), _lib.stream()), "mg_conv_wino_prepare")
        weight._mg_u_ok = weight._version
    return u


def _wgrad_fused(ctx_u, g, weight, x, gy, v, md):
    """The fused call when the scope, the layer and the image allow it; returns False to fall back to the separate kernels."""
    opt = _FusedAdamScope.opt
    if opt is None or getattr(weight, "_mg_opt", None) is not opt or not getattr(weight, "_mg_fresh", True):
        return False
    u = getattr(weight, "_mg_u_persist", None)
    if u is None or ctx_u is None or ctx_u.data_ptr() != u.data_ptr() or weight._mg_u_ok != weight._version:
        return False
    # One Adam update per weight and optimiser step: a module applied twice in one forward whose weight was not annotated
    # with _mg_writes = 2 would otherwise be stepped twice from the same bias-correction terms, with U rewritten between the
    # two data gradients.  opt._step is the host's count of optimizer.step() calls.
    stamp = (id(opt), opt._step)      # (a rebuilt optimiser restarts _step at 0: the identity keeps its first step from matching a stale stamp)
    if getattr(weight, "_mg_fused_stamp", None) == stamp:
        raise RuntimeError("a weight whose gradient, Adam update and transform are fused received a second gradient "
                           "contribution in one backward pass: set weight._mg_writes = <uses per forward> on shared modules")
    weight._mg_fused_stamp = stamp
    grp = opt.param_groups[0]
    (b1, b2), eps = grp["betas"], grp["eps"]
    opt.sync_lr()
    ops.conv_wgrad_adam(g, x, gy, weight.detach(), weight._mg_m, weight._mg_v, u, opt.state, b1, b2, eps, opt.grad_scale, v=v, md=md)
    # The gradient buffer stays "fresh": optimizer.step() skips this weight; u now matches the new weights.  weight.grad
    # was NOT written this step -- it holds whatever an earlier unfused step left there; gradient readers (a norm logger)
    # must run under MG_NO_WINO_ADAM_FUSION=1.
    return True


def _weight_image(g, weight, weight_grad=None):
    """Weight operand image of a training-step convolution: the Winograd transform / float16 copy made by
    ops.wino_weights, or -- for layers whose image is the plain float16 copy -- the slice of the optimiser's float16
    shadow arena (FusedAdam(half_shadow=True)), which the Adam kernel keeps current."""
    if weight_grad is not None and _fusable(g, weight, weight_grad):
        u = _persistent_u(g, weight)
        if u is not None:
            return u
    h = getattr(weight, "_mg_h", None)
    if h is not None and ops.weights_are_casts(g):
        if weight._version == weight._mg_h_version:
            return h
        if not torch.cuda.is_current_stream_capturing():       # the parameter was rewritten by a torch op: re-sync
            h.copy_(weight._mg_flat)
            weight._mg_h_version = weight._version
            return h
    return ops.wino_weights(g, weight.detach())


# ------------------------------------------------------------------------------------------------
# one discriminator forward, two backward passes
# ------------------------------------------------------------------------------------------------
# train.py:160-202 evaluates D three times per iteration: D(fake.detach()) and D(real) for the D loss, D(fake) for the
# G loss.  The two fake passes see the same weights and the same input, so their activations are identical: the model
# runs ONE forward over the batch-stacked [fake, real] input (weight_grad="D0") and backpropagates twice through it --
#   * pass "G" (loss_G): data gradients only, and only for the first `rows` samples (the fake half; every discriminator
#     layer is per-sample).  Gradient buffers keep the stacked shape autograd expects; their second half is never
#     written and never read.
#   * pass "D" (loss_D): weight gradients + data gradients over the whole stack, nothing flows back into the generator.
class _BackwardPass:
    kind, rows = None, 0


class backward_pass:
    """with backward_pass("G", B): loss_G.backward(retain_graph=True) / with backward_pass("D", B): loss_D.backward(...)"""

    def __init__(self, kind, rows):
        self.kind, self.rows = kind, int(rows)

    def __enter__(self):
        self.prev = (_BackwardPass.kind, _BackwardPass.rows)
        _BackwardPass.kind, _BackwardPass.rows = self.kind, self.rows
        return self

    def __exit__(self, *exc):
        _BackwardPass.kind, _BackwardPass.rows = self.prev
        return False


def _live_rows(t):
    """Samples of a stacked tensor that carry gradient in the running pass (None: all of them)."""
    if _BackwardPass.kind == "G" and 0 < _BackwardPass.rows < t.shape[0]:
        return _BackwardPass.rows
    return None


def _is_shared(weight_grad):
    return weight_grad in ("D", "D0")


# ------------------------------------------------------------------------------------------------
# a second gradient for a tensor with two consumers
# ------------------------------------------------------------------------------------------------
# A discriminator feature map feeds the next layer AND the feature-matching loss (pix2pixHD_model.py:443-451): autograd would add the
# two gradients with an elementwise launch per map (8 / 12 per configs[1] / configs[2] step, up to 100 MB each).  The node that
# PRODUCED the map (conv + LeakyReLU, or InstanceNorm + LeakyReLU) hangs an ExtraGrad on its output; the loss node
# (_L1HalvesSumFn) parks its gradient there instead of returning it, and the producer's backward -- which runs later in the same
# pass, reached through the map's other consumer -- reads dy + extra inside its own kernel (mg_instnorm_bwd_add / mg_act_bwd_add:
# the same float32 addition).  Valid because the producer's backward always runs after the loss's in a pass that contains both
# (topological order), and the map's other consumer always carries gradient in such a pass (the GAN term behind the last layer).
class ExtraGrad:
    __slots__ = ("g",)

    def __init__(self):
        self.g = None


def _hang_extra(ctx, out):
    ctx.extra = None
    if any(ctx.needs_input_grad) and os.environ.get("MG_NO_EXTRA_GRAD", "0") != "1":      # (grad mode reads False inside a forward)
        ctx.extra = out._mg_extra = ExtraGrad()
    return out


def _take_extra(ctx, like):
    """The parked second gradient of this node's output (None when nobody parked one), as a tensor shaped like `like`."""
    h = getattr(ctx, "extra", None)
    if h is None or h.g is None:
        return None
    g, h.g = h.g, None
    assert g.shape == like.shape
    return g


# ------------------------------------------------------------------------------------------------
# convolution
# ------------------------------------------------------------------------------------------------
class _ConvFn(torch.autograd.Function):
    """y = act(conv(x, w) + b); transposed=True runs the data-gradient kernel forward (ConvTranspose2d)."""

    @staticmethod
    def forward(ctx, x, weight, bias, cfg):
        x = to_cl(x)
        stride, pad, reflect, act, transposed, weight_grad = cfg
        B, _, H, W = x.shape
        w = weight.detach()
        assert w.is_contiguous(memory_format=CL), "conv weights must be channels_last (OHWI) tensors"
        KH, KW = w.shape[2], w.shape[3]
        b = bias.detach() if bias is not None else None
        if not transposed:
            Co, Ci = w.shape[0], w.shape[1]
            assert Ci == x.shape[1]
            g = ops.conv_geom(B, H, W, Ci, Co, KH, KW, stride, pad, reflect, amp.current_precision())
            _tag_g16(g, weight, bias, weight_grad)
            # Winograd layers: transform the weights once, reuse the image for the data gradient of this step
            if ctx.needs_input_grad[0]:
                u = _weight_image(g, weight, weight_grad)
            elif x.is_cuda and not torch.is_grad_enabled():
                # inference: once per weight version (a captured inference graph reads the images its warm-up made)
                u = _cached_wino_weights(g, weight, fill=not torch.cuda.is_current_stream_capturing())
            else:
                u = None
            # ... and keep B^T x B for the weight gradient (A dy A^T is shared between dgrad and wgrad in backward)
            v, v_filled = None, False
            x16 = _h16_of(x) if ops.precast_ok(0, g) else None
            if x16 is not None:                  # x's producer already wrote float16(x): the call skips its cast pass
                v, v_filled = x16, True
                H16_STATS["used_fwd"] += 1
            elif u is not None and weight_grad and weight.requires_grad:
                v, _ = ops.wino_tile_buffers(g, x.device, want_md=False)
            y = ops.conv_fwd(g, nhwc_view(x), w, b, act, u, v, v_filled)
            ctx.u, ctx.v = u, (v if (weight_grad and weight.requires_grad) else None)
        else:
            # nn.ConvTranspose2d(k, stride, pad, output_padding = stride - 1): the data gradient of the conv
            # high-res [B, sH, sW, Cout_T] -> low-res [B, H, W, Cin_T]
            cin_t, cout_t = w.shape[0], w.shape[1]
            assert cin_t == x.shape[1]
            g = ops.conv_geom(B, stride * H, stride * W, cout_t, cin_t, KH, KW, stride, pad, False,
                              amp.current_precision())
            assert (g.OH, g.OW) == (H, W), "unsupported ConvTranspose2d geometry"
            # the weight image (Winograd U / float16 copy) serves this call and the backward's data gradient
            u = _weight_image(g, weight) if (ctx.needs_input_grad[0] and x.is_cuda) else None
            x16 = _h16_of(x) if ops.precast_ok(1, g) else None       # x plays dy in this call
            if x16 is not None:
                H16_STATS["used_fwd"] += 1
            y = ops.conv_dgrad(g, nhwc_view(x), w, b, act, u=u, md_out=x16, md_filled=x16 is not None)
            ctx.u, ctx.v = u, (x16 if (weight_grad and weight.requires_grad) else None)
        y = nchw_view(y)
        ctx.g, ctx.cfg = g, cfg
        ctx.weight, ctx.bias = weight, bias
        ctx.save_for_backward(x, y if act != ACT_NONE else None)
        return _hang_extra(ctx, y) if act in (ACT_RELU, ACT_LRELU02) else y

    @staticmethod
    def backward(ctx, gy):
        x, y = ctx.saved_tensors
        return _conv_backward(ctx, gy, x, y), None, None, None


def _conv_backward(ctx, gy, x, y, add=None):
    """Data and weight gradient of a _ConvFn / _ConvInstNormFn node; returns dx (weight gradients go to the arena).
    add (NHWC view, stride-1 convolutions without live rows only): dx += add inside the data gradient's last kernel where the
    layer's path has one that takes it (mg_wino_tiles.add: Winograd gather, reflection fold, split-K epilogue), by one add
    launch of the library otherwise."""
    stride, pad, reflect, act, transposed, weight_grad = ctx.cfg
    assert add is None or not transposed
    g, weight, bias = ctx.g, ctx.weight, ctx.bias
    gy = to_cl(gy)
    kind = _BackwardPass.kind
    shared = _is_shared(weight_grad)
    last_use = not shared or kind != "G"                   # a shared layer is walked again by the D pass
    want_dx = ctx.needs_input_grad[0] and not (weight_grad == "D0" and kind == "D")
    want_dw = bool(weight_grad) and weight.requires_grad and not (shared and kind == "G")
    rows = _live_rows(x) if shared else None
    x_full, gy_full = x, gy
    rec16 = getattr(gy, "_mg_h16_rows", None) if rows is not None else None
    if rows is not None:                                   # pass "G" over a stacked batch: the fake half only
        assert not transposed and not want_dw
        x, gy = x[:rows], gy[:rows]
        y = y[:rows] if y is not None else None
        g = ops.conv_geom(rows, g.H, g.W, g.Ci, g.Co, g.KH, g.KW, g.stride, g.pad, g.reflect, g.precision)
    # float16(gy) written by gy's producer (an InstanceNorm backward): only for the whole, untouched tensor
    g16 = _h16_of(gy) if (rows is None and act == ACT_NONE) else None
    if act != ACT_NONE:
        extra = _take_extra(ctx, gy_full)          # the feature-matching loss's gradient of this layer's output (see ExtraGrad)
        if extra is not None and rows is not None:
            extra = extra[:rows]
        gy = nchw_view(ops.act_bwd(nhwc_view(gy), nhwc_view(y), act, dy2=nhwc_view(to_cl(extra)) if extra is not None else None))
    w = weight.detach()
    dx = None
    md = None
    md_filled = False
    u_used = getattr(ctx, "u", None)
    if want_dx:
        if not transposed:
            u = getattr(ctx, "u", None)
            if g16 is not None and ops.precast_ok(1, g):
                md, md_filled = g16, True
                H16_STATS["used_bwd"] += 1
            elif u is not None and want_dw and (getattr(ctx, "v", None) is not None or ops.tiles_are_casts(g)):
                _, md = ops.wino_tile_buffers(g, gy.device, want_v=False)
            if rows is not None:
                assert add is None
                if u is not None and ops.wino_weights_bytes(g) != ops.wino_weights_bytes(ctx.g):
                    u = None          # the weight image was made for the stacked batch's plan; half the rows take another path
                dx = torch.empty_like(x_full)
                g16r = None
                if rec16 is not None and act == ACT_NONE and ops.precast_ok(1, g):
                    buf, ptr, ver, r_ = rec16
                    if ptr == gy_full.data_ptr() and ver == gy_full._version and r_ == rows and gy_full.is_contiguous(memory_format=CL):
                        g16r = buf
                        H16_STATS["used_bwd"] += 1
                ops.conv_dgrad(g, nhwc_view(gy), w, u=u, out=nhwc_view(dx[:rows]), md_out=g16r, md_filled=g16r is not None)
            else:
                dx = nchw_view(ops.conv_dgrad(g, nhwc_view(gy), w, u=u, md_out=md, md_filled=md_filled, add=add))
                add = None
            if last_use:
                ctx.u = None
        else:
            if g16 is not None and ops.precast_ok(0, g):             # gy plays x in this call
                dx = nchw_view(ops.conv_fwd(g, nhwc_view(gy), w, u=getattr(ctx, "u", None), v_out=g16, v_filled=True))
                H16_STATS["used_bwd"] += 1
            else:
                g16 = None
                dx = nchw_view(ops.conv_fwd(g, nhwc_view(gy), w, u=getattr(ctx, "u", None)))
            if last_use:
                ctx.u = None
    if want_dw and not transposed and rows is None and (bias is None or not bias.requires_grad or (
            getattr(bias, "_mg_zero_grad", False) and not COMPUTE_DEAD_BIAS_GRADS)):
        vv = getattr(ctx, "v", None) if (md is not None or ops.tiles_are_casts(g)) else None
        if _wgrad_fused(u_used, g, weight, nhwc_view(x), nhwc_view(gy), vv, md):
            if bias is not None and bias.requires_grad:
                _zero_grad_bias(bias)
                _notify(bias)
            ctx.v = None
            want_dw = False
    if want_dw and not transposed and getattr(weight, "_mg_g16", None) is not None:
        # --fp16, FusedAdam GRAD_F16: the kernel stores the gradient as float16 (and runs the GradScaler's check on it)
        w16, wacc = grad_buffer16(weight)
        if bias is not None and bias.requires_grad:
            if getattr(bias, "_mg_zero_grad", False) and not COMPUTE_DEAD_BIAS_GRADS:
                _zero_grad_bias(bias)
            else:
                bbuf, bacc = grad_buffer(bias)
                ops.colsum(nhwc_view(gy).reshape(-1, g.Co), bbuf, bacc)
        flag = _producer_flag(weight, g)
        if ops.wgrad_h16_ok(g):
            ops.conv_wgrad_h16(g, nhwc_view(x), nhwc_view(gy), w16, wacc, found_inf=flag)
            weight._mg_inf_checked = flag is not None
        else:
            # another geometry than the one the arena was laid out for (a batch the float16-storing kernel does not take): the
            # float32 kernel, then one cast into the float16 slot the optimiser reads
            tmp = torch.empty(weight.numel(), dtype=torch.float32, device=gy.device)
            ops.conv_wgrad(g, nhwc_view(x), nhwc_view(gy), tmp, None, False)
            if wacc:
                w16.add_(tmp)
            else:
                w16.copy_(tmp)
            weight._mg_inf_checked = False
        ctx.v = None
        _notify(weight)
        if bias is not None and bias.requires_grad:
            _notify(bias)
        want_dw = False
    if want_dw:
        wbuf, wacc = grad_buffer(weight)
        bbuf = bacc = None
        if bias is not None and bias.requires_grad:
            if getattr(bias, "_mg_zero_grad", False) and not COMPUTE_DEAD_BIAS_GRADS:
                _zero_grad_bias(bias)
            else:
                bbuf, bacc = grad_buffer(bias)
        if not transposed:
            if bbuf is not None and bacc != wacc:      # keep one accumulate flag per launch
                ops.colsum(nhwc_view(gy).reshape(-1, g.Co), bbuf, bacc)
                bbuf = None
            # Winograd images come as a pair; the float16 copies of the implicit-GEMM layers are independent
            v = getattr(ctx, "v", None) if (md is not None or ops.tiles_are_casts(g)) else None
            flag = _producer_flag(weight, g)
            ops.conv_wgrad(g, nhwc_view(x), nhwc_view(gy), wbuf, bbuf, wacc, v=v, md=md, found_inf=flag)
            weight._mg_inf_checked = flag is not None
            ctx.v = None
        else:
            flag = _producer_flag(weight, g)
            half = ops.tiles_are_casts(g) and g.precision == _lib.PRECISION_F16 and g.Co != 1
            ops.conv_wgrad(g, nhwc_view(gy), nhwc_view(x), wbuf, None, wacc, found_inf=flag,
                           v=g16 if half else None, md=getattr(ctx, "v", None) if half else None)
            ctx.v = None
            weight._mg_inf_checked = flag is not None
            if bbuf is not None:
                ops.colsum(nhwc_view(gy).reshape(-1, g.Ci), bbuf, bacc)
        _notify(weight)
        if bias is not None and bias.requires_grad:
            _notify(bias)
    return dx


def conv2d(x, weight, bias, stride=1, padding=0, reflect=False, act=ACT_NONE, weight_grad=True):
    return _ConvFn.apply(x, weight, bias, (stride, padding, bool(reflect), act, False, weight_grad))


def conv_transpose2d(x, weight, bias, stride=2, padding=1, act=ACT_NONE, weight_grad=True):
    return _ConvFn.apply(x, weight, bias, (stride, padding, False, act, True, weight_grad))


# ------------------------------------------------------------------------------------------------
# instance norm (+ activation, + residual)
# ------------------------------------------------------------------------------------------------
class _ConvInstNormFn(torch.autograd.Function):
    """y = act(InstanceNorm2d(conv(x, w) + b)) + residual as ONE node (ResnetBlock's two conv / norm pairs,
    models/networks.py:440-462): forward is ops.conv_fwd_instnorm -- the Winograd inverse transform, the statistics and
    the normalisation in one kernel on the trunk's small maps -- backward is the InstanceNorm backward followed by the
    convolution's (_conv_backward), exactly what the two separate nodes do."""

    @staticmethod
    def forward(ctx, x, weight, bias, residual, cfg):
        x = to_cl(x)
        pad, reflect, weight_grad, act, eps = cfg[:5]
        ctx.skip = cfg[5] if len(cfg) > 5 else None      # ("give" | "take", SkipGrad): see SkipGrad
        res = to_cl(residual) if residual is not None else None
        B, _, H, W = x.shape
        w = weight.detach()
        assert w.is_contiguous(memory_format=CL), "conv weights must be channels_last (OHWI) tensors"
        g = ops.conv_geom(B, H, W, w.shape[1], w.shape[0], w.shape[2], w.shape[3], 1, pad, reflect, amp.current_precision())
        _tag_g16(g, weight, bias, weight_grad)
        if ctx.needs_input_grad[0]:
            u = _weight_image(g, weight, weight_grad)
        elif x.is_cuda and not torch.is_grad_enabled():
            u = _cached_wino_weights(g, weight, fill=not torch.cuda.is_current_stream_capturing())
        else:
            u = None
        v, v_filled = None, False
        x16 = _h16_of(x) if ops.precast_ok(0, g) else None
        xv = (_wino_v_of(x, (B, H, W, w.shape[1], bool(reflect), g.precision))
              if (pad == 1 and w.shape[2] == 3 and w.shape[3] == 3 and ops.wino_vnext_ok(g)) else None)
        if x16 is not None:
            v, v_filled = x16, True
            H16_STATS["used_fwd"] += 1
        elif xv is not None:          # x's producer (the trunk layer in front) wrote B^T x B: this layer's own V, kept for its weight gradient
            v, v_filled = xv, True
            WINO_NEXT_STATS["used"] += 1
        elif u is not None and weight_grad and weight.requires_grad:
            v, _ = ops.wino_tile_buffers(g, x.device, want_md=False)
        # ... and this layer writes the NEXT trunk layer's image when told that one follows (cfg[7]: its padding mode)
        next_reflect = cfg[7] if len(cfg) > 7 else None
        v_next = None
        if next_reflect is not None and ops.wino_vnext_ok(g) and os.environ.get("MG_NO_WINO_NEXT", "0") != "1":
            v_next = torch.empty(16 * B * (g.OH // 2) * (g.OW // 2) * g.Co, dtype=torch.float32, device=x.device)
        y16 = (torch.empty(B * g.OH * g.OW * g.Co, dtype=torch.float16, device=x.device)
               if _want_h16(B, g.OH * g.OW, g.Co) else None)
        # under torch.no_grad() nobody reads the raw convolution output again (grad mode is read by conv_instnorm(): inside a
        # Function's forward it is always off, and needs_input_grad ignores it)
        need_raw = bool(cfg[6]) and any(ctx.needs_input_grad) if len(cfg) > 6 else any(ctx.needs_input_grad)
        y, y_raw, mean, rstd = ops.conv_fwd_instnorm(g, nhwc_view(x), w, bias.detach() if bias is not None else None, act,
                                                     nhwc_view(res) if res is not None else None, eps, u, v, v_filled, y16,
                                                     need_raw=need_raw, v_next=v_next, next_reflect=bool(next_reflect))
        ctx.u, ctx.v = u, (v if (weight_grad and weight.requires_grad) else None)
        ctx.g, ctx.cfg = g, (1, pad, reflect, ACT_NONE, False, weight_grad)
        ctx.weight, ctx.bias, ctx.norm_act = weight, bias, act
        if need_raw:
            ctx.save_for_backward(x, nchw_view(y_raw), mean, rstd)
        out = _attach_h16(nchw_view(y), y16)
        if v_next is not None:
            _attach_wino_v(out, v_next, (B, g.OH, g.OW, g.Co, bool(next_reflect), g.precision))
        return out

    @staticmethod
    def backward(ctx, gy):
        x, y_raw, mean, rstd = ctx.saved_tensors
        gy = to_cl(gy)
        dres = gy if ctx.needs_input_grad[3] else None
        skip_g = None
        if ctx.skip is not None:
            role, holder = ctx.skip
            if role == "give" and dres is not None:      # the block's first convolution adds it to its data gradient
                holder.g, dres = dres, None
            elif role == "take":
                skip_g, holder.g = holder.g, None
        g, weight, bias = ctx.g, ctx.weight, ctx.bias
        weight_grad = ctx.cfg[5]
        dead_bias = bias is None or not bias.requires_grad or (getattr(bias, "_mg_zero_grad", False) and not COMPUTE_DEAD_BIAS_GRADS)
        u, v = getattr(ctx, "u", None), getattr(ctx, "v", None)
        if (ctx.needs_input_grad[0] and weight_grad is True and weight.requires_grad and dead_bias and u is not None
                and v is not None and ops.wino_md_from_norm_ok(g)):
            # InstanceNorm backward and the A dy A^T transform in one kernel; data and weight gradient start from the images
            _, md = ops.wino_tile_buffers(g, gy.device, want_v=False)
            ops.instnorm_bwd_wino_md(g, nhwc_view(gy), nhwc_view(y_raw), mean, rstd, ctx.norm_act, md)
            w = weight.detach()
            dx = nchw_view(ops.conv_dgrad(g, None, w, u=u, md_out=md,
                                          add=nhwc_view(skip_g) if skip_g is not None else None))
            if bias is not None and bias.requires_grad:
                _zero_grad_bias(bias)
            if not _wgrad_fused(u, g, weight, None, None, v, md):
                wbuf, wacc = grad_buffer(weight)
                ops.conv_wgrad(g, None, None, wbuf, None, wacc, v=v, md=md)
            ctx.u = ctx.v = None
            _notify(weight)
            if bias is not None and bias.requires_grad:
                _notify(bias)
            return dx, None, None, dres, None
        d16 = (torch.empty(gy.numel(), dtype=torch.float16, device=gy.device)
               if (_want_h16(g.B, g.OH * g.OW, g.Co, g.precision == _lib.PRECISION_F16) and ops.precast_ok(1, g)) else None)
        d_raw = _attach_h16(nchw_view(ops.instnorm_bwd(nhwc_view(gy), nhwc_view(y_raw), mean, rstd, ctx.norm_act, dx16=d16)), d16)
        # the skip connection's gradient joins the data gradient inside its last kernel (no elementwise add launch)
        dx = _conv_backward(ctx, d_raw, x, None, add=nhwc_view(skip_g) if (skip_g is not None and ctx.needs_input_grad[0]) else None)
        if skip_g is not None and dx is None:
            dx = skip_g
        return dx, None, None, dres, None


class SkipGrad:
    """Carries the gradient of a skip connection from the node that receives it (the block's last conv_instnorm, role "give":
    it hands over d(residual) instead of returning it to autograd) to the node that produced the skipped tensor's other use
    (the block's first conv_instnorm, role "take": dx += g inside its data gradient's last kernel) -- the engine would add
    the two with an elementwise kernel.  Valid because both nodes take the SAME tensor and "take" runs after "give" in every
    backward pass (its output feeds "give")."""
    __slots__ = ("g",)

    def __init__(self):
        self.g = None


def conv_instnorm(x, weight, bias, padding=0, reflect=False, act=ACT_NONE, residual=None, eps=1e-5, weight_grad=True, skip=None,
                  next_reflect=None):
    """act(InstanceNorm2d(affine=False)(conv2d(x, weight, bias, stride 1))) + residual.  skip: ("give" | "take", SkipGrad).
    next_reflect (None | bool): the output feeds another 3x3 stride-1 pad-1 convolution with that padding mode -- where the fused
    kernel runs it also writes that layer's Winograd input image (see _attach_wino_v)."""
    cfg = (padding, bool(reflect), weight_grad, act, eps, skip, torch.is_grad_enabled(), next_reflect)
    return _ConvInstNormFn.apply(x, weight, bias, residual, cfg)


class _InstNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, residual, act, eps):
        x = to_cl(x)
        res = to_cl(residual) if residual is not None else None
        B, C, H, W = x.shape
        y16 = torch.empty(x.numel(), dtype=torch.float16, device=x.device) if _want_h16(B, H * W, C) else None
        y, mean, rstd = ops.instnorm_fwd(nhwc_view(x), act, nhwc_view(res) if res is not None else None, eps, y16=y16)
        ctx.act, ctx.half = act, amp.current_precision() == _lib.PRECISION_F16
        ctx.save_for_backward(x, mean, rstd)
        return _hang_extra(ctx, _attach_h16(nchw_view(y), y16))

    @staticmethod
    def backward(ctx, gy):
        x, mean, rstd = ctx.saved_tensors
        gy = to_cl(gy)
        dx = None
        extra = _take_extra(ctx, gy)               # the feature-matching loss's gradient of this layer's output (see ExtraGrad)
        if extra is not None:
            extra = to_cl(extra)
        if ctx.needs_input_grad[0]:
            rows = _live_rows(x)
            if rows is not None:                               # pass "G" over a stacked discriminator batch
                dx = torch.empty_like(x)
                B, C, H, W = x.shape
                # the float16 copy of the live rows for the data gradient in front (round 6: the fake half only)
                d16 = (torch.empty(rows * H * W * C, dtype=torch.float16, device=x.device)
                       if _want_h16(rows, H * W, C, ctx.half) else None)
                ops.instnorm_bwd(nhwc_view(gy[:rows]), nhwc_view(x[:rows]), mean[:rows], rstd[:rows], ctx.act,
                                 out=nhwc_view(dx[:rows]), dx16=d16, dy2=nhwc_view(extra[:rows]) if extra is not None else None)
                if d16 is not None:
                    dx._mg_h16_rows = (d16, dx.data_ptr(), dx._version, rows)
                    H16_STATS["made"] += 1
            else:
                B, C, H, W = x.shape
                d16 = torch.empty(x.numel(), dtype=torch.float16, device=x.device) if _want_h16(B, H * W, C, ctx.half) else None
                dx = _attach_h16(nchw_view(ops.instnorm_bwd(nhwc_view(gy), nhwc_view(x), mean, rstd, ctx.act, dx16=d16,
                                                            dy2=nhwc_view(extra) if extra is not None else None)), d16)
        dres = gy if ctx.needs_input_grad[1] else None
        return dx, dres, None, None


def instance_norm_act(x, act=ACT_NONE, residual=None, eps=1e-5):
    """act(InstanceNorm2d(affine=False)(x)) + residual."""
    return _InstNormFn.apply(x, residual, act, eps)


class _BatchNormFn(torch.autograd.Function):
    """act(BatchNorm2d(x) + residual), training or eval statistics; gamma / beta gradients go straight to .grad."""

    @staticmethod
    def forward(ctx, x, gamma, beta, residual, bn, act):
        x = to_cl(x)
        res = to_cl(residual) if residual is not None else None
        training = bn.training or bn.running_mean is None
        y, mean, rstd = ops.batchnorm_fwd(nhwc_view(x), gamma.detach(), beta.detach(), bn.running_mean, bn.running_var,
                                          bn.eps, bn.momentum if bn.momentum is not None else 0.1, training,
                                          nhwc_view(res) if res is not None else None, act)
        if training and bn.num_batches_tracked is not None and not getattr(bn, "_mg_counter_external", False):
            bn.num_batches_tracked.add_(1)
        y = nchw_view(y)
        ctx.act, ctx.training, ctx.gamma, ctx.beta = act, training, gamma, beta
        ctx.save_for_backward(x, y, mean, rstd)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, y, mean, rstd = ctx.saved_tensors
        gamma, beta = ctx.gamma, ctx.beta
        gy = to_cl(gy)
        gbuf = bbuf = None
        acc = False
        if gamma.requires_grad:
            gbuf, acc = grad_buffer(gamma)
            bbuf, acc2 = grad_buffer(beta)
            assert acc == acc2
        dx, dres = ops.batchnorm_bwd(nhwc_view(gy), nhwc_view(x), nhwc_view(y), gamma.detach(), mean, rstd, ctx.act,
                                     ctx.training, gbuf, bbuf, acc, ctx.needs_input_grad[3])
        if gamma.requires_grad:
            _notify(gamma)
            _notify(beta)
        return nchw_view(dx), None, None, (nchw_view(dres) if dres is not None else None), None, None


def batch_norm_act(x, bn, act=ACT_NONE, residual=None):
    """act(bn(x) + residual) for an nn.BatchNorm2d module `bn` (its buffers are updated in training mode)."""
    return _BatchNormFn.apply(x, bn.weight, bn.bias, residual, bn, act)


class _AttentionFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, qkv, emb_h, emb_w, heads, dim_head):
        qkv = to_cl(qkv)
        out, P = ops.attention_fwd(nhwc_view(qkv), emb_h.detach().contiguous(), emb_w.detach().contiguous(), heads, dim_head)
        ctx.heads, ctx.d, ctx.eh, ctx.ew = heads, dim_head, emb_h, emb_w
        ctx.save_for_backward(qkv, P)
        return nchw_view(out)

    @staticmethod
    def backward(ctx, go):
        qkv, P = ctx.saved_tensors
        eh, ew = ctx.eh, ctx.ew
        go = to_cl(go)
        hbuf = wbuf = None
        acc = False
        if eh.requires_grad:
            hbuf, acc = grad_buffer(eh)
            wbuf, acc2 = grad_buffer(ew)
            assert acc == acc2
        dqkv = ops.attention_bwd(nhwc_view(qkv), eh.detach().contiguous(), ew.detach().contiguous(), nhwc_view(go), P,
                                 ctx.heads, ctx.d, hbuf, wbuf, acc)
        if eh.requires_grad:
            _notify(eh)
            _notify(ew)
        return nchw_view(dqkv), None, None, None, None


def bot_attention(qkv, emb_h, emb_w, heads, dim_head):
    """Multi-head self attention of the bottleneck-transformer block on the to_qkv output (abs. position embeddings)."""
    return _AttentionFn.apply(qkv, emb_h, emb_w, heads, dim_head)


class _AddFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        a, b = to_cl(a), to_cl(b)
        return nchw_view(ops.add(nhwc_view(a), nhwc_view(b)))

    @staticmethod
    def backward(ctx, g):
        return g, g


def add(a, b):
    return _AddFn.apply(a, b)


class _AvgPoolFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = to_cl(x)
        ctx.in_shape = nhwc_view(x).shape
        return nchw_view(ops.avgpool_fwd(nhwc_view(x)))

    @staticmethod
    def backward(ctx, gy):
        gy = to_cl(gy)
        rows = _live_rows(gy)
        if rows is not None:                                   # pass "G" over a stacked discriminator batch
            B, H, W, Cc = ctx.in_shape
            dx = torch.empty(B, H, W, Cc, dtype=torch.float32, device=gy.device)
            ops.avgpool_bwd(nhwc_view(gy[:rows]), (rows, H, W, Cc), out=dx[:rows])
            return nchw_view(dx)
        return nchw_view(ops.avgpool_bwd(nhwc_view(gy), tuple(ctx.in_shape)))


def avg_pool_3s2(x):
    """nn.AvgPool2d(3, stride=2, padding=[1, 1], count_include_pad=False)."""
    return _AvgPoolFn.apply(x)


class _UpsampleFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return nchw_view(ops.upsample_fwd(nhwc_view(to_cl(x))))

    @staticmethod
    def backward(ctx, gy):
        return nchw_view(ops.upsample_bwd(nhwc_view(to_cl(gy))))


def upsample_nearest2x(x):
    return _UpsampleFn.apply(x)


class _ActFn(torch.autograd.Function):
    """Stand-alone activation (only used when a pattern could not be fused): computed by the add-free path
    y = act(x) through the conv epilogue helpers is not available, so use instnorm-free kernels."""

    @staticmethod
    def forward(ctx, x, act):
        raise NotImplementedError("stand-alone activations are always fused into the producing kernel")


# ------------------------------------------------------------------------------------------------
# discriminator input / generator input assembly
# ------------------------------------------------------------------------------------------------
class _DInputFn(torch.autograd.Function):
    """cat(lr, s, 2|s| + nr0) along channels (pix2pixHD_model.py:420-424, 439-440)."""

    @staticmethod
    def forward(ctx, lr, s, nr0):
        lr, s = to_cl(lr), to_cl(s)
        assert lr.shape[1] == 1 and s.shape[1] == 1
        ctx.save_for_backward(s)
        return nchw_view(ops.dinput_fwd(nhwc_view(lr), nhwc_view(s), nr0))

    @staticmethod
    def backward(ctx, g):
        (s,) = ctx.saved_tensors
        ds = nchw_view(ops.dinput_bwd(nhwc_view(to_cl(g)), nhwc_view(s))) if ctx.needs_input_grad[1] else None
        return None, ds, None


def d_input(lr_spectro, s_spectro, nr0):
    return _DInputFn.apply(lr_spectro, s_spectro, nr0)


class _Cat2Fn(torch.autograd.Function):
    """torch.cat((a, b), dim=1) (the discriminator input without --abs_spectro --arcsinh_transform,
    pix2pixHD_model.py:425-427, 440) on channels_last tensors."""

    @staticmethod
    def forward(ctx, a, b):
        from . import _lib
        a, b = to_cl(a), to_cl(b)
        B, Ca, H, W = a.shape
        Cb = b.shape[1]
        assert b.shape[0] == B and b.shape[2:] == a.shape[2:]
        out = torch.empty(B, H, W, Ca + Cb, dtype=torch.float32, device=a.device)
        _lib.check(_lib.load().mg_cat2_fwd(_lib.ptr(nhwc_view(a).contiguous()), Ca, _lib.ptr(nhwc_view(b).contiguous()), Cb,
                                           B * H * W, _lib.ptr(out), _lib.stream()), "mg_cat2_fwd")
        ctx.dims = (B, Ca, Cb, H, W)
        return nchw_view(out)

    @staticmethod
    def backward(ctx, g):
        from . import _lib
        B, Ca, Cb, H, W = ctx.dims
        g = nhwc_view(to_cl(g)).contiguous()
        ga = torch.empty(B, H, W, Ca, dtype=torch.float32, device=g.device) if ctx.needs_input_grad[0] else None
        gb = torch.empty(B, H, W, Cb, dtype=torch.float32, device=g.device) if ctx.needs_input_grad[1] else None
        if ga is None and gb is None:
            return None, None
        _lib.check(_lib.load().mg_cat2_bwd(_lib.ptr(g), Ca, Cb, B * H * W, _lib.ptr(ga), _lib.ptr(gb), _lib.stream()), "mg_cat2_bwd")
        return (nchw_view(ga) if ga is not None else None), (nchw_view(gb) if gb is not None else None)


def cat_channels(a, b):
    return _Cat2Fn.apply(a, b)


def d_input_pair(lr_spectro, a_spectro, b_spectro, nr0):
    """The discriminator inputs of two spectrogram batches stacked along the batch axis, [2B, 3, F, W]: rows [0, B)
    from ``a``, rows [B, 2B) from ``b``.  No gradient path (used for the two passes of the discriminator loss, whose
    inputs are the detached fake and the real spectrogram)."""
    lr, a, b = to_cl(lr_spectro.detach()), to_cl(a_spectro.detach()), to_cl(b_spectro.detach())
    B, _, H, W = a.shape
    out = torch.empty(2 * B, H, W, 3, dtype=torch.float32, device=a.device)
    ops.dinput_fwd(nhwc_view(lr), nhwc_view(a), nr0, out=out[:B])
    ops.dinput_fwd(nhwc_view(lr), nhwc_view(b), nr0, out=out[B:])
    return nchw_view(out)


class _DInputSharedFn(torch.autograd.Function):
    """d_input_pair whose first half keeps its gradient path (the generator output): the input of the single
    discriminator forward that serves both the D loss and the G loss (see backward_pass)."""

    @staticmethod
    def forward(ctx, lr, a, b, nr0):
        lr, a, b = to_cl(lr), to_cl(a), to_cl(b)
        B, _, H, W = a.shape
        out = torch.empty(2 * B, H, W, 3, dtype=torch.float32, device=a.device)
        ops.dinput_fwd(nhwc_view(lr), nhwc_view(a), nr0, out=out[:B])
        ops.dinput_fwd(nhwc_view(lr), nhwc_view(b), nr0, out=out[B:])
        if TAP is not None:
            TAP.append(("abs_sign", torch.sign(a.detach())))
        ctx.save_for_backward(a)
        return nchw_view(out)

    @staticmethod
    def backward(ctx, g):
        (a,) = ctx.saved_tensors
        if not ctx.needs_input_grad[1] or _BackwardPass.kind == "D":
            return None, None, None, None
        g = to_cl(g)
        return None, nchw_view(ops.dinput_bwd(nhwc_view(g[:a.shape[0]]), nhwc_view(a))), None, None


def d_input_shared(lr_spectro, fake_spectro, real_spectro, nr0):
    """[2B, 3, F, W]: rows [0, B) from the (attached) generator output, rows [B, 2B) from the real spectrogram."""
    return _DInputSharedFn.apply(lr_spectro.detach(), fake_spectro, real_spectro.detach(), nr0)


def g_input(spectro, nr0):
    """cat(s, 2|s| + nr0) (pix2pixHD_model.py:400-402); no gradient path (the generator input is data)."""
    s = to_cl(spectro.detach())
    return nchw_view(ops.pair_fwd(nhwc_view(s), nr0))


# ------------------------------------------------------------------------------------------------
# losses
# ------------------------------------------------------------------------------------------------
class _MseConstFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, target, scale):
        pred = pred.contiguous(memory_format=CL) if pred.dim() == 4 else pred.contiguous()
        loss = torch.empty(1, dtype=torch.float32, device=pred.device)
        ops.mse_const_fwd(pred, target, scale, loss, False)
        ctx.target, ctx.scale = target, scale
        ctx.save_for_backward(pred)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, go):
        (pred,) = ctx.saved_tensors
        go = go.reshape(1).float().contiguous()
        return ops.mse_const_bwd(pred, ctx.target, ctx.scale, go), None, None


class _MseConstPairFn(torch.autograd.Function):
    """(mean((pred[:B] - t0)^2), mean((pred[B:] - t1)^2)) of a batch-stacked prediction; one gradient buffer."""

    @staticmethod
    def forward(ctx, pred, t0, t1):
        pred = pred.contiguous(memory_format=CL) if pred.dim() == 4 else pred.contiguous()
        B = pred.shape[0] // 2
        l0 = torch.empty(1, dtype=torch.float32, device=pred.device)
        l1 = torch.empty(1, dtype=torch.float32, device=pred.device)
        ops.mse_const_fwd(pred[:B], t0, 1.0, l0, False)
        ops.mse_const_fwd(pred[B:], t1, 1.0, l1, False)
        ctx.targets = (t0, t1)
        ctx.save_for_backward(pred)
        return l0.reshape(()), l1.reshape(())

    @staticmethod
    def backward(ctx, g0, g1):
        (pred,) = ctx.saved_tensors
        B = pred.shape[0] // 2
        g = torch.empty_like(pred)
        ops.mse_const_bwd(pred[:B], ctx.targets[0], 1.0, g0.reshape(1).float().contiguous(), out=g[:B])
        ops.mse_const_bwd(pred[B:], ctx.targets[1], 1.0, g1.reshape(1).float().contiguous(), out=g[B:])
        return g, None, None


def mse_const_pair_loss(pred, target_first: float, target_second: float):
    """The two LSGAN terms of a prediction whose batch stacks two passes (first half / second half)."""
    assert pred.shape[0] % 2 == 0
    return _MseConstPairFn.apply(pred, float(target_first), float(target_second))


class _MseConstFirstHalfFn(torch.autograd.Function):
    """mean((pred[:B] - t)^2) of a batch-stacked prediction; the gradient buffer has the stacked shape, second half
    unwritten (consumed by pass "G" only, which never reads it)."""

    @staticmethod
    def forward(ctx, pred, target):
        pred = pred.contiguous(memory_format=CL) if pred.dim() == 4 else pred.contiguous()
        B = pred.shape[0] // 2
        loss = torch.empty(1, dtype=torch.float32, device=pred.device)
        ops.mse_const_fwd(pred[:B], target, 1.0, loss, False)
        ctx.target = target
        ctx.save_for_backward(pred)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, go):
        (pred,) = ctx.saved_tensors
        B = pred.shape[0] // 2
        g = torch.empty_like(pred)
        if _BackwardPass.kind != "G":
            g[B:].zero_()
        ops.mse_const_bwd(pred[:B], ctx.target, 1.0, go.reshape(1).float().contiguous(), out=g[:B])
        return g, None


def mse_const_first_half_loss(pred, target: float):
    assert pred.shape[0] % 2 == 0
    return _MseConstFirstHalfFn.apply(pred, float(target))


class _L1HalvesFn(torch.autograd.Function):
    """scale * mean(|t[:B] - t[B:].detach()|): the feature-matching term of a batch-stacked [fake, real] feature map."""

    @staticmethod
    def forward(ctx, t, scale):
        t = t.contiguous(memory_format=CL) if t.dim() == 4 else t.contiguous()
        B = t.shape[0] // 2
        loss = torch.empty(1, dtype=torch.float32, device=t.device)
        ops.l1_fwd(t[:B], t[B:], scale, loss, False)
        ctx.scale = scale
        ctx.save_for_backward(t)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, go):
        (t,) = ctx.saved_tensors
        B = t.shape[0] // 2
        g = torch.empty_like(t)
        if _BackwardPass.kind != "G":
            g[B:].zero_()
        ops.l1_bwd(t[:B], t[B:], ctx.scale, go.reshape(1).float().contiguous(), out=g[:B])
        return g, None


def l1_halves_loss(t, scale: float = 1.0):
    assert t.shape[0] % 2 == 0
    return _L1HalvesFn.apply(t, float(scale))


# Sums of such terms over the discriminator scales / layers as ONE autograd node each: the kernels accumulate into the
# loss scalar (mg_*_fwd's accumulate flag, the same float32 additions in the same order as `loss = loss + term`), so the
# step launches no at::native add for them.
def _cl(t):
    return t.contiguous(memory_format=CL) if t.dim() == 4 else t.contiguous()


class _L1HalvesSumFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, scale, *ts):
        ctx.holders = [getattr(t, "_mg_extra", None) for t in ts]      # where the producers take a second gradient (ExtraGrad)
        ts = [_cl(t) for t in ts]
        if TAP is not None:
            TAP.append(("l1_sign", [torch.sign(t[:t.shape[0] // 2].detach() - t[t.shape[0] // 2:].detach()) for t in ts]))
        loss = torch.empty(1, dtype=torch.float32, device=ts[0].device)
        # all layers in one partial + one final launch (mg_loss_multi_fwd): same blocks, same order of additions
        ops.loss_multi_fwd(ops.LOSS_L1, [(t[:t.shape[0] // 2], t[t.shape[0] // 2:], None, 0) for t in ts], 0.0, scale, loss)
        ctx.scale = scale
        ctx.save_for_backward(*ts)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, go):
        go = go.reshape(1).float().contiguous()
        grads, rows = [], []
        for t in ctx.saved_tensors:
            B = t.shape[0] // 2
            g = torch.empty_like(t)
            # the real half gets no gradient from this loss: cleared by the same launch (pass "G" never reads it)
            rows.append((t[:B], t[B:], g, (t.numel() - t[:B].numel()) if _BackwardPass.kind != "G" else 0))
            grads.append(g)
        ops.loss_multi_bwd(ops.LOSS_L1, rows, 0.0, ctx.scale, go)
        if _BackwardPass.kind == "G":
            # optimize_parameters' generator pass: every map's producer runs later in this pass (reached through the next layer) and
            # adds the parked gradient inside its own kernel -- no accumulation launch
            for i, h in enumerate(ctx.holders):
                if h is not None:
                    assert h.g is None, "an ExtraGrad was parked twice without being taken"
                    h.g, grads[i] = grads[i], None
        return (None, *grads)


def l1_halves_loss_sum(ts, scale: float = 1.0):
    """sum_i l1_halves_loss(ts[i], scale), one node."""
    ts = list(ts)
    assert ts and all(t.shape[0] % 2 == 0 for t in ts)
    return _L1HalvesSumFn.apply(float(scale), *ts)


class _MseConstFirstHalfSumFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, target, *preds):
        preds = [_cl(t) for t in preds]
        loss = torch.empty(1, dtype=torch.float32, device=preds[0].device)
        ops.loss_multi_fwd(ops.LOSS_MSE_CONST, [(t[:t.shape[0] // 2], None, None, 0) for t in preds], target, 1.0, loss)
        ctx.target = target
        ctx.save_for_backward(*preds)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, go):
        go = go.reshape(1).float().contiguous()
        grads, rows = [], []
        for t in ctx.saved_tensors:
            B = t.shape[0] // 2
            g = torch.empty_like(t)
            rows.append((t[:B], None, g, (t.numel() - t[:B].numel()) if _BackwardPass.kind != "G" else 0))
            grads.append(g)
        ops.loss_multi_bwd(ops.LOSS_MSE_CONST, rows, ctx.target, 1.0, go)
        return (None, *grads)


def mse_const_first_half_loss_sum(preds, target: float):
    """sum_i mse_const_first_half_loss(preds[i], target), one node."""
    preds = list(preds)
    assert preds and all(t.shape[0] % 2 == 0 for t in preds)
    return _MseConstFirstHalfSumFn.apply(float(target), *preds)


class _MseConstPairSumFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, t0, t1, *preds):
        preds = [_cl(t) for t in preds]
        l0 = torch.empty(1, dtype=torch.float32, device=preds[0].device)
        l1 = torch.empty(1, dtype=torch.float32, device=preds[0].device)
        ops.loss_multi_fwd(ops.LOSS_MSE_CONST, [(t[:t.shape[0] // 2], None, None, 0) for t in preds], t0, 1.0, l0)
        ops.loss_multi_fwd(ops.LOSS_MSE_CONST, [(t[t.shape[0] // 2:], None, None, 0) for t in preds], t1, 1.0, l1)
        ctx.targets = (t0, t1)
        ctx.save_for_backward(*preds)
        return l0.reshape(()), l1.reshape(())

    @staticmethod
    def backward(ctx, g0, g1):
        g0 = g0.reshape(1).float().contiguous()
        g1 = g1.reshape(1).float().contiguous()
        grads, rows0, rows1 = [], [], []
        for t in ctx.saved_tensors:
            B = t.shape[0] // 2
            g = torch.empty_like(t)
            rows0.append((t[:B], None, g[:B], 0))
            rows1.append((t[B:], None, g[B:], 0))
            grads.append(g)
        ops.loss_multi_bwd(ops.LOSS_MSE_CONST, rows0, ctx.targets[0], 1.0, g0)
        ops.loss_multi_bwd(ops.LOSS_MSE_CONST, rows1, ctx.targets[1], 1.0, g1)
        return (None, None, *grads)


def mse_const_pair_loss_sum(preds, target_first: float, target_second: float):
    """(sum_i first-half term, sum_i second-half term) of mse_const_pair_loss over preds, one node."""
    preds = list(preds)
    assert preds and all(t.shape[0] % 2 == 0 for t in preds)
    return _MseConstPairSumFn.apply(float(target_first), float(target_second), *preds)


class _BceConstFn(torch.autograd.Function):
    """scale * nn.BCELoss()(pred, full_like(pred, target)) (networks.py:106-109 with use_lsgan=False)."""

    @staticmethod
    def forward(ctx, pred, target, scale):
        from . import _lib
        pred = pred.contiguous(memory_format=CL) if pred.dim() == 4 else pred.contiguous()
        loss = torch.empty(1, dtype=torch.float32, device=pred.device)
        lib = _lib.load()
        ws = _lib.workspace(lib.mg_loss_workspace(), pred.device)
        _lib.check(lib.mg_bce_const_fwd(_lib.ptr(pred), pred.numel(), target, scale, _lib.ptr(loss), 0, _lib.ptr(ws), _lib.stream()),
                   "mg_bce_const_fwd")
        ctx.target, ctx.scale = target, scale
        ctx.save_for_backward(pred)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, go):
        from . import _lib
        (pred,) = ctx.saved_tensors
        go = go.reshape(1).float().contiguous()
        g = torch.empty_like(pred)
        _lib.check(_lib.load().mg_bce_const_bwd(_lib.ptr(pred), pred.numel(), ctx.target, ctx.scale, _lib.ptr(go), _lib.ptr(g),
                                                _lib.stream()), "mg_bce_const_bwd")
        return g, None, None


def bce_const_loss(pred, target: float, scale: float = 1.0):
    return _BceConstFn.apply(pred, float(target), float(scale))


class _SigmoidFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        from . import _lib
        x = to_cl(x) if x.dim() == 4 else x.contiguous()
        y = torch.empty_like(x)
        _lib.check(_lib.load().mg_sigmoid_fwd(_lib.ptr(x), _lib.ptr(y), x.numel(), _lib.stream()), "mg_sigmoid_fwd")
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, gy):
        from . import _lib
        (y,) = ctx.saved_tensors
        gy = gy.contiguous(memory_format=CL) if gy.dim() == 4 else gy.contiguous()
        dx = torch.empty_like(y)
        _lib.check(_lib.load().mg_sigmoid_bwd(_lib.ptr(gy), _lib.ptr(y), _lib.ptr(dx), y.numel(), _lib.stream()), "mg_sigmoid_bwd")
        return dx


def sigmoid(x):
    return _SigmoidFn.apply(x)


def mse_const_loss(pred, target: float, scale: float = 1.0):
    """scale * mean((pred - target)^2)  == nn.MSELoss()(pred, full_like(pred, target)) * scale."""
    return _MseConstFn.apply(pred, float(target), float(scale))


class _L1Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b, scale):
        fmt = CL if a.dim() == 4 else torch.contiguous_format
        a, b = a.contiguous(memory_format=fmt), b.contiguous(memory_format=fmt)
        loss = torch.empty(1, dtype=torch.float32, device=a.device)
        ops.l1_fwd(a, b, scale, loss, False)
        ctx.scale = scale
        ctx.save_for_backward(a, b)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, go):
        a, b = ctx.saved_tensors
        go = go.reshape(1).float().contiguous()
        ga = ops.l1_bwd(a, b, ctx.scale, go) if ctx.needs_input_grad[0] else None
        gb = None
        if ctx.needs_input_grad[1]:
            gb = ops.l1_bwd(b, a, ctx.scale, go)
        return ga, gb, None


def l1_loss(a, b, scale: float = 1.0):
    """scale * mean(|a - b|) == nn.L1Loss()(a, b) * scale."""
    return _L1Fn.apply(a, b, float(scale))
