"""Thin launchers over the C ABI (one Python function per entry point of include/mdctgan_hip.h).
Tensors are float32, contiguous, NHWC, resident in HBM.  No arithmetic happens here."""
from __future__ import annotations

import torch

from . import _lib
from ._lib import ACT_LRELU02, ACT_NONE, ACT_RELU, ACT_TANH, ConvGeom  # noqa: F401


def conv_geom(B, H, W, Ci, Co, KH, KW, stride, pad, reflect, precision=0) -> ConvGeom:
    """precision: _lib.PRECISION_F32 (exact float32) or _lib.PRECISION_F16 (autocast arithmetic: f16 MFMA products,
    float32 accumulation, forward / data-gradient outputs rounded through float16)."""
    OH = (H + 2 * pad - KH) // stride + 1
    OW = (W + 2 * pad - KW) // stride + 1
    return ConvGeom(B, H, W, Ci, OH, OW, Co, KH, KW, stride, pad, int(bool(reflect)), int(precision))


def _ws(nbytes, dev):
    return _lib.workspace(nbytes, dev)


# ---- optional per-kernel event timing (bench.py's roofline leg) ----------------------------------------------
PROFILER = None     # object with .begin(name, flops) / .end(); set by bench.py, None in normal operation


def plan_name(pass_id: int, g: ConvGeom) -> str:
    import ctypes
    buf = ctypes.create_string_buffer(96)
    _lib.check(_lib.load().mg_conv_plan_name(pass_id, g, buf, 96), "mg_conv_plan_name")
    return buf.value.decode()


def plan_flops(pass_id: int, g: ConvGeom) -> float:
    """FLOPs the main GEMM kernel of this pass issues (Winograd layers: the 16 transformed-domain GEMMs)."""
    return float(_lib.load().mg_conv_plan_flops(pass_id, g))


def conv_flops(g: ConvGeom) -> float:
    """Algorithmic FLOPs of one pass (fwd == dgrad == wgrad): 2 * B*OH*OW * Co * KH*KW*Ci."""
    return 2.0 * g.B * g.OH * g.OW * g.Co * g.KH * g.KW * g.Ci


def wino_weights_bytes(g: ConvGeom) -> int:
    """Size of the transformed-weight image of this geometry (0: not a Winograd layer in the current configuration)."""
    return int(_lib.load().mg_conv_wino_weights_bytes(g))


def wino_weights(g: ConvGeom, w):
    """Transformed weights U = G w G^T of a Winograd layer (None when the geometry is not one): lets the forward
    and the data-gradient pass of one step share a single weight transform."""
    lib = _lib.load()
    nbytes = lib.mg_conv_wino_weights_bytes(g)
    if not nbytes:
        return None
    u = torch.empty(nbytes // 4, dtype=torch.float32, device=w.device)
    _lib.check(lib.mg_conv_wino_prepare(g, _lib.ptr(w), _lib.ptr(u), _lib.stream()), "mg_conv_wino_prepare")
    return u


def wino_tile_buffers(g: ConvGeom, device, want_v=True, want_md=True):
    """(v, md) float32 buffers for the Winograd images a training step shares between its passes (None where the
    geometry / configuration does not use one)."""
    lib = _lib.load()
    nv = lib.mg_conv_wino_tiles_bytes(g, 0) if want_v else 0
    nm = lib.mg_conv_wino_tiles_bytes(g, 1) if want_md else 0
    v = torch.empty(nv // 4, dtype=torch.float32, device=device) if nv else None
    md = torch.empty(nm // 4, dtype=torch.float32, device=device) if nm else None
    return v, md


def tiles_are_casts(g: ConvGeom) -> bool:
    """True for layers whose v / md "tiles" are usable on their own (Winograd images only work as a pair): the autocast
    layers on the float16 implicit GEMMs (csrc/conv_dma.h: plain float16 copies of x / dy) and the single-output-channel
    tap GEMMs (csrc/conv_co1.h: md = the scattered dy)."""
    key = (g.B, g.H, g.W, g.Ci, g.Co, g.KH, g.KW, g.stride, g.pad, g.reflect, g.precision)
    hit = _CASTS.get(key)
    if hit is None:
        name = plan_name(2, g)
        half = g.precision == _lib.PRECISION_F16 and name.startswith("conv_wgrad_dma_kernel") and ", true" in name
        # ... and the single-output-channel tap GEMMs (csrc/conv_co1.h): md is the scattered dy, there is no v
        co1 = g.Co == 1 and wino_weights_bytes(g) == 64 * g.Ci * 4
        hit = _CASTS[key] = half or co1
    return hit


_CASTS = {}


TILES_V_FILLED, TILES_MD_FILLED = 1, 2      # include/mdctgan_hip.h: MG_TILES_*_FILLED


def _tiles(u=None, v=None, md=None, add=None, flags=0):
    if u is None and v is None and md is None and add is None:
        return None
    return _lib.WinoTiles(_lib.ptr(u), _lib.ptr(v), _lib.ptr(md), _lib.ptr(add), flags)


def precast_ok(pass_id: int, g: ConvGeom) -> bool:
    """True when this pass of the layer reads its activation operand from a plain float16 copy and honours MG_TILES_V_FILLED
    (pass 0) / MG_TILES_MD_FILLED (pass 1): the HALF instances of the implicit GEMMs (csrc/conv_dma.h)."""
    if g.precision != _lib.PRECISION_F16:
        return False
    key = ("pc", pass_id, g.B, g.H, g.W, g.Ci, g.Co, g.KH, g.KW, g.stride, g.pad, g.reflect, g.precision)
    hit = _CASTS.get(key)
    if hit is None:
        name = plan_name(pass_id, g)
        hit = name.startswith(("conv_fwd_dma_kernel", "conv_dgrad_dma_kernel")[pass_id]) and ", true" in name
        # ... and (round 6) the forward pass of the weight-streaming layers, whose GEMM gathers its rows from float16(x) itself
        hit = _CASTS[key] = hit or (pass_id == 0 and name.startswith("hgemm_sa_kernel<false, true>"))
    return hit


def conv_fwd(g: ConvGeom, x, w, bias=None, act=ACT_NONE, u=None, v_out=None, v_filled=False):
    """v_filled: v_out already holds float16(x) (written by x's producer; only where precast_ok(0, g))."""
    lib = _lib.load()
    y = torch.empty(g.B, g.OH, g.OW, g.Co, dtype=torch.float32, device=x.device)
    ws = _ws(lib.mg_conv_fwd_workspace(g), x.device)
    if PROFILER is not None:
        PROFILER.begin(0, g)
    _lib.check(lib.mg_conv_fwd_w(g, _lib.ptr(x), _lib.ptr(w), _lib.ptr(bias), _lib.ptr(y), act, _lib.ptr(ws),
                                 ws.numel(), _lib.stream(), _tiles(u, v_out, None, None, TILES_V_FILLED if v_filled else 0)),
               "mg_conv_fwd")
    if PROFILER is not None:
        PROFILER.end()
    return y


def wino_vnext_ok(g: ConvGeom) -> bool:
    """True when conv_fwd_instnorm can also write the next 3x3 layer's Winograd input image (float32 F(2x2,3x3), small maps)."""
    key = ("vn", g.B, g.H, g.W, g.Ci, g.Co, g.KH, g.KW, g.stride, g.pad, g.reflect, g.precision)
    hit = _CASTS.get(key)
    if hit is None:
        hit = _CASTS[key] = bool(_lib.load().mg_conv_wino_vnext_ok(g))
    return hit


def conv_fwd_instnorm(g: ConvGeom, x, w, bias=None, act=ACT_NONE, residual=None, eps=1e-5, u=None, v_out=None, v_filled=False,
                      y16=None, need_raw=True, v_next=None, next_reflect=False):
    """conv + InstanceNorm2d(affine=False) (+ act, + residual) -> (y, y_raw, mean, rstd); one kernel does the Winograd
    inverse transform and the normalisation when the layer and the map size allow (csrc/wino.h: wino_out_norm_kernel).
    need_raw=False (no backward pass will follow): y_raw is None and never written -- a fifth of the fused kernel's HBM bytes.
    v_next (only where wino_vnext_ok(g)): receives B^T y B of the output for a following 3x3 stride-1 pad-1 layer with padding mode
    next_reflect, which then passes it as v_out with v_filled=True."""
    lib = _lib.load()
    y = torch.empty(g.B, g.OH, g.OW, g.Co, dtype=torch.float32, device=x.device)
    y_raw = torch.empty_like(y) if need_raw else None
    mean = torch.empty(g.B, g.Co, dtype=torch.float32, device=x.device)
    rstd = torch.empty(g.B, g.Co, dtype=torch.float32, device=x.device)
    ws = _ws(lib.mg_conv_fwd_instnorm_workspace(g), x.device)
    if PROFILER is not None:
        PROFILER.begin(0, g)
    tiles = _tiles(u, v_out, None, None, TILES_V_FILLED if v_filled else 0)
    if v_next is not None and y16 is not None:
        # the two entry points each write ONE of the side outputs (v_next: float32 layers, y16: autocast layers); a caller that
        # asks for both would get an uninitialised float16 copy attached to the output
        raise ValueError("conv_fwd_instnorm: v_next and y16 are mutually exclusive")
    if v_next is not None:
        _lib.check(lib.mg_conv_fwd_instnorm_next(g, _lib.ptr(x), _lib.ptr(w), _lib.ptr(bias), _lib.ptr(y_raw), eps, act,
                                                 _lib.ptr(residual), _lib.ptr(y), _lib.ptr(mean), _lib.ptr(rstd), _lib.ptr(ws),
                                                 ws.numel(), _lib.stream(), tiles, _lib.ptr(v_next), int(bool(next_reflect))),
                   "mg_conv_fwd_instnorm_next")
    else:
        _lib.check(lib.mg_conv_fwd_instnorm_h(g, _lib.ptr(x), _lib.ptr(w), _lib.ptr(bias), _lib.ptr(y_raw), eps, act,
                                              _lib.ptr(residual), _lib.ptr(y), _lib.ptr(mean), _lib.ptr(rstd), _lib.ptr(ws),
                                              ws.numel(), _lib.stream(), tiles, _lib.ptr(y16)),
                   "mg_conv_fwd_instnorm")
    if PROFILER is not None:
        PROFILER.end()
    return y, y_raw, mean, rstd


def wino_md_from_norm_ok(g: ConvGeom) -> bool:
    """True when instnorm_bwd_wino_md can produce this layer's A dy A^T image (F(2x2,3x3) layer, small map)."""
    key = ("mdn", g.B, g.H, g.W, g.Ci, g.Co, g.KH, g.KW, g.stride, g.pad, g.reflect, g.precision)
    hit = _CASTS.get(key)
    if hit is None:
        hit = _CASTS[key] = bool(_lib.load().mg_conv_wino_md_from_norm_ok(g))
    return hit


def instnorm_bwd_wino_md(g: ConvGeom, gy, y_raw, mean, rstd, act, md):
    """InstanceNorm backward + the data gradient's A dy A^T transform in one kernel: fills md, writes no dy."""
    _lib.check(_lib.load().mg_instnorm_bwd_wino_md(g, _lib.ptr(gy), _lib.ptr(y_raw), _lib.ptr(mean), _lib.ptr(rstd), act,
                                                   _lib.ptr(md), _lib.stream()), "mg_instnorm_bwd_wino_md")


def conv_dgrad(g: ConvGeom, dy, w, bias=None, act=ACT_NONE, u=None, md_out=None, out=None, add=None, md_filled=False):
    """dy None: md_out already holds the layer's A dy A^T image (instnorm_bwd_wino_md).  add: dx += add (a skip connection's
    gradient), inside the last kernel where the layer's path allows.  md_filled: md_out already holds float16(dy) (written by
    dy's producer; only where precast_ok(1, g))."""
    lib = _lib.load()
    dx = torch.empty(g.B, g.H, g.W, g.Ci, dtype=torch.float32, device=w.device) if out is None else out
    ws = _ws(lib.mg_conv_dgrad_workspace(g), w.device)
    if PROFILER is not None:
        PROFILER.begin(1, g)
    _lib.check(lib.mg_conv_dgrad_w(g, _lib.ptr(dy), _lib.ptr(w), _lib.ptr(bias), _lib.ptr(dx), act, _lib.ptr(ws),
                                   ws.numel(), _lib.stream(), _tiles(u, None, md_out, add, TILES_MD_FILLED if md_filled else 0)),
               "mg_conv_dgrad")
    if PROFILER is not None:
        PROFILER.end()
    return dx


def conv_wgrad(g: ConvGeom, x, dy, dw, dbias=None, accumulate=False, v=None, md=None, found_inf=None):
    """dw: float32 buffer of Co*KH*KW*Ci elements in OHWI order (written / accumulated in place).
    found_inf (a float32 device scalar, only where wgrad_checks_finite(g)): set to 1 by the kernel when dw is not finite."""
    lib = _lib.load()
    nbytes = lib.mg_conv_wgrad_workspace(g)
    ws = _ws(nbytes, dw.device)
    if PROFILER is not None:
        PROFILER.begin(2, g)
    _lib.check(lib.mg_conv_wgrad_chk(g, _lib.ptr(x), _lib.ptr(dy), _lib.ptr(dw), _lib.ptr(dbias), int(accumulate),
                                     _lib.ptr(ws), ws.numel(), _lib.stream(), _tiles(None, v, md), _lib.ptr(found_inf)),
               "mg_conv_wgrad")
    if PROFILER is not None:
        PROFILER.end()


def wgrad_checks_finite(g: ConvGeom) -> bool:
    """True when the layer's weight-gradient kernel can do the GradScaler's inf / nan check on its own results
    (mg_conv_wgrad_chk: the float16 A-stationary GEMM of the small-spatial trunk layers)."""
    key = ("wf", g.B, g.H, g.W, g.Ci, g.Co, g.KH, g.KW, g.stride, g.pad, g.reflect, g.precision)
    hit = _CASTS.get(key)
    if hit is None:
        hit = _CASTS[key] = bool(_lib.load().mg_conv_wgrad_checks_finite(g))
    return hit


def wgrad_h16_ok(g: ConvGeom) -> bool:
    """True when the layer's weight gradient can be STORED as float16 by its own kernel (mg_conv_wgrad_h16: the weight-streaming
    trunk layers under --fp16)."""
    key = ("wh", g.B, g.H, g.W, g.Ci, g.Co, g.KH, g.KW, g.stride, g.pad, g.reflect, g.precision)
    hit = _CASTS.get(key)
    if hit is None:
        hit = _CASTS[key] = bool(_lib.load().mg_conv_wgrad_h16_ok(g))
    return hit


def conv_wgrad_h16(g: ConvGeom, x, dy, dw16, accumulate=False, found_inf=None):
    """dw16: float16 buffer of Co*KH*KW*Ci elements in OHWI order (written / accumulated in place); found_inf as in conv_wgrad."""
    lib = _lib.load()
    ws = _ws(lib.mg_conv_wgrad_workspace(g), dw16.device)
    if PROFILER is not None:
        PROFILER.begin(2, g)
    _lib.check(lib.mg_conv_wgrad_h16(g, _lib.ptr(x), _lib.ptr(dy), _lib.ptr(dw16), int(accumulate), _lib.ptr(ws), ws.numel(),
                                     _lib.stream(), _lib.ptr(found_inf)), "mg_conv_wgrad_h16")
    if PROFILER is not None:
        PROFILER.end()


def wgrad_adam_ok(g: ConvGeom) -> bool:
    """True when the layer's weight gradient can run as mg_conv_wgrad_adam_w (Winograd F(2x2,3x3), float32)."""
    key = ("wa", g.B, g.H, g.W, g.Ci, g.Co, g.KH, g.KW, g.stride, g.pad, g.reflect, g.precision)
    hit = _CASTS.get(key)
    if hit is None:
        hit = _CASTS[key] = bool(_lib.load().mg_conv_wgrad_adam_ok(g))
    return hit


def conv_wgrad_adam(g: ConvGeom, x, dy, w, m, v_mom, u, state, beta1, beta2, eps, grad_scale, v=None, md=None):
    """Weight gradient + Adam update of (w, m, v_mom) + refresh of the transformed weights u, in place (include/mdctgan_hip.h:
    mg_conv_wgrad_adam_w).  x / dy may be None when both Winograd images (v, md) are handed over."""
    lib = _lib.load()
    ws = _ws(lib.mg_conv_wgrad_workspace(g), w.device)
    ad = _lib.WinoAdam(_lib.ptr(m), _lib.ptr(v_mom), _lib.ptr(u), _lib.ptr(state), beta1, beta2, eps, grad_scale)
    if PROFILER is not None:
        PROFILER.begin(2, g)
    _lib.check(lib.mg_conv_wgrad_adam_w(g, _lib.ptr(x), _lib.ptr(dy), _lib.ptr(w), ad, _lib.ptr(ws), ws.numel(), _lib.stream(),
                                        _tiles(None, v, md)), "mg_conv_wgrad_adam_w")
    if PROFILER is not None:
        PROFILER.end()


def adam_prime(state, beta1, beta2):
    lib = _lib.load()
    _lib.check(lib.mg_adam_prime(_lib.ptr(state), beta1, beta2, _lib.stream()), "mg_adam_prime")


def colsum(a2d, out, accumulate=False):
    lib = _lib.load()
    M, Cc = a2d.shape
    ws = _ws(lib.mg_colsum_workspace(M, Cc), a2d.device)
    _lib.check(lib.mg_colsum(_lib.ptr(a2d), M, Cc, _lib.ptr(out), int(accumulate), _lib.ptr(ws), ws.numel(),
                             _lib.stream()), "mg_colsum")


def instnorm_fwd(x, act=ACT_NONE, residual=None, eps=1e-5, y16=None):
    """x [B, H, W, C] -> (y, mean [B,C], rstd [B,C]).  y16: a float16 buffer of y's size that also receives float16(y)."""
    lib = _lib.load()
    B, H, W, Cc = x.shape
    y = torch.empty_like(x)
    mean = torch.empty(B, Cc, dtype=torch.float32, device=x.device)
    rstd = torch.empty(B, Cc, dtype=torch.float32, device=x.device)
    ws = _ws(lib.mg_instnorm_workspace(B, H * W, Cc), x.device)
    _lib.check(lib.mg_instnorm_fwd_h(_lib.ptr(x), B, H * W, Cc, eps, act, _lib.ptr(residual), _lib.ptr(y),
                                     _lib.ptr(mean), _lib.ptr(rstd), _lib.ptr(ws), ws.numel(), _lib.stream(), _lib.ptr(y16)),
               "mg_instnorm_fwd")
    return y, mean, rstd


def instnorm_bwd(dy, x, mean, rstd, act=ACT_NONE, out=None, dx16=None, dy2=None):
    """dy2 (optional): a second gradient of the same tensor, added inside the kernels (mg_instnorm_bwd_add)."""
    lib = _lib.load()
    B, H, W, Cc = x.shape
    dx = torch.empty_like(x) if out is None else out
    ws = _ws(lib.mg_instnorm_workspace(B, H * W, Cc), x.device)
    _lib.check(lib.mg_instnorm_bwd_add(_lib.ptr(dy), _lib.ptr(dy2), _lib.ptr(x), _lib.ptr(mean), _lib.ptr(rstd), B, H * W, Cc, act,
                                       _lib.ptr(dx), _lib.ptr(ws), ws.numel(), _lib.stream(), _lib.ptr(dx16)), "mg_instnorm_bwd")
    return dx


def act_bwd(dy, y, act, out=None, dy2=None):
    lib = _lib.load()
    out = torch.empty_like(dy) if out is None else out
    _lib.check(lib.mg_act_bwd_add(_lib.ptr(dy), _lib.ptr(dy2), _lib.ptr(y), _lib.ptr(out), dy.numel(), act, _lib.stream()), "mg_act_bwd")
    return out


def add(a, b, out=None):
    lib = _lib.load()
    out = torch.empty_like(a) if out is None else out
    _lib.check(lib.mg_add(_lib.ptr(a), _lib.ptr(b), _lib.ptr(out), a.numel(), _lib.stream()), "mg_add")
    return out


def avgpool_fwd(x):
    lib = _lib.load()
    B, H, W, Cc = x.shape
    y = torch.empty(B, (H - 1) // 2 + 1, (W - 1) // 2 + 1, Cc, dtype=torch.float32, device=x.device)
    _lib.check(lib.mg_avgpool3s2_fwd(_lib.ptr(x), B, H, W, Cc, _lib.ptr(y), _lib.stream()), "mg_avgpool3s2_fwd")
    return y


def avgpool_bwd(dy, in_shape, out=None):
    lib = _lib.load()
    B, H, W, Cc = in_shape
    dx = torch.empty(B, H, W, Cc, dtype=torch.float32, device=dy.device) if out is None else out
    _lib.check(lib.mg_avgpool3s2_bwd(_lib.ptr(dy), B, H, W, Cc, _lib.ptr(dx), _lib.stream()), "mg_avgpool3s2_bwd")
    return dx


def upsample_fwd(x):
    lib = _lib.load()
    B, H, W, Cc = x.shape
    y = torch.empty(B, 2 * H, 2 * W, Cc, dtype=torch.float32, device=x.device)
    _lib.check(lib.mg_upsample2x_fwd(_lib.ptr(x), B, H, W, Cc, _lib.ptr(y), _lib.stream()), "mg_upsample2x_fwd")
    return y


def upsample_bwd(dy):
    lib = _lib.load()
    B, H2, W2, Cc = dy.shape
    dx = torch.empty(B, H2 // 2, W2 // 2, Cc, dtype=torch.float32, device=dy.device)
    _lib.check(lib.mg_upsample2x_bwd(_lib.ptr(dy), B, H2 // 2, W2 // 2, Cc, _lib.ptr(dx), _lib.stream()),
               "mg_upsample2x_bwd")
    return dx


def dinput_fwd(lr, s, nr0, out=None):
    """lr, s: [B, H, W] or [B, H, W, 1] -> [B, H, W, 3] = (lr, s, 2|s| + nr0)."""
    lib = _lib.load()
    B, H, W = s.shape[:3]
    if out is None:
        out = torch.empty(B, H, W, 3, dtype=torch.float32, device=s.device)
    _lib.check(lib.mg_dinput_fwd(_lib.ptr(lr), _lib.ptr(s), s.numel(), nr0, _lib.ptr(out), _lib.stream()),
               "mg_dinput_fwd")
    return out


def dinput_bwd(dout, s):
    lib = _lib.load()
    ds = torch.empty_like(s)
    _lib.check(lib.mg_dinput_bwd(_lib.ptr(dout), _lib.ptr(s), s.numel(), _lib.ptr(ds), _lib.stream()),
               "mg_dinput_bwd")
    return ds


def pair_fwd(s, nr0):
    lib = _lib.load()
    B, H, W = s.shape[:3]
    out = torch.empty(B, H, W, 2, dtype=torch.float32, device=s.device)
    _lib.check(lib.mg_pair_fwd(_lib.ptr(s), s.numel(), nr0, _lib.ptr(out), _lib.stream()), "mg_pair_fwd")
    return out


def mse_const_fwd(pred, target, scale, loss, accumulate):
    lib = _lib.load()
    ws = _ws(lib.mg_loss_workspace(), pred.device)
    _lib.check(lib.mg_mse_const_fwd(_lib.ptr(pred), pred.numel(), target, scale, _lib.ptr(loss), int(accumulate),
                                    _lib.ptr(ws), _lib.stream()), "mg_mse_const_fwd")


def mse_const_bwd(pred, target, scale, grad_out, out=None):
    lib = _lib.load()
    g = torch.empty_like(pred) if out is None else out
    _lib.check(lib.mg_mse_const_bwd(_lib.ptr(pred), pred.numel(), target, scale, _lib.ptr(grad_out), _lib.ptr(g),
                                    _lib.stream()), "mg_mse_const_bwd")
    return g


LOSS_MSE_CONST, LOSS_L1, LOSS_BCE_CONST = 0, 1, 2
LOSS_MAX_ITEMS = 16


def _loss_items(rows):
    """rows: [(a, b | None, grad | None, zero_tail)] -> the ctypes mg_loss_item array (a's element count is n)."""
    arr = (_lib.LossItem * len(rows))()
    for i, (a, b, grad, tail) in enumerate(rows):
        arr[i] = _lib.LossItem(_lib.ptr(a), _lib.ptr(b), _lib.ptr(grad), a.numel(), int(tail))
    return arr


def loss_multi_fwd(kind, rows, target, scale, loss, accumulate=False):
    """loss (+)= sum_i scale * loss_kind(rows[i]) in one partial + one final launch (mg_loss_multi_fwd); bit-identical to the
    single-tensor calls accumulated in list order."""
    lib = _lib.load()
    ws = _ws(lib.mg_loss_multi_workspace(), loss.device)
    for lo in range(0, len(rows), LOSS_MAX_ITEMS):
        part = rows[lo:lo + LOSS_MAX_ITEMS]
        _lib.check(lib.mg_loss_multi_fwd(kind, _loss_items(part), len(part), float(target), float(scale), _lib.ptr(loss),
                                         int(accumulate or lo > 0), _lib.ptr(ws), ws.numel(), _lib.stream()), "mg_loss_multi_fwd")


def loss_multi_bwd(kind, rows, target, scale, grad_out):
    """rows carry grad buffers (and the number of elements behind each to clear): one launch for all of them."""
    lib = _lib.load()
    for lo in range(0, len(rows), LOSS_MAX_ITEMS):
        part = rows[lo:lo + LOSS_MAX_ITEMS]
        _lib.check(lib.mg_loss_multi_bwd(kind, _loss_items(part), len(part), float(target), float(scale), _lib.ptr(grad_out),
                                         _lib.stream()), "mg_loss_multi_bwd")


def l1_fwd(a, b, scale, loss, accumulate):
    lib = _lib.load()
    ws = _ws(lib.mg_loss_workspace(), a.device)
    _lib.check(lib.mg_l1_fwd(_lib.ptr(a), _lib.ptr(b), a.numel(), scale, _lib.ptr(loss), int(accumulate),
                             _lib.ptr(ws), _lib.stream()), "mg_l1_fwd")


def l1_bwd(a, b, scale, grad_out, out=None):
    lib = _lib.load()
    g = torch.empty_like(a) if out is None else out
    _lib.check(lib.mg_l1_bwd(_lib.ptr(a), _lib.ptr(b), a.numel(), scale, _lib.ptr(grad_out), _lib.ptr(g),
                             _lib.stream()), "mg_l1_bwd")
    return g


def adam_step(p, g, m, v, lr, beta1, beta2, eps, step, grad_scale=1.0):
    lib = _lib.load()
    _lib.check(lib.mg_adam_step(_lib.ptr(p), _lib.ptr(g), _lib.ptr(m), _lib.ptr(v), p.numel(), lr, beta1, beta2, eps,
                                step, grad_scale, _lib.stream()), "mg_adam_step")


def adam_tick(state, beta1, beta2):
    lib = _lib.load()
    _lib.check(lib.mg_adam_tick(_lib.ptr(state), beta1, beta2, _lib.stream()), "mg_adam_tick")


def _stream_probe(name, nbytes):
    """bench.py's KernelTimer also times the optimiser launches (single elementwise kernels on torch's current stream): events on
    that stream around the launch.  nbytes = the algorithmic HBM bytes of the launch.  None in normal operation."""
    begin = getattr(PROFILER, "stream_begin", None) if PROFILER is not None else None
    return begin(name, nbytes) if begin is not None else None


def _stream_probe_end(token):
    if token is not None:
        PROFILER.stream_end(token)


def adam_step_dev(p, g, m, v, state, beta1, beta2, eps, grad_scale=1.0):
    lib = _lib.load()
    tok = _stream_probe("adam_dev_kernel", 28 * p.numel())      # reads p, g, m, v; writes p, m, v
    _lib.check(lib.mg_adam_step_dev(_lib.ptr(p), _lib.ptr(g), _lib.ptr(m), _lib.ptr(v), p.numel(), _lib.ptr(state),
                                    beta1, beta2, eps, grad_scale, _lib.stream()), "mg_adam_step_dev")
    _stream_probe_end(tok)


def scaler_check(g, scaler, slot):
    lib = _lib.load()
    tok = _stream_probe("scaler_check_kernel", 4 * g.numel())
    _lib.check(lib.mg_scaler_check(_lib.ptr(g), g.numel(), _lib.ptr(scaler), slot, _lib.stream()), "mg_scaler_check")
    _stream_probe_end(tok)


def scaler_update(scaler, growth_factor, backoff_factor, growth_interval):
    lib = _lib.load()
    _lib.check(lib.mg_scaler_update(_lib.ptr(scaler), growth_factor, backoff_factor, growth_interval, _lib.stream()),
               "mg_scaler_update")


def adam_tick_amp(state, beta1, beta2, scaler, slot):
    lib = _lib.load()
    _lib.check(lib.mg_adam_tick_amp(_lib.ptr(state), beta1, beta2, _lib.ptr(scaler), slot, _lib.stream()),
               "mg_adam_tick_amp")


def adam_step_amp(p, g, m, v, state, beta1, beta2, eps, grad_scale, scaler, slot):
    lib = _lib.load()
    tok = _stream_probe("adam_dev_kernel (GradScaler)", 28 * p.numel())
    _lib.check(lib.mg_adam_step_amp(_lib.ptr(p), _lib.ptr(g), _lib.ptr(m), _lib.ptr(v), p.numel(), _lib.ptr(state),
                                    beta1, beta2, eps, grad_scale, _lib.ptr(scaler), slot, _lib.stream()),
               "mg_adam_step_amp")
    _stream_probe_end(tok)


def adam_step_h(p, g, m, v, p16, state, beta1, beta2, eps, grad_scale, scaler=None, slot=0):
    """Adam update that also refreshes the float16 shadow p16 of the parameters (scaler: GradScaler state or None)."""
    lib = _lib.load()
    tok = _stream_probe("adam_dev_kernel (float16 shadow)", 30 * p.numel())      # ... + the 2-byte shadow write
    _lib.check(lib.mg_adam_step_h(_lib.ptr(p), _lib.ptr(g), _lib.ptr(m), _lib.ptr(v), _lib.ptr(p16), p.numel(),
                                  _lib.ptr(state), beta1, beta2, eps, grad_scale, _lib.ptr(scaler), slot, _lib.stream()),
               "mg_adam_step_h")
    _stream_probe_end(tok)


def scaler_check_segs(g, g16, segs, nsegs, n_total, nbytes, scaler, slot):
    """mg_scaler_check over a segmented arena (segs: uint8 device tensor holding nsegs mg_grad_seg records)."""
    lib = _lib.load()
    tok = _stream_probe("scaler_check_seg_kernel", nbytes)
    _lib.check(lib.mg_scaler_check_segs(_lib.ptr(g), _lib.ptr(g16), _lib.ptr(segs), nsegs, n_total, _lib.ptr(scaler), slot,
                                        _lib.stream()), "mg_scaler_check_segs")
    _stream_probe_end(tok)


def adam_step_segs(p, g, g16, m, v, p16, segs, nsegs, n_total, nbytes, state, beta1, beta2, eps, grad_scale, scaler=None, slot=0):
    """mg_adam_step_h over a segmented arena: one launch, per-segment gradient source (float32 / float32 rounded through float16 /
    float16 storage).  nbytes: the launch's algorithmic HBM bytes (for bench.py's probe)."""
    lib = _lib.load()
    tok = _stream_probe("adam_seg_kernel", nbytes)
    _lib.check(lib.mg_adam_step_segs(_lib.ptr(p), _lib.ptr(g), _lib.ptr(g16), _lib.ptr(m), _lib.ptr(v), _lib.ptr(p16), _lib.ptr(segs),
                                     nsegs, n_total, _lib.ptr(state), beta1, beta2, eps, grad_scale, _lib.ptr(scaler), slot,
                                     _lib.stream()), "mg_adam_step_segs")
    _stream_probe_end(tok)


def weights_are_casts(g: ConvGeom) -> bool:
    """True when the layer's cached weight image (mg_conv_wino_prepare) is the plain float16 copy of its OHWI weights."""
    if g.precision != _lib.PRECISION_F16:
        return False
    key = ("w", g.B, g.H, g.W, g.Ci, g.Co, g.KH, g.KW, g.stride, g.pad, g.reflect)
    hit = _CASTS.get(key)
    if hit is None:
        hit = _CASTS[key] = wino_weights_bytes(g) == g.Co * g.KH * g.KW * g.Ci * 2
    return hit


# SyncBN (SURVEY 8e opt-in; ddp.enable_sync_batchnorm): when set, the per-slice partial sums of a training-mode BatchNorm
# are all-reduced over this process group between the sums kernel and the apply kernel -- two collectives of
# slices x 2 x C doubles per layer and pass -- and the statistics cover world x R rows.
SYNC_BN_GROUP = None       # (group, world) or None


def _bn_sync(part):
    """part: double [slices, 2, C] partial sums -> (partials of the whole batch, world)."""
    if SYNC_BN_GROUP is None:
        return part, 1
    import torch.distributed as dist
    group, world = SYNC_BN_GROUP
    tot = part.clone()
    dist.all_reduce(tot, op=dist.ReduceOp.SUM, group=group)
    return tot, world


def batchnorm_fwd(x, gamma, beta, running_mean, running_var, eps, momentum, training, residual=None, act=ACT_NONE):
    """x [B, H, W, C] NHWC -> (y, save_mean [C], save_rstd [C])."""
    lib = _lib.load()
    Cc = x.shape[-1]
    R = x.numel() // Cc
    y = torch.empty_like(x)
    mean = torch.empty(Cc, dtype=torch.float32, device=x.device)
    rstd = torch.empty(Cc, dtype=torch.float32, device=x.device)
    part, world = None, 1
    if training:
        part = torch.empty(lib.mg_batchnorm_slices(), 2, Cc, dtype=torch.float64, device=x.device)
        _lib.check(lib.mg_batchnorm_sums(_lib.ptr(x), R, Cc, _lib.ptr(part), _lib.stream()), "mg_batchnorm_sums")
        part, world = _bn_sync(part)
    _lib.check(lib.mg_batchnorm_fwd(_lib.ptr(x), R, Cc, eps, momentum, int(training), _lib.ptr(gamma), _lib.ptr(beta),
                                    _lib.ptr(running_mean), _lib.ptr(running_var), _lib.ptr(residual), act, _lib.ptr(y),
                                    _lib.ptr(mean), _lib.ptr(rstd), _lib.ptr(part), float(R * world), _lib.stream()),
               "mg_batchnorm_fwd")
    return y, mean, rstd


def batchnorm_bwd(dy, x, y, gamma, mean, rstd, act, training, dgamma, dbeta, accumulate, want_dres):
    lib = _lib.load()
    Cc = x.shape[-1]
    R = x.numel() // Cc
    dx = torch.empty_like(x)
    dres = torch.empty_like(x) if want_dres else None
    part = torch.empty(lib.mg_batchnorm_slices(), 2, Cc, dtype=torch.float64, device=x.device)
    _lib.check(lib.mg_batchnorm_bwd_sums(_lib.ptr(dy), _lib.ptr(x), _lib.ptr(y), R, Cc, _lib.ptr(mean), _lib.ptr(rstd), act,
                                         _lib.ptr(part), _lib.stream()), "mg_batchnorm_bwd_sums")
    tot, world = _bn_sync(part) if training else (part, 1)
    _lib.check(lib.mg_batchnorm_bwd(_lib.ptr(dy), _lib.ptr(x), _lib.ptr(y), R, Cc, _lib.ptr(gamma), _lib.ptr(mean),
                                    _lib.ptr(rstd), act, int(training), _lib.ptr(dx), _lib.ptr(dres), _lib.ptr(dgamma),
                                    _lib.ptr(dbeta), int(accumulate), _lib.ptr(part), _lib.ptr(tot), float(R * world),
                                    _lib.stream()), "mg_batchnorm_bwd")
    return dx, dres


def attention_fwd(qkv, emb_h, emb_w, heads, d):
    """qkv [B, fh, fw, 3*heads*d] NHWC -> (out [B, fh, fw, heads*d], P [B, heads, n, n])."""
    lib = _lib.load()
    B, fh, fw, _ = qkv.shape
    n = fh * fw
    out = torch.empty(B, fh, fw, heads * d, dtype=torch.float32, device=qkv.device)
    P = torch.empty(B, heads, n, n, dtype=torch.float32, device=qkv.device)
    _lib.check(lib.mg_attention_fwd(_lib.ptr(qkv), _lib.ptr(emb_h), _lib.ptr(emb_w), B, fh, fw, heads, d, _lib.ptr(out),
                                    _lib.ptr(P), _lib.stream()), "mg_attention_fwd")
    return out, P


def attention_bwd(qkv, emb_h, emb_w, dout, P, heads, d, demb_h, demb_w, accumulate):
    lib = _lib.load()
    B, fh, fw, _ = qkv.shape
    dqkv = torch.empty_like(qkv)
    ws = _ws(lib.mg_attention_bwd_workspace(B, fh, fw, heads, d), qkv.device)
    _lib.check(lib.mg_attention_bwd(_lib.ptr(qkv), _lib.ptr(emb_h), _lib.ptr(emb_w), _lib.ptr(dout), _lib.ptr(P), B, fh,
                                    fw, heads, d, _lib.ptr(dqkv), _lib.ptr(demb_h), _lib.ptr(demb_w), int(accumulate),
                                    _lib.ptr(ws), ws.numel(), _lib.stream()), "mg_attention_bwd")
    return dqkv


def stitch_segments(audio, segment_length: int, gen_overlap: int = 0):
    """generate_audio.py:40-53 on the device: audio [n_seg, 1, 1, T] (or [n_seg, T]) float32 / float64 -> [1, total]."""
    lib = _lib.load()
    a = audio.reshape(audio.shape[0], -1)
    if a.shape[1] != segment_length:
        raise ValueError("segments are %d samples long, segment_length says %d" % (a.shape[1], segment_length))
    if a.dtype not in (torch.float32, torch.float64):
        a = a.float()
    a = a.contiguous()
    n = lib.mg_stitch_length(a.shape[0], segment_length, gen_overlap)
    if n <= 0:
        raise ValueError("invalid stitching geometry: n_seg=%d segment_length=%d gen_overlap=%d"
                         % (a.shape[0], segment_length, gen_overlap))
    out = torch.empty(1, n, dtype=a.dtype, device=a.device)
    _lib.check(lib.mg_stitch_segments(_lib.ptr(a), a.shape[0], segment_length, gen_overlap, _lib.ptr(out),
                                      int(a.dtype == torch.float64), _lib.stream()), "mg_stitch_segments")
    return out
