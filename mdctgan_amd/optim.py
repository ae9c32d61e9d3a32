"""FusedAdam: torch.optim.Adam semantics (reference: pix2pixHD_model.py:350-351, 363-364 -- Adam(lr, betas=
(beta1, 0.999)), eps 1e-8, no weight decay) as ONE multi-hundred-MB elementwise HIP launch (K12) over a flat
float32 arena that holds every parameter, its gradient and both moments contiguously.

* parameters are re-pointed at slices of the arena (keeping their channels_last strides), gradients likewise, so
  the wgrad kernels write straight into the arena and a data-parallel reducer can all-reduce arena slices;
* zero_grad() does not touch memory: it flags every gradient "fresh" so the next wgrad kernel overwrites;
* param_groups[0]['lr'] is honoured every step (Pix2PixHDModel.update_learning_rate edits it).
"""
from __future__ import annotations

import os

import numpy as np
import torch

from . import _lib
from . import functional as Fh
from . import ops


def _padded(n):
    """Arena slice length: every float32 slice AND its float16 shadow slice start 16-byte aligned."""
    return (n + 7) // 8 * 8


def _arena_view(flat, off, p):
    n = p.numel()
    if p.dim() == 4 and p.is_contiguous(memory_format=torch.channels_last) and not p.is_contiguous():
        a, b, c, d = p.shape
        return flat[off:off + n].view(a, c, d, b).permute(0, 3, 1, 2)
    return flat[off:off + n].view(p.shape)


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=2e-4, betas=(0.5, 0.999), eps=1e-8, half_shadow=False):
        """half_shadow (--fp16): keep a float16 copy of the whole parameter arena, rewritten by the Adam kernel itself; the
        autocast convolutions read their weight operand from it (functional._weight_image) instead of casting per layer."""
        params = [p for p in params]
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))
        self._params = [p for g in self.param_groups for p in g["params"]]
        self._step = 0
        self._built = False
        self.grad_scale = 1.0          # 1/world_size under data parallelism (gradients arrive summed)
        self.pre_step_hook = None      # reducer.finish() under data parallelism
        self.shard = None              # ddp.ArenaReducer in "sharded" mode: this rank updates its shards only
        self._pending = None           # ... and the all-gather of the updated shards still in flight
        self.half_shadow = bool(half_shadow)
        self.flat_h = None
        self._scaler_flag = None       # GradScaler.state[2 + slot : 3 + slot] once a scaler has stepped this optimiser
        # --fp16 (round 6): how each parameter's gradient is carried (mg_grad_seg.mode, include/mdctgan_hip.h) -- an autocast
        # layer's weight / bias gradient is a float16 tensor in the reference (train.py:161-164), so it is rounded through float16
        # where it is consumed (GRAD_AUTOCAST) or stored as float16 by its own kernel (GRAD_F16, flat_g16); BatchNorm and
        # position-embedding parameters stay float32 (GRAD_F32: they are float32 under torch.autocast too)
        self.flat_g16 = None
        self._modes = None
        self._seg_cache = {}
        self._g16_allowed = os.environ.get("MG_NO_G16", "0") != "1"
        self._index = {}

    # -- arena ---------------------------------------------------------------------------------
    def _build(self):
        ps = self._params
        if not ps:
            self._built = True
            return
        dev = ps[0].device
        if dev.type != "cuda":
            raise RuntimeError("FusedAdam drives the HIP Adam kernel: parameters must live in HBM")
        offs, total = [], 0
        for p in ps:
            offs.append(total)
            total += _padded(p.numel())
        self.flat_p = torch.zeros(total, dtype=torch.float32, device=dev)
        self.flat_g = torch.zeros(total, dtype=torch.float32, device=dev)
        self.flat_m = torch.zeros(total, dtype=torch.float32, device=dev)
        self.flat_v = torch.zeros(total, dtype=torch.float32, device=dev)
        self.offsets, self.total = offs, total
        self._index = {id(p): i for i, p in enumerate(ps)}
        # device-resident clock {step, lr, step_size, sqrt(bias_correction2)}: keeps the step hipGraph-replayable
        # {step, lr, lr / (1 - b1^step), sqrt(1 - b2^step), 1 - b1^(step+1), sqrt(1 - b2^(step+1))}: the last two are the NEXT step's
        # bias-correction terms, read by the kernels that update weights during backward (functional.fused_adam_scope)
        self.state = torch.zeros(6, dtype=torch.float64, device=dev)
        ops.adam_prime(self.state, *self.param_groups[0]["betas"])
        self._lr_on_device = None
        with torch.no_grad():
            for p, off in zip(ps, offs):
                view = _arena_view(self.flat_p, off, p)
                view.copy_(p.data)
                p.data = view
                gview = _arena_view(self.flat_g, off, p)
                if p.grad is not None:
                    gview.copy_(p.grad)
                    fresh = False
                else:
                    fresh = True
                p.grad = gview
                p._mg_fresh = fresh
                n = p.numel()
                # for the layers whose weight gradient, Adam update and weight transform run as one kernel (mg_conv_wgrad_adam_w)
                p._mg_opt, p._mg_m, p._mg_v, p._mg_u_ok = self, self.flat_m[off:off + n], self.flat_v[off:off + n], None
            if self.half_shadow:
                self.flat_h = self.flat_p.to(torch.float16)
                for p, off in zip(ps, offs):
                    n = p.numel()
                    # flat views in arena (= OHWI memory) order; valid while the parameter's version counter stands still
                    # (the Adam kernel writes both through raw pointers, which does not move it)
                    p._mg_h, p._mg_flat, p._mg_h_version = self.flat_h[off:off + n], self.flat_p[off:off + n], p._version
                self._modes = []
                store16 = self._g16_allowed
                for p in ps:
                    if getattr(p, "_mg_grad_f32", False):
                        self._modes.append(_lib.GRAD_F32)
                    elif store16 and getattr(p, "_mg_g16_ok", False):
                        self._modes.append(_lib.GRAD_F16)
                    else:
                        self._modes.append(_lib.GRAD_AUTOCAST)
                for i, p in enumerate(ps):
                    if i and self._rides_f16(i):
                        self._modes[i] = _lib.GRAD_F16
                if _lib.GRAD_F16 in self._modes:
                    # same element index as the float32 arenas; only the GRAD_F16 parameters' slices are ever touched
                    self.flat_g16 = torch.zeros(total, dtype=torch.float16, device=dev)
                    for p, off, mode in zip(ps, offs, self._modes):
                        p._mg_g16 = self.flat_g16[off:off + p.numel()] if mode == _lib.GRAD_F16 else None
        self._built = True

    def _rides_f16(self, i):
        """A bias whose gradient is identically zero (it feeds an InstanceNorm: functional.mark_bias_feeds_norm -- nobody ever writes
        it) right behind a weight whose gradient is stored as float16 is carried in the float16 arena too: its slot there stays
        zero, the optimiser's segments and a data-parallel reducer's float16 pieces then run across whole blocks of the trunk
        instead of being cut at every 2048-element bias (36 -> 1 pieces on configs[2]'s trunk)."""
        p = self._params[i]
        return (self._modes[i] == _lib.GRAD_AUTOCAST and self._modes[i - 1] == _lib.GRAD_F16 and p.dim() == 1
                and getattr(p, "_mg_zero_grad", False) and not Fh.COMPUTE_DEAD_BIAS_GRADS and getattr(p, "_mg_fresh", True))

    @torch.no_grad()
    def adopt_g16(self, p):
        """A weight learned AFTER the arenas were laid out that its gradient kernel can store float16 (functional._tag_g16 at the
        first forward pass; ddp.attach builds the arenas before any forward pass has run): move it to GRAD_F16."""
        i = self._index.get(id(p))
        if (i is None or not self._g16_allowed or self._modes is None or self._modes[i] != _lib.GRAD_AUTOCAST
                or not getattr(p, "_mg_fresh", True)):
            return
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("a gradient's storage format changed inside a graph capture: run one eager step first")
        if self.flat_g16 is None:
            self.flat_g16 = torch.zeros(self.total, dtype=torch.float16, device=self.flat_p.device)
        off = self.offsets[i]
        p._mg_g16 = self.flat_g16[off:off + p.numel()]
        self._modes[i] = _lib.GRAD_F16
        if i + 1 < len(self._params) and self._rides_f16(i + 1):
            q, off = self._params[i + 1], self.offsets[i + 1]
            q._mg_g16 = self.flat_g16[off:off + q.numel()]
            self._modes[i + 1] = _lib.GRAD_F16
        self._seg_cache = {}

    def disable_g16(self):
        """Gradients back into the float32 arena (a reducer mode that reads flat_g alone: ddp.attach_optimizer calls this)."""
        self._g16_allowed = False
        if self.flat_g16 is None:
            return
        for i, p in enumerate(self._params):
            if self._modes[i] == _lib.GRAD_F16:
                self._modes[i] = _lib.GRAD_AUTOCAST
            p._mg_g16 = None
        self.flat_g16 = None
        self._seg_cache = {}

    def grad_of(self, p):
        """The gradient of p as a float32 tensor shaped like p -- p.grad, or the widened float16-stored gradient."""
        g16 = getattr(p, "_mg_g16", None)
        if g16 is None:
            return p.grad
        return _arena_view(g16.float(), 0, p)

    def _segments(self, kinds, spans):
        """(device table of mg_grad_seg records, count, elements, bytes moved by Adam, any unchecked) for this step's live
        parameters: runs of equal (mode, skip_check) over the stepped parameters, cut to `spans`.  Cached per live set -- the table
        a captured step reads was made by its eager warm-up iterations."""
        ps = self._params
        flags = tuple((k != 0, bool(getattr(p, "_mg_inf_checked", False)) and self.shard is None) for k, p in zip(kinds, ps))
        key = (flags, tuple(spans))
        hit = self._seg_cache.get(key)
        if hit is not None:
            return hit
        runs = []          # [lo, hi, mode, skip]
        n = len(ps)
        for i, p in enumerate(ps):
            if kinds[i] == 0:
                continue
            lo = self.offsets[i]
            hi = self.total if i + 1 == n else self.offsets[i + 1]
            mode = self._modes[i]
            zero = bool(getattr(p, "_mg_known_zero", False))
            skip = int(flags[i][1] or zero)        # an exactly-zero bias gradient needs no look
            if runs and runs[-1][1] == lo and runs[-1][2] == mode and (runs[-1][3] == skip or zero):
                runs[-1][1] = hi
            else:
                runs.append([lo, hi, mode, skip])
        segs = []
        for lo, hi, mode, skip in runs:
            for a, b in spans:
                x, y = max(lo, a), min(hi, b)
                if x < y:
                    segs.append((x, y - x, mode, skip))
        rec = np.zeros(len(segs), dtype=np.dtype([("off", "<i8"), ("n", "<i8"), ("mode", "<i4"), ("skip", "<i4")]))
        for j, sg in enumerate(segs):
            rec[j] = sg
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("the set of stepped parameters changed inside a graph capture: run one eager step with it first")
        table = torch.from_numpy(rec.view(np.uint8).copy()).to(self.flat_p.device) if len(segs) else None
        n_total = sum(sg[1] for sg in segs)
        # p, m, v read + written (24), float16 shadow written (2), gradient read (4 or 2)
        nbytes = sum(sg[1] * (26 + (2 if sg[2] == _lib.GRAD_F16 else 4)) for sg in segs)
        check_bytes = sum(sg[1] * (2 if sg[2] == _lib.GRAD_F16 else 4) for sg in segs if not sg[3])
        hit = self._seg_cache[key] = (table, len(segs), n_total, nbytes, check_bytes)
        return hit

    @torch.no_grad()
    def resync_shadow(self):
        """Re-cast the float16 shadow from the parameter arena.  Needed after any raw write into flat_p that does not go
        through a parameter (dist.broadcast of the arena, an all-gather of updated shards): such writes do not move the
        parameters' version counters, so functional._weight_image would keep serving the pre-write shadow."""
        if self.flat_h is not None:
            self.flat_h.copy_(self.flat_p)

    def can_fuse(self):
        """Weight-side fusion (gradient inverse transform + Adam + next forward transform in the weight-gradient call of the
        Winograd trunk layers) is valid for a single process in float32: no gradient reduction has to see the gradient, no
        GradScaler has to inspect it.  MG_NO_WINO_ADAM_FUSION=1 turns it off."""
        import os
        return (self._built and not self.half_shadow and self.pre_step_hook is None and self.shard is None
                and os.environ.get("MG_NO_WINO_ADAM_FUSION", "0") != "1")

    def producer_flag(self):
        """The found_inf slot the weight-gradient kernels may set themselves (functional._producer_flag), or None.  Known from
        the first GradScaler.step on; single process only -- a reduced gradient can be non-finite on a rank whose own
        contribution was finite, so under data parallelism every rank checks the reduced arena.  MG_NO_PRODUCER_INF_CHECK=1
        turns it off."""
        import os
        if (self._scaler_flag is None or self.pre_step_hook is not None or self.shard is not None
                or os.environ.get("MG_NO_PRODUCER_INF_CHECK", "0") == "1"):
            return None
        return self._scaler_flag

    def arena_slices(self):
        """[(param, offset, padded_numel)] in arena order (used by the data-parallel reducer)."""
        if not self._built:
            self._build()
        return [(p, o, _padded(p.numel())) for p, o in zip(self._params, self.offsets)]

    # -- torch.optim API -----------------------------------------------------------------------
    def zero_grad(self, set_to_none: bool = False):
        if not self._built:
            self._build()
        Fh.mark_fresh(self._params)

    @torch.no_grad()
    def sync_lr(self):
        """Copy param_groups[0]['lr'] into the device-resident clock when it changed.  step() calls this; a captured
        (hipGraph) step never runs step() again, so its replay wrapper calls it before every replay
        (Pix2PixHDModel.make_graphed_step) -- otherwise update_learning_rate() would have no effect on graphed training."""
        if not self._built:
            self._build()
        if not self._params:
            return
        lr = float(self.param_groups[0]["lr"])
        if lr != self._lr_on_device:
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError("learning-rate changes must happen outside graph capture")
            self.state[1] = lr
            self._lr_on_device = lr

    @torch.no_grad()
    def step(self, closure=None, scaler_state=None, scaler_slot=0):
        """scaler_state (mdctgan_amd.amp.GradScaler.state): the gradients hold loss-scaled values; check them for
        inf / nan, divide by the scale inside the Adam kernel and skip the whole update -- step counter included --
        when the check fired (GradScaler.step semantics), all on the device."""
        if not self._built:
            self._build()
        self.finish_pending()
        if self.pre_step_hook is not None:
            self.pre_step_hook()
        from .functional import bump_weight_epoch
        bump_weight_epoch()                   # parameters change through raw pointers: invalidate cached weight images
        self._step += 1
        g = self.param_groups[0]
        (b1, b2), eps = g["betas"], g["eps"]
        self.sync_lr()                        # schedule change (update_learning_rate): refresh the device copy
        # contiguous runs of parameters that received a gradient this step (normally a single run = everything)
        fuse = self.can_fuse()
        kinds = []                            # per parameter: 1 = step it, 0 = nothing to do, 2 = may ride along
        for p in self._params:
            has = (p.grad is not None) and not getattr(p, "_mg_fresh", False)
            if has and fuse and getattr(p, "_mg_known_zero", False) and getattr(p, "_mg_never_stepped", True):
                # a bias whose gradient is identically zero (it feeds an InstanceNorm) and whose moments are still zero: Adam
                # leaves it where it is (m = v = 0 -> the update is lr * 0 / (0 + eps)), exactly.  Between two parameters that
                # are stepped it rides along (one launch over the run instead of one per weight); next to the big weights that
                # were already updated during backward it is left out, which keeps those out of the spans.
                kinds.append(2)
            elif has:
                p._mg_never_stepped = False
                p._mg_u_ok = None          # rewritten through the arena: a transformed-weight image kept for it is stale
                kinds.append(1)
            else:
                kinds.append(0)
        runs, start, last = [], None, None    # [start, last] = first / last stepped parameter of the current run
        for i, k in enumerate(kinds):
            if k == 1:
                if start is None:
                    start = i
                last = i
            elif k == 0 and start is not None:
                runs.append((start, last + 1))
                start = None
        if start is not None:
            runs.append((start, last + 1))
        spans = []
        for a, b in runs:
            lo = self.offsets[a]
            hi = self.total if b == len(self._params) else self.offsets[b]
            spans.append((lo, hi))
        if self.shard is not None:
            spans = self.shard.restrict(spans)         # reduce-scattered gradients: only this rank's shards are complete
        if self._modes is not None:
            # --fp16: one segmented launch each for the GradScaler's check and the update (see __init__)
            table, nsegs, n_total, nbytes, check_bytes = self._segments(kinds, spans)
            if scaler_state is None:
                ops.adam_tick(self.state, b1, b2)
            else:
                if nsegs and check_bytes:
                    ops.scaler_check_segs(self.flat_g, self.flat_g16, table, nsegs, n_total, check_bytes, scaler_state, scaler_slot)
                self._scaler_flag = scaler_state[2 + scaler_slot:3 + scaler_slot]
                if self.shard is not None:
                    self.shard.agree(scaler_state[2 + scaler_slot:3 + scaler_slot])
                ops.adam_tick_amp(self.state, b1, b2, scaler_state, scaler_slot)
            if nsegs:
                ops.adam_step_segs(self.flat_p, self.flat_g, self.flat_g16, self.flat_m, self.flat_v, self.flat_h, table, nsegs,
                                   n_total, nbytes, self.state, b1, b2, eps, self.grad_scale, scaler_state, scaler_slot)
            if self.shard is not None:
                self._pending = self.shard.gather(self.flat_p)
            return None
        if scaler_state is None:
            ops.adam_tick(self.state, b1, b2)
        else:
            for lo, hi in self._check_spans(spans, kinds):
                ops.scaler_check(self.flat_g[lo:hi], scaler_state, scaler_slot)
            self._scaler_flag = scaler_state[2 + scaler_slot:3 + scaler_slot]
            if self.shard is not None:
                self.shard.agree(scaler_state[2 + scaler_slot:3 + scaler_slot])
            ops.adam_tick_amp(self.state, b1, b2, scaler_state, scaler_slot)
        for lo, hi in spans:
            args = (self.flat_p[lo:hi], self.flat_g[lo:hi], self.flat_m[lo:hi], self.flat_v[lo:hi])
            if self.flat_h is not None:
                ops.adam_step_h(*args, self.flat_h[lo:hi], self.state, b1, b2, eps, self.grad_scale, scaler_state, scaler_slot)
            elif scaler_state is None:
                ops.adam_step_dev(*args, self.state, b1, b2, eps, self.grad_scale)
            else:
                ops.adam_step_amp(*args, self.state, b1, b2, eps, self.grad_scale, scaler_state, scaler_slot)
        if self.shard is not None:
            self._pending = self.shard.gather(self.flat_p)
        return None

    def _check_spans(self, spans, kinds):
        """The part of `spans` the GradScaler's inf / nan pass has to read: everything except gradients their own kernel
        already checked (`_mg_inf_checked`).  Exactly-zero bias gradients need no look either, but may sit inside a span."""
        if self.shard is not None or not any(getattr(p, "_mg_inf_checked", False) for p in self._params):
            return spans
        out, start, last = [], None, None
        n = len(self._params)
        for i, p in enumerate(self._params):
            stepped = kinds[i] != 0
            need = stepped and not getattr(p, "_mg_inf_checked", False) and not getattr(p, "_mg_known_zero", False)
            ride = stepped and getattr(p, "_mg_known_zero", False) and not getattr(p, "_mg_inf_checked", False)
            if need:
                if start is None:
                    start = i
                last = i
            elif not ride and start is not None:
                out.append((start, last + 1))
                start = None
        if start is not None:
            out.append((start, last + 1))
        return [(self.offsets[a], self.total if b == n else self.offsets[b]) for a, b in out]

    @torch.no_grad()
    def finish_pending(self):
        """Sharded data parallelism: wait for the all-gather of the updated parameter shards (launched by step(), left
        running under whatever followed) and refresh the float16 shadow of the other ranks' shards.  Every reader of the
        parameters calls this first (Pix2PixHDModel.optimize_parameters / inference / save)."""
        works, self._pending = self._pending, None
        if works:
            for w in works:
                w.wait()
            self.resync_shadow()
