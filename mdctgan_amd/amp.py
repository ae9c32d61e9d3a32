"""Mixed precision for the hot path: the arithmetic of ``torch.autocast(float16)`` + ``torch.cuda.amp.GradScaler``
as the reference uses them (train.py:65-70, 161-164, 183-199), on the MI355X.

* ``autocast(True)``: convolutions (every FLOP-carrying op of G and D) run with MG_PRECISION_F16 -- operands rounded
  to float16 as they are staged into LDS, ``v_mfma_f32_32x32x16_f16`` products, float32 accumulation, forward and
  data-gradient outputs rounded through float16 (so an overflow becomes inf exactly where a float16 tensor would
  overflow), weight gradients accumulated and kept in float32.  Tensors stay float32 in HBM; normalisation,
  losses and the optimiser run in float32 (autocast runs the losses in float32 as well and keeps float32 master
  weights).
* ``GradScaler``: loss scale, inf / nan check, skipped optimiser steps and scale growth / back-off, all as device
  kernels (no ``.item()``), so the AMP iteration is hipGraph-capturable.
"""
from __future__ import annotations

import contextlib

import torch

from . import _lib, ops

_state = {"precision": _lib.PRECISION_F32}


def current_precision() -> int:
    return _state["precision"]


@contextlib.contextmanager
def autocast(enabled: bool = True):
    prev = _state["precision"]
    _state["precision"] = _lib.PRECISION_F16 if enabled else _lib.PRECISION_F32
    try:
        yield
    finally:
        _state["precision"] = prev


class GradScaler:
    """torch.cuda.amp.GradScaler(init_scale=65536, growth_factor=2, backoff_factor=0.5, growth_interval=2000) with
    its state {scale, growth tracker, found_inf per optimiser} in HBM."""

    SLOTS = 2

    def __init__(self, init_scale=2.0 ** 16, growth_factor=2.0, backoff_factor=0.5, growth_interval=2000, enabled=True,
                 device="cuda"):
        self.enabled = enabled
        self.growth_factor, self.backoff_factor, self.growth_interval = growth_factor, backoff_factor, growth_interval
        self.state = torch.zeros(2 + self.SLOTS, dtype=torch.float32, device=device)
        self.state[0] = init_scale
        self._slots = {}

    def _slot(self, optimizer) -> int:
        """One found_inf slot per live optimiser.  Slots are keyed by the optimiser object (a weak reference, so a
        recycled id() can never alias a dead optimiser's slot); an optimiser that was replaced
        (Pix2PixHDModel.update_fixed_params) hands its slot over through release()."""
        import weakref
        for slot, ref in list(self._slots.items()):
            o = ref()
            if o is optimizer:
                return slot
            if o is None:
                del self._slots[slot]
        for slot in range(self.SLOTS):
            if slot not in self._slots:
                self._slots[slot] = weakref.ref(optimizer)
                return slot
        raise RuntimeError("GradScaler tracks at most %d optimisers" % self.SLOTS)

    def release(self, optimizer):
        """Free the slot of an optimiser that is being discarded."""
        for slot, ref in list(self._slots.items()):
            if ref() is optimizer or ref() is None:
                del self._slots[slot]

    def scale(self, loss):
        return loss * self.state[0] if self.enabled else loss

    def step(self, optimizer):
        if not self.enabled:
            return optimizer.step()
        return optimizer.step(scaler_state=self.state, scaler_slot=self._slot(optimizer))

    def update(self):
        if self.enabled:
            ops.scaler_update(self.state, self.growth_factor, self.backoff_factor, self.growth_interval)

    def get_scale(self) -> float:
        return float(self.state[0].item())

    def state_dict(self):
        s = self.state.cpu()
        return {"scale": float(s[0]), "growth_factor": self.growth_factor, "backoff_factor": self.backoff_factor,
                "growth_interval": self.growth_interval, "_growth_tracker": int(s[1])}

    def load_state_dict(self, d):
        self.growth_factor, self.backoff_factor = d["growth_factor"], d["backoff_factor"]
        self.growth_interval = d["growth_interval"]
        self.state[0] = d["scale"]
        self.state[1] = d["_growth_tracker"]
