"""ctypes binding of libmdctgan_hip.so (the C ABI declared in include/mdctgan_hip.h).

There is deliberately NO fallback: if the library is missing, or a tensor is not resident in
HBM, the call raises.  PyTorch is used only for device memory, streams and autograd plumbing.
"""
from __future__ import annotations

import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MDCTGAN_HIP_LIB", os.path.join(_HERE, "libmdctgan_hip.so"))   # override: kernel A/B experiments

MG_CODEC_RAW, MG_CODEC_ARCSINH, MG_CODEC_RANGE, MG_CODEC_DB, MG_CODEC_EXPLICIT = 0, 1, 2, 3, 4
ACT_NONE, ACT_RELU, ACT_LRELU02, ACT_TANH = 0, 1, 2, 3


class HipLibraryError(RuntimeError):
    pass


class ConvGeom(C.Structure):
    _fields_ = [(n, C.c_int) for n in
                ("B", "H", "W", "Ci", "OH", "OW", "Co", "KH", "KW", "stride", "pad", "reflect", "precision")]


PRECISION_F32, PRECISION_F16 = 0, 1


_p, _i, _f, _ll, _sz = C.c_void_p, C.c_int, C.c_float, C.c_longlong, C.c_size_t
_G = C.POINTER(ConvGeom)


class WinoTiles(C.Structure):
    """mg_wino_tiles: caller-held Winograd images (u: transformed weights, v: B^T x B, md: A dy A^T)."""
    _fields_ = [("u", C.c_void_p), ("v", C.c_void_p), ("md", C.c_void_p), ("add", C.c_void_p), ("flags", C.c_uint)]


_W = C.POINTER(WinoTiles)


class LossItem(C.Structure):
    """mg_loss_item: one tensor of a multi-tensor loss call."""
    _fields_ = [("a", C.c_void_p), ("b", C.c_void_p), ("grad", C.c_void_p), ("n", C.c_longlong), ("zero_tail", C.c_longlong)]


class WinoAdam(C.Structure):
    """mg_wino_adam: the optimiser side of mg_conv_wgrad_adam_w (moments, the U image to refresh, the device clock)."""
    _fields_ = [("m", C.c_void_p), ("v", C.c_void_p), ("u", C.c_void_p), ("state", C.c_void_p),
                ("beta1", C.c_float), ("beta2", C.c_float), ("eps", C.c_float), ("grad_scale", C.c_float)]


_WA = C.POINTER(WinoAdam)

# name -> (restype, argtypes); must list every symbol include/mdctgan_hip.h declares
SIGNATURES = {
    "mg_abi_version": (_i, []),
    "mg_conv_geom_size": (_i, []),
    "mg_mdct4_forward": (_i, [_p, _i, _i, _i, _p, _p, _p, _i, _f, _f, _f, _f, _f, _i, _p, _p, _p, _p, _p, _p, _p, _p]),
    "mg_mdct4_num_frames": (_i, [_i, _i]),
    "mg_dct4_image": (_i, [_p, _p, _p]),
    "mg_dct4_image_floats": (_ll, [_i]),
    "mg_imdct4_forward": (_i, [_p, _i, _i, _i, _p, _p, _p, _i, _f, _f, _f, _f, _f, _p, _p, _p, _i, _i, _p, _p]),
    "mg_imdct4_stitched": (_i, [_p, _i, _i, _i, _p, _p, _p, _i, _f, _f, _f, _f, _f, _p, _p, _p, _ll, _i, _i, _ll, _i, _i, _p]),
    "mg_mdct_last_kernel": (C.c_char_p, [_i]),
    "mg_conv_fwd": (_i, [_G, _p, _p, _p, _p, _i, _p, _sz, _p]),
    "mg_conv_fwd_workspace": (_sz, [_G]),
    "mg_conv_wino_weights_bytes": (_sz, [_G]),
    "mg_conv_wino_prepare": (_i, [_G, _p, _p, _p]),
    "mg_conv_wino_tiles_bytes": (_sz, [_G, _i]),
    "mg_conv_fwd_w": (_i, [_G, _p, _p, _p, _p, _i, _p, _sz, _p, _W]),
    "mg_conv_fwd_instnorm_workspace": (_sz, [_G]),
    "mg_conv_wino_md_from_norm_ok": (_i, [_G]),
    "mg_instnorm_bwd_wino_md": (_i, [_G, _p, _p, _p, _p, _i, _p, _p]),
    "mg_conv_fwd_instnorm_w": (_i, [_G, _p, _p, _p, _p, _f, _i, _p, _p, _p, _p, _p, _sz, _p, _W]),
    "mg_conv_fwd_instnorm_h": (_i, [_G, _p, _p, _p, _p, _f, _i, _p, _p, _p, _p, _p, _sz, _p, _W, _p]),
    "mg_conv_fwd_instnorm_next": (_i, [_G, _p, _p, _p, _p, _f, _i, _p, _p, _p, _p, _p, _sz, _p, _W, _p, _i]),
    "mg_conv_wino_vnext_ok": (_i, [_G]),
    "mg_conv_dgrad_w": (_i, [_G, _p, _p, _p, _p, _i, _p, _sz, _p, _W]),
    "mg_conv_wgrad_w": (_i, [_G, _p, _p, _p, _p, _i, _p, _sz, _p, _W]),
    "mg_conv_wgrad_chk": (_i, [_G, _p, _p, _p, _p, _i, _p, _sz, _p, _W, _p]),
    "mg_conv_wgrad_checks_finite": (_i, [_G]),
    "mg_conv_wgrad_adam_ok": (_i, [_G]),
    "mg_conv_wgrad_adam_w": (_i, [_G, _p, _p, _p, _WA, _p, _sz, _p, _W]),
    "mg_conv_dgrad": (_i, [_G, _p, _p, _p, _p, _i, _p, _sz, _p]),
    "mg_conv_dgrad_workspace": (_sz, [_G]),
    "mg_conv_wgrad": (_i, [_G, _p, _p, _p, _p, _i, _p, _sz, _p]),
    "mg_conv_wgrad_workspace": (_sz, [_G]),
    "mg_conv_plan_name": (_i, [_i, _G, C.c_char_p, _i]),
    "mg_conv_plan_splits": (_i, [_i, _G]),
    "mg_conv_plan_flops": (C.c_double, [_i, _G]),
    "mg_probe_arm": (None, [_p, _p]),
    "mg_resample_length": (_ll, [_ll, _i, _i]),
    "mg_resample": (_i, [_p, _i, _i, _p, _i, _i, _i, _p, _i, _p]),
    "mg_metrics_rows": (_i, [_p, _p, _p, _i, _i, _p, _p]),
    "mg_stft_num_frames": (_i, [_i, _i, _i, _i]),
    "mg_stft_frames": (_i, [_p, _i, _i, _p, _i, _i, _i, _p, _p]),
    "mg_lsd_frames": (_i, [_p, _p, _ll, _i, _p, _p]),
    "mg_frames_window": (_i, [_p, _i, _i, _i, _i, _i, _i, _p, _p, _p]),
    "mg_codec_forward": (_i, [_p, _i, _i, _i, _f, _f, _f, _f, _f, _f, _f, _i, _p, _p, _p, _p, _p, _p, _p]),
    "mg_codec_inverse": (_i, [_p, _i, _i, _i, _f, _f, _f, _f, _f, _f, _f, _p, _p, _p, _p]),
    "mg_overlap_add": (_i, [_p, _i, _i, _i, _i, _i, _p, _i, _p, _i, _i, _p]),
    "mg_stitch_length": (_ll, [_i, _i, _i]),
    "mg_stitch_segments": (_i, [_p, _i, _i, _i, _p, _i, _p]),
    "mg_colsum": (_i, [_p, _ll, _i, _p, _i, _p, _sz, _p]),
    "mg_colsum_workspace": (_sz, [_ll, _i]),
    "mg_instnorm_fwd": (_i, [_p, _i, _i, _i, _f, _i, _p, _p, _p, _p, _p, _sz, _p]),
    "mg_instnorm_bwd": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _p, _p, _sz, _p]),
    "mg_instnorm_fwd_h": (_i, [_p, _i, _i, _i, _f, _i, _p, _p, _p, _p, _p, _sz, _p, _p]),
    "mg_instnorm_bwd_h": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _p, _p, _sz, _p, _p]),
    "mg_instnorm_bwd_add": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, _p, _p, _sz, _p, _p]),
    "mg_instnorm_workspace": (_sz, [_i, _i, _i]),
    "mg_batchnorm_workspace": (_sz, [_i]),
    "mg_batchnorm_slices": (_i, []),
    "mg_batchnorm_sums": (_i, [_p, _i, _i, _p, _p]),
    "mg_batchnorm_fwd": (_i, [_p, _i, _i, _f, _f, _i, _p, _p, _p, _p, _p, _i, _p, _p, _p, _p, C.c_double, _p]),
    "mg_batchnorm_bwd_sums": (_i, [_p, _p, _p, _i, _i, _p, _p, _i, _p, _p]),
    "mg_batchnorm_bwd": (_i, [_p, _p, _p, _i, _i, _p, _p, _p, _i, _i, _p, _p, _p, _p, _i, _p, _p, C.c_double, _p]),
    "mg_attention_fwd": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _p, _p, _p]),
    "mg_attention_bwd": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p, _p, _p, _i, _p, _sz, _p]),
    "mg_attention_bwd_workspace": (_sz, [_i, _i, _i, _i, _i]),
    "mg_act_bwd": (_i, [_p, _p, _p, _ll, _i, _p]),
    "mg_act_bwd_add": (_i, [_p, _p, _p, _p, _ll, _i, _p]),
    "mg_add": (_i, [_p, _p, _p, _ll, _p]),
    "mg_avgpool3s2_fwd": (_i, [_p, _i, _i, _i, _i, _p, _p]),
    "mg_avgpool3s2_bwd": (_i, [_p, _i, _i, _i, _i, _p, _p]),
    "mg_upsample2x_fwd": (_i, [_p, _i, _i, _i, _i, _p, _p]),
    "mg_upsample2x_bwd": (_i, [_p, _i, _i, _i, _i, _p, _p]),
    "mg_dinput_fwd": (_i, [_p, _p, _ll, _f, _p, _p]),
    "mg_cat2_fwd": (_i, [_p, _i, _p, _i, _ll, _p, _p]),
    "mg_cat2_bwd": (_i, [_p, _i, _i, _ll, _p, _p, _p]),
    "mg_dinput_bwd": (_i, [_p, _p, _ll, _p, _p]),
    "mg_pair_fwd": (_i, [_p, _ll, _f, _p, _p]),
    "mg_stats_finalize": (_i, [_p, _ll, _p, _p]),
    "mg_loss_workspace": (_sz, []),
    "mg_mse_const_fwd": (_i, [_p, _ll, _f, _f, _p, _i, _p, _p]),
    "mg_mse_const_bwd": (_i, [_p, _ll, _f, _f, _p, _p, _p]),
    "mg_bce_const_fwd": (_i, [_p, _ll, _f, _f, _p, _i, _p, _p]),
    "mg_bce_const_bwd": (_i, [_p, _ll, _f, _f, _p, _p, _p]),
    "mg_sigmoid_fwd": (_i, [_p, _p, _ll, _p]),
    "mg_sigmoid_bwd": (_i, [_p, _p, _p, _ll, _p]),
    "mg_loss_multi_workspace": (_sz, []),
    "mg_loss_multi_fwd": (_i, [_i, C.POINTER(LossItem), _i, _f, _f, _p, _i, _p, _sz, _p]),
    "mg_loss_multi_bwd": (_i, [_i, C.POINTER(LossItem), _i, _f, _f, _p, _p]),
    "mg_l1_fwd": (_i, [_p, _p, _ll, _f, _p, _i, _p, _p]),
    "mg_l1_bwd": (_i, [_p, _p, _ll, _f, _p, _p, _p]),
    "mg_adam_step": (_i, [_p, _p, _p, _p, _ll, _f, _f, _f, _f, _i, _f, _p]),
    "mg_adam_tick": (_i, [_p, _f, _f, _p]),
    "mg_adam_prime": (_i, [_p, _f, _f, _p]),
    "mg_adam_step_dev": (_i, [_p, _p, _p, _p, _ll, _p, _f, _f, _f, _f, _p]),
    "mg_scaler_check": (_i, [_p, _ll, _p, _i, _p]),
    "mg_scaler_update": (_i, [_p, _f, _f, _i, _p]),
    "mg_adam_tick_amp": (_i, [_p, _f, _f, _p, _i, _p]),
    "mg_adam_step_amp": (_i, [_p, _p, _p, _p, _ll, _p, _f, _f, _f, _f, _p, _i, _p]),
    "mg_adam_step_h": (_i, [_p, _p, _p, _p, _p, _ll, _p, _f, _f, _f, _f, _p, _i, _p]),
    "mg_scaler_check_segs": (_i, [_p, _p, _p, _i, _ll, _p, _i, _p]),
    "mg_adam_step_segs": (_i, [_p, _p, _p, _p, _p, _p, _p, _i, _ll, _p, _f, _f, _f, _f, _p, _i, _p]),
    "mg_conv_wgrad_h16_ok": (_i, [C.POINTER(ConvGeom)]),
    "mg_conv_wgrad_h16": (_i, [C.POINTER(ConvGeom), _p, _p, _p, _i, _p, _sz, _p, _p]),
}

_lib = None


ABI_VERSION = 4       # include/mdctgan_hip.h: mg_abi_version (4: segmented --fp16 optimiser passes, float16-stored weight gradients)
GRAD_F32, GRAD_AUTOCAST, GRAD_F16 = 0, 1, 2      # mg_grad_seg.mode


def load():
    """Load the shared library (once).  Raises HipLibraryError when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HipLibraryError(
            "libmdctgan_hip.so not found at %s -- run `python -m mdctgan_amd.build` "
            "(there is no CPU / eager fallback for the hot path)" % LIB_PATH)
    try:
        lib = C.CDLL(LIB_PATH)
    except OSError as e:  # pragma: no cover
        raise HipLibraryError("cannot load %s: %s" % (LIB_PATH, e)) from e
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise HipLibraryError("libmdctgan_hip.so lacks symbol %s (stale build?)" % name) from e
        fn.restype, fn.argtypes = res, args
    if lib.mg_abi_version() != ABI_VERSION:
        raise HipLibraryError("libmdctgan_hip.so speaks C-ABI version %d, these bindings version %d (stale build: run "
                              "`python -m mdctgan_amd.build`)" % (lib.mg_abi_version(), ABI_VERSION))
    _lib = lib
    return lib


def check(rc: int, what: str):
    if rc == 0:
        return
    if rc == -1:
        raise ValueError("%s: argument rejected by the HIP library (MG_ERR_ARG)" % what)
    if rc == -2:
        raise NotImplementedError("%s: configuration not supported by the HIP kernels (MG_ERR_UNSUPPORTED)" % what)
    raise HipLibraryError("%s: HIP error %d" % (what, rc))


def ptr(t):
    """Device pointer of a tensor (None -> NULL).  Refuses host tensors: the hot path has no CPU leg."""
    if t is None:
        return None
    if not t.is_cuda:
        raise HipLibraryError("the MI355X hot path needs device tensors (got a %s tensor); "
                              "there is no CPU fallback" % t.device)
    # the kernels address their operands densely: a strided view (a channel slice of an NHWC pair, say) must go through
    # f32c() / .contiguous() first -- handing out its data_ptr would silently read the neighbouring elements
    if not (t.is_contiguous() or _dense(t)):
        raise HipLibraryError("the HIP kernels take dense tensors; got a strided view of shape %s, strides %s"
                              % (tuple(t.shape), tuple(t.stride())))
    return t.data_ptr()


def _dense(t):
    if t.dim() == 4 and t.is_contiguous(memory_format=torch.channels_last):
        return True
    if t.dim() == 5 and t.is_contiguous(memory_format=torch.channels_last_3d):
        return True
    # any permutation of a dense block (NHWC views of channels_last storage and the like)
    n, expect = t.numel(), 1
    if n == 0:
        return True
    for size, stride in sorted(((s, st) for s, st in zip(t.shape, t.stride()) if s != 1), key=lambda p: p[1]):
        if stride != expect:
            return False
        expect *= size
    return True


def stream():
    return torch.cuda.current_stream().cuda_stream


def f32c(t, name="tensor"):
    """float32 + contiguous, converting when needed (plumbing only)."""
    if t.dtype != torch.float32:
        t = t.float()
    return t if t.is_contiguous() else t.contiguous()


_workspaces = {}


def workspace(nbytes: int, device) -> torch.Tensor:
    """Persistent scratch per (device, launch stream), grown on demand; launches on one stream are ordered."""
    if torch.device(device).type != "cuda":
        raise HipLibraryError("the MI355X hot path needs device tensors (got a %s tensor); "
                              "there is no CPU fallback" % device)
    dev_index = torch.device(device).index if torch.device(device).index is not None else torch.cuda.current_device()
    key = (dev_index, torch.cuda.current_stream(dev_index).cuda_stream)     # one scratch per launch stream
    ws = _workspaces.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(max(int(nbytes * 1.25), 1 << 22), dtype=torch.uint8, device=device)
        _workspaces[key] = ws
    return ws
