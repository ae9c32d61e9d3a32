"""The C-ABI library loads without a GPU and exports every symbol include/mdctgan_hip.h declares;
the ctypes binding table covers exactly that set.  No compute calls here."""
import ctypes
import os
import re

import pytest

from mdctgan_amd import _lib

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    text = open(os.path.join(REPO, "include", "mdctgan_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mg_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_table_agree():
    syms = header_symbols()
    assert len(syms) >= 25
    assert sorted(_lib.SIGNATURES) == syms


def test_library_exports_every_declared_symbol():
    if not os.path.exists(_lib.LIB_PATH):
        from mdctgan_amd import build
        build.build_hip(verbose=False)
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for s in header_symbols():
        assert hasattr(lib, s), s
    assert _lib.load().mg_abi_version() == 4 == _lib.ABI_VERSION
    # ... and nothing else: every exported mg_* symbol is declared (library-internal cross-file helpers are hidden)
    import shutil
    import subprocess
    nm = shutil.which("nm") or shutil.which("llvm-nm") or "/opt/rocm/lib/llvm/bin/llvm-nm"
    if os.path.exists(nm):
        out = subprocess.run([nm, "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True).stdout
        exported = {ln.split()[-1] for ln in out.splitlines() if ln.split() and ln.split()[-1].startswith("mg_")}
        assert exported == set(header_symbols()), (exported ^ set(header_symbols()))
    assert _lib.load().mg_mdct4_num_frames(32512, 512) == 128
    assert _lib.load().mg_mdct4_num_frames(7936, 512) == 32


def test_conv_geom_layout_matches_library_and_integration_doc():
    """The binding's struct, the library's sizeof and the struct INTEGRATION.md tells a maintainer to write agree
    (a 12-field struct would make the library read garbage as `precision`)."""
    assert ctypes.sizeof(_lib.ConvGeom) == _lib.load().mg_conv_geom_size() == 52
    doc = open(os.path.join(REPO, "INTEGRATION.md")).read()
    m = re.search(r"class ConvGeom\(ctypes\.Structure\):.*?for n in \((.*?)\)\]", doc, flags=re.S)
    assert m, "INTEGRATION.md no longer shows the ConvGeom binding"
    fields = re.findall(r'"(\w+)"', m.group(1))
    assert fields == [f[0] for f in _lib.ConvGeom._fields_]


def test_ladder_weight_gradient_plans_fill_the_resident_slots():
    """Host-side planner of the LDS-DMA weight gradient (csrc/conv_dma.h, HISTORY.md section 3 "Round 5"): a CU holds 4 / 3 / 2 workgroups
    of the 64x64 / 64x128, 128x64 / 128x128 tiles; for the stride-2 rungs of configs[1] at the bench's batch the plan must fit in
    whole rounds of them with the last round at least three quarters full (the old fixed ladder of split counts left a round of
    one workgroup per CU), take the row-regular gather, and reproduce itself (the workspace query and the launch call it twice)."""
    from mdctgan_amd import ops
    slots = {(64, 64): 4, (64, 128): 3, (128, 64): 3, (128, 128): 2}
    for prec in (0, _lib.PRECISION_F16):
        for (H, W, Ci, Co) in ((128, 256, 64, 128), (64, 128, 128, 256), (32, 64, 256, 512), (16, 32, 512, 1024)):
            g = ops.conv_geom(8, H, W, Ci, Co, 3, 3, 2, 1, False, prec)
            name = ops.plan_name(2, g)
            m = re.match(r"conv_wgrad_dma_kernel<(\d+), (\d+), (true|false), 2, true>", name)
            assert m, name
            assert (m.group(3) == "true") == (prec == _lib.PRECISION_F16)
            bm, bn = int(m.group(1)), int(m.group(2))
            splits = _lib.load().mg_conv_plan_splits(2, g)
            assert splits == _lib.load().mg_conv_plan_splits(2, g) >= 1
            chunks = 8 * (H // 2) * (W // 2) // (64 if prec else 32)
            assert splits <= chunks
            wgs = (Co // bm) * (9 * Ci // bn) * splits
            per_cu = -(-wgs // 256)
            last = per_cu % slots[(bm, bn)] or slots[(bm, bn)]
            if prec == 0:
                assert 4 * last >= 3 * slots[(bm, bn)] or per_cu >= 2 * slots[(bm, bn)], (name, splits, wgs)
    # a geometry the LDS-DMA kernels do not take reports 0 splits
    g = ops.conv_geom(2, 32, 64, 2, 16, 7, 7, 1, 3, True, 0)
    assert _lib.load().mg_conv_plan_splits(0, g) == 0
    # a float16 forward GEMM so large that EVERY tile shape needs more than 65536 workgroups (6.3 M pixels x 256 channels: 98304
    # tiles of 128 x 128) is still costed unsplit -- the workgroup cap bounds split plans only -- and takes the tile it takes one
    # batch size below the cap, not the uncosted 64 x 64 default
    big, below = (ops.conv_geom(B, 256, 512, 64, 256, 3, 3, 1, 1, False, _lib.PRECISION_F16) for B in (48, 32))
    assert ops.plan_name(0, big) == ops.plan_name(0, below) == "conv_fwd_dma_kernel<128, 128, true, 2>"
    assert _lib.load().mg_conv_plan_splits(0, big) == 1


def test_host_tensors_are_refused():
    import torch
    with pytest.raises(_lib.HipLibraryError):
        _lib.ptr(torch.zeros(4))
    from mdctgan_amd.mdct import MDCT4, kbdwin
    m = MDCT4(512, 256, 512, kbdwin, device="cpu")
    with pytest.raises(_lib.HipLibraryError):
        m(torch.zeros(2, 7936))


def test_unsupported_geometry_raises():
    from mdctgan_amd.mdct import IMDCT4, MDCT4, kbdwin
    MDCT4(2048, 512, 2048, kbdwin, device="cpu")           # legal geometry: the generic path (csrc/codec_generic.hip)
    with pytest.raises(NotImplementedError):
        MDCT4(511, 128, 511, None, device="cpu")            # odd n_fft
    with pytest.raises(AssertionError):
        MDCT4(512, 600, 512, kbdwin, device="cpu")          # hop longer than the window (mdct.py:384)
    with pytest.raises(AssertionError):
        MDCT4(512, 256, 1024, kbdwin, device="cpu")        # window longer than n_fft (mdct.py:383)
    im = IMDCT4(512, 256, 512, kbdwin, device="cpu")
    import torch
    with pytest.raises(AssertionError):
        im(torch.zeros(2, 256))                              # mdct.py:458
    with pytest.raises(AssertionError):
        im(torch.zeros(1, 4, 255))                           # mdct.py:460


def test_kbdwin_is_the_reference_window(golden):
    import numpy as np
    from mdctgan_amd.mdct import kbdwin
    from oracle import transform
    g = golden("g1_kbdwin")
    for n in (512, 1024):
        # same float32 op chain as util/util.py:179-186 -> identical to the oracle's on this machine, and within
        # 2 ulp of the vector captured from the reference on the build container's CPU
        np.testing.assert_array_equal(kbdwin(n).numpy(), transform.kbd_window(n))
        np.testing.assert_allclose(kbdwin(n).numpy(), g["w%d" % n], atol=2.4e-7, rtol=0)
