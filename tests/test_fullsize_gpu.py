"""The convolution passes at the FULL layer sizes of BASELINE configs[1] (the shapes bench.py runs: tuned tile / split
plans, Winograd, 32-deep kernels, per-class split-K, tiled single-channel kernels), where a float64 CPU reference of the
whole tensor would take minutes.  Size-independent properties instead:

* exactness on a sample: 48 random output pixels (all output channels) of the forward pass against a float64
  evaluation of those pixels' receptive fields;
* adjointness: <conv(x), dy> == <x, dgrad(dy)> -- the data gradient is the transpose of the forward map;
* bilinearity: <conv(x; w), dy> == <w, wgrad(x, dy)> -- the weight gradient is the transpose in w.

Inner products are accumulated in float64 on the device.  Float32 passes: 2e-5 relative (3e-5 per element on the
sample); MG_PRECISION_F16 passes: 2e-3 (operands rounded to float16)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"

# (name, B, H, W, Ci, Co, k, stride, pad, reflect) -- scripts/bench_conv.py's table of configs[1] layers
SHAPES = [
    ("bottleneck", 8, 8, 16, 1024, 1024, 3, 1, 1, True),
    ("down512", 8, 16, 32, 512, 1024, 3, 2, 1, False),
    ("down256", 8, 32, 64, 256, 512, 3, 2, 1, False),
    ("down128", 8, 64, 128, 128, 256, 3, 2, 1, False),
    ("down64", 8, 128, 256, 64, 128, 3, 2, 1, False),
    ("stem", 8, 128, 256, 2, 64, 7, 1, 3, True),
    ("head", 8, 128, 256, 64, 1, 7, 1, 3, True),
    ("d3_64_b16", 16, 128, 256, 3, 64, 4, 2, 2, False),
    ("d64_128_b16", 16, 65, 129, 64, 128, 4, 2, 2, False),
    ("d128_256_b16", 16, 33, 65, 128, 256, 4, 2, 2, False),
    ("d256_512_b16", 16, 17, 33, 256, 512, 4, 1, 2, False),
    ("dlast_b16", 16, 18, 34, 512, 1, 4, 1, 2, False),
    ("d1_256_512", 8, 9, 17, 256, 512, 4, 1, 2, False),
]


def dot64(a, b):
    return (a.double() * b.double()).sum().item()


def sample_reference(x, w, bias, B, H, W, Ci, Co, k, s, p, reflect, OH, OW, n=48, seed=0):
    """float64 outputs at n random (b, oy, ox): returns indices and [n, Co] values.  x [B,H,W,Ci], w [Co,k,k,Ci] on CPU."""
    rng = np.random.default_rng(seed)
    idx = np.stack([rng.integers(0, B, n), rng.integers(0, OH, n), rng.integers(0, OW, n)], 1)
    idx[0] = (0, 0, 0)
    idx[1] = (B - 1, OH - 1, OW - 1)                     # corners: padding / reflection on both sides
    xd, wd = x.double(), w.double()
    out = torch.zeros(n, Co, dtype=torch.float64)
    for j, (b, oy, ox) in enumerate(idx):
        acc = bias.double().clone() if bias is not None else torch.zeros(Co, dtype=torch.float64)
        for ky in range(k):
            iy = oy * s - p + ky
            if reflect:
                iy = -iy if iy < 0 else (2 * (H - 1) - iy if iy >= H else iy)
            elif iy < 0 or iy >= H:
                continue
            for kx in range(k):
                ix = ox * s - p + kx
                if reflect:
                    ix = -ix if ix < 0 else (2 * (W - 1) - ix if ix >= W else ix)
                elif ix < 0 or ix >= W:
                    continue
                acc += wd[:, ky, kx, :] @ xd[b, iy, ix, :]
        out[j] = acc
    return idx, out


@pytest.mark.parametrize("prec", ["f32", "f16"])
@pytest.mark.parametrize("shape", SHAPES, ids=[s[0] for s in SHAPES])
def test_full_size_layer_properties(shape, prec):
    from mdctgan_amd import _lib, ops
    name, B, H, W, Ci, Co, k, s, p, reflect = shape
    hp = prec == "f16"
    g = ops.conv_geom(B, H, W, Ci, Co, k, k, s, p, reflect, _lib.PRECISION_F16 if hp else _lib.PRECISION_F32)
    gen = torch.Generator().manual_seed(len(name) + Ci)
    x = torch.randn(B, H, W, Ci, generator=gen)
    w = torch.randn(Co, k, k, Ci, generator=gen) / np.sqrt(Ci * k * k)
    bias = torch.randn(Co, generator=gen)
    dy = torch.randn(B, g.OH, g.OW, Co, generator=gen)
    if hp:   # the properties are stated for the values the kernels actually multiply
        x, w, dy = x.half().float(), w.half().float(), dy.half().float()
    xd, wd, bd, dyd = x.to(DEV), w.to(DEV), bias.to(DEV), dy.to(DEV)
    rtol = 2e-3 if hp else 2e-5

    y = ops.conv_fwd(g, xd, wd, bd)
    idx, want = sample_reference(x, w, bias, B, H, W, Ci, Co, k, s, p, reflect, g.OH, g.OW)
    got = y[idx[:, 0], idx[:, 1], idx[:, 2]].double().cpu()
    tol = (4e-3 if hp else 3e-5) * want.abs().max().item()
    assert (got - want).abs().max().item() <= tol, (name, (got - want).abs().max().item(), want.abs().max().item())

    y0 = ops.conv_fwd(g, xd, wd, None)                           # linear part only for the transpose identities
    lhs = dot64(y0, dyd)
    scale = np.sqrt(dot64(y0, y0) * dot64(dyd, dyd))              # Cauchy-Schwarz scale of the inner product
    dx = ops.conv_dgrad(g, dyd, wd)
    assert abs(lhs - dot64(xd, dx)) <= rtol * scale, (name, "adjoint", lhs, dot64(xd, dx), scale)
    dw = torch.empty(Co, k, k, Ci, device=DEV)
    db = torch.empty(Co, device=DEV)
    ops.conv_wgrad(g, xd, dyd, dw, db)
    assert abs(lhs - dot64(wd, dw)) <= rtol * scale, (name, "wgrad", lhs, dot64(wd, dw), scale)
    assert (db.double().cpu() - dy.double().sum((0, 1, 2))).abs().max().item() <= 1e-4 * np.sqrt(B * g.OH * g.OW) * 4
