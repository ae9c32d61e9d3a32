"""Micro-benchmark of the implicit-GEMM passes on the layer shapes of configs[1] (tuning harness).
    python scripts/bench_conv.py [--iters 20] [--only bottleneck]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mdctgan_amd import ops

SHAPES = {   # name: (B, H, W, Ci, Co, k, stride, pad, reflect)
    "bottleneck": (8, 8, 16, 1024, 1024, 3, 1, 1, True),
    "down512": (8, 16, 32, 512, 1024, 3, 2, 1, False),
    "down256": (8, 32, 64, 256, 512, 3, 2, 1, False),
    "down128": (8, 64, 128, 128, 256, 3, 2, 1, False),
    "down64": (8, 128, 256, 64, 128, 3, 2, 1, False),
    "d256_512": (8, 17, 33, 256, 512, 4, 1, 2, False),
    "d128_256": (8, 33, 65, 128, 256, 4, 2, 2, False),
    "d64_128": (8, 65, 129, 64, 128, 4, 2, 2, False),
    "d1_256_512": (8, 9, 17, 256, 512, 4, 1, 2, False),     # second discriminator scale (64x128 input)
    "d1_128_256": (8, 17, 33, 128, 256, 4, 2, 2, False),
    "d1_64_128": (8, 33, 65, 64, 128, 4, 2, 2, False),
    "bottleneck_b64": (64, 8, 16, 1024, 1024, 3, 1, 1, True),   # configs[4]: the same layer at the inference batch
    "trunk2048": (8, 4, 8, 2048, 2048, 3, 1, 1, True),      # configs[2]: trunk ResNet blocks of the LocalEnhancer
    "local128": (8, 64, 128, 128, 128, 3, 1, 1, True),      #             half-resolution local blocks
    "down1024": (8, 8, 16, 1024, 2048, 3, 2, 1, False),     #             last rung of its 64 -> 2048 ladder
    "d256_512_b16": (16, 17, 33, 256, 512, 4, 1, 2, False),  #             stacked [fake, real] discriminator pass
    "d3_64": (8, 128, 256, 3, 64, 4, 2, 2, False),          # first discriminator layers
    "d1_3_64": (8, 64, 128, 3, 64, 4, 2, 2, False),
    "head": (8, 128, 256, 64, 1, 7, 1, 3, True),
    "dlast": (8, 18, 34, 512, 1, 4, 1, 2, False),
    "stem": (8, 128, 256, 2, 64, 7, 1, 3, True),
}


def timeit(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--only", default="")
    ap.add_argument("--f16", action="store_true", help="MG_PRECISION_F16 (autocast arithmetic)")
    a = ap.parse_args()
    for name, (B, H, W, Ci, Co, k, s, p, refl) in SHAPES.items():
        if a.only and name not in a.only.split(","):
            continue
        B = int(os.environ.get("MG_BENCH_BATCH", B))
        g = ops.conv_geom(B, H, W, Ci, Co, k, k, s, p, refl, 1 if a.f16 else 0)
        x = torch.randn(B, H, W, Ci, device="cuda")
        w = torch.randn(Co, k, k, Ci, device="cuda") * 0.02
        b = torch.randn(Co, device="cuda")
        dy = torch.randn(B, g.OH, g.OW, Co, device="cuda")
        dw = torch.empty_like(w)
        fl = ops.conv_flops(g)
        res = []
        for tag, fn in (("fwd", lambda: ops.conv_fwd(g, x, w, b)), ("dgrad", lambda: ops.conv_dgrad(g, dy, w)),
                        ("wgrad", lambda: ops.conv_wgrad(g, x, dy, dw, None))):
            t = timeit(fn, a.iters)
            res.append("%s %7.1f us %6.1f TF [%s]" % (tag, t * 1e6, fl / t / 1e12, ops.plan_name(("fwd", "dgrad", "wgrad").index(tag), g).split("kernel")[1]))
        print("%-11s %6.2f GF | %s" % (name, fl / 1e9, " | ".join(res)), flush=True)


if __name__ == "__main__":
    main()
