"""Forced (tile, split) sweep of the LDS-DMA implicit-GEMM passes on the stride-2 ladder shapes (tuning harness; in-process:
the planner reads MG_FORCE_CONV_DMA on every call).   python scripts/tune_ladder.py [--batch 8] [--only down64]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mdctgan_amd import ops
from bench_conv import SHAPES, timeit

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--only", default="down64,down128,down256,down512")
    ap.add_argument("--iters", type=int, default=15)
    ap.add_argument("--splits", default="1,2,3,4,5,6,7,8,9,10,11,12,13,14,15,16,18,20,21,22,24,26,28,30,32,36,40,42,48,56,64")
    ap.add_argument("--f16", action="store_true", help="MG_PRECISION_F16 (the float16 instances)")
    a = ap.parse_args()
    splits = [int(v) for v in a.splits.split(",")]
    for name in a.only.split(","):
        B, H, W, Ci, Co, k, s, p, refl = SHAPES[name]
        B = a.batch
        g = ops.conv_geom(B, H, W, Ci, Co, k, k, s, p, refl, 1 if a.f16 else 0)
        x = torch.randn(B, H, W, Ci, device="cuda"); w = torch.randn(Co, k, k, Ci, device="cuda") * 0.02
        b = torch.zeros(Co, device="cuda"); dy = torch.randn(B, g.OH, g.OW, Co, device="cuda"); dw = torch.empty_like(w)
        fl = ops.conv_flops(g)
        fns = {"fwd": lambda: ops.conv_fwd(g, x, w, b), "dgrad": lambda: ops.conv_dgrad(g, dy, w), "wgrad": lambda: ops.conv_wgrad(g, x, dy, dw, None)}
        for tag, fn in fns.items():
            os.environ.pop("MG_FORCE_CONV_DMA", None)
            t0 = timeit(fn, a.iters)
            res = []
            for bm, bn in ((64, 64), (64, 128), (128, 64), (128, 128)):
                for sp in splits:
                    os.environ["MG_FORCE_CONV_DMA"] = "%d,%d,%d" % (bm, bn, sp)
                    try:
                        pn = ops.plan_name(("fwd", "dgrad", "wgrad").index(tag), g)
                        if "<%d, %d," % (bm, bn) not in pn:
                            continue
                        t = timeit(fn, a.iters)
                    except Exception as e:
                        continue
                    res.append((t, bm, bn, sp))
            os.environ.pop("MG_FORCE_CONV_DMA", None)
            res.sort()
            print("%-8s %-5s default %6.1f us (%5.1f TF) | best: %s" % (name, tag, t0 * 1e6, fl / t0 / 1e12,
                  "  ".join("%dx%d/%d %.1f" % (bm, bn, sp, t * 1e6) for t, bm, bn, sp in res[:10])), flush=True)
            for bm, bn in ((64, 64), (64, 128), (128, 64), (128, 128)):
                row = ["%d:%.0f" % (sp, t * 1e6) for t, m, n, sp in sorted(res, key=lambda r: r[3]) if (m, n) == (bm, bn)]
                if row: print("      %dx%d  %s" % (bm, bn, " ".join(row)), flush=True)

if __name__ == "__main__":
    main()
