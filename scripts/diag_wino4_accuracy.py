"""Accuracy of the discriminators' 4x4 layers with and without the 25-position Winograd families (csrc/wino4.h, wino42.h):
rel-L2 and max-abs/max error of fwd / dgrad / wgrad against a float64 CPU convolution, on the layer shapes of the
batch-4 stacked pass of tests/test_fullsize_step_gpu.py.  Run on the GPU box: python scripts/diag_wino4_accuracy.py"""
import os, subprocess, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

SHAPES = [("s0_l1_64_128_s2", 4, 64, 33, 65, 128, 4, 2, 2), ("s0_l2_128_256_s2", 4, 128, 17, 33, 256, 4, 2, 2),
          ("s0_l3_256_512_s1", 4, 256, 9, 17, 512, 4, 1, 2), ("s1_l3_256_512_s1", 4, 256, 17, 33, 512, 4, 1, 2),
          ("s1_l2_128_256_s2", 4, 128, 33, 65, 256, 4, 2, 2)]


def main():
    from mdctgan_amd import ops
    for name, B, Ci, H, W, Co, k, s, p in SHAPES:
        gen = torch.Generator().manual_seed(len(name))
        x = torch.randn(B, Ci, H, W, generator=gen)
        w = torch.randn(Co, Ci, k, k, generator=gen) * 0.02
        x64, w64 = x.double().requires_grad_(), w.double().requires_grad_()
        y64 = torch.nn.functional.conv2d(x64, w64, stride=s, padding=p)
        gy = torch.randn(y64.shape, generator=gen)
        y64.backward(gy.double())
        g = ops.conv_geom(B, H, W, Ci, Co, k, k, s, p, False)
        nh = lambda t: t.permute(0, 2, 3, 1).contiguous()
        xd, wd, gyd = nh(x).cuda(), nh(w).cuda(), nh(gy).cuda()
        y = ops.conv_fwd(g, xd, wd)
        dx = ops.conv_dgrad(g, gyd, wd)
        dw = torch.empty_like(wd)
        ops.conv_wgrad(g, xd, gyd, dw, None)
        out = []
        for got, want in ((y, nh(y64.detach())), (dx, nh(x64.grad)), (dw, nh(w64.grad))):
            d = got.double().cpu() - want
            out.append("%.2e/%.2e" % (d.norm() / want.norm(), d.abs().max() / want.abs().max()))
        print("%-20s %-34s fwd %s dgrad %s wgrad %s" % (name, ops.plan_name(0, g)[:34], *out), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1:
        main()
    else:
        for env in ({}, {"MG_NO_WINOGRAD4": "1", "MG_NO_WINOGRAD42": "1"}):
            print("== env", env, "(rel-L2 / max-abs over max)", flush=True)
            subprocess.run([sys.executable, __file__, "run"], env={**os.environ, **env})
