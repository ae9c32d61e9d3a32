"""configs[2] float32, batch 1: (a) is the plain (unfused) step deterministic run to run -- parameters and BatchNorm running
statistics after 3 iterations; (b) after fused iterations, is every persistent transformed-weight image U bit-identical to
mg_conv_wino_prepare of the updated weights; (c) which tensor differs first between fused and unfused."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch  # noqa: E402

import test_fullsize_step_gpu as T  # noqa: E402
from mdctgan_amd import ops, options  # noqa: E402
from mdctgan_amd.pix2pixHD_model import create_model  # noqa: E402

tag = sys.argv[1] if len(sys.argv) > 1 else "configs2"
cfg = T.CONFIGS[tag]
B = cfg["batch"]
lr, hr = T.synth(B, 5)
lr, hr = lr.cuda(), hr.cuda()


def run(n_it, fused):
    if fused:
        os.environ.pop("MG_NO_WINO_ADAM_FUSION", None)
    else:
        os.environ["MG_NO_WINO_ADAM_FUSION"] = "1"
    torch.manual_seed(42)
    opt = options.make_opt(*options.SPECTRAL_FLAGS, "--lr_sampling_rate", "12000", *cfg["flags"], "--batchSize", str(B), "--gpu_ids", "0")
    m = create_model(opt)
    for _ in range(n_it):
        m.optimize_parameters(lr, hr)
    torch.cuda.synchronize()
    return m


def first_diff(a, b, what):
    bad = []
    for (k, x), (_, y) in zip(a.state_dict().items(), b.state_dict().items()):
        if not torch.equal(x, y):
            bad.append((k, float((x.double() - y.double()).abs().max())))
    print("%s: %d tensors differ%s" % (what, len(bad), (": first " + repr(bad[:4])) if bad else ""))


for n_it in (1, 2, 3):
    p1, p2 = run(n_it, False), run(n_it, False)
    first_diff(p1.netG, p2.netG, "unfused vs unfused, %d iterations, G" % n_it)
    f = run(n_it, True)
    first_diff(p1.netG, f.netG, "unfused vs FUSED,   %d iterations, G" % n_it)
    first_diff(p1.netD, f.netD, "unfused vs FUSED,   %d iterations, D" % n_it)
    n_u = n_bad = 0
    for k, p in f.netG.named_parameters():
        u = getattr(p, "_mg_u_persist", None)
        if u is None:
            continue
        n_u += 1
        co, ci = p.shape[0], p.shape[1]
        # trunk geometry: configs2 2048 ch @ 4 x 8, local 128 ch @ 64 x 128; configs1 1024 ch @ 8 x 16
        hw = {2048: (4, 8), 1024: (8, 16), 128: (64, 128)}[co]
        g = ops.conv_geom(B, hw[0], hw[1], ci, co, 3, 3, 1, 1, True)
        want = ops.wino_weights(g, p.detach())
        if want is None or want.numel() != u.numel() or not torch.equal(u, want):
            n_bad += 1
            if n_bad <= 3:
                print("   U image of %s differs from the transform of its weights (ok flag %r, version %r)" % (k, p._mg_u_ok, p._version),
                      None if want is None else float((u - want).abs().max()))
    print("   %d persistent U images, %d differ from mg_conv_wino_prepare(weights)" % (n_u, n_bad))
    del p1, p2, f
    torch.cuda.empty_cache()
