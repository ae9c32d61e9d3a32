"""Per-layer, per-pass time of the convolution calls of one eager training step (whole op: staging casts, GEMM, split-K
epilogue), grouped by geometry -- the target list for tile plans.
    python scripts/layer_times.py [--config 1|2] [--fp16] [--top 40]"""
import argparse, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch


class OpTimer:
    def __init__(self, ops):
        self.ops, self.rec, self.cur = ops, [], None

    def begin(self, pass_id, g):
        e0 = torch.cuda.Event(enable_timing=True)
        e0.record()
        self.cur = (pass_id, (g.B, g.H, g.W, g.Ci, g.Co, g.KH, g.stride, g.pad, g.reflect, g.precision), self.ops.plan_name(pass_id, g),
                    self.ops.conv_flops(g), e0)

    def end(self):
        e1 = torch.cuda.Event(enable_timing=True)
        e1.record()
        self.rec.append(self.cur + (e1,))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=int, default=2)
    ap.add_argument("--fp16", action="store_true")
    ap.add_argument("--top", type=int, default=45)
    a = ap.parse_args()
    import bench
    from mdctgan_amd import ops, options
    from mdctgan_amd.pix2pixHD_model import create_model
    if a.config == 1:
        net = ["--netG", "global", "--ngf", "64", "--n_downsample_global", "4", "--n_blocks_global", "9", "--n_blocks_attn_g", "0", "--num_D", "2"]
    else:
        net = ["--netG", "local", "--ngf", "64", "--n_downsample_global", "4", "--n_blocks_global", "9", "--n_blocks_local", "3",
               "--n_blocks_attn_g", "2", "--heads_g", "8", "--dim_head_g", "64", "--num_D", "3"]
    if a.fp16:
        net.append("--fp16")
    opt = options.make_opt(*options.SPECTRAL_FLAGS, "--lr_sampling_rate", "12000", *net, "--batchSize", "8", "--gpu_ids", "0")
    model = create_model(opt)
    lr, hr = bench.synth_batch(8, 42, "cuda:0", lr_rate=12000)
    for _ in range(6):
        model.optimize_parameters(lr, hr)
    torch.cuda.synchronize()
    t = OpTimer(ops)
    ops.PROFILER = t
    for _ in range(3):
        model.optimize_parameters(lr, hr)
    ops.PROFILER = None
    torch.cuda.synchronize()
    agg = {}
    for pass_id, key, name, flops, e0, e1 in t.rec:
        k = (pass_id, key, name)
        v = agg.setdefault(k, [0, 0.0, flops])
        v[0] += 1
        v[1] += e0.elapsed_time(e1) * 1e3
    rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
    tot = sum(v[1] for v in agg.values()) / 3
    print("conv ops per step: %.2f ms (event-bracketed whole ops, eager)" % (tot / 1e3))
    for (pass_id, key, name), (n, us, flops) in rows[:a.top]:
        B, H, W, Ci, Co, K, s, p, refl, prec = key
        print("%-5s B%-2d %4dx%-4d Ci%-4d Co%-4d k%d s%d %s | x%-2d %7.1f us/call %6.1f TF | %s" % (
            ("fwd", "dgrad", "wgrad")[pass_id], B, H, W, Ci, Co, K, s, "R" if refl else "Z", n // 3, us / n, flops / (us / n) / 1e6, name))


if __name__ == "__main__":
    main()
