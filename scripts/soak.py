"""Soak test: N optimisation steps of configs[1] on synthetic data (hipGraph replay), printing the losses -- checks that
nothing drifts to inf / nan over a few hundred steps in float32 and under --fp16 (GradScaler state included)."""
import sys, os, argparse
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import synth_batch, BATCH
from mdctgan_amd import options
from mdctgan_amd.pix2pixHD_model import create_model

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=300)
ap.add_argument("--fp16", action="store_true")
a = ap.parse_args()
torch.manual_seed(0)
flags = ["--netG", "global", "--ngf", "64", "--n_downsample_global", "4", "--n_blocks_global", "9", "--n_blocks_attn_g", "0",
         "--num_D", "2"] + (["--fp16"] if a.fp16 else [])
opt = options.make_opt(*options.SPECTRAL_FLAGS, "--lr_sampling_rate", "12000", *flags, "--batchSize", str(BATCH), "--gpu_ids", "0")
model = create_model(opt)
batches = [synth_batch(BATCH, 100 + i, "cuda:0") for i in range(4)]
step = model.make_graphed_step(*batches[0], warmup=2)
for it in range(a.steps):
    lr, hr = batches[it % 4]
    ld = step(lr, hr)
    if it % 50 == 0 or it == a.steps - 1:
        vals = {k: float(v) for k, v in ld.items()}
        extra = " scale %.0f" % model.scaler.get_scale() if model.scaler is not None else ""
        print("step %4d " % it + " ".join("%s %.4f" % kv for kv in vals.items()) + extra, flush=True)
        assert all(v == v and abs(v) < 1e6 for v in vals.values()), "loss diverged"
p = torch.cat([q.detach().reshape(-1) for q in model.netG.parameters()])
assert torch.isfinite(p).all()
print("ok: %d steps, |G params| max %.3f" % (a.steps, p.abs().max().item()))
