"""Which Python lines launch torch (at::native) kernels inside one optimize_parameters() of a bench configuration: the step's own
HIP kernels go through the C ABI, everything torch launches beside them (accumulate-adds, fills, copies) is overhead to remove.
    python scripts/trace_aten_ops.py [--config 2] [--fp16]
Prints, per aten op that launched a device kernel: calls per step, device time per step, and the innermost repo frames."""
import argparse
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402

from bench import BATCH, synth_batch  # noqa: E402
from mdctgan_amd import options  # noqa: E402
from mdctgan_amd.pix2pixHD_model import create_model  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--config", type=int, default=2)
ap.add_argument("--fp16", action="store_true")
ap.add_argument("--steps", type=int, default=3)
args = ap.parse_args()
if args.config == 1:
    net = ["--netG", "global", "--ngf", "64", "--n_downsample_global", "4", "--n_blocks_global", "9", "--n_blocks_attn_g", "0", "--num_D", "2"]
else:
    net = ["--netG", "local", "--ngf", "64", "--n_downsample_global", "4", "--n_blocks_global", "9", "--n_blocks_local", "3",
           "--n_blocks_attn_g", "2", "--heads_g", "8", "--dim_head_g", "64", "--num_D", "3"]
opt = options.make_opt(*options.SPECTRAL_FLAGS, "--lr_sampling_rate", "12000", *net, "--batchSize", str(BATCH), "--gpu_ids", "0",
                       *(["--fp16"] if args.fp16 else []))
model = create_model(opt)
if args.fp16:
    model.scaler.state[0] = 1024.0
lr, hr = synth_batch(BATCH, 42, "cuda:0")
for _ in range(4):
    model.optimize_parameters(lr, hr)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    for _ in range(args.steps):
        model.optimize_parameters(lr, hr)
    torch.cuda.synchronize()
repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
agg = collections.defaultdict(lambda: [0, 0.0, ""])
for ev in prof.events():
    if ev.device_time_total <= 0 or not ev.name.startswith("aten::"):
        continue
    frames = [f for f in (ev.stack or []) if repo in f or "mdctgan_amd" in f]
    site = " <- ".join(f.replace(repo + "/", "") for f in frames[:3]) or "(no repo frame)"
    shapes = str(ev.input_shapes)[:60]
    key = (ev.name, site, shapes)
    a = agg[key]
    a[0] += 1
    a[1] += ev.device_time_total
rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
print("%-18s %8s %10s  %s" % ("op", "calls/it", "us/it", "site"))
tot = 0.0
for (name, site, shapes), (n, us, _) in rows[:60]:
    tot += us / args.steps
    print("%-18s %8.1f %10.1f  %s  %s" % (name, n / args.steps, us / args.steps, shapes, site))
print("total aten device time per iteration (listed): %.1f us" % tot)
