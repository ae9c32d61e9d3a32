"""Which Python lines make torch launch its own (at::native) kernels inside one optimize_parameters() of a bench configuration:
the step's HIP kernels go through the C ABI; everything torch launches beside them (accumulate-adds, fills, copies) is overhead
to remove.  A TorchDispatchMode logs every aten op with the innermost repo frames (the profiler's with_stack aborts on this
ROCm build).
    python scripts/trace_aten_ops.py [--config 2] [--fp16]"""
import argparse
import collections
import os
import sys
import traceback

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from torch.utils._python_dispatch import TorchDispatchMode  # noqa: E402

from bench import BATCH, synth_batch  # noqa: E402
from mdctgan_amd import options  # noqa: E402
from mdctgan_amd.pix2pixHD_model import create_model  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--config", type=int, default=2)
ap.add_argument("--fp16", action="store_true")
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
repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SKIP = ("aten.view", "aten.detach", "aten._unsafe_view", "aten.t.", "aten.permute", "aten.transpose", "aten.slice", "aten.select",
        "aten.as_strided", "aten.alias", "aten.expand", "aten.unsqueeze", "aten.squeeze", "aten.reshape", "aten.empty", "aten._local_scalar",
        "aten.is_", "aten.split", "aten.unbind", "aten.lift_fresh", "aten.stride", "aten.sym_", "aten.size")
log = collections.Counter()


class Log(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, fargs=(), kwargs=None):
        name = str(func)
        if not name.startswith(SKIP):
            big = 0
            for a in fargs:
                if isinstance(a, torch.Tensor):
                    big = max(big, a.numel())
            frames = [f for f in traceback.extract_stack() if repo in f.filename and "trace_aten_ops" not in f.filename]
            site = " <- ".join("%s:%d" % (f.filename.replace(repo + "/", ""), f.lineno) for f in frames[-3:][::-1])
            log[(name, site, big)] += 1
        return func(*fargs, **(kwargs or {}))


with Log():
    model.optimize_parameters(lr, hr)
torch.cuda.synchronize()
print("%6s %12s  %-28s %s" % ("calls", "max numel", "op", "site (innermost first)"))
for (name, site, big), n in sorted(log.items(), key=lambda kv: (-kv[0][2] * kv[1], kv[0][0])):
    print("%6d %12d  %-28s %s" % (n, big, name, site))
print("total logged aten calls per iteration:", sum(log.values()))
