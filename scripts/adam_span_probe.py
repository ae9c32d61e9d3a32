"""Which arena spans does FusedAdam.step launch on configs[1]?  (diagnostic; GPU)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mdctgan_amd import ops, options
from mdctgan_amd.pix2pixHD_model import create_model
import bench

opt = options.make_opt(*options.SPECTRAL_FLAGS, "--lr_sampling_rate", "12000", "--netG", "global", "--ngf", "64",
                       "--n_downsample_global", "4", "--n_blocks_global", "9", "--n_blocks_attn_g", "0", "--num_D", "2",
                       "--batchSize", "8", "--gpu_ids", "0")
model = create_model(opt)
lr, hr = bench.synth_batch(8, 42, "cuda:0", lr_rate=12000)
real = ops.adam_step_dev
log = []
def spy(p, *a, **k):
    log.append(p.numel())
    return real(p, *a, **k)
ops.adam_step_dev = spy
for it in range(4):
    log.clear()
    model.optimize_parameters(lr, hr)
    print("iteration", it, "launches", len(log), "sizes", log)
og = model.optimizer_G
names = {id(p): n for n, p in model.netG.named_parameters()}
for p in og._params[:80]:
    print(names.get(id(p)), p.numel(), "known_zero", getattr(p, "_mg_known_zero", None), "never_stepped", getattr(p, "_mg_never_stepped", None),
          "zero_grad", getattr(p, "_mg_zero_grad", None))
