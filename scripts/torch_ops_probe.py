"""Which torch-native (aten) kernels still run inside one eager G+D step of configs[1], with shapes and Python call sites?
(diagnostic; GPU)   python scripts/torch_ops_probe.py [--fp16] [--config 2]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mdctgan_amd import options
from mdctgan_amd.pix2pixHD_model import create_model
import bench

fp16 = "--fp16" in sys.argv
cfg2 = "--config" in sys.argv and sys.argv[sys.argv.index("--config") + 1] == "2"
if cfg2:
    net = ["--netG", "local", "--ngf", "64", "--n_downsample_global", "4", "--n_blocks_global", "9", "--n_blocks_local", "3",
           "--n_blocks_attn_g", "2", "--heads_g", "8", "--dim_head_g", "64", "--num_D", "3"]
else:
    net = ["--netG", "global", "--ngf", "64", "--n_downsample_global", "4", "--n_blocks_global", "9", "--n_blocks_attn_g", "0",
           "--num_D", "2"]
opt = options.make_opt(*options.SPECTRAL_FLAGS, "--lr_sampling_rate", "12000", *net, "--batchSize", "8", "--gpu_ids", "0",
                       *(["--fp16"] if fp16 else []))
model = create_model(opt)
lr, hr = bench.synth_batch(8, 42, "cuda:0", lr_rate=12000)
for it in range(3):
    model.optimize_parameters(lr, hr)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU], record_shapes=True) as prof:
    model.optimize_parameters(lr, hr)
    torch.cuda.synchronize()
rows = {}
for e in prof.events():
    if not e.name.startswith("aten::"):
        continue
    if e.name in ("aten::empty", "aten::empty_like", "aten::empty_strided", "aten::view", "aten::as_strided", "aten::permute",
                  "aten::reshape", "aten::detach", "aten::alias", "aten::slice", "aten::select", "aten::unsqueeze", "aten::squeeze",
                  "aten::expand", "aten::t", "aten::transpose", "aten::_unsafe_view", "aten::resize_", "aten::contiguous",
                  "aten::result_type", "aten::to", "aten::lift_fresh", "aten::narrow", "aten::unbind", "aten::is_nonzero",
                  "aten::item", "aten::_local_scalar_dense", "aten::view_as", "aten::flatten", "aten::chunk", "aten::split"):
        continue
    key = (e.name, str(e.input_shapes)[:120], "")
    rows[key] = rows.get(key, 0) + 1
for k, n in sorted(rows.items(), key=lambda kv: -kv[1]):
    print("%3d  %-22s %-92s %s" % (n, k[0], k[1], k[2]))
