"""Which convolution backward calls receive a producer-written float16 copy of their gradient operand (functional._h16_of)?
(diagnostic for the --fp16 hand-over of MG_TILES_V_FILLED / MG_TILES_MD_FILLED; GPU)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mdctgan_amd import functional as Fh, ops, options
from mdctgan_amd.pix2pixHD_model import create_model
from oracle import nets as onets
opt = options.make_opt(*options.SPECTRAL_FLAGS, "--lr_sampling_rate", "12000", "--netG", "local", "--ngf", "32",
                       "--n_downsample_global", "2", "--n_blocks_global", "2", "--n_blocks_local", "2",
                       "--n_blocks_attn_g", "0", "--num_D", "2", "--ndf", "64", "--batchSize", "2", "--bins", "64",
                       "--segment_length", "16128", "--gpu_ids", "0", "--fp16")
model = create_model(opt)
orig = Fh._conv_backward
def spy(ctx, gy, x, y):
    g = ctx.g
    print("conv_bwd", (g.B, g.H, g.W, g.Ci, g.Co, g.KH, g.stride), "attr", hasattr(gy, "_mg_h16"), "valid", Fh._h16_of(Fh.to_cl(gy)) is not None,
          "same_obj", Fh.to_cl(gy) is gy, "pc1", ops.precast_ok(1, g), "act", ctx.cfg[3], "transposed", ctx.cfg[4], ops.plan_name(1, g)[:40])
    return orig(ctx, gy, x, y)
Fh._conv_backward = spy
g = torch.Generator().manual_seed(1)
hr = (0.1 * torch.randn(2, 16128, generator=g)).cuda(); lr = (0.1 * torch.randn(2, 16128, generator=g)).cuda()
model.optimize_parameters(lr, hr)
print(Fh.H16_STATS)
