import sys, os, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import synth_batch, BATCH
from mdctgan_amd import options
from mdctgan_amd.pix2pixHD_model import create_model
opt = options.make_opt(*options.SPECTRAL_FLAGS, "--lr_sampling_rate", "12000", "--netG", "global", "--ngf", "64",
                       "--n_blocks_attn_g", "0", "--num_D", "2", "--batchSize", "8", "--gpu_ids", "0")
model = create_model(opt)
lr, hr = synth_batch(BATCH, 42, "cuda:0")
for _ in range(3): model.optimize_parameters(lr, hr)
torch.cuda.synchronize()
import time
t0=time.perf_counter()
for _ in range(5): model.optimize_parameters(lr, hr)
t1=time.perf_counter(); torch.cuda.synchronize(); t2=time.perf_counter()
print("host enqueue %.1f ms/step, + drain %.1f ms" % ((t1-t0)/5*1e3, (t2-t1)*1e3))
pr = cProfile.Profile(); pr.enable()
for _ in range(3): model.optimize_parameters(lr, hr)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
