import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import nets as onets, step as ostep
sys.path.insert(0, '.')
from bench import synth_batch
gen = torch.Generator().manual_seed(0)
netG = onets.init_weights(onets.build_generator("global", 2, 1, 64, 4, 9, input_size=(128, 256)), gen)
netD = onets.init_weights(onets.MultiscaleDRef(3, 64, 3, 2), gen)
ref = ostep.HotPathRef(netG, netD, ostep.CodecCfg(), num_D=2)
lr, hr = synth_batch(2, 1, "cpu")
print("cpus", os.cpu_count())
for th in (16, 32, 64, 128):
    torch.set_num_threads(th)
    ref.train_step(lr[:1].numpy(), hr[:1].numpy())
    t0 = time.perf_counter(); ref.train_step(lr.numpy(), hr.numpy()); dt = time.perf_counter() - t0
    print("threads %d: B=2 step %.2f s" % (th, dt), flush=True)
