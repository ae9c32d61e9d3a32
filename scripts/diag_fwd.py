import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, 'tests')
import numpy as np, torch
from test_nets_gpu import hip_g, oracle_g
from mdctgan_amd import networks as N
gen = torch.Generator().manual_seed(7)
x = torch.rand(2, 2, 32, 256, generator=gen) * 2 - 1
outs = {}
for name, dt in (("f64", torch.float64), ("f32", torch.float32)):
    net = oracle_g("global", dt)
    h = x.to(dt); o = []
    with torch.no_grad():
        for i, m in enumerate(net.model):
            h = m(h); o.append(h.numpy())
    outs[name] = o
net = hip_g("global")
seq = N.FusedSequence(net.model)
h = x.cuda()
bound = [3, 6, 9, 12, 15, 16, 17, 20, 23, 26, 29, 32]
with torch.no_grad():
    for st, bi in zip(seq.steps, bound):
        sub = N.FusedSequence([]); sub.steps = [st]
        h = sub(h)
        got = h.cpu().numpy().astype(np.float64); f64 = outs["f64"][bi]; f32 = outs["f32"][bi]
        s = np.abs(f64).max()
        print("after model[%2d] %-18s scale %.2e hip %.2e f32 %.2e" % (bi, str(tuple(got.shape)), s, np.abs(got-f64).max()/s, np.abs(f32-f64).max()/s))
