import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, 'tests')
import numpy as np, torch
from test_nets_gpu import hip_g, oracle_g
from mdctgan_amd import networks as N
gen = torch.Generator().manual_seed(7)
x = torch.rand(2, 2, 32, 256, generator=gen) * 2 - 1
gy = torch.randn(2, 1, 32, 256, generator=gen)
bound = [3, 6, 9, 12, 15, 16, 17, 20, 23, 26, 29, 32]
gr = {}
for name, dt in (("f64", torch.float64), ("f32", torch.float32)):
    net = oracle_g("global", dt)
    h = x.to(dt); d = {}
    for i, m in enumerate(net.model):
        h = m(h)
        if i in bound:
            h.register_hook(lambda g, i=i, d=d: d.__setitem__(i, g.numpy().copy()))
    (h * gy.to(dt)).sum().backward()
    gr[name] = d
net = hip_g("global")
seq = N.FusedSequence(net.model)
h = x.cuda(); dh = {}
for st, bi in zip(seq.steps, bound):
    sub = N.FusedSequence([]); sub.steps = [st]
    h = sub(h)
    h.register_hook(lambda g, bi=bi: dh.__setitem__(bi, g.detach().cpu().numpy().astype(np.float64)))
(h * gy.cuda()).sum().backward()
for bi in bound[::-1]:
    f64 = gr["f64"][bi]; f32 = gr["f32"][bi]; got = dh[bi]
    s = np.abs(f64).max()
    d = np.abs(got - f64); idx = np.unravel_index(d.argmax(), d.shape)
    print("grad at model[%2d] out  scale %.2e hip %.2e f32 %.2e  worst %s" % (bi, s, d.max()/s, np.abs(f32-f64).max()/s, idx))
