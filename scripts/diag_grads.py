import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
sys.path.insert(0, 'tests')
from test_nets_gpu import hip_g, oracle_g
tag = "global"
gen = torch.Generator().manual_seed(7)
x = torch.rand(2, 2, 32, 256, generator=gen) * 2 - 1
gy = torch.randn(2, 1, 32, 256, generator=gen)
grads = {}
acts = {}
for name, dt in (("f64", torch.float64), ("f32", torch.float32)):
    net = oracle_g(tag, dt)
    (net(x.to(dt)) * gy.to(dt)).sum().backward()
    grads[name] = {k: p.grad.numpy() for k, p in net.named_parameters()}
net = hip_g(tag)
(net(x.to("cuda")) * gy.to("cuda")).sum().backward()
for k, p in net.named_parameters():
    g64 = grads["f64"][k].astype(np.float64); g32 = grads["f32"][k]; gh = p.grad.cpu().numpy()
    s = np.abs(g64).max()
    print("%-32s scale %.2e  hip %.2e  f32 %.2e  ratio %.1f" % (k, s, np.abs(gh-g64).max()/s, np.abs(g32-g64).max()/s, np.abs(gh-g64).max()/max(np.abs(g32-g64).max(),1e-30)))
