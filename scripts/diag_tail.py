import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.nn as nn
from mdctgan_amd import networks as N, functional as Fh
from oracle import nets as onets
torch.manual_seed(0)
def tail_o():
    return nn.Sequential(nn.ConvTranspose2d(16, 8, 3, 2, 1, 1), nn.InstanceNorm2d(8), nn.ReLU(),
                         nn.ReflectionPad2d(3), nn.Conv2d(8, 1, 7), nn.Tanh())
def tail_h():
    return nn.Sequential(N.ConvTranspose2d(16, 8, 3, 2, 1, 1), nn.InstanceNorm2d(8), nn.ReLU(True),
                         nn.ReflectionPad2d(3), N.Conv2d(8, 1, 7), nn.Tanh())
gen = torch.Generator().manual_seed(3)
x = torch.randn(2, 16, 16, 128, generator=gen)
gy = torch.randn(2, 1, 32, 256, generator=gen)
res = {}
for name, dt in (("f64", torch.float64), ("f32", torch.float32)):
    net = onets.fill_deterministic(tail_o()).to(dt)
    xx = x.to(dt).requires_grad_()
    hs = []
    h = xx
    for m in net:
        h = m(h); h.retain_grad(); hs.append(h)
    (h * gy.to(dt)).sum().backward()
    res[name] = dict(dx=xx.grad.numpy(), w0=net[0].weight.grad.numpy(), w4=net[4].weight.grad.numpy(),
                     g_in=hs[1].grad.numpy(), g_relu=hs[2].grad.numpy(), g_ct=hs[0].grad.numpy(), y=h.detach().numpy())
net = onets.fill_deterministic(tail_h()).cuda()
xx = x.cuda().requires_grad_()
seq = N.FusedSequence(net)
# run manually to capture intermediates
cap = {}
ct = net[0](xx); ct.register_hook(lambda g: cap.__setitem__("g_ct", g))
inr = Fh.instance_norm_act(ct, Fh.ACT_RELU); inr.register_hook(lambda g: cap.__setitem__("g_relu", g))
y = net[4](inr, 3, Fh.ACT_TANH)
(y * gy.cuda()).sum().backward()
got = dict(dx=xx.grad, w0=net[0].weight.grad, w4=net[4].weight.grad, g_relu=cap.get("g_relu"), g_ct=cap.get("g_ct"), y=y.detach())
for k, v in got.items():
    if v is None:
        print(k, "is None"); continue
    v = v.cpu().numpy().astype(np.float64); f64 = res["f64"][k]; f32 = res["f32"][k]
    s = np.abs(f64).max()
    d = np.abs(v - f64)
    idx = np.unravel_index(d.argmax(), d.shape)
    print("%-8s scale %.2e hip %.2e f32 %.2e  worst idx %s" % (k, s, d.max()/s, np.abs(f32-f64).max()/s, idx))
