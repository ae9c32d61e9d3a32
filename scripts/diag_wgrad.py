import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, 'tests')
import numpy as np, torch, torch.nn.functional as F
from test_nets_gpu import hip_g
from mdctgan_amd import functional as Fh, ops
gen = torch.Generator().manual_seed(7)
x = torch.rand(2, 2, 32, 256, generator=gen) * 2 - 1
gy = torch.randn(2, 1, 32, 256, generator=gen)
net = hip_g("global")
cap = {}
layer = net.model[27]
orig_fwd = layer.forward
def fwd(xin, act=0, weight_grad=True):
    cap["x"] = xin.detach()
    y = orig_fwd(xin, act, weight_grad)
    y.register_hook(lambda g: cap.__setitem__("gy", g.detach().clone()))
    return y
layer.forward = fwd
(net(x.cuda()) * gy.cuda()).sum().backward()
xin, g = cap["x"], cap["gy"]
w = layer.weight.detach()
print("x", tuple(xin.shape), "gy", tuple(g.shape), "w", tuple(w.shape))
# reference wgrad from the SAME captured tensors
for dt in (torch.float64, torch.float32):
    xc = xin.cpu().to(dt).contiguous(); gc = g.cpu().to(dt).contiguous()
    wc = w.cpu().to(dt).contiguous().requires_grad_()
    y = F.conv_transpose2d(xc, wc, None, stride=2, padding=1, output_padding=1)
    (y * gc).sum().backward()
    if dt == torch.float64: w64 = wc.grad.numpy()
    else: w32 = wc.grad.numpy()
wh = layer.weight.grad.cpu().numpy().astype(np.float64)
s = np.abs(w64).max()
print("from captured tensors: hip %.2e  cpu32 %.2e (rel to max %.2e)" % (np.abs(wh - w64).max()/s, np.abs(w32 - w64).max()/s, s))
# cancellation yardstick: sum of |terms|
xa = xin.cpu().double().abs(); ga = g.cpu().double().abs()
ya = F.conv_transpose2d(xa, torch.ones_like(w.cpu().double()), None, stride=2, padding=1, output_padding=1)
print("sum|terms| / |result| ~ %.1e" % ((ya * ga).sum().item() / w.numel() / s))
