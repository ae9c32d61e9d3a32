import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, 'tests')
import numpy as np, torch, torch.nn.functional as F
from test_nets_gpu import hip_g
from mdctgan_amd import networks as N, functional as Fh
gen = torch.Generator().manual_seed(7)
x = torch.rand(2, 2, 32, 256, generator=gen) * 2 - 1
gy = torch.randn(2, 1, 32, 256, generator=gen)
net = hip_g("global")
m = net.model
cap = {}
h = x.cuda()
seq = N.FusedSequence(m)
for st in seq.steps[:-2]:
    sub = N.FusedSequence([]); sub.steps = [st]; h = sub(h)
x26 = h; x26.register_hook(lambda g: cap.__setitem__("g26", g.detach().clone()))
ct = m[27](x26); ct.register_hook(lambda g: cap.__setitem__("gct", g.detach().clone()))
a29 = Fh.instance_norm_act(ct, Fh.ACT_RELU); a29.register_hook(lambda g: cap.__setitem__("g29", g.detach().clone()))
y = m[31](a29, 3, Fh.ACT_TANH)
(y * gy.cuda()).sum().backward()
# recompute in fp64 from captured tensors
ctc = ct.detach().cpu().double().requires_grad_()
a = torch.relu(F.instance_norm(ctc))
a.backward(cap["g29"].cpu().double())
ref_gct = ctc.grad
d = (cap["gct"].cpu().double() - ref_gct).abs()
print("IN bwd from captured: max err %.2e (scale %.2e)" % (d.max().item(), ref_gct.abs().max().item()))
xc = x26.detach().cpu().double().requires_grad_()
w = m[27].weight.detach().cpu().double().contiguous()
yy = F.conv_transpose2d(xc, w, None, stride=2, padding=1, output_padding=1)
yy.backward(cap["gct"].cpu().double())
d = (cap["g26"].cpu().double() - xc.grad).abs()
idx = np.unravel_index(d.numpy().argmax(), d.shape)
print("convT dgrad from captured: max err %.2e (scale %.2e) at %s" % (d.max().item(), xc.grad.abs().max().item(), idx))
print("g26 is_contig_cl", cap["g26"].is_contiguous(memory_format=torch.channels_last), cap["gct"].is_contiguous(memory_format=torch.channels_last), cap["gct"].shape, cap["gct"].stride())
