import sys, torch
sys.path.insert(0, '/root/repo')
from mdctgan_amd import ops
n = 182_000_000
p = torch.randn(n, device='cuda'); g = torch.randn(n, device='cuda') * 1e-3; m = torch.zeros(n, device='cuda'); v = torch.zeros(n, device='cuda')
state = torch.zeros(4, dtype=torch.float64, device='cuda'); state[0] = 2e-4
ops.adam_tick(state, 0.5, 0.999)
for _ in range(3): ops.adam_step_dev(p, g, m, v, state, 0.5, 0.999, 1e-8)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): ops.adam_step_dev(p, g, m, v, state, 0.5, 0.999, 1e-8)
e1.record(); torch.cuda.synchronize()
t = e0.elapsed_time(e1) / 10
print("adam %.1f us  %.2f TB/s" % (t * 1e3, n * 28 / t / 1e9))
