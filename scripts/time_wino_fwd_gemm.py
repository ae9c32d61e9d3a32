"""Event-timed Winograd forward GEMM of the 1024-channel 8x16 layer under MG_FORCE_PLAN (library-side probe events)."""
import sys, os
sys.path.insert(0, "/root/repo")
import torch
from mdctgan_amd import ops, _lib
B,H,W,Ci,Co=8,8,16,1024,1024
g = ops.conv_geom(B,H,W,Ci,Co,3,3,1,1,True)
x = torch.randn(B,H,W,Ci,device="cuda"); w = torch.randn(Co,3,3,Ci,device="cuda")*0.02
u = ops.wino_weights(g, w)
lib=_lib.load()
from bench import HipEvents
import ctypes
def timeit(n=30):
    ev=HipEvents(); ts=[]
    for i in range(n+5):
        e0,e1=ev.new(),ev.new()
        lib.mg_probe_arm(e0,e1)
        ops.conv_fwd(g,x,w,None,0,u)
        torch.cuda.synchronize()
        if i>=5: ts.append(ev.elapsed_s(e0,e1)*1e6)
    ts.sort(); return ts[len(ts)//2]
print("PLAN", os.environ.get("MG_FORCE_PLAN","-"), "gemm us", round(timeit(),1))
