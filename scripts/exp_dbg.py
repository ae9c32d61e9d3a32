import sys, os
sys.path.insert(0, "/root/repo")
os.environ["MG_EXP_DBG"] = "1"
import torch, numpy as np
from mdctgan_amd import ops
B,H,W,Ci,Co=8,8,16,1024,1024
g = ops.conv_geom(B,H,W,Ci,Co,3,3,1,1,True)
x = torch.randn(B,H,W,Ci,device="cuda"); w = torch.randn(Co,3,3,Ci,device="cuda")*0.02
u = ops.wino_weights(g, w)
nwg = 4096
dbg = torch.zeros(nwg*40*5, dtype=torch.int64, device="cuda")
for _ in range(3):
    ops.conv_fwd(g,x,w,dbg.view(torch.float32),0,u)
torch.cuda.synchronize()
d = dbg.cpu().numpy().reshape(nwg,40,5)
used = [i for i in range(nwg) if d[i,0,0] != 0]
print("wgs with data", len(used))
for i in used[:3] + used[len(used)//2:len(used)//2+2]:
    t = d[i]
    n = int((t[:,0]!=0).sum())
    seg = np.diff(np.concatenate([t[:n], t[1:n+1,0:1] if n<40 else t[:n,4:5]],1), axis=1) if False else None
    print("wg", i, "chunks", n)
    for c in range(2, min(n-1, 10)):
        a = t[c]; nxt = t[c+1][0]
        print("   c%2d: s0+stash %5d | s1+loads %5d | s2,s3 %5d | barrier %5d | to next %4d | total %5d" % (c, a[1]-a[0], a[2]-a[1], a[3]-a[2], a[4]-a[3], nxt-a[4], nxt-a[0]))
tot = [ (d[i,n-2,0]-d[i,2,0])/(n-4) for i in used for n in [int((d[i,:,0]!=0).sum())] if n>6]
print("mean clk per chunk", np.mean(tot), "min", np.min(tot), "max", np.max(tot))
