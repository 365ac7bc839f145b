import sys; sys.path.insert(0,'/root/repo')
import torch
from musicgan_amd import ops
dev='cuda:0'
def t(fn, iters=200):
    for _ in range(5): fn()
    g=torch.cuda.CUDAGraph()
    torch.cuda.synchronize()
    with torch.cuda.graph(g):
        for _ in range(20): fn()
    g.replay(); torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): g.replay()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1)/200*1e3
for (n,hw) in ((96,4),(96,8),(96,2),(32,4)):
  for cin,cout in ((8,128),(32,128),(64,128),(128,128),(128,16),(128,32)):
    x=torch.randn(n,cin,hw,hw,device=dev); w=torch.randn(cout,cin,3,3,device=dev)*0.05; b=torch.randn(cout,device=dev)
    wp=ops.pack_conv3x3(w,False)
    us=t(lambda: ops.conv3x3(x,wp,b,cout,lrelu=True))
    fl=18*cin*cout*hw*hw*n
    print(f"n={n} {hw}x{hw} cin={cin:3d} cout={cout:3d}: {us:6.1f} us  {fl/us/1e6:6.1f} TF/s")
# trivial kernel floor
y=torch.empty(1024,device=dev)
print("axpby tiny floor", t(lambda: ops.axpby(1.0, y, out=y)))
