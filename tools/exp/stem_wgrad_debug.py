import torch, sys
sys.path.insert(0, "/root/repo")
from peclr_amd import _capi as capi
DEV="cuda:0"
g = torch.Generator().manual_seed(1)
n,h,w = 2,64,64
x = torch.randn(n,3,h,w,generator=g).to(DEV).contiguous(memory_format=torch.channels_last)
gy = torch.randn(n,64,h//2,w//2,generator=g).to(DEV).contiguous(memory_format=torch.channels_last)
wt = torch.zeros(64,3,7,7,device=DEV)
dw = capi.stem_wgrad(gy,x)
ref = torch.ops.aten.convolution_backward(gy.double(), x.double(), wt.double(), None,[2,2],[3,3],[1,1],False,[0,0],1,[False,True,False])[1]
d = (dw.double()-ref).abs()
print("max err", float(d.max()), "scale", float(ref.abs().max()))
print("per kh max err", [float(d[:,:,k,:].max()) for k in range(7)])
print("per kw max err", [float(d[:,:,:,k].max()) for k in range(7)])
print("per c max err", [float(d[:,c].max()) for c in range(3)])
print("per n-block", [float(d[32*m:32*m+32].max()) for m in range(2)])
print(dw[0,0,0,:4].tolist(), ref[0,0,0,:4].tolist())
r = dw.double()/ref
print("ratio sample", r[0,0,3,:].tolist())
