import sys, os
sys.path.insert(0, 'dl-unet_amd'); sys.path.insert(0, '.')
import torch, torch.nn.functional as F, numpy as np
import _hip
L = _hip.lib()
B,H,C,K = 1, 21, 64, 64
g = torch.Generator().manual_seed(1)
x = torch.randn(B,C,H,H, generator=g, dtype=torch.float64)
w = torch.randn(K,C,3,3, generator=g, dtype=torch.float64)*0.05
b = torch.zeros(K, dtype=torch.float64)
ref = F.conv2d(x,w,b)
y = torch.zeros(B,H-2,H-2,K, device='cuda')
nb = L.unet_conv3x3_scratch_bytes(C,K)
sc = torch.zeros(nb, dtype=torch.uint8, device='cuda')
xd = x.permute(0,2,3,1).contiguous().float().cuda()
wd = w.float().cuda(); bd = b.float().cuda()
_hip.check(L.unet_conv3x3_fwd(_hip.ptr(xd), H,H,C,0,None,0,B,H,H,_hip.ptr(wd), _hip.ptr(bd), K, 0, _hip.ptr(y), _hip.ptr(sc), _hip.stream()))
torch.cuda.synchronize()
wt = sc.view(torch.float32)[:K*C*9].reshape(K,9,C).cpu()
wref = w.permute(0,2,3,1).reshape(K,9,C).float()
d = (wt != wref)
print("pack mismatches:", d.sum().item(), "rows:", d.any(2).any(1).nonzero().flatten().tolist()[:10], "taps", d.any(2).any(0).nonzero().flatten().tolist(), "chans", d.any(1).any(0).nonzero().flatten().tolist()[:16])
print("wt[0,0,:8]", wt[0,0,:8].tolist()); print("wref[0,0,:8]", wref[0,0,:8].tolist())
yy = y.permute(0,3,1,2).double().cpu()
err = (yy-ref).abs()
print("y err max", err.max().item(), "bad chans", (err>1e-3).sum((0,2,3)).nonzero().flatten().tolist())
# second run of same call
_hip.check(L.unet_conv3x3_fwd(_hip.ptr(xd), H,H,C,0,None,0,B,H,H,_hip.ptr(wd), _hip.ptr(bd), K, 0, _hip.ptr(y), _hip.ptr(sc), _hip.stream()))
torch.cuda.synchronize()
wt = sc.view(torch.float32)[:K*C*9].reshape(K,9,C).cpu()
print("2nd run pack mismatches:", (wt != wref).sum().item())
yy = y.permute(0,3,1,2).double().cpu(); err=(yy-ref).abs()
print("2nd y err max", err.max().item(), "bad chans", (err>1e-3).sum((0,2,3)).nonzero().flatten().tolist())
