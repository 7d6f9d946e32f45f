import sys
sys.path.insert(0, 'dl-unet_amd'); sys.path.insert(0, '.')
import torch, torch.nn.functional as F
import _hip
L = _hip.lib()
def nhwc(t): return t.permute(0,2,3,1).contiguous().float().cuda()
def nchw(t): return t.permute(0,3,1,2).double().cpu()
def ne(a,b): return ((a-b).abs().max()/b.abs().max()).item()
for (B,H,Ci,Co) in [(2,8,512,256),(2,4,512,256),(2,8,256,128),(1,8,512,256),(2,6,1024,512),(2,8,128,64)]:
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B,Ci,H,H,generator=g,dtype=torch.float64).clamp_min(0).requires_grad_(True)
    w = (torch.randn(Ci,Co,2,2,generator=g,dtype=torch.float64)*0.05).requires_grad_(True)
    dy = torch.randn(B,Co,2*H,2*H,generator=g,dtype=torch.float64)
    F.conv_transpose2d(x,w,stride=2).backward(dy)
    dx_ref = x.grad*(x.detach()>0)
    dx = torch.zeros(B,H,H,Ci,device='cuda'); dw=torch.zeros(Ci,Co,2,2,device='cuda'); db=torch.zeros(Co,device='cuda')
    sc = torch.zeros(L.unet_upconv2_scratch_bytes(B,H,H,Ci,Co),dtype=torch.uint8,device='cuda')
    xd=nhwc(x.detach()); wd=w.detach().float().cuda(); dyd=nhwc(dy)
    _hip.check(L.unet_upconv2_bwd(_hip.ptr(xd),B,H,H,Ci,_hip.ptr(wd),Co,_hip.ptr(dyd),_hip.ptr(dx),_hip.ptr(xd),_hip.ptr(dw),_hip.ptr(db),_hip.ptr(sc),_hip.stream()))
    torch.cuda.synchronize()
    e = (nchw(dx)-dx_ref).abs()
    print((B,H,Ci,Co), "dx err %.3g dw %.3g db %.3g" % (ne(nchw(dx),dx_ref), ne(dw.double().cpu(),w.grad), ne(db.double().cpu(), dy.sum((0,2,3)))),
          "bad chans:", (e>1e-3).sum((0,2,3)).nonzero().flatten().tolist()[:12], "bad pix:", (e>1e-3).sum(1).nonzero().tolist()[:6])
