import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "dl-unet_amd"))
import torch, torch.nn.functional as F
import _hip
L = _hip.lib()
B, H, C, K = [int(v) for v in (sys.argv[1:5] if len(sys.argv) >= 5 else (2, 21, 64, 64))]
torch.manual_seed(1)
x = torch.randn(B, C, H, H, dtype=torch.float64); dz = torch.randn(B, K, H - 2, H - 2, dtype=torch.float64)
w = torch.zeros(K, C, 3, 3, dtype=torch.float64, requires_grad=True)
F.conv2d(x, w).backward(dz)
ref = w.grad
xg = x.permute(0, 2, 3, 1).contiguous().float().cuda(); dzg = dz.permute(0, 2, 3, 1).contiguous().float().cuda()
wg = torch.zeros(K, C, 3, 3, device="cuda")
dw = torch.empty(K, C, 3, 3, device="cuda"); db = torch.empty(K, device="cuda")
sc = torch.empty(L.unet_conv3x3_bwd_scratch_bytes(B, H, H, C, K), dtype=torch.uint8, device="cuda")
_hip.check(L.unet_set_math(3))
_hip.check(L.unet_conv3x3_bwd(_hip.ptr(xg), H, H, C, 0, None, 0, B, H, H, _hip.ptr(wg), K, _hip.ptr(dzg), None, None, None, None, None,
                              _hip.ptr(dw), _hip.ptr(db), _hip.ptr(sc), _hip.stream()))
torch.cuda.synchronize()
e = (dw.double().cpu() - ref).abs() / ref.abs().max()
print("max err", e.max().item(), "db err", ((db.double().cpu() - dz.sum((0, 2, 3))).abs().max() / dz.sum((0, 2, 3)).abs().max()).item())
bad = (e > 1e-4)
print("bad fraction", bad.float().mean().item())
print("bad by tap", bad.float().mean((0, 1)))
print("bad by cj block of 16", bad.float().mean((1, 2, 3)).view(-1, 16).mean(1))
print("bad by ci block of 16", bad.float().mean((0, 2, 3)).view(-1, 16).mean(1))
