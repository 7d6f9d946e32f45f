import sys, os
sys.path.insert(0, "dl-unet_amd")
import torch, _hip
L = _hip.lib()
_hip.check(L.unet_set_math(2))
B, H, C, K = 8, 570, 64, 64
x = torch.randn(B, H, H, C, device="cuda").to(torch.bfloat16); w = torch.randn(K, C, 3, 3, device="cuda") * 0.05; b = torch.randn(K, device="cuda")
y = torch.empty(B, H - 2, H - 2, K, device="cuda", dtype=torch.bfloat16)
sc = torch.empty(L.unet_conv3x3_scratch_bytes(C, K), dtype=torch.uint8, device="cuda")
for _ in range(3):
    _hip.check(L.unet_conv3x3_fwd(_hip.ptr(x), H, H, C, 0, None, 0, B, H, H, _hip.ptr(w), _hip.ptr(b), K, 1, _hip.ptr(y), _hip.ptr(sc), _hip.stream()))
torch.cuda.synchronize()
