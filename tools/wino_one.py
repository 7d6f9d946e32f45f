"""One Winograd (math mode 3) forward conv launch shape, a few repetitions — the target of tools/wino_pmc.sh."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "dl-unet_amd"))
import torch
import _hip
L = _hip.lib()
B, H, C, K = [int(v) for v in (sys.argv[1:5] if len(sys.argv) >= 5 else (8, 282, 128, 128))]
mode = int(sys.argv[5]) if len(sys.argv) > 5 else 3
torch.manual_seed(1)
x = torch.randn(B, H, H, C, device="cuda"); w = torch.randn(K, C, 3, 3, device="cuda") * 0.05; b = torch.randn(K, device="cuda")
y = torch.empty(B, H - 2, H - 2, K, device="cuda")
sc = torch.empty(L.unet_conv3x3_scratch_bytes(C, K), dtype=torch.uint8, device="cuda")
_hip.check(L.unet_set_math(mode))
for _ in range(3):
    _hip.check(L.unet_conv3x3_fwd(_hip.ptr(x), H, H, C, 0, None, 0, B, H, H, _hip.ptr(w), _hip.ptr(b), K, 1, _hip.ptr(y), _hip.ptr(sc), _hip.stream()))
torch.cuda.synchronize()
