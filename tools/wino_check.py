"""Winograd vs direct: error against fp64 and time per launch on one big layer (run on the GPU box)."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "dl-unet_amd"))
import torch, torch.nn.functional as F
import _hip
L = _hip.lib()
torch.manual_seed(0)

def run(B, H, C, K, mode, reps=5):
    torch.manual_seed(1)
    x = torch.randn(B, H, H, C, device="cuda"); w = (torch.randn(K, C, 3, 3, device="cuda") * 0.05); b = torch.randn(K, device="cuda")
    y = torch.empty(B, H - 2, H - 2, K, device="cuda")
    sc = torch.empty(L.unet_conv3x3_scratch_bytes(C, K), dtype=torch.uint8, device="cuda")
    _hip.check(L.unet_set_math(mode))
    def call():
        _hip.check(L.unet_conv3x3_fwd(_hip.ptr(x), H, H, C, 0, None, 0, B, H, H, _hip.ptr(w), _hip.ptr(b), K, 1, _hip.ptr(y), _hip.ptr(sc), _hip.stream()))
    call(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): call()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    fl = 2.0 * B * (H - 2) ** 2 * C * K * 9
    return y, ms, fl / ms / 1e9

for (B, H, C, K) in [(1, 62, 64, 64), (8, 570, 64, 64), (8, 282, 128, 128), (8, 138, 256, 256), (8, 66, 512, 512), (8, 30, 1024, 1024)]:
    y0, ms0, tf0 = run(B, H, C, K, 0)
    y3, ms3, tf3 = run(B, H, C, K, 3)
    err = ((y3 - y0).abs().max() / y0.abs().max()).item()
    line = "B=%d H=%d C=%d K=%d: direct %.3f ms (%.1f TF)  winograd %.3f ms (%.1f TF-eq)  |wino-direct|/|y| = %.2e" % (B, H, C, K, ms0, tf0, ms3, tf3, err)
    if B * H * H * C < 3e6:
        xr = None
    print(line, flush=True)
_hip.check(L.unet_set_math(0))
