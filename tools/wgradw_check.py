"""Weight gradient: direct (mode 0) vs Winograd (mode 3) on layer shapes — error between them and time per call."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "dl-unet_amd"))
import torch
import _hip
L = _hip.lib()

def run(B, H, C, K, mode, reps=5):
    torch.manual_seed(1)
    x = torch.randn(B, H, H, C, device="cuda"); w = torch.randn(K, C, 3, 3, device="cuda") * 0.05
    dz = torch.randn(B, H - 2, H - 2, K, device="cuda")
    dw = torch.empty(K, C, 3, 3, device="cuda"); db = torch.empty(K, device="cuda")
    sc = torch.empty(L.unet_conv3x3_bwd_scratch_bytes(B, H, H, C, K), dtype=torch.uint8, device="cuda")
    _hip.check(L.unet_set_math(mode))
    def call():
        _hip.check(L.unet_conv3x3_bwd(_hip.ptr(x), H, H, C, 0, None, 0, B, H, H, _hip.ptr(w), K, _hip.ptr(dz), None, None, None, None, None,
                                      _hip.ptr(dw), _hip.ptr(db), _hip.ptr(sc), _hip.stream()))
    call(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): call()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    return dw.clone(), db.clone(), ms, 2.0 * B * (H - 2) ** 2 * C * K * 9 / ms / 1e9

for (B, H, C, K) in [(1, 62, 64, 64), (8, 570, 64, 64), (8, 282, 128, 128), (8, 138, 256, 256), (8, 66, 512, 512), (8, 30, 1024, 1024), (8, 392, 128, 64)]:
    dw0, db0, ms0, tf0 = run(B, H, C, K, 0)
    dw3, db3, ms3, tf3 = run(B, H, C, K, 3)
    e = ((dw3 - dw0).abs().max() / dw0.abs().max()).item(); eb = ((db3 - db0).abs().max() / db0.abs().max()).item()
    print("B=%d H=%d C=%d K=%d: direct %.3f ms (%.1f TF)  winograd %.3f ms (%.1f TF-eq)  dw err %.2e  db err %.2e" % (B, H, C, K, ms0, tf0, ms3, tf3, e, eb), flush=True)
_hip.check(L.unet_set_math(3))
