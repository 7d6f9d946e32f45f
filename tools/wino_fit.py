"""Per-workgroup fixed cost and per-step cost of the Winograd kernel: same M and N, growing channel count."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "dl-unet_amd"))
import torch
import _hip
L = _hip.lib()

def run(B, H, C, K, mode, reps=10):
    torch.manual_seed(1)
    x = torch.randn(B, H, H, C, device="cuda"); w = torch.randn(K, C, 3, 3, device="cuda") * 0.05; b = torch.randn(K, device="cuda")
    y = torch.empty(B, H - 2, H - 2, K, device="cuda")
    sc = torch.empty(L.unet_conv3x3_scratch_bytes(C, K), dtype=torch.uint8, device="cuda")
    _hip.check(L.unet_set_math(mode))
    L.unet_profile_enable(1)
    def call():
        _hip.check(L.unet_conv3x3_fwd(_hip.ptr(x), H, H, C, 0, None, 0, B, H, H, _hip.ptr(w), _hip.ptr(b), K, 1, _hip.ptr(y), _hip.ptr(sc), _hip.stream()))
    call(); torch.cuda.synchronize()
    L.unet_profile_reset()
    for _ in range(reps): call()
    torch.cuda.synchronize()
    import ctypes as C_
    ms = C_.c_double(); n = C_.c_long(); fl = C_.c_double()
    L.unet_profile_read(3, C_.byref(ms), C_.byref(n), C_.byref(fl))
    L.unet_profile_enable(0)
    return ms.value / max(n.value, 1)

B, H, K = 8, 282, 128
tiles = B * ((H - 2) // 2) ** 2
wgs = (tiles + 63) // 64 * (K // 64)
rounds = wgs / 256.0
prev = None
for C in (64, 128, 256, 512):
    ms = run(B, H, C, K, 3)
    per_wg = ms * 1e3 / rounds
    line = "C=%d: %.3f ms, %.1f rounds, %.2f us/WG, %d steps" % (C, ms, rounds, per_wg, C // 8)
    if prev:
        line += "  -> %.3f us/step, fixed %.2f us" % ((per_wg - prev[1]) / (C // 8 - prev[0]), per_wg - (per_wg - prev[1]) / (C // 8 - prev[0]) * (C // 8))
    prev = (C // 8, per_wg)
    print(line, flush=True)
_hip.check(L.unet_set_math(0))
