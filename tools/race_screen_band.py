"""Race screen for the band kernel's sync structure (counted vmcnt + raw barriers, igemmb.hip igemmb3_kernel): every shape is
run ITER times in one process and each result compared bit for bit with the first (the kernel is deterministic by construction;
a DMA that lands after its reader shows as a rare mismatch).  usage (GPU box): python tools/race_screen_band.py [ITER]"""
import hashlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dl-unet_amd"))
import torch  # noqa: E402
import _hip  # noqa: E402

ITER = int(sys.argv[1]) if len(sys.argv) > 1 else 200
L = _hip.lib()
_hip.check(L.unet_set_math(2), "set_math")
torch.cuda.set_device(0)
g = torch.Generator().manual_seed(11)


def rnd(*s, scale=1.0):
    return torch.randn(*s, generator=g) * scale


bad = 0
# (B, H, C, K): the layers of the B = 8 step that take the band kernel, shallow to deep, plus a narrow ragged one
for B, H, C, K in [(8, 282, 128, 128), (8, 138, 256, 256), (8, 66, 512, 512), (8, 30, 1024, 1024), (3, 21, 128, 128)]:
    x = rnd(B, H, H, C).to(torch.bfloat16).cuda()
    w = rnd(K, C, 3, 3, scale=0.05).to(torch.bfloat16).float().cuda()
    b = rnd(K).cuda()
    dz = rnd(B, H - 2, H - 2, K).to(torch.bfloat16).cuda()
    mask = rnd(B, H, H, C).clamp_min(0).to(torch.bfloat16).cuda()
    y = torch.empty(B, H - 2, H - 2, K, device="cuda", dtype=torch.bfloat16)
    dx = torch.empty(B, H, H, C, device="cuda", dtype=torch.bfloat16)
    dw = torch.empty(K, C, 3, 3, device="cuda"); db = torch.empty(K, device="cuda")
    sc = torch.empty(max(int(L.unet_conv3x3_scratch_bytes(C, K)), 256), dtype=torch.uint8, device="cuda")
    sc2 = torch.empty(max(int(L.unet_conv3x3_bwd_scratch_bytes(B, H, H, C, K)), 256), dtype=torch.uint8, device="cuda")
    first = None
    for it in range(ITER):
        y.zero_(); dx.zero_()
        _hip.check(L.unet_conv3x3_fwd(_hip.ptr(x), H, H, C, 0, None, 0, B, H, H, _hip.ptr(w), _hip.ptr(b), K, 1, _hip.ptr(y), _hip.ptr(sc), _hip.stream()), "fwd")
        _hip.check(L.unet_conv3x3_bwd(_hip.ptr(x), H, H, C, 0, None, 0, B, H, H, _hip.ptr(w), K, _hip.ptr(dz), _hip.ptr(dx), _hip.ptr(mask), None,
                                      None, None, _hip.ptr(dw), _hip.ptr(db), _hip.ptr(sc2), _hip.stream()), "bwd")
        h = (int(y.view(torch.int16).to(torch.int64).sum().item()), int(dx.view(torch.int16).to(torch.int64).sum().item()),
             hashlib.sha256(y.view(torch.int16)[0, :8].cpu().numpy().tobytes()).hexdigest()[:8])
        if first is None:
            first = h
        elif h != first:
            bad += 1
            print("MISMATCH", (B, H, C, K), it, h, first, flush=True)
    print("shape", (B, H, C, K), "iterations", ITER, "checksum", first, flush=True)
print("race screen:", "FAILED (%d mismatches)" % bad if bad else "clean")
sys.exit(1 if bad else 0)
