import sys, ctypes as C
sys.path.insert(0, 'dl-unet_amd'); sys.path.insert(0, '.')
import torch, _hip
L = _hip.lib()
B, H, Co = 8, 280, 64
for Ci in (32, 64, 128, 256, 512):
    x = torch.randn(B, H, H, Ci, device='cuda'); w = torch.randn(Ci, Co, 2, 2, device='cuda'); b = torch.randn(Co, device='cuda')
    y = torch.empty(B, 2*H, 2*H, Co, device='cuda')
    sc = torch.empty(L.unet_upconv2_scratch_bytes(B, H, H, max(Ci,64), Co), dtype=torch.uint8, device='cuda')
    for _ in range(2):
        _hip.check(L.unet_upconv2_fwd(_hip.ptr(x), B, H, H, Ci, _hip.ptr(w), _hip.ptr(b), Co, _hip.ptr(y), _hip.ptr(sc), _hip.stream()))
    torch.cuda.synchronize()
    L.unet_profile_reset(); L.unet_profile_enable(1)
    for _ in range(5):
        _hip.check(L.unet_upconv2_fwd(_hip.ptr(x), B, H, H, Ci, _hip.ptr(w), _hip.ptr(b), Co, _hip.ptr(y), _hip.ptr(sc), _hip.stream()))
    torch.cuda.synchronize(); L.unet_profile_enable(0)
    ms = C.c_double(); n = C.c_long(); fl = C.c_double()
    L.unet_profile_read(0, C.byref(ms), C.byref(n), C.byref(fl))
    nblk = ((B*H*H + 127)//128) * (4*Co//128)
    t = ms.value / n.value
    print("Ci=%4d nk=%3d blocks=%d: %.3f ms/launch  -> %.2f us per block-slot (512 slots)  %.1f TF" % (Ci, Ci//32, nblk, t, t*1e3/(nblk/512), fl.value/ms.value/1e9))
