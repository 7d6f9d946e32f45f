"""GPU versions of the reference's data.py transforms that sit directly either side of the hot path
(SURVEY §8f).  Same names and meaning as the reference functions; tensors stay on the HIP device.

  mirror_transform(image)            data.py:249-277   overlap-tile border (N2), asymmetric reflection kept
  mirror_transform_tensor(image)     data.py:281-312
  test_input(images)                 data.py:184-188   mirror + (x-min)/ptp for a batch (ImageDataset_test)
  elastic_transform(images, a, s)    data.py:225-245   Simard-2003 elastic deformation (N1)

Datasets, file readers, downloaders and the OpenCV label preprocessing are out of scope.
"""
import numpy as np
import torch

import _hip
from functions import input_size_compute


def _as_batch(image):
    if image.dim() == 2:
        return image[None]
    return image.reshape(-1, image.shape[-2], image.shape[-1])


def mirror_transform(image, normalise=False):
    """image: device tensor [n,n] (or [B,n,n] / [B,1,n,n]); returns [input_size,input_size] (or
    [B,1,S,S]) mirrored outwards like the reference (reflection without the edge pixel on the top/left
    band, with it on the bottom/right band)."""
    single = image.dim() == 2
    x = _as_batch(image).contiguous().float()
    if x.shape[-1] != x.shape[-2]:
        raise ValueError("mirror_transform expects square images")
    B, n, _ = x.shape
    _, S, _ = input_size_compute(x)
    mm = None
    if normalise:
        mm = torch.empty(B, 2, dtype=torch.float32, device=x.device)
        _hip.run("unet_minmax", x.device, _hip.ptr(x), B, n * n, _hip.ptr(mm))
    out = torch.empty(B, 1, S, S, dtype=torch.float32, device=x.device)
    _hip.run("unet_mirror_pad", x.device, _hip.ptr(x), B, n, S, _hip.ptr(mm), _hip.ptr(out))
    return out[0, 0] if single else out


def mirror_transform_tensor(image):
    """[(1),(1),n,n] -> [1,1,S,S] (data.py:281-312)."""
    n = image.shape[-1]
    return mirror_transform(image.reshape(1, n, n))


def test_input(images):
    """What ImageDataset_test.__getitem__ produces for the network (data.py:184-188), batched on the
    device: mirror to the input size, then (x - min) / ptp per image."""
    return mirror_transform(images if images.dim() >= 3 else images[None], normalise=True)


def gaussian_taps(sigma, truncate=4.0):
    radius = int(truncate * float(sigma) + 0.5)
    k = np.arange(-radius, radius + 1, dtype=np.float64)
    w = np.exp(-0.5 / (sigma * sigma) * k * k)
    return (w / w.sum()).astype(np.float32), radius


def elastic_transform(images, alpha, sigma, random_state=None, fields=None):
    """images: tuple of device tensors of equal shape [H,W] or [B,H,W]; every image of a sample is warped
    with the SAME displacement field (image and mask, data.py:128).  The two uniform[0,1) fields come
    from `fields`, from a numpy RandomState (drawn on the host exactly like the reference), or from
    torch.rand on the device.  Returns a list of warped tensors."""
    ref = _as_batch(images[0]).contiguous().float()
    B, H, W = ref.shape
    dev = ref.device
    if fields is not None:
        f0, f1 = [torch.as_tensor(f, dtype=torch.float32, device=dev).reshape(B, H, W).contiguous() for f in fields]
    elif random_state is not None:
        draws = [(random_state.rand(H, W), random_state.rand(H, W)) for _ in range(B)]    # per sample: dx field, then dy field
        f0 = torch.from_numpy(np.stack([d[0] for d in draws])).float().to(dev)
        f1 = torch.from_numpy(np.stack([d[1] for d in draws])).float().to(dev)
    else:
        f0 = torch.rand(B, H, W, device=dev); f1 = torch.rand(B, H, W, device=dev)
    w, radius = gaussian_taps(sigma)
    wd = torch.from_numpy(w).to(dev)
    tmp = torch.empty_like(f0)
    dx = torch.empty_like(f0); dy = torch.empty_like(f0)
    # dx displaces rows (axis 0), dy columns (axis 1) — the reference's naming (data.py:238-243)
    _hip.run("unet_gaussian_filter", dev, _hip.ptr((f0 * 2 - 1).contiguous()), B, H, W, _hip.ptr(wd), radius, float(alpha), _hip.ptr(tmp), _hip.ptr(dx))
    _hip.run("unet_gaussian_filter", dev, _hip.ptr((f1 * 2 - 1).contiguous()), B, H, W, _hip.ptr(wd), radius, float(alpha), _hip.ptr(tmp), _hip.ptr(dy))
    outs = []
    for im in images:
        x = _as_batch(im).contiguous().float()
        o = torch.empty_like(x)
        _hip.run("unet_warp_bilinear", dev, _hip.ptr(x), _hip.ptr(dx), _hip.ptr(dy), B, H, W, _hip.ptr(o))
        outs.append(o.reshape(im.shape))
    return outs
