"""GPU versions of the reference's data.py transforms that sit directly either side of the hot path
(SURVEY §8f).  Same names and meaning as the reference functions; tensors stay on the HIP device.

  mirror_transform(image)            data.py:249-277   overlap-tile border (N2), asymmetric reflection kept
  mirror_transform_tensor(image)     data.py:281-312
  test_input(images)                 data.py:184-188   mirror + (x-min)/ptp for a batch (ImageDataset_test)
  elastic_transform(images, a, s)    data.py:225-245   Simard-2003 elastic deformation (N1)
  reflect_rotate_crop(images, deg)   data.py:106-125   reflect pad + cubic-spline rotation + centre crop, fused (N1)
  augment(image, target, ...)        data.py:97-135    ImageDataset.__getitem__ after the file reads, on the device

Datasets, file readers, downloaders and the OpenCV label preprocessing are out of scope.
"""
import ctypes as _C

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
    # (host arrays are narrowed to fp32 by numpy: a torch CPU op on half a million elements goes through the intra-op thread
    #  pool, which on a GPU host's share of cores costs tens of milliseconds per call - measured 54 ms for one .float())
    if fields is not None:
        f0, f1 = [(f.to(dev, torch.float32) if torch.is_tensor(f) else torch.from_numpy(np.ascontiguousarray(f, dtype=np.float32)).to(dev))
                  .reshape(B, H, W).contiguous() for f in fields]
    elif random_state is not None:
        draws = [(random_state.rand(H, W), random_state.rand(H, W)) for _ in range(B)]    # per sample: dx field, then dy field
        f0 = torch.from_numpy(np.stack([d[0] for d in draws]).astype(np.float32)).to(dev)
        f1 = torch.from_numpy(np.stack([d[1] for d in draws]).astype(np.float32)).to(dev)
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


def reflect_rotate_crop(images, angles_deg, input_size=None, levels=255):
    """images: device tensor [n,n] or [B,n,n] (the random crop of the sample, e.g. 388^2); angles_deg: one angle or B angles
    (the reference draws one of 0,30,...,330 per sample, data.py:113).  Returns [S,S] / [B,S,S] with S = input_size: the centre
    of scipy.ndimage.rotate(np.pad(image, S, 'reflect'), deg) (data.py:106-125), fused so the 1532^2 padded / rotated
    images never exist.  levels=255 / 65535 reproduces scipy's rounding and clamping for the uint8 / uint16 images the reference loads;
    levels=0 keeps the float spline value."""
    single = images.dim() == 2
    x = _as_batch(images).contiguous().float()
    B, n, _ = x.shape
    if input_size is None:
        _, input_size, _ = input_size_compute(x)
    S = int(input_size)
    ang = np.atleast_1d(np.asarray(angles_deg, dtype=np.float32))
    if ang.size == 1 and B > 1:
        ang = np.repeat(ang, B)
    if ang.size != B:
        raise ValueError("need one angle per image")
    outs = []
    for lo in range(0, B, 64):                                    # the kernel takes up to 64 samples per call
        xb = x[lo:lo + 64]
        b = xb.shape[0]
        out = torch.empty(b, S, S, dtype=torch.float32, device=x.device)
        sc = torch.empty(_hip.lib().unet_rotate_scratch_bytes(b, S), dtype=torch.uint8, device=x.device)
        arr = (_C.c_float * b)(*[float(a) for a in ang[lo:lo + b]])
        _hip.run("unet_reflect_rotate_crop", x.device, _hip.ptr(xb), b, n, S, S, arr, int(levels), _hip.ptr(out), _hip.ptr(sc))
        outs.append(out)
    out = torch.cat(outs) if len(outs) > 1 else outs[0]
    return out[0] if single else out


def augment(image, target, crop_xy, crop, rot_deg, alpha, sigma, random_state=None, fields=None, levels=255):
    """What ImageDataset.__getitem__ does to one sample after reading it (data.py:97-135), on the device.  The random draws
    stay with the caller (crop origin from the weighted distribution + jitter, rot_deg from np.arange(0,360,30), the elastic
    fields), so the host RNG sequence can follow the reference's:  crop -> reflect pad + rotate + centre crop -> the same
    elastic deformation for image and mask -> mask cropped to the label extent and thresholded at 127 -> image to [0,1].
    image / target: device tensors [H,W] (grey levels / {0,255}); returns (inp [1,S,S] float32, gt [1,crop,crop] int64).
    A whole batch in one call (the DataLoader's collate, vectorised): image / target [B,H,W], crop_xy a list of B origins,
    rot_deg B angles; returns (inp [B,1,S,S], gt [B,1,crop,crop]) - every kernel then runs once for the batch."""
    batched = image.dim() == 3
    imgs = image if batched else image[None]
    tgts = target if batched else target[None]
    B = imgs.shape[0]
    origins = list(crop_xy) if batched else [crop_xy]
    angles = [float(a) for a in (rot_deg if batched else [rot_deg])]
    if len(origins) != B or len(angles) != B:
        raise ValueError("need one crop origin and one angle per sample")
    img = torch.stack([imgs[b, x0:x0 + crop, y0:y0 + crop] for b, (x0, y0) in enumerate(origins)]).float()
    tgt = torch.stack([tgts[b, x0:x0 + crop, y0:y0 + crop] for b, (x0, y0) in enumerate(origins)]).float()
    _, S, _ = input_size_compute(img)
    both = reflect_rotate_crop(torch.cat((img, tgt)), angles + angles, S, levels=levels)            # [2B,S,S]: images, then masks
    inp, gt = elastic_transform((both[:B], both[B:]), alpha, sigma, random_state=random_state, fields=fields)
    if levels:
        # the reference warps the uint8 / uint16 arrays it loaded: scipy's map_coordinates writes its result in the input's
        # type, i.e. rounds t + 0.5 down and clamps to the type's range (data.py:245 on the rotated integer images)
        inp = torch.floor(inp + 0.5).clamp_(0, levels)
        gt = torch.floor(gt + 0.5).clamp_(0, levels)
    pad = int((S - crop) / 2)
    gt = (gt[:, pad:crop + pad, pad:crop + pad] > 127).long()
    lo, hi = inp.amin(dim=(1, 2), keepdim=True), inp.amax(dim=(1, 2), keepdim=True)
    inp = (inp - lo) / (hi - lo)
    if batched:
        return inp[:, None], gt[:, None]
    return inp[0][None], gt[0][None]
