"""TEST INFRASTRUCTURE — numpy/scipy restatement of the reference's helpers either side of the hot
path.  Pinned by tests/golden/aux_golden.npz (tests/golden/make_golden_aux.py runs
the reference's own functions here)."""
import numpy as np
from scipy.ndimage import gaussian_filter, map_coordinates, rotate


def input_size_compute(n):
    """functions.py:121-146."""
    L = 20
    while 16 * L - 124 < n:
        L += 2
    return n, 16 * L + 60, 16 * L - 124


def mirror_index(S, n):
    """Source row/column of every output row/column of mirror_transform (data.py:249-277): top/left band
    reflects WITHOUT the edge pixel (P - Y), bottom/right band WITH it (n-1-(Y-n-P))."""
    P = (S - n) // 2
    if (S - n) % 2 or P > n - 1:
        raise ValueError("cannot mirror %d into %d" % (n, S))
    Y = np.arange(S)
    return np.where(Y < P, P - Y, np.where(Y < P + n, Y - P, n - 1 - (Y - n - P)))


def mirror_transform(image):
    n = image.shape[-1]
    _, S, _ = input_size_compute(n)
    idx = mirror_index(S, n)
    return image.reshape(n, n)[np.ix_(idx, idx)]


def normalise01(x):
    """(x - min) / ptp per image (data.py:134,188)."""
    return (x - x.min()) / np.ptp(x)


def class_balance(gt):
    """functions.py:82-117 for one [H,W] label image; raises IndexError like the reference when a class is missing."""
    uval, counts = np.unique(gt, return_counts=True)
    w = np.ones(gt.shape, dtype=np.float32)
    for pos in range(len(uval)):
        w[gt == uval[pos]] = np.float32(counts[1]) / np.float32(counts[pos])
    return w


def gaussian_taps(sigma, truncate=4.0):
    """The normalised 1-D kernel scipy.ndimage.gaussian_filter uses (radius = int(truncate*sigma + 0.5))."""
    radius = int(truncate * float(sigma) + 0.5)
    k = np.arange(-radius, radius + 1, dtype=np.float64)
    w = np.exp(-0.5 / (sigma * sigma) * k * k)
    return (w / w.sum()), radius


def elastic_transform(images, alpha, sigma, fields):
    """data.py:225-245 with the two uniform[0,1) fields given explicitly (the reference draws them from
    a numpy RandomState; parity is defined for equal fields)."""
    shape = images[0].shape
    dx = gaussian_filter(fields[0] * 2 - 1, sigma, mode="constant", cval=0) * alpha
    dy = gaussian_filter(fields[1] * 2 - 1, sigma, mode="constant", cval=0) * alpha
    x, y = np.meshgrid(np.arange(shape[0]), np.arange(shape[1]), indexing="ij")
    idx = np.reshape(x + dx, (-1, 1)), np.reshape(y + dy, (-1, 1))
    return [map_coordinates(im, idx, order=1).reshape(shape) for im in images], dx, dy


def eval_counts(pred, label):
    """IoU / Pixel_error numerators and denominators (functions.py:174-213)."""
    inter = np.logical_and(pred, label).sum()
    union = np.logical_or(pred, label).sum()
    diff = np.abs(pred.astype(np.int64) - label.astype(np.int64)).sum()
    return int(inter), int(union), int(diff)


def reflect_rotate_crop(image, rot_deg, input_size=None):
    """data.py:106-125 for one image (any dtype the reference loads: uint8 / uint16 / float): reflect-pad by input_size on
    every side, scipy.ndimage.rotate (cubic spline, reshape=True, mode='constant', same dtype out), centre crop of
    input_size x input_size."""
    if input_size is None:
        _, input_size, _ = input_size_compute(image.shape[-1])
    pad = np.pad(image, pad_width=input_size, mode="reflect")
    rot = rotate(pad, rot_deg)
    h, w = rot.shape
    l = w // 2 - input_size // 2; r = w // 2 + input_size // 2
    t = h // 2 - input_size // 2; b = h // 2 + input_size // 2
    return rot[t:b, l:r]


def cells(seed, n=512):
    """The synthetic 'cell' sample of tests/golden/make_golden_augment.py (BASELINE configs[3] shape): 12 discs on a noisy
    background, from the libm-free PRNG so that the GPU box regenerates it: (uint8 image [n,n], {0,255} uint8 mask [n,n])."""
    from . import prng
    u = prng.uniform01(21, seed, 36)
    yy, xx = np.mgrid[0:n, 0:n]
    mask = np.zeros((n, n), bool)
    for i in range(12):
        cy, cx, r = 40 + u[3 * i] * (n - 80), 40 + u[3 * i + 1] * (n - 80), 15 + u[3 * i + 2] * 30
        mask |= (yy - cy) ** 2 + (xx - cx) ** 2 < r * r
    noise = prng.uniform01(22, seed, n * n).reshape(n, n)
    img = np.floor((0.3 + 0.5 * mask + 0.1 * noise) * 255).astype(np.uint8)
    return img, (mask * 255).astype(np.uint8)


def augment(image, target, rot_deg, alpha, sigma, random_state):
    """ImageDataset.__getitem__ after the crop (data.py:103-134) for one sample: reflect pad + rotation + centre crop, the same
    elastic deformation for image and mask (fields drawn from random_state: dx field, then dy field), mask cropped to the label
    extent (returned BEFORE the 127 threshold, as grey levels) and the image normalised to [0,1]."""
    n = image.shape[-1]
    _, S, _ = input_size_compute(n)
    ri, rt = reflect_rotate_crop(image, rot_deg), reflect_rotate_crop(target, rot_deg)
    f0 = random_state.rand(S, S); f1 = random_state.rand(S, S)
    (ei, et), _, _ = elastic_transform((ri, rt), alpha, sigma, (f0, f1))
    pad = int((S - n) / 2)
    return normalise01(ei), et[pad:n + pad, pad:n + pad]
