"""Drop-in for the reference's tester.py: `from tester import testing`, same signature (tester.py:13),
output tree (images/ labels/ preds/ + test_iou.out, test_pe.out) and metrics; forward and argmax run
on the HIP path (unet_forward, unet_argmax2).  torchvision is absent here, so the three TIFFs are
written with PIL using torchvision.utils.save_image's conversion (clamp to [0,1], x255, round, RGB)."""
import os
from time import time

import numpy as np
import torch

from functions import evaluation_metrics, metrics_from_counts
import optim as hip_optim


def maybe_mkdir_p(path):
    os.makedirs(path, exist_ok=True)


def save_image(tensor, path):
    from PIL import Image
    t = tensor.detach().float().cpu()
    if t.dim() == 2:
        t = t.unsqueeze(0)
    if t.shape[0] == 1:
        t = t.expand(3, -1, -1)
    arr = t.mul(255).add_(0.5).clamp_(0, 255).permute(1, 2, 0).to(torch.uint8).numpy()
    Image.fromarray(arr).save(path)


def testing(unet, test_loader, batch_size, device, output_dir):
    t0 = time()
    for sub in ('images', 'preds', 'labels'):
        maybe_mkdir_p(os.path.join(output_dir, sub))
    first = None                      # Q5: the reference keeps only the first sample's metrics

    for n, (image, label) in enumerate(test_loader):
        with torch.no_grad():      # the reference leaves autograd on here (Q8); only memory differs
            logits = unet(image.to(device))
        side = label.shape[-1]
        off = int((logits.shape[-1] - side) / 2)
        # crop + argmax + IoU / pixel-error counts in one pass on the device (no per-image mask download)
        mask, stats = hip_optim.crop_argmax_metrics(logits, label)
        for sub, stem, t in (('images', 'image', image[0, 0, off:side + off, off:side + off]),
                             ('labels', 'label', label[0, 0].float()),
                             ('preds', 'pred', mask[0].float())):
            save_image(t, os.path.join(output_dir, sub, '%s%d.tif' % (stem, n)))
        if first is None:
            inter, union, diff = (int(v) for v in stats[0].tolist())
            first = metrics_from_counts(inter, union, diff, side * label.shape[-2])

    mean, std = np.mean(first, axis=1), np.std(first, axis=1)
    for k, fname in enumerate(('test_iou.out', 'test_pe.out')):
        np.savetxt(os.path.join(output_dir, fname), [mean[k], std[k]])

    for label_, value in (('Mean IoU testing:', mean[0]), ('Mean PE testing :', mean[1])):
        print(label_, "{:.6f}".format(value))
    print('Testing took    :', "{:.6f}".format(time() - t0), 's')
    print(' ')
    print('Testing is finished')
