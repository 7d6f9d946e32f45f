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
    start = time()
    test_eval = None
    idx = 0

    maybe_mkdir_p(os.path.join(output_dir, 'images'))
    maybe_mkdir_p(os.path.join(output_dir, 'preds'))
    maybe_mkdir_p(os.path.join(output_dir, 'labels'))

    for image, label in test_loader:
        with torch.no_grad():      # the reference leaves autograd on here (Q8); only memory differs
            pred = unet(image.to(device))
        pad = int((pred.shape[-1] - label.shape[-1]) / 2)
        # crop + argmax + IoU / pixel-error counts in one pass on the device (no per-image mask download)
        pred, stats = hip_optim.crop_argmax_metrics(pred, label)

        save_image(image[0, 0, pad:label.shape[-1] + pad, pad:label.shape[-1] + pad], os.path.join(output_dir, 'images', f'image{idx}.tif'))
        save_image(label[0, 0, :, :].float(), os.path.join(output_dir, 'labels', f'label{idx}.tif'))
        save_image(pred[0, :, :].float(), os.path.join(output_dir, 'preds', f'pred{idx}.tif'))
        idx += 1

        if test_eval is None:      # Q5: the reference keeps only the first sample's metrics
            inter, union, diff = [int(v) for v in stats[0].tolist()]
            test_eval = metrics_from_counts(inter, union, diff, label.shape[-1] * label.shape[-2])

    test = np.mean(test_eval, axis=1)
    test_std = np.std(test_eval, axis=1)
    np.savetxt(os.path.join(output_dir, 'test_iou.out'), [test[0], test_std[0]])
    np.savetxt(os.path.join(output_dir, 'test_pe.out'), [test[1], test_std[1]])

    print('Mean IoU testing:', "{:.6f}".format(test[0]))
    print('Mean PE testing :', "{:.6f}".format(test[1]))
    print('Testing took    :', "{:.6f}".format(time() - start), 's')
    print(' ')
    print('Testing is finished')
