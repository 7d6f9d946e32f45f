"""Drop-in for the reference's functions.py helpers that sit either side of the hot path.

input_size_compute / evaluation_metrics / IoU / Pixel_error / class_balance keep the reference's
names, arguments and results (functions.py:82-213).  They are host-side bookkeeping on labels and
388^2 masks (SURVEY §2 rows 5,7,8: out of scope as kernels); weighted_map needs OpenCV, which this
image does not have, and is dead code at run time in the reference (quirk Q3).
"""
import numpy as np
import torch


def weighted_map(gt_batch):
    raise NotImplementedError("weighted_map needs OpenCV (cv2.connectedComponents / distanceTransform), absent in this "
                              "image; the reference never reaches it at run time (trainer.py:68 `is` comparison, SURVEY Q3)")


def class_balance(gt_batch):
    """Per-image class-frequency weight map (functions.py:82-117): every pixel of value v gets
    count(second unique value) / count(v).  gt_batch: [B,H,W] -> float [B,H,W].
    Host tensors take the reference's CPU algorithm; {0,1} int64 labels already on the HIP device
    use unet_class_balance (two counting reductions and a select, no host round trip)."""
    if gt_batch.is_cuda:
        import _hip
        gt = gt_batch.contiguous()
        if gt.dtype != torch.int64:
            gt = gt.long()
        B, H, W = gt.shape
        w = torch.empty(B, H, W, dtype=torch.float32, device=gt.device)
        counts = torch.empty(B, dtype=torch.int64, device=gt.device)
        _hip.run("unet_class_balance", gt.device, _hip.ptr(gt), B, H, W, _hip.ptr(w), _hip.ptr(counts))
        if bool(((counts == 0) | (counts == H * W)).any()):      # the reference indexes counts[1]: a one-class image raises
            raise IndexError("index 1 is out of bounds for dimension 0 with size 1")
        return w
    gt_batch = gt_batch.cpu()
    w_batch = torch.empty_like(gt_batch).float()
    for b in range(gt_batch.shape[0]):
        gt = gt_batch[b]
        uval, counts = torch.unique(gt, return_counts=True)
        w_c = torch.ones(gt.shape)
        for pos in range(len(uval)):
            w_c[gt == uval[pos]] = counts[1].float() / counts[pos].float()
        w_batch[b] = w_c
    return w_batch


def input_size_compute(image):
    """(original, input, output) sizes for the overlap-tile strategy (functions.py:121-146):
    smallest even L >= 20 with 16L-124 >= original; input = 16L+60; output = 16L-124."""
    original_size = image.shape[-1]
    lowest_res = 20
    while 16 * lowest_res - 124 < original_size:
        lowest_res += 2
    return original_size, 16 * lowest_res + 60, 16 * lowest_res - 124


def Pixel_error(pred, label):
    pred_np = pred.cpu().numpy()
    label_np = label.cpu().numpy()
    return np.sum(abs(pred_np - label_np)) / pred_np.size


def IoU(pred, label):
    pred_np = pred.cpu().numpy()
    label_np = label.cpu().numpy()
    return np.sum(np.logical_and(pred_np, label_np)) / np.sum(np.logical_or(pred_np, label_np))


def metrics_from_counts(inter, union, diff, size):
    """IoU and pixel error from the integer counts unet_eval_masks produces on the device."""
    out = np.empty([2, 1])
    out[0] = inter / union
    out[1] = diff / size
    return out


def evaluation_metrics(pred, label):
    """[[IoU],[pixel error]] as a (2,1) array (functions.py:150-170)."""
    out = np.empty([2, 1])
    out[0] = IoU(pred, label)
    out[1] = Pixel_error(pred, label)
    return out
