"""Step-side HIP kernels behind torch-shaped interfaces.

  bce_with_logits  — nn.BCEWithLogitsLoss(weight=w)(preds, ll) + backward   (trainer.py:63-77)
  argmax2          — preds.argmax(dim=1) for the 2-class head               (trainer.py:82, tester.py:30)
  SGD              — optim.SGD(lr, momentum).step()                          (trainer.py:30,78)
"""
import ctypes as C

import torch

import _hip


class _BCEFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, target, weight, wstrides, grad_scale):
        B, two, H, W = logits.shape
        assert two == 2
        logits = logits.contiguous()
        target = target.contiguous()
        loss = torch.empty((), dtype=torch.float32, device=logits.device)
        dl = torch.empty_like(logits)
        sc = torch.empty(_hip.lib().unet_bce_scratch_bytes(logits.numel()), dtype=torch.uint8, device=logits.device)
        ws = wstrides if weight is not None else (0, 0, 0, 0)
        _hip.run("unet_bce_logits", logits.device, _hip.ptr(logits), _hip.ptr(target), _hip.ptr(weight), ws[0], ws[1], ws[2], ws[3],
                 B, H, W, _hip.ptr(loss), _hip.ptr(dl), float(grad_scale), _hip.ptr(sc))
        ctx.save_for_backward(dl)
        return loss

    @staticmethod
    def backward(ctx, g):
        (dl,) = ctx.saved_tensors
        return dl * g, None, None, None, None


def _weight_strides(weight, logits):
    """torch broadcasting of `weight` against [B,2,H,W] as four element strides (0 on broadcast axes); like the reference
    (trainer.py:72-75), a [B,H,W] map is right-aligned: its first axis meets the CLASS axis (quirk Q4), other sizes raise."""
    weight = weight.to(logits.device, torch.float32).contiguous()
    shape = (1,) * (4 - weight.dim()) + tuple(weight.shape)
    st = (0,) * (4 - weight.dim()) + tuple(weight.stride())
    wstrides = []
    for d in range(4):
        if shape[d] == logits.shape[d]:
            wstrides.append(st[d])
        elif shape[d] == 1:
            wstrides.append(0)
        else:
            raise RuntimeError("The size of tensor a (%d) must match the size of tensor b (%d) at non-singleton "
                               "dimension %d" % (logits.shape[d], shape[d], d))
    return weight, wstrides


class _BCEStepFn(torch.autograd.Function):
    """unet_bce_step: loss (+ its gradient) and the argmax mask from logits and integer labels in one device pass."""

    @staticmethod
    def forward(ctx, logits, labels, weight, wstrides, grad_scale, want_mask):
        B, two, H, W = logits.shape
        assert two == 2
        # (read before any copy: forward runs with grad mode off, so a contiguous() copy never requires grad)
        need_grad = ctx.needs_input_grad[0]
        if logits.stride(3) != 1:
            logits = logits.contiguous()
        labels = labels.to(logits.device).reshape(B, H, W).contiguous()
        if labels.dtype != torch.int64:
            labels = labels.long()
        loss = torch.empty((), dtype=torch.float32, device=logits.device)
        dl = torch.empty(B, 2, H, W, dtype=torch.float32, device=logits.device) if need_grad else None
        mask = torch.empty(B, H, W, dtype=torch.int64, device=logits.device) if want_mask else None
        sc = torch.empty(_hip.lib().unet_bce_step_scratch_bytes(B * H * W), dtype=torch.uint8, device=logits.device)
        ws = wstrides if weight is not None else (0, 0, 0, 0)
        _hip.run("unet_bce_step", logits.device, _hip.ptr(logits), logits.stride(0), logits.stride(1), logits.stride(2), _hip.ptr(labels),
                 _hip.ptr(weight), ws[0], ws[1], ws[2], ws[3], B, H, W, _hip.ptr(loss), _hip.ptr(dl), float(grad_scale), _hip.ptr(mask), _hip.ptr(sc))
        if need_grad:
            ctx.save_for_backward(dl)
        if mask is not None:
            ctx.mark_non_differentiable(mask)
            return loss, mask
        return loss, torch.empty(0, dtype=torch.int64, device=logits.device)

    @staticmethod
    def backward(ctx, g, _gmask):
        (dl,) = ctx.saved_tensors
        return dl * g, None, None, None, None, None


def bce_argmax_step(preds, labels, weight=None, grad_scale=1.0, want_mask=True):
    """Both halves of the step either side of the network in one kernel (trainer.py:60-82):
         loss = BCEWithLogitsLoss(weight)(preds, [1 - y, y])      mask = preds.argmax(dim=1)
    preds [B,2,H,W] (any view with unit last stride, e.g. the trainer's centre crop), labels int64 [B,1,H,W] or [B,H,W].
    Returns (loss, mask | None); loss.backward() feeds the gradient the same pass produced."""
    wstrides = None
    if weight is not None:
        weight, wstrides = _weight_strides(weight, preds)
    loss, mask = _BCEStepFn.apply(preds, labels, weight, wstrides, grad_scale, want_mask)
    return loss, (mask if want_mask else None)


def bce_with_logits(logits, target, weight=None, grad_scale=1.0):
    """mean(weight * bce(logits, target)).  `weight` follows torch broadcasting against
    [B,2,H,W]: like the reference (trainer.py:72-75), a [B,H,W] map is right-aligned, i.e. its
    first axis meets the CLASS axis and must be 1 or 2 (quirk Q4) — other sizes raise."""
    wstrides = None
    if weight is not None:
        weight, wstrides = _weight_strides(weight, logits)
    return _BCEFn.apply(logits, target, weight, wstrides, grad_scale)


def onehot2(labels, like):
    """ll[:,0] = 1 - y, ll[:,1] = y on the device (trainer.py:63-66)."""
    labels = labels.to(like.device).contiguous()
    B, _, H, W = like.shape
    out = torch.empty(B, 2, H, W, dtype=torch.float32, device=like.device)
    _hip.run("unet_onehot2", like.device, _hip.ptr(labels), _hip.ptr(out), B, H, W)
    return out


def argmax2(preds):
    """preds [B,2,H,W] (any view with unit last stride) -> int64 [B,H,W]; ties -> class 0."""
    B, two, H, W = preds.shape
    assert two == 2 and preds.stride(3) == 1
    out = torch.empty(B, H, W, dtype=torch.int64, device=preds.device)
    _hip.run("unet_argmax2", preds.device, _hip.ptr(preds), preds.stride(0), preds.stride(1), preds.stride(2), _hip.ptr(out), B, H, W)
    return out


class SGD(torch.optim.Optimizer):
    """optim.SGD(params, lr, momentum) with the update done by one multi-tensor HIP kernel per
    <=46 tensors.  No dampening / nesterov / weight decay (the reference uses none).  param_groups
    and state keep torch's shape so lr schedulers (ReduceLROnPlateau, trainer.py:31) work."""

    def __init__(self, params, lr=1e-4, momentum=0.99):
        super().__init__(params, dict(lr=lr, momentum=momentum))

    @torch.no_grad()
    def step(self, closure=None):
        for group in self.param_groups:
            ps = [p for p in group["params"] if p.grad is not None]
            first = [p for p in ps if "momentum_buffer" not in self.state[p]]
            rest = [p for p in ps if "momentum_buffer" in self.state[p]]
            for batch, is_first in ((first, 1), (rest, 0)):
                for i in range(0, len(batch), _hip.N_PARAMS):
                    chunk = batch[i:i + _hip.N_PARAMS]
                    if is_first:
                        for p in chunk:
                            self.state[p]["momentum_buffer"] = torch.empty_like(p)
                    bufs = [self.state[p]["momentum_buffer"] for p in chunk]
                    grads = [p.grad.contiguous() for p in chunk]
                    numel = (C.c_size_t * len(chunk))(*[p.numel() for p in chunk])
                    _hip.run("unet_sgd_momentum", chunk[0].device, _hip.ptr_table(chunk), _hip.ptr_table(grads), _hip.ptr_table(bufs),
                             numel, len(chunk), float(group["lr"]), float(group["momentum"]), is_first)


def crop_argmax_metrics(preds, labels=None):
    """Fused back end of the test/validation step (tester.py:29-42): centre-crop `preds` [B,2,So,So] to the
    label size, argmax over the two classes, and (with int64 labels [B,1,n,n] or [B,n,n]) count
    sum(pred&label), sum(pred|label), sum|pred-label| per image.  Returns (mask int64 [B,n,n], stats int64 [B,3] | None)."""
    B, two, So, _ = preds.shape
    assert two == 2 and preds.stride(3) == 1
    if labels is not None:
        labels = labels.to(preds.device).reshape(B, labels.shape[-2], labels.shape[-1]).contiguous()
        n = labels.shape[-1]
    else:
        n = So
    pad = int((So - n) / 2)
    mask = torch.empty(B, n, n, dtype=torch.int64, device=preds.device)
    stats = torch.empty(B, 3, dtype=torch.int64, device=preds.device) if labels is not None else None
    _hip.run("unet_eval_masks", preds.device, _hip.ptr(preds), preds.stride(0), preds.stride(1), preds.stride(2), pad, _hip.ptr(labels),
             _hip.ptr(mask), B, n, _hip.ptr(stats))
    return mask, stats
