"""TEST INFRASTRUCTURE — drives the HIP path through the C ABI and checks it against the fp64 C oracle
evaluated on the SAME piecewise-linear branch (same ReLU masks and max-pool winners).

Why "same branch": a ReLU whose pre-activation is within fp32 noise of zero, or a pool window with a
near-tie, lands on a different linear piece in two correct fp32 evaluations (and in fp64); the
gradient of that element then differs by O(1) and, through the small deep feature maps, moves whole
weight-gradient tensors by ~1e-2 (measured: at S=220 a single flipped element of conv42e's output
changes every upstream gradient by 0.5-1%; the plain-C fp32 oracle shows the same 4e-3..2e-2 at
S=572).  The reference's own fp32 CPU run has the same property.  On the same branch the
function is linear in every ReLU/pool and parity is tight.
"""
import ctypes as C

import numpy as np
import torch

from . import oracle_c, prng

RELU_BUFS = ["a1_0", "a2_0", "a1_1", "a2_1", "a1_2", "a2_2", "a1_3", "a2_3", "a1_4", "a2_4",
             "d1_0", "d2_0", "d1_1", "d2_1", "d1_2", "d2_2", "d1_3", "d2_3"]


def nerr(a, ref):
    a = np.asarray(a, dtype=np.float64); ref = np.asarray(ref, dtype=np.float64)
    return float(np.abs(a - ref).max() / max(np.abs(ref).max(), 1e-300))


def hip_forward_backward(params, x, dlogits, device=0):
    """Runs unet_forward(training) + unet_backward through the C ABI on caller-owned buffers.
    Returns (logits, grads dict, handle, workspace, keepalive)."""
    import _hip
    import network
    L = _hip.lib()
    dev = torch.device("cuda", device)
    names = list(params.keys())
    B, _, S, _ = x.shape
    h = network._handle(device)
    plist = [torch.from_numpy(np.ascontiguousarray(params[k])).to(dev) for k in names]
    nbytes = h.workspace_bytes(B, S, True)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    xd = torch.from_numpy(x).to(dev)
    logits = torch.empty(B, 2, S - 184, S - 184, device=dev)
    ptab = _hip.ptr_table(plist)
    _hip.check(L.unet_forward(h.h, ptab, _hip.ptr(xd), _hip.ptr(logits), B, S, _hip.ptr(ws), nbytes, 1, _hip.stream()), "unet_forward")
    grads = [torch.empty_like(p) for p in plist]
    gtab = _hip.ptr_table(grads)
    dld = torch.from_numpy(np.ascontiguousarray(dlogits)).to(dev)
    _hip.check(L.unet_backward(h.h, ptab, _hip.ptr(dld), gtab, _hip.ptr(ws), nbytes, _hip.stream()), "unet_backward")
    torch.cuda.synchronize()
    return logits.cpu().numpy(), {k: g.cpu().numpy() for k, g in zip(names, grads)}, h, ws


def branch_of(h, ws, B, S):
    """ReLU masks (18) and pool winners (4) of the HIP forward, NCHW uint8."""
    masks = []
    for name in RELU_BUFS:
        v = h.buffer_view(ws, B, S, True, name)
        masks.append((v > 0).permute(0, 3, 1, 2).contiguous().to(torch.uint8).cpu().numpy())
    sels = []
    for l in range(4):
        v = h.buffer_view(ws, B, S, True, "a2_%d" % l)                      # [B,H,W,C]
        Bq, H, W, Cc = v.shape
        win = v.view(Bq, H // 2, 2, W // 2, 2, Cc).permute(0, 5, 1, 3, 2, 4).reshape(Bq, Cc, H // 2, W // 2, 4)
        sels.append(win.argmax(dim=-1).to(torch.uint8).cpu().numpy())       # first maximum on ties
    return masks, sels


def check_same_branch(S, B, seed_w=0, seed_x=1, seed_dl=2, dlogits=None):
    """Returns dict(fwd=..., grads={name: err}) of the HIP path against the fp64 oracle on HIP's branch."""
    params = prng.make_params(seed_w)
    x = prng.make_input(seed_x, B, S)
    dl = dlogits if dlogits is not None else prng.make_cotangent(seed_dl, (B, 2, S - 184, S - 184))
    logits, grads, h, ws = hip_forward_backward(params, x, dl)
    masks, sels = branch_of(h, ws, B, S)
    del ws
    p64 = {k: v.astype(np.float64) for k, v in params.items()}
    ref_logits, ref_grads = oracle_c.unet_fwd_bwd(p64, x.astype(np.float64), dlogits=np.asarray(dl, dtype=np.float64),
                                                  relu_masks=masks, pool_sel=sels)
    return {"fwd": nerr(logits, ref_logits), "grads": {k: nerr(grads[k], ref_grads[k]) for k in grads},
            "logits": logits, "hip_grads": grads, "ref_grads": ref_grads, "masks": masks, "sels": sels}


# layer whose ReLU output each entry of RELU_BUFS is (network.py:131-188)
RELU_LAYER = {"a1_%d" % l: "conv%d1c" % (l + 1) for l in range(5)}
RELU_LAYER.update({"a2_%d" % l: "conv%d2c" % (l + 1) for l in range(5)})
RELU_LAYER.update({"d1_%d" % l: "conv%d1e" % (l + 1) for l in range(4)})
RELU_LAYER.update({"d2_%d" % l: "conv%d2e" % (l + 1) for l in range(4)})


def branch_disagreements(masks, sels, S, B, seed_w=0, seed_x=1):
    """Where does the HIP forward's piecewise-linear branch differ from the fp64 forward's, and by how much did the
    fp64 value miss the decision boundary there?  Returns (n_relu, n_pool, worst): worst = the largest fp64 margin of
    a disagreeing element, normalised by its layer's scale (|z|max): a legitimate fp32 evaluation only ever disagrees
    where that margin is at the level of the forward rounding error."""
    import torch
    from . import torch_ref
    p = torch_ref.params_to_torch(prng.make_params(seed_w), torch.float64)
    x = torch.from_numpy(prng.make_input(seed_x, B, S)).double()
    pre = {}
    torch_ref.unet_forward(p, x, pre=pre)
    n_relu = n_pool = 0
    worst = 0.0
    for name, m in zip(RELU_BUFS, masks):
        z = pre[RELU_LAYER[name]].numpy()
        diff = (z > 0) != (m > 0)
        n_relu += int(diff.sum())
        if diff.any():
            worst = max(worst, float(np.abs(z[diff]).max() / np.abs(z).max()))
    for l, sel in enumerate(sels):
        a = np.maximum(pre["conv%d2c" % (l + 1)].numpy(), 0.0)                       # pooled tensor = ReLU output
        Bq, Cc, H, W = a.shape
        win = a.reshape(Bq, Cc, H // 2, 2, W // 2, 2).transpose(0, 1, 2, 4, 3, 5).reshape(Bq, Cc, H // 2, W // 2, 4)
        ref = win.argmax(axis=-1)                                                    # first maximum on ties
        diff = ref != sel
        n_pool += int(diff.sum())
        if diff.any():
            chosen = np.take_along_axis(win, sel[..., None].astype(np.int64), axis=-1)[..., 0]
            gap = (win.max(axis=-1) - chosen)[diff]
            worst = max(worst, float(gap.max() / max(np.abs(a).max(), 1e-300)))
    return n_relu, n_pool, worst


def shadow_training(net, train, val, epochs_arg, batch_size=2, lr=1e-4, mu=0.99, device=0):
    """The trainer's step sequence (trainer.py:39-135: epochs_arg + 1 epochs of train batches with class-balanced
    BCE-with-logits, backward, SGD(momentum), then the validation batches forward-only) run twice in lock step:
      * on the HIP path through the module API (the same calls dl-unet_amd/trainer.py makes), and
      * by the fp64 C oracle on its own fp64 weight trajectory, with every training forward/backward evaluated on the
        ReLU / pool branch the HIP forward of that step took (validation is forward-only: the forward is continuous in
        those decisions, so it needs no pinning).
    `net` is modified (trained).  Returns (hip, ref, final): the six progress series of each side and the normalised
    distance of the final HIP weights from the oracle's."""
    import network  # noqa: F401
    import optim as hip_optim
    from functions import class_balance, metrics_from_counts
    dev = torch.device("cuda", device)
    names = [k for k, _ in net.named_parameters()]
    p64 = {k: v.detach().cpu().numpy().astype(np.float64) for k, v in net.named_parameters()}
    buf64 = {}
    opt = hip_optim.SGD(net.parameters(), lr=lr, momentum=mu)
    keys = ("loss", "loss_val", "train_eval_iou", "train_eval_pe", "val_eval_iou", "val_eval_pe")
    hip = {k: [] for k in keys}
    ref = {k: [] for k in keys}

    def ref_weights(labels):
        # functions.py:82-117 in the reference's own arithmetic: float32 count ratio, then widened (the loss is fp64)
        lab = labels[:, 0]
        w = np.empty(lab.shape, dtype=np.float32)
        for b in range(lab.shape[0]):
            uval, counts = np.unique(lab[b], return_counts=True)
            for pos in range(len(uval)):
                w[b][lab[b] == uval[pos]] = np.float32(counts[1]) / np.float32(counts[pos])
        return w.astype(np.float64)

    def ref_metrics(logits64, labels):
        pred = oracle_c.argmax2(logits64)[0]
        lab = labels[0, 0]
        return float(np.logical_and(pred, lab).sum() / np.logical_or(pred, lab).sum()), float(np.abs(pred - lab).sum() / pred.size)

    for _ in range(epochs_arg + 1):
        tot_h = tot_r = 0.0
        m_h = m_r = None
        for images, labels in train:
            x_np, lab_np = images.numpy(), labels.numpy()
            B, _, S, _ = x_np.shape
            opt.zero_grad()
            logits = net(images.to(dev))
            ctx = logits.grad_fn                                        # the autograd node of _UnetFunction holds the workspace
            masks, sels = branch_of(net._get_handle(device), ctx.ws, B, S)
            lab_d = labels.to(dev)
            loss, _ = hip_optim.bce_argmax_step(logits, lab_d, weight=class_balance(lab_d.squeeze(1)), want_mask=False)
            loss.backward()
            opt.step()
            tot_h += float(loss.item())
            if m_h is None:
                _, stats = hip_optim.crop_argmax_metrics(logits.detach(), lab_d)
                i, u, d = [int(v) for v in stats[0].tolist()]
                mm = metrics_from_counts(i, u, d, lab_np.shape[-1] * lab_np.shape[-2])
                m_h = (float(mm[0, 0]), float(mm[1, 0]))
            # the shadow
            x64 = x_np.astype(np.float64)
            lg64, _ = oracle_c.unet_fwd_bwd(p64, x64, relu_masks=masks, pool_sel=sels)
            tgt = np.concatenate([1 - lab_np, lab_np], axis=1).astype(np.float64)
            w64 = ref_weights(lab_np)                                   # [B,H,W] right-aligned against [B,2,H,W] (quirk Q4)
            l64, dl64 = oracle_c.bce_logits(lg64, tgt, w=w64)
            _, g64 = oracle_c.unet_fwd_bwd(p64, x64, dlogits=dl64, relu_masks=masks, pool_sel=sels)
            first = not buf64
            for k in names:
                buf64[k] = g64[k].copy() if first else mu * buf64[k] + g64[k]
                p64[k] = p64[k] - lr * buf64[k]
            tot_r += float(l64)
            if m_r is None:
                m_r = ref_metrics(lg64, lab_np)
        tv_h = tv_r = 0.0
        v_h = v_r = None
        with torch.no_grad():
            for images, labels in val:
                x_np, lab_np = images.numpy(), labels.numpy()
                lab_d = labels.to(dev)
                logits = net(images.to(dev))
                tv_h += float(hip_optim.bce_argmax_step(logits, lab_d, weight=class_balance(lab_d.squeeze(1)), want_mask=False)[0].item())
                if v_h is None:
                    _, stats = hip_optim.crop_argmax_metrics(logits, lab_d)
                    i, u, d = [int(v) for v in stats[0].tolist()]
                    mm = metrics_from_counts(i, u, d, lab_np.shape[-1] * lab_np.shape[-2])
                    v_h = (float(mm[0, 0]), float(mm[1, 0]))
                lg64, _ = oracle_c.unet_fwd_bwd(p64, x_np.astype(np.float64))
                tgt = np.concatenate([1 - lab_np, lab_np], axis=1).astype(np.float64)
                l64, _ = oracle_c.bce_logits(lg64, tgt, w=ref_weights(lab_np), need_grad=False)
                tv_r += float(l64)
                if v_r is None:
                    v_r = ref_metrics(lg64, lab_np)
        for side, tot, totv, m, v in ((hip, tot_h, tv_h, m_h, v_h), (ref, tot_r, tv_r, m_r, v_r)):
            side["loss"].append(tot / (len(train) * batch_size))
            side["loss_val"].append(totv / (len(val) * batch_size))
            side["train_eval_iou"].append(m[0]); side["train_eval_pe"].append(m[1])
            side["val_eval_iou"].append(v[0]); side["val_eval_pe"].append(v[1])
    final = max(nerr(v.detach().cpu().numpy(), p64[k]) for k, v in net.named_parameters())
    return hip, ref, final


# ---- bf16 tensors (arithmetic mode 2, BASELINE configs[2]): the error model the S=572 test holds the path to ---------------------
# Storage roundings on the longest path input -> logits (network.py:129-192; 23 layers):
#   * 22 layer outputs are stored as bf16 (every conv / up-conv output; the head's logits stay fp32, pooling picks a stored value);
#   * 21 layers read bf16 copies of their filters (conv11c and the head multiply by the fp32 parameters).
# One round-to-nearest to bf16's 8 significant bits is a relative error uniform in [-u, u], u = 2^-8: variance u^2 / 3.
# First order, independent roundings: the output rounding adds u^2/3 to an element's relative variance; the filter roundings add
# (u^2/3) sum_k (w_k a_k)^2 / (sum_k w_k a_k)^2 = u^2/3 for terms of random sign (no systematic cancellation at the reference's
# random init: the tests below check this premise by emulation); a layer passes its input's relative error on with unit gain
# (it is linear up to the ReLU/pool selection, and relative error does not see the layer's scale).  Hence
#       sigma_rel = u sqrt((22 + 21) / 3) = 1.48e-2
# of an element's typical magnitude (the tensor's rms), and over N elements the largest error is the Gaussian tail
#       max |err| <= sqrt(2 ln N) sigma_rel rms(tensor).
BF16_U = 2.0 ** -8
BF16_ROUNDED_OUTPUTS = 22
BF16_ROUNDED_FILTERS = 21


def bf16_sigma_rel(outputs=BF16_ROUNDED_OUTPUTS, filters=BF16_ROUNDED_FILTERS):
    return BF16_U * np.sqrt((outputs + filters) / 3.0)


def bf16_max_err(n_elements, rms, outputs=BF16_ROUNDED_OUTPUTS, filters=BF16_ROUNDED_FILTERS):
    """Largest absolute error the model allows among n_elements values of a tensor with the given rms."""
    return float(np.sqrt(2.0 * np.log(max(int(n_elements), 2))) * bf16_sigma_rel(outputs, filters) * rms)


def bf16_expected_flips(margins, sigma_abs):
    """Expected number (and its standard deviation) of sign changes of `margins` under independent N(0, sigma_abs) errors."""
    from math import erfc, sqrt
    m = np.abs(np.asarray(margins, dtype=np.float64).ravel())
    m = m[m < 8 * sigma_abs]
    pr = np.array([0.5 * erfc(v / (sigma_abs * sqrt(2.0))) for v in m])
    return float(pr.sum()), float(np.sqrt((pr * (1 - pr)).sum()))
