"""TEST INFRASTRUCTURE — torch restatement of the reference graph (CPU baseline).

The same torch.nn.functional calls, shapes and concat order as Unet.forward
(network.py:129-192), taking an explicit parameter dict.  It is what bench.py times
as `cpu_baseline` (kind "port") on the GPU host's cores and what the parity tests
compare the HIP path against at full size; it is validated against the imported
reference by tests/golden/make_golden.py (here only; the reference cannot travel).
"""
import torch
import torch.nn.functional as F


def crop_and_concat(A, B):
    # network.py:108-127: c = int((A-B)/2), negative pad == crop, c<0 zero-pads.
    c = int((A.size(2) - B.size(2)) * 0.5)
    return torch.cat((F.pad(A, (-c, -c, -c, -c)), B), 1)


def unet_forward(p, t, pre=None, store=None, wstore=None):
    """pre: optional dict that receives every ReLU'd conv's PRE-activation by layer name (branch analysis in the tests).
    store / wstore: optional emulation of a reduced-precision tensor format: store(x) is applied to every conv / up-conv output
    as it would be written to memory (after bias and ReLU), wstore(w) to the filters of every layer but conv11c and the head -
    the roundings of the HIP path's bf16-tensor mode (oracle/parity.py, bf16 error model)."""
    store = store or (lambda v: v)

    def wt(name):
        w = p[name + ".weight"]
        return wstore(w) if wstore is not None and name not in ("conv11c", "finalconv") else w

    def cr(name, t):
        z = F.conv2d(t, wt(name), p[name + ".bias"])
        if pre is not None:
            pre[name] = z.detach()
        return store(F.relu(z))

    def up(name, t):
        return store(F.conv_transpose2d(t, wt(name), p[name + ".bias"], stride=2))

    skips = []
    for lvl in "1234":
        t = cr("conv%s1c" % lvl, t)
        t = cr("conv%s2c" % lvl, t)
        t = F.max_pool2d(t, kernel_size=2, stride=2)
        skips.append(t)                      # skip is taken AFTER the pool
    t = cr("conv51c", t)
    t = cr("conv52c", t)
    for lvl in "4321":
        t = up("upconv" + lvl, t)
        t = crop_and_concat(skips[int(lvl) - 1], t)
        t = cr("conv%s1e" % lvl, t)
        t = cr("conv%s2e" % lvl, t)
    return F.conv2d(t, p["finalconv.weight"], p["finalconv.bias"])


def params_to_torch(np_params, dtype=torch.float32, requires_grad=False):
    return {k: torch.from_numpy(v).to(dtype).requires_grad_(requires_grad) for k, v in np_params.items()}


def train_step(p, mom, x, target, lr=1e-4, mu=0.99, first=False):
    """fwd + unweighted BCE-with-logits + bwd + SGD(momentum) (trainer.py:52-82; Q4)."""
    for v in p.values():
        v.grad = None
    logits = unet_forward(p, x)
    loss = F.binary_cross_entropy_with_logits(logits, target)
    loss.backward()
    with torch.no_grad():
        for k, v in p.items():
            if first:
                mom[k] = v.grad.clone()
            else:
                mom[k].mul_(mu).add_(v.grad)
            v.sub_(mom[k], alpha=lr)
    return loss.detach(), logits.detach()
