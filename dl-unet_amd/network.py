"""Drop-in for the reference's network.py: `from network import Unet`.

Same class, constructor (no arguments), submodule names, 46 state-dict keys/shapes/layouts and call
surface as the reference (network.py:8-192), but `forward` runs the hand-written HIP path for gfx950
through the C ABI of libunet_hip.so (include/unet_hip.h) and backward is the library's own
dgrad/wgrad kernels behind one torch.autograd.Function.  PyTorch supplies parameter storage,
device memory and streams; for data parallel the gradient all-reduce is the library's RCCL call
(unet_dp_*), torch.distributed only carries the 128-byte rendezvous id between the ranks.

There is no CPU fallback: calling the module with a host tensor raises.
"""
import ctypes as C

import torch
import torch.nn as nn
import torch.nn.functional as F

import _hip
import dp as dp_mod

_BASE = 64          # hard-coded in the reference (network.py:23-58)


def _layers(base=_BASE):
    """(name, kind, cin, cout, k) in the reference's declaration order; base = first-level width
    (64 in the reference; 32 is the BASELINE config #5 variant)."""
    c = [base, 2 * base, 4 * base, 8 * base, 16 * base]
    return [
        ("conv11c", "conv", 1, c[0], 3), ("conv12c", "conv", c[0], c[0], 3),
        ("conv21c", "conv", c[0], c[1], 3), ("conv22c", "conv", c[1], c[1], 3),
        ("conv31c", "conv", c[1], c[2], 3), ("conv32c", "conv", c[2], c[2], 3),
        ("conv41c", "conv", c[2], c[3], 3), ("conv42c", "conv", c[3], c[3], 3),
        ("conv51c", "conv", c[3], c[4], 3), ("conv52c", "conv", c[4], c[4], 3),
        ("upconv4", "up", c[4], c[3], 2), ("conv41e", "conv", c[4], c[3], 3), ("conv42e", "conv", c[3], c[3], 3),
        ("upconv3", "up", c[3], c[2], 2), ("conv31e", "conv", c[3], c[2], 3), ("conv32e", "conv", c[2], c[2], 3),
        ("upconv2", "up", c[2], c[1], 2), ("conv21e", "conv", c[2], c[1], 3), ("conv22e", "conv", c[1], c[1], 3),
        ("upconv1", "up", c[1], c[0], 2), ("conv11e", "conv", c[1], c[0], 3), ("conv12e", "conv", c[0], c[0], 3),
        ("finalconv", "conv", c[0], 2, 1),
    ]


_LAYERS = _layers()


def _init_std(name, cin):
    """std of the reference's weight re-initialisation (network.py:70-105).  As written there the
    expression is 2 / sqrt(N) (operator precedence), N counted with 3x3 kernels
    even for the up-convs and the 1x1 head; conv_k1e counts its two halves as C*9 + C*4."""
    if name == "conv11c":
        return 2 ** 0.5
    if name.endswith("1e"):
        half = cin // 2
        return 2.0 / (half * 9 + half * 4) ** 0.5
    return 2.0 / (cin * 9) ** 0.5


_handles = {}


def _handle(device_index, base=_BASE):
    h = _handles.get((device_index, base))
    if h is None:
        h = _hip.Handle(base, device_index)
        _handles[(device_index, base)] = h
    return h


def _stage_layout():
    """Gradient flat-buffer order = completion order of the backward stages (reverse layer order),
    so every stage's parameters form one contiguous bucket for the all-reduce."""
    L = _hip.lib()
    order, bounds = [], [0]
    for s in range(L.unet_backward_stages()):
        buf = (C.c_int * 64)()
        n = L.unet_backward_stage_params(s, buf, 64)
        order.extend(int(buf[i]) for i in range(n))
        bounds.append(len(order))
    return order, bounds


class _UnetFunction(torch.autograd.Function):
    """forward: unet_forward (replaces network.py:129-192); backward: unet_backward_stage x6
    (replaces the autograd graph of every op in it), with an optional bucketed gradient all-reduce."""

    @staticmethod
    def forward(ctx, x, module, *params):
        dev = x.device.index
        h = module._get_handle(dev)
        B, _, S, _ = x.shape
        So = S - 184
        with torch.cuda.device(x.device):               # the library checks that the handle's device is current
            nbytes = h.workspace_bytes(B, S, True)
            ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
            logits = torch.empty(B, 2, So, So, dtype=torch.float32, device=x.device)
            ptab = _hip.ptr_table(params)
            _hip.check(_hip.lib().unet_forward(h.h, ptab, _hip.ptr(x), _hip.ptr(logits), B, S, _hip.ptr(ws), nbytes, 1,
                                               _hip.stream(x.device)), "unet_forward")
        ctx.save_for_backward(*params)
        ctx.ws, ctx.nbytes, ctx.dev, ctx.module, ctx.x_shape = ws, nbytes, dev, module, tuple(x.shape)
        return logits

    @staticmethod
    def backward(ctx, dlogits):
        params = ctx.saved_tensors
        module = ctx.module
        h = module._get_handle(ctx.dev)
        L = _hip.lib()
        buckets = module._buckets
        dp = module._dp
        with torch.cuda.device(dlogits.device):
            flat, grads = buckets.allocate([p.shape for p in params], dlogits.device)
            dlogits = dlogits.contiguous()     # with data parallel the handle reads it as dlogits / world (unet_set_grad_scale)
            ptab, gtab = _hip.ptr_table(params), _hip.ptr_table(grads)
            st = _hip.stream(dlogits.device)
            for s in range(buckets.n_stages()):
                _hip.check(L.unet_backward_stage(h.h, s, ptab, _hip.ptr(dlogits), gtab, _hip.ptr(ctx.ws), ctx.nbytes, st),
                           "unet_backward_stage %d" % s)
                if dp is not None:
                    # all-reduce of this stage's bucket on the communicator's stream, overlapping the next stage
                    dp.reduce_stage(buckets, flat, s, h, st)
            if dp is not None:
                dp.join(h, st)
            dx = None
            if ctx.needs_input_grad[0]:
                # d loss / d image (the reference's callers never ask for it): conv11c's dgrad from the stashed dz
                dx = torch.empty(ctx.x_shape, dtype=torch.float32, device=dlogits.device)
                _hip.check(L.unet_backward_input(h.h, ptab, _hip.ptr(dx), _hip.ptr(ctx.ws), ctx.nbytes, st), "unet_backward_input")
        ctx.ws = None
        return (dx, None) + tuple(grads)


class Unet(nn.Module):
    """Unet 2D (Ronneberger et al. 2015) with the reference's exact structure and quirks:
    valid 3x3 convs, skips taken AFTER the pool and zero-padded to the up-conv size,
    1x1 head without activation; fp32; input [B,1,S,S] with S = 16L+60, L even."""

    def __init__(self, base_ch=_BASE):
        """Unet() as in the reference (no arguments).  base_ch is an extension: 32 gives the half-width
        net of BASELINE config #5 (forward and backward; its 32-channel layers take the implicit-GEMM
        kernels and half-filled weight-gradient tiles instead of the Winograd ones)."""
        super(Unet, self).__init__()
        self.base_ch = base_ch
        self._layer_table = _layers(base_ch)
        # nn.Conv2d / nn.ConvTranspose2d serve ONLY as parameter containers with the reference's
        # names, shapes and default-init RNG consumption (their forward is never called).
        for name, kind, cin, cout, k in self._layer_table:
            if kind == "conv":
                setattr(self, name, nn.Conv2d(in_channels=cin, out_channels=cout, kernel_size=k))
            else:
                setattr(self, name, nn.ConvTranspose2d(in_channels=cin, out_channels=cout, kernel_size=k, stride=k))
        # weights are then re-drawn in declaration order (network.py:70-105), biases keep the default
        for name, kind, cin, cout, k in self._layer_table:
            m = getattr(self, name)
            m.weight = nn.Parameter(torch.empty_like(m.weight).normal_(mean=0, std=_init_std(name, cin)))
        self._dp = None
        self._buckets = None
        self._own_handle = None

    # -- data parallel: batch-sharded replicas, gradients averaged with RCCL over xGMI -------------
    def enable_data_parallel(self, process_group=None, backend="rccl"):
        """Batch-sharded data parallel, one process per GPU.  backend "rccl": the library's own RCCL
        communicator (unet_dp_*; torch.distributed, if initialised, only carries the rendezvous id — without it a
        one-rank communicator is made); backend "torch": torch.distributed.all_reduce on the given group (what the
        CPU/gloo tests use).  Parameters are broadcast from rank 0."""
        params = self._params()
        if not all(p.is_cuda for p in params):
            raise RuntimeError("enable_data_parallel: move the module to its HIP device first (.to('cuda:N'))")
        dev = params[0].device
        # a handle of this module's own: the communicator, its stream and the 1/world gradient scale live in the handle,
        # and the per-device handle is shared by every other Unet on the device
        self._own_handle = _hip.Handle(self.base_ch, dev.index)
        self._dp = dp_mod.DataParallel(self._own_handle, dev, process_group, backend)
        self._dp.broadcast_parameters(params)
        return self

    def _get_handle(self, device_index):
        h = self._own_handle
        if h is not None:
            if h.device != device_index:
                raise RuntimeError("this Unet was set up for data parallel on device %d; it cannot run on device %d" % (h.device, device_index))
            return h
        return _handle(device_index, self.base_ch)

    def _params(self):
        out = []
        for name, _, _, _, _ in self._layer_table:
            m = getattr(self, name)
            out.append(m.weight)
            out.append(m.bias)
        return out

    def crop_and_concat(self, A, B):
        """Public helper kept for API parity (network.py:108-127): crop (A larger) or ZERO-PAD (A
        smaller, the case every valid input takes) A to B's extent and concatenate on channels.
        The forward pass never materialises this tensor — the consuming conv reads both sources."""
        crop_factor = (A.size()[2] - B.size()[2]) * 0.5
        c = int(crop_factor)
        A = F.pad(A, (-c, -c, -c, -c))
        return torch.cat((A, B), 1)

    def forward(self, t):
        if not t.is_cuda:
            raise RuntimeError("Unet.forward: the HIP path needs the input on a HIP device (got %s); "
                               "there is no CPU fallback — use unet.to('cuda:0') and images.to('cuda:0')" % t.device)
        if t.dim() != 4 or t.shape[1] != 1 or t.shape[2] != t.shape[3]:
            raise RuntimeError("Unet.forward: expected input [B,1,S,S], got %s" % (tuple(t.shape),))
        if t.dtype != torch.float32:
            raise RuntimeError("Unet.forward: expected float32 input, got %s" % t.dtype)
        t = t.contiguous()
        params = self._params()
        if self._buckets is None:
            order, bounds = _stage_layout()
            self._buckets = dp_mod.GradBuckets([p.numel() for p in params], order, bounds)
        if torch.is_grad_enabled() and (t.requires_grad or any(p.requires_grad for p in params)):
            return _UnetFunction.apply(t, self, *params)
        # inference (trainer.py:95 no_grad): no activation-gradient storage
        h = self._get_handle(t.device.index)
        B, _, S, _ = t.shape
        with torch.cuda.device(t.device):
            nbytes = h.workspace_bytes(B, S, False)
            ws = torch.empty(nbytes, dtype=torch.uint8, device=t.device)
            logits = torch.empty(B, 2, S - 184, S - 184, dtype=torch.float32, device=t.device)
            _hip.check(_hip.lib().unet_forward(h.h, _hip.ptr_table([p.detach() for p in params]), _hip.ptr(t), _hip.ptr(logits),
                                               B, S, _hip.ptr(ws), nbytes, 0, _hip.stream(t.device)), "unet_forward")
        return logits
