"""ctypes binding of libunet_hip.so (include/unet_hip.h) — the only way the host reaches the GPU path.

There is deliberately no CPU fallback: if the shared library is missing or a tensor is not on a
HIP device, the call raises.  torch is used for device memory and streams only.
"""
import ctypes as C
import os
import subprocess

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.environ.get("UNET_SO") or os.path.join(_HERE, "libunet_hip.so")
_lib = None

N_PARAMS = 46

vp = C.c_void_p
_SIGS = {
    "unet_last_error": (C.c_char_p, []),
    "unet_abi_version": (C.c_int, []),
    "unet_set_math": (C.c_int, [C.c_int]),
    "unet_get_math": (C.c_int, []),
    "unet_set_lds_dma": (C.c_int, [C.c_int]),
    "unet_set_grad_scale": (C.c_int, [vp, C.c_float]),
    "unet_set_overlap": (C.c_int, [C.c_int]),
    "unet_dp_unique_id": (C.c_int, [vp]),
    "unet_dp_init": (C.c_int, [vp, C.c_int, C.c_int, vp]),
    "unet_dp_destroy": (C.c_int, [vp]),
    "unet_dp_world": (C.c_int, [vp]),
    "unet_dp_rccl_version": (C.c_int, []),
    "unet_dp_allreduce": (C.c_int, [vp, vp, C.c_size_t, vp]),
    "unet_dp_broadcast": (C.c_int, [vp, vp, C.c_size_t, C.c_int, vp]),
    "unet_dp_join": (C.c_int, [vp, vp]),
    "unet_create": (C.c_int, [C.POINTER(vp), vp]),
    "unet_destroy": (C.c_int, [vp]),
    "unet_output_size": (C.c_int, [C.c_int, C.POINTER(C.c_int)]),
    "unet_param_count": (C.c_int, [vp, C.c_int, C.POINTER(C.c_size_t)]),
    "unet_workspace_bytes": (C.c_size_t, [vp, C.c_int, C.c_int, C.c_int]),
    "unet_forward": (C.c_int, [vp, vp, vp, vp, C.c_int, C.c_int, vp, C.c_size_t, C.c_int, vp]),
    "unet_backward": (C.c_int, [vp, vp, vp, vp, vp, C.c_size_t, vp]),
    "unet_backward_stages": (C.c_int, []),
    "unet_backward_stage": (C.c_int, [vp, C.c_int, vp, vp, vp, vp, C.c_size_t, vp]),
    "unet_backward_stage_params": (C.c_int, [C.c_int, C.POINTER(C.c_int), C.c_int]),
    "unet_backward_input": (C.c_int, [vp, vp, vp, vp, C.c_size_t, vp]),
    "unet_flops": (C.c_double, [vp, C.c_int, C.c_int, C.c_int]),
    "unet_activation_bytes": (C.c_int, [vp]),
    "unet_debug_buffer": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, C.c_char_p, C.POINTER(C.c_size_t), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "unet_profile_enable": (C.c_int, [C.c_int]),
    "unet_profile_select": (C.c_int, [C.c_uint]),
    "unet_profile_reset": (C.c_int, []),
    "unet_profile_read": (C.c_int, [C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_long), C.POINTER(C.c_double),
                                    C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "unet_profile_dump": (C.c_int, [C.c_char_p]),
    "unet_bce_scratch_bytes": (C.c_size_t, [C.c_size_t]),
    "unet_bce_logits": (C.c_int, [vp, vp, vp, C.c_long, C.c_long, C.c_long, C.c_long, C.c_int, C.c_int, C.c_int,
                                  vp, vp, C.c_float, vp, vp]),
    "unet_bce_step_scratch_bytes": (C.c_size_t, [C.c_size_t]),
    "unet_bce_step": (C.c_int, [vp, C.c_long, C.c_long, C.c_long, vp, vp, C.c_long, C.c_long, C.c_long, C.c_long, C.c_int, C.c_int, C.c_int,
                                vp, vp, C.c_float, vp, vp, vp]),
    "unet_onehot2": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, vp]),
    "unet_argmax2": (C.c_int, [vp, C.c_long, C.c_long, C.c_long, vp, C.c_int, C.c_int, C.c_int, vp]),
    "unet_sgd_momentum": (C.c_int, [vp, vp, vp, C.POINTER(C.c_size_t), C.c_int, C.c_float, C.c_float, C.c_int, vp]),
    "unet_minmax": (C.c_int, [vp, C.c_int, C.c_size_t, vp, vp]),
    "unet_mirror_pad": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, vp, vp, vp]),
    "unet_eval_masks": (C.c_int, [vp, C.c_long, C.c_long, C.c_long, C.c_int, vp, vp, C.c_int, C.c_int, vp, vp]),
    "unet_class_balance": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, vp, vp, vp]),
    "unet_gaussian_filter": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, vp, C.c_int, C.c_float, vp, vp, vp]),
    "unet_warp_bilinear": (C.c_int, [vp, vp, vp, C.c_int, C.c_int, C.c_int, vp, vp]),
    "unet_rotate_scratch_bytes": (C.c_size_t, [C.c_int, C.c_int]),
    "unet_reflect_rotate_crop": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float), C.c_int, vp, vp, vp]),
    "unet_conv3x3_scratch_bytes": (C.c_size_t, [C.c_int, C.c_int]),
    "unet_conv3x3_fwd": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, C.c_int, C.c_int, C.c_int, C.c_int,
                                   vp, vp, C.c_int, C.c_int, vp, vp, vp]),
    "unet_conv3x3_bwd_scratch_bytes": (C.c_size_t, [C.c_int] * 5),
    "unet_conv3x3_bwd": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, C.c_int, C.c_int, C.c_int, C.c_int,
                                   vp, C.c_int, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]),
    "unet_maxpool2_fwd": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    "unet_maxpool2_bwd": (C.c_int, [vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    "unet_upconv2_scratch_bytes": (C.c_size_t, [C.c_int] * 5),
    "unet_upconv2_fwd": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, C.c_int, vp, vp, vp]),
    "unet_upconv2_bwd": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, C.c_int, vp, vp, vp, vp, vp, vp, vp]),
    "unet_head1x1_fwd": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp]),
    "unet_head1x1_bwd_scratch_bytes": (C.c_size_t, [C.c_int] * 4),
    "unet_head1x1_bwd": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, vp, vp, vp]),
    "unet_conv1ch_fwd": (C.c_int, [vp, C.c_int, C.c_int, vp, vp, C.c_int, vp, vp]),
    "unet_conv1ch_bwd_scratch_bytes": (C.c_size_t, [C.c_int] * 3),
    "unet_conv1ch_bwd": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, vp]),
}
EXPORTS = tuple(_SIGS.keys())


class UnetConfig(C.Structure):
    _fields_ = [("base_ch", C.c_int), ("device", C.c_int), ("math", C.c_int)]


def build(force=False):
    """Compile libunet_hip.so for gfx950 with hipcc (cross-compiles without a GPU)."""
    csrc = os.path.join(_HERE, "csrc")
    if force and os.path.exists(_SO):
        os.remove(_SO)
    subprocess.check_call(["make", "-s", "-C", csrc, "-j4"])
    return _SO


def lib():
    """Load the HIP library; raises (never falls back) when it is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            raise RuntimeError(
                "libunet_hip.so is not built (%s). Run `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C dl-unet_amd/csrc`. There is no CPU fallback for the HIP path." % _SO)
        L = C.CDLL(_SO)
        for name, (res, args) in _SIGS.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(rc, what=""):
    if rc != 0:
        msg = lib().unet_last_error().decode("utf-8", "replace")
        raise RuntimeError("%s failed (rc=%d): %s" % (what or "libunet_hip", rc, msg))


def ptr(t):
    """Device pointer of a torch tensor (None -> NULL). Refuses host tensors loudly."""
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("the HIP path needs tensors on a HIP device (got %s); there is no CPU fallback" % t.device)
    return C.c_void_p(t.data_ptr())


def stream(device=None):
    """The current HIP stream of `device` (default: the current device) as a void*."""
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def run(name, device, *args):
    """lib().<name>(*args, stream): `device` is made current for the call (everything the library keeps per device —
    zero page, kernel attributes — is keyed on the current device) and the work goes to that device's current stream."""
    with torch.cuda.device(device):
        check(getattr(lib(), name)(*args, stream(device)), name)


def ptr_table(tensors):
    arr = (C.c_void_p * len(tensors))()
    for i, t in enumerate(tensors):
        if not t.is_cuda:
            raise RuntimeError("parameter %d is on %s; move the module to a HIP device" % (i, t.device))
        arr[i] = t.data_ptr()
    return arr


class Handle:
    """RAII wrapper of unet_handle (one per device / module)."""

    def __init__(self, base_ch=64, device=0, math=-1):
        """math: arithmetic mode of this handle (include/unet_hip.h unet_set_math codes); -1 follows unet_set_math."""
        self._h = C.c_void_p()
        cfg = UnetConfig(base_ch, device, math)
        check(lib().unet_create(C.byref(self._h), C.byref(cfg)), "unet_create")
        self.base_ch = base_ch
        self.device = device
        self.math = math

    def __del__(self):
        try:
            if self._h:
                lib().unet_destroy(self._h)
                self._h = None
        except Exception:
            pass

    @property
    def h(self):
        return self._h

    def workspace_bytes(self, B, S, training):
        n = lib().unet_workspace_bytes(self._h, B, S, int(training))
        if n == 0:
            check(-1, "unet_workspace_bytes")
        return n

    def buffer_view(self, ws, B, S, training, name):
        """NHWC [B,e,e,C] float32 view of a named plan buffer inside workspace tensor `ws` (debug/tests)."""
        off, e, c = C.c_size_t(), C.c_int(), C.c_int()
        check(lib().unet_debug_buffer(self._h, B, S, int(training), name.encode(), C.byref(off), C.byref(e), C.byref(c)),
              "unet_debug_buffer")
        n = B * e.value * e.value * c.value
        eb = 4 if name == "xin" else lib().unet_activation_bytes(self._h)          # bf16 tensors in arithmetic mode 2
        return ws[off.value:off.value + eb * n].view(torch.bfloat16 if eb == 2 else torch.float32).view(B, e.value, e.value, c.value)

    def flops(self, B, S, backward):
        return lib().unet_flops(self._h, B, S, int(backward))
