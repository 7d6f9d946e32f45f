"""TEST INFRASTRUCTURE — ctypes binding of liboracle.so (oracle/unet_oracle.c)."""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build():
    so = os.path.join(_HERE, "liboracle.so")
    srcs = [os.path.join(_HERE, f) for f in ("unet_oracle.c", "unet_oracle_body.h")]
    if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-s", "-C", _HERE, "liboracle.so"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        for sfx in ("_f32", "_f64"):
            getattr(_LIB, "oracle_bce_logits" + sfx).restype = C.c_double
    return _LIB


def _sfx(dt):
    return "_f64" if np.dtype(dt) == np.float64 else "_f32"


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def unet_fwd_bwd(params, x, base=64, dlogits=None, relu_masks=None, pool_sel=None):
    """params: ordered dict of 46 arrays; x [N,1,S,S]. Returns (logits, grads|None).
    relu_masks (18 uint8 NCHW arrays: a1_l,a2_l at 2l,2l+1; d1_l,d2_l at 10+2l,11+2l) and pool_sel
    (4 uint8 NCHW arrays, winner 0..3) evaluate the net on the SAME piecewise-linear branch as
    another run (see unet_oracle_body.h)."""
    dt = x.dtype
    names = list(params.keys())
    arrs = [np.ascontiguousarray(params[k], dtype=dt) for k in names]
    N, _, S, _ = x.shape
    So = S - 184
    logits = np.empty((N, 2, So, So), dtype=dt)
    PT = C.c_void_p * len(arrs)
    pp = PT(*[a.ctypes.data for a in arrs])
    grads = gp = None
    if dlogits is not None:
        grads = [np.empty_like(a) for a in arrs]
        gp = PT(*[g.ctypes.data for g in grads])
        dlogits = np.ascontiguousarray(dlogits, dtype=dt)
    mp = sp = None
    if relu_masks is not None:
        relu_masks = [np.ascontiguousarray(m, dtype=np.uint8) for m in relu_masks]
        pool_sel = [np.ascontiguousarray(m, dtype=np.uint8) for m in pool_sel]
        assert len(relu_masks) == 18 and len(pool_sel) == 4
        mp = (C.c_void_p * 18)(*[m.ctypes.data for m in relu_masks])
        sp = (C.c_void_p * 4)(*[m.ctypes.data for m in pool_sel])
    fn = getattr(lib(), "oracle_unet_fwd_bwd" + _sfx(dt))
    rc = fn(pp, _p(np.ascontiguousarray(x)), N, S, base, _p(logits), _p(dlogits), gp, mp, sp)
    if rc != 0:
        raise RuntimeError("oracle: size rejected (reference would raise), rc=%d" % rc)
    return logits, (dict(zip(names, grads)) if grads is not None else None)


def conv_valid_fwd(x, w, b, relu):
    dt = x.dtype
    N, Cc, H, W = x.shape
    K, _, R, _ = w.shape
    y = np.empty((N, K, H - R + 1, W - R + 1), dtype=dt)
    getattr(lib(), "oracle_conv_valid_fwd" + _sfx(dt))(_p(x), _p(w), _p(b), _p(y), N, Cc, H, W, K, R, int(relu))
    return y


def conv_valid_bwd(x, w, dy, need_dx=True):
    dt = x.dtype
    N, Cc, H, W = x.shape
    K, _, R, _ = w.shape
    dx = np.empty_like(x) if need_dx else None
    dw = np.empty_like(w)
    db = np.empty((K,), dtype=dt)
    getattr(lib(), "oracle_conv_valid_bwd" + _sfx(dt))(_p(x), _p(w), _p(dy), _p(dx), _p(dw), _p(db), N, Cc, H, W, K, R)
    return dx, dw, db


def maxpool2_fwd(x):
    dt = x.dtype
    N, Cc, H, W = x.shape
    y = np.empty((N, Cc, H // 2, W // 2), dtype=dt)
    idx = np.empty((N, Cc, H // 2, W // 2), dtype=np.uint8)
    getattr(lib(), "oracle_maxpool2_fwd" + _sfx(dt))(_p(x), _p(y), _p(idx), N, Cc, H, W)
    return y, idx


def maxpool2_bwd(dy, idx, H, W):
    dt = dy.dtype
    N, Cc = dy.shape[:2]
    dx = np.empty((N, Cc, H, W), dtype=dt)
    getattr(lib(), "oracle_maxpool2_bwd" + _sfx(dt))(_p(dy), _p(idx), _p(dx), N, Cc, H, W)
    return dx


def upconv2_fwd(x, w, b):
    dt = x.dtype
    N, Ci, H, W = x.shape
    Co = w.shape[1]
    y = np.empty((N, Co, 2 * H, 2 * W), dtype=dt)
    getattr(lib(), "oracle_upconv2_fwd" + _sfx(dt))(_p(x), _p(w), _p(b), _p(y), N, Ci, H, W, Co)
    return y


def upconv2_bwd(x, w, dy):
    dt = x.dtype
    N, Ci, H, W = x.shape
    Co = w.shape[1]
    dx = np.empty_like(x); dw = np.empty_like(w); db = np.empty((Co,), dtype=dt)
    getattr(lib(), "oracle_upconv2_bwd" + _sfx(dt))(_p(x), _p(w), _p(dy), _p(dx), _p(dw), _p(db), N, Ci, H, W, Co)
    return dx, dw, db


def crop_and_concat(A, B):
    dt = A.dtype
    N, Ca, Ha, _ = A.shape
    _, Cb, Hb, _ = B.shape
    out = np.empty((N, Ca + Cb, Hb, Hb), dtype=dt)
    rc = getattr(lib(), "oracle_crop_and_concat_fwd" + _sfx(dt))(_p(A), _p(B), _p(out), N, Ca, Ha, Cb, Hb)
    if rc:
        raise RuntimeError("crop_and_concat: sizes do not match (reference torch.cat raises, Q7)")
    return out


def bce_logits(x, z, w=None, need_grad=True):
    dt = x.dtype
    x = np.ascontiguousarray(x); z = np.ascontiguousarray(z, dtype=dt)
    if w is not None:
        w = np.ascontiguousarray(np.broadcast_to(w, x.shape), dtype=dt)
    dx = np.empty_like(x) if need_grad else None
    loss = getattr(lib(), "oracle_bce_logits" + _sfx(dt))(_p(x), _p(z), _p(w), _p(dx), C.c_size_t(x.size))
    return loss, dx


def argmax2(x):
    dt = x.dtype
    N, two, H, W = x.shape
    assert two == 2
    out = np.empty((N, H, W), dtype=np.int64)
    getattr(lib(), "oracle_argmax2" + _sfx(dt))(_p(np.ascontiguousarray(x)), _p(out), N, C.c_size_t(H * W))
    return out


def sgd_momentum(p, g, buf, lr, mu, first):
    dt = p.dtype
    ct = C.c_double if dt == np.float64 else C.c_float
    getattr(lib(), "oracle_sgd_momentum" + _sfx(dt))(_p(p), _p(g), _p(buf), C.c_size_t(p.size), ct(lr), ct(mu), int(first))


def input_size_compute(original):
    a = C.c_int(); b = C.c_int()
    lib().oracle_input_size_compute(int(original), C.byref(a), C.byref(b))
    return original, a.value, b.value
