"""Accuracy and speed of the three arithmetic modes of the dense contractions (UNET_MATH / unet_set_math)."""
import sys, json, subprocess, os
sys.path.insert(0, 'dl-unet_amd'); sys.path.insert(0, '.')
import numpy as np, torch
import _hip, network
from oracle import parity, prng
L = _hip.lib()
for mode, name in ((0, "fp32 MFMA"), (1, "bf16x3"), (2, "bf16")):
    _hip.check(L.unet_set_math(mode))
    r = parity.check_same_branch(220, 2)
    worst = max(r["grads"].items(), key=lambda kv: kv[1])
    r5 = parity.check_same_branch(380, 1)
    print("mode %d %-10s S=220: logits err %.3g, worst same-branch grad err %.3g (%s) | S=380: logits %.3g grads %.3g" % (
        mode, name, r["fwd"], worst[1], worst[0], r5["fwd"], max(r5["grads"].values())), flush=True)
_hip.check(L.unet_set_math(0))
