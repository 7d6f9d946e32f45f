"""Checkpoint / resume for the HIP training path.

The reference saves `unet.state_dict()` only (trainer.py:143-144,187-219) and its resume (`-sf`, main_main.py:202-205,
:244-249) parses an integer out of `unet_weight_save_<N>.pth` although the trainer only ever writes `best`, `latest`
and `<DATASET>` (quirk Q6: int('best') raises).  Kept: the 46-key fp32 OIHW/IOHW state dict, so the reference's .pth
files load here and ours load there.  Added: the SGD momentum buffers (and scheduler state), without which a resumed
run is not the run that was interrupted — momentum 0.99 remembers ~100 steps.

  save_checkpoint(path, unet, optimizer, scheduler=None, **extra)   one file: model + optimizer (+ scheduler) state
  load_checkpoint(path, unet, optimizer=None, scheduler=None)       restores them; returns the `extra` dict
  find_resume_checkpoint(models_dir)                                'latest' if present, else 'best' (what -sf meant to do)
Only tensors, numbers and strings are stored: files load with torch.load(weights_only=True).
"""
import os

import torch


def save_checkpoint(path, unet, optimizer, scheduler=None, **extra):
    ckpt = {"format": "dl-unet_amd/1", "model": unet.state_dict(), "optimizer": optimizer.state_dict(), "extra": dict(extra)}
    if scheduler is not None:
        ckpt["scheduler"] = {k: v for k, v in scheduler.state_dict().items() if isinstance(v, (int, float, str, bool, list, type(None)))}
    tmp = path + ".tmp"
    torch.save(ckpt, tmp)
    os.replace(tmp, path)                     # a crash mid-write never leaves a truncated checkpoint behind
    return path


def load_checkpoint(path, unet, optimizer=None, scheduler=None, map_location=None):
    """Accepts our checkpoints and the reference's bare state dicts (then only the weights are restored)."""
    ckpt = torch.load(path, map_location=map_location, weights_only=True)
    if not (isinstance(ckpt, dict) and "model" in ckpt and "format" in ckpt):
        unet.load_state_dict(ckpt)
        return {}
    unet.load_state_dict(ckpt["model"])
    if optimizer is not None and "optimizer" in ckpt:
        optimizer.load_state_dict(ckpt["optimizer"])          # momentum buffers land on the parameters' device
    if scheduler is not None and "scheduler" in ckpt:
        st = scheduler.state_dict()
        st.update(ckpt["scheduler"])
        scheduler.load_state_dict(st)
    return ckpt.get("extra", {})


def find_resume_checkpoint(models_dir):
    for name in ("checkpoint_latest.pth", "unet_weight_save_latest.pth", "unet_weight_save_best.pth"):
        p = os.path.join(models_dir, name)
        if os.path.exists(p):
            return p
    return None
