"""Generates tests/golden/*.npz by IMPORTING the reference (this container only).

Run:  python tests/golden/make_golden.py          (needs /root/reference; ~2-3 min)

The reference's network.py / functions.py are imported from /root/reference with inert
stand-ins for modules that are absent here and unused on the path (torchvision in
network.py:5-6; cv2 in functions.py:1, used only by weighted_map).  Weights, inputs and
cotangents come from oracle/prng.py so that the GPU box can regenerate them; only
OUTPUTS are stored.  Every fixture records torch version and thread count.

Nothing from the reference is stored: fixtures are numbers (inputs' seeds + outputs).
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import prng  # noqa: E402

REF = "/root/reference"


def import_reference():
    for name in ("torchvision", "torchvision.transforms", "cv2"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    sys.path.insert(0, REF)
    import network  # noqa
    import functions  # noqa
    return network, functions


def checksum(a, nsamp=16):
    a = np.asarray(a, dtype=np.float64).ravel()
    idx = (np.arange(nsamp, dtype=np.int64) * max(1, a.size // nsamp)) % a.size
    return np.array([a.sum(), np.sqrt((a * a).sum()), np.abs(a).max()]), idx, a[idx]


def run_net(network, params_np, x_np, dl_np, dtype):
    torch.manual_seed(0)
    net = network.Unet()
    sd = {k: torch.from_numpy(v.astype(np.float64)).to(dtype) for k, v in params_np.items()}
    net = net.to(dtype)
    net.load_state_dict(sd)
    x = torch.from_numpy(x_np.astype(np.float64)).to(dtype)
    y = net(x)
    out = {"logits": y.detach().numpy().astype(np.float64)}
    if dl_np is not None:
        y.backward(torch.from_numpy(dl_np.astype(np.float64)).to(dtype))
        for k, p in net.named_parameters():
            out["grad." + k] = p.grad.detach().numpy().astype(np.float64)
    return out


def main_s572_grad():
    """G3b: the BASELINE tile size itself, B=1: logits and all 46 gradients of the reference in fp32 and fp64 as per-tensor
    checksums + 64 strided samples (unet_S572_grad.npz; ~10 min, the fp64 backward dominates).
    Run:  python tests/golden/make_golden.py s572grad"""
    network, _ = import_reference()
    torch.set_num_threads(8)
    S, B = 572, 1
    params = prng.make_params(seed=0)
    names = list(params.keys())
    x = prng.make_input(1, B, S)
    dl = prng.make_cotangent(2, (B, 2, S - 184, S - 184))
    fx = {"meta": np.array(repr(dict(torch=torch.__version__, threads=torch.get_num_threads(), seed_w=0, S=S, B=B, seed_x=1, seed_dl=2)))}
    for tag, dt in (("f32", torch.float32), ("f64", torch.float64)):
        r = run_net(network, params, x, dl, dt)
        fx["logits_sample_" + tag] = r["logits"][:, :, ::6, ::6].copy()
        sums, samp_idx, samp = [], [], []
        for k in names:
            s, i, v = checksum(r["grad." + k], nsamp=64)
            sums.append(s); samp_idx.append(i); samp.append(v)
        fx["grad_sums_" + tag] = np.stack(sums)
        fx["grad_samp_" + tag] = np.stack(samp)
        fx["grad_samp_idx"] = np.stack(samp_idx)
        for k in ("conv11c.weight", "conv11c.bias", "finalconv.weight", "finalconv.bias", "conv52c.bias", "upconv4.bias"):
            fx["grad_full_%s_%s" % (k, tag)] = r["grad." + k]
        print("S=572 %s done" % tag, flush=True)
    fx["names"] = np.array(names)
    np.savez_compressed(os.path.join(HERE, "unet_S572_grad.npz"), **fx)
    e = np.abs(fx["grad_samp_f32"] - fx["grad_samp_f64"]).max(axis=1) / fx["grad_sums_f64"][:, 2]
    print("S=572 gradients done: reference's own fp32-vs-fp64 distance per tensor: worst %.3g (%s)" % (e.max(), names[int(e.argmax())]))


def main_s572_margin():
    """G3c: the class margin d = logit1 - logit0 of the reference's fp64 forward at S=572, B=1, for EVERY output pixel
    (float32, 602 KB): what a reduced-precision path (bf16 tensors, BASELINE configs[2]) needs to show that each argmax
    pixel it flips sat closer to the decision boundary than its own error bound (unet_S572_margin.npz).
    Run:  python tests/golden/make_golden.py s572margin"""
    network, _ = import_reference()
    torch.set_num_threads(8)
    S, B = 572, 1
    params = prng.make_params(seed=0)
    x = prng.make_input(1, B, S)
    r64 = run_net(network, params, x, None, torch.float64)["logits"]
    d = r64[:, 1] - r64[:, 0]
    g = np.load(os.path.join(HERE, "unet_S572_fwd.npz"))
    assert np.array_equal(np.packbits((d > 0).ravel()), g["argmax_packed"]), "the fp64 forward no longer reproduces unet_S572_fwd.npz"
    fx = {"meta": np.array(repr(dict(torch=torch.__version__, threads=torch.get_num_threads(), seed_w=0, S=S, B=B, seed_x=1))),
          "margin_f64_as_f32": d.astype(np.float32),
          "margin_rms": np.array(np.sqrt((d * d).mean())),
          "logits_absmax": np.array(np.abs(r64).max()),
          "logits_rms": np.array(np.sqrt((r64 * r64).mean()))}
    np.savez_compressed(os.path.join(HERE, "unet_S572_margin.npz"), **fx)
    print("S=572 margins done: rms %.4g, |d| < 1: %d px, < 10: %d px of %d" % (fx["margin_rms"], (np.abs(d) < 1).sum(), (np.abs(d) < 10).sum(), d.size))


def main():
    network, functions = import_reference()
    torch.set_num_threads(8)
    meta = dict(torch=torch.__version__, threads=torch.get_num_threads(), seed_w=0)
    params = prng.make_params(seed=0)
    names = list(params.keys())

    # ---- G2: whole net fwd+bwd at S=188 (pad 0) and S=220 (pad > 0), B=2, fp32 and fp64
    for S, B in ((188, 2), (220, 2)):
        So = S - 184
        x = prng.make_input(1, B, S)
        dl = prng.make_cotangent(2, (B, 2, So, So))
        fx = {"meta": np.array(repr(dict(meta, S=S, B=B, seed_x=1, seed_dl=2)))}
        for tag, dt in (("f32", torch.float32), ("f64", torch.float64)):
            r = run_net(network, params, x, dl, dt)
            fx["logits_" + tag] = r["logits"]
            sums, samp_idx, samp = [], [], []
            for k in names:
                s, i, v = checksum(r["grad." + k])
                sums.append(s); samp_idx.append(i); samp.append(v)
            fx["grad_sums_" + tag] = np.stack(sums)          # [46,3] sum, l2, absmax
            fx["grad_samp_" + tag] = np.stack(samp)          # [46,16]
            fx["grad_samp_idx"] = np.stack(samp_idx)
            # a few small tensors in full
            for k in ("conv11c.weight", "conv11c.bias", "finalconv.weight", "finalconv.bias", "conv52c.bias", "upconv4.bias"):
                fx["grad_full_%s_%s" % (k, tag)] = r["grad." + k]
        fx["names"] = np.array(names)
        np.savez_compressed(os.path.join(HERE, "unet_S%d.npz" % S), **fx)
        print("S=%d done: logits absmax %.4g" % (S, np.abs(fx["logits_f64"]).max()))

    # ---- G3: S=572 forward, B=1: norms, strided samples, packed argmax, min margin
    S, B = 572, 1
    x = prng.make_input(1, B, S)
    r32 = run_net(network, params, x, None, torch.float32)["logits"]
    r64 = run_net(network, params, x, None, torch.float64)["logits"]
    margin = np.abs(r64[:, 0] - r64[:, 1])
    am = (r64[:, 1] > r64[:, 0])
    am32 = (r32[:, 1] > r32[:, 0])
    fx = {
        "meta": np.array(repr(dict(meta, S=S, B=B, seed_x=1))),
        "logits_sample_f64": r64[:, :, ::6, ::6].copy(),
        "logits_sample_f32": r32[:, :, ::6, ::6].copy(),
        "logits_norms_f64": np.array([np.abs(r64).max(), np.sqrt((r64 ** 2).sum()), r64.sum()]),
        "argmax_packed": np.packbits(am.ravel()),
        "argmax_shape": np.array(am.shape),
        "min_margin": np.array(margin.min()),
        # pixels whose |logit0-logit1| is within 5e-3 (1.6e-5 of |y|max): argmax there is
        # decided by fp32 rounding order, so bit-exactness is asserted on all OTHER pixels
        "low_margin_idx": np.flatnonzero(margin.ravel() < 5e-3).astype(np.int64),
        "low_margin_thr": np.array(5e-3),
        "f32_vs_f64_maxabs": np.array(np.abs(r32 - r64).max()),
        "argmax_f32_equals_f64": np.array(bool((am == am32).all())),
    }
    np.savez_compressed(os.path.join(HERE, "unet_S572_fwd.npz"), **fx)
    print("S=572 done: |y|max %.4g  min margin %.4g  f32-f64 maxabs %.3g  argmax equal %s"
          % (fx["logits_norms_f64"][0], margin.min(), fx["f32_vs_f64_maxabs"], fx["argmax_f32_equals_f64"]))

    # ---- G5: known answers of the host-side helpers (functions.py)
    ka = {}
    for orig in (196, 388, 512, 1024, 100, 20):
        ka["isc_%d" % orig] = np.array(functions.input_size_compute(torch.zeros(1, 1, orig, orig)))
    p = torch.tensor([[1, 1, 0], [0, 1, 0], [0, 0, 0]])
    l = torch.tensor([[1, 0, 0], [0, 1, 1], [0, 0, 0]])
    ka["evalm"] = functions.evaluation_metrics(p, l)
    ka["class_balance"] = functions.class_balance(l[None]).numpy()
    lab = torch.from_numpy(prng.make_labels(3, 2, 36)[:, 0])
    ka["class_balance_rand"] = functions.class_balance(lab).numpy()
    # crop_and_concat cases (Q2, Q7): pad (A smaller), crop (A larger), odd difference raises
    net = network.Unet.__new__(network.Unet)
    A = torch.arange(2 * 3 * 4 * 4, dtype=torch.float32).reshape(2, 3, 4, 4)
    Bt = torch.arange(2 * 2 * 8 * 8, dtype=torch.float32).reshape(2, 2, 8, 8)
    ka["cac_pad"] = network.Unet.crop_and_concat(net, A, Bt).numpy()
    ka["cac_crop"] = network.Unet.crop_and_concat(net, Bt, A).numpy()
    try:
        network.Unet.crop_and_concat(net, A, torch.zeros(2, 2, 7, 7))
        ka["cac_odd_raises"] = np.array(False)
    except RuntimeError:
        ka["cac_odd_raises"] = np.array(True)
    # L1: BCE-with-logits (weighted per Q4 for B=2, and unweighted), L2 argmax, on S=220 logits
    g = np.load(os.path.join(HERE, "unet_S220.npz"))
    lg = torch.from_numpy(g["logits_f64"])
    labels = torch.from_numpy(prng.make_labels(3, 2, 36))
    ll = torch.empty_like(lg)
    ll[:, 0] = 1 - labels[:, 0]; ll[:, 1] = labels[:, 0]
    wm = functions.class_balance(labels.squeeze(1)).double()
    lgr = lg.clone().requires_grad_(True)
    loss_w = torch.nn.BCEWithLogitsLoss(weight=wm)(lgr, ll)
    loss_w.backward()
    ka["bce_weighted_loss"] = np.array(loss_w.item()); ka["bce_weighted_grad"] = lgr.grad.numpy()
    lgr = lg.clone().requires_grad_(True)
    loss_u = torch.nn.BCEWithLogitsLoss()(lgr, ll)
    loss_u.backward()
    ka["bce_plain_loss"] = np.array(loss_u.item()); ka["bce_plain_grad"] = lgr.grad.numpy()
    ka["argmax_S220"] = lg.argmax(dim=1).numpy()
    for Bq in (3, 8):   # Q4: the reference's weight broadcast raises for B not in {1,2}
        try:
            torch.nn.BCEWithLogitsLoss(weight=torch.ones(Bq, 4, 4))(torch.zeros(Bq, 2, 4, 4), torch.zeros(Bq, 2, 4, 4))
            ka["bce_B%d_raises" % Bq] = np.array(False)
        except RuntimeError:
            ka["bce_B%d_raises" % Bq] = np.array(True)
    # init std (Q1) measured on the reference's own init
    torch.manual_seed(0)
    net = network.Unet()
    ka["init_std"] = np.array([p.std().item() for k, p in net.named_parameters() if k.endswith("weight")])
    ka["init_keys"] = np.array([k for k, _ in net.named_parameters()])
    ka["init_shapes"] = np.array([repr(tuple(p.shape)) for _, p in net.named_parameters()])
    ka["init_seed0_first"] = np.stack([p.detach().flatten()[:2].numpy() for _, p in net.named_parameters()])
    ka["meta"] = np.array(repr(meta))
    np.savez_compressed(os.path.join(HERE, "known_answers.npz"), **ka)
    print("known answers done")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "s572grad":
        main_s572_grad()
    elif len(sys.argv) > 1 and sys.argv[1] == "s572margin":
        main_s572_margin()
    else:
        main()
