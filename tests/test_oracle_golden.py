"""Pins the CPU oracle (oracle/unet_oracle.c and oracle/torch_ref.py) against golden
vectors produced by the imported reference (tests/golden/make_golden.py).  CPU only."""
import os

import numpy as np
import pytest
import torch

from oracle import oracle_c, prng, torch_ref


def nerr(a, ref):
    ref = np.asarray(ref, dtype=np.float64)
    return np.abs(np.asarray(a, dtype=np.float64) - ref).max() / max(np.abs(ref).max(), 1e-300)


def _checks(grads, g, tag, tol):
    names = [str(n) for n in g["names"]]
    sums = g["grad_sums_" + tag]; samp = g["grad_samp_" + tag]; idx = g["grad_samp_idx"]
    worst = 0.0
    for i, k in enumerate(names):
        a = np.asarray(grads[k], dtype=np.float64).ravel()
        l2 = np.sqrt((a * a).sum())
        amax = sums[i, 2]
        assert abs(l2 - sums[i, 1]) <= tol * max(sums[i, 1], 1e-30), (k, l2, sums[i, 1])
        e = np.abs(a[idx[i]] - samp[i]).max() / max(amax, 1e-30)
        worst = max(worst, e)
        assert e <= tol, (k, e)
    return worst


@pytest.mark.parametrize("S", [188, 220])
def test_c_oracle_f64_matches_reference(golden_dir, S):
    g = np.load(os.path.join(golden_dir, "unet_S%d.npz" % S))
    B, So = 2, S - 184
    p32 = prng.make_params(0)                         # float32 values, widened
    params = {k: p32[k].astype(np.float64) for k in p32}
    x = prng.make_input(1, B, S).astype(np.float64)
    dl = prng.make_cotangent(2, (B, 2, So, So)).astype(np.float64)
    logits, grads = oracle_c.unet_fwd_bwd(params, x, dlogits=dl)
    assert nerr(logits, g["logits_f64"]) < 1e-12
    _checks(grads, g, "f64", 1e-10)
    for k in ("conv11c.weight", "conv11c.bias", "finalconv.weight", "finalconv.bias", "conv52c.bias", "upconv4.bias"):
        assert nerr(grads[k], g["grad_full_%s_f64" % k]) < 1e-10, k


def test_c_oracle_f32_within_fp32_noise(golden_dir):
    S = 188
    g = np.load(os.path.join(golden_dir, "unet_S%d.npz" % S))
    B, So = 2, S - 184
    params = prng.make_params(0)
    x = prng.make_input(1, B, S)
    dl = prng.make_cotangent(2, (B, 2, So, So))
    logits, grads = oracle_c.unet_fwd_bwd(params, x, dlogits=dl)
    # judged against the f64 truth, normalised by tensor scale
    assert nerr(logits, g["logits_f64"]) < 2e-5
    _checks(grads, g, "f64", 5e-3)


def test_reference_f32_vs_f64_noise_floor(golden_dir):
    """Records the reference's own fp32-vs-fp64 distance: the bar the HIP path is held to."""
    for S in (188, 220):
        g = np.load(os.path.join(golden_dir, "unet_S%d.npz" % S))
        assert nerr(g["logits_f32"], g["logits_f64"]) < 1e-5
        rel = np.abs(g["grad_samp_f32"] - g["grad_samp_f64"]).max(axis=1) / g["grad_sums_f64"][:, 2]
        assert rel.max() < 5e-3


def test_torch_ref_matches_reference(golden_dir):
    S = 220
    g = np.load(os.path.join(golden_dir, "unet_S%d.npz" % S))
    p = torch_ref.params_to_torch(prng.make_params(0), torch.float64, requires_grad=True)
    x = torch.from_numpy(prng.make_input(1, 2, S)).double()
    y = torch_ref.unet_forward(p, x)
    assert nerr(y.detach().numpy(), g["logits_f64"]) < 1e-12
    y.backward(torch.from_numpy(prng.make_cotangent(2, (2, 2, 36, 36))).double())
    grads = {k: v.grad.numpy() for k, v in p.items()}
    _checks(grads, g, "f64", 1e-10)


def test_c_oracle_S572_forward_and_argmax(golden_dir):
    g = np.load(os.path.join(golden_dir, "unet_S572_fwd.npz"))
    params = prng.make_params(0)
    x = prng.make_input(1, 1, 572)
    logits, _ = oracle_c.unet_fwd_bwd(params, x)
    assert nerr(logits[:, :, ::6, ::6], g["logits_sample_f64"]) < 1e-5
    am = oracle_c.argmax2(logits).ravel()
    ref = np.unpackbits(g["argmax_packed"])[: am.size]
    safe = np.ones(am.size, bool); safe[g["low_margin_idx"]] = False
    assert (am[safe] == ref[safe]).all()          # bit-exact where the margin allows
    assert (am != ref).sum() <= g["low_margin_idx"].size


def test_known_answers(golden_dir):
    ka = np.load(os.path.join(golden_dir, "known_answers.npz"))
    for orig in (196, 388, 512, 1024, 100, 20):
        assert tuple(ka["isc_%d" % orig]) == oracle_c.input_size_compute(orig)
    # crop_and_concat: zero-pad (Q2), crop, odd difference raises (Q7)
    A = np.arange(2 * 3 * 4 * 4, dtype=np.float32).reshape(2, 3, 4, 4)
    Bt = np.arange(2 * 2 * 8 * 8, dtype=np.float32).reshape(2, 2, 8, 8)
    assert np.array_equal(oracle_c.crop_and_concat(A, Bt), ka["cac_pad"])
    assert np.array_equal(oracle_c.crop_and_concat(Bt, A), ka["cac_crop"])
    assert bool(ka["cac_odd_raises"])
    with pytest.raises(RuntimeError):
        oracle_c.crop_and_concat(A, np.zeros((2, 2, 7, 7), np.float32))
    # L1 / L2
    g = np.load(os.path.join(golden_dir, "unet_S220.npz"))
    lg = g["logits_f64"]
    labels = prng.make_labels(3, 2, 36)
    ll = np.empty_like(lg); ll[:, 0] = 1 - labels[:, 0]; ll[:, 1] = labels[:, 0]
    loss, dx = oracle_c.bce_logits(lg, ll)
    assert abs(loss - ka["bce_plain_loss"]) < 1e-12 * abs(ka["bce_plain_loss"])
    assert nerr(dx, ka["bce_plain_grad"]) < 1e-12
    w = ka["class_balance_rand"].astype(np.float64)          # [B,H,W] broadcast against [B,2,H,W] (Q4)
    loss, dx = oracle_c.bce_logits(lg, ll, w)
    assert abs(loss - ka["bce_weighted_loss"]) < 1e-12 * abs(ka["bce_weighted_loss"])
    assert nerr(dx, ka["bce_weighted_grad"]) < 1e-12
    assert np.array_equal(oracle_c.argmax2(lg), ka["argmax_S220"])
    assert bool(ka["bce_B3_raises"]) and bool(ka["bce_B8_raises"])
    # init std formulas (Q1) against the reference's measured init
    stds = np.array([t[5] for t in prng.layer_table()])
    assert np.allclose(stds, ka["init_std"], rtol=0.1)   # sqrt(2/N) would be 41% off
    assert [str(k) for k in ka["init_keys"]] == list(prng.make_params(0, base=4).keys())
    assert [str(s) for s in ka["init_shapes"]] == [repr(tuple(v.shape)) for v in prng.make_params(0).values()]


def test_sgd_oracle_matches_torch():
    rng = np.random.default_rng(0)
    p = rng.standard_normal(1000).astype(np.float32); g1 = rng.standard_normal(1000).astype(np.float32)
    g2 = rng.standard_normal(1000).astype(np.float32)
    tp = torch.nn.Parameter(torch.from_numpy(p.copy()))
    opt = torch.optim.SGD([tp], lr=1e-4, momentum=0.99)
    buf = np.zeros_like(p); q = p.copy()
    for i, gg in enumerate((g1, g2)):
        tp.grad = torch.from_numpy(gg.copy()); opt.step()
        oracle_c.sgd_momentum(q, gg, buf, 1e-4, 0.99, i == 0)
    assert np.allclose(q, tp.detach().numpy(), rtol=0, atol=1e-7)


def test_torch_restatement_f64_matches_reference_at_the_baseline_tile(golden_dir):
    """S=572, B=1 (BASELINE tile): the torch restatement in fp64 reproduces the reference's fp64 logits samples and all 46
    gradients (checksums + 64 samples per tensor, tests/golden/unet_S572_grad.npz) — the fixture the GPU accounting test at
    572 stands on."""
    g = np.load(os.path.join(golden_dir, "unet_S572_grad.npz"))
    S, B = 572, 1
    torch.set_num_threads(8)
    p = torch_ref.params_to_torch(prng.make_params(0), torch.float64, requires_grad=True)
    y = torch_ref.unet_forward(p, torch.from_numpy(prng.make_input(1, B, S)).double())
    y.backward(torch.from_numpy(prng.make_cotangent(2, (B, 2, S - 184, S - 184))).double())
    assert nerr(y.detach().numpy()[:, :, ::6, ::6], g["logits_sample_f64"]) < 1e-12
    _checks({k: v.grad.numpy() for k, v in p.items()}, g, "f64", 1e-10)


def test_bf16_error_model_against_emulated_storage_roundings(golden_dir):
    """oracle/parity.py's bf16 error model (sigma_rel = 2^-8 sqrt(43/3), the bound tests/test_bf16_gpu.py holds the HIP path's
    bf16-tensor mode to at S=572) checked on the CPU against an independent emulation: the fp64 torch restatement with every
    stored activation and every bf16 filter copy rounded to bf16 (round-to-nearest-even), at S=220 where it takes seconds.
    The model must be an upper estimate (emulated rms error <= model) without being idle (>= 1/4 of it), and the largest
    emulated error must respect the Gaussian-tail bound; the margin fixture of the S=572 test must agree with the argmax golden."""
    from oracle import parity
    S, B = 220, 2
    torch.set_num_threads(8)
    p = torch_ref.params_to_torch(prng.make_params(0), torch.float64)
    x = torch.from_numpy(prng.make_input(1, B, S)).double()
    rb = lambda v: v.float().to(torch.bfloat16).double()       # one RNE rounding to 8 significant bits
    with torch.no_grad():
        ref = torch_ref.unet_forward(p, x).numpy()
        emu = torch_ref.unet_forward(p, x, store=rb, wstore=rb).numpy()
    g = np.load(os.path.join(golden_dir, "unet_S220.npz"))
    assert nerr(ref, g["logits_f64"]) < 1e-12                    # the un-rounded run is the reference's
    rms = float(np.sqrt((ref * ref).mean()))
    e_rms = float(np.sqrt(((emu - ref) ** 2).mean())) / rms
    e_max = float(np.abs(emu - ref).max())
    sig = parity.bf16_sigma_rel()
    print("bf16 storage emulation at S=%d: rms error %.3e of the logits' rms (model %.3e), max %.3g (bound %.3g)"
          % (S, e_rms, sig, e_max, parity.bf16_max_err(ref.size, rms)))
    assert 0.25 * sig <= e_rms <= sig
    assert e_max <= parity.bf16_max_err(ref.size, rms)
    m = np.load(os.path.join(golden_dir, "unet_S572_margin.npz"))
    f = np.load(os.path.join(golden_dir, "unet_S572_fwd.npz"))
    d = m["margin_f64_as_f32"]
    assert np.array_equal(np.packbits((d > 0).ravel()), f["argmax_packed"])
    assert abs(float(np.abs(d).min()) - float(f["min_margin"])) < 1e-6 * float(f["min_margin"]) + 1e-7
    assert abs(float(m["logits_absmax"]) - float(f["logits_norms_f64"][0])) < 1e-9
