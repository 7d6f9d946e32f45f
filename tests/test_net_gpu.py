"""Whole-path parity of the HIP U-Net (through network.Unet -> C ABI) against the golden vectors
captured from the imported reference, the C oracle and the torch restatement.

Tolerances (north_star "within 1e-3 rel fp32"): errors are normalised by tensor scale,
|d|_inf/|ref|_inf, and judged against fp64.
  forward logits <= 2e-5 (measured 3-5e-6; the reference's own fp32 CPU run sits at ~1e-6).
  gradients, SAME branch (same ReLU masks / pool winners as the HIP forward) <= 3e-4 per tensor: the
      rigorous check of every backward kernel (oracle/parity.py explains why the branch is pinned).
  gradients, free-running, against the reference's fp64 goldens: any fp32 evaluation, the reference's own
      included, lands on a different ReLU/pool piece for a few near-zero activations, and one such element moves
      whole upstream tensors by up to ~1e-2.  So the free-running check is an ACCOUNTING, not a loose bound
      (test_free_running_gradients_are_explained_by_legitimate_branch_choices): (a) every element where the HIP
      branch differs from the fp64 branch missed the decision boundary by no more than the forward rounding error,
      (b) where no decision differs, every sampled gradient element agrees with the reference's golden within the
      same-branch tolerance; where some do, the distance to the golden is reported, not asserted (a bound built from the
      branch effect would follow from the same-branch check by the triangle inequality): (a) carries the argument.
      The accounting runs at S = 188, 220 and at the BASELINE tile size 572 (tests/golden/unet_S572_grad.npz holds the
      reference's fp32 and fp64 gradients there); no loose free-running bound is left.
argmax masks: bit-exact on every pixel whose fp64 margin exceeds the recorded threshold."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

FWD_TOL = 2e-5
GRAD_TOL = 3e-4          # same-branch
TRAINER_SAME_BRANCH_TOL = 1e-5   # relative: 4 SGD steps, HIP (fp32) against the fp64 oracle following the same ReLU/pool branch
TRAINER_LOSS_FLOOR = 2e-5    # relative: one forward rounding (FWD_TOL) on top of twice the reference's own fp32 distance
TRAINER_PIXEL_FLOOR = 1.0    # pixels of the mask on top of twice the reference's own fp32 distance (IoU / pixel-error series)


@pytest.fixture(scope="module")
def net():
    import network
    from oracle import prng
    assert torch.cuda.is_available()
    torch.cuda.set_device(0)
    m = network.Unet()
    sd = {k: torch.from_numpy(v) for k, v in prng.make_params(0).items()}
    m.load_state_dict(sd)
    return m.to("cuda:0")


def nerr(a, ref):
    a = np.asarray(a, dtype=np.float64); ref = np.asarray(ref, dtype=np.float64)
    return np.abs(a - ref).max() / max(np.abs(ref).max(), 1e-300)


def run(net, S, B, with_grad, seed_x=1):
    from oracle import prng
    x = torch.from_numpy(prng.make_input(seed_x, B, S)).cuda()
    net.zero_grad(set_to_none=True)
    if not with_grad:
        with torch.no_grad():
            return net(x).cpu().numpy(), None
    y = net(x)
    dl = torch.from_numpy(prng.make_cotangent(2, (B, 2, S - 184, S - 184))).cuda()
    y.backward(dl)
    grads = {k: p.grad.detach().cpu().numpy() for k, p in net.named_parameters()}
    return y.detach().cpu().numpy(), grads


@pytest.mark.parametrize("S", [188, 220])
def test_forward_backward_vs_reference_golden(net, golden_dir, S):
    g = np.load(os.path.join(golden_dir, "unet_S%d.npz" % S))
    logits, grads = run(net, S, 2, True)
    assert nerr(logits, g["logits_f64"]) < FWD_TOL
    names = [str(n) for n in g["names"]]
    assert set(names) == set(grads)
    # the head's gradients do not pass through any ReLU/pool decision: tight against the reference
    assert nerr(grads["finalconv.weight"], g["grad_full_finalconv.weight_f64"]) < 2e-5
    assert nerr(grads["finalconv.bias"], g["grad_full_finalconv.bias_f64"]) < 2e-5
    # inference path (no stash) gives the same logits bit for bit
    logits_ng, _ = run(net, S, 2, False)
    assert np.array_equal(logits_ng, logits)


@pytest.mark.parametrize("S,B", [(188, 2), (220, 2), (252, 1), (380, 1)])
def test_every_gradient_element_on_same_branch_vs_c_oracle_f64(net, S, B):
    """All 31,030,658 gradient elements (and the logits) through the C ABI against the fp64 C oracle
    evaluated on the HIP forward's own ReLU masks / pool winners.  188/220/252 take the crop branch of
    crop_and_concat at some levels, 220 zero-pads level 4, 380 zero-pads every level (like 572)."""
    from oracle import parity
    r = parity.check_same_branch(S, B)
    assert r["fwd"] < FWD_TOL, r["fwd"]
    worst = max(r["grads"].items(), key=lambda kv: kv[1])
    assert worst[1] < GRAD_TOL, worst


@pytest.mark.parametrize("S,B,fixture", [(188, 2, "unet_S188.npz"), (220, 2, "unet_S220.npz"), (572, 1, "unet_S572_grad.npz")])
def test_free_running_gradients_are_explained_by_legitimate_branch_choices(golden_dir, S, B, fixture):
    """Free-running HIP gradients against the reference's fp64 goldens (see the module docstring).  S=572 is the BASELINE
    tile: the fp64 C oracle on the HIP branch and the fp64 torch forward for the margins take ~2 min of host time there."""
    from oracle import parity
    g = np.load(os.path.join(golden_dir, fixture))
    r = parity.check_same_branch(S, B)
    assert r["fwd"] < FWD_TOL and max(r["grads"].values()) < GRAD_TOL                # same-branch parity of this very run
    # (a) the HIP forward only leaves the fp64 branch where fp64 itself is within rounding of the boundary
    n_relu, n_pool, worst = parity.branch_disagreements(r["masks"], r["sels"], S, B)
    print("S=%d: %d ReLU and %d pool decisions differ from fp64; largest fp64 margin among them %.2e of the layer scale" % (S, n_relu, n_pool, worst))
    assert worst < FWD_TOL, (n_relu, n_pool, worst)
    # (b) against the reference's fp64 golden itself.  Where no decision differs (S = 188) the two runs are on the same branch and
    # every sampled gradient element must agree with the GOLDEN within the same-branch tolerance - an independent check, the C
    # oracle is not involved.  Where decisions differ, the distance to the golden is whatever those legitimate flips do to the
    # gradient; it is printed next to the reference's own fp32-vs-fp64 distance and the fp64 effect of the HIP branch (evaluated by
    # the C oracle), but not asserted: an assertion built from that branch effect would follow from the same-branch check above
    # by the triangle inequality.  (a) carries the argument there.
    names = [str(n) for n in g["names"]]
    s64, samp64, samp32, idx = g["grad_sums_f64"], g["grad_samp_f64"], g["grad_samp_f32"], g["grad_samp_idx"]
    table = []
    for i, k in enumerate(names):
        hip = r["hip_grads"][k].astype(np.float64).ravel()[idx[i]]
        onb = r["ref_grads"][k].astype(np.float64).ravel()[idx[i]]
        e_hip = np.abs(hip - samp64[i]).max() / s64[i, 2]
        e_branch = np.abs(onb - samp64[i]).max() / s64[i, 2]
        e_ref = np.abs(samp32[i] - samp64[i]).max() / s64[i, 2]
        table.append((k, e_hip, e_ref, e_branch))
        if n_relu + n_pool == 0:
            assert e_hip < GRAD_TOL, (k, e_hip)
    worst_t = max(table, key=lambda t: t[1])
    print("S=%d: worst free-running tensor %s: HIP %.2e from the golden, the reference's own fp32 run %.2e, fp64 effect of the HIP branch %.2e" % ((S,) + worst_t))


def test_S572_forward_and_bit_exact_argmax(net, golden_dir):
    import optim as hip_optim
    g = np.load(os.path.join(golden_dir, "unet_S572_fwd.npz"))
    from oracle import prng
    x = torch.from_numpy(prng.make_input(1, 1, 572)).cuda()
    with torch.no_grad():
        y = net(x)
    assert nerr(y.cpu().numpy()[:, :, ::6, ::6], g["logits_sample_f64"]) < FWD_TOL
    am = hip_optim.argmax2(y).cpu().numpy().ravel()
    ref = np.unpackbits(g["argmax_packed"])[: am.size]
    safe = np.ones(am.size, bool); safe[g["low_margin_idx"]] = False
    assert (am[safe] == ref[safe]).all()
    assert (am != ref).sum() <= g["low_margin_idx"].size


def test_S572_batch8_properties(net):
    """BASELINE config #2 size (B=8, 572): batch independence and additivity of gradients over the
    batch.  Which tiles fall into a split-K tail round depends on the batch size, so the summation order
    of those tiles (not the result beyond fp32 rounding) differs between B=8 and B=1."""
    from oracle import prng
    S, B = 572, 8
    x = torch.from_numpy(prng.make_input(5, B, S)).cuda()
    dl = torch.from_numpy(prng.make_cotangent(6, (B, 2, 388, 388))).cuda()
    net.zero_grad(set_to_none=True)
    y = net(x)
    y.backward(dl)
    g_all = [p.grad.clone() for p in net.parameters()]
    with torch.no_grad():
        y0 = net(x[0:1].contiguous()); y7 = net(x[7:8].contiguous())
    scale = y.abs().max()
    assert ((y0[0] - y[0]).abs().max() / scale).item() < FWD_TOL and ((y7[0] - y[7]).abs().max() / scale).item() < FWD_TOL
    acc = None
    for lo, hi in ((0, 3), (3, 8)):                 # ragged split
        net.zero_grad(set_to_none=True)
        net(x[lo:hi].contiguous()).backward(dl[lo:hi].contiguous())
        part = [p.grad.clone() for p in net.parameters()]
        acc = part if acc is None else [a + b for a, b in zip(acc, part)]
    for a, b in zip(acc, g_all):
        assert ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item() < GRAD_TOL      # same forward values, sums in another order


def test_full_size_vs_torch_restatement(net):
    """Full-size (S=572) forward against the torch restatement run on the host CPU in fp32, and the gradients that pass
    through no ReLU/pool decision (the head's) against the reference's fp64 golden.  All other gradients at this size:
    test_free_running_gradients_are_explained_by_legitimate_branch_choices[572] and the same-branch test."""
    from oracle import prng, torch_ref
    S, B = 572, 1
    logits, grads = run(net, S, B, True)
    p = torch_ref.params_to_torch(prng.make_params(0), torch.float32)
    with torch.no_grad():
        y = torch_ref.unet_forward(p, torch.from_numpy(prng.make_input(1, B, S)))
    assert nerr(logits, y.numpy()) < FWD_TOL
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "unet_S572_grad.npz"))
    assert nerr(logits[:, :, ::6, ::6], g["logits_sample_f64"]) < FWD_TOL
    assert nerr(grads["finalconv.weight"], g["grad_full_finalconv.weight_f64"]) < 2e-5
    assert nerr(grads["finalconv.bias"], g["grad_full_finalconv.bias_f64"]) < 2e-5


def test_gradient_with_respect_to_the_input_image(net):
    """Unet.forward of an input that requires grad (the reference supports it through autograd; its trainer / tester never ask):
    conv11c's dgrad (unet_backward_input) against the fp64 torch restatement.  S = 188, where the HIP forward takes the fp64
    branch at every ReLU / pool (test_free_running_gradients...: 0 decisions differ), so the two gradients are comparable
    directly; the parameter gradients of the same backward must not change."""
    from oracle import prng, torch_ref
    S, B = 188, 2
    x = torch.from_numpy(prng.make_input(1, B, S))
    dl = torch.from_numpy(prng.make_cotangent(2, (B, 2, S - 184, S - 184)))
    p64 = torch_ref.params_to_torch(prng.make_params(0), torch.float64)
    x64 = x.double().requires_grad_(True)
    torch_ref.unet_forward(p64, x64).backward(dl.double())
    ref = x64.grad.numpy()
    net.zero_grad(set_to_none=True)
    y0 = net(x.cuda())
    y0.backward(dl.cuda())
    g0 = [p.grad.clone() for p in net.parameters()]
    net.zero_grad(set_to_none=True)
    xg = x.cuda().requires_grad_(True)
    y = net(xg)
    y.backward(dl.cuda())
    assert xg.grad is not None and xg.grad.shape == xg.shape
    assert nerr(xg.grad.cpu().numpy(), ref) < GRAD_TOL, nerr(xg.grad.cpu().numpy(), ref)
    assert torch.equal(y, y0) and all(torch.equal(a, p.grad) for a, p in zip(g0, net.parameters()))
    # frozen parameters, only the image asks for a gradient
    for p_ in net.parameters():
        p_.requires_grad_(False)
    try:
        xg2 = x.cuda().requires_grad_(True)
        net(xg2).backward(dl.cuda())
        assert torch.equal(xg2.grad, xg.grad)
    finally:
        for p_ in net.parameters():
            p_.requires_grad_(True)


def test_bad_sizes_raise_like_the_reference(net):
    # odd size difference in crop_and_concat -> torch.cat raises in the reference (Q7)
    for S in (570, 200, 187, 60):
        with pytest.raises(RuntimeError):
            net(torch.zeros(1, 1, S, S, device="cuda"))
    with pytest.raises(RuntimeError):
        net(torch.zeros(1, 2, 188, 188, device="cuda"))
    with pytest.raises(RuntimeError):
        net(torch.zeros(1, 1, 188, 188))            # host tensor: no CPU fallback


def test_odd_batch_and_state_dict_roundtrip(net, tmp_path):
    import network
    from oracle import prng
    x = torch.from_numpy(prng.make_input(9, 3, 188)).cuda()
    with torch.no_grad():
        y3 = net(x)
    path = os.path.join(tmp_path, "w.pth")
    torch.save(net.state_dict(), path)
    m2 = network.Unet()
    m2.load_state_dict(torch.load(path, weights_only=True))
    m2 = m2.to("cuda:0")
    with torch.no_grad():
        assert torch.equal(m2(x), y3)


def test_training_and_testing_loops(net, tmp_path, capsys):
    """trainer.training / tester.testing keep the reference's call surface and artefacts
    (progress/*.out, models/*.pth, images|labels|preds/*.tif, test_iou.out, test_pe.out)."""
    import copy
    from oracle import prng
    from trainer import training
    from tester import testing
    m = copy.deepcopy(net)
    S, So = 188, 4

    def loader(seed, n):
        out = []
        for i in range(n):
            out.append((torch.from_numpy(prng.make_input(seed + i, 2, S)), torch.from_numpy(prng.make_labels(seed + i, 2, So))))
        return out

    before = [p.detach().clone() for p in m.parameters()]
    training(m, loader(10, 2), loader(20, 1), 1, 2, torch.device("cuda:0"), str(tmp_path), "ISBI2012")
    for f in ("train_eval_iou", "train_eval_pe", "val_eval_iou", "val_eval_pe", "loss", "loss_val"):
        assert np.loadtxt(os.path.join(tmp_path, "progress", f + ".out")).size == 2
    assert os.path.exists(os.path.join(tmp_path, "models", "unet_weight_save_best.pth"))
    assert any(not torch.equal(a, b.detach()) for a, b in zip(before, m.parameters()))
    tl = [(torch.from_numpy(prng.make_input(30, 1, S)), torch.from_numpy(prng.make_labels(30, 1, So)))]
    out = os.path.join(tmp_path, "test_out")
    testing(m, tl, 1, torch.device("cuda:0"), out)
    for sub, f in (("images", "image0.tif"), ("labels", "label0.tif"), ("preds", "pred0.tif")):
        assert os.path.exists(os.path.join(out, sub, f))
    assert np.loadtxt(os.path.join(out, "test_iou.out")).shape == (2,)


def test_trainer_loss_matches_reference_bce(net, golden_dir):
    """The trainer's loss on golden logits reproduces the reference's weighted BCE (Q4 broadcast)."""
    import optim as hip_optim
    from oracle import prng
    ka = np.load(os.path.join(golden_dir, "known_answers.npz"))
    g = np.load(os.path.join(golden_dir, "unet_S220.npz"))
    lg = torch.from_numpy(g["logits_f64"]).float().cuda().requires_grad_(True)
    labels = torch.from_numpy(prng.make_labels(3, 2, 36))
    ll = hip_optim.onehot2(labels, lg)
    wm = torch.from_numpy(ka["class_balance_rand"]).float().cuda()
    loss = hip_optim.bce_with_logits(lg, ll, weight=wm)
    loss.backward()
    assert abs(loss.item() - float(ka["bce_weighted_loss"])) < 1e-5 * float(ka["bce_weighted_loss"])
    assert nerr(lg.grad.cpu().numpy(), ka["bce_weighted_grad"]) < 1e-5
    with pytest.raises(RuntimeError):                # Q4: B=3 cannot broadcast, like the reference
        hip_optim.bce_with_logits(torch.zeros(3, 2, 4, 4, device="cuda"), torch.zeros(3, 2, 4, 4, device="cuda"),
                                  weight=torch.ones(3, 4, 4, device="cuda"))


_DP_GPU_WORKER = r'''
import os, sys
import numpy as np, torch, torch.distributed as dist
root = sys.argv[1]
sys.path.insert(0, os.path.join(root, "dl-unet_amd")); sys.path.insert(0, root)
import network
from oracle import prng
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dist.init_process_group("gloo", rank=rank, world_size=world)       # both ranks share the one GPU of the box
S, Bg = 188, 4
net = network.Unet()
net.load_state_dict({k: torch.from_numpy(v) for k, v in prng.make_params(0).items()})
net = net.to("cuda:0").enable_data_parallel(backend="torch")     # RCCL wants one GPU per rank: gloo carries the buckets here
x = torch.from_numpy(prng.make_input(1, Bg, S)).cuda()
dl = torch.from_numpy(prng.make_cotangent(2, (Bg, 2, 4, 4))).cuda()
per = Bg // world
net(x[rank * per:(rank + 1) * per].contiguous()).backward(dl[rank * per:(rank + 1) * per].contiguous())
torch.cuda.synchronize()
grads = {k: p.grad.detach().cpu().numpy() for k, p in net.named_parameters()}
if rank == 0:
    ref = network.Unet()
    ref.load_state_dict({k: torch.from_numpy(v) for k, v in prng.make_params(0).items()})
    ref = ref.to("cuda:0")
    ref(x).backward(dl / world)               # single process, global batch, mean over ranks' shards
    worst = 0.0
    for k, p in ref.named_parameters():
        g = p.grad.cpu().numpy()
        worst = max(worst, float(np.abs(grads[k] - g).max() / max(np.abs(g).max(), 1e-30)))
    print("WORST", worst)
    assert worst < 1e-5, worst
dist.barrier()
dist.destroy_process_group()
'''


def test_data_parallel_module_path_two_ranks_one_gpu(tmp_path):
    """The module's DP backward (pre-scaled dlogits, per-stage bucketed async all-reduce, stream sync)
    with 2 processes sharing this box's GPU over gloo == the single-process global-batch gradient.
    (RCCL needs one GPU per rank: test_rccl_* below runs the library's own communicator with one rank, the
    driver's N>1 runs exercise it across GPUs.)"""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = os.path.join(tmp_path, "dp_gpu_worker.py")
    with open(script, "w") as f:
        f.write(_DP_GPU_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", WORLD_SIZE="2", HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, script, root], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=600)[0] for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o[-3000:]
    assert "WORST" in outs[0]


def test_config4_shaped_training_loop_with_gpu_augmentation(tmp_path, golden_dir):
    """BASELINE configs[3]: 512x512-shaped samples -> data.augment on the device (data.py:97-135: reflect pad + 30-degree-step
    cubic-spline rotation + centre crop to the 700 input, elastic deformation alpha=200 sigma=10 shared by image and mask,
    label crop + threshold, normalisation) -> trainer.training() for one epoch at the reference's batch size 2 -> IoU / pixel
    error files.  The two training samples are checked against tests/golden/augment_golden_S700.npz (the reference's own
    statements executed on the same inputs, angles 330 and 0 as it drew them): image within 2.5 grey levels of 255 and more than
    0.6 of a level off on fewer than 2 % of the pixels (scipy rounds the rotated and the warped uint8 image at t + 0.5: a value
    within fp32 noise of k + 0.5 may land on the neighbouring level, twice), mask differing on fewer than 0.5 % of the pixels."""
    import data
    import network
    from oracle import aux_ref
    from trainer import training
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    n, S = 512, 700
    g = np.load(os.path.join(golden_dir, "augment_golden_S700.npz"))

    def sample(seed, deg, eseed):
        img, tgt = aux_ref.cells(seed, n)
        return data.augment(torch.from_numpy(img.astype(np.float32)).to(dev), torch.from_numpy(tgt.astype(np.float32)).to(dev),
                            (0, 0), n, deg, 200.0, 10.0, random_state=np.random.RandomState(eseed))

    train = []
    for tag in ("a", "b"):
        crop, seed, deg, S_ref, eseed, _ = [int(v) for v in g["%s_params" % tag]]
        assert crop == n and S_ref == S
        inp, gt = sample(seed, deg, eseed)
        assert inp.shape == (1, S, S) and gt.shape == (1, n, n) and gt.dtype == torch.int64
        d_img = np.abs(inp[0].cpu().numpy()[::5, ::5] - g["%s_inp_sample" % tag])
        e_img = d_img.max()
        assert (d_img > 0.6 / 255).mean() < 2e-2, (tag, (d_img > 0.6 / 255).mean())
        ref_gt = g["%s_gt_sample" % tag] > 127
        mism = (gt[0].cpu().numpy()[::4, ::4].astype(bool) != ref_gt).mean()
        print("config 4 sample %s (rotation %d deg): image err %.2e, mask mismatch %.2e" % (tag, deg, e_img, mism))
        assert e_img < 2.5 / 255 and mism < 5e-3, (tag, e_img, mism)
        train.append((inp, gt))
    val = [sample(3, 90.0, 77), sample(4, 210.0, 78)]

    def batch(pairs):
        return torch.stack([p[0] for p in pairs]).contiguous(), torch.stack([p[1] for p in pairs]).contiguous()

    net = network.Unet().to(dev)
    training(net, [batch(train)], [batch(val)], 0, 2, dev, str(tmp_path), "DIC-C2DH-HeLa")
    iou = np.loadtxt(os.path.join(tmp_path, "progress", "val_eval_iou.out"))
    pe = np.loadtxt(os.path.join(tmp_path, "progress", "val_eval_pe.out"))
    assert 0.0 <= float(iou) <= 1.0 and 0.0 <= float(pe) <= 1.0
    assert np.isfinite(np.loadtxt(os.path.join(tmp_path, "progress", "loss.out")))


@pytest.fixture
def math_mode():
    import _hip
    L = _hip.lib()
    default = L.unet_get_math()
    yield lambda m: _hip.check(L.unet_set_math(m), "unet_set_math")
    _hip.check(L.unet_set_math(default), "unet_set_math")


def test_bf16x3_mode_keeps_fp32_class_accuracy(math_mode, golden_dir):
    """unet_set_math(1): fp32 operands split into two bf16 terms, three bf16 MFMAs per product, fp32 accumulation.
    Measured: logits 3e-5, same-branch gradients 6e-5 of tensor scale (fp32 MFMA: 4e-6 / 7e-6) — bounds below are
    ~5x that and far inside the path's 1e-3."""
    from oracle import parity
    math_mode(1)
    for S, B in ((220, 2), (380, 1)):
        r = parity.check_same_branch(S, B)
        assert r["fwd"] < 2e-4, r["fwd"]
        worst = max(r["grads"].items(), key=lambda kv: kv[1])
        assert worst[1] < 5e-4, worst
    # S=572 argmax against the reference golden on every pixel whose fp64 margin exceeds 10x the logit error bound
    import network, optim as hip_optim
    from oracle import prng
    g = np.load(os.path.join(golden_dir, "unet_S572_fwd.npz"))
    net = network.Unet()
    net.load_state_dict({k: torch.from_numpy(v) for k, v in prng.make_params(0).items()})
    net = net.to("cuda:0")
    with torch.no_grad():
        y = net(torch.from_numpy(prng.make_input(1, 1, 572)).cuda())
    assert nerr(y.cpu().numpy()[:, :, ::6, ::6], g["logits_sample_f64"]) < 2e-4
    am = hip_optim.argmax2(y).cpu().numpy().ravel()
    ref = np.unpackbits(g["argmax_packed"])[: am.size]
    yc = y.cpu().numpy()
    margin = np.abs(yc[0, 0] - yc[0, 1]).ravel()
    safe = margin > 2e-3 * float(g["logits_norms_f64"][0])            # 10 x (2e-4 x |y|max)
    assert safe.mean() > 0.99 and (am[safe] == ref[safe]).all()


def test_winograd_mode_is_fp32_parity_grade(math_mode, golden_dir):
    """unet_set_math(3): the stride-1 3x3 layers (forward and dgrad) run as fp32 Winograd F(2x2,3x3) on the fp32 MFMA;
    everything else as in mode 0.  Same tolerances as the direct fp32 path: forward 2e-5, same-branch gradients 3e-4,
    argmax bit-exact on every pixel outside the golden's low-margin set."""
    from oracle import parity
    math_mode(3)
    for S, B in ((188, 2), (220, 2), (380, 1)):
        r = parity.check_same_branch(S, B)
        worst = max(r["grads"].items(), key=lambda kv: kv[1])
        print("winograd S=%d: fwd %.2e, worst grad %s %.2e" % (S, r["fwd"], worst[0], worst[1]))
        assert r["fwd"] < FWD_TOL, r["fwd"]
        assert worst[1] < GRAD_TOL, worst
    import network, optim as hip_optim
    from oracle import prng
    g = np.load(os.path.join(golden_dir, "unet_S572_fwd.npz"))
    net = network.Unet()
    net.load_state_dict({k: torch.from_numpy(v) for k, v in prng.make_params(0).items()})
    net = net.to("cuda:0")
    with torch.no_grad():
        y = net(torch.from_numpy(prng.make_input(1, 1, 572)).cuda())
    assert nerr(y.cpu().numpy()[:, :, ::6, ::6], g["logits_sample_f64"]) < FWD_TOL
    am = hip_optim.argmax2(y).cpu().numpy().ravel()
    ref = np.unpackbits(g["argmax_packed"])[: am.size]
    safe = np.ones(am.size, bool); safe[g["low_margin_idx"]] = False
    assert (am[safe] == ref[safe]).all()


def test_direct_fp32_mode(math_mode):
    """unet_set_math(0): every layer as the direct correlation (exact fmaf chains on the fp32 MFMA)."""
    from oracle import parity
    math_mode(0)
    r = parity.check_same_branch(220, 2)
    assert r["fwd"] < FWD_TOL, r["fwd"]
    assert max(r["grads"].values()) < GRAD_TOL


def test_bf16_compute_mode(math_mode):
    """unet_set_math(2): bf16 tensors in HBM (activations, their gradients, packed filters), bf16 MFMA, fp32 accumulation;
    parameters, their gradients and the logits stay fp32 (BASELINE configs[2]'s arithmetic).  Same-branch check against the fp64
    oracle at S = 220; the bounds are oracle/parity.py's storage-rounding model (sigma_rel = 2^-8 sqrt(43/3) of a tensor's rms,
    Gaussian tail over its elements) expressed against the tensor maximum, rms <= max - the model itself is checked against
    an emulation by tests/test_oracle_golden.py and held at the BASELINE tile by tests/test_bf16_gpu.py."""
    from oracle import parity
    math_mode(2)
    r = parity.check_same_branch(220, 2)
    assert r["fwd"] < 5e-2, r["fwd"]
    assert max(r["grads"].values()) < 6e-2


def test_bf16_tensors_overlap_default_is_bit_identical_to_one_stream(math_mode):
    """With bf16 tensors the weight gradients run on the handle's auxiliary stream by default (unet_set_overlap(-1): per
    arithmetic mode).  Same kernels on the same data: logits and all 46 gradients must be bit-identical to the one-stream
    order, run after run."""
    import _hip
    import network
    from oracle import prng
    L = _hip.lib()
    math_mode(2)
    m = network.Unet()
    m.load_state_dict({k: torch.from_numpy(v) for k, v in prng.make_params(0).items()})
    m = m.to("cuda:0")
    x = torch.from_numpy(prng.make_input(1, 2, 380)).cuda()
    dl = torch.from_numpy(prng.make_cotangent(2, (2, 2, 196, 196))).cuda()

    def once():
        m.zero_grad(set_to_none=True)
        y = m(x)
        y.backward(dl)
        torch.cuda.synchronize()
        return y.detach().clone(), [p.grad.clone() for p in m.parameters()]

    try:
        _hip.check(L.unet_set_overlap(0), "unet_set_overlap")
        y0, g0 = once()
        _hip.check(L.unet_set_overlap(-1), "unet_set_overlap")
        for _ in range(3):
            y1, g1 = once()
            assert torch.equal(y0, y1) and all(torch.equal(a, b) for a, b in zip(g0, g1))
    finally:
        _hip.check(L.unet_set_overlap(-1), "unet_set_overlap")


@pytest.mark.parametrize("B,S", [(1, 188), (3, 252), (2, 700), (5, 380), (1, 1212), (16, 572)])
def test_shape_sweep_forward_backward(net, B, S):
    """Sizes of the BASELINE configs and ragged batches: finite results, gradients of a batch equal the sum of the
    per-sample gradients (first and last sample re-run alone), no size-dependent failure in the tilings."""
    from oracle import prng
    x = torch.from_numpy(prng.make_input(40 + B, B, S)).cuda()
    dl = torch.from_numpy(prng.make_cotangent(41, (B, 2, S - 184, S - 184))).cuda()
    net.zero_grad(set_to_none=True)
    y = net(x)
    assert y.shape == (B, 2, S - 184, S - 184) and bool(torch.isfinite(y).all())
    y.backward(dl)
    for p in net.parameters():
        assert bool(torch.isfinite(p.grad).all())
    with torch.no_grad():
        for i in (0, B - 1):
            yi = net(x[i:i + 1].contiguous())
            assert ((yi[0] - y[i]).abs().max() / y.abs().max()).item() < FWD_TOL


def test_overlap_knob_gives_bit_identical_gradients(net):
    """unet_set_overlap(1): weight gradients on the handle's auxiliary stream next to the dgrad chain, re-joined at every stage
    end.  Same kernels on the same data, only their interleaving differs: logits and all 46 gradients must be bit-identical
    (any missing stream dependency shows up as a difference here or as a failure of the fp64 comparison at S=380)."""
    import _hip
    from oracle import parity, prng
    L = _hip.lib()
    x = torch.from_numpy(prng.make_input(1, 2, 252)).cuda()
    dl = torch.from_numpy(prng.make_cotangent(2, (2, 2, 68, 68))).cuda()

    def once():
        net.zero_grad(set_to_none=True)
        y = net(x)
        y.backward(dl)
        torch.cuda.synchronize()
        return y.detach().clone(), [p.grad.clone() for p in net.parameters()]

    _hip.check(L.unet_set_overlap(0), "unet_set_overlap")
    y0, g0 = once()
    _hip.check(L.unet_set_overlap(1), "unet_set_overlap")
    try:
        for _ in range(3):                                   # repeated: a race would not lose every time
            y1, g1 = once()
            assert torch.equal(y0, y1) and all(torch.equal(a, b) for a, b in zip(g0, g1))
        r = parity.check_same_branch(380, 1)
        assert r["fwd"] < FWD_TOL and max(r["grads"].values()) < GRAD_TOL
    finally:
        _hip.check(L.unet_set_overlap(-1), "unet_set_overlap")       # the default: per arithmetic mode


def test_rccl_communicator_behind_the_c_abi_single_rank(net):
    """unet_dp_* (csrc/dp.hip): unique id, ncclCommInitRank, broadcast, per-stage all-reduce on the communicator stream,
    join.  With one rank the collectives are identities, so the gradients must equal the plain backward bit for bit —
    what this pins is that the RCCL code path, its stream fork/join and the bucket ranges execute on the GPU."""
    import copy
    import ctypes as C
    import _hip
    import network
    from oracle import prng
    L = _hip.lib()
    assert L.unet_dp_rccl_version() > 0
    m = copy.deepcopy(net)
    x = torch.from_numpy(prng.make_input(1, 2, 188)).cuda()
    dl = torch.from_numpy(prng.make_cotangent(2, (2, 2, 4, 4))).cuda()
    m.zero_grad(set_to_none=True)
    m(x).backward(dl)
    plain = [p.grad.clone() for p in m.parameters()]
    before = [p.detach().clone() for p in m.parameters()]
    m.enable_data_parallel(backend="rccl")              # no torch.distributed: a communicator of one rank
    h = m._get_handle(0)                                # the module's own handle: communicator + gradient scale live there
    assert h is not network._handle(0, m.base_ch)
    assert L.unet_dp_world(h.h) == 1 and L.unet_dp_world(network._handle(0, m.base_ch).h) == 0
    assert all(torch.equal(a, b.detach()) for a, b in zip(before, m.parameters()))        # broadcast from rank 0 == identity
    m.zero_grad(set_to_none=True)
    m(x).backward(dl)
    torch.cuda.synchronize()
    assert all(torch.equal(a, p.grad) for a, p in zip(plain, m.parameters()))
    # raw entry points: SUM all-reduce of a buffer, ordered after the producer on the caller's stream
    buf = torch.arange(1000, dtype=torch.float32, device="cuda") * 0.5
    _hip.check(L.unet_dp_allreduce(h.h, _hip.ptr(buf), buf.numel(), _hip.stream()), "allreduce")
    _hip.check(L.unet_dp_join(h.h, _hip.stream()), "join")
    torch.cuda.synchronize()
    assert torch.equal(buf.cpu(), torch.arange(1000, dtype=torch.float32) * 0.5)
    _hip.check(L.unet_dp_destroy(h.h), "destroy")
    assert L.unet_dp_world(h.h) == 0
    assert L.unet_dp_allreduce(h.h, _hip.ptr(buf), 10, _hip.stream()) != 0                 # loud, not silent, without a communicator


def test_training_stop_goal_follows_the_reference_identity_comparisons(net, golden_dir, tmp_path, capsys):
    """trainer.py:18-27,185-214: a caller's literal 'ISBI2012' arms the stop goal, the same text built at run time (argv)
    and 'DIC-C2DH-HeLa' do not.  Golden: the reference's own training() run on the same inputs
    (tests/golden/make_golden_trainer.py): files written, goal lines, and the six progress series."""
    import copy
    import json
    from oracle import prng
    from trainer import training
    gold = json.load(open(os.path.join(golden_dir, "trainer_golden.json")))
    S, So = gold["meta"]["S"], gold["meta"]["S"] - 184

    def loader(seeds):
        return [(torch.from_numpy(prng.make_input(s, 2, S)), torch.from_numpy(prng.make_labels(s, 2, So))) for s in seeds]

    cases = {"literal_ISBI2012": "ISBI2012", "runtime_ISBI2012": "".join(["ISBI", "2012"]), "literal_DIC-C2DH-HeLa": "DIC-C2DH-HeLa"}
    got_series = {}
    for case, name in cases.items():
        m = copy.deepcopy(net)
        out = os.path.join(tmp_path, case)
        capsys.readouterr()
        training(m, loader(gold["meta"]["train_seeds"]), loader(gold["meta"]["val_seeds"]), gold["meta"]["epochs_arg"], 2,
                 torch.device("cuda:0"), out, name)
        printed = capsys.readouterr().out.splitlines()
        want = gold["cases"][case]
        files = sorted(os.path.relpath(os.path.join(r, f), out) for r, _, fs in os.walk(out) for f in fs)
        assert files == want["files"], (case, files)
        assert [ln for ln in printed if ln.startswith("The goal was reached")] == want["goal_lines"], case
        assert sum(ln == "Model has been saved:" for ln in printed) == want["n_model_saved_lines"], case
        got_series[case] = {series: np.atleast_1d(np.loadtxt(os.path.join(out, "progress", series + ".out"))) for series in want["progress"]}
    # the three cases differ only in the dataset string: the same arithmetic, run-to-run deterministic -> identical series
    # (their values are checked against the reference's fp64 series in test_trainer_series_within_the_reference_fp32_distance)
    for case in cases:
        for series, v in got_series[case].items():
            assert np.array_equal(v, got_series["literal_ISBI2012"][series]), (case, series)


@pytest.mark.parametrize("S", [188, 220])
def test_trainer_series_against_fp64_on_the_same_branch_and_the_reference_series(net, golden_dir, tmp_path, S):
    """The six progress series of training() (2 epochs of 2 train batches + 1 validation batch, B=2; 4 SGD steps), three ways:
      1. dl-unet_amd/trainer.training() on the HIP path;
      2. the same step sequence through the module API with an fp64 C-oracle SHADOW that follows its own fp64 weights but
         evaluates every training step on the ReLU/pool branch the HIP forward took (oracle/parity.shadow_training): the
         rigorous check of the whole step chain (forward, class-balanced BCE, backward, SGD momentum) over several steps -
         losses within 1e-5 relative, final weights within 1e-5 of their scale, masks identical;
      3. the reference's own training() in fp32 and fp64 (tests/golden/trainer_series.json): the HIP series may differ from
         the reference's fp64 series by twice the larger of the reference's OWN fp32 distance from it and the exact fp64
         effect of the HIP run's branch choices (|shadow - f64|), plus one forward rounding.
    (1) and (2) must be bit-identical on the HIP side: same kernels, same order."""
    import copy
    import json
    from oracle import parity, prng
    from trainer import training
    gold = json.load(open(os.path.join(golden_dir, "trainer_series.json")))
    meta, series = gold["meta"], gold["sizes"]["S%d" % S]
    So = S - 184

    def loader(seeds):
        return [(torch.from_numpy(prng.make_input(s, 2, S)), torch.from_numpy(prng.make_labels(s, 2, So))) for s in seeds]

    out = os.path.join(tmp_path, "run")
    training(copy.deepcopy(net), loader(meta["train_seeds"]), loader(meta["val_seeds"]), meta["epochs_arg"], 2, torch.device("cuda:0"),
             out, "".join(["ISBI", "2012"]))
    hip, shadow, final = parity.shadow_training(copy.deepcopy(net), loader(meta["train_seeds"]), loader(meta["val_seeds"]), meta["epochs_arg"])
    print("S=%d final weights: HIP vs fp64 shadow on the same branch %.2e" % (S, final))
    assert final < TRAINER_SAME_BRANCH_TOL, final
    npx = So * So
    for name in ("loss", "loss_val", "train_eval_iou", "train_eval_pe", "val_eval_iou", "val_eval_pe"):
        got = np.atleast_1d(np.loadtxt(os.path.join(out, "progress", name + ".out")))
        is_loss = name.startswith("loss")
        # (1) == (2): the files hold float32 losses printed with 18 digits, the stepwise loop summed python floats of the same values
        assert np.allclose(got, np.array(hip[name]), rtol=1e-6 if is_loss else 0, atol=0), (name, got, hip[name])
        sh = np.array(shadow[name])
        e_sh = np.abs(got - sh)
        f32, f64, spread = (np.array(series[name][k]) for k in ("f32", "f64", "thread_spread"))
        ref_dist = np.maximum(np.abs(f32 - f64), spread)
        branch = np.abs(sh - f64)
        floor = TRAINER_LOSS_FLOOR * np.abs(f64) if is_loss else np.full_like(f64, TRAINER_PIXEL_FLOOR / npx)
        err = np.abs(got - f64)
        print("S=%d %-15s HIP %s | shadow(f64, same branch) %s | reference f64 %s | |HIP-shadow| %s  |HIP-f64| %s  ref fp32 dist %s  branch effect %s"
              % (S, name, got, sh, f64, e_sh, err, ref_dist, branch))
        if is_loss:
            assert (e_sh <= TRAINER_SAME_BRANCH_TOL * np.abs(sh)).all(), (name, got.tolist(), sh.tolist())
        else:
            assert (e_sh <= TRAINER_PIXEL_FLOOR / npx + 1e-12).all(), (name, got.tolist(), sh.tolist())
        assert (err <= 2 * np.maximum(ref_dist, branch) + floor).all(), (S, name, got.tolist(), f64.tolist(), err.tolist(), ref_dist.tolist(), branch.tolist())


def test_training_resume_is_the_interrupted_run(net, tmp_path, capsys):
    """training(resume_from=...) continues the interrupted run for everything it writes: the six progress series (the earlier
    epochs are kept, not overwritten), the best-model file and the stop-goal state (a goal reached before the interruption
    is not re-armed).  Uninterrupted: epochs 0..3; interrupted after epoch 1 and resumed for 2..3 from checkpoint_latest.pth."""
    import copy
    from oracle import prng
    from trainer import training
    S, So = 188, 4

    def loader(seeds):
        return [(torch.from_numpy(prng.make_input(s, 2, S)), torch.from_numpy(prng.make_labels(s, 2, So))) for s in seeds]

    dev = torch.device("cuda:0")
    full, part = os.path.join(tmp_path, "full"), os.path.join(tmp_path, "part")
    training(copy.deepcopy(net), loader([10, 11]), loader([20]), 3, 2, dev, full, "ISBI2012", save_optimizer=True)
    goal_full = [ln for ln in capsys.readouterr().out.splitlines() if ln.startswith("The goal was reached")]
    training(copy.deepcopy(net), loader([10, 11]), loader([20]), 1, 2, dev, part, "ISBI2012", save_optimizer=True)
    goal_a = [ln for ln in capsys.readouterr().out.splitlines() if ln.startswith("The goal was reached")]
    training(copy.deepcopy(net), loader([10, 11]), loader([20]), 3, 2, dev, part, "ISBI2012", save_optimizer=True,
             resume_from=os.path.join(part, "models", "checkpoint_latest.pth"))
    goal_b = [ln for ln in capsys.readouterr().out.splitlines() if ln.startswith("The goal was reached")]
    assert goal_a + goal_b == goal_full and len(goal_full) == 1                  # reached once, not again after the resume
    for f in sorted(os.listdir(os.path.join(full, "progress"))):
        a = np.atleast_1d(np.loadtxt(os.path.join(full, "progress", f)))
        b = np.atleast_1d(np.loadtxt(os.path.join(part, "progress", f)))
        assert a.shape == (4,) and np.array_equal(a, b), (f, a, b)
    assert sorted(os.listdir(os.path.join(full, "models"))) == sorted(os.listdir(os.path.join(part, "models")))
    for f in ("unet_weight_save_best.pth", "unet_weight_save_ISBI2012.pth"):
        sa = torch.load(os.path.join(full, "models", f), weights_only=True)
        sb = torch.load(os.path.join(part, "models", f), weights_only=True)
        assert all(torch.equal(sa[k], sb[k]) for k in sa), f


def test_checkpoint_resume_with_momentum_is_bit_exact(net, tmp_path):
    """the reference saves weights only; checkpoint.py adds the SGD momentum buffers.  An interrupted-and-resumed
    run must be the uninterrupted run, bit for bit; resuming from weights alone (the reference's format) must not be."""
    import copy
    import checkpoint
    import optim as hip_optim
    from oracle import prng
    x = torch.from_numpy(prng.make_input(1, 2, 188)).cuda()
    tgt = hip_optim.onehot2(torch.from_numpy(prng.make_labels(3, 2, 4)), torch.empty(2, 2, 4, 4, device="cuda"))

    def steps(m, opt, n):
        for _ in range(n):
            opt.zero_grad(set_to_none=True)
            hip_optim.bce_with_logits(m(x), tgt).backward()
            opt.step()

    a = copy.deepcopy(net); oa = hip_optim.SGD(a.parameters(), lr=1e-4, momentum=0.99)
    steps(a, oa, 2)
    path = checkpoint.save_checkpoint(os.path.join(tmp_path, "ck.pth"), a, oa, epoch=7)
    torch.save(a.state_dict(), os.path.join(tmp_path, "weights_only.pth"))
    steps(a, oa, 2)
    b = copy.deepcopy(net); ob = hip_optim.SGD(b.parameters(), lr=1e-4, momentum=0.99)
    extra = checkpoint.load_checkpoint(path, b, ob)
    assert extra["epoch"] == 7
    steps(b, ob, 2)
    assert all(torch.equal(p, q) for p, q in zip(a.parameters(), b.parameters()))
    c = copy.deepcopy(net); oc = hip_optim.SGD(c.parameters(), lr=1e-4, momentum=0.99)
    assert checkpoint.load_checkpoint(os.path.join(tmp_path, "weights_only.pth"), c, oc) == {}      # a reference-format file loads too
    steps(c, oc, 2)
    assert any(not torch.equal(p, q) for p, q in zip(a.parameters(), c.parameters()))             # ...but the momentum is gone
    assert checkpoint.find_resume_checkpoint(str(tmp_path)) is None


def test_arithmetic_mode_is_fixed_per_forward(net, math_mode):
    """The mode is read when a forward is planned and kept with its plan: changing the process default between a
    forward and its backward (another thread's call, a scheduler) must not change that backward."""
    from oracle import prng
    x = torch.from_numpy(prng.make_input(1, 2, 188)).cuda()
    dl = torch.from_numpy(prng.make_cotangent(2, (2, 2, 4, 4))).cuda()
    math_mode(3)
    net.zero_grad(set_to_none=True)
    net(x).backward(dl)
    ref = [p.grad.clone() for p in net.parameters()]
    net.zero_grad(set_to_none=True)
    y = net(x)
    math_mode(0)
    y.backward(dl)
    assert all(torch.equal(a, p.grad) for a, p in zip(ref, net.parameters()))


def test_config3_arithmetic_at_its_real_size(math_mode):
    """BASELINE configs[2] per-GPU work: batch 8 x 572^2 in bf16 compute (mode 2).  Same-branch fp64 oracle at S=380 (the
    largest size the oracle finishes in seconds), then at B=8 x 572 the size-independent properties: batch independence of
    the logits and additivity of the gradients over a ragged split of the batch."""
    import network
    from oracle import parity, prng
    math_mode(2)
    r = parity.check_same_branch(380, 1)
    assert r["fwd"] < 5e-2, r["fwd"]
    assert max(r["grads"].values()) < 6e-2
    m = network.Unet()
    m.load_state_dict({k: torch.from_numpy(v) for k, v in prng.make_params(0).items()})
    m = m.to("cuda:0")
    S, B = 572, 8
    x = torch.from_numpy(prng.make_input(5, B, S)).cuda()
    dl = torch.from_numpy(prng.make_cotangent(6, (B, 2, 388, 388))).cuda()
    y = m(x)
    y.backward(dl)
    g_all = [p.grad.clone() for p in m.parameters()]
    assert bool(torch.isfinite(y).all())
    with torch.no_grad():
        y0 = m(x[0:1].contiguous()); y7 = m(x[7:8].contiguous())
    scale = y.abs().max()
    assert ((y0[0] - y[0]).abs().max() / scale).item() < 1e-5 and ((y7[0] - y[7]).abs().max() / scale).item() < 1e-5
    acc = None
    for lo, hi in ((0, 3), (3, 8)):
        m.zero_grad(set_to_none=True)
        m(x[lo:hi].contiguous()).backward(dl[lo:hi].contiguous())
        part = [p.grad.clone() for p in m.parameters()]
        acc = part if acc is None else [a + b for a, b in zip(acc, part)]
    for a, b in zip(acc, g_all):
        assert ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item() < 1e-3     # same bf16 products, fp32 sums in another order


def test_bench_self_launch_two_ranks_on_this_gpu(tmp_path):
    """`python3 bench.py --gpus 2` without WORLD_SIZE: the launcher starts both ranks itself (child processes), rank 0's line
    is relayed.  With --share-gpu both ranks use this box's one GPU over gloo (RCCL refuses two ranks on a device), which
    exercises the launch path, the per-module data-parallel handle, the 1/world gradient scale and the bucket logic on
    hardware; the driver's multi-GPU runs take the same path with one device per rank and the RCCL communicator."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    detail = os.path.join(tmp_path, "detail.json")
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--share-gpu", "--steps", "2", "--warmup", "1", "--batch", "1",
                          "--detail", detail], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1 and len(lines[0]) < 2048, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 2 and d["config"]["parallelism"] == "dp2"
    assert d["comm"]["self_launched"] is True and d["comm"]["ranks"] == 2 and "gloo" in d["comm"]["gradient_allreduce"]
    assert d["value"] > 0 and "roofline" in d and "cpu_baseline" not in d            # the CPU baseline belongs to the N = 1 line
    assert len(json.load(open(detail))["layers"]) > 50
