"""Per-op parity of the HIP kernels (through the C ABI) against fp64 torch CPU references and the
C oracle.  Floating point: tolerance = normalised max error (|d|_inf / |ref|_inf) <= 2e-5 for fp32
accumulation over up to 9216 terms (tolerances are normalised by tensor scale)."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

TOL = 2e-5


@pytest.fixture(scope="module")
def hip():
    import _hip
    _hip.lib()
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    torch.cuda.set_device(0)
    return _hip


def nerr(a, ref):
    a = a.detach().double().cpu(); ref = ref.detach().double().cpu()
    return ((a - ref).abs().max() / ref.abs().max().clamp_min(1e-300)).item()


def nhwc(t):   # NCHW cpu -> NHWC cuda fp32
    return t.permute(0, 2, 3, 1).contiguous().float().cuda()


def nchw(t):   # NHWC cuda -> NCHW cpu fp64
    return t.permute(0, 3, 1, 2).double().cpu()


class Keep(list):
    """Owns the device tensors whose raw pointers are passed to the library."""

    def __call__(self, t):
        self.append(t)
        return t


def scratch(nbytes):
    return torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device="cuda")


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g, dtype=torch.float64) * scale


@pytest.fixture(params=[(0, 1), (3, 1), (0, 0), (3, 0)], ids=["direct", "winograd", "direct-glds", "winograd-glds"])
def conv_mode(hip, request):
    """The 3x3 layers have two fp32 evaluations: the direct fmaf chain (math mode 0) and Winograd F(2x2,3x3) (mode 3);
    each stages its operands either by buffer-descriptor LDS-DMA (default) or by global_load_lds — the instantiation
    tensors >= 2 GiB take (BASELINE config #5 at batch 16), forced here with unet_set_lds_dma(0)."""
    mode, dma = request.param
    default = hip.lib().unet_get_math()
    hip.check(hip.lib().unet_set_math(mode), "set_math")
    hip.check(hip.lib().unet_set_lds_dma(dma), "set_lds_dma")
    yield mode
    hip.check(hip.lib().unet_set_lds_dma(1), "set_lds_dma")
    hip.check(hip.lib().unet_set_math(default), "set_math")


@pytest.mark.parametrize("B,H,C,K", [(2, 21, 64, 64), (1, 37, 64, 128), (3, 14, 128, 128), (1, 12, 256, 512), (2, 9, 128, 64), (1, 30, 32, 32),
                                     (3, 26, 128, 128), (1, 22, 256, 512), (2, 45, 64, 64),
                                     (1, 66, 512, 512),
                                     # minimal Winograd tile grids (9 x 9, 13 x 13 tiles per image): every 64-tile workgroup crosses an image boundary, ragged last one
                                     (4, 20, 64, 64), (5, 28, 64, 128),
                                     # 32 filter rows (the base-32 net of BASELINE configs[4]): half a 64-row block of transformed filters
                                     (2, 40, 32, 32), (1, 34, 32, 64), (1, 44, 64, 32), (2, 22, 96, 96)])
def test_conv3x3_fwd(hip, conv_mode, B, H, C, K):
    keep = Keep()
    x = rnd(B, C, H, H, seed=1); w = rnd(K, C, 3, 3, seed=2, scale=0.05); b = rnd(K, seed=3)
    ref = F.relu(F.conv2d(x, w, b))
    y = torch.empty(B, H - 2, H - 2, K, device="cuda")
    sc = scratch(hip.lib().unet_conv3x3_scratch_bytes(C, K))
    hip.check(hip.lib().unet_conv3x3_fwd(hip.ptr(keep(nhwc(x))), H, H, C, 0, None, 0, B, H, H, hip.ptr(keep(w.float().cuda())),
                                         hip.ptr(keep(b.float().cuda())), K, 1, hip.ptr(y), hip.ptr(sc), hip.stream()), "conv3x3_fwd")
    assert nerr(nchw(y), ref) < TOL


def test_conv3x3_random_shapes(hip):
    """30 random shapes (channel counts incl. 192, batch 1-3, extents 6-64, virtual concat with pad -3..7) through the
    forward and backward entry points in the default mode, against fp64 (tools/fuzz_conv.py)."""
    import importlib.util, os
    spec = importlib.util.spec_from_file_location("fuzz_conv", os.path.join(os.path.dirname(__file__), "..", "tools", "fuzz_conv.py"))
    fz = importlib.util.module_from_spec(spec); spec.loader.exec_module(fz)
    assert fz.run(30, 11, verbose=False) < TOL


@pytest.mark.parametrize("B,Hs,pad,C1,C2,K", [(2, 8, 6, 64, 64, 64), (1, 10, 3, 128, 128, 128), (1, 6, 0, 64, 64, 128),
                                               (1, 24, 4, 64, 64, 64), (2, 30, -3, 64, 64, 128), (1, 20, 1, 128, 128, 64),
                                               (1, 26, 5, 32, 32, 32), (2, 32, -2, 32, 32, 32)])
def test_conv3x3_fwd_virtual_concat(hip, conv_mode, B, Hs, pad, C1, C2, K):
    keep = Keep()
    """crop_and_concat (network.py:108-127) is never materialised: the conv reads two sources."""
    H = Hs + 2 * pad
    a = rnd(B, C1, Hs, Hs, seed=1); u = rnd(B, C2, H, H, seed=2)
    w = rnd(K, C1 + C2, 3, 3, seed=3, scale=0.05); b = rnd(K, seed=4)
    cat = torch.cat((F.pad(a, (pad,) * 4), u), 1)
    ref = F.relu(F.conv2d(cat, w, b))
    y = torch.empty(B, H - 2, H - 2, K, device="cuda")
    sc = scratch(hip.lib().unet_conv3x3_scratch_bytes(C1 + C2, K))
    hip.check(hip.lib().unet_conv3x3_fwd(hip.ptr(keep(nhwc(a))), Hs, Hs, C1, pad, hip.ptr(keep(nhwc(u))), C2, B, H, H,
                                         hip.ptr(keep(w.float().cuda())), hip.ptr(keep(b.float().cuda())), K, 1, hip.ptr(y), hip.ptr(sc),
                                         hip.stream()), "conv3x3_fwd concat")
    assert nerr(nchw(y), ref) < TOL


@pytest.mark.parametrize("B,H,C,K,use_mask,use_add", [(2, 21, 64, 64, True, False), (1, 38, 64, 128, False, True),
                                                      (2, 13, 128, 256, True, True), (1, 70, 64, 64, True, False), (2, 25, 128, 256, True, True),
                                                      (1, 66, 512, 512, True, True), (4, 20, 64, 64, True, True),
                                                      (2, 40, 32, 32, True, True), (1, 34, 32, 64, True, False), (1, 44, 64, 32, False, True)])
def test_conv3x3_bwd(hip, conv_mode, B, H, C, K, use_mask, use_add):
    keep = Keep()
    x = rnd(B, C, H, H, seed=1).requires_grad_(True)
    w = rnd(K, C, 3, 3, seed=2, scale=0.05).requires_grad_(True)
    dz = rnd(B, K, H - 2, H - 2, seed=3)
    mask = rnd(B, C, H, H, seed=4).clamp_min(0) if use_mask else None     # a ReLU output: many exact zeros
    add = rnd(B, C, H, H, seed=5) if use_add else None
    F.conv2d(x, w).backward(dz)
    dx_ref = x.grad.clone()
    if add is not None:
        dx_ref = dx_ref + add
    if mask is not None:
        dx_ref = dx_ref * (mask > 0)
    dx = torch.empty(B, H, H, C, device="cuda"); dw = torch.empty(K, C, 3, 3, device="cuda"); db = torch.empty(K, device="cuda")
    sc = scratch(hip.lib().unet_conv3x3_bwd_scratch_bytes(B, H, H, C, K))
    hip.check(hip.lib().unet_conv3x3_bwd(hip.ptr(keep(nhwc(x.detach()))), H, H, C, 0, None, 0, B, H, H, hip.ptr(keep(w.detach().float().cuda())), K,
                                         hip.ptr(keep(nhwc(dz))), hip.ptr(dx), hip.ptr(keep(nhwc(mask))) if use_mask else None,
                                         hip.ptr(keep(nhwc(add))) if use_add else None, None, None, hip.ptr(dw), hip.ptr(db),
                                         hip.ptr(sc), hip.stream()), "conv3x3_bwd")
    assert nerr(nchw(dx), dx_ref) < TOL
    assert nerr(dw, w.grad) < TOL
    assert nerr(db, dz.sum((0, 2, 3))) < TOL


@pytest.mark.parametrize("B,Hs,pad,C,K", [(2, 8, 6, 64, 64), (1, 12, 3, 128, 128), (1, 8, 0, 64, 64), (1, 24, 4, 64, 64), (2, 30, -3, 64, 128),
                                          (1, 26, 5, 32, 32), (2, 32, -2, 32, 64)])
def test_conv3x3_bwd_virtual_concat(hip, conv_mode, B, Hs, pad, C, K):
    keep = Keep()
    H = Hs + 2 * pad
    a = rnd(B, C, Hs, Hs, seed=1).requires_grad_(True); u = rnd(B, C, H, H, seed=2).requires_grad_(True)
    w = rnd(K, 2 * C, 3, 3, seed=3, scale=0.05).requires_grad_(True)
    dz = rnd(B, K, H - 2, H - 2, seed=4)
    F.conv2d(torch.cat((F.pad(a, (pad,) * 4), u), 1), w).backward(dz)
    dx1 = torch.empty(B, Hs, Hs, C, device="cuda"); dx2 = torch.empty(B, H, H, C, device="cuda")
    dw = torch.empty(K, 2 * C, 3, 3, device="cuda"); db = torch.empty(K, device="cuda")
    sc = scratch(hip.lib().unet_conv3x3_bwd_scratch_bytes(B, H, H, 2 * C, K))
    hip.check(hip.lib().unet_conv3x3_bwd(hip.ptr(keep(nhwc(a.detach()))), Hs, Hs, C, pad, hip.ptr(keep(nhwc(u.detach()))), C, B, H, H,
                                         hip.ptr(keep(w.detach().float().cuda())), K, hip.ptr(keep(nhwc(dz))), hip.ptr(dx1), None, None,
                                         hip.ptr(dx2), None, hip.ptr(dw), hip.ptr(db), hip.ptr(sc), hip.stream()), "conv3x3_bwd concat")
    assert nerr(nchw(dx1), a.grad) < TOL      # pad-backward == crop of the padded gradient
    assert nerr(nchw(dx2), u.grad) < TOL
    assert nerr(dw, w.grad) < TOL
    assert nerr(db, dz.sum((0, 2, 3))) < TOL


@pytest.mark.parametrize("B,H,Ci,Co", [(2, 7, 128, 64), (1, 13, 256, 128), (1, 4, 1024, 512), (3, 17, 64, 32)])
def test_upconv2_fwd(hip, B, H, Ci, Co):
    keep = Keep()
    x = rnd(B, Ci, H, H, seed=1); w = rnd(Ci, Co, 2, 2, seed=2, scale=0.05); b = rnd(Co, seed=3)
    ref = F.conv_transpose2d(x, w, b, stride=2)
    y = torch.empty(B, 2 * H, 2 * H, Co, device="cuda")
    sc = scratch(hip.lib().unet_upconv2_scratch_bytes(B, H, H, max(Ci, 64), max(Co, 64)))
    hip.check(hip.lib().unet_upconv2_fwd(hip.ptr(keep(nhwc(x))), B, H, H, Ci, hip.ptr(keep(w.float().cuda())), hip.ptr(keep(b.float().cuda())), Co,
                                         hip.ptr(y), hip.ptr(sc), hip.stream()), "upconv2_fwd")
    assert nerr(nchw(y), ref) < TOL


@pytest.fixture(params=[1, 0], ids=["pixel-linear", "row-walking"])
def up_staging(hip, request):
    """The up-conv weight gradient runs as the pixel-linear kernel (buffer-descriptor LDS-DMA); with unet_set_lds_dma(0) - what
    tensors >= 2 GiB take by themselves - it falls back to the row-walking kernel."""
    hip.check(hip.lib().unet_set_lds_dma(request.param), "set_lds_dma")
    yield request.param
    hip.check(hip.lib().unet_set_lds_dma(1), "set_lds_dma")


# (the weight gradient is the pixel-linear kernel: fewer pixels than one 32-pixel chunk, a ragged last chunk, chunks that cross
#  image boundaries, the half-filled channel tiles of the base-32 net, several channel tiles in both directions)
@pytest.mark.parametrize("B,H,Ci,Co", [(2, 7, 128, 64), (1, 13, 256, 128), (1, 18, 128, 64), (1, 5, 64, 64), (3, 11, 64, 32), (5, 6, 256, 256)])
def test_upconv2_bwd(hip, B, H, Ci, Co, up_staging):
    keep = Keep()
    x = rnd(B, Ci, H, H, seed=1).clamp_min(0).requires_grad_(True)     # the producer's ReLU output
    w = rnd(Ci, Co, 2, 2, seed=2, scale=0.05).requires_grad_(True)
    dy = rnd(B, Co, 2 * H, 2 * H, seed=3)
    F.conv_transpose2d(x, w, stride=2).backward(dy)
    dx_ref = x.grad * (x.detach() > 0)
    dx = torch.empty(B, H, H, Ci, device="cuda"); dw = torch.empty(Ci, Co, 2, 2, device="cuda"); db = torch.empty(Co, device="cuda")
    sc = scratch(hip.lib().unet_upconv2_scratch_bytes(B, H, H, Ci, Co))
    xd = nhwc(x.detach())
    hip.check(hip.lib().unet_upconv2_bwd(hip.ptr(xd), B, H, H, Ci, hip.ptr(keep(w.detach().float().cuda())), Co, hip.ptr(keep(nhwc(dy))),
                                         hip.ptr(dx), hip.ptr(xd), hip.ptr(dw), hip.ptr(db), hip.ptr(sc), hip.stream()), "upconv2_bwd")
    assert nerr(nchw(dx), dx_ref) < TOL
    assert nerr(dw, w.grad) < TOL
    assert nerr(db, dy.sum((0, 2, 3))) < TOL


def test_maxpool2_fwd_bwd_exact(hip):
    keep = Keep()
    """Pool is a selection: bit-exact, including first-max-wins ties (all-zero windows after ReLU)."""
    B, H, Cc = 2, 12, 64
    pre = rnd(B, Cc, H, H, seed=1).clamp_min(0).float()
    pre[0, :, 0:2, 0:2] = 0.0
    pre[1, 3, 4:6, 4:6] = 1.25                                   # positive tie: first (row-major) wins
    p = pre.clone().requires_grad_(True)
    y_ref = F.max_pool2d(F.relu(p), 2, 2)
    dy = rnd(B, Cc, H // 2, H // 2, seed=2).float()
    y_ref.backward(dy)
    y = torch.empty(B, H // 2, H // 2, Cc, device="cuda"); dpre = torch.empty(B, H, H, Cc, device="cuda")
    xd = nhwc(pre)
    hip.check(hip.lib().unet_maxpool2_fwd(hip.ptr(xd), hip.ptr(y), B, H, H, Cc, hip.stream()))
    hip.check(hip.lib().unet_maxpool2_bwd(hip.ptr(xd), hip.ptr(keep(nhwc(dy))), hip.ptr(dpre), B, H, H, Cc, hip.stream()))
    assert torch.equal(y.permute(0, 3, 1, 2).cpu(), y_ref.detach())
    assert torch.equal(dpre.permute(0, 3, 1, 2).cpu(), p.grad)


def test_head1x1_fwd_bwd(hip):
    keep = Keep()
    B, H, Cc = 2, 37, 64
    x = rnd(B, Cc, H, H, seed=1).clamp_min(0).requires_grad_(True)
    w = rnd(2, Cc, 1, 1, seed=2, scale=0.1).requires_grad_(True); b = rnd(2, seed=3)
    ref = F.conv2d(x, w, b)
    dl = rnd(B, 2, H, H, seed=4)
    ref.backward(dl)
    logits = torch.empty(B, 2, H, H, device="cuda")
    xd = nhwc(x.detach())
    hip.check(hip.lib().unet_head1x1_fwd(hip.ptr(xd), B, H, H, Cc, hip.ptr(keep(w.detach().float().cuda())), hip.ptr(keep(b.float().cuda())),
                                         hip.ptr(logits), hip.stream()))
    assert nerr(logits, ref) < TOL
    dz = torch.empty(B, H, H, Cc, device="cuda"); dw = torch.empty(2, Cc, 1, 1, device="cuda"); db = torch.empty(2, device="cuda")
    sc = scratch(hip.lib().unet_head1x1_bwd_scratch_bytes(B, H, H, Cc))
    hip.check(hip.lib().unet_head1x1_bwd(hip.ptr(xd), B, H, H, Cc, hip.ptr(keep(w.detach().float().cuda())), hip.ptr(keep(dl.float().cuda())),
                                         hip.ptr(dz), hip.ptr(dw), hip.ptr(db), hip.ptr(sc), hip.stream()))
    assert nerr(nchw(dz), x.grad * (x.detach() > 0)) < TOL
    assert nerr(dw, w.grad) < TOL
    assert nerr(db, dl.sum((0, 2, 3))) < TOL


@pytest.mark.parametrize("B,S,K", [(2, 44, 64), (1, 188, 64), (3, 60, 32), (1, 572, 64)])
def test_conv1ch_fwd_bwd_vs_c_oracle(hip, B, S, K):
    keep = Keep()
    from oracle import oracle_c
    x = rnd(B, 1, S, S, seed=1).float(); w = rnd(K, 1, 3, 3, seed=2).float(); b = rnd(K, seed=3).float()
    ref = oracle_c.conv_valid_fwd(x.double().numpy(), w.double().numpy(), b.double().numpy(), True)
    y = torch.empty(B, S - 2, S - 2, K, device="cuda")
    hip.check(hip.lib().unet_conv1ch_fwd(hip.ptr(keep(x.cuda())), B, S, hip.ptr(keep(w.cuda())), hip.ptr(keep(b.cuda())), K, hip.ptr(y), hip.stream()))
    assert nerr(nchw(y), torch.from_numpy(ref)) < TOL
    dz = rnd(B, K, S - 2, S - 2, seed=4).float()
    _, dw_ref, db_ref = oracle_c.conv_valid_bwd(x.double().numpy(), w.double().numpy(), dz.double().numpy(), need_dx=False)
    dw = torch.empty(K, 1, 3, 3, device="cuda"); db = torch.empty(K, device="cuda")
    sc = scratch(hip.lib().unet_conv1ch_bwd_scratch_bytes(B, S, K))
    hip.check(hip.lib().unet_conv1ch_bwd(hip.ptr(keep(x.cuda())), B, S, K, hip.ptr(keep(nhwc(dz))), hip.ptr(dw), hip.ptr(db), hip.ptr(sc), hip.stream()))
    assert nerr(dw, torch.from_numpy(dw_ref)) < TOL
    assert nerr(db, torch.from_numpy(db_ref)) < TOL


def test_step_side_kernels(hip, golden_dir):
    keep = Keep()
    import os
    from oracle import oracle_c, prng
    L = hip.lib()
    ka = np.load(os.path.join(golden_dir, "known_answers.npz"))
    g = np.load(os.path.join(golden_dir, "unet_S220.npz"))
    lg = torch.from_numpy(g["logits_f64"]).float().cuda()
    labels = torch.from_numpy(prng.make_labels(3, 2, 36)).cuda()
    B, _, H, W = lg.shape
    ll = torch.empty_like(lg)
    hip.check(L.unet_onehot2(hip.ptr(labels), hip.ptr(ll), B, H, W, hip.stream()))
    assert torch.equal(ll[:, 1].cpu(), labels[:, 0].float().cpu()) and torch.equal(ll[:, 0].cpu(), 1 - labels[:, 0].float().cpu())
    loss = torch.empty(1, device="cuda"); dx = torch.empty_like(lg)
    sc = scratch(L.unet_bce_scratch_bytes(lg.numel()))
    # L1 unweighted: golden from the reference's nn.BCEWithLogitsLoss
    hip.check(L.unet_bce_logits(hip.ptr(lg), hip.ptr(ll), None, 0, 0, 0, 0, B, H, W, hip.ptr(loss), hip.ptr(dx), 1.0, hip.ptr(sc), hip.stream()))
    assert abs(loss.item() - float(ka["bce_plain_loss"])) < 1e-5 * abs(float(ka["bce_plain_loss"]))
    assert nerr(dx, torch.from_numpy(ka["bce_plain_grad"])) < 1e-5
    # L1 weighted with the reference's broadcast (Q4): weight [B,H,W] aligned so that B meets the class axis
    wm = torch.from_numpy(ka["class_balance_rand"]).float().cuda()           # [2,H,W]
    hip.check(L.unet_bce_logits(hip.ptr(lg), hip.ptr(ll), hip.ptr(wm), 0, H * W, W, 1, B, H, W, hip.ptr(loss), hip.ptr(dx), 1.0,
                                hip.ptr(sc), hip.stream()))
    assert abs(loss.item() - float(ka["bce_weighted_loss"])) < 1e-5 * abs(float(ka["bce_weighted_loss"]))
    assert nerr(dx, torch.from_numpy(ka["bce_weighted_grad"])) < 1e-5
    # L2 argmax: integer result, bit-exact against the reference
    am = torch.empty(B, H, W, dtype=torch.int64, device="cuda")
    hip.check(L.unet_argmax2(hip.ptr(lg), 2 * H * W, H * W, W, hip.ptr(am), B, H, W, hip.stream()))
    assert np.array_equal(am.cpu().numpy(), ka["argmax_S220"])
    tie = torch.zeros(1, 2, 4, 4, device="cuda"); am2 = torch.empty(1, 4, 4, dtype=torch.int64, device="cuda")
    hip.check(L.unet_argmax2(hip.ptr(tie), 32, 16, 4, hip.ptr(am2), 1, 4, 4, hip.stream()))
    assert int(am2.sum()) == 0                                               # ties -> class 0
    # L1 + L2 fused (unet_bce_step): integer labels instead of the one-hot target, loss + gradient + argmax in one pass;
    # the logits are a centre-cropped VIEW of a larger tensor, as in the trainer (trainer.py:60-61)
    big = torch.full((B, 2, H + 4, W + 6), 123.0, device="cuda")
    big[:, :, 2:2 + H, 3:3 + W] = lg
    view = big[:, :, 2:2 + H, 3:3 + W]
    sc2 = scratch(L.unet_bce_step_scratch_bytes(B * H * W))
    for wptr, wstr, kl, kg in ((None, (0, 0, 0, 0), "bce_plain_loss", "bce_plain_grad"), (hip.ptr(wm), (0, H * W, W, 1), "bce_weighted_loss", "bce_weighted_grad")):
        dx2 = torch.full_like(lg, -7.0); am3 = torch.full((B, H, W), -1, dtype=torch.int64, device="cuda")
        hip.check(L.unet_bce_step(hip.ptr(view), view.stride(0), view.stride(1), view.stride(2), hip.ptr(labels), wptr, wstr[0], wstr[1], wstr[2], wstr[3],
                                  B, H, W, hip.ptr(loss), hip.ptr(dx2), 1.0, hip.ptr(am3), hip.ptr(sc2), hip.stream()))
        assert abs(loss.item() - float(ka[kl])) < 1e-5 * abs(float(ka[kl]))
        assert nerr(dx2, torch.from_numpy(ka[kg])) < 1e-5
        assert np.array_equal(am3.cpu().numpy(), ka["argmax_S220"])
    # grad_scale, and the optional outputs left out
    hip.check(L.unet_bce_step(hip.ptr(lg), 2 * H * W, H * W, W, hip.ptr(labels), None, 0, 0, 0, 0, B, H, W, hip.ptr(loss), hip.ptr(dx2), 0.25,
                              None, hip.ptr(sc2), hip.stream()))
    assert nerr(dx2, 0.25 * torch.from_numpy(ka["bce_plain_grad"])) < 1e-5
    hip.check(L.unet_bce_step(hip.ptr(lg), 2 * H * W, H * W, W, hip.ptr(labels), None, 0, 0, 0, 0, B, H, W, hip.ptr(loss), None, 1.0,
                              None, hip.ptr(sc2), hip.stream()))
    assert abs(loss.item() - float(ka["bce_plain_loss"])) < 1e-5 * abs(float(ka["bce_plain_loss"]))
    # module-level wrapper: loss.backward() delivers that gradient to the logits
    import optim as hip_optim
    lgr = lg.clone().requires_grad_(True)
    l2, m2 = hip_optim.bce_argmax_step(lgr, labels, weight=wm)
    l2.backward()
    assert nerr(lgr.grad, torch.from_numpy(ka["bce_weighted_grad"])) < 1e-5 and np.array_equal(m2.cpu().numpy(), ka["argmax_S220"])
    # L3 SGD momentum, two steps, against torch.optim.SGD
    torch.manual_seed(0)
    ps = [torch.randn(n, device="cuda") for n in (5000, 3, 70001)]
    ref = [torch.nn.Parameter(p.clone()) for p in ps]
    opt = torch.optim.SGD(ref, lr=1e-4, momentum=0.99)
    bufs = [torch.zeros_like(p) for p in ps]
    numel = (C.c_size_t * 3)(*[p.numel() for p in ps])
    for step in range(2):
        gs = [torch.randn_like(p) for p in ps]
        for r, gg in zip(ref, gs):
            r.grad = gg.clone()
        opt.step()
        hip.check(L.unet_sgd_momentum(hip.ptr_table(ps), hip.ptr_table(gs), hip.ptr_table(bufs), numel, 3, 1e-4, 0.99, int(step == 0), hip.stream()))
    for p, r in zip(ps, ref):
        assert torch.allclose(p, r.detach(), rtol=0, atol=5e-7)      # <= 1 ulp at |p| ~ 4 (fma vs mul+add)
    # pointers that are not 16-byte aligned take the 4-byte kernel: same result
    base = [torch.randn(n + 1, device="cuda") for n in (5000, 70001)]
    ps2 = [b[1:] for b in base]
    ref2 = [torch.nn.Parameter(p.clone()) for p in ps2]
    opt2 = torch.optim.SGD(ref2, lr=1e-4, momentum=0.99)
    bufs2 = [torch.zeros_like(p) for p in ps2]
    numel2 = (C.c_size_t * 2)(*[p.numel() for p in ps2])
    for step in range(2):
        gs = [torch.randn_like(p) for p in ps2]
        for r, gg in zip(ref2, gs):
            r.grad = gg.clone()
        opt2.step()
        hip.check(L.unet_sgd_momentum(hip.ptr_table(ps2), hip.ptr_table(gs), hip.ptr_table(bufs2), numel2, 2, 1e-4, 0.99, int(step == 0), hip.stream()))
    for p, r in zip(ps2, ref2):
        assert torch.allclose(p, r.detach(), rtol=0, atol=5e-7)
