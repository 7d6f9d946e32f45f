"""GPU kernels either side of the hot path against the pinned numpy/scipy oracle
(oracle/aux_ref.py) and the reference's goldens.  Integer results bit-exact; float results <= 2e-6."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    import _hip
    _hip.lib()
    assert torch.cuda.is_available()
    torch.cuda.set_device(0)
    return torch.device("cuda:0")


def test_mirror_transform_bit_exact(dev, golden_dir):
    import data
    from oracle import aux_ref, prng
    g = np.load(os.path.join(golden_dir, "aux_golden.npz"))
    for n in (196, 388, 512):
        img = prng.uniform01(7, 50 + n, n * n).reshape(n, n).astype(np.float32)
        m = data.mirror_transform(torch.from_numpy(img).to(dev)).cpu().numpy()
        ref = aux_ref.mirror_transform(img)
        assert m.shape == ref.shape and np.array_equal(m, ref)           # a pure gather: bit-exact
    n = 196                                                              # index-encoding image against the reference golden
    enc = (np.arange(n)[:, None] * 1000.0 + np.arange(n)[None, :]).astype(np.float32)
    m = data.mirror_transform(torch.from_numpy(enc).to(dev)).cpu().numpy()
    assert np.array_equal(m[0], g["mirror_196_row0"]) and np.array_equal(m[:, 0], g["mirror_196_col0"])
    assert np.array_equal(np.diagonal(m), g["mirror_196_diag"]) and np.array_equal(m[-1], g["mirror_196_last"])
    # batched + normalised front end (ImageDataset_test): mirror then (x-min)/ptp per image
    batch = np.stack([prng.uniform01(8, i, n * n).reshape(n, n) * (i + 1) + i for i in range(3)]).astype(np.float32)
    out = data.test_input(torch.from_numpy(batch).to(dev)).cpu().numpy()
    for i in range(3):
        ref = aux_ref.normalise01(aux_ref.mirror_transform(batch[i].astype(np.float64)))
        assert np.abs(out[i, 0] - ref).max() < 2e-6
    with pytest.raises(RuntimeError):                                    # too small to mirror into its input size
        data.mirror_transform(torch.zeros(36, 36, device=dev))


def test_elastic_transform_vs_reference_golden(dev, golden_dir):
    import data
    from oracle import prng
    g = np.load(os.path.join(golden_dir, "aux_golden.npz"))
    for tag in ("a", "b"):
        alpha, sigma, H, seed = g["elastic_%s_params" % tag]
        H = int(H)
        img = prng.uniform01(7, 1, H * H).reshape(H, H) * 255.0
        tgt = (prng.uniform01(7, 2, H * H).reshape(H, H) > 0.5) * 255.0
        a, b = data.elastic_transform((torch.from_numpy(img).float().to(dev), torch.from_numpy(tgt).float().to(dev)),
                                      alpha=float(alpha), sigma=float(sigma), random_state=np.random.RandomState(int(seed)))
        # fp32 warp of 0..255 images: the displacement field itself carries fp32 rounding (alpha * 1e-7),
        # which moves a bilinear sample of a 0/255 mask by up to ~1e-3; tolerance 2e-5 of the value range
        assert np.abs(a.cpu().numpy() - g["elastic_%s_img" % tag]).max() < 255 * 2e-5
        assert np.abs(b.cpu().numpy() - g["elastic_%s_tgt" % tag]).max() < 255 * 2e-5
    # batched, device RNG: shapes and the identity for alpha = 0
    x = torch.rand(3, 40, 40, device=dev)
    (y,) = data.elastic_transform((x,), alpha=0.0, sigma=3.0)
    assert torch.equal(x, y)


def test_class_balance_and_metrics_bit_exact(dev, golden_dir):
    import functions
    import optim as hip_optim
    from oracle import aux_ref, prng
    ka = np.load(os.path.join(golden_dir, "known_answers.npz"))
    lab = torch.from_numpy(prng.make_labels(3, 2, 36)[:, 0]).to(dev)
    w = functions.class_balance(lab)
    assert w.is_cuda and np.array_equal(w.cpu().numpy(), ka["class_balance_rand"])
    # fused crop + argmax + counts against the oracle on a strided, padded logits tensor
    B, So, n = 3, 52, 36
    g = torch.Generator().manual_seed(5)
    logits = torch.randn(B, 2, So, So, generator=g)
    labels = torch.from_numpy(prng.make_labels(21, B, n))
    mask, stats = hip_optim.crop_argmax_metrics(logits.to(dev), labels.to(dev))
    pad = (So - n) // 2
    ref_mask = logits[:, :, pad:pad + n, pad:pad + n].argmax(dim=1).numpy()
    assert np.array_equal(mask.cpu().numpy(), ref_mask)
    for b in range(B):
        assert tuple(stats[b].tolist()) == aux_ref.eval_counts(ref_mask[b], labels[b, 0].numpy())
    m = functions.metrics_from_counts(*[int(v) for v in stats[0].tolist()], n * n)
    assert np.allclose(m, functions.evaluation_metrics(torch.from_numpy(ref_mask[0]), labels[0, 0]))


def test_config5_overlap_tile_inference_base32(dev):
    """BASELINE config #5 shape at reduced batch: 1024^2 images -> mirror to 1212 -> 32-base-ch net ->
    crop/argmax/pixel-error, end to end on the device, against the torch restatement on the host."""
    import data
    import network
    import optim as hip_optim
    from oracle import aux_ref, prng, torch_ref
    torch.manual_seed(0)
    net = network.Unet(base_ch=32).to(dev)
    n, B = 1024, 1
    imgs = torch.from_numpy(prng.uniform01(31, 0, B * n * n).reshape(B, n, n).astype(np.float32) * 255)
    labels = torch.from_numpy(prng.make_labels(32, B, n))
    x = data.test_input(imgs.to(dev))
    assert x.shape == (B, 1, 1212, 1212)
    with torch.no_grad():
        y = net(x)
    assert y.shape == (B, 2, 1028, 1028)
    mask, stats = hip_optim.crop_argmax_metrics(y, labels.to(dev))
    p = {k: v.detach().cpu() for k, v in net.state_dict().items()}
    xr = torch.from_numpy(aux_ref.normalise01(aux_ref.mirror_transform(imgs[0].double().numpy()))).float()[None, None]
    yr = torch_ref.unet_forward(p, xr)
    assert ((y.cpu() - yr).abs().max() / yr.abs().max()).item() < 2e-5


def test_base32_net_trains(dev):
    """Unet(base_ch=32) (BASELINE config #5's width) through forward + backward: its 32-channel layers fill half a 64-channel
    weight-gradient tile.  Every gradient against the torch restatement in fp64 on the HIP forward's own ReLU / pool branch is
    not available for this width (the C oracle's same-branch entry point is), so: fp64 C oracle, same branch, all elements."""
    import network
    from oracle import oracle_c, parity, prng
    import _hip
    S, B, base = 188, 2, 32
    params = prng.make_params(0, base=base)
    x = prng.make_input(1, B, S)
    dl = prng.make_cotangent(2, (B, 2, S - 184, S - 184))
    net = network.Unet(base_ch=base)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()})
    net = net.to(dev)
    L = _hip.lib()
    h = network._handle(0, base)
    plist = [p.detach() for p in net._params()]
    nbytes = h.workspace_bytes(B, S, True)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    logits = torch.empty(B, 2, S - 184, S - 184, device=dev)
    ptab = _hip.ptr_table(plist)
    _hip.check(L.unet_forward(h.h, ptab, _hip.ptr(torch.from_numpy(x).to(dev)), _hip.ptr(logits), B, S, _hip.ptr(ws), nbytes, 1, _hip.stream()), "fwd")
    grads = [torch.empty_like(p) for p in plist]
    _hip.check(L.unet_backward(h.h, ptab, _hip.ptr(torch.from_numpy(dl).to(dev)), _hip.ptr_table(grads), _hip.ptr(ws), nbytes, _hip.stream()), "bwd")
    torch.cuda.synchronize()
    masks, sels = parity.branch_of(h, ws, B, S)
    p64 = {k: v.astype(np.float64) for k, v in params.items()}
    ref_logits, ref_grads = oracle_c.unet_fwd_bwd(p64, x.astype(np.float64), base=base, dlogits=dl.astype(np.float64), relu_masks=masks, pool_sel=sels)
    assert parity.nerr(logits.cpu().numpy(), ref_logits) < 2e-5
    for (k, _), g in zip(net.named_parameters(), grads):
        assert parity.nerr(g.cpu().numpy(), ref_grads[k]) < 3e-4, k
    # and through the module / autograd surface
    net.zero_grad(set_to_none=True)
    net(torch.from_numpy(x).to(dev)).backward(torch.from_numpy(dl).to(dev))
    for p_, g in zip(net.parameters(), grads):
        assert torch.equal(p_.grad, g)


def test_config5_at_its_real_batch_crosses_2GiB(dev):
    """BASELINE configs[4] as stated: batch 16 of 1024^2 images -> 1212^2 inputs, 32-base-ch net, crop/argmax/pixel error.
    The first activation is 16 x 1210^2 x 32 x 4 B = 3.0 GB: above 2 GiB the MFMA kernels stage with global_load_lds
    instead of buffer descriptors (32-bit num_records).  Every tile of the batch must equal the same image run alone
    (the B=1 path is oracle-checked in the test above)."""
    import data
    import network
    import optim as hip_optim
    from oracle import prng
    torch.manual_seed(0)
    net = network.Unet(base_ch=32).to(dev)
    n, B = 1024, 16
    imgs = torch.stack([torch.from_numpy(prng.uniform01(31, i, n * n).reshape(n, n).astype(np.float32) * 255) for i in range(B)])
    labels = torch.stack([torch.from_numpy(prng.make_labels(32 + i, 1, n)[0]) for i in range(B)])
    x = data.test_input(imgs.to(dev))
    assert x.shape == (B, 1, 1212, 1212)
    with torch.no_grad():
        y = net(x)
        assert y.shape == (B, 2, 1028, 1028) and bool(torch.isfinite(y).all())
        mask, stats = hip_optim.crop_argmax_metrics(y, labels.to(dev))
        scale = y.abs().max()
        for i in (0, 7, 15):
            yi = net(x[i:i + 1].contiguous())
            assert ((yi[0] - y[i]).abs().max() / scale).item() < 2e-6
            mi, si = hip_optim.crop_argmax_metrics(yi, labels[i:i + 1].to(dev))
            margin = (yi[0, 0] - yi[0, 1]).abs()
            pad = (1028 - n) // 2
            safe = margin[pad:pad + n, pad:pad + n] > 1e-5 * scale
            assert bool((mi[0][safe] == mask[i][safe]).all())
            assert (si[0] - stats[i]).abs().max().item() <= int((~safe).sum())


def test_reflect_rotate_crop_vs_scipy_oracle(dev, golden_dir):
    """N1: np.pad(reflect) + scipy.ndimage.rotate (cubic spline) + centre crop (data.py:106-125), fused on the device.
    Oracle: oracle/aux_ref.reflect_rotate_crop (pinned bit-exactly to the reference's own statements by the CPU suite).
    Float result: <= 2e-5 of the value range (fp32 prefilter and weights against scipy's fp64).  uint8 result: scipy rounds
    t + 0.5, so a pixel whose spline value lies within fp32 noise of k + 0.5 may land on the neighbouring level: at most 1 level
    off, on fewer than 0.1 % of the pixels; rotations by multiples of 90 degrees are pure permutations and must be exact."""
    import data
    from oracle import aux_ref, prng
    g = np.load(os.path.join(golden_dir, "rotate_golden.npz"))
    for tag in ("a", "b", "c", "d"):
        crop, seed, deg, S = [int(v) for v in g["%s_params" % tag]]
        st = int(g["%s_stride" % tag])
        img = (prng.uniform01(9, seed, crop * crop).reshape(crop, crop) * 255).astype(np.uint8)
        tgt = ((prng.uniform01(9, 100 + seed, crop * crop).reshape(crop, crop) > 0.5) * 255).astype(np.uint8)
        both = torch.from_numpy(np.stack([img, tgt]).astype(np.float32)).to(dev)
        out = data.reflect_rotate_crop(both, [deg, deg], S, levels=255).cpu().numpy()
        for k, src in enumerate((img, tgt)):
            ref = aux_ref.reflect_rotate_crop(src, deg).astype(np.int64)
            d = np.abs(out[k].astype(np.int64) - ref)
            assert d.max() <= 1 and (d > 0).mean() < 1e-3, (tag, k, d.max(), (d > 0).mean())
            assert np.abs(out[k][::st, ::st].astype(np.int64) - g["%s_%s_sample" % (tag, ("img", "tgt")[k])].astype(np.int64)).max() <= 1
        fl = data.reflect_rotate_crop(both[0], deg, S, levels=0).cpu().numpy()
        ref = aux_ref.reflect_rotate_crop(img.astype(np.float64), deg)
        assert np.abs(fl - ref).max() < 255 * 2e-5
    # all twelve angles of the reference (np.arange(0, 360, 30)) on one image; 0/90/180/270 are exact
    crop = 36
    img = (prng.uniform01(9, 50, crop * crop).reshape(crop, crop) * 255).astype(np.uint8)
    x = torch.from_numpy(np.repeat(img[None].astype(np.float32), 12, axis=0)).to(dev)
    out = data.reflect_rotate_crop(x, list(range(0, 360, 30)), levels=255).cpu().numpy()
    for i, deg in enumerate(range(0, 360, 30)):
        ref = aux_ref.reflect_rotate_crop(img, deg).astype(np.int64)
        d = np.abs(out[i].astype(np.int64) - ref)
        assert d.max() <= (0 if deg % 90 == 0 else 1) and (d > 0).mean() < 1e-3, (deg, d.max(), (d > 0).mean())
    # the reference's real size: 388 crop -> 572 input, angle as drawn by the reference
    crop, seed, deg, S = [int(v) for v in g["full_params"]]
    img = (prng.uniform01(9, seed, crop * crop).reshape(crop, crop) * 255).astype(np.uint8)
    out = data.reflect_rotate_crop(torch.from_numpy(img.astype(np.float32)).to(dev), deg, levels=255).cpu().numpy()
    assert out.shape == (S, S) and np.array_equal(out[::4, ::4].astype(np.uint8), g["full_img_sample"])
    assert int(out.astype(np.int64).sum()) == int(g["full_sums"][0])


def test_augment_pipeline_on_device(dev):
    """ImageDataset.__getitem__ after the file reads (data.py:97-135): crop -> reflect pad + rotation + centre crop -> elastic
    deformation shared by image and mask -> label crop + threshold -> normalisation, against the same steps done with the
    numpy/scipy oracle."""
    import data
    from oracle import aux_ref, prng
    H, crop = 120, 36
    img = (prng.uniform01(12, 1, H * H).reshape(H, H) * 255).astype(np.uint8)
    yy, xx = np.mgrid[0:H, 0:H]
    tgt = ((((yy - 60) ** 2 + (xx - 50) ** 2) < 30 ** 2) * 255).astype(np.uint8)
    x0, y0, deg, alpha, sigma = 40, 30, 210, 200.0, 10.0
    S = aux_ref.input_size_compute(crop)[1]
    rs = np.random.RandomState(77)
    f0, f1 = rs.rand(S, S), rs.rand(S, S)
    inp, gt = data.augment(torch.from_numpy(img.astype(np.float32)).to(dev), torch.from_numpy(tgt.astype(np.float32)).to(dev),
                           (x0, y0), crop, deg, alpha, sigma, fields=(f0, f1))
    # the oracle works on the uint8 arrays like the reference (scipy writes rotation and warp results in the input's type)
    rs2 = np.random.RandomState(77)
    inp_ref, et = aux_ref.augment(img[x0:x0 + crop, y0:y0 + crop], tgt[x0:x0 + crop, y0:y0 + crop], deg, alpha, sigma, rs2)
    gt_ref = (et > 127).astype(np.int64)
    assert inp.shape == (1, S, S) and gt.shape == (1, crop, crop) and gt.dtype == torch.int64
    # two roundings to grey levels on the way (rotation, warp), each of which a value within fp32 noise of k + 0.5 may flip
    d = np.abs(inp.cpu().numpy()[0] - inp_ref)
    assert d.max() < 2.5 / 255 and (d > 0.6 / 255).mean() < 2e-2, (d.max(), (d > 0.6 / 255).mean())
    assert (gt.cpu().numpy()[0] != gt_ref).mean() < 5e-3


def test_augment_batch_equals_per_sample_calls(dev):
    """data.augment on a batch ([B,H,W] images, one crop origin and one angle per sample) gives exactly what B per-sample calls
    give with the same elastic fields: the kernels are per-image, only the launches are shared."""
    import data
    from oracle import aux_ref
    n, B = 96, 3
    ims, tgs = zip(*[aux_ref.cells(10 + b, 160) for b in range(B)])
    img = torch.from_numpy(np.stack(ims).astype(np.float32)).to(dev)
    tgt = torch.from_numpy(np.stack(tgs).astype(np.float32)).to(dev)
    origins = [(5, 9), (40, 0), (64, 64)]
    angles = [30.0, 0.0, 240.0]
    S = aux_ref.input_size_compute(n)[1]
    rs = np.random.RandomState(3)
    f0 = rs.rand(B, S, S); f1 = rs.rand(B, S, S)
    xb, gb = data.augment(img, tgt, origins, n, angles, 200.0, 10.0, fields=(f0, f1))
    assert xb.shape == (B, 1, S, S) and gb.shape == (B, 1, n, n) and gb.dtype == torch.int64
    for b in range(B):
        x1, g1 = data.augment(img[b], tgt[b], origins[b], n, angles[b], 200.0, 10.0, fields=(f0[b], f1[b]))
        assert torch.equal(x1, xb[b]) and torch.equal(g1, gb[b])
