"""Pins oracle/aux_ref.py (numpy/scipy restatement of mirror_transform, elastic_transform, metrics,
class_balance) against goldens produced by the reference's own functions.  CPU only."""
import os

import numpy as np

from oracle import aux_ref, prng


def test_mirror_transform_matches_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, "aux_golden.npz"))
    for n in (196, 388):
        img = (np.arange(n)[:, None] * 1000.0 + np.arange(n)[None, :]).astype(np.float64)
        m = aux_ref.mirror_transform(img)
        assert tuple(g["mirror_%d_shape" % n]) == m.shape
        assert np.array_equal(m[0], g["mirror_%d_row0" % n]) and np.array_equal(m[:, 0], g["mirror_%d_col0" % n])
        assert np.array_equal(np.diagonal(m), g["mirror_%d_diag" % n]) and np.array_equal(m[-1], g["mirror_%d_last" % n])
        rnd = prng.uniform01(7, 50 + n, n * n).reshape(n, n)
        mr = aux_ref.mirror_transform(rnd)
        assert np.allclose([mr.sum(), (mr * mr).sum(), mr[::7, ::5].sum()], g["mirror_%d_rand_sum" % n], rtol=1e-13)


def test_elastic_transform_matches_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, "aux_golden.npz"))
    for tag in ("a", "b"):
        alpha, sigma, H, seed = g["elastic_%s_params" % tag]
        H = int(H)
        img = prng.uniform01(7, 1, H * H).reshape(H, H) * 255.0
        tgt = (prng.uniform01(7, 2, H * H).reshape(H, H) > 0.5) * 255.0
        rs = np.random.RandomState(int(seed))
        (a, b), _, _ = aux_ref.elastic_transform((img, tgt), alpha, sigma, (rs.rand(H, H), rs.rand(H, H)))
        assert np.abs(a - g["elastic_%s_img" % tag]).max() < 1e-9
        assert np.abs(b - g["elastic_%s_tgt" % tag]).max() < 1e-9


def test_metrics_and_class_balance(golden_dir):
    g = np.load(os.path.join(golden_dir, "aux_golden.npz"))
    ka = np.load(os.path.join(golden_dir, "known_answers.npz"))
    p = prng.make_labels(11, 1, 64)[0, 0]; l = prng.make_labels(12, 1, 64)[0, 0]
    inter, union, diff = aux_ref.eval_counts(p, l)
    assert np.allclose([inter / union, diff / p.size], g["evalm_rand"].ravel(), rtol=1e-14)
    lab = prng.make_labels(3, 2, 36)[:, 0]
    for b in range(2):
        assert np.array_equal(aux_ref.class_balance(lab[b]), ka["class_balance_rand"][b])
    w, r = aux_ref.gaussian_taps(10.0)
    assert r == 40 and abs(w.sum() - 1) < 1e-15


def test_reflect_rotate_crop_oracle_vs_reference_golden(golden_dir):
    """oracle/aux_ref.reflect_rotate_crop against the outputs of the reference's own statements (data.py:103-125, executed by
    tests/golden/make_golden_rotate.py): integer images, bit-exact."""
    import os
    from oracle import aux_ref, prng
    g = np.load(os.path.join(golden_dir, "rotate_golden.npz"))
    for tag in ("a", "b", "c", "d", "full"):
        crop, seed, deg, S = [int(v) for v in g["%s_params" % tag]]
        st = int(g["%s_stride" % tag])
        img = (prng.uniform01(9, seed, crop * crop).reshape(crop, crop) * 255).astype(np.uint8)
        tgt = ((prng.uniform01(9, 100 + seed, crop * crop).reshape(crop, crop) > 0.5) * 255).astype(np.uint8)
        ri = aux_ref.reflect_rotate_crop(img, deg); rt = aux_ref.reflect_rotate_crop(tgt, deg)
        assert ri.shape == (S, S) and ri.dtype == np.uint8
        assert np.array_equal(ri[::st, ::st], g["%s_img_sample" % tag]) and np.array_equal(rt[::st, ::st], g["%s_tgt_sample" % tag])
        assert [int(ri.astype(np.int64).sum()), int(rt.astype(np.int64).sum())] == [int(v) for v in g["%s_sums" % tag]]


def test_augment_oracle_at_the_config4_size_vs_reference_golden(golden_dir):
    """BASELINE configs[3] shape (512^2 sample -> 700^2 input): oracle/aux_ref.augment against tests/golden/augment_golden_S700.npz,
    made by executing the reference's own __getitem__ statements and elastic_transform (make_golden_augment.py): rotation angle
    as the reference drew it (330 and 0 degrees), alpha = 200, sigma = 10, the RandomState the reference would have created."""
    g = np.load(os.path.join(golden_dir, "augment_golden_S700.npz"))
    for tag in ("a", "b"):
        crop, seed, deg, S, eseed, _ = [int(v) for v in g["%s_params" % tag]]
        img, tgt = aux_ref.cells(seed, crop)
        inp, gt = aux_ref.augment(img, tgt, deg, 200, 10, np.random.RandomState(eseed))
        assert inp.shape == (S, S) and gt.shape == (crop, crop)
        assert np.abs(inp[::5, ::5] - g["%s_inp_sample" % tag]).max() < 1e-6          # the fixture is float32
        assert np.abs(gt[::4, ::4] - g["%s_gt_sample" % tag]).max() < 1e-3
        assert np.allclose([inp.sum(), (inp * inp).sum(), gt.sum(), float((gt > 127).sum())], g["%s_sums" % tag], rtol=1e-12)
