"""Generates tests/golden/aux_golden.npz by running the REFERENCE's own helper functions (this container only).

data.py cannot be imported whole (cv2, wget, torchvision are absent and `scipy.ndimage.interpolation` no
longer exists in scipy 1.15), so the two pure numpy/scipy functions are taken from its AST and executed
in a namespace that supplies exactly the names they use.  Nothing of the reference is stored: the
fixture holds inputs' seeds and outputs."""
import ast
import os
import sys
import types

import numpy as np
import torch
from scipy.ndimage import gaussian_filter, map_coordinates

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import prng  # noqa: E402

REF = "/root/reference"


def reference_functions():
    for name in ("cv2",):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.path.insert(0, REF)
    import functions                                   # input_size_compute, class_balance, evaluation_metrics
    src = open(os.path.join(REF, "data.py")).read()
    tree = ast.parse(src)
    ns = {"np": np, "torch": torch, "gaussian_filter": gaussian_filter, "map_coordinates": map_coordinates,
          "input_size_compute": functions.input_size_compute}
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name in ("elastic_transform", "mirror_transform", "mirror_transform_tensor"):
            exec(compile(ast.Module([node], []), "reference:data.py", "exec"), ns)
    return ns, functions


def main():
    ns, functions = reference_functions()
    out = {}
    # mirror_transform: an index-encoding image makes the (separable) map readable from the output
    for n in (196, 388):
        img = (np.arange(n)[:, None] * 1000.0 + np.arange(n)[None, :]).astype(np.float64)
        m = ns["mirror_transform"](img)
        out["mirror_%d_shape" % n] = np.array(m.shape)
        out["mirror_%d_row0" % n] = m[0].copy(); out["mirror_%d_col0" % n] = m[:, 0].copy()
        out["mirror_%d_diag" % n] = np.diagonal(m).copy()
        out["mirror_%d_last" % n] = m[-1].copy()
        rnd = prng.uniform01(7, 50 + n, n * n).reshape(n, n)
        mr = ns["mirror_transform"](rnd)
        out["mirror_%d_rand_sum" % n] = np.array([mr.sum(), (mr * mr).sum(), mr[::7, ::5].sum()])
        mt = ns["mirror_transform_tensor"](torch.from_numpy(rnd).reshape(1, 1, n, n))
        assert np.array_equal(mt.numpy().reshape(mr.shape), mr)
    # elastic_transform with a seeded RandomState: record the uniform fields it draws, and its outputs
    H = 64
    img = prng.uniform01(7, 1, H * H).reshape(H, H) * 255.0
    tgt = (prng.uniform01(7, 2, H * H).reshape(H, H) > 0.5) * 255.0
    for tag, alpha, sigma in (("a", 30.0, 4.0), ("b", 200.0, 10.0)):
        rs = np.random.RandomState(1234)
        inp, gt = ns["elastic_transform"]((img, tgt), alpha=alpha, sigma=sigma, random_state=rs)
        # the two uniform fields are np.random.RandomState(1234).rand(H, H) twice (MT19937 is stable): not stored
        out["elastic_%s_params" % tag] = np.array([alpha, sigma, H, 1234])
        out["elastic_%s_img" % tag] = np.asarray(inp); out["elastic_%s_tgt" % tag] = np.asarray(gt)
    # evaluation metrics + class_balance on random masks
    p = prng.make_labels(11, 1, 64)[0, 0]; l = prng.make_labels(12, 1, 64)[0, 0]
    out["evalm_rand"] = functions.evaluation_metrics(torch.from_numpy(p), torch.from_numpy(l))
    out["meta"] = np.array(repr(dict(numpy=np.__version__, scipy=__import__("scipy").__version__, torch=torch.__version__)))
    np.savez_compressed(os.path.join(HERE, "aux_golden.npz"), **out)
    print("aux goldens written:", sorted(out.keys()))


if __name__ == "__main__":
    main()
