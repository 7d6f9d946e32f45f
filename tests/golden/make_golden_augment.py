"""Generates tests/golden/augment_golden_S700.npz by EXECUTING the reference's own statements (this container only).

BASELINE configs[3] shape: a 512x512 sample -> 700x700 network input.  ImageDataset.__getitem__ (data.py:97-137) cannot run as
a whole (the class reads files with OpenCV), so its statements from `original_size = image.shape[-1]` (data.py:103) to the crop
of the deformed mask `gt = gt[pad:original_size+pad, ...]` (data.py:130), and the normalisation statement
`inp = (inp - np.min(inp))/np.ptp(inp)` (data.py:134), are taken from the file's AST and executed in a namespace holding
exactly the names they use: np, rotate (scipy.ndimage.rotate, as data.py:11 imports it), input_size_compute (the reference's
functions.py) and elastic_transform (the reference's own function, data.py:225-245, from the same AST).  The two statements
in between use OpenCV, which this image does not have: `cv.threshold(gt, 127, 255, cv.THRESH_BINARY)` + `gt / 255`
(data.py:131-132) are NOT executed; the fixture stores the deformed mask BEFORE the threshold, and the test applies `> 127`
(OpenCV's published THRESH_BINARY rule) to both sides - that one statement is "parity unpinned".
np.random is seeded so that the `rot_deg` the reference draws is reproducible, and np.random.RandomState(None) - what
elastic_transform creates for itself - is pinned to a seeded generator for the duration of the call.
Nothing of the reference is stored: inputs are prng seeds, the fixture holds the drawn angle and strided samples / sums of the
outputs."""
import ast
import os
import sys
import types

import numpy as np
from scipy.ndimage import gaussian_filter, map_coordinates, rotate

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import aux_ref  # noqa: E402

REF = "/root/reference"
ELASTIC_SEED = 4321


def reference_code():
    sys.modules.setdefault("cv2", types.ModuleType("cv2"))
    sys.path.insert(0, REF)
    import functions
    tree = ast.parse(open(os.path.join(REF, "data.py")).read())
    ns = {"np": np, "rotate": rotate, "gaussian_filter": gaussian_filter, "map_coordinates": map_coordinates,
          "input_size_compute": functions.input_size_compute}
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name == "elastic_transform":
            exec(compile(ast.Module([node], []), "reference:data.py:elastic_transform", "exec"), ns)
    cls = [n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "ImageDataset"][0]
    fn = [n for n in cls.body if isinstance(n, ast.FunctionDef) and n.name == "__getitem__"][0]

    def idx_of(pred):
        return [i for i, st in enumerate(fn.body) if pred(st)][0]
    first = idx_of(lambda st: isinstance(st, ast.Assign) and getattr(st.targets[0], "id", "") == "original_size")
    crop_gt = idx_of(lambda st: isinstance(st, ast.Assign) and getattr(st.targets[0], "id", "") == "gt" and "original_size" in ast.dump(st.value))
    norm = idx_of(lambda st: isinstance(st, ast.Assign) and getattr(st.targets[0], "id", "") == "inp" and "ptp" in ast.dump(st.value))
    part1 = compile(ast.Module(fn.body[first:crop_gt + 1], []), "reference:data.py:__getitem__", "exec")
    part2 = compile(ast.Module([fn.body[norm]], []), "reference:data.py:__getitem__", "exec")
    return ns, part1, part2


class _Self:
    alpha, sigma = 200, 10            # main_main.py:175


def main():
    ns, part1, part2 = reference_code()
    out = {}
    real_rs = np.random.RandomState
    for tag, seed in (("a", 1), ("b", 2)):
        img, tgt = aux_ref.cells(seed)          # the sample generator lives with the test oracle so that the GPU box regenerates it
        run = dict(ns, image=img, target=tgt, self=_Self())
        np.random.seed(100 + seed)
        np.random.RandomState = lambda s=None: real_rs(ELASTIC_SEED + seed) if s is None else real_rs(s)
        try:
            exec(part1, run)
            exec(part2, run)
        finally:
            np.random.RandomState = real_rs
        inp, gt = np.asarray(run["inp"], dtype=np.float64), np.asarray(run["gt"], dtype=np.float64)
        assert inp.shape == (700, 700) and gt.shape == (512, 512), (inp.shape, gt.shape)
        out["%s_params" % tag] = np.array([512, seed, int(run["rot_deg"]), int(run["input_size"]), ELASTIC_SEED + seed, 100 + seed])
        out["%s_inp_sample" % tag] = inp[::5, ::5].astype(np.float32)
        out["%s_gt_sample" % tag] = gt[::4, ::4].astype(np.float32)           # deformed mask BEFORE the threshold (grey levels)
        out["%s_sums" % tag] = np.array([inp.sum(), (inp * inp).sum(), gt.sum(), float((gt > 127).sum())])
        print(tag, "rot_deg", run["rot_deg"], "input_size", run["input_size"], "mask fraction %.3f" % (gt > 127).mean())
    out["meta"] = np.array(repr(dict(numpy=np.__version__, scipy=__import__("scipy").__version__)))
    np.savez_compressed(os.path.join(HERE, "augment_golden_S700.npz"), **out)


if __name__ == "__main__":
    main()
