"""Generates tests/golden/rotate_golden.npz by EXECUTING the reference's own statements (this container only).

The reflect pad + random rotation + centre crop live inside ImageDataset.__getitem__ (data.py:103-125), which cannot run
as a whole (the class reads image files with OpenCV).  Its statements from `original_size = image.shape[-1]` to
`target = target_rot[t:b, l:r]` are taken from the file's AST and executed in a namespace holding exactly the names they
use (np, rotate = scipy.ndimage.rotate as data.py:11 imports it, input_size_compute from the reference's functions.py),
with np.random seeded so that the `rot_deg` the reference draws is reproducible.  Nothing of the reference is stored:
inputs are prng seeds, the fixture holds the drawn angle and the outputs (uint8, as the reference's images are)."""
import ast
import os
import sys
import types

import numpy as np
from scipy.ndimage import rotate

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import prng  # noqa: E402

REF = "/root/reference"


def reference_statements():
    sys.modules.setdefault("cv2", types.ModuleType("cv2"))
    sys.path.insert(0, REF)
    import functions
    tree = ast.parse(open(os.path.join(REF, "data.py")).read())
    cls = [n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "ImageDataset"][0]
    fn = [n for n in cls.body if isinstance(n, ast.FunctionDef) and n.name == "__getitem__"][0]
    first = [i for i, st in enumerate(fn.body) if isinstance(st, ast.Assign) and getattr(st.targets[0], "id", "") == "original_size"][0]
    last = [i for i, st in enumerate(fn.body) if isinstance(st, ast.Assign) and getattr(st.targets[0], "id", "") == "target"
            and "target_rot" in ast.dump(st.value)][0]
    code = compile(ast.Module(fn.body[first:last + 1], []), "reference:data.py:__getitem__", "exec")
    return code, functions


def main():
    code, functions = reference_statements()
    out = {}
    cases = [("a", 36, 1), ("b", 36, 2), ("c", 68, 3), ("d", 36, 7), ("full", 388, 5)]
    for tag, crop, seed in cases:
        img = (prng.uniform01(9, seed, crop * crop).reshape(crop, crop) * 255).astype(np.uint8)
        blob = prng.uniform01(9, 100 + seed, crop * crop).reshape(crop, crop)
        tgt = ((blob > 0.5) * 255).astype(np.uint8)
        ns = {"np": np, "rotate": rotate, "input_size_compute": functions.input_size_compute, "image": img, "target": tgt}
        np.random.seed(seed)
        exec(code, ns)
        out["%s_params" % tag] = np.array([crop, seed, ns["rot_deg"], ns["input_size"]])
        # strided samples + exact sums keep the fixture small (noise images do not compress)
        st = 4 if tag == "full" else 3
        out["%s_stride" % tag] = np.array(st)
        out["%s_img_sample" % tag] = ns["image"][::st, ::st].copy(); out["%s_tgt_sample" % tag] = ns["target"][::st, ::st].copy()
        out["%s_sums" % tag] = np.array([ns["image"].astype(np.int64).sum(), ns["target"].astype(np.int64).sum()])
        print(tag, crop, "rot_deg", ns["rot_deg"], ns["image"].shape, ns["image"].dtype)
    out["meta"] = np.array(repr(dict(numpy=np.__version__, scipy=__import__("scipy").__version__)))
    np.savez_compressed(os.path.join(HERE, "rotate_golden.npz"), **out)


if __name__ == "__main__":
    main()
