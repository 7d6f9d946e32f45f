"""Generates tests/golden/trainer_golden.json by RUNNING the reference's trainer.training() (this container only).

Run:  python tests/golden/make_golden_trainer.py        (needs /root/reference; ~1 min on CPU)

What is pinned: the behaviour of `training()` that depends on how the DATASET string reaches it
(trainer.py:18-27, :68, :185-214 compare it with `is` against literals):
  * 'ISBI2012' passed as a literal from another module IS the interned constant -> the stop goal is armed
    (when_to_stop = 1, goal = 0.0611, tested against the validation IoU), the goal checkpoint
    models/unet_weight_save_ISBI2012.pth is written when it is passed, and the epoch's "latest" checkpoint /
    LR-floor logic is skipped while the goal is armed;
  * the same text built at run time (what sys.argv delivers) is a different object -> no goal;
  * 'DIC-C2DH-HeLa' is not identifier-like, so CPython does not intern the constant: even a literal from another
    module is a different object -> no goal, class_balance (not weighted_map) is used.
Recorded per case: files written, the epochs in which "The goal was reached" was printed, and the six progress
series.  Inputs and weights come from oracle/prng.py; nothing of the reference's text is stored.

Stand-ins for modules that are absent here and unused on this path: batchgenerators (maybe_mkdir_p ==
os.makedirs(exist_ok=True), trainer.py:4), cv2 (functions.py:1, only weighted_map), torchvision (network.py:5-6).
"""
import contextlib
import io
import json
import os
import sys
import tempfile
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import prng  # noqa: E402

REF = "/root/reference"
S, SO = 188, 4


def import_reference():
    for name in ("torchvision", "torchvision.transforms", "cv2", "batchgenerators", "batchgenerators.utilities",
                 "batchgenerators.utilities.file_and_folder_operations"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    sys.modules["batchgenerators.utilities.file_and_folder_operations"].maybe_mkdir_p = lambda p: os.makedirs(p, exist_ok=True)
    sys.path.insert(0, REF)
    import network  # noqa
    import trainer  # noqa
    return network, trainer


def loaders():
    def mk(seed, n):
        return [(torch.from_numpy(prng.make_input(seed + i, 2, S)), torch.from_numpy(prng.make_labels(seed + i, 2, SO))) for i in range(n)]
    return mk(10, 2), mk(20, 1)


def run_case(network, trainer, dataset):
    torch.manual_seed(0)
    torch.set_num_threads(8)
    net = network.Unet()
    net.load_state_dict({k: torch.from_numpy(v) for k, v in prng.make_params(0).items()})
    tr, va = loaders()
    out = io.StringIO()
    with tempfile.TemporaryDirectory() as d:
        with contextlib.redirect_stdout(out):
            trainer.training(net, tr, va, 1, 2, torch.device("cpu"), d, dataset)
        files = sorted(os.path.relpath(os.path.join(r, f), d) for r, _, fs in os.walk(d) for f in fs)
        prog = {f[:-4]: np.atleast_1d(np.loadtxt(os.path.join(d, "progress", f))).tolist()
                for f in sorted(os.listdir(os.path.join(d, "progress")))}
    lines = out.getvalue().splitlines()
    goal = [ln for ln in lines if ln.startswith("The goal was reached")]
    return {"files": files, "goal_lines": goal, "progress": prog,
            "n_model_saved_lines": sum(ln == "Model has been saved:" for ln in lines)}


def run_series(network, trainer, S_, threads, dtype=torch.float32):
    """One run of the reference's training() (2 epochs of 2 train batches + 1 validation batch of 2) at tile size S_ on
    `threads` CPU threads: the six progress series."""
    torch.manual_seed(0)
    torch.set_num_threads(threads)
    net = network.Unet().to(dtype)
    net.load_state_dict({k: torch.from_numpy(v).to(dtype) for k, v in prng.make_params(0).items()})

    def mk(seed, n):
        return [(torch.from_numpy(prng.make_input(seed + i, 2, S_)).to(dtype), torch.from_numpy(prng.make_labels(seed + i, 2, S_ - 184))) for i in range(n)]
    out = io.StringIO()
    with tempfile.TemporaryDirectory() as d:
        with contextlib.redirect_stdout(out):
            trainer.training(net, mk(10, 2), mk(20, 1), 1, 2, torch.device("cpu"), d, "".join(["ISBI", "2012"]))
        return {f[:-4]: np.atleast_1d(np.loadtxt(os.path.join(d, "progress", f))).tolist()
                for f in sorted(os.listdir(os.path.join(d, "progress")))}


def main_series():
    """tests/golden/trainer_series.json: the six progress series of the reference's training() at S=188 (4x4 masks) and S=220
    (36x36 masks), each run in fp32 on 1, 2, 4 and 8 CPU threads and once in fp64 (model and images converted; the loop is
    the reference's own).  Per series: `f32` (8 threads), `f64`, and `thread_spread` = max - min over the thread counts.
    What they show: oneDNN's result hardly depends on the thread count (spread <= 5e-5 relative), so that is no
    measure of what fp32 evaluation order does to the trajectory; the distance |f32 - f64| is: 1.4 % on the validation loss
    after 4 SGD steps at S=188 (one pixel of the 4x4 mask flips with it) and 5e-6 at S=220.  A HIP run is therefore held to
    the fp64 series within twice the reference's own fp32 distance from it (tests/test_net_gpu.py).
    Run:  python tests/golden/make_golden_trainer.py series"""
    network, trainer = import_reference()
    out = {}
    for S_ in (188, 220):
        runs = {t: run_series(network, trainer, S_, t) for t in (1, 2, 4, 8)}
        r64 = run_series(network, trainer, S_, 8, torch.float64)
        series = {}
        for name in runs[8]:
            vals = np.array([runs[t][name] for t in (1, 2, 4, 8)])
            series[name] = {"f32": runs[8][name], "f64": r64[name], "thread_spread": (vals.max(axis=0) - vals.min(axis=0)).tolist()}
            print(S_, name, "f32", runs[8][name], "f64", r64[name], "thread spread", series[name]["thread_spread"])
        out["S%d" % S_] = series
    meta = {"torch": torch.__version__, "threads": [1, 2, 4, 8], "sizes": [188, 220], "epochs_arg": 1, "batch": 2,
            "train_seeds": [10, 11], "val_seeds": [20], "weights_seed": 0, "dataset_arg": "run-time string 'ISBI2012' (no stop goal)"}
    with open(os.path.join(HERE, "trainer_series.json"), "w") as f:
        json.dump({"meta": meta, "sizes": out}, f, indent=1)


def main():
    network, trainer = import_reference()
    runtime_name = "".join(["ISBI", "2012"])            # equal text, different object (what argv gives)
    assert runtime_name == "ISBI2012" and runtime_name is not "ISBI2012"  # noqa: F632
    cases = {
        "literal_ISBI2012": run_case(network, trainer, "ISBI2012"),
        "runtime_ISBI2012": run_case(network, trainer, runtime_name),
        "literal_DIC-C2DH-HeLa": run_case(network, trainer, "DIC-C2DH-HeLa"),
    }
    meta = {"torch": torch.__version__, "threads": 8, "S": S, "epochs_arg": 1, "batch": 2,
            "train_seeds": [10, 11], "val_seeds": [20], "weights_seed": 0}
    with open(os.path.join(HERE, "trainer_golden.json"), "w") as f:
        json.dump({"meta": meta, "cases": cases}, f, indent=1)
    for k, v in cases.items():
        print(k, v["files"], v["goal_lines"], v["progress"]["loss"])


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "series":
        main_series()
    else:
        main()
