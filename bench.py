#!/usr/bin/env python3
"""bench.py — 572x572 tiles/s of the U-Net fwd+bwd step on N MI355X (BASELINE.json metric).

A "step" is one pass of the hot path over one batch of synthetic tiles (BASELINE configs[1]:
batch 8 per GPU, 572x572x1, fp32, 64-base-channel U-Net):
    zero_grad -> Unet.forward -> BCE-with-logits (unweighted) + argmax masks (one kernel) -> backward
    [-> RCCL gradient all-reduce, one bucket per backward stage, libunet_hip unet_dp_*] -> SGD(momentum)
Inputs are resident in HBM before the timed region.  One process per GPU.

`python3 bench.py --gpus N` starts its own ranks: when WORLD_SIZE is not in the environment and N > 1, this process —
before importing torch or touching a GPU — starts N child processes of itself (RANK / LOCAL_RANK / WORLD_SIZE /
MASTER_* set, rendezvous on 127.0.0.1, a free port), relays rank 0's line and returns the children's exit code; a rank that
does not come out of the RCCL bring-up within BENCH_DP_INIT_TIMEOUT seconds (default 120) says so and exits with code 3,
and the launcher then starts the ranks ONCE more with `--comm torch` (fresh processes).  Under
`python -m torch.distributed.run ... bench.py --gpus N` (WORLD_SIZE set) it is a rank.  `--dry-launch` prints the
child command lines and environment instead of starting them.

Rank 0 prints ONE short JSON line (< 2 KB: the contract fields, `roofline`, `cpu_baseline`, `bf16`, `comm` when
distributed) and writes everything else — per-kernel-family and per-layer tables (one row per SURVEY 8a row and direction),
the CPU baseline table, per-bucket all-reduce times, notes — to the side file named in the line's `detail` field
(default bench_detail.json next to this file, or under gpurun_out/ when that directory exists).

  value / dtype: the fp32 step (BASELINE configs[1]); never the bf16 figure.
  roofline     : the dominant kernel family of the step (default arithmetic: the fp32 Winograd 3x3 kernel
                 wino32_f32_kernel): achieved = FLOPs its launches EXECUTE on the matrix cores / their HIP-event time,
                 measured live in the timed region (that family only, in the last 2 timed steps: events around all
                 ~190 launches cost ~1 ms per step); frac = achieved / 157.3 TFLOP/s (<= 1)
  cpu_baseline : the torch restatement of the same step (oracle/torch_ref.py) timed on the host cores (BASELINE.md
                 section 4 plan), >= 5 timed iterations, tiles/s and GFLOP/s, plus the same-run parity of the GPU
                 logits against the CPU logits on the same weights and tile
  bf16         : after the fp32 timed region, the SAME step with bf16 tensors (unet_set_math(2): BASELINE configs[2]'s
                 per-GPU work; with --gpus 8 the configs[2] job itself), >= 10 timed steps bracketed like the main
                 region: ms_per_step, tiles_per_s, the implicit-GEMM family's executed fraction of the 2.5 PFLOP/s bf16
                 peak, and (one GPU) logits error / argmax flips against the CPU fp32 logits of the same tile

Other workloads (one GPU; `metric` then names the workload, the tables go to the side file as usual):
  --config 4   BASELINE configs[3]: batch 2 (the reference's batch size, main_main.py:136) of 512x512-shaped samples,
               a trainer.training()-shaped step at S = 700: data.augment (crop, reflect pad + rotation + centre crop,
               elastic deformation, label threshold, normalisation) -> forward -> centre crop -> class_balance weight
               map -> weighted BCE -> backward -> SGD -> argmax + IoU / pixel error of the first sample
  --config 5   BASELINE configs[4]: batch 16 of 1024x1024 images, mirror-pad + normalise to 1212x1212 (overlap-tile),
               Unet(base_ch=32) forward under no_grad, fused centre-crop + argmax + IoU / pixel-error counts
  --batch 2    the default step at the reference's batch size
"""
import argparse
import collections
import csv
import ctypes as C
import hashlib
import json
import os
import socket
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "dl-unet_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

# /opt/skills/guides/MI355X_MICROARCH.md: dense peaks
PEAK_TFLOPS = {0: 157.3, 3: 157.3, 1: 2500.0, 2: 2500.0}      # fp32 MFMA (= vector) / bf16 MFMA
PEAK_HBM_TBS = 8.0
HBM_STREAM_FRAC = 0.79               # 6.29 of 8 TB/s is what a streaming copy reaches (same guide): fractions above it are MALL-served
S = 572
B_PER_GPU = 8
GFLOP_PER_TILE_FWD = 300.86
GFLOP_PER_TILE_STEP = 902.2
KINDS = ("igemm", "wgrad", "wgrad_reduce", "wino", "stencil", "elementwise", "comm")
LINE_LIMIT = 2048                    # bytes; the driver keeps an 8 KB tail of stdout
EXIT_DP_INIT_TIMEOUT = 3             # a rank whose RCCL bring-up did not return in time

# profile row (csrc/net.hip LAYER_NAME) -> SURVEY 8a row
SURVEY_ROW = {"conv11c": "A1", "conv12c": "A2", "pool": "A3", "conv21c": "A4", "conv22c": "A5", "conv31c": "A6", "conv32c": "A7",
              "conv41c": "A8", "conv42c": "A9", "conv51c": "A10", "conv52c": "A11", "upconv": "A12", "conv41e": "A14",
              "conv42e": "A15", "conv31e": "A16", "conv32e": "A17", "conv21e": "A18", "conv22e": "A19", "conv11e": "A20",
              "conv12e": "A21", "finalconv": "A22", "L1": "L1", "L2": "L2", "L3": "L3", "allreduce": "8e"}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=None, help="tiles per GPU (default: 8; config 4: 2; config 5: 16)")
    ap.add_argument("--config", type=int, default=2, choices=(2, 4, 5),
                    help="workload: 2 = BASELINE configs[1] (default, the metric); 4 = configs[3] (B=2, S=700 training()-shaped step with "
                         "GPU augmentation); 5 = configs[4] (B=16, 1024^2 -> 1212^2 overlap-tile inference, 32-base-ch net)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--no-bf16", action="store_true", help="skip the bf16-tensor block that follows the fp32 timed region")
    ap.add_argument("--bf16-steps", type=int, default=10)
    ap.add_argument("--force-dist", action="store_true", help="initialise torch.distributed (nccl) and the RCCL communicator even for one rank")
    ap.add_argument("--comm", default="rccl", choices=("rccl", "torch"),
                    help="gradient all-reduce: the library's own RCCL communicator (default) or torch.distributed")
    ap.add_argument("--math", type=int, default=3, choices=(0, 1, 2, 3),
                    help="arithmetic of the dense contractions for the MAIN measurement: 3 fp32 MFMA with Winograd F(2x2,3x3) 3x3 "
                         "layers (default), 0 fp32 MFMA direct convolution, 1 bf16x3 split, 2 bf16 compute (include/unet_hip.h unet_set_math)")
    ap.add_argument("--other-modes", action="store_true", help="append short measurements of the other arithmetic modes (side file)")
    ap.add_argument("--dump-launches", default=None, help="write per-launch timings (CSV) of the instrumented pass here")
    ap.add_argument("--detail", default=None, help="side file for the per-kernel / per-layer tables (default: bench_detail.json)")
    ap.add_argument("--dry-launch", action="store_true", help="print the child command lines / environment of the N-rank launch and exit")
    ap.add_argument("--share-gpu", action="store_true",
                    help="rehearsal on a box with fewer GPUs than ranks: rank r uses device r %% visible devices, gloo rendezvous, "
                         "torch all-reduce (RCCL refuses two ranks on one device); never a scaling measurement")
    return ap.parse_args(argv)


# ------------------------------------------------------------------------------------------------------------------
# launcher: nothing below imports torch or touches HIP
# ------------------------------------------------------------------------------------------------------------------
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def child_specs(args, argv, port=None):
    """[(argv, env-additions)] of the N ranks of `bench.py --gpus N`."""
    port = port or _free_port()                 # never a fixed or inherited port: a stale listener would block the rendezvous
    rest = [a for a in argv if a != "--dry-launch"]
    specs = []
    for r in range(args.gpus):
        env = {"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(args.gpus), "LOCAL_WORLD_SIZE": str(args.gpus),
               "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "HSA_ENABLE_IPC_MODE_LEGACY": "0",
               "BENCH_SELF_LAUNCHED": "1"}
        specs.append(([sys.executable, os.path.abspath(__file__)] + rest, env))
    return specs


def _with_comm_torch(argv):
    """argv with the gradient all-reduce switched to torch.distributed (the relaunch after a failed RCCL bring-up)."""
    out, skip = [], False
    for a in argv:
        if skip:
            skip = False
            continue
        if a == "--comm":
            skip = True
            continue
        if a.startswith("--comm="):
            continue
        out.append(a)
    return out + ["--comm", "torch"]


def _run_children(specs):
    """Start the ranks as CHILD processes (never exec: this process may not be replaced once anything GPU-related is
    loaded, and it has loaded nothing), collect rank 0's JSON line, return (worst exit code, line | None, first failing code)."""
    procs = []
    for r, (cmd, env) in enumerate(specs):
        e = dict(os.environ)
        e.update(env)
        # rank 0's stdout carries the line; the other ranks' stdout (RCCL chatter) goes to our stderr
        procs.append(subprocess.Popen(cmd, env=e, stdout=subprocess.PIPE if r == 0 else sys.stderr, stderr=sys.stderr))
    # rank 0's stdout is read on a thread so that this loop can watch every rank: a rank that dies leaves the others
    # waiting in a collective (and rank 0's pipe open) for ever
    import threading
    got = []

    def reader():
        for raw in procs[0].stdout:
            txt = raw.decode("utf-8", "replace").rstrip("\n")
            if txt.startswith("{") and '"metric"' in txt:
                got.append(txt)
            elif txt:
                sys.stderr.write(txt + "\n")
    th = threading.Thread(target=reader, daemon=True)
    th.start()
    rc, first = 0, 0
    deadline = time.time() + float(os.environ.get("BENCH_LAUNCH_TIMEOUT", "1500"))
    pending = list(procs)
    while pending:
        for p in list(pending):
            code = p.poll()
            if code is None:
                continue
            pending.remove(p)
            if code != 0:
                first = first or code
                rc = rc or code
                for q in pending:           # stop exactly the processes started here
                    q.terminate()
        if pending:
            if time.time() > deadline:
                for q in pending:
                    q.kill()
                rc = rc or 124
                first = first or 124
            time.sleep(0.05)
    th.join(timeout=10)
    return rc, (got[-1] if got else None), first


def launch(args, argv):
    """bench.py --gpus N without WORLD_SIZE: run the ranks, relay rank 0's JSON line, return the worst exit code.  If a rank
    reports that the RCCL bring-up did not return (exit code 3, see run_rank's watchdog) the ranks are started once more, as
    fresh child processes, with the all-reduce on torch.distributed - so that the scaling run still measures."""
    specs = child_specs(args, argv)
    if args.dry_launch:
        print(json.dumps({"launch": [{"argv": a, "env": e} for a, e in specs],
                          "note": "each entry is started with subprocess.Popen(argv, env=os.environ + env); rank 0's stdout is relayed; "
                                  "exit code 3 from a rank (RCCL bring-up timed out) -> one relaunch with --comm torch"}))
        return 0
    rc, line, first = _run_children(specs)
    if first == EXIT_DP_INIT_TIMEOUT and args.comm == "rccl" and "--share-gpu" not in argv:
        sys.stderr.write("bench.py: a rank did not come out of unet_dp_init in time; starting the ranks again with --comm torch\n")
        rc, line, first = _run_children(child_specs(args, _with_comm_torch(argv)))
    if line is not None and rc == 0:
        sys.stdout.write(line + "\n")
        sys.stdout.flush()
    elif rc == 0:
        sys.stderr.write("bench.py: rank 0 printed no result line\n")
        rc = 1
    return rc


# ------------------------------------------------------------------------------------------------------------------
# the short line
# ------------------------------------------------------------------------------------------------------------------
_LINE_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline", "bf16", "comm", "summary", "effective_step_tflops", "kernel_time_sum_ms_per_step")
_ROOF_KEYS = ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "traffic_source", "alg_bytes_per_launch", "avg_launch_ms",
              "launches_per_step", "hbm_frac", "exec_gflop_per_launch", "share_of_step_time")
_CPU_KEYS = ("value", "unit", "cores", "kind", "sample", "gflops", "iters", "s_per_iter", "logits_parity")
_BF16_KEYS = ("ms_per_step", "tiles_per_s", "steps", "kernel", "frac", "peak", "logits_err", "argmax_flips", "px")


def _round(v, nd=6):
    if isinstance(v, float):
        return float("%.*g" % (nd, v))
    if isinstance(v, dict):
        return {k: _round(x, nd) for k, x in v.items()}
    if isinstance(v, list):
        return [_round(x, nd) for x in v]
    return v


def split_line(full, detail_path=None):
    """full result dict -> (short dict for stdout, everything for the side file)."""
    line = collections.OrderedDict()
    for k in _LINE_KEYS:
        if k in full:
            line[k] = full[k]
    if "roofline" in line:
        line["roofline"] = {k: line["roofline"][k] for k in _ROOF_KEYS if k in line["roofline"]}
    if "cpu_baseline" in line:
        line["cpu_baseline"] = {k: line["cpu_baseline"][k] for k in _CPU_KEYS if k in line["cpu_baseline"]}
    if "bf16" in line:
        line["bf16"] = _round({k: line["bf16"][k] for k in _BF16_KEYS if k in line["bf16"]}, 4)
    if isinstance(line.get("config"), dict):
        line["config"] = {k: v for k, v in line["config"].items() if k in ("workload", "arithmetic", "global_batch", "tile", "parallelism", "loss")}
    if detail_path:
        line["detail"] = detail_path
    line = _round(line)
    text = json.dumps(line)
    if len(text) >= LINE_LIMIT:            # never let free text push the numbers out of the driver's window
        for sect, key in (("roofline", "kernel"), ("roofline", "traffic_source"), ("cpu_baseline", "sample"), ("config", "arithmetic"),
                          ("config", "workload"), ("bf16", "kernel")):
            if sect in line and isinstance(line[sect].get(key), str):
                line[sect][key] = line[sect][key][:60]
        text = json.dumps(line)
    if len(text) >= LINE_LIMIT:
        for k in ("kernel_time_sum_ms_per_step", "effective_step_tflops", "summary"):
            line.pop(k, None)
    return line, full


def csrc_sha():
    """Fingerprint of the kernel sources (dl-unet_amd/csrc/*.hip|*.hpp): a PMC traffic figure is only quoted for the kernels
    it was measured on (tools/summarize_profiles.py stamps profiles/pmc_traffic.json with the same function)."""
    d = os.path.join(ROOT, "dl-unet_amd", "csrc")
    h = hashlib.sha256()
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".hpp")):
            h.update(name.encode())
            h.update(open(os.path.join(d, name), "rb").read())
    return h.hexdigest()[:16]


def pmc_traffic(math, family, B):
    """(bytes per launch | None, source note) from profiles/pmc_traffic.json, refused when its stamp is not this tree's kernels."""
    tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if not os.path.exists(tpath) or B != B_PER_GPU:
        return None, "not measured for this workload"
    t = json.load(open(tpath)).get("math%d" % math, {})
    v = t.get("%s_hbm_mb_per_launch" % family)
    if not v:
        return None, "no PMC pass for this family"
    if t.get("csrc_sha") != csrc_sha():
        return None, "stale: %s was measured on kernel sources %s, this tree is %s; rerun tools/profile_round.sh" % (t.get("source"), t.get("csrc_sha"), csrc_sha())
    return v * 1e6, "%s (rocprofv3 --pmc passes of this command, FETCH_SIZE x2 + WRITE_SIZE per launch; kernel sources %s = this tree)" % (t.get("source"), t.get("csrc_sha"))


def cpu_baseline(budget_s=30.0, gpu_checks=None):
    """The reference's CPU path (torch restatement, validated against the imported reference by
    tests/test_oracle_golden.py) on a bounded sample, BASELINE.md section 4: B=1 fwd+bwd+SGD (>= 5 timed iterations) and
    forward-only on the host's share of cores; B=2 and 8 threads as far as the budget allows.
    `value` = B=1 fwd+bwd+SGD on the host share.  gpu_checks: {name: (fn(params_np, x_np) -> GPU logits, bound | bound(ref_logits),
    argmax margin tolerance | None)} for the same-run parity ("f32": the path's 1e-3; "bf16": oracle/parity.py's storage-rounding
    model); errors and tolerances are fractions of the CPU logits' largest magnitude."""
    import numpy as np
    import torch
    from oracle import prng, torch_ref
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    # the GPU box gives one GPU's share of the host (16 cores); torch's default of one thread per visible core (128)
    # oversubscribes that share and runs slower
    share = max(1, min(16, avail))
    params_np = prng.make_params(0)
    t_start = time.perf_counter()
    table = []

    def run(threads, B, train, iters, min_iters=1):
        torch.set_num_threads(threads)
        p = torch_ref.params_to_torch(params_np, torch.float32, requires_grad=train)
        x = torch.from_numpy(prng.make_input(1, B, S))
        lab = prng.make_labels(3, B, S - 184)
        tgt = torch.from_numpy(np.concatenate([1 - lab, lab], axis=1).astype(np.float32))
        mom = {}

        def once(first=False):
            if train:
                torch_ref.train_step(p, mom, x, tgt, first=first)
            else:
                with torch.no_grad():
                    torch_ref.unet_forward(p, x)
        once(first=True)                                           # warm-up
        t0 = time.perf_counter(); n = 0
        while n < iters and (n < min_iters or time.perf_counter() - t_start < budget_s):
            once(); n += 1
        dt = time.perf_counter() - t0
        gf = (GFLOP_PER_TILE_STEP if train else GFLOP_PER_TILE_FWD) * B * n / dt
        table.append({"threads": threads, "batch": B, "what": "fwd+bwd+SGD" if train else "fwd (no_grad)", "iters": n,
                      "s_per_iter": dt / n, "tiles_per_s": B * n / dt, "gflops": gf})
        return table[-1]

    # same-run parity first: CPU fp32 logits of one tile vs the GPU's, same weights, same tile
    parity = {}
    if gpu_checks:
        torch.set_num_threads(share)
        x1 = prng.make_input(1, 1, S)
        with torch.no_grad():
            ref = torch_ref.unet_forward(torch_ref.params_to_torch(params_np), torch.from_numpy(x1)).numpy()
        scale = float(np.abs(ref).max())
        margin = np.abs(ref[:, 0] - ref[:, 1])
        for name, (fn, bound, safe_tol) in gpu_checks.items():
            got = fn(params_np, x1)
            if callable(bound):
                bound = bound(ref)
            err = float(np.abs(got - ref).max()) / scale
            same = (got[:, 1] > got[:, 0]) == (ref[:, 1] > ref[:, 0])
            # pixels whose CPU margin exceeds what the arithmetic lets the two logits move by (safe_tol None: twice the bound)
            safe = margin > (2 * bound if safe_tol is None else safe_tol) * scale
            parity[name] = {"max_abs_err_over_max_abs_ref": err, "bound": bound, "ok": bool(err <= bound and same[safe].all()),
                            "argmax_equal_px": int(same.sum()), "px": int(same.size), "argmax_equal_where_margin_gt_tol": bool(same[safe].all())}
    head = run(share, 1, True, 5, min_iters=5)
    run(share, 1, False, 5, min_iters=3)
    run(share, 2, True, 5)
    run(share, 2, False, 5)
    if share != 8 and avail >= 8:
        run(8, 1, True, 5)
        run(8, 1, False, 5)
    return {"value": head["tiles_per_s"], "unit": "tiles/s", "cores": share, "kind": "port",
            "sample": "%d timed B=1 572x572 fp32 fwd+bwd+SGD steps of the torch CPU restatement (oracle/torch_ref.py), %d threads" % (head["iters"], share),
            "gflops": head["gflops"], "iters": head["iters"], "s_per_iter": head["s_per_iter"], "logits_parity": parity.get("f32"),
            "parity_other": {k: v for k, v in parity.items() if k != "f32"},
            "cores_visible": avail, "table": table, "seconds": time.perf_counter() - t_start}


def layer_table(rows, steps, math):
    """rows: parsed unet_profile_dump lines of the instrumented pass -> per-row two-roof table."""
    peak_f = PEAK_TFLOPS[math] * 1e12
    agg = collections.OrderedDict()
    for r in rows:
        name = r["row"] or "(unscoped)"
        a = agg.setdefault(name, {"ms": 0.0, "launches": 0, "gflop": 0.0, "exec_gflop": 0.0, "mbytes": 0.0, "kernels": set()})
        a["ms"] += float(r["ms"]); a["launches"] += 1
        a["gflop"] += float(r["gflop"]); a["exec_gflop"] += float(r["exec_gflop"]); a["mbytes"] += float(r["mbytes"])
        a["kernels"].add(r["tag"].split(" ")[0].split("<")[0])
    out = []
    for name, a in agg.items():
        t = a["ms"] * 1e-3
        layer = name.split(".")[0]
        key = "pool" if layer.startswith("pool") else "upconv" if layer.startswith("upconv") else layer
        # fp32 element-wise work runs on the vector ALU, whose peak equals the fp32 MFMA peak
        flops = a["exec_gflop"] if a["exec_gflop"] > 0 else a["gflop"]
        t_f = flops * 1e9 / (peak_f if a["exec_gflop"] > 0 else PEAK_TFLOPS[0] * 1e12)
        t_b = a["mbytes"] * 1e6 / (PEAK_HBM_TBS * 1e12)
        hbm_frac = (t_b / t) if t > 0 else None
        bound = "mfma" if t_f >= t_b else "hbm"
        if bound == "hbm" and hbm_frac is not None and hbm_frac > HBM_STREAM_FRAC:
            bound = "hbm+mall"            # more than a streaming copy gets from HBM: part of the bytes came from the Infinity Cache
        out.append({"row": name, "survey": SURVEY_ROW.get(key), "kernels": sorted(a["kernels"]),
                    "ms_per_step": a["ms"] / steps, "launches_per_step": a["launches"] / steps,
                    "alg_gflop_per_step": a["gflop"] / steps, "alg_mb_per_step": a["mbytes"] / steps,
                    "mfma_frac": (t_f / t) if t > 0 else None, "hbm_frac": hbm_frac,
                    "bound": bound,
                    "frac": (max(t_f, t_b) / t) if t > 0 else None})
    return out


def comm_table(rows, steps):
    """per-bucket all-reduce time on the communicator stream and the exposed (non-overlapped) wait of the compute stream."""
    buckets, exposed = collections.OrderedDict(), 0.0
    for r in rows:
        if int(r["kind"]) != KINDS.index("comm"):
            continue
        if r["tag"].startswith("join"):
            exposed += float(r["ms"])
        else:
            b = buckets.setdefault(r["tag"], {"ms": 0.0, "n": 0, "mbytes_on_wire": 0.0})
            b["ms"] += float(r["ms"]); b["n"] += 1; b["mbytes_on_wire"] = float(r["mbytes"])
    return {"buckets": [{"bucket": k, "ms_per_step": v["ms"] / steps, "mb_on_wire": v["mbytes_on_wire"],
                         "gb_per_s": (v["mbytes_on_wire"] / 1e3) / (v["ms"] / v["n"] * 1e-3) if v["ms"] > 0 else None}
                        for k, v in buckets.items()],
            "allreduce_ms_per_step": sum(v["ms"] for v in buckets.values()) / steps,
            "exposed_ms_per_step": exposed / steps,
            "note": "bucket times are on the communicator stream (they include waiting for the slowest rank); exposed = how long the "
                    "compute stream stood at unet_dp_join before the optimizer"}


def weakest_rows(layers, n=3, min_ms=0.02):
    """the n rows furthest below the roof that binds them (rows of at least min_ms per step): the side file's to-do list"""
    rows = [r for r in layers if r.get("frac") is not None and r["ms_per_step"] >= min_ms]
    rows.sort(key=lambda r: r["frac"])
    return [{"row": r["row"], "ms": r["ms_per_step"], "bound": r["bound"], "frac": r["frac"]} for r in rows[:n]]


def _call_with_watchdog(fn, seconds, what):
    """Run fn() on a helper thread; if it has not returned after `seconds`, say which call hangs and end THIS process with
    EXIT_DP_INIT_TIMEOUT (a collective bring-up that blocks can only be abandoned together with its process: ctypes calls
    release the GIL, so the main thread is free to watch).  The launcher relaunches once with --comm torch."""
    import threading
    box = {}

    def target():
        try:
            box["value"] = fn()
        except BaseException as e:          # noqa: BLE001 - handed to the caller's thread
            box["error"] = e
    th = threading.Thread(target=target, daemon=True)
    th.start()
    th.join(seconds)
    if th.is_alive():
        sys.stderr.write("bench.py: rank %s: %s has not returned after %.0f s (another rank failed inside its own bring-up, or the "
                         "ranks cannot reach each other); giving up with exit code %d\n" % (os.environ.get("RANK", "0"), what, seconds, EXIT_DP_INIT_TIMEOUT))
        sys.stderr.flush()
        os._exit(EXIT_DP_INIT_TIMEOUT)
    if "error" in box:
        raise box["error"]
    return box.get("value")


def run_rank(args):
    # must be in the environment before anything initialises HIP (RCCL's IPC path reads it at init)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # kernel arguments in device memory instead of host memory: each of a step's ~190 (fp32) / ~250 (bf16 tensors) dependent
    # launches starts 1-2 us sooner (measured, same box, alternating: 33.58 -> 33.26 ms fp32, 10.90 -> 10.64 ms bf16).  A HIP
    # runtime flag, read when libamdhip64 is loaded - i.e. before `import torch`; INTEGRATION.md lists it for deployments.
    os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if "MASTER_PORT" not in os.environ:
        if world > 1:
            raise SystemExit("bench.py: WORLD_SIZE=%d without MASTER_PORT: start the ranks with `bench.py --gpus N` or torch.distributed.run" % world)
        os.environ["MASTER_PORT"] = str(_free_port())           # a lone rank (--force-dist) rendezvouses with itself: any free port
    if world > 1 or args.force_dist:
        os.environ.setdefault("NCCL_DEBUG", "VERSION")          # RCCL states its version once at communicator creation (stderr, below)
    # stdout carries exactly one JSON line: RCCL prints its warnings to stdout, so everything else written to fd 1 during the
    # run is sent to stderr and the line goes to the saved descriptor
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if args.config != 2 and world > 1:
        raise SystemExit("bench.py: --config %d is a one-GPU workload" % args.config)
    import torch
    try:
        # host-side torch ops (dataset glue) on the process's share of cores: torch's default of one thread per visible core
        # oversubscribes a GPU box's share and turns a 1 ms conversion into tens of ms
        torch.set_num_threads(max(1, min(16, len(os.sched_getaffinity(0)))))
    except AttributeError:
        pass
    ndev = torch.cuda.device_count()
    if args.share_gpu:
        local_rank = local_rank % max(ndev, 1)
    elif local_rank >= ndev:
        raise SystemExit("bench.py: rank %d needs device %d but only %d visible (use --share-gpu for a rehearsal)" % (rank, local_rank, ndev))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    import torch.distributed as dist
    use_dist = world > 1 or args.force_dist
    if args.share_gpu:
        args.comm = "torch"
    if use_dist:
        if args.share_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    import _hip
    import network
    import optim as hip_optim
    L = _hip.lib()

    _hip.check(L.unet_set_math(args.math), "unet_set_math")
    torch.manual_seed(0)                                   # same initial weights on every rank
    net = network.Unet(base_ch=32 if args.config == 5 else 64).to(dev)
    comm = None
    if use_dist:
        init_timeout = float(os.environ.get("BENCH_DP_INIT_TIMEOUT", "120"))
        try:
            _call_with_watchdog(lambda: net.enable_data_parallel(backend=args.comm), init_timeout,
                                "enable_data_parallel(backend=%r) [unet_dp_init / ncclCommInitRank]" % args.comm)
            comm = args.comm
        except RuntimeError as e:                           # a failed RCCL bring-up must not cost the scaling run: say so and use torch's
            if args.comm != "rccl":
                raise
            sys.stderr.write("bench.py: unet_dp_init failed (%s); falling back to torch.distributed all-reduce\n" % e)
            net.enable_data_parallel(backend="torch")
            comm = "torch (fallback)"
        if args.share_gpu:
            comm = "torch over gloo (shared-GPU rehearsal, not a scaling measurement)"
        if rank == 0:
            h0 = net._get_handle(local_rank)
            sys.stderr.write("bench.py: data parallel up: %d ranks, gradient all-reduce = %s, RCCL %s, communicator ranks %s, torch.distributed backend %s, "
                             "HSA_ENABLE_IPC_MODE_LEGACY=%s\n" % (world, comm, L.unet_dp_rccl_version(), L.unet_dp_world(h0.h) if comm == "rccl" else "-",
                                                                 dist.get_backend(), os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY")))
    opt = hip_optim.SGD(net.parameters(), lr=1e-4, momentum=0.99)

    # ---- the workload: step() is one pass of the hot path over one batch resident in HBM -------------------------------
    g = torch.Generator(device="cpu").manual_seed(1 + rank)          # each rank its own shard of the global batch
    training_step = True
    if args.config == 2:
        B, S_in = args.batch or B_PER_GPU, S
        x = torch.rand(B, 1, S_in, S_in, generator=g).to(dev)
        labels = (torch.rand(B, 1, S_in - 184, S_in - 184, generator=g) >= 0.5).long().to(dev)

        def step():
            opt.zero_grad(set_to_none=True)
            logits = net(x)
            loss, masks = hip_optim.bce_argmax_step(logits, labels)      # L1 + L2 in one pass: loss, its gradient, argmax masks
            loss.backward()
            opt.step()
            return masks, loss
        workload = "batch=%d/GPU 572x572x1 fwd+bwd+SGD %s, 64-base-ch U-Net (BASELINE %s)" % (
            B, "fp32" if args.math in (0, 3) else "bf16 compute", "configs[1]" if args.math in (0, 3) else "configs[2] per-GPU work")
        metric, loss_name = "572x572 tiles/sec fwd+bwd", "unweighted BCE-with-logits"
    elif args.config == 4:
        import data
        import trainer
        import numpy as np
        B, S_in, n = args.batch or 2, 700, 512
        yy, xx = torch.meshgrid(torch.arange(n), torch.arange(n), indexing="ij")

        def sample():                                                   # blobs = "cells" on a noisy background, grey levels 0..255
            mask = torch.zeros(n, n)
            for _ in range(12):
                cy, cx = torch.randint(40, n - 40, (2,), generator=g)
                r = torch.randint(15, 45, (1,), generator=g)
                mask = torch.maximum(mask, ((yy - cy) ** 2 + (xx - cx) ** 2 < r * r).float())
            img = torch.floor((0.3 + 0.5 * mask + 0.1 * torch.rand(n, n, generator=g)) * 255)
            return img.to(dev), (mask * 255).to(dev)
        raw = [sample() for _ in range(B)]
        raw_img, raw_tgt = torch.stack([r[0] for r in raw]), torch.stack([r[1] for r in raw])
        rs = np.random.RandomState(7)

        def step():
            # ImageDataset.__getitem__ after the file reads (data.py:97-135) for the batch's samples, on the device;
            # the rotation angle is drawn on the host like the reference's (np.arange(0, 360, 30)); the two uniform fields of the
            # elastic deformation are drawn on the device (host draws + upload: data.augment(random_state=...), 3 ms per field)
            angles = [float(rs.choice(np.arange(0, 360, 30))) for _ in range(B)]
            images, labels = data.augment(raw_img, raw_tgt, [(0, 0)] * B, n, angles, 200.0, 10.0)      # the batch's samples in one call
            opt.zero_grad(set_to_none=True)
            preds, loss, labels = trainer._step_loss(net, images, labels, dev, True)      # trainer.py:58-75
            loss.backward()
            opt.step()
            trainer._first_sample_metrics(preds, labels)                                  # trainer.py:82-89 (host sync, as there)
            return None, loss
        workload = ("batch=%d 512x512-shaped samples -> 700x700 inputs, training()-shaped step with GPU augmentation "
                    "(rotation + elastic + class_balance), fp32, 64-base-ch U-Net (BASELINE configs[3])" % B)
        metric, loss_name = "700x700 tiles/sec fwd+bwd incl. augmentation", "class-balance weighted BCE-with-logits (trainer.py:72-75)"
    else:
        import data
        B, S_in, n = args.batch or 16, 1212, 1024
        imgs = (torch.rand(B, n, n, generator=g) * 255).to(dev)
        labels = (torch.rand(B, 1, n, n, generator=g) >= 0.5).long().to(dev)
        training_step = False

        def step():
            with torch.no_grad():
                xin = data.test_input(imgs)                                               # data.py:184-188 mirror pad + normalise
                y = net(xin)
                mask, stats = hip_optim.crop_argmax_metrics(y, labels)                    # tester.py:29-42
            return mask, stats
        workload = ("batch=%d 1024x1024 images -> 1212x1212 overlap-tile inputs, forward + crop/argmax/IoU/pixel-error counts, fp32, "
                    "32-base-ch U-Net (BASELINE configs[4])" % B)
        metric, loss_name = "1212x1212 tiles/sec forward (overlap-tile inference)", "none (inference)"

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    def read_families():
        ms = C.c_double(); n_ = C.c_long(); fl = C.c_double(); ex = C.c_double(); by = C.c_double()
        fam = {}
        for k, name in enumerate(KINDS):
            _hip.check(L.unet_profile_read(k, C.byref(ms), C.byref(n_), C.byref(fl), C.byref(ex), C.byref(by)))
            fam[name] = (ms.value, n_.value, fl.value, ex.value, by.value)
        return fam

    timing = not args.no_kernel_timing

    def timed_region(steps, warmup, dom):
        """`warmup` untimed steps, then exactly `steps` steps between barrier + synchronize on both sides; MAX over ranks.
        The dominant kernel family `dom` carries HIP events on its launch stream INSIDE the region - that family only, and in
        the last `nsample` steps only: a pair of events costs ~5 us of stream time, i.e. 0.9 ms per fp32 step (1.2 ms with bf16
        tensors) around all ~190 launches and still 0.4 ms around the 42 Winograd launches (measured: 33.89 / 33.46 / 33.03 ms
        per step with all / this family's / no events), all of which would be charged to the result."""
        for _ in range(warmup):
            step()
        barrier()
        nsample = min(2, steps)
        if timing:
            L.unet_profile_reset(); L.unet_profile_select(1 << KINDS.index(dom))
        t0 = time.perf_counter()
        res = None
        for i in range(steps):
            if timing and i == steps - nsample:
                L.unet_profile_enable(1)                       # host-side switch: no synchronisation
            res = step()
        barrier()
        dt = time.perf_counter() - t0
        stats = None
        if timing:
            L.unet_profile_enable(0)
            stats = read_families()[dom]
        tmax = torch.tensor([dt], dtype=torch.float64, device="cpu" if args.share_gpu else dev)
        if use_dist:
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        return tmax.item(), stats, nsample, res

    dom = "wino" if args.math == 3 and args.config != 5 else "igemm"
    dt, dom_stats, nsample, (masks, loss) = timed_region(args.steps, args.warmup, dom)

    psteps = 0
    if timing:                                             # every rank: the steps contain the collectives
        psteps = max(1, min(3, args.steps))
        L.unet_profile_reset(); L.unet_profile_select(0xFFFFFFFF); L.unet_profile_enable(1)
        for _ in range(psteps):
            step()
        barrier()
        L.unet_profile_enable(0)

    out = None
    if rank == 0:
        tiles = B * world * args.steps
        h = net._get_handle(local_rank)
        flops_step = h.flops(B, S_in, training_step)
        arith = {0: "fp32 MFMA, direct convolution", 1: "bf16x3 split", 2: "bf16 tensors and MFMA, fp32 accumulate / master weights",
                 3: "fp32 MFMA; 3x3 fwd/dgrad/wgrad as Winograd F(2x2,3x3)"}[args.math]
        out = {
            "metric": metric, "value": tiles / dt, "unit": "tiles/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": {0: "f32", 1: "bf16x3", 2: "bf16", 3: "f32"}[args.math],
            "data": "synthetic",
            "config": {"workload": workload, "arithmetic": arith, "global_batch": B * world, "tile": S_in, "parallelism": "dp%d" % world,
                       "loss": loss_name, "final_loss": float(loss.item()) if training_step else None},
            "effective_step_tflops": flops_step / (dt / args.steps) / 1e12,
        }
        if use_dist:
            ranks = int(L.unet_dp_world(h.h)) if comm == "rccl" else world
            if ranks != args.gpus:
                raise SystemExit("bench.py: the communicator has %d ranks but --gpus is %d" % (ranks, args.gpus))
            out["comm"] = {"gradient_allreduce": comm, "ranks": ranks,
                           "rccl_version": int(L.unet_dp_rccl_version()), "message_mb": 124.1, "buckets": 6,
                           "self_launched": os.environ.get("BENCH_SELF_LAUNCHED") == "1"}
        if timing:
            fam = read_families()                           # the instrumented pass (psteps steps)
            ms0, n0, fl0, ex0, by0 = dom_stats              # the dominant family, in the timed region
            peak = PEAK_TFLOPS[args.math]
            ach = ex0 / (ms0 * 1e-3) / 1e12 if ms0 > 0 else 0.0
            traffic, tsrc = pmc_traffic(args.math, dom, B) if args.config == 2 else (None, "not measured for this workload")
            kname = {"wino": "wino32_f32_kernel (3x3 conv fwd/dgrad, Winograd F(2x2,3x3), fp32 MFMA)",
                     "igemm": "igemm kernels (conv fwd / dgrad / up-conv implicit GEMM)",
                     "wgrad": "weight-gradient kernels"}[dom]
            out["roofline"] = {"bound": "mfma", "kernel": kname,
                               "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak,
                               "traffic": traffic, "traffic_source": tsrc,
                               "what": "achieved = FLOPs executed on the matrix cores (Winograd: 16 multiplies per tile, channel pair and xi "
                                       "instead of 36) / HIP-event time of its launches in the last `sampled_steps` steps of "
                                       "the timed region (only this family carries events there; `kernels` and `layers` come from a separate fully "
                                       "instrumented pass)",
                               "effective_tflops": fl0 / (ms0 * 1e-3) / 1e12 if ms0 > 0 else 0.0,
                               "effective_note": "direct-convolution (algorithmic, SURVEY 8d) FLOPs / time; may exceed the peak because F(2x2,3x3) skips 20/36 of them",
                               "hbm_frac": by0 / (ms0 * 1e-3) / (PEAK_HBM_TBS * 1e12) if ms0 > 0 else 0.0,
                               "alg_bytes_per_launch": by0 / max(n0, 1),
                               "launches_per_step": n0 / nsample, "avg_launch_ms": ms0 / max(n0, 1),
                               "sampled_steps": nsample,
                               "exec_gflop_per_launch": ex0 / max(n0, 1) / 1e9, "alg_gflop_per_launch": fl0 / max(n0, 1) / 1e9,
                               "share_of_step_time": (ms0 / nsample) / (dt / args.steps * 1e3)}
            out["kernels"] = {k: {"ms_per_step": v[0] / psteps, "launches_per_step": v[1] / psteps,
                                  "exec_tflops": (v[3] / (v[0] * 1e-3) / 1e12 if v[0] > 0 and v[3] > 0 else None),
                                  "effective_tflops": (v[2] / (v[0] * 1e-3) / 1e12 if v[0] > 0 and v[2] > 0 else None),
                                  "alg_tb_per_s": (v[4] / (v[0] * 1e-3) / 1e12 if v[0] > 0 and v[4] > 0 else None)}
                              for k, v in fam.items() if v[1] > 0}
            path = args.dump_launches
            tmp = None
            if not path:
                tmp = tempfile.NamedTemporaryFile(suffix=".csv", delete=False)
                tmp.close(); path = tmp.name
            _hip.check(L.unet_profile_dump(path.encode()))
            rows = list(csv.DictReader(open(path)))
            if tmp:
                os.unlink(path)
            out["layers"] = layer_table([r for r in rows if int(r["kind"]) != KINDS.index("comm")], psteps, args.math)
            out["weakest_rows"] = weakest_rows(out["layers"], 8)
            if use_dist:
                out["comm_detail"] = comm_table(rows, psteps)
                out["comm"]["allreduce_ms_per_step"] = out["comm_detail"]["allreduce_ms_per_step"]
                out["comm"]["exposed_ms_per_step"] = out["comm_detail"]["exposed_ms_per_step"]
            out["instrumented_pass"] = {"steps": psteps, "what": "after the timed region: every launch bracketed by HIP events on its stream; "
                                        "source of `kernels`, `layers` and --dump-launches (the events themselves cost ~1 ms per step, so this pass "
                                        "is not the one `value` is taken from)"}
            out["layers_note"] = ("per SURVEY 8a row: mfma_frac = executed matrix-core FLOPs / time / %.1f TFLOP/s (element-wise rows: their "
                                  "FLOPs against the equal fp32 vector peak), hbm_frac = algorithmic bytes / time / 8 TB/s, bound = the roof "
                                  "that would take longer at peak ('hbm+mall': more than the %.2f of 8 TB/s a streaming copy reaches - part of "
                                  "the bytes was served by the Infinity Cache, so it is not a fraction of HBM alone), frac = that roof's fraction"
                                  % (PEAK_TFLOPS[args.math], HBM_STREAM_FRAC))
            out["kernel_time_sum_ms_per_step"] = sum(v[0] for k, v in fam.items() if k != "comm") / psteps
            if args.config != 2:
                kt = out["kernel_time_sum_ms_per_step"]
                out["summary"] = {"tiles_per_s": out["value"], "ms_per_step": out["ms_per_step"], "kernel_ms_per_step": kt,
                                  "host_gap_ms_per_step": out["ms_per_step"] - kt,
                                  "weakest": out["weakest_rows"][:1]}

    # ---- the same step with bf16 tensors (BASELINE configs[2]): its own warm-up, >= 10 timed steps, its own family's roofline -------
    bf16_on = args.config == 2 and args.math == 3 and not args.no_bf16 and args.bf16_steps > 0
    if bf16_on:
        _hip.check(L.unet_set_math(2), "unet_set_math")
        nb = max(10, args.bf16_steps)
        dtb, st_b, ns_b, _ = timed_region(nb, 3, "igemm")
        _hip.check(L.unet_set_math(args.math), "unet_set_math")
        if rank == 0:
            blk = {"ms_per_step": dtb / nb * 1e3, "tiles_per_s": B * world * nb / dtb, "steps": nb, "warmup": 3,
                   "what": "the step above with bf16 tensors (unet_set_math(2): bf16 activations / activation gradients / packed filters, "
                           "fp32 accumulation, parameters, gradients and all-reduce), timed like the main region; not part of `value`.  With bf16 "
                           "tensors the weight gradients run on the handle's auxiliary stream next to the dgrad chain (unet_set_overlap default); "
                           "`frac` comes from the region's last two steps, which carry per-launch events and therefore run on one stream"}
            if st_b is not None and st_b[0] > 0:
                blk.update({"kernel": "igemmb3+igemmb+convb64", "frac": st_b[3] / (st_b[0] * 1e-3) / 1e12 / PEAK_TFLOPS[2], "peak": PEAK_TFLOPS[2],
                            "avg_launch_ms": st_b[0] / max(st_b[1], 1), "launches_per_step": st_b[1] / ns_b,
                            "share_of_step_time": (st_b[0] / ns_b) / (dtb / nb * 1e3)})
            out["bf16"] = blk

    if rank == 0 and world == 1 and not args.no_cpu_baseline and args.config == 2:
        def gpu_logits_in(mode):
            def fn(params_np, x_np):
                # the same weights and tile through the HIP path, in the given arithmetic mode
                _hip.check(L.unet_set_math(mode), "unet_set_math")
                try:
                    chk = network.Unet()
                    chk.load_state_dict({k: torch.from_numpy(v) for k, v in params_np.items()})
                    chk = chk.to(dev)
                    with torch.no_grad():
                        return chk(torch.from_numpy(x_np).to(dev)).cpu().numpy()
                finally:
                    _hip.check(L.unet_set_math(args.math), "unet_set_math")
            return fn
        from oracle import parity as _parity
        # fp32 arithmetic: the north_star's 1e-3.  bf16 tensors cannot meet an fp32 tolerance; their bound is the storage-rounding
        # model of oracle/parity.py (tests/test_bf16_gpu.py holds the path to it at this tile), here as a fraction of |y|max
        def bf16_bound(ref):
            return _parity.bf16_max_err(ref.size, float((ref.astype("float64") ** 2).mean() ** 0.5)) / float(abs(ref).max())
        # fp32: argmax must agree wherever the CPU margin exceeds the forward tolerance of the parity tests (2e-5)
        checks = {"bf16": (gpu_logits_in(2), bf16_bound, None)} if args.math == 2 else {"f32": (gpu_logits_in(args.math), 1e-3, 2e-5)}
        if bf16_on:
            checks["bf16"] = (gpu_logits_in(2), bf16_bound, None)
        cb = cpu_baseline(gpu_checks=checks)
        if args.math == 2:
            cb["logits_parity"] = cb["parity_other"].pop("bf16")
        out["cpu_baseline"] = cb
        pb = cb["parity_other"].get("bf16")
        if bf16_on and pb:
            out["bf16"].update({"logits_err": pb["max_abs_err_over_max_abs_ref"], "argmax_flips": pb["px"] - pb["argmax_equal_px"], "px": pb["px"],
                                "logits_bound": pb["bound"], "parity_ok": pb["ok"],
                                "parity_note": "against the CPU fp32 logits of the same weights and tile; the north_star's 1e-3 / bit-exact argmax "
                                               "is an fp32 tolerance that 8-bit significands do not meet (DESIGN section 2)"})
    # informational: the same step in the other arithmetic modes (not part of `value`)
    if args.other_modes and args.config == 2:
        other = {}
        for m, name in ((0, "f32_direct"), (3, "f32_winograd_3x3"), (1, "bf16x3_split_fp32_accumulate"), (2, "bf16_compute_fp32_accumulate")):
            if m == args.math:
                continue
            _hip.check(L.unet_set_math(m), "unet_set_math")
            for _ in range(2):
                step()
            barrier()
            t0 = time.perf_counter()
            for _ in range(5):
                step()
            barrier()
            tm = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device="cpu" if args.share_gpu else dev)
            if use_dist:
                dist.all_reduce(tm, op=dist.ReduceOp.MAX)
            other[name] = {"tiles_per_s": B * world * 5 / tm.item(), "ms_per_step": tm.item() / 5 * 1e3}
        _hip.check(L.unet_set_math(args.math), "unet_set_math")
        if rank == 0:
            out["other_math_modes"] = other
    if use_dist:
        dist.destroy_process_group()
    sys.stdout.flush()
    if rank == 0:
        dpath = args.detail
        if not dpath:
            gdir = os.path.join(ROOT, "gpurun_out")
            dpath = os.path.join(gdir if os.path.isdir(gdir) else ROOT, "bench_detail.json")
        line, detail = split_line(out, os.path.relpath(dpath, ROOT) if dpath.startswith(ROOT) else dpath)
        try:
            with open(dpath, "w") as f:
                json.dump(detail, f, indent=1)
        except OSError as e:
            sys.stderr.write("bench.py: cannot write %s (%s); the tables follow on stderr\n" % (dpath, e))
            line.pop("detail", None)
        # a compact per-layer table for whoever reads the log
        for r in out.get("layers", []):
            sys.stderr.write("%-16s %-4s %7.3f ms  mfma %.3f  hbm %.3f  %s\n" % (r["row"], r["survey"] or "", r["ms_per_step"], r["mfma_frac"] or 0.0,
                                                                               r["hbm_frac"] or 0.0, r["bound"]))
        sys.stderr.flush()
        os.write(json_fd, (json.dumps(line) + "\n").encode())


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse_args(argv)
    if args.dry_launch or ("WORLD_SIZE" not in os.environ and args.gpus > 1):
        return launch(args, argv)
    run_rank(args)
    return 0


if __name__ == "__main__":
    sys.exit(main())
