#!/usr/bin/env python3
"""bench.py — 572x572 tiles/s of the U-Net fwd+bwd step on N MI355X (BASELINE.json metric).

A "step" is one pass of the hot path over one batch of synthetic tiles (BASELINE configs[1]:
batch 8 per GPU, 572x572x1, fp32, 64-base-channel U-Net):
    zero_grad -> Unet.forward -> BCE-with-logits (unweighted) + argmax masks (one kernel) -> backward
    [-> RCCL gradient all-reduce, one bucket per backward stage, libunet_hip unet_dp_*] -> SGD(momentum)
Inputs are resident in HBM before the timed region.  One process per GPU.

`python3 bench.py --gpus N` starts its own ranks: when WORLD_SIZE is not in the environment and N > 1, this process —
before importing torch or touching a GPU — starts N child processes of itself (RANK / LOCAL_RANK / WORLD_SIZE /
MASTER_* set, rendezvous on 127.0.0.1), relays rank 0's line and returns the children's exit code.  Under
`python -m torch.distributed.run ... bench.py --gpus N` (WORLD_SIZE set) it is a rank.  `--dry-launch` prints the
child command lines and environment instead of starting them.

Rank 0 prints ONE short JSON line (< 2 KB: the contract fields, `roofline`, `cpu_baseline`, `comm` when distributed)
and writes everything else — per-kernel-family and per-layer tables (one row per SURVEY 8a row and direction), the CPU
baseline table, per-bucket all-reduce times, notes — to the side file named in the line's `detail` field
(default bench_detail.json next to this file, or under gpurun_out/ when that directory exists).

  roofline     : the dominant kernel family of the step (default arithmetic: the fp32 Winograd 3x3 kernel
                 wino32_f32_kernel): achieved = FLOPs its launches EXECUTE on the matrix cores / their HIP-event time,
                 measured live in the timed region (that family only, in the last 2 timed steps: events around all
                 ~190 launches cost ~1 ms per step); frac = achieved / 157.3 TFLOP/s (<= 1)
  cpu_baseline : the torch restatement of the same step (oracle/torch_ref.py) timed on the host cores (BASELINE.md
                 section 4 plan), >= 5 timed iterations, tiles/s and GFLOP/s, plus the same-run parity of the GPU
                 logits against the CPU logits on the same weights and tile
"""
import argparse
import collections
import csv
import ctypes as C
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
S = 572
B_PER_GPU = 8
GFLOP_PER_TILE_FWD = 300.86
GFLOP_PER_TILE_STEP = 902.2
KINDS = ("igemm", "wgrad", "wgrad_reduce", "wino", "stencil", "elementwise", "comm")
LINE_LIMIT = 2048                    # bytes; the driver keeps an 8 KB tail of stdout

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
    ap.add_argument("--batch", type=int, default=B_PER_GPU, help="tiles per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
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
    port = port or int(os.environ.get("MASTER_PORT", "0")) or _free_port()
    rest = [a for a in argv if a != "--dry-launch"]
    specs = []
    for r in range(args.gpus):
        env = {"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(args.gpus), "LOCAL_WORLD_SIZE": str(args.gpus),
               "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "HSA_ENABLE_IPC_MODE_LEGACY": "0",
               "BENCH_SELF_LAUNCHED": "1"}
        specs.append(([sys.executable, os.path.abspath(__file__)] + rest, env))
    return specs


def launch(args, argv):
    """Start the ranks as CHILD processes (never exec: this process may not be replaced once anything GPU-related is
    loaded, and it has loaded nothing), relay rank 0's JSON line, return the worst exit code."""
    specs = child_specs(args, argv)
    if args.dry_launch:
        print(json.dumps({"launch": [{"argv": a, "env": e} for a, e in specs],
                          "note": "each entry is started with subprocess.Popen(argv, env=os.environ + env); rank 0's stdout is relayed"}))
        return 0
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
    rc = 0
    deadline = time.time() + float(os.environ.get("BENCH_LAUNCH_TIMEOUT", "1500"))
    pending = list(procs)
    while pending:
        for p in list(pending):
            code = p.poll()
            if code is None:
                continue
            pending.remove(p)
            if code != 0:
                rc = rc or code
                for q in pending:           # stop exactly the processes started here
                    q.terminate()
        if pending:
            if time.time() > deadline:
                for q in pending:
                    q.kill()
                rc = rc or 124
            time.sleep(0.05)
    th.join(timeout=10)
    line = got[-1] if got else None
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
              "dtype", "data", "config", "roofline", "cpu_baseline", "comm", "effective_step_tflops", "kernel_time_sum_ms_per_step")
_ROOF_KEYS = ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "traffic_source", "alg_bytes_per_launch", "avg_launch_ms",
              "launches_per_step", "hbm_frac", "exec_gflop_per_launch", "share_of_step_time")
_CPU_KEYS = ("value", "unit", "cores", "kind", "sample", "gflops", "iters", "s_per_iter", "logits_parity")


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
    if isinstance(line.get("config"), dict):
        line["config"] = {k: v for k, v in line["config"].items() if k in ("workload", "arithmetic", "global_batch", "tile", "parallelism", "loss")}
    if detail_path:
        line["detail"] = detail_path
    line = _round(line)
    text = json.dumps(line)
    if len(text) >= LINE_LIMIT:            # never let free text push the numbers out of the driver's window
        for sect, key in (("roofline", "kernel"), ("cpu_baseline", "sample"), ("config", "arithmetic"), ("config", "workload")):
            if sect in line and isinstance(line[sect].get(key), str):
                line[sect][key] = line[sect][key][:80]
        text = json.dumps(line)
    return line, full


def cpu_baseline(budget_s=30.0, gpu_check=None, parity_bound=1e-3):
    """The reference's CPU path (torch restatement, validated against the imported reference by
    tests/test_oracle_golden.py) on a bounded sample, BASELINE.md section 4: B=1 fwd+bwd+SGD (>= 5 timed iterations) and
    forward-only on the host's share of cores; B=2 and 8 threads as far as the budget allows.
    `value` = B=1 fwd+bwd+SGD on the host share.  gpu_check(params_np, x_np) -> GPU logits for the same-run parity."""
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
    parity = None
    if gpu_check is not None:
        torch.set_num_threads(share)
        x1 = prng.make_input(1, 1, S)
        with torch.no_grad():
            ref = torch_ref.unet_forward(torch_ref.params_to_torch(params_np), torch.from_numpy(x1)).numpy()
        got = gpu_check(params_np, x1)
        scale = float(np.abs(ref).max())
        err = float(np.abs(got - ref).max()) / scale
        margin = np.abs(ref[:, 0] - ref[:, 1])
        safe = margin > 2e-5 * scale                                 # pixels whose CPU margin exceeds the forward tolerance
        same = (got[:, 1] > got[:, 0]) == (ref[:, 1] > ref[:, 0])
        safe = margin > 2 * parity_bound * scale if parity_bound > 1e-3 else safe
        parity = {"max_abs_err_over_max_abs_ref": err, "bound": parity_bound, "ok": bool(err <= parity_bound and same[safe].all()),
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
            "gflops": head["gflops"], "iters": head["iters"], "s_per_iter": head["s_per_iter"], "logits_parity": parity,
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
        out.append({"row": name, "survey": SURVEY_ROW.get(key), "kernels": sorted(a["kernels"]),
                    "ms_per_step": a["ms"] / steps, "launches_per_step": a["launches"] / steps,
                    "alg_gflop_per_step": a["gflop"] / steps, "alg_mb_per_step": a["mbytes"] / steps,
                    "mfma_frac": (t_f / t) if t > 0 else None, "hbm_frac": (t_b / t) if t > 0 else None,
                    "bound": "mfma" if t_f >= t_b else "hbm",
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


def run_rank(args):
    # must be in the environment before anything initialises HIP (RCCL's IPC path reads it at init)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # kernel arguments in device memory instead of host memory: each of a step's ~190 (fp32) / ~250 (bf16 tensors) dependent
    # launches starts 1-2 us sooner (measured, same box, alternating: 33.58 -> 33.26 ms fp32, 10.90 -> 10.64 ms bf16).  A HIP
    # runtime flag, read when libamdhip64 is loaded - i.e. before `import torch`; INTEGRATION.md lists it for deployments.
    os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29517")
    # stdout carries exactly one JSON line: RCCL prints its warnings to stdout, so everything else written to fd 1 during the
    # run is sent to stderr and the line goes to the saved descriptor
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    import torch
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
    net = network.Unet().to(dev)
    comm = None
    if use_dist:
        try:
            net.enable_data_parallel(backend=args.comm)
            comm = args.comm
        except RuntimeError as e:                           # a failed RCCL bring-up must not cost the scaling run: say so and use torch's
            if args.comm != "rccl":
                raise
            sys.stderr.write("bench.py: unet_dp_init failed (%s); falling back to torch.distributed all-reduce\n" % e)
            net.enable_data_parallel(backend="torch")
            comm = "torch (fallback)"
        if args.share_gpu:
            comm = "torch over gloo (shared-GPU rehearsal, not a scaling measurement)"
    opt = hip_optim.SGD(net.parameters(), lr=1e-4, momentum=0.99)

    B = args.batch
    g = torch.Generator(device="cpu").manual_seed(1 + rank)          # each rank its own shard of the global batch
    x = torch.rand(B, 1, S, S, generator=g).to(dev)
    labels = (torch.rand(B, 1, S - 184, S - 184, generator=g) >= 0.5).long().to(dev)

    def step():
        opt.zero_grad(set_to_none=True)
        logits = net(x)
        loss, masks = hip_optim.bce_argmax_step(logits, labels)      # L1 + L2 in one pass: loss, its gradient, argmax masks
        loss.backward()
        opt.step()
        return masks, loss

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    timing = not args.no_kernel_timing
    # The dominant kernel family (Winograd 3x3 forward/dgrad in the default arithmetic, the implicit GEMM otherwise) is timed
    # with HIP events on its launch stream INSIDE the timed region - that family only, and in the last `nsample` of the timed
    # steps only: a pair of events costs ~5 us of stream time, i.e. 0.9 ms per fp32 step (1.2 ms with bf16 tensors) around all
    # ~190 launches and still 0.4 ms around the 42 Winograd launches (measured: 33.89 / 33.46 / 33.03 ms per step with all /
    # this family's / no events), all of which would be charged to `value`.  The per-family and per-layer tables come from a
    # separate, fully instrumented pass after the timed region.
    dom = "wino" if args.math == 3 else "igemm"
    nsample = min(2, args.steps)

    def read_families():
        ms = C.c_double(); n = C.c_long(); fl = C.c_double(); ex = C.c_double(); by = C.c_double()
        fam = {}
        for k, name in enumerate(KINDS):
            _hip.check(L.unet_profile_read(k, C.byref(ms), C.byref(n), C.byref(fl), C.byref(ex), C.byref(by)))
            fam[name] = (ms.value, n.value, fl.value, ex.value, by.value)
        return fam

    if timing:
        L.unet_profile_reset(); L.unet_profile_select(1 << KINDS.index(dom))
    t0 = time.perf_counter()
    for i in range(args.steps):
        if timing and i == args.steps - nsample:
            L.unet_profile_enable(1)                       # host-side switch: no synchronisation
        masks, loss = step()
    barrier()
    dt = time.perf_counter() - t0
    dom_stats = None
    if timing:
        L.unet_profile_enable(0)
        if rank == 0:
            dom_stats = read_families()[dom]

    tmax = torch.tensor([dt], dtype=torch.float64, device="cpu" if args.share_gpu else dev)
    if use_dist:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = tmax.item()

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
        flops_step = h.flops(B, S, True)
        arith = {0: "fp32 MFMA, direct convolution", 1: "bf16x3 split", 2: "bf16 tensors and MFMA, fp32 accumulate / master weights",
                 3: "fp32 MFMA; 3x3 fwd/dgrad/wgrad as Winograd F(2x2,3x3)"}[args.math]
        out = {
            "metric": "572x572 tiles/sec fwd+bwd", "value": tiles / dt, "unit": "tiles/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": {0: "f32", 1: "bf16x3", 2: "bf16", 3: "f32"}[args.math],
            "data": "synthetic",
            "config": {"workload": "batch=%d/GPU 572x572x1 fwd+bwd+SGD %s, 64-base-ch U-Net (BASELINE %s)"
                                   % (B, "fp32" if args.math in (0, 3) else "bf16 compute", "configs[1]" if args.math in (0, 3) else "configs[2] per-GPU work"),
                       "arithmetic": arith, "global_batch": B * world, "tile": S, "parallelism": "dp%d" % world,
                       "loss": "unweighted BCE-with-logits", "final_loss": float(loss.item())},
            "effective_step_tflops": flops_step / (dt / args.steps) / 1e12,
        }
        if use_dist:
            out["comm"] = {"gradient_allreduce": comm, "ranks": int(L.unet_dp_world(h.h)) if comm == "rccl" else world,
                           "rccl_version": int(L.unet_dp_rccl_version()), "message_mb": 124.1, "buckets": 6,
                           "self_launched": os.environ.get("BENCH_SELF_LAUNCHED") == "1"}
        if timing:
            fam = read_families()                           # the instrumented pass (psteps steps)
            ms0, n0, fl0, ex0, by0 = dom_stats              # the dominant family, in the timed region
            peak = PEAK_TFLOPS[args.math]
            ach = ex0 / (ms0 * 1e-3) / 1e12 if ms0 > 0 else 0.0
            traffic = None
            tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")     # from tools/summarize_profiles.py (rocprofv3 --pmc passes)
            if os.path.exists(tpath) and B == B_PER_GPU:
                traffic = json.load(open(tpath)).get("math%d" % args.math, {}).get("%s_hbm_mb_per_launch" % dom)
                traffic = traffic * 1e6 if traffic else None
            kname = {"wino": "wino32_f32_kernel (3x3 conv fwd/dgrad, Winograd F(2x2,3x3), fp32 MFMA)",
                     "igemm": "igemm kernels (conv fwd / dgrad / up-conv implicit GEMM)",
                     "wgrad": "weight-gradient kernels"}[dom]
            out["roofline"] = {"bound": "mfma", "kernel": kname,
                               "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak,
                               "traffic": traffic,
                               "traffic_source": "profiles/pmc_traffic.json (rocprofv3 --pmc pass of this command, FETCH_SIZE x2 + WRITE_SIZE per launch; not re-measured in this run)",
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
            if use_dist:
                out["comm_detail"] = comm_table(rows, psteps)
                out["comm"]["allreduce_ms_per_step"] = out["comm_detail"]["allreduce_ms_per_step"]
                out["comm"]["exposed_ms_per_step"] = out["comm_detail"]["exposed_ms_per_step"]
            out["instrumented_pass"] = {"steps": psteps, "what": "after the timed region: every launch bracketed by HIP events on its stream; "
                                        "source of `kernels`, `layers` and --dump-launches (the events themselves cost ~1 ms per step, so this pass "
                                        "is not the one `value` is taken from)"}
            out["layers_note"] = ("per SURVEY 8a row: mfma_frac = executed matrix-core FLOPs / time / %.1f TFLOP/s (element-wise rows: their "
                                  "FLOPs against the equal fp32 vector peak), hbm_frac = algorithmic bytes / time / 8 TB/s, bound = the roof "
                                  "that would take longer at peak, frac = that roof's fraction" % PEAK_TFLOPS[args.math])
            out["kernel_time_sum_ms_per_step"] = sum(v[0] for k, v in fam.items() if k != "comm") / psteps
        if world == 1 and not args.no_cpu_baseline:
            def gpu_logits(params_np, x_np):
                # the same weights and tile through the HIP path, in this run's arithmetic mode
                chk = network.Unet()
                chk.load_state_dict({k: torch.from_numpy(v) for k, v in params_np.items()})
                chk = chk.to(dev)
                with torch.no_grad():
                    return chk(torch.from_numpy(x_np).to(dev)).cpu().numpy()
            # fp32 arithmetic: the north_star's 1e-3; bf16 tensors (mode 2, 8 significant bits per stored activation): the 5e-2 of
            # tests/test_bf16_gpu.py's whole-net bound
            out["cpu_baseline"] = cpu_baseline(gpu_check=gpu_logits, parity_bound=5e-2 if args.math == 2 else 1e-3)
    # informational: the same step in the other arithmetic modes (not part of `value`)
    if args.other_modes:
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
