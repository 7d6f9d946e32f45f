#!/usr/bin/env python3
"""bench.py — 572x572 tiles/s of the U-Net fwd+bwd step on N MI355X (BASELINE.json metric).

A "step" is one pass of the hot path over one batch of synthetic tiles (BASELINE configs[1]:
batch 8 per GPU, 572x572x1, fp32, 64-base-channel U-Net):
    zero_grad -> Unet.forward -> BCE-with-logits (unweighted) -> backward
    [-> RCCL gradient all-reduce, one bucket per backward stage, libunet_hip unet_dp_*] -> SGD(momentum) -> argmax
Inputs are resident in HBM before the timed region.  One process per GPU (torch.distributed.run).

Prints ONE JSON line on rank 0 with the contract fields plus
  roofline     : the dominant kernel family of the step (default arithmetic: the fp32 Winograd 3x3 kernel wino32_f32_kernel):
                 achieved = FLOPs its launches EXECUTE on the matrix cores / their HIP-event time, measured live in the timed
                 region; frac = achieved / 157.3 TFLOP/s (<= 1).  The direct-convolution-equivalent rate is `effective_tflops`.
                 (Only this family carries events in the timed region, in its last 2 steps: events around all ~190 launches
                 cost ~1 ms per step.)
  kernels, layers : from a fully instrumented pass of 3 steps after the timed region — per kernel family, and per SURVEY 8a row
                 (every conv / pool / up-conv / head layer and the step-side kernels): ms per step, executed-MFMA fraction, HBM
                 fraction (algorithmic bytes / time / 8 TB/s) and which roof binds
  cpu_baseline : the torch restatement of the same step (oracle/torch_ref.py) timed on the host cores (BASELINE.md section 4 plan)
"""
import argparse
import collections
import csv
import ctypes as C
import json
import os
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
KINDS = ("igemm", "wgrad", "wgrad_reduce", "wino", "stencil", "elementwise", "comm")

# profile row (csrc/net.hip LAYER_NAME) -> SURVEY 8a row
SURVEY_ROW = {"conv11c": "A1", "conv12c": "A2", "pool": "A3", "conv21c": "A4", "conv22c": "A5", "conv31c": "A6", "conv32c": "A7",
              "conv41c": "A8", "conv42c": "A9", "conv51c": "A10", "conv52c": "A11", "upconv": "A12", "conv41e": "A14",
              "conv42e": "A15", "conv31e": "A16", "conv32e": "A17", "conv21e": "A18", "conv22e": "A19", "conv11e": "A20",
              "conv12e": "A21", "finalconv": "A22", "L1": "L1", "L2": "L2", "L3": "L3", "allreduce": "8e"}


def cpu_baseline(budget_s=28.0):
    """The reference's CPU path (torch restatement, validated against the imported reference by
    tests/test_oracle_golden.py) on a bounded sample, BASELINE.md section 4: B=1 and B=2, forward-only (no_grad) and
    fwd+bwd+SGD, on the host's share of cores and on 8 threads.  `value` = B=1 fwd+bwd+SGD on the host share."""
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

    def run(threads, B, train, iters):
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
        while n < iters and (n == 0 or time.perf_counter() - t_start < budget_s):
            once(); n += 1
        dt = time.perf_counter() - t0
        table.append({"threads": threads, "batch": B, "what": "fwd+bwd+SGD" if train else "fwd (no_grad)", "iters": n,
                      "s_per_iter": dt / n, "tiles_per_s": B * n / dt})
        return B * n / dt

    head = run(share, 1, True, 3)
    run(share, 1, False, 3)
    run(share, 2, True, 2)
    run(share, 2, False, 2)
    if share != 8 and avail >= 8:
        run(8, 1, True, 2)
        run(8, 1, False, 2)
    return {"value": head, "unit": "tiles/s", "cores": share, "kind": "port",
            "sample": "B=1 572x572 fp32 fwd+bwd+SGD steps of the torch CPU restatement on %d threads (host share; %d cores visible); "
                      "table: B=1/B=2, forward-only and full step, %d and 8 threads (BASELINE.md section 4)" % (share, avail, share),
            "table": table, "seconds": time.perf_counter() - t_start}


def layer_table(rows, steps, math):
    """rows: parsed unet_profile_dump lines of the timed region -> per-row two-roof table."""
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


def main():
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
    ap.add_argument("--other-modes", action="store_true", help="append short measurements of the other arithmetic modes (informational)")
    ap.add_argument("--dump-launches", default=None, help="write per-launch timings (CSV) of the timed region here")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nnodes=1 --nproc-per-node %d "
                             "--master-addr 127.0.0.1 --master-port P bench.py --gpus %d ..." % (args.gpus, args.gpus))
    import torch
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    import torch.distributed as dist
    use_dist = world > 1 or args.force_dist
    if use_dist:
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
    opt = hip_optim.SGD(net.parameters(), lr=1e-4, momentum=0.99)

    B = args.batch
    g = torch.Generator(device="cpu").manual_seed(1 + rank)          # each rank its own shard of the global batch
    x = torch.rand(B, 1, S, S, generator=g).to(dev)
    labels = (torch.rand(B, 1, S - 184, S - 184, generator=g) >= 0.5).long().to(dev)
    target = hip_optim.onehot2(labels, torch.empty(B, 2, S - 184, S - 184, device=dev))

    def step():
        opt.zero_grad(set_to_none=True)
        logits = net(x)
        loss = hip_optim.bce_with_logits(logits, target)
        loss.backward()
        opt.step()
        return hip_optim.argmax2(logits.detach()), loss

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

    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
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
        h = network._handle(local_rank)
        flops_step = h.flops(B, S, True)
        arith = {0: "fp32 MFMA, direct convolution", 1: "bf16x3 split", 2: "bf16 operands, fp32 accumulate/storage",
                 3: "fp32 MFMA; 3x3 fwd/dgrad/wgrad as Winograd F(2x2,3x3)"}[args.math]
        out = {
            "metric": "572x572 tiles/sec fwd+bwd", "value": tiles / dt, "unit": "tiles/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": {0: "f32", 1: "bf16x3 (fp32 accumulate/storage)", 2: "bf16 (fp32 accumulate/storage)", 3: "f32"}[args.math],
            "data": "synthetic",
            "config": {"workload": "batch=%d/GPU 572x572x1 fwd+bwd+SGD %s, 64-base-ch U-Net (BASELINE %s)"
                                   % (B, "fp32" if args.math in (0, 3) else "bf16 compute", "configs[1]" if args.math in (0, 3) else "configs[2] per-GPU work"),
                       "arithmetic": arith, "global_batch": B * world, "tile": S, "parallelism": "dp%d" % world,
                       "loss": "unweighted BCE-with-logits", "final_loss": float(loss.item())},
            "effective_step_tflops": flops_step / (dt / args.steps) / 1e12,
        }
        if use_dist:
            out["comm"] = {"gradient_allreduce": comm, "ranks": int(L.unet_dp_world(h.h)) if comm == "rccl" else world,
                           "rccl_version": int(L.unet_dp_rccl_version()), "message_mb": 124.1, "buckets": 6}
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
            kname = {"wino": "wino32_f32_kernel (3x3 conv fwd / dgrad, Winograd F(2x2,3x3) on the fp32 MFMA)",
                     "igemm": "igemm kernels (conv fwd / dgrad / up-conv implicit GEMM)",
                     "wgrad": "weight-gradient kernels"}[dom]
            out["roofline"] = {"bound": "mfma", "kernel": kname,
                               "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak,
                               "traffic": traffic, "traffic_unit": "HBM bytes per launch (FETCH_SIZE x2 + WRITE_SIZE, profiles/)",
                               "what": "achieved = FLOPs executed on the matrix cores (Winograd: 16 multiplies per tile, channel pair and xi "
                                       "instead of 36; ragged-tile padding included) / HIP-event time of its launches in the last `sampled_steps` steps of "
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
            out["layers"] = layer_table(rows, psteps, args.math)
            out["instrumented_pass"] = {"steps": psteps, "what": "after the timed region: every launch bracketed by HIP events on its stream; "
                                        "source of `kernels`, `layers` and --dump-launches (the events themselves cost ~1 ms per step, so this pass "
                                        "is not the one `value` is taken from)"}
            out["layers_note"] = ("per SURVEY 8a row: mfma_frac = executed matrix-core FLOPs / time / %.1f TFLOP/s (element-wise rows: their "
                                  "FLOPs against the equal fp32 vector peak), hbm_frac = algorithmic bytes / time / 8 TB/s, bound = the roof "
                                  "that would take longer at peak, frac = that roof's fraction" % PEAK_TFLOPS[args.math])
            out["kernel_time_sum_ms_per_step"] = sum(v[0] for k, v in fam.items() if k != "comm") / psteps
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
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
            tm = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
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
        os.write(json_fd, (json.dumps(out) + "\n").encode())


if __name__ == "__main__":
    main()
