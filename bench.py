#!/usr/bin/env python3
"""bench.py — 572x572 tiles/s of the U-Net fwd+bwd step on N MI355X (BASELINE.json metric).

A "step" is one pass of the hot path over one batch of synthetic tiles (BASELINE configs[1]:
batch 8 per GPU, 572x572x1, fp32, 64-base-channel U-Net):
    zero_grad -> Unet.forward -> BCE-with-logits (unweighted) -> backward
    [-> RCCL gradient all-reduce, bucketed per backward stage] -> SGD(momentum) -> argmax
Inputs are resident in HBM before the timed region.  One process per GPU (torch.distributed.run).

Prints ONE JSON line on rank 0 with the contract fields plus
  roofline     : dominant kernel = the fp32-MFMA implicit GEMM (igemm_f32_kernel); achieved =
                 algorithmic FLOPs of its launches / their HIP-event time, measured in the timed region
  cpu_baseline : the torch restatement of the same step (oracle/torch_ref.py) timed on the host cores
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "dl-unet_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402

PEAK_F32_MFMA_TFLOPS = 157.3      # /opt/skills/guides/MI355X_MICROARCH.md: dense fp32 MFMA (= vector) peak
S = 572
B_PER_GPU = 8


def cpu_baseline(max_seconds=30.0):
    """The reference's CPU path (torch restatement, validated against the imported reference by
    tests/test_oracle_golden.py) on a bounded sample: B=1 tiles of 572^2, fwd+bwd+SGD steps."""
    import numpy as np
    from oracle import prng, torch_ref
    # the GPU box gives one GPU's share of the host (16 cores); torch's default of one thread per visible
    # core (128) oversubscribes that share and runs slower
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cores = max(1, min(16, avail))
    torch.set_num_threads(cores)
    p = torch_ref.params_to_torch(prng.make_params(0), torch.float32, requires_grad=True)
    mom = {}
    x = torch.from_numpy(prng.make_input(1, 1, S))
    lab = prng.make_labels(3, 1, S - 184)
    tgt = torch.from_numpy(np.concatenate([1 - lab, lab], axis=1).astype(np.float32))
    torch_ref.train_step(p, mom, x, tgt, first=True)                 # warm-up
    t0 = time.perf_counter(); n = 0
    while True:
        torch_ref.train_step(p, mom, x, tgt)
        n += 1
        dt = time.perf_counter() - t0
        if dt > max_seconds * 0.6 or n >= 6:
            break
    return {"value": n / dt, "unit": "tiles/s", "cores": cores, "kind": "port",
            "sample": "%d fwd+bwd+SGD steps of B=1 572x572 fp32 (torch CPU restatement, %d threads)" % (n, cores)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=B_PER_GPU, help="tiles per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--force-dist", action="store_true", help="initialise torch.distributed (nccl) even for one rank")
    ap.add_argument("--math", type=int, default=3, choices=(0, 1, 2, 3),
                    help="arithmetic of the dense contractions for the MAIN measurement: 3 fp32 MFMA with Winograd F(2x2,3x3) 3x3 "
                         "layers (default), 0 fp32 MFMA direct convolution, 1 bf16x3 split, 2 bf16 compute (include/unet_hip.h unet_set_math)")
    ap.add_argument("--no-other-modes", action="store_true", help="skip the short extra measurements of the other math modes")
    ap.add_argument("--dump-launches", default=None, help="write per-launch timings (CSV) of the timed region here")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nnodes=1 --nproc-per-node %d "
                             "--master-addr 127.0.0.1 --master-port P bench.py --gpus %d ..." % (args.gpus, args.gpus))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    import torch.distributed as dist
    use_dist = world > 1 or args.force_dist
    if use_dist:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    import _hip
    import network
    import optim as hip_optim
    L = _hip.lib()

    _hip.check(L.unet_set_math(args.math), "unet_set_math")
    torch.manual_seed(0)                                   # same initial weights on every rank
    net = network.Unet().to(dev)
    if use_dist:
        net.enable_data_parallel()
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
    if timing:
        L.unet_profile_reset(); L.unet_profile_enable(1)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        masks, loss = step()
    barrier()
    dt = time.perf_counter() - t0
    if timing:
        L.unet_profile_enable(0)

    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    if use_dist:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = tmax.item()

    if rank == 0:
        tiles = B * world * args.steps
        h = network._handle(local_rank)
        flops_step = h.flops(B, S, True)
        out = {
            "metric": "572x572 tiles/sec fwd+bwd", "value": tiles / dt, "unit": "tiles/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": {0: "f32", 1: "bf16x3 (fp32 accumulate/storage)", 2: "bf16 (fp32 accumulate/storage)", 3: "f32"}[args.math],
            "data": "synthetic",
            "config": {"workload": "batch=%d/GPU 572x572x1 fwd+bwd+SGD fp32, 64-base-ch U-Net (BASELINE configs[1])" % B,
                       "arithmetic": {0: "fp32 MFMA, direct convolution", 1: "bf16x3 split", 2: "bf16", 3: "fp32 MFMA; 3x3 fwd/dgrad as Winograd F(2x2,3x3)"}[args.math],
                       "global_batch": B * world, "tile": S, "parallelism": "dp%d" % world,
                       "loss": "unweighted BCE-with-logits", "final_loss": float(loss.item())},
            "step_tflops": flops_step / (dt / args.steps) / 1e12,
        }
        if timing:
            ms = C.c_double(); n = C.c_long(); fl = C.c_double()
            fam = {}
            for f, name in ((0, "igemm_f32"), (1, "wgrad_f32"), (2, "wgrad_reduce"), (3, "wino_f32")):
                _hip.check(L.unet_profile_read(f, C.byref(ms), C.byref(n), C.byref(fl)))
                fam[name] = (ms.value, n.value, fl.value)
            # the dominant kernel: Winograd 3x3 (math mode 3) or the implicit GEMM (other modes)
            dom = "wino_f32" if fam["wino_f32"][0] > fam["igemm_f32"][0] else "igemm_f32"
            ms0, n0, fl0 = fam[dom]
            ach = fl0 / (ms0 * 1e-3) / 1e12 if ms0 > 0 else 0.0
            traffic = None
            tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")     # from tools/summarize_profiles.py (rocprofv3 --pmc passes)
            if os.path.exists(tpath) and B == B_PER_GPU:
                traffic = json.load(open(tpath)).get("%s_hbm_mb_per_launch" % dom.split("_")[0])
                traffic = traffic * 1e6 if traffic else None
            kname = {"wino_f32": "wino_f32_kernel / wino32_f32_kernel (3x3 conv fwd / dgrad, Winograd F(2x2,3x3) on the fp32 MFMA)",
                     "igemm_f32": "igemm_f32_kernel (conv fwd / dgrad / up-conv implicit GEMM)"}[dom]
            out["roofline"] = {"bound": "mfma", "kernel": kname,
                               "achieved": ach, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                               "frac": ach / PEAK_F32_MFMA_TFLOPS, "traffic": traffic, "traffic_unit": "HBM bytes per launch (FETCH_SIZE x2 + WRITE_SIZE, profiles/)",
                               "launches_per_step": n0 / args.steps, "avg_launch_ms": ms0 / max(n0, 1),
                               "gflop_per_launch": fl0 / max(n0, 1) / 1e9,
                               "share_of_step_time": ms0 / (dt * 1e3)}
            if dom == "wino_f32":
                # `achieved` counts the ALGORITHMIC flops of the direct 3x3 correlation; F(2x2,3x3) executes
                # 16/36 of those multiplies on the matrix cores, so the fraction of the MFMA peak actually kept busy is:
                out["roofline"]["executed_mfma_frac"] = ach * (16.0 / 36.0) / PEAK_F32_MFMA_TFLOPS
                out["roofline"]["note"] = "achieved = direct-convolution flops / time; Winograd executes 16/36 of them, see executed_mfma_frac"
            if args.dump_launches:
                _hip.check(L.unet_profile_dump(args.dump_launches.encode()))
            out["kernels"] = {k: {"ms_per_step": v[0] / args.steps, "launches_per_step": v[1] / args.steps,
                                  "tflops": (v[2] / (v[0] * 1e-3) / 1e12 if v[0] > 0 and v[2] > 0 else None)}
                              for k, v in fam.items()}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
    # informational: the same step in the other arithmetic modes (not part of `value`)
    if not args.no_other_modes:
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
    if rank == 0:
        print(json.dumps(out))
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
