"""Data parallelism for the U-Net path: batch-sharded replicas (tiles are independent in forward and
backward), one process per GPU, and ONE exchange step — the gradient all-reduce.

The backward is cut into stages in reverse layer order (unet_backward_stage); all parameters a stage
completes are contiguous in one flat fp32 buffer, so each stage is one bucket handed to RCCL — the
library's own communicator and stream, unet_dp_allreduce (csrc/dp.hip) — as soon as its kernels are
enqueued, overlapping the remaining stages.  Gradients are linear in dlogits, so the caller pre-scales
dlogits by 1/world and the buckets are SUM-reduced: the result equals the single-process gradient of
the global-batch mean loss.

GradBuckets touches neither HIP nor RCCL, so the bucket logic is testable with gloo on CPU; the
DataParallel object's "torch" backend keeps that path usable on a GPU box with one GPU (two gloo ranks
sharing it), its "rccl" backend is what bench.py and training use.
"""
import ctypes as C

import torch
import torch.distributed as dist

ALIGN = 64      # elements; keeps every tensor 256-byte aligned inside the flat buffer


class GradBuckets:
    def __init__(self, numels, order, bounds):
        """numels[i]: elements of parameter i; order: parameter indices in completion order;
        bounds[s]..bounds[s+1]: slice of `order` completed by backward stage s."""
        self.numels = list(numels)
        self.order = list(order)
        self.bounds = list(bounds)
        self.offs = {}
        tot = 0
        for i in self.order:
            self.offs[i] = tot
            tot += (self.numels[i] + ALIGN - 1) // ALIGN * ALIGN
        self.total = tot

    def n_stages(self):
        return len(self.bounds) - 1

    def allocate(self, shapes, device, dtype=torch.float32):
        """One flat buffer + per-parameter views (in parameter-index order)."""
        # not zero-filled (124 MB per step): the <= 63-element alignment gaps between tensors ride along in the bucket
        # all-reduce, which is element-wise - whatever they hold stays in the gaps and is never read
        flat = torch.empty(self.total, dtype=dtype, device=device)
        views = [flat[self.offs[i]:self.offs[i] + self.numels[i]].view(shapes[i]) for i in range(len(self.numels))]
        return flat, views

    def bucket_range(self, s):
        ids = self.order[self.bounds[s]:self.bounds[s + 1]]
        a = self.offs[ids[0]]
        last = ids[-1]
        b = self.offs[last] + (self.numels[last] + ALIGN - 1) // ALIGN * ALIGN
        return a, b

    def reduce_stage(self, flat, s, group=None):
        """Asynchronous SUM all-reduce of stage s's bucket; returns the work handle."""
        a, b = self.bucket_range(s)
        return dist.all_reduce(flat[a:b], op=dist.ReduceOp.SUM, group=group, async_op=True)


class DataParallel:
    """Rank/world bookkeeping + the collective calls of one module replica."""

    def __init__(self, handle, device, group=None, backend="rccl"):
        if backend not in ("rccl", "torch"):
            raise ValueError("backend must be 'rccl' or 'torch'")
        self.group, self.backend, self.device = group, backend, device
        inited = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size(group) if inited else 1
        self.rank = dist.get_rank(group) if inited else 0
        self._works = []
        import _hip
        L = _hip.lib()
        if backend == "torch" and not inited:
            raise RuntimeError("backend 'torch' needs torch.distributed.init_process_group first")
        # gradients are linear in dlogits: the handle's backward reads dlogits / world (inside the head's backward kernel), so
        # the SUM all-reduce of the buckets is the gradient of the global-batch mean loss
        _hip.check(L.unet_set_grad_scale(handle.h, 1.0 / self.world), "unet_set_grad_scale")
        self._handle = handle
        if backend == "torch":
            return
        # rendezvous: rank 0 makes the id, torch.distributed (any backend) carries the 128 bytes (+ one flag byte: a failure on
        # rank 0 must reach the other ranks, or they would wait in the broadcast / inside ncclCommInitRank for ever)
        idbuf = (C.c_ubyte * 128)()
        ok0, err0 = 1, ""
        if self.rank == 0:
            try:
                _hip.check(L.unet_dp_unique_id(idbuf), "unet_dp_unique_id")
            except RuntimeError as e:
                ok0, err0 = 0, str(e)
        on_gpu = self.world > 1 and dist.get_backend(group) == "nccl"
        cdev = device if on_gpu else "cpu"
        if self.world > 1:
            t = torch.tensor(list(idbuf) + [ok0], dtype=torch.uint8, device=cdev)
            dist.broadcast(t, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
            vals = t.cpu().tolist()
            idbuf = (C.c_ubyte * 128)(*vals[:128])
            ok0 = vals[128]
        if not ok0:
            raise RuntimeError("data parallel: rank 0 could not create the RCCL rendezvous id %s" % err0)
        err = ""
        with torch.cuda.device(device):
            try:
                _hip.check(L.unet_dp_init(handle.h, self.rank, self.world, idbuf), "unet_dp_init")
            except RuntimeError as e:
                err = str(e)
        if self.world > 1:
            # every rank learns whether all of them have a communicator: a caller that falls back to another backend
            # (bench.py) must do so on all ranks together
            flag = torch.tensor([0 if err else 1], dtype=torch.int32, device=cdev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
            if flag.item() == 0:
                if not err:
                    with torch.cuda.device(device):
                        L.unet_dp_destroy(handle.h)
                    err = "unet_dp_init failed on another rank"
        if err:
            raise RuntimeError(err)
        self._handle = handle

    def broadcast_parameters(self, params):
        if self.world <= 1 and self.backend == "torch":
            return
        with torch.no_grad(), torch.cuda.device(self.device):
            if self.backend == "torch":
                for p in params:
                    dist.broadcast(p, src=dist.get_global_rank(self.group, 0) if self.group is not None else 0, group=self.group)
                return
            import _hip
            L = _hip.lib()
            st = _hip.stream(self.device)
            for p in params:
                _hip.check(L.unet_dp_broadcast(self._handle.h, _hip.ptr(p), p.numel(), 0, st), "unet_dp_broadcast")
            _hip.check(L.unet_dp_join(self._handle.h, st), "unet_dp_join")

    def reduce_stage(self, buckets, flat, s, handle, stream):
        """SUM all-reduce of backward stage s's bucket, asynchronous with respect to `stream`."""
        if self.backend == "torch":
            self._works.append(buckets.reduce_stage(flat, s, self.group))
            return
        import _hip
        a, b = buckets.bucket_range(s)
        _hip.check(_hip.lib().unet_dp_allreduce(handle.h, C.c_void_p(flat.data_ptr() + 4 * a), b - a, stream), "unet_dp_allreduce")

    def join(self, handle, stream):
        """After this, work enqueued on `stream` sees the reduced gradients."""
        if self.backend == "torch":
            for w in self._works:
                w.wait()
            self._works = []
            return
        import _hip
        _hip.check(_hip.lib().unet_dp_join(handle.h, stream), "unet_dp_join")


def world_size(group=None):
    return dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1


def shard_batch(n_items, rank, world):
    """Contiguous shard [lo, hi) of a global batch for this rank (global batch must divide evenly)."""
    if n_items % world:
        raise ValueError("global batch %d is not divisible by world size %d" % (n_items, world))
    per = n_items // world
    return rank * per, (rank + 1) * per
