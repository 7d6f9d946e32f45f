"""Data parallelism for the U-Net path: batch-sharded replicas (tiles are independent in forward and
backward), one process per GPU, and ONE exchange step — the gradient all-reduce.

The backward is cut into stages in reverse layer order (unet_backward_stage); all parameters a stage
completes are contiguous in one flat fp32 buffer, so each stage is one bucket handed to RCCL
(`torch.distributed` backend "nccl") as soon as its kernels are enqueued, overlapping the remaining
stages.  Gradients are linear in dlogits, so the caller pre-scales dlogits by 1/world and the buckets
are SUM-reduced: the result equals the single-process gradient of the global-batch mean loss.

Nothing here touches HIP directly, so the bucket logic is testable with gloo on CPU.
"""
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
        # zero-filled so the alignment gaps never carry garbage into the reduction
        flat = torch.zeros(self.total, dtype=dtype, device=device)
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


def world_size(group=None):
    return dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1


def shard_batch(n_items, rank, world):
    """Contiguous shard [lo, hi) of a global batch for this rank (global batch must divide evenly)."""
    if n_items % world:
        raise ValueError("global batch %d is not divisible by world size %d" % (n_items, world))
    per = n_items // world
    return rank * per, (rank + 1) * per
