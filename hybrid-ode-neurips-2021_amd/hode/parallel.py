"""Data parallelism over the patient axis: one process per GPU, contiguous batch shards, one flat-bucket all-reduce.

Patients are independent in the ODE, the encoder, the likelihood and the KL; the only exchange of the path is the
parameter gradient.  ``torch.distributed`` backend "nccl" is RCCL on ROCm (xGMI within a node);
"gloo" is used by the CPU tests.  The bucket is a few hundred KB at most (160 528 floats at dim12), i.e. latency
bound: a single collective per step, no per-layer bucketing.
"""

from __future__ import annotations

import torch
import torch.distributed as dist


def is_distributed() -> bool:
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def rank0_print(*args, **kw):
    if not (dist.is_available() and dist.is_initialized()) or dist.get_rank() == 0:
        print(*args, **kw)


def shard_bounds(n: int, rank: int, world: int):
    """Contiguous split of n patients; the first n % world ranks get one extra."""
    base, extra = divmod(n, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_batch(data: dict, rank: int = None, world: int = None) -> dict:
    """Slice dim 1 (patients) of every (T, B, .) tensor of a reference-style data dict for this rank."""
    if rank is None:
        rank, world = dist.get_rank(), dist.get_world_size()
    n = next(iter(data.values())).shape[1]
    lo, hi = shard_bounds(n, rank, world)
    out = {k: v[:, lo:hi] for k, v in data.items()}
    for k, v in data.items():   # a batch source's precomputed dose schedule (hode.batches.DeviceFolds) travels with the shard
        sched = getattr(v, "hode_schedule", None)
        if sched is not None:
            out[k].hode_schedule = tuple(x[lo:hi] for x in sched)
    return out


class GradBucket:
    """Flat fp32 view over the gradients of a parameter list; ``all_reduce_mean`` averages them across ranks.

    Losses are normalised per local batch (reference model.py:1179,1188), so with equal shards the mean of the
    per-rank gradients equals the global-batch gradient; with unequal shards pass ``weight = local_B / global_B * world``.
    """

    def __init__(self, params):
        self.params = [p for p in params if p.requires_grad]
        self.numel = sum(p.numel() for p in self.params)
        first = self.params[0]
        # one extra slot behind the gradients: the "this rank's step failed" flag of the training loop rides along
        self._buf = torch.zeros(self.numel + 1, device=first.device, dtype=torch.float32)
        self.flat = self._buf[:self.numel]
        self._flag = self._buf[self.numel:]
        self.views, off = [], 0
        for p in self.params:
            self.views.append(self.flat[off:off + p.numel()].view_as(p))
            off += p.numel()

    def _gather(self, weight=1.0):
        """Copy the gradients into the flat buffer: ONE multi-tensor copy (a dozen single copies cost 0.1-0.2 ms per step on
        the launch path, measured in bench.py's full step at world size 1), zeros where a parameter has no gradient.  A
        gradient that already IS its view (see _scatter) is left where it is."""
        src, dst = [], []
        for p, v in zip(self.params, self.views):
            if p.grad is None:
                v.zero_()
            elif p.grad.data_ptr() != v.data_ptr():
                src.append(p.grad.detach().to(torch.float32))
                dst.append(v)
        if dst:
            torch._foreach_copy_(dst, src)
        if weight != 1.0:
            self.flat.mul_(weight)

    def _scatter(self):
        """Hand the averaged gradients back WITHOUT a copy: p.grad becomes the parameter's view into the flat buffer.
        ``optimizer.zero_grad()`` (set_to_none=True, torch's default) drops the views again; with set_to_none=False the
        next backward accumulates into them in place and _gather finds them already in position."""
        for p, v in zip(self.params, self.views):
            p.grad = v

    def all_reduce_mean(self, weight=1.0, failed=None):
        """Average the gradients over the ranks in place.  With ``failed`` given (a bool: did THIS rank's step fail) the
        flag travels in the same collective and the call returns whether ANY rank failed -- one host read-back, so that
        all ranks leave the training loop in the same iteration; otherwise it returns the flat gradient view."""
        self._gather(weight)
        self._flag.fill_(1.0 if failed else 0.0)   # (a kernel: `buf[i] = python_float` is a synchronous host-to-device copy)
        if is_distributed():
            if dist.get_backend() == "gloo":  # gloo has no AVG
                dist.all_reduce(self._buf, op=dist.ReduceOp.SUM)
                self._buf.div_(dist.get_world_size())
            else:
                dist.all_reduce(self._buf, op=dist.ReduceOp.AVG)
        self._scatter()
        if failed is None:
            return self.flat
        return bool(self._buf[self.numel].item() > 0.0)


class AlternatingExchange:
    """Pipelines a small asynchronous collective under the NEXT step's compute with two (or ``n``) buffers.

    Step k fills buffer ``i = k % n`` and hands it to ``issue(buf)`` (e.g. ``dist.all_reduce(buf, async_op=True)``), which
    returns a work handle; the exchange runs while step k+1 fills the other buffer.  ``acquire()`` returns the buffer
    index for the coming step after waiting for the exchange issued on it ``n`` steps ago -- a buffer is never refilled
    while its collective is in flight; ``drain()`` waits for everything outstanding, so a caller that drains before it
    stops its clock has every exchange inside the timed region (bench.py).
    """

    def __init__(self, issue, n=2):
        self.issue, self.n = issue, n
        self.works = [None] * n
        self.k = self.issued = self.completed = 0

    def _finish(self, i):
        if self.works[i] is not None:
            self.works[i].wait()
            self.works[i] = None
            self.completed += 1

    def acquire(self):
        i = self.k % self.n
        self._finish(i)
        return i

    def release(self, i, buf):
        assert i == self.k % self.n and self.works[i] is None, "release() must follow the acquire() of the same step"
        self.works[i] = self.issue(buf)
        self.issued += 1
        self.k += 1

    def drain(self):
        for i in range(self.n):
            self._finish(i)
