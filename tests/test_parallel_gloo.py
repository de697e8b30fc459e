"""World-size-2 CPU (gloo) tests of the data-parallel path: batch sharding + one flat-bucket gradient all-reduce
reproduces the single-process full-batch gradient.  The decoder's solver hook is pointed at the CPU oracle (tests only)."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _build(seed=3):
    import model
    from oracle.solvers import odeint as oracle_odeint
    torch.manual_seed(seed)
    cpu = torch.device("cpu")
    enc = model.EncoderLSTM(7, 12, 8, device=cpu)
    dec = model.RocheExpertDecoder(6, 8, 1, 1.0, 0.125, method="rk4", device=cpu)
    dec._odeint = oracle_odeint
    return model.VariationalInference(enc, dec, elbo=False), enc, dec


def _data(B=8, T=9, obs=6):
    from hode import synth
    gen = torch.Generator().manual_seed(11)
    return {"measurements": torch.randn(T, B, obs, generator=gen), "actions": synth.one_dose_actions(T, B, gen),
            "masks": (torch.rand(T, B, obs, generator=gen) < 0.5).float()}


def _worker(rank, world, port, out_dir):
    for p in (ROOT, os.path.join(ROOT, "hybrid-ode-neurips-2021_amd")):
        sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from hode.parallel import GradBucket, shard_batch
    vi, enc, dec = _build()
    params = list(enc.parameters()) + list(dec.output_function.parameters()) + list(dec.ode.ml_net.parameters())
    loss = vi.loss(shard_batch(_data()))
    loss.backward()
    bucket = GradBucket(params)
    flat = bucket.all_reduce_mean().clone()
    torch.save({"flat": flat, "w": dec.ode.ml_net[0].weight.grad.clone()}, os.path.join(out_dir, "r%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gradient_equals_full_batch(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0 = torch.load(str(tmp_path / "r0.pt"))
    r1 = torch.load(str(tmp_path / "r1.pt"))
    assert torch.equal(r0["flat"], r1["flat"])  # both ranks hold the same averaged bucket
    vi, enc, dec = _build()
    params = list(enc.parameters()) + list(dec.output_function.parameters()) + list(dec.ode.ml_net.parameters())
    vi.loss(_data()).backward()
    # elbo=False: the log_var head gets no gradient (None) -> zeros in the bucket
    full = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in params])
    # loss is normalised by the local batch (sum/B): mean over equal shards == full-batch gradient
    torch.testing.assert_close(r0["flat"], full, rtol=2e-4, atol=1e-6)
    torch.testing.assert_close(r0["w"], dec.ode.ml_net[0].weight.grad, rtol=2e-4, atol=1e-6)


def test_shard_bounds_cover_ragged_batches():
    from hode.parallel import shard_bounds, shard_batch
    for n in (1, 7, 8, 10000, 10001):
        for world in (1, 2, 3, 8):
            spans = [shard_bounds(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans[:-1], spans[1:]))
            assert max(hi - lo for lo, hi in spans) - min(hi - lo for lo, hi in spans) <= 1
    d = {"x": torch.arange(30).reshape(3, 10, 1)}
    assert shard_batch(d, 1, 3)["x"].shape == (3, 3, 1)
