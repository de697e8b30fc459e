"""Can the gradient all-reduce be captured INSIDE the HIP graph of the solver step?  (torchrun, any world size.)
Prints ms per step for: graph + eager all-reduce, graph with the all-reduce captured."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "hybrid-ode-neurips-2021_amd"))
import torch, torch.distributed as dist
import bench
local = int(os.environ.get("LOCAL_RANK", 0))
torch.cuda.set_device(local)
dev = torch.device("cuda", local)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
dist.init_process_group("nccl", device_id=dev)
plan, _, _ = bench.build_plan(dev, dist.get_rank())
def timeit(fn, n=200):
    for _ in range(20): fn()
    torch.cuda.synchronize(); dist.barrier(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
plan.capture()
def eager():
    plan.replay(); dist.all_reduce(plan.grad_flat, op=dist.ReduceOp.AVG)
print("rank %d graph + eager all-reduce: %.4f ms" % (dist.get_rank(), timeit(eager)), flush=True)
# capture the collective too
dist.all_reduce(plan.grad_flat, op=dist.ReduceOp.AVG)  # warm the communicator outside capture
torch.cuda.synchronize()
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    plan.step(); dist.all_reduce(plan.grad_flat, op=dist.ReduceOp.AVG)
torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
try:
    with torch.cuda.graph(g):
        plan.step()
        dist.all_reduce(plan.grad_flat, op=dist.ReduceOp.AVG)
    print("rank %d captured-collective graph: %.4f ms" % (dist.get_rank(), timeit(g.replay)), flush=True)
    ref = plan.grad_flat.clone(); g.replay(); torch.cuda.synchronize()
    print("rank %d replay reproducible: %s" % (dist.get_rank(), bool(torch.equal(ref, plan.grad_flat))), flush=True)
except Exception as e:
    print("capture failed:", type(e).__name__, str(e)[:300], flush=True)
dist.barrier(); dist.destroy_process_group()
