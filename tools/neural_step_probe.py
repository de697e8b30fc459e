"""NeuralODE decoder (reference model.py:969-1026, 13 -> 120 -> 12 MLP rhs) with the fixed-grid rk4 kernels: time of the
solve + backward at the bench shape."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "hybrid-ode-neurips-2021_amd"))
import torch, model
from hode import synth
dev = torch.device("cuda:0")
B, T, D, obs = 10000, 100, 12, 80
torch.manual_seed(0)
dec = model.RocheExpertDecoder(obs, D, 1, (T - 1) * synth.STEP, synth.STEP, roche=False, method="rk4", device=dev)
sol = synth.solver_inputs(B, T, D)
z0 = sol["z0"].to(dev).requires_grad_(True); a = sol["actions"].to(dev)
cot = torch.randn(T, B, D, device=dev)
def step():
    for p in dec.parameters(): p.grad = None
    h = dec.latent(z0, a)
    (h * cot).sum().backward()
for _ in range(2): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): step()
torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 5 * 1e3
print("NeuralODE rk4 solve + adjoint, B=%d T=%d D=%d: %.2f ms (%.0f trajectories/s)" % (B, T, D, ms, B / ms * 1e3))
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    for _ in range(2): step()
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=8, max_name_column_width=60))
