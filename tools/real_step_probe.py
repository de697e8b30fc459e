"""Config 5 (DDW-shaped real-data model: obs 24, statics 11, D 20, encoder 37 -> 44, decoder hidden 43, t0 24, T 120,
midpoint + perturb): time of one VariationalInferenceReal loss + backward, and its per-kernel split."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "hybrid-ode-neurips-2021_amd"))
import torch, model
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
obs, act, stat, D, T, t0 = 24, 1, 11, 20, 120, 24
input_dim = obs + act + stat + 1
hidden = int((obs + act + stat) * 1.2)
torch.manual_seed(0)
enc = model.EncoderLSTMReal(input_dim, int(input_dim * 1.2), D, output_all=False, reverse=False, device=dev)
dec = model.DecoderReal(obs, D, act, stat, hidden, T, 1, method="midpoint", ode_step_size=1.0, ode_type="hybrid", t0=t0, device=dev)
vi = model.VariationalInferenceReal(enc, dec, elbo=True, t0=t0)
g = torch.Generator().manual_seed(1)
data = {"measurements": torch.randn(T, B, obs, generator=g).to(dev),
        "actions": ((torch.rand(T, B, 1, generator=g) < 0.1).float() * torch.rand(T, B, 1, generator=g)).to(dev),
        "masks": (torch.rand(T, B, obs, generator=g) < 0.5).float().to(dev),
        "statics": torch.rand(T, B, stat, generator=g).to(dev)}
def step():
    for p in vi.parameters(): p.grad = None
    vi.loss(data).backward()
for _ in range(2): step()
torch.cuda.synchronize(); t0_ = time.perf_counter()
for _ in range(5): step()
torch.cuda.synchronize()
ms = (time.perf_counter() - t0_) / 5 * 1e3
print("B=%d: %.2f ms per training step (%.0f trajectories/s)" % (B, ms, B / ms * 1e3))
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    for _ in range(3): step()
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=70, max_name_column_width=110))
