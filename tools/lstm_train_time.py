"""Per-kernel time of one fwd+bwd of the MFMA LSTM encoder at the bench shape (torch profiler, device time)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "hybrid-ode-neurips-2021_amd"))
import torch
from hode.lstm import lstm_encode
dev = torch.device("cuda:0")
N, T, obs, H = 10000, 100, 80, 160
torch.manual_seed(0)
lstm = torch.nn.LSTM(obs + 1, H).to(dev)
x = torch.randn(T, N, obs, device=dev); a = torch.rand(T, N, 1, device=dev); m = (torch.rand(T, N, obs, device=dev) < 0.5).float()
def run():
    for p in lstm.parameters(): p.grad = None
    h = lstm_encode(x, a, m, lstm.weight_ih_l0, lstm.weight_hh_l0, lstm.bias_ih_l0, lstm.bias_hh_l0)
    h.sum().backward()
for _ in range(3): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5): run()
e1.record(); torch.cuda.synchronize()
print("fwd+bwd %.2f ms" % (e0.elapsed_time(e1) / 5))
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    for _ in range(3): run()
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=10, max_name_column_width=70))
