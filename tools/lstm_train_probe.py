"""One fwd+bwd of the MFMA LSTM encoder at the bench shape (for rocprofv3 --pmc passes)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "hybrid-ode-neurips-2021_amd"))
import torch
from hode.lstm import lstm_encode
dev = torch.device("cuda:0")
N, T, obs, H = 10000, 100, 80, 160
torch.manual_seed(0)
lstm = torch.nn.LSTM(obs + 1, H).to(dev)
x = torch.randn(T, N, obs, device=dev); a = torch.rand(T, N, 1, device=dev); m = (torch.rand(T, N, obs, device=dev) < 0.5).float()
for _ in range(12):   # the first ~25 ms of work run below the full clock (profiles/r03_v0_clock_ramp.txt)
    for p in lstm.parameters(): p.grad = None
    h = lstm_encode(x, a, m, lstm.weight_ih_l0, lstm.weight_hh_l0, lstm.bias_ih_l0, lstm.bias_hh_l0)
    h.sum().backward()
torch.cuda.synchronize()
print("ok")
