"""In-kernel s_memtime stamps of the BPTT kernel: where wave 0 of block 0 spends a step.
Needs: HODE_LSTM_FLAGS=-DHODE_LSTM_STAMPS python build_hip.py --force   (product builds carry no stamps)"""
import os, sys, torch, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "hybrid-ode-neurips-2021_amd"))
from hode.lstm import lstm_encode
dev = torch.device("cuda:0")
T, B, obs, H = 100, 10000, 80, 160
g = torch.Generator().manual_seed(0)
x = torch.randn(T, B, obs, generator=g).to(dev); a = torch.rand(T, B, 1, generator=g).to(dev)
m = (torch.rand(T, B, obs, generator=g) < 0.5).float().to(dev)
lstm = torch.nn.LSTM(obs + 1, H).to(dev)
dbg = torch.zeros(T * 8, dtype=torch.int64, device=dev)
os.environ["HODE_LSTM_DBG_PTR"] = hex(dbg.data_ptr())
prm = [p for p in (lstm.weight_ih_l0, lstm.weight_hh_l0, lstm.bias_ih_l0, lstm.bias_hh_l0)]
for _ in range(2):
    for p in prm: p.grad = None
    h = lstm_encode(x, a, m, *prm, reverse=True)
    h.sum().backward()
torch.cuda.synchronize()
s = dbg.cpu().numpy().reshape(T, 8)[5:95]
d = np.diff(s, axis=1)
names = ["tile loop (elementwise + 1200 MFMAs + LDS transposes)", "barrier", "store phase (dG, h_prev rows)", "barrier", "slab write", "barrier", "carry_h read"]
for n, v in zip(names, np.median(d, axis=0)): print("%-58s %7d cycles  %.2f us" % (n, v, v / 2400.0))
per = np.median(np.abs(np.diff(s[:, 0])))
print("step period: %d cycles = %.2f us" % (per, per / 2400.0))
