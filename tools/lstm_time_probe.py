"""Encoder forward and backward (BPTT + weight-gradient GEMM) time at the bench shape, from HIP events."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "hybrid-ode-neurips-2021_amd"))
from hode.lstm import lstm_encode
dev = torch.device("cuda:0")
T, B, obs, H = 100, 10000, 80, 160
g = torch.Generator().manual_seed(0)
x = torch.randn(T, B, obs, generator=g).to(dev); a = torch.rand(T, B, 1, generator=g).to(dev)
m = (torch.rand(T, B, obs, generator=g) < 0.5).float().to(dev)
lstm = torch.nn.LSTM(obs + 1, H).to(dev)
prm = [lstm.weight_ih_l0, lstm.weight_hh_l0, lstm.bias_ih_l0, lstm.bias_hh_l0]
def run(n=6):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    fw, bw = [], []
    for i in range(n + 2):
        for p in prm: p.grad = None
        ev[0].record(); h = lstm_encode(x, a, m, *prm, reverse=True); ev[1].record(); h.sum().backward(); ev[2].record()
        torch.cuda.synchronize()
        if i >= 2: fw.append(ev[0].elapsed_time(ev[1])); bw.append(ev[1].elapsed_time(ev[2]))
    return sorted(fw)[len(fw)//2], sorted(bw)[len(bw)//2]
for ns in sys.argv[1:] or ["0"]:
    os.environ["HODE_LSTM_SKEW_NS"] = ns
    f, b = run()
    print("run %s: forward %.2f ms, backward (BPTT + weight-gradient GEMM) %.2f ms" % (ns, f, b), flush=True)
