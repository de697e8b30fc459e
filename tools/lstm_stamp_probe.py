"""In-kernel s_memtime stamps of the BPTT kernel: where wave 0 of block 0 spends a step.
Needs: HODE_LSTM_FLAGS=-DHODE_LSTM_STAMPS python build_hip.py --force   (product builds carry no stamps)"""
import os, sys, torch, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "hybrid-ode-neurips-2021_amd"))
from hode.lstm import lstm_encode
dev = torch.device("cuda:0")
T, B, obs, H = 100, int(os.environ.get("PROBE_B", "10000")), 80, 160   # PROBE_B=144: three blocks, the tape stays in cache (isolates memory latency)
g = torch.Generator().manual_seed(0)
x = torch.randn(T, B, obs, generator=g).to(dev); a = torch.rand(T, B, 1, generator=g).to(dev)
m = (torch.rand(T, B, obs, generator=g) < 0.5).float().to(dev)
lstm = torch.nn.LSTM(obs + 1, H).to(dev)
dbg = torch.zeros(T * 8 + 2 * T * 16, dtype=torch.int64, device=dev)
os.environ["HODE_LSTM_DBG_PTR"] = hex(dbg.data_ptr())
dbgf = torch.zeros(T * 8 + 2 * T * 16, dtype=torch.int64, device=dev)
os.environ["HODE_LSTM_FWD_DBG_PTR"] = hex(dbgf.data_ptr())
prm = [p for p in (lstm.weight_ih_l0, lstm.weight_hh_l0, lstm.bias_ih_l0, lstm.bias_hh_l0)]
for _ in range(2):
    for p in prm: p.grad = None
    h = lstm_encode(x, a, m, *prm, reverse=True)
    h.sum().backward()
torch.cuda.synchronize()
allb = dbg.cpu().numpy()
s = allb[:T * 8].reshape(T, 8)[5:95]
body = allb[T * 16: 2 * T * 16].reshape(T, 16)[5:95].astype(np.int64)
print('BPTT tile bodies, cycles (120 MFMAs = 3840 when the pipe never waits): first body starts %d after the step stamp;' % int(np.median(body[:, 0] - s[:, 0].astype(np.int64))), np.median(np.diff(body[:, :10], axis=1), axis=0).astype(int).tolist(), '; last body + tail', int(np.median(s[:, 1].astype(np.int64) - body[:, 9])))
d = np.diff(s, axis=1)
names = ["tile loop (elementwise + 1200 MFMAs + LDS transposes)", "barrier", "store phase (dG, h_prev rows)", "barrier", "slab write", "barrier", "carry_h read"]
for n, v in zip(names, np.median(d, axis=0)): print("%-58s %7d cycles  %.2f us" % (n, v, v / 2400.0))
per = np.median(np.abs(np.diff(s[:, 0])))
print("step period: %d cycles = %.2f us" % (per, per / 2400.0))

print("forward kernel (with tape):")
allf = dbgf.cpu().numpy()
s = allf[:T * 8].reshape(T, 8)[5:95]
d = np.diff(s[:, :5], axis=1)
for n, v in zip(["MFMA loop (weights from L2, B fragments from LDS)", "cell update + h to LDS + tape stores", "x staging (next step)", "barrier"], np.median(d, axis=0)):
    print("%-58s %7d cycles  %.2f us" % (n, v, v / 2400.0))
per = np.median(np.abs(np.diff(s[:, 0])))
print("step period: %d cycles = %.2f us" % (per, per / 2400.0))
grp = allf[T * 16: 2 * T * 16].reshape(T, 16)[5:95].astype(np.int64)
st0 = s[:, 0].astype(np.int64)
med = lambda v: int(np.median(v))
print("forward MFMA section, cycles (a full 4-quad group = 120 MFMAs = 3840 cycles when the pipe never waits):")
print("  step start -> loop entry (x fetch issue, accumulator init, first B read):", med(grp[:, 10] - st0))
print("  pairs of groups:", [med(grp[:, i + 1] - grp[:, i]) for i in range(7)], "last pair:", med(grp[:, 8] - grp[:, 7]))
print("  odd group + tail group:", med(grp[:, 9] - grp[:, 8]))
print("  -> stamp after s_waitcnt(0):", med(s[:, 1].astype(np.int64) - grp[:, 9]))
