"""In-kernel s_memtime stamps of the dopri5 attempt: where a wave of block 0 / block n/2 spends its life.
Needs a library built with the stamps compiled in: HODE_DP_FLAGS=-DHODE_DP_STAMPS python build_hip.py --force"""
import os, sys, torch, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "hybrid-ode-neurips-2021_amd"))
import bench
from hode import adaptive
dev = torch.device("cuda:0")
prob = bench.solver_problem(0)
y0 = prob["inp"]["z0"].to(dev).requires_grad_(True); th = prob["theta"].to(dev); w = prob["w"].to(dev); b = prob["b"].to(dev)
t = prob["inp"]["t"].to(dev); dosage = prob["dosage"].to(dev); times = prob["times"].to(dev)
adaptive.keep_workspace = True
for _ in range(2):
    adaptive.roche_dopri5(y0, th, w, b, t, dosage, times, rtol=1e-7, atol=1e-8)
torch.cuda.synchronize()
ws, d, n_acc = adaptive._last_ws
raw = ws.cpu().numpy()
P = 8 * 12 + 8 + 15
gp = ((625 * P * 4 + 255) // 256) * 256
off = raw.size - gp
st = adaptive.last_stats; n_att = min(2300, st["n_accepted"] + st["n_rejected"])
dbg = np.frombuffer(raw[off:off + n_att * 2 * 8 * 8].tobytes(), dtype=np.uint64).reshape(n_att, 2, 8).astype(np.int64)
for blk in (0, 1):
    s = dbg[100:n_att - 10, blk, :6]
    d_ = np.diff(s, axis=1)
    print("block %s: median cycles  entry->L1 back %d | ->L2 back %d | ->decision %d | ->stages done %d | ->stores drained %d | total %d" % (
        ("0" if blk == 0 else "n/2",) + tuple(np.median(d_, axis=0)) + (np.median(s[:, 5] - s[:, 0]),)))
    gap = dbg[101:n_att - 10, blk, 0] - dbg[100:n_att - 11, blk, 5]
    print("   end of attempt k -> entry of attempt k+1 (same block): median %d cycles; attempt period median %d cycles" % (
        np.median(gap), np.median(np.diff(dbg[100:n_att - 10, blk, 0]))))
