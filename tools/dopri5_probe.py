"""Timing of the adaptive (dopri5) path at the bench shape: forward (attempt launches) and tape-driven backward."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "hybrid-ode-neurips-2021_amd"))
import torch
from hode import synth, adaptive
from hode.solver import pack_theta
dev = torch.device("cuda:0")
N, T, D = (int(v) for v in (sys.argv[1].split(",") if len(sys.argv) > 1 else (10000, 100, 12)))
inp = synth.solver_inputs(N, T, D)
w, b = synth.default_ml_weights(D)
theta = torch.tensor((2.0, 2.0) + (1.0,) * 11 + (0.0,) * 3, device=dev)
chan = inp["actions"][..., 0]
dosage = chan.max(dim=0)[0].to(dev)
times = (torch.nonzero((chan != 0).t())[:, 1].reshape(N, -1) * synth.STEP).float().to(dev)
y0 = inp["z0"].to(dev).requires_grad_(True); wg = w.to(dev).requires_grad_(True); bg = b.to(dev).requires_grad_(True)
t = inp["t"].to(dev)
cot = torch.randn(T, N, D, device=dev)
REPS = int(os.environ.get("REPS", "3"))
fw, bw = [], []
for rep in range(REPS):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    h = adaptive.roche_dopri5(y0, theta, wg, bg, t, dosage, times, rtol=1e-7, atol=1e-8)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    (h * cot).sum().backward()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    fw.append((t1 - t0) * 1e3); bw.append((t2 - t1) * 1e3)
    print("N=%d T=%d D=%d: fwd %.2f ms (%d accepted, %d rejected) bwd %.2f ms" % (N, T, D, (t1 - t0) * 1e3, adaptive.last_stats["n_accepted"], adaptive.last_stats["n_rejected"], (t2 - t1) * 1e3), flush=True)
n_att = adaptive.last_stats["n_accepted"] + adaptive.last_stats["n_rejected"]
print("summary: fwd min %.2f median %.2f ms = %.2f us per attempt (min), bwd min %.2f ms" % (min(fw[1:]), sorted(fw[1:])[len(fw[1:]) // 2], min(fw[1:]) / n_att * 1e3, min(bw[1:])), flush=True)
