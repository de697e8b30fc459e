"""(needs an experiment build: HODE_DP_FLAGS=-DHODE_DP_EXPERIMENTS python build_hip.py --force)
A/B of the dopri5 forward: persistent attempt loop (default) vs one launch per attempt (HODE_DP_PERSIST=0), same process;
checks that both leave the SAME tape and trajectory."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "hybrid-ode-neurips-2021_amd"))
import bench
from hode import adaptive
dev = torch.device("cuda:0")
prob = bench.solver_problem(0)
y0 = prob["inp"]["z0"].to(dev); th = prob["theta"].to(dev); w = prob["w"].to(dev); b = prob["b"].to(dev)
t = prob["inp"]["t"].to(dev); dosage = prob["dosage"].to(dev); times = prob["times"].to(dev)
def fwd(iters=6):
    ts = []
    for i in range(iters + 2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        h = adaptive.roche_dopri5(y0, th, w, b, t, dosage, times, rtol=1e-7, atol=1e-8)
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    ts = sorted(ts[2:]); return ts[len(ts) // 2], h
res = {}
for rep in range(2):
    for mode in ("1", "0"):
        os.environ["HODE_DP_PERSIST"] = mode
        ms, h = fwd(); st = dict(adaptive.last_stats); n = st["n_accepted"] + st["n_rejected"]
        res[mode] = (h.clone(), st)
        print("rep %d persist %s: fwd %.2f ms, %d + %d attempts, %.2f us per attempt" % (rep, mode, ms, st["n_accepted"], st["n_rejected"], ms * 1e3 / n), flush=True)
print("same step counts:", res["1"][1] == res["0"][1], " trajectories bit-identical:", torch.equal(res["1"][0], res["0"][0]))
