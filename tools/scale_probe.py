"""Solver throughput vs batch size per GPU for the three layouts (split = default, quad, one lane per patient)."""
import sys, os, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "hybrid-ode-neurips-2021_amd"))
import torch, bench
for n in (2500, 5000, 10000, 12288, 20000, 40000, 80000, 160000):
    for lanes in (0, 4, 1):
        bench.N_PER_GPU = n
        plan, _, _ = bench.build_plan(torch.device("cuda:0"), 0, lanes=lanes)
        for _ in range(3): plan.step()
        f, b, bc = bench.kernel_times(plan, iters=10)
        print("N=%6d lanes=%d fwd %.1f us bwd %.1f us (call %.1f)  -> %.2f Mtraj/s" % (n, lanes, f*1e6, b*1e6, bc*1e6, n/(f+bc)/1e6), flush=True)
        del plan
        torch.cuda.empty_cache()
