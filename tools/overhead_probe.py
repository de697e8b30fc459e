"""Fixed per-launch cost of the solver kernels: time them at T = 2, 3, 4, 100 grid points (10 000 patients, D = 12)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "hybrid-ode-neurips-2021_amd"))
import torch
import bench

dev = torch.device("cuda:0")
for T in (2, 3, 4, 10, 100):
    bench.T = T
    plan, _, _ = bench.build_plan(dev, 0)
    def t(fn, n=200):
        for _ in range(20): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e3
    print("T=%3d  fwd %.2f us   bwd(kernel only) %.2f us   bwd(call) %.2f us" % (T, t(plan.forward), t(plan.backward_kernel_only), t(plan.backward)))
