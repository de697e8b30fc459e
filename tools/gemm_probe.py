"""Which formulation of the LSTM weight-gradient products (K = T*B = 1M, skinny outputs) does the BLAS library run fastest?"""
import sys, os, time
import torch
dev = torch.device("cuda:0")
TB, G, H, O = 1000000, 640, 160, 80
dg = torch.randn(TB, G, device=dev); xm = torch.randn(TB, O, device=dev); hp = torch.randn(TB, H, device=dev); a = torch.randn(TB, 1, device=dev)
def timeit(fn, n=5):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): r = fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3, r
ref = None
for name, fn in [
    ("dg.t() @ xm", lambda: dg.t() @ xm),
    ("(xm.t() @ dg).t()", lambda: (xm.t() @ dg).t()),
    ("bmm P=64", lambda: torch.bmm(dg.view(64, -1, G).transpose(1, 2), xm.view(64, -1, O)).sum(0)),
    ("bmm P=250", lambda: torch.bmm(dg.view(250, -1, G).transpose(1, 2), xm.view(250, -1, O)).sum(0)),
    ("bmm P=64 swapped", lambda: torch.bmm(xm.view(64, -1, O).transpose(1, 2), dg.view(64, -1, G)).sum(0).t()),
    ("hh: dg.t() @ hp", lambda: dg.t() @ hp),
    ("hh: bmm P=64", lambda: torch.bmm(dg.view(64, -1, G).transpose(1, 2), hp.view(64, -1, H)).sum(0)),
    ("hh: bmm P=64 swapped", lambda: torch.bmm(hp.view(64, -1, H).transpose(1, 2), dg.view(64, -1, G)).sum(0).t()),
    ("cat acts then one product bmm P=64", lambda: torch.bmm(dg.view(64, -1, G).transpose(1, 2), torch.cat([xm, a, hp], 1).view(64, -1, O + 1 + H)).sum(0)),
    ("a: dg.t() @ a", lambda: dg.t() @ a),
    ("a: (dg * a).sum(0)", lambda: (dg * a).sum(0)),
    ("bias: dg.sum(0)", lambda: dg.sum(0)),
]:
    ms, r = timeit(fn)
    print("%-40s %.2f ms" % (name, ms), flush=True)
