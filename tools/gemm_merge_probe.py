"""Weight-gradient GEMMs of the LSTM backward: two products (N = 80 and N = 162) vs one against the concatenated operand."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "hybrid-ode-neurips-2021_amd"))
import torch
from hode.lstm import _splitk_tn
dev = torch.device("cuda:0")
K, M = 1000000, 640
dg = torch.randn(K, M, device=dev)
def t(fn, n=5):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for widths in ((80,), (162,), (242,), (244,), (256,)):
    rhs = [torch.randn(K, w, device=dev) for w in widths]
    print(widths, "%.2f ms" % t(lambda: [_splitk_tn(dg, r) for r in rhs]), flush=True)
