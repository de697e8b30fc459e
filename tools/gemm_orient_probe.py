"""Weight-gradient GEMM of the LSTM backward (K = 1e6, M = 640, N = 244, fp32): split count P and orientation
(dG^T h vs (h^T dG)^T -- the BLAS macro tile 256 x 128 wastes 20 % on 640 x 244 and 5 % on 244 x 640)."""
import sys, os, time
import torch
dev = torch.device("cuda:0")
K, M, N = 1000000, 640, 244
dg = torch.randn(K, M, device=dev)
hp = torch.randn(K, N, device=dev)
def t(fn, n=5):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
def fold(part, P):
    ones = torch.ones((1, P), device=dev)
    return (ones @ part.view(P, -1)).view(part.shape[1], part.shape[2])
ref = None
for P in [int(v) for v in os.environ.get("PROBE_P", "250,200,160,125,100,80,64,50").split(",")]:
    if K % P: continue
    a = lambda: fold(torch.bmm(dg.view(P, K // P, M).transpose(1, 2), hp.view(P, K // P, N)), P)
    b = lambda: fold(torch.bmm(hp.view(P, K // P, N).transpose(1, 2), dg.view(P, K // P, M)), P)
    ta, tb = t(a), t(b)
    if ref is None: ref = a()
    err = float((b().t() - ref).abs().max() / ref.abs().max())
    print("P=%3d  dG^T h: %.2f ms   h^T dG: %.2f ms   (max rel diff %.1e)" % (P, ta, tb, err), flush=True)
