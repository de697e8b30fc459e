"""Forward time of the MFMA LSTM encoder vs nn.LSTM (MIOpen).  nn.LSTM is skipped when T*B*4H exceeds int32
(MIOpen faults there: observed at T=100, B=40000, H=160)."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "hybrid-ode-neurips-2021_amd"))
import torch
from hode.lstm import lstm_final_state
dev = torch.device("cuda:0")
shapes = [(10000, 100, 80, 160), (10000, 100, 40, 80), (40000, 100, 80, 160)]
if len(sys.argv) > 1:
    shapes = [tuple(int(v) for v in sys.argv[1].split(","))]
for (N, T, obs, H) in shapes:
    torch.manual_seed(0)
    lstm = torch.nn.LSTM(obs + 1, H).to(dev)
    x = torch.randn(T, N, obs, device=dev); a = torch.rand(T, N, 1, device=dev); m = (torch.rand(T, N, obs, device=dev) < 0.5).float()
    args = (x, a, m, lstm.weight_ih_l0, lstm.weight_hh_l0, lstm.bias_ih_l0, lstm.bias_hh_l0)
    with torch.no_grad():
        for _ in range(2): lstm_final_state(*args)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5): h, c = lstm_final_state(*args)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    fl = 2.0 * T * N * (obs + 1 + H) * 4 * H
    print("N=%d T=%d obs=%d H=%d: hode %.2f ms (%.1f TFLOP/s fp32), finite=%s" % (N, T, obs, H, dt * 1e3, fl / dt / 1e12, bool(torch.isfinite(h).all())), flush=True)
    if T * N * 4 * H < 2 ** 31:
        with torch.no_grad():
            seq = torch.cat([x * m, a], -1).flip(0)
            for _ in range(2): lstm(seq)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(3): _, (h2, _) = lstm(seq)
            torch.cuda.synchronize(); dt2 = (time.perf_counter() - t0) / 3
        print("     nn.LSTM %.2f ms   max|dh| %.2e" % (dt2 * 1e3, (h - h2[0]).abs().max().item()), flush=True)
