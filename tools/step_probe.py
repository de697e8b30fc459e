"""Timing of the full loss step of the mirror model on one GPU: encoder (nn.LSTM / MIOpen) vs solver vs readout+loss."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "hybrid-ode-neurips-2021_amd"))
import torch, model
from hode import synth
dev = torch.device("cuda:0")
N, T, D, obs = 10000, 100, 12, 80
torch.manual_seed(0)
enc = model.EncoderLSTM(obs + 1, obs * 2, D, device=dev)
dec = model.RocheExpertDecoder(obs, D, 1, (T - 1) * synth.STEP, synth.STEP, method="rk4", device=dev)
vi = model.VariationalInference(enc, dec, prior_log_pdf=model.ExponentialPrior.log_density)
sol = synth.solver_inputs(N, T, D); ob = synth.observation_inputs(N, T, obs)
data = {k: v.to(dev) for k, v in {"measurements": ob["measurements"], "actions": sol["actions"], "masks": ob["masks"]}.items()}
def timeit(fn, n=5):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
def full():
    for p in vi.parameters(): p.grad = None
    vi.loss(data).backward()
print("full loss+backward: %.2f ms" % timeit(full))
def enc_only():
    for p in enc.parameters(): p.grad = None
    mu, lv = enc(data["measurements"], data["actions"], data["masks"]); (mu.sum() + lv.sum()).backward()
print("encoder fwd+bwd (hode MFMA LSTM): %.2f ms" % timeit(enc_only))
with torch.no_grad():
    print("encoder fwd only: %.2f ms" % timeit(lambda: enc(data["measurements"], data["actions"], data["masks"])))
z = torch.rand(N, D, device=dev) * 0.01
def dec_only():
    for p in dec.parameters(): p.grad = None
    zz = z.clone().requires_grad_(True)
    xh, h = dec(zz, data["actions"]); (((data["measurements"] - xh) ** 2 * data["masks"]).sum() / N).backward()
print("decoder+lik fwd+bwd: %.2f ms" % timeit(dec_only))
def sol_only():
    zz = z.clone().requires_grad_(True)
    dec.ode.set_action(data["actions"])
    h = model.hode.odeint(dec.ode, zz, dec.t, method="rk4"); h.sum().backward()
print("solver via autograd fwd+bwd: %.2f ms" % timeit(sol_only))
print("set_action: %.2f ms" % timeit(lambda: dec.ode.set_action(data["actions"])))
