"""Per-kernel breakdown (torch profiler) of one full training step of the mirror model at the bench shape."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "hybrid-ode-neurips-2021_amd"))
import torch, model
from hode import synth
from torch.profiler import profile, ProfilerActivity
dev = torch.device("cuda:0")
N, T, D, obs = 10000, 100, 12, 80
torch.manual_seed(0)
enc = model.EncoderLSTM(obs + 1, obs * 2, D, device=dev)
dec = model.RocheExpertDecoder(obs, D, 1, (T - 1) * synth.STEP, synth.STEP, method="rk4", device=dev)
vi = model.VariationalInference(enc, dec, prior_log_pdf=model.ExponentialPrior.log_density)
sol = synth.solver_inputs(N, T, D); ob = synth.observation_inputs(N, T, obs)
data = {k: v.to(dev) for k, v in {"measurements": ob["measurements"], "actions": sol["actions"], "masks": ob["masks"]}.items()}
def step():
    for p in vi.parameters(): p.grad = None
    vi.loss(data).backward()
for _ in range(3): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    for _ in range(3): step()
    torch.cuda.synchronize()
rows = sorted(prof.key_averages(), key=lambda e: -e.self_device_time_total)
tot = sum(e.self_device_time_total for e in rows)
print("total device time per step: %.3f ms" % (tot / 3e3))
for e in rows[:22]:
    print("%8.1f us/step  x%-3d %s" % (e.self_device_time_total / 3, e.count // 3, e.key[:110]))
if "--ops" in sys.argv:   # the framework-level view: which aten op (with operand shapes) each small launch belongs to
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
        step()
        torch.cuda.synchronize()
    ops = [(e.self_device_time_total, e.key, e.count, str(e.input_shapes)[:120]) for e in prof.key_averages(group_by_input_shape=True)
           if e.self_device_time_total > 0 and e.key.startswith("aten::")]
    for r in sorted(ops, reverse=True)[:40]:
        print("%8.1f us  %-26s x%-3d %s" % r)
