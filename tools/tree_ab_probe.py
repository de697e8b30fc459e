"""Full training step (bench shape) and config-5 step of two source TREES, alternated in one call (boxes differ by 1-3 %, so
two versions of the Python layer + library are only ever compared inside one call):

    git archive <commit> hybrid-ode-neurips-2021_amd include build_hip.py | tar -x -C _ab_old && (cd _ab_old && python build_hip.py)
    python tools/tree_ab_probe.py _ab_old .          # one child process per tree and repetition

A tree is a directory that holds `hybrid-ode-neurips-2021_amd/` with its library built in place."""
import json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child(tree):
    pkg = os.path.join(os.path.abspath(tree), "hybrid-ode-neurips-2021_amd")
    sys.path.insert(0, pkg)
    os.environ["HODE_LIBRARY"] = os.path.join(pkg, "hode", "libhode.so")
    import torch, model
    from hode import synth
    dev = torch.device("cuda:0")
    N, T, D, obs = 10000, 100, 12, 80
    torch.manual_seed(0)
    enc = model.EncoderLSTM(obs + 1, obs * 2, D, device=dev)
    dec = model.RocheExpertDecoder(obs, D, 1, (T - 1) * synth.STEP, synth.STEP, method="rk4", device=dev)
    vi = model.VariationalInference(enc, dec, prior_log_pdf=model.ExponentialPrior.log_density)
    opt = torch.optim.Adam(vi.parameters(), lr=1e-3)
    sol = synth.solver_inputs(N, T, D); ob = synth.observation_inputs(N, T, obs)
    data = {k: v.to(dev) for k, v in {"measurements": ob["measurements"], "actions": sol["actions"], "masks": ob["masks"]}.items()}

    def step():
        opt.zero_grad(set_to_none=True)
        vi.loss(data).backward()
        opt.step()

    def timeit(fn, warm, n):
        for _ in range(warm): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): fn()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3

    full = timeit(step, 6, 20)
    # config 5
    B, obs5, act, stat, D5, T5, t0 = 8192, 24, 1, 11, 20, 120, 24
    inp = obs5 + act + stat + 1
    enc5 = model.EncoderLSTMReal(inp, int(inp * 1.2), D5, output_all=False, reverse=False, device=dev)
    dec5 = model.DecoderReal(obs5, D5, act, stat, int((obs5 + act + stat) * 1.2), T5, 1, method="midpoint", ode_step_size=1.0, ode_type="hybrid", t0=t0, device=dev)
    vi5 = model.VariationalInferenceReal(enc5, dec5, elbo=True, t0=t0)
    g = torch.Generator().manual_seed(1)
    d5 = {"measurements": torch.randn(T5, B, obs5, generator=g).to(dev),
          "actions": ((torch.rand(T5, B, 1, generator=g) < 0.1).float() * torch.rand(T5, B, 1, generator=g)).to(dev),
          "masks": (torch.rand(T5, B, obs5, generator=g) < 0.5).float().to(dev), "statics": torch.rand(T5, B, stat, generator=g).to(dev)}

    def step5():
        for p in vi5.parameters(): p.grad = None
        vi5.loss(d5).backward()

    c5 = timeit(step5, 5, 20)
    print("ROW " + json.dumps({"full_step_ms": full, "config5_step_ms": c5}), flush=True)


if __name__ == "__main__":
    if sys.argv[1] == "--child":
        child(sys.argv[2])
        sys.exit(0)
    trees = sys.argv[1:]
    for rep in range(3):
        for t in trees:
            out = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", t], stdout=subprocess.PIPE, text=True, cwd=ROOT).stdout
            for line in out.splitlines():
                if line.startswith("ROW "):
                    d = json.loads(line[4:])
                    print("rep %d %-10s full training step %.3f ms   config-5 step %.3f ms" % (rep, t, d["full_step_ms"], d["config5_step_ms"]), flush=True)
