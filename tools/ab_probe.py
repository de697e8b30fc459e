"""Same-call A/B of the headline kernels between library builds: python tools/ab_probe.py libA.so libB.so [...] [--reps 3]
(each library in a fresh child process per repetition, interleaved; kernel times from HIP events after a clock-ramp burst).
Boxes differ by a few per cent between calls, so variants are only ever compared inside one call."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "hybrid-ode-neurips-2021_amd"))


def child():
    import torch, bench
    dev = torch.device("cuda:0")
    prob = bench.solver_problem(0)
    plan = bench.build_plan(dev, prob)
    plan.capture()
    for _ in range(600):
        plan.replay()
    torch.cuda.synchronize()
    f, b, bc = bench.kernel_times(plan, iters=50)
    import time
    t0 = time.perf_counter()
    for _ in range(200):
        plan.replay()
    torch.cuda.synchronize()
    step = (time.perf_counter() - t0) / 200
    print("ROW " + json.dumps({"fwd_us": f * 1e6, "bwd_us": b * 1e6, "bwd_call_us": bc * 1e6, "step_us": step * 1e6,
                               "mtraj": bench.N_PER_GPU / step / 1e6}), flush=True)


if __name__ == "__main__":
    if "--child" in sys.argv:
        child()
        sys.exit(0)
    args = [a for a in sys.argv[1:]]
    reps = 3
    if "--reps" in args:
        i = args.index("--reps"); reps = int(args[i + 1]); del args[i:i + 2]
    libdir = os.path.join(ROOT, "hybrid-ode-neurips-2021_amd", "hode")
    for r in range(reps):
        for lib in args:
            path = lib if os.path.isabs(lib) else os.path.join(libdir, lib)
            out = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=dict(os.environ, HODE_LIBRARY=path),
                                 stdout=subprocess.PIPE, text=True).stdout
            for line in out.splitlines():
                if line.startswith("ROW "):
                    d = json.loads(line[4:])
                    print("rep %d %-24s fwd %6.1f us  bwd %6.1f us (call %6.1f)  step %6.1f us = %6.2f M trajectories/s"
                          % (r, lib, d["fwd_us"], d["bwd_us"], d["bwd_call_us"], d["step_us"], d["mtraj"]), flush=True)
