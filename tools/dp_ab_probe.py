"""dopri5 leg of the bench (10 000 patients, rtol 1e-7) on two builds of the library, alternated inside one call.

    python tools/dp_ab_probe.py libhode_dpold.so libhode.so      # one child process per library, A B A B

Prints forward / backward / total ms of the adaptive solve and its attempt counts (they must agree between builds)."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "hybrid-ode-neurips-2021_amd"))


def child():
    import torch, bench
    dev = torch.device("cuda:0")
    prob = bench.solver_problem(0)
    for _ in range(2):   # the first pass ramps the clock
        r, _ = bench.dopri5_step(dev, 0, prob, iters=5, cpu=False)
    keep = {k: r[k] for k in ("ms", "fwd_ms", "bwd_ms", "n_accepted", "n_rejected")}
    keep["us_per_attempt"] = r["roofline"]["avg_launch_us"]
    print("ROW " + json.dumps(keep, default=str), flush=True)


if __name__ == "__main__":
    if "--child" in sys.argv:
        child()
        sys.exit(0)
    libdir = os.path.join(ROOT, "hybrid-ode-neurips-2021_amd", "hode")
    libs = sys.argv[1:] or ["libhode.so"]
    for rep in range(2):
        for lib in libs:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=dict(os.environ, HODE_LIBRARY=os.path.join(libdir, lib)),
                               stdout=subprocess.PIPE, text=True)
            for line in r.stdout.splitlines():
                if line.startswith("ROW "):
                    print("%-20s %s" % (lib, line[4:]), flush=True)
