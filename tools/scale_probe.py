"""Solver throughput vs batch size per GPU, product build against the occupancy variants of the split kernels.

    python build_hip.py --variant wpeB --unit-flags hode_rk_split="-DHODE_SPLIT_WPE_FWD=2 -DHODE_SPLIT_WPE_BWD=3"
    python build_hip.py --variant wpeC --unit-flags hode_rk_split="-DHODE_SPLIT_WPE_FWD=4 -DHODE_SPLIT_WPE_BWD=4"
    python tools/scale_probe.py            # on the GPU box: one child process per library, same call

The product build pins ONE workgroup per CU (`amdgpu_waves_per_eu(1, 1)` makes the backend pad the forward kernel's
register allocation to 264, tools/kernel_descriptor.py); variant B lets two forward workgroups share a CU (176 registers),
C four (104).  The backward cannot co-reside in any variant: its workgroup holds 94 KB of LDS (two do not fit 160 KB) and
216 registers x 5 waves.  Rows: kernel times from HIP events (bench.kernel_times), after a pre-conditioning burst."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "hybrid-ode-neurips-2021_amd"))
SIZES = (10000, 12288, 20000, 24576, 40000, 80000, 160000)


def child():
    import torch, bench
    dev = torch.device("cuda:0")
    rows = []
    for n in SIZES:
        bench.N_PER_GPU = n
        prob = bench.solver_problem(0)
        plan = bench.build_plan(dev, prob)
        for _ in range(300):      # clock ramp (profiles/r03_v0_clock_ramp.txt)
            plan.step()
        torch.cuda.synchronize()
        f, b, bc = bench.kernel_times(plan, iters=20)
        rows.append({"patients": n, "fwd_us": f * 1e6, "bwd_us": b * 1e6, "bwd_call_us": bc * 1e6, "mtraj_per_s": n / (f + bc) / 1e6,
                     "fwd_only_mtraj_per_s": n / f / 1e6})
        print("  N=%6d fwd %7.1f us  bwd %7.1f us (call %7.1f)  -> %6.2f M trajectories/s (forward alone %6.1f M/s)"
              % (n, f * 1e6, b * 1e6, bc * 1e6, n / (f + bc) / 1e6, n / f / 1e6), flush=True)
        del plan, prob
        torch.cuda.empty_cache()
    print("ROWS " + json.dumps(rows), flush=True)


if __name__ == "__main__":
    if "--child" in sys.argv:
        child()
        sys.exit(0)
    libdir = os.path.join(ROOT, "hybrid-ode-neurips-2021_amd", "hode")
    out = {}
    for tag, lib in (("product (forward <= 2 workgroups / CU, backward 1)", "libhode.so"), ("B: forward <= 2 workgroups / CU", "libhode_wpeB.so"),
                     ("C: forward <= 4 workgroups / CU", "libhode_wpeC.so"),
                     ("bwd3: backward forced to <= 168 registers (two workgroups / CU fit)", "libhode_bwd3.so"), ("product again", "libhode.so")):
        path = os.path.join(libdir, lib)
        if not os.path.exists(path):
            print("skip %s (%s not built)" % (tag, lib))
            continue
        print(tag, flush=True)
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=dict(os.environ, HODE_LIBRARY=path),
                           stdout=subprocess.PIPE, text=True)
        for line in r.stdout.splitlines():
            if line.startswith("ROWS "):
                out[tag] = json.loads(line[5:])
            else:
                print(line, flush=True)
    print("JSON " + json.dumps(out))
