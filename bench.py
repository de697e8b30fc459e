#!/usr/bin/env python
"""bench.py -- patient-trajectories/sec (fwd + discrete adjoint) of the hybrid-ODE solver path on MI355X.

Workload (BASELINE.json configs[1]): dim=12 synthetic, 10 000 patients PER GPU, T=100 grid points (dt=0.125),
3/8-rule RK4, fused rhs+step HIP kernels.  One "step" = one forward solve + one discrete-adjoint solve over the
rank's batch (+ one RCCL all-reduce of the parameter-gradient bucket when N>1).  Inputs are resident in HBM
before the timed region.  Rank 0 prints ONE JSON line.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--no-cpu] [--no-graph]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W
"""
import argparse
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "hybrid-ode-neurips-2021_amd"))

import torch  # noqa: E402

N_PER_GPU, T, D = 10000, 100, 12
def host_cores():
    """CPU cores this process may actually use: affinity mask, capped by the cgroup quota (the GPU box gives a
    16-core share of a much larger host; sizing thread pools by os.cpu_count() there oversubscribes badly)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(float(txt[0]) / float(txt[1]) + 0.5)))
            else:
                q = int(txt[0])
                if q > 0:
                    per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                    n = min(n, max(1, int(q / per + 0.5)))
        except (OSError, ValueError, IndexError):
            pass
    return max(1, min(n, int(os.environ.get("HODE_CPU_THREADS", "16"))))


def log(msg):
    print("[bench] " + msg, file=sys.stderr, flush=True)


HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6.3 TB/s is the measured copy ceiling


def build_plan(dev, rank, lanes=0, need_theta=True, tape=True):
    from hode import synth
    from hode.plan import RocheRKPlan
    from hode.solver import pack_theta
    inp = synth.solver_inputs(N_PER_GPU, T, D, seed=synth.SEED + rank)
    w, b = synth.default_ml_weights(D)
    theta = torch.tensor((2.0, 2.0) + (1.0,) * 11 + (0.0,) * 3)  # RochConfig defaults (sim_config.py:4-18)
    chan = inp["actions"][..., 0]
    dosage = chan.max(dim=0)[0]
    times = (torch.nonzero((chan != 0).t())[:, 1].reshape(N_PER_GPU, -1) * synth.STEP).float()
    plan = RocheRKPlan(inp["z0"].to(dev), theta.to(dev), w.to(dev), b.to(dev), inp["t"].to(dev), dosage.to(dev),
                       times.to(dev), method="rk4", lanes_per_patient=lanes, need_theta_grad=need_theta, tape=tape)
    gen = torch.Generator().manual_seed(99 + rank)
    plan.grad_h.copy_(torch.randn(T, N_PER_GPU, D, generator=gen))  # synthetic cotangent (what the readout+loss would send)
    return plan, inp, (w, b)


def cpu_baseline(inp, wb, n_sample=2000, reps=3):
    """The CPU oracle (op-for-op PyTorch eager restatement of the reference path, autograd backward) on a bounded
    sample of the same workload: first `n_sample` patients, all host threads."""
    from oracle.rhs import RocheRHS
    from oracle.solvers import odeint
    from hode import synth
    torch.set_num_threads(host_cores())
    f = RocheRHS(D, synth.STEP)
    with torch.no_grad():
        f.ml_net[0].weight.copy_(wb[0])
        f.ml_net[0].bias.copy_(wb[1])
    a = inp["actions"][:, :n_sample]
    z0 = inp["z0"][:n_sample]
    cot = torch.randn(T, n_sample, D, generator=torch.Generator().manual_seed(99))
    times = []
    dt = 0.0
    for i in range(reps + 1):
        if i == 1 and dt > 15.0:  # bounded: keep the whole CPU leg to tens of seconds
            reps = 1
        if i > reps:
            break
        t0 = time.perf_counter()
        f.set_action(a)
        y0 = z0.clone().requires_grad_(True)
        f.zero_grad()
        h = odeint(f, y0, inp["t"], method="rk4")
        (h * cot).sum().backward()
        dt = time.perf_counter() - t0
        log("cpu_baseline rep %d: %.2f s (%d threads)" % (i, dt, torch.get_num_threads()))
        if i > 0:
            times.append(dt)
    med = statistics.median(times)
    return {"value": n_sample / med, "unit": "trajectories/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": "first %d of %d patients, T=%d, D=%d, rk4, fwd+autograd bwd, 1 warm-up + %d reps (median %.3f s)"
                      % (n_sample, N_PER_GPU, T, D, reps, med)}


def pmc_traffic(tape=True):
    """HBM bytes per launch of the dominant (backward) kernel from the committed rocprofv3 --pmc passes
    (profiles/*_pmc_summary.json, FETCH_SIZE doubled per MI355X_MICROARCH.md section HBM); None if absent.
    The tape variant of the kernel (template arguments end in `true, true>`) reads the forward's stage tape on top of
    the algorithmic bytes; the newest summary that holds the variant being timed wins."""
    import glob
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_summary.json"))):
        try:
            d = json.load(open(f))
        except (OSError, ValueError):
            continue
        for k, v in d.items():
            if ("split_bwd_kernel<12" in k or "rk_bwd_kernel<12" in k) and "hbm_bytes_per_launch" in v:
                if best is not None and "split" in best.get("kernel", "") and "split" not in k:
                    continue
                if "split" in k and (k.count(",") == 4) and (k.rstrip().endswith("true, true>") != tape):
                    continue
                if "split" in k and k.count(",") < 4 and tape:
                    continue
                best = {"bytes_per_launch": v["hbm_bytes_per_launch"], "source": os.path.relpath(f, ROOT), "kernel": k}
    return best


def full_step_ms(dev, iters=5):
    """Extra (not the headline): one full training step of the mirror model at the same shape -- MFMA LSTM encoder
    (obs 80 -> H 160), HIP solver, readout, masked SSE + MC-KL, backward through everything."""
    import model
    from hode import synth
    obs = 80
    torch.manual_seed(synth.SEED)
    enc = model.EncoderLSTM(obs + 1, obs * 2, D, device=dev)
    dec = model.RocheExpertDecoder(obs, D, 1, (T - 1) * synth.STEP, synth.STEP, method="rk4", device=dev)
    vi = model.VariationalInference(enc, dec, prior_log_pdf=model.ExponentialPrior.log_density)
    sol = synth.solver_inputs(N_PER_GPU, T, D)
    ob = synth.observation_inputs(N_PER_GPU, T, obs)
    data = {k: v.to(dev) for k, v in {"measurements": ob["measurements"], "actions": sol["actions"], "masks": ob["masks"]}.items()}

    def step():
        for p in vi.parameters():
            p.grad = None
        vi.loss(data).backward()

    step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


def dopri5_step(dev, rank, dist=None, iters=3):
    """Extra (not the headline; BASELINE config 3 per GPU): the same patients through the adaptive Dormand-Prince solve
    (rtol 1e-7, atol 1e-8, per-rank batch-global controller) + its tape adjoint, and -- when distributed -- the all-reduce
    of the parameter gradients.  Max over ranks, like the headline."""
    from hode import synth, adaptive
    inp = synth.solver_inputs(N_PER_GPU, T, D, seed=synth.SEED + rank)
    w, b = synth.default_ml_weights(D)
    theta = torch.tensor((2.0, 2.0) + (1.0,) * 11 + (0.0,) * 3, device=dev)
    chan = inp["actions"][..., 0]
    dosage = chan.max(dim=0)[0].to(dev)
    times = (torch.nonzero((chan != 0).t())[:, 1].reshape(N_PER_GPU, -1) * synth.STEP).float().to(dev)
    y0 = inp["z0"].to(dev).requires_grad_(True)
    wg, bg = w.to(dev).requires_grad_(True), b.to(dev).requires_grad_(True)
    t = inp["t"].to(dev)
    cot = torch.randn(T, N_PER_GPU, D, device=dev)

    def step():
        y0.grad = wg.grad = bg.grad = None
        h = adaptive.roche_dopri5(y0, theta, wg, bg, t, dosage, times, rtol=1e-7, atol=1e-8)
        (h * cot).sum().backward()
        if dist is not None:
            flat = torch.cat([wg.grad.flatten(), bg.grad.flatten()])
            dist.all_reduce(flat, op=dist.ReduceOp.AVG)

    step()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        step()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / iters * 1e3
    if dist is not None:
        tt = torch.tensor([ms], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        ms = float(tt.item())
    return ms, dict(adaptive.last_stats)


def kernel_times(plan, iters=20):
    """Average duration of the forward kernel and of the adjoint kernel ALONE (no memset, no partial fold -- the quantity
    rocprofv3's kernel stats report), plus the whole backward call, from HIP events on the launch stream."""
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(iters)]
    for i in range(iters):
        ev[i][0].record()
        plan.forward()
        ev[i][1].record()
        plan.backward_kernel_only()
        ev[i][2].record()
        plan.backward()
        ev[i][3].record()
    torch.cuda.synchronize()
    fwd = statistics.mean(e[0].elapsed_time(e[1]) for e in ev) * 1e-3
    bwd_kernel = statistics.mean(e[1].elapsed_time(e[2]) for e in ev) * 1e-3
    bwd_call = statistics.mean(e[2].elapsed_time(e[3]) for e in ev) * 1e-3
    return fwd, bwd_kernel, bwd_call


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--no-graph", action="store_true", help="launch eagerly instead of replaying a HIP graph")
    ap.add_argument("--lanes", type=int, default=0, help="force lanes per patient (1|4), 0 = library default")
    ap.add_argument("--no-theta-grad", action="store_true", help="skip the 13 expert-constant gradients")
    ap.add_argument("--graph-allreduce", action="store_true", help="N>1: capture the gradient all-reduce inside the step's HIP graph")
    ap.add_argument("--sync-allreduce", action="store_true", help="N>1: wait for each step's gradient all-reduce before the next solve")
    ap.add_argument("--no-tape", action="store_true", help="backward re-integrates the expert stages instead of reading the forward's tape")
    ap.add_argument("--full-step", action="store_true", help="also time one full training step (encoder + loss) as an extra field")
    ap.add_argument("--dopri5", action="store_true", help="also time the adaptive (dopri5) solve + adjoint at the same shape (BASELINE config 3 per GPU) as an extra field")
    args = ap.parse_args()

    # The contract is ONE JSON line on stdout.  Native libraries write there too (RCCL prints a version banner on
    # communicator creation): keep the real stdout aside for the result line and point fd 1 at stderr for everything else.
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py: --gpus %d needs `python -m torch.distributed.run --nproc-per-node %d ...`" % (args.gpus, args.gpus))
    if not torch.cuda.is_available():
        sys.exit("bench.py: no HIP device visible (the solver path has no CPU fallback)")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1 or "RANK" in os.environ:  # launched by torch.distributed.run: always take the distributed path
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)  # "nccl" IS RCCL on ROCm

    plan, inp, wb = build_plan(dev, rank, lanes=args.lanes, need_theta=not args.no_theta_grad, tape=not args.no_tape)
    use_graph = not args.no_graph
    log("rank %d: plan built (B=%d, T=%d, D=%d)" % (rank, N_PER_GPU, T, D))
    overlap = dist is not None and use_graph and not args.sync_allreduce and not args.graph_allreduce
    in_graph = dist is not None and use_graph and args.graph_allreduce
    if use_graph:
        plan.capture(n_buckets=2 if overlap else 1)
        log("rank %d: graph captured" % rank)
    step_graph = None
    if in_graph:
        # opt-in: the all-reduce captured INSIDE the step's HIP graph (RCCL supports capture): no hand-over between
        # torch's and RCCL's streams per step (0.155 ms per step at world size 1 against 0.168 eager).  Not the default
        # because it cannot be rehearsed at world size > 1 on the one-GPU development box.
        dist.all_reduce(plan.grad_flat, op=dist.ReduceOp.AVG)  # communicator warm-up outside capture
        torch.cuda.synchronize()
        step_graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(step_graph):
            plan.step()
            dist.all_reduce(plan.grad_flat, op=dist.ReduceOp.AVG)
        log("rank %d: step graph with the all-reduce captured" % rank)

    # Data-parallel gradient exchange over xGMI: one RCCL all-reduce(AVG) of the flat bucket per step.  It is issued
    # asynchronously on alternating buckets (one captured graph per bucket), so the exchange of step k runs on RCCL's stream
    # while the solver kernels of step k+1 run -- in the full training step the same bucket is exchanged under the
    # encoder's BPTT, which follows the solver backward.  Every exchange completes inside the timed region (fence()).
    # --sync-allreduce serialises it instead (replay -> all-reduce -> replay).
    works = [None, None]
    state = {"k": 0}

    def step():
        if step_graph is not None:
            step_graph.replay()
            return
        if overlap:
            i = state["k"] & 1
            if works[i] is not None:
                works[i].wait()  # stream-side wait for the exchange issued two steps ago before its bucket is refilled
            plan.replay(i)       # the graph captured for bucket i
            works[i] = dist.all_reduce(plan.buckets[i], op=dist.ReduceOp.AVG, async_op=True)
            state["k"] += 1
            return
        if use_graph:
            plan.replay()
        else:
            plan.step()
        if dist is not None:
            dist.all_reduce(plan.grad_flat, op=dist.ReduceOp.AVG)

    def fence():
        if dist is not None:
            for w in works:
                if w is not None:
                    w.wait()
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    t_enqueued = time.perf_counter() - t0
    fence()
    elapsed = time.perf_counter() - t0
    log("rank %d: %d steps in %.4f s (host had enqueued them after %.4f s)" % (rank, args.steps, elapsed, t_enqueued))
    if dist is not None:
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    dp = dopri5_step(dev, rank, dist) if args.dopri5 else None  # every rank takes part (its all-reduce is collective)
    out = None
    if rank == 0:
        fwd_s, bwd_s, bwd_call_s = kernel_times(plan)
        ms = elapsed / args.steps * 1e3
        total = N_PER_GPU * world
        ach = plan.bwd_bytes / bwd_s / 1e9
        out = {
            "metric": "patient-trajectories/sec (fwd+adjoint) at dim=12, T=100",
            "value": total * args.steps / elapsed,
            "unit": "trajectories/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "dim12 synthetic, %d patients/GPU, T=%d, dt=0.125, rk4(3/8), fused rhs+step kernel "
                                   "+ discrete-adjoint kernel (split expert/learned wave pipelines)" % (N_PER_GPU, T),
                       "patients_total": total, "launch": "hipGraph" if use_graph else "eager",
                       "lanes_per_patient": args.lanes or "auto", "theta_grad": not args.no_theta_grad,
                       "parallelism": "dp%d" % world,
                       "grad_exchange": None if dist is None else ("rccl all-reduce(AVG), async under the next solve" if overlap
                                                                   else ("rccl all-reduce(AVG), captured in the step graph" if in_graph
                                                                         else "rccl all-reduce(AVG), serialised"))},
            "roofline": {"bound": "hbm", "kernel": "split_bwd_kernel<12, rk4> (adjoint kernel alone; the whole backward call incl. "
                                                      "accumulator memset and partial folds is bwd_call_us)", "achieved": ach,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": None,
                         "bytes_per_launch": plan.bwd_bytes, "avg_launch_us": bwd_s * 1e6, "bwd_call_us": bwd_call_s * 1e6,
                         "fwd": {"bytes_per_launch": plan.fwd_bytes, "avg_launch_us": fwd_s * 1e6,
                                 "achieved": plan.fwd_bytes / fwd_s / 1e9},
                         "step_frac": (plan.fwd_bytes + plan.bwd_bytes) / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBS},
        }
        tr = pmc_traffic(tape=not args.no_tape)
        if tr is not None:
            out["roofline"]["traffic"] = tr["bytes_per_launch"]
            out["roofline"]["traffic_source"] = tr["source"]
        if not args.no_tape:
            # deliberate recompute <-> traffic trade: the forward leaves the 3 intermediate expert stage
            # states of every step (16 B each) and the backward reads them back instead of re-integrating
            out["roofline"]["tape_bytes_per_launch"] = (T - 1) * 3 * N_PER_GPU * 16
            out["config"]["stage_tape"] = True
        if args.full_step:
            ms_full = full_step_ms(dev)
            out["full_training_step"] = {"ms": ms_full, "trajectories_per_s": N_PER_GPU / ms_full * 1e3,
                                         "what": "EncoderLSTM(81->160, MFMA) + rk4 solve + readout + masked SSE + MC-KL, fwd+bwd"}
        if dp is not None:
            out["dopri5_step"] = {"ms": dp[0], "trajectories_per_s": N_PER_GPU * world / dp[0] * 1e3, "rtol": 1e-7, "atol": 1e-8,
                                  "n_accepted": dp[1]["n_accepted"], "n_rejected": dp[1]["n_rejected"],
                                  "what": "dopri5 solve (one launch per attempted step, batch-global controller per rank) + "
                                          "tape adjoint" + ("" if dist is None else " + rccl all-reduce of the parameter grads")}
        if world == 1 and not args.no_cpu:
            out["cpu_baseline"] = cpu_baseline(inp, wb)
            out["speedup_vs_cpu"] = out["value"] / out["cpu_baseline"]["value"]
        sys.stdout.flush()
        os.write(result_fd, (json.dumps(out) + "\n").encode())
    os.close(result_fd)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
