#!/usr/bin/env python
"""bench.py -- patient-trajectories/sec (fwd + discrete adjoint) of the hybrid-ODE solver path on MI355X.

Workload (BASELINE.json configs[1]): dim=12 synthetic, 10 000 patients PER GPU, T=100 grid points (dt=0.125),
3/8-rule RK4, fused rhs+step HIP kernels.  One "step" = one forward solve + one discrete-adjoint solve over the
rank's batch (+ one RCCL all-reduce of the parameter-gradient bucket when N>1).  Inputs are resident in HBM
before the timed region.  Rank 0 prints ONE JSON line.

Besides the headline the default run also measures (same shape, same patients):
  * "parity"             -- the shipped kernels (split layout + stage tape + hipGraph) against the CPU oracle on ALL
                            10 000 patients of this very run: trajectory MSE / max-abs, gradient rel-L2
  * "cpu_baseline"       -- that oracle, timed: 3 warm-up + 5 repetitions, median, all host cores of the share
  * "full_training_step" -- encoder (MFMA LSTM) + solver + readout + loss, forward + backward, with its own roofline
                            (MFMA fraction), cpu_baseline and parity (oracle pipeline on a sample)
  * "dopri5_step"        -- BASELINE config 3 per GPU: adaptive solve + tape adjoint, with roofline (per attempt),
                            cpu_baseline and parity (the oracle's step algebra replayed along the run's own tape)
`--headline-only` skips the two extra blocks, `--no-cpu` every CPU leg.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W
`python bench.py --gpus N` without a launcher starts that command itself (before anything touches the GPU).

N > 1: every process the launcher starts is first a SUPERVISOR that has not touched the GPU; it runs the real rank as a
child.  The default gradient exchange is the all-reduce captured inside the step's HIP graph (no per-step hand-over
between torch's and RCCL's streams); each rank guards the first replays with its own watchdog (exit code 17 if they do
not finish), and the supervisors then start FRESH ranks on the overlap path (async all-reduce on alternating buckets)
under a new store prefix; `config.grad_exchange` / `config.grad_exchange_fallback` say which path the numbers come from.
At N > 1 every rank also runs `full_training_step` (with the flat-bucket all-reduce of `training_utils`) and
`dopri5_step`, so a scaling run yields the solver-only, the full-step and the dopri5 curve.  CPU rehearsal of this
control flow: tests/test_bench_dist.py (gloo, world size 2, HODE_BENCH_STUB).

Before the W warm-up steps the same step is replayed, untimed, for `--precondition-ms` (60) of wall time: from idle the
card's clock ramps for ~25 ms (profiles/r03_v0_clock_ramp.txt), longer than 5 + 20 steps take; `config.preconditioning`
discloses it.
"""
import argparse
import json
import os
import socket
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "hybrid-ode-neurips-2021_amd"))

N_PER_GPU, T, D, OBS = 10000, 100, 12, 80
HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6.3 TB/s is the measured copy ceiling
MFMA_F32_PEAK_TFLOPS = 157.3  # v_mfma_f32_16x16x4_f32: exact fp32 in / fp32 accumulate (same guide); no TF32 on gfx950
PRECONDITION_MS = 60.0        # graph replays ahead of the W warm-up steps: the clock ramps for ~25 ms from idle (profiles/r03_v0_clock_ramp.txt)


def encoder_flops(n, t=T, i=OBS + 1, h=2 * OBS):
    """EXECUTED matrix flops of the encoder per forward + backward over n patients: the forward's
    G = Wcat [x*mask | a | h] (2 (I+H) 4H per patient and step), the BPTT's dh = W_hh^T dG (2 H 4H; there is no dX product:
    the inputs need no gradient), the weight-gradient GEMM dG^T [x*mask | a | h | 1] (2 4H (I+H+1)).  SURVEY 8d's
    "3 x forward" estimate (92.5 Mflop per trajectory) over-counts by the missing dX product."""
    fwd = 2.0 * (i + h) * 4 * h * t * n
    bwd = 2.0 * h * 4 * h * t * n
    wgrad = 2.0 * 4 * h * (i + h + 1) * t * n
    return {"lstm_fwd": fwd, "lstm_bwd": bwd, "wgrad_gemm": wgrad, "total": fwd + bwd + wgrad}


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


def self_launch(args):
    """`python bench.py --gpus N` (N > 1) without a launcher: start N fresh ranks through torch.distributed.run as a
    CHILD process -- this parent has not touched the GPU -- relay its stdout (the one JSON line) and exit with its code."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    log("self-launch: " + " ".join(cmd))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE)
    sys.stdout.write(r.stdout.decode())
    sys.stdout.flush()
    sys.exit(r.returncode)


def _rel(a, b):
    a, b = a.double().flatten().cpu(), b.double().flatten().cpu()
    return float((a - b).norm() / (b.norm() + 1e-300))


def solver_problem(rank):
    import torch
    from hode import synth
    inp = synth.solver_inputs(N_PER_GPU, T, D, seed=synth.SEED + rank)
    w, b = synth.default_ml_weights(D)
    theta = torch.tensor((2.0, 2.0) + (1.0,) * 11 + (0.0,) * 3)  # RochConfig defaults (sim_config.py:4-18)
    chan = inp["actions"][..., 0]
    dosage = chan.max(dim=0)[0]
    times = (torch.nonzero((chan != 0).t())[:, 1].reshape(N_PER_GPU, -1) * synth.STEP).float()
    cot = torch.randn(T, N_PER_GPU, D, generator=torch.Generator().manual_seed(99 + rank))  # what readout+loss would send
    return {"inp": inp, "w": w, "b": b, "theta": theta, "dosage": dosage, "times": times, "cot": cot}


def build_plan(dev, prob, lanes=0, need_theta=True, tape=True):
    from hode.plan import RocheRKPlan
    plan = RocheRKPlan(prob["inp"]["z0"].to(dev), prob["theta"].to(dev), prob["w"].to(dev), prob["b"].to(dev),
                       prob["inp"]["t"].to(dev), prob["dosage"].to(dev), prob["times"].to(dev), method="rk4",
                       lanes_per_patient=lanes, need_theta_grad=need_theta, tape=tape)
    plan.grad_h.copy_(prob["cot"])
    return plan


def oracle_rhs(prob):
    import torch
    from hode import synth
    from oracle.rhs import RocheRHS
    f = RocheRHS(D, synth.STEP)
    with torch.no_grad():
        f.ml_net[0].weight.copy_(prob["w"])
        f.ml_net[0].bias.copy_(prob["b"])
    return f


def cpu_baseline_and_parity(prob, gpu, warm=3, reps=5):
    """The CPU oracle (op-for-op PyTorch eager restatement of the reference path, autograd backward) on the SAME
    10 000 patients, the same cotangent: timed per SURVEY.md 8d (3 warm-up + 5 repetitions, median; the batch is cut
    to 2 000 only if one repetition exceeds 60 s), and its outputs compared with what the GPU run left in HBM."""
    import torch
    from oracle.rhs import THETA_NAMES
    from oracle.solvers import odeint
    torch.set_num_threads(host_cores())
    f = oracle_rhs(prob)
    n = N_PER_GPU
    times, first = [], None
    i = 0
    while i < warm + reps:
        a, z0, cot = prob["inp"]["actions"][:, :n], prob["inp"]["z0"][:n], prob["cot"][:, :n]
        t0 = time.perf_counter()
        f.set_action(a)
        y0 = z0.clone().requires_grad_(True)
        f.zero_grad()
        h = odeint(f, y0, prob["inp"]["t"], method="rk4")
        (h * cot).sum().backward()
        dt = time.perf_counter() - t0
        log("cpu_baseline rep %d: %.2f s (%d patients, %d threads)" % (i, dt, n, torch.get_num_threads()))
        if first is None and n == N_PER_GPU:
            first = {"h": h.detach(), "gy0": y0.grad.clone(), "gw": f.ml_net[0].weight.grad.clone(),
                     "gb": f.ml_net[0].bias.grad.clone(), "gth": torch.stack([getattr(f, k).grad for k in THETA_NAMES])}
        if i == 0 and dt > 60.0 and n == N_PER_GPU:
            n = 2000  # reduce only past 60 s per repetition, and say so
            continue
        if i >= warm:
            times.append(dt)
        i += 1
    med = statistics.median(times)
    base = {"value": n / med, "unit": "trajectories/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": "%s %d patients, T=%d, D=%d, rk4, fwd + autograd bwd, %d warm-up + %d reps (median %.3f s)"
                      % ("all" if n == N_PER_GPU else "first", n, T, D, warm, reps, med)}
    par = None
    if first is not None:
        hg = gpu["h"].cpu()
        err = (hg - first["h"]).double()
        par = {"vs": "CPU oracle, all %d patients of this run, shipped kernels (%s)" % (N_PER_GPU, gpu["what"]),
               "mse": float((err ** 2).mean()), "max_abs": float(err.abs().max()), "max_abs_h": float(first["h"].abs().max()),
               "grad_y0_rel": _rel(gpu["gy0"], first["gy0"]), "grad_w_rel": _rel(gpu["gw"], first["gw"]),
               "grad_b_rel": _rel(gpu["gb"], first["gb"]), "grad_theta_rel": _rel(gpu["gth"][:13], first["gth"]),
               "tolerance": {"mse": 1e-5, "grad_rel": 1e-3}}
        rels = [par["grad_y0_rel"], par["grad_w_rel"], par["grad_b_rel"]]
        if gpu.get("need_theta", True):
            rels.append(par["grad_theta_rel"])  # the theta wave of split_bwd_kernel is part of the shipped adjoint: it gates too
        else:
            par["grad_theta_rel"] = None
        par["ok"] = bool(par["mse"] <= 1e-5 and max(rels) <= 1e-3)
    return base, par


def _profile_order(path):
    """Natural order of the committed profile names (r03_v10 after r03_v9): the newest set wins."""
    import re
    return [int(x) if x.isdigit() else x for x in re.split(r"(\d+)", os.path.basename(path))]


def pmc_traffic(tape=True):
    """HBM bytes per launch of the dominant (backward) kernel from the committed rocprofv3 --pmc passes
    (profiles/*_pmc_summary.json, FETCH_SIZE doubled per MI355X_MICROARCH.md section HBM); None if absent.  The counters
    cannot be collected from inside this process, so this is the newest COMMITTED profile of the kernel variant being
    timed (the JSON line says so in `traffic_source`)."""
    import glob
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_summary.json")), key=_profile_order):
        try:
            d = json.load(open(f))
        except (OSError, ValueError):
            continue
        for k, v in d.items():
            if not isinstance(v, dict):
                continue
            if ("split_bwd_kernel<12" in k or "rk_bwd_kernel<12" in k) and "hbm_bytes_per_launch" in v:
                if best is not None and "split" in best.get("kernel", "") and "split" not in k:
                    continue
                if "split" in k and (k.count(",") == 4) and (k.rstrip().endswith("true, true>") != tape):
                    continue
                if "split" in k and k.count(",") < 4 and tape:
                    continue
                best = {"bytes_per_launch": v["hbm_bytes_per_launch"], "source": os.path.relpath(f, ROOT), "kernel": k,
                        "valu_active_frac": v.get("valu_active_fraction_of_wave_cycles")}
    return best


# ---------------------------------------------------------------------------------------------------------------------
# extra block 1: full training step

def lstm_pmc():
    """Counter evidence for the encoder kernels from the newest COMMITTED rocprofv3 --pmc pass set
    (profiles/*lstm*_pmc_summary.json: FETCH_SIZE / WRITE_SIZE in separate passes, matrix-pipe busy cycles, LDS bank
    conflicts); None if absent.  Counters cannot be collected from inside this process."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*lstm*_pmc_summary.json")), key=_profile_order)
    if not files:
        return None
    d = json.load(open(files[-1]))
    pick = {"lstm_fwd": "hode::lstm_fwd_kernel", "lstm_bwd": "hode::lstm_bwd_kernel", "wgrad_gemm": "Cijk_Ailk_Bjlk"}
    out = {"source": os.path.relpath(files[-1], ROOT)}
    for name, prefix in pick.items():
        for k, v in d.items():
            if k.startswith(prefix) and "hbm_bytes_per_launch" in v:
                out[name] = {"kernel": k, "hbm_bytes_per_launch": v["hbm_bytes_per_launch"],
                             "mfma_busy_fraction_of_chip_simd_cycles": v.get("mfma_busy_fraction_of_chip_simd_cycles"),
                             "lds_bank_conflict_fraction_of_lds_cycles": v.get("lds_bank_conflict_fraction_of_lds_cycles")}
                break
    return out


def full_training_step(dev, iters=10, cpu=True, n_cpu=1000, dist=None, rank=0):
    """One full training step of the mirror model at the bench shape -- MFMA LSTM encoder (obs 80 -> H 160), HIP solver,
    fused readout + masked SSE, MC-KL, backward through everything, Adam update -- plus the encoder alone (forward + BPTT +
    weight gradients) for the MFMA roofline, with the three encoder pieces timed by HIP events on the launch stream.
    Distributed (`dist` given): every rank runs the step on its own 10 000 patients and the flat gradient bucket of all
    trainable parameters is averaged over RCCL before the optimiser update (`hode.parallel.GradBucket`, the path
    `training_utils.variational_training_loop` takes); the time is the max over ranks.
    Returns the result block and a closure that adds the CPU legs (the oracle pipeline on the first `n_cpu` patients as
    baseline and checker): every GPU measurement of the run is taken before the first CPU leg, because the card clocks
    down while the host computes for tens of seconds."""
    import torch
    import model
    from hode import lstm as hlstm
    from hode import synth
    from hode.parallel import GradBucket
    torch.manual_seed(synth.SEED)
    enc = model.EncoderLSTM(OBS + 1, OBS * 2, D, device=dev)
    dec = model.RocheExpertDecoder(OBS, D, 1, (T - 1) * synth.STEP, synth.STEP, method="rk4", device=dev)
    vi = model.VariationalInference(enc, dec, prior_log_pdf=model.ExponentialPrior.log_density)
    sol = synth.solver_inputs(N_PER_GPU, T, D, seed=synth.SEED + rank)
    ob = synth.observation_inputs(N_PER_GPU, T, OBS, seed=synth.SEED + rank)
    host = {"measurements": ob["measurements"], "actions": sol["actions"], "masks": ob["masks"]}
    data = {k: v.to(dev) for k, v in host.items()}
    world = 1 if dist is None else dist.get_world_size()

    # the reference's optimiser (experiments/run_simulation.py:131: Adam over encoder + decoder parameters): its update is part of
    # a training step and is inside the timed region
    opt_params = list(enc.parameters()) + list(dec.output_function.parameters()) + list(dec.ode.ml_net.parameters())
    try:
        opt = torch.optim.Adam(opt_params, lr=1e-3, fused=True)   # the same update in one launch instead of ~20
    except (RuntimeError, TypeError):
        opt = torch.optim.Adam(opt_params, lr=1e-3)
    bucket = GradBucket(opt_params) if dist is not None else None

    def step():
        opt.zero_grad(set_to_none=True)
        vi.loss(data).backward()
        if bucket is not None:
            bucket.all_reduce_mean()   # one RCCL all-reduce(AVG) of the flat bucket (160 528 floats), equal shards
        opt.step()

    def enc_only():
        for p in enc.parameters():
            p.grad = None
        mu, lv = enc(data["measurements"], data["actions"], data["masks"])
        (mu.sum() + lv.sum()).backward()

    host_ms = []

    def timed(fn):
        for _ in range(3):
            fn()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            fn()
        host_ms.append((time.perf_counter() - t0) / iters * 1e3)   # time the HOST needed to enqueue a step
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / iters * 1e3
        if dist is not None:
            tt = torch.tensor([ms], device=dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            ms = float(tt.item())
        return ms

    ms = timed(step)
    hlstm.timeline = []
    try:
        ms_enc = timed(enc_only)
        torch.cuda.synchronize()
        spans = {}
        for name, e0, e1 in hlstm.timeline[-3 * iters:]:      # the timed repetitions only
            spans.setdefault(name, []).append(e0.elapsed_time(e1))
    finally:
        hlstm.timeline = None
    fl = encoder_flops(N_PER_GPU)
    pieces = {}
    for name in ("lstm_fwd", "lstm_bwd", "wgrad_gemm"):
        t_ms = statistics.mean(spans[name])
        pieces[name] = {"ms": t_ms, "gflop": fl[name] / 1e9, "achieved": fl[name] / (t_ms * 1e-3) / 1e12,
                        "frac": fl[name] / (t_ms * 1e-3) / 1e12 / MFMA_F32_PEAK_TFLOPS}
    ach = fl["total"] / (ms_enc * 1e-3) / 1e12
    out = {"ms": ms, "host_enqueue_ms": host_ms[0], "trajectories_per_s": N_PER_GPU * world / ms * 1e3, "n_ranks": world,
           "per_gpu_trajectories_per_s": N_PER_GPU / ms * 1e3,
           "what": "EncoderLSTM(81->160, fp32 MFMA) + rk4 solve + fused readout / masked SSE + MC-KL, fwd+bwd%s + Adam update, "
                   "%d patients per GPU; the batch is the same tensors at every step, so set_action's dose schedule comes from its identity "
                   "cache after the first step (no host synchronisation inside the step: host_enqueue_ms)"
                   % ("" if dist is None else " + rccl all-reduce(AVG) of the flat gradient bucket", N_PER_GPU),
           "roofline": {"bound": "mfma", "kernel": "lstm_fwd_kernel + lstm_bwd_kernel (v_mfma_f32_16x16x4_f32) + weight-gradient GEMM (hipBLASLt)",
                        "achieved": ach, "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach / MFMA_F32_PEAK_TFLOPS,
                        "traffic": None,
                        "executed_gflop": fl["total"] / 1e9,
                        "flop_formula": "T B [2 (I+H) 4H (forward) + 2 H 4H (BPTT dh) + 2 4H (I+H+1) (weight gradients)], T=%d B=%d I=%d H=%d; "
                                        "no dX product" % (T, N_PER_GPU, OBS + 1, 2 * OBS),
                        "encoder_fwd_bwd_ms": ms_enc,
                        "per_kernel": pieces,
                        "per_kernel_note": "HIP events on the launch stream around each piece of the encoder call: lstm_fwd = weight pack "
                                           "kernels + lstm_fwd_kernel; lstm_bwd = lstm_bwd_kernel with the x*mask operand fill running beside it on a side "
                                           "stream (2.73 ms alone); wgrad_gemm = split-K hipBLASLt GEMM + fold; frac = executed flops / time / 157.3 TFLOP/s",
                        "frac_of_whole_step": fl["total"] / (ms * 1e-3) / 1e12 / MFMA_F32_PEAK_TFLOPS}}
    pm = lstm_pmc()
    if pm is not None and all(k in pm for k in ("lstm_fwd", "lstm_bwd", "wgrad_gemm")):
        out["roofline"]["traffic"] = sum(pm[k]["hbm_bytes_per_launch"] for k in ("lstm_fwd", "lstm_bwd", "wgrad_gemm"))
        out["roofline"]["traffic_source"] = ("committed profile %s (rocprofv3 --pmc, FETCH_SIZE doubled, not collected in this run): "
                                             "HBM bytes of the three encoder kernels per forward + backward" % pm["source"])
        out["roofline"]["algorithmic_bytes"] = 2 * 4 * T * N_PER_GPU * OBS + 4 * T * N_PER_GPU   # x, mask, action read once
        out["roofline"]["counters"] = {k: pm[k] for k in ("lstm_fwd", "lstm_bwd", "wgrad_gemm")}
    if not cpu or rank != 0:
        return out, lambda: None
    # the parity pass on the GPU (elbo=False: no sampling noise in the way), first n_cpu patients
    small = {k: v[:, :n_cpu].contiguous() for k, v in host.items()}
    vi_p = model.VariationalInference(enc, dec, elbo=False)
    for p in vi_p.parameters():
        p.grad = None
    loss_g = vi_p.loss({k: v.to(dev) for k, v in small.items()})
    loss_g.backward()
    keys = ("lstm.weight_ih_l0", "lstm.weight_hh_l0", "lin.weight", "output_function.0.weight", "ode.ml_net.0.weight")
    gp = dict(list(enc.named_parameters()) + list(dec.named_parameters()))
    g_gpu = {k: gp[k].grad.detach().cpu().clone() for k in keys}
    loss_gpu = loss_g.item()
    state = ({k: v.cpu() for k, v in enc.state_dict().items()}, {k: v.cpu() for k, v in dec.state_dict().items()})

    def cpu_legs():
        from oracle import vi as ovi
        from oracle.encoder import EncoderLSTMOracle
        torch.set_num_threads(host_cores())
        enc_o = EncoderLSTMOracle(OBS + 1, OBS * 2, D)
        dec_o = ovi.DecoderOracle(OBS, D, (T - 1) * synth.STEP, synth.STEP, method="rk4")
        enc_o.load_state_dict(state[0])
        dec_o.load_state_dict(state[1])
        times = []
        for i in range(4):
            t0 = time.perf_counter()
            enc_o.zero_grad()
            dec_o.zero_grad()
            loss_o = ovi.vi_loss(enc_o, dec_o, small, elbo=False)
            loss_o.backward()
            times.append(time.perf_counter() - t0)
            log("full-step cpu rep %d: %.2f s (%d patients)" % (i, times[-1], n_cpu))
        med = statistics.median(times[1:])
        out["cpu_baseline"] = {"value": n_cpu / med, "unit": "trajectories/s", "cores": torch.get_num_threads(), "kind": "port",
                               "sample": "first %d of %d patients, oracle encoder + rk4 + readout + masked SSE (elbo=False), "
                                         "fwd + autograd bwd, 1 warm-up + 3 reps (median %.3f s)" % (n_cpu, N_PER_GPU, med)}
        op = dict(list(enc_o.named_parameters()) + list(dec_o.named_parameters()))
        rels = {k: _rel(g_gpu[k], op[k].grad) for k in keys}
        out["parity"] = {"vs": "CPU oracle pipeline, first %d patients, T=%d, obs=%d, elbo=False" % (n_cpu, T, OBS),
                         "loss_rel": abs(loss_gpu - loss_o.item()) / abs(loss_o.item()), "grad_rel": rels,
                         "tolerance": {"loss_rel": 2e-4, "grad_rel": 2e-3}}
        out["parity"]["ok"] = bool(out["parity"]["loss_rel"] <= 2e-4 and max(rels.values()) <= 2e-3)

    return out, cpu_legs


# ---------------------------------------------------------------------------------------------------------------------
# extra block 2: dopri5 (BASELINE config 3 per GPU)

def dopri5_step(dev, rank, prob, dist=None, iters=3, cpu=True, n_cpu=256, n_par=128):
    """The same patients through the adaptive Dormand-Prince solve (rtol 1e-7, atol 1e-8, per-rank batch-global controller)
    + its tape adjoint, and -- when distributed -- the all-reduce of the parameter gradients.  Max over ranks."""
    import torch
    from hode import adaptive
    inp = prob["inp"]
    theta = prob["theta"].to(dev)
    dosage, times = prob["dosage"].to(dev), prob["times"].to(dev)
    y0 = inp["z0"].to(dev).requires_grad_(True)
    wg, bg = prob["w"].to(dev).requires_grad_(True), prob["b"].to(dev).requires_grad_(True)
    t = inp["t"].to(dev)
    cot = prob["cot"].to(dev)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    split = {"fwd": [], "bwd": []}

    def step(detach=False):
        y0.grad = wg.grad = bg.grad = None
        ev[0].record()
        h = adaptive.roche_dopri5(y0, theta, wg, bg, t, dosage, times, rtol=1e-7, atol=1e-8, detach_first_step=detach)
        ev[1].record()
        (h * cot).sum().backward()
        ev[2].record()
        if dist is not None:
            flat = torch.cat([wg.grad.flatten(), bg.grad.flatten()])
            dist.all_reduce(flat, op=dist.ReduceOp.AVG)
        return h

    step()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        step()
        torch.cuda.synchronize()
        split["fwd"].append(ev[0].elapsed_time(ev[1]))
        split["bwd"].append(ev[1].elapsed_time(ev[2]))
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / iters * 1e3
    if dist is not None:
        tt = torch.tensor([ms], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        ms = float(tt.item())
    st = dict(adaptive.last_stats)
    attempts = st["n_accepted"] + st["n_rejected"]
    fwd_ms, bwd_ms = statistics.mean(split["fwd"]), statistics.mean(split["bwd"])
    world = 1 if dist is None else dist.get_world_size()
    # algorithmic bytes of one attempt: read y_n and f_n (FSAL), write the candidate y_{n+1} and its k7 = 16 D bytes per
    # patient; of the sweep per accepted step: read y_n (4D) -- plus h and grad_h once (8 T D + 4 D, as for rk4)
    att_bytes = N_PER_GPU * 16 * D
    us_att = fwd_ms * 1e3 / max(attempts, 1)
    out = {"ms": ms, "trajectories_per_s": N_PER_GPU * world / ms * 1e3, "rtol": 1e-7, "atol": 1e-8,
           "n_accepted": st["n_accepted"], "n_rejected": st["n_rejected"], "fwd_ms": fwd_ms, "bwd_ms": bwd_ms,
           "what": "dopri5 solve (one launch per attempted step, batch-global controller per rank) + tape adjoint incl. the "
                   "derivative of the first step size" + ("" if dist is None else " + rccl all-reduce of the parameter grads"),
           "roofline": {"bound": "hbm", "kernel": "dp_fwd_kernel<12, 4> (one attempt per launch)", "achieved": att_bytes / (us_att * 1e-6) / 1e9,
                        "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": att_bytes / (us_att * 1e-6) / 1e9 / HBM_PEAK_GBS, "traffic": None,
                        "bytes_per_launch": att_bytes, "avg_launch_us": us_att, "attempts": attempts,
                        "note": "us per attempt = forward wall time on the stream / attempts (launch floor ~1.2 us included); "
                                "the attempt is issue / latency bound, not traffic bound (DESIGN.md section 5)"}}
    if not cpu or rank != 0:
        return out, lambda: None
    # ---- parity: the oracle's accepted-step algebra replayed along THIS run's tape on a slice of the patients (the
    # first step size detached on both sides: its derivative is batch-global and cannot be formed from a slice;
    # tests/test_hip_dopri5.py pins that term on whole batches).  GPU pass now, CPU legs later.
    adaptive.keep_workspace = True
    try:
        h = step(detach=True)
        tape = adaptive.read_tape()
    finally:
        adaptive.keep_workspace = False
    h_gpu, gy0_gpu = h.detach()[:, :n_par].cpu(), y0.grad[:n_par].cpu().clone()

    def cpu_legs():
        from oracle.solvers import odeint, odeint_dopri5_replay
        torch.set_num_threads(host_cores())
        f = oracle_rhs(prob)
        f.set_action(inp["actions"][:, :n_par])
        y0c = inp["z0"][:n_par].clone().requires_grad_(True)
        t0 = time.perf_counter()
        hr = odeint_dopri5_replay(f, y0c, inp["t"], 1e-7, 1e-8, list(zip(tape["t"], tape["dt"])), False)
        (hr * prob["cot"][:, :n_par]).sum().backward()
        log("dopri5 replay oracle on %d patients along %d accepted steps: %.1f s" % (n_par, len(tape["t"]), time.perf_counter() - t0))
        err = (h_gpu - hr.detach()).double()
        out["parity"] = {"vs": "CPU oracle step algebra replayed along this run's (t_n, dt_n) tape, first %d patients, first "
                               "step size detached on both sides" % n_par,
                         "mse": float((err ** 2).mean()), "max_abs": float(err.abs().max()),
                         "grad_y0_rel": _rel(gy0_gpu, y0c.grad), "tolerance": {"mse": 1e-5, "grad_rel": 1e-3}}
        out["parity"]["ok"] = bool(out["parity"]["mse"] <= 1e-5 and out["parity"]["grad_y0_rel"] <= 1e-3)
        # CPU baseline: the free-running oracle (its own controller) on a bounded sample
        f.set_action(inp["actions"][:, :n_cpu])
        times_, stc = [], {}
        for i in range(2):
            y0o = inp["z0"][:n_cpu].clone().requires_grad_(True)
            f.zero_grad()
            t0 = time.perf_counter()
            ho = odeint(f, y0o, inp["t"], method="dopri5", rtol=1e-7, atol=1e-8, stats=stc)
            (ho * prob["cot"][:, :n_cpu]).sum().backward()
            times_.append(time.perf_counter() - t0)
            log("dopri5 cpu rep %d: %.1f s (%d patients, %d + %d attempts)" % (i, times_[-1], n_cpu, stc["n_accepted"], stc["n_rejected"]))
        out["cpu_baseline"] = {"value": n_cpu / times_[-1], "unit": "trajectories/s", "cores": torch.get_num_threads(), "kind": "port",
                               "sample": "first %d of %d patients (own batch-global controller: %d accepted + %d rejected attempts), "
                                         "dopri5 rtol 1e-7 atol 1e-8, fwd + autograd bwd, 1 warm-up + 1 rep (%.1f s)"
                                         % (n_cpu, N_PER_GPU, stc["n_accepted"], stc["n_rejected"], times_[-1])}

    return out, cpu_legs


def other_configs(dev, iters=5):
    """Timings of the paths the headline does not touch, so that every number quoted in DESIGN.md is in the driver's line:
    BASELINE config 5 (DDW-shaped real-data model: one VariationalInferenceReal training step) and the NeuralODE rhs
    (`run_simulation --method=neural`: rk4 and the reference's default dopri5) at the bench shape.  GPU only; parity of
    these paths is held by tests/test_hip_fullsize.py, test_hip_real.py, test_hip_neural.py."""
    import torch
    import model
    from hode import adaptive, synth
    from hode.neural import neural_solve
    out = {}

    def timed(fn):
        fn()
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / iters * 1e3

    # ---- config 5
    obs, act, stat, Dr, Tr, t0r, Br = 24, 1, 11, 20, 120, 24, 8192
    input_dim = obs + act + stat + 1
    torch.manual_seed(0)
    enc = model.EncoderLSTMReal(input_dim, int(input_dim * 1.2), Dr, output_all=False, reverse=False, device=dev)
    dec = model.DecoderReal(obs, Dr, act, stat, int((obs + act + stat) * 1.2), Tr, 1, method="midpoint", ode_step_size=1.0,
                            ode_type="hybrid", t0=t0r, device=dev)
    vi = model.VariationalInferenceReal(enc, dec, elbo=True, t0=t0r)
    g = torch.Generator().manual_seed(1)
    data = {"measurements": torch.randn(Tr, Br, obs, generator=g).to(dev),
            "actions": ((torch.rand(Tr, Br, 1, generator=g) < 0.1).float() * torch.rand(Tr, Br, 1, generator=g)).to(dev),
            "masks": (torch.rand(Tr, Br, obs, generator=g) < 0.5).float().to(dev),
            "statics": torch.rand(Tr, Br, stat, generator=g).to(dev)}

    def real_step():
        for p in vi.parameters():
            p.grad = None
        vi.loss(data).backward()

    ms = timed(real_step)
    out["config5_real_data_step"] = {"ms": ms, "trajectories_per_s": Br / ms * 1e3,
                                     "what": "VariationalInferenceReal loss + backward, %d patients x T=%d (t0=%d), obs 24, statics 11, D=20, "
                                             "encoder 37->44 (MFMA LSTM), RocheODEReal midpoint+perturb (MFMA rhs, weight gradients on chip), "
                                             "fused two-layer readout + masked SSE" % (Br, Tr, t0r)}
    del vi, enc, dec, data
    # ---- NeuralODE rhs at the bench shape
    inp = synth.solver_inputs(N_PER_GPU, T, D)
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(D + 1, 10 * D), torch.nn.Tanh(), torch.nn.Linear(10 * D, D), torch.nn.Tanh())
    prm = [p.detach().clone().to(dev).requires_grad_(True) for p in (net[0].weight, net[0].bias, net[2].weight, net[2].bias)]
    y0 = inp["z0"].to(dev).requires_grad_(True)
    chan = inp["actions"][..., 0]
    dosage = chan.max(dim=0)[0].to(dev)
    times = (torch.nonzero((chan != 0).t())[:, 1].reshape(N_PER_GPU, -1) * synth.STEP).float().to(dev)
    tt = inp["t"].to(dev)
    cot = torch.randn(T, N_PER_GPU, D, device=dev)

    def run(solve):
        def f():
            y0.grad = None
            for q in prm:
                q.grad = None
            (solve() * cot).sum().backward()
        return f

    ms_rk4 = timed(run(lambda: neural_solve(y0, *prm, tt, dosage, times, method="rk4")))
    ms_dp = timed(run(lambda: adaptive.neural_dopri5(y0, *prm, tt, dosage, times, rtol=1e-7, atol=1e-8)))
    st = dict(adaptive.last_stats)
    out["neural_rhs"] = {"rk4_solve_adjoint_ms": ms_rk4, "dopri5_solve_adjoint_ms": ms_dp,
                         "dopri5_attempts": st["n_accepted"] + st["n_rejected"],
                         "what": "NeuralODE rhs (13 -> 120 -> 12 on the matrix cores), %d patients x T=%d, forward + adjoint incl. "
                                 "on-chip weight gradients" % (N_PER_GPU, T)}
    return out


def kernel_times(plan, iters=20):
    """Average duration of the forward kernel and of the adjoint kernel ALONE (no memset, no partial fold -- the quantity
    rocprofv3's kernel stats report), plus the whole backward call, from HIP events on the launch stream."""
    import torch
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


GUARD_EXIT = 17   # exit code of a rank whose guarded phase (graph-captured collective) did not finish in time


class Watchdog:
    """`with Watchdog("what", seconds):` -- if the block does not finish in time the PROCESS exits with GUARD_EXIT.

    A collective that hangs inside a HIP-graph replay is not caught by c10d's watchdog and cannot be cancelled from the
    host; the only safe reaction is to end this rank (stream waits release the GIL, so the timer thread runs) and let the
    supervising parent -- which has never touched the GPU -- start a fresh rank on the fallback path."""

    def __init__(self, what, seconds):
        self.what, self.seconds = what, seconds

    def _fire(self):
        log("guard: '%s' did not finish within %.0f s -- leaving with exit code %d" % (self.what, self.seconds, GUARD_EXIT))
        os._exit(GUARD_EXIT)

    def __enter__(self):
        import threading
        self.t = threading.Timer(self.seconds, self._fire)
        self.t.daemon = True
        self.t.start()
        return self

    def __exit__(self, *exc):
        self.t.cancel()
        return False


def supervise(args, modes):
    """Rank-level guard for N > 1 (runs in every process torch.distributed.run starts, BEFORE anything touches the GPU):
    start the real rank as a child with the first exchange mode; if it leaves with a non-zero code (GUARD_EXIT: its
    graph-captured all-reduce hung, on every rank at once since it is a collective) or exceeds the overall limit, kill
    its process group and start a FRESH child with the next mode.  The children of attempt k > 0 rendezvous under a new
    store prefix.  Rank 0's supervisor relays the successful child's stdout (the one JSON line).  Exits non-zero if
    every mode failed.  No process that has touched the GPU is ever re-executed."""
    import signal
    rank = int(os.environ.get("RANK", "0"))
    limit = float(os.environ.get("HODE_BENCH_GUARD_S", "900"))
    argv = [a for a in sys.argv[1:]]
    for flag in ("--graph-allreduce", "--sync-allreduce"):
        while flag in argv:
            argv.remove(flag)
    if "--grad-exchange" in argv:
        i = argv.index("--grad-exchange")
        del argv[i:i + 2]
    failed = []
    for attempt, mode in enumerate(modes):
        env = dict(os.environ, HODE_BENCH_CHILD="1", HODE_BENCH_ATTEMPT=str(attempt), HODE_BENCH_FAILED_MODES=",".join(failed),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        cmd = [sys.executable, os.path.abspath(__file__)] + argv + ["--grad-exchange", mode]
        log("rank %d supervisor: attempt %d, gradient exchange '%s'" % (rank, attempt, mode))
        p = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, start_new_session=True)
        try:
            out, _ = p.communicate(timeout=limit * (attempt + 1))
            rc = p.returncode
        except subprocess.TimeoutExpired:
            rc = -9
            out = b""
        finally:
            try:
                os.killpg(p.pid, signal.SIGKILL)   # the child's whole process group, by its exact id
            except (ProcessLookupError, PermissionError):
                pass
            p.wait()
        if rc == 0:
            sys.stdout.write(out.decode())
            sys.stdout.flush()
            sys.exit(0)
        log("rank %d supervisor: exchange '%s' failed (exit code %s)%s" % (rank, mode, rc, " = guard" if rc == GUARD_EXIT else ""))
        failed.append(mode)
    sys.exit(1)


class StubPlan:
    """CPU stand-in for RocheRKPlan (tests/test_bench_dist.py: bench.py's N > 1 control flow on gloo, no GPU, no solver)."""

    def __init__(self, rank):
        import torch
        self.rank, self.k = rank, 0
        self.grad_flat = torch.zeros(120)
        self.buckets = [self.grad_flat]
        self.fwd_bytes = self.bwd_bytes = 1

    def capture(self, n_buckets=1):
        import torch
        self.buckets = [self.grad_flat] + [torch.zeros(120) for _ in range(n_buckets - 1)]

    def step(self, bucket=0):
        self.k += 1
        self.buckets[bucket].fill_(float(self.rank + 1))

    def replay(self, bucket=0):
        self.step(bucket)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-cpu", action="store_true", help="skip every CPU leg (baselines and parity)")
    ap.add_argument("--headline-only", action="store_true", help="skip the full_training_step and dopri5_step blocks")
    ap.add_argument("--no-graph", action="store_true", help="launch eagerly instead of replaying a HIP graph")
    ap.add_argument("--lanes", type=int, default=0, help="force lanes per patient (1|4), 0 = library default")
    ap.add_argument("--no-theta-grad", action="store_true", help="skip the 13 expert-constant gradients")
    ap.add_argument("--grad-exchange", choices=("auto", "graph", "overlap", "sync"), default="auto",
                    help="N>1: how the gradient all-reduce is issued.  graph = captured inside the step's HIP graph (no stream "
                         "hand-over per step); overlap = asynchronous on alternating buckets under the next solve; sync = serialised. "
                         "auto = graph, guarded by a supervising parent that falls back to overlap in a fresh process")
    ap.add_argument("--graph-allreduce", action="store_true", help="alias of --grad-exchange graph")
    ap.add_argument("--sync-allreduce", action="store_true", help="alias of --grad-exchange sync")
    ap.add_argument("--no-guard", action="store_true", help="N>1: run the rank in this process (no supervising parent, no fallback)")
    ap.add_argument("--precondition-ms", type=float, default=PRECONDITION_MS,
                    help="untimed graph replays ahead of the warm-up steps until this much wall time has passed (clock ramp)")
    ap.add_argument("--no-tape", action="store_true", help="backward re-integrates the expert stages instead of reading the forward's tape")
    ap.add_argument("--full-step", action="store_true", help="(default now) kept for compatibility")
    ap.add_argument("--dopri5", action="store_true", help="(default now) kept for compatibility")
    args = ap.parse_args()
    if args.graph_allreduce:
        args.grad_exchange = "graph"
    if args.sync_allreduce:
        args.grad_exchange = "sync"

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    stub = bool(os.environ.get("HODE_BENCH_STUB"))   # CPU test of the N > 1 control flow (gloo, StubPlan)
    if world == 1 and args.gpus > 1 and "RANK" not in os.environ:
        self_launch(args)  # does not return
    # HODE_BENCH_FORCE_SUPERVISOR: rehearse the supervisor / fallback machinery at world size 1 (all a one-GPU box allows)
    if (world > 1 or os.environ.get("HODE_BENCH_FORCE_SUPERVISOR")) and "RANK" in os.environ and not os.environ.get("HODE_BENCH_CHILD") \
            and not args.no_guard:
        if args.no_graph and args.grad_exchange in ("auto", "graph"):
            args.grad_exchange = "sync"
        supervise(args, ["graph", "overlap"] if args.grad_exchange == "auto" else [args.grad_exchange])  # does not return
    if args.grad_exchange == "auto":
        args.grad_exchange = "graph" if world > 1 else "overlap"
    if os.environ.get("HODE_BENCH_FAIL_GRAPH") and args.grad_exchange == "graph" and os.environ.get("HODE_BENCH_CHILD"):
        log("HODE_BENCH_FAIL_GRAPH: leaving with the guard's exit code (rehearsal of the fallback)")
        os._exit(GUARD_EXIT)
    attempt = int(os.environ.get("HODE_BENCH_ATTEMPT", "0"))
    failed_modes = [m for m in os.environ.get("HODE_BENCH_FAILED_MODES", "").split(",") if m]

    # The contract is ONE JSON line on stdout.  Native libraries write there too (RCCL prints a version banner on
    # communicator creation): keep the real stdout aside for the result line and point fd 1 at stderr for everything else.
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)

    import torch
    if world != args.gpus:
        sys.exit("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if stub:
        dev = torch.device("cpu")
        sync = lambda: None
    else:
        if not torch.cuda.is_available():
            sys.exit("bench.py: no HIP device visible (the solver path has no CPU fallback)")
        torch.cuda.set_device(local)
        dev = torch.device("cuda", local)
        sync = torch.cuda.synchronize
    dist = None
    n_ranks = 1
    if world > 1 or "RANK" in os.environ:  # launched by torch.distributed.run: always take the distributed path
        import datetime
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = "gloo" if stub else "nccl"   # "nccl" IS RCCL on ROCm
        kw = {} if stub else {"device_id": dev}
        if attempt == 0:
            dist.init_process_group(backend, **kw)
        else:
            # a fallback attempt: the keys of the failed attempt are still in the launcher's store -> new prefix
            agent_store = os.environ.get("TORCHELASTIC_USE_AGENT_STORE") == "True"
            port = int(os.environ["MASTER_PORT"]) + (0 if agent_store else attempt)
            store = dist.TCPStore(os.environ["MASTER_ADDR"], port, world, is_master=(rank == 0 and not agent_store),
                                  timeout=datetime.timedelta(seconds=300))
            dist.init_process_group(backend, store=dist.PrefixStore("hode_bench_attempt%d" % attempt, store), rank=rank,
                                    world_size=world, **kw)
        ones = torch.ones(1, device=dev)
        dist.all_reduce(ones, op=dist.ReduceOp.SUM)  # the rank count as the collective itself sees it
        n_ranks = int(ones.item())
        log("rank %d: %s communicator up (attempt %d), all-reduce of ones = %d ranks" % (rank, backend, attempt, n_ranks))

    def avg(buf, async_op=False):
        if stub:   # gloo has no AVG
            w = dist.all_reduce(buf, op=dist.ReduceOp.SUM, async_op=async_op)
            return w
        return dist.all_reduce(buf, op=dist.ReduceOp.AVG, async_op=async_op)

    if stub:
        prob, plan = None, StubPlan(rank)
    else:
        prob = solver_problem(rank)
        plan = build_plan(dev, prob, lanes=args.lanes, need_theta=not args.no_theta_grad, tape=not args.no_tape)
    use_graph = not args.no_graph
    log("rank %d: plan built (B=%d, T=%d, D=%d)" % (rank, N_PER_GPU, T, D))
    mode = args.grad_exchange if dist is not None else None
    overlap = mode == "overlap" and use_graph
    in_graph = mode == "graph" and use_graph
    if use_graph:
        plan.capture(n_buckets=2 if overlap else 1)
        log("rank %d: graph captured" % rank)
    step_graph = None
    guard_s = float(os.environ.get("HODE_BENCH_PHASE_GUARD_S", "90"))
    if in_graph:
        # The all-reduce captured INSIDE the step's HIP graph (RCCL supports capture): no hand-over between torch's and RCCL's
        # streams per step (140.3 us per step at world size 1 against 155.4 for the overlap path).  A
        # collective that hangs inside a graph replay is invisible to c10d's watchdog, so the first replays run under this
        # rank's own guard and the supervising parent falls back to the overlap path in a fresh process.
        avg(plan.grad_flat)  # communicator warm-up outside capture
        sync()
        with Watchdog("capture + first replays of the step graph with the all-reduce", guard_s):
            if stub:
                if os.environ.get("HODE_BENCH_STUB_HANG") == "graph":
                    time.sleep(1e6)
                step_graph = type("G", (), {"replay": lambda self: (plan.step(), avg(plan.grad_flat))})()
            else:
                step_graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(step_graph):
                    plan.step()
                    dist.all_reduce(plan.grad_flat, op=dist.ReduceOp.AVG)
            for _ in range(3):
                step_graph.replay()
            sync()
        log("rank %d: step graph with the all-reduce captured and replayed" % rank)

    # Data-parallel gradient exchange over xGMI: one RCCL all-reduce(AVG) of the flat bucket per step.
    #   graph   (default for N > 1): the collective is a node of the step's HIP graph
    #   overlap (fallback): issued asynchronously on alternating buckets (one captured graph per bucket), so the exchange of
    #           step k runs on RCCL's stream while the solver kernels of step k+1 run; every exchange completes inside the
    #           timed region (fence())
    #   sync:   replay -> all-reduce -> replay
    from hode.parallel import AlternatingExchange
    exchange = AlternatingExchange(lambda buf: avg(buf, async_op=True)) if overlap else None

    def step():
        if step_graph is not None:
            step_graph.replay()
            return
        if overlap:
            i = exchange.acquire()          # waits (stream-side) for the exchange issued two steps ago on this bucket
            plan.replay(i)                  # the graph captured for bucket i
            exchange.release(i, plan.buckets[i])
            return
        if use_graph:
            plan.replay()
        else:
            plan.step()
        if dist is not None:
            avg(plan.grad_flat)

    def fence():
        if dist is not None:
            if exchange is not None:
                exchange.drain()
            dist.barrier()
        sync()

    # Pre-conditioning (disclosed in config.preconditioning): from idle the card needs ~25 ms of work to reach its clock
    # (profiles/r03_v0_clock_ramp.txt: 160 -> 134 us per step over the first 23 ms), more than the driver's 5 + 20 steps
    # take.  The same step is replayed, untimed, until `--precondition-ms` of wall time have passed; then the W warm-up
    # steps and the K timed steps follow as the contract says.
    n_pre = 0
    with Watchdog("pre-conditioning + warm-up + timed region", max(guard_s, 30.0 + 0.01 * (args.steps + args.warmup))):
        t_pre = time.perf_counter()
        while (time.perf_counter() - t_pre) * 1e3 < args.precondition_ms:
            for _ in range(25):
                step()
            n_pre += 25
            fence() if dist is not None else sync()
        pre_ms = (time.perf_counter() - t_pre) * 1e3
        for _ in range(args.warmup):
            step()
        fence()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        t_enqueued = time.perf_counter() - t0
        fence()
        elapsed = time.perf_counter() - t0
    log("rank %d: %d steps in %.4f s (host had enqueued them after %.4f s; %d pre-conditioning replays in %.0f ms before the warm-up)"
        % (rank, args.steps, elapsed, t_enqueued, n_pre, pre_ms))
    if dist is not None:
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    extras = not args.headline_only and not stub
    cpu = not args.no_cpu and world == 1 and not stub
    # ---- every GPU measurement first (the card clocks down while the host runs the CPU legs), then the CPU legs
    gpu = kt = None
    if rank == 0 and not stub:
        kt = kernel_times(plan)
        # what the timed kernels left in HBM for one step (single rank: the bucket IS this rank's gradient)
        if use_graph:
            plan.replay(0)
        else:
            plan.step()
        torch.cuda.synchronize()
        gpu = {"h": plan.h.clone(), "gy0": plan.grad_y0.clone(), "gw": plan.grad_w.clone(), "gb": plan.grad_b.clone(),
               "gth": plan.grad_theta.clone(), "need_theta": not args.no_theta_grad,
               "what": "split layout%s, %s" % ("" if args.no_tape else " + stage tape", "hipGraph replay" if use_graph else "eager launches")}
    # every rank takes part in the two extra blocks (they contain collectives when distributed)
    fs, fs_cpu = full_training_step(dev, cpu=cpu, dist=dist, rank=rank) if extras else (None, None)
    dp, dp_cpu = dopri5_step(dev, rank, prob, dist, cpu=cpu) if extras else (None, None)
    if rank == 0:
        others = other_configs(dev) if extras and world == 1 else None
        fwd_s, bwd_s, bwd_call_s = kt if kt is not None else (1.0, 1.0, 1.0)
        ms = elapsed / args.steps * 1e3
        total = N_PER_GPU * world
        ach = plan.bwd_bytes / bwd_s / 1e9
        exchange_text = {None: None, "graph": "rccl all-reduce(AVG), captured in the step graph",
                         "overlap": "rccl all-reduce(AVG), async under the next solve", "sync": "rccl all-reduce(AVG), serialised"}[mode]
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
                       "parallelism": "dp%d" % world, "n_ranks": n_ranks,
                       "grad_exchange": exchange_text,
                       "grad_exchange_fallback": ({"failed": failed_modes, "attempt": attempt} if failed_modes else None),
                       "preconditioning": {"replays": n_pre, "ms": pre_ms,
                                           "why": "untimed replays of the same step ahead of the W warm-up steps: from idle the card's "
                                                  "clock ramps for ~25 ms (profiles/r03_v0_clock_ramp.txt)"}},
            "roofline": {"bound": "hbm", "kernel": "split_bwd_kernel<12, rk4> (adjoint kernel alone; the whole backward call incl. "
                                                      "the partial fold is bwd_call_us)", "achieved": ach,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": None,
                         "bytes_per_launch": plan.bwd_bytes, "avg_launch_us": bwd_s * 1e6, "bwd_call_us": bwd_call_s * 1e6,
                         "fwd": {"bytes_per_launch": plan.fwd_bytes, "avg_launch_us": fwd_s * 1e6,
                                 "achieved": plan.fwd_bytes / fwd_s / 1e9, "frac": plan.fwd_bytes / fwd_s / 1e9 / HBM_PEAK_GBS},
                         "step_frac": (plan.fwd_bytes + plan.bwd_bytes) / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBS},
        }
        tr = pmc_traffic(tape=not args.no_tape) if not stub else None
        if tr is not None:
            out["roofline"]["traffic"] = tr["bytes_per_launch"]
            out["roofline"]["traffic_source"] = "committed profile " + tr["source"] + " (rocprofv3 --pmc, not collected in this run)"
            if tr.get("valu_active_frac") is not None:
                # the second roofline of this kernel: it is issue bound: share of the wave's cycles
                # in which it issues a VALU instruction, from the same committed --pmc pass
                out["roofline"]["issue"] = {"valu_active_frac": tr["valu_active_frac"], "source": tr["source"]}
        if not args.no_tape:
            # deliberate recompute <-> traffic trade: the forward leaves the 3 intermediate expert stage
            # states of every step (16 B each) and the backward reads them back instead of re-integrating
            out["roofline"]["tape_bytes_per_launch"] = (T - 1) * 3 * N_PER_GPU * 16 + (T - 1) * N_PER_GPU * (D - 4) * 2 * 4  # expert stage states + 2 learned stage derivatives
            out["config"]["stage_tape"] = True
        if cpu:
            out["cpu_baseline"], out["parity"] = cpu_baseline_and_parity(prob, gpu)
            out["speedup_vs_cpu"] = out["value"] / out["cpu_baseline"]["value"]
        if extras:
            fs_cpu()
            dp_cpu()
            out["full_training_step"] = fs
            out["dopri5_step"] = dp
            if others is not None:
                out["other_configs"] = others
        sys.stdout.flush()
        os.write(result_fd, (json.dumps(out) + "\n").encode())
    os.close(result_fd)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
