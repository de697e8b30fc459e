"""bench.py's N > 1 control flow on the CPU (gloo, world size 2, solver stubbed: HODE_BENCH_STUB): the launcher contract
(`python -m torch.distributed.run ... bench.py --gpus N`), the rank-level supervisor, the guarded graph-captured exchange
and its fallback to the overlap path in FRESH rank processes, and the self-launch of `python bench.py --gpus N`.
Nothing here touches a GPU or the solver; what is exercised is the branch structure a multi-GPU node will see first."""
import json
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(cmd, extra_env, timeout=240):
    env = dict(os.environ, HODE_BENCH_STUB="1", **extra_env)
    env.pop("RANK", None)
    env.pop("WORLD_SIZE", None)
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout, cwd=ROOT)
    lines = [l for l in r.stdout.decode().splitlines() if l.strip()]
    return r, lines


def _launcher(port, n=2):
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
            "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "5", "--warmup", "2",
            "--precondition-ms", "1"]


def test_two_ranks_default_to_the_graph_captured_exchange():
    r, lines = _run(_launcher(_free_port()), {})
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    assert len(lines) == 1, lines   # ONE JSON line, from rank 0
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["n_ranks"] == 2 and out["steps"] == 5 and out["warmup"] == 2
    assert out["config"]["patients_total"] == 2 * 10000 and out["scaling"] == "weak"
    assert "captured in the step graph" in out["config"]["grad_exchange"]
    assert out["config"]["grad_exchange_fallback"] is None
    assert out["config"]["preconditioning"]["replays"] >= 25


def test_a_hanging_graph_exchange_falls_back_to_overlap_in_fresh_ranks():
    """The stubbed graph-captured collective never returns: every rank's own guard ends it (exit code 17), every
    supervisor starts a fresh rank on the overlap path under a new store prefix, and the line says so."""
    r, lines = _run(_launcher(_free_port()), {"HODE_BENCH_STUB_HANG": "graph", "HODE_BENCH_PHASE_GUARD_S": "3"})
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    assert len(lines) == 1, lines
    out = json.loads(lines[0])
    assert "async under the next solve" in out["config"]["grad_exchange"]
    assert out["config"]["grad_exchange_fallback"] == {"failed": ["graph"], "attempt": 1}
    assert out["config"]["n_ranks"] == 2
    err = r.stderr.decode()
    assert "guard:" in err and "exit code 17" in err


def test_every_mode_failing_exits_non_zero():
    r, lines = _run(_launcher(_free_port()) + ["--grad-exchange", "graph"], {"HODE_BENCH_STUB_HANG": "graph", "HODE_BENCH_PHASE_GUARD_S": "3"})
    assert r.returncode != 0 and lines == []


def test_self_launch_without_a_launcher():
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--precondition-ms", "1",
           "--grad-exchange", "sync"]
    r, lines = _run(cmd, {})
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    out = json.loads(lines[-1])
    assert out["n_gpus"] == 2 and "serialised" in out["config"]["grad_exchange"]
