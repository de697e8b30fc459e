#!/usr/bin/env python
"""Build libhode.so (gfx950) in-tree with hipcc: `python build_hip.py [-j N] [--force]`."""
import argparse
import concurrent.futures as cf
import hashlib
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(ROOT, "hybrid-ode-neurips-2021_amd", "csrc")
OUT = os.path.join(ROOT, "hybrid-ode-neurips-2021_amd", "hode", "libhode.so")
OBJ = os.path.join(CSRC, "build")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-variable",
         "-Wno-unused-but-set-variable"]
RK_DIMS = (4, 6, 8, 12, 20)
DP_DIMS = (4, 6, 8, 12)
LSTM_TPWS = (1, 2, 3, 4, 5, 6, 8, 10)   # padded hidden sizes 16 * TPW (csrc/hode_lstm_tpw.hip)
# per-unit flags (measured on MI355X)
# -fno-slp-vectorize on the split kernels: packed-fp32 pairing costs more v_mov than it saves (step 0.228 -> 0.207 ms)
EXTRA_FLAGS = {"hode_rk_split": os.environ.get("HODE_SPLIT_FLAGS", "-fno-slp-vectorize").split()}
DP_FLAGS = os.environ.get("HODE_DP_FLAGS", "").split()  # experiments on the dopri5 units only; product builds: empty
EXTRA_FLAGS["hode_dopri5"] = DP_FLAGS
EXTRA_FLAGS["hode_lstm"] = os.environ.get("HODE_LSTM_FLAGS", "").split()  # diagnostics (-DHODE_LSTM_STAMPS); product builds: empty


def units():
    u = [("hode_api", os.path.join(CSRC, "hode_api.hip"), [])]
    for d in RK_DIMS:
        u.append(("hode_rk_d%d" % d, os.path.join(CSRC, "hode_rk_dim.hip"), ["-DHODE_DIM=%d" % d]))
    for d in DP_DIMS:
        u.append(("hode_dp_d%d" % d, os.path.join(CSRC, "hode_dopri5_dim.hip"), ["-DHODE_DIM=%d" % d] + DP_FLAGS))
    for n in LSTM_TPWS:
        u.append(("hode_lstm_tpw%d" % n, os.path.join(CSRC, "hode_lstm_tpw.hip"), ["-DHODE_LSTM_TPW=%d" % n] + EXTRA_FLAGS["hode_lstm"]))
    for name in ("hode_dopri5", "hode_lstm", "hode_neural", "hode_real", "hode_rk_mf", "hode_readout", "hode_rk_split", "hode_crps", "hode_mckl", "hode_neural_mf", "hode_real_mf", "hode_neural_dopri5", "hode_readout_mlp"):
        src = os.path.join(CSRC, name + ".hip")
        if os.path.exists(src):
            u.append((name, src, EXTRA_FLAGS.get(name, [])))
    return u


def newest_dep():
    ts = [os.path.getmtime(os.path.join(ROOT, "include", "hode.h")), os.path.getmtime(__file__)]
    for f in os.listdir(CSRC):
        if f.endswith((".hpp", ".hip", ".h")):
            ts.append(os.path.getmtime(os.path.join(CSRC, f)))
    return max(ts)


def source_digest():
    """sha256 over everything the library is built from (sources, ABI header, flags).  Written next to libhode.so after a
    build; tests/test_abi.py compares, so a library left over from other sources (e.g. after `git checkout`) is caught."""
    h = hashlib.sha256()
    files = [os.path.join(ROOT, "include", "hode.h")] + sorted(
        os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hpp", ".hip", ".h")))
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    h.update(repr((FLAGS, sorted(EXTRA_FLAGS.items()), DP_FLAGS, RK_DIMS, DP_DIMS, LSTM_TPWS)).encode())
    return h.hexdigest()


def _deps_newest(obj, src):
    """Newest mtime among the files `obj` was compiled from (the -MD depfile hipcc left next to it), the ABI header and this
    script; None if there is no usable depfile (then the unit is rebuilt)."""
    dfile = obj[:-2] + ".d"
    try:
        txt = open(dfile).read()
    except OSError:
        return None
    deps = [x for x in txt.replace("\\\n", " ").split() if not x.endswith(":")]
    ts = [os.path.getmtime(__file__), os.path.getmtime(src)]
    for d in deps:
        if d.startswith("/opt/") or d.startswith("/usr/"):
            continue   # toolchain headers do not change inside a container
        try:
            ts.append(os.path.getmtime(d))
        except OSError:
            return None
    return max(ts)


def compile_one(name, src, extra, force, dep_time):
    obj = os.path.join(OBJ, name + ".o")
    flags_txt = " ".join(FLAGS + extra)
    stamp = obj[:-2] + ".flags"
    if not force and os.path.exists(obj):
        newest = _deps_newest(obj, src)
        try:
            same_flags = open(stamp).read() == flags_txt
        except OSError:
            same_flags = False
        if newest is not None and same_flags and os.path.getmtime(obj) >= newest:
            return name, 0.0, ""
    t0 = time.time()
    cmd = [HIPCC] + FLAGS + extra + ["-MD", "-MF", obj[:-2] + ".d", "-c", src, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (name, " ".join(cmd), r.stderr[-6000:]))
    with open(stamp, "w") as f:
        f.write(flags_txt)
    return name, time.time() - t0, r.stderr


def build(jobs=7, force=False, verbose=True):
    os.makedirs(OBJ, exist_ok=True)
    dep = newest_dep()
    us = units()
    with cf.ThreadPoolExecutor(max_workers=jobs) as ex:
        futs = [ex.submit(compile_one, n, s, e, force, dep) for n, s, e in us]
        for f in futs:
            name, dt, err = f.result()
            if verbose and dt:
                print("  hipcc %-14s %.1fs" % (name, dt), flush=True)
            if verbose and err.strip():
                print(err[-2000:], file=sys.stderr)
    objs = [os.path.join(OBJ, n + ".o") for n, _, _ in us]
    if force or not os.path.exists(OUT) or os.path.getmtime(OUT) < max(os.path.getmtime(o) for o in objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-Wl,--no-undefined", "-o", OUT] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n" + r.stderr[-4000:])
        if verbose:
            print("  linked", os.path.relpath(OUT, ROOT), flush=True)
    with open(OUT + ".digest", "w") as f:
        f.write(source_digest() + "\n")
    return OUT


def build_variant(tag, unit_flags, verbose=True):
    """Experiment builds (never loaded by the product): libhode_<tag>.so = the main build's objects with the units named in
    `unit_flags` ({unit: [extra flags]}) recompiled with those flags ON TOP of their product flags.  Select it at run time
    with HODE_LIBRARY=<path>.  The main library must be built first."""
    build(verbose=False)
    objs = []
    for name, src, extra in units():
        if name in unit_flags:
            obj = os.path.join(OBJ, "%s__%s.o" % (name, tag))
            cmd = [HIPCC] + FLAGS + extra + list(unit_flags[name]) + ["-c", src, "-o", obj]
            t0 = time.time()
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode != 0:
                raise RuntimeError("hipcc failed for variant %s of %s:\n%s" % (tag, name, r.stderr[-6000:]))
            if verbose:
                print("  hipcc %s [%s] %.1fs" % (name, tag, time.time() - t0), flush=True)
            objs.append(obj)
        else:
            objs.append(os.path.join(OBJ, name + ".o"))
    out = os.path.join(os.path.dirname(OUT), "libhode_%s.so" % tag)
    r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-Wl,--no-undefined", "-o", out] + objs,
                       capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n" + r.stderr[-4000:])
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("-j", type=int, default=7)
    ap.add_argument("--force", action="store_true")
    ap.add_argument("--variant", help="tag of an experiment build (libhode_<tag>.so), with --unit-flags")
    ap.add_argument("--unit-flags", action="append", default=[], help='unit="extra flags" (repeatable), e.g. hode_rk_split="-DHODE_SPLIT_WPE_FWD=2"')
    a = ap.parse_args()
    if a.variant:
        uf = {}
        for item in a.unit_flags:
            k, v = item.split("=", 1)
            uf[k] = v.split()
        print(build_variant(a.variant, uf))
    else:
        build(a.j, a.force)
