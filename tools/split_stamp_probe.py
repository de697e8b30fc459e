"""Which wave of a split-adjoint workgroup arrives last at the step barrier (s_memtime stamps, block 0).
Needs a stamp build (product builds carry no stamps):
    python build_hip.py --variant stamps --unit-flags hode_rk_split="-DHODE_SPLIT_STAMPS"
    HODE_LIBRARY=.../hode/libhode_stamps.so python tools/split_stamp_probe.py"""
import os, sys, torch, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "hybrid-ode-neurips-2021_amd"))
import bench
dev = torch.device("cuda:0")
T = bench.T
dbg = torch.zeros(6 * (T + 2) + 64 + 8 * (T + 2), dtype=torch.int64, device=dev)
os.environ["HODE_SPLIT_DBG_PTR"] = hex(dbg.data_ptr())
dbgf = torch.zeros(4 * T, dtype=torch.int64, device=dev)
os.environ["HODE_SPLIT_FWD_DBG_PTR"] = hex(dbgf.data_ptr())
prob = bench.solver_problem(0)
plan = bench.build_plan(dev, prob, lanes=0, need_theta=True, tape=True)
for _ in range(3):
    plan.step()
torch.cuda.synchronize()
raw = dbg.cpu().numpy().astype(np.int64)
s = raw[:6 * (T + 2)].reshape(6, T + 2)[:, :T]
hw = raw[6 * (T + 2):6 * (T + 2) + 64].reshape(8, 8)[:, :6]
print('SIMD of waves 0..5 (expert, learned x3, theta, c-wave) in workgroups 0..7:', [[int((v >> 4) & 3) for v in row] for row in hw])
names = ["expert", "learned 1", "learned 2", "learned 3", "theta", "c-wave"]
ks = np.arange(10, T - 10)
arr = s[:, ks]                      # arrival of each wave at the barrier of iteration k
last = arr.max(axis=0)
period = np.diff(last)
print("step period (last arrival to last arrival): median %d cycles = %.0f ns" % (np.median(period), np.median(period) / 2.4))
for w in range(6):
    slack = last - arr[w]
    print("%-10s arrives %5d cycles (median) before the last wave; last in %2d %% of the steps" % (names[w], np.median(slack), 100 * np.mean(arr[w] == last)))
ph = raw[6 * (T + 2) + 64:].reshape(T + 2, 8)[10:T - 10]
names_ph = ["iteration start -> ring reads issued", "-> stage states + recompute done (adjoint starts)", "-> VJP of stage 3 entered",
            "-> VJP of stage 2 entered", "-> VJP of stage 1 entered", "-> VJP of stage 0 entered", "-> adjoint done"]
if ph[:, 0].any():   # phase slots: 0 start, 1 reads, 2 adjoint start, 3 = VJP s=3, 4 = s=2, 5 = s=1, 6 = s=0, 7 end
    print("learned wave 3, phases of an iteration (median cycles; the stamps themselves add ~40 % to the step):")
    for i, nm in enumerate(names_ph):
        print("   %-52s %5d" % (nm, np.median(ph[:, i + 1] - ph[:, i])))
    print("   %-52s %5d" % ("whole iteration body (start -> adjoint done)", np.median(ph[:, 7] - ph[:, 0])))
    print("   %-52s %5d" % ("adjoint done -> next iteration start (barrier)", np.median(ph[1:, 0] - ph[:-1, 7])))
print("forward kernel:")
sf = dbgf.cpu().numpy().astype(np.int64).reshape(4, T)
ks = np.arange(10, T - 10)
arr = sf[:, ks]
last = arr.max(axis=0)
print("step period: median %d cycles = %.0f ns" % (np.median(np.diff(last)), np.median(np.diff(last)) / 2.4))
for w in range(4):
    print("%-10s arrives %5d cycles (median) before the last wave; last in %2d %% of the steps" % (names[w], np.median(last - arr[w]), 100 * np.mean(arr[w] == last)))
