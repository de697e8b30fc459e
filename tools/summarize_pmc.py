"""Fold rocprofv3 --pmc CSVs (one pass per counter group) into profiles/<tag>_pmc_summary.json.

    python tools/summarize_pmc.py <tag> [<dir with pmc_*/ sub-directories>, default gpurun_out] [--stats <kernel_stats.csv>]

HBM bytes per launch follow MI355X_MICROARCH.md section HBM: FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports
half the bytes of a wide (16 B/lane) coalesced read stream, so the read side is doubled; WRITE_SIZE is exact.
Matrix-pipe share: SQ_VALU_MFMA_BUSY_CYCLES counts cycles per SIMD-MFMA (32 per v_mfma_f32_16x16x4_f32), GRBM_GUI_ACTIVE is the
sum over the 8 XCDs of the cycles the launch was active: busy / (1024 SIMDs x GUI_ACTIVE / 8) is the clock-independent
fraction of ALL the chip's SIMD cycles in which the matrix pipe worked (idle CUs count against it)."""
import collections, csv, glob, json, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = [a for a in sys.argv[1:]]
stats = None
if "--stats" in args:
    i = args.index("--stats")
    stats = args[i + 1]
    del args[i:i + 2]
tag = args[0] if args else "r01"
base = args[1] if len(args) > 1 else os.path.join(ROOT, "gpurun_out")
N_SIMD, N_XCD = 1024, 8


def short(name):
    name = name.split("(")[0].replace("void ", "")
    return name if len(name) <= 80 else name[:77] + "..."


out = {}
for f in glob.glob(os.path.join(base, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "hode::" in r["Kernel_Name"] or r["Kernel_Name"].startswith("Cijk_"):
            agg[(short(r["Kernel_Name"]), r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in agg.items():
        out.setdefault(k, {})[c] = sum(v) / len(v)
        out[k]["launches_averaged"] = len(v)
dur = {}
if stats:
    for r in csv.DictReader(open(stats)):
        dur[short(r["Name"])] = float(r["AverageNs"])
for k, d in out.items():
    if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
        d["hbm_read_bytes_corrected"] = 2.0 * d["FETCH_SIZE"] * 1024
        d["hbm_write_bytes"] = d["WRITE_SIZE"] * 1024
        d["hbm_bytes_per_launch"] = d["hbm_read_bytes_corrected"] + d["hbm_write_bytes"]
    if "SQ_INSTS_VALU" in d and "SQ_WAVES" in d:
        d["valu_insts_per_wave"] = d["SQ_INSTS_VALU"] / d["SQ_WAVES"]
        d["wave_cycles_per_valu_inst"] = 4.0 * d["SQ_WAVE_CYCLES"] / d["SQ_INSTS_VALU"]  # SQ_WAVE_CYCLES counts quad-cycles
        d["valu_active_fraction_of_wave_cycles"] = d["SQ_ACTIVE_INST_VALU"] / d["SQ_WAVE_CYCLES"]
    if "SQ_VALU_MFMA_BUSY_CYCLES" in d and "GRBM_GUI_ACTIVE" in d:
        d["mfma_busy_fraction_of_chip_simd_cycles"] = d["SQ_VALU_MFMA_BUSY_CYCLES"] / (N_SIMD * d["GRBM_GUI_ACTIVE"] / N_XCD)
        if d.get("SQ_INSTS_MFMA"):
            d["mfma_busy_cycles_per_mfma"] = d["SQ_VALU_MFMA_BUSY_CYCLES"] / d["SQ_INSTS_MFMA"]
    if "SQ_LDS_BANK_CONFLICT" in d and d.get("SQ_LDS_IDX_ACTIVE"):
        d["lds_bank_conflict_fraction_of_lds_cycles"] = d["SQ_LDS_BANK_CONFLICT"] / d["SQ_LDS_IDX_ACTIVE"]
    if k in dur:
        d["avg_duration_ns_kernel_trace_same_call"] = dur[k]
        if "GRBM_GUI_ACTIVE" in d:
            d["effective_clock_ghz"] = d["GRBM_GUI_ACTIVE"] / N_XCD / dur[k]
dst = os.path.join(ROOT, "profiles", "%s_pmc_summary.json" % tag)
json.dump(out, open(dst, "w"), indent=1, sort_keys=True)
print(json.dumps({k: {c: v for c, v in d.items() if not c.isupper()} for k, d in out.items()}, indent=1, sort_keys=True))
