"""Fold rocprofv3 --pmc CSVs (one pass per counter group) into profiles/<tag>_pmc_summary.json.

HBM bytes per launch follow MI355X_MICROARCH.md section HBM: FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports
half the bytes of a wide (16 B/lane) coalesced read stream, so the read side is doubled; WRITE_SIZE is exact."""
import collections, csv, glob, json, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
out = {}
for f in glob.glob(os.path.join(ROOT, "gpurun_out", "pmc_*", "*", "*_counter_collection.csv")):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "hode::" in r["Kernel_Name"]:
            agg[(r["Kernel_Name"].split("(")[0].replace("void ", ""), r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in agg.items():
        out.setdefault(k, {})[c] = sum(v) / len(v)
        out[k]["launches_averaged"] = len(v)
for k, d in out.items():
    if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
        d["hbm_read_bytes_corrected"] = 2.0 * d["FETCH_SIZE"] * 1024
        d["hbm_write_bytes"] = d["WRITE_SIZE"] * 1024
        d["hbm_bytes_per_launch"] = d["hbm_read_bytes_corrected"] + d["hbm_write_bytes"]
    if "SQ_INSTS_VALU" in d and "SQ_WAVES" in d:
        d["valu_insts_per_wave"] = d["SQ_INSTS_VALU"] / d["SQ_WAVES"]
        d["wave_cycles_per_valu_inst"] = 4.0 * d["SQ_WAVE_CYCLES"] / d["SQ_INSTS_VALU"]  # SQ_WAVE_CYCLES counts quad-cycles
        d["valu_active_fraction_of_wave_cycles"] = d["SQ_ACTIVE_INST_VALU"] / d["SQ_WAVE_CYCLES"]
dst = os.path.join(ROOT, "profiles", "%s_pmc_summary.json" % tag)
json.dump(out, open(dst, "w"), indent=1, sort_keys=True)
print(json.dumps(out, indent=1, sort_keys=True))
