#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection CSVs (one pass per directory) per kernel.
usage: pmc_summary.py <dir with sq/ sq2/ fetch/ write/> [samples_channels_per_launch]
FETCH_SIZE / WRITE_SIZE are in KiB; FETCH_SIZE is doubled (gfx950 tallies 128-B read requests at 64 B, MI355X_MICROARCH.md "HBM")."""
import csv, glob, os, re, sys
from collections import defaultdict
d = sys.argv[1]
nsc = float(sys.argv[2]) if len(sys.argv) > 2 else None
agg = defaultdict(lambda: defaultdict(float)); cnt = defaultdict(lambda: defaultdict(int)); meta = {}
for f in glob.glob(os.path.join(d, "*", "*_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        k = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("wam::", "").replace("void ", "")
        k = {"k_warm_punch<true, false>": "k_warm", "k_warm_punch<false, true>": "k_punch"}.get(k, re.sub(r"<.*", "", k))
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
        meta[k] = (r["VGPR_Count"], r["Accum_VGPR_Count"], r["SGPR_Count"], r["LDS_Block_Size"], r["Scratch_Size"], r["Grid_Size"], r["Workgroup_Size"])
# kernels whose launches SPLIT the batch (the exact de-harsh analysis goes through in groups of channels that fit its
# magnitude scratch): all their launches together are one pass, so per-launch figures are summed, not averaged
SPLIT = {"k_dh_bands_x", "k_dh_bandsum"}
cols = ["SQ_WAVES", "SQ_INSTS_VALU", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU",
        "SQ_INSTS_LDS", "SQ_INSTS_SALU", "SQ_WAIT_INST_LDS", "FETCH_SIZE", "WRITE_SIZE"]
w = csv.writer(sys.stdout)
w.writerow(["kernel", "launches", "vgpr", "agpr", "sgpr", "lds", "scratch", "grid", "wg"] + cols +
           ["valu_lane_instr_per_sample_channel", "wait_any_frac", "wait_inst_frac", "active_frac", "fetch_GB_per_launch_x2", "write_GB_per_launch"])
for k in sorted(agg, key=lambda k: -agg[k].get("SQ_WAVE_CYCLES", 0)):
    a = agg[k]; nl = max(cnt[k].values()); n = 1 if k in SPLIT else nl
    wc = a.get("SQ_WAVE_CYCLES", 0) or 1
    row = [k, str(nl)] + list(meta[k]) + ["%.4g" % a.get(c, float("nan")) for c in cols]
    row += ["%.1f" % (a.get("SQ_INSTS_VALU", 0) * 64 / n / nsc) if nsc else "", "%.3f" % (a.get("SQ_WAIT_ANY", 0) / wc), "%.3f" % (a.get("SQ_WAIT_INST_ANY", 0) / wc),
            "%.3f" % (a.get("SQ_ACTIVE_INST_ANY", 0) / wc), "%.3f" % (a.get("FETCH_SIZE", 0) * 2 * 1024 / 1e9 / n), "%.3f" % (a.get("WRITE_SIZE", 0) * 1024 / 1e9 / n)]
    w.writerow(row)
