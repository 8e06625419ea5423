#!/usr/bin/env python3
"""Summarise the two rocprofv3 --pmc passes of tools/pmc_probe.py (FETCH_SIZE, WRITE_SIZE) into HBM bytes per
limb-NTT.  Corrections per MI355X_MICROARCH.md (HBM / rocprofv3 section): both counters are in KiB; on gfx950
FETCH_SIZE reports half of the bytes of wide coalesced reads -- calibrated here on the probe's own row copy
(k_select, 8 bytes per lane, known byte count), whose measured/true ratio is applied to the NTT kernels.
usage: pmc_summarize.py <dir with pmc_FETCH_SIZE/ pmc_WRITE_SIZE/> <rows> > profiles/<name>.json"""
import csv, json, sys
root, rows = sys.argv[1], int(sys.argv[2])
true_bytes = rows * 16384 * 8
def load(counter):
    out = {}
    for r in csv.DictReader(open(f"{root}/pmc_{counter}/p_counter_collection.csv")):
        k = r["Kernel_Name"]
        key = "copy" if k.startswith("k_select") else "fwd" if "k_ntt_fwd" in k else "inv" if "k_ntt_inv" in k else None
        if key: out[key] = {"kernel": k.split("(")[0], "kib": float(r["Counter_Value"])}
    return out
F, W = load("FETCH_SIZE"), load("WRITE_SIZE")
fcal = true_bytes / (F["copy"]["kib"] * 1024.0)       # ~2.0 on gfx950
wcal = true_bytes / (W["copy"]["kib"] * 1024.0)       # ~1.0
res = {"rows": rows, "algorithmic_bytes_per_limb_ntt": 2 * 16384 * 8, "fetch_calibration": fcal, "write_calibration": wcal,
       "method": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE in separate passes over tools/pmc_probe.py; KiB units; "
                 "calibrated on the probe's k_select row copy of known size"}
for key in ("fwd", "inv"):
    rd, wr = F[key]["kib"] * 1024.0 * fcal, W[key]["kib"] * 1024.0 * wcal
    res[key] = {"kernel": F[key]["kernel"], "hbm_read_bytes_per_limb_ntt": rd / rows, "hbm_write_bytes_per_limb_ntt": wr / rows,
                "hbm_bytes_per_limb_ntt": (rd + wr) / rows, "ratio_to_algorithmic": (rd + wr) / rows / (2 * 16384 * 8)}
print(json.dumps(res, indent=1))
