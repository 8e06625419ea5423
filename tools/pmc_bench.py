#!/usr/bin/env python3
"""HBM bytes per launch of the bench's kernels from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; one counter per
pass, CSV output) over `python bench.py --steps 1 --warmup 0 --no-cpu`.  Units / corrections as MI355X_MICROARCH.md
prescribes and as calibrated by tools/pmc_probe.py on this access width: both counters in KiB, FETCH_SIZE x2 on gfx950.
usage: pmc_bench.py <dir with pmc_FETCH_SIZE/ pmc_WRITE_SIZE/> > profiles/<name>.json"""
import csv, json, sys, collections
root = sys.argv[1]
FETCH_CAL, WRITE_CAL = 2.0, 1.0
def load(counter):
    acc = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f"{root}/pmc_{counter}/p_counter_collection.csv")):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        acc[k][0] += 1
        acc[k][1] += float(r["Counter_Value"]) * 1024.0
    return acc
F, W = load("FETCH_SIZE"), load("WRITE_SIZE")
out = {"method": "rocprofv3 --kernel-trace --pmc <counter> --output-format csv, one pass per counter; KiB units; FETCH_SIZE x2 (gfx950)",
       "kernels": {}}
for k in sorted(F, key=lambda k: -F[k][1]):
    if not (k.startswith("k_") or "k_" in k):
        continue
    n = F[k][0]
    rd, wr = F[k][1] * FETCH_CAL / n, W.get(k, [1, 0.0])[1] * WRITE_CAL / max(1, W.get(k, [1, 0.0])[0])
    out["kernels"][k] = {"launches": n, "hbm_read_bytes_per_launch": rd, "hbm_write_bytes_per_launch": wr, "hbm_bytes_per_launch": rd + wr}
print(json.dumps(out, indent=1))
