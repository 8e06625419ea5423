#!/usr/bin/env python3
"""Dispatch count and kernel time of the LAST evaluation pass in a rocprofv3 kernel trace of tools/prof_mnist_eval.py: the pass
ends with the decryption's k_ckks_decode_start; it starts after the previous pass's one.
usage: csv_last_pass.py <dir with *kernel_trace.csv>"""
import csv, glob, os, sys, collections
f = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
ends = [i for i, r in enumerate(rows) if "k_ckks_decode_start" in r["Kernel_Name"]]
a, b = ends[-2] + 1, ends[-1]
sel = [r for r in rows[a:b + 1] if "k_ckks_gather" not in r["Kernel_Name"]]
agg = collections.Counter(); t = collections.Counter()
for r in sel:
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")[:60]
    agg[k] += 1; t[k] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
span = (int(sel[-1]["End_Timestamp"]) - int(sel[0]["Start_Timestamp"])) / 1e6
print(f"# last evaluation pass: {len(sel)} dispatches, {sum(t.values()) / 1e6:.2f} ms kernel time in a {span:.2f} ms window")
for k, _ in t.most_common(24):
    print(f"{k:62s} {agg[k]:6d} {t[k] / 1e6:9.3f} ms")
