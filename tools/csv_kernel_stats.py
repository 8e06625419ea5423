#!/usr/bin/env python3
"""Per-kernel summary (calls, total, average, min, max, share) of a rocprofv3 --kernel-trace CSV run.
usage: csv_kernel_stats.py <dir containing *kernel_trace.csv> > profiles/<tag>_bench_kernel_stats.txt"""
import csv, glob, os, sys
root = sys.argv[1]
files = glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True)
agg = {}
for f in files:
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "")
        d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        a = agg.setdefault(name, [0, 0, 1 << 62, 0])
        a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
tot = sum(a[1] for a in agg.values()) or 1
print(f"# rocprofv3 --kernel-trace --stats: {sum(a[0] for a in agg.values())} dispatches, {tot / 1e6:.3f} ms total kernel time ({', '.join(os.path.basename(f) for f in files)})")
print(f"{'kernel':70s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} {'pct':>6s}")
for name, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    short = name if len(name) <= 70 else name[:67] + "..."
    print(f"{short:70s} {a[0]:7d} {a[1] / 1e6:10.3f} {a[1] / a[0] / 1e3:10.2f} {a[2] / 1e3:10.2f} {a[3] / 1e3:10.2f} {100 * a[1] / tot:6.2f}")
