#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd (.db) kernel trace per kernel: calls, total/avg/min/max duration, share.
Usage: python tools/rocpd_stats.py <results.db> [> profiles/xxx_kernel_stats.txt]"""
import sqlite3
import sys


def main(path):
    c = sqlite3.connect(path)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    kd = next(t for t in tabs if t.startswith("rocpd_kernel_dispatch"))
    ks = next(t for t in tabs if t.startswith("rocpd_info_kernel_symbol"))
    cols = [r[1] for r in c.execute(f"pragma table_info({kd})")]
    scols = [r[1] for r in c.execute(f"pragma table_info({ks})")]
    name_col = "kernel_name" if "kernel_name" in scols else ("display_name" if "display_name" in scols else scols[-1])
    q = f"select s.{name_col}, k.start, k.end from {kd} k join {ks} s on k.kernel_id = s.id"
    agg = {}
    for name, st, en in c.execute(q):
        d = en - st
        a = agg.setdefault(name, [0, 0, 1 << 62, 0])
        a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
    tot = sum(a[1] for a in agg.values())
    print(f"# {path}: {sum(a[0] for a in agg.values())} dispatches, {tot/1e6:.3f} ms total kernel time")
    print(f"{'kernel':70s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} {'pct':>6s}")
    for name, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        short = name if len(name) <= 70 else name[:67] + "..."
        print(f"{short:70s} {a[0]:7d} {a[1]/1e6:10.3f} {a[1]/a[0]/1e3:10.2f} {a[2]/1e3:10.2f} {a[3]/1e3:10.2f} {100*a[1]/tot:6.2f}")


if __name__ == "__main__":
    main(sys.argv[1])
