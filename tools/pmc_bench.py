#!/usr/bin/env python3
"""Fold the rocprofv3 --pmc passes of tools/pmc_round.sh (one directory per pass, CSV) into one JSON: per kernel, summed
over its launches -- HBM-side bytes (FETCH_SIZE / WRITE_SIZE in KiB; FETCH_SIZE x2 on gfx950 as MI355X_MICROARCH.md
prescribes and tools/pmc_probe.py calibrated on this access width), the SQ issue / wait counters (quad-cycle units for
SQ_WAVE_CYCLES, SQ_WAIT_*, SQ_ACTIVE_INST_*; SQ_INSTS_* count wave instructions), GRBM_GUI_ACTIVE (shader-clock cycles of
the dispatch, reported as the SUM over the 8 XCDs: a 2.2 ms kernel shows 35 M, i.e. 8 x 2.0 GHz x 2.2 ms) and the dispatch durations of the GRBM pass, from which the effective clock and the VALU issue utilisation
follow:  valu_util = SQ_INSTS_VALU x 4 clk / (1024 SIMDs x GRBM_GUI_ACTIVE / 8).
usage: pmc_bench.py <dir with pmc_*/> <ciphertexts per launch> <tag> > profiles/pmc_bench_kernels.json"""
import csv, glob, hashlib, json, os, sys, collections
root, batch, tag = sys.argv[1], int(sys.argv[2]), sys.argv[3]


def source_id():  # same as bench.py source_id(): which sources the profiled library was built from
    R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    h = hashlib.sha256()
    src = os.path.join(R, "toyfhe.jl_amd", "csrc")
    for f in sorted(os.listdir(src)) + [os.path.join("..", "..", "include", "toyfhe_hip.h")]:
        h.update(f.encode())
        h.update(open(os.path.join(src, f), "rb").read())
    return h.hexdigest()[:16]
FETCH_CAL, WRITE_CAL, N_SIMD, N_XCD = 2.0, 1.0, 1024, 8


def short(name):
    return name.split("(")[0].replace("void ", "")


kern = collections.defaultdict(lambda: collections.defaultdict(float))
launch = collections.defaultdict(dict)
for d in sorted(glob.glob(os.path.join(root, "pmc_*"))):
    if not os.path.isdir(d):
        continue
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        rows = list(csv.DictReader(open(f)))
        # steady state: the profiled command runs 3 warm-up steps before the counted one; the dispatches of each
        # kernel's first step (cold clocks, cold caches) are dropped -- the first quarter of its dispatches in this pass
        order = collections.defaultdict(list)
        for r in rows:
            k = short(r["Kernel_Name"])
            if "k_" in k and int(r["Dispatch_Id"]) not in order[k]:
                order[k].append(int(r["Dispatch_Id"]))
        drop = {k: set(sorted(v)[:len(v) // 4]) for k, v in order.items()}
        seen = set()
        for r in rows:
            k = short(r["Kernel_Name"])
            if "k_" not in k or int(r["Dispatch_Id"]) in drop[k]:
                continue
            c = r["Counter_Name"]
            kern[k][c] += float(r["Counter_Value"])
            launch[k].setdefault(c, set()).add(r["Dispatch_Id"])
            if c == "GRBM_GUI_ACTIVE" and r["Dispatch_Id"] not in seen:
                seen.add(r["Dispatch_Id"])
                kern[k]["duration_ns"] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
out = {"method": "rocprofv3 --kernel-trace --pmc <counters> --output-format csv, separate passes (FETCH_SIZE | WRITE_SIZE | 8 SQ counters | "
                 "GRBM_GUI_ACTIVE + SQ_INSTS_LDS/SALU + SQ_WAVES) over `bench.py --steps 1 --warmup 3 --batch 256 --no-cpu --no-ntt` (the first quarter of each kernel's dispatches dropped: steady state); "
                 "FETCH_SIZE/WRITE_SIZE in KiB, FETCH_SIZE x2 (gfx950); per-kernel sums over the launches",
       "batch": batch, "source_id": source_id(), "note": f"{tag}: one launch of each fused kernel covers {batch} ciphertexts", "kernels": {}}
for k, c in sorted(kern.items(), key=lambda kv: -kv[1].get("FETCH_SIZE", 0)):
    n = max(len(s) for s in launch[k].values())
    e = {"launches": n}
    if "FETCH_SIZE" in c:
        e["hbm_read_bytes"] = c["FETCH_SIZE"] * 1024.0 * FETCH_CAL
    if "WRITE_SIZE" in c:
        e["hbm_write_bytes"] = c["WRITE_SIZE"] * 1024.0 * WRITE_CAL
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        e["hbm_bytes"] = e["hbm_read_bytes"] + e["hbm_write_bytes"]
    for name, v in c.items():
        if name not in ("FETCH_SIZE", "WRITE_SIZE"):
            e[name] = v
    if c.get("GRBM_GUI_ACTIVE") and c.get("duration_ns"):
        e["clock_GHz"] = c["GRBM_GUI_ACTIVE"] / N_XCD / c["duration_ns"]
        if "SQ_INSTS_VALU" in c:
            e["valu_issue_util"] = c["SQ_INSTS_VALU"] * 4.0 / (N_SIMD * c["GRBM_GUI_ACTIVE"] / N_XCD)
        if "hbm_bytes" in e:
            e["hbm_GBs_in_profiled_pass"] = e["hbm_bytes"] / c["duration_ns"]
    if c.get("SQ_WAVE_CYCLES"):
        for w in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS"):
            if w in c:
                e[w + "_share_of_wave_cycles"] = c[w] / c["SQ_WAVE_CYCLES"]
    out["kernels"][k] = e
print(json.dumps(out, indent=1))
