#!/usr/bin/env python3
"""Fold the rocprofv3 runs of tools/pmc_configs.sh into one JSON: per case of tools/bench_configs.py and per kernel of that
case -- duration (mean over the launches of the un-instrumented --kernel-trace run, first quarter dropped: steady state), and per launch, from
the --pmc passes with the first quarter of each kernel's dispatches dropped (warm-up): HBM-side bytes (FETCH_SIZE x2 on gfx950 +
WRITE_SIZE, both in KiB: MI355X_MICROARCH.md, HBM section; calibration tools/pmc_probe.py), wave64 VALU instructions, the SQ
issue / wait shares (quad-cycle counters over SQ_WAVE_CYCLES) and the shader clock of the profiled launches
(GRBM_GUI_ACTIVE / 8 XCDs / duration).  Derived per kernel:
  valu_frac        = SQ_INSTS_VALU / mean duration / 614.4 G/s  (1024 SIMDs x 2.4 GHz / 4 clk per wave64 instruction)
  valu_issue_util  = SQ_INSTS_VALU x 4 / (1024 x GRBM_GUI_ACTIVE / 8)   (issue slots used at the clock the launch ran at)
  hbm_frac         = HBM-side bytes / mean duration / 8 TB/s
  bound            = the larger of the two
usage: pmc_configs.py <dir with case*/> <tag> [--table]"""
import collections
import csv
import glob
import hashlib
import json
import os
import statistics
import sys

root, tag = sys.argv[1], sys.argv[2]
table = "--table" in sys.argv
FETCH_CAL, WRITE_CAL, N_SIMD, N_XCD, PEAK_HZ, HBM_PEAK = 2.0, 1.0, 1024, 8, 2.4e9, 8e12
VALU_PEAK = N_SIMD * PEAK_HZ / 4


GATHER_KERNELS = ("k_md_acc", "k_md_special_perm", "k_galois", "k_ks_top_tail_rot", "k_ks_rot_tail", "k_ntt_perm", "k_ckks_gather")


def source_id():  # same as bench.py source_id()
    R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    h = hashlib.sha256()
    src = os.path.join(R, "toyfhe.jl_amd", "csrc")
    for f in sorted(os.listdir(src)) + [os.path.join("..", "..", "include", "toyfhe_hip.h")]:
        h.update(f.encode())
        h.update(open(os.path.join(src, f), "rb").read())
    return h.hexdigest()[:16]


def short(name):
    return name.split("(")[0].replace("void ", "")


def steady(rows):
    """rows of one kernel in dispatch order with the first quarter dropped"""
    rows = sorted(rows, key=lambda r: r[0])
    return rows[len(rows) // 4:] if len(rows) >= 4 else rows


out = {"method": "tools/pmc_configs.sh: per case one un-instrumented rocprofv3 --kernel-trace run (durations: mean per kernel over its steady launches) and four "
                 "--pmc passes in profile mode (FETCH_SIZE | WRITE_SIZE | 8 SQ counters | GRBM_GUI_ACTIVE + SQ_INSTS_LDS/SALU + SQ_WAVES), "
                 "first quarter of each kernel's dispatches dropped; FETCH_SIZE / WRITE_SIZE in KiB, FETCH_SIZE x2 (gfx950)",
       "tag": tag, "source_id": source_id(), "valu_peak_G_per_s": VALU_PEAK / 1e9, "hbm_peak_GBs": HBM_PEAK / 1e9, "cases": {}}
for cdir in sorted(glob.glob(os.path.join(root, "case*")), key=lambda p: int(os.path.basename(p)[4:])):
    case = os.path.basename(cdir)[4:]
    rec = None
    for line in open(os.path.join(cdir, "trace.log"), errors="replace"):
        if line.startswith("{"):
            rec = json.loads(line)
    dur = collections.defaultdict(list)
    is_mnist = "mnist" in str((rec or {}).get("config", "")).lower()
    passes = [3]   # evaluation passes inside the counted window of a MNIST case (below)

    def in_window(all_rows):
        """MNIST cases (r06): only the dispatches of the evaluation passes AFTER the first one -- between the first and the last
        k_ckks_decode_start of the file (a pass ends with the decryption's decode).  The set-up of the profiled process (key generation:
        one-polynomial transforms by the hundred, encryption, weight encoding) and the first pass (which encodes the weights) stay out:
        VERDICT r05 read the set-up's launch counts as the pass's."""
        if not is_mnist:
            return all_rows
        marks = sorted({int(r["Dispatch_Id"]) for r in all_rows if "k_ckks_decode_start" in r["Kernel_Name"]})
        if len(marks) < 2:
            return all_rows
        passes[0] = len(marks) - 1
        return [r for r in all_rows if marks[0] < int(r["Dispatch_Id"]) <= marks[-1]]
    for f in glob.glob(os.path.join(cdir, "trace", "**", "*kernel_trace.csv"), recursive=True):
        rows = collections.defaultdict(list)
        for r in in_window(list(csv.DictReader(open(f)))):
            rows[short(r["Kernel_Name"])].append((int(r["Dispatch_Id"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
        for k, v in rows.items():   # steady state: the first quarter of each kernel's dispatches dropped, as in the counter passes
            dur[k] += [d for _, d in steady([(i, d) for i, d in v])]
    cnt = collections.defaultdict(lambda: collections.defaultdict(list))   # kernel -> counter -> [(dispatch, value, dur)]
    for d in sorted(glob.glob(os.path.join(cdir, "pmc_*"))):
        if not os.path.isdir(d):
            continue
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for r in in_window(list(csv.DictReader(open(f)))):
                cnt[short(r["Kernel_Name"])][r["Counter_Name"]].append(
                    (int(r["Dispatch_Id"]), float(r["Counter_Value"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    total = sum(sum(v) for k, v in dur.items() if "k_" in k) or 1
    kernels = {}
    for k, ds in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
        if "k_" not in k or k.startswith("k_sample"):
            continue
        med = sum(ds) / len(ds)   # MEAN over the steady launches: a kernel is launched with several shapes per call (chunks, special
                                  # limb / ciphertext limbs), and the counter passes see the same mix -- means match, medians do not
        e = {"launches": len(ds), "mean_us": med / 1e3, "median_us": statistics.median(ds) / 1e3, "share_of_case": sum(ds) / total}
        c = {}
        for name, rows in cnt.get(k, {}).items():
            st = steady(rows)
            c[name] = sum(v for _, v, _ in st) / len(st)
            if name == "GRBM_GUI_ACTIVE":
                c["_pmc_dur_ns"] = sum(t for _, _, t in st) / len(st)
        if "FETCH_SIZE" in c:
            e["hbm_read_bytes"] = c["FETCH_SIZE"] * 1024.0 * FETCH_CAL
        if "WRITE_SIZE" in c:
            e["hbm_write_bytes"] = c["WRITE_SIZE"] * 1024.0 * WRITE_CAL
        # FETCH_SIZE x2 is calibrated (MI355X_MICROARCH.md) on 16 B/lane streaming reads only; the row kernels confirm it (1.01-1.03 x
        # their algorithmic bytes).  Kernels that gather or scatter 8 B per lane are marked: their byte figures are counter readings,
        # not calibrated traffic.
        if any(t in k for t in GATHER_KERNELS):
            e["byte_calibration"] = "uncalibrated (8 B/lane gather / scatter)"
        if "hbm_read_bytes" in e and "hbm_write_bytes" in e:
            e["hbm_bytes"] = e["hbm_read_bytes"] + e["hbm_write_bytes"]
            e["hbm_GBs"] = e["hbm_bytes"] / med
            e["hbm_frac"] = e["hbm_bytes"] / (med * 1e-9) / HBM_PEAK
        for name in ("SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_INSTS_SALU", "SQ_WAVES"):
            if name in c:
                e[name] = c[name]
        if "SQ_INSTS_VALU" in c:
            e["valu_frac"] = c["SQ_INSTS_VALU"] / (med * 1e-9) / VALU_PEAK
            if "SQ_WAVES" in c and c["SQ_WAVES"]:
                e["valu_per_wave"] = c["SQ_INSTS_VALU"] / c["SQ_WAVES"]
        if c.get("GRBM_GUI_ACTIVE") and c.get("_pmc_dur_ns"):
            e["clock_GHz"] = c["GRBM_GUI_ACTIVE"] / N_XCD / c["_pmc_dur_ns"]
            if "SQ_INSTS_VALU" in c:
                e["valu_issue_util"] = c["SQ_INSTS_VALU"] * 4.0 / (N_SIMD * c["GRBM_GUI_ACTIVE"] / N_XCD)
        if c.get("SQ_WAVE_CYCLES"):
            for w in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS"):
                if w in c:
                    e[w + "_share"] = c[w] / c["SQ_WAVE_CYCLES"]
        if "valu_frac" in e and "hbm_frac" in e:
            e["bound"] = "valu" if e["valu_frac"] >= e["hbm_frac"] else "hbm"
        kernels[k] = e
    # Per CALL of each timed operation of the case (tools/bench_configs.py in profile mode makes 1 oracle-check call + 8 counted
    # calls of every operation; a rotation runs the key-switch kernels too): kernel time, HBM-side bytes and VALU instructions
    # summed over the kernels the operation launches -- all dispatches of the counter passes, warm-up included (the counts do not
    # depend on the clock).  `per_call[op]["dominant"]` = the kernel with the largest share of the operation's kernel time.
    def op_of(k):
        if k.startswith("k_rescale"):
            return "rescale"
        if k.startswith("k_galois"):
            return "galois"
        ntt_case = "limbs" in (rec or {}) and "level" not in (rec or {})        # the stand-alone transform cases
        if ntt_case and k.startswith("k_ntt_fwd"):
            return "nntt"
        if ntt_case and k.startswith("k_ntt_inv"):
            return "inntt"
        return "main"
    # main: key switch (9 calls) + rotation (9 calls) in the key-switch cases; the encrypted-MNIST cases: the evaluation passes inside the
    # window of in_window() (the passes after the first; set-up excluded since r06)
    calls = {"rescale": 9, "galois": 9, "nntt": 9, "inntt": 9, "main": 18 if (rec or {}).get("keyswitch_per_s") else passes[0]}
    per_call = {}
    for k, e in kernels.items():
        op = op_of(k)
        tot = {name: sum(v for _, v, _ in rows) for name, rows in cnt.get(k, {}).items()}
        n_disp = max((len(rows) for rows in cnt.get(k, {}).values()), default=0)
        pc = per_call.setdefault(op, {"calls_in_counter_pass": calls[op], "kernel_us": 0.0, "hbm_bytes": 0.0, "valu_insts": 0.0, "kernels": {}})
        us = e["mean_us"] * n_disp / calls[op]
        pc["kernel_us"] += us
        pc["hbm_bytes"] += (tot.get("FETCH_SIZE", 0.0) * 1024.0 * FETCH_CAL + tot.get("WRITE_SIZE", 0.0) * 1024.0 * WRITE_CAL) / calls[op]
        pc["valu_insts"] += tot.get("SQ_INSTS_VALU", 0.0) / calls[op]
        pc["kernels"][k] = us
    for op, pc in per_call.items():
        dom = max(pc["kernels"], key=pc["kernels"].get)
        pc["dominant"] = dom
        pc["dominant_share"] = pc["kernels"][dom] / pc["kernel_us"] if pc["kernel_us"] else None
        pc["kernels"] = {k: round(v, 1) for k, v in pc["kernels"].items()}
    out["cases"][case] = {"config": (rec or {}).get("config"), "record": rec, "kernels": kernels, "per_call": per_call}

if not table:
    print(json.dumps(out, indent=1))
else:
    for case, cv in out["cases"].items():
        print(f"## case {case}: {cv['config']}")
        if "mnist" in str(cv["config"]).lower():
            # (VERDICT r05 read the many small u64 launches of these tables as the pass's; they are the set-up's)
            print("   (evaluation passes after the first only: the dispatches between the first and the last k_ckks_decode_start of each profile; the set-up -- key "
                  "generation, encryption, weight encoding -- is outside the window.  One pass by itself: profiles/r06_mnist16_*_last_pass.txt)")
        print(f"{'kernel':52s} {'n':>5s} {'mean_us':>9s} {'share':>6s} {'valu_frac':>9s} {'util@clk':>8s} {'GHz':>5s} {'hbm_GB':>8s} {'hbm_frac':>8s} "
              f"{'valu/wave':>9s} {'wait':>5s} {'stall':>5s} {'issue':>5s} {'bound':>5s}")
        for k, e in cv["kernels"].items():
            g = lambda key, fmt, d="": (fmt % e[key]) if key in e else d
            print(f"{k[:52]:52s} {e['launches']:5d} {e['mean_us']:9.1f} {e['share_of_case']:6.3f} {g('valu_frac', '%9.3f'):>9s} "
                  f"{g('valu_issue_util', '%8.3f'):>8s} {g('clock_GHz', '%5.2f'):>5s} {(('%8.3f' % (e['hbm_bytes'] / 1e9)) if 'hbm_bytes' in e else ''):>8s} "
                  f"{g('hbm_frac', '%8.3f'):>8s} {g('valu_per_wave', '%9.0f'):>9s} {g('SQ_WAIT_ANY_share', '%5.2f'):>5s} "
                  f"{g('SQ_WAIT_INST_ANY_share', '%5.2f'):>5s} {g('SQ_ACTIVE_INST_ANY_share', '%5.2f'):>5s} {e.get('bound', ''):>5s}"
                  + ("  (bytes uncalibrated: gather / scatter)" if "byte_calibration" in e else ""))
        rec_, pm = cv.get("record") or {}, (cv.get("per_call") or {}).get("main")
        if pm and "level" in rec_ and str(pm.get("dominant", "")).startswith("k_ks_fused"):
            # what the traffic ratio of a fused key switch is made of: key rows are shared by all items of a call and
            # come from the L2s / the Infinity Cache (FETCH_SIZE counts the hits); the rest streams (what the x2 calibration was made on)
            lv, n_, b_ = rec_["level"], rec_["N"], rec_.get("batch") or 1
            alg, keyb, tot = 4 * lv * n_ * 8, lv * 2 * (lv + 1) * n_ * 8, pm["hbm_bytes"] / b_
            print(f"   per key switch (key switch + rotation calls averaged): HBM-side {tot / 1e6:.1f} MB = {tot / alg:.1f} x the algorithmic {alg / 1e6:.1f} MB: "
                  f"key rows {keyb / 1e6:.1f} MB ({keyb / alg:.1f} x; cache-served, counted) + rows {max(0.0, tot - keyb) / 1e6:.1f} MB "
                  f"({max(0.0, tot - keyb) / alg:.1f} x; streaming, calibrated)")
        print()
