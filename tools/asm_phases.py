#!/usr/bin/env python3
"""Per-phase instruction budget of one kernel in a gfx950 .s listing (design aid; r05).

The kernel is cut at its barriers, labels and branches; every segment gets the count of its instructions by class:
  fp64   v_fma_f64 v_mul_f64 v_add_f64 (the butterflies, products and sweeps)      rnd    v_rndne_f64 (one per modular product / sweep)
  cvt    v_cvt_* / v_ldexp / v_frexp (u64 <-> double conversions)                   vint   other VALU (addresses, selects, 64-bit integer work)
  mov    v_mov / v_accvgpr (moves, spill traffic through AGPRs)                       salu   scalar ALU + s_nop
  lds    ds_*      vmem  global_* / buffer_* / scratch_*      wait  s_waitcnt        bar    s_barrier
usage: asm_phases.py full.s <substring of the mangled kernel name> [lo:hi:trips ...]
  lo:hi:trips  -- lines lo..hi (relative to the kernel's first line, as printed) form a loop body executed `trips` times; the
                  weighted totals at the end use it (default: every segment once)."""
import collections
import re
import sys

CLASSES = ("fp64", "rnd", "cvt", "vint", "mov", "salu", "lds", "vmem", "wait", "bar")


def classify(op):
    if op.startswith(("v_fma_f64", "v_mul_f64", "v_add_f64", "v_fmac_f64")):
        return "fp64"
    if op.startswith("v_rndne_f64"):
        return "rnd"
    if op.startswith(("v_cvt_", "v_ldexp", "v_frexp", "v_trunc_f64", "v_floor_f64", "v_fract")):
        return "cvt"
    if op.startswith(("v_mov", "v_accvgpr", "v_swap")):
        return "mov"
    if op.startswith("v_"):
        return "vint"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "scratch_", "flat_")):
        return "vmem"
    if op == "s_waitcnt":
        return "wait"
    if op == "s_barrier":
        return "bar"
    if op.startswith("s_"):
        return "salu"
    return None


def main():
    path, key = sys.argv[1], sys.argv[2]
    loops = [tuple(int(x) for x in a.split(":")) for a in sys.argv[3:]]
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and key in l and ":" in l)
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
    print(lines[start].split(":")[0])
    segs, cur, cur_start = [], collections.Counter(), 0
    ops_all = collections.Counter()

    def close(at, why):
        nonlocal cur, cur_start
        if sum(cur.values()):
            segs.append((cur_start, at, why, cur))
        cur, cur_start = collections.Counter(), at

    for i in range(start, end + 1):
        l = lines[i]
        rel = i - start
        if l.startswith(".LBB"):
            close(rel, "label " + l.split(":")[0])
            continue
        m = re.match(r"^\s+([a-z][a-z_0-9]+)", l)
        if not m:
            continue
        op = m.group(1)
        c = classify(op)
        if c is None:
            continue
        cur[c] += 1
        ops_all[op] += 1
        if op == "s_barrier":
            close(rel + 1, "barrier")
        elif op.startswith(("s_cbranch", "s_branch")):
            close(rel + 1, op + " " + l.split()[-1])
    close(end - start, "end")
    hdr = f"{'lines':>13s} " + " ".join(f"{c:>5s}" for c in CLASSES) + "  total  ends with"
    print(hdr)
    tot_w = collections.Counter()
    for a, b, why, cnt in segs:
        w = 1
        for lo, hi, trips in loops:
            if a >= lo and b <= hi + 1:
                w = trips
        for c in CLASSES:
            tot_w[c] += cnt[c] * w
        n = sum(cnt.values())
        if n >= 8:
            print(f"{a:6d}-{b:6d} " + " ".join(f"{cnt[c]:5d}" for c in CLASSES) + f" {n:6d}  {why}" + (f"   x{w}" if w != 1 else ""))
    n = sum(tot_w.values())
    print("weighted      " + " ".join(f"{tot_w[c]:5d}" for c in CLASSES) + f" {n:6d}")
    valu = sum(tot_w[c] for c in ("fp64", "rnd", "cvt", "vint", "mov"))
    print(f"VALU {valu} ({valu / n:.3f} of all)   fp64+rnd {tot_w['fp64'] + tot_w['rnd']} ({(tot_w['fp64'] + tot_w['rnd']) / valu:.3f} of VALU)")
    print("top opcodes: " + ", ".join(f"{k} {v}" for k, v in ops_all.most_common(14)))


main()
