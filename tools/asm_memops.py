#!/usr/bin/env python3
"""Memory-operation skeleton of one kernel in a gfx950 .s file (design aid): every vector-memory / LDS / scratch instruction,
barrier and s_waitcnt in program order with the number of VALU instructions between them -- how far ahead of its wait a load
is issued, and where a load is awaited on its own.  (This is how the serialised addend loads of the fused key switch and the
just-in-time constant loads of the narrow BFV conversions were found.)
usage: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -S --cuda-device-only -o full.s toyfhe.jl_amd/csrc/toyfhe_hip.hip
       asm_memops.py full.s <substring of the mangled kernel name> [vmcnt0]   (vmcnt0: only the full waits, with what precedes them)"""
import re, sys
path, key = sys.argv[1], sys.argv[2]
only0 = len(sys.argv) > 3 and sys.argv[3] == "vmcnt0"
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and key in l and ":" in l)
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
out, nv = [], 0
for i in range(start, end + 1):
    l = lines[i].strip()
    m = re.match(r"^([a-z][a-z_0-9]+)", l)
    if not m:
        continue
    op = m.group(1)
    if op.startswith("v_"):
        nv += 1
        continue
    if op.startswith(("global_", "scratch_", "flat_", "buffer_", "ds_")) or op in ("s_barrier", "s_waitcnt"):
        k = l if op == "s_waitcnt" else op
        if out and out[-1][1] == k and nv == 0:
            out[-1][2] += 1
        else:
            if nv:
                out.append([i - start, "VALU", nv])
            out.append([i - start, k, 1])
        nv = 0
prev = None
for o in out:
    if only0:
        if o[1].startswith("s_waitcnt") and "vmcnt(0)" in o[1] and prev:
            print(f"{prev[0]:6d} {prev[1]} x{prev[2]}  ||  {o[0]:6d} {o[1]}")
    else:
        print(f"{o[0]:6d} {o[1]} x{o[2]}")
    prev = o
