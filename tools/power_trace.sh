#!/bin/bash
# Power / clock samples of the GPU while a command runs (run ON THE GPU BOX): bash tools/power_trace.sh <tag> <command...>
# writes gpurun_out/power_<tag>.txt (rocm-smi samples every ~0.25 s) and prints min / median / max of each column.
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/power_$TAG.txt
mkdir -p "$R/gpurun_out"; : > "$OUT"
"$@" > "$R/gpurun_out/power_$TAG.cmd.log" 2>&1 &
PID=$!
while kill -0 $PID 2>/dev/null; do
  rocm-smi --showpower --showclocks --showuse --csv 2>/dev/null | tail -n +2 | head -1 >> "$OUT"
  sleep 0.25
done
wait $PID
rocm-smi --showpower --showclocks --showuse --csv 2>/dev/null | head -1
python3 - "$OUT" <<'PY'
import sys, re
rows = [l.strip().split(',') for l in open(sys.argv[1]) if l.strip()]
print(len(rows), "samples")
for c in range(len(rows[0])):
    vals = []
    for r in rows:
        m = re.search(r'[-+]?\d+\.?\d*', r[c]) if c < len(r) else None
        if m: vals.append(float(m.group()))
    if vals:
        vals.sort(); print("col %d: min %.1f median %.1f max %.1f" % (c, vals[0], vals[len(vals)//2], vals[-1]))
PY
tail -1 "$R/gpurun_out/power_$TAG.cmd.log" | cut -c1-160
