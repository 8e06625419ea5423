#!/bin/bash
# usage: ab_lanes.sh <out-tag> [cases]: alternating runs of the mixed-ring cases of bench_configs.py with TFHE_LANES=0 (one stream) and 1 (two lanes)
OUTTAG=$1; CASES=${2:-3,5,9,10}
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/$OUTTAG
for i in 1 2; do
  for l in 0 1; do
    TFHE_LANES=$l python $R/tools/bench_configs.py 1 $CASES 2>/dev/null > $R/gpurun_out/$OUTTAG/lanes${l}_$i.jsonl
    python - $l $R/gpurun_out/$OUTTAG/lanes${l}_$i.jsonl <<'PY'
import sys, json
for ln in open(sys.argv[2]):
    try: d = json.loads(ln)
    except Exception: continue
    keys = [k for k in ("keyswitch_per_s", "rotate_per_s", "fwd_GBs", "inv_GBs", "images_per_s", "ms_per_pass", "error") if k in d]
    print("lanes", sys.argv[1], d.get("name", d.get("config", ""))[:60], {k: (round(d[k], 1) if isinstance(d[k], float) else d[k]) for k in keys})
PY
  done
done | tee $R/gpurun_out/$OUTTAG/ab.log
