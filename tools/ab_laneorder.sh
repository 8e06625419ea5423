#!/bin/bash
# usage: ab_laneorder.sh <out-tag>: which policy's launches go first in the two-lane transforms of a mixed ring (TFHE_LANE_FP_FIRST)
OUTTAG=$1
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$OUTTAG; mkdir -p $O
for i in 1 2; do
  for f in 0 1; do
    TFHE_LANE_FP_FIRST=$f python $R/tools/bench_configs.py 1 3,5,10 2>/dev/null > $O/fpfirst${f}_$i.jsonl
    python - $f $O/fpfirst${f}_$i.jsonl <<'PY'
import sys, json
for ln in open(sys.argv[2]):
    try: d = json.loads(ln)
    except Exception: continue
    keys = [k for k in ("keyswitch_per_s", "rotate_per_s", "fwd_GBs", "inv_GBs", "images_per_s", "ms_per_pass", "error") if k in d]
    print("fp_first", sys.argv[1], d.get("config", "")[:70], {k: (round(d[k], 1) if isinstance(d[k], float) else d[k]) for k in keys})
PY
  done
done | tee $O/ab.log
