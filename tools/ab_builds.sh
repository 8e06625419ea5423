#!/bin/bash
# Same-box A/B of two library builds (tools/ab_old.so, tools/ab_new.so) on the headline bench: alternating runs.
# usage (on the GPU box, repo root): bash tools/ab_builds.sh [rounds]
R=${1:-3}
SO=toyfhe.jl_amd/libtoyfhe_hip.so
cp $SO /tmp/keep.so
for i in $(seq $R); do
  for w in old new; do
    cp tools/ab_$w.so $SO
    python bench.py --steps 6 --warmup 3 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$w', round(d['value']), d['ms_per_step'], d['roofline']['achieved'])"
  done
done
cp /tmp/keep.so $SO
