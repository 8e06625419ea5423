#!/bin/bash
# usage: ab_full.sh <out-tag> <tag> <tag> ... : alternating full bench.py lines (headline + other_configs) per tools/ab_<tag>.so
OUTTAG=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/$OUTTAG
for i in 1 2; do
  for w in "$@"; do
    TFHE_HIP_LIB=$R/tools/ab_$w.so python $R/bench.py --steps 10 --warmup 3 --no-cpu 2>/dev/null | tail -1 > $R/gpurun_out/$OUTTAG/${w}_$i.json
    python - "$w" $R/gpurun_out/$OUTTAG/${w}_$i.json <<'PY'
import sys, json
d = json.load(open(sys.argv[2]))
oc = {c["name"][:6]: c for c in d.get("other_configs", [])}
f = lambda c, k: round(c.get(k, 0)) if c else None
print(sys.argv[1], "ctmul", round(d["value"]), "ntt", round(d["ntt"]["fwd_GBs"]), round(d["ntt"]["inv_GBs"]),
      "| cfg3", f(oc.get("cfg#3 "), "ks_s"), f(oc.get("cfg#3 "), "rot_s"), "| cfg4", f(oc.get("cfg#4 "), "ks_s"), f(oc.get("cfg#4 "), "rot_s"),
      "| cfg5", f(oc.get("cfg#5 "), "ks_s"), f(oc.get("cfg#5'"), "ks_s"), "| mnist", [round(c["img_s"]) for c in d.get("other_configs", []) if "img_s" in c],
      "| ntt16", [(round(c["fwd_GBs"]), round(c["inv_GBs"])) for c in d.get("other_configs", []) if "fwd_GBs" in c])
PY
  done
done | tee $R/gpurun_out/$OUTTAG/ab.log
