#!/bin/bash
# usage: ab_ablate.sh <out-tag>: the upper-bound (ablation) builds of round 6 next to the product build, alternating, on one box.
#  headline (bench.py, no other legs): product | nobar (TFHE_ABL_NOXCHG=1) | noxchg (=3) | norows | nokeys
#  cfg#3 (bench_configs case 1, oracle check off for the wrong-result builds): product | toponce | norows
OUTTAG=$1
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$OUTTAG; mkdir -p $O
for i in 1 2; do
  for w in base nobar nowr nowrbar nord noxchg norows nokeys; do
    L=$R/tools/ab_$w.so; [ $w = base ] && L=$R/toyfhe.jl_amd/libtoyfhe_hip.so
    [ -f $L ] || continue
    TFHE_HIP_LIB=$L timeout 300 python $R/bench.py --steps 10 --warmup 3 --no-cpu --no-ntt --no-configs 2>/dev/null | tail -1 > $O/head_${w}_$i.json
    python -c "import json,sys; d=json.load(open('$O/head_${w}_$i.json')); print('headline', '$w', $i, round(d['value']), 'ct-mul/s', round(d['ms_per_step'],3), 'ms/step')"
  done
  for w in base toponce norows; do
    L=$R/tools/ab_$w.so; [ $w = base ] && L=$R/toyfhe.jl_amd/libtoyfhe_hip.so
    [ -f $L ] || continue
    TFHE_CFG_NOCHECK=1 TFHE_HIP_LIB=$L timeout 300 python $R/tools/bench_configs.py 1 1 2>/dev/null | tail -1 > $O/cfg3_${w}_$i.json
    python -c "import json,sys; d=json.load(open('$O/cfg3_${w}_$i.json')); print('cfg#3', '$w', $i, round(d.get('keyswitch_per_s',0)), 'ks/s', round(d.get('rotate_per_s',0)), 'rot/s', d.get('error',''))"
  done
done | tee $O/ab.log
