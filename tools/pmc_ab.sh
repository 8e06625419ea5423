#!/bin/bash
# SQ / GRBM counter passes of one bench step (256 ciphertexts) for each tools/ab_<tag>.so; per-kernel table to stdout.
# usage (GPU box): bash tools/pmc_ab.sh <out-tag> <tag> [<tag> ...]
OUTTAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp; export TMPDIR=/tmp
for w in "$@"; do
  OUT=$R/gpurun_out/$OUTTAG/pmc_$w; mkdir -p $OUT
  CMD="python $R/bench.py --steps 1 --warmup 1 --batch 256 --no-cpu --no-ntt --no-configs"
  TFHE_HIP_LIB=$R/tools/ab_$w.so rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU -d $OUT/pmc_SQ -o p --output-format csv -- $CMD > $OUT/sq.log 2>&1
  TFHE_HIP_LIB=$R/tools/ab_$w.so rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_INSTS_SMEM SQ_WAVES SQ_INSTS_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM -d $OUT/pmc_GRBM -o p --output-format csv -- $CMD > $OUT/grbm.log 2>&1
  python $R/tools/pmc_bench.py $OUT 256 $w > $R/gpurun_out/$OUTTAG/${w}_pmc.json
  python - $R/gpurun_out/$OUTTAG/${w}_pmc.json $w <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
for k,v in d['kernels'].items():
    if 'k_' not in k: continue
    w=v.get('SQ_WAVES',0) or 1; n=v['launches']
    print(sys.argv[2], k[:40].ljust(40), 'launches',n, 'valu/wave %.0f'%(v.get('SQ_INSTS_VALU',0)/w), 'salu/wave %.0f'%(v.get('SQ_INSTS_SALU',0)/w), 'smem/wave %.0f'%(v.get('SQ_INSTS_SMEM',0)/w),
          'dur_us %.0f'%(v.get('duration_ns',0)/n/1e3), 'clk %.2f'%v.get('clock_GHz',0), 'valu_util %.2f'%v.get('valu_issue_util',0),
          'wait_any %.2f'%v.get('SQ_WAIT_ANY_share_of_wave_cycles',0), 'wait_inst %.2f'%v.get('SQ_WAIT_INST_ANY_share_of_wave_cycles',0), 'active_valu %.2f'%v.get('SQ_ACTIVE_INST_VALU_share_of_wave_cycles',0))
PY
done
