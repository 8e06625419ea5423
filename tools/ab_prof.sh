#!/bin/bash
# Same-box comparison of several library builds: per-kernel rocprofv3 statistics of the headline bench for each
# tools/ab_<tag>.so (TFHE_HIP_LIB selects the build), then alternating un-profiled bench runs.
# usage (on the GPU box, repo root): bash tools/ab_prof.sh <out-tag> <tag> [<tag> ...]
OUTTAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p $R/gpurun_out/$OUTTAG; cd /tmp; export TMPDIR=/tmp
for w in "$@"; do
  OUT=$R/gpurun_out/$OUTTAG/prof_$w; mkdir -p $OUT
  TFHE_HIP_LIB=$R/tools/ab_$w.so rocprofv3 --kernel-trace --stats -d $OUT -o t --output-format csv -- python $R/bench.py --steps 8 --warmup 3 --no-cpu --no-ntt --no-configs > $OUT/cmd.log 2>&1
  python $R/tools/csv_kernel_stats.py $OUT > $R/gpurun_out/$OUTTAG/${w}_kernel_stats.txt
  echo "== $w"; head -8 $R/gpurun_out/$OUTTAG/${w}_kernel_stats.txt | cut -c1-60,68-130
done
for i in 1 2; do
  for w in "$@"; do
    TFHE_HIP_LIB=$R/tools/ab_$w.so python $R/bench.py --steps 8 --warmup 3 --no-cpu --no-ntt --no-configs 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$w', round(d['value']), round(d['ms_per_step'],3))"
  done
done | tee $R/gpurun_out/$OUTTAG/ab.log
