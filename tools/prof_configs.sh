#!/bin/bash
# Per-kernel rocprofv3 statistics of the other BASELINE.json configurations, one case per run (tools/bench_configs.py).
# usage (on the GPU box, repo root): bash tools/prof_configs.sh <out-tag>
TAG=$1
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p $R/gpurun_out/$TAG; cd /tmp; export TMPDIR=/tmp
for i in 1 2 3 4 5 6 7 8; do
  sel=$i; OUT=$R/gpurun_out/$TAG/prof_$i; mkdir -p $OUT
  rocprofv3 --kernel-trace --stats -d $OUT -o t --output-format csv -- python $R/tools/bench_configs.py 1 "$sel" > $OUT/cmd.log 2>&1
  { echo "## case $sel"; grep -h "^{" $OUT/cmd.log | cut -c1-300 | head -2; python $R/tools/csv_kernel_stats.py $OUT | head -16 | cut -c1-150; echo; } >> $R/gpurun_out/$TAG/configs_kernel_stats.txt
done
cat $R/gpurun_out/$TAG/configs_kernel_stats.txt
