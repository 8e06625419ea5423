#!/bin/bash
# rocprofv3 --kernel-trace --stats over a command, per-kernel summary to stdout (run ON THE GPU BOX):
#   bash tools/rocprof_stats.sh <tag> <command...>
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p "$OUT"; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d "$OUT" -o t --output-format csv -- "$@" > "$OUT/cmd.log" 2>&1
python "$R/tools/csv_kernel_stats.py" "$OUT" | tee "$OUT/${TAG}_kernel_stats.txt"
