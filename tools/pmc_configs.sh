#!/bin/bash
# Counters for the kernels of the OTHER BASELINE.json configurations (cfg#3 / #4 / #5, the N = 2^16 and 60-bit transforms, the
# encrypted-MNIST pass), run ON THE GPU BOX:   gpurun -- 'bash tools/pmc_configs.sh r04 "1 2 3"'
# Per case of tools/bench_configs.py (positions 1..10):
#   1. rocprofv3 --kernel-trace --stats over the case as bench.py runs it (durations; steady state: warm-up + 3 x 8 calls)
#   2. separate --pmc passes over the same case in profile mode (TFHE_CFG_PROFILE=1: 3 warm calls, then 5 counted ones; the
#      folding script drops the first quarter of every kernel's dispatches): FETCH_SIZE | WRITE_SIZE | 8 SQ counters |
#      GRBM_GUI_ACTIVE + 3 SQ counters.  Never combined with --sys-trace & co (MI355X_MICROARCH.md, rocprofv3 PMC slots).
#   3. tools/pmc_configs.py folds everything into <tag>_pmc_configs.json (copy it to profiles/).
set -u
TAG=${1:-r04}
CASES=${2:-"1 2 3 4 5 6 7 8 9 10"}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/pmcc_$TAG
mkdir -p "$OUT" && cd /tmp && export TMPDIR=/tmp
for c in $CASES; do
  D=$OUT/case$c; mkdir -p "$D"
  rocprofv3 --kernel-trace --stats -d "$D/trace" -o t --output-format csv -- python "$R/tools/bench_configs.py" 1 "$c" > "$D/trace.log" 2> "$D/trace.err"
  pass() { name=$1; shift; TFHE_CFG_PROFILE=1 rocprofv3 --kernel-trace --pmc "$@" -d "$D/pmc_$name" -o p --output-format csv -- python "$R/tools/bench_configs.py" 1 "$c" > "$D/pmc_$name.log" 2>&1; }
  pass FETCH_SIZE FETCH_SIZE
  pass WRITE_SIZE WRITE_SIZE
  pass SQ SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS
  pass GRBM GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES
done
python "$R/tools/pmc_configs.py" "$OUT" "$TAG" > "$OUT/${TAG}_pmc_configs.json"
python "$R/tools/pmc_configs.py" "$OUT" "$TAG" --table > "$OUT/${TAG}_pmc_configs.txt"
cat "$OUT/${TAG}_pmc_configs.txt"
