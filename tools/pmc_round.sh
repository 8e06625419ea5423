#!/bin/bash
# One round's profile evidence for bench.py, run ON THE GPU BOX (gpurun -- 'bash tools/pmc_round.sh r02a'):
#   1. rocprofv3 --kernel-trace --stats over the default bench (kernel durations; summary -> profiles/<tag>_bench_kernel_stats.txt)
#   2. separate --pmc passes over `bench.py --steps 1 --warmup 3 --batch 256` (steady state: warm-up dispatches dropped) (one launch of each fused kernel = 256
#      ciphertexts): FETCH_SIZE, WRITE_SIZE (TCC: cannot share a pass), the SQ issue/wait counters (8 SQ slots), and
#      GRBM_GUI_ACTIVE (+ SQ_INSTS_LDS); never combined with --sys-trace & co (MI355X_MICROARCH.md, rocprofv3 PMC slots).
#   3. tools/pmc_bench.py folds the passes into profiles/pmc_bench_kernels.json (what bench.py reads) and a tagged copy.
set -u
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p "$OUT" && cd /tmp && export TMPDIR=/tmp
BENCH1="python $R/bench.py --steps 1 --warmup 3 --batch 256 --no-cpu --no-ntt --no-configs"   # 3 warm steps, the folding drops the first
rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o t --output-format csv -- python "$R/bench.py" --no-cpu --no-ntt --no-configs > "$OUT/bench_under_trace.json" 2> "$OUT/trace.err"
python "$R/tools/csv_kernel_stats.py" "$OUT/trace" > "$OUT/${TAG}_bench_kernel_stats.txt" 2>> "$OUT/trace.err"
pass() { name=$1; shift; rocprofv3 --kernel-trace --pmc "$@" -d "$OUT/pmc_$name" -o p --output-format csv -- $BENCH1 > "$OUT/pmc_$name.log" 2>&1; }
pass FETCH_SIZE FETCH_SIZE
pass WRITE_SIZE WRITE_SIZE
pass SQ SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS
pass GRBM GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES
python "$R/tools/pmc_bench.py" "$OUT" 256 "$TAG" > "$OUT/${TAG}_pmc_bench_kernels.json"
ls -la "$OUT"
