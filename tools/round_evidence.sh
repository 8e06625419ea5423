#!/bin/bash
# One round's evidence against ONE build, run ON THE GPU BOX (gpurun -- 'bash tools/round_evidence.sh r06e'):
#   GPU tests -> counters of the headline (pmc_round.sh) and of every configuration (pmc_configs.sh), installed as profiles/pmc_*.json on the
#   box so that the bench line that follows reads counters of ITS OWN source id -> the full default bench line -> last-pass traces of the MNIST example.
# Everything lands under gpurun_out/<tag>_evidence/; copy what is to be judged into profiles/ afterwards.
set -u
TAG=${1:-r06e}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
E=$R/gpurun_out/${TAG}_evidence; mkdir -p $E
cd $R
python -m pytest tests -m gpu -q 2>&1 | grep -vE "^(RCCL|HIP|ROCm|Hostname|Librccl)" | tail -15 > $E/${TAG}_gpu_tests.txt
bash tools/pmc_round.sh $TAG > $E/pmc_round.log 2>&1
cp gpurun_out/pmc_$TAG/${TAG}_pmc_bench_kernels.json profiles/pmc_bench_kernels.json
cp gpurun_out/pmc_$TAG/${TAG}_pmc_bench_kernels.json gpurun_out/pmc_$TAG/${TAG}_bench_kernel_stats.txt $E/
bash tools/pmc_configs.sh $TAG > $E/pmc_configs.log 2>&1
cp gpurun_out/pmcc_$TAG/${TAG}_pmc_configs.json profiles/pmc_configs.json
cp gpurun_out/pmcc_$TAG/${TAG}_pmc_configs.json gpurun_out/pmcc_$TAG/${TAG}_pmc_configs.txt $E/
cd $R && python bench.py 2> $E/bench.err | tail -1 > $E/${TAG}_bench.json
cp profiles/bench_full.json $E/${TAG}_bench_full.json
bash tools/prof_mnist_last_pass.sh ${TAG}_evidence > /dev/null 2>&1
# keep the merge small: the raw traces stay on the box
rm -rf gpurun_out/pmc_$TAG/trace gpurun_out/pmc_$TAG/pmc_* gpurun_out/pmcc_$TAG/case*/trace gpurun_out/pmcc_$TAG/case*/pmc_*
tail -3 $E/${TAG}_gpu_tests.txt; cat $E/${TAG}_bench.json | cut -c1-600
