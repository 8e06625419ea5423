#!/bin/bash
# usage (GPU box): prof_mnist_last_pass.sh <out-tag>: kernel traces of the LAST evaluation pass of the encrypted-MNIST example at N = 2^16 in
# both circuit shapes -> gpurun_out/<tag>/r06_mnist16_{refshape,restructured}_last_pass.txt (tools/csv_last_pass.py); the raw traces are removed
OUTTAG=$1
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$OUTTAG; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace -d /tmp/prof_ref -o t --output-format csv -- python $R/tools/prof_mnist_ref.py > $O/prof_ref.log 2>&1
python $R/tools/csv_last_pass.py /tmp/prof_ref > $O/r06_mnist16_refshape_last_pass.txt
rocprofv3 --kernel-trace -d /tmp/prof_eval -o t --output-format csv -- python $R/tools/prof_mnist_eval.py 16 16 1 > $O/prof_eval.log 2>&1
python $R/tools/csv_last_pass.py /tmp/prof_eval > $O/r06_mnist16_restructured_last_pass.txt
head -32 $O/r06_mnist16_refshape_last_pass.txt
