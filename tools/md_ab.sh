# usage: md_ab.sh <tag for profile> <lib tags...>
R=$PWD; P=$1; shift
export TFHE_HIP_LIB=$R/tools/ab_$1.so
timeout 900 python -m pytest tests/test_gpu_reference_mirrors.py -x -q -m gpu -k "matmul_diag" 2>&1 | tail -2
for rep in 1 2; do for l in "$@"; do echo -n "$l: "; TFHE_HIP_LIB=$R/tools/ab_$l.so python examples/encrypted_mnist.py --logn 16 --batches 4 --hoisted --fused --repeat 3 2>&1 | tail -3 | head -1 | cut -c38-100; done; done
cd /tmp; export TMPDIR=/tmp
for l in "$@"; do TFHE_HIP_LIB=$R/tools/ab_$l.so rocprofv3 --kernel-trace -d $R/gpurun_out/prof_$P$l -o t --output-format csv -- python $R/tools/prof_mnist_eval.py 16 4 1 > /dev/null 2>&1; echo "== $l"; TFHE_HIP_LIB=$R/tools/ab_$l.so python $R/tools/csv_last_pass.py $R/gpurun_out/prof_$P$l | head -8; done
