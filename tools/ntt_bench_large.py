#!/usr/bin/env python3
"""NTT throughput for N > 2^14 (top stages on global memory + 2^14-point LDS blocks): fp64 block kernels (variant 0)
against the u64 ones (variant 2).  usage: ntt_bench_large.py [logN] [rows]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import toyfhe_jl_amd as tf
from tests import helpers as H

logn = int(sys.argv[1]) if len(sys.argv) > 1 else 15
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
N, L = 1 << logn, 4
ctx = tf.Context(N, H.chain(50, L, N))
a, b = tf.DeviceBuffer(rows * N), tf.DeviceBuffer(rows * N)
tf.native.check(tf.native.lib().tfhe_memset(ctx.h, a.ptr, 1, rows * N * 8))
count, reps = rows // L, 10
gb = rows * N * 8 * 2 / 1e9
def timed(f):
    for _ in range(reps): f()
    ctx.sync(); t = time.perf_counter()
    for _ in range(reps): f()
    ctx.sync(); return (time.perf_counter() - t) / reps
for rnd in range(2):
    for v in (0, 2):
        ctx.set_ntt_variant(v)
        tfw = timed(lambda: ctx.nntt(a.ptr, b.ptr, count, L)); tiv = timed(lambda: ctx.inntt(b.ptr, a.ptr, count, L))
        print("N=2^%d variant %d  fwd %6.0f GB/s  inv %6.0f GB/s" % (logn, v, gb / tfw, gb / tiv))
