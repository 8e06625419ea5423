#!/usr/bin/env python3
"""NTT throughput at a given N <= 2^14 (same total bytes), to compare workgroup-per-CU regimes.  usage: ntt_bench_n.py logN"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import toyfhe_jl_amd as tf
from tests import helpers as H
logn = int(sys.argv[1]); N, L = 1 << logn, 8
rows = (8192 << 14) >> logn
ctx = tf.Context(N, H.chain(50, L, N))
a, b = tf.DeviceBuffer(rows * N), tf.DeviceBuffer(rows * N)
tf.native.check(tf.native.lib().tfhe_memset(ctx.h, a.ptr, 1, rows * N * 8))
count, reps, gb = rows // L, 20, rows * N * 16 / 1e9
def timed(f):
    for _ in range(reps): f()
    ctx.sync(); t = time.perf_counter()
    for _ in range(reps): f()
    ctx.sync(); return (time.perf_counter() - t) / reps
for rnd in range(2):
    tfw = timed(lambda: ctx.nntt(a.ptr, b.ptr, count, L)); tiv = timed(lambda: ctx.inntt(b.ptr, a.ptr, count, L))
    print("N=2^%d rows %d  fwd %6.0f GB/s  inv %6.0f GB/s" % (logn, rows, gb / tfw, gb / tiv))
