#!/usr/bin/env python3
"""NTT throughput on a ring that mixes modulus sizes (60-bit q0 + five 40-bit primes + 60-bit special prime, infer.jl:97-112):
variant 0 = two launches, fp64 kernels for the 40-bit limbs and u64 kernels for the 60-bit ones; variant 2 = u64 everywhere.
usage: ntt_bench_mixed.py [logN] [polys]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import toyfhe_jl_amd as tf
from tests import helpers as H
logn = int(sys.argv[1]) if len(sys.argv) > 1 else 14
count = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
N = 1 << logn
qs = H.chain(60, 1, N) + H.chain(40, 5, N) + H.chain(61, 1, N)
L = len(qs)
ctx = tf.Context(N, qs)
a, b = tf.DeviceBuffer(count * L * N), tf.DeviceBuffer(count * L * N)
tf.native.check(tf.native.lib().tfhe_memset(ctx.h, a.ptr, 1, count * L * N * 8))
gb = count * L * N * 16 / 1e9
def timed(f, reps=10):
    for _ in range(reps): f()
    ctx.sync(); t = time.perf_counter()
    for _ in range(reps): f()
    ctx.sync(); return (time.perf_counter() - t) / reps
for rnd in range(2):
    for v in (0, 2):
        ctx.set_ntt_variant(v)
        tfw = timed(lambda: ctx.nntt(a.ptr, b.ptr, count, L)); tiv = timed(lambda: ctx.inntt(b.ptr, a.ptr, count, L))
        print("N=2^%d mixed ring, variant %d: fwd %6.0f GB/s  inv %6.0f GB/s" % (logn, v, gb / tfw, gb / tiv))
