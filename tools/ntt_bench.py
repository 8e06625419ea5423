#!/usr/bin/env python3
"""Forward / inverse 2^14-point NTT throughput (GB/s of algorithmic traffic: one read + one write per row),
staged kernel (variant 0) vs unstaged fp64 kernel (variant 3), with a plain row copy for scale."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import toyfhe_jl_amd as tf
from tests import helpers as H

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
N, L = 1 << 14, int(os.environ.get("NTT_L", "8"))
ctx = tf.Context(N, H.chain(50, L, N))
a, b = tf.DeviceBuffer(rows * N), tf.DeviceBuffer(rows * N)
tf.native.check(tf.native.lib().tfhe_memset(ctx.h, a.ptr, 1, rows * N * 8))
count = rows // L
gb = rows * N * 8 * 2 / 1e9

def timed(f):
    for _ in range(reps): f()          # clocks ramp over milliseconds: warm up with as much work as is measured
    ctx.sync()
    t = time.perf_counter()
    for _ in range(reps): f()
    ctx.sync()
    return (time.perf_counter() - t) / reps

variants = [int(v) for v in sys.argv[3].split(",")] if len(sys.argv) > 3 else [0, 3]
for rnd in range(3):                   # interleaved rounds: order effects show up as round-to-round spread
    print("round %d  copy %7.0f GB/s" % (rnd, gb / timed(lambda: ctx.select_limbs(a.ptr, b.ptr, count, L, list(range(L))))))
    for v in variants:
        ctx.set_ntt_variant(v)
        tfw = timed(lambda: ctx.nntt(a.ptr, b.ptr, count, L))
        tiv = timed(lambda: ctx.inntt(b.ptr, a.ptr, count, L))
        print("  variant %d  fwd %7.0f GB/s (%.1f us)   inv %7.0f GB/s (%.1f us)" % (v, gb / tfw, tfw * 1e6, gb / tiv, tiv * 1e6))
