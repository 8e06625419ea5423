#!/usr/bin/env python3
"""Secondary measurements on the other BASELINE.json configurations (not the bench.py metric): device time per
operation on synthetic residues, 1 GPU.
  cfg#3  CKKS  N=2^15, 10 limbs @40 bit + special prime: rotate (galois + key switch), rescale
  cfg#4        N=2^14,  6 limbs @50 bit + special prime: key switch only
  cfg#5        N=2^16,  6 limbs @50 bit + special prime: key switch / rotate; 7 limbs: forward / inverse NTT
usage: bench_configs.py [batch]"""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import toyfhe_jl_amd as tf
from tests import helpers as H

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 128

def timed(ctx, f, reps=5):
    for _ in range(2): f()
    ctx.sync(); t = time.perf_counter()
    for _ in range(reps): f()
    ctx.sync(); return (time.perf_counter() - t) / reps

def fill(ctx, words):
    b = tf.DeviceBuffer(words)
    tf.native.check(tf.native.lib().tfhe_memset(ctx.h, b.ptr, 1, words * 8))
    return b

def keyswitch_case(name, logn, bits, level, b):
    N = 1 << logn
    Lk = level + 1
    qs = H.chain(bits, Lk, N)
    ctx = tf.Context(N, qs)
    evk = fill(ctx, level * 2 * Lk * N)
    ct, out = fill(ctx, b * 2 * level * N), fill(ctx, b * 2 * level * N)
    t_ks = timed(ctx, lambda: ctx.keyswitch(Lk, level, True, evk.ptr, level, ct.ptr, 2, out.ptr, b))
    t_rot = timed(ctx, lambda: ctx.rotate(Lk, level, True, evk.ptr, level, 3, ct.ptr, out.ptr, b))
    res = fill(ctx, b * 2 * (level - 1) * N)
    t_rs = timed(ctx, lambda: ctx.rescale(ct.ptr, res.ptr, b * 2, level))
    print(f"{name}: N=2^{logn}, level {level} (+special), batch {b}: keyswitch {b / t_ks:9.0f}/s  rotate {b / t_rot:9.0f}/s  "
          f"rescale {b / t_rs:9.0f} ct/s")

keyswitch_case("cfg#3", 15, 40, 10, batch)
keyswitch_case("cfg#4", 14, 50, 6, batch * 4)
keyswitch_case("cfg#5", 16, 50, 6, max(8, batch // 2))
N = 1 << 16
ctx = tf.Context(N, H.chain(50, 7, N))
rows = 7 * max(8, batch // 4)
a, b2 = fill(ctx, rows * N), fill(ctx, rows * N)
gb = rows * N * 16 / 1e9
tf_ = timed(ctx, lambda: ctx.nntt(a.ptr, b2.ptr, rows // 7, 7)); ti = timed(ctx, lambda: ctx.inntt(b2.ptr, a.ptr, rows // 7, 7))
print(f"cfg#5: N=2^16, 7 limbs: NTT fwd {gb / tf_:6.0f} GB/s  inv {gb / ti:6.0f} GB/s")
