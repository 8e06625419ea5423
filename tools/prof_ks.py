#!/usr/bin/env python3
"""One key-switch configuration in a loop, for rocprofv3 --kernel-trace --stats.
usage: prof_ks.py <logN> <mixed|50|40|60> <level> <batch> [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import toyfhe_jl_amd as tf

logn, kind, level, batch = int(sys.argv[1]), sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 5
N = 1 << logn


def chain(start, n):
    out, p = [], tf.nextprime(start, 1, 2 * N)
    for _ in range(n):
        out.append(p); p = tf.nextprime(p + 2 * N, 1, 2 * N)
    return out


if kind == "mixed":
    q0, ps = chain(2**60 + 1, 2)
    qs = [q0] + chain(2**40 + 1, level - 1) + [ps]
else:
    qs = chain(2**int(kind) + 1, level + 1)
Lk = level + 1
ctx = tf.Context(N, qs)
evk = tf.DeviceBuffer(Lk * 2 * Lk * N); ctx.sample_uniform(Lk, 1, 0, 0, evk.ptr, Lk * 2)
ct = tf.DeviceBuffer(batch * 2 * level * N); ctx.sample_uniform(level, 2, 0, 0, ct.ptr, batch * 2)
out = tf.DeviceBuffer(batch * 2 * level * N)
for _ in range(reps):
    ctx.keyswitch(Lk, level, True, evk.ptr, Lk, ct.ptr, 2, out.ptr, batch)
ctx.sync()
print("done", qs)
