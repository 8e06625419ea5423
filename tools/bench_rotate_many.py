#!/usr/bin/env python3
"""Hoisted rotations (tfhe_rotate_many) against one tfhe_rotate per Galois element: device time per rotation.
usage: bench_rotate_many.py <logN> <bits|mixed> <level> <batch> <n_rot>"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import toyfhe_jl_amd as tf
logn, kind, level, batch, nrot = int(sys.argv[1]), sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
N = 1 << logn
def chain(start, n):
    out, p = [], tf.nextprime(start, 1, 2 * N)
    for _ in range(n):
        out.append(p); p = tf.nextprime(p + 2 * N, 1, 2 * N)
    return out
if kind == "mixed":
    q0, ps = chain(2**60 + 1, 2); qs = [q0] + chain(2**40 + 1, level - 1) + [ps]
else:
    qs = chain(2**int(kind) + 1, level + 1)
Lk = level + 1
ctx = tf.Context(N, qs)
evks = []
for r in range(nrot):
    e = tf.DeviceBuffer(Lk * 2 * Lk * N); ctx.sample_uniform(Lk, 1 + r, 0, 0, e.ptr, Lk * 2); evks.append(e)
gs = [pow(3, r + 1, 2 * N) for r in range(nrot)]
ct = tf.DeviceBuffer(batch * 2 * level * N); ctx.sample_uniform(level, 99, 0, 0, ct.ptr, batch * 2)
out = tf.DeviceBuffer(nrot * batch * 2 * level * N)
def timed(f, reps=3):
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.2:
        f(); ctx.sync()
    best = 1e9
    for _ in range(3):
        ctx.sync(); t = time.perf_counter()
        for _ in range(reps): f()
        ctx.sync(); best = min(best, (time.perf_counter() - t) / reps)
    return best
sz = batch * 2 * level * N * 8
prep = []
for r in range(nrot):
    p_ = tf.DeviceBuffer(Lk * 2 * Lk * N); ctx.galois_key_prepare(Lk, Lk, gs[r], evks[r].ptr, p_.ptr); prep.append(p_)
t_many = timed(lambda: ctx.rotate_many(Lk, level, True, [e.ptr for e in prep], Lk, gs, ct.ptr, out.ptr, batch, prepared=True))
def one_by_one():
    for r in range(nrot):
        ctx.rotate(Lk, level, True, evks[r].ptr, Lk, gs[r], ct.ptr, out.ptr + r * sz, batch)
t_one = timed(one_by_one)
print(f"N=2^{logn} {kind} level {level} batch {batch}, {nrot} rotations: hoisted {nrot * batch / t_many:8.0f} rot/s   one by one {nrot * batch / t_one:8.0f} rot/s   x{t_one / t_many:.2f}")
