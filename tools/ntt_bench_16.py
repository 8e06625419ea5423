#!/usr/bin/env python3
"""N = 2^16 / 2^17 transform rates (algorithmic GB/s = 2 N 8 bytes per limb row), uniform 50-bit and the infer.jl mixed ring."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import toyfhe_jl_amd as tf


def chain(start, n, N):
    out, p = [], tf.nextprime(start, 1, 2 * N)
    for _ in range(n):
        out.append(p); p = tf.nextprime(p + 2 * N, 1, 2 * N)
    return out


def timed(ctx, f, reps=6):
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.1:
        f(); ctx.sync()
    best = 1e9
    for _ in range(3):
        ctx.sync(); t = time.perf_counter()
        for _ in range(reps): f()
        ctx.sync(); best = min(best, (time.perf_counter() - t) / reps)
    return best


for logn in (16, 17):
    N = 1 << logn
    q0, ps = chain(2**60 + 1, 2, N)
    rings = {"7x50": chain(2**50 + 1, 7, N), "60+5x40+60": [q0] + chain(2**40 + 1, 5, N) + [ps], "8x60": chain(2**60 + 1, 8, N)}
    for name, qs in rings.items():
        L = len(qs)
        ctx = tf.Context(N, qs)
        for polys in (128, 512):
            a = tf.DeviceBuffer(polys * L * N); ctx.sample_uniform(L, 9, 0, 0, a.ptr, polys)
            b = tf.DeviceBuffer(polys * L * N)
            gb = polys * L * N * 16 / 1e9
            tf_ = timed(ctx, lambda: ctx.nntt(a.ptr, b.ptr, polys, L))
            ti = timed(ctx, lambda: ctx.inntt(b.ptr, a.ptr, polys, L))
            tip = timed(ctx, lambda: ctx.inntt(b.ptr, b.ptr, polys, L))
            print(f"N=2^{logn} {name:11s} {polys:4d} polys: fwd {gb / tf_:6.0f}  inv {gb / ti:6.0f}  inv in place {gb / tip:6.0f} GB/s")
