#!/usr/bin/env python3
"""apply_galois_element rate at N = 2^15 / 2^16 (GB/s of 2 N 8 bytes per limb row) and the rotate rate of cfg#3 / cfg#5."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import toyfhe_jl_amd as tf


def chain(start, n, N):
    out, p = [], tf.nextprime(start, 1, 2 * N)
    for _ in range(n):
        out.append(p); p = tf.nextprime(p + 2 * N, 1, 2 * N)
    return out


def timed(ctx, f, reps=8):
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.1:
        f(); ctx.sync()
    best = 1e9
    for _ in range(3):
        ctx.sync(); t = time.perf_counter()
        for _ in range(reps): f()
        ctx.sync(); best = min(best, (time.perf_counter() - t) / reps)
    return best


for logn, L, polys in ((15, 10, 1024), (16, 6, 128), (16, 6, 1024)):
    N = 1 << logn
    qs = chain(2**40 + 1, L + 1, N)
    ctx = tf.Context(N, qs)
    a = tf.DeviceBuffer(polys * L * N); ctx.sample_uniform(L, 5, 0, 0, a.ptr, polys)
    b = tf.DeviceBuffer(polys * L * N)
    g = pow(3, 2 * N - 1, 2 * N)
    t = timed(ctx, lambda: ctx.galois(a.ptr, b.ptr, g, polys, L))
    print(f"galois N=2^{logn} {L} limbs x {polys} polys: {polys * L * N * 16 / t / 1e9:7.0f} GB/s")
    evk = tf.DeviceBuffer((L + 1) * 2 * (L + 1) * N); ctx.sample_uniform(L + 1, 6, 0, 0, evk.ptr, (L + 1) * 2)
    nb = polys // 2
    out = tf.DeviceBuffer(nb * 2 * L * N)
    t = timed(ctx, lambda: ctx.rotate(L + 1, L, True, evk.ptr, L + 1, g, a.ptr, out.ptr, nb), reps=4)
    print(f"rotate N=2^{logn} level {L} batch {nb}: {nb / t:9.0f} /s")
