#!/usr/bin/env python3
"""BFV ct*ct + relinearise rate at other ring degrees (device time, synthetic residues): usage bench_bfv_n.py [logn ...]
N = 2^12 / 2^13 are the sizes of the reference's own BFV tests and MNIST parameters (test/bfv_crt.jl:8, infer.jl:97)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import toyfhe_jl_amd as tf


def chain(bits, n, N):
    out, p = [], tf.nextprime(2**bits + 1, 1, 2 * N)
    for _ in range(n):
        out.append(p); p = tf.nextprime(p + 2 * N, 1, 2 * N)
    return out


for logn in [int(a) for a in sys.argv[1:]] or [12, 13, 14]:
    N = 1 << logn
    for ns, ne in ((3, 4), (8, 9)):
        ch = chain(50, ns + ne, N)
        ctx = tf.Context(N, ch)
        plan = tf.BfvPlan(ctx, ctx, 65537, idx_s=list(range(ns)))
        B = 1024 if ns == 8 else 4096
        d1 = tf.DeviceBuffer(B * 2 * ns * N); ctx.sample_uniform(ns, 1, 0, 0, d1.ptr, B * 2)
        d2 = tf.DeviceBuffer(B * 2 * ns * N); ctx.sample_uniform(ns, 2, 0, 0, d2.ptr, B * 2)
        ek = tf.DeviceBuffer(ns * 2 * ns * N); ctx.sample_uniform(ns, 3, 0, 0, ek.ptr, ns * 2)
        out = tf.DeviceBuffer(B * 2 * ns * N)
        f = lambda: plan.mul_relin(ek.ptr, ns, d1.ptr, d2.ptr, out.ptr, B)
        for _ in range(3): f()
        ctx.sync(); best = 1e9
        for _ in range(3):
            t = time.perf_counter()
            for _ in range(4): f()
            ctx.sync(); best = min(best, (time.perf_counter() - t) / 4)
        print(f"N=2^{logn} ns={ns} ext={ne} batch={B}: {B / best:10.0f} ct-mul/s")
