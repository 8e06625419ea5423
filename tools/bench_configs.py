#!/usr/bin/env python3
"""Secondary measurements on the other BASELINE.json configurations (not the bench.py metric): device time per
operation on synthetic residues, 1 GPU.  Every case first checks one ciphertext of its own batch against the C oracle
(bit for bit) and only then times.
  cfg#3  CKKS  N=2^15, 10 limbs @40 bit + special prime: rotate (galois + key switch), rescale       batch 512
  cfg#4        N=2^14,  6 limbs @50 bit + special prime: key switch only                             batch 512 (4096 / 8 GPUs)
  cfg#5  CKKS  N=2^16, the infer.jl ring 60 + 5 x 40 + 60 bit: key switch / rotate / rescale / NTT   batch 64
  cfg#5' same shape on a uniform 50-bit chain (fp64 policy throughout), for comparison
usage: bench_configs.py [scale]   (scale divides the batches; default 1)"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import toyfhe_jl_amd as tf
from oracle import ref_cpu            # checker only

scale = int(sys.argv[1]) if len(sys.argv) > 1 else 1


def chain(start, n, N):
    out, p = [], tf.nextprime(start, 1, 2 * N)
    for _ in range(n):
        out.append(p)
        p = tf.nextprime(p + 2 * N, 1, 2 * N)
    return out


def timed(ctx, f, reps=8):
    """best of three rounds after >= 0.1 s of warm-up (the oracle check before each case idles the GPU for seconds and the
    clocks take tens of milliseconds to come back)"""
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.1:
        f()
        ctx.sync()
    best = 1e9
    for _ in range(3):
        ctx.sync()
        t = time.perf_counter()
        for _ in range(reps):
            f()
        ctx.sync()
        best = min(best, (time.perf_counter() - t) / reps)
    return best


def uniform(ctx, level, count, seed):
    b = tf.DeviceBuffer(count * level * ctx.N)
    ctx.sample_uniform(level, seed, 0, 0, b.ptr, count)
    return b


def rows(buf, shape, k):
    n = int(np.prod(shape[1:]))
    out = np.empty(n, dtype=np.uint64)
    tf.native.check(tf.native.lib().tfhe_memcpy_d2h(out.ctypes.data, buf.ptr + k * n * 8, n * 8))
    return out.reshape((1,) + tuple(shape[1:]))


def keyswitch_case(name, N, qs, b):
    Lk, level = len(qs), len(qs) - 1
    ctx, ref = tf.Context(N, qs), ref_cpu.RefCtx(N, qs)
    evk = uniform(ctx, Lk, Lk * 2, 7)                        # [Lk digits][2][Lk][N]
    ct = uniform(ctx, level, b * 2, 8)
    out = tf.DeviceBuffer(b * 2 * level * N)
    g = pow(3, 2 * N - 1, 2 * N)
    evk_h = evk.to_numpy((Lk, 2, Lk, N))
    k = b // 2
    cin = rows(ct, (b, 2, level, N), k)
    ctx.keyswitch(Lk, level, True, evk.ptr, Lk, ct.ptr, 2, out.ptr, b)
    assert np.array_equal(rows(out, (b, 2, level, N), k), ref.keyswitch(level, True, evk_h, cin)), name + ": key switch differs from the oracle"
    ctx.rotate(Lk, level, True, evk.ptr, Lk, g, ct.ptr, out.ptr, b)
    want = ref.keyswitch(level, True, evk_h, ref.galois(g, cin.reshape(-1, level, N), idx=range(level)).reshape(cin.shape))
    assert np.array_equal(rows(out, (b, 2, level, N), k), want), name + ": rotate differs from the oracle"
    res = tf.DeviceBuffer(b * 2 * (level - 1) * N)
    ctx.rescale(out.ptr, res.ptr, b * 2, level)
    assert np.array_equal(rows(res, (b, 2, level - 1, N), k), ref.modswitch(want.reshape(-1, level, N), idx=range(level)).reshape(1, 2, level - 1, N))
    t_ks = timed(ctx, lambda: ctx.keyswitch(Lk, level, True, evk.ptr, Lk, ct.ptr, 2, out.ptr, b))
    t_rot = timed(ctx, lambda: ctx.rotate(Lk, level, True, evk.ptr, Lk, g, ct.ptr, out.ptr, b))
    t_rs = timed(ctx, lambda: ctx.rescale(ct.ptr, res.ptr, b * 2, level))
    print(f"{name}: N=2^{N.bit_length() - 1}, level {level} (+special), batch {b}: keyswitch {b / t_ks:9.0f}/s  rotate {b / t_rot:9.0f}/s  "
          f"rescale {b / t_rs:9.0f} ct/s   [oracle-checked]")
    return ctx, ref


def ntt_case(name, N, qs, polys):
    L = len(qs)
    ctx, ref = tf.Context(N, qs), ref_cpu.RefCtx(N, qs)
    a = uniform(ctx, L, polys, 9)
    b2 = tf.DeviceBuffer(polys * L * N)
    ctx.nntt(a.ptr, b2.ptr, polys, L)
    k = polys // 2
    assert np.array_equal(rows(b2, (polys, L, N), k), ref.nntt(rows(a, (polys, L, N), k))), name + ": NTT differs from the oracle"
    c = tf.DeviceBuffer(polys * L * N)
    ctx.inntt(b2.ptr, c.ptr, polys, L)
    assert np.array_equal(rows(c, (polys, L, N), k), rows(a, (polys, L, N), k))
    gb = polys * L * N * 16 / 1e9
    t_f = timed(ctx, lambda: ctx.nntt(a.ptr, b2.ptr, polys, L))
    t_i = timed(ctx, lambda: ctx.inntt(b2.ptr, c.ptr, polys, L))
    print(f"{name}: N=2^{N.bit_length() - 1}, {L} limbs, {polys} polys: NTT fwd {gb / t_f:6.0f} GB/s  inv {gb / t_i:6.0f} GB/s   [oracle-checked]")


N = 1 << 15
keyswitch_case("cfg#3", N, chain(2**40 + 1, 11, N), max(8, 512 // scale))
N = 1 << 14
keyswitch_case("cfg#4", N, chain(2**50 + 1, 7, N), max(8, 512 // scale))
N = 1 << 16
q0, ps = chain(2**60 + 1, 2, N)
mnist = [q0] + chain(2**40 + 1, 5, N) + [ps]
keyswitch_case("cfg#5 (infer.jl ring 60+5x40+60 bit)", N, mnist, max(8, 64 // scale))
keyswitch_case("cfg#5' (7 x 50 bit)", N, chain(2**50 + 1, 7, N), max(8, 64 // scale))
ntt_case("cfg#5 (infer.jl ring)", N, mnist, max(8, 128 // scale))
ntt_case("cfg#5' (7 x 50 bit)", N, chain(2**50 + 1, 7, N), max(8, 128 // scale))
N = 1 << 14
ntt_case("60-bit primes", N, chain(2**60 + 1, 8, N), max(8, 1024 // scale))
ntt_case("50-bit primes", N, chain(2**50 + 1, 8, N), max(8, 1024 // scale))
