"""Shared helpers for the parity tests: seeded inputs, evaluation keys (built with the spec/C oracle
on the host, like the Julia callers keygen/make_eval_key would), device round trips."""
import random

import numpy as np

from oracle import ref_cpu, spec


def chain(bits, n, N):
    return spec.prime_chain(2**bits + 1, n, N)


def rand_residues(rng: np.random.Generator, qs, shape_prefix, N):
    """uniform residues, layout [*shape_prefix][len(qs)][N]"""
    cols = [rng.integers(0, int(q), size=tuple(shape_prefix) + (N,), dtype=np.uint64) for q in qs]
    return np.stack(cols, axis=len(shape_prefix))


def uniform_evk(rng, qs, ndig, N):
    """an evaluation key with uniform components (throughput runs): [ndig][2][Lk][N], NTT domain"""
    return rand_residues(rng, qs, (ndig, 2), N)


def real_evk(seed, N, qs, special, sigma=3.2):
    """a genuine RNS-gadget key switching key for s^2 -> s via the fast C oracle NTT.
    Returns (secret [Lk][N] coefficient domain, evk [Lk][2][Lk][N] NTT domain)."""
    prng = random.Random(seed)
    rng = np.random.default_rng(seed)
    Lk = len(qs)
    ctx = ref_cpu.RefCtx(N, qs)
    s_int = spec.sample_gauss_ints(prng, N, sigma)
    secret = np.array([[x % q for x in s_int] for q in qs], dtype=np.uint64)
    s_ntt = ctx.nntt(secret[None])[0]
    s2_ntt = ctx.pointwise("mul", s_ntt[None], s_ntt[None])[0]
    s2 = ctx.inntt(s2_ntt[None])[0]
    premul = int(qs[-1]) if special else 1
    evk = np.empty((Lk, 2, Lk, N), dtype=np.uint64)
    for i in range(Lk):
        mask = rand_residues(rng, qs, (), N)
        e_int = spec.sample_gauss_ints(prng, N, sigma)
        e = np.array([[x % q for x in e_int] for q in qs], dtype=np.uint64)
        g = np.zeros((Lk, N), dtype=np.uint64)
        g[i] = (s2[i].astype(object) * (premul % int(qs[i])) % int(qs[i])).astype(np.uint64)
        m_ntt = ctx.nntt(mask[None])[0]
        ms = ctx.inntt(ctx.pointwise("mul", m_ntt[None], s_ntt[None]))[0]
        masked = ctx.pointwise("sub", g[None], ctx.pointwise("add", ms[None], e[None]))[0]
        evk[i, 0] = m_ntt
        evk[i, 1] = ctx.nntt(masked[None])[0]
    return secret, evk
