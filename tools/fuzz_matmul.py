#!/usr/bin/env python3
"""Random rings for tfhe_matmul_diag at N = 2^10 .. 2^16 (one to six ciphertext limbs of 30 / 40 / 50 / 60 bits in any mix, special
prime of 40 / 50 / 60 bits or none, one to eight rotations, batch 1-4): the product against rotate_many + dot_plain, word for word
(the evaluation-domain form against the coefficient-domain path: fused and unfused lifts, both masked walks, every policy split).
About five minutes on the GPU box.  usage: python tools/fuzz_matmul.py [first-seed] [seconds]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import toyfhe_jl_amd as tf

S0 = int(sys.argv[1]) if len(sys.argv) > 1 else 0
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 300.0
t0, bad, n = time.time(), [], 0
for seed in range(S0, S0 + 100000):
    if time.time() - t0 > budget:
        break
    rs = np.random.default_rng(seed)
    logn = int(rs.choice([10, 12, 13, 14, 15, 16, 16])); N = 1 << logn
    L = int(rs.integers(1, 7)); raised = bool(rs.integers(0, 5))
    qs, used = [], set()
    for k in range(L + (1 if raised else 0)):
        bits = int(rs.choice([40, 50, 60])) if (raised and k == L) else int(rs.choice([30, 40, 50, 60]))
        q = tf.nextprime(2 ** bits + 1, 1, 2 * N)
        while q in used:
            q = tf.nextprime(q + 2 * N, 1, 2 * N)
        used.add(q); qs.append(q)
    desc = (seed, logn, [q.bit_length() for q in qs], raised)
    try:
        params = tf.CKKSParams(tf.NegacyclicRing(N, qs), 0, 3.2)
        if raised:
            params = tf.ModulusRaised(params)
        rng = tf.DeviceRng(seed)
        kp = tf.keygen(rng, params)
        n_rot = int(rs.integers(1, 9)); batch = [None, 2, 3, 4][int(rs.integers(0, 4))]
        shape = (N // 2,) if batch is None else (batch, N // 2)
        scale = 2 ** 20
        c = tf.encrypt(rng, kp, tf.ckks_encode(rs.normal(0, 1, shape).astype(complex), params.R_cipher(), scale), scale=scale)
        if L > 1 and rs.integers(0, 2):
            c = tf.modswitch(c)                           # a lower level of the same keys
        gks = [tf.keygen_galois(rng, kp.priv, steps=int(k)) for k in rs.choice(np.arange(1, N // 2), n_rot, replace=False)]
        dv = rs.normal(0, 1, (n_rot + 1, N // 2)).astype(complex)
        singles = [tf.ckks_encode(dv[k], c.ring(), scale) for k in range(n_rot + 1)]
        want = tf.CipherText.dot_plain([c] + list(tf.rotate_many(gks, c)), [d if batch is None else d.broadcast_to(batch) for d in singles])
        got = tf.matmul_diag(gks, singles, c)
        ok = all(np.array_equal(a.to_numpy("dual"), b.to_numpy("dual")) for a, b in zip(got.cs, want.cs))
    except Exception as e:  # noqa: BLE001
        ok = False; desc = desc + (repr(e)[:200],)
    n += 1
    if not ok:
        bad.append(desc); print("FAIL", desc, flush=True)
print(f"{n} cases from seed {S0}; failures: {bad}")
