"""Generate tests/golden/golden_v1.npz from the pure-Python spec oracle (oracle/spec.py).

The reference cannot be executed here (no Julia), so these vectors are NOT reference
outputs: they are the spec oracle's outputs at exactly the parameter sets the reference's own tests
use (test/bfv_crt.jl:8-36, test/ckks_rotate.jl:8-16, test/ckks_modraise.jl:10-20,
test/ckks_modswitch.jl:7-16, docs/src/man/background/rlwe.md:186-212), with seeded inputs.  They pin
the C oracle and the HIP engine to one another and to the doc known answers across refactors.

Run:  python tests/golden/make_golden.py     (deterministic; rewrites golden_v1.npz)
"""
import os
import random
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import spec  # noqa: E402


def arr(p):
    return np.array(p, dtype=np.uint64)


def rand_poly(rng, ring):
    return [[rng.randrange(q) for _ in range(ring.N)] for q in ring.qs]


def main():
    out = {}
    rng = random.Random(20260928)

    # --- doc vectors: 𝔽₉₇[x]/(x⁴+1), ψ = 33 (rlwe.md:186-212) ---
    out["doc97_q"] = arr([97]); out["doc97_psi"] = arr([33])
    out["doc97_in"] = arr([[1, 1, 0, 0], [0, 0, 0, 1], [4, 0, 0, 0], [5, 0, 0, 0]])
    out["doc97_prod"] = arr([[20, 0, 0, 0], [1, 2, 1, 0], [96, 0, 0, 1]])  # p3*p4, p1^2, p1*p2
    out["doc97_ntt"] = arr([spec.nntt(list(map(int, r)), 97, 33) for r in out["doc97_in"]])

    # --- NTT vectors on the reference's test rings ---
    for name, N, start, n in (("n16", 16, 2**40 + 1, 2), ("n32", 32, 2**40 + 1, 3), ("n2048", 2048, 2**50 + 1, 2)):
        qs = spec.prime_chain(start, n, N)
        ring = spec.Ring(N, qs)
        a = [rand_poly(rng, ring) for _ in range(2)]
        out[f"{name}_q"] = arr(qs); out[f"{name}_psi"] = arr(ring.psis)
        out[f"{name}_in"] = arr(a)
        out[f"{name}_ntt"] = arr([spec.poly_nntt(p, ring) for p in a])
    # PALISADE ring with explicit ψ (cryptparams.jl:25), one limb
    q, N, psi = 1152921504606830593, 2048, 811032584449645127
    a = [rng.randrange(q) for _ in range(N)]
    out["pal_q"] = arr([q]); out["pal_psi"] = arr([psi]); out["pal_in"] = arr([[a]]); out["pal_ntt"] = arr([[spec.nntt(a, q, psi)]])

    # --- modswitch + galois (ckks_modswitch.jl / ckks_rotate.jl rings) ---
    qs = spec.prime_chain(2**40 + 1, 3, 32); ring = spec.Ring(32, qs)
    a = [rand_poly(rng, ring) for _ in range(3)]
    out["ms_q"] = arr(qs); out["ms_in"] = arr(a); out["ms_out"] = arr([spec.modswitch_poly(p, ring) for p in a])
    for g in (3, 5, 63, pow(3, 15, 64)):
        out[f"gal{g}_out"] = arr([spec.poly_galois(p, g, ring) for p in a])

    # --- keyswitch with special prime at ckks_modraise.jl parameters (N=32, 3x40-bit, last = special) ---
    keyring = ring
    secret, _ = spec.keygen(rng, keyring, 3.2)
    evk = spec.make_eval_key(rng, secret, secret, keyring, 3.2, premul=qs[-1])
    cring = keyring.drop_last()
    cts = [[rand_poly(rng, cring) for _ in range(2)] for _ in range(2)]
    out["ksS_q"] = arr(qs)
    out["ksS_evk_ntt"] = arr([[spec.poly_nntt(m, keyring), spec.poly_nntt(md, keyring)] for m, md in evk])
    out["ksS_ct"] = arr(cts)
    out["ksS_out"] = arr([spec.keyswitch(evk, ct, cring, keyring, special=True) for ct in cts])
    # --- plain RNS-digit keyswitch of a 3-element ciphertext (relinearisation), N=32, 3 limbs ---
    evk2 = spec.make_eval_key(rng, spec.poly_mul(secret, secret, keyring), secret, keyring, 3.2)
    cts3 = [[rand_poly(rng, keyring) for _ in range(3)] for _ in range(2)]
    out["ksR_evk_ntt"] = arr([[spec.poly_nntt(m, keyring), spec.poly_nntt(md, keyring)] for m, md in evk2])
    out["ksR_ct"] = arr(cts3)
    out["ksR_out"] = arr([spec.keyswitch(evk2, ct, keyring, keyring, special=False) for ct in cts3])

    # --- BFV multiplication at exactly test/bfv_crt.jl parameters: N=2048, ℛ = 2x50-bit, ℛbig = next 4 ---
    N, t = 2048, 53
    ch = spec.prime_chain(2**50 + 1, 6, N)
    small, big = spec.Ring(N, ch[:2]), spec.Ring(N, ch[2:])
    s, pub = spec.keygen(rng, small, 3.2)
    ct = spec.encrypt_zero(rng, pub, small, 3.2)
    ct[0] = spec.poly_add(ct[0], spec.bfv_encode([6] + [0] * (N - 1), small, t), small)
    prod = spec.bfv_enc_mul(ct, ct, small, big, t)
    assert spec.bfv_decode(spec.decrypt_raw(s, prod, small), small, t)[0] == 36  # test/bfv_crt.jl:45-47
    out["bfvcrt_q"] = arr(ch); out["bfvcrt_t"] = arr([t])
    out["bfvcrt_ct"] = arr(ct); out["bfvcrt_prod"] = arr(prod); out["bfvcrt_secret"] = arr(s)
    # --- BFV multiplication, ℛbig ⊇ ℛ (the bench's basis relation), N=64, 3 + 4 limbs, t = 65537 ---
    N, t = 64, 65537
    ch = spec.prime_chain(2**50 + 1, 7, N)
    small, big = spec.Ring(N, ch[:3]), spec.Ring(N, ch)
    c1 = [rand_poly(rng, small) for _ in range(2)]; c2 = [rand_poly(rng, small) for _ in range(2)]
    out["bfvsup_q"] = arr(ch); out["bfvsup_t"] = arr([t])
    out["bfvsup_c1"] = arr(c1); out["bfvsup_c2"] = arr(c2)
    out["bfvsup_prod"] = arr(spec.bfv_enc_mul(c1, c2, small, big, t))

    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden_v1.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
