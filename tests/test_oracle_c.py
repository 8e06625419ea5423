"""The C restatement (oracle/ref_cpu.c) against the pure-Python spec oracle, bit for bit."""
import random

import numpy as np
import pytest

from oracle import ref_cpu, spec


def _rand_polys(rng, ring, count):
    return np.array([[[rng.randrange(q) for _ in range(ring.N)] for q in ring.qs] for _ in range(count)],
                    dtype=np.uint64)


def _L(a):
    return [[int(x) for x in limb] for limb in a]


def test_bignum_divrem():
    rng = random.Random(7)
    for _ in range(300):
        ub, vb = rng.randrange(1, 1400), rng.randrange(1, 900)
        u, v = rng.getrandbits(ub), rng.getrandbits(vb) | 1
        assert ref_cpu.divrem(u, v) == divmod(u, v)
    # Knuth-D corner: qhat over-estimate and add-back
    B = 2**64
    for u, v in [((B - 1) * B**3, (B - 1) * B + 1), (B**4 - 1, B**2 - 1), (B**3, B**2 // 2 + 1),
                 (0x7fffffffffffffff * B**2 + (B - 1), 0x8000000000000000 * B + 1)]:
        assert ref_cpu.divrem(u, v) == divmod(u, v)


@pytest.mark.parametrize("N,nq,start", [(4, 1, 97), (16, 2, 2**40 + 1), (64, 3, 2**50 + 1), (256, 2, 2**60 + 1)])
def test_ring_ops(N, nq, start):
    rng = random.Random(N)
    qs = [97] if start == 97 else spec.prime_chain(start, nq, N)
    ring = spec.Ring(N, qs)
    ctx = ref_cpu.RefCtx(N, qs)
    assert ctx.psis == ring.psis
    a = _rand_polys(rng, ring, 3); b = _rand_polys(rng, ring, 3)
    fw = ctx.nntt(a)
    for i in range(3):
        assert _L(fw[i]) == spec.poly_nntt(_L(a[i]), ring)
    assert np.array_equal(ctx.inntt(fw), a)
    for op, f in (("add", spec.poly_add), ("sub", spec.poly_sub), ("mul", spec.poly_pointwise)):
        r = ctx.pointwise(op, a, b)
        for i in range(3):
            assert _L(r[i]) == f(_L(a[i]), _L(b[i]), ring)
    assert _L(ctx.pointwise("neg", a)[0]) == spec.poly_neg(_L(a[0]), ring)
    s = rng.getrandbits(200)
    assert _L(ctx.scalar_mul([s % q for q in qs], a)[1]) == spec.poly_scalar_mul(s, _L(a[1]), ring)
    for g in (3, 5, 2 * N - 1, pow(3, N // 2 - 1, 2 * N)):
        assert _L(ctx.galois(g, a)[2]) == spec.poly_galois(_L(a[2]), g, ring)
    if nq > 1:
        ms = ctx.modswitch(a)
        for i in range(3):
            assert _L(ms[i]) == spec.modswitch_poly(_L(a[i]), ring)


def test_explicit_psi_context():
    # cryptparams.jl:25 — a non-derived ψ must be honoured
    q, N, psi = 1152921504606830593, 2048, 811032584449645127
    ctx = ref_cpu.RefCtx(N, [q], [psi])
    assert ctx.psis == [psi]
    rng = random.Random(1)
    a = np.array([[[rng.randrange(q) for _ in range(N)]]], dtype=np.uint64)
    assert _L(ctx.nntt(a)[0]) == [spec.nntt(_L(a[0])[0], q, psi)]
    with pytest.raises(ValueError):
        ref_cpu.RefCtx(N, [q], [3])


@pytest.mark.parametrize("disjoint", [True, False])
def test_bfv_switch_contract_and_mul(disjoint):
    N, t = 32, 65537
    ch = spec.prime_chain(2**50 + 1, 7, N)
    small = spec.Ring(N, ch[:3])
    big = spec.Ring(N, ch[3:] if disjoint else ch)
    cs, cb = ref_cpu.RefCtx(N, small.qs), ref_cpu.RefCtx(N, big.qs)
    rng = random.Random(11)
    a = _rand_polys(rng, small, 4)
    # force the centring edge cases: 0, 1, q-1, floor(q/2), floor(q/2)+1
    edge = [0, 1, small.Q - 1, small.Q // 2, small.Q // 2 + 1]
    for k, x in enumerate(edge):
        for l, q in enumerate(small.qs):
            a[0, l, k] = x % q
    sw = ref_cpu.switch(cs, cb, a)
    for i in range(4):
        assert _L(sw[i]) == spec.switch_poly(_L(a[i]), small, big)
    y = _rand_polys(rng, big, 3)
    edge = [0, 1, big.Q - 1, big.Q // 2, big.Q // 2 + 1, small.Q // 2, small.Q // 2 + 1, small.Q]
    tinv = pow(t, -1, big.Q)
    for k, x in enumerate(edge):
        for l, q in enumerate(big.qs):
            y[0, l, k] = (x * tinv) % big.Q % q     # so that t*y mod Qbig hits the edge value
    ct = ref_cpu.contract(cb, cs, t, y)
    for i in range(3):
        want = spec.switch_poly(spec.multround_poly(_L(y[i]), big, t, small.Q), big, small)
        assert _L(ct[i]) == want
    c1 = np.stack([_rand_polys(rng, small, 2) for _ in range(2)])
    c2 = np.stack([_rand_polys(rng, small, 2) for _ in range(2)])
    out = ref_cpu.bfv_mul(cs, cb, t, c1, c2)
    for b in range(2):
        want = spec.bfv_enc_mul([_L(p) for p in c1[b]], [_L(p) for p in c2[b]], small, big, t)
        assert [_L(p) for p in out[b]] == want


def test_enc_mul_plain():
    N = 32
    qs = spec.prime_chain(2**40 + 1, 3, N)
    ring = spec.Ring(N, qs); ctx = ref_cpu.RefCtx(N, qs)
    rng = random.Random(3)
    c1 = np.stack([_rand_polys(rng, ring, 2)]); c2 = np.stack([_rand_polys(rng, ring, 3)])
    out = ctx.enc_mul(c1, c2)
    want = spec.tensor([_L(p) for p in c1[0]], [_L(p) for p in c2[0]], ring)
    assert [_L(p) for p in out[0]] == want


@pytest.mark.parametrize("special,npolys", [(True, 2), (True, 3), (False, 3), (False, 2)])
def test_keyswitch(special, npolys):
    N = 32
    qs = spec.prime_chain(2**40 + 1, 4, N)
    keyring = spec.Ring(N, qs)
    rng = random.Random(5)
    secret, _ = spec.keygen(rng, keyring, 3.2)
    old = spec.poly_mul(secret, secret, keyring)
    evk = spec.make_eval_key(rng, old, secret, keyring, 3.2, premul=qs[-1] if special else 1)
    kctx = ref_cpu.RefCtx(N, qs)
    evk_ntt = np.array([[spec.poly_nntt(m, keyring), spec.poly_nntt(md, keyring)] for m, md in evk],
                       dtype=np.uint64)
    for l in ((3, 2) if special else (4, 3)):
        cring = keyring.select(range(l))
        ct = np.stack([_rand_polys(rng, cring, npolys) for _ in range(2)])
        out = kctx.keyswitch(l, special, evk_ntt, ct)
        for b in range(2):
            want = spec.keyswitch(evk, [_L(p) for p in ct[b]], cring, keyring, special=special)
            assert [_L(p) for p in out[b]] == want
