"""The HIP kernels' per-thread bodies (toyfhe.jl_amd/csrc/*_core.h), run on the CPU by
tests/emul/ one thread id at a time, against the oracle.  This checks index logic, lazy-range
bounds and the exact-conversion slow path without a GPU; the GPU tests (-m gpu) then check the
real kernels through the C ABI."""
import random

import numpy as np
import pytest

from oracle import ref_cpu, spec
from tests.emul import emul


def _chain(bits, n, N):
    return spec.prime_chain(2**bits + 1, n, N)


@pytest.mark.parametrize("logn", [1, 2, 5, 9, 10, 11, 12, 13, 14])
@pytest.mark.parametrize("bits", [30, 50, 61])
def test_ntt_bodies_match_oracle(logn, bits):
    N = 1 << logn
    q = _chain(bits, 1, N)[0]
    ctx = ref_cpu.RefCtx(N, [q])
    rng = np.random.default_rng(logn * 100 + bits)
    a = rng.integers(0, q, size=N, dtype=np.uint64)
    a[0] = q - 1; a[-1] = 0   # range edges
    want = ctx.nntt(a.reshape(1, 1, N)).reshape(N)
    # variant 0: register-blocked (fp64 butterflies for the 30- and 50-bit primes, u64 for the 61-bit one),
    # 1: generic radix-2, 2: register-blocked with u64 butterflies forced
    for variant in ((0, 1, 2) if logn >= 10 else (1,)):
        got = emul.ntt(a, q, variant=variant)
        assert np.array_equal(got, want), (logn, bits, variant)
        back = emul.ntt(want, q, inverse=True, variant=variant)
        assert np.array_equal(back, a), (logn, bits, variant)


@pytest.mark.parametrize("logn", [10, 12, 14])
def test_fp64_butterflies_at_the_modulus_limit_and_worst_case_inputs(logn):
    """fp64arith.h is valid for q < 2^50 + 2^40: take the largest NTT-friendly prime below that bound and
    inputs that maximise growth (all q-1, alternating 0 / q-1, all (q-1)/2)."""
    N = 1 << logn
    q = (1126999418470400 // (2 * N)) * (2 * N) + 1
    while q >= 1126999418470400 or not spec.is_prime(q):
        q -= 2 * N
    assert q > 2**50 + 2**40 - 2**30
    ctx = ref_cpu.RefCtx(N, [q])
    pats = [np.full(N, q - 1, dtype=np.uint64), np.array([0, q - 1] * (N // 2), dtype=np.uint64),
            np.full(N, (q - 1) // 2, dtype=np.uint64), np.random.default_rng(logn).integers(0, q, size=N, dtype=np.uint64)]
    emul.fp_max_ratio_reset()
    for a in pats:
        want = ctx.nntt(a.reshape(1, 1, N)).reshape(N)
        assert np.array_equal(emul.ntt(a, q, variant=0), want)
        assert np.array_equal(emul.ntt(a, q, variant=2), want)
        assert np.array_equal(emul.ntt(want, q, inverse=True, variant=0), a)
        # inverse on growth-maximising NTT-domain inputs as well
        wi = ctx.inntt(a.reshape(1, 1, N)).reshape(N)
        assert np.array_equal(emul.ntt(a, q, inverse=True, variant=0), wi)
    # every operand of a modular product / reduction stayed inside the exactness budget: |v| < 2^53 (7.1 p at this modulus)
    worst = emul.fp_max_ratio_reset()
    assert 0 < worst < 2.0**53 / q, worst


@pytest.mark.parametrize("logn", [10, 13, 14, 15, 16])
def test_fp64_small_modulus_range_plan_at_its_limit(logn):
    """ArithFpS (fp64arith.h, r04): moduli below 2^42 run without forward sweeps and with one sweep per inverse.  At the largest
    NTT-friendly prime below 2^42, on growth-maximising inputs, the block passes give the oracle's words and every operand of a
    modular product / reduction stays below 2^53 (2048 p there; the plan's own limit is 2000)."""
    N = 1 << logn
    q = ((2**42) // (2 * N)) * (2 * N) + 1
    while q >= 2**42 or not spec.is_prime(q):
        q -= 2 * N
    assert q > 2**42 - 2**30
    ctx = ref_cpu.RefCtx(N, [q])
    pats = [np.full(N, q - 1, dtype=np.uint64), np.array([0, q - 1] * (N // 2), dtype=np.uint64),
            np.full(N, (q - 1) // 2, dtype=np.uint64), np.random.default_rng(logn).integers(0, q, size=N, dtype=np.uint64)]
    emul.fp_max_ratio_reset()
    for a in pats:
        want = ctx.nntt(a.reshape(1, 1, N)).reshape(N)
        assert np.array_equal(emul.ntt(a, q, variant=3), want)
        assert np.array_equal(emul.ntt(want, q, inverse=True, variant=3), a)
        wi = ctx.inntt(a.reshape(1, 1, N)).reshape(N)
        assert np.array_equal(emul.ntt(a, q, inverse=True, variant=3), wi)
    worst = emul.fp_max_ratio_reset()
    assert 1.0 < worst < 2000.0, worst          # it does use the room (no sweeps) and stays inside the plan's limit


@pytest.mark.parametrize("logn", [15, 16])
def test_ntt_bodies_large_n(logn):
    N = 1 << logn
    q = _chain(50, 1, N)[0]
    ctx = ref_cpu.RefCtx(N, [q])
    rng = np.random.default_rng(logn)
    a = rng.integers(0, q, size=N, dtype=np.uint64)
    want = ctx.nntt(a.reshape(1, 1, N)).reshape(N)
    assert np.array_equal(emul.ntt(a, q), want)
    assert np.array_equal(emul.ntt(want, q, inverse=True), a)


def test_ntt_bodies_explicit_psi_and_max_modulus():
    # explicit (non-minimal) psi: cryptparams.jl:25; and a prime just below 2^62 (lazy range 4q < 2^64)
    q, N, psi = 1152921504606830593, 2048, 811032584449645127
    rng = np.random.default_rng(1)
    a = rng.integers(0, q, size=N, dtype=np.uint64)
    want = np.array(spec.nntt([int(x) for x in a], q, psi), dtype=np.uint64)
    assert np.array_equal(emul.ntt(a, q, psi=psi), want)
    N = 4096
    q = 2**62 - 2 * N + 1
    while not (spec.is_prime(q)):
        q -= 2 * N
    a = rng.integers(0, q, size=N, dtype=np.uint64); a[:4] = q - 1
    ctx = ref_cpu.RefCtx(N, [q])
    want = ctx.nntt(a.reshape(1, 1, N)).reshape(N)
    assert np.array_equal(emul.ntt(a, q), want)
    assert np.array_equal(emul.ntt(want, q, inverse=True), a)
    with pytest.raises(RuntimeError):
        emul.ntt(a, q, psi=5)


def _conv_ref(a, t, res, centred):
    A = 1
    for x in a:
        A *= x
    out = []
    for r in res:
        x = spec.rns_to_int([int(v) for v in r], a)
        if centred:
            x = spec.centred(x, A)
        out.append([x % ti for ti in t])
    return np.array(out, dtype=np.uint64)


@pytest.mark.parametrize("bits,k,m", [(50, 8, 9), (50, 9, 8), (40, 3, 2), (61, 4, 5), (30, 17, 3), (50, 2, 4)])
@pytest.mark.parametrize("centred", [False, True])
def test_exact_conversion_random_and_edges(bits, k, m, centred):
    N = 64
    ch = _chain(bits, k + m, N)
    a, t = ch[:k], ch[k:]
    A = 1
    for x in a:
        A *= x
    rng = random.Random(bits * k + m)
    vals = [rng.randrange(A) for _ in range(200)]
    # structured values that sit on the alpha decision boundary / centring boundary
    vals += [0, 1, 2, A - 1, A - 2, A // 2, A // 2 + 1, A // 2 - 1, A // 2 + 2, 12345, A - 12345,
             (A // 2 + 1 + 5) % A, a[0], A // a[0], A - A // a[0]]
    res = np.array([[v % x for x in a] for v in vals], dtype=np.uint64)
    got, slow = emul.conv(a, t, res, centred)
    assert np.array_equal(got, _conv_ref(a, t, res, centred))
    assert slow > 0  # the structured values force the exact multi-word branch


def test_exact_conversion_target_equals_source():
    N = 64
    ch = _chain(50, 5, N)
    a, t = ch[:3], [ch[1], ch[3], ch[0], ch[4]]
    rng = random.Random(9)
    A = a[0] * a[1] * a[2]
    vals = [rng.randrange(A) for _ in range(50)] + [0, 1, A - 1, A // 2, A // 2 + 1]
    res = np.array([[v % x for x in a] for v in vals], dtype=np.uint64)
    for centred in (False, True):
        got, _ = emul.conv(a, t, res, centred)
        assert np.array_equal(got, _conv_ref(a, t, res, centred))


@pytest.mark.parametrize("mode", ["superset", "disjoint"])
@pytest.mark.parametrize("bits,ns,nextra", [(50, 3, 4), (40, 2, 3), (60, 2, 3)])
def test_bfv_expand_contract_bodies(mode, bits, ns, nextra):
    N, t = 32, 65537
    ch = _chain(bits, ns + nextra + ns, N)
    qs = ch[:ns]
    pb = (ch[: ns + nextra] if mode == "superset" else ch[ns: 2 * ns + nextra + 1])
    if mode == "superset":
        pb = pb[::-1]  # ℛbig limb order is arbitrary: put the shared primes last
    cs, cb = ref_cpu.RefCtx(N, qs), ref_cpu.RefCtx(N, pb)
    small, big = spec.Ring(N, qs), spec.Ring(N, pb)
    rng = np.random.default_rng(bits + ns)
    a = np.stack([rng.integers(0, q, size=(5, N), dtype=np.uint64) for q in qs], axis=1)
    for k, x in enumerate([0, 1, small.Q - 1, small.Q // 2, small.Q // 2 + 1, 7, small.Q - 7]):
        for l, q in enumerate(qs):
            a[0, l, k] = x % q
    got, _ = emul.bfv(qs, pb, t, a, N, contract=False)
    assert np.array_equal(got, ref_cpu.switch(cs, cb, a))
    y = np.stack([rng.integers(0, p, size=(6, N), dtype=np.uint64) for p in pb], axis=1)
    tinv = pow(t, -1, big.Q)
    edges = [0, 1, big.Q - 1, big.Q // 2, big.Q // 2 + 1, small.Q // 2, small.Q // 2 + 1, small.Q, small.Q - 1,
             3 * small.Q + small.Q // 2, 3 * small.Q + small.Q // 2 + 1, big.Q - small.Q // 2, big.Q - small.Q // 2 - 1]
    for k, x in enumerate(edges):
        for l, p in enumerate(pb):
            y[0, l, k] = (x * tinv) % big.Q % p
    got, slow = emul.bfv(qs, pb, t, y, N, contract=True)
    assert np.array_equal(got, ref_cpu.contract(cb, cs, t, y))
    assert slow > 0


def test_bfv_bodies_reject_partial_overlap():
    N = 32
    ch = _chain(50, 6, N)
    with pytest.raises(RuntimeError):
        emul.bfv(ch[:3], ch[1:6], 65537, np.zeros((1, 3, N), dtype=np.uint64), N, contract=False)


@pytest.mark.parametrize("bits,ns,np_", [(50, 8, 9), (50, 3, 4), (40, 2, 3), (60, 3, 4), (61, 6, 7)])
def test_bfv_fast_path_bodies(bits, ns, np_):
    """bfv_fast.h (folded constants, register-resident) against the oracle, incl. edge values that force the
    exact-alpha branch; ℛbig limb order shuffled."""
    N, t = 32, 65537
    ch = _chain(bits, ns + np_, N)
    qs = ch[:ns]
    pb = ch[ns:] + ch[:ns]          # P limbs first, shared limbs last
    pb = pb[1:] + pb[:1]            # rotate: positions are arbitrary
    cs, cb = ref_cpu.RefCtx(N, qs), ref_cpu.RefCtx(N, pb)
    small, big = spec.Ring(N, qs), spec.Ring(N, pb)
    rng = np.random.default_rng(bits + ns)
    a = np.stack([rng.integers(0, q, size=(5, N), dtype=np.uint64) for q in qs], axis=1)
    for k, x in enumerate([0, 1, small.Q - 1, small.Q // 2, small.Q // 2 + 1, 7, small.Q - 7]):
        for l, q in enumerate(qs):
            a[0, l, k] = x % q
    assert np.array_equal(emul.bfv_fast(qs, pb, t, a, N, contract=False), ref_cpu.switch(cs, cb, a))
    y = np.stack([rng.integers(0, p, size=(6, N), dtype=np.uint64) for p in pb], axis=1)
    tinv = pow(t, -1, big.Q)
    edges = [0, 1, big.Q - 1, big.Q // 2, big.Q // 2 + 1, small.Q // 2, small.Q // 2 + 1, small.Q, small.Q - 1,
             3 * small.Q + small.Q // 2, 3 * small.Q + small.Q // 2 + 1, big.Q - small.Q // 2, big.Q - small.Q // 2 - 1,
             small.Q * (big.Q // small.Q // 2), small.Q * (big.Q // small.Q // 2) + small.Q // 2 + 1]
    for k, x in enumerate(edges):
        for l, p in enumerate(pb):
            y[0, l, k] = (x * tinv) % big.Q % p
    assert np.array_equal(emul.bfv_fast(qs, pb, t, y, N, contract=True), ref_cpu.contract(cb, cs, t, y))
    if bits <= 50:
        # the same contraction on tensor rows handed over as reduced doubles (k_bfv_core_fused<.., OUTD> -> bfv_contract_narrow<.., TD>):
        # centred representatives, and the two extreme lazy forms +-(p - 1)/2 -+ 1 a transform output may take
        pbv = np.array(pb, dtype=np.int64).reshape(1, -1, 1)
        yc = y.astype(np.int64)
        yc = np.where(yc > pbv // 2, yc - pbv, yc)
        yd = yc.astype(np.float64).view(np.uint64)
        assert np.array_equal(emul.bfv_fast(qs, pb, t, yd, N, contract=2), ref_cpu.contract(cb, cs, t, y))
        ylazy = np.where(yc == pbv // 2, yc - pbv, yc)                # p/2 -> -(p/2 + 1): the other representative at the edge
        assert np.array_equal(emul.bfv_fast(qs, pb, t, ylazy.astype(np.float64).view(np.uint64), N, contract=2), ref_cpu.contract(cb, cs, t, y))



@pytest.mark.parametrize("bits,k,w", [(60, 1, 1), (61, 1, 7), (50, 2, 10), (40, 3, 13), (61, 4, 32), (50, 8, 16)])
def test_window_digit_bodies(bits, k, w):
    """conv_core.h window_digits_coeff (the body of k_ks_window_digits): base-2^w digits of the exactly reconstructed
    integer, against Python big integers (rlwe_she.jl:333-334: digits(convert(Integer, x), base = 2^w, pad = nwindows))."""
    qs = _chain(bits, k, 64)
    Q = 1
    for q in qs:
        Q *= q
    nwin = -(-Q.bit_length() // w)
    rng = random.Random(bits * 100 + k)
    xs = [0, 1, Q - 1, Q // 2, Q // 2 + 1, (1 << (Q.bit_length() - 1)), (1 << 64) % Q, ((1 << 64) - 1) % Q] + [rng.randrange(Q) for _ in range(300)]
    res = np.array([[x % q for q in qs] for x in xs], dtype=np.uint64)
    got = emul.window_digits(qs, w, nwin, res)
    want = np.array([[(x >> (i * w)) & ((1 << w) - 1) for i in range(nwin)] for x in xs], dtype=np.uint64)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("smant,sexp", [(1, 40), (1, 80), (1, 0), (12345, 30), (2**63 + 12345, -20), (1, 200)])
def test_ckks_fixed_point_conversions(smant, sexp):
    """ckks_core.h against exact rational arithmetic: round(BigInt, big(x) * scale) with ties to even, reduced mod q
    (ckks.jl:42-45), and Float64(n / scale) for multi-word n (ckks.jl:52-58)."""
    from fractions import Fraction
    scale = Fraction(smant) * Fraction(2) ** sexp
    q = _chain(50, 1, 64)[0]
    rng = random.Random(smant % 1000 + sexp)
    xs = [0.0, -0.0, 1.0, -1.0, 0.5, 1.5, 2.5, -2.5, 1e-30, -1e-30, 5e-324, 123456.789, -3.75e10, 2.0**-41, 3 * 2.0**-41, -(2.0**-41),
          float(Fraction(1, 2) / scale) if scale < 2**1000 else 0.0, float(Fraction(3, 2) / scale) if scale < 2**1000 else 0.0]
    xs += [rng.uniform(-4, 4) * 2.0 ** rng.randint(-60, 20) for _ in range(400)]
    got = emul.ckks_round(np.array(xs), smant, sexp, q)

    def rne(fr):
        fl = fr.numerator // fr.denominator
        rem = fr - fl
        return fl + (1 if rem > Fraction(1, 2) or (rem == Fraction(1, 2) and fl % 2 == 1) else 0)
    want = [rne(Fraction(x) * scale) % q for x in xs]
    assert [int(g) for g in got] == want
    # magnitude -> double: nearest double of n / scale (one rounding for power-of-two scales)
    for _ in range(300):
        bits = rng.randint(1, 400)
        mag = rng.getrandbits(bits) | (1 << (bits - 1))
        neg = rng.random() < 0.5
        got_d = emul.ckks_to_double(mag, neg, smant, sexp)
        exact = Fraction(-mag if neg else mag) / scale
        if smant == 1:
            assert got_d == float(exact), (mag, neg)
        else:
            assert abs(Fraction(got_d) - exact) <= abs(exact) * Fraction(1, 2**51)


@pytest.mark.parametrize("bits", [20, 40, 50])
def test_centred_double_hand_over(bits):
    """bfv_fast.h centred_double_bits (c2 between the BFV contraction and the fused key switch): the centred residue
    r - q (r > q/2) or r, as an exact double -- edges of the centring and random values."""
    q = _chain(bits, 1, 1 << 10)[0]
    rng = np.random.default_rng(bits)
    vals = [0, 1, 2, q // 2 - 1, q // 2, q // 2 + 1, q // 2 + 2, q - 2, q - 1] + [int(v) for v in rng.integers(0, q, size=200)]
    for r in vals:
        want = r - q if r > q // 2 else r          # SignedMod-style centred representative (rlwe_she.jl:326-329)
        got = emul.centred_double(r, q)
        assert got == float(want) and float(int(got)) == got, (r, q, got, want)


@pytest.mark.parametrize("bits,pbits", [(50, 50), (40, 50), (50, 40), (30, 45)])
def test_fused_contraction_store_arithmetic(bits, pbits):
    """ntt_core.h ArithFpMD::out_moddown (k_ks_fused SPMODE 2): (v - [t_P]) P^-1 + c mod q from a LAZY inverse-transform value
    (any integer |v| < 7.9 q), the special limb's coefficient t_P in [0, P) and the addend c in [0, q) -- the canonical residue
    of modulusraising.jl:35-42 / crt.jl:215-220 with the unsigned representative of t_P, at the edges of every range."""
    N = 1 << 10
    q = _chain(bits, 1, N)[0]
    P = _chain(pbits, 2, N)[1]
    pinv = pow(P % q, -1, q)
    rng = np.random.default_rng(bits * 64 + pbits)
    lim = int(7.9 * q)
    vs = [0, 1, -1, q, -q, q // 2, q // 2 + 1, -(q // 2) - 1, lim - 1, -(lim - 1)] + [int(x) for x in rng.integers(-lim + 1, lim, size=300)]
    ts = [0, 1, P - 1, P // 2, P // 2 + 1, q % P, (q - 1) % P] + [int(x) for x in rng.integers(0, P, size=40)]
    cs = [0, 1, q - 1, q // 2] + [int(x) for x in rng.integers(0, q, size=8)]
    for i, v in enumerate(vs):
        for t in (ts if i < 12 else ts[i % len(ts):][:3]):
            for c in (cs if i < 12 else cs[i % len(cs):][:2]):
                want = ((v - t) * pinv + c) % q
                assert emul.out_moddown(float(v), t, c, q, pinv) == want, (v, t, c)
