"""Mirrors of the reference tests that round 1 left out (VERDICT r01 "Missing 5"), driven through the host mirror with
every ring operation on the device:

  test/ckks_rotate.jl:8-45    rotation with relin_window = 1 digit keys (K14 on a two-limb ring)
  test/ckks_matmul.jl:8-44    encrypted_matmul (diagonal method) with relin_window = 1 keys, steps = 4
  test/bfv_simd.jl:10-31 + docs/src/man/encoding.md:69-91   SlotEncoding = the user-visible NTT order
  test/bfv_noise.jl:5-34      invariant_noise_budget decreases along mul / keyswitch chains
  test/ckks_triv.jl:7-33      encode -> square -> decode, encrypt -> c*c -> decrypt at scale 2^40

Parameter sets the reference derives with BFVParams(p; eval_mult_count) (single big-integer moduli from the PALISADE-style
estimator, out of scope per SURVEY §1) are replaced by RNS rings of at least that size; everything else is as in the tests."""
import numpy as np
import pytest

import toyfhe_jl_amd as tf

pytestmark = pytest.mark.gpu


def chain(start, n, N):
    out, p = [], tf.nextprime(start, 1, 2 * N)
    for _ in range(n):
        out.append(p)
        p = tf.nextprime(p + 2 * N, 1, 2 * N)
    return out


def test_ckks_rotate_with_digit_window_keys():
    """test/ckks_rotate.jl:8-45: N = 2^4, (q0, ps) two 40-bit primes, scale 2^60, CKKSParams(R, 1, 3.2) -- NO special prime:
    the Galois keys are base-2 digit keys over the two-limb ring (81 key components)."""
    N = 16
    R = tf.NegacyclicRing(N, chain(2**40 + 1, 2, N))
    scale = 2**60
    plain = np.arange(1, N // 2 + 1).astype(complex)
    plain[0] += 1j                                                             # ckks_rotate.jl:21
    re = tf.ckks_encode(plain, R, scale)
    assert np.abs(tf.ckks_decode(re.apply_galois_element(3), scale) - np.roll(plain, -1)).max() < 1e-9   # :24
    params = tf.CKKSParams(R, 1, 3.2)
    rng = np.random.default_rng(8)
    kp = tf.keygen(rng, params)
    c = tf.encrypt(rng, kp, re, scale=scale)
    # :32-38: apply the automorphism to the ciphertext, switch from sigma_3(s) back to s
    c3 = tf.apply_galois_element(c, 3)
    ek = tf.make_eval_key(rng, kp.priv.secret.apply_galois_element(3), kp.priv)
    assert len(ek.key) == R.modulus().bit_length()                            # ndigits(Q, base = 2), rlwe_she.jl:282
    rt = tf.ckks_decode(tf.decrypt(kp, tf.keyswitch(ek, c3)), scale)
    assert np.allclose(rt, np.roll(plain, -1), atol=1e-6)                      # :39
    gk = tf.keygen_galois(rng, kp.priv, steps=1)
    got = tf.ckks_decode(tf.decrypt(kp, tf.rotate(gk, tf.encrypt(rng, kp, re, scale=scale))), scale)
    assert np.allclose(got, np.roll(plain, 1), atol=1e-6)                      # :43-45


def test_ckks_matmul_with_digit_window_keys():
    """test/ckks_matmul.jl:8-44: N = 2^5, three 40-bit primes, scale 2^40, relin_window = 1, Galois key for 4 steps,
    encrypted_matmul by diagonals with a 4 x 4 matrix of ones."""
    N = 32
    R = tf.NegacyclicRing(N, chain(2**40 + 1, 3, N))
    scale = 2**40
    plain = np.arange(1, N // 2 + 1).astype(complex)
    W = np.ones((4, 4), dtype=np.float32)
    params = tf.CKKSParams(R, 1, 3.2)
    rng = np.random.default_rng(9)
    kp = tf.keygen(rng, params)
    c = tf.encrypt(rng, kp, tf.ckks_encode(plain, R, scale), scale=scale)
    gk = tf.keygen_galois(rng, kp.priv, steps=4)

    def encrypted_matmul(gk, weights, x):                                      # ckks_matmul.jl:33-41
        n = weights.shape[1]
        result = x.mul_plain(np.tile(np.diag(weights), n))                    # repeat(diag(weights), n) .* x
        rotated = x
        for k in range(2, n + 1):
            rotated = tf.rotate(gk, rotated)
            result = result + rotated.mul_plain(np.tile(np.diag(np.roll(weights, k - 1, axis=1)), n))
        return result

    res = encrypted_matmul(gk, W, c)
    got = tf.ckks_decode(tf.decrypt(kp, res), res.scale)
    want = (W.astype(float) @ plain.reshape((4, 4), order="F").T).T            # (W * reshape(plain,4,4)')'
    assert np.allclose(got.reshape((4, 4), order="F"), want, atol=1e-5)        # :43


def test_slot_encoding_is_the_device_ntt_order():
    """docs/src/man/encoding.md:69-91 (and src/encoding.jl:35-52): the plaintext slots are the NTT-domain coefficients in the
    library's natural order -- a[0:9] = 1:10, b[:] .= 10, SlotEncoding(a * b)[0:10] = 10, 20, ..., 100, 0 -- through the
    device transforms of the plaintext ring Z_65537[x]/(x^2048 + 1)."""
    Rt = tf.NegacyclicRing(2048, [65537])
    a = tf.she.slot_encode(Rt, list(range(1, 11)) + [0] * 2038)
    b = tf.she.slot_encode(Rt, [10] * 2048)
    assert a.primal is None and b.primal is None
    a = tf.RingElement(Rt, a.coeffs_primal(), None)                            # force the product through inntt -> nntt
    b = tf.RingElement(Rt, b.coeffs_primal(), None)
    prod = a * b
    slots = tf.she.slot_decode(prod)
    assert slots[:11] == [10, 20, 30, 40, 50, 60, 70, 80, 90, 100, 0] and not any(slots[11:])
    # a constant slot vector is the constant polynomial
    assert b.to_ints() == [10] + [0] * 2047


def test_bfv_simd():
    """test/bfv_simd.jl:10-31: t = 65537, slot-wise product under encryption."""
    n, t = 2048, 65537
    ch = chain(2**50 + 1, 5, n)
    Rbig = tf.NegacyclicRing(n, ch)
    R = Rbig.crtselect(range(2))
    Rt = tf.NegacyclicRing(n, [t])
    params = tf.BFVParams(R, Rbig, t, 0, 3.2)
    rng = np.random.default_rng(10)
    kp = tf.keygen(rng, params)
    plain = tf.she.slot_encode(Rt, [1, 1] + [0] * (n - 2))                     # plain[0] = plain[1] = 1
    plain2 = tf.she.slot_encode(Rt, [5] + [10] * (n - 1))                      # plain2[:] .= 10; plain2[0] = 5
    c1 = tf.encrypt(rng, kp, plain.to_ints())
    c2 = tf.encrypt(rng, kp, plain2.to_ints())
    y = c1 * c2
    data = tf.she.slot_decode(Rt(tf.decrypt(kp, y)))
    assert data[0] == 5 and data[1] == 10 and not any(data[2:])               # :27-31


def test_bfv_noise_budget_decreases():
    """test/bfv_noise.jl:5-34: invariant_noise_budget (bfv.jl:137-166) shrinks with every multiplication, and a key switch
    costs little; t = 7, three multiplications."""
    n, t = 2048, 7
    ch = chain(2**50 + 1, 7, n)
    Rbig = tf.NegacyclicRing(n, ch)
    R = Rbig.crtselect(range(3))
    params = tf.BFVParams(R, Rbig, t, 0, 3.2)
    rng = np.random.default_rng(11)
    kp1 = tf.keygen(rng, params)
    ek = tf.keygen_evalmult(rng, kp1.priv)
    c1 = tf.encrypt(rng, kp1, [2] + [0] * (n - 1))
    b1 = tf.invariant_noise_budget(kp1.priv, c1)
    c1squared = c1 * c1
    b2 = tf.invariant_noise_budget(kp1.priv, c1squared)
    assert b2 < b1                                                             # :19
    cswitch1 = tf.keyswitch(ek, c1squared)
    bswitch1 = tf.invariant_noise_budget(kp1.priv, cswitch1)
    cswitchmul = cswitch1 * c1
    bswitchmul = tf.invariant_noise_budget(kp1.priv, cswitchmul)
    assert bswitchmul < bswitch1 < b1                                          # :24
    cswitch2 = tf.keyswitch(ek, cswitchmul)
    bswitch2 = tf.invariant_noise_budget(kp1.priv, cswitch2)
    cswitchmul2 = cswitch2 * c1
    bswitchmul2 = tf.invariant_noise_budget(kp1.priv, cswitchmul2)
    assert bswitchmul2 < bswitch2 < bswitch1                                   # :29
    assert bswitchmul2 > 1                                                     # still decrypts: 2^4 = 16 = 2 (mod 7)
    assert tf.decrypt(kp1, cswitchmul2)[0] == 16 % t


def test_ckks_triv():
    """test/ckks_triv.jl:7-33: N/2 = 2048 slots, LinRange(0, 1, 2048) at scale 2^40; encode -> re*re -> decode at scale^2;
    encrypt -> decrypt; c*c (3-element) -> decrypt; all atol 1e-4."""
    N = 4096
    R = tf.NegacyclicRing(N, chain(2**50 + 1, 3, N))                           # >= 2^80 * headroom, as the estimator's ring
    scale = 2**40
    x = np.linspace(0.0, 1.0, N // 2)
    re = tf.ckks_encode(x.astype(complex), R, scale)
    assert np.allclose(tf.ckks_decode(re * re, scale * scale).real, x**2, atol=1e-4)     # :24
    params = tf.CKKSParams(R, 0, 3.2)
    rng = np.random.default_rng(12)
    kp = tf.keygen(rng, params)
    c = tf.encrypt(rng, kp, re, scale=scale)
    assert np.allclose(tf.ckks_decode(tf.decrypt(kp, c), scale).real, x, atol=1e-4)       # :31
    cc = c * c
    assert len(cc) == 3
    assert np.allclose(tf.ckks_decode(tf.decrypt(kp, cc), cc.scale).real, x**2, atol=1e-4)  # :32


def test_matmul_by_hoisted_rotations():
    """the diagonal-method product of ckks_matmul.jl:33-41 / infer.jl:140-149 with every rotation taken from the SAME ciphertext
    (one Galois key per step, tf.rotate_many: one digit decomposition for all of them) -- each rotation bit-identical to
    rotate(gk_k, c), the product within the test's tolerance."""
    N = 64
    R = tf.NegacyclicRing(N, chain(2**40 + 1, 4, N))
    params = tf.ModulusRaised(tf.CKKSParams(R, 0, 3.2))
    rng = np.random.default_rng(13)
    kp = tf.keygen(rng, params)
    scale = 2**40
    n = 8                                                                       # 8 x 8 matrix, 4 copies side by side in the 32 slots
    x = rng.normal(0, 1, N // 2)
    W = rng.normal(0, 1, (n, n))
    c = tf.encrypt(rng, kp, tf.ckks_encode(x.astype(complex), params.R_cipher(), scale), scale=scale)
    B = N // 2 // n
    gks = [tf.keygen_galois(rng, kp.priv, steps=k * B) for k in range(1, n)]
    rots = tf.rotate_many(gks, c)
    for gk, r in zip(gks, rots):                                                # bit-identical to the one-at-a-time rotation
        one = tf.rotate(gk, c)
        assert all(np.array_equal(a.to_numpy(), b.to_numpy()) for a, b in zip(r.cs, one.cs))
    diag = lambda k: np.repeat(np.array([W[i, (i - k) % n] for i in range(n)]), B)
    res = c.mul_plain(diag(0))
    for k in range(1, n):
        res = res + rots[k - 1].mul_plain(diag(k))
    got = tf.ckks_decode(tf.decrypt(kp, res), res.scale).real.reshape(n, B)
    want = W @ x.reshape(n, B)
    assert np.allclose(got, want, atol=1e-5)


@pytest.mark.parametrize("logn,bits,n_rot,batch,raised", [(6, [40, 40, 40, 40], 7, None, True), (12, [60, 40, 40, 60], 5, 3, True), (14, [50, 50, 50, 50], 4, 2, True),
                                                          (15, [40, 40, 40, 40], 3, 2, True), (16, [60, 40, 40, 40, 60], 3, 2, True),
                                                          (16, [60, 40, 40, 40, 40, 60], 2, 3, True), (16, [50, 50, 50, 50, 50], 2, 2, True),
                                                          (16, [40, 40, 40, 60], 2, 2, True),
                                                          (12, [60, 40, 40], 5, 3, False), (16, [50, 50, 50], 2, 2, False)])
def test_matmul_diag_is_bit_identical_to_rotate_many_and_dot(logn, bits, n_rot, batch, raised):
    """tfhe_matmul_diag (every step of the hoisted diagonal product run over all rotations at once) against the composition it
    replaces -- rotate_many, the forward transforms and CipherText.dot_plain -- word for word, on uniform rings, on the
    reference's mixed 60/40-bit CKKS rings (infer.jl:97-112) and through the sub-block transforms of N = 2^15 / 2^16; the
    diagonals given as a list of plaintext elements and as one stacked element.  With the special prime (raised) the product
    finishes the rotations in the evaluation domain (k_md_*: one inverse transform per key sum, the unsigned lift of the
    permuted special limb; at N = 2^16 fused into the forward transforms, with an even number of limbs through the dense
    masked walk, with fp64-size limbs under a 60-bit special prime through the two-halves lift); rotate_many is the
    coefficient-domain path (k_ks_rot_tail / ks_finish): the two must agree bit for bit.  Without the special prime both
    take the coefficient-domain tail."""
    N = 1 << logn
    qs, used = [], set()
    for b in bits:
        q = tf.nextprime(2**b + 1, 1, 2 * N)
        while q in used:
            q = tf.nextprime(q + 2 * N, 1, 2 * N)
        used.add(q); qs.append(q)
    R = tf.NegacyclicRing(N, qs)
    params = tf.CKKSParams(R, 0, 3.2)
    if raised:
        params = tf.ModulusRaised(params)
    rng = tf.DeviceRng(1000 + logn)
    kp = tf.keygen(rng, params)
    scale = 2**30
    nrng = np.random.default_rng(logn)
    shape = (N // 2,) if batch is None else (batch, N // 2)
    x = nrng.normal(0, 1, shape).astype(complex)
    c = tf.encrypt(rng, kp, tf.ckks_encode(x, params.R_cipher(), scale), scale=scale)
    gks = [tf.keygen_galois(rng, kp.priv, steps=k) for k in range(1, n_rot + 1)]
    dv = nrng.normal(0, 1, (n_rot + 1, N // 2)).astype(complex)
    stacked = tf.ckks_encode(dv, params.R_cipher(), scale)
    singles = [tf.ckks_encode(dv[k], params.R_cipher(), scale) for k in range(n_rot + 1)]
    rots = tf.rotate_many(gks, c)
    bcast = [d if batch is None else d.broadcast_to(batch) for d in singles]
    want = tf.CipherText.dot_plain([c] + list(rots), bcast)
    for diags in (singles, stacked):
        got = tf.matmul_diag(gks, diags, c)
        assert got.scale == want.scale and len(got) == 2
        for a, b in zip(got.cs, want.cs):
            assert a.primal is None                                             # NTT-domain result, like dot_plain's
            assert np.array_equal(a.to_numpy("dual"), b.to_numpy("dual"))
    dec = tf.ckks_decode(tf.decrypt(kp, got), got.scale)
    ref = dv[0] * x + sum(dv[k] * np.roll(x, k, axis=-1) for k in range(1, n_rot + 1))
    if raised:   # (without the special prime the RNS-digit key switch adds noise of the size of a limb: only the words are compared)
        assert np.abs(dec - ref).max() < (1e-4 if logn < 10 else 5e-2)         # scale 2^30: fresh noise ~ sqrt(N) sigma / 2^30 per term
    # a lower level of the same keys (downswitch_keyelement, modulusraising.jl:43-49) and no rotation at all
    lo = tf.modswitch(c)
    dl = tf.ckks_encode(dv, lo.ring(), lo.scale)
    got = tf.matmul_diag(gks, dl, lo)
    want = tf.CipherText.dot_plain([lo] + list(tf.rotate_many(gks, lo)), [tf.ckks_encode(dv[k], lo.ring(), lo.scale) if batch is None else
                                                                          tf.ckks_encode(dv[k], lo.ring(), lo.scale).broadcast_to(batch) for k in range(n_rot + 1)])
    assert all(np.array_equal(a.to_numpy("dual"), b.to_numpy("dual")) for a, b in zip(got.cs, want.cs))
    only = tf.matmul_diag([], [singles[0]], c)
    w0 = c.mul_plain(singles[0] if batch is None else singles[0].broadcast_to(batch))
    assert all(np.array_equal(a.to_numpy("dual"), b.to_numpy("dual")) for a, b in zip(only.cs, w0.cs))


@pytest.mark.parametrize("seed", range(12))
def test_matmul_diag_random_small_shapes(seed):
    """Random small rings (N = 2^4 .. 2^9, one to four ciphertext limbs of 30-60 bits, with and without the special prime, one
    to six rotations, single and batched): tfhe_matmul_diag against rotate_many + dot_plain, word for word -- the
    evaluation-domain form at every limb count (a single limb, odd and even counts: both masked walks) and the k_md_lift path."""
    rs = np.random.default_rng(900 + seed)
    logn = int(rs.integers(4, 10)); N = 1 << logn
    L = int(rs.integers(1, 5)); raised = bool(rs.integers(0, 4))           # mostly with the special prime
    qs, used = [], set()
    for _ in range(L + (1 if raised else 0)):
        q = tf.nextprime(2 ** int(rs.choice([30, 40, 50, 60])) + 1, 1, 2 * N)
        while q in used:
            q = tf.nextprime(q + 2 * N, 1, 2 * N)
        used.add(q); qs.append(q)
    params = tf.CKKSParams(tf.NegacyclicRing(N, qs), 0, 3.2)
    if raised:
        params = tf.ModulusRaised(params)
    rng = tf.DeviceRng(7000 + seed)
    kp = tf.keygen(rng, params)
    n_rot = int(rs.integers(1, 7)); batch = [None, 2, 3][int(rs.integers(0, 3))]
    shape = (N // 2,) if batch is None else (batch, N // 2)
    scale = 2**20
    c = tf.encrypt(rng, kp, tf.ckks_encode(rs.normal(0, 1, shape).astype(complex), params.R_cipher(), scale), scale=scale)
    gks = [tf.keygen_galois(rng, kp.priv, steps=int(k)) for k in rs.choice(np.arange(1, N // 2), n_rot, replace=False)]
    dv = rs.normal(0, 1, (n_rot + 1, N // 2)).astype(complex)
    singles = [tf.ckks_encode(dv[k], params.R_cipher(), scale) for k in range(n_rot + 1)]
    want = tf.CipherText.dot_plain([c] + list(tf.rotate_many(gks, c)), [d if batch is None else d.broadcast_to(batch) for d in singles])
    got = tf.matmul_diag(gks, singles, c)
    for a, b in zip(got.cs, want.cs):
        assert np.array_equal(a.to_numpy("dual"), b.to_numpy("dual")), (logn, qs, raised, n_rot, batch)


def test_lincomb_equals_the_sum_of_scalar_products():
    """CipherText.lincomb (tfhe_lincomb: the 49 scalar-weighted terms of a convolution channel, infer.jl:127-129, in one pass per
    component) against sum(c.mul_plain(w)) word for word, in both domains, single and batched."""
    N = 256
    R = tf.NegacyclicRing(N, chain(2**40 + 1, 2, N) + [tf.nextprime(2**60 + 1, 1, 2 * N)])
    params = tf.CKKSParams(R, 0, 3.2)
    rng = tf.DeviceRng(77)
    kp = tf.keygen(rng, params)
    scale = 2**40
    nrng = np.random.default_rng(5)
    for batch in (None, 3):
        shape = (N // 2,) if batch is None else (batch, N // 2)
        cts = [tf.encrypt(rng, kp, tf.ckks_encode(nrng.normal(0, 1, shape).astype(complex), R, scale), scale=scale) for _ in range(49)]
        ws = list(nrng.normal(0, 0.2, 47)) + [0.0, -1.5]
        want = None
        for c, w in zip(cts, ws):
            t = c.mul_plain(float(w))
            want = t if want is None else want + t
        got = tf.CipherText.lincomb(cts, ws)
        assert got.scale == want.scale
        for a, b in zip(got.cs, want.cs):
            assert np.array_equal(a.to_numpy(), b.to_numpy())
        for c in cts:                                                           # NTT-domain operands
            for x in c.cs:
                x.coeffs_dual(); x.primal = None
        got = tf.CipherText.lincomb(cts, ws)
        for a, b in zip(got.cs, want.cs):
            assert a.primal is None and np.array_equal(a.to_numpy("dual"), b.to_numpy("dual"))
        # several sums of the same operands in one pass (tfhe_lincomb_many: 4 per launch, so 6 rows take two) = one call per row
        rows = [ws] + [list(nrng.normal(0, 0.3, 49)) for _ in range(5)]
        many = tf.CipherText.lincomb_many(cts, rows)
        assert len(many) == 6
        for row, m in zip(rows, many):
            one = tf.CipherText.lincomb(cts, row)
            assert m.scale == one.scale
            for a, b in zip(m.cs, one.cs):
                assert np.array_equal(a.to_numpy("dual"), b.to_numpy("dual"))
    with pytest.raises(AssertionError):
        tf.CipherText.lincomb_many(cts, [ws, ws[:-1]])
    with pytest.raises(AssertionError):
        tf.CipherText.lincomb(cts, ws[:-1])


def test_more_than_one_device_pass_of_terms_and_rotations():
    """A device pass takes 64 operands / rotations (TFHE_DOT_MAX): longer weighted sums and diagonal products are split on the
    host into partial sums (she.DOT_MAX) -- same words as the term-by-term / rotate_many + dot_plain compositions."""
    N = 256                                                                    # 127 distinct rotation steps
    qs = chain(2**40 + 1, 3, N)
    params = tf.ModulusRaised(tf.CKKSParams(tf.NegacyclicRing(N, qs), 0, 3.2))
    rng = tf.DeviceRng(4100)
    kp = tf.keygen(rng, params)
    R = params.R_cipher()
    scale = 2**30
    nrng = np.random.default_rng(41)
    cts = [tf.encrypt(rng, kp, tf.ckks_encode(nrng.normal(0, 1, (N // 2,)).astype(complex), R, scale), scale=scale) for _ in range(70)]
    ws = list(nrng.normal(0, 0.3, 70))
    want = None
    for c, w in zip(cts, ws):
        t = c.mul_plain(float(w))
        want = t if want is None else want + t
    got = tf.CipherText.lincomb(cts, ws)
    for a, b in zip(got.cs, want.cs):
        assert np.array_equal(a.to_numpy(), b.to_numpy())
    n_rot = 67
    c = cts[0]
    gks = [tf.keygen_galois(rng, kp.priv, steps=int(k)) for k in range(1, n_rot + 1)]
    singles = [tf.ckks_encode(nrng.normal(0, 1, (N // 2,)).astype(complex), R, scale) for _ in range(n_rot + 1)]
    rots = list(tf.rotate_many(gks[:64], c)) + list(tf.rotate_many(gks[64:], c))
    want = tf.CipherText.dot_plain([c] + rots[:63], singles[:64]) + tf.CipherText.dot_plain(rots[63:], singles[64:])
    got = tf.matmul_diag(gks, singles, c)
    for a, b in zip(got.cs, want.cs):
        assert np.array_equal(a.to_numpy("dual"), b.to_numpy("dual"))
    stacked = tf.RingElement.concat(singles)
    got2 = tf.matmul_diag(gks, stacked, c)
    for a, b in zip(got2.cs, got.cs):
        assert np.array_equal(a.to_numpy("dual"), b.to_numpy("dual"))
