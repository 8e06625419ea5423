"""Spec oracle for the ToyFHE.jl power-of-two-cyclotomic RNS path (pure-Python big-int).

TEST INFRASTRUCTURE ONLY.  Nothing under ``toyfhe.jl_amd/`` may import this module; only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg use ``oracle/``.

Every function restates one reference function *by definition* (schoolbook sums, exact big-int
CRT, exact rational rounding), with the reference file:line it follows.  It is slow and meant to
be obviously correct; ``oracle/ref_cpu.c`` is the fast restatement that is checked against this
one and then used at full size.

Pinning status: the reference cannot be executed here (no Julia; the NTT and
field arithmetic live in FourierTransforms.jl / GaloisFields.jl which are pinned only by git tree
hash in Manifest.toml:219-225,235-241 and are not on disk).  No reference test pins NTT outputs
or ciphertext bits.  This oracle is pinned against every known-answer datum the reference holds
for the path (docs/src/man/background/rlwe.md:186,207-212; docs/src/man/encoding.md:14-38,69-91;
src/cryptparams.jl:22-25) -- see tests/test_oracle_known_answers.py -- and against the
mathematical definition in src/pow2_cyc_rings.jl:278-294.  Beyond those data: PARITY UNPINNED.

All paths below are relative to /root/reference/src/.
"""
from __future__ import annotations

import math
import random
from fractions import Fraction

# --------------------------------------------------------------------------------------------
# number theory helpers (Primes.jl / GaloisFields.jl call sites: crt.jl:282-295,
# pow2_cyc_rings.jl:38-44)
# --------------------------------------------------------------------------------------------

_SMALL_PRIMES = (2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37)


def is_prime(n: int) -> bool:
    """Deterministic Miller-Rabin for n < 3.3e24 (bases = first 12 primes); BPSW-free."""
    if n < 2:
        return False
    for p in _SMALL_PRIMES:
        if n % p == 0:
            return n == p
    d, s = n - 1, 0
    while d % 2 == 0:
        d //= 2
        s += 1
    for a in _SMALL_PRIMES:
        x = pow(a, d, n)
        if x in (1, n - 1):
            continue
        for _ in range(s - 1):
            x = x * x % n
            if x == n - 1:
                break
        else:
            return False
    return True


def nextprime(n: int, i: int = 1, interval: int = 1) -> int:
    """Primes.nextprime(n, i; interval): i-th prime p >= n with p ≡ n (mod interval).

    Call sites: crt.jl:287, test/bfv_crt.jl:9-20, test/ckks_*.jl.
    """
    p = n
    found = 0
    while True:
        if is_prime(p):
            found += 1
            if found == i:
                return p
        p += interval


def minimal_primitive_root(q: int, n: int) -> int:
    """GaloisFields.minimal_primitive_root(𝔽q, n): the numerically smallest element of exact
    multiplicative order n (n a power of two here).  Known answer: (97, 8) -> 33
    (docs/src/man/background/rlwe.md:186).  Call sites: pow2_cyc_rings.jl:40, crt.jl:142-144."""
    assert (q - 1) % n == 0 and n >= 2 and n & (n - 1) == 0
    # find any element of order exactly n
    g = 2
    while True:
        z = pow(g, (q - 1) // n, q)
        if pow(z, n // 2, q) == q - 1:
            break
        g += 1
    # all primitive n-th roots are the odd powers of z
    best = z
    z2 = z * z % q
    cur = z
    for _ in range(n // 2 - 1):
        cur = cur * z2 % q
        if cur < best:
            best = cur
    return best


def rns_ring_primes(N: int, logqs) -> list:
    """NegacyclicRing(N, logqs) prime search, crt.jl:282-291."""
    perm = sorted(range(len(logqs)), key=lambda i: logqs[i])
    primes = [0] * len(logqs)
    lastp = 0
    for i in perm:
        p = nextprime(max(2 ** logqs[i] + 1, lastp + 2 * N), 1, 2 * N)
        lastp = p
        primes[i] = p
    return primes


def prime_chain(start: int, count: int, N: int) -> list:
    """The chain used by test/bfv_crt.jl:9-20 and test/ckks_*.jl: first prime ≡ 1 mod 2N at or
    above ``start`` (start ≡ 1 mod 2N), then nextprime(p + 2N; interval=2N) repeatedly."""
    out = []
    p = nextprime(start, 1, 2 * N)
    for _ in range(count):
        out.append(p)
        p = nextprime(p + 2 * N, 1, 2 * N)
    return out


# --------------------------------------------------------------------------------------------
# SignedMod / rounding (signedmod.jl:8-36, div_hacks.jl:120-135)
# --------------------------------------------------------------------------------------------

def centred(n: int, m: int) -> int:
    """convert(Integer, ::SignedMod), signedmod.jl:12-19: n > m÷2 ? n - m : n."""
    return n - m if n > m // 2 else n


def div_ties_away(x: int, y: int) -> int:
    """div(x, y, RoundNearestTiesAway) for y > 0, div_hacks.jl:120-135 (exact rational)."""
    assert y > 0
    fr = Fraction(x, y)
    fl = math.floor(fr)
    rem = fr - fl
    if rem > Fraction(1, 2):
        return fl + 1
    if rem < Fraction(1, 2):
        return fl
    return fl + 1 if x >= 0 else fl  # tie: away from zero


# --------------------------------------------------------------------------------------------
# single-modulus negacyclic ring (pow2_cyc_rings.jl)
# --------------------------------------------------------------------------------------------

def nntt_def(a, q, psi):
    """nntt by definition, pow2_cyc_rings.jl:278-303: â[k] = Σ_i a[i] ψ^i ω^{ik}, ω = ψ²,
    i.e. â[k] = a(ψ^{2k+1}); natural order in and out.  O(N²)."""
    N = len(a)
    return [sum(a[i] * pow(psi, i * (2 * k + 1), q) for i in range(N)) % q for k in range(N)]


def inntt_def(ahat, q, psi):
    """inntt by definition, pow2_cyc_rings.jl:308-318: c[i] = N⁻¹ ψ^{-i} Σ_k â[k] ω^{-ik}."""
    N = len(ahat)
    ninv = pow(N, -1, q)
    pinv = pow(psi, -1, q)
    return [ninv * sum(ahat[k] * pow(pinv, i * (2 * k + 1), q) for k in range(N)) % q
            for i in range(N)]


def _bitrev(x, bits):
    r = 0
    for _ in range(bits):
        r = (r << 1) | (x & 1)
        x >>= 1
    return r


def nntt(a, q, psi):
    """Same function as nntt_def in O(N log N): ψ-twist then radix-2 Cooley-Tukey, the structure
    of pow2_cyc_rings.jl:295-303 (powmulp at :298, CTPlan at :301).  Checked against nntt_def."""
    N = len(a)
    lg = N.bit_length() - 1
    x = [0] * N
    pw = 1
    for i in range(N):
        x[_bitrev(i, lg)] = a[i] * pw % q
        pw = pw * psi % q
    w = psi * psi % q
    m = 2
    while m <= N:
        wm = pow(w, N // m, q)
        for s in range(0, N, m):
            t = 1
            for j in range(m // 2):
                u = x[s + j]
                v = x[s + j + m // 2] * t % q
                x[s + j] = (u + v) % q
                x[s + j + m // 2] = (u - v) % q
                t = t * wm % q
        m *= 2
    return x


def inntt(ahat, q, psi):
    """O(N log N) inverse, structure of pow2_cyc_rings.jl:308-318."""
    N = len(ahat)
    lg = N.bit_length() - 1
    x = [0] * N
    for i in range(N):
        x[_bitrev(i, lg)] = ahat[i]
    winv = pow(psi * psi % q, -1, q)
    m = 2
    while m <= N:
        wm = pow(winv, N // m, q)
        for s in range(0, N, m):
            t = 1
            for j in range(m // 2):
                u = x[s + j]
                v = x[s + j + m // 2] * t % q
                x[s + j] = (u + v) % q
                x[s + j + m // 2] = (u - v) % q
                t = t * wm % q
        m *= 2
    ninv = pow(N, -1, q)
    pinv = pow(psi, -1, q)
    pw = ninv
    out = [0] * N
    for i in range(N):
        out[i] = x[i] * pw % q
        pw = pw * pinv % q
    return out


def negacyclic_mul_naive(a, b, q):
    """ring_multiply with ψ == 0, pow2_cyc_rings.jl:150-165 (schoolbook, x^N = -1)."""
    N = len(a)
    res = [0] * N
    for i in range(N):
        for j in range(N):
            idx = i + j
            if idx < N:
                res[idx] = (res[idx] + a[i] * b[j]) % q
            else:
                res[idx - N] = (res[idx - N] - a[i] * b[j]) % q
    return res


def apply_galois_element(a, g, q):
    """pow2_cyc_rings.jl:321-329: out[(g*i) mod N] = (-1)^{floor(g*i/N)} a[i] (coefficient domain)."""
    N = len(a)
    out = [0] * N
    for i in range(N):
        qq, r = divmod(g * i, N)
        out[r] = (-a[i]) % q if qq % 2 == 1 else a[i]
    return out


def galois_element_for_steps(steps: int, N: int) -> int:
    """rlwe_she.jl:304."""
    return pow(3, 2 * N - steps, 2 * N) if steps > 0 else pow(3, -steps, 2 * N)


# --------------------------------------------------------------------------------------------
# RNS ring: a polynomial is a list of L limb lists ("SoA", crt.jl:150-156); a ring is
# (N, qs, psis)
# --------------------------------------------------------------------------------------------

class Ring:
    """NegacyclicRing{CRTEncoded{L,...},N} (pow2_cyc_rings.jl:27-65, crt.jl:282-295)."""

    def __init__(self, N, qs, psis=None):
        self.N = N
        self.qs = list(qs)
        if psis is None:
            psis = [minimal_primitive_root(q, 2 * N) for q in self.qs]
        self.psis = list(psis)
        for q, p in zip(self.qs, self.psis):
            assert pow(p, 2 * N, q) == 1  # pow2_cyc_rings.jl:31
        self.Q = math.prod(self.qs)

    @property
    def L(self):
        return len(self.qs)

    def select(self, which):
        """crtselect(ring, which), crt.jl:189-192."""
        return Ring(self.N, [self.qs[i] for i in which], [self.psis[i] for i in which])

    def drop_last(self):
        """crt.jl:213."""
        return self.select(range(self.L - 1))

    def __eq__(self, o):
        return isinstance(o, Ring) and (self.N, self.qs, self.psis) == (o.N, o.qs, o.psis)


def rns_from_int(x: int, qs):
    """CRTEncoded{N,M}(x::Integer), crt.jl:91-95."""
    return [x % q for q in qs]


def rns_to_int(res, qs) -> int:
    """convert(Integer, ::CRTEncoded), crt.jl:98-112: exact CRT into [0, Q)."""
    Q = math.prod(qs)
    x = 0
    for r, q in zip(res, qs):
        Qi = Q // q
        x += r * Qi * pow(Qi, -1, q)
    return x % Q


def poly_from_ints(coeffs, ring: Ring):
    return [[c % q for c in coeffs] for q in ring.qs]


def poly_to_ints(p, ring: Ring):
    return [rns_to_int([p[l][i] for l in range(ring.L)], ring.qs) for i in range(ring.N)]


def poly_nntt(p, ring: Ring):
    """crt.jl:247-256: per-limb nntt."""
    return [nntt(p[l], ring.qs[l], ring.psis[l]) for l in range(ring.L)]


def poly_inntt(p, ring: Ring):
    """crt.jl:258-267."""
    return [inntt(p[l], ring.qs[l], ring.psis[l]) for l in range(ring.L)]


def poly_add(a, b, ring):
    """crt.jl:120-122 limb-wise."""
    return [[(x + y) % q for x, y in zip(a[l], b[l])] for l, q in enumerate(ring.qs)]


def poly_sub(a, b, ring):
    """crt.jl:132-134."""
    return [[(x - y) % q for x, y in zip(a[l], b[l])] for l, q in enumerate(ring.qs)]


def poly_neg(a, ring):
    """crt.jl:128-130."""
    return [[(-x) % q for x in a[l]] for l, q in enumerate(ring.qs)]


def poly_pointwise(a, b, ring):
    """crt.jl:124-126 (used on duals by pow2_cyc_rings.jl:167)."""
    return [[x * y % q for x, y in zip(a[l], b[l])] for l, q in enumerate(ring.qs)]


def poly_scalar_mul(s: int, a, ring):
    """scalar_mul, pow2_cyc_rings.jl:177-180 with an Integer scalar promoted by crt.jl:87-95."""
    return [[(s % q) * x % q for x in a[l]] for l, q in enumerate(ring.qs)]


def poly_mul(a, b, ring):
    """RingElement * RingElement on coefficient-domain inputs, coefficient-domain output
    (pow2_cyc_rings.jl:147-173 then coeffs_primal :124-130)."""
    return poly_inntt(poly_pointwise(poly_nntt(a, ring), poly_nntt(b, ring), ring), ring)


def poly_galois(a, g, ring):
    return [apply_galois_element(a[l], g, q) for l, q in enumerate(ring.qs)]


def poly_zero(ring):
    return [[0] * ring.N for _ in ring.qs]


# --------------------------------------------------------------------------------------------
# level operations (crt.jl:215-236)
# --------------------------------------------------------------------------------------------

def modswitch_poly(a, ring: Ring):
    """modswitch(::RingElement), crt.jl:226-228 with modswitch(::CRTEncoded) :215-220:
    c'_j = q_last^{-1} (c_j - [c_last]) mod q_j, c_last taken as its unsigned representative
    (convert(Integer, ·) == x.n, utils.jl:39).  Coefficient domain in and out."""
    L = ring.L
    ql = ring.qs[-1]
    out = []
    for j in range(L - 1):
        qj = ring.qs[j]
        inv = pow(ql % qj, -1, qj)
        out.append([inv * ((a[j][i] - a[L - 1][i] % qj) % qj) % qj for i in range(ring.N)])
    return out


def modswitch_drop_poly(a, ring: Ring):
    """crt.jl:222-232."""
    return [list(x) for x in a[:-1]]


# --------------------------------------------------------------------------------------------
# BFV expand / contract (bfv.jl:34-40,172-226)
# --------------------------------------------------------------------------------------------

def switchel(x: int, q: int, T: int) -> int:
    """bfv.jl:202-220 on the integer representative en ∈ [0, q); returns an integer that the
    caller reduces into the target basis (the `T(...)` constructor, crt.jl:91-95)."""
    halfq = q >> 1
    diff = T - q if T > q else q - T
    if q < T:
        return x + diff if x > halfq else x
    return x - diff if x > halfq else x


def switch_poly(a, src: Ring, dst: Ring):
    """switch(ℛ, e), bfv.jl:222-226."""
    ints = poly_to_ints(a, src)
    return poly_from_ints([switchel(x, src.Q, dst.Q) for x in ints], dst)


def multround_poly(a, big: Ring, t: int, q: int):
    """multround(e, a, b), bfv.jl:172-190: per coefficient
    SignedMod(x)*t  (modular product in ℛbig, signedmod.jl:24-28)
    → div(centred, q, RoundNearestTiesAway) (signedmod.jl:30-32) → back into ℛbig."""
    ints = poly_to_ints(a, big)
    out = []
    for x in ints:
        v = centred(x * t % big.Q, big.Q)
        out.append(div_ties_away(v, q) % big.Q)
    return poly_from_ints(out, big)


def bfv_mul_expand(ct, ring: Ring, big: Ring):
    """mul_expand, bfv.jl:34."""
    return [switch_poly(c, ring, big) for c in ct]


def bfv_mul_contract(cs, ring: Ring, big: Ring, t: int):
    """mul_contract, bfv.jl:35-40."""
    return [switch_poly(multround_poly(c, big, t, ring.Q), big, ring) for c in cs]


def tensor(c1, c2, ring: Ring):
    """enc_mul body, rlwe_she.jl:255-258, coefficient-domain in/out."""
    n = len(c1) + len(c2) - 1
    out = [poly_zero(ring) for _ in range(n)]
    for i in range(len(c1)):
        for j in range(len(c2)):
            out[i + j] = poly_add(out[i + j], poly_mul(c1[i], c2[j], ring), ring)
    return out


def bfv_enc_mul(c1, c2, ring: Ring, big: Ring, t: int):
    """enc_mul for BFVParams, rlwe_she.jl:247-262 + bfv.jl:34-40."""
    e1 = bfv_mul_expand(c1, ring, big)
    e2 = bfv_mul_expand(c2, ring, big)
    return bfv_mul_contract(tensor(e1, e2, big), ring, big, t)


# --------------------------------------------------------------------------------------------
# key switching (rlwe_she.jl:273-349, modulusraising.jl)
# --------------------------------------------------------------------------------------------

def rns_digits(c_end, ring: Ring, target: Ring):
    """rlwe_she.jl:326-329: digit i = centred residue of limb i (SignedMod, signedmod.jl:12-19)
    re-reduced into every limb of ``target`` (the ring of c1 after keyswitch_expand)."""
    ps = []
    for l, q in enumerate(ring.qs):
        ps.append(poly_from_ints([centred(x, q) for x in c_end[l]], target))
    return ps


def window_digits(c_end, ring: Ring, w: int, target: Ring = None):
    """rlwe_she.jl:330-338: base-2^w digits of the unsigned integer representative of c[end] in ℛ = ring(c.cs[1]) (:333:
    nwindows = ndigits(modulus(coefftype(ℛ))), the CIPHERTEXT ring), each digit polynomial an element of typeof(c1) (:336):
    with ModulusRaised that is the expanded ring [q_1..q_l, P] (modulusraising.jl:35-41) -- ``target``."""
    nwin = ndigits(ring.Q, 2 ** w)
    ints = poly_to_ints(c_end, ring)
    ps = []
    for i in range(nwin):
        ps.append(poly_from_ints([(x >> (i * w)) & ((1 << w) - 1) for x in ints], target or ring))
    return ps


def ndigits(x: int, base: int) -> int:
    n = 0
    while x > 0:
        x //= base
        n += 1
    return max(n, 1)


def keyswitch_expand_modraise(c, cring: Ring, keyring: Ring):
    """keyswitch_expand for ModulusRaised, modulusraising.jl:35-41 with CRTExpand crt.jl:38-40:
    multiply by the special prime P (last key modulus) limb-wise and append a zero limb.  The
    result lives in limbs [1..l, L+1] of the key ring."""
    P = keyring.qs[-1]
    l = cring.L
    exp_ring = keyring.select(list(range(l)) + [keyring.L - 1])
    out = [[(P % q) * x % q for x in c[j]] for j, q in enumerate(cring.qs)]
    out.append([0] * cring.N)
    return out, exp_ring


def keyswitch(evk, ct, cring: Ring, keyring: Ring, special: bool, relin_window: int = 0):
    """keyswitch(ek, c), rlwe_she.jl:315-347; all polys coefficient domain in and out.

    evk: list of (mask, masked) pairs over ``keyring`` (coefficient domain), rlwe_she.jl:297.
    special=True is ModulusRaised (modulusraising.jl:35-49); then keyring = cring's limbs plus
    extra levels plus the special prime last, and key limbs [1..l, L+1] are used (:43-49).
    """
    assert len(ct) in (2, 3)  # rlwe_she.jl:318
    l = cring.L
    if special:
        which = list(range(l)) + [keyring.L - 1]
        c1, wring = keyswitch_expand_modraise(ct[0], cring, keyring)
        c2 = poly_zero(wring) if len(ct) == 2 else keyswitch_expand_modraise(ct[1], cring, keyring)[0]
    else:
        which = list(range(l))
        wring = cring
        c1 = [list(x) for x in ct[0]]
        c2 = poly_zero(wring) if len(ct) == 2 else [list(x) for x in ct[1]]
    if relin_window == 0:
        ps = rns_digits(ct[-1], cring, wring)
    else:
        ps = window_digits(ct[-1], cring, relin_window, wring)   # the first len(ps) key components are used (rlwe_she.jl:340)
    for i, p in enumerate(ps):
        mask = [evk[i][0][j] for j in which]      # downswitch_keyelement, crt.jl:238-244 /
        masked = [evk[i][1][j] for j in which]    # modulusraising.jl:43-49
        c2 = poly_add(c2, poly_mul(mask, p, wring), wring)     # rlwe_she.jl:342
        c1 = poly_add(c1, poly_mul(masked, p, wring), wring)   # rlwe_she.jl:343
    if special:
        return [modswitch_poly(c1, wring), modswitch_poly(c2, wring)]  # modulusraising.jl:42
    return [c1, c2]


# --------------------------------------------------------------------------------------------
# host-side producers used to make *valid* inputs (rlwe_she.jl:155-217, 273-310).  Randomness
# cannot match Julia's; parity is defined on the deterministic ops given inputs.
# --------------------------------------------------------------------------------------------

def sample_uniform(rng: random.Random, ring: Ring):
    """RingSampler(ℛ, DiscreteUniform(coefftype)), crt.jl:146-148: limb-wise independent."""
    return [[rng.randrange(q) for _ in range(ring.N)] for q in ring.qs]


def sample_gauss_ints(rng: random.Random, N: int, sigma: float):
    """DiscreteNormal(0, σ) stand-in: rounded Gaussian (the fork of Distributions is not on disk,
    Manifest.toml:182-188)."""
    return [int(round(rng.gauss(0.0, sigma))) for _ in range(N)]


def keygen(rng, ring: Ring, sigma: float, noise_scale: int = 1):
    """keygen, rlwe_she.jl:155-167. noise_scale = plaintext modulus for BGV (bgv.jl:27-34)."""
    mask = sample_uniform(rng, ring)
    secret = poly_from_ints(sample_gauss_ints(rng, ring.N, sigma), ring)
    err = poly_from_ints([noise_scale * e for e in sample_gauss_ints(rng, ring.N, sigma)], ring)
    masked = poly_neg(poly_add(poly_mul(mask, secret, ring), err, ring), ring)
    return secret, (mask, masked)


def encrypt_zero(rng, pub, ring: Ring, sigma: float, noise_scale: int = 1):
    """encrypt(rng, pk, ::Zero), rlwe_she.jl:176-186."""
    mask, masked = pub
    u = poly_from_ints(sample_gauss_ints(rng, ring.N, sigma), ring)
    e1 = poly_from_ints([noise_scale * e for e in sample_gauss_ints(rng, ring.N, sigma)], ring)
    e2 = poly_from_ints([noise_scale * e for e in sample_gauss_ints(rng, ring.N, sigma)], ring)
    return [poly_add(poly_mul(masked, u, ring), e1, ring),
            poly_add(poly_mul(mask, u, ring), e2, ring)]


def decrypt_raw(secret, ct, ring: Ring):
    """decrypt body, rlwe_she.jl:206-212: b = c1 + s c2 + s² c3 ..."""
    b = ct[0]
    spow = secret
    for i in range(1, len(ct)):
        b = poly_add(b, poly_mul(spow, ct[i], ring), ring)
        spow = poly_mul(spow, secret, ring)
    return b


def make_eval_key(rng, old, secret, ring: Ring, sigma: float, relin_window: int = 0,
                  noise_scale: int = 1, premul: int = 1):
    """make_eval_key, rlwe_she.jl:273-298 (premul = special prime for ModulusRaised,
    modulusraising.jl:28-32).  Returns [(mask_i, masked_i)] over ``ring``."""
    old = poly_scalar_mul(premul, old, ring)
    if relin_window != 0:
        nwin = ndigits(ring.Q, 2 ** relin_window)
        evala = [poly_scalar_mul(pow(2, i * relin_window), old, ring) for i in range(nwin)]
    else:
        # CRTResidual gadget, rlwe_she.jl:287 + crt.jl:64-77: keep limb i, zero the others
        evala = []
        for i in range(ring.L):
            g = poly_zero(ring)
            g[i] = list(old[i])
            evala.append(g)
    key = []
    for a in evala:
        mask = sample_uniform(rng, ring)
        e = poly_from_ints([noise_scale * x for x in sample_gauss_ints(rng, ring.N, sigma)], ring)
        masked = poly_sub(a, poly_add(poly_mul(mask, secret, ring), e, ring), ring)
        key.append((mask, masked))
    return key


# --------------------------------------------------------------------------------------------
# scheme plaintext maps
# --------------------------------------------------------------------------------------------

def bfv_encode(plain, ring: Ring, t: int):
    """π⁻¹ for BFV, bfv.jl:21-24: Δ * plaintext, Δ = q ÷ t."""
    delta = ring.Q // t
    return poly_from_ints([delta * (m % t) for m in plain], ring)


def bfv_decode(b, ring: Ring, t: int):
    """π for BFV, bfv.jl:26-29: mod(SignedMod(divround(x, Δ)), t)."""
    delta = ring.Q // t
    out = []
    for x in poly_to_ints(b, ring):
        y = div_ties_away(centred(x, ring.Q), delta) % ring.Q   # divround → SignedMod → .x
        out.append(centred(y, ring.Q) % t)
    return out


def bgv_decode(b, ring: Ring, t: int):
    """π for BGV, bgv.jl:21-25: mod(SignedMod(x), t)."""
    return [centred(x, ring.Q) % t for x in poly_to_ints(b, ring)]


# --------------------------------------------------------------------------------------------
# CKKS encode / decode (float; ckksencoding.jl:43-97, ckks.jl:35-59)
# --------------------------------------------------------------------------------------------

def zmstar_row1(M: int, ncols: int):
    """ℤmstarPermutation(M)[1, 1:ncols], ckksencoding.jl:49-54: 3^col mod M."""
    return [pow(3, c, M) for c in range(1, ncols + 1)]


def ckks_encode(slots, ring: Ring, scale):
    """convert(RingElement, ::CKKSEncoding), ckksencoding.jl:72-97."""
    import numpy as np
    n2 = len(slots)
    N = 2 * n2
    assert N == ring.N
    M = 4 * n2
    cm = np.zeros(N, dtype=np.complex128)
    for i in range(n2):
        e = pow(3, i + 1, M)
        cm[e >> 1] = slots[i]
        cm[(M - e) % M >> 1] = np.conj(slots[i])
    ip = np.fft.ifft(cm)
    tw = np.array([complex(np.exp(1j * np.float64(2 * k / (2 * N)) * np.pi)) for k in range(N)])
    real = (ip * tw).real
    out = []
    for x in real:
        n = int((Fraction(float(x)) * Fraction(scale)).__round__())  # round(BigInt, big(x)*den)
        out.append(n)
    return poly_from_ints(out, ring)


def ckks_decode(p, ring: Ring, scale):
    """CKKSEncoding{ScaleT}(plain), ckksencoding.jl:56-66."""
    import numpy as np
    N = ring.N
    vals = np.array([float(Fraction(centred(x, ring.Q)) / Fraction(scale)) for x in poly_to_ints(p, ring)])
    tw = np.array([complex(np.exp(-1j * np.float64(2 * k / (2 * N)) * np.pi)) for k in range(N)])
    f = np.fft.fft(vals * tw)
    idx = [e >> 1 for e in zmstar_row1(2 * N, N // 2)]
    return f[idx]


# --------------------------------------------------------------------------------------------
# Samplers (poly.jl:7-23, crt.jl:277-279).  The reference draws from Julia's generator, which cannot be
# reproduced here: PARITY UNPINNED for random streams.  What is pinned is the stream this build
# DEFINES for its device samplers -- Philox4x32-10 (Salmon et al., SC'11; known-answer vectors from the
# Random123 distribution in tests/test_oracle_known_answers.py) -- restated here for the GPU tests.
# --------------------------------------------------------------------------------------------

def philox4x32_10(c, k):
    """counter c = 4 words, key k = 2 words (32-bit) -> 4 words"""
    c0, c1, c2, c3 = c
    k0, k1 = k
    M = 0xFFFFFFFF
    for _ in range(10):
        p0, p1 = 0xD2511F53 * c0, 0xCD9E8D57 * c2
        c0, c1, c2, c3 = (p1 >> 32) ^ c1 ^ k0, p1 & M, (p0 >> 32) ^ c3 ^ k1, p0 & M
        k0, k1 = (k0 + 0x9E3779B9) & M, (k1 + 0xBB67AE85) & M
    return c0, c1, c2, c3


def sample_uniform_mod(idx: int, limb: int, stream: int, seed: int, q: int) -> int:
    """csrc/sample_kernels.h sample_uniform_mod: rejection sampling on 64-bit draws"""
    lim = ((2**64 - 1) // q) * q
    a = 0
    while True:
        b = philox4x32_10((idx & 0xFFFFFFFF, idx >> 32, ((a << 8) | limb) & 0xFFFFFFFF, stream), (seed & 0xFFFFFFFF, seed >> 32))
        for r in ((b[1] << 32) | b[0], (b[3] << 32) | b[2]):
            if r < lim:
                return r % q
        a += 1


def sample_gauss_int(idx: int, stream: int, seed: int, sigma: float) -> int:
    """csrc/sample_kernels.h sample_gauss_int: Box-Muller, rint (ties to even)"""
    import math
    b = philox4x32_10((idx & 0xFFFFFFFF, idx >> 32, 0xFFFFFFFF, stream), (seed & 0xFFFFFFFF, seed >> 32))
    r0, r1 = (b[1] << 32) | b[0], (b[3] << 32) | b[2]
    u1, u2 = (float(r0 >> 11) + 1.0) * 2.0**-53, float(r1 >> 11) * 2.0**-53
    z = math.sqrt(-2.0 * math.log(u1)) * math.cos(6.283185307179586476925286766559 * u2)
    x = sigma * z
    return int(round(x)) if abs(x - math.floor(x) - 0.5) > 1e-300 else int(2 * round(x / 2))
