"""Pin the spec oracle (oracle/spec.py) against every known-answer datum the reference holds for
the path: docs/src/man/background/rlwe.md:186,207-212,
docs/src/man/encoding.md:14-38,69-91, src/cryptparams.jl:22-25, and the nntt definition
src/pow2_cyc_rings.jl:278-294."""
import random

import pytest

from oracle import spec


def test_psi_97_8_is_33():
    # rlwe.md:186  NegacyclicRing{𝔽₉₇,4}() prints NegacyclicRing{𝔽₉₇,4}(33)
    assert spec.minimal_primitive_root(97, 8) == 33


def test_rlwe_md_products():
    # rlwe.md:194-212
    q, N = 97, 4
    psi = spec.minimal_primitive_root(q, 2 * N)
    ring = spec.Ring(N, [q], [psi])
    p1, p2, p3, p4 = [1, 1, 0, 0], [0, 0, 0, 1], [4, 0, 0, 0], [5, 0, 0, 0]
    mul = lambda a, b: spec.poly_mul([a], [b], ring)[0]
    assert mul(p3, p4) == [20, 0, 0, 0]
    assert mul(p1, p1) == [1, 2, 1, 0]
    assert mul(p1, p2) == [96, 0, 0, 1]
    # and the naive path agrees
    assert spec.negacyclic_mul_naive(p1, p2, q) == [96, 0, 0, 1]


def test_encoding_md_naive_paths():
    # encoding.md:14-24: 𝔽₇, N=2, ψ=nothing → naive convolution: 3*4 = [5, 0]
    assert spec.negacyclic_mul_naive([3, 0], [4, 0], 7) == [5, 0]
    # encoding.md:29-38: UInt8 (mod 256): 3*15 = 0x2d
    assert spec.negacyclic_mul_naive([3, 0], [15, 0], 256) == [0x2D, 0]


def test_encoding_md_slot_encoding():
    # encoding.md:69-91: 𝔽₆₅₅₃₇, N=2048; slots (=dual) 1:10 times all-10 → 10,20,...,100,0
    q, N = 65537, 2048
    psi = spec.minimal_primitive_root(q, 2 * N)
    a_dual = list(range(1, 11)) + [0] * (N - 10)
    b_dual = [10] * N
    a = spec.inntt(a_dual, q, psi)
    b = spec.inntt(b_dual, q, psi)
    ring = spec.Ring(N, [q], [psi])
    prod = spec.poly_mul([a], [b], ring)[0]
    slots = spec.nntt(prod, q, psi)
    assert slots[:11] == [10, 20, 30, 40, 50, 60, 70, 80, 90, 100, 0]


@pytest.mark.parametrize("q,N,psi", [
    (1099511627873, 8, 108163207722),
    (525313, 512, 513496),
    (34359724033, 1024, 7225104974),
    (1152921504606830593, 2048, 811032584449645127),
])
def test_cryptparams_triples(q, N, psi):
    # cryptparams.jl:22-25: explicit (q, N, ψ); ctor asserts ψ^{2N} == 1 (pow2_cyc_rings.jl:31)
    assert spec.is_prime(q)
    assert pow(psi, 2 * N, q) == 1
    assert pow(psi, N, q) == q - 1  # primitive


@pytest.mark.parametrize("q,N", [(97, 4), (1099511627873, 16), (1099511628161, 32), (65537, 64)])
def test_fast_nntt_matches_definition(q, N):
    rng = random.Random(q + N)
    psi = spec.minimal_primitive_root(q, 2 * N)
    a = [rng.randrange(q) for _ in range(N)]
    assert spec.nntt(a, q, psi) == spec.nntt_def(a, q, psi)
    assert spec.inntt(spec.nntt(a, q, psi), q, psi) == a
    assert spec.inntt_def(spec.nntt_def(a, q, psi), q, psi) == a
    # evaluation form of the definition: â[k] = a(ψ^{2k+1})
    ev = [sum(c * pow(psi, (2 * k + 1) * i, q) for i, c in enumerate(a)) % q for k in range(N)]
    assert ev == spec.nntt(a, q, psi)
    # convolution theorem vs schoolbook
    b = [rng.randrange(q) for _ in range(N)]
    ring = spec.Ring(N, [q], [psi])
    assert spec.poly_mul([a], [b], ring)[0] == spec.negacyclic_mul_naive(a, b, q)


def test_prime_chains_match_survey():
    # test/ckks_rotate.jl:9-10 (N=16), test/ckks_*.jl (N=32), test/bfv_crt.jl:9-20 (N=2048)
    assert spec.prime_chain(2**40 + 1, 2, 16) == [1099511627873, 1099511628161]
    assert spec.prime_chain(2**40 + 1, 3, 32) == [1099511628161, 1099511629121, 1099511629889]
    ch = spec.prime_chain(2**50 + 1, 6, 2048)
    assert ch[0] == 1125899906949121 and ch[1] == 1125899906977793
    assert [spec.minimal_primitive_root(q, 32) for q in (1099511627873, 1099511628161)] == \
        [49985231946, 32623736631]


def test_ties_away_and_centred():
    # div_hacks.jl:120-135, signedmod.jl:12-19
    assert spec.div_ties_away(5, 2) == 3 and spec.div_ties_away(-5, 2) == -3
    assert spec.div_ties_away(7, 3) == 2 and spec.div_ties_away(-7, 3) == -2
    assert spec.div_ties_away(8, 3) == 3 and spec.div_ties_away(-8, 3) == -3
    assert spec.centred(48, 97) == 48 and spec.centred(49, 97) == -48
    assert spec.centred(5, 10) == 5 and spec.centred(6, 10) == -4


def test_crtexpand_docstring():
    # crt.jl:21-33: CRTEncoded{2,(𝔽₅,𝔽₇)}(3) * CRTExpand{𝔽₁₁} == (3,5,0).  The (non-doctest)
    # docstring then prints the integer as "333"; that is inconsistent with its own tuple
    # (333 mod 7 = 4): 3*11 = 33 is the value whose residues are (3,5,0).
    x = spec.rns_from_int(3, [5, 7])
    y = [(11 % q) * r % q for r, q in zip(x, [5, 7])] + [0]
    assert y == [3, 5, 0]
    assert spec.rns_to_int(y, [5, 7, 11]) == 33
    # crt.jl:42-58: CRTResidual(c) = c*[(q/q_i)^{-1}]_{q_i}*q/q_i; the docstring's arithmetic line
    # `mod(3*invmod(77, 5), 5)*77 == 308` is the (5,7,11) basis with c = 𝔽₅(3): residues (3,0,0)
    assert spec.rns_to_int([3, 0, 0], [5, 7, 11]) == 308 == (3 * pow(77, -1, 5) % 5) * 77


def test_philox4x32_10_known_answers():
    """Random123 kat_vectors (philox4x32, 10 rounds): the stream definition of the device samplers."""
    from oracle import spec
    kat = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
           ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
           ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0), (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for c, k, want in kat:
        assert spec.philox4x32_10(c, k) == want
