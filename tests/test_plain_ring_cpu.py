"""The psi = 0 plaintext rings of the host mirror (ring.PlainRing): the reference's naive negacyclic convolution
(pow2_cyc_rings.jl:150-165) against the oracle's definition, and plaintext_space's choice (rlwe_she.jl:380-392).  Host only."""
import random

import pytest

import toyfhe_jl_amd as tf
from oracle import spec


@pytest.mark.parametrize("N,t", [(8, 53), (64, 256), (256, 65537), (128, 2 ** 61 - 1)])
def test_naive_product_matches_the_definition(N, t):
    rng = random.Random(N * 1000 + t % 997)
    R = tf.PlainRing(N, t)
    a, b = [rng.randrange(t) for _ in range(N)], [rng.randrange(t) for _ in range(N)]
    x, y = R(a), R(b)
    assert (x * y).to_ints() == spec.negacyclic_mul_naive(a, b, t)
    assert (x * y) == (y * x)
    assert ((x + y) - y) == x and (-x + x) == R.zero()
    assert (x * 3).to_ints() == [3 * v % t for v in a]
    assert (x ** 3) == x * x * x


def test_wraparound_sign_and_indexing():
    R = tf.PlainRing(4, 53)                       # x^3 * x = x^4 = -1
    x3, x1 = R.zero(), R.zero()
    x3[3] = 1
    x1[1] = 1
    assert (x3 * x1).to_ints() == [52, 0, 0, 0]
    p = R.zero()
    p[0] = 6                                      # test/bfv_crt.jl:39-47
    assert (p * p)[0] == 0x24
    with pytest.raises(AssertionError):
        tf.PlainRing(12, 53)
    with pytest.raises(tf.UsageError):
        R.zero() + tf.PlainRing(8, 53).zero()


def test_plaintext_space_choice():
    assert isinstance(tf.plaintext_space(1024, 53), tf.PlainRing)            # prime below 2N
    assert isinstance(tf.plaintext_space(1024, 256), tf.PlainRing)           # composite
    assert isinstance(tf.plaintext_space(1024, 2053), tf.PlainRing)          # prime above 2N without a 2N-th root of unity
