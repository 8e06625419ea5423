"""Host-side mirror of ToyFHE's generic RLWE scheme and its BFV / BGV / CKKS / ModulusRaised parameter
wrappers, over device-resident ``RingElement``s (see ring.py).

Mirrors ``src/rlwe_she.jl`` (keys, CipherText, keygen/encrypt/decrypt, ``+ - *``, ``keyswitch``,
``rotate``), ``src/bfv.jl`` (π, π⁻¹, mul_expand / mul_contract), ``src/bgv.jl``, ``src/ckks.jl`` +
``src/ckksencoding.jl`` (encode / decode, ct ``modswitch``) and ``src/modulusraising.jl``.  The heavy
operations -- ring products, ``enc_mul`` for BFV, ``keyswitch``, ``rotate``, ``modswitch`` -- run as
single fused calls into libtoyfhe_hip.so; key generation, encryption and decryption are, as in the
reference, thin host-side callers of ring arithmetic.

Randomness cannot match Julia's (MersenneTwister + a forked Distributions); parity is defined on the
deterministic operations given inputs.  Samplers take a ``numpy.random.Generator``.
"""
from __future__ import annotations

import math
import os
from fractions import Fraction

import numpy as np

from . import native
from .native import BfvPlan, DeviceBuffer, UsageError
from .ring import NegacyclicRing, RingElement, plaintext_space as _plaintext_space

DOT_MAX = 64   # operands / rotations one device pass of tfhe_dot, tfhe_lincomb[_many], tfhe_matmul_diag takes (TFHE_DOT_MAX, csrc/kernels.h)

# --------------------------------------------------------------------------------------------------
# samplers (poly.jl:7-23, crt.jl:146-148,277-279, bgv.jl:27-34)
# --------------------------------------------------------------------------------------------------


class DeviceRng:
    """Counter-based generator for the device samplers (tfhe_sample_uniform / tfhe_sample_gaussian): Philox4x32-10 keyed by
    the 64-bit `seed`.  Every draw of a polynomial takes a fresh value of the polynomial counter, which is its own counter
    word (separate from the coefficient index), so no two draws -- for rings of any degree -- ever share a counter; uniform
    and Gaussian draws additionally use different stream ids (0 / 1).  Passing a DeviceRng where the mirror takes an `rng`
    makes keygen / encrypt sample on the GPU (no host round trip).

    This is a reproducible statistical generator for tests and benchmarks, NOT a cryptographically secure one (64-bit key,
    Philox is not a CSPRNG): production keys must come from a CSPRNG on the host (`numpy.random.Generator` over
    `os.urandom`-seeded state is accepted wherever an `rng` is)."""

    def __init__(self, seed: int):
        self.seed, self.next_poly = int(seed) & (2**64 - 1), 0

    def take(self, n: int) -> int:
        first, self.next_poly = self.next_poly, self.next_poly + n
        return first


def _device_sample(rng: DeviceRng, ring: NegacyclicRing, batch, gaussian=None):
    if ring.idx != list(range(ring.L)):
        raise UsageError("device sampling: the ring must be a prefix of its context")
    n = 1 if batch is None else int(batch)
    out = DeviceBuffer(n * ring.L * ring.N)
    first = rng.take(n)
    if gaussian is None:
        ring.ctx.sample_uniform(ring.L, rng.seed, 0, first, out.ptr, n)
    else:
        sigma, mult = gaussian
        ring.ctx.sample_gaussian(ring.L, sigma, mult, rng.seed, 1, first, out.ptr, n)
    return RingElement(ring, out, None, batch)


def sample_uniform(rng, ring: NegacyclicRing, batch=None) -> RingElement:
    """RingSampler(ℛ, DiscreteUniform(coefftype)), poly.jl:7-23 / crt.jl:146-148,277-279: limb-wise independent."""
    if isinstance(rng, DeviceRng):
        return _device_sample(rng, ring, batch)
    shape = (ring.N,) if batch is None else (batch, ring.N)
    cols = [rng.integers(0, q, size=shape, dtype=np.uint64) for q in ring.moduli]
    return RingElement.from_host(ring, np.stack(cols, axis=len(shape) - 1))


def sample_noise(rng, ring: NegacyclicRing, sigma: float, batch=None, scale: int = 1) -> RingElement:
    """scale * round(N(0, sigma^2)) coefficients (𝒩 / 𝒢 of the schemes; rounded Gaussian as in test/bfv_crt.jl:34)."""
    if isinstance(rng, DeviceRng):
        return _device_sample(rng, ring, batch, gaussian=(sigma, scale))
    return lift_ints(ring, sample_normal_ints(rng, ring.N, sigma, batch), scale)


def sample_normal_ints(rng: np.random.Generator, N: int, sigma: float, batch=None):
    shape = (N,) if batch is None else (batch, N)
    return np.rint(rng.normal(0.0, sigma, size=shape)).astype(np.int64)


def lift_ints(ring: NegacyclicRing, ints: np.ndarray, scale: int = 1) -> RingElement:
    """small signed integers -> RNS residues (CRTEncoded(x::Integer), crt.jl:91-95)"""
    cols = [np.mod(ints.astype(object) * scale, q).astype(np.uint64) for q in ring.moduli]
    return RingElement.from_host(ring, np.stack(cols, axis=ints.ndim - 1))


# --------------------------------------------------------------------------------------------------
# scheme parameters
# --------------------------------------------------------------------------------------------------


class SHESchemeParams:
    relin_window = 0

    def R_cipher(self) -> NegacyclicRing:
        raise NotImplementedError

    def R_key(self) -> NegacyclicRing:      # ℛ_key, rlwe_she.jl:26
        return self.R_cipher()

    def noise(self, rng, ring, batch=None):  # 𝒩
        return sample_noise(rng, ring, self.sigma, batch)

    def secret_dist(self, rng, ring, batch=None):  # 𝒢
        return sample_noise(rng, ring, self.sigma, batch)

    def mul_expand(self, c):                 # rlwe_she.jl:39
        return None

    def encode(self, plaintext):             # π⁻¹
        raise NotImplementedError

    def decode(self, b):                     # π
        raise NotImplementedError


class BFVParams(SHESchemeParams):
    """BFVParams(ℛ, ℛbig, ℛplain, relin_window, σ, Δ), bfv.jl:5-19 (ℛplain given by its modulus t)."""

    scheme_name = "BFV"

    def __init__(self, ring: NegacyclicRing, ring_big: NegacyclicRing, t: int, relin_window: int = 0, sigma: float = 3.2):
        self.ring, self.ring_big, self.t, self.relin_window, self.sigma = ring, ring_big, int(t), relin_window, sigma
        self.delta = ring.modulus() // self.t            # test/bfv_crt.jl:35
        self._plan = None

    def R_cipher(self):
        return self.ring

    def plan(self) -> BfvPlan:
        if self._plan is None:
            self._plan = BfvPlan(self.ring.ctx, self.ring_big.ctx, self.t, self.ring.idx, self.ring_big.idx)
        return self._plan

    def mul_expand(self, c: RingElement) -> RingElement:
        """mul_expand for one component, bfv.jl:34: switch(ℛbig, c) (:202-226) -- the centred lift q -> Qbig, exact."""
        n = c.count
        out = DeviceBuffer(n * self.ring_big.L * self.ring_big.N)
        self.plan().expand(c.coeffs_primal().ptr, out.ptr, n)
        return RingElement(self.ring_big, out, None, c.batch)

    def mul_contract(self, e: RingElement) -> RingElement:
        """mul_contract for one component, bfv.jl:35-40: switch(ℛ, multround(e, t, q)) -- round-ties-away(t e / q), exact."""
        n = e.count
        out = DeviceBuffer(n * self.ring.L * self.ring.N)
        self.plan().contract(e.coeffs_primal().ptr, out.ptr, n)
        return RingElement(self.ring, out, None, e.batch)

    def plaintext_space(self):
        """ℛ_plain (bfv.jl:18, rlwe_she.jl:380-392): the psi = 0 host ring unless t is a prime with a 2N-th root."""
        return _plaintext_space(self.ring, self.t)

    def encode(self, plain) -> RingElement:
        """π⁻¹, bfv.jl:21-24: Δ * plaintext (a list of coefficients, or a list of such lists = a batch)."""
        return self.ring(_map_plain(plain, lambda m: self.delta * (int(m) % self.t)))

    def decode(self, b: RingElement):
        """π, bfv.jl:26-29: mod(SignedMod(divround(x, Δ)), t)."""
        Q = self.ring.modulus()

        def one(x):
            v = x - Q if x > Q // 2 else x
            y = _div_ties_away(v, self.delta) % Q
            y = y - Q if y > Q // 2 else y
            return y % self.t
        return _map_plain(b.to_ints(), one)


class BGVParams(SHESchemeParams):
    """BGVParams(ℛ, ℛplain, σ), bgv.jl:5-34 (no relin_window field: keyswitch is unsupported, as upstream)."""

    scheme_name = "BGV"

    def __init__(self, ring: NegacyclicRing, t: int, sigma: float = 8 / math.sqrt(2 * math.pi)):
        self.ring, self.t, self.sigma = ring, int(t), sigma

    def R_cipher(self):
        return self.ring

    def noise(self, rng, ring, batch=None):  # ShiftedDiscreteNormal, bgv.jl:27-34
        return sample_noise(rng, ring, self.sigma, batch, self.t)

    def plaintext_space(self):
        """ℛ_plain (bgv.jl:18)."""
        return _plaintext_space(self.ring, self.t)

    def encode(self, plain):
        return self.ring(_map_plain(plain, lambda m: int(m) % self.t))

    def decode(self, b):
        Q = b.ring.modulus()                 # the ciphertext's own level (a ModulusRaised / modswitched ring is a sub-basis)
        return _map_plain(b.to_ints(), lambda x: (x - Q if x > Q // 2 else x) % self.t)


class CKKSParams(SHESchemeParams):
    """CKKSParams(ℛ, relin_window, σ), ckks.jl:7-25."""

    scheme_name = "CKKS"

    def __init__(self, ring: NegacyclicRing, relin_window: int = 0, sigma: float = 8 / math.sqrt(2 * math.pi)):
        self.ring, self.relin_window, self.sigma = ring, relin_window, sigma

    def R_cipher(self):
        return self.ring

    def encode(self, plain):
        return plain  # π⁻¹(params, plaintext) = ℛ(plaintext), ckks.jl:21

    def decode(self, b):
        return b      # π(params, b) = b, ckks.jl:22


class ModulusRaised(SHESchemeParams):
    """ModulusRaised{P}, modulusraising.jl:12-21: the last CRT prime is a special prime reserved for keys."""

    def __init__(self, params: SHESchemeParams):
        self.params = params
        self.sigma = params.sigma
        self.relin_window = params.relin_window
        self.scheme_name = params.scheme_name + " (with special prime)"

    def R_cipher(self):
        return self.params.R_cipher().drop_last()

    def R_key(self):
        return self.params.R_key()

    def noise(self, rng, ring, batch=None):
        return self.params.noise(rng, ring, batch)

    def secret_dist(self, rng, ring, batch=None):
        return self.params.secret_dist(rng, ring, batch)

    def encode(self, plain):
        return self.params.encode(plain)

    def decode(self, b):
        return self.params.decode(b)


def _map_plain(plain, f):
    if len(plain) and hasattr(plain[0], "__len__"):
        return [[f(m) for m in row] for row in plain]
    return [f(m) for m in plain]


def _div_ties_away(x: int, y: int) -> int:
    """div(x, y, RoundNearestTiesAway), div_hacks.jl:120-135 (y > 0)."""
    q, r = divmod(abs(x), y)
    if 2 * r >= y:
        q += 1
    return q if x >= 0 else -q


# --------------------------------------------------------------------------------------------------
# keys and ciphertexts (rlwe_she.jl:67-149)
# --------------------------------------------------------------------------------------------------


class KeyComponent:
    def __init__(self, mask: RingElement, masked: RingElement):
        self.mask, self.masked = mask, masked


class PrivKey:
    def __init__(self, params, secret: RingElement):
        self.params, self.secret = params, secret


class PubKey:
    def __init__(self, params, key: KeyComponent):
        self.params, self.key = params, key


class KeyPair:
    def __init__(self, priv: PrivKey, pub: PubKey):
        self.priv, self.pub = priv, pub


class KeySwitchKey:
    """KeySwitchKey{P}(params, key::Vector{KeyComponent}), rlwe_she.jl:87-91.  ``packed()`` is the device
    layout the C ABI consumes: [n_digits][2 = mask, masked][Lk][N], NTT domain."""

    def __init__(self, params, key):
        self.params, self.key = params, list(key)
        self._packed = None

    def packed(self) -> DeviceBuffer:
        if self._packed is None:
            ring = self.key[0].mask.ring
            sz = ring.L * ring.N
            buf = DeviceBuffer(len(self.key) * 2 * sz)
            for i, kc in enumerate(self.key):
                for s, el in enumerate((kc.mask, kc.masked)):
                    native.check(native.lib().tfhe_memcpy_d2d(ring.ctx.h, buf.ptr + ((i * 2 + s) * sz) * 8,
                                                              el.coeffs_dual().ptr, sz * 8))   # once per key
            ring.ctx.sync()
            self._packed = buf
        return self._packed


class EvalMultKey:
    def __init__(self, key: KeySwitchKey):
        self.key = key


class GaloisKey:
    def __init__(self, galois_element: int, key: KeySwitchKey):
        self.galois_element, self.key = galois_element, key
        self._prepared = None

    def prepared(self) -> DeviceBuffer:
        """the packed key with its NTT-domain rows permuted by g^-1 (tfhe_galois_key_prepare), once per key: what the hoisted
        rotations consume"""
        if self._prepared is None:
            ring = self.key.key[0].mask.ring
            src = self.key.packed()
            dst = DeviceBuffer(src.n)
            ring.ctx.galois_key_prepare(ring.L, len(self.key.key), self.galois_element, src.ptr, dst.ptr)
            self._prepared = dst
        return self._prepared


# the forward transforms of many small ring elements in one launch (CipherText.dot_plain).  Above _BATCH_NTT_MAX_WORDS a single element's
# transform fills the chip by itself; a chunk stages at most _BATCH_NTT_CHUNK_WORDS (1 GiB) of coefficients, and as much of results.
_BATCH_NTT_MAX_WORDS = 0 if os.environ.get("TFHE_BATCH_NTT", "1") == "0" else 1 << 25   # TFHE_BATCH_NTT=0: one transform per element (comparisons)
_BATCH_NTT_CHUNK_WORDS = 1 << 27


def _stage_buffers(ctx, words):
    """the context's two staging buffers (coefficients in, evaluation-domain rows out), grown on demand and kept for the context's
    lifetime: everything that touches them is enqueued on the context's own stream, so a chunk may reuse them as soon as the previous
    chunk's calls are enqueued.  (r06: fresh multi-GiB buffers per call went through the recycling allocator -- past its soft threshold
    every first-time size is a device drain, `hipFree` of stale blocks and a `hipMalloc`: the reference-shaped MNIST pass ran 215 ms on its
    own and 520-680 ms behind the restructured case in one process.)"""
    st = getattr(ctx, "_ntt_stage", None)
    if st is None or st[0].n < words:
        ctx._ntt_stage = st = (DeviceBuffer(words), DeviceBuffer(words))
    return st


def release_staging(ctx):
    """give the staging buffers of `_stage_buffers` back (they are re-created on the next use)"""
    if getattr(ctx, "_ntt_stage", None) is not None:
        ctx._ntt_stage = None


class _PackedComponent:
    """component p of a key-switch result that is still one packed buffer [count][P][L][N] (_PackedResult): an operand of _dot_batched
    that is staged by the strided copy which would otherwise have split it off"""
    dual = None

    def __init__(self, owner, image, P, p):
        self.owner, self.image, self.P, self.p = owner, image, P, p

    def coeffs_dual(self):                                   # (the un-batched paths: the owner's ring element after all)
        return self.owner.cs[self.p].coeffs_dual()


def _component_source(c, p):
    """component p of ciphertext c as an operand of _dot_batched: the ring element, or -- for a result that has not been split --
    its place in the packed buffer"""
    if isinstance(c, _PackedResult) and c._cs is None:
        return _PackedComponent(c, c._packed_image[0], 2, p)
    return c.cs[p]


def _dot_batched(ring, n, elems, pb_ptrs, dst):
    """dst = sum_k elems[k] .* plain_k for ring elements of ONE ring and batch size, the plaintexts given by their evaluation-domain
    pointers.  Elements whose transform is cached are used where they lie.  When two or more are still in the coefficient domain
    and small against the chip (r06: a batch of 16 ciphertext components at N = 2^16 is 16 rows of a 60-bit limb -- 64 of the 256
    workgroup slots -- per call; the reference-shaped matrix product of infer.jl:140-149 asks for 2 x 63 such transforms, one per
    rotated ciphertext) the sum runs chunk by chunk: a chunk's operands are copied side by side into the context's staging buffer,
    transformed in ONE call, and accumulated onto dst (tfhe_dot with dst as its running sum: residues are canonical, so the partial
    sums are the same words as one pass over all terms).  The transforms are the same per row: bit for bit `coeffs_dual()`.
    Nothing is cached on the elements."""
    words = n * ring.L * ring.N
    todo = sum(1 for e in elems if e.dual is None)
    if todo < 2 or words > _BATCH_NTT_MAX_WORDS:
        ring.ctx.dot(None, [e.coeffs_dual().ptr for e in elems], pb_ptrs, dst.ptr, n, ring.L, ring.idx)
        return
    per = max(2, _BATCH_NTT_CHUNK_WORDS // words)
    lib = native.lib()
    acc = None
    for a in range(0, len(elems), per):
        part = elems[a:a + per]
        stage = [e for e in part if e.dual is None]
        ptrs = [e.dual.ptr if e.dual is not None else None for e in part]
        if len(stage) == 1:                                  # (a lone straggler of the last chunk: its own transform)
            i = ptrs.index(None)
            ptrs[i] = stage[0].coeffs_dual().ptr
        elif stage:
            src, dual = _stage_buffers(ring.ctx, len(stage) * words)
            for k, e in enumerate(stage):
                if isinstance(e, _PackedComponent):          # straight out of an unsplit key-switch result: the one strided copy
                    native.check(lib.tfhe_unpack_poly(ring.ctx.h, src.ptr + k * words * 8, e.image.ptr, e.P, e.p, ring.L * ring.N, n))
                else:
                    native.check(lib.tfhe_memcpy_d2d(ring.ctx.h, src.ptr + k * words * 8, e.coeffs_primal().ptr, words * 8))
            ring.ctx.nntt(src.ptr, dual.ptr, len(stage) * n, ring.L, ring.idx)
            k = 0
            for i, e in enumerate(part):
                if ptrs[i] is None:
                    ptrs[i] = dual.ptr + k * words * 8
                    k += 1
        ring.ctx.dot(acc, ptrs, pb_ptrs[a:a + per], dst.ptr, n, ring.L, ring.idx)
        acc = dst.ptr


class CipherText:
    """CipherText{Plain,P,T,N}(params, cs), rlwe_she.jl:131-149.  ``scale`` carries the CKKS FixedRational
    denominator (a type parameter upstream, ckksencoding.jl:3-15)."""

    def __init__(self, params, cs, scale=None):
        self.params, self._cs, self.scale = params, tuple(cs), scale
        # (packed [count][polys][L][N] image, the component buffers it was split into): a key switch / rotation returns its packed
        # result split into ring elements (strided copies); a chained caller (infer.jl:140-149: rotated = rotate(gk, rotated)) hands
        # the same elements straight back, and the next call takes the image instead of packing them again.  Ring elements never
        # change a device buffer in place (setindex! replaces it), so identity of the buffers is validity of the image.
        self._packed_image = None

    def _remember_packed(self, image, ctx):
        self._packed_image = (image, ctx, tuple(x.primal for x in self.cs))
        return self

    def _packed_for(self, prim, ctx, consume=False):
        """the packed image of exactly these coefficient buffers on this context, or None.  ``consume``: the image is handed
        over and forgotten (r06, ADVICE r05: a result ciphertext otherwise holds its packed buffer next to the unpacked
        components -- twice the device memory -- for as long as it lives; a chained caller uses it exactly once)"""
        pi = self._packed_image
        if pi is not None and pi[1] is ctx and len(pi[2]) == len(prim) and all(a is b for a, b in zip(pi[2], prim)):
            if consume:
                self._packed_image = None
            return pi[0]
        if consume:
            self._packed_image = None      # stale (a component was replaced): nothing can use it any more
        return None

    @property
    def cs(self):
        """the components (ring elements).  A deferred form (_ScalarSum) evaluates itself on the first access."""
        if self._cs is None:
            self._cs = tuple(self._evaluate())
        return self._cs

    @cs.setter
    def cs(self, v):
        self._cs = tuple(v)

    def _evaluate(self):
        raise AssertionError("a plain ciphertext always has its components")

    def __len__(self):
        return len(self.cs)

    def __getitem__(self, i):
        return self.cs[i]

    def ring(self):
        return self.cs[0].ring

    def _shape(self):
        """(polynomials per component, batch) without evaluating a deferred form"""
        return self.cs[0].count, self.cs[0].batch

    def __repr__(self):
        return f"{self.params.scheme_name} ciphertext (length {len(self)})"

    # homomorphic arithmetic, rlwe_she.jl:231-266
    def _addsub(self, o, sub):
        if self.params is not o.params:
            raise UsageError("Attempting to add ciphertexts with differing parameters")
        if isinstance(self, _ScalarSum) and isinstance(o, _ScalarSum):
            both = self._joined(o, sub)
            if both is not None:
                return both
        n = max(len(self), len(o))
        cs = []
        for i in range(n):
            if i >= len(self):
                cs.append(-o[i] if sub else o[i])
            elif i >= len(o):
                cs.append(self[i])
            else:
                cs.append(self[i] - o[i] if sub else self[i] + o[i])
        return CipherText(self.params, cs, self.scale)

    def __add__(self, o):
        if isinstance(o, RingElement):  # +(c, b::T), rlwe_she.jl:243-245
            return CipherText(self.params, (self.cs[0] + o,) + self.cs[1:], self.scale)
        return self._addsub(o, False)

    def __sub__(self, o):
        return self._addsub(o, True)

    def __mul__(self, o):
        if isinstance(o, CipherText):
            if (self.scale is None) != (o.scale is None):
                raise UsageError("multiplying a scaled (CKKS) ciphertext by an unscaled one")
            scale = None if self.scale is None else self.scale * o.scale  # ckksencoding.jl:133-135
            return CipherText(self.params, enc_mul(self, o), scale)
        if isinstance(o, int):
            return CipherText(self.params, [c * o for c in self.cs], self.scale)
        if isinstance(o, float):                     # ct * b::AbstractFloat, ckksencoding.jl:99-102
            return self.mul_plain(o)
        raise TypeError(type(o))

    __rmul__ = __mul__

    # ---- CKKS plaintext operands (ckksencoding.jl:99-124); the ciphertext carries its scale ----
    def _need_scale(self):
        if self.scale is None:
            raise UsageError("plaintext operands need a CKKS ciphertext (with a scale)")

    def mul_plain(self, x) -> "CipherText":
        """ct * float (:99-102) or vector .* ct (:104-109): the operand is brought to the ciphertext's scale, every
        component is multiplied by it; the result's scale is the square."""
        self._need_scale()
        if np.isscalar(x):
            fr = Fraction(float(x)) * Fraction(self.scale)
            fl = fr.numerator // fr.denominator
            rem = fr - fl
            scaled = fl + (1 if rem > Fraction(1, 2) or (rem == Fraction(1, 2) and fl % 2) else 0)   # FixedRational(b).x, ckks.jl:42
            if _LAZY_SCALAR_SUMS:
                return _ScalarSum.term(self, int(scaled))
            cs = [c * int(scaled) for c in self.cs]
        elif isinstance(x, RingElement):             # a plaintext the caller has encoded at this ciphertext's scale already
            if x.ring != self.ring():
                raise UsageError("pre-encoded plaintext belongs to another ring")
            cs = [c * x for c in self.cs]
        else:
            re = ckks_encode(np.asarray(x, dtype=np.complex128), self.ring(), self.scale)
            cs = [c * re for c in self.cs]
        return CipherText(self.params, cs, Fraction(self.scale) ** 2)

    @staticmethod
    def dot_plain(cts, plains) -> "CipherText":
        """sum_k cts[k] .* plains[k] for ciphertexts of one ring, length and scale and plaintexts already encoded at that scale
        (ring elements): the accumulation loop of the diagonal matrix product (infer.jl:140-149) as one device pass per
        component (tfhe_dot) instead of a ring multiplication and a ring addition per term; bit-identical to
        `sum(c.mul_plain(p) for c, p in zip(cts, plains))`."""
        cts, plains = list(cts), list(plains)
        if not cts or len(cts) != len(plains):
            raise AssertionError("dot_plain: as many ciphertexts as plaintexts, at least one")
        c0 = cts[0]
        c0._need_scale()
        ring = c0.ring()
        n, batch = c0._shape()
        for c in cts:
            if c.ring() != ring or len(c) != len(c0) or c.scale != c0.scale or c._shape()[0] != n:
                raise UsageError("dot_plain: ciphertexts of one ring, length, batch and scale")
        for p in plains:
            if not isinstance(p, RingElement) or p.ring != ring or p.count != n:
                raise UsageError("dot_plain: plaintexts must be ring elements of the ciphertexts' ring and batch")
        pb = [p.coeffs_dual() for p in plains]
        out = []
        for s_ in range(len(c0)):
            o = DeviceBuffer(n * ring.L * ring.N)
            _dot_batched(ring, n, [_component_source(c, s_) for c in cts], [x.ptr for x in pb], o)
            out.append(RingElement(ring, None, o, batch))
        return CipherText(c0.params, out, Fraction(c0.scale) ** 2)

    @staticmethod
    def _weight_residues(w: float, scale, moduli):
        """FixedRational(w).x at the ciphertext's scale, ckks.jl:42 (ties to even), as residues.  Integer arithmetic on the
        float's exact ratio (a layer has hundreds of weights: Fraction objects cost a millisecond per call)."""
        n, d = float(w).as_integer_ratio()
        sc = scale if isinstance(scale, Fraction) else Fraction(scale)
        num, den = n * sc.numerator, d * sc.denominator
        fl, rem = divmod(num, den)
        v = fl + (1 if 2 * rem > den or (2 * rem == den and fl % 2) else 0)
        return [v % q for q in moduli]

    @staticmethod
    def lincomb(cts, weights) -> "CipherText":
        """sum_k cts[k] * weights[k] for float weights (ct * float, ckksencoding.jl:99-102, and the + of rlwe_she.jl:231-245 per
        term): the scalar-weighted sums of a convolution over encrypted inputs (infer.jl:127-129) as ONE device pass per
        component (tfhe_lincomb); bit-identical to `sum(c.mul_plain(w) for c, w in zip(cts, weights))`."""
        return CipherText.lincomb_many(cts, [weights])[0]

    @staticmethod
    def lincomb_many(cts, weight_rows) -> "list[CipherText]":
        """[sum_k cts[k] * row[k] for row in weight_rows]: several weighted sums of the SAME ciphertexts -- the output channels of
        a convolution layer (infer.jl:127-131: every channel weighs the same 49 encrypted inputs) -- with one pass over the
        operands per component (tfhe_lincomb_many); each result bit-identical to `lincomb(cts, row)`."""
        cts = list(cts)
        rows = [[float(w) for w in row] for row in weight_rows]
        if not cts or not rows or any(len(r) != len(cts) for r in rows):
            raise AssertionError("lincomb: as many ciphertexts as weights, at least one")
        if len(cts) > DOT_MAX:      # a device pass takes DOT_MAX operands: longer sums as partial sums (the same residues: + is exact)
            parts = [CipherText.lincomb_many(cts[k:k + DOT_MAX], [r[k:k + DOT_MAX] for r in rows]) for k in range(0, len(cts), DOT_MAX)]
            outs = parts[0]
            for more in parts[1:]:
                outs = [a + b for a, b in zip(outs, more)]
            return outs
        c0 = cts[0]
        c0._need_scale()
        ring, n, batch = c0.ring(), c0[0].count, c0[0].batch
        for c in cts:
            if c.ring() != ring or len(c) != len(c0) or c.scale != c0.scale or c[0].count != n:
                raise UsageError("lincomb: ciphertexts of one ring, length, batch and scale")
        sc, mods = Fraction(c0.scale), list(ring.moduli)
        scal = [[CipherText._weight_residues(w, sc, mods) for w in row] for row in rows]
        primal = all(c.cs[0].primal is not None for c in cts)   # stay in the domain the operands are in
        comps = [[] for _ in rows]
        for s_ in range(len(c0)):
            ab = [(c.cs[s_].coeffs_primal() if primal else c.cs[s_].coeffs_dual()) for c in cts]
            outs = [DeviceBuffer(n * ring.L * ring.N) for _ in rows]
            if len(rows) == 1:
                ring.ctx.lincomb(scal[0], [x.ptr for x in ab], outs[0].ptr, n, ring.L, ring.idx)
            else:
                ring.ctx.lincomb_many(scal, [x.ptr for x in ab], [o.ptr for o in outs], n, ring.L, ring.idx)
            for k, o in enumerate(outs):
                comps[k].append(RingElement(ring, o, None, batch) if primal else RingElement(ring, None, o, batch))
        return [CipherText(c0.params, out, Fraction(c0.scale) ** 2) for out in comps]

    @staticmethod
    def concat(cts) -> "CipherText":
        """the batches of several ciphertexts (same parameters, length and scale) as one batched ciphertext: the independent
        ciphertexts of one circuit layer share every device call that follows"""
        cts = list(cts)
        c0 = cts[0]
        if any(len(c) != len(c0) or c.scale != c0.scale or c.ring() != c0.ring() for c in cts):
            raise UsageError("concat: ciphertexts of one ring, length and scale")
        return CipherText(c0.params, [RingElement.concat([c.cs[s_] for c in cts]) for s_ in range(len(c0))], c0.scale)

    def split(self, sizes) -> "list[CipherText]":
        """inverse of concat"""
        parts = [x.split(sizes) for x in self.cs]
        return [CipherText(self.params, [p[k] for p in parts], self.scale) for k in range(len(sizes))]

    def add_plain(self, x) -> "CipherText":
        """ct .+ float / ct .+ vector (:111-124): encoded at the ciphertext's scale and added to the first component."""
        self._need_scale()
        n2 = self.ring().N // 2
        v = np.full(n2, x, dtype=np.complex128) if np.isscalar(x) else np.asarray(x, dtype=np.complex128)
        re = ckks_encode(v, self.ring(), self.scale)
        return CipherText(self.params, (self.cs[0] + re,) + self.cs[1:], self.scale)


# ct * float + ct * float + ... (infer.jl:127-129: `sum(C[i,j] * w[i,j] for i, j)`, 49 terms per channel) term by term is a scalar
# multiplication and an addition per term and component -- two passes over a ciphertext each, 27 us apiece at the example's batch, 784
# launches per pass.  TFHE_LAZY_SUMS=0 keeps that.
_LAZY_SCALAR_SUMS = os.environ.get("TFHE_LAZY_SUMS", "1") != "0"
_LAZY_UNPACK = os.environ.get("TFHE_LAZY_UNPACK", "1") != "0"      # key-switch / rotation results stay packed until their components are asked for


class _ScalarSum(CipherText):
    """sum_k s_k x_k, s_k integers (FixedRational(w).x of `ct * float`, ckksencoding.jl:99-102), NOT YET EVALUATED: `mul_plain(float)`
    returns one term, `+` / `-` of two such sums joins their terms, and the first use of the components evaluates the whole sum in one
    device pass per component and DOT_MAX terms (tfhe_lincomb) -- the same residues as the term-by-term evaluation, every operation being
    exact modulo each q.  What is captured per term is the DEVICE BUFFERS of the operand's components at the time of the
    multiplication (ring elements never change a buffer in place: setindex! replaces it), so a later change of the operand does not
    reach the sum -- the value semantics of the eager form.  Errors of the eager form are raised where it would raise them
    (`_need_scale` in mul_plain, differing parameters in `+`); sums that do not fit together (ring, batch, length, scale, domain)
    are evaluated and added as ordinary ciphertexts."""

    def __init__(self, params, scale, ring, n, batch, domain, terms):
        self.params, self._cs, self.scale = params, None, scale
        self._packed_image = None
        self._ring, self._n, self._batch, self._domain, self._terms = ring, n, batch, domain, terms

    @staticmethod
    def term(c: CipherText, scalar: int) -> "_ScalarSum":
        cs = c.cs
        primal = all(x.primal is not None for x in cs)             # stay in the domain the operand is in
        bufs = tuple(x.coeffs_primal() if primal else x.coeffs_dual() for x in cs)
        return _ScalarSum(c.params, Fraction(c.scale) ** 2, cs[0].ring, cs[0].count, cs[0].batch, primal, [(bufs, scalar)])

    def _joined(self, o, sub):
        if self._cs is not None or o._cs is not None:               # one of them has been evaluated already: ordinary ciphertexts
            return None
        if (self._ring != o._ring or self._n != o._n or self._batch != o._batch or self._domain != o._domain or self.scale != o.scale
                or len(self._terms[0][0]) != len(o._terms[0][0])):
            return None
        more = [(b, -k) for b, k in o._terms] if sub else list(o._terms)
        return _ScalarSum(self.params, self.scale, self._ring, self._n, self._batch, self._domain, self._terms + more)

    def __len__(self):
        return len(self._terms[0][0]) if self._cs is None else len(self._cs)

    def ring(self):
        return self._ring

    def _shape(self):
        return self._n, self._batch

    def _evaluate(self):
        ring, n = self._ring, self._n
        mods = list(ring.moduli)
        out = []
        for s_ in range(len(self._terms[0][0])):
            acc = None
            for a in range(0, len(self._terms), DOT_MAX):           # a device pass takes DOT_MAX operands (+ is exact)
                part = self._terms[a:a + DOT_MAX]
                o = DeviceBuffer(n * ring.L * ring.N)
                ring.ctx.lincomb([[k % q for q in mods] for _, k in part], [b[s_].ptr for b, _ in part], o.ptr, n, ring.L, ring.idx)
                el = RingElement(ring, o, None, self._batch) if self._domain else RingElement(ring, None, o, self._batch)
                acc = el if acc is None else acc + el
            out.append(acc)
        self._terms = None                                          # the operands' buffers are no longer needed
        return out


class _PackedResult(CipherText):
    """The result of a key switch / rotation as the device call left it: ONE packed buffer [count][2][L][N] on the key ring's context,
    split into ring elements only when somebody asks for the components (r06).  The chained rotations of infer.jl:140-149 never do:
    rotation k + 1 takes the packed buffer as its input, and the accumulation (`CipherText.dot_plain`) stages its operands straight
    out of the packed buffers -- two strided copies and two buffers per rotation less than splitting every result.  Evaluated, it
    is an ordinary ciphertext that remembers its packed image (`_packed_image`, as before)."""

    def __init__(self, params, scale, image, ctx, ring, n, batch):
        self.params, self._cs, self.scale = params, None, scale
        self._packed_image = (image, ctx, None)
        self._ring, self._n, self._batch = ring, n, batch

    def _deferred_image(self, ctx):
        """the packed buffer if the components have not been asked for and it lives on `ctx`, else None (it stays with this ciphertext)"""
        pi = self._packed_image
        return pi[0] if self._cs is None and pi is not None and pi[1] is ctx else None

    def __len__(self):
        return 2

    def ring(self):
        return self._ring

    def _shape(self):
        return self._n, self._batch

    def _evaluate(self):
        image, ctx, _ = self._packed_image
        cs = _unpack(image, self._ring, self._n, 2, self._batch, primal=True, ctx=ctx)
        if self._ring.ctx is not ctx:
            self._ring.ctx.wait_for(ctx)                   # the components are elements of the ciphertext ring: its stream must see them written
        self._packed_image = (image, ctx, tuple(x.primal for x in cs))
        return cs


# --------------------------------------------------------------------------------------------------
# keygen / encrypt / decrypt (rlwe_she.jl:155-217, modulusraising.jl:23-26)
# --------------------------------------------------------------------------------------------------


def keygen(rng, params) -> KeyPair:
    ring = params.R_key()
    mask = sample_uniform(rng, ring)
    secret = params.secret_dist(rng, ring)
    error = params.noise(rng, ring)
    masked = -(mask * secret + error)
    return KeyPair(PrivKey(params, secret), PubKey(params, KeyComponent(mask, masked)))


def encrypt_zero(rng, pub: PubKey, batch=None) -> CipherText:
    params = pub.params
    ring = params.R_key()
    u = params.secret_dist(rng, ring, batch)
    e1, e2 = params.noise(rng, ring, batch), params.noise(rng, ring, batch)
    mask, masked = pub.key.mask, pub.key.masked
    if batch is not None:  # broadcast the key over the batch
        mask, masked = _broadcast(mask, batch), _broadcast(masked, batch)
    c1 = masked * u + e1
    c2 = mask * u + e2
    cs = (c1, c2)
    if isinstance(params, ModulusRaised):  # modulusraising.jl:23-26: drop the special limb
        cs = tuple(c.modswitch_drop() for c in cs)
    return CipherText(params, cs)


def _broadcast(el: RingElement, batch: int) -> RingElement:
    el.coeffs_dual()                                       # keys are used as multiplicands: broadcast the NTT form
    return RingElement(el.ring, None, el.dual, None).broadcast_to(batch)


def encrypt(rng, key, plaintext, scale=None) -> CipherText:
    pub = key.pub if isinstance(key, KeyPair) else key
    enc = pub.params.encode(plaintext)
    c = encrypt_zero(rng, pub, enc.batch)
    c = c + enc  # rlwe_she.jl:190
    c.scale = scale
    return c


def decrypt(key, c: CipherText):
    priv = key.priv if isinstance(key, KeyPair) else key
    secret = priv.secret
    while secret.ring.L != c[0].ring.L:  # rlwe_she.jl:202-204
        secret = secret.modswitch_drop()
    if c[0].batch is not None:
        secret = _broadcast(secret, c[0].batch)
    b, spow = c[0], secret
    for i in range(1, len(c)):
        b = b + spow * c[i]
        if i + 1 < len(c):
            spow = spow * secret
    return priv.params.decode(b)


def invariant_noise_budget(key, c: CipherText) -> float:
    """invariant_noise_budget(pk::PrivKey{BFVParams}, c), bfv.jl:137-166:
    log2(q) - log2(t) - 1 - max_i log2(birem(b_i)) with b = c_1 + s c_2 + s^2 c_3 ... and birem(x) = min(x mod Δ, Δ - x mod Δ).
    A host-side diagnostic over the device ring ops (one ciphertext, not a batch)."""
    priv = key.priv if isinstance(key, KeyPair) else key
    params = priv.params
    if not isinstance(params, BFVParams) or c[0].batch is not None:
        raise NotImplementedError("invariant_noise_budget: single BFV ciphertexts")
    b, spow = c[0], priv.secret
    for i in range(1, len(c)):
        b = b + spow * c[i]
        if i + 1 < len(c):
            spow = spow * priv.secret
    delta = params.delta

    def birem(x):
        r = x % delta
        return delta - r if r > delta // 2 else r
    worst = max(birem(x) for x in b.to_ints())
    return math.log2(params.ring.modulus()) - math.log2(params.t) - 1 - (math.log2(worst) if worst else 0.0)


def slot_encode(plain_ring: NegacyclicRing, slots) -> RingElement:
    """SlotEncoding -> RingElement (encoding.jl:35-52): the slots ARE the NTT-domain (dual) coefficients, natural order."""
    vals = [int(v) % plain_ring.moduli[0] for v in slots]
    res = np.array([vals], dtype=np.uint64)
    return RingElement.from_host(plain_ring, res, dual=True)


def slot_decode(el: RingElement):
    """SlotEncoding(r) (encoding.jl:35-43): coeffs_dual of a plaintext ring element."""
    return [int(v) for v in el.to_numpy("dual")[0]]


# --------------------------------------------------------------------------------------------------
# multiplication (rlwe_she.jl:247-266 + bfv.jl:34-40)
# --------------------------------------------------------------------------------------------------


def enc_mul(c1: CipherText, c2: CipherText):
    if c1.params is not c2.params:
        raise UsageError("Attempting to multiply ciphertexts with differing parameters")
    params = c1.params
    ring, batch = c1[0].ring, c1[0].batch
    n = c1[0].count
    sz = ring.L * ring.N
    if isinstance(params, BFVParams):
        if len(c1) != 2 or len(c2) != 2:
            # any component counts (rlwe_she.jl:247-262 as written: e.g. (c*c)*c without relinearisation): mul_expand per
            # component (bfv.jl:34), the convolution over ℛbig, mul_contract per output component (bfv.jl:35-40).  The
            # conversions order ℛ's and ℛbig's streams themselves (tfhe_bfv_expand / tfhe_bfv_contract).
            e1, e2 = [params.mul_expand(c) for c in c1.cs], [params.mul_expand(c) for c in c2.cs]
            cs = [None] * (len(e1) + len(e2) - 1)
            for i, x in enumerate(e1):
                for j, y in enumerate(e2):
                    p = x * y
                    cs[i + j] = p if cs[i + j] is None else cs[i + j] + p
            return tuple(params.mul_contract(c) for c in cs)
        a, b = _pack([c.coeffs_primal() for c in c1.cs], ring, n), _pack([c.coeffs_primal() for c in c2.cs], ring, n)
        out = DeviceBuffer(n * 3 * sz)
        params.plan().mul(a.ptr, b.ptr, out.ptr, n)
        return _unpack(out, ring, n, 3, batch, primal=True)
    if len(c1) == 2 and len(c2) == 2:  # mul_expand / mul_contract are the identity (rlwe_she.jl:39-40)
        a, b = _pack([c.coeffs_dual() for c in c1.cs], ring, n), _pack([c.coeffs_dual() for c in c2.cs], ring, n)
        out = DeviceBuffer(n * 3 * sz)
        ring.ctx.tensor(a.ptr, b.ptr, out.ptr, n, ring.L, ring.idx)
        return _unpack(out, ring, n, 3, batch, primal=False)
    cs = [None] * (len(c1) + len(c2) - 1)  # generic convolution of components
    for i in range(len(c1)):
        for j in range(len(c2)):
            p = c1[i] * c2[j]
            cs[i + j] = p if cs[i + j] is None else cs[i + j] + p
    return tuple(cs)


def _pack(bufs, ring, n, ctx=None) -> DeviceBuffer:
    """[poly](n, L, N) -> (n, polys, L, N): one strided device copy per component, on the stream of `ctx` (the context that
    will consume the packed batch)."""
    ctx = ctx or ring.ctx
    P, sz = len(bufs), ring.L * ring.N
    out = DeviceBuffer(n * P * sz)
    lib = native.lib()
    for p, b in enumerate(bufs):
        native.check(lib.tfhe_pack_poly(ctx.h, out.ptr, b.ptr, P, p, sz, n))
    return out


def _unpack(buf, ring, n, P, batch, primal=True, ctx=None):
    """(n, P, L, N) -> P ring elements of shape (n, L, N), on the stream of `ctx` (the context that produced `buf`)."""
    ctx = ctx or ring.ctx
    sz = ring.L * ring.N
    lib = native.lib()
    outs = []
    for p in range(P):
        o = DeviceBuffer(n * sz)
        native.check(lib.tfhe_unpack_poly(ctx.h, o.ptr, buf.ptr, P, p, sz, n))
        outs.append(RingElement(ring, o, None, batch) if primal else RingElement(ring, None, o, batch))
    # no host wait: `buf` may be released by the caller as soon as we return -- the allocator parks a released block until
    # the work submitted so far on every context stream has finished (tfhe_free, dev_alloc.h)
    return tuple(outs)


# --------------------------------------------------------------------------------------------------
# key switching (rlwe_she.jl:273-360, modulusraising.jl:28-49)
# --------------------------------------------------------------------------------------------------


def make_eval_key(rng, old: RingElement, new: PrivKey) -> KeySwitchKey:
    """make_eval_key(rng, old => new), rlwe_she.jl:273-298 with the RNS gadget (:287); ModulusRaised
    pre-multiplies ``old`` by the special prime (modulusraising.jl:28-32)."""
    params = new.params
    ring = old.ring
    if isinstance(params, ModulusRaised):
        old = old * ring.moduli[-1]                       # P * old over the key ring Q P; the gadget below is the parent's
    key = []
    if params.relin_window != 0:                          # base-2^w gadget, rlwe_she.jl:281-283
        w = params.relin_window
        nwin = -(-ring.modulus().bit_length() // w)       # ndigits(Q, base = 2^w)
        gadget = [old * pow(2, i * w, ring.modulus()) for i in range(nwin)]
    else:
        res = old.to_numpy("primal")
        gadget = []
        for i in range(ring.L):
            g = np.zeros_like(res)
            g[i] = res[i]                                 # CRTResidual, crt.jl:64-77
            gadget.append(RingElement.from_host(ring, g))
    for g in gadget:
        mask = sample_uniform(rng, ring)
        e = params.noise(rng, ring)
        key.append(KeyComponent(mask, g - (mask * new.secret + e)))
    return KeySwitchKey(params, key)


def keygen_evalmult(rng, priv: PrivKey) -> EvalMultKey:
    return EvalMultKey(make_eval_key(rng, priv.secret * priv.secret, priv))  # rlwe_she.jl:299


def galois_element_for_steps(steps: int, N: int) -> int:
    return pow(3, 2 * N - steps, 2 * N) if steps > 0 else pow(3, -steps, 2 * N)  # rlwe_she.jl:304


def keygen_galois(rng, priv: PrivKey, galois_element=None, steps=None) -> GaloisKey:
    assert (galois_element is None) != (steps is None)  # rlwe_she.jl:301
    if galois_element is None:
        galois_element = galois_element_for_steps(steps, priv.secret.ring.N)
    return GaloisKey(galois_element, make_eval_key(rng, priv.secret.apply_galois_element(galois_element), priv))


def keyswitch(ek, c: CipherText, _galois=None, _gk=None) -> CipherText:
    """keyswitch(ek, c), rlwe_she.jl:315-349 -- one fused device call."""
    if isinstance(ek, (EvalMultKey, GaloisKey)):
        ek = ek.key
    if len(c) not in (2, 3):
        raise AssertionError("keyswitch needs a 2- or 3-element ciphertext")  # rlwe_she.jl:318
    params = ek.params
    keyring = ek.key[0].mask.ring
    if params.relin_window != 0:                          # base-2^w digits, rlwe_she.jl:330-338
        special = isinstance(params, ModulusRaised)       # digits of c[end] over Q_level, keys over [q_0..q_{l-1}, P] (modulusraising.jl:35-49)
        ring, n, batch = c[0].ring, c[0].count, c[0].batch
        level = ring.L
        if keyring.idx != list(range(keyring.L)) or ring.idx != list(range(level)) or level > keyring.L - (1 if special else 0):
            raise UsageError("window key switch: the ciphertext ring must be a prefix of the key ring")
        cs = c.cs if _galois is None else [x.apply_galois_element(_galois) for x in c.cs]
        prim = [x.coeffs_primal() for x in cs]            # may enqueue inverse transforms on the ciphertext ring's stream: BEFORE the hand-over
        if ring.ctx is not keyring.ctx:
            if ring.N != keyring.N or ring.moduli != keyring.moduli[:level] or ring.psi != keyring.psi[:level]:
                raise UsageError("ciphertext and key belong to different rings")
            keyring.ctx.wait_for(ring.ctx)
        ct = _pack(prim, ring, n, ctx=keyring.ctx)
        out = DeviceBuffer(n * 2 * level * ring.N)
        keyring.ctx.keyswitch_window(level, params.relin_window, ek.packed().ptr, len(ek.key), ct.ptr, len(c), out.ptr, n,
                                     key_limbs=keyring.L, special=special)
        res = _unpack(out, ring, n, 2, batch, primal=True, ctx=keyring.ctx)
        if ring.ctx is not keyring.ctx:
            ring.ctx.wait_for(keyring.ctx)
        return CipherText(c.params, res, c.scale)
    special = isinstance(params, ModulusRaised)
    ring = c.ring()
    n, batch = c._shape()
    level = ring.L
    if keyring.idx != list(range(keyring.L)) or ring.idx != list(range(level)):
        raise UsageError("ciphertext ring is not a prefix of the key ring")
    if ring.ctx is not keyring.ctx and (ring.N != keyring.N or ring.moduli != keyring.moduli[:level] or ring.psi != keyring.psi[:level]):
        raise UsageError("ciphertext and key belong to different rings")
    sz = level * ring.N
    ct = c._deferred_image(keyring.ctx) if isinstance(c, _PackedResult) else None   # an unsplit result of this very context: in stream order already
    if ct is None:
        prim = [x.coeffs_primal() for x in c.cs]           # may enqueue inverse transforms on the ciphertext ring's stream ...
        if ring.ctx is not keyring.ctx:
            keyring.ctx.wait_for(ring.ctx)                 # ... so the hand-over to the key ring's stream comes after them
        ct = c._packed_for(prim, keyring.ctx, consume=True) or _pack(prim, ring, n, ctx=keyring.ctx)
    out = DeviceBuffer(n * 2 * sz)
    if _galois is None:
        keyring.ctx.keyswitch(keyring.L, level, special, ek.packed().ptr, len(ek.key), ct.ptr, len(c), out.ptr, n)
    elif _gk is not None and ring.N >= (1 << 15) and n >= 8:
        # N >= 2^15, 8 ciphertexts or more: the rotation is finished in the key switch's tail on the key as
        # tfhe_galois_key_prepare leaves it -- prepared once per key (GaloisKey.prepared), not once per call (r06:
        # infer.jl:140-149 rotates by ONE key 63 times per matrix product)
        keyring.ctx.rotate(keyring.L, level, special, _gk.prepared().ptr, len(ek.key), _galois, ct.ptr, out.ptr, n, prepared=True)
    else:
        keyring.ctx.rotate(keyring.L, level, special, ek.packed().ptr, len(ek.key), _galois, ct.ptr, out.ptr, n)
    if not _LAZY_UNPACK:
        cs = _unpack(out, ring, n, 2, batch, primal=True, ctx=keyring.ctx)
        if ring.ctx is not keyring.ctx:
            ring.ctx.wait_for(keyring.ctx)                 # the results are elements of `ring`: its stream must see them written
        return CipherText(c.params, cs, c.scale)._remember_packed(out, keyring.ctx)
    if ring.ctx is not keyring.ctx:
        ring.ctx.wait_for(keyring.ctx)                     # the result belongs to `ring`: its stream must see it written
    return _PackedResult(c.params, c.scale, out, keyring.ctx, ring, n, batch)


def apply_galois_element(c: CipherText, g: int) -> CipherText:
    return CipherText(c.params, [x.apply_galois_element(g) for x in c.cs], c.scale)  # rlwe_she.jl:355-357


def rotate(gk: GaloisKey, c: CipherText) -> CipherText:
    """rotate(gk, c) = keyswitch(gk, apply_galois_element(c, g)), rlwe_she.jl:359 (fused on the device)."""
    if len(c) != 2:
        raise AssertionError("rotate takes a 2-element ciphertext")
    return keyswitch(gk.key, c, _galois=gk.galois_element, _gk=gk)


def rotate_many(gks, c: CipherText):
    """[rotate(gk, c) for gk in gks] from one digit decomposition of c (tfhe_rotate_many: the forward transforms of the RNS
    digits are shared by all rotations); each result is bit-identical to `rotate(gk, c)`.  RNS-digit keys (relin_window = 0)."""
    gks = list(gks)
    if len(c) != 2:
        raise AssertionError("rotate takes a 2-element ciphertext")
    if not gks:
        return []
    params = gks[0].key.params
    if any(g.key.params is not params for g in gks) or params.relin_window != 0:
        raise UsageError("hoisted rotations need Galois keys of one parameter set with RNS digits")
    keyring = gks[0].key.key[0].mask.ring
    ring, n, batch = c[0].ring, c[0].count, c[0].batch
    level, special = ring.L, isinstance(params, ModulusRaised)
    if keyring.idx != list(range(keyring.L)) or ring.idx != list(range(level)):
        raise UsageError("ciphertext ring is not a prefix of the key ring")
    prim = [x.coeffs_primal() for x in c.cs]               # before the hand-over (see keyswitch)
    if ring.ctx is not keyring.ctx:
        if ring.N != keyring.N or ring.moduli != keyring.moduli[:level] or ring.psi != keyring.psi[:level]:
            raise UsageError("ciphertext and key belong to different rings")
        keyring.ctx.wait_for(ring.ctx)
    sz = level * ring.N
    ct = _pack(prim, ring, n, ctx=keyring.ctx)
    out = DeviceBuffer(len(gks) * n * 2 * sz)
    keyring.ctx.rotate_many(keyring.L, level, special, [g.prepared().ptr for g in gks], len(gks[0].key.key),
                            [g.galois_element for g in gks], ct.ptr, out.ptr, n, prepared=True)
    res = []
    for r in range(len(gks)):
        view = _View(out, r * n * 2 * sz)
        res.append(CipherText(c.params, _unpack(view, ring, n, 2, batch, primal=True, ctx=keyring.ctx), c.scale))
    if ring.ctx is not keyring.ctx:
        ring.ctx.wait_for(keyring.ctx)                     # as in keyswitch: the results live on `ring`, written on the key ring's stream
    return res


def matmul_diag(gks, diags, c: CipherText) -> CipherText:
    """diags[0] .* c + sum_k diags[k] .* rotate(gks[k-1], c): the diagonal matrix-vector product of infer.jl:140-149 /
    test/ckks_matmul.jl:33-41 with every rotation starting from `c` (one Galois key per step), in ONE device call
    (tfhe_matmul_diag).  `diags`: len(gks) + 1 plaintext ring elements of c's ring encoded at c's scale (one polynomial each:
    the same diagonal multiplies every ciphertext of the batch), or ONE batched element holding them back to back (what
    ckks_encode returns for a [len(gks) + 1][N/2] array of slot vectors).  Bit-identical to
    `CipherText.dot_plain([c] + rotate_many(gks, c), diags)`; the result is in the NTT domain, at the squared scale."""
    gks = list(gks)
    if len(c) != 2:
        raise AssertionError("rotate takes a 2-element ciphertext")
    c._need_scale()
    ring, n, batch = c[0].ring, c[0].count, c[0].batch
    level = ring.L
    if len(gks) > DOT_MAX:          # a device call takes DOT_MAX rotations: more of them as partial products (+ is exact); the
        dl = diags.split([1] * diags.count) if isinstance(diags, RingElement) else list(diags)   # later ones weigh c itself by 0
        if len(dl) != len(gks) + 1:
            raise AssertionError("matmul_diag: one diagonal for the ciphertext itself and one per rotation")
        out = matmul_diag(gks[:DOT_MAX], dl[:DOT_MAX + 1], c)
        for k in range(DOT_MAX, len(gks), DOT_MAX):
            out = out + matmul_diag(gks[k:k + DOT_MAX], [ring.zero()] + dl[k + 1:k + DOT_MAX + 1], c)
        return out
    stacked = isinstance(diags, RingElement)              # one batched element holding the len(gks) + 1 diagonals back to back
    if stacked:
        if diags.ring != ring or diags.count != len(gks) + 1:
            raise UsageError("matmul_diag: a stacked plaintext element must hold one diagonal per rotation plus one")
    else:
        diags = list(diags)
        if len(diags) != len(gks) + 1:
            raise AssertionError("matmul_diag: one diagonal for the ciphertext itself and one per rotation")
        for d in diags:
            if not isinstance(d, RingElement) or d.ring != ring or d.count != 1:
                raise UsageError("matmul_diag: the diagonals are single plaintext elements of the ciphertext's ring")
    if gks:
        params = gks[0].key.params
        if any(g.key.params is not params for g in gks) or params.relin_window != 0:
            raise UsageError("hoisted rotations need Galois keys of one parameter set with RNS digits")
        keyring = gks[0].key.key[0].mask.ring
        special = isinstance(params, ModulusRaised)
    else:
        keyring, special = ring, False
    if keyring.idx != list(range(keyring.L)) or ring.idx != list(range(level)):
        raise UsageError("ciphertext ring is not a prefix of the key ring")
    prim = [x.coeffs_primal() for x in c.cs]               # both before the hand-over (see keyswitch)
    duals = [diags.coeffs_dual()] if stacked else [d.coeffs_dual() for d in diags]
    if ring.ctx is not keyring.ctx:
        if ring.N != keyring.N or ring.moduli != keyring.moduli[:level] or ring.psi != keyring.psi[:level]:
            raise UsageError("ciphertext and key belong to different rings")
        keyring.ctx.wait_for(ring.ctx)
    sz = level * ring.N
    dg = duals[0] if stacked else _pack(duals, ring, 1, ctx=keyring.ctx)   # [R+1][level][N]
    ct = _pack(prim, ring, n, ctx=keyring.ctx)
    out = DeviceBuffer(n * 2 * sz)
    keyring.ctx.matmul_diag(keyring.L, level, special, [g.prepared().ptr for g in gks], len(gks[0].key.key) if gks else level,
                            [g.galois_element for g in gks], dg.ptr, ct.ptr, out.ptr, n)
    res = _unpack(out, ring, n, 2, batch, primal=False, ctx=keyring.ctx)
    if ring.ctx is not keyring.ctx:
        ring.ctx.wait_for(keyring.ctx)
    return CipherText(c.params, res, Fraction(c.scale) ** 2)


class _View:
    """a window into a DeviceBuffer (keeps the parent alive)"""

    def __init__(self, parent, word_offset):
        self.parent, self.ptr = parent, parent.ptr + word_offset * 8


def modswitch(c: CipherText) -> CipherText:
    """modswitch(::CipherText{CKKSEncoding}), ckksencoding.jl:127-130: rescale every component by the
    last modulus; the scale is divided by it."""
    q_last = c[0].ring.moduli[-1]
    scale = None if c.scale is None else c.scale / q_last
    return CipherText(c.params, [x.modswitch() for x in c.cs], scale)


# --------------------------------------------------------------------------------------------------
# CKKS encoding (float; ckksencoding.jl:43-97, ckks.jl:35-59) -- host side, tolerance-level parity
# --------------------------------------------------------------------------------------------------


def scale_parts(scale):
    """scale = mant * 2^exp2 with a 64-bit integer mant: exact for 2^k and for integers below 2^64 (times 2^k),
    to 2^-63 relative otherwise (the boundary type of tfhe_ckks_encode / decode)."""
    fr = Fraction(scale)
    if fr <= 0:
        raise AssertionError("scale must be positive")
    num, den = fr.numerator, fr.denominator
    if den & (den - 1) == 0:                               # dyadic: strip common powers of two
        exp2 = -(den.bit_length() - 1)
        while num % 2 == 0:
            num //= 2; exp2 += 1
        if num < 2**64:
            return num, exp2
    e = (num.bit_length() - den.bit_length()) - 63         # approximate: 63-64 significant bits
    mant = int(round(fr / Fraction(2) ** e)) if e >= 0 else int(round(fr * Fraction(2) ** (-e)))
    if mant >= 2**64:
        mant //= 2; e += 1
    return mant, e


def ckks_encode(slots, ring: NegacyclicRing, scale) -> RingElement:
    """convert(RingElement, ::CKKSEncoding), ckksencoding.jl:72-97 -- on the device (tfhe_ckks_encode).
    slots: [N/2] complex, or [batch][N/2]."""
    z = np.ascontiguousarray(np.asarray(slots, dtype=np.complex128))
    batch = None if z.ndim == 1 else z.shape[0]
    z2 = z.reshape(-1, z.shape[-1])
    if 2 * z2.shape[1] != ring.N:
        raise AssertionError("CKKS plaintexts have N/2 slots")
    if ring.idx != list(range(ring.L)):
        raise UsageError("CKKS encoding: the ring must be a prefix of its context")
    mant, exp2 = scale_parts(scale)
    dz = DeviceBuffer.from_numpy(z2.view(np.uint64))
    out = DeviceBuffer(z2.shape[0] * ring.L * ring.N)
    ring.ctx.ckks_encode(ring.L, mant, exp2, dz.ptr, out.ptr, z2.shape[0])
    return RingElement(ring, out, None, batch)             # (dz is parked by the allocator until the kernels have read it)


def ckks_decode(el: RingElement, scale) -> np.ndarray:
    """CKKSEncoding{ScaleT}(plain), ckksencoding.jl:56-66 -- on the device (tfhe_ckks_decode)."""
    ring = el.ring
    if ring.idx != list(range(ring.L)):
        raise UsageError("CKKS decoding: the ring must be a prefix of its context")
    mant, exp2 = scale_parts(scale)
    n = el.count
    out = DeviceBuffer(n * ring.N)                         # N/2 complex doubles = N words per plaintext
    ring.ctx.ckks_decode(ring.L, mant, exp2, el.coeffs_primal().ptr, out.ptr, n)
    ring.ctx.sync()
    z = out.to_numpy().view(np.complex128).reshape(n, ring.N // 2)
    return z if el.batch is not None else z[0]


# --------------------------------------------------------------------------------------------------
# on-wire format (wire.py): device <-> bytes
# --------------------------------------------------------------------------------------------------

def dump_ciphertext(c: CipherText) -> bytes:
    """CipherText -> wire blob (coefficient domain, [count][polys][L][N])."""
    from . import wire
    ring = c[0].ring
    res = np.stack([x.to_numpy("primal").reshape(x.count, ring.L, ring.N) for x in c.cs], axis=1)
    scale = scale_parts(c.scale) if c.scale is not None else (0, 0)
    return wire.dump(res, ring.moduli, ring.psi, kind=wire.KIND_CIPHERTEXT, domain=0, scale=scale, unbatched=c[0].batch is None)


def load_ciphertext(blob: bytes, params) -> CipherText:
    """wire blob -> CipherText on the device; the ring (moduli, psi) must be the scheme's ciphertext ring at that level."""
    from . import wire
    d = wire.load(blob)
    if d["kind"] != wire.KIND_CIPHERTEXT or d["domain"] != 0:
        raise UsageError("not a coefficient-domain ciphertext blob")
    ring = params.R_cipher()
    L = len(d["moduli"])
    if L != ring.L:
        ring = ring.crtselect(range(L))
    if d["N"] != ring.N or d["moduli"] != list(ring.moduli) or d["psis"] != list(ring.psi):
        raise UsageError("blob ring does not match the parameters' ciphertext ring")
    res = d["residues"]
    batch = res.shape[0]
    cs = [RingElement.from_host(ring, res[0, p] if d["unbatched"] else res[:, p]) for p in range(d["polys"])]
    smant, sexp = d["scale"]
    scale = None if smant == 0 else Fraction(smant) * Fraction(2) ** sexp
    return CipherText(params, cs, scale)
