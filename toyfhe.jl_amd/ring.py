"""Host-side mirror of ToyFHE's ``NegacyclicRing`` / ``RingElement`` over device storage.

This is the Python stand-in for the Julia shim (``julia/ToyFHEHIP.jl``; no Julia toolchain exists in the
build image): same names, argument meaning and error behaviour as ``src/pow2_cyc_rings.jl`` and the RNS
hooks of ``src/crt.jl``, with every polynomial held in device memory and every operation executed by
libtoyfhe_hip.so through the C ABI.  Nothing here computes on the CPU except the per-coefficient
``getindex``/``setindex!`` conveniences and host<->device conversion, which download/upload -- and ``PlainRing``, the
reference's psi = 0 plaintext rings (naive convolution, "just for plaintexts and testing", pow2_cyc_rings.jl:150-165), which
have no transform and stay on the host by design.

A ``RingElement`` may carry a leading batch dimension (a batch of independent ring elements sharing one
ring): that is how batches of ciphertexts reach the batched kernels.
"""
from __future__ import annotations

import math

import numpy as np

from . import native
from .native import Context, DeviceBuffer, UsageError  # noqa: F401

# --------------------------------------------------------------------------------------------------
# Primes.jl stand-ins used by the ring constructors (crt.jl:282-295, test/*.jl prime chains)
# --------------------------------------------------------------------------------------------------
_MR_BASES = (2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37)


def isprime(n: int) -> bool:
    if n < 2:
        return False
    for p in _MR_BASES:
        if n % p == 0:
            return n == p
    d, s = n - 1, 0
    while d % 2 == 0:
        d //= 2
        s += 1
    for a in _MR_BASES:
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
    """Primes.nextprime(n, i; interval)."""
    p, found = n, 0
    while True:
        if isprime(p):
            found += 1
            if found == i:
                return p
        p += interval


class NegacyclicRing:
    """ℤ_q[x]/(x^N + 1) in RNS form with an identified 2N-th root per limb.

    Mirrors ``NegacyclicRing{CRTEncoded{L,...},N}(ψ)`` (pow2_cyc_rings.jl:27-65) and the RNS constructor
    ``NegacyclicRing(N, logqs)`` (crt.jl:282-295).  A single-modulus ring is L = 1.  Rings obtained by
    ``crtselect`` / ``drop_last`` (crt.jl:185-213) share the parent's device context.
    """

    def __init__(self, N: int, moduli, psi=None, *, _ctx: Context | None = None, _idx=None):
        if _ctx is None:
            _ctx = Context(N, moduli, psi)          # raises AssertionError like pow2_cyc_rings.jl:31
            _idx = list(range(len(moduli)))
        self.ctx, self.idx = _ctx, list(_idx)
        self.N = int(N)

    @classmethod
    def from_logqs(cls, N: int, logqs) -> "NegacyclicRing":
        """NegacyclicRing(N, logqs), crt.jl:282-295."""
        perm = sorted(range(len(logqs)), key=lambda i: logqs[i])
        primes, lastp = [0] * len(logqs), 0
        for i in perm:
            lastp = nextprime(max(2 ** logqs[i] + 1, lastp + 2 * N), 1, 2 * N)
            primes[i] = lastp
        return cls(N, primes)

    # -- accessors (pow2_cyc_rings.jl:48-65, crt.jl:80-85) --
    @property
    def moduli(self):
        return [self.ctx.qs[i] for i in self.idx]

    @property
    def psi(self):
        return [self.ctx.psis[i] for i in self.idx]

    @property
    def L(self):
        return len(self.idx)

    def degree(self):
        return self.N

    def modulus(self) -> int:
        return math.prod(self.moduli)

    def __eq__(self, o):
        return isinstance(o, NegacyclicRing) and (self.N, self.moduli, self.psi) == (o.N, o.moduli, o.psi)

    def __hash__(self):
        return hash((self.N, tuple(self.moduli), tuple(self.psi)))

    def __repr__(self):  # Base.show, pow2_cyc_rings.jl:69-75
        return f"ℤ_{self.modulus()}/(x^{self.N} + 1)"

    def crtselect(self, which) -> "NegacyclicRing":
        """crtselect(ℛ, which), crt.jl:189-192 (0-based positions)."""
        return NegacyclicRing(self.N, None, _ctx=self.ctx, _idx=[self.idx[i] for i in which])

    def drop_last(self) -> "NegacyclicRing":
        return self.crtselect(range(self.L - 1))

    # -- element constructors (pow2_cyc_rings.jl:77-85) --
    def __call__(self, coeffs) -> "RingElement":
        """ℛ(coeffs): integer coefficients (any sign/size) of one element [N] or a batch [B][N], or residues
        already in RNS form as a uint64 array [..., L, N]."""
        return RingElement.from_host(self, coeffs)

    def from_residues(self, residues, dual=False) -> "RingElement":
        return RingElement.from_residues(self, residues, dual=dual)

    def zero(self, batch=None) -> "RingElement":
        n = (batch or 1) * self.L * self.N
        buf = DeviceBuffer(n)
        native.check(native.lib().tfhe_memset(self.ctx.h, buf.ptr, 0, n * 8))
        return RingElement(self, primal=buf, batch=batch)


class RingElement:
    """Element(s) of a NegacyclicRing with lazily cached primal (coefficient) and dual (NTT) forms --
    the semantics of the mutable struct at pow2_cyc_rings.jl:93-145."""

    def __init__(self, ring: NegacyclicRing, primal: DeviceBuffer | None = None, dual: DeviceBuffer | None = None, batch=None):
        assert primal is not None or dual is not None      # pow2_cyc_rings.jl:116
        self.ring, self.primal, self.dual, self.batch = ring, primal, dual, batch

    # -- sizes --
    @property
    def count(self):
        return self.batch or 1

    @property
    def _words(self):
        return self.count * self.ring.L * self.ring.N

    def _new(self):
        return DeviceBuffer(self._words)

    # -- host conversion --
    @classmethod
    def from_host(cls, ring, coeffs, dual=False) -> "RingElement":
        a = coeffs
        L, N = ring.L, ring.N
        if isinstance(a, np.ndarray) and a.dtype == np.uint64 and a.shape[-2:] == (L, N):
            # a uint64 array is ALWAYS taken as RNS residues [..., L, N] (see from_residues); integer coefficients go in as
            # Python ints or signed arrays.  The kernels assume reduced inputs, so check them here.
            res = a
            batch = None if a.ndim == 2 else int(np.prod(a.shape[:-2]))
            for l, q in enumerate(ring.moduli):
                if res.size and int(res[..., l, :].max()) >= q:
                    raise AssertionError(f"residue out of range on limb {l} (must be < {q})")
        else:
            rows = [list(a)] if not hasattr(a[0], "__len__") else [list(r) for r in a]
            batch = None if not hasattr(a[0], "__len__") else len(rows)
            res = np.empty((len(rows), L, N), dtype=np.uint64)
            for b, r in enumerate(rows):
                assert len(r) == N
                for l, q in enumerate(ring.moduli):          # CRTEncoded{N,M}(x::Integer), crt.jl:91-95
                    res[b, l] = [int(x) % q for x in r]
        buf = DeviceBuffer.from_numpy(res)
        return cls(ring, dual=buf, batch=batch) if dual else cls(ring, primal=buf, batch=batch)

    @classmethod
    def from_residues(cls, ring, residues, dual=False) -> "RingElement":
        """explicit residue path: `residues` uint64 [L][N] (one element) or [B][L][N] (a batch), each limb reduced"""
        a = np.ascontiguousarray(residues, dtype=np.uint64)
        if a.shape[-2:] != (ring.L, ring.N) or a.ndim not in (2, 3):
            raise AssertionError(f"residues must have shape [{ring.L}][{ring.N}] or [B][{ring.L}][{ring.N}]")
        return cls.from_host(ring, a, dual=dual)

    def to_numpy(self, domain="primal") -> np.ndarray:
        buf = self.coeffs_primal() if domain == "primal" else self.coeffs_dual()
        shape = (self.ring.L, self.ring.N) if self.batch is None else (self.batch, self.ring.L, self.ring.N)
        self.ring.ctx.sync()        # the one host wait of the device path (a caller's stream may be non-blocking)
        return buf.to_numpy(shape)

    def to_ints(self):
        """convert(Integer, ::CRTEncoded) per coefficient (crt.jl:98-112): exact CRT into [0, Q)."""
        res = self.to_numpy().reshape(self.count, self.ring.L, self.ring.N)
        qs = self.ring.moduli
        Q = math.prod(qs)
        cof = [(Q // q) * pow(Q // q, -1, q) for q in qs]
        out = [[sum(int(res[b, l, i]) * cof[l] for l in range(len(qs))) % Q for i in range(self.ring.N)]
               for b in range(self.count)]
        return out[0] if self.batch is None else out

    # -- lazy domains (pow2_cyc_rings.jl:124-138) --
    def coeffs_primal(self) -> DeviceBuffer:
        if self.primal is None:
            self.primal = self._new()
            self.ring.ctx.inntt(self.dual.ptr, self.primal.ptr, self.count, self.ring.L, self.ring.idx)
        return self.primal

    def coeffs_dual(self) -> DeviceBuffer:
        if self.dual is None:
            self.dual = self._new()
            self.ring.ctx.nntt(self.primal.ptr, self.dual.ptr, self.count, self.ring.L, self.ring.idx)
        return self.dual

    def __getitem__(self, i):  # Base.getindex, pow2_cyc_rings.jl:140
        assert self.batch is None
        return self.to_ints()[i]

    def __setitem__(self, i, v):  # Base.setindex!, pow2_cyc_rings.jl:141-145 (invalidates the dual)
        assert self.batch is None
        res = self.to_numpy()
        for l, q in enumerate(self.ring.moduli):
            res[l, i] = int(v) % q
        self.primal = DeviceBuffer.from_numpy(res)
        self.dual = None

    # -- arithmetic --
    def _check(self, o):
        if not isinstance(o, RingElement) or o.ring != self.ring:
            raise UsageError("ring elements belong to different rings")
        if self.batch is not None and o.batch is not None and o.batch != self.batch:
            raise UsageError("ring element batches differ")

    def broadcast_to(self, batch: int) -> "RingElement":
        """one ring element against a batch (the scalar-broadcast patterns of rlwe_she.jl:143, ckksencoding.jl:99-124):
        the present domain(s) repeated `batch` times on the device."""
        assert self.batch is None
        if batch == 1:                                   # a batch of one shares the element's buffers
            return RingElement(self.ring, self.primal, self.dual, 1)
        words = self.ring.L * self.ring.N
        def rep(b):
            if b is None:
                return None
            out = DeviceBuffer(batch * words)
            native.check(native.lib().tfhe_broadcast_poly(self.ring.ctx.h, out.ptr, b.ptr, words, batch))
            return out
        return RingElement(self.ring, rep(self.primal), rep(self.dual), batch)

    @staticmethod
    def concat(els) -> "RingElement":
        """the batches of several elements of one ring back to back (an unbatched element counts as a batch of one): independent
        polynomials ride in the leading batch dimension of every device call.  The result holds the domain(s) ALL inputs hold."""
        els = list(els)
        ring = els[0].ring
        assert all(e.ring == ring for e in els)
        words = ring.L * ring.N
        total = sum(e.count for e in els)
        def cat(get):
            if any(get(e) is None for e in els):
                return None
            out = DeviceBuffer(total * words)
            off = 0
            for e in els:
                native.check(native.lib().tfhe_memcpy_d2d(ring.ctx.h, out.ptr + off * 8, get(e).ptr, e.count * words * 8))
                off += e.count * words
            return out
        p, d = cat(lambda e: e.primal), cat(lambda e: e.dual)
        if p is None and d is None:                      # mixed domains: bring everything to the coefficient domain
            for e in els:
                e.coeffs_primal()
            p = cat(lambda e: e.primal)
        return RingElement(ring, p, d, total)

    def split(self, sizes) -> "list[RingElement]":
        """inverse of concat: consecutive sub-batches of the given sizes (copies)"""
        sizes = [int(x) for x in sizes]
        assert sum(sizes) == self.count
        words = self.ring.L * self.ring.N
        out, off = [], 0
        for sz in sizes:
            def part(b):
                if b is None:
                    return None
                o = DeviceBuffer(sz * words)
                native.check(native.lib().tfhe_memcpy_d2d(self.ring.ctx.h, o.ptr, b.ptr + off * 8, sz * words * 8))
                return o
            out.append(RingElement(self.ring, part(self.primal), part(self.dual), sz))
            off += sz * words
        return out

    def _align(self, o):
        """(a, b) with equal batch: an unbatched operand is broadcast against a batched one"""
        self._check(o)
        if self.batch == o.batch:
            return self, o
        return (self.broadcast_to(o.batch), o) if self.batch is None else (self, o.broadcast_to(self.batch))

    def _binary(self, o, op):
        """+ and - , pow2_cyc_rings.jl:192-219: operate in whichever domain(s) both operands have; if the
        domains are disjoint compute both."""
        self, o = self._align(o)
        ctx, L, idx, n = self.ring.ctx, self.ring.L, self.ring.idx, self.count
        f = ctx.add if op == "+" else ctx.sub
        new_p = new_d = None
        if self.primal is not None and o.primal is not None:
            new_p = self._new(); f(self.primal.ptr, o.primal.ptr, new_p.ptr, n, L, idx)
        if self.dual is not None and o.dual is not None:
            new_d = self._new(); f(self.dual.ptr, o.dual.ptr, new_d.ptr, n, L, idx)
        if new_p is None and new_d is None:
            new_p = self._new(); f(self.coeffs_primal().ptr, o.coeffs_primal().ptr, new_p.ptr, n, L, idx)
            new_d = self._new(); f(self.coeffs_dual().ptr, o.coeffs_dual().ptr, new_d.ptr, n, L, idx)
        return RingElement(self.ring, new_p, new_d, self.batch)

    def __add__(self, o):
        return self._binary(o, "+")

    def __sub__(self, o):
        return self._binary(o, "-")

    def __neg__(self):  # pow2_cyc_rings.jl:187-190
        ctx, L, idx, n = self.ring.ctx, self.ring.L, self.ring.idx, self.count
        p = d = None
        if self.primal is not None:
            p = self._new(); ctx.neg(self.primal.ptr, p.ptr, n, L, idx)
        if self.dual is not None:
            d = self._new(); ctx.neg(self.dual.ptr, d.ptr, n, L, idx)
        return RingElement(self.ring, p, d, self.batch)

    def __mul__(self, o):
        if isinstance(o, int):                       # scalar_mul, pow2_cyc_rings.jl:177-185
            ctx, L, idx, n = self.ring.ctx, self.ring.L, self.ring.idx, self.count
            scal = [o % q for q in self.ring.moduli]
            p = d = None
            if self.primal is not None:
                p = self._new(); ctx.scalar_mul(scal, self.primal.ptr, p.ptr, n, L, idx)
            if self.dual is not None:
                d = self._new(); ctx.scalar_mul(scal, self.dual.ptr, d.ptr, n, L, idx)
            return RingElement(self.ring, p, d, self.batch)
        self, o = self._align(o)                     # ring_multiply, pow2_cyc_rings.jl:147-173: dual-only result
        out = self._new()
        self.ring.ctx.mul(self.coeffs_dual().ptr, o.coeffs_dual().ptr, out.ptr, self.count, self.ring.L, self.ring.idx)
        return RingElement(self.ring, None, out, self.batch)

    __rmul__ = __mul__

    def __pow__(self, n: int):  # pow2_cyc_rings.jl:221-224
        assert n >= 1
        r, base = None, self
        while n:
            if n & 1:
                r = base if r is None else r * base
            n >>= 1
            if n:
                base = base * base
        return r

    # -- structure maps --
    def apply_galois_element(self, g: int) -> "RingElement":
        """pow2_cyc_rings.jl:321-329 (coefficient domain; result is primal-only)."""
        out = self._new()
        self.ring.ctx.galois(self.coeffs_primal().ptr, out.ptr, g, self.count, self.ring.L, self.ring.idx)
        return RingElement(self.ring, out, None, self.batch)

    def modswitch(self) -> "RingElement":
        """modswitch(::RingElement), crt.jl:226-228: rescale by the last modulus; primal-only result."""
        if self.ring.L < 2:
            raise UsageError("modswitch needs at least two CRT moduli")
        new = self.ring.drop_last()
        out = DeviceBuffer(self.count * new.L * self.ring.N)
        self.ring.ctx.rescale(self.coeffs_primal().ptr, out.ptr, self.count, self.ring.L, self.ring.idx)
        return RingElement(new, out, None, self.batch)

    def crtselect(self, which) -> "RingElement":
        """crtselect(x, which), crt.jl:199-211: keeps whichever of primal/dual are present."""
        new = self.ring.crtselect(which)
        which = list(which)
        p = d = None
        if self.primal is not None:
            p = DeviceBuffer(self.count * new.L * self.ring.N)
            self.ring.ctx.select_limbs(self.primal.ptr, p.ptr, self.count, self.ring.L, which)
        if self.dual is not None:
            d = DeviceBuffer(self.count * new.L * self.ring.N)
            self.ring.ctx.select_limbs(self.dual.ptr, d.ptr, self.count, self.ring.L, which)
        return RingElement(new, p, d, self.batch)

    def modswitch_drop(self) -> "RingElement":
        """modswitch_drop(::RingElement), crt.jl:230-232 (primal-only, last limb dropped)."""
        new = self.ring.drop_last()
        out = DeviceBuffer(self.count * new.L * self.ring.N)
        self.ring.ctx.select_limbs(self.coeffs_primal().ptr, out.ptr, self.count, self.ring.L, list(range(new.L)))
        return RingElement(new, out, None, self.batch)

    def copy(self):
        def dup(b):
            if b is None:
                return None
            n = self._new()
            native.check(native.lib().tfhe_memcpy_d2d(self.ring.ctx.h, n.ptr, b.ptr, self._words * 8))
            return n
        return RingElement(self.ring, dup(self.primal), dup(self.dual), self.batch)


def zero(x):
    """Base.zero(r::RingElement) / zero(ℛ), pow2_cyc_rings.jl:83-85,121-122."""
    return x.ring.zero(x.batch) if isinstance(x, RingElement) else x.zero()


# --------------------------------------------------------------------------------------------------
# psi = 0 rings: NegacyclicRing{coefft, N}() (pow2_cyc_rings.jl:39-41) -- the plaintext spaces of BFV / BGV whose modulus has no
# 2N-th root of unity (plaintext_space, rlwe_she.jl:380-392).  The reference multiplies them by the naive negacyclic convolution
# (pow2_cyc_rings.jl:150-165, "just for plaintexts and testing"); so does this mirror, on the host: there is nothing to accelerate.
# --------------------------------------------------------------------------------------------------
class PlainRing:
    """Z_t[x]/(x^N + 1) without a root of unity (psi = 0).  ``ring(coeffs)`` / ``ring.zero()`` make elements."""

    def __init__(self, N: int, t: int):
        if N < 1 or N & (N - 1):
            raise AssertionError("degree must be a power of two")          # pow2_cyc_rings.jl:29
        if t < 2:
            raise AssertionError("plaintext modulus must be at least 2")
        self.N, self.t = int(N), int(t)

    psi = 0

    def degree(self):
        return self.N

    def modulus(self) -> int:
        return self.t

    def __eq__(self, o):
        return isinstance(o, PlainRing) and (self.N, self.t) == (o.N, o.t)

    def __hash__(self):
        return hash((self.N, self.t, 0))

    def __repr__(self):
        return f"ℤ_{self.t}/(x^{self.N} + 1)"

    def __call__(self, coeffs) -> "PlainElement":
        return PlainElement(self, coeffs)

    def zero(self) -> "PlainElement":
        return PlainElement(self, [0] * self.N)


class PlainElement:
    """RingElement of a psi = 0 ring: primal coefficients only (Python integers mod t), 0-based like the reference's OffsetArray."""

    def __init__(self, ring: PlainRing, coeffs):
        coeffs = list(coeffs)
        if len(coeffs) > ring.N:
            raise AssertionError("more coefficients than the degree")
        self.ring = ring
        self.c = [int(x) % ring.t for x in coeffs] + [0] * (ring.N - len(coeffs))

    def __len__(self):
        return self.ring.N

    def __iter__(self):
        return iter(self.c)

    def __getitem__(self, i):
        return self.c[i]

    def __setitem__(self, i, v):
        self.c[i] = int(v) % self.ring.t

    def to_ints(self):
        return list(self.c)

    def _same(self, o):
        if not isinstance(o, PlainElement) or o.ring != self.ring:
            raise UsageError("operands belong to different rings")

    def __eq__(self, o):
        if isinstance(o, PlainElement):
            return self.ring == o.ring and self.c == o.c
        try:
            o = list(o)
        except TypeError:
            return NotImplemented
        return len(o) == self.ring.N and all(int(a) % self.ring.t == b for a, b in zip(o, self.c))

    __hash__ = None

    def __add__(self, o):
        self._same(o)
        return PlainElement(self.ring, [a + b for a, b in zip(self.c, o.c)])

    def __sub__(self, o):
        self._same(o)
        return PlainElement(self.ring, [a - b for a, b in zip(self.c, o.c)])

    def __neg__(self):
        return PlainElement(self.ring, [-a for a in self.c])

    def __mul__(self, o):
        if isinstance(o, (int, np.integer)):                                   # scalar_mul, pow2_cyc_rings.jl:177-180
            return PlainElement(self.ring, [int(o) * a for a in self.c])
        self._same(o)
        N, t = self.ring.N, self.ring.t
        # naive negacyclic convolution (pow2_cyc_rings.jl:150-165): res[i+j] += a_i b_j, wrapped with a sign change past x^N
        if N * t * t < 2 ** 62:
            full = np.convolve(np.array(self.c, dtype=np.int64), np.array(o.c, dtype=np.int64))
        else:
            full = np.convolve(np.array(self.c, dtype=object), np.array(o.c, dtype=object))
        res = [int(x) for x in full[:N]]
        for k in range(N, 2 * N - 1):
            res[k - N] -= int(full[k])
        return PlainElement(self.ring, res)

    __rmul__ = __mul__

    def __pow__(self, e: int):                                                 # Base.:^ by repeated multiplication (:182-186)
        if e < 1:
            raise UsageError("positive powers only")
        r = self
        for _ in range(e - 1):
            r = r * self
        return r


def plaintext_space(ring, t: int):
    """plaintext_space(ℛ, p), rlwe_she.jl:380-392: a prime p > 2N gets the ring with its minimal 2N-th root (device storage, NTT
    products -- what SlotEncoding needs); anything else the psi = 0 ring.  (The reference leaves the existence of the root as a
    TODO and fails in minimal_primitive_root when it does not exist; here such a prime falls back to psi = 0.)"""
    N = ring.N if hasattr(ring, "N") else int(ring)
    if isprime(t) and t > 2 * N and (t - 1) % (2 * N) == 0:
        return NegacyclicRing(N, [t])
    return PlainRing(N, t)
