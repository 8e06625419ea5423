"""ctypes binding of oracle/libref_cpu.so (the C restatement, see ref_cpu.c).

TEST INFRASTRUCTURE ONLY -- never imported by the product package.  Arrays are numpy uint64 with
layout [count][limbs][N]."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
u64p = C.POINTER(C.c_uint64)
i32p = C.POINTER(C.c_int)


def build(force: bool = False) -> str:
    so = os.path.join(_HERE, "libref_cpu.so")
    src = os.path.join(_HERE, "ref_cpu.c")
    if force or not os.path.exists(so) or (os.path.exists(src) and os.path.getmtime(so) < os.path.getmtime(src)):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libref_cpu.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        L.ref_ctx_create.restype = C.c_void_p
        L.ref_ctx_create.argtypes = [C.c_int, C.c_int, u64p, u64p]
        L.ref_ctx_destroy.argtypes = [C.c_void_p]
        L.ref_ctx_psi.argtypes = [C.c_void_p, u64p]
        for f in (L.ref_nntt, L.ref_inntt):
            f.argtypes = [C.c_void_p, i32p, C.c_int, u64p, C.c_long]
        L.ref_pointwise.argtypes = [C.c_void_p, i32p, C.c_int, C.c_int, u64p, u64p, u64p, C.c_long]
        L.ref_scalar_mul.argtypes = [C.c_void_p, i32p, C.c_int, u64p, u64p, u64p, C.c_long]
        L.ref_galois.argtypes = [C.c_void_p, i32p, C.c_int, C.c_uint64, u64p, u64p, C.c_long]
        L.ref_modswitch.argtypes = [C.c_void_p, i32p, C.c_int, u64p, u64p, C.c_long]
        L.ref_bfv_mul.argtypes = [C.c_void_p, i32p, C.c_int, C.c_void_p, i32p, C.c_int, C.c_uint64,
                                  u64p, u64p, u64p, C.c_long]
        L.ref_enc_mul.argtypes = [C.c_void_p, i32p, C.c_int, u64p, C.c_int, u64p, C.c_int, u64p, C.c_long]
        L.ref_keyswitch.argtypes = [C.c_void_p, C.c_int, C.c_int, u64p, u64p, C.c_int, u64p, C.c_long]
        L.ref_switch.argtypes = [C.c_void_p, i32p, C.c_int, C.c_void_p, i32p, C.c_int, u64p, u64p, C.c_long]
        L.ref_contract.argtypes = [C.c_void_p, i32p, C.c_int, C.c_void_p, i32p, C.c_int, C.c_uint64,
                                   u64p, u64p, C.c_long]
        L.ref_test_divrem.argtypes = [u64p, C.c_int, u64p, C.c_int, u64p, i32p, u64p, i32p]
        L.ref_num_threads.restype = C.c_int
        L.ref_set_threads.argtypes = [C.c_int]
        _LIB = L
    return _LIB


def _p(a):
    return a.ctypes.data_as(u64p)


def _idx(idx):
    arr = (C.c_int * len(idx))(*idx)
    return arr, len(idx)


def _c(a):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    return a


class RefCtx:
    """A ring context NegacyclicRing{CRTEncoded{L},N}; psis=None derives the minimal roots."""

    def __init__(self, N, qs, psis=None):
        self.N, self.qs = N, [int(q) for q in qs]
        q = np.array(self.qs, dtype=np.uint64)
        ps = np.array([0] * len(qs) if psis is None else [int(p) for p in psis], dtype=np.uint64)
        self.h = lib().ref_ctx_create(N, len(qs), _p(q), _p(ps))
        if not self.h:
            raise ValueError("ref_ctx_create failed (psi^(2N) != 1 or bad N)")
        out = np.zeros(len(qs), dtype=np.uint64)
        lib().ref_ctx_psi(self.h, _p(out))
        self.psis = [int(x) for x in out]

    @property
    def L(self):
        return len(self.qs)

    def __del__(self):
        if getattr(self, "h", None):
            lib().ref_ctx_destroy(self.h)
            self.h = None

    def _sel(self, idx):
        return list(range(self.L)) if idx is None else list(idx)

    def nntt(self, data, idx=None):
        idx = self._sel(idx); a = _c(data).copy(); ia, nl = _idx(idx)
        lib().ref_nntt(self.h, ia, nl, _p(a), a.size // (nl * self.N)); return a

    def inntt(self, data, idx=None):
        idx = self._sel(idx); a = _c(data).copy(); ia, nl = _idx(idx)
        lib().ref_inntt(self.h, ia, nl, _p(a), a.size // (nl * self.N)); return a

    def pointwise(self, op, a, b=None, idx=None):
        idx = self._sel(idx); a = _c(a); ia, nl = _idx(idx)
        out = np.empty_like(a)
        bb = _p(_c(b)) if b is not None else None
        lib().ref_pointwise(self.h, ia, nl, {"add": 0, "sub": 1, "mul": 2, "neg": 3}[op], _p(a), bb, _p(out),
                            a.size // (nl * self.N))
        return out

    def scalar_mul(self, scal, a, idx=None):
        idx = self._sel(idx); a = _c(a); ia, nl = _idx(idx)
        s = np.array([int(x) for x in scal], dtype=np.uint64); out = np.empty_like(a)
        lib().ref_scalar_mul(self.h, ia, nl, _p(s), _p(a), _p(out), a.size // (nl * self.N)); return out

    def galois(self, g, a, idx=None):
        idx = self._sel(idx); a = _c(a); ia, nl = _idx(idx); out = np.empty_like(a)
        lib().ref_galois(self.h, ia, nl, int(g), _p(a), _p(out), a.size // (nl * self.N)); return out

    def modswitch(self, a, idx=None):
        idx = self._sel(idx); a = _c(a); ia, nl = _idx(idx)
        cnt = a.size // (nl * self.N)
        out = np.empty((cnt, nl - 1, self.N), dtype=np.uint64)
        lib().ref_modswitch(self.h, ia, nl, _p(a), _p(out), cnt); return out

    def enc_mul(self, c1, c2, idx=None):
        idx = self._sel(idx); ia, nl = _idx(idx)
        c1, c2 = _c(c1), _c(c2)  # [batch][n][nl][N]
        batch, n1, n2 = c1.shape[0], c1.shape[1], c2.shape[1]
        out = np.empty((batch, n1 + n2 - 1, nl, self.N), dtype=np.uint64)
        lib().ref_enc_mul(self.h, ia, nl, _p(c1), n1, _p(c2), n2, _p(out), batch); return out

    def keyswitch(self, l, special, evk, ct):
        evk, ct = _c(evk), _c(ct)  # evk [ndig][2][Lk][N] NTT domain; ct [batch][np][l][N]
        batch, npol = ct.shape[0], ct.shape[1]
        out = np.empty((batch, 2, l, self.N), dtype=np.uint64)
        lib().ref_keyswitch(self.h, l, int(bool(special)), _p(evk), _p(ct), npol, _p(out), batch); return out


def bfv_mul(small: RefCtx, big: RefCtx, t, c1, c2, idx_s=None, idx_b=None):
    idx_s, idx_b = small._sel(idx_s), big._sel(idx_b)
    ia, ns = _idx(idx_s); ib, nb = _idx(idx_b)
    c1, c2 = _c(c1), _c(c2)
    batch = c1.shape[0]
    out = np.empty((batch, 3, ns, small.N), dtype=np.uint64)
    lib().ref_bfv_mul(small.h, ia, ns, big.h, ib, nb, int(t), _p(c1), _p(c2), _p(out), batch)
    return out


def switch(src_ctx: RefCtx, dst_ctx: RefCtx, a, idx_s=None, idx_d=None):
    idx_s, idx_d = src_ctx._sel(idx_s), dst_ctx._sel(idx_d)
    ia, ns = _idx(idx_s); ib, nd = _idx(idx_d)
    a = _c(a); cnt = a.size // (ns * src_ctx.N)
    out = np.empty((cnt, nd, src_ctx.N), dtype=np.uint64)
    lib().ref_switch(src_ctx.h, ia, ns, dst_ctx.h, ib, nd, _p(a), _p(out), cnt); return out


def contract(big: RefCtx, small: RefCtx, t, a, idx_b=None, idx_s=None):
    idx_b, idx_s = big._sel(idx_b), small._sel(idx_s)
    ib, nb = _idx(idx_b); ia, ns = _idx(idx_s)
    a = _c(a); cnt = a.size // (nb * big.N)
    out = np.empty((cnt, ns, big.N), dtype=np.uint64)
    lib().ref_contract(big.h, ib, nb, small.h, ia, ns, int(t), _p(a), _p(out), cnt); return out


def divrem(u: int, v: int):
    def words(x):
        w = []
        while x:
            w.append(x & (2**64 - 1)); x >>= 64
        return np.array(w or [0], dtype=np.uint64)
    uw, vw = words(u), words(v)
    q = np.zeros(64, dtype=np.uint64); r = np.zeros(64, dtype=np.uint64)
    qn, rn = C.c_int(0), C.c_int(0)
    lib().ref_test_divrem(_p(uw), len(uw), _p(vw), len(vw), _p(q), C.byref(qn), _p(r), C.byref(rn))
    toint = lambda a, n: sum(int(a[i]) << (64 * i) for i in range(n))
    return toint(q, qn.value), toint(r, rn.value)
