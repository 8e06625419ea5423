"""ctypes binding of tests/emul/libemul.so (CPU emulation of the HIP kernel bodies; tests only)."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
u64p = C.POINTER(C.c_uint64)
_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        subprocess.check_call(["make", "-C", _HERE, "libemul.so"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        L = C.CDLL(os.path.join(_HERE, "libemul.so"))
        L.emul_ntt.argtypes = [C.c_int, C.c_uint64, C.c_uint64, C.c_int, C.c_int, u64p, u64p]
        L.emul_conv.restype = C.c_long
        L.emul_conv.argtypes = [u64p, C.c_int, u64p, C.c_int, C.c_int, u64p, u64p, C.c_long]
        L.emul_window_digits.argtypes = [u64p, C.c_int, C.c_int, C.c_int, u64p, u64p, C.c_long]
        L.emul_window_digits.restype = None
        L.emul_bfv.argtypes = [u64p, C.c_int, u64p, C.c_int, C.c_uint64, C.c_int, C.c_int64, u64p, u64p, C.c_long,
                               C.POINTER(C.c_long)]
        L.emul_bfv_fast.argtypes = [u64p, C.c_int, u64p, C.c_int, C.c_uint64, C.c_int, C.c_int64, u64p, u64p, C.c_long]
        L.emul_fp_max_ratio_reset.restype = C.c_double
        L.emul_ckks_round.argtypes = [C.POINTER(C.c_double), C.c_long, C.c_uint64, C.c_int, C.c_uint64, u64p]
        L.emul_ckks_round.restype = None
        L.emul_ckks_to_double.argtypes = [u64p, C.c_int, C.c_int, C.c_uint64, C.c_int]
        L.emul_ckks_to_double.restype = C.c_double
        L.emul_centred_double.argtypes = [C.c_uint64, C.c_uint64]
        L.emul_centred_double.restype = C.c_double
        _LIB = L
    return _LIB


def _p(a):
    return a.ctypes.data_as(u64p)


def ntt(a, q, psi=0, inverse=False, variant=0):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    out = np.empty_like(a)
    rc = lib().emul_ntt(int(a.size).bit_length() - 1, int(q), int(psi), int(inverse), int(variant), _p(a), _p(out))
    if rc:
        raise RuntimeError(f"emul_ntt rc={rc}")
    return out


def conv(a, t, res, centred):
    a = np.array(a, dtype=np.uint64); t = np.array(t, dtype=np.uint64)
    res = np.ascontiguousarray(res, dtype=np.uint64)
    out = np.empty((res.shape[0], len(t)), dtype=np.uint64)
    slow = lib().emul_conv(_p(a), len(a), _p(t), len(t), int(centred), _p(res), _p(out), res.shape[0])
    return out, slow


def window_digits(a, wbits, nwin, res):
    """digits [count][nwin] of the integers with residues res [count][k] over the basis a"""
    a_ = np.array(a, dtype=np.uint64)
    res = np.ascontiguousarray(res, dtype=np.uint64)
    out = np.empty((res.shape[0], nwin), dtype=np.uint64)
    lib().emul_window_digits(_p(a_), len(a), int(wbits), int(nwin), _p(res), _p(out), res.shape[0])
    return out


def bfv(qs, pb, t, src, N, contract):
    qs_a = np.array(qs, dtype=np.uint64); pb_a = np.array(pb, dtype=np.uint64)
    src = np.ascontiguousarray(src, dtype=np.uint64)
    nin, nout = (len(pb), len(qs)) if contract else (len(qs), len(pb))
    count = src.size // (nin * N)
    out = np.empty((count, nout, N), dtype=np.uint64)
    slow = C.c_long(0)
    rc = lib().emul_bfv(_p(qs_a), len(qs), _p(pb_a), len(pb), int(t), int(contract), N, _p(src), _p(out), count,
                        C.byref(slow))
    if rc:
        raise RuntimeError(f"emul_bfv rc={rc}")
    return out, slow.value


def bfv_fast(qs, pb, t, src, N, contract):
    qs_a = np.array(qs, dtype=np.uint64); pb_a = np.array(pb, dtype=np.uint64)
    src = np.ascontiguousarray(src, dtype=np.uint64)
    nin, nout = (len(pb), len(qs)) if contract else (len(qs), len(pb))
    count = src.size // (nin * N)
    out = np.empty((count, nout, N), dtype=np.uint64)
    rc = lib().emul_bfv_fast(_p(qs_a), len(qs), _p(pb_a), len(pb), int(t), int(contract), N, _p(src), _p(out), count)
    if rc:
        raise RuntimeError(f"emul_bfv_fast rc={rc}")
    return out


def fp_max_ratio_reset():
    """largest |operand| / p that entered an fp64 modular product or reduction since the last call"""
    return lib().emul_fp_max_ratio_reset()


def ckks_round(x, smant, sexp, q):
    x = np.ascontiguousarray(x, dtype=np.float64)
    out = np.empty(x.size, dtype=np.uint64)
    lib().emul_ckks_round(x.ctypes.data_as(C.POINTER(C.c_double)), x.size, int(smant), int(sexp), int(q), _p(out))
    return out


def ckks_to_double(mag: int, neg: bool, smant, sexp):
    nwords = max(1, -(-mag.bit_length() // 64))
    w = np.array([(mag >> (64 * i)) & (2**64 - 1) for i in range(nwords)], dtype=np.uint64)
    return lib().emul_ckks_to_double(_p(w), nwords, int(neg), int(smant), int(sexp))


def centred_double(r: int, q: int) -> float:
    return lib().emul_centred_double(int(r), int(q))


def out_moddown(v: float, tsp: int, cw: int, q: int, pinv_modq: int) -> int:
    f = lib().emul_out_moddown
    f.restype = C.c_uint64
    f.argtypes = [C.c_double, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64]
    return int(f(float(v), int(tsp), int(cw), int(q), int(pinv_modq)))
