"""ctypes binding of libtoyfhe_hip.so (include/toyfhe_hip.h).

This is the only way the package computes anything: there is no CPU fallback.  If the HIP library
is missing or no device is usable, every entry point raises -- loudly.
"""
from __future__ import annotations

import ctypes as C
import os
import shutil
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libtoyfhe_hip.so")
_SRC = os.path.join(_HERE, "csrc")
_LIB = None

u64p = C.POINTER(C.c_uint64)
i32p = C.POINTER(C.c_int32)

OK, E_BADARG, E_DOMAIN, E_LEVEL, E_PARAMS, E_NOMEM, E_HIP, E_UNSUPPORTED = 0, -1, -2, -3, -4, -5, -6, -7


class UsageError(Exception):
    """Mirror of ToyFHE.UsageError (src/rlwe_she.jl:223-225): operands with differing parameters."""


class HipError(RuntimeError):
    pass


def build(force: bool = False) -> str:
    """Compile csrc/toyfhe_hip.hip for gfx950 into libtoyfhe_hip.so (in-tree)."""
    srcs = [os.path.join(_SRC, f) for f in os.listdir(_SRC)] + [os.path.join(_HERE, "..", "include", "toyfhe_hip.h")]
    if not force and os.path.exists(_SO) and all(os.path.getmtime(_SO) >= os.path.getmtime(s) for s in srcs):
        return _SO
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-Wno-unused-value", "-shared", "-fPIC",
           os.path.join(_SRC, "toyfhe_hip.hip"), "-o", _SO]
    subprocess.check_call(cmd)
    return _SO


def _one_hip_runtime():
    """One HIP runtime per process.  A PyTorch-ROCm wheel bundles its own libamdhip64.so and loads it by file name, so an
    engine that was loaded first (against /opt/rocm's copy, same SONAME) would leave the process with two runtimes whose
    streams and allocations do not mix (torch streams passed to tfhe_ctx_set_stream, RCCL from torch/lib).  If such a wheel
    is installed and not yet imported, load ITS runtime first: the engine's NEEDED libamdhip64.so.7 then resolves to it, and
    a later `import torch` shares it.  TFHE_USE_SYSTEM_HIP=1 skips this."""
    import importlib.util
    import sys
    if "torch" in sys.modules or os.environ.get("TFHE_USE_SYSTEM_HIP", "0") not in ("", "0"):
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    so = os.environ.get("TFHE_HIP_LIB") or _SO   # TFHE_HIP_LIB: another build of the same engine (A/B measurements)
    if not os.path.exists(so):
        raise HipError(f"{so} is missing: the HIP engine was not built "
                       "(run `python -c 'import __graft_entry__ as g; g.build()'`). There is no CPU fallback.")
    _one_hip_runtime()
    L = C.CDLL(so)
    L.tfhe_last_error.restype = C.c_char_p
    vp, i64, i32, u64, sz = C.c_void_p, C.c_int64, C.c_int, C.c_uint64, C.c_size_t
    sig = {
        "tfhe_device_count": [C.POINTER(C.c_int)],
        "tfhe_set_device": [i32],
        "tfhe_ctx_create": [i64, i32, u64p, u64p, C.POINTER(vp)],
        "tfhe_ctx_destroy": [vp],
        "tfhe_ctx_psi": [vp, u64p],
        "tfhe_ctx_set_stream": [vp, vp],
        "tfhe_ctx_sync": [vp],
        "tfhe_ctx_wait_for": [vp, vp],
        "tfhe_ctx_set_ntt_variant": [vp, i32],
        "tfhe_malloc": [sz, C.POINTER(vp)],
        "tfhe_free": [vp],
        "tfhe_memcpy_h2d": [vp, vp, sz],
        "tfhe_memcpy_d2h": [vp, vp, sz],
        "tfhe_memcpy_d2d": [vp, vp, vp, sz],
        "tfhe_memset": [vp, vp, i32, sz],
        "tfhe_pack_poly": [vp, vp, vp, i32, i32, sz, i64],
        "tfhe_unpack_poly": [vp, vp, vp, i32, i32, sz, i64],
        "tfhe_broadcast_poly": [vp, vp, vp, sz, i64],
        "tfhe_alloc_stats": [u64p, u64p, u64p, u64p],
        "tfhe_alloc_trim": [],
        "tfhe_comm_id": [vp],
        "tfhe_comm_create": [vp, i32, i32, C.POINTER(vp)],
        "tfhe_comm_destroy": [vp],
        "tfhe_gather": [vp, vp, vp, vp, sz],
        "tfhe_nntt": [vp, vp, vp, i64, i32, i32p],
        "tfhe_inntt": [vp, vp, vp, i64, i32, i32p],
        "tfhe_add": [vp, vp, vp, vp, i64, i32, i32p],
        "tfhe_sub": [vp, vp, vp, vp, i64, i32, i32p],
        "tfhe_neg": [vp, vp, vp, i64, i32, i32p],
        "tfhe_mul": [vp, vp, vp, vp, i64, i32, i32p],
        "tfhe_mad": [vp, vp, vp, vp, vp, i64, i32, i32p],
        "tfhe_dot": [vp, vp, C.POINTER(vp), C.POINTER(vp), i32, vp, i64, i32, i32p],
        "tfhe_lincomb": [vp, u64p, C.POINTER(vp), i32, vp, i64, i32, i32p],
        "tfhe_scalar_mul": [vp, u64p, vp, vp, i64, i32, i32p],
        "tfhe_tensor": [vp, vp, vp, vp, i64, i32, i32p],
        "tfhe_rescale": [vp, vp, vp, i64, i32, i32p],
        "tfhe_select_limbs": [vp, vp, vp, i64, i32, i32p, i32],
        "tfhe_galois": [vp, vp, vp, u64, i64, i32, i32p],
        "tfhe_keyswitch": [vp, i32, i32, i32, vp, i32, vp, i32, vp, i64],
        "tfhe_rotate": [vp, i32, i32, i32, vp, i32, u64, vp, vp, i64],
        "tfhe_rotate_prepared": [vp, i32, i32, i32, vp, i32, u64, vp, vp, i64],
        "tfhe_keyswitch_window": [vp, i32, i32, i32, i32, vp, i32, vp, i32, vp, i64],
        "tfhe_rotate_many": [vp, i32, i32, i32, C.POINTER(vp), i32, i32, u64p, i32, vp, vp, i64],
        "tfhe_galois_key_prepare": [vp, i32, i32, u64, vp, vp],
        "tfhe_matmul_diag": [vp, i32, i32, i32, C.POINTER(vp), i32, u64p, i32, vp, vp, vp, i64],
        "tfhe_lincomb_many": [vp, u64p, C.POINTER(vp), i32, C.POINTER(vp), i32, i64, i32, i32p],
        "tfhe_sample_uniform": [vp, i32, u64, C.c_uint32, u64, vp, i64],
        "tfhe_sample_gaussian": [vp, i32, C.c_double, u64, u64, C.c_uint32, u64, vp, i64],
        "tfhe_ckks_encode": [vp, i32, u64, i32, vp, vp, i64],
        "tfhe_ckks_decode": [vp, i32, u64, i32, vp, vp, i64],
        "tfhe_bfv_plan_create": [vp, i32p, i32, vp, i32p, i32, u64, C.POINTER(vp)],
        "tfhe_bfv_plan_destroy": [vp],
        "tfhe_bfv_plan_set_chunk": [vp, i32],
        "tfhe_bfv_plan_set_variant": [vp, i32],
        "tfhe_bfv_mul": [vp, vp, vp, vp, i64],
        "tfhe_bfv_expand": [vp, vp, vp, i64],
        "tfhe_bfv_contract": [vp, vp, vp, i64],
        "tfhe_bfv_mul_relin": [vp, vp, i32, vp, vp, vp, i64],
        "tfhe_prof_enable": [vp, i32],
        "tfhe_prof_read": [vp, C.POINTER(i64), C.POINTER(i64), C.POINTER(C.c_double)],
        "tfhe_event_create": [C.POINTER(vp)],
        "tfhe_event_destroy": [vp],
        "tfhe_event_record": [vp, vp],
        "tfhe_event_elapsed_ms": [vp, vp, C.POINTER(C.c_float)],
    }
    for name, args in sig.items():
        f = getattr(L, name)
        f.argtypes = args
        f.restype = C.c_int
    _LIB = L
    return L


EXPORTED_SYMBOLS = [
    "tfhe_last_error", "tfhe_device_count", "tfhe_set_device", "tfhe_ctx_create", "tfhe_ctx_destroy", "tfhe_ctx_psi",
    "tfhe_ctx_set_stream", "tfhe_ctx_sync", "tfhe_ctx_wait_for", "tfhe_ctx_set_ntt_variant", "tfhe_malloc", "tfhe_free", "tfhe_memcpy_h2d",
    "tfhe_memcpy_d2h", "tfhe_memcpy_d2d", "tfhe_memset", "tfhe_pack_poly", "tfhe_unpack_poly", "tfhe_broadcast_poly", "tfhe_alloc_stats", "tfhe_alloc_trim", "tfhe_comm_id", "tfhe_comm_create", "tfhe_comm_destroy", "tfhe_gather", "tfhe_nntt", "tfhe_inntt", "tfhe_add", "tfhe_sub", "tfhe_neg",
    "tfhe_mul", "tfhe_mad", "tfhe_dot", "tfhe_scalar_mul", "tfhe_tensor", "tfhe_rescale", "tfhe_select_limbs", "tfhe_galois",
    "tfhe_keyswitch", "tfhe_rotate", "tfhe_rotate_prepared", "tfhe_rotate_many", "tfhe_galois_key_prepare", "tfhe_matmul_diag", "tfhe_lincomb", "tfhe_lincomb_many", "tfhe_keyswitch_window", "tfhe_ckks_encode", "tfhe_ckks_decode", "tfhe_sample_uniform", "tfhe_sample_gaussian", "tfhe_bfv_plan_create", "tfhe_bfv_plan_destroy", "tfhe_bfv_plan_set_chunk",
    "tfhe_bfv_plan_set_variant", "tfhe_bfv_mul", "tfhe_bfv_expand", "tfhe_bfv_contract", "tfhe_bfv_mul_relin", "tfhe_prof_enable", "tfhe_prof_read",
    "tfhe_event_create", "tfhe_event_destroy", "tfhe_event_record", "tfhe_event_elapsed_ms",
]


def check(rc: int):
    """Map a tfhe_status to the exception class the reference raises for the same condition."""
    if rc == OK:
        return
    msg = lib().tfhe_last_error().decode("utf-8", "replace")
    if rc == E_BADARG:
        raise AssertionError(msg)            # @assert (pow2_cyc_rings.jl:31,61,116; rlwe_she.jl:318)
    if rc in (E_PARAMS, E_LEVEL):
        raise UsageError(msg)                # UsageError (rlwe_she.jl:223-225,233-235,248-250)
    if rc == E_UNSUPPORTED:
        raise NotImplementedError(msg)       # error("... only implemented ...") (crt.jl:270,274)
    if rc == E_NOMEM:
        raise MemoryError(msg)
    raise HipError(msg)


def alloc_stats() -> dict:
    """live / cached bytes and hipMalloc / reuse counts of the library's recycling allocator (csrc/dev_alloc.h)"""
    v = [C.c_uint64(0) for _ in range(4)]
    check(lib().tfhe_alloc_stats(*[C.byref(x) for x in v]))
    return dict(zip(("live_bytes", "cached_bytes", "hip_mallocs", "reuses"), (x.value for x in v)))


def device_count() -> int:
    n = C.c_int(0)
    rc = lib().tfhe_device_count(C.byref(n))
    return n.value if rc == OK else 0


def _idx(idx):
    if idx is None:
        return None
    return (C.c_int32 * len(idx))(*[int(i) for i in idx])


class DeviceBuffer:
    """A flat device allocation of uint64 residues (owned by the library's allocator)."""

    def __init__(self, n_words: int):
        self.n = int(n_words)
        p = C.c_void_p()
        check(lib().tfhe_malloc(self.n * 8, C.byref(p)))
        self.ptr = p.value

    @classmethod
    def from_numpy(cls, a) -> "DeviceBuffer":
        a = np.ascontiguousarray(a, dtype=np.uint64)
        b = cls(a.size)
        check(lib().tfhe_memcpy_h2d(b.ptr, a.ctypes.data, a.size * 8))
        return b

    def to_numpy(self, shape=None) -> np.ndarray:
        out = np.empty(self.n, dtype=np.uint64)
        check(lib().tfhe_memcpy_d2h(out.ctypes.data, self.ptr, self.n * 8))
        return out.reshape(shape) if shape is not None else out

    def free(self):
        if getattr(self, "ptr", None):
            lib().tfhe_free(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Context:
    """NegacyclicRing{CRTEncoded{L,...},N} on the device (tfhe_ctx)."""

    def __init__(self, N: int, qs, psis=None):
        self.N, self.qs = int(N), [int(q) for q in qs]
        q = (C.c_uint64 * len(qs))(*self.qs)
        ps = (C.c_uint64 * len(qs))(*([0] * len(qs) if psis is None else [int(p) for p in psis]))
        h = C.c_void_p()
        check(lib().tfhe_ctx_create(self.N, len(self.qs), q, ps, C.byref(h)))
        self.h = h.value
        out = (C.c_uint64 * len(qs))()
        check(lib().tfhe_ctx_psi(self.h, out))
        self.psis = [int(x) for x in out]

    @property
    def L(self):
        return len(self.qs)

    def close(self):
        if getattr(self, "h", None):
            lib().tfhe_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def sync(self):
        check(lib().tfhe_ctx_sync(self.h))

    def wait_for(self, producer: "Context"):
        """Order this context's later work after everything already submitted to `producer` (device-side, no host wait)."""
        check(lib().tfhe_ctx_wait_for(self.h, producer.h))

    def set_stream(self, stream_ptr):
        check(lib().tfhe_ctx_set_stream(self.h, stream_ptr))

    def set_ntt_variant(self, v):
        check(lib().tfhe_ctx_set_ntt_variant(self.h, int(v)))

    # ---- raw ops on device pointers (ints) ----
    def nntt(self, src, dst, count, limbs, idx=None):
        check(lib().tfhe_nntt(self.h, src, dst, count, limbs, _idx(idx)))

    def inntt(self, src, dst, count, limbs, idx=None):
        check(lib().tfhe_inntt(self.h, src, dst, count, limbs, _idx(idx)))

    def add(self, a, b, dst, count, limbs, idx=None):
        check(lib().tfhe_add(self.h, a, b, dst, count, limbs, _idx(idx)))

    def sub(self, a, b, dst, count, limbs, idx=None):
        check(lib().tfhe_sub(self.h, a, b, dst, count, limbs, _idx(idx)))

    def neg(self, a, dst, count, limbs, idx=None):
        check(lib().tfhe_neg(self.h, a, dst, count, limbs, _idx(idx)))

    def mul(self, a, b, dst, count, limbs, idx=None):
        check(lib().tfhe_mul(self.h, a, b, dst, count, limbs, _idx(idx)))

    def mad(self, acc, a, b, dst, count, limbs, idx=None):
        check(lib().tfhe_mad(self.h, acc, a, b, dst, count, limbs, _idx(idx)))

    def dot(self, acc, a_ptrs, b_ptrs, dst, count, limbs, idx=None):
        """dst = (acc +) sum_k a_k .* b_k (tfhe_dot); a_ptrs / b_ptrs: equally long lists of device pointers"""
        if len(a_ptrs) != len(b_ptrs):
            raise AssertionError("tfhe_dot: as many a operands as b operands")
        n = len(a_ptrs)
        A = (C.c_void_p * n)(*a_ptrs)
        B = (C.c_void_p * n)(*b_ptrs)
        check(lib().tfhe_dot(self.h, acc, A, B, n, dst, count, limbs, _idx(idx)))

    def lincomb(self, scalars, a_ptrs, dst, count, limbs, idx=None):
        """dst = sum_k scalars[k] * a_k (tfhe_lincomb); scalars: [n_terms][limbs] residues, a_ptrs: device pointers"""
        n = len(a_ptrs)
        flat = [int(x) for row in scalars for x in row]
        if len(flat) != n * limbs:
            raise AssertionError("tfhe_lincomb: one scalar per term and limb")
        S = (C.c_uint64 * len(flat))(*flat)
        A = (C.c_void_p * n)(*a_ptrs)
        check(lib().tfhe_lincomb(self.h, S, A, n, dst, count, limbs, _idx(idx)))

    def lincomb_many(self, scalars, a_ptrs, dst_ptrs, count, limbs, idx=None):
        """dst[o] = sum_k scalars[o][k] * a_k for every output o in one pass over the operands (tfhe_lincomb_many);
        scalars: [n_out][n_terms][limbs] residues"""
        n, no = len(a_ptrs), len(dst_ptrs)
        flat = [int(x) for out in scalars for row in out for x in row]
        if len(flat) != no * n * limbs:
            raise AssertionError("tfhe_lincomb_many: one scalar per output, term and limb")
        S = (C.c_uint64 * len(flat))(*flat)
        A = (C.c_void_p * n)(*a_ptrs)
        D = (C.c_void_p * no)(*dst_ptrs)
        check(lib().tfhe_lincomb_many(self.h, S, A, n, D, no, count, limbs, _idx(idx)))

    def matmul_diag(self, key_limbs, level, special, evks, n_digits, gs, diags, ct, out, batch):
        ptrs = (C.c_void_p * max(1, len(evks)))(*[int(p) for p in evks])
        garr = (C.c_uint64 * max(1, len(gs)))(*[int(g) for g in gs])
        check(lib().tfhe_matmul_diag(self.h, key_limbs, level, int(bool(special)), ptrs, n_digits, garr, len(evks), diags, ct, out, batch))

    def scalar_mul(self, scal, a, dst, count, limbs, idx=None):
        s = (C.c_uint64 * limbs)(*[int(x) for x in scal])
        check(lib().tfhe_scalar_mul(self.h, s, a, dst, count, limbs, _idx(idx)))

    def tensor(self, a, b, out, batch, limbs, idx=None):
        check(lib().tfhe_tensor(self.h, a, b, out, batch, limbs, _idx(idx)))

    def rescale(self, src, dst, count, limbs, idx=None):
        check(lib().tfhe_rescale(self.h, src, dst, count, limbs, _idx(idx)))

    def select_limbs(self, src, dst, count, src_limbs, which):
        check(lib().tfhe_select_limbs(self.h, src, dst, count, src_limbs, _idx(which), len(which)))

    def galois(self, src, dst, g, count, limbs, idx=None):
        check(lib().tfhe_galois(self.h, src, dst, int(g), count, limbs, _idx(idx)))

    def keyswitch(self, key_limbs, level, special, evk, n_digits, ct, polys, out, batch):
        check(lib().tfhe_keyswitch(self.h, key_limbs, level, int(bool(special)), evk, n_digits, ct, polys, out, batch))

    def keyswitch_window(self, level, window_bits, evk, n_windows, ct, polys, out, batch, key_limbs=None, special=False):
        check(lib().tfhe_keyswitch_window(self.h, level if key_limbs is None else key_limbs, level, int(bool(special)), window_bits,
                                          evk, n_windows, ct, polys, out, batch))

    def sample_uniform(self, level, seed, stream, first_poly, out, count):
        check(lib().tfhe_sample_uniform(self.h, level, int(seed), int(stream), int(first_poly), out, count))

    def sample_gaussian(self, level, sigma, multiplier, seed, stream, first_poly, out, count):
        check(lib().tfhe_sample_gaussian(self.h, level, float(sigma), int(multiplier), int(seed), int(stream), int(first_poly), out, count))

    def ckks_encode(self, level, scale_mant, scale_exp2, slots, out, batch):
        check(lib().tfhe_ckks_encode(self.h, level, int(scale_mant), int(scale_exp2), slots, out, batch))

    def ckks_decode(self, level, scale_mant, scale_exp2, src, slots, batch):
        check(lib().tfhe_ckks_decode(self.h, level, int(scale_mant), int(scale_exp2), src, slots, batch))

    def rotate(self, key_limbs, level, special, evk, n_digits, g, ct, out, batch, prepared=False):
        f = lib().tfhe_rotate_prepared if prepared else lib().tfhe_rotate
        check(f(self.h, key_limbs, level, int(bool(special)), evk, n_digits, int(g), ct, out, batch))

    def rotate_many(self, key_limbs, level, special, evks, n_digits, gs, ct, out, batch, prepared=False):
        ptrs = (C.c_void_p * len(evks))(*[int(p) for p in evks])
        garr = (C.c_uint64 * len(gs))(*[int(g) for g in gs])
        check(lib().tfhe_rotate_many(self.h, key_limbs, level, int(bool(special)), ptrs, n_digits, int(bool(prepared)), garr, len(evks),
                                     ct, out, batch))

    def galois_key_prepare(self, key_limbs, n_digits, g, evk, evk_out):
        check(lib().tfhe_galois_key_prepare(self.h, key_limbs, n_digits, int(g), evk, evk_out))

    def prof_enable(self, on=True):
        check(lib().tfhe_prof_enable(self.h, int(on)))

    def prof_read(self):
        a, b, ms = C.c_int64(0), C.c_int64(0), C.c_double(0)
        check(lib().tfhe_prof_read(self.h, C.byref(a), C.byref(b), C.byref(ms)))
        return a.value, b.value, ms.value


class BfvPlan:
    """(ℛ, ℛbig, t) of a BFVParams (src/bfv.jl:5-19) with the exact-conversion tables on the device."""

    def __init__(self, small: Context, big: Context, t: int, idx_s=None, idx_b=None):
        self.small, self.big, self.t = small, big, int(t)
        idx_s = list(range(small.L)) if idx_s is None else list(idx_s)
        idx_b = list(range(big.L)) if idx_b is None else list(idx_b)
        self.ns, self.nb = len(idx_s), len(idx_b)
        h = C.c_void_p()
        check(lib().tfhe_bfv_plan_create(small.h, _idx(idx_s), self.ns, big.h, _idx(idx_b), self.nb, self.t, C.byref(h)))
        self.h = h.value

    def close(self):
        if getattr(self, "h", None):
            lib().tfhe_bfv_plan_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_chunk(self, n):
        check(lib().tfhe_bfv_plan_set_chunk(self.h, int(n)))

    def set_variant(self, v):
        check(lib().tfhe_bfv_plan_set_variant(self.h, int(v)))

    def mul(self, c1, c2, out, batch):
        check(lib().tfhe_bfv_mul(self.h, c1, c2, out, batch))

    def expand(self, src, dst, count):
        check(lib().tfhe_bfv_expand(self.h, src, dst, count))

    def contract(self, src, dst, count):
        check(lib().tfhe_bfv_contract(self.h, src, dst, count))

    def mul_relin(self, evk, n_digits, c1, c2, out, batch):
        check(lib().tfhe_bfv_mul_relin(self.h, evk, n_digits, c1, c2, out, batch))


class Comm:
    """The ranks of a multi-GPU job for the final gather (tfhe_comm over RCCL).  `exchange(id_bytes_or_None) -> id_bytes` is
    the host-side broadcast of rank 0's rendezvous id (torch.distributed, MPI, a file ...)."""

    def __init__(self, nranks: int, rank: int, exchange):
        buf = C.create_string_buffer(128)
        if rank == 0:
            check(lib().tfhe_comm_id(buf))
        data = exchange(bytes(buf.raw) if rank == 0 else None)
        h = C.c_void_p()
        check(lib().tfhe_comm_create(C.create_string_buffer(data, 128), nranks, rank, C.byref(h)))
        self.h, self.nranks, self.rank = h.value, nranks, rank

    def gather(self, ctx: "Context", src_ptr: int, dst_ptr: int, words_per_rank: int):
        check(lib().tfhe_gather(self.h, ctx.h, src_ptr, dst_ptr, words_per_rank))

    def close(self):
        if getattr(self, "h", None):
            lib().tfhe_comm_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Event:
    def __init__(self):
        p = C.c_void_p()
        check(lib().tfhe_event_create(C.byref(p)))
        self.p = p.value

    def record(self, ctx: Context):
        check(lib().tfhe_event_record(ctx.h, self.p))

    def elapsed_ms(self, stop: "Event") -> float:
        ms = C.c_float(0)
        check(lib().tfhe_event_elapsed_ms(self.p, stop.p, C.byref(ms)))
        return ms.value

    def __del__(self):
        try:
            if self.p:
                lib().tfhe_event_destroy(self.p)
        except Exception:
            pass
