"""On-wire format for ring elements, ciphertexts and evaluation keys.

The reference has no serialisation; its in-memory layout is the `StructArray` SoA of `src/crt.jl:150-156`
(one contiguous length-N array per limb).  The wire image is exactly the device layout, so a blob can be
`tfhe_memcpy_h2d`'d without any reshuffle:

    offset  size  field
    0       8     magic  b"TFHEWIRE"
    8       4     version (= 1)                      little-endian throughout
    12      4     kind (low 8 bits): 1 ring elements / ciphertext, 2 key-switch key; bit 8: a single unbatched element (count = 1)
    16      4     log2(N)
    20      4     L (limbs)
    24      4     polys (ciphertext components; key: 2 = (mask, masked))
    28      4     domain: 0 coefficient (primal), 1 NTT (dual, natural order, psi below)
    32      8     count (batch of ciphertexts; key: number of digits / windows)
    40      8     scale mantissa  (CKKS; 0 = no scale)     scale = mantissa * 2^exp2
    48      4     scale exp2 (signed)
    52      4     relin_window (keys; 0 = RNS digits)
    56      8*L   moduli q_l
    ...     8*L   psi_l (primitive 2N-th roots the NTT domain refers to)
    ...     8*count*polys*L*N   residues, u64, [count][polys][L][N]

Pure numpy: usable on any host; `she.py` adds the device round trips."""
import struct

import numpy as np

MAGIC = b"TFHEWIRE"
VERSION = 1
KIND_CIPHERTEXT, KIND_KEY = 1, 2
FLAG_UNBATCHED = 0x100
_HDR = struct.Struct("<8sIIIIIIQQiI")


class WireError(ValueError):
    pass


def dump(residues, moduli, psis, *, kind=KIND_CIPHERTEXT, domain=0, scale=(0, 0), relin_window=0, unbatched=False) -> bytes:
    """residues: uint64 array [count][polys][L][N] (or [polys][L][N] for count = 1)."""
    a = np.ascontiguousarray(residues, dtype="<u8")
    if a.ndim == 3:
        a = a[None]
    if a.ndim != 4:
        raise WireError("residues must be [count][polys][L][N]")
    count, polys, L, N = a.shape
    logn = int(N).bit_length() - 1
    if N != 1 << logn or L != len(moduli) or L != len(psis):
        raise WireError("shape does not match the ring")
    for l, q in enumerate(moduli):
        if a[:, :, l, :].max(initial=0) >= q:
            raise WireError(f"residue out of range in limb {l}")
    if unbatched and count != 1:
        raise WireError("an unbatched element has count 1")
    hdr = _HDR.pack(MAGIC, VERSION, kind | (FLAG_UNBATCHED if unbatched else 0), logn, L, polys, domain, count, int(scale[0]), int(scale[1]), relin_window)
    return hdr + np.asarray(moduli, dtype="<u8").tobytes() + np.asarray(psis, dtype="<u8").tobytes() + a.tobytes()


def load(blob: bytes):
    """-> dict(kind, N, moduli, psis, polys, domain, count, scale, relin_window, unbatched, residues[count][polys][L][N])"""
    if len(blob) < _HDR.size:
        raise WireError("truncated header")
    magic, ver, kind, logn, L, polys, domain, count, smant, sexp, window = _HDR.unpack_from(blob, 0)
    if magic != MAGIC:
        raise WireError("bad magic")
    if ver != VERSION:
        raise WireError(f"unsupported version {ver}")
    unbatched, kind = bool(kind & FLAG_UNBATCHED), kind & ~FLAG_UNBATCHED
    if unbatched and count != 1:
        raise WireError("bad header field")
    if kind not in (KIND_CIPHERTEXT, KIND_KEY) or domain not in (0, 1) or not (1 <= logn <= 17) or L < 1 or polys < 1:
        raise WireError("bad header field")
    N = 1 << logn
    off = _HDR.size
    need = off + 16 * L + 8 * count * polys * L * N
    if len(blob) != need:
        raise WireError(f"size mismatch: {len(blob)} bytes, header implies {need}")
    moduli = [int(x) for x in np.frombuffer(blob, dtype="<u8", count=L, offset=off)]
    psis = [int(x) for x in np.frombuffer(blob, dtype="<u8", count=L, offset=off + 8 * L)]
    res = np.frombuffer(blob, dtype="<u8", count=count * polys * L * N, offset=off + 16 * L).reshape(count, polys, L, N)
    for l, q in enumerate(moduli):
        if res[:, :, l, :].max(initial=0) >= q:
            raise WireError(f"residue out of range in limb {l}")
    return {"kind": kind, "N": N, "moduli": moduli, "psis": psis, "polys": polys, "domain": domain, "count": count,
            "scale": (smant, sexp), "relin_window": window, "unbatched": unbatched, "residues": res.astype(np.uint64)}
