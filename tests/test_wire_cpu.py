"""On-wire format (toyfhe.jl_amd/wire.py): pure-numpy round trips and rejection of malformed blobs."""
import numpy as np
import pytest

import toyfhe_jl_amd as tf
from toyfhe_jl_amd import wire
from tests import helpers as H


def test_wire_round_trip_and_layout():
    N, qs = 64, H.chain(50, 3, 64)
    psis = [3, 5, 7]
    rng = np.random.default_rng(1)
    res = H.rand_residues(rng, qs, (4, 2), N)                        # [count][polys][L][N]
    blob = wire.dump(res, qs, psis, kind=wire.KIND_CIPHERTEXT, domain=0, scale=(1, 40))
    assert blob[:8] == b"TFHEWIRE" and len(blob) == 56 + 16 * 3 + res.size * 8
    # the payload is the device layout verbatim, little-endian
    assert np.array_equal(np.frombuffer(blob[56 + 48:], dtype="<u8").reshape(res.shape), res)
    d = wire.load(blob)
    assert d["N"] == N and d["moduli"] == qs and d["psis"] == psis and d["polys"] == 2 and d["count"] == 4
    assert d["scale"] == (1, 40) and d["domain"] == 0 and d["kind"] == wire.KIND_CIPHERTEXT
    assert np.array_equal(d["residues"], res)
    # single (unbatched) ciphertext and a key blob
    one = wire.load(wire.dump(res[0], qs, psis))
    assert one["count"] == 1 and np.array_equal(one["residues"][0], res[0])
    key = wire.load(wire.dump(res, qs, psis, kind=wire.KIND_KEY, domain=1, relin_window=3))
    assert key["kind"] == wire.KIND_KEY and key["domain"] == 1 and key["relin_window"] == 3


def test_wire_rejects_malformed():
    N, qs = 16, H.chain(40, 2, 16)
    res = H.rand_residues(np.random.default_rng(2), qs, (1, 2), N)
    blob = wire.dump(res, qs, [1, 1])
    with pytest.raises(wire.WireError):
        wire.load(blob[:-8])                                         # truncated payload
    with pytest.raises(wire.WireError):
        wire.load(b"XXXXXXXX" + blob[8:])                            # magic
    with pytest.raises(wire.WireError):
        wire.load(blob[:8] + (2).to_bytes(4, "little") + blob[12:])  # version
    bad = bytearray(blob)
    bad[-8:] = (qs[1]).to_bytes(8, "little")                         # residue == modulus
    with pytest.raises(wire.WireError):
        wire.load(bytes(bad))
    with pytest.raises(wire.WireError):
        wire.dump(res, qs[:1], [1])                                  # ring / shape mismatch
    res[0, 0, 0, 0] = qs[0]
    with pytest.raises(wire.WireError):
        wire.dump(res, qs, [1, 1])
    with pytest.raises(wire.WireError):
        wire.load(b"short")
