"""The C oracle against the committed golden fixtures (tests/golden/golden_v1.npz, generated from the
spec oracle by tests/golden/make_golden.py at the reference's own test parameter sets)."""
import os

import numpy as np
import pytest

from oracle import ref_cpu

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "golden_v1.npz"))


def test_doc_vectors_97():
    ctx = ref_cpu.RefCtx(4, [97])
    assert ctx.psis == [33]                       # rlwe.md:186
    x = G["doc97_in"].reshape(4, 1, 4)
    nt = ctx.nntt(x)
    assert np.array_equal(nt.reshape(4, 4), G["doc97_ntt"])
    prods = [(2, 3), (0, 0), (0, 1)]              # p3*p4, p1^2, p1*p2 (rlwe.md:207-212)
    for (i, j), want in zip(prods, G["doc97_prod"]):
        got = ctx.inntt(ctx.pointwise("mul", nt[i:i + 1], nt[j:j + 1]))
        assert np.array_equal(got.reshape(4), want)


@pytest.mark.parametrize("name,N", [("n16", 16), ("n32", 32), ("n2048", 2048), ("pal", 2048)])
def test_ntt_vectors(name, N):
    ctx = ref_cpu.RefCtx(N, G[f"{name}_q"], G[f"{name}_psi"])
    assert np.array_equal(ctx.nntt(G[f"{name}_in"]), G[f"{name}_ntt"])
    assert np.array_equal(ctx.inntt(G[f"{name}_ntt"]), G[f"{name}_in"])


def test_modswitch_galois_vectors():
    ctx = ref_cpu.RefCtx(32, G["ms_q"])
    assert np.array_equal(ctx.modswitch(G["ms_in"]), G["ms_out"])
    for g in (3, 5, 63, pow(3, 15, 64)):
        assert np.array_equal(ctx.galois(g, G["ms_in"]), G[f"gal{g}_out"])


def test_keyswitch_vectors():
    ctx = ref_cpu.RefCtx(32, G["ksS_q"])
    assert np.array_equal(ctx.keyswitch(2, True, G["ksS_evk_ntt"], G["ksS_ct"]), G["ksS_out"])
    assert np.array_equal(ctx.keyswitch(3, False, G["ksR_evk_ntt"], G["ksR_ct"]), G["ksR_out"])


def test_bfv_vectors():
    q = G["bfvcrt_q"]
    cs, cb = ref_cpu.RefCtx(2048, q[:2]), ref_cpu.RefCtx(2048, q[2:])
    ct = G["bfvcrt_ct"][None]
    assert np.array_equal(ref_cpu.bfv_mul(cs, cb, int(G["bfvcrt_t"][0]), ct, ct)[0], G["bfvcrt_prod"])
    q = G["bfvsup_q"]
    cs, cb = ref_cpu.RefCtx(64, q[:3]), ref_cpu.RefCtx(64, q)
    got = ref_cpu.bfv_mul(cs, cb, int(G["bfvsup_t"][0]), G["bfvsup_c1"][None], G["bfvsup_c2"][None])
    assert np.array_equal(got[0], G["bfvsup_prod"])
