"""GPU parity tests: the HIP engine, called through the C ABI (libtoyfhe_hip.so), against the oracle on
the same seeded inputs -- bit-exact (integer work) -- plus the committed golden fixtures and
size-independent properties at the BASELINE.json size (N = 2^14, L = 8)."""
import os

import numpy as np
import pytest

import toyfhe_jl_amd as tf
from oracle import ref_cpu, spec
from tests import helpers as H

pytestmark = pytest.mark.gpu

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "golden_v1.npz"))


def dev(a):
    return tf.DeviceBuffer.from_numpy(a)


def run_ntt(ctx, a, inverse=False, idx=None):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    limbs = a.shape[-2]
    count = a.size // (limbs * ctx.N)
    d = dev(a)
    (ctx.inntt if inverse else ctx.nntt)(d.ptr, d.ptr, count, limbs, idx)
    return d.to_numpy(a.shape)


# ---------------------------------------------------------------------------------------------------
# K1/K2
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("logn", [1, 2, 4, 5, 9, 10, 11, 12, 13, 14])
@pytest.mark.parametrize("bits,nq", [(40, 3), (50, 2), (61, 1)])
def test_nntt_inntt_match_oracle(logn, bits, nq):
    N = 1 << logn
    qs = H.chain(bits, nq, N)
    rng = np.random.default_rng(logn * 7 + bits)
    a = H.rand_residues(rng, qs, (5,), N)
    a[0, :, 0] = np.array(qs, dtype=np.uint64) - 1
    ref = ref_cpu.RefCtx(N, qs)
    ctx = tf.Context(N, qs)
    assert ctx.psis == ref.psis
    want = ref.nntt(a)
    for variant in ((0, 1, 2, 3) if logn >= 10 else (0,)):   # 1-3: cross-check kernel families (include/toyfhe_hip.h)
        ctx.set_ntt_variant(variant)
        got = run_ntt(ctx, a)
        assert np.array_equal(got, want), (logn, bits, variant)
        assert np.array_equal(run_ntt(ctx, want, inverse=True), a), (logn, bits, variant)


@pytest.mark.parametrize("rows", [1, 255, 3 * 256 + 17])
def test_nntt_many_rows_per_workgroup(rows):
    """N = 2^14 with more rows than compute units: every workgroup walks several rows, which is where the staged
    inverse kernel copies the next row into LDS underneath the last pass (and where the tail of the walk is ragged)."""
    N = 1 << 14
    qs = H.chain(50, 3, N)
    rng = np.random.default_rng(rows)
    ref1, ref3 = ref_cpu.RefCtx(N, qs[:1]), ref_cpu.RefCtx(N, qs)
    ctx = tf.Context(N, qs)
    a = H.rand_residues(rng, qs[:1], (rows,), N)                    # rows x 1 limb, all on modulus 0
    want = ref1.nntt(a)
    for variant in (0, 3):
        ctx.set_ntt_variant(variant)
        assert np.array_equal(run_ntt(ctx, a, idx=[0]), want), variant
        assert np.array_equal(run_ntt(ctx, want, inverse=True, idx=[0]), a), variant
    ctx.set_ntt_variant(0)
    if rows >= 3:                                                   # rows cycling through three moduli
        b = H.rand_residues(rng, qs, (rows // 3,), N)
        wantb = ref3.nntt(b)
        assert np.array_equal(run_ntt(ctx, b), wantb)
        assert np.array_equal(run_ntt(ctx, wantb, inverse=True), b)


def test_keyswitch_more_digit_rows_than_compute_units():
    """Key switch at N = 2^14 with batch * level > 256: the read-once digit-lift kernel and the staged inverse
    (addend mode) both walk several items per workgroup."""
    N, Lk, level, batch = 1 << 14, 3, 3, 90
    qs = H.chain(50, Lk, N)
    ref = ref_cpu.RefCtx(N, qs); ctx = tf.Context(N, qs)
    rng = np.random.default_rng(90)
    evk = H.uniform_evk(rng, qs, Lk, N)
    ct = H.rand_residues(rng, qs, (batch, 2), N)
    devk, dct, dout = dev(evk), dev(ct), tf.DeviceBuffer(batch * 2 * level * N)
    want = ref.keyswitch(level, False, evk, ct)
    for variant in (0, 3):
        ctx.set_ntt_variant(variant)
        ctx.keyswitch(Lk, level, False, devk.ptr, Lk, dct.ptr, 2, dout.ptr, batch)
        assert np.array_equal(dout.to_numpy(want.shape), want), variant


@pytest.mark.parametrize("logn", [15, 16])
def test_nntt_large_n(logn):
    N = 1 << logn
    qs = H.chain(50, 2, N)
    rng = np.random.default_rng(logn)
    a = H.rand_residues(rng, qs, (3,), N)
    ref = ref_cpu.RefCtx(N, qs); ctx = tf.Context(N, qs)
    want = ref.nntt(a)
    d_in, d_out = dev(a), tf.DeviceBuffer(a.size)
    ctx.nntt(d_in.ptr, d_out.ptr, 3, 2)                # out of place
    assert np.array_equal(d_out.to_numpy(a.shape), want)
    assert np.array_equal(d_in.to_numpy(a.shape), a)   # source untouched
    assert np.array_equal(run_ntt(ctx, a), want)       # in place
    assert np.array_equal(run_ntt(ctx, want, inverse=True), a)


def test_nntt_2_15_many_rows_single_kernel():
    """N = 2^15 with more rows than compute units: the one-kernel path (top stage in registers, both 2^14 sub-blocks per
    workgroup) walks several rows per workgroup; variant 2 (u64 blocks + separate top-stage kernel) must agree."""
    N, rows = 1 << 15, 2 * 256 + 7
    qs = H.chain(50, 2, N)
    rng = np.random.default_rng(15)
    a = H.rand_residues(rng, qs[:1], (rows,), N)
    a[0, 0, :4] = [0, 1, qs[0] - 1, qs[0] // 2]
    want = ref_cpu.RefCtx(N, qs[:1]).nntt(a)
    ctx = tf.Context(N, qs)
    for variant in (0, 2):
        ctx.set_ntt_variant(variant)
        assert np.array_equal(run_ntt(ctx, a, idx=[0]), want), variant
        assert np.array_equal(run_ntt(ctx, want, inverse=True, idx=[0]), a), variant


@pytest.mark.parametrize("logn,rows", [(16, 2 * 256 + 7), (17, 256 + 5)])
def test_nntt_2_16_17_many_rows_subblock_walk(logn, rows):
    """N >= 2^16 with more rows than compute units: the sub-block kernels walk their items XCD-aware and (fp64 policy)
    take two sub-blocks per workgroup with 16-byte pieces on the natural-order side; every row must still land where
    the oracle puts it, and the u64 one-sub-block path (variant 2) must agree."""
    N = 1 << logn
    qs = H.chain(50, 2, N)
    rng = np.random.default_rng(logn)
    a = H.rand_residues(rng, qs, (rows,), N)        # rows alternate over both limbs inside the kernels
    a[0, 0, :4] = [0, 1, qs[0] - 1, qs[0] // 2]
    want = ref_cpu.RefCtx(N, qs).nntt(a)
    ctx = tf.Context(N, qs)
    for variant in (0, 2):
        ctx.set_ntt_variant(variant)
        assert np.array_equal(run_ntt(ctx, a), want), variant
        assert np.array_equal(run_ntt(ctx, want, inverse=True), a), variant
    # out of place (N = 2^16: the one-kernel forward path, which reads the source row from two workgroups)
    ctx.set_ntt_variant(0)
    d_in, d_out = dev(a), tf.DeviceBuffer(a.size)
    ctx.nntt(d_in.ptr, d_out.ptr, rows, 2)
    assert np.array_equal(d_out.to_numpy(a.shape), want)
    assert np.array_equal(d_in.to_numpy(a.shape), a)
    ctx.inntt(d_out.ptr, d_in.ptr, rows, 2)
    assert np.array_equal(d_in.to_numpy(a.shape), a)
    # rows that start 8 bytes off a 16-byte boundary: the paired kernels do not apply, the result must not change
    few = a[:3]
    pad = tf.DeviceBuffer.from_numpy(np.concatenate([[0], few.ravel(), [0]]).astype(np.uint64))
    ctx.nntt(pad.ptr + 8, pad.ptr + 8, 3, 2)
    assert np.array_equal(pad.to_numpy((few.size + 2,))[1:-1].reshape(few.shape), want[:3])
    ctx.inntt(pad.ptr + 8, pad.ptr + 8, 3, 2)
    assert np.array_equal(pad.to_numpy((few.size + 2,))[1:-1].reshape(few.shape), few)


@pytest.mark.parametrize("logn", [10, 12, 14, 15, 16, 17])
def test_nntt_mixed_modulus_sizes(logn):
    """A ring that mixes a 60-bit q0 / special prime with 40-bit primes (infer.jl:97-112): the 40-bit limbs go through the
    fp64 kernels and the 60-bit ones through the u64 kernels in two passes over the same rows (ntt_io_t::limb_mask; N > 2^14:
    every kernel of a pass, top stages included, skips the other policy's limbs);
    more rows than compute units, in place and out of place, and the forced-u64 path (variant 2) must agree."""
    N = 1 << logn
    qs = H.chain(60, 1, N) + H.chain(40, 3, N) + H.chain(61, 1, N)
    rows = max(2 * 256 // len(qs) + 3 if logn in (14, 15, 16) else 7, (1 << 20) // (len(qs) * N) + 1)   # enough words for the two-pass mode
    rng = np.random.default_rng(logn)
    a = H.rand_residues(rng, qs, (rows,), N)
    a[0, :, 0] = np.array(qs, dtype=np.uint64) - 1
    ref = ref_cpu.RefCtx(N, qs); ctx = tf.Context(N, qs)
    want = ref.nntt(a)
    for variant in (0, 2):
        ctx.set_ntt_variant(variant)
        assert np.array_equal(run_ntt(ctx, a), want), variant
        assert np.array_equal(run_ntt(ctx, want, inverse=True), a), variant
    ctx.set_ntt_variant(0)
    d_in, d_out = dev(a), tf.DeviceBuffer(a.size)
    ctx.nntt(d_in.ptr, d_out.ptr, rows, len(qs))
    assert np.array_equal(d_out.to_numpy(a.shape), want)
    sub = [3, 0, 2]                                   # a selection in another order, still mixed
    b = H.rand_residues(rng, [qs[i] for i in sub], (max(5, (1 << 20) // (3 * N) + 1),), N)
    assert np.array_equal(run_ntt(ctx, b, idx=sub), ref.nntt(b, sub))


def test_nntt_limb_selection_and_explicit_psi():
    N = 2048
    q, psi = 1152921504606830593, 811032584449645127    # cryptparams.jl:25
    qs = H.chain(50, 3, N) + [q]
    ref = ref_cpu.RefCtx(N, qs, [0, 0, 0, psi]); ctx = tf.Context(N, qs, [0, 0, 0, psi])
    assert ctx.psis[3] == psi
    rng = np.random.default_rng(3)
    idx = [3, 1]
    a = H.rand_residues(rng, [qs[i] for i in idx], (4,), N)
    assert np.array_equal(run_ntt(ctx, a, idx=idx), ref.nntt(a, idx))
    assert np.array_equal(run_ntt(ctx, G["pal_in"], idx=[3]), G["pal_ntt"])
    with pytest.raises(AssertionError):
        tf.Context(N, [q], [3])                          # psi^(2N) != 1 (pow2_cyc_rings.jl:31)
    with pytest.raises(tf.UsageError):
        run_ntt(ctx, a, idx=[0, 7])                      # limb outside the ring


def test_golden_ntt_vectors():
    for name, N in (("doc97", 4), ("n16", 16), ("n32", 32), ("n2048", 2048)):
        ctx = tf.Context(N, G[f"{name}_q"], G[f"{name}_psi"])
        x = G[f"{name}_in"]
        x = x.reshape(-1, 1, N) if name == "doc97" else x
        want = G[f"{name}_ntt"].reshape(x.shape)
        assert np.array_equal(run_ntt(ctx, x), want), name
        assert np.array_equal(run_ntt(ctx, want, inverse=True), x), name
    # rlwe.md:207-212 products through the device
    ctx = tf.Context(4, [97])
    assert ctx.psis == [33]
    nt = dev(G["doc97_ntt"])
    for (i, j), want in zip([(2, 3), (0, 0), (0, 1)], G["doc97_prod"]):
        o = tf.DeviceBuffer(4)
        ctx.mul(nt.ptr + i * 32, nt.ptr + j * 32, o.ptr, 1, 1)
        ctx.inntt(o.ptr, o.ptr, 1, 1)
        assert np.array_equal(o.to_numpy(), want)


# ---------------------------------------------------------------------------------------------------
# K3/K4/K5
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("N,bits,nq", [(16, 40, 2), (4096, 50, 3), (1024, 61, 2)])
def test_limbwise_ops(N, bits, nq):
    qs = H.chain(bits, nq, N)
    ref = ref_cpu.RefCtx(N, qs); ctx = tf.Context(N, qs)
    rng = np.random.default_rng(N + bits)
    a, b, c = (H.rand_residues(rng, qs, (3,), N) for _ in range(3))
    a[0, :, :2] = 0; b[0, :, 0] = np.array(qs, dtype=np.uint64) - 1; a[1, :, 0] = np.array(qs, dtype=np.uint64) - 1
    da, db, dc, do = dev(a), dev(b), dev(c), tf.DeviceBuffer(a.size)
    for name in ("add", "sub", "mul"):
        getattr(ctx, name)(da.ptr, db.ptr, do.ptr, 3, nq)
        assert np.array_equal(do.to_numpy(a.shape), ref.pointwise(name, a, b)), name
    ctx.neg(da.ptr, do.ptr, 3, nq)
    assert np.array_equal(do.to_numpy(a.shape), ref.pointwise("neg", a))
    ctx.mad(dc.ptr, da.ptr, db.ptr, do.ptr, 3, nq)
    assert np.array_equal(do.to_numpy(a.shape), ref.pointwise("add", c, ref.pointwise("mul", a, b)))
    s = 0x1234567890ABCDEF1234567890ABCDEF
    ctx.scalar_mul([s % q for q in qs], da.ptr, do.ptr, 3, nq)
    assert np.array_equal(do.to_numpy(a.shape), ref.scalar_mul([s % q for q in qs], a))
    # tensor (rlwe_she.jl:255-258) in the NTT domain
    x, y = H.rand_residues(rng, qs, (2, 2), N), H.rand_residues(rng, qs, (2, 2), N)
    dx, dy, dt = dev(x), dev(y), tf.DeviceBuffer(2 * 3 * nq * N)
    ctx.tensor(dx.ptr, dy.ptr, dt.ptr, 2, nq)
    got = dt.to_numpy((2, 3, nq, N))
    m = lambda u, v: ref.pointwise("mul", u, v)
    assert np.array_equal(got[:, 0], m(x[:, 0], y[:, 0]))
    assert np.array_equal(got[:, 1], ref.pointwise("add", m(x[:, 0], y[:, 1]), m(x[:, 1], y[:, 0])))
    assert np.array_equal(got[:, 2], m(x[:, 1], y[:, 1]))


# ---------------------------------------------------------------------------------------------------
# K6/K7/K8
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("N,bits,nq", [(32, 40, 3), (2048, 50, 4), (16384, 50, 3)])
def test_rescale_select_galois(N, bits, nq):
    qs = H.chain(bits, nq, N)
    ref = ref_cpu.RefCtx(N, qs); ctx = tf.Context(N, qs)
    rng = np.random.default_rng(N)
    a = H.rand_residues(rng, qs, (3,), N)
    da = dev(a)
    do = tf.DeviceBuffer(3 * (nq - 1) * N)
    ctx.rescale(da.ptr, do.ptr, 3, nq)
    assert np.array_equal(do.to_numpy((3, nq - 1, N)), ref.modswitch(a))
    idx = [0, nq - 1]   # rescale by the last selected limb of a sub-basis
    sub = np.ascontiguousarray(a[:, idx])
    ds, d1 = dev(sub), tf.DeviceBuffer(3 * N)
    ctx.rescale(ds.ptr, d1.ptr, 3, 2, idx)
    assert np.array_equal(d1.to_numpy((3, 1, N)), ref.modswitch(sub, idx))
    which = [nq - 1, 0]
    dsel = tf.DeviceBuffer(3 * 2 * N)
    ctx.select_limbs(da.ptr, dsel.ptr, 3, nq, which)
    assert np.array_equal(dsel.to_numpy((3, 2, N)), a[:, which])
    dg = tf.DeviceBuffer(a.size)
    for g in (3, 5, 2 * N - 1, pow(3, N // 2 - 1, 2 * N), spec.galois_element_for_steps(1, N)):
        ctx.galois(da.ptr, dg.ptr, g, 3, nq)
        assert np.array_equal(dg.to_numpy(a.shape), ref.galois(g, a)), g
    with pytest.raises(AssertionError):
        ctx.galois(da.ptr, dg.ptr, 4, 3, nq)   # even element is not an automorphism


def test_golden_modswitch_galois():
    ctx = tf.Context(32, G["ms_q"])
    da = dev(G["ms_in"]); do = tf.DeviceBuffer(3 * 2 * 32); dg = tf.DeviceBuffer(G["ms_in"].size)
    ctx.rescale(da.ptr, do.ptr, 3, 3)
    assert np.array_equal(do.to_numpy(G["ms_out"].shape), G["ms_out"])
    for g in (3, 5, 63, pow(3, 15, 64)):
        ctx.galois(da.ptr, dg.ptr, g, 3, 3)
        assert np.array_equal(dg.to_numpy(G["ms_in"].shape), G[f"gal{g}_out"])


# ---------------------------------------------------------------------------------------------------
# K9-K11: keyswitch / rotate
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("N,bits,Lk", [(32, 40, 4), (2048, 50, 3), (8192, 50, 4), (16384, 50, 4), (1 << 15, 50, 3), (1 << 16, 50, 3),
                                       (1 << 15, 40, 4), (1 << 16, 40, 4)])
@pytest.mark.parametrize("special", [True, False])
def test_keyswitch_matches_oracle(N, bits, Lk, special):
    qs = H.chain(bits, Lk, N)
    ref = ref_cpu.RefCtx(N, qs); ctx = tf.Context(N, qs)
    rng = np.random.default_rng(N + Lk)
    evk = H.uniform_evk(rng, qs, Lk, N)
    devk = dev(evk)
    levels = (Lk - 1, 1) if special else (Lk, 2)
    for level in levels:
        for polys in (2, 3):
            batch = 3
            ct = H.rand_residues(rng, qs[:level], (batch, polys), N)
            # centring edges of the RNS digits: 0, 1, q-1, floor(q/2), floor(q/2)+1
            for l in range(level):
                ct[0, polys - 1, l, :5] = [0, 1, qs[l] - 1, qs[l] // 2, qs[l] // 2 + 1]
            dct, dout = dev(ct), tf.DeviceBuffer(batch * 2 * level * N)
            ctx.keyswitch(Lk, level, special, devk.ptr, Lk, dct.ptr, polys, dout.ptr, batch)
            want = ref.keyswitch(level, special, evk, ct)
            assert np.array_equal(dout.to_numpy(want.shape), want), (level, polys)
    with pytest.raises(AssertionError):
        ctx.keyswitch(Lk, 1, special, devk.ptr, Lk, devk.ptr, 4, devk.ptr, 1)      # rlwe_she.jl:318
    with pytest.raises(tf.UsageError):
        ctx.keyswitch(Lk, Lk + 1, special, devk.ptr, Lk, devk.ptr, 2, devk.ptr, 1)  # level outside the key ring


@pytest.mark.parametrize("bits", [50, 40])
def test_keyswitch_sub_block_fused_at_2_16_more_items_than_workgroups(bits):
    """k_ks_fused_sub at X = 2 (four sub-blocks per row, quarters streamed through the LDS) with several items per workgroup:
    44 ciphertexts x 3 working limbs x 4 sub-blocks = 528 items on 256 workgroups; fp64 policy (50 bit) and the small-modulus
    range plan (40 bit).  Picks against the oracle, every ciphertext against the three-kernel path's bits (single-ciphertext
    calls go the same fused way, so the cross-check is the batch permuted)."""
    N, Lk, level, batch = 1 << 16, 3, 2, 44
    qs = H.chain(bits, Lk, N)
    ref = ref_cpu.RefCtx(N, qs); ctx = tf.Context(N, qs)
    rng = np.random.default_rng(bits)
    evk = H.uniform_evk(rng, qs, Lk, N)
    devk = dev(evk)
    ct = H.rand_residues(rng, qs[:level], (batch, 2), N)
    dct, dout = dev(ct), tf.DeviceBuffer(batch * 2 * level * N)
    ctx.keyswitch(Lk, level, True, devk.ptr, Lk, dct.ptr, 2, dout.ptr, batch)
    got = dout.to_numpy((batch, 2, level, N))
    pick = [0, 21, batch - 1]
    assert np.array_equal(got[pick], ref.keyswitch(level, True, evk, ct[pick]))
    perm = rng.permutation(batch)
    dct2, dout2 = dev(ct[perm]), tf.DeviceBuffer(batch * 2 * level * N)
    ctx.keyswitch(Lk, level, True, devk.ptr, Lk, dct2.ptr, 2, dout2.ptr, batch)
    assert np.array_equal(dout2.to_numpy((batch, 2, level, N)), got[perm])


@pytest.mark.parametrize("N,bits,Lk,level,batch", [(8192, 40, 6, 3, 300), (16384, 50, 5, 4, 270)])
def test_keyswitch_special_prime_in_kernel_contraction_many_ciphertexts(N, bits, Lk, level, batch):
    """The two-launch form of the fused key switch with a special prime (k_ks_fused SPMODE 1 / 2: the special limb's rows first,
    then the ciphertext limbs with the ModulusRaised contraction in their final store, modulusraising.jl:35-49) on more
    ciphertexts than there are workgroups (several items per workgroup in both launches), at a lower level of a longer key
    (downswitch_keyelement, :43-49): the first and last ciphertexts against the oracle, every ciphertext against its own
    single-ciphertext call."""
    qs = H.chain(bits, Lk, N)
    ref = ref_cpu.RefCtx(N, qs); ctx = tf.Context(N, qs)
    rng = np.random.default_rng(N + batch)
    evk = H.uniform_evk(rng, qs, Lk, N)
    devk = dev(evk)
    ct = H.rand_residues(rng, qs[:level], (batch, 2), N)
    dct, dout = dev(ct), tf.DeviceBuffer(batch * 2 * level * N)
    ctx.keyswitch(Lk, level, True, devk.ptr, Lk, dct.ptr, 2, dout.ptr, batch)
    got = dout.to_numpy((batch, 2, level, N))
    pick = [0, 1, batch // 2, batch - 2, batch - 1]
    want = ref.keyswitch(level, True, evk, ct[pick])
    assert np.array_equal(got[pick], want)
    one = tf.DeviceBuffer(2 * level * N)
    for b in range(0, batch, 17):
        ctx.keyswitch(Lk, level, True, devk.ptr, Lk, dev(ct[b:b + 1]).ptr, 2, one.ptr, 1)
        assert np.array_equal(one.to_numpy((1, 2, level, N))[0], got[b]), b


@pytest.mark.parametrize("special", [True, False])
def test_keyswitch_mixed_modulus_sizes(special):
    """Key switch on a ring that mixes a 60-bit q0 with 40-bit primes (+ a 61-bit special prime): the digit lift stays on
    the u64 kernels, the inverse transforms (addend mode) run as one pass per arithmetic policy; batch large enough for
    the two-pass mode."""
    N, batch = 1 << 12, 48
    qs = H.chain(60, 1, N) + H.chain(40, 2, N) + H.chain(61, 1, N)
    Lk = len(qs)
    level = Lk - 1 if special else Lk
    ref = ref_cpu.RefCtx(N, qs); ctx = tf.Context(N, qs)
    rng = np.random.default_rng(7 + special)
    evk = H.uniform_evk(rng, qs, Lk, N)
    ct = H.rand_residues(rng, qs[:level], (batch, 2), N)
    devk, dct, dout = dev(evk), dev(ct), tf.DeviceBuffer(batch * 2 * level * N)
    want = ref.keyswitch(level, special, evk, ct)
    for variant in (0, 2):
        ctx.set_ntt_variant(variant)
        ctx.keyswitch(Lk, level, special, devk.ptr, Lk, dct.ptr, 2, dout.ptr, batch)
        assert np.array_equal(dout.to_numpy(want.shape), want), variant


@pytest.mark.parametrize("logn,batch", [(14, 6), (15, 3), (16, 2)])
@pytest.mark.parametrize("special", [True, False])
def test_keyswitch_mixed_modulus_sizes_large_n(logn, batch, special):
    """the reference's CKKS ring shape (60-bit q0 and special prime around 40-bit primes, infer.jl:97-107) at N = 2^14 ... 2^16:
    the fp64-size working limbs take the lift-fused transforms -- their lift reads the 60-bit source limb through the integer
    path -- and the 60-bit limbs the digit buffer + u64 kernels; 2- and 3-element inputs, centring edges on every limb."""
    N = 1 << logn
    qs = H.chain(60, 1, N) + H.chain(40, 3, N) + [H.chain(60, 2, N)[1]]
    Lk = len(qs)
    level = Lk - 1 if special else Lk
    ref = ref_cpu.RefCtx(N, qs); ctx = tf.Context(N, qs)
    rng = np.random.default_rng(logn * 2 + special)
    evk = H.uniform_evk(rng, qs, Lk, N)
    devk = dev(evk)
    for polys in (2, 3):
        ct = H.rand_residues(rng, qs[:level], (batch, polys), N)
        for l in range(level):
            ct[0, polys - 1, l, :5] = [0, 1, qs[l] - 1, qs[l] // 2, qs[l] // 2 + 1]
        dct, dout = dev(ct), tf.DeviceBuffer(batch * 2 * level * N)
        ctx.keyswitch(Lk, level, special, devk.ptr, Lk, dct.ptr, polys, dout.ptr, batch)
        want = ref.keyswitch(level, special, evk, ct)
        assert np.array_equal(dout.to_numpy(want.shape), want), polys
    if logn == 14:                                                    # the all-u64 path must agree
        ctx.set_ntt_variant(2)
        ctx.keyswitch(Lk, level, special, devk.ptr, Lk, dct.ptr, 3, dout.ptr, batch)
        assert np.array_equal(dout.to_numpy(want.shape), want)


@pytest.mark.parametrize("N,bits,L,w", [(32, 60, 1, 1), (64, 50, 1, 8), (32, 40, 2, 10), (2048, 50, 3, 16), (16, 61, 4, 32),
                                        (1 << 15, 50, 1, 20)])
def test_keyswitch_window_matches_oracle(N, bits, L, w):
    """K14: key switch with base-2^w digits of convert(Integer, x) (rlwe_she.jl:330-338), bit-exact against the spec
    oracle; multi-limb rings need the exact integer reconstruction on the device."""
    qs = H.chain(bits, L, N)
    ring = spec.Ring(N, qs)
    nwin = spec.ndigits(ring.Q, 2 ** w)
    rng = np.random.default_rng(N + w)
    ctx = tf.Context(N, qs)
    batch = 1 if N > 4096 else 3
    evk = np.stack([H.rand_residues(rng, qs, (2,), N) for _ in range(nwin)])          # [nwin][2][L][N], coefficient domain
    evk_ntt = np.array([[spec.poly_nntt([list(map(int, l)) for l in comp], ring) for comp in pair] for pair in evk], dtype=np.uint64)
    devk = dev(evk_ntt)
    for polys in (2, 3):
        ct = H.rand_residues(rng, qs, (batch, polys), N)
        ct[0, polys - 1, :, 0] = 0                                                     # x = 0
        ct[0, polys - 1, :, 1] = [q - 1 for q in qs]                                   # x = Q - 1
        ct[0, polys - 1, :, 2] = 1                                                     # x = 1 (exact-alpha fallback)
        dct, dout = dev(ct), tf.DeviceBuffer(batch * 2 * L * N)
        ctx.keyswitch_window(L, w, devk.ptr, nwin, dct.ptr, polys, dout.ptr, batch)
        got = dout.to_numpy((batch, 2, L, N))
        for b in range(batch):
            want = spec.keyswitch([([list(map(int, l)) for l in p[0]], [list(map(int, l)) for l in p[1]]) for p in evk],
                                  [[list(map(int, l)) for l in c] for c in ct[b]], ring, ring, False, relin_window=w)
            assert np.array_equal(got[b], np.array(want, dtype=np.uint64)), (polys, b)
    with pytest.raises(tf.UsageError):
        ctx.keyswitch_window(L, w, devk.ptr, nwin - 1, devk.ptr, 2, devk.ptr, 1)      # too few key components for this ring
    with pytest.raises(AssertionError):
        ctx.keyswitch_window(L, 33, devk.ptr, nwin, devk.ptr, 2, devk.ptr, 1)
    with pytest.raises(AssertionError):
        ctx.keyswitch_window(L, w, devk.ptr, nwin, devk.ptr, 4, devk.ptr, 1)           # rlwe_she.jl:318


@pytest.mark.parametrize("N,bits,Lk,level,w", [(32, 40, 3, 2, 10), (64, 50, 4, 3, 16), (64, 50, 4, 2, 7), (2048, 50, 3, 2, 20), (16, 61, 2, 1, 32)])
def test_keyswitch_window_with_special_prime_matches_oracle(N, bits, Lk, level, w):
    """ModulusRaised + digit window (rlwe_she.jl:330-338 under modulusraising.jl:35-49): the base-2^w digits of c[end] over
    the ciphertext modulus, keys over the key ring restricted to [q_1..q_l, P], P c raised and the sums contracted by
    floor(./P); at the top level and at a lower level of the same key (more key components than digits), bit-exact against
    the spec oracle."""
    qs = H.chain(bits, Lk, N)
    keyring, cring = spec.Ring(N, qs), spec.Ring(N, qs[:level])
    nkey = spec.ndigits(spec.Ring(N, qs[:Lk - 1]).Q * qs[-1], 2 ** w)   # a key made over Q P has ndigits(Q P) components
    need = spec.ndigits(cring.Q, 2 ** w)
    assert nkey >= need
    rng = np.random.default_rng(N + w + level)
    ctx = tf.Context(N, qs)
    evk = np.stack([H.rand_residues(rng, qs, (2,), N) for _ in range(nkey)])            # [nkey][2][Lk][N], coefficient domain
    evk_ntt = np.array([[spec.poly_nntt([list(map(int, l)) for l in comp], keyring) for comp in pair] for pair in evk], dtype=np.uint64)
    devk = dev(evk_ntt)
    batch = 3
    for polys in (2, 3):
        ct = H.rand_residues(rng, qs[:level], (batch, polys), N)
        ct[0, polys - 1, :, 0] = 0
        ct[0, polys - 1, :, 1] = [q - 1 for q in qs[:level]]
        ct[0, polys - 1, :, 2] = 1
        dct, dout = dev(ct), tf.DeviceBuffer(batch * 2 * level * N)
        ctx.keyswitch_window(level, w, devk.ptr, nkey, dct.ptr, polys, dout.ptr, batch, key_limbs=Lk, special=True)
        got = dout.to_numpy((batch, 2, level, N))
        for b in range(batch):
            want = spec.keyswitch([([list(map(int, l)) for l in p[0]], [list(map(int, l)) for l in p[1]]) for p in evk],
                                  [[list(map(int, l)) for l in c] for c in ct[b]], cring, keyring, True, relin_window=w)
            assert np.array_equal(got[b], np.array(want, dtype=np.uint64)), (polys, b)
    with pytest.raises(tf.UsageError):
        ctx.keyswitch_window(level, w, devk.ptr, need - 1, devk.ptr, 2, devk.ptr, 1, key_limbs=Lk, special=True)
    with pytest.raises(tf.UsageError):
        ctx.keyswitch_window(Lk, w, devk.ptr, nkey, devk.ptr, 2, devk.ptr, 1, key_limbs=Lk, special=True)   # no room for P


@pytest.mark.parametrize("N,qspec,batch", [(2048, "60x3", 3), (1 << 12, "mixed", 40), (1 << 14, "50x4", 5), (1 << 15, "50x3", 2),
                                           (1 << 16, "50x3", 2), (1 << 16, "mixed", 2)])
@pytest.mark.parametrize("special", [True, False])
def test_rotate_many_equals_individual_rotations(N, qspec, batch, special):
    """tfhe_rotate_many (hoisted rotations: one digit decomposition, the automorphism applied as an index permutation of the
    transformed digits) against tfhe_rotate per Galois element, bit for bit, and one of them against the oracle."""
    if qspec == "mixed":
        qs = H.chain(60, 1, N) + H.chain(40, 2, N) + [H.chain(60, 2, N)[1]]
    else:
        bits, n = qspec.split("x")
        qs = H.chain(int(bits), int(n), N)
    Lk = len(qs)
    level = Lk - 1 if special else Lk
    ctx, ref = tf.Context(N, qs), ref_cpu.RefCtx(N, qs)
    rng = np.random.default_rng(N % 1000 + special)
    gs = [3, pow(3, 2 * N - 1, 2 * N), 2 * N - 1, pow(3, 5, 2 * N)]
    evks = [H.uniform_evk(rng, qs, Lk, N) for _ in gs]
    devks = [dev(e) for e in evks]
    ct = H.rand_residues(rng, qs[:level], (batch, 2), N)
    for l in range(level):
        ct[0, 1, l, :5] = [0, 1, qs[l] - 1, qs[l] // 2, qs[l] // 2 + 1]
    dct = dev(ct)
    out = tf.DeviceBuffer(len(gs) * batch * 2 * level * N)
    ctx.rotate_many(Lk, level, special, [d.ptr for d in devks], Lk, gs, dct.ptr, out.ptr, batch)
    got = out.to_numpy((len(gs), batch, 2, level, N))
    one = tf.DeviceBuffer(batch * 2 * level * N)
    for r, g in enumerate(gs):
        ctx.rotate(Lk, level, special, devks[r].ptr, Lk, g, dct.ptr, one.ptr, batch)
        assert np.array_equal(got[r], one.to_numpy((batch, 2, level, N))), (r, g)
    r = 1
    want = ref.keyswitch(level, special, evks[r], ref.galois(gs[r], ct.reshape(-1, level, N), idx=range(level)).reshape(ct.shape))
    assert np.array_equal(got[r], want)
    # prepared keys (rows permuted by g^-1 once) give the same bits
    prep = [tf.DeviceBuffer(e.size) for e in evks]
    for r, g in enumerate(gs):
        ctx.galois_key_prepare(Lk, Lk, g, devks[r].ptr, prep[r].ptr)
    out2 = tf.DeviceBuffer(len(gs) * batch * 2 * level * N)
    ctx.rotate_many(Lk, level, special, [d.ptr for d in prep], Lk, gs, dct.ptr, out2.ptr, batch, prepared=True)
    assert np.array_equal(out2.to_numpy(got.shape), got)
    with pytest.raises(AssertionError):
        ctx.rotate_many(Lk, level, special, [devks[0].ptr], Lk, [4], dct.ptr, out.ptr, batch)   # even element


@pytest.mark.parametrize("special", [True, False])
@pytest.mark.parametrize("N,qspec", [(1 << 14, "50x4"), (1 << 15, "40x4"), (1 << 15, "50x3"), (1 << 16, "50x3"), (1 << 16, "mixed")])
def test_rotation_finished_in_the_tail(N, qspec, special):
    """tfhe_rotate at N = 2^15 / 2^16 on enough ciphertexts (>= 8) takes no rotated copy of its input: the key is prepared on the
    way, the key sums are those of the unrotated digits and the automorphism rides on the tail's stores (k_ks_top_tail_rot, signs
    before the ModulusRaised floor).  Against the hoisted path (tfhe_rotate_many: inverse transform, automorphism pass, tail --
    other kernels throughout) on every ciphertext, and against the oracle's rotate = keyswitch o apply_galois_element
    (rlwe_she.jl:355-359) on two of them; a Galois element with many sign wraps and the conjugation.  At N = 2^14 with the special
    prime the rotation rides on the in-kernel contraction's stores (k_ks_fused SPMODE 3), on more ciphertexts than workgroups; the
    hoisted path is reached there through a prepared key."""
    if qspec == "mixed":
        qs = H.chain(60, 1, N) + H.chain(40, 2, N) + [H.chain(60, 2, N)[1]]
    else:
        bits, n = qspec.split("x")
        qs = H.chain(int(bits), int(n), N)
    Lk = len(qs)
    level = Lk - 1 if special else Lk
    batch = 70 if N == 1 << 14 else 9      # 70 x 3 limbs = 210 items + 70 special-limb items on 256 workgroups, then > 1 item each at level 4
    ctx, ref = tf.Context(N, qs), ref_cpu.RefCtx(N, qs)
    rng = np.random.default_rng(N % 1000 + 7 * special + len(qspec))
    ct = H.rand_residues(rng, qs[:level], (batch, 2), N)
    for l in range(level):   # zeros (no sign to flip), the centring edges
        ct[0, :, l, :6] = [0, 1, qs[l] - 1, qs[l] // 2, qs[l] // 2 + 1, 0]
    dct = dev(ct)
    for g in (pow(3, N // 2 + 3, 2 * N), 2 * N - 1):
        evk = H.uniform_evk(rng, qs, Lk, N)
        devk = dev(evk)
        one, many = tf.DeviceBuffer(batch * 2 * level * N), tf.DeviceBuffer(batch * 2 * level * N)
        ctx.rotate(Lk, level, special, devk.ptr, Lk, g, dct.ptr, one.ptr, batch)
        if N == 1 << 14:   # (unprepared, tfhe_rotate_many goes through tfhe_rotate there)
            prep = tf.DeviceBuffer(evk.size)
            ctx.galois_key_prepare(Lk, Lk, g, devk.ptr, prep.ptr)
            ctx.rotate_many(Lk, level, special, [prep.ptr], Lk, [g], dct.ptr, many.ptr, batch, prepared=True)
        else:
            ctx.rotate_many(Lk, level, special, [devk.ptr], Lk, [g], dct.ptr, many.ptr, batch)
        got = one.to_numpy((batch, 2, level, N))
        assert np.array_equal(got, many.to_numpy((batch, 2, level, N))), g
        pick = [0, batch - 1]
        want = ref.keyswitch(level, special, evk, ref.galois(g, ct[pick].reshape(-1, level, N), idx=range(level)).reshape(ct[pick].shape))
        assert np.array_equal(got[pick], want), g


@pytest.mark.parametrize("special", [False, True])
def test_rotate_full_degree_matches_oracle(special):
    """rotate = keyswitch o apply_galois_element (rlwe_she.jl:355-359) at N = 2^14 -- the fused key-switch kernel behind the
    Galois permutation, with and without the special prime -- against the oracle composition."""
    N, Lk = 1 << 14, 4
    qs = H.chain(50, Lk, N)
    level = Lk - 1 if special else Lk
    ref = ref_cpu.RefCtx(N, qs); ctx = tf.Context(N, qs)
    rng = np.random.default_rng(14 + special)
    evk = H.uniform_evk(rng, qs, Lk, N)
    ct = H.rand_residues(rng, qs[:level], (3, 2), N)
    g = pow(3, 5, 2 * N)
    devk, dct, dout = dev(evk), dev(ct), tf.DeviceBuffer(3 * 2 * level * N)
    ctx.rotate(Lk, level, special, devk.ptr, Lk, g, dct.ptr, dout.ptr, 3)
    rot = ref.galois(g, ct, idx=list(range(level)))
    want = ref.keyswitch(level, special, evk, rot)
    assert np.array_equal(dout.to_numpy(want.shape), want)


def test_golden_keyswitch_and_rotate():
    ctx = tf.Context(32, G["ksS_q"])
    devk, dct = dev(G["ksS_evk_ntt"]), dev(G["ksS_ct"])
    dout = tf.DeviceBuffer(G["ksS_out"].size)
    ctx.keyswitch(3, 2, True, devk.ptr, 3, dct.ptr, 2, dout.ptr, 2)
    assert np.array_equal(dout.to_numpy(G["ksS_out"].shape), G["ksS_out"])
    devk, dct = dev(G["ksR_evk_ntt"]), dev(G["ksR_ct"])
    dout = tf.DeviceBuffer(G["ksR_out"].size)
    ctx.keyswitch(3, 3, False, devk.ptr, 3, dct.ptr, 3, dout.ptr, 2)
    assert np.array_equal(dout.to_numpy(G["ksR_out"].shape), G["ksR_out"])
    # rotate = keyswitch ∘ apply_galois_element (rlwe_she.jl:359), against the oracle composition
    ref = ref_cpu.RefCtx(32, G["ksS_q"])
    g = spec.galois_element_for_steps(1, 32)
    ct = G["ksS_ct"]
    want = ref.keyswitch(2, True, G["ksS_evk_ntt"], ref.galois(g, ct.reshape(-1, 2, 32), [0, 1]).reshape(ct.shape))
    devk, dct = dev(G["ksS_evk_ntt"]), dev(ct)
    dout = tf.DeviceBuffer(want.size)
    ctx.rotate(3, 2, True, devk.ptr, 3, g, dct.ptr, dout.ptr, 2)
    assert np.array_equal(dout.to_numpy(want.shape), want)


def test_keyswitch_decrypts_ckks_modraise():
    """test/ckks_modraise.jl:10-30 end to end: encrypt (host), keyswitch s->s on the device, decrypt (host)."""
    import random
    N = 32
    qs = spec.prime_chain(2**40 + 1, 3, N)
    keyring = spec.Ring(N, qs); cring = keyring.drop_last()
    prng = random.Random(77)
    s, pub = spec.keygen(prng, keyring, 3.2)
    ct = [spec.modswitch_drop_poly(c, keyring) for c in spec.encrypt_zero(prng, pub, keyring, 3.2)]
    slots = np.arange(1, N // 2 + 1).astype(complex)
    ct[0] = spec.poly_add(ct[0], spec.ckks_encode(slots, cring, 2**40), cring)
    evk = spec.make_eval_key(prng, s, s, keyring, 3.2, premul=qs[-1])
    evk_ntt = np.array([[spec.poly_nntt(m, keyring), spec.poly_nntt(md, keyring)] for m, md in evk], dtype=np.uint64)
    ctx = tf.Context(N, qs)
    dct, devk, dout = dev(np.array([ct], dtype=np.uint64)), dev(evk_ntt), tf.DeviceBuffer(2 * 2 * N)
    ctx.keyswitch(3, 2, True, devk.ptr, 3, dct.ptr, 2, dout.ptr, 1)
    out = dout.to_numpy((2, 2, N))
    s_c = spec.modswitch_drop_poly(s, keyring)
    dec = spec.ckks_decode(spec.decrypt_raw(s_c, [[list(map(int, l)) for l in p] for p in out], cring), cring, 2**40)
    assert np.abs(dec - slots).max() < 1e-8          # the reference test's atol


# ---------------------------------------------------------------------------------------------------
# K12/K13: BFV
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("mode", ["superset", "disjoint"])
@pytest.mark.parametrize("N,bits,ns,nextra", [(32, 50, 3, 4), (2048, 50, 2, 3), (4096, 40, 2, 4), (1024, 60, 2, 3), (64, 50, 8, 9), (256, 61, 6, 7)])
def test_bfv_expand_contract_mul(mode, N, bits, ns, nextra):
    t = 65537
    ch = H.chain(bits, 2 * ns + nextra + 1, N)
    qs = ch[:ns]
    pb = ch[: ns + nextra] if mode == "superset" else ch[ns: 2 * ns + nextra + 1]
    rs, rb = ref_cpu.RefCtx(N, qs), ref_cpu.RefCtx(N, pb)
    small, big = spec.Ring(N, qs), spec.Ring(N, pb)
    if mode == "superset":
        cbig = tf.Context(N, pb); csmall = cbig
        plan = tf.BfvPlan(csmall, cbig, t, idx_s=list(range(ns)), idx_b=None)
    else:
        csmall, cbig = tf.Context(N, qs), tf.Context(N, pb)
        plan = tf.BfvPlan(csmall, cbig, t)
    rng = np.random.default_rng(N + ns)
    a = H.rand_residues(rng, qs, (4,), N)
    for k, x in enumerate([0, 1, small.Q - 1, small.Q // 2, small.Q // 2 + 1]):
        a[0, :, k] = [x % q for q in qs]
    da, de = dev(a), tf.DeviceBuffer(4 * len(pb) * N)
    plan.expand(da.ptr, de.ptr, 4)
    assert np.array_equal(de.to_numpy((4, len(pb), N)), ref_cpu.switch(rs, rb, a))
    y = H.rand_residues(rng, pb, (3,), N)
    tinv = pow(t, -1, big.Q)
    edges = [0, 1, big.Q - 1, big.Q // 2, big.Q // 2 + 1, small.Q // 2, small.Q // 2 + 1, small.Q,
             5 * small.Q + small.Q // 2, 5 * small.Q + small.Q // 2 + 1, big.Q - small.Q // 2 - 1]
    for k, x in enumerate(edges):
        y[0, :, k] = [(x * tinv) % big.Q % p for p in pb]
    dy, dc = dev(y), tf.DeviceBuffer(3 * ns * N)
    plan.contract(dy.ptr, dc.ptr, 3)
    assert np.array_equal(dc.to_numpy((3, ns, N)), ref_cpu.contract(rb, rs, t, y))
    if mode == "superset":   # the general kernels must agree with the register-resident fast path (when instantiated)
        plan.set_variant(1)
        plan.expand(da.ptr, de.ptr, 4)
        assert np.array_equal(de.to_numpy((4, len(pb), N)), ref_cpu.switch(rs, rb, a))
        plan.contract(dy.ptr, dc.ptr, 3)
        assert np.array_equal(dc.to_numpy((3, ns, N)), ref_cpu.contract(rb, rs, t, y))
        plan.set_variant(0)
    batch = 5
    plan.set_chunk(2)   # exercise the chunked pipeline incl. a ragged last chunk
    c1, c2 = H.rand_residues(rng, qs, (batch, 2), N), H.rand_residues(rng, qs, (batch, 2), N)
    d1, d2, do = dev(c1), dev(c2), tf.DeviceBuffer(batch * 3 * ns * N)
    plan.mul(d1.ptr, d2.ptr, do.ptr, batch)
    assert np.array_equal(do.to_numpy((batch, 3, ns, N)), ref_cpu.bfv_mul(rs, rb, t, c1, c2))


def test_golden_bfv_vectors_and_decrypt():
    q = G["bfvcrt_q"]                                   # test/bfv_crt.jl parameters, disjoint ℛbig
    cs, cb = tf.Context(2048, q[:2]), tf.Context(2048, q[2:])
    plan = tf.BfvPlan(cs, cb, int(G["bfvcrt_t"][0]))
    ct = G["bfvcrt_ct"][None]
    d, do = dev(ct), tf.DeviceBuffer(3 * 2 * 2048)
    plan.mul(d.ptr, d.ptr, do.ptr, 1)
    prod = do.to_numpy((3, 2, 2048))
    assert np.array_equal(prod, G["bfvcrt_prod"])
    small = spec.Ring(2048, [int(x) for x in q[:2]])
    s = [[int(v) for v in l] for l in G["bfvcrt_secret"]]
    dec = spec.bfv_decode(spec.decrypt_raw(s, [[[int(v) for v in l] for l in p] for p in prod], small), small, 53)
    assert dec[0] == 36                                  # test/bfv_crt.jl:45-47
    q = G["bfvsup_q"]
    c = tf.Context(64, q)
    plan = tf.BfvPlan(c, c, int(G["bfvsup_t"][0]), idx_s=[0, 1, 2])
    d1, d2, do = dev(G["bfvsup_c1"][None]), dev(G["bfvsup_c2"][None]), tf.DeviceBuffer(3 * 3 * 64)
    plan.mul(d1.ptr, d2.ptr, do.ptr, 1)
    assert np.array_equal(do.to_numpy((3, 3, 64)), G["bfvsup_prod"])
    with pytest.raises(NotImplementedError):
        tf.BfvPlan(c, c, 65537, idx_s=[0, 1, 2], idx_b=[1, 2, 3, 4])   # partial overlap


def test_bfv_mul_relin_matches_oracle_and_decrypts():
    import random
    N, t, ns = 1024, 65537, 3
    ch = H.chain(50, 2 * ns + 2, N)
    qs, pb = ch[:ns], ch
    rs, rb = ref_cpu.RefCtx(N, qs), ref_cpu.RefCtx(N, pb)
    ring = spec.Ring(N, qs)
    secret, evk = H.real_evk(5, N, qs, special=False)
    # genuine encryptions of 6 and 7 under `secret`
    prng = random.Random(6); rng = np.random.default_rng(6)
    s_l = [[int(v) for v in l] for l in secret]
    def enc(m):
        mask = [[prng.randrange(q) for _ in range(N)] for q in qs]
        e = spec.poly_from_ints(spec.sample_gauss_ints(prng, N, 3.2), ring)
        c0 = spec.poly_sub(spec.poly_add(spec.bfv_encode([m] + [0] * (N - 1), ring, t), e, ring), spec.poly_mul(mask, s_l, ring), ring)
        return [c0, mask]
    c1 = np.array([enc(6), enc(3)], dtype=np.uint64); c2 = np.array([enc(7), enc(5)], dtype=np.uint64)
    ctx = tf.Context(N, pb)
    plan = tf.BfvPlan(ctx, ctx, t, idx_s=list(range(ns)))
    d1, d2, devk, do = dev(c1), dev(c2), dev(evk), tf.DeviceBuffer(2 * 2 * ns * N)
    plan.mul_relin(devk.ptr, ns, d1.ptr, d2.ptr, do.ptr, 2)
    got = do.to_numpy((2, 2, ns, N))
    want = rs.keyswitch(ns, False, evk, ref_cpu.bfv_mul(rs, rb, t, c1, c2))
    assert np.array_equal(got, want)
    for b, m in enumerate((42, 15)):
        dec = spec.bfv_decode(spec.decrypt_raw(s_l, [[[int(v) for v in l] for l in p] for p in got[b]], ring), ring, t)
        assert dec[0] == m and not any(dec[1:])


# ---------------------------------------------------------------------------------------------------
# BASELINE.json size: N = 2^14, L = 8 (+9 extension limbs); properties + oracle on a sub-batch
# ---------------------------------------------------------------------------------------------------
def test_full_size_bfv_mul_relin():
    N, L, t = 1 << 14, 8, 65537
    ch = H.chain(50, 17, N)
    assert ch[:2] == [1125899908022273, 1125899908612097]     # BASELINE.md §3 moduli
    qs = ch[:L]
    ctx = tf.Context(N, ch)
    plan = tf.BfvPlan(ctx, ctx, t, idx_s=list(range(L)))
    rng = np.random.default_rng(2026)
    batch = 24
    c1, c2 = H.rand_residues(rng, qs, (batch, 2), N), H.rand_residues(rng, qs, (batch, 2), N)
    evk = H.uniform_evk(rng, qs, L, N)
    d1, d2, devk = dev(c1), dev(c2), dev(evk)
    do = tf.DeviceBuffer(batch * 2 * L * N)
    plan.set_chunk(16)
    plan.mul_relin(devk.ptr, L, d1.ptr, d2.ptr, do.ptr, batch)
    got = do.to_numpy((batch, 2, L, N))
    # (1) oracle on a sub-batch (first, a middle one across the chunk boundary, last)
    rs, rb = ref_cpu.RefCtx(N, qs), ref_cpu.RefCtx(N, ch)
    pick = [0, 15, 16, batch - 1]
    want = rs.keyswitch(L, False, evk, ref_cpu.bfv_mul(rs, rb, t, c1[pick], c2[pick]))
    assert np.array_equal(got[pick], want)
    # (2) commutativity of the ciphertext product: c1*c2 == c2*c1 bit for bit
    do2 = tf.DeviceBuffer(batch * 2 * L * N)
    plan.mul_relin(devk.ptr, L, d2.ptr, d1.ptr, do2.ptr, batch)
    assert np.array_equal(do2.to_numpy(got.shape), got)
    # (3) batch independence: a permuted batch gives the permuted result
    perm = rng.permutation(batch)
    d1p, d2p = dev(c1[perm]), dev(c2[perm])
    plan.mul_relin(devk.ptr, L, d1p.ptr, d2p.ptr, do2.ptr, batch)
    assert np.array_equal(do2.to_numpy(got.shape), got[perm])


@pytest.mark.parametrize("logn,ns,next_", [(12, 3, 4), (13, 3, 4), (13, 2, 3), (12, 6, 7)])
def test_fused_bfv_core_at_the_reference_test_sizes(logn, ns, next_):
    """k_bfv_core_fused at N = 2^12 and 2^13 (the sizes of the reference's own BFV tests and MNIST parameters: test/bfv_crt.jl:8,
    infer.jl:97): 50-bit chains on the (ns, np) pairs of the register-resident conversions, more items than compute units,
    ragged chunks -- against the oracle on picks across the chunk boundaries and against the one-operation-per-launch path
    (NTT variant 3) on every ciphertext."""
    N, t = 1 << logn, 65537
    ch = H.chain(50, ns + next_, N)
    qs = ch[:ns]
    ctx = tf.Context(N, ch)
    plan = tf.BfvPlan(ctx, ctx, t, idx_s=list(range(ns)))
    rng = np.random.default_rng(1300 + logn + ns)
    batch = 90                                                   # 90 * 7 = 630 items at (3, 4): more than 2 x 256 workgroups
    c1, c2 = H.rand_residues(rng, qs, (batch, 2), N), H.rand_residues(rng, qs, (batch, 2), N)
    evk = H.uniform_evk(rng, qs, ns, N)
    d1, d2, devk = dev(c1), dev(c2), dev(evk)
    do3 = tf.DeviceBuffer(batch * 3 * ns * N)
    plan.set_chunk(64)
    plan.mul(d1.ptr, d2.ptr, do3.ptr, batch)
    got3 = do3.to_numpy((batch, 3, ns, N))
    rs, rb = ref_cpu.RefCtx(N, qs), ref_cpu.RefCtx(N, ch)
    pick = [0, 63, 64, batch - 1]
    want3 = ref_cpu.bfv_mul(rs, rb, t, c1[pick], c2[pick])
    assert np.array_equal(got3[pick], want3)
    do = tf.DeviceBuffer(batch * 2 * ns * N)
    plan.mul_relin(devk.ptr, ns, d1.ptr, d2.ptr, do.ptr, batch)
    got = do.to_numpy((batch, 2, ns, N))
    assert np.array_equal(got[pick], rs.keyswitch(ns, False, evk, want3))
    ctx.set_ntt_variant(3)
    do_b = tf.DeviceBuffer(batch * 3 * ns * N)
    plan.mul(d1.ptr, d2.ptr, do_b.ptr, batch)
    assert np.array_equal(do_b.to_numpy(got3.shape), got3)
    do_c = tf.DeviceBuffer(batch * 2 * ns * N)
    plan.mul_relin(devk.ptr, ns, d1.ptr, d2.ptr, do_c.ptr, batch)
    ctx.set_ntt_variant(0)
    assert np.array_equal(do_c.to_numpy(got.shape), got)


def test_fused_pipeline_equals_three_kernel_pipeline():
    """BASELINE configuration with more (ciphertext, limb) items than compute units: the fused kernels (k_bfv_core_fused,
    k_ks_fused: several items per workgroup, ragged last chunk) against the separate transform / tensor / inner-product
    kernels (NTT variant 3), bit for bit; both paths are checked against the oracle on small batches elsewhere."""
    N, L, t = 1 << 14, 8, 65537
    ch = H.chain(50, 17, N)
    qs = ch[:L]
    ctx = tf.Context(N, ch)
    plan = tf.BfvPlan(ctx, ctx, t, idx_s=list(range(L)))
    rng = np.random.default_rng(77)
    batch = 70                                                   # 70 * 17 = 1190 core items, 560 key-switch items
    c1, c2 = H.rand_residues(rng, qs, (batch, 2), N), H.rand_residues(rng, qs, (batch, 2), N)
    evk = H.uniform_evk(rng, qs, L, N)
    d1, d2, devk = dev(c1), dev(c2), dev(evk)
    outs = {}
    for variant, chunk in ((0, 48), (3, 48), (0, 256)):
        ctx.set_ntt_variant(variant)
        plan.set_chunk(chunk)
        do = tf.DeviceBuffer(batch * 2 * L * N)
        plan.mul_relin(devk.ptr, L, d1.ptr, d2.ptr, do.ptr, batch)
        outs[(variant, chunk)] = do.to_numpy((batch, 2, L, N))
    assert np.array_equal(outs[(0, 48)], outs[(3, 48)])
    assert np.array_equal(outs[(0, 256)], outs[(3, 48)])
    rs, rb = ref_cpu.RefCtx(N, qs), ref_cpu.RefCtx(N, ch)
    pick = [0, 47, 48, batch - 1]
    want = rs.keyswitch(L, False, evk, ref_cpu.bfv_mul(rs, rb, t, c1[pick], c2[pick]))
    assert np.array_equal(outs[(0, 48)][pick], want)


@pytest.mark.parametrize("bits_s", [(40, 50), (50, 50)])
def test_bfv_mul_relin_prelift_condition(bits_s):
    """tfhe_bfv_mul_relin at N = 2^14 on two-limb rings: with moduli of one size class c2 goes to the fused key switch as
    centred doubles (bit-cast lift); with a 40-bit next to a 50-bit modulus (q_i > 2 q_j) that hand-over is not taken and
    the in-kernel lift with its reduction runs.  Both against the oracle."""
    N, t = 1 << 14, 65537
    qs = H.chain(bits_s[0], 1, N) + [q for q in H.chain(bits_s[1], 2, N) if q not in H.chain(bits_s[0], 1, N)][:1]
    ext = [q for q in H.chain(50, 8, N) if q not in qs][:3]
    ch = qs + ext
    ctx = tf.Context(N, ch)
    plan = tf.BfvPlan(ctx, ctx, t, idx_s=[0, 1])
    rng = np.random.default_rng(sum(bits_s))
    batch = 3
    c1, c2 = H.rand_residues(rng, qs, (batch, 2), N), H.rand_residues(rng, qs, (batch, 2), N)
    evk = H.uniform_evk(rng, qs, 2, N)
    do = tf.DeviceBuffer(batch * 2 * 2 * N)
    devk, d1, d2 = dev(evk), dev(c1), dev(c2)                    # keep the buffers alive across the asynchronous call
    plan.mul_relin(devk.ptr, 2, d1.ptr, d2.ptr, do.ptr, batch)
    rs, rb = ref_cpu.RefCtx(N, qs), ref_cpu.RefCtx(N, ch)
    want = rs.keyswitch(2, False, evk, ref_cpu.bfv_mul(rs, rb, t, c1, c2))
    assert np.array_equal(do.to_numpy((batch, 2, 2, N)), want)


def test_full_size_ntt_properties():
    N, L = 1 << 14, 8
    qs = H.chain(50, L, N)
    ctx = tf.Context(N, qs)
    rng = np.random.default_rng(99)
    batch = 64
    a, b = H.rand_residues(rng, qs, (batch,), N), H.rand_residues(rng, qs, (batch,), N)
    da, db, ds, dn = dev(a), dev(b), tf.DeviceBuffer(a.size), tf.DeviceBuffer(a.size)
    ctx.nntt(da.ptr, dn.ptr, batch, L)
    fa = dn.to_numpy(a.shape)
    ctx.inntt(dn.ptr, dn.ptr, batch, L)
    assert np.array_equal(dn.to_numpy(a.shape), a)                       # round trip
    ctx.add(da.ptr, db.ptr, ds.ptr, batch, L); ctx.nntt(ds.ptr, ds.ptr, batch, L)
    ctx.nntt(db.ptr, db.ptr, batch, L)
    ctx.add(dn.ptr, dn.ptr, dn.ptr, 0, L)                                # empty batch is a no-op
    fb = db.to_numpy(a.shape)
    ref = ref_cpu.RefCtx(N, qs)
    assert np.array_equal(ds.to_numpy(a.shape), ref.pointwise("add", fa, fb))   # linearity
    assert np.array_equal(fa[:2], ref.nntt(a[:2]))                       # oracle on a sub-batch
    # â[k] = a(ψ^(2k+1)) spot check straight from the definition (pow2_cyc_rings.jl:278-294)
    q, psi = qs[0], ctx.psis[0]
    for k in (0, 1, N // 2, N - 1):
        x = pow(psi, 2 * k + 1, q)
        acc = 0
        for c in reversed([int(v) for v in a[0, 0]]):
            acc = (acc * x + c) % q
        assert acc == int(fa[0, 0, k])


# ---------------------------------------------------------------------------------------------------
# CKKS encode / decode on the device (float -- tolerances as stated in SURVEY §8 a18)
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("N,bits,L,scale", [(32, 40, 3, 2**40), (4096, 50, 2, 2**40), (64, 60, 1, 2**30), (32, 40, 3, 2**80),
                                             (16, 50, 2, 12345 * 2**20), (1 << 15, 50, 1, 2**30)])
def test_ckks_encode_decode_match_oracle(N, bits, L, scale):
    """tfhe_ckks_encode / tfhe_ckks_decode against the numpy restatement of ckksencoding.jl:56-97 (oracle/spec.py).
    encode: integer coefficients equal to the oracle's except +-1 at rounding boundaries (different FFT summation
    order; for scales beyond 2^53 a last-place difference of the double is worth scale * ulp); decode: per-slot error
    <= 8 log2(N) eps max|slot|."""
    qs = H.chain(bits, L, N)
    ring = spec.Ring(N, qs)
    ctx = tf.Context(N, qs)
    rng = np.random.default_rng(N + L)
    batch = 1 if N > 4096 else 3
    slots = rng.normal(size=(batch, N // 2)) * 3 + 1j * rng.normal(size=(batch, N // 2))
    slots[0, :4] = [1.0, -2.5, 0.0, 1j]
    mant, exp2 = tf.she.scale_parts(scale)
    dz = tf.DeviceBuffer.from_numpy(np.ascontiguousarray(slots).view(np.uint64))
    dout = tf.DeviceBuffer(batch * L * N)
    ctx.ckks_encode(L, mant, exp2, dz.ptr, dout.ptr, batch)
    res = dout.to_numpy((batch, L, N))
    eps = 2.0 ** -53
    for b in range(batch):
        want = spec.poly_to_ints(spec.ckks_encode(list(slots[b]), ring, scale), ring)
        got = spec.poly_to_ints([list(map(int, l)) for l in res[b]], ring)
        diff = [spec.centred(g - w, ring.Q) for g, w in zip(got, want)]
        # one unit at a rounding boundary; beyond 53 bits of scale a last-place difference of x scales up with it
        allowed = max(1, int(8 * np.log2(N) * eps * np.abs(slots[b]).max() * float(scale)))
        assert max(abs(d) for d in diff) <= allowed, (max(abs(d) for d in diff), allowed)
        if allowed == 1:
            assert sum(1 for d in diff if d) <= max(2, N // 20)      # boundary flips are rare
        # decode of the device encoding against the oracle's decode of the same residues
        dslots = tf.DeviceBuffer(N)
        one = tf.DeviceBuffer.from_numpy(res[b])
        ctx.ckks_decode(L, mant, exp2, one.ptr, dslots.ptr, 1)
        dec = dslots.to_numpy().view(np.complex128)
        ref = spec.ckks_decode([list(map(int, l)) for l in res[b]], ring, scale)
        tol = 8 * np.log2(N) * eps * max(1.0, np.abs(ref).max()) * 4
        assert np.abs(dec - ref).max() <= tol, (np.abs(dec - ref).max(), tol)
        assert np.abs(dec - slots[b]).max() <= N * 2.0 / float(scale) + tol   # round trip: quantisation 1/scale per coefficient
    # decode of arbitrary ring elements (uniform residues: magnitudes up to Q / 2 scale)
    a = H.rand_residues(rng, qs, (1,), N)
    da, dslots = dev(a), tf.DeviceBuffer(N)
    ctx.ckks_decode(L, mant, exp2, da.ptr, dslots.ptr, 1)
    dec = dslots.to_numpy().view(np.complex128)
    ref = spec.ckks_decode([list(map(int, l)) for l in a[0]], ring, scale)
    assert np.abs(dec - ref).max() <= 32 * np.log2(N) * eps * np.abs(ref).max()
    with pytest.raises(AssertionError):
        ctx.ckks_encode(L, 0, 0, dz.ptr, dout.ptr, 1)                 # scale must be positive
    with pytest.raises(tf.UsageError):
        ctx.ckks_encode(L + 1, 1, 40, dz.ptr, dout.ptr, 1)


def test_recycling_allocator_reuses_blocks_without_draining_the_device():
    """csrc/dev_alloc.h: tfhe_free parks a block behind events on the live contexts' streams, tfhe_malloc hands it out again once
    they have completed -- same results, far fewer hipMallocs, and tfhe_alloc_trim gives the cache back."""
    import os
    if os.environ.get("TFHE_ALLOC_CACHE", "1") == "0":
        pytest.skip("the recycling allocator is switched off (TFHE_ALLOC_CACHE=0)")
    N = 4096
    qs = H.chain(50, 2, N)
    ctx, ref = tf.Context(N, qs), ref_cpu.RefCtx(N, qs)
    rng = np.random.default_rng(7)
    a = H.rand_residues(rng, qs, (4,), N)
    want = ref.nntt(a)
    tf.native.check(tf.native.lib().tfhe_alloc_trim())
    s0 = tf.native.alloc_stats()
    for _ in range(200):                                     # allocate / transform / free, never synchronising explicitly
        d, o = dev(a), tf.DeviceBuffer(a.size)
        ctx.nntt(d.ptr, o.ptr, 4, 2)
        del d
        got = o
    assert np.array_equal(got.to_numpy(a.shape), want)
    s1 = tf.native.alloc_stats()
    assert s1["reuses"] - s0["reuses"] >= 300                # 400 allocations, almost all recycled
    assert s1["hip_mallocs"] - s0["hip_mallocs"] <= 100
    del got, o
    ctx.sync()
    tf.native.check(tf.native.lib().tfhe_alloc_trim())
    s2 = tf.native.alloc_stats()
    assert s2["cached_bytes"] == 0


def test_allocator_holds_back_a_host_that_runs_far_ahead():
    """csrc/dev_alloc.h back-pressure: a host that enqueues far ahead of the device frees its temporaries long before their release
    events complete, so nothing is ready when the same sizes are requested again.  Past the soft threshold a request waits for
    the oldest parked block of its size instead of allocating -- the cache stays near the threshold (here 64 MiB, set through
    TFHE_ALLOC_SOFT_GIB in a fresh process) instead of growing by a block per operation, and the results are the same."""
    import os, subprocess, sys, textwrap
    if os.environ.get("TFHE_ALLOC_CACHE", "1") == "0":
        pytest.skip("the recycling allocator is switched off (TFHE_ALLOC_CACHE=0)")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = textwrap.dedent('''
        import sys, numpy as np
        sys.path.insert(0, %r)
        import toyfhe_jl_amd as tf
        from tests import helpers as H
        from oracle import ref_cpu
        N = 1 << 14
        qs = H.chain(50, 4, N)
        ctx, ref = tf.Context(N, qs), ref_cpu.RefCtx(N, qs)
        rng = np.random.default_rng(3)
        a = H.rand_residues(rng, qs, (8,), N)                 # 4 MiB per buffer
        cur = tf.DeviceBuffer.from_numpy(a)
        for i in range(600):                                  # a chain of transforms, each into a fresh buffer; the old one is dropped
            o = tf.DeviceBuffer(a.size)
            if i %% 2 == 0: ctx.nntt(cur.ptr, o.ptr, 8, 4)
            else: ctx.inntt(cur.ptr, o.ptr, 8, 4)
            cur = o
        s = tf.native.alloc_stats()
        got = cur.to_numpy(a.shape)
        assert np.array_equal(got, a), "600 transforms = 300 round trips"
        print("STATS", s["cached_bytes"] + s["live_bytes"], s["hip_mallocs"])
    ''' % root)
    env = dict(os.environ, TFHE_ALLOC_SOFT_GIB="0.0625")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("STATS")][-1].split()
    held, mallocs = int(line[1]), int(line[2])
    assert held <= (64 << 20) + 16 * (4 << 20), held        # the threshold plus a few blocks, not 600 x 4 MiB
    assert mallocs <= 64, mallocs


@pytest.mark.parametrize("special", [True, False])
def test_fused_keyswitch_at_2_13_many_items(special):
    """N = 2^13 on the 256 x 32 geometry: the fused key switch with two workgroups per CU walking several (ciphertext, limb)
    items each, against the one-operation-per-launch path on the whole batch and the oracle on a sub-batch."""
    N, Lk, batch = 1 << 13, 7, 200
    qs = H.chain(50, Lk, N)
    level = Lk - 1 if special else Lk
    ctx, ref = tf.Context(N, qs), ref_cpu.RefCtx(N, qs)
    rng = np.random.default_rng(13 + special)
    evk = H.uniform_evk(rng, qs, Lk, N)
    ct = H.rand_residues(rng, qs[:level], (batch, 3), N)
    devk, dct = dev(evk), dev(ct)
    out, out3 = tf.DeviceBuffer(batch * 2 * level * N), tf.DeviceBuffer(batch * 2 * level * N)
    ctx.keyswitch(Lk, level, special, devk.ptr, Lk, dct.ptr, 3, out.ptr, batch)
    ctx.set_ntt_variant(3)
    ctx.keyswitch(Lk, level, special, devk.ptr, Lk, dct.ptr, 3, out3.ptr, batch)
    ctx.set_ntt_variant(0)
    got = out.to_numpy((batch, 2, level, N))
    assert np.array_equal(got, out3.to_numpy(got.shape))
    pick = [0, 99, batch - 1]
    assert np.array_equal(got[pick], ref.keyswitch(level, special, evk, ct[pick]))


@pytest.mark.parametrize("logn", [11, 12, 14, 15, 16])
def test_galois_many_rows_lds_scatter_path(logn):
    """apply_galois_element on enough rows to take the LDS scatter kernel (N <= 2^14), against the gather kernel (NTT variant 3 keeps
    the one-operation path) on every row and the oracle on a few; several elements incl. conjugation (2N - 1)."""
    N = 1 << logn
    qs = H.chain(50, 3, N)
    ctx, ref = tf.Context(N, qs), ref_cpu.RefCtx(N, qs)
    rng = np.random.default_rng(logn)
    count = 200 if logn <= 14 else 60                                   # 3 limbs each: 600 / 180 rows
    a = H.rand_residues(rng, qs, (count,), N)
    a[0, :, :3] = [[0, 1, q - 1] for q in qs]
    da, d1, d2 = dev(a), tf.DeviceBuffer(a.size), tf.DeviceBuffer(a.size)
    for g in (3, 5, pow(3, N // 4 + 1, 2 * N), 2 * N - 1):
        ctx.set_ntt_variant(0)
        ctx.galois(da.ptr, d1.ptr, g, count, 3)
        ctx.set_ntt_variant(3)
        ctx.galois(da.ptr, d2.ptr, g, count, 3)
        got = d1.to_numpy(a.shape)
        assert np.array_equal(got, d2.to_numpy(a.shape)), g
        assert np.array_equal(got[[0, 37, count - 1]], ref.galois(g, a[[0, 37, count - 1]])), g
    ctx.set_ntt_variant(0)


@pytest.mark.parametrize("N,qspec,terms,count", [(64, "40x3", 5, 3), (4096, "mixed", 64, 2), (4096, "61x2", 7, 2), (1 << 13, "50x4", 70, 1),
                                                  (1 << 16, "mixed", 9, 1)])
def test_dot_equals_the_term_by_term_sum(N, qspec, terms, count):
    """tfhe_dot (sum_k a_k .* b_k in one pass, 128-bit lazy sums reduced every 2^(62 - bits(q)) terms) against exact integer
    arithmetic and against tfhe_mad term by term; > 64 terms take a second launch that accumulates; acc and limb subsets."""
    if qspec == "mixed":
        qs = H.chain(60, 1, N) + H.chain(40, 2, N) + [H.chain(60, 2, N)[1]]
    else:
        bits, n = qspec.split("x")
        qs = H.chain(int(bits), int(n), N)
    L = len(qs)
    ctx = tf.Context(N, qs)
    rng = np.random.default_rng(N + terms)
    a = [H.rand_residues(rng, qs, (count,), N) for _ in range(terms)]
    b = [H.rand_residues(rng, qs, (count,), N) for _ in range(terms)]
    for l in range(L):                                             # worst case for the lazy sums: every product (q - 1)^2
        for k in range(terms):
            a[k][0, l, :3] = qs[l] - 1
            b[k][0, l, :3] = qs[l] - 1
    acc = H.rand_residues(rng, qs, (count,), N)
    da, db, dacc = [dev(x) for x in a], [dev(x) for x in b], dev(acc)
    out = tf.DeviceBuffer(count * L * N)
    ctx.dot(None, [x.ptr for x in da], [x.ptr for x in db], out.ptr, count, L)
    qv = np.array(qs, dtype=object)[None, :, None]
    want = sum(x.astype(object) * y.astype(object) for x, y in zip(a, b)) % qv
    assert np.array_equal(out.to_numpy((count, L, N)), want.astype(np.uint64))
    ctx.dot(dacc.ptr, [x.ptr for x in da], [x.ptr for x in db], out.ptr, count, L)
    assert np.array_equal(out.to_numpy((count, L, N)), ((want + acc.astype(object)) % qv).astype(np.uint64))
    run = dev(acc)                                                 # the reference's loop: acc += a_k * b_k, one call per term
    for k in range(min(terms, 6)):
        ctx.mad(run.ptr, da[k].ptr, db[k].ptr, run.ptr, count, L)
    ctx.dot(dacc.ptr, [x.ptr for x in da[:6]], [x.ptr for x in db[:6]], out.ptr, count, L)
    assert np.array_equal(out.to_numpy((count, L, N)), run.to_numpy((count, L, N)))
    idx = [L - 1, 0]                                               # limb subset, in the caller's order
    sa = [dev(np.ascontiguousarray(x[:, idx])) for x in a[:3]]
    sb = [dev(np.ascontiguousarray(x[:, idx])) for x in b[:3]]
    o2 = tf.DeviceBuffer(count * 2 * N)
    ctx.dot(None, [x.ptr for x in sa], [x.ptr for x in sb], o2.ptr, count, 2, idx)
    w2 = (sum(x[:, idx].astype(object) * y[:, idx].astype(object) for x, y in zip(a[:3], b[:3])) % np.array([qs[i] for i in idx], dtype=object)[None, :, None])
    assert np.array_equal(o2.to_numpy((count, 2, N)), w2.astype(np.uint64))
    with pytest.raises(AssertionError):
        ctx.dot(None, [], [], out.ptr, count, L)


# ---------------------------------------------------------------------------------------------------
# randomised shapes: degree, limb count, modulus sizes (mixed 30..61 bit), batch, level, special prime, component count
# drawn from a seeded generator -- every operation of the path against the C oracle, bit for bit
# ---------------------------------------------------------------------------------------------------
def _random_ring(rng, logn, L):
    N = 1 << logn
    qs, used = [], set()
    for _ in range(L):
        bits = int(rng.choice([30, 36, 40, 45, 50, 50, 55, 60, 61]))
        q = tf.nextprime(2**bits + 1 + 2 * N * int(rng.integers(0, 50)), 1, 2 * N)
        while q in used:
            q = tf.nextprime(q + 2 * N, 1, 2 * N)
        used.add(q)
        qs.append(q)
    return N, qs


@pytest.mark.parametrize("seed", range(24))
def test_random_shapes_against_the_oracle(seed):
    rng = np.random.default_rng(9000 + seed)
    logn = int(rng.choice([3, 6, 9, 10, 11, 12, 13, 14, 15, 16], p=[.05, .05, .05, .1, .1, .1, .15, .2, .1, .1]))
    L = int(rng.integers(2, 7 if logn <= 14 else 5))
    N, qs = _random_ring(rng, logn, L)
    ctx, ref = tf.Context(N, qs), ref_cpu.RefCtx(N, qs)
    batch = int(rng.integers(1, 6 if logn >= 15 else 12))
    # transforms (all limbs, then a random limb subset), round trip
    a = H.rand_residues(rng, qs, (batch,), N)
    f = run_ntt(ctx, a)
    assert np.array_equal(f, ref.nntt(a)), ("nntt", logn, qs)
    assert np.array_equal(run_ntt(ctx, f, inverse=True), a), ("inntt", logn, qs)
    idx = sorted(rng.choice(L, size=int(rng.integers(1, L + 1)), replace=False).tolist())
    sub = np.ascontiguousarray(a[:, idx])
    assert np.array_equal(run_ntt(ctx, sub, idx=idx), ref.nntt(sub, idx=idx)), ("nntt subset", idx)
    # rescale (drop the last limb) and a Galois automorphism
    g = int(rng.choice([3, 5, 2 * N - 1, pow(3, int(rng.integers(1, N)), 2 * N)]))
    da = dev(a)
    dg = tf.DeviceBuffer(a.size)
    ctx.galois(da.ptr, dg.ptr, g, batch, L)
    assert np.array_equal(dg.to_numpy(a.shape), ref.galois(g, a)), ("galois", g)
    dr = tf.DeviceBuffer(batch * (L - 1) * N)
    ctx.rescale(da.ptr, dr.ptr, batch, L)
    assert np.array_equal(dr.to_numpy((batch, L - 1, N)), ref.modswitch(a)), "rescale"
    # key switch / rotation at a random level, with and without the special prime, 2 or 3 components
    special = bool(rng.integers(0, 2))
    maxlevel = L - 1 if special else L
    level = int(rng.integers(1, maxlevel + 1))
    polys = int(rng.choice([2, 3]))
    evk = H.uniform_evk(rng, qs, L, N)
    ct = H.rand_residues(rng, qs[:level], (batch, polys), N)
    devk, dct, dout = dev(evk), dev(ct), tf.DeviceBuffer(batch * 2 * level * N)
    ctx.keyswitch(L, level, special, devk.ptr, L, dct.ptr, polys, dout.ptr, batch)
    assert np.array_equal(dout.to_numpy((batch, 2, level, N)), ref.keyswitch(level, special, evk, ct)), ("keyswitch", level, special, polys, qs)
    ct2 = np.ascontiguousarray(ct[:, :2])
    dct2 = dev(ct2)
    ctx.rotate(L, level, special, devk.ptr, L, g, dct2.ptr, dout.ptr, batch)
    rot = ref.galois(g, ct2.reshape(-1, level, N), idx=range(level)).reshape(ct2.shape)
    assert np.array_equal(dout.to_numpy((batch, 2, level, N)), ref.keyswitch(level, special, evk, rot)), ("rotate", g, level, special)
    # hoisted rotations: the same key under two more Galois elements, against the one-by-one rotations
    gs = [g, int(pow(3, int(rng.integers(1, N)), 2 * N)), 2 * N - 1]
    many = tf.DeviceBuffer(len(gs) * batch * 2 * level * N)
    ctx.rotate_many(L, level, special, [devk.ptr] * len(gs), L, gs, dct2.ptr, many.ptr, batch)
    got = many.to_numpy((len(gs), batch, 2, level, N))
    for r, gr in enumerate(gs):
        ctx.rotate(L, level, special, devk.ptr, L, gr, dct2.ptr, dout.ptr, batch)
        assert np.array_equal(got[r], dout.to_numpy((batch, 2, level, N))), ("rotate_many", gr, level, special)


@pytest.mark.parametrize("seed", range(12))
def test_random_bfv_multiplications_against_the_oracle(seed):
    """tfhe_bfv_mul_relin on random parameter sets: degree 2^8..2^14, 1..5 limbs of mixed sizes, an extension basis that
    contains the ciphertext basis (superset) or is disjoint from it, random plaintext modulus, ragged chunks."""
    rng = np.random.default_rng(7000 + seed)
    logn = int(rng.choice([8, 10, 11, 12, 13, 14], p=[.1, .1, .15, .15, .2, .3]))
    N = 1 << logn
    ns = int(rng.integers(1, 6))
    t = int(rng.choice([2, 257, 65537, 786433, (1 << 20) + 7]))
    superset = bool(rng.integers(0, 2))
    sizes = [int(rng.choice([40, 45, 50, 50, 50, 55, 60])) for _ in range(ns)]
    uniform50 = bool(rng.integers(0, 2))                          # half of the cases on the fp64-size class (the fused kernels)
    if uniform50:
        sizes = [50] * ns
    qs, used = [], set()
    for bits in sizes:
        q = tf.nextprime(2**bits + 1, 1, 2 * N)
        while q in used:
            q = tf.nextprime(q + 2 * N, 1, 2 * N)
        used.add(q); qs.append(q)
    ext_bits = 50 if uniform50 else int(rng.choice([50, 60]))
    # extension primes: enough for Qbig >= Q^2 * t * N * 4 (bfv.jl:47-118 sizes it the same way)
    need = 2 * sum(q.bit_length() for q in qs) + t.bit_length() + logn + 3
    ext, q = [], tf.nextprime(2**ext_bits + 1, 1, 2 * N)
    have = sum(x.bit_length() - 1 for x in qs) if superset else 0
    while have < need:
        if q not in used:
            ext.append(q); used.add(q); have += q.bit_length() - 1
        q = tf.nextprime(q + 2 * N, 1, 2 * N)
    pb = (qs + ext) if superset else ext
    rs, rb = ref_cpu.RefCtx(N, qs), ref_cpu.RefCtx(N, pb)
    if superset:
        cbig = tf.Context(N, pb); csmall = cbig
        plan = tf.BfvPlan(csmall, cbig, t, idx_s=list(range(ns)))
    else:
        csmall, cbig = tf.Context(N, qs), tf.Context(N, pb)
        plan = tf.BfvPlan(csmall, cbig, t)
    batch = int(rng.integers(1, 7))
    plan.set_chunk(int(rng.integers(1, batch + 1)))
    c1, c2 = H.rand_residues(rng, qs, (batch, 2), N), H.rand_residues(rng, qs, (batch, 2), N)
    evk = H.uniform_evk(rng, qs, ns, N)
    d1, d2, devk = dev(c1), dev(c2), dev(evk)
    do3 = tf.DeviceBuffer(batch * 3 * ns * N)
    plan.mul(d1.ptr, d2.ptr, do3.ptr, batch)
    prod = ref_cpu.bfv_mul(rs, rb, t, c1, c2)
    assert np.array_equal(do3.to_numpy((batch, 3, ns, N)), prod), ("mul", logn, qs, pb, t)
    do2 = tf.DeviceBuffer(batch * 2 * ns * N)
    plan.mul_relin(devk.ptr, ns, d1.ptr, d2.ptr, do2.ptr, batch)
    assert np.array_equal(do2.to_numpy((batch, 2, ns, N)), rs.keyswitch(ns, False, evk, prod)), ("mul_relin", logn, qs, pb, t)
