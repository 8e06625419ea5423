"""Every BASELINE.json configuration at its REAL parameters (ring degree, limb count, modulus sizes, special prime,
batch shape), through the C ABI: the engine runs the full shape, the C oracle checks a sub-batch bit for bit, and
size-independent properties (batch permutation, composition, decrypt-level round trips) cover the rest.

  cfg#1  BFV  N = 2^12, single ~60-bit q          test/bfv_triv.jl:5-22        encrypt -> add -> mul -> decrypt
  cfg#2  BFV  N = 2^14, L = 8                     tests/test_gpu_parity.py::test_full_size_bfv_mul_relin
  cfg#3  CKKS N = 2^15, L = 10 + special, b 512   test/ckks_rotate.jl:43-45    rotate + rescale
  cfg#4  BGV-style key switch N = 2^14, 6 + special, b 512 per GPU            rlwe_she.jl:315-347, modulusraising.jl:35-49
  cfg#5  CKKS N = 2^16, 60 + 5 x 40 + 60-bit ring examples/encrypted_mnist/infer.jl:97-112  rotate, rescale, ct*ct -> relin
"""
import numpy as np
import pytest

import toyfhe_jl_amd as tf
from oracle import ref_cpu
from tests import helpers as H

pytestmark = pytest.mark.gpu


def dev(a):
    return tf.DeviceBuffer.from_numpy(a)


def chain(start, n, N):
    """q_1 = nextprime(start; interval = 2N), q_{i+1} = nextprime(q_i + 2N; interval = 2N) -- the rule of the reference's
    tests (ckks_rotate.jl:9-10, infer.jl:98-105)."""
    out, p = [], tf.nextprime(start, 1, 2 * N)
    for _ in range(n):
        out.append(p)
        p = tf.nextprime(p + 2 * N, 1, 2 * N)
    return out


def device_uniform(ctx, level, count, seed):
    """[count][level][N] uniform residues on limbs 0..level-1, drawn on the device (no host round trip of the full batch)"""
    buf = tf.DeviceBuffer(count * level * ctx.N)
    ctx.sample_uniform(level, seed, 0, 0, buf.ptr, count)
    return buf


def fetch(buf, shape, picks):
    """rows `picks` of a device array whose leading dimension is shape[0]"""
    row = int(np.prod(shape[1:]))
    out = np.empty((len(picks),) + tuple(shape[1:]), dtype=np.uint64)
    for k, b in enumerate(picks):
        tmp = np.empty(row, dtype=np.uint64)
        tf.native.check(tf.native.lib().tfhe_memcpy_d2h(tmp.ctypes.data, buf.ptr + b * row * 8, row * 8))
        out[k] = tmp.reshape(shape[1:])
    return out


# ---------------------------------------------------------------------------------------------------
# cfg#1: test/bfv_triv.jl:5-22 shape
# ---------------------------------------------------------------------------------------------------
def test_cfg1_bfv_triv_shape():
    """BFVParams(53; eval_mult_count = 2) resolves (BASELINE.json configs[0]) to N = 2^12 with one ~60-bit q
    (bfv.jl:104-123: q' = nextprime(2^(bits+1) + 1; interval 2n), a single-prime extension ring above q'^2 t 2^3).  The
    device takes the extension ring in RNS form -- q' plus two more primes of the same chain, 183 bits > 2*61+6+3 -- so the
    tensor never wraps and c*c is the exact rounded product either way.  encrypt -> add -> mul -> decrypt, the 3-element
    product decrypted as b + s c2 + s^2 c3 (bfv_triv.jl:17-22), plus the oracle on the raw product."""
    n, t = 4096, 53
    ch = chain(2**60 + 1, 3, n)
    Rbig = tf.NegacyclicRing(n, ch)
    R = Rbig.crtselect([0])
    assert R.moduli[0].bit_length() == 61
    params = tf.BFVParams(R, Rbig, t, relin_window=1)
    rng = np.random.default_rng(53)
    kp = tf.keygen(rng, params)
    plain = [6] + [0] * (n - 1)
    c = tf.encrypt(rng, kp, plain)
    assert tf.decrypt(kp, c)[0] == 6                               # bfv_triv.jl:18
    y = c * c
    assert len(y) == 3
    dec = tf.decrypt(kp, y)
    assert dec[0] == 0x24 and not any(dec[1:])                     # bfv_triv.jl:20-22
    s = tf.decrypt(kp, c + c)
    assert s[0] == 12 and not any(s[1:])
    d = tf.decrypt(kp, (c + c) * c)
    assert d[0] == 72 % t and not any(d[1:])
    # bit-exact product against the oracle on the same ciphertext
    cin = np.stack([x.to_numpy("primal") for x in c.cs])[None]     # [1][2][1][N]
    rs, rb = ref_cpu.RefCtx(n, ch[:1]), ref_cpu.RefCtx(n, ch)
    want = ref_cpu.bfv_mul(rs, rb, t, cin, cin)
    got = np.stack([x.to_numpy("primal") for x in y.cs])[None]
    assert np.array_equal(got, want)
    # relinearise with the reference's default digit window (relin_window = 1: one key component per bit of q)
    ek = tf.keygen_evalmult(rng, kp.priv)
    assert len(ek.key.key) == ch[0].bit_length()
    z = tf.keyswitch(ek, y)
    dz = tf.decrypt(kp, z)
    assert len(z) == 2 and dz[0] == 36 and not any(dz[1:])


# ---------------------------------------------------------------------------------------------------
# cfg#3: CKKS N = 2^15, 10 x 40-bit + special prime, rotate + rescale, batch 512
# ---------------------------------------------------------------------------------------------------
def test_cfg3_ckks_rotate_rescale_full_shape():
    N, L, batch = 1 << 15, 10, 512
    qs = chain(2**40 + 1, L + 1, N)                                # ckks_rotate.jl:9-10 rule; the last one is the special prime
    Lk = L + 1
    ctx = tf.Context(N, qs)
    ref = ref_cpu.RefCtx(N, qs)
    assert ctx.psis == ref.psis
    rng = np.random.default_rng(315)
    evk = H.uniform_evk(rng, qs, Lk, N)                            # [Lk][2][Lk][N]: L+1 gadget components (a15)
    devk = dev(evk)
    g = pow(3, 2 * N - 1, 2 * N)                                   # steps = 1 (rlwe_she.jl:304)
    ct = device_uniform(ctx, L, batch * 2, 0xC3)                   # [batch][2][L][N]
    rot = tf.DeviceBuffer(batch * 2 * L * N)
    ctx.rotate(Lk, L, True, devk.ptr, Lk, g, ct.ptr, rot.ptr, batch)
    res = tf.DeviceBuffer(batch * 2 * (L - 1) * N)
    ctx.rescale(rot.ptr, res.ptr, batch * 2, L)
    picks = [0, 255, batch - 1]
    cin = fetch(ct, (batch, 2, L, N), picks)
    want_rot = ref.keyswitch(L, True, evk, ref.galois(g, cin.reshape(-1, L, N), idx=range(L)).reshape(cin.shape))
    assert np.array_equal(fetch(rot, (batch, 2, L, N), picks), want_rot)
    want_res = ref.modswitch(want_rot.reshape(-1, L, N), idx=range(L)).reshape(len(picks), 2, L - 1, N)
    assert np.array_equal(fetch(res, (batch, 2, L - 1, N), picks), want_res)
    # plain key switch of the same batch, 3-element input as well (relinearisation shape)
    ct3 = device_uniform(ctx, L, 24 * 3, 0xC33)
    out3 = tf.DeviceBuffer(24 * 2 * L * N)
    ctx.keyswitch(Lk, L, True, devk.ptr, Lk, ct3.ptr, 3, out3.ptr, 24)
    p3 = [0, 23]
    assert np.array_equal(fetch(out3, (24, 2, L, N), p3), ref.keyswitch(L, True, evk, fetch(ct3, (24, 3, L, N), p3)))
    # a lower level of the same key (downswitch_keyelement, modulusraising.jl:43-49): level 9 after the rescale
    lo = tf.DeviceBuffer(batch * 2 * (L - 1) * N)
    ctx.rotate(Lk, L - 1, True, devk.ptr, Lk, g, res.ptr, lo.ptr, batch)
    rin = fetch(res, (batch, 2, L - 1, N), picks[:2])
    want_lo = ref.keyswitch(L - 1, True, evk, ref.galois(g, rin.reshape(-1, L - 1, N), idx=range(L - 1)).reshape(rin.shape))
    assert np.array_equal(fetch(lo, (batch, 2, L - 1, N), picks[:2]), want_lo)


def test_cfg3_ckks_rotate_decrypts_at_full_degree():
    """ckks_rotate.jl:43-45 with its plaintext (1:N/2, plain[0] += im) at N = 2^15 on the 10 + 1 limb ModulusRaised ring:
    rotate -> rescale -> decrypt ~ circshift(plain, 1).  The reference's rescale takes c_last as its unsigned representative
    (crt.jl:215-220), i.e. floor, whose rounding error has mean 1/2 per coefficient: in the slots that is
    1/2 * (1 + s(z)) / (1 - z), largest at the roots z next to 1 -- about 5 * 2^20 / scale' here.  The input scale is
    2^75 so that the rescaled scale (2^35) leaves that bias at 2e-4."""
    N, L = 1 << 15, 10
    R = tf.NegacyclicRing(N, chain(2**40 + 1, L + 1, N))
    params = tf.ModulusRaised(tf.CKKSParams(R, 0, 3.2))
    rng = tf.DeviceRng(315)
    kp = tf.keygen(rng, params)
    scale = 2**75
    plain = np.arange(1, N // 2 + 1).astype(complex)
    plain[0] += 1j
    c = tf.encrypt(rng, kp, tf.ckks_encode(plain, params.R_cipher(), scale), scale=scale)
    gk = tf.keygen_galois(rng, kp.priv, steps=1)
    assert len(gk.key.key) == L + 1
    r = tf.rotate(gk, c)
    got = tf.ckks_decode(tf.decrypt(kp, r), scale)
    assert np.abs(got - np.roll(plain, 1)).max() < 1e-6
    r2 = tf.modswitch(r)
    assert r2.ring().L == L - 1
    got2 = tf.ckks_decode(tf.decrypt(kp, r2), r2.scale)
    assert np.abs(got2 - np.roll(plain, 1)).max() < 1e-2          # scale 2^75 / 2^40 = 2^35 after the rescale


# ---------------------------------------------------------------------------------------------------
# cfg#4: key switch N = 2^14, 6 x 50-bit + special prime (7 working limbs), batch 512 per GPU
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("polys", [2, 3])
def test_cfg4_keyswitch_full_shape(polys):
    N, L, batch = 1 << 14, 6, 512                                  # 4096 over 8 GPUs = 512 per GPU
    qs = H.chain(50, L + 1, N)
    Lk = L + 1
    ctx = tf.Context(N, qs)
    ref = ref_cpu.RefCtx(N, qs)
    rng = np.random.default_rng(46 + polys)
    evk = H.uniform_evk(rng, qs, Lk, N)
    devk = dev(evk)
    ct = device_uniform(ctx, L, batch * polys, 0xC4 + polys)
    out = tf.DeviceBuffer(batch * 2 * L * N)
    ctx.keyswitch(Lk, L, True, devk.ptr, Lk, ct.ptr, polys, out.ptr, batch)
    picks = [0, 1, 255, 256, batch - 1]
    want = ref.keyswitch(L, True, evk, fetch(ct, (batch, polys, L, N), picks))
    assert np.array_equal(fetch(out, (batch, 2, L, N), picks), want)
    # the one-operation-per-launch path (variant 3) must agree on the whole batch
    ctx.set_ntt_variant(3)
    out3 = tf.DeviceBuffer(batch * 2 * L * N)
    ctx.keyswitch(Lk, L, True, devk.ptr, Lk, ct.ptr, polys, out3.ptr, batch)
    ctx.set_ntt_variant(0)
    assert np.array_equal(out3.to_numpy(), out.to_numpy())
    # without the special prime on the same 7-limb ring (the bfv_keyswitch.jl-style relinearisation): 7 working limbs
    ct7 = device_uniform(ctx, Lk, 300 * polys, 0xC47 + polys)
    out7 = tf.DeviceBuffer(300 * 2 * Lk * N)
    ctx.keyswitch(Lk, Lk, False, devk.ptr, Lk, ct7.ptr, polys, out7.ptr, 300)
    p7 = [0, 128, 299]
    assert np.array_equal(fetch(out7, (300, 2, Lk, N), p7), ref.keyswitch(Lk, False, evk, fetch(ct7, (300, polys, Lk, N), p7)))


def test_cfg4_bgv_product_and_keyswitch_on_genuine_ciphertexts():
    """BGV ciphertexts (t e noise, bgv.jl:27-34) on the cfg#4 ring: c*c (3 elements, tensor in the NTT domain) decrypts to the
    products; the key switch of that product with a genuine s^2 -> s key (special prime) is bit-identical to the oracle's.
    (The reference's BGV has no keyswitch of its own -- bgv.jl defines no relin_window -- so there is no decrypt-level
    statement to mirror after the switch: floor(./P) is not a BGV-compatible rounding.)"""
    N, L, t = 1 << 14, 6, 257
    qs = H.chain(50, L + 1, N)
    R = tf.NegacyclicRing(N, qs)
    params = tf.ModulusRaised(tf.BGVParams(R, t))
    rng = tf.DeviceRng(46)
    kp = tf.keygen(rng, params)
    ms = [[m] + [0] * (N - 1) for m in (6, 11, 200)]
    # encrypt = encryption of zero (special limb dropped, modulusraising.jl:23-26) + the plaintext in the ciphertext ring
    c = tf.she.encrypt_zero(rng, kp.pub, batch=3) + params.R_cipher()(ms)
    assert [d[0] for d in tf.decrypt(kp, c)] == [6, 11, 200]
    y = c * c
    assert len(y) == 3 and [d[0] for d in tf.decrypt(kp, y)] == [36, 121, 200 * 200 % t]
    ek = tf.keygen_evalmult(rng, kp.priv)
    z = tf.keyswitch(ek, y)
    assert len(z) == 2
    cin = np.stack([x.to_numpy("primal") for x in y.cs], axis=1)   # [batch 3][polys 3][L][N]
    ref = ref_cpu.RefCtx(N, qs)
    evk = ek.key.packed().to_numpy((L + 1, 2, L + 1, N))
    want = ref.keyswitch(L, True, evk, cin)
    got = np.stack([x.to_numpy("primal") for x in z.cs], axis=1)
    assert np.array_equal(got, want)


# ---------------------------------------------------------------------------------------------------
# cfg#5: CKKS N = 2^16 on the mixed 60 + 5 x 40 + 60-bit ring of infer.jl:97-112
# ---------------------------------------------------------------------------------------------------
def mnist_ring_moduli(N):
    q0, ps = chain(2**60 + 1, 2, N)                                # infer.jl:98-99
    return [q0] + chain(2**40 + 1, 5, N) + [ps]                    # infer.jl:101-107: (q0, q1..q5, ps)


def test_cfg5_ckks_mnist_ring_rotate_rescale_square_relin():
    N, batch = 1 << 16, 16
    qs = mnist_ring_moduli(N)
    Lk, L = 7, 6
    assert [q.bit_length() for q in qs] == [61, 41, 41, 41, 41, 41, 61]
    ctx = tf.Context(N, qs)
    ref = ref_cpu.RefCtx(N, qs)
    assert ctx.psis == ref.psis
    rng = np.random.default_rng(516)
    evk = H.uniform_evk(rng, qs, Lk, N)
    devk = dev(evk)
    g = pow(3, 2 * N - 1, 2 * N)
    picks = [0, batch - 1]
    # rotate at level 6 (infer.jl:146), rescale (infer.jl:130)
    ct = device_uniform(ctx, L, batch * 2, 0xC5)
    rot = tf.DeviceBuffer(batch * 2 * L * N)
    ctx.rotate(Lk, L, True, devk.ptr, Lk, g, ct.ptr, rot.ptr, batch)
    cin = fetch(ct, (batch, 2, L, N), picks)
    want_rot = ref.keyswitch(L, True, evk, ref.galois(g, cin.reshape(-1, L, N), idx=range(L)).reshape(cin.shape))
    assert np.array_equal(fetch(rot, (batch, 2, L, N), picks), want_rot)
    res = tf.DeviceBuffer(batch * 2 * (L - 1) * N)
    ctx.rescale(rot.ptr, res.ptr, batch * 2, L)
    want_res = ref.modswitch(want_rot.reshape(-1, L, N), idx=range(L)).reshape(len(picks), 2, L - 1, N)
    assert np.array_equal(fetch(res, (batch, 2, L - 1, N), picks), want_res)
    # x -> x*x -> keyswitch(ek, .) -> modswitch at level 5 (infer.jl:134-137)
    lv = L - 1
    dual = tf.DeviceBuffer(batch * 2 * lv * N)
    ctx.nntt(res.ptr, dual.ptr, batch * 2, lv)
    ten = tf.DeviceBuffer(batch * 3 * lv * N)
    ctx.tensor(dual.ptr, dual.ptr, ten.ptr, batch, lv)
    ctx.inntt(ten.ptr, ten.ptr, batch * 3, lv)
    want_ten = ref.enc_mul(want_res, want_res, idx=range(lv))
    assert np.array_equal(fetch(ten, (batch, 3, lv, N), picks), want_ten)
    rel = tf.DeviceBuffer(batch * 2 * lv * N)
    ctx.keyswitch(Lk, lv, True, devk.ptr, Lk, ten.ptr, 3, rel.ptr, batch)
    want_rel = ref.keyswitch(lv, True, evk, want_ten)
    assert np.array_equal(fetch(rel, (batch, 2, lv, N), picks), want_rel)
    out = tf.DeviceBuffer(batch * 2 * (lv - 1) * N)
    ctx.rescale(rel.ptr, out.ptr, batch * 2, lv)
    want_out = ref.modswitch(want_rel.reshape(-1, lv, N), idx=range(lv)).reshape(len(picks), 2, lv - 1, N)
    assert np.array_equal(fetch(out, (batch, 2, lv - 1, N), picks), want_out)


def test_cfg5_ckks_mnist_ring_decrypt_level():
    """genuine encryptions on the infer.jl ring at N = 2^16: x*x -> relinearise (special prime) -> rescale, and a rotation,
    decrypted and decoded."""
    N = 1 << 16
    R = tf.NegacyclicRing(N, mnist_ring_moduli(N))
    params = tf.ModulusRaised(tf.CKKSParams(R, 0, 3.2))
    rng = tf.DeviceRng(516)
    kp = tf.keygen(rng, params)
    scale = 2**40                                                  # infer.jl:114
    x = np.linspace(-1.5, 1.5, N // 2).astype(complex)
    c = tf.encrypt(rng, kp, tf.ckks_encode(x, params.R_cipher(), scale), scale=scale)
    ek = tf.keygen_evalmult(rng, kp.priv)
    sq = tf.modswitch(tf.keyswitch(ek, c * c))
    assert sq.ring().L == 5
    got = tf.ckks_decode(tf.decrypt(kp, sq), sq.scale)
    assert np.abs(got - x * x).max() < 1e-4
    gk = tf.keygen_galois(rng, kp.priv, steps=1)
    r = tf.rotate(gk, sq)
    got = tf.ckks_decode(tf.decrypt(kp, r), r.scale)
    assert np.abs(got - np.roll(x * x, 1)).max() < 1e-4


# ---------------------------------------------------------------------------------------------------
# whole-batch cross-checks (an item-walk bug at ciphertext 300 of a sub-block walk must not pass): the
# one-operation-per-launch path (NTT variant 3) against the default fused / sub-block path on EVERY ciphertext, plus the
# batch-permutation property
# ---------------------------------------------------------------------------------------------------
def _whole_batch_keyswitch_crosscheck(N, qs, L, batch, seed, g):
    Lk = len(qs)
    ctx = tf.Context(N, qs)
    rng = np.random.default_rng(seed)
    devk = dev(H.uniform_evk(rng, qs, Lk, N))
    ct = device_uniform(ctx, L, batch * 2, seed)
    n = batch * 2 * L * N
    out, out3 = tf.DeviceBuffer(n), tf.DeviceBuffer(n)
    ctx.rotate(Lk, L, True, devk.ptr, Lk, g, ct.ptr, out.ptr, batch)
    ctx.set_ntt_variant(3)
    ctx.rotate(Lk, L, True, devk.ptr, Lk, g, ct.ptr, out3.ptr, batch)
    ctx.set_ntt_variant(0)
    a = out.to_numpy((batch, 2, L, N))
    assert np.array_equal(out3.to_numpy((batch, 2, L, N)), a)
    del out3
    # batch independence: the permuted batch gives the permuted result
    perm = rng.permutation(batch)
    ctp = dev(ct.to_numpy((batch, 2, L, N))[perm])
    ctx.rotate(Lk, L, True, devk.ptr, Lk, g, ctp.ptr, out.ptr, batch)
    assert np.array_equal(out.to_numpy((batch, 2, L, N)), a[perm])
    # rescale of the permuted batch: the oracle on ciphertexts spread over the whole batch (crt.jl:215-220, unsigned c_last)
    res = tf.DeviceBuffer(batch * 2 * (L - 1) * N)
    ctx.rescale(out.ptr, res.ptr, batch * 2, L)
    ref = ref_cpu.RefCtx(N, qs)
    picks = sorted({0, batch // 3, (2 * batch) // 3 + 1, batch - 1})
    want = ref.modswitch(a[perm][picks].reshape(-1, L, N), idx=range(L)).reshape(len(picks), 2, L - 1, N)
    assert np.array_equal(fetch(res, (batch, 2, L - 1, N), picks), want)


def test_cfg3_whole_batch_crosscheck_and_permutation():
    N = 1 << 15
    _whole_batch_keyswitch_crosscheck(N, chain(2**40 + 1, 11, N), 10, 512, 0xC3A, pow(3, 2 * N - 1, 2 * N))


def test_cfg5_whole_batch_crosscheck_and_permutation():
    N = 1 << 16
    _whole_batch_keyswitch_crosscheck(N, mnist_ring_moduli(N), 6, 16, 0xC5A, pow(3, 2 * N - 1, 2 * N))


# ---------------------------------------------------------------------------------------------------
# cfg#2 at the bench shape itself: batch 1024 in 256-ciphertext chunks, the oracle on both sides of every chunk boundary
# ---------------------------------------------------------------------------------------------------
def test_cfg2_bench_shape_batch_1024():
    N, L, t, batch = 1 << 14, 8, 65537, 1024
    ch = H.chain(50, 17, N)
    qs = ch[:L]
    ctx = tf.Context(N, ch)
    plan = tf.BfvPlan(ctx, ctx, t, idx_s=list(range(L)))
    d1, d2 = device_uniform(ctx, L, batch * 2, 0xC21), device_uniform(ctx, L, batch * 2, 0xC22)
    rng = np.random.default_rng(1024)
    evk = H.uniform_evk(rng, qs, L, N)
    devk = dev(evk)
    do = tf.DeviceBuffer(batch * 2 * L * N)
    plan.mul_relin(devk.ptr, L, d1.ptr, d2.ptr, do.ptr, batch)      # default chunk (256): what bench.py times
    picks = [0, 255, 256, 511, 512, 1023]
    c1, c2 = fetch(d1, (batch, 2, L, N), picks), fetch(d2, (batch, 2, L, N), picks)
    rs, rb = ref_cpu.RefCtx(N, qs), ref_cpu.RefCtx(N, ch)
    want = rs.keyswitch(L, False, evk, ref_cpu.bfv_mul(rs, rb, t, c1, c2))
    assert np.array_equal(fetch(do, (batch, 2, L, N), picks), want)
    # every ciphertext of the batch: another chunking (96: ragged last chunk) must give the same bits
    do2 = tf.DeviceBuffer(batch * 2 * L * N)
    plan.set_chunk(96)
    plan.mul_relin(devk.ptr, L, d1.ptr, d2.ptr, do2.ptr, batch)
    plan.set_chunk(0)
    a = do.to_numpy((batch, 2, L, N))
    assert np.array_equal(do2.to_numpy((batch, 2, L, N)), a)
    # and commutativity on the whole batch
    plan.mul_relin(devk.ptr, L, d2.ptr, d1.ptr, do2.ptr, batch)
    assert np.array_equal(do2.to_numpy((batch, 2, L, N)), a)
