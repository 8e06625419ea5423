#!/usr/bin/env python3
"""Encrypted CNN inference in the shape of the reference's examples/encrypted_mnist/infer.jl,
driven through the host mirror with every ring operation on the MI355X.

The reference evaluates its trained Flux model (examples/encrypted_mnist/mnist_conv.bson) on MNIST.  The weights are
exported from that file by tools/export_mnist_bson.py (a BSON reader, run in the build container) into
tests/golden/mnist_conv.npz and used here by default (`--model synthetic` draws a random model of the same architecture);
the MNIST images are not on disk, so the batch is synthetic (seeded; sparse stroke-like 28x28 images in [0, 1]).  The
homomorphic result is checked against the same arithmetic in float64 (infer.jl:55-88 `do_encrypted_inference`):

    conv 7x7 stride 3, 4 channels (49 ciphertexts x plaintext scalars)  ->  + bias  ->  rescale        infer.jl:127-131
    square + relinearise + rescale                                                                      :136-138
    dense 256 -> 64 as 4 diagonal-packed 64x64 products (63 rotations by 64 slots each)  + bias, rescale :142-165
    square + relinearise + rescale                                                                      :167-169
    dense 64 -> 10 (zero-padded to 64x64, 63 rotations) + bias                                          :171-179

Packing: N/2 slots = 64 windows x B images (B = N/128; the reference uses N = 2^13, B = 64), slot = window * B + image.

  python examples/encrypted_mnist.py [--logn 13] [--seed 0]"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import toyfhe_jl_amd as tf  # noqa: E402


def public_preprocess(batch):
    """infer.jl:57-64: I[i][j] = matrix [image k][window l] of pixel (i, j) of each 7x7 window (stride 3, 8x8 windows)"""
    B = batch.shape[0]
    I = np.empty((7, 7, B, 64))
    for a in range(8):
        for b in range(8):
            I[:, :, :, a + 8 * b] = batch[:, a * 3:a * 3 + 7, b * 3:b * 3 + 7].transpose(1, 2, 0)
    return I


def plain_matmul(W, x):
    """infer.jl:44-46 on plaintext: sum_k diag(circshift(W, (0, k-1))) .* circshift(x, (k-1, 0)) == W @ x"""
    n = x.shape[0]
    out = np.zeros_like(x)
    for k in range(n):
        d = np.array([W[i, (i - k) % n] for i in range(n)])
        out += d[:, None] * np.roll(x, k, axis=0)
    return out


def plain_model(model, batch):
    I = public_preprocess(batch)                                   # [7][7][B][64]
    conv = [sum(I[i, j] * model["conv_w"][i, j, ch] for i in range(7) for j in range(7)) + model["conv_b"][ch] for ch in range(4)]
    sq1 = [(c ** 2).T for c in conv]                               # [64 windows][B]
    fq1 = sum(plain_matmul(model["fq1_w"][:, 64 * i:64 * (i + 1)], sq1[i]) for i in range(4)) + model["fq1_b"][:, None]
    sq2 = fq1 ** 2
    W2 = np.vstack([model["fq2_w"], np.zeros((54, 64))])
    return (plain_matmul(W2, sq2) + np.concatenate([model["fq2_b"], np.zeros(54)])[:, None])[:10]


def encrypted_matmul(gk, W, x, B, cache=None, fused=False):
    """infer.jl:140-149: diagonal method; `rotate` by B slots moves every window's value to the next window.
    gk = one Galois key (the reference's loop: 63 chained rotations by B slots) or a list of 63 keys for the steps B, 2B, ...,
    63B (hoisted: every rotation starts from x and they share one digit decomposition, tfhe_rotate_many -- fewer transforms
    and 63 times less rotation noise, at the price of 63 keys)."""
    n = 64
    K = x[0].count

    def diag(k):
        # the k-th generalised diagonal as a plaintext at x's scale; encoded once per (matrix, k) when a cache is given
        # (a deployed model encodes its weights once, not per inference)
        key = (id(W), k, x.ring().L, x.scale)
        if cache is not None and key in cache:
            return cache[key]
        v = np.repeat(np.array([W[i, (i - k) % n] for i in range(n)]), B)
        if cache is None:
            return v
        pt = tf.ckks_encode(np.repeat(v[None].astype(np.complex128), K, axis=0), x.ring(), x.scale)
        pt.coeffs_dual()
        cache[key] = pt
        return pt
    if isinstance(gk, (list, tuple)) and cache is not None and fused:
        # the whole product in one device call (tfhe_matmul_diag): the diagonals are single plaintexts shared by the batch
        key = (id(W), "fused", x.ring().L, x.scale)
        if key not in cache:
            vs = np.stack([np.repeat(np.array([W[i, (i - k) % n] for i in range(n)]), B) for k in range(n)]).astype(np.complex128)
            enc = tf.ckks_encode(vs, x.ring(), x.scale)                       # the 64 diagonals as one stacked plaintext element
            enc.coeffs_dual()
            cache[key] = enc
        return tf.matmul_diag(gk, cache[key], x)
    if isinstance(gk, (list, tuple)):
        rots = [x] + list(tf.rotate_many(gk, x))
    else:
        rots = [x]
        for k in range(1, n):
            rots.append(tf.rotate(gk, rots[-1]))
    if cache is not None:      # plaintexts are ring elements already: the whole accumulation is one device pass per component
        return tf.CipherText.dot_plain(rots, [diag(k) for k in range(n)])
    result = rots[0].mul_plain(diag(0))
    for k in range(1, n):
        result = result + rots[k].mul_plain(diag(k))
    return result


def add_bias(ct, x, cache=None, name=None):
    """ct .+ bias (infer.jl: the `.+ b` of every layer).  With a cache the bias is encoded once per (layer, level, scale) like the
    weight plaintexts -- each encoding is a host-to-device copy the pass would otherwise wait on -- and added to the first
    component as CipherText.add_plain does."""
    if cache is None:
        return ct.add_plain(x)
    key = ("bias", name, ct.ring().L, ct.scale)
    if key not in cache:
        n2 = ct.ring().N // 2
        v = np.full(n2, x, dtype=np.complex128) if np.isscalar(x) else np.asarray(x, dtype=np.complex128)
        cache[key] = tf.ckks_encode(v, ct.ring(), ct.scale)
    return tf.CipherText(ct.params, (ct.cs[0] + cache[key],) + tuple(ct.cs[1:]), ct.scale)


GOLDEN_MODEL = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "mnist_conv.npz")


def load_model(path=GOLDEN_MODEL):
    """the reference's trained weights; infer.jl:118-119 flips the Flux convolution kernel in both spatial dimensions"""
    d = np.load(path)
    m = {k: d[k].astype(np.float64) for k in d.files}
    m["conv_w"] = m["conv_w"][::-1, ::-1, :].copy()
    return m


def run(logn=13, seed=0, verbose=True, model="reference", batches=1, hoisted=False, repeat=1, fused=False, return_logits=False, stats=None):
    """`batches` = K ciphertext sets evaluated together (K * B images): every ring element carries a leading batch dimension
    of K, so each device call covers K ciphertexts (the batch the engine shards across GPUs)."""
    N = 1 << logn
    B = N // 128                                                   # images per ciphertext
    K = int(batches)
    rs = np.random.default_rng(seed)
    if model == "reference":
        model = load_model()
    else:
        model = {"conv_w": rs.normal(0, 0.15, (7, 7, 4)), "conv_b": rs.normal(0, 0.1, 4),
                 "fq1_w": rs.normal(0, 0.06, (64, 256)), "fq1_b": rs.normal(0, 0.1, 64),
                 "fq2_w": rs.normal(0, 0.1, (10, 64)), "fq2_b": rs.normal(0, 0.1, 10)}
    batch = (rs.random((K * B, 28, 28)) < 0.15) * rs.random((K * B, 28, 28))   # sparse strokes: MNIST-like pixel statistics
    want = plain_model(model, batch)

    # infer.jl:97-112: q0 (60 bit), five 40-bit primes, 60-bit special prime; ModulusRaised CKKS, sigma 3.2
    q0 = tf.nextprime(2**60 + 1, 1, 2 * N)
    ps = tf.nextprime(q0 + 2 * N, 1, 2 * N)
    qs = [tf.nextprime(2**40 + 1, 1, 2 * N)]
    for _ in range(4):
        qs.append(tf.nextprime(qs[-1] + 2 * N, 1, 2 * N))
    ring = tf.NegacyclicRing(N, [q0] + qs + [ps])
    params = tf.ModulusRaised(tf.CKKSParams(ring, 0, 3.2))
    rng = tf.DeviceRng(seed + 1)                                   # all sampling on the GPU
    t0 = time.perf_counter()
    kp = tf.keygen(rng, params)
    ek = tf.keygen_evalmult(rng, kp.priv)
    if hoisted:
        gk = [tf.keygen_galois(rng, kp.priv, steps=k * B) for k in range(1, 64)]
    else:
        gk = tf.keygen_galois(rng, kp.priv, steps=B)               # infer.jl:134 (steps = 64 there)
    scale = 2**40
    I = public_preprocess(batch)
    cring = params.R_cipher()
    def slots(m):                                                  # [K*B images][64 windows] -> [K][N/2], slot = window * B + image
        return m.reshape(K, B, 64).transpose(0, 2, 1).reshape(K, -1)
    C = [[tf.encrypt(rng, kp, tf.ckks_encode(slots(I[i, j]), cring, scale), scale=scale) for j in range(7)] for i in range(7)]
    t_setup = time.perf_counter() - t0

    fq1_blocks = [np.ascontiguousarray(model["fq1_w"][:, 64 * i:64 * (i + 1)]) for i in range(4)]   # stable ids for the cache
    W2 = np.vstack([model["fq2_w"], np.zeros((54, 64))])
    cache = {} if (repeat > 1 or fused) else None                 # pre-encoded weight plaintexts, filled by the first pass
    for rep in range(repeat):
      ev0, ev1 = tf.Event(), tf.Event()                             # device-side span of the pass next to the host's clock
      ev0.record(cring.ctx)
      t0 = time.perf_counter()
      conved = []
      if fused:        # the 4 x 49 scalar-weighted terms in one device pass per component over the 49 inputs (tfhe_lincomb_many)
          accs = tf.CipherText.lincomb_many([C[i][j] for i in range(7) for j in range(7)],
                                            [[float(model["conv_w"][i, j, ch]) for i in range(7) for j in range(7)] for ch in range(4)])
      for ch in range(4):
          if fused:
              acc = accs[ch]
          else:
              acc = None
              for i in range(7):
                  for j in range(7):
                      term = C[i][j].mul_plain(float(model["conv_w"][i, j, ch]))
                      acc = term if acc is None else acc + term
          conved.append(tf.modswitch(add_bias(acc, float(model["conv_b"][ch]), cache, ("conv", ch))))
      if fused:    # the four channels' squares, relinearisations and rescales as ONE batch of 4 K ciphertexts
          big = tf.CipherText.concat(conved)
          sq1 = tf.modswitch(tf.keyswitch(ek, big * big)).split([K] * 4)
      else:
          sq1 = [tf.modswitch(tf.keyswitch(ek, c * c)) for c in conved]
      fq1 = None
      for i in range(4):
          part = encrypted_matmul(gk, fq1_blocks[i], sq1[i], B, cache, fused)
          fq1 = part if fq1 is None else fq1 + part
      fq1 = tf.modswitch(add_bias(fq1, np.repeat(model["fq1_b"], B), cache, "fq1"))
      sq2 = tf.modswitch(tf.keyswitch(ek, fq1 * fq1))
      res = add_bias(encrypted_matmul(gk, W2, sq2, B, cache, fused), np.repeat(np.concatenate([model["fq2_b"], np.zeros(54)]), B), cache, "fq2")
      ev1.record(res[0].ring.ctx)
      t_enq = time.perf_counter() - t0                                # every launch of the circuit is enqueued; the device may still be running
      dec = tf.ckks_decode(tf.decrypt(kp, res), res.scale).real       # [K][N/2]
      got = dec.reshape(K, 64, B)[:, :10].transpose(1, 0, 2).reshape(10, K * B)
      t_eval = time.perf_counter() - t0
    err = float(np.abs(got - want).max())
    if stats is not None:   # (tools/bench_configs.py)
        stats.update(images=K * B, eval_s=t_eval, setup_s=t_setup, images_per_s=K * B / t_eval, host_enqueue_s=t_enq,
                     device_span_s=ev0.elapsed_ms(ev1) * 1e-3)
    if verbose:
        print(f"N=2^{logn}, {K} x {B} images: setup {t_setup:.2f} s, encrypted evaluation {t_eval:.2f} s = {K * B / t_eval:.0f} images/s "
              f"(49 encrypted inputs, 5 x 63 {'hoisted ' if hoisted else ''}rotations, 5 relinearisations per ciphertext set"
              f"{'; pass ' + str(repeat) + ' with the weight plaintexts encoded by pass 1' if repeat > 1 else ''})")
        print(f"max |encrypted - plaintext| over the 10 x {K * B} logits: {err:.3e}   (logit range +-{np.abs(want).max():.2f})")
        print("argmax agreement:", float((got.argmax(0) == want.argmax(0)).mean()))
    if return_logits:
        return err, float(np.abs(want).max()), float((got.argmax(0) == want.argmax(0)).mean()), got
    return err, float(np.abs(want).max()), float((got.argmax(0) == want.argmax(0)).mean())


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--logn", type=int, default=13)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--model", default="reference", choices=["reference", "synthetic"])
    ap.add_argument("--batches", type=int, default=1, help="ciphertext sets evaluated together (images = batches * N/128)")
    ap.add_argument("--hoisted", action="store_true", help="63 Galois keys and tfhe_rotate_many instead of 63 chained rotations")
    ap.add_argument("--repeat", type=int, default=1, help="evaluate this many times; from the second pass on the weight plaintexts are cached")
    ap.add_argument("--fused", action="store_true",
                    help="with --hoisted: each matrix product as one tfhe_matmul_diag call and each convolution channel as one tfhe_lincomb per component")
    a = ap.parse_args()
    run(a.logn, a.seed, model=a.model, batches=a.batches, hoisted=a.hoisted, repeat=a.repeat, fused=a.fused)
