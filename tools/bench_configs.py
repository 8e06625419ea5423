#!/usr/bin/env python3
"""Secondary measurements on the other BASELINE.json configurations (not the bench.py metric): device time per
operation on synthetic residues, 1 GPU.  Every case first checks one ciphertext of its own batch against the C oracle
(bit for bit) and only then times.
  cfg#3  CKKS  N=2^15, 10 limbs @40 bit + special prime: rotate (galois + key switch), rescale       batch 512
  cfg#4        N=2^14,  6 limbs @50 bit + special prime: key switch only                             batch 512 (4096 / 8 GPUs)
  cfg#5  CKKS  N=2^16, the infer.jl ring 60 + 5 x 40 + 60 bit: key switch / rotate / rescale / NTT   batch 64
  cfg#5' same shape on a uniform 50-bit chain (fp64 policy throughout), for comparison
usage: bench_configs.py [scale] [cases]   (scale divides the batches, default 1; cases = positions to run, e.g. 1,3 -- default all)"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import toyfhe_jl_amd as tf
from oracle import ref_cpu            # checker only

HBM_PEAK_GBS = 8000.0
PROFILE = os.environ.get("TFHE_CFG_PROFILE", "0") not in ("", "0")
RECORDS = []                          # one dict per case (bench.py puts them into its JSON line as `other_configs`)


def chain(start, n, N):
    out, p = [], tf.nextprime(start, 1, 2 * N)
    for _ in range(n):
        out.append(p)
        p = tf.nextprime(p + 2 * N, 1, 2 * N)
    return out


def timed(ctx, f, reps=8):
    """best of three rounds after >= 0.1 s of warm-up (the oracle check before each case idles the GPU for seconds and the
    clocks take tens of milliseconds to come back).  TFHE_CFG_PROFILE=1 (tools/pmc_configs.sh, counter passes): three warm calls
    and five counted ones -- the folding script drops the first quarter of every kernel's dispatches."""
    if PROFILE:
        for _ in range(8):
            f()
        ctx.sync()
        return 1.0
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.1:
        f()
        ctx.sync()
    best = 1e9
    for _ in range(3):
        ctx.sync()
        t = time.perf_counter()
        for _ in range(reps):
            f()
        ctx.sync()
        best = min(best, (time.perf_counter() - t) / reps)
    return best


def uniform(ctx, level, count, seed):
    b = tf.DeviceBuffer(count * level * ctx.N)
    ctx.sample_uniform(level, seed, 0, 0, b.ptr, count)
    return b


def rows(buf, shape, k):
    n = int(np.prod(shape[1:]))
    out = np.empty(n, dtype=np.uint64)
    tf.native.check(tf.native.lib().tfhe_memcpy_d2h(out.ctypes.data, buf.ptr + k * n * 8, n * 8))
    return out.reshape((1,) + tuple(shape[1:]))


def keyswitch_case(name, N, qs, b):
    Lk, level = len(qs), len(qs) - 1
    ctx, ref = tf.Context(N, qs), ref_cpu.RefCtx(N, qs)
    evk = uniform(ctx, Lk, Lk * 2, 7)                        # [Lk digits][2][Lk][N]
    ct = uniform(ctx, level, b * 2, 8)
    out = tf.DeviceBuffer(b * 2 * level * N)
    g = pow(3, 2 * N - 1, 2 * N)
    evk_h = evk.to_numpy((Lk, 2, Lk, N))
    k = b // 2
    cin = rows(ct, (b, 2, level, N), k)
    ctx.keyswitch(Lk, level, True, evk.ptr, Lk, ct.ptr, 2, out.ptr, b)
    assert np.array_equal(rows(out, (b, 2, level, N), k), ref.keyswitch(level, True, evk_h, cin)), name + ": key switch differs from the oracle"
    ctx.rotate(Lk, level, True, evk.ptr, Lk, g, ct.ptr, out.ptr, b)
    want = ref.keyswitch(level, True, evk_h, ref.galois(g, cin.reshape(-1, level, N), idx=range(level)).reshape(cin.shape))
    assert np.array_equal(rows(out, (b, 2, level, N), k), want), name + ": rotate differs from the oracle"
    res = tf.DeviceBuffer(b * 2 * (level - 1) * N)
    ctx.rescale(out.ptr, res.ptr, b * 2, level)
    assert np.array_equal(rows(res, (b, 2, level - 1, N), k), ref.modswitch(want.reshape(-1, level, N), idx=range(level)).reshape(1, 2, level - 1, N))
    t_ks = timed(ctx, lambda: ctx.keyswitch(Lk, level, True, evk.ptr, Lk, ct.ptr, 2, out.ptr, b))
    t_rot = timed(ctx, lambda: ctx.rotate(Lk, level, True, evk.ptr, Lk, g, ct.ptr, out.ptr, b))
    t_rs = timed(ctx, lambda: ctx.rescale(ct.ptr, res.ptr, b * 2, level))
    print(f"{name}: N=2^{N.bit_length() - 1}, level {level} (+special), batch {b}: keyswitch {b / t_ks:9.0f}/s  rotate {b / t_rot:9.0f}/s  "
          f"rescale {b / t_rs:9.0f} ct/s   [oracle-checked]", file=sys.stderr)
    # algorithmic bytes per unit (BASELINE.md section 3 / SURVEY 8d): a key switch reads the ciphertext (2 polys) and writes 2
    # polys at the ciphertext's level (the key is shared by the batch); a rescale reads `level` and writes `level - 1` limbs
    ks_bytes, rs_bytes = 4 * level * N * 8, 2 * (2 * level - 1) * N * 8
    RECORDS.append({"config": name, "N": N, "level": level, "special_prime": True, "batch": b, "oracle_checked": True,
                    "moduli_bits": [int(q).bit_length() for q in qs],
                    "keyswitch_per_s": b / t_ks, "rotate_per_s": b / t_rot, "rescale_ct_per_s": b / t_rs,
                    "keyswitch_algorithmic_GBs": b / t_ks * ks_bytes / 1e9, "keyswitch_frac_of_hbm_peak": b / t_ks * ks_bytes / 1e9 / HBM_PEAK_GBS,
                    "rescale_algorithmic_GBs": b / t_rs * rs_bytes / 1e9, "rescale_frac_of_hbm_peak": b / t_rs * rs_bytes / 1e9 / HBM_PEAK_GBS,
                    "limb_ntts_per_keyswitch": level * (level + 1) + 2 * (level + 1)})
    return ctx, ref


def ntt_case(name, N, qs, polys):
    L = len(qs)
    ctx, ref = tf.Context(N, qs), ref_cpu.RefCtx(N, qs)
    a = uniform(ctx, L, polys, 9)
    b2 = tf.DeviceBuffer(polys * L * N)
    ctx.nntt(a.ptr, b2.ptr, polys, L)
    k = polys // 2
    assert np.array_equal(rows(b2, (polys, L, N), k), ref.nntt(rows(a, (polys, L, N), k))), name + ": NTT differs from the oracle"
    c = tf.DeviceBuffer(polys * L * N)
    ctx.inntt(b2.ptr, c.ptr, polys, L)
    assert np.array_equal(rows(c, (polys, L, N), k), rows(a, (polys, L, N), k))
    gb = polys * L * N * 16 / 1e9
    t_f = timed(ctx, lambda: ctx.nntt(a.ptr, b2.ptr, polys, L))
    t_i = timed(ctx, lambda: ctx.inntt(b2.ptr, c.ptr, polys, L))
    print(f"{name}: N=2^{N.bit_length() - 1}, {L} limbs, {polys} polys: NTT fwd {gb / t_f:6.0f} GB/s  inv {gb / t_i:6.0f} GB/s   [oracle-checked]", file=sys.stderr)
    RECORDS.append({"config": "NTT " + name, "N": N, "limbs": L, "polys": polys, "oracle_checked": True,
                    "moduli_bits": [int(q).bit_length() for q in qs], "bytes_per_pass": polys * L * N * 16,
                    "fwd_GBs": gb / t_f, "inv_GBs": gb / t_i, "fwd_frac_of_hbm_peak": gb / t_f / HBM_PEAK_GBS,
                    "inv_frac_of_hbm_peak": gb / t_i / HBM_PEAK_GBS})


def mnist_case(name, logn, sets):
    """BASELINE.json configs[4] end to end: the encrypted CNN of examples/encrypted_mnist.py (infer.jl's circuit on the reference's
    trained weights, synthetic images) with hoisted rotations and the fused diagonal products; third pass, weights encoded by the
    first.  The logits are checked against the float64 model inside run()."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples"))
    import encrypted_mnist as em
    st = {}
    err, rng_, agree = em.run(logn, 0, verbose=False, batches=sets, hoisted=True, repeat=3, fused=True, stats=st)
    assert err < 5e-3 * max(1.0, rng_) and agree > 0.99, name + ": logits differ from the float64 model"
    print(f"{name}: {st['images']} images in {st['eval_s'] * 1e3:.1f} ms = {st['images_per_s']:.0f} images/s  (max logit error {err:.2e}, argmax agreement {agree})")
    RECORDS.append({"config": name, "N": 1 << logn, "ciphertext_sets": sets, "images": st["images"], "images_per_s": st["images_per_s"],
                    "ms_per_pass": st["eval_s"] * 1e3, "max_logit_error": err, "logit_range": rng_, "argmax_agreement": agree,
                    "oracle_checked": "float64 model (infer.jl:55-88)", "includes": "host (Python) time between launches"})


def run(scale=1, out=None, only=None):
    """All cases; returns the list of records (each case asserts one ciphertext against the oracle before it is timed).  A case that
    fails is recorded as {"config": ..., "error": ...} and does not stop the others."""
    del RECORDS[:]

    seen = [0]

    def guarded(f, name, *a):
        seen[0] += 1
        if only and str(seen[0]) not in str(only).split(","):   # cases by position, 1-based ("1,3")
            return
        try:
            f(name, *a)
        except Exception as e:  # noqa: BLE001
            RECORDS.append({"config": name, "error": f"{type(e).__name__}: {e}"})

    N = 1 << 15
    guarded(keyswitch_case, "cfg#3 CKKS N=2^15 10x40-bit + special prime (ckks_rotate.jl path)", N, chain(2**40 + 1, 11, N), max(8, 512 // scale))
    N = 1 << 14
    guarded(keyswitch_case, "cfg#4 N=2^14 6x50-bit + special prime (keyswitch with RNS basis extension)", N, chain(2**50 + 1, 7, N), max(8, 512 // scale))
    N = 1 << 16
    q0, ps = chain(2**60 + 1, 2, N)
    mnist = [q0] + chain(2**40 + 1, 5, N) + [ps]
    guarded(keyswitch_case, "cfg#5 N=2^16 infer.jl ring 60+5x40+60 bit", N, mnist, max(8, 64 // scale))
    guarded(keyswitch_case, "cfg#5' N=2^16 7x50 bit", N, chain(2**50 + 1, 7, N), max(8, 64 // scale))
    guarded(ntt_case, "cfg#5 N=2^16 infer.jl ring", N, mnist, max(8, 128 // scale))
    guarded(ntt_case, "cfg#5' N=2^16 7x50 bit", N, chain(2**50 + 1, 7, N), max(8, 128 // scale))
    N = 1 << 14
    guarded(ntt_case, "N=2^14 60-bit primes", N, chain(2**60 + 1, 8, N), max(8, 1024 // scale))
    guarded(ntt_case, "N=2^14 50-bit primes", N, chain(2**50 + 1, 8, N), max(8, 1024 // scale))
    guarded(mnist_case, "cfg#5 encrypted MNIST inference N=2^16, infer.jl ring (examples/encrypted_mnist.py)", 16, max(1, 16 // scale))
    return list(RECORDS)


if __name__ == "__main__":
    import json
    recs = run(int(sys.argv[1]) if len(sys.argv) > 1 else 1, only=sys.argv[2] if len(sys.argv) > 2 else None)
    for r in recs:
        print(json.dumps(r))
