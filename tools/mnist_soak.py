#!/usr/bin/env python3
"""Soak (r06): the encrypted-MNIST example at N = 2^16, 16 ciphertext sets, 12 passes per circuit shape in ONE process -- last-pass time, logit
error against the float64 model, allocator statistics.  TFHE_ALLOC_DEBUG=1 adds the allocator's account at exit."""
import sys, os
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "examples"))
import encrypted_mnist as m
import toyfhe_jl_amd as tf
for shape, kw in (("reference-shaped", dict(hoisted=False, fused=False)), ("restructured", dict(hoisted=True, fused=True))):
    st = {}
    err, rng_, agree = m.run(16, 0, verbose=False, batches=16, repeat=12, stats=st, **kw)
    a = tf.native.alloc_stats()
    print(shape, "12 passes: last", round(st["eval_s"] * 1e3, 1), "ms  err", err, "agree", agree,
          "live GiB", round(a["live_bytes"] / 2**30, 2), "cached GiB", round(a["cached_bytes"] / 2**30, 2), "mallocs", a["hip_mallocs"], "reuses", a["reuses"], flush=True)
