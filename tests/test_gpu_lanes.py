"""Two lanes (r06, toyfhe_hip.hip lanes_t): on a ring that mixes fp64-size moduli with 60-bit ones (the reference's CKKS rings,
infer.jl:97-112) the launches of the two arithmetic policies run on two streams forked from / joined into the context's stream.
The single-call parity of every such path against the oracle is in tests/test_gpu_parity.py and tests/test_gpu_configs.py (they run
with the lanes on); here: the ORDERING the fork / join must keep -- back-to-back calls without a host wait in between, buffers released
while the calls that use them are still queued (the allocator sees one stream), a second context on its own stream, and the
one-stream build of the same calls (TFHE_LANES=0, a subprocess) bit for bit."""
import hashlib
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import toyfhe_jl_amd as tf
from oracle import ref_cpu
from tests import helpers as H

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def dev(a):
    return tf.DeviceBuffer.from_numpy(a)


def mixed_chain(N, small=3):
    return H.chain(60, 1, N) + H.chain(40, small, N) + [H.chain(60, 2, N)[1]]


@pytest.mark.parametrize("logn,batch", [(14, 70), (16, 16)])
def test_lanes_keep_the_stream_order_over_queued_calls_and_released_buffers(logn, batch):
    N = 1 << logn
    qs = mixed_chain(N)
    Lk, level = len(qs), len(qs) - 1
    ctx, ref = tf.Context(N, qs), ref_cpu.RefCtx(N, qs)
    other = tf.Context(N, qs)                                             # a second context: its own stream, the same allocator
    rng = np.random.default_rng(logn)
    evk = H.uniform_evk(rng, qs, Lk, N)
    devk = dev(evk)
    ct = H.rand_residues(rng, qs[:level], (batch, 2), N)
    g = pow(3, 5, 2 * N)
    want_ks = ref.keyswitch(level, True, evk, ct[:2])
    results = []
    for it in range(12):                                                  # nothing waits on the host inside the loop
        dct, dout, drot = dev(ct), tf.DeviceBuffer(batch * 2 * level * N), tf.DeviceBuffer(batch * 2 * level * N)
        a, b = tf.DeviceBuffer(batch * 2 * level * N), tf.DeviceBuffer(batch * 2 * level * N)
        ctx.keyswitch(Lk, level, True, devk.ptr, Lk, dct.ptr, 2, dout.ptr, batch)
        ctx.rotate(Lk, level, True, devk.ptr, Lk, g, dct.ptr, drot.ptr, batch)
        ctx.nntt(dct.ptr, a.ptr, batch * 2, level)                        # out of place: the one-pass kernels of N = 2^16
        ctx.inntt(a.ptr, b.ptr, batch * 2, level)
        ctx.inntt(a.ptr, a.ptr, batch * 2, level)                         # in place: the two-kernel path through the transform scratch
        if it % 3 == 0:                                                   # the other context works beside it and releases its buffers at once
            oa = tf.DeviceBuffer(batch * 2 * level * N)
            other.nntt(dct.ptr, oa.ptr, batch * 2, level)
            del oa
        results.append((dout, drot, a, b))
        del dct                                                           # released while the calls above are still queued
    first = None
    for dout, drot, a, b in results:
        got = (dout.to_numpy((batch, 2, level, N)), drot.to_numpy((batch, 2, level, N)))
        if first is None:
            first = got
            assert np.array_equal(got[0][:2], want_ks)                    # the first one is the oracle's
        else:
            assert np.array_equal(got[0], first[0]) and np.array_equal(got[1], first[1])
        assert np.array_equal(a.to_numpy(ct.shape), ct) and np.array_equal(b.to_numpy(ct.shape), ct)   # transform round trips


WORKER = r'''
import sys, json, hashlib
sys.path.insert(0, %(root)r)
import numpy as np
import toyfhe_jl_amd as tf
from tests import helpers as H
out = {}
for logn, batch in ((14, 70), (16, 16)):
    N = 1 << logn
    qs = H.chain(60, 1, N) + H.chain(40, 3, N) + [H.chain(60, 2, N)[1]]
    Lk, level = len(qs), len(qs) - 1
    ctx = tf.Context(N, qs)
    rng = np.random.default_rng(100 + logn)
    evk = tf.DeviceBuffer.from_numpy(H.uniform_evk(rng, qs, Lk, N))
    ct = tf.DeviceBuffer.from_numpy(H.rand_residues(rng, qs[:level], (batch, 2), N))
    o = tf.DeviceBuffer(batch * 2 * level * N)
    sha = lambda b: hashlib.sha256(b.to_numpy().tobytes()).hexdigest()
    ctx.keyswitch(Lk, level, True, evk.ptr, Lk, ct.ptr, 2, o.ptr, batch); out["ks%%d" %% logn] = sha(o)
    ctx.rotate(Lk, level, True, evk.ptr, Lk, pow(3, 7, 2 * N), ct.ptr, o.ptr, batch); out["rot%%d" %% logn] = sha(o)
    ctx.nntt(ct.ptr, o.ptr, batch * 2, level); out["nntt%%d" %% logn] = sha(o)
    ctx.inntt(ct.ptr, o.ptr, batch * 2, level); out["inntt%%d" %% logn] = sha(o)
    full = tf.DeviceBuffer.from_numpy(H.rand_residues(rng, qs, (batch,), N))
    of = tf.DeviceBuffer(batch * Lk * N)
    ctx.nntt(full.ptr, of.ptr, batch, Lk); out["nntt_all%%d" %% logn] = sha(of)
print(json.dumps(out))
'''


def test_two_lanes_and_one_stream_give_the_same_words():
    script = os.path.join(ROOT, "gpurun_out", "_lanes_worker.py")
    os.makedirs(os.path.dirname(script), exist_ok=True)
    open(script, "w").write(WORKER % {"root": ROOT})
    res = {}
    for lanes in ("0", "1"):
        env = dict(os.environ, TFHE_LANES=lanes)
        out = subprocess.run([sys.executable, script], capture_output=True, text=True, timeout=900, env=env)
        assert out.returncode == 0, out.stderr[-3000:]
        res[lanes] = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert res["0"] == res["1"] and len(res["0"]) == 10
