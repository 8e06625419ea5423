#!/usr/bin/env python3
"""Probe: BFV mul+relin throughput with ONE pipeline vs TWO pipelines on two HIP streams (two contexts / plans sharing
the GPU), each taking half of the batch -- does a second chunk's streaming kernels run underneath the transforms?"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import toyfhe_jl_amd as tf
from tests import helpers as H

N, L, LBIG, T = 1 << 14, 8, 17, 65537
primes = H.chain(50, LBIG, N)
dev = torch.device("cuda", 0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
chunk = int(sys.argv[2]) if len(sys.argv) > 2 else 256

def mk(stream):
    ctx = tf.Context(N, primes)
    ctx.set_stream(stream.cuda_stream)
    plan = tf.BfvPlan(ctx, ctx, T, idx_s=list(range(L)))
    plan.set_chunk(chunk)
    return ctx, plan

def rnd(shape):
    out = torch.empty(tuple(shape) + (L, N), dtype=torch.int64, device=dev)
    for l, q in enumerate(primes[:L]):
        out[..., l, :] = torch.randint(0, q, tuple(shape) + (N,), dtype=torch.int64, device=dev)
    return out

c1, c2, evk, out = rnd((B, 2)), rnd((B, 2)), rnd((L, 2)), torch.empty((B, 2, L, N), dtype=torch.int64, device=dev)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
(ctxa, pa), (ctxb, pb) = mk(s1), mk(s2)
sz = 2 * L * N * 8

def one():
    pa.mul_relin(evk.data_ptr(), L, c1.data_ptr(), c2.data_ptr(), out.data_ptr(), B)

def two():
    h = B // 2
    pa.mul_relin(evk.data_ptr(), L, c1.data_ptr(), c2.data_ptr(), out.data_ptr(), h)
    pb.mul_relin(evk.data_ptr(), L, c1.data_ptr() + h * sz, c2.data_ptr() + h * sz, out.data_ptr() + h * sz, B - h)

for name, f in (("one lane", one), ("two lanes", two), ("one lane", one), ("two lanes", two)):
    f(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(3): f()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / 3
    print(f"{name}: {B / dt:8.0f} ct-mul/s")
