#!/usr/bin/env python3
"""Same-box A/B of NTT kernel variants inside the full BFV mul+relin pipeline (box-to-box spread is +-3 %).
usage: ab_pipeline.py v1,v2[,...] [batch]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import toyfhe_jl_amd as tf
from tests import helpers as H
N, L, LBIG, T = 1 << 14, 8, 17, 65537
primes = H.chain(50, LBIG, N)
dev = torch.device("cuda", 0)
variants = [int(v) for v in sys.argv[1].split(",")]
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
ctx = tf.Context(N, primes)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
plan = tf.BfvPlan(ctx, ctx, T, idx_s=list(range(L)))
def rnd(shape):
    out = torch.empty(tuple(shape) + (L, N), dtype=torch.int64, device=dev)
    for l, q in enumerate(primes[:L]):
        out[..., l, :] = torch.randint(0, q, tuple(shape) + (N,), dtype=torch.int64, device=dev)
    return out
c1, c2, evk, out = rnd((B, 2)), rnd((B, 2)), rnd((L, 2)), torch.empty((B, 2, L, N), dtype=torch.int64, device=dev)
def step(): plan.mul_relin(evk.data_ptr(), L, c1.data_ptr(), c2.data_ptr(), out.data_ptr(), B)
for rnd_ in range(3):
    for v in variants:
        ctx.set_ntt_variant(v)
        step(); torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(3): step()
        torch.cuda.synchronize()
        print(f"round {rnd_} variant {v}: {B * 3 / (time.perf_counter() - t):8.0f} ct-mul/s")
