#!/usr/bin/env python3
"""Shader clock / power while the headline pipeline runs (rocm-smi sampled from a thread).  usage: clock_probe.py [seconds]"""
import sys, os, time, subprocess, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import toyfhe_jl_amd as tf
from tests import helpers as H
N, L, LBIG, T, B = 1 << 14, 8, 17, 65537, 1024
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 15
primes = H.chain(50, LBIG, N)
dev = torch.device("cuda", 0)
ctx = tf.Context(N, primes)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
plan = tf.BfvPlan(ctx, ctx, T, idx_s=list(range(L)))
def rnd(shape):
    out = torch.empty(tuple(shape) + (L, N), dtype=torch.int64, device=dev)
    for l, q in enumerate(primes[:L]):
        out[..., l, :] = torch.randint(0, q, tuple(shape) + (N,), dtype=torch.int64, device=dev)
    return out
c1, c2, evk, out = rnd((B, 2)), rnd((B, 2)), rnd((L, 2)), torch.empty((B, 2, L, N), dtype=torch.int64, device=dev)
stop = False
def sampler():
    while not stop:
        o = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True).stdout
        print(" | ".join(l.split(":", 1)[1].strip() for l in o.splitlines() if "sclk" in l or "Power (W)" in l or "fclk" in l), flush=True)
        time.sleep(1.0)
th = threading.Thread(target=sampler); th.start()
t0 = time.perf_counter(); n = 0
while time.perf_counter() - t0 < secs:
    for _ in range(10): plan.mul_relin(evk.data_ptr(), L, c1.data_ptr(), c2.data_ptr(), out.data_ptr(), B)
    torch.cuda.synchronize(); n += 10
dt = time.perf_counter() - t0
stop = True; th.join()
print("%.0f ct-mul/s over %.1f s" % (n * B / dt, dt))
