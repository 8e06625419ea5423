#!/usr/bin/env python3
"""More seeds of tests/test_gpu_parity.py::test_random_shapes_against_the_oracle (random degree / limb count / mixed modulus
sizes / batch / level / special prime / component count; transforms, Galois, rescale, key switch, rotation against the C
oracle, bit for bit).  Runs for about seven minutes on the GPU box; r02e: seeds 24..399, no failure.
usage: python tools/fuzz_parity.py"""
import sys, time, traceback
sys.path.insert(0, "/root/repo")
from tests import test_gpu_parity as T
bad = []
t0 = time.time()
for seed in range(24, 400):
    if time.time() - t0 > 420: print("stopped at", seed); break
    try:
        T.test_random_shapes_against_the_oracle(seed)
    except Exception as e:
        bad.append(seed); print("FAIL seed", seed, repr(e)[:300])
print("done; failures:", bad)
