#!/usr/bin/env python3
"""More seeds of tests/test_gpu_parity.py::test_random_shapes_against_the_oracle (random degree / limb count / mixed modulus
sizes / batch / level / special prime / component count; transforms, Galois, rescale, key switch, rotation against the C
oracle, bit for bit).  Then the same for
test_random_bfv_multiplications_against_the_oracle.  About eleven minutes on the GPU box; r02e: no failure.
usage: python tools/fuzz_parity.py [first-seed]   (default 24: the seeds after the ones the test suite runs)"""
import sys, time, traceback
sys.path.insert(0, "/root/repo")
from tests import test_gpu_parity as T
bad = []
t0 = time.time()
S0 = int(sys.argv[1]) if len(sys.argv) > 1 else 24
for seed in range(S0, S0 + 1000):
    if time.time() - t0 > 420: print("stopped at", seed); break
    try:
        T.test_random_shapes_against_the_oracle(seed)
    except Exception as e:
        bad.append(seed); print("FAIL seed", seed, repr(e)[:300])
print("done; failures:", bad)
bad, t0 = [], time.time()
for seed in range(S0, S0 + 1000):
    if time.time() - t0 > 240: print("bfv: stopped at", seed); break
    try:
        T.test_random_bfv_multiplications_against_the_oracle(seed)
    except Exception as e:
        bad.append(seed); print("FAIL bfv seed", seed, repr(e)[:300])
print("bfv done; failures:", bad)
