#!/usr/bin/env python3
"""Probe (r06): where the host time of the reference-shaped MNIST pass goes when it runs behind the restructured case in one process:
time inside tfhe_malloc per size, inside the rotations, inside dot_plain."""
import collections, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bench_configs as bc
import toyfhe_jl_amd as tf
from toyfhe_jl_amd import native, she
if "alone" not in sys.argv:
    bc.mnist_case("restructured", 16, 16)
T = collections.defaultdict(float); C = collections.Counter()
orig_init = native.DeviceBuffer.__init__
def timed_init(self, n_words):
    t = time.perf_counter(); orig_init(self, n_words); dt = time.perf_counter() - t
    T["malloc %d MiB" % (n_words * 8 >> 20)] += dt; C["malloc %d MiB" % (n_words * 8 >> 20)] += 1
native.DeviceBuffer.__init__ = timed_init
def wrap(mod, name):
    f = getattr(mod, name)
    def g(*a, **k):
        t = time.perf_counter(); r = f(*a, **k); T[name] += time.perf_counter() - t; C[name] += 1; return r
    setattr(mod, name, g)
wrap(she, "_dot_batched"); wrap(she, "keyswitch"); wrap(she, "_unpack"); wrap(she, "_pack")
bc.mnist_case("refshape", 16, 16, True)
r = bc.RECORDS[-1]
print("refshape", round(r["ms_per_pass"], 1), "ms; host enqueue", round(r["host_enqueue_ms"], 1), "(3 passes + set-up in the totals below)")
for k, v in sorted(T.items(), key=lambda kv: -kv[1])[:14]:
    print(f"{k:24s} {C[k]:6d} calls {v * 1e3:9.1f} ms")
