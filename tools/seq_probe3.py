#!/usr/bin/env python3
"""Probe (r06): cProfile of the reference-shaped MNIST case behind the restructured one (or alone)."""
import cProfile, os, pstats, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bench_configs as bc
if "alone" not in sys.argv:
    bc.mnist_case("restructured", 16, 16)
pr = cProfile.Profile()
pr.enable()
bc.mnist_case("refshape", 16, 16, True)
pr.disable()
r = bc.RECORDS[-1]
print("refshape", round(r["ms_per_pass"], 1), "ms; host enqueue", round(r["host_enqueue_ms"], 1))
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
