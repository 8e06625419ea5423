#!/usr/bin/env python3
"""cProfile of the encrypted MNIST example (host-side cost per scheme-layer call).
usage: prof_mnist.py <logN> [batches] [hoisted 0|1] [repeat]   e.g.  prof_mnist.py 16 4 1 2"""
import cProfile, os, pstats, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "examples"))
import encrypted_mnist as m
logn = int(sys.argv[1])
batches = int(sys.argv[2]) if len(sys.argv) > 2 else 1
hoisted = bool(int(sys.argv[3])) if len(sys.argv) > 3 else False
repeat = int(sys.argv[4]) if len(sys.argv) > 4 else 1
m.run(logn, 0, verbose=False, batches=batches, hoisted=hoisted, repeat=repeat)   # warm
pr = cProfile.Profile(); pr.enable()
m.run(logn, 0, batches=batches, hoisted=hoisted, repeat=repeat)
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(24)
