import cProfile, pstats, sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/examples")
import encrypted_mnist as m
m.run(16, 0, verbose=False, repeat=2, batches=4, hoisted=True)
pr = cProfile.Profile(); pr.enable()
m.run(16, 0, verbose=True, repeat=2, batches=4, hoisted=True)
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
