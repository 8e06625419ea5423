import cProfile, pstats, sys, os, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "examples"))
import encrypted_mnist as m
logn = int(sys.argv[1])
m.run(logn, 0)   # warm
pr = cProfile.Profile(); pr.enable()
m.run(logn, 0)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
