import os, sys
R = "/root/repo"
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "examples"))
import encrypted_mnist as m
st = {}
m.run(16, 0, verbose=False, batches=16, hoisted=False, repeat=3, fused=False, stats=st)
print(st)
