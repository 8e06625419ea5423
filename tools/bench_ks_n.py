#!/usr/bin/env python3
"""Key switches per second at N = 2^logN on a 6-limb + special-prime ring of 50-bit primes (the fused kernels at
logN = 13 / 14).  usage: bench_ks_n.py <logN> <batch> [bits]   (TFHE_HIP_LIB=<other build> for A/B runs; bits = size of the primes, default 50)"""
import sys, os, time
sys.path.insert(0, os.getcwd())
import toyfhe_jl_amd as tf
N=1<<int(sys.argv[1]); batch=int(sys.argv[2])
def chain(start,n):
    out,p=[],tf.nextprime(start,1,2*N)
    for _ in range(n):
        out.append(p); p=tf.nextprime(p+2*N,1,2*N)
    return out
bits=int(sys.argv[3]) if len(sys.argv)>3 else 50
qs=chain(2**bits+1,7); Lk=7; level=6
ctx=tf.Context(N,qs)
evk=tf.DeviceBuffer(Lk*2*Lk*N); ctx.sample_uniform(Lk,1,0,0,evk.ptr,Lk*2)
ct=tf.DeviceBuffer(batch*2*level*N); ctx.sample_uniform(level,2,0,0,ct.ptr,batch*2)
out=tf.DeviceBuffer(batch*2*level*N)
f=lambda: ctx.keyswitch(Lk,level,True,evk.ptr,Lk,ct.ptr,2,out.ptr,batch)
t0=time.perf_counter()
while time.perf_counter()-t0<0.3: f(); ctx.sync()
best=1e9
for _ in range(3):
    ctx.sync(); t=time.perf_counter()
    for _ in range(8): f()
    ctx.sync(); best=min(best,(time.perf_counter()-t)/8)
print("N=2^%s batch %d %d-bit fused13=%s: %.0f keyswitch/s = %.2f G coefficient-limbs/s" % (sys.argv[1], batch, bits, os.environ.get("TFHE_FUSED13","1"), batch/best, batch/best*N*level/1e9))
