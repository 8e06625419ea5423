#!/usr/bin/env python3
"""Phase times of workgroup 0 in the fused key switch (k_ks_fused), from a -DTFHE_KS_TRACE build of the engine:
  hipcc ... -DTFHE_KS_TRACE toyfhe_hip.hip -o tools/bin/libtoyfhe_trace.so ;  TFHE_HIP_LIB=tools/bin/libtoyfhe_trace.so python tools/ks_trace.py
tags: 1 digit start | 2 after load + lift + pass 1 + exchange | 3 after pass 2 + exchange | 4 after pass 3 | 5 after the last key product |
6 after inverse 1 | 7 after inverse 2.  (100 MHz clock: 10 ns resolution.)"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import toyfhe_jl_amd as tf
N = 1 << 14; batch = 512
def chain(start, n):
    out, p = [], tf.nextprime(start, 1, 2 * N)
    for _ in range(n):
        out.append(p); p = tf.nextprime(p + 2 * N, 1, 2 * N)
    return out
qs = chain(2**50 + 1, 7); Lk = 7; level = 6
ctx = tf.Context(N, qs)
evk = tf.DeviceBuffer(Lk * 2 * Lk * N); ctx.sample_uniform(Lk, 1, 0, 0, evk.ptr, Lk * 2)
ct = tf.DeviceBuffer(batch * 2 * level * N); ctx.sample_uniform(level, 2, 0, 0, ct.ptr, batch * 2)
out = tf.DeviceBuffer(batch * 2 * level * N)
L = tf.native.lib()
buf = (C.c_ulonglong * 4096)(); n = C.c_uint(0)
for _ in range(20): ctx.keyswitch(Lk, level, True, evk.ptr, Lk, ct.ptr, 2, out.ptr, batch)
L.tfhe_debug_kstrace(buf, C.byref(n), 1)
ctx.keyswitch(Lk, level, True, evk.ptr, Lk, ct.ptr, 2, out.ptr, batch)
L.tfhe_debug_kstrace(buf, C.byref(n), 1)
ev = [(int(x) >> 56, (int(x) & ((1 << 56) - 1)) * 0.01) for x in buf[:n.value]]
names = {2: "load+lift+pass1+xchg", 3: "pass2+xchg", 4: "pass3", 1: "key product (or item setup)", 5: "key product (last digit)", 6: "inverse 1", 7: "inverse 2"}
acc, cnt = {}, {}
for (t0, a), (t1, b) in zip(ev, ev[1:]):
    k = (t0, t1)
    acc[k] = acc.get(k, 0.0) + (b - a); cnt[k] = cnt.get(k, 0) + 1
print("events", n.value, "span %.1f us" % (ev[-1][1] - ev[0][1]))
for k in sorted(acc):
    print("  %d -> %d  %-28s n=%4d  mean %.2f us" % (k[0], k[1], names.get(k[1], ""), cnt[k], acc[k] / cnt[k]))
