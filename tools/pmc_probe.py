#!/usr/bin/env python3
"""Workload for the HBM-traffic PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, one counter per pass):
a calibration copy with the NTT's own access width (8 bytes per lane: tfhe_select_limbs over a known byte count)
followed by forward and inverse 2^14-point NTT launches over `rows` limb-polynomials."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import toyfhe_jl_amd as tf
from tests import helpers as H

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
N, L = 1 << 14, 8
qs = H.chain(50, L, N)
ctx = tf.Context(N, qs)
a = tf.DeviceBuffer(rows * N)
b = tf.DeviceBuffer(rows * N)
tf.native.check(tf.native.lib().tfhe_memset(ctx.h, a.ptr, 1, rows * N * 8))
ctx.sync()
count = rows // L
ctx.select_limbs(a.ptr, b.ptr, count, L, list(range(L)))      # calibration: reads rows*128 KiB, writes the same
ctx.nntt(a.ptr, b.ptr, count, L)
ctx.inntt(b.ptr, a.ptr, count, L)
ctx.sync()
print("rows", rows, "bytes_per_direction", rows * N * 8)
