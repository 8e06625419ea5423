#!/usr/bin/env python3
"""Probe (r06): the reference-shaped MNIST case after the restructured one in ONE process -- pass time and allocator statistics, with
the environment's switches (TFHE_BATCH_NTT, TFHE_ALLOC_SOFT_GIB) and optionally a trim in between.  usage: seq_probe.py [trim]"""
import gc, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bench_configs as bc
import toyfhe_jl_amd as tf
st = lambda: {k: (round(v / 2**30, 2) if "bytes" in k else v) for k, v in tf.native.alloc_stats().items()}
bc.mnist_case("restructured", 16, 16)
print("after case 9", st(), flush=True)
if "trim" in sys.argv:
    gc.collect()
    tf.native.check(tf.native.lib().tfhe_alloc_trim())
    print("after trim", st(), flush=True)
bc.mnist_case("refshape", 16, 16, True)
print("after case 10", st(), flush=True)
for r in bc.RECORDS:
    print(r["config"], round(r["ms_per_pass"], 1), "ms; host enqueue", round(r["host_enqueue_ms"], 1))
