#!/usr/bin/env python3
"""Encrypted-MNIST pass (restructured circuit, N = 2^16, infer.jl ring) over 1 .. 20 ciphertext sets on ONE GPU: what a rank of an 8-GPU run
of BASELINE configs[4] gets out of its shard (20 sets -> 3/3/3/3/2/2/2/2).  usage: mnist_sets_sweep.py [sets ...]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bench_configs as bc
sets = [int(x) for x in sys.argv[1:]] or [1, 2, 3, 4, 8, 16, 20]
for s in sets:
    del bc.RECORDS[:]
    bc.mnist_case("cfg#5 MNIST restructured, %d sets" % s, 16, s)
    r = bc.RECORDS[-1]
    print(json.dumps({"sets": s, "images": r["images"], "ms_per_pass": round(r["ms_per_pass"], 2), "images_per_s": round(r["images_per_s"]),
                      "ms_per_set": round(r["ms_per_pass"] / s, 2)}), flush=True)
