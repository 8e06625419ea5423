#!/usr/bin/env python3
"""Kernel dispatches of ONE encrypted-MNIST evaluation pass (setup -- keys, encryption, weight encoding -- excluded): run
under `rocprofv3 --kernel-trace`; prints the wall-clock window of the last pass so that tools/csv_window_stats.py can cut
the trace.  usage: prof_mnist_eval.py <logN> <batches> <fused 0|1>"""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "examples"))
import encrypted_mnist as m
logn, batches, fused = int(sys.argv[1]), int(sys.argv[2]), bool(int(sys.argv[3]))
m.run(logn, 0, verbose=True, batches=batches, hoisted=True, repeat=3, fused=fused)
