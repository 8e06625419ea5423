#!/usr/bin/env python3
"""bench.py -- BASELINE.json metric: BFV ciphertext-mul (+ relinearise) per second at N = 2^14, L = 8 RNS
limbs, batch = 1024 per GPU (configs[1]); NTT GB/s against the HBM roofline from the same timed region.

A "step" is one pass of the hot path (tfhe_bfv_mul_relin: exact expand 8->17 limbs; 68 forward limb-NTTs, tensor and
51 inverse limb-NTTs in one fused kernel; exact scale-and-round back to 8 limbs; RNS-digit key switch with its 64 + 16
transforms in a second fused kernel) over one batch
of synthetic ciphertexts already resident in HBM.  One process per GPU; ranks shard the batch (weak
scaling: the per-GPU batch is fixed), no data-path collective.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--no-cpu]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

LOGN, L, LBIG, T_PLAIN = 14, 8, 17, 65537
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E datasheet peak (MI355X_MICROARCH.md)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=1024, help="ciphertext pairs per GPU per step")
    ap.add_argument("--chunk", type=int, default=0, help="ciphertexts per internal pipeline chunk (0 = default)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--cpu-sample", type=int, default=0, help="ciphertext pairs in the CPU sample (0 = auto)")
    args = ap.parse_args()

    import numpy as np
    import torch

    import toyfhe_jl_amd as tf

    from toyfhe_jl_amd import dist as tdist
    world, rank, local_rank = tdist.env_world()
    torch.cuda.set_device(local_rank)
    tdist.init(backend="nccl", device_id=torch.device("cuda", local_rank))
    tf.native.check(tf.native.lib().tfhe_set_device(local_rank))
    dev = torch.device("cuda", local_rank)

    N = 1 << LOGN
    from tests import helpers as H  # prime chain helper only (no oracle compute in the timed path)
    primes = H.chain(50, LBIG, N)   # first NTT-friendly primes above 2^50 for N = 2^14
    qs = primes[:L]
    ctx = tf.Context(N, primes)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    plan = tf.BfvPlan(ctx, ctx, T_PLAIN, idx_s=list(range(L)))
    if args.chunk:
        plan.set_chunk(args.chunk)

    B = args.batch
    gen = torch.Generator(device=dev)
    gen.manual_seed(0xF4E5EED + rank)

    def uniform(shape_prefix):
        out = torch.empty(tuple(shape_prefix) + (L, N), dtype=torch.int64, device=dev)
        for l, q in enumerate(qs):
            out[..., l, :] = torch.randint(0, q, tuple(shape_prefix) + (N,), dtype=torch.int64, device=dev, generator=gen)
        return out

    c1, c2 = uniform((B, 2)), uniform((B, 2))
    evk = uniform((L, 2))
    out = torch.empty((B, 2, L, N), dtype=torch.int64, device=dev)

    def step():
        plan.mul_relin(evk.data_ptr(), L, c1.data_ptr(), c2.data_ptr(), out.data_ptr(), B)

    barrier = tdist.barrier

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    ctx.prof_enable(True)
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    barrier()
    t1 = time.perf_counter()
    launches, limb_polys, ntt_ms = ctx.prof_read()
    ctx.prof_enable(False)
    elapsed = t1 - t0
    elapsed = tdist.max_over_ranks(elapsed, device=dev)

    total_units = B * world * args.steps
    value = total_units / elapsed
    ntt_bytes = limb_polys * 2 * N * 8                   # one read + one write per limb transform
    achieved = ntt_bytes / (ntt_ms * 1e-3) / 1e9 if ntt_ms > 0 else 0.0
    # HBM bytes per launch from the PMC passes (FETCH_SIZE / WRITE_SIZE, separate rocprofv3 runs, gfx950 corrections;
    # tools/pmc_bench.py over this very script at batch 256).  bench.py cannot run the counters itself, so it scales the
    # committed per-ciphertext figure of the two transform-carrying kernels to this run's launches.
    traffic = None
    pmc_path = os.path.join(ROOT, "profiles", "r01r_pmc_bench_kernels.json")
    if os.path.exists(pmc_path) and launches:
        pmc = json.load(open(pmc_path))
        per_ct = sum(k["hbm_bytes_per_launch"] for name, k in pmc["kernels"].items() if "fused" in name) / pmc["batch"]
        traffic = per_ct * B * args.steps / launches
    result = {
        "metric": "ciphertext-mul/s (BFV ct*ct + relinearize, N=2^14, L=8 RNS)",
        "value": value,
        "unit": "ciphertext-mul/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u64",
        "data": "synthetic",
        "config": {"workload": "BFV N=2^14, L=8 RNS limbs (50-bit primes), extension basis 17 limbs, t=65537, "
                               "ciphertext-mul + relinearize (RNS-digit keyswitch), bit-exact", "batch_per_gpu": B,
                   "global_batch": B * world, "sharding": f"batch x{world}, no data-path collective"},
        "roofline": {"bound": "hbm", "kernel": "k_bfv_core_fused + k_ks_fused: the 2^14-point negacyclic NTTs (fp64 butterflies) fused with the "
                               "tensor product / key inner product -- 7 and 10 limb transforms per workgroup item; units = limb "
                               "transforms x 2*N*8 algorithmic bytes (SURVEY 8d), durations by HIP events around these launches",
                     "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                     "traffic": traffic, "traffic_unit": "HBM-side bytes per launch (PMC FETCH_SIZE x2 + WRITE_SIZE of the two kernels, "
                                                         "profiles/r01r_pmc_bench_kernels.json, scaled to this run's launches); below the "
                                                         "algorithmic bytes because the transforms are fused",
                     "algorithmic_bytes_per_launch": ntt_bytes / launches if launches else None,
                     "launches": launches, "limb_ntts": limb_polys,
                     "avg_launch_ms": ntt_ms / launches if launches else None,
                     "ntt_share_of_step": ntt_ms * 1e-3 / elapsed if elapsed > 0 else None},
    }

    if rank == 0 and world == 1 and not args.no_cpu:
        from oracle import ref_cpu
        cores = ref_cpu.lib().ref_num_threads()
        ns = args.cpu_sample or max(8 * cores, 8)
        rng = np.random.default_rng(1)
        s1, s2 = H.rand_residues(rng, qs, (ns, 2), N), H.rand_residues(rng, qs, (ns, 2), N)
        sk = H.uniform_evk(rng, qs, L, N)
        rs, rb = ref_cpu.RefCtx(N, qs), ref_cpu.RefCtx(N, primes)
        tc = time.perf_counter()
        prod = ref_cpu.bfv_mul(rs, rb, T_PLAIN, s1, s2)
        rs.keyswitch(L, False, sk, prod)
        tc = time.perf_counter() - tc
        result["cpu_baseline"] = {"value": ns / tc, "unit": "ciphertext-mul/s", "cores": cores, "kind": "port",
                                  "sample": f"{ns} ciphertext pairs of the same workload, oracle/ref_cpu.c "
                                            f"(exact BigInt-style conversions, radix-2 NTT), OpenMP over the batch, {tc:.1f} s"}
    if rank == 0:
        print(json.dumps(result))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
