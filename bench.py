#!/usr/bin/env python3
"""bench.py -- BASELINE.json metric: BFV ciphertext-mul (+ relinearise) per second at N = 2^14, L = 8 RNS
limbs, batch = 1024 per GPU (configs[1]); NTT GB/s against the HBM roofline (BASELINE.md #2').

A "step" is one pass of the hot path (tfhe_bfv_mul_relin: exact expand 8->17 limbs; 68 forward limb-NTTs, tensor and
51 inverse limb-NTTs in one fused kernel; exact scale-and-round back to 8 limbs; RNS-digit key switch with its 64 + 16
transforms in a second fused kernel) over one batch of synthetic ciphertexts already resident in HBM.

One process per GPU; ranks shard the batch (weak scaling: the per-GPU batch is fixed), no data-path collective.
`--gpus N` with N > 1 and no torchrun environment re-executes this script under `python -m torch.distributed.run`
with N ranks (RCCL); under torchrun the environment's WORLD_SIZE is what runs and what `n_gpus` reports.  After the
timed region the optional final gather (north_star: "RCCL over xGMI only for the final gather") is timed on its own, so
the JSON carries the rate without (`value`) and with it (`gather.value_with_gather`).

After the headline line's legs, rank 0 of a single-GPU run also measures the other BASELINE.json configurations
(`other_configs`: cfg#3 / #4 / #5 key switch, rotation and rescale rates, N = 2^16 and 60-bit NTT rates, each case checked
against the oracle on one ciphertext before it is timed -- tools/bench_configs.py); `--no-configs` skips that leg.

A multi-rank run (--gpus N > 1) also carries `configs_multi`: the configurations BASELINE.json defines across the GPUs of a node --
cfg#4 (4096 key switches sharded over the ranks, strong scaling) and cfg#5 (the ciphertext sets of the MNIST test set sharded over
the ranks) -- each with per-rank rates, max/min imbalance, nranks_seen and the final gather timed on its own.  `--config cfg3|cfg4|cfg5`
makes one of them THE line of the run instead of the BFV metric.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--config bfv|cfg3|cfg4|cfg5] [--no-cpu] [--no-ntt] [--no-gather] [--no-configs]
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

LOGN, L, LBIG, T_PLAIN = 14, 8, 17, 65537
HBM_PEAK_GBS = 8000.0            # MI355X HBM3E datasheet peak (MI355X_MICROARCH.md)
N_SIMD, PEAK_CLOCK_HZ = 256 * 4, 2.4e9   # 256 CUs x 4 SIMD16; peak engine clock (78.6 TFLOP/s fp64 vector = 1024 x 16 x 2 x 2.4e9)
VALU_PEAK_GIPS = N_SIMD * PEAK_CLOCK_HZ / 4 / 1e9   # one wave64 VALU instruction per 4 clocks per SIMD (fp64 is full rate)
PMC_FILE = os.path.join(ROOT, "profiles", "pmc_bench_kernels.json")   # regenerated per round by tools/pmc_round.sh


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=1024, help="ciphertext pairs per GPU per step")
    ap.add_argument("--chunk", type=int, default=0, help="ciphertexts per internal pipeline chunk (0 = default)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-ntt", action="store_true", help="skip the stand-alone NTT record")
    ap.add_argument("--no-gather", action="store_true", help="skip the timed final gather (multi-rank runs)")
    ap.add_argument("--no-configs", action="store_true", help="skip the other BASELINE.json configurations (other_configs leg)")
    ap.add_argument("--configs-scale", type=int, default=1, help="divide the batches of the other_configs cases by this")
    ap.add_argument("--cpu-sample", type=int, default=0, help="ciphertext pairs in the all-core CPU sample (0 = auto)")
    ap.add_argument("--cabi-gather", action="store_true",
                    help="time the final gather through the C ABI (tfhe_gather, the library's own RCCL communicator) instead of torch.distributed")
    ap.add_argument("--config", default="bfv", choices=["bfv", "cfg3", "cfg4", "cfg5"],
                    help="what the line measures: bfv = the BASELINE metric (configs[1]); cfg3 / cfg4 / cfg5 = the other BASELINE.json "
                         "configurations as multi-rank jobs (tools/bench_configs.py MULTI: cfg#4 = 4096 key switches sharded over the ranks, "
                         "cfg#5 = the MNIST test set's ciphertext sets sharded over the ranks)")
    ap.add_argument("--total", type=int, default=0, help="--config cfg*: global units instead of the configuration's own (tests)")
    ap.add_argument("--backend", default=os.environ.get("TFHE_BENCH_BACKEND", "nccl"),
                    help="torch.distributed backend; 'gloo' + ranks sharing a GPU is a functional check only")
    return ap.parse_args()


def respawn(args):
    """`python bench.py --gpus N` outside torchrun: become the launcher of N ranks, one per GPU."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    return subprocess.call(cmd, env=env)


def prime_chain(tf, bits, n, N):
    """first n primes = 1 (mod 2N) above 2^bits (crt.jl:282-295 rule)"""
    out, p = [], tf.nextprime(2**bits + 1, 1, 2 * N)
    for _ in range(n):
        out.append(p)
        p = tf.nextprime(p + 2 * N, 1, 2 * N)
    return out


def load_pmc():
    if not os.path.exists(PMC_FILE):
        return None
    try:
        return json.load(open(PMC_FILE))
    except Exception:
        return None


def source_id():
    """sha256 over the engine's sources (csrc/* and the header): what the library running here was built from.  The PMC file
    carries the same id of the build it profiled (tools/pmc_bench.py); a mismatch means its counters describe other kernels."""
    h = hashlib.sha256()
    src = os.path.join(ROOT, "toyfhe.jl_amd", "csrc")
    for f in sorted(os.listdir(src)) + [os.path.join("..", "..", "include", "toyfhe_hip.h")]:
        h.update(f.encode())
        h.update(open(os.path.join(src, f), "rb").read())
    return h.hexdigest()[:16]


# ---- the stdout line: compact (the driver's capture is ~9.6 KB; BENCH_r04 lost a 22.8 KB line), the full record goes to a file -----
LINE_LIMIT = 4096
FULL_RECORD = os.path.join(ROOT, "profiles", "bench_full.json")


def _r(x, sig=5):
    """floats to `sig` significant digits (the line is a summary; the full record keeps every digit)"""
    if isinstance(x, bool) or x is None:
        return x
    if isinstance(x, float):
        return float(f"{x:.{sig}g}")
    if isinstance(x, dict):
        return {k: _r(v, sig) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_r(v, sig) for v in x]
    return x


def _compact_roofline(roof):
    keep = ("bound", "achieved", "peak", "unit", "frac", "essential_frac", "hbm_frac", "ctmul_hbm_frac", "traffic",
            "traffic_per_ctmul_over_algorithmic", "avg_launch_ms", "launches", "pmc_source_id", "pmc_stale")
    out = {k: roof.get(k) for k in keep if k in roof}
    out["kernel"] = str(roof.get("kernel", "")).split(":")[0]
    if "ctmul_algorithmic_bytes" in roof:
        out["algorithmic_bytes"] = roof["ctmul_algorithmic_bytes"]
    if "traffic_per_ctmul_over_algorithmic" in out:
        out["traffic_over_algorithmic"] = out.pop("traffic_per_ctmul_over_algorithmic")
    te = roof.get("transform_equiv") or {}
    if te:
        out["transform_equiv_GBs"] = te.get("achieved")
    return out


def _compact_config(c):
    """one short dict per other_configs / configs_multi record: name, rates, dominant kernel, its fraction and traffic ratio"""
    name = str(c.get("config", "?"))
    out = {"name": name.split(" (")[0][:44]}
    if "circuit" in c:
        out["name"] = ("MNIST 2^16 " if c.get("N") == 65536 else f"MNIST N={c.get('N')} ") + \
            ("restructured (63 keys, hoisted)" if str(c["circuit"]).startswith("restructured") else "reference-shaped (1 key, chained)")
    for src, dst in (("keyswitch_per_s", "ks_s"), ("rotate_per_s", "rot_s"), ("rescale_ct_per_s", "resc_s"), ("fwd_GBs", "fwd_GBs"),
                     ("inv_GBs", "inv_GBs"), ("images_per_s", "img_s"), ("ms_per_pass", "ms_pass"), ("value", "value"), ("unit", "unit"),
                     ("ms_per_step", "ms_step"), ("nranks_seen", "nranks_seen"), ("imbalance_max_over_min", "imbalance"), ("global_units", "units"), ("scaling", "scaling"),
                     ("batch", "batch"), ("oracle_checked", "ok"), ("error", "error")):
        if src in c and c[src] is not None:
            out[dst] = c[src] if not isinstance(c[src], str) else c[src][:40]
    roof = c.get("roofline") or {}
    for op in ("keyswitch", "nntt", "inntt", "pass"):                 # the dominant kernel of the case's main operation
        r = roof.get(op)
        if r:
            out.setdefault("kern", {})[op] = [str(r.get("kernel", "")).replace(" ", "")[:36], r.get("bound"), _r(r.get("frac"), 3),
                                              _r(r.get("traffic_over_algorithmic"), 3)]          # kernel, bound, frac, traffic / algorithmic
            ts = r.get("traffic_split")
            if ts:                                                    # ... of which key rows (cache-served, counted) / streaming rows (calibrated)
                out["kern"][op] += [_r(ts.get("key_over_algorithmic"), 3), _r(ts.get("rows_over_algorithmic"), 3)]
    if (c.get("roofline_source") or {}).get("pmc_stale"):
        out["pmc_stale"] = True
    return out


def compact_line(result, limit=LINE_LIMIT):
    """The ONE stdout line of a run, built from the full record: every key of the bench contract, `roofline` and `cpu_baseline`,
    a short `ntt` record and one short dict per other configuration.  Guaranteed to serialise below `limit` bytes: optional parts
    are dropped in a fixed order (per-config kernel notes, then the configs' secondary rates) until it fits."""
    line = {k: result[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                                   "vs_baseline", "dtype", "data") if k in result}
    for k in ("ms_per_step_median", "nranks_seen", "imbalance_max_over_min", "global_units", "algorithmic_GBs", "frac_of_hbm_peak_per_gpu", "oracle_checked"):
        if k in result:
            line[k] = result[k] if not isinstance(result[k], str) else result[k][:60]
    cfg = dict(result.get("config") or {})
    if isinstance(cfg.get("workload"), str):
        cfg["workload"] = cfg["workload"][:160]
    line["config"] = cfg
    if "roofline" in result:
        line["roofline"] = _compact_roofline(result["roofline"]) if "transform_equiv" in result["roofline"] or "essential_note" in result["roofline"] \
            else {k: (v if not isinstance(v, str) else v[:80]) for k, v in result["roofline"].items() if not isinstance(v, (dict, list))}
    cb = result.get("cpu_baseline")
    if cb:
        line["cpu_baseline"] = {"value": cb.get("value"), "unit": cb.get("unit"), "cores": cb.get("cores"), "kind": cb.get("kind"),
                                "sample": str(cb.get("sample", ""))[:120]}
        if cb.get("single_thread"):
            line["cpu_baseline"]["single_thread"] = {"value": cb["single_thread"].get("value"), "cores": 1}
    nt = result.get("ntt")
    if nt:
        line["ntt"] = {"fwd_GBs": nt.get("fwd_GBs"), "inv_GBs": nt.get("inv_GBs"), "fwd_frac": nt.get("fwd_frac_of_hbm_peak"),
                       "inv_frac": nt.get("inv_frac_of_hbm_peak"), "peak_GBs": nt.get("peak_GBs")}
    g = result.get("gather")
    if g:
        line["gather"] = {k: (v if not isinstance(v, str) else v[:80]) for k, v in g.items()
                          if k in ("collective", "ms_per_step", "ms", "GBs_per_rank", "value_with_gather", "error", "fallback")}
    if result.get("cabi_gather_error"):                              # tfhe_comm_create / tfhe_gather failed: torch's collective ran instead
        line.setdefault("gather", {})["fallback"] = str(result["cabi_gather_error"])[:120]
    for key in ("other_configs", "configs_multi"):
        if result.get(key):
            line[key] = [_compact_config(c) for c in result[key]]
    if result.get("errors"):
        line["errors"] = [str(e)[:120] for e in result["errors"]][:4]
    line["full_record"] = os.path.relpath(FULL_RECORD, ROOT)
    line = _r(line)
    shrink = [lambda: [c.pop("kern", None) for k in ("other_configs", "configs_multi") for c in line.get(k, [])],
              lambda: [c.pop(f, None) for k in ("other_configs", "configs_multi") for c in line.get(k, []) for f in ("resc_s", "batch", "ms_pass", "ok")],
              lambda: line.pop("other_configs", None), lambda: line.pop("configs_multi", None), lambda: line.pop("gather", None),
              lambda: line.pop("errors", None), lambda: line.pop("ntt", None)]
    for step in shrink:
        if len(json.dumps(line)) < limit:
            break
        step()
    # the guarantee, enforced: the unbounded parts that survive every step above (config extras, roofline, cpu_baseline)
    # go next, last of all everything but the contract keys with the strings cut -- an oversized line is what the driver truncates
    for k in ("full_record", "cpu_baseline", "roofline"):
        if len(json.dumps(line)) < limit:
            break
        if k == "roofline" and isinstance(line.get(k), dict):
            line[k] = {f: line[k][f] for f in ("bound", "achieved", "peak", "unit", "frac", "traffic") if f in line[k]}
        elif k == "cpu_baseline" and isinstance(line.get(k), dict):
            line[k] = {f: line[k][f] for f in ("value", "unit", "cores", "kind") if f in line[k]}
        else:
            line.pop(k, None)
    if len(json.dumps(line)) >= limit:
        cfgw = str((line.get("config") or {}).get("workload", ""))[:80]
        line = {k: (v if not isinstance(v, str) else v[:60]) for k, v in line.items()
                if k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")}
        line["config"] = {"workload": cfgw}
        line["truncated"] = True
    assert len(json.dumps(line)) < limit or limit < 600, "compact_line: contract keys alone exceed the limit"
    return line


def emit(result):
    """full record -> profiles/bench_full.json (+ gpurun_out/ when present) and stderr; compact line -> the LAST stdout line"""
    full = json.dumps(result)
    for path in (FULL_RECORD, os.path.join(ROOT, "gpurun_out", "bench_full.json")):
        try:
            if os.path.isdir(os.path.dirname(path)):
                with open(path, "w") as f:
                    f.write(full + "\n")
        except OSError:
            pass
    print(full, file=sys.stderr, flush=True)
    print(json.dumps(compact_line(result)), flush=True)


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(respawn(args))

    import numpy as np
    import torch

    import toyfhe_jl_amd as tf
    from toyfhe_jl_amd import dist as tdist

    world, rank, local_rank = tdist.env_world()
    ndev = torch.cuda.device_count()
    if ndev == 0:
        raise SystemExit("bench.py needs a GPU (the engine has no CPU path)")
    if world > ndev and args.backend == "nccl":
        raise SystemExit(f"{world} ranks but {ndev} visible GPUs (one rank per GPU)")
    dev_index = local_rank % ndev
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if args.backend == "nccl":
        tdist.init(backend="nccl", device_id=dev)
    else:
        tdist.init(backend=args.backend)
    coll_dev = dev if args.backend == "nccl" else None
    tf.native.check(tf.native.lib().tfhe_set_device(dev_index))

    if args.config != "bfv":
        # one of the other BASELINE.json configurations as THE line of this run (same contract: barrier + sync around exactly
        # `steps` steps after `warmup`, max over ranks, whole-job rate)
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import bench_configs                                        # imports oracle.ref_cpu as the checker of each rank's shard
        rec = bench_configs.multi_case(args.config, tdist, world, rank, coll_dev, steps=args.steps, warmup=args.warmup, total=args.total or None,
                                       gather="cabi" if args.cabi_gather else "torch", backend=args.backend)
        if rank == 0:
            line = {"metric": f"{rec.get('unit', '')} -- {rec['config']}", "value": rec.get("value"), "unit": rec.get("unit"), "n_gpus": world,
                    "steps": args.steps, "warmup": args.warmup, "ms_per_step": rec.get("ms_per_step"), "higher_is_better": True,
                    "scaling": rec.get("scaling"), "vs_baseline": None, "dtype": "u64", "data": "synthetic",
                    "config": {"workload": rec["config"], "global_units": rec.get("global_units"), "sharding": f"units x{world}, no data-path collective"}}
            line.update({k: v for k, v in rec.items() if k not in line and k != "config"})
            if rec.get("algorithmic_GBs") is not None:                  # SURVEY 8(d) bytes of a key switch (4 level N 8) x rate, per GPU, against HBM
                line["roofline"] = {"bound": "hbm", "achieved": rec["algorithmic_GBs"] / world, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                    "frac": rec["algorithmic_GBs"] / world / HBM_PEAK_GBS, "traffic": None,
                                    "note": "algorithmic bytes per GPU; counters per kernel: profiles/pmc_configs.json"}
            else:                                                       # cfg#5 (the MNIST pass: ~30 kernels): no single launch to price
                line["roofline"] = {"bound": "hbm", "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None, "traffic": None,
                                    "note": "a pass of ~500 launches; per-kernel counters: profiles/pmc_configs.json (MNIST case)"}
            emit(line)
        if world > 1:
            torch.distributed.destroy_process_group()
        return

    N = 1 << LOGN
    primes = prime_chain(tf, 50, LBIG, N)
    assert primes[:2] == [1125899908022273, 1125899908612097]
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
    gen.manual_seed(0xF4E5EED)                                      # the evaluation key is shared by every rank
    evk = uniform((L, 2))
    out = torch.empty((B, 2, L, N), dtype=torch.int64, device=dev)

    def step():
        plan.mul_relin(evk.data_ptr(), L, c1.data_ptr(), c2.data_ptr(), out.data_ptr(), B)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    ctx.prof_enable(True)
    tdist.barrier()
    torch.cuda.synchronize()
    evs = [tf.Event() for _ in range(args.steps + 1)]                 # per-step device time (median of >= 5)
    t0 = time.perf_counter()
    for k in range(args.steps):
        evs[k].record(ctx)
        step()
    evs[args.steps].record(ctx)
    torch.cuda.synchronize()
    tdist.barrier()
    t1 = time.perf_counter()
    step_ms = sorted(evs[k].elapsed_ms(evs[k + 1]) for k in range(args.steps))
    median_ms = step_ms[len(step_ms) // 2] if len(step_ms) % 2 else 0.5 * (step_ms[len(step_ms) // 2 - 1] + step_ms[len(step_ms) // 2])
    median_ms = tdist.max_over_ranks(median_ms, device=coll_dev)
    launches, limb_polys, ntt_ms = ctx.prof_read()
    ctx.prof_enable(False)
    elapsed = tdist.max_over_ranks(t1 - t0, device=coll_dev)

    total_units = B * world * args.steps
    value = total_units / elapsed

    # ---- optional final gather (all ranks end up with every result), timed on its own -----------------------------
    gather = None
    if world > 1 and not args.no_gather:
      try:                                                              # never let the optional leg cost the headline line
          import torch.distributed as dist
          flat = out.view(-1)
          comm, fallback, run_gather = None, None, None
          if args.backend == "nccl":
              full = torch.empty(world * flat.numel(), dtype=flat.dtype, device=dev)
          if args.backend == "nccl" and args.cabi_gather:
              try:                                                       # the library's own communicator; a failure is a one-line reason, not a lost leg
                  comm = tdist.make_comm()                               # tfhe_comm_create over the same ranks (rendezvous with a deadline)
              except Exception as e:                                     # noqa: BLE001
                  comm, fallback = None, f"tfhe_comm_create: {type(e).__name__}: {e}"[:160]
              # every rank must take the same branch BEFORE the first collective on the communicator is issued (a rank
              # that failed to create it would leave the others waiting inside tfhe_gather): one rank's failure sends all to torch's
              ok_all = tdist.max_over_ranks(0.0 if comm is not None else 1.0, device=coll_dev) == 0.0
              if not ok_all and comm is not None:
                  comm, fallback = None, "tfhe_comm_create failed on another rank"
              if comm is not None:
                  run_gather = lambda: comm.gather(ctx, flat.data_ptr(), full.data_ptr(), flat.numel())
          if run_gather is None and args.backend == "nccl":
              run_gather = lambda: dist.all_gather_into_tensor(full, flat)
          elif run_gather is None:
              run_gather = lambda: tdist.gather_results(out[:8].cpu())    # functional check only
          run_gather()
          torch.cuda.synchronize(); tdist.barrier()
          g0 = time.perf_counter()
          greps = 3
          for _ in range(greps):
              run_gather()
          torch.cuda.synchronize(); tdist.barrier()
          g_s = tdist.max_over_ranks((time.perf_counter() - g0) / greps, device=coll_dev)
          gbytes = flat.numel() * 8 * (world - 1)                        # received per rank
          gather = {"collective": ("tfhe_gather (C ABI, ncclAllGather over xGMI)" if comm is not None else "all_gather_into_tensor (RCCL over xGMI)")
                    if args.backend == "nccl" else f"{args.backend} functional check",
                    "ms_per_step": g_s * 1e3, "bytes_received_per_rank": gbytes, "GBs_per_rank": gbytes / g_s / 1e9,
                    "value_with_gather": B * world / (elapsed / args.steps + g_s)}
          if fallback:
              gather["fallback"] = fallback
          if args.backend == "nccl":
              del full
      except Exception as e:                                          # noqa: BLE001
        gather = {"error": f"{type(e).__name__}: {e}"}

    # ---- roofline of the dominant kernels (k_bfv_core_fused + k_ks_fused) --------------------------------------------
    # (1) transform-equivalent bytes: SURVEY 8(d)'s unit (one limb transform = 2*N*8 B) x the transforms the launches carry
    ntt_bytes = limb_polys * 2 * N * 8
    ntt_s = ntt_ms * 1e-3
    teq = ntt_bytes / ntt_s / 1e9 if ntt_s > 0 else 0.0
    # (2) what binds: the fp64 vector ALU.  Wave64 VALU instructions per ciphertext-mul of the two kernels come from the PMC
    # pass over this very script (SQ_INSTS_VALU, profiles/pmc_bench_kernels.json, tools/pmc_round.sh); bench.py cannot
    # run counters itself, so it scales the committed per-ciphertext figures to this run's launches and divides by the
    # HIP-event kernel time measured in this run.
    pmc = load_pmc()
    valu_gips = valu_frac = hbm_frac = traffic = clock_ghz = valu_frac_clk = None
    sq = {}
    src_id = source_id()
    pmc_stale = bool(pmc) and pmc.get("source_id") != src_id           # counters of another build: not scaled onto this run
    if pmc and launches and not pmc_stale:
        fused = {k: v for k, v in pmc["kernels"].items() if "k_bfv_core_fused" in k or "k_ks_fused" in k}
        cts = B * args.steps                                          # ciphertext-muls carried by this rank's launches
        per_ct = lambda key: sum(v.get(key, 0.0) / v["launches"] for v in fused.values()) / pmc["batch"]
        traffic = per_ct("hbm_bytes") * cts / launches
        hbm_frac = per_ct("hbm_bytes") * cts / ntt_s / (HBM_PEAK_GBS * 1e9)
        if all("SQ_INSTS_VALU" in v for v in fused.values()):
            insts = per_ct("SQ_INSTS_VALU") * cts
            valu_gips = insts / ntt_s / 1e9
            valu_frac = valu_gips / VALU_PEAK_GIPS
            gui, dur = per_ct("GRBM_GUI_ACTIVE"), per_ct("duration_ns")
            if gui and dur:
                clock_ghz = gui / 8 / dur                             # GRBM_GUI_ACTIVE is summed over the 8 XCDs; shader clock of the profiled pass
                valu_frac_clk = per_ct("SQ_INSTS_VALU") * 4.0 / (N_SIMD * gui / 8)   # counters only: issue slots used / available in that pass
            for k in ("SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU",
                      "SQ_ACTIVE_INST_LDS", "SQ_INSTS_VALU", "SQ_INSTS_LDS"):
                if all(k in v for v in fused.values()):
                    sq[k + "_per_ctmul"] = per_ct(k)
    # (3) the same on the USEFUL-work scale: only the butterflies (8 fp64 instructions each: one 6-instruction modular
    # product, one sum, one difference) and the modular products of the tensor / key inner product (6 + 1 accumulate) count --
    # range sweeps, conversions, address arithmetic and moves do not.  Per ciphertext-mul: 17 limbs x 7 + 8 limbs x 10 = 199 limb
    # transforms of 14 x 2^13 butterflies, 17 x 4 tensor products and 8 x 8 x 2 key products per coefficient.
    ess_per_ct = (LBIG * 7 + L * (L + 2)) * LOGN * (N // 2) * 8 / 64.0 + LBIG * 4 * N * 6 / 64.0 + L * L * 2 * N * 7 / 64.0   # wave64 instructions
    essential_frac = ess_per_ct * B * args.steps / ntt_s / 1e9 / VALU_PEAK_GIPS if ntt_s > 0 else None
    # (4) on SURVEY 8(d)'s own unit: a ciphertext-mul moves 6 L N 8 = 6 MiB compulsory bytes (4 polynomials in, 2 out)
    ctmul_bytes = 6 * L * N * 8
    traffic_per_ctmul = None
    if pmc and not pmc_stale:
        # every kernel of the step: per launch (a launch covers pmc["batch"] ciphertexts; the expansion runs twice per step)
        per_launch = lambda k, v: v.get("hbm_bytes", 0.0) / max(1, v["launches"]) * (2 if "k_bfv_expand" in k else 1)
        traffic_per_ctmul = sum(per_launch(k, v) for k, v in pmc["kernels"].items()) / pmc["batch"]
    roof = {
        "bound": "valu-fp64" if valu_frac is not None else "hbm",
        "kernel": "k_bfv_core_fused + k_ks_fused: the 2^14-point negacyclic NTTs (exact-integer fp64 butterflies) fused with the "
                  "tensor product / key inner product -- 7 and 10 limb transforms per workgroup item; durations by HIP events "
                  "around these launches inside the timed region",
        "achieved": valu_gips if valu_frac is not None else teq,
        "peak": VALU_PEAK_GIPS if valu_frac is not None else HBM_PEAK_GBS,
        "unit": "G wave64-VALU-instr/s (1024 SIMDs x 2.4 GHz / 4 clk)" if valu_frac is not None else "GB/s",
        "frac": valu_frac if valu_frac is not None else teq / HBM_PEAK_GBS,
        "valu_issue_util_profiled_pass": valu_frac_clk, "profiled_clock_GHz": clock_ghz,
        "hbm_frac": hbm_frac,
        "essential_frac": essential_frac, "essential_wave_instr_per_ctmul": ess_per_ct,
        "essential_note": "butterflies (8 fp64 instructions) + tensor / key products only, / kernel time of the two fused kernels / 614.4 G/s: "
                          "the fraction of the fp64 vector peak spent on arithmetic the algorithm needs (frac counts every issued VALU instruction)",
        "ctmul_hbm_frac": value / max(1, world) * ctmul_bytes / 1e9 / HBM_PEAK_GBS,
        "ctmul_algorithmic_bytes": ctmul_bytes, "traffic_per_ctmul": traffic_per_ctmul,
        "traffic_per_ctmul_over_algorithmic": (traffic_per_ctmul / ctmul_bytes) if traffic_per_ctmul else None,
        "traffic": traffic,
        "traffic_unit": "HBM-side bytes per launch (PMC FETCH_SIZE x2 + WRITE_SIZE of the two kernels, profiles/pmc_bench_kernels.json, "
                        "scaled to this run's launches)",
        "transform_equiv": {"achieved": teq, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": teq / HBM_PEAK_GBS,
                            "note": "limb transforms x 2*N*8 algorithmic bytes (SURVEY 8d) / kernel time: prices transform throughput, "
                                    "not HBM utilisation (the fused kernels move fewer bytes than this)"},
        "algorithmic_bytes_per_launch": ntt_bytes / launches if launches else None,
        "launches": launches, "limb_ntts": limb_polys,
        "avg_launch_ms": ntt_ms / launches if launches else None,
        "ntt_share_of_step": ntt_s / (t1 - t0) if t1 > t0 else None,
        "counters": sq or None,
        "pmc_source": (pmc or {}).get("note"),
        "pmc_stale": pmc_stale, "source_id": src_id, "pmc_source_id": (pmc or {}).get("source_id"),
    }

    result = {
        "metric": "ciphertext-mul/s (BFV ct*ct + relinearize, N=2^14, L=8 RNS)",
        "value": value,
        "unit": "ciphertext-mul/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "ms_per_step_median": median_ms, "value_at_median_step": B * world / (median_ms * 1e-3),
        "nranks_seen": tdist.world_size_seen(),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u64",
        "data": "synthetic",
        "config": {"workload": "BFV N=2^14, L=8 RNS limbs (50-bit primes), extension basis 17 limbs, t=65537, "
                               "ciphertext-mul + relinearize (RNS-digit keyswitch), bit-exact", "batch_per_gpu": B,
                   "global_batch": B * world, "sharding": f"batch x{world}, no data-path collective"},
        "roofline": roof,
    }
    if gather is not None:
        result["gather"] = gather

    # ---- BASELINE.md #2': stand-alone NTT, 4096 polys x 8 limbs, N = 2^14 -------------------------------------------
    def ntt_record():
        polys = 4096
        rows = polys * L
        a = torch.randint(0, qs[0], (rows, N), dtype=torch.int64, device=dev, generator=gen)
        b = torch.empty_like(a)
        ev = [tf.Event() for _ in range(4)]
        reps = 10
        for _ in range(3):
            ctx.nntt(a.data_ptr(), b.data_ptr(), polys, L)
            ctx.inntt(b.data_ptr(), a.data_ptr(), polys, L)
        ev[0].record(ctx)
        for _ in range(reps):
            ctx.nntt(a.data_ptr(), b.data_ptr(), polys, L)
        ev[1].record(ctx)
        for _ in range(reps):
            ctx.inntt(b.data_ptr(), a.data_ptr(), polys, L)
        ev[2].record(ctx)
        fwd_s, inv_s = ev[0].elapsed_ms(ev[1]) / reps * 1e-3, ev[1].elapsed_ms(ev[2]) / reps * 1e-3
        gb = rows * N * 8 * 2 / 1e9
        result["ntt"] = {"workload": f"{polys} polys x {L} limbs, N=2^14, 50-bit primes, out of place (BASELINE.md #2')",
                         "bytes_per_pass": rows * N * 8 * 2, "fwd_GBs": gb / fwd_s, "inv_GBs": gb / inv_s,
                         "fwd_frac_of_hbm_peak": gb / fwd_s / HBM_PEAK_GBS, "inv_frac_of_hbm_peak": gb / inv_s / HBM_PEAK_GBS,
                         "limb_ntts_per_s_fwd": rows / fwd_s, "limb_ntts_per_s_inv": rows / inv_s, "peak_GBs": HBM_PEAK_GBS}

    # ---- CPU baseline: the C restatement of the reference algorithm, 1 thread (the reference is single-threaded) and all cores
    def cpu_record():
        from oracle import ref_cpu                                   # checker / baseline only (never on the product path)
        rl = ref_cpu.lib()
        cores = rl.ref_num_threads()
        rs, rb = ref_cpu.RefCtx(N, qs), ref_cpu.RefCtx(N, primes)
        rng = np.random.default_rng(1)

        def sample(n):
            cols = lambda pre: np.stack([rng.integers(0, q, size=tuple(pre) + (N,), dtype=np.uint64) for q in qs], axis=len(pre))
            return cols((n, 2)), cols((n, 2)), cols((L, 2))

        def run(n):
            s1, s2, sk = sample(n)
            tc = time.perf_counter()
            prod = ref_cpu.bfv_mul(rs, rb, T_PLAIN, s1, s2)
            rs.keyswitch(L, False, sk, prod)
            return time.perf_counter() - tc

        ns = args.cpu_sample or max(8 * cores, 8)
        t_all = run(ns)
        rl.ref_set_threads(1)
        n1 = 4
        t_one = run(n1)
        rl.ref_set_threads(cores)
        result["cpu_baseline"] = {"value": ns / t_all, "unit": "ciphertext-mul/s", "cores": cores, "kind": "port",
                                  "sample": f"{ns} ciphertext pairs of the same workload, oracle/ref_cpu.c "
                                            f"(exact BigInt-style conversions, radix-2 NTT), OpenMP over the batch, {t_all:.1f} s",
                                  "single_thread": {"value": n1 / t_one, "cores": 1,
                                                    "sample": f"{n1} ciphertext pairs, 1 thread (the reference is single-threaded), {t_one:.1f} s"}}
    # ---- the other BASELINE.json configurations (secondary rates; every case oracle-checked before it is timed) -----------
    def configs_record():
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import bench_configs                                        # imports oracle.ref_cpu as the checker of each case
        result["other_configs"] = bench_configs.run(args.configs_scale)

    # ---- multi-rank runs: the configurations BASELINE.json defines across the GPUs of a node (cfg#4 strong-scaled key switches,
    # cfg#5 the MNIST test set), every rank on its shard; a failure anywhere is a record with "error", never a lost line ------------
    if world > 1 and not args.no_configs:
        try:
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import bench_configs
            multi = []
            for cfg, st_, wu_ in (("cfg4", 5, 2), ("cfg5", 1, 1)):
                r_ = bench_configs.multi_case(cfg, tdist, world, rank, coll_dev, steps=st_, warmup=wu_, total=(args.total or None) if cfg == "cfg4" else (512 * world if args.total else None),
                                              gather="cabi" if args.cabi_gather else "torch", backend=args.backend)
                if rank == 0:
                    multi.append(r_)
            if rank == 0:
                result["configs_multi"] = multi
        except Exception as e:                                          # noqa: BLE001
            result.setdefault("errors", []).append(f"configs_multi: {type(e).__name__}: {e}")

    for enabled, leg in ((rank == 0 and not args.no_ntt, ntt_record), (rank == 0 and world == 1 and not args.no_cpu, cpu_record),
                         (rank == 0 and world == 1 and not args.no_configs, configs_record)):
        if enabled:
            try:                                                        # a failing side record must not cost the headline line
                leg()
            except Exception as e:                                      # noqa: BLE001
                result.setdefault("errors", []).append(f"{leg.__name__}: {type(e).__name__}: {e}")
    if rank == 0:
        emit(result)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
