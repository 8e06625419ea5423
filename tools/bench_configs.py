#!/usr/bin/env python3
"""Secondary measurements on the other BASELINE.json configurations (not the bench.py metric): device time per
operation on synthetic residues, 1 GPU.  Every case first checks one ciphertext of its own batch against the C oracle
(bit for bit) and only then times.
  cfg#3  CKKS  N=2^15, 10 limbs @40 bit + special prime: rotate (galois + key switch), rescale       batch 512
  cfg#4        N=2^14,  6 limbs @50 bit + special prime: key switch only                             batch 512 (4096 / 8 GPUs)
  cfg#5  CKKS  N=2^16, the infer.jl ring 60 + 5 x 40 + 60 bit: key switch / rotate / rescale / NTT   batch 64
  cfg#5' same shape on a uniform 50-bit chain (fp64 policy throughout), for comparison
usage: bench_configs.py [scale] [cases]   (scale divides the batches, default 1; cases = positions to run, e.g. 1,3 -- default all)"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import toyfhe_jl_amd as tf
from oracle import ref_cpu            # checker only

HBM_PEAK_GBS = 8000.0
PROFILE = os.environ.get("TFHE_CFG_PROFILE", "0") not in ("", "0")
NOCHECK = os.environ.get("TFHE_CFG_NOCHECK", "0") not in ("", "0")   # ablation builds (-DTFHE_ABL_*: wrong results by design); records say so
RECORDS = []                          # one dict per case (bench.py puts them into its JSON line as `other_configs`)


def chain(start, n, N):
    out, p = [], tf.nextprime(start, 1, 2 * N)
    for _ in range(n):
        out.append(p)
        p = tf.nextprime(p + 2 * N, 1, 2 * N)
    return out


def timed(ctx, f, reps=8):
    """best of three rounds after >= 0.1 s of warm-up (the oracle check before each case idles the GPU for seconds and the
    clocks take tens of milliseconds to come back).  TFHE_CFG_PROFILE=1 (tools/pmc_configs.sh, counter passes): three warm calls
    and five counted ones -- the folding script drops the first quarter of every kernel's dispatches."""
    if PROFILE:
        for _ in range(8):
            f()
        ctx.sync()
        return 1.0
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.1:
        f()
        ctx.sync()
    best = 1e9
    for _ in range(3):
        ctx.sync()
        t = time.perf_counter()
        for _ in range(reps):
            f()
        ctx.sync()
        best = min(best, (time.perf_counter() - t) / reps)
    return best


def uniform(ctx, level, count, seed):
    b = tf.DeviceBuffer(count * level * ctx.N)
    ctx.sample_uniform(level, seed, 0, 0, b.ptr, count)
    return b


def rows(buf, shape, k):
    n = int(np.prod(shape[1:]))
    out = np.empty(n, dtype=np.uint64)
    tf.native.check(tf.native.lib().tfhe_memcpy_d2h(out.ctypes.data, buf.ptr + k * n * 8, n * 8))
    return out.reshape((1,) + tuple(shape[1:]))


def keyswitch_case(name, N, qs, b):
    Lk, level = len(qs), len(qs) - 1
    ctx, ref = tf.Context(N, qs), ref_cpu.RefCtx(N, qs)
    evk = uniform(ctx, Lk, Lk * 2, 7)                        # [Lk digits][2][Lk][N]
    ct = uniform(ctx, level, b * 2, 8)
    out = tf.DeviceBuffer(b * 2 * level * N)
    g = pow(3, 2 * N - 1, 2 * N)
    evk_h = evk.to_numpy((Lk, 2, Lk, N))
    k = b // 2
    cin = rows(ct, (b, 2, level, N), k)
    ctx.keyswitch(Lk, level, True, evk.ptr, Lk, ct.ptr, 2, out.ptr, b)
    assert NOCHECK or np.array_equal(rows(out, (b, 2, level, N), k), ref.keyswitch(level, True, evk_h, cin)), name + ": key switch differs from the oracle"
    ctx.rotate(Lk, level, True, evk.ptr, Lk, g, ct.ptr, out.ptr, b)
    want = ref.keyswitch(level, True, evk_h, ref.galois(g, cin.reshape(-1, level, N), idx=range(level)).reshape(cin.shape))
    assert NOCHECK or np.array_equal(rows(out, (b, 2, level, N), k), want), name + ": rotate differs from the oracle"
    res = tf.DeviceBuffer(b * 2 * (level - 1) * N)
    ctx.rescale(out.ptr, res.ptr, b * 2, level)
    assert NOCHECK or np.array_equal(rows(res, (b, 2, level - 1, N), k), ref.modswitch(want.reshape(-1, level, N), idx=range(level)).reshape(1, 2, level - 1, N))
    t_ks = timed(ctx, lambda: ctx.keyswitch(Lk, level, True, evk.ptr, Lk, ct.ptr, 2, out.ptr, b))
    t_rot = timed(ctx, lambda: ctx.rotate(Lk, level, True, evk.ptr, Lk, g, ct.ptr, out.ptr, b))
    t_rs = timed(ctx, lambda: ctx.rescale(ct.ptr, res.ptr, b * 2, level))
    print(f"{name}: N=2^{N.bit_length() - 1}, level {level} (+special), batch {b}: keyswitch {b / t_ks:9.0f}/s  rotate {b / t_rot:9.0f}/s  "
          f"rescale {b / t_rs:9.0f} ct/s   [oracle-checked]", file=sys.stderr)
    # algorithmic bytes per unit (BASELINE.md section 3 / SURVEY 8d): a key switch reads the ciphertext (2 polys) and writes 2
    # polys at the ciphertext's level (the key is shared by the batch); a rescale reads `level` and writes `level - 1` limbs
    ks_bytes, rs_bytes = 4 * level * N * 8, 2 * (2 * level - 1) * N * 8
    RECORDS.append({"config": name, "N": N, "level": level, "special_prime": True, "batch": b, "oracle_checked": not NOCHECK,
                    "moduli_bits": [int(q).bit_length() for q in qs],
                    "keyswitch_per_s": b / t_ks, "rotate_per_s": b / t_rot, "rescale_ct_per_s": b / t_rs,
                    "keyswitch_algorithmic_GBs": b / t_ks * ks_bytes / 1e9, "keyswitch_frac_of_hbm_peak": b / t_ks * ks_bytes / 1e9 / HBM_PEAK_GBS,
                    "rescale_algorithmic_GBs": b / t_rs * rs_bytes / 1e9, "rescale_frac_of_hbm_peak": b / t_rs * rs_bytes / 1e9 / HBM_PEAK_GBS,
                    "limb_ntts_per_keyswitch": level * (level + 1) + 2 * (level + 1)})
    return ctx, ref


def ntt_case(name, N, qs, polys):
    L = len(qs)
    ctx, ref = tf.Context(N, qs), ref_cpu.RefCtx(N, qs)
    a = uniform(ctx, L, polys, 9)
    b2 = tf.DeviceBuffer(polys * L * N)
    ctx.nntt(a.ptr, b2.ptr, polys, L)
    k = polys // 2
    assert np.array_equal(rows(b2, (polys, L, N), k), ref.nntt(rows(a, (polys, L, N), k))), name + ": NTT differs from the oracle"
    c = tf.DeviceBuffer(polys * L * N)
    ctx.inntt(b2.ptr, c.ptr, polys, L)
    assert np.array_equal(rows(c, (polys, L, N), k), rows(a, (polys, L, N), k))
    gb = polys * L * N * 16 / 1e9
    t_f = timed(ctx, lambda: ctx.nntt(a.ptr, b2.ptr, polys, L))
    t_i = timed(ctx, lambda: ctx.inntt(b2.ptr, c.ptr, polys, L))
    print(f"{name}: N=2^{N.bit_length() - 1}, {L} limbs, {polys} polys: NTT fwd {gb / t_f:6.0f} GB/s  inv {gb / t_i:6.0f} GB/s   [oracle-checked]", file=sys.stderr)
    RECORDS.append({"config": "NTT " + name, "N": N, "limbs": L, "polys": polys, "oracle_checked": True,
                    "moduli_bits": [int(q).bit_length() for q in qs], "bytes_per_pass": polys * L * N * 16,
                    "fwd_GBs": gb / t_f, "inv_GBs": gb / t_i, "fwd_frac_of_hbm_peak": gb / t_f / HBM_PEAK_GBS,
                    "inv_frac_of_hbm_peak": gb / t_i / HBM_PEAK_GBS})


def mnist_case(name, logn, sets, reference_shape=False):
    """BASELINE.json configs[4] end to end: the encrypted CNN of examples/encrypted_mnist.py (infer.jl's circuit on the reference's
    trained weights, synthetic images); third pass, weights encoded by the first.  The logits are checked against the float64
    model inside run().  Two circuit shapes: the restructured one (63 Galois keys, hoisted rotations, one tfhe_matmul_diag per
    matrix product, one tfhe_lincomb_many per convolution) and -- reference_shape -- the reference's own (infer.jl:140-149: ONE Galois
    key, 63 CHAINED rotations per product, a ring product and sum per diagonal)."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples"))
    import encrypted_mnist as em
    st = {}
    err, rng_, agree = em.run(logn, 0, verbose=False, batches=sets, hoisted=not reference_shape, repeat=3, fused=not reference_shape, stats=st)
    # 63 chained rotations accumulate 63 times the rotation noise of the hoisted form: the logits stay within the same tolerance
    assert err < 5e-3 * max(1.0, rng_) and agree > 0.99, name + ": logits differ from the float64 model"
    print(f"{name}: {st['images']} images in {st['eval_s'] * 1e3:.1f} ms = {st['images_per_s']:.0f} images/s  (max logit error {err:.2e}, argmax agreement {agree})", file=sys.stderr)
    RECORDS.append({"config": name, "N": 1 << logn, "ciphertext_sets": sets, "images": st["images"], "images_per_s": st["images_per_s"],
                    "ms_per_pass": st["eval_s"] * 1e3, "host_enqueue_ms": st["host_enqueue_s"] * 1e3, "device_span_ms": st["device_span_s"] * 1e3,
                    "host_share_of_pass": st["host_enqueue_s"] / st["eval_s"],
                    "timing_note": "ms_per_pass = host clock from the first launch to the decoded logits; host_enqueue_ms = until the last launch of the "
                                   "circuit is enqueued (the host runs ahead of the device when this is well below the pass); device_span_ms = HIP events "
                                   "around the circuit on its stream",
                    "circuit": "infer.jl:140-149 as written: one Galois key, 63 chained rotations per matrix product" if reference_shape
                               else "restructured: 63 Galois keys, hoisted rotations (one digit decomposition per product), tfhe_matmul_diag / tfhe_lincomb_many",
                    "images_data": "synthetic (no MNIST on disk); weights: the reference's trained model (tests/golden/mnist_conv.npz)",
                    "max_logit_error": err, "logit_range": rng_, "argmax_agreement": agree,
                    "oracle_checked": "float64 model (infer.jl:55-88)"})


# ---------------------------------------------------------------------------------------------------------------------------------
# Multi-rank modes (bench.py --gpus N --config cfg3|cfg4|cfg5, and `configs_multi` of the default multi-rank line): BASELINE.json
# defines cfg#4 as 4096 key switches sharded over 8 GPUs and cfg#5 as the full MNIST test set across 8 GPUs.  One process per
# GPU; every rank owns a contiguous shard (dist.shard), replicates the context and the key, no data-path collective; the timed
# region is bracketed by barrier + device sync, the rate is the whole job's units / the slowest rank's time; the per-rank
# times are all-gathered (imbalance), and the final gather of the results is timed separately.
# ---------------------------------------------------------------------------------------------------------------------------------
MULTI = {
    "cfg3": dict(name="cfg#3 CKKS N=2^15 10x40-bit + special prime: rotate (galois + key switch) + rescale", logn=15, bits=40, limbs=11,
                 per_gpu=512, scaling="weak", op="rotate_rescale", unit="rotate+rescale/s"),
    "cfg4": dict(name="cfg#4 N=2^14 6x50-bit + special prime: key switch, batch 4096 sharded over the ranks", logn=14, bits=50, limbs=7,
                 total=4096, scaling="strong", op="keyswitch", unit="keyswitch/s"),
    "cfg5": dict(name="cfg#5 encrypted MNIST inference N=2^16 (infer.jl ring), the 10 000-image test set sharded over the ranks",
                 scaling="strong", op="mnist", unit="images/s"),
}


def multi_case(cfg, tdist, world, rank, coll_dev, steps=3, warmup=1, total=None, gather="torch", backend="nccl"):
    """One multi-rank case; every rank calls it, rank 0 gets the record (others None)."""
    import torch
    import torch.distributed as dist
    spec = MULTI[cfg]
    have_pg = world > 1 and dist.is_available() and dist.is_initialized()

    def all_ranks(x):
        if not have_pg:
            return [x]
        outl = [None] * world
        dist.all_gather_object(outl, x)
        return outl

    rec = {"config": spec["name"], "scaling": spec["scaling"], "n_gpus": world, "nranks_seen": tdist.world_size_seen(), "steps": steps, "warmup": warmup}
    import gc
    gc.collect()                                                     # as run(): a mode starts with an empty allocator cache on every rank
    tf.native.check(tf.native.lib().tfhe_alloc_trim())
    if spec["op"] == "mnist":
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples"))
        import encrypted_mnist as em
        images_total = total or 10000
        per_set = (1 << 16) // 128
        sets_total = -(-images_total // per_set)                     # 20 ciphertext sets of 512 images cover the test set
        start, count = tdist.shard(sets_total, rank, world)
        st, fail_ = {}, None
        tdist.barrier()
        try:
            if count:
                err, rng_, agree = em.run(16, rank, verbose=False, batches=count, hoisted=True, repeat=warmup + steps, fused=True, stats=st)
                assert err < 5e-3 * max(1.0, rng_) and agree > 0.99, "logits differ from the float64 model"
        except Exception as e:  # noqa: BLE001 -- reported by rank 0; the other ranks must still reach the collectives below
            fail_ = f"rank {rank}: {type(e).__name__}: {e}"
        mine = st.get("eval_s", 0.0)                                 # the last pass of this rank's sets
        per_rank = all_ranks({"rank": rank, "sets": count, "pass_ms": mine * 1e3, "images_per_s": count * per_set / mine if mine else 0.0, "error": fail_})
        if any(r["error"] for r in per_rank):
            return {"config": spec["name"], "error": "; ".join(r["error"] for r in per_rank if r["error"])} if rank == 0 else None
        slow = max(r["pass_ms"] for r in per_rank) * 1e-3
        rec.update(unit=spec["unit"], value=sets_total * per_set / slow, ms_per_step=slow * 1e3, global_units=sets_total * per_set,
                   images="synthetic (no MNIST on disk; the reference's trained weights)", per_rank=per_rank, oracle_checked="float64 model (infer.jl:55-88), every rank",
                   imbalance_max_over_min=max(r["pass_ms"] for r in per_rank) / max(1e-9, min(r["pass_ms"] for r in per_rank if r["sets"])),
                   gather=None, note="one evaluation pass per rank over its own ciphertext sets after %d warm passes; logits decrypted per rank "
                                     "(nothing to gather but 10 numbers per image)" % warmup)
        return rec if rank == 0 else None

    N = 1 << spec["logn"]
    qs = chain(2 ** spec["bits"] + 1, spec["limbs"], N)
    Lk, level = len(qs), len(qs) - 1
    if spec["scaling"] == "strong":
        g = total or spec["total"]
        start, b = tdist.shard(g, rank, world)
    else:
        b = total or spec["per_gpu"]
        g, start = b * world, rank * b
    fail_ = None
    try:
        ctx, ref = tf.Context(N, qs), ref_cpu.RefCtx(N, qs)
        setup = _multi_setup(spec, ctx, ref, N, qs, Lk, level, b, start, warmup, rank)
    except Exception as e:  # noqa: BLE001 -- every rank must reach the collectives below: agree on the outcome first
        fail_ = f"rank {rank}: {type(e).__name__}: {e}"
    fails = [f for f in all_ranks(fail_) if f]
    if fails:
        return {"config": spec["name"], "error": "; ".join(fails)} if rank == 0 else None
    op, out, res = setup
    return _multi_timed(spec, rec, tdist, world, rank, coll_dev, steps, gather, backend, ctx, op, out, res, N, qs, level, b, g, all_ranks)


def _multi_setup(spec, ctx, ref, N, qs, Lk, level, b, start, warmup, rank):
    """this rank's shard, the shared key, one ciphertext checked against the oracle, warm-up; returns (op, out, res)"""
    evk = uniform(ctx, Lk, Lk * 2, 7)                                # the key is shared: same seed on every rank
    ct = uniform(ctx, level, max(b, 1) * 2, 8 + 1000 * (start + 1))  # this rank's shard of the global batch
    out = tf.DeviceBuffer(max(b, 1) * 2 * level * N)
    gal = pow(3, 2 * N - 1, 2 * N)
    res = tf.DeviceBuffer(max(b, 1) * 2 * (level - 1) * N) if spec["op"] == "rotate_rescale" else None

    def op():
        if not b:
            return
        if spec["op"] == "keyswitch":
            ctx.keyswitch(Lk, level, True, evk.ptr, Lk, ct.ptr, 2, out.ptr, b)
        else:
            ctx.rotate(Lk, level, True, evk.ptr, Lk, gal, ct.ptr, out.ptr, b)
            ctx.rescale(out.ptr, res.ptr, b * 2, level)
    if b:                                                            # one ciphertext of this rank's shard against the oracle
        k = b // 2
        cin = rows(ct, (b, 2, level, N), k)
        evk_h = evk.to_numpy((Lk, 2, Lk, N))
        op()
        if spec["op"] == "keyswitch":
            want = ref.keyswitch(level, True, evk_h, cin)
            got = rows(out, (b, 2, level, N), k)
        else:
            rot = ref.keyswitch(level, True, evk_h, ref.galois(gal, cin.reshape(-1, level, N), idx=range(level)).reshape(cin.shape))
            want = ref.modswitch(rot.reshape(-1, level, N), idx=range(level)).reshape(1, 2, level - 1, N)
            got = rows(res, (b, 2, level - 1, N), k)
        assert np.array_equal(got, want), spec["name"] + ": differs from the oracle on rank %d" % rank
    for _ in range(max(1, warmup)):
        op()
    ctx.sync()
    op.keep = (evk, ct)
    return op, out, res


def _multi_timed(spec, rec, tdist, world, rank, coll_dev, steps, gather, backend, ctx, op, out, res, N, qs, level, b, g, all_ranks):
    import torch
    import torch.distributed as dist
    tdist.barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        op()
    ctx.sync()
    mine = time.perf_counter() - t0
    tdist.barrier()
    elapsed = tdist.max_over_ranks(time.perf_counter() - t0, device=coll_dev)
    per_rank = all_ranks({"rank": rank, "units": b, "s": mine, "units_per_s": b * steps / mine if mine else 0.0})
    # the final gather (north_star: RCCL over xGMI only for the final gather), timed on its own: every rank ends up with all results
    gat = None
    final = res if res is not None else out
    row_words = 2 * (level - (1 if res is not None else 0)) * N                                 # one result ciphertext
    maxw = max(1, -(-g // world)) * row_words                                                   # shards padded to the largest
    if world > 1:
        try:
            d2d = lambda dst, src, nbytes: tf.native.check(tf.native.lib().tfhe_memcpy_d2d(ctx.h, dst, src, nbytes))
            used, run_g, keep = None, None, []
            if gather == "cabi" and backend == "nccl":
                try:                                                   # never lose the run to the optional communicator: fall back and say so
                    comm = tdist.make_comm()
                    pad, full = tf.DeviceBuffer(maxw), tf.DeviceBuffer(world * maxw)
                    if b:
                        d2d(pad.ptr, final.ptr, b * row_words * 8)
                    run_g = lambda: comm.gather(ctx, pad.ptr, full.ptr, maxw)
                    run_g(); ctx.sync()
                    used, keep = "tfhe_gather (C ABI, ncclAllGather over xGMI)", [comm, pad, full]
                except Exception as e:  # noqa: BLE001
                    rec["cabi_gather_error"] = f"{type(e).__name__}: {e}"
            if used is None and backend == "nccl":
                loc = torch.zeros(maxw, dtype=torch.int64, device=coll_dev)
                fullt = torch.empty(world * maxw, dtype=torch.int64, device=coll_dev)
                if b:
                    d2d(loc.data_ptr(), final.ptr, b * row_words * 8)
                ctx.sync()
                run_g = lambda: dist.all_gather_into_tensor(fullt, loc)
                used = "all_gather_into_tensor (RCCL over xGMI)" + (" -- fallback: the C-ABI communicator failed" if "cabi_gather_error" in rec else "")
            elif used is None:
                small = torch.from_numpy(rows(final, (max(b, 1), row_words), 0).astype(np.int64))
                run_g = lambda: tdist.gather_results(small)
                used = f"{backend} functional check (one ciphertext per rank)"
            run_g()
            torch.cuda.synchronize(); ctx.sync(); tdist.barrier()
            g0 = time.perf_counter()
            for _ in range(3):
                run_g()
            torch.cuda.synchronize(); ctx.sync(); tdist.barrier()
            g_s = tdist.max_over_ranks((time.perf_counter() - g0) / 3, device=coll_dev)
            nbytes = (maxw if backend == "nccl" else row_words) * 8 * (world - 1)
            gat = {"collective": used, "ms": g_s * 1e3, "bytes_received_per_rank": nbytes, "GBs_per_rank": nbytes / g_s / 1e9,
                   "value_with_gather": g / (elapsed / steps + g_s)}
            del keep
        except Exception as e:  # noqa: BLE001
            gat = {"error": f"{type(e).__name__}: {e}"}
    ks_bytes = 4 * level * N * 8
    rates = [r["units_per_s"] for r in per_rank if r["units"]]
    rec.update(unit=spec["unit"], value=g * steps / elapsed, ms_per_step=elapsed / steps * 1e3, global_units=g, N=N, level=level, special_prime=True,
               moduli_bits=[int(q).bit_length() for q in qs], per_rank=per_rank, oracle_checked="one ciphertext per rank, bit for bit",
               imbalance_max_over_min=(max(rates) / min(rates)) if rates else None, gather=gat,
               algorithmic_GBs=g * steps / elapsed * ks_bytes / 1e9, frac_of_hbm_peak_per_gpu=g * steps / elapsed * ks_bytes / 1e9 / HBM_PEAK_GBS / world)
    return rec if rank == 0 else None


PMC_CONFIGS = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "pmc_configs.json")   # tools/pmc_configs.sh, per round


def _source_id():  # bench.py source_id(): which sources the library running here was built from
    import hashlib
    R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    h = hashlib.sha256()
    src = os.path.join(R, "toyfhe.jl_amd", "csrc")
    for f in sorted(os.listdir(src)) + [os.path.join("..", "..", "include", "toyfhe_hip.h")]:
        h.update(f.encode())
        h.update(open(os.path.join(src, f), "rb").read())
    return h.hexdigest()[:16]


def attach_rooflines(records):
    """Every record gets `roofline` objects from the committed counter passes of ITS case (profiles/pmc_configs.json: rocprofv3
    --pmc over tools/bench_configs.py, steady-state launches): per timed operation the dominant kernel, what binds it (wave64
    VALU issue against 614.4 G/s, or HBM-side bytes against 8 TB/s -- whichever fraction is larger), the achieved rate, and the
    HBM-side traffic per unit next to the unit's algorithmic bytes.  bench_configs cannot collect counters itself;
    `pmc_stale` says whether the counters describe the build that is running."""
    import json
    if not os.path.exists(PMC_CONFIGS):
        return
    try:
        pmc = json.load(open(PMC_CONFIGS))
    except Exception:  # noqa: BLE001
        return
    stale = pmc.get("source_id") != _source_id()
    by_cfg = {v.get("config"): v for v in pmc.get("cases", {}).values()}
    for r in records:
        c = by_cfg.get(r.get("config"))
        if not c or "error" in r:
            continue
        b = r.get("batch") or r.get("polys") or 1
        alg = {"main": 4 * r["level"] * r["N"] * 8 if "level" in r else None,                      # a key switch: 2 polys in, 2 out
               "rescale": 2 * (2 * r["level"] - 1) * r["N"] * 8 if "level" in r else None,
               "nntt": 2 * r.get("limbs", 0) * r["N"] * 8, "inntt": 2 * r.get("limbs", 0) * r["N"] * 8}
        roofs = {}
        for op, pc in c.get("per_call", {}).items():
            if op == "galois":
                continue
            k = c["kernels"].get(pc["dominant"], {})
            if "valu_frac" not in k and "hbm_frac" not in k:
                continue
            valu = k.get("bound", "valu") == "valu"
            e = {"kernel": pc["dominant"], "share_of_the_operations_kernel_time": pc["dominant_share"], "bound": "valu-issue" if valu else "hbm",
                 "achieved": k["valu_frac"] * pmc["valu_peak_G_per_s"] if valu else k["hbm_GBs"],
                 "peak": pmc["valu_peak_G_per_s"] if valu else pmc["hbm_peak_GBs"],
                 "unit": "G wave64-VALU-instr/s (1024 SIMDs x 2.4 GHz / 4 clk)" if valu else "GB/s (HBM-side: FETCH_SIZE x2 + WRITE_SIZE)",
                 "frac": k["valu_frac"] if valu else k["hbm_frac"], "valu_frac": k.get("valu_frac"), "hbm_frac": k.get("hbm_frac"),
                 "valu_issue_util_at_profiled_clock": k.get("valu_issue_util"), "profiled_clock_GHz": k.get("clock_GHz"),
                 "wave_cycles": {"waiting": k.get("SQ_WAIT_ANY_share"), "issue_stalled": k.get("SQ_WAIT_INST_ANY_share"), "issuing": k.get("SQ_ACTIVE_INST_ANY_share")},
                 "kernel_us_per_call": pc["kernel_us"]}
            if k.get("byte_calibration"):
                e["byte_calibration"] = k["byte_calibration"]
            if alg.get(op) and "mnist" not in r.get("config", "").lower():
                e["traffic"] = pc["hbm_bytes"] / b                                    # HBM-side bytes per unit (all kernels of the operation)
                e["algorithmic_bytes"] = alg[op]
                e["traffic_over_algorithmic"] = e["traffic"] / alg[op]
                if op == "main" and str(pc["dominant"]).startswith("k_ks_fused"):
                    # what the ratio is made of.  A fused key switch reads, per (ciphertext, working limb) item, the
                    # key words of every digit for that limb (doubles, both components): level x 2 x nw rows of N words per ciphertext.
                    # The 32-128 MiB of key rows of a call are shared by all items and are served by the L2s / the Infinity Cache --
                    # FETCH_SIZE counts those hits like HBM reads.  The rest (digit source rows, sub-block sums, results) streams and is
                    # what the x2 calibration of FETCH_SIZE was made on.
                    nw = r["level"] + (1 if r.get("special_prime", True) else 0)
                    keyb = float(r["level"] * 2 * nw * r["N"] * 8)
                    e["traffic_split"] = {"key_rows_bytes_cache_served_counted": keyb, "row_bytes_streaming_calibrated": max(0.0, e["traffic"] - keyb),
                                          "key_over_algorithmic": keyb / alg[op], "rows_over_algorithmic": max(0.0, e["traffic"] - keyb) / alg[op]}
            roofs["keyswitch" if (op == "main" and "level" in r) else ("pass" if op == "main" else op)] = e
        if roofs:
            r["roofline"] = roofs
            r["roofline_source"] = {"file": "profiles/pmc_configs.json", "tag": pmc.get("tag"), "pmc_source_id": pmc.get("source_id"), "pmc_stale": stale}


def run(scale=1, out=None, only=None):
    """All cases; returns the list of records (each case asserts one ciphertext against the oracle before it is timed).  A case that
    fails is recorded as {"config": ..., "error": ...} and does not stop the others."""
    del RECORDS[:]

    seen = [0]

    def guarded(f, name, *a):
        seen[0] += 1
        if only and str(seen[0]) not in str(only).split(","):   # cases by position, 1-based ("1,3")
            return
        try:
            # every case starts as a fresh process would: the objects of the case before are collected and the recycling allocator's
            # cache goes back to the driver (r06: the blocks a case leaves behind become "stale" 4 096 requests later and are freed
            # then -- hundreds of hipFree calls, each a device drain, inside the NEXT case's timed passes: the reference-shaped MNIST
            # case read 520-680 ms behind the restructured one and 215 ms on its own; TFHE_ALLOC_DEBUG=1 prints the allocator's account)
            import gc
            gc.collect()
            tf.native.check(tf.native.lib().tfhe_alloc_trim())
            f(name, *a)
        except Exception as e:  # noqa: BLE001
            RECORDS.append({"config": name, "error": f"{type(e).__name__}: {e}"})

    N = 1 << 15
    guarded(keyswitch_case, "cfg#3 CKKS N=2^15 10x40-bit + special prime (ckks_rotate.jl path)", N, chain(2**40 + 1, 11, N), max(8, 512 // scale))
    N = 1 << 14
    guarded(keyswitch_case, "cfg#4 N=2^14 6x50-bit + special prime (keyswitch with RNS basis extension)", N, chain(2**50 + 1, 7, N), max(8, 512 // scale))
    N = 1 << 16
    q0, ps = chain(2**60 + 1, 2, N)
    mnist = [q0] + chain(2**40 + 1, 5, N) + [ps]
    guarded(keyswitch_case, "cfg#5 N=2^16 infer.jl ring 60+5x40+60 bit", N, mnist, max(8, 64 // scale))
    guarded(keyswitch_case, "cfg#5' N=2^16 7x50 bit", N, chain(2**50 + 1, 7, N), max(8, 64 // scale))
    guarded(ntt_case, "cfg#5 N=2^16 infer.jl ring", N, mnist, max(8, 128 // scale))
    guarded(ntt_case, "cfg#5' N=2^16 7x50 bit", N, chain(2**50 + 1, 7, N), max(8, 128 // scale))
    N = 1 << 14
    guarded(ntt_case, "N=2^14 60-bit primes", N, chain(2**60 + 1, 8, N), max(8, 1024 // scale))
    guarded(ntt_case, "N=2^14 50-bit primes", N, chain(2**50 + 1, 8, N), max(8, 1024 // scale))
    guarded(mnist_case, "cfg#5 encrypted MNIST inference N=2^16, infer.jl ring (examples/encrypted_mnist.py)", 16, max(1, 16 // scale))
    guarded(mnist_case, "cfg#5 encrypted MNIST inference N=2^16, infer.jl ring, the reference's circuit shape (one Galois key, chained rotations)",
            16, max(1, 16 // scale), True)
    attach_rooflines(RECORDS)
    return list(RECORDS)


if __name__ == "__main__":
    import json
    recs = run(int(sys.argv[1]) if len(sys.argv) > 1 else 1, only=sys.argv[2] if len(sys.argv) > 2 else None)
    for r in recs:
        print(json.dumps(r))
