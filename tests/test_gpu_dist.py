"""Multi-rank runs of the ENGINE: one process per rank, each running tfhe_bfv_mul_relin on its contiguous
shard of a global batch, no data-path collective, results gathered at the end and compared with the single-rank run of
the whole batch and with the oracle.  With >= 2 visible GPUs the ranks take one GPU each and talk RCCL; on a 1-GPU box
both ranks share GPU 0 and rendezvous over gloo (same engine code path per rank, collective on host tensors)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys, json
sys.path.insert(0, %(root)r)
import numpy as np, torch
import toyfhe_jl_amd as tf
from toyfhe_jl_amd import dist as tdist
world, rank, local_rank = tdist.env_world()
ndev = torch.cuda.device_count()
nccl = ndev >= world
dev_index = local_rank %% ndev
torch.cuda.set_device(dev_index)
dev = torch.device("cuda", dev_index)
tdist.init(backend="nccl" if nccl else "gloo", **({"device_id": dev} if nccl else {}))
tf.native.check(tf.native.lib().tfhe_set_device(dev_index))
N, ns, t, G = 4096, 2, 65537, 7                  # ragged global batch: shards of 4 and 3
p, ch = tf.nextprime(2**50 + 1, 1, 2 * N), []
for _ in range(5):
    ch.append(p); p = tf.nextprime(p + 2 * N, 1, 2 * N)
qs = ch[:ns]
rng = np.random.default_rng(0)                   # every rank draws the same global batch and key, owns one shard
res = lambda pre: np.stack([rng.integers(0, q, size=tuple(pre) + (N,), dtype=np.uint64) for q in qs], axis=len(pre))
c1, c2, evk = res((G, 2)), res((G, 2)), res((ns, 2))
ctx = tf.Context(N, ch)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
plan = tf.BfvPlan(ctx, ctx, t, idx_s=list(range(ns)))
def run(a, b):
    n = a.shape[0]
    da, db, dk = (torch.from_numpy(x.astype(np.int64)).to(dev) for x in (a, b, evk))
    out = torch.empty((n, 2, ns, N), dtype=torch.int64, device=dev)
    plan.mul_relin(dk.data_ptr(), ns, da.data_ptr(), db.data_ptr(), out.data_ptr(), n)
    torch.cuda.synchronize()
    return out
start, count = tdist.shard(G, rank, world)
local = run(c1[start:start + count], c2[start:start + count])
tdist.barrier()
parts = tdist.gather_results(local if nccl else local.cpu())
cabi = None
if nccl:                                            # tfhe_gather over RCCL on device memory, shards padded to the largest
    comm = tdist.make_comm()
    mx = -(-G // world)
    pad = torch.zeros((mx, 2, ns, N), dtype=torch.int64, device=dev)
    pad[:count] = local
    full_d = torch.empty((world, mx, 2, ns, N), dtype=torch.int64, device=dev)
    comm.gather(ctx, pad.data_ptr(), full_d.data_ptr(), pad.numel())
    ctx.sync()
    cabi = torch.cat([full_d[r, :tdist.shard(G, r, world)[1]] for r in range(world)]).cpu().numpy().astype(np.uint64)
if rank == 0:
    full = torch.cat([x.cpu() for x in parts]).numpy().astype(np.uint64)
    alone = run(c1, c2).cpu().numpy().astype(np.uint64)
    from oracle import ref_cpu                   # checker
    rs, rb = ref_cpu.RefCtx(N, qs), ref_cpu.RefCtx(N, ch)
    pick = [0, 3, 4, 6]                          # both sides of the shard boundary
    want = rs.keyswitch(ns, False, evk, ref_cpu.bfv_mul(rs, rb, t, c1[pick], c2[pick]))
    print(json.dumps({"same_as_single_rank": bool(np.array_equal(full, alone)), "oracle": bool(np.array_equal(full[pick], want)),
                      "cabi_gather": None if cabi is None else bool(np.array_equal(cabi, alone)),
                      "counts": [int(x.shape[0]) for x in parts], "backend": "nccl" if nccl else "gloo", "world": world}))
torch.distributed.destroy_process_group()
'''


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_two_ranks_run_the_engine_on_their_shards():
    script = os.path.join(ROOT, "gpurun_out", "_dist_gpu_worker.py")
    os.makedirs(os.path.dirname(script), exist_ok=True)
    open(script, "w").write(WORKER % {"root": ROOT})
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), script]
    env = dict(os.environ, OMP_NUM_THREADS="4", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    res = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert res["same_as_single_rank"] and res["oracle"] and res["counts"] == [4, 3] and res["world"] == 2
    import torch
    if torch.cuda.device_count() >= 2:                # two physical devices: the C-ABI communicator MUST come up and agree
        assert res["backend"] == "nccl" and res["cabi_gather"] is True, res
    else:
        assert res["cabi_gather"] is None            # 1-GPU box (RCCL refuses two ranks on one device): gloo, no C-ABI gather


def test_cabi_gather_world_of_one():
    """tfhe_comm_id / tfhe_comm_create / tfhe_gather with a single rank: RCCL is found and bound at run time, the communicator
    comes up on this GPU and the all-gather (a copy for one rank) lands on the context's stream."""
    import numpy as np

    import toyfhe_jl_amd as tf
    from toyfhe_jl_amd import dist as tdist
    ctx = tf.Context(64, [tf.nextprime(2**40 + 1, 1, 128)])
    comm = tdist.make_comm()
    assert comm.nranks == 1 and comm.rank == 0
    a = np.arange(1000, dtype=np.uint64)
    src, dst = tf.DeviceBuffer.from_numpy(a), tf.DeviceBuffer(1000)
    comm.gather(ctx, src.ptr, dst.ptr, 1000)
    ctx.sync()
    assert np.array_equal(dst.to_numpy(), a)
    comm.close()


def _line_and_record(out):
    """bench.py prints ONE compact line on stdout (the driver parses it; < 4 KB) and the full record on stderr + profiles/bench_full.json"""
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and len(lines[-1]) < 4096, [len(l) for l in lines]
    line = json.loads(lines[-1])
    full = json.loads([l for l in out.stderr.splitlines() if l.startswith('{"metric"')][-1])
    assert json.load(open(os.path.join(ROOT, line["full_record"]))) == full
    return line, full


def test_bench_spawns_the_ranks_it_is_asked_for():
    """`python bench.py --gpus 2` outside torchrun launches two ranks itself and reports n_gpus = 2 with the whole-job
    rate (one GPU each over RCCL when two are visible; otherwise both on GPU 0 over gloo as a functional check)."""
    import torch
    backend = "nccl" if torch.cuda.device_count() >= 2 else "gloo"
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "64",
           "--no-cpu", "--no-ntt", "--backend", backend, "--total", "96"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=1500, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    line, res = _line_and_record(out)
    assert line["n_gpus"] == 2 and line["nranks_seen"] == 2 and line["config"]["global_batch"] == 128 and line["value"] > 0
    assert line["scaling"] == "weak" and line["gather"]["value_with_gather"] <= line["value"] * 1.0001 and "roofline" in line
    assert {m["name"][:5] for m in line["configs_multi"]} == {"cfg#4", "cfg#5"} and all(m["nranks_seen"] == 2 and m["value"] > 0 for m in line["configs_multi"])
    assert res["n_gpus"] == 2 and res["config"]["global_batch"] == 128 and res["value"] > 0
    assert res["scaling"] == "weak" and "gather" in res and res["gather"]["value_with_gather"] <= res["value"] * 1.0001
    # the configurations BASELINE.json defines across the GPUs of a node ride in the same line (here scaled down by --total)
    multi = {m["config"][:5]: m for m in res["configs_multi"]}
    assert set(multi) == {"cfg#4", "cfg#5"} and not any("error" in m for m in multi.values()), multi
    c4 = multi["cfg#4"]
    assert c4["global_units"] == 96 and [r["units"] for r in c4["per_rank"]] == [48, 48] and c4["nranks_seen"] == 2
    assert c4["value"] > 0 and c4["imbalance_max_over_min"] >= 1.0 and c4["gather"] and "error" not in c4["gather"]
    c5 = multi["cfg#5"]
    assert [r["sets"] for r in c5["per_rank"]] == [1, 1] and c5["value"] > 0 and c5["global_units"] == 1024


@pytest.mark.parametrize("cfg,total,units", [("cfg4", 67, [34, 33]), ("cfg3", 8, [8, 8]), ("cfg5", 1024, [1, 1])])
def test_bench_multi_rank_modes_of_the_other_configurations(cfg, total, units):
    """`bench.py --gpus 2 --config cfg4|cfg3|cfg5`: the configuration is THE line of the run -- every rank on its shard (ragged for
    cfg#4), each rank's shard checked against the oracle before it is timed, per-rank rates, imbalance, the final gather timed
    separately.  RCCL with two GPUs; both ranks on GPU 0 over gloo on a 1-GPU box (functional check)."""
    import torch
    backend = "nccl" if torch.cuda.device_count() >= 2 else "gloo"
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--config", cfg, "--steps", "2", "--warmup", "1",
           "--total", str(total), "--backend", backend]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=1500, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    line, res = _line_and_record(out)
    assert "error" not in res, res
    assert line["n_gpus"] == 2 and line["nranks_seen"] == 2 and line["value"] > 0 and line["roofline"]["bound"] == "hbm" and line["scaling"] == res["scaling"]
    assert (line["roofline"]["frac"] is not None) == (cfg != "cfg5")
    assert res["n_gpus"] == 2 and res["nranks_seen"] == 2 and res["value"] > 0 and res["steps"] == 2 and res["warmup"] == 1
    key = "sets" if cfg == "cfg5" else "units"
    assert [r[key] for r in res["per_rank"]] == units
    assert res["scaling"] == ("weak" if cfg == "cfg3" else "strong")
    if cfg != "cfg5":
        assert res["gather"] and "error" not in res["gather"] and res["gather"]["value_with_gather"] <= res["value"] * 1.0001


@pytest.mark.parametrize("mode,extra,units", [
    ("bfv", ["--batch", "8", "--no-cpu", "--no-ntt", "--no-configs"], [8] * 8),
    ("cfg4", ["--total", "4096"], [512] * 8),                                  # BASELINE configs[3] as it is defined: 4096 -> 8 x 512
    ("cfg5", ["--total", "10000"], [3, 3, 3, 3, 2, 2, 2, 2]),                  # configs[4]: 20 ciphertext sets of 512 images
])
def test_eight_rank_rehearsal(mode, extra, units):
    """the world = 8 code paths (sharding, per-rank records, agreement on errors, the compact line of an 8-rank
    record, the gather leg) run BEFORE they meet eight physical GPUs.  Eight GPUs visible: one rank each over RCCL; fewer: all ranks on
    the devices there are over gloo (functional: the rates mean nothing).  Every rank checks a ciphertext of its own shard against the
    oracle before the timed region (bench.py / bench_configs._multi_setup) -- the union of the shards is the global batch by
    construction of dist.shard (tests/test_dist_cpu.py)."""
    import torch
    backend = "nccl" if torch.cuda.device_count() >= 8 else "gloo"
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(OMP_NUM_THREADS="2", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1", "--backend", backend] + extra
    if mode != "bfv":
        cmd += ["--config", mode]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=2400, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    line, res = _line_and_record(out)                                          # one stdout line, < 4 KB, equal to the full record on disk
    assert "error" not in res and not res.get("errors"), res
    assert line["n_gpus"] == 8 and line["nranks_seen"] == 8 and line["value"] > 0 and res["nranks_seen"] == 8
    if mode == "bfv":
        assert line["config"]["global_batch"] == 64 and line["scaling"] == "weak"
        assert res["gather"] and "error" not in res["gather"]
        return
    key = "sets" if mode == "cfg5" else "units"
    assert [r[key] for r in res["per_rank"]] == units and [r["rank"] for r in res["per_rank"]] == list(range(8))
    assert res["scaling"] == "strong" and res["global_units"] == (4096 if mode == "cfg4" else 20 * 512)
    assert res["imbalance_max_over_min"] >= 1.0
    if mode == "cfg4":
        assert res["gather"] and "error" not in res["gather"]
