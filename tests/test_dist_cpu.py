"""World-size-2 gloo test of the multi-GPU plumbing (toyfhe.jl_amd/dist.py): batch sharding with no
data-path collective, barrier, max-over-ranks timing and the optional final gather.  Each rank runs the
oracle on its shard (the GPU ranks run the HIP engine on theirs); the union must equal the unsharded run."""
import os
import socket
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys, json
sys.path.insert(0, %(root)r)
import numpy as np, torch
import toyfhe_jl_amd as tf
from toyfhe_jl_amd import dist as tdist
from oracle import ref_cpu
from tests import helpers as H
world, rank, _ = tdist.init(backend="gloo")
assert world == 2
N, qs, G = 64, H.chain(40, 2, 64), 7          # ragged global batch
rng = np.random.default_rng(0)
a = H.rand_residues(rng, qs, (G,), N)         # every rank draws the same global batch, owns one shard
start, count = tdist.shard(G, rank, world)
ctx = ref_cpu.RefCtx(N, qs)
local = ctx.nntt(a[start:start + count])
tdist.barrier()
tmax = tdist.max_over_ranks(1.0 + rank)
parts = tdist.gather_results(torch.from_numpy(local.astype(np.int64)))
if rank == 0:
    full = np.concatenate([p.numpy().astype(np.uint64) for p in parts])
    ok = bool(np.array_equal(full, ctx.nntt(a)))
    print(json.dumps({"ok": ok, "tmax": tmax, "counts": [int(p.shape[0]) for p in parts]}))
torch.distributed.destroy_process_group()
'''


def test_shard_partition():
    import toyfhe_jl_amd as tf
    from toyfhe_jl_amd import dist as tdist
    for G in (0, 1, 7, 8, 1024, 4096):
        for world in (1, 2, 3, 8):
            cover = []
            for r in range(world):
                s, c = tdist.shard(G, r, world)
                cover += list(range(s, s + c))
            assert cover == list(range(G))
    assert tdist.shard(4096, 3, 8) == (1536, 512)      # BASELINE.json configs[3]: 4096 over 8 GPUs


def test_two_rank_gloo():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    script = os.path.join(ROOT, "gpurun_out", "_dist_worker.py")
    os.makedirs(os.path.dirname(script), exist_ok=True)
    open(script, "w").write(WORKER % {"root": ROOT})
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), script]
    env = dict(os.environ, OMP_NUM_THREADS="1")
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    import json
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    res = json.loads(line)
    assert res["ok"] and res["tmax"] == 2.0 and res["counts"] == [4, 3]
