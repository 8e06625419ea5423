"""One-process-per-GPU plumbing for batches of independent ciphertexts.

The path shards by ciphertext: no collective on the data path.  torch.distributed (RCCL on the GPUs,
gloo in the CPU tests) is used only for rendezvous, barriers, max-over-ranks timing and an optional
final gather of per-rank results."""
from __future__ import annotations

import os


def env_world():
    return int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))


def shard(global_batch: int, rank: int, world: int):
    """Contiguous partition of `global_batch` ciphertexts: returns (start, count); the first
    `global_batch % world` ranks take one extra (ragged batches are allowed)."""
    base, extra = divmod(global_batch, world)
    count = base + (1 if rank < extra else 0)
    start = rank * base + min(rank, extra)
    return start, count


def init(backend: str | None = None, device_id=None):
    """init_process_group from the torchrun environment (no-op for world size 1). Returns (world, rank, local_rank)."""
    world, rank, local_rank = env_world()
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if not dist.is_initialized():
            kw = {"device_id": device_id} if device_id is not None else {}
            dist.init_process_group(backend=backend or "nccl", rank=rank, world_size=world, **kw)
    return world, rank, local_rank


def world_size_seen() -> int:
    """ranks the process group actually has (1 without one): lets a bench line answer "did RCCL see N ranks"."""
    import torch.distributed as dist
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def barrier():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        dist.barrier()


def max_over_ranks(value: float, device=None) -> float:
    """all-reduce MAX of a host scalar (the bench's step time)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_results(local, dst: int = 0):
    """Optional final gather of per-rank result tensors (ragged first dimension allowed) to `dst`."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return [local]
    world, rank = dist.get_world_size(), dist.get_rank()
    n = torch.tensor([local.shape[0]], dtype=torch.int64, device=local.device)
    counts = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(counts, n)
    mx = int(max(int(c.item()) for c in counts))
    pad = torch.zeros((mx,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    bufs = [torch.zeros_like(pad) for _ in range(world)] if rank == dst else None
    dist.gather(pad, bufs, dst=dst)
    if rank != dst:
        return None
    return [b[: int(c.item())] for b, c in zip(bufs, counts)]


def make_comm():
    """tfhe_comm over this job's ranks (the C-ABI gather boundary, include/toyfhe_hip.h): rank 0's RCCL rendezvous id is
    broadcast with torch.distributed (any backend), then every rank joins.  World size 1 works without a process group."""
    import torch
    import torch.distributed as dist

    from . import native
    world, rank, _ = env_world()
    if not (dist.is_available() and dist.is_initialized()):
        world, rank = 1, 0

    def exchange(data):
        if world == 1:
            return data
        on_gpu = dist.get_backend() == "nccl"
        t = torch.zeros(128, dtype=torch.uint8, device="cuda" if on_gpu else "cpu")
        if rank == 0:
            t.copy_(torch.frombuffer(bytearray(data), dtype=torch.uint8))
        dist.broadcast(t, src=0)
        return bytes(t.cpu().numpy().tobytes())
    return native.Comm(world, rank, exchange)
