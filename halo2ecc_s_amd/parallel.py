"""Multi-GPU sharding of independent units (MSM tiles / pairing instances), SURVEY.md §8(e).

Units shard round-robin over ranks with no data-path collective; the only collective is one all_gather of
small per-unit records (status word + result digest) at the end — RCCL over xGMI on GPUs ("nccl" backend),
gloo in the CPU tests.  The reference has no multi-context driver (each unit is its own `Context`,
src/context.rs:136-143), so this file has no reference counterpart."""
import torch
import torch.distributed as dist


def shard_units(n_units, world, rank):
    """unit indices owned by `rank` (round-robin, so ragged counts differ by at most one)"""
    return list(range(rank, n_units, world))


def gather_unit_records(local_units, local_records, n_units, world):
    """all_gather per-unit records.  local_records: int64 tensor [len(local_units), R].
    Returns an int64 tensor [n_units, R] on every rank, rows ordered by unit index."""
    R = local_records.shape[1]
    cap = (n_units + world - 1) // world
    dev = local_records.device
    buf = torch.full((cap, R + 1), -1, dtype=torch.int64, device=dev)
    if len(local_units):
        buf[:len(local_units), 0] = torch.tensor(local_units, dtype=torch.int64, device=dev)
        buf[:len(local_units), 1:] = local_records
    if world > 1:
        parts = [torch.empty_like(buf) for _ in range(world)]
        dist.all_gather(parts, buf)
    else:
        parts = [buf]
    out = torch.zeros((n_units, R), dtype=torch.int64, device=dev)
    seen = torch.zeros((n_units,), dtype=torch.bool, device=dev)
    for p in parts:
        valid = p[:, 0] >= 0
        idx = p[valid, 0]
        out[idx] = p[valid, 1:]
        seen[idx] = True
    assert bool(seen.all()), "some units were not produced by any rank"
    return out
