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


RECORD_WORDS = 29   # status, Offset (3), result point cells (13 words), digests (3 x 4 words)


def unit_records(status, offsets, base, out_refs, digests=None):
    """Per-unit records of one finished step, built on the device:
    [status, base/range/select Offset, result point as its cells' words (x limbs, y limbs: 3 x 2 words each; z: 1 word;
    empty for workloads without a result point), 32-byte digest of each advice array].
    status int32 [units]; offsets int64 [3]; base = batch-interleaved base array [rows][5][2][units][2];
    out_refs = Program.outputs() (x limbs, x native, y limbs, y native, z); digests int64 [3][units][4] or None."""
    units = status.shape[0]
    rec = torch.zeros((units, RECORD_WORDS), dtype=torch.int64, device=status.device)
    rec[:, 0] = status.to(torch.int64)
    rec[:, 1:4] = offsets.to(status.device)
    if out_refs:
        L = (len(out_refs) - 3) // 2
        limb_refs = list(out_refs[0:L]) + list(out_refs[L + 1:2 * L + 1])
        for i, ref in enumerate(limb_refs[:6]):
            rec[:, 4 + 2 * i:6 + 2 * i] = base[ref & 0x3FFFFFF, (ref >> 27) & 7, 0]      # low half: limbs are < 2^128
        z = out_refs[2 * L + 2]
        rec[:, 16] = base[z & 0x3FFFFFF, (z >> 27) & 7, 0, :, 0]
    if digests is not None:
        rec[:, 17:29] = digests.permute(1, 0, 2).reshape(units, 12)
    return rec


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
