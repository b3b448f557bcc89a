"""Multi-GPU sharding of independent units (MSM tiles / pairing instances), SURVEY.md §8(e).

Units shard round-robin over ranks with no data-path collective; the only collective is ONE all_gather of
small per-unit records (status word + Offset + result point + digests) at the end of a job — RCCL over xGMI on
GPUs ("nccl" backend), gloo in the CPU tests.  Nothing here synchronises with the host: records are built on the
device per finished step (`unit_records`, into a preallocated job table), the gather plan (unit indices) is uploaded
once before the job (`GatherPlan`), and the completeness check is a device tensor the caller reads after its timed
region.  The reference has no multi-context driver (each unit is its own `Context`, src/context.rs:136-143), so this
file has no reference counterpart."""
import torch
import torch.distributed as dist


def shard_units(n_units, world, rank):
    """unit indices owned by `rank` (round-robin, so ragged counts differ by at most one)"""
    return list(range(rank, n_units, world))


def record_words(limbs=3):
    """status, Offset (3), result point cells (x limbs, y limbs: 2 words each; z: 1 word), digests (3 x 4 words)"""
    return 1 + 3 + 2 * (2 * limbs) + 1 + 12


RECORD_WORDS = record_words(3)   # 29: the bn256 workloads


def unit_records(status, offsets, base, out_refs, digests=None, out=None, limbs=3):
    """Per-unit records of one finished step, built on the device:
    [status, base/range/select Offset, result point as its cells' words (x limbs, y limbs: `limbs` x 2 words each; z: 1
    word; zero for workloads without a result point), 32-byte digest of each advice array].
    status int32 [units]; offsets int64 [3] (device); base = batch-interleaved base array [rows][5][2][units][2];
    out_refs = Program.outputs() (x limbs, x native, y limbs, y native, z) or []; digests int64 [3][units][4] or None;
    out = int64 [units][record_words(limbs)] to fill (a row block of the job's record table) or None."""
    units = status.shape[0]
    R = record_words(limbs)
    rec = out if out is not None else torch.zeros((units, R), dtype=torch.int64, device=status.device)
    assert rec.shape == (units, R)
    rec[:, 0] = status
    rec[:, 1:4] = offsets
    if out_refs:
        L = (len(out_refs) - 3) // 2
        assert L == limbs, f"result point has {L}-limb coordinates, the record layout was sized for {limbs}"
        limb_refs = list(out_refs[0:L]) + list(out_refs[L + 1:2 * L + 1])
        for i, ref in enumerate(limb_refs):
            rec[:, 4 + 2 * i:6 + 2 * i] = base[ref & 0x3FFFFFF, (ref >> 27) & 7, 0]      # low half: limbs are < 2^128
        z = out_refs[2 * L + 2]
        rec[:, 4 + 4 * L] = base[z & 0x3FFFFFF, (z >> 27) & 7, 0, :, 0]
    if digests is not None:
        rec[:, R - 12:R] = digests.permute(1, 0, 2).reshape(units, 12)
    return rec


class GatherPlan:
    """Who owns which unit, uploaded once: `idx` = this rank's global unit indices padded with -1 to the per-rank
    capacity (ragged shards), on the device the collective runs on."""

    def __init__(self, local_units, n_units, world, device, cap=None):
        """cap: rows every rank contributes (the largest share; default: ceil(n_units / world), right when the job's units are
        dealt round-robin as a whole - a job of several steps whose every step is dealt round-robin needs steps x ceil(step / world))"""
        self.n_units, self.world = n_units, world
        self.cap = cap if cap is not None else (n_units + world - 1) // world
        assert len(local_units) <= self.cap
        idx = torch.full((self.cap,), -1, dtype=torch.int64)
        if len(local_units):
            idx[:len(local_units)] = torch.tensor(local_units, dtype=torch.int64)
        self.idx = idx.to(device)
        self.n_local = len(local_units)


def job_table(plan, record_words_, device=None):
    """The table a job's steps fill and its one collective sends as it is: int64 [plan.cap][1 + R], column 0 = the global unit
    index of the row (-1: padding of a ragged share), the rest written per finished step by the engine's h2e_unit_records kernel
    (Engine.unit_records(..., out=table[rows], col0=1)) - no indexing kernels on the host side of the boundary."""
    t = torch.zeros((plan.cap, 1 + record_words_), dtype=torch.int64, device=device if device is not None else plan.idx.device)
    t[:, 0] = plan.idx.to(t.device)
    return t


def gather_records(plan, local_records):
    """ONE all_gather of the job's per-unit records.  local_records: int64 [plan.n_local][R] on plan.idx's device.
    Returns (table int64 [n_units][R] with rows ordered by unit index, seen bool [n_units]) - both on the device and
    without a host synchronisation; the caller checks `seen.all()` when it is allowed to wait."""
    R = local_records.shape[1]
    dev = plan.idx.device
    buf = torch.zeros((plan.cap, R + 1), dtype=torch.int64, device=dev)
    buf[:, 0] = plan.idx
    buf[:plan.n_local, 1:] = local_records
    return gather_table(plan, buf)


def gather_table(plan, buf):
    """The collective itself: `buf` = a job_table (index column + records, on plan.idx's device) -> all_gather -> rows ordered
    by unit index.  A host in another language issues the same ONE ncclAllGather on the same table."""
    R = buf.shape[1] - 1
    dev = plan.idx.device
    assert buf.shape[0] == plan.cap and buf.device == dev
    # (a one-rank process group still runs the collective: on a GPU that is RCCL's communicator and all_gather kernel, the exact
    # code an N-rank job executes - tests/test_bench_gpu.py::test_bench_one_rank_rccl)
    if plan.world > 1 or (dist.is_available() and dist.is_initialized()):
        lst = [torch.empty_like(buf) for _ in range(plan.world)]
        dist.all_gather(lst, buf)
        parts = torch.cat(lst)
    else:
        parts = buf
    # padding rows (index -1) are scattered into one extra row that is dropped: no data-dependent shapes, no host sync
    idx = torch.where(parts[:, 0] >= 0, parts[:, 0], torch.full_like(parts[:, 0], plan.n_units))
    out = torch.zeros((plan.n_units + 1, R), dtype=torch.int64, device=dev)
    out[idx] = parts[:, 1:]
    seen = torch.zeros((plan.n_units + 1,), dtype=torch.bool, device=dev)
    seen[idx] = True
    return out[:plan.n_units], seen[:plan.n_units]


def gather_unit_records(local_units, local_records, n_units, world):
    """Convenience form (tests, small jobs): plan + gather + completeness check on the host."""
    plan = GatherPlan(local_units, n_units, world, local_records.device)
    out, seen = gather_records(plan, local_records)
    assert bool(seen.all()), "some units were not produced by any rank"
    return out
