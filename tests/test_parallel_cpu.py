"""world_size-2 gloo test of the N>1 path, through the same two functions bench.py's step uses
(parallel.unit_records + parallel.gather_unit_records): units shard round-robin, each rank produces its units' arrays
(here the oracle stands in for the GPU worker and its rows are laid out batch-interleaved like the engine's), builds the
per-unit records {status, Offset, result point cells, digests}, one all_gather, every rank sees all units."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, n_units, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle_lib
    from halo2ecc_s_amd import synth
    from halo2ecc_s_amd.parallel import gather_unit_records, shard_units
    from halo2ecc_s_amd.parallel import unit_records, RECORD_WORDS
    from halo2ecc_s_amd import Program
    units = shard_units(n_units, world, rank)
    n = 2
    prog = Program.msm_bn256_tile(n)
    runs = []
    for u in units:
        inp, _ = synth.msm_bn256_tile_inputs(n, tile=u)
        runs.append(oracle_lib.run_msm_bn256_tile(n, inp))
    # this rank's base array, batch-interleaved [rows][5][half][unit][2] like the engine's output
    rows = np.stack([r.adv(0, prog.base_rows)[0] for r in runs], axis=0)                  # [unit][row][5][4]
    base = torch.from_numpy(np.ascontiguousarray(rows.reshape(len(units), prog.base_rows, 5, 2, 2).transpose(1, 2, 3, 0, 4)).view(np.int64))
    status = torch.tensor([r.info.status for r in runs], dtype=torch.int32)
    offsets = torch.tensor([prog.base_offset, prog.range_offset, prog.select_offset], dtype=torch.int64)
    dig = torch.from_numpy(np.stack([[r.digest(region) for r in runs] for region in range(3)]).view(np.int64))   # [3][unit][4]
    local = unit_records(status, offsets, base, prog.outputs(), dig)
    assert local.shape == (len(units), RECORD_WORDS)
    # the record's result point is the MSM result the oracle asserted equal to the expected input
    for i, r in enumerate(runs):
        x = sum((int(local[i, 4 + 2 * k]) & (2**64 - 1) | (int(local[i, 5 + 2 * k]) & (2**64 - 1)) << 64) << (108 * k) for k in range(3))
        assert x % synth.BN_Q == sum(int(w) << (64 * j) for j, w in enumerate(synth.msm_bn256_tile_inputs(n, tile=units[i])[0][4 * n + 6]))
    allrec = gather_unit_records(units, local, n_units, world)
    for r in runs:
        r.close()
    q.put((rank, units, allrec.numpy().tolist()))
    dist.barrier()
    dist.destroy_process_group()


def test_shard_units_ragged():
    from halo2ecc_s_amd.parallel import shard_units
    for n_units in (0, 1, 5, 16, 64):
        for world in (1, 2, 3, 8):
            got = sorted(sum((shard_units(n_units, world, r) for r in range(world)), []))
            assert got == list(range(n_units))
            sizes = [len(shard_units(n_units, world, r)) for r in range(world)]
            assert max(sizes) - min(sizes) <= 1


def test_gather_world2_gloo(oracle):
    world, n_units = 2, 5  # ragged: 3 + 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 1000)
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_units, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    by_rank = {r: (u, rec) for r, u, rec in results}
    assert by_rank[0][0] == [0, 2, 4] and by_rank[1][0] == [1, 3]
    assert by_rank[0][1] == by_rank[1][1]  # every rank sees the same gathered table
    table = np.array(by_rank[0][1])
    assert table.shape == (n_units, 29)
    assert (table[:, 0] == 0).all()          # every unit OK
    assert len(set(table[:, 1])) == 1        # same shape -> same offsets
    assert len(set(map(tuple, table[:, 17:29]))) == n_units   # different inputs -> different digests


def _plan_worker(rank, world, port, steps, total, q):
    """a job of `steps` steps whose every step deals `total` units round-robin over the ranks (bench.py --scaling strong): the gather
    plan's capacity is steps x ceil(total / world), not ceil(steps x total / world)"""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from halo2ecc_s_amd.parallel import GatherPlan, gather_records
    mine = [k * total + rank + t * world for k in range(steps) for t in range(len(range(rank, total, world)))]
    plan = GatherPlan(mine, steps * total, world, "cpu", cap=steps * ((total + world - 1) // world))
    local = torch.tensor([[u, 1000 + u] for u in mine], dtype=torch.int64)
    table, seen = gather_records(plan, local)
    q.put((rank, mine, table.tolist(), bool(seen.all())))
    dist.barrier()
    dist.destroy_process_group()


def test_gather_per_step_ragged_shares_world2_gloo():
    world, steps, total = 2, 3, 5   # every step: 3 + 2 units; rank 0 holds 9 of the job's 15 units (more than ceil(15 / 2))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + ((os.getpid() + 17) % 1000)
    procs = [ctx.Process(target=_plan_worker, args=(r, world, port, steps, total, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, mine, table, complete in results:
        assert complete
        assert table == [[u, 1000 + u] for u in range(steps * total)]
    assert sorted(sum((m for _, m, _, _ in results), [])) == list(range(steps * total))
