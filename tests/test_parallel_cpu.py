"""world_size-2 gloo test of the N>1 path: units shard round-robin, each rank produces its units (here with
the oracle standing in for the GPU worker), one all_gather of per-unit records, every rank sees all units."""
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
    units = shard_units(n_units, world, rank)
    recs = []
    for u in units:
        inp, _ = synth.msm_bn256_tile_inputs(2, tile=u)
        run = oracle_lib.run_msm_bn256_tile(2, inp)
        recs.append([run.info.status, run.info.base_offset, run.info.n_advice_cells])
        run.close()
    local = torch.tensor(recs, dtype=torch.int64).reshape(len(units), 3)
    allrec = gather_unit_records(units, local, n_units, world)
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
    assert table.shape == (n_units, 3)
    assert (table[:, 0] == 0).all()          # every unit OK
    assert len(set(table[:, 1])) == 1        # same shape -> same offsets
