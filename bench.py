#!/usr/bin/env python3
"""Headline benchmark: bn256 G1 select-chip MSM witness generation, 2^16 points per GPU = 64 tiles of
1024 points (BASELINE.json configs[1]); each tile replays the reference's own test body
(src/tests/native_scalar_ecc_chip.rs:34-47) in its own row space.

    python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run)

A step = one pass of the hot path over the batch: inputs resident in HBM -> every advice array resident in
HBM, all tiles' status words == 0 (the in-circuit `ecc_assert_equal(msm, expected)` holds).  Multi-GPU:
tiles are independent units, sharded with no data-path collective; one RCCL all_gather of the per-tile
status/result digest at the end of each step.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); measured copy ceiling is 6.29 TB/s
Q = 0x30644E72E131A029B85045B68181585D97816A916871CA8D3C208C16D87CFD47


def cpu_baseline(points):
    """The oracle (CPU restatement = 'port') timed on this box's host cores on a bounded sample: one tile of
    `points` points, window-parallel like the reference's rayon region (src/circuit/ecc_chip.rs:317-343)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib
    from halo2ecc_s_amd import synth
    cores = os.cpu_count() or 1
    inp, _ = synth.msm_bn256_tile_inputs(points, cheap_points=True, with_expected=False)
    run = oracle_lib.run_msm_bn256_tile(points, inp, threads=cores)
    secs, cells = run.info.seconds, run.info.n_advice_cells
    run.close()
    return {"value": cells / secs, "unit": "cells/s", "cores": cores, "kind": "port",
            "sample": f"one {points}-point bn256 MSM tile (test body incl. assign_point), {cells} advice cells, "
                      f"{secs:.1f} s, oracle C++ restatement, {cores} threads over MSM windows",
            "points_per_s": points / secs}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--tiles", type=int, default=64, help="tiles per GPU (64 x 1024 = 2^16 points)")
    ap.add_argument("--points", type=int, default=1024, help="points per tile")
    ap.add_argument("--cpu-sample-points", type=int, default=1024)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-check", action="store_true", help="A/B experiments with deliberately broken arithmetic")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL) on GPUs; gloo only to exercise the N>1 path on one GPU")
    ap.add_argument("--device", type=int, default=None, help="override the CUDA device index (default: LOCAL_RANK)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from halo2ecc_s_amd import Engine, Program, synth

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.device is not None:
        local_rank = args.device
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(f"cuda:{local_rank}"))
        else:
            dist.init_process_group(args.dist_backend, rank=rank, world_size=world)
    torch.cuda.set_device(local_rank)
    dev = f"cuda:{local_rank}"
    coll_dev = dev if args.dist_backend == "nccl" else "cpu"

    n, tiles = args.points, args.tiles
    eng = Engine(local_rank)
    prog = Program.msm_bn256_tile(n, emit_shape=False)
    shape_prog = Program.msm_bn256_tile(n, emit_shape=True)  # shape-only artefacts, once per shape (not timed)
    cells_per_tile = shape_prog.n_advice_cells
    launches = shape_prog.launches()
    shape_prog.close()

    # synthetic inputs, different per tile and per rank; two batches that the steps alternate between, so that a step
    # can never pass on data a previous step left behind (same inputs every step would hide a missing dependency)
    n_batches = 2
    base, rng, sel, status = eng.alloc(prog, tiles)
    out_refs = prog.outputs()
    L = 3
    batches = []
    for bi in range(n_batches):
        ins = np.stack([synth.msm_bn256_tile_inputs(n, tile=(bi * world + rank) * tiles + t, cheap_points=True, with_expected=False)[0]
                        for t in range(tiles)])
        d_in = eng.upload_inputs(prog, ins)
        # pass 0: learn each tile's MSM result, then feed it back as the `expected` input so that the in-circuit
        # ecc_assert_equal holds in every timed pass (the reference test computes it with the native library)
        eng.run(prog, d_in, base, rng, sel, status)
        torch.cuda.synchronize()
        exp = np.zeros((tiles, 3, 4), dtype=np.uint64)
        for t in range(tiles):
            xs = [eng.read_cell(base, r, t) for r in out_refs[0:L]]
            ys = [eng.read_cell(base, r, t) for r in out_refs[L + 1:2 * L + 1]]
            z = eng.read_cell(base, out_refs[2 * L + 2], t)
            x = sum(v << (108 * i) for i, v in enumerate(xs)) % Q
            y = sum(v << (108 * i) for i, v in enumerate(ys)) % Q
            if z:
                x = y = 0
            exp[t] = synth.pack([x, y, z], 4)
        d_in[:, 4 * n + 6:4 * n + 9, :] = torch.from_numpy(exp.view(np.int64)).to(dev)
        batches.append(d_in)
    step_no = [0]
    status_any = torch.zeros_like(status)   # OR of every step's status words

    def step():
        status.zero_()
        eng.run(prog, batches[step_no[0] % n_batches], base, rng, sel, status)
        status_any.bitwise_or_(status)
        step_no[0] += 1
        if world > 1:  # final gather of per-tile status words (the only collective on the path)
            st = status.to(coll_dev)
            gathered = [torch.empty_like(st) for _ in range(world)]
            dist.all_gather(gathered, st)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if args.warmup > 0 and not args.no_check:
        assert int(status_any.abs().max()) == 0, f"tile status {status_any.cpu().numpy()}"

    eng.set_profiling(True)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    launch_ms, step_ms = [], []
    for _ in range(args.steps):
        ts = time.perf_counter()
        step()
        torch.cuda.current_stream().synchronize()
        launch_ms.append(eng.last_run_launch_ms())
        step_ms.append(1e3 * (time.perf_counter() - ts))
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if not args.no_check:
        assert int(status_any.abs().max()) == 0, f"tile status {status_any.cpu().numpy()}"
    if world > 1:
        tmax = torch.tensor([elapsed], device=coll_dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    total_cells = cells_per_tile * tiles * world * args.steps
    total_points = n * tiles * world * args.steps
    # dominant kernel = the launch with the most cells (the MSM window strands)
    dom = max(range(len(launches)), key=lambda i: launches[i]["cells"])
    # a big expansion goes out as two back-to-back kernel launches over a prefix / the rest of its sub-ranges (h2e.h):
    # the events bracket both, so the per-launch figures are bracket / n and bytes / n
    dom_n = eng.last_run_expansion_launches()[dom]
    dom_ms = float(np.mean([ms[dom][1] for ms in launch_ms if len(ms) > dom])) / dom_n
    dom_bytes = 32.0 * launches[dom]["cells"] * tiles / dom_n
    achieved = dom_bytes / (dom_ms * 1e-3) / 1e9
    out = {
        "metric": "witness_cells_per_sec",
        "value": total_cells / elapsed,
        "unit": "cells/s",
        "msm_points_per_sec": total_points / elapsed,
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps,
        "step_ms": [round(x, 2) for x in step_ms],   # rank 0's wall time of each timed step (diagnostic)
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u64",
        "data": "synthetic",
        "config": {"workload": f"bn256 G1 select-chip MSM witness, {tiles} tiles x {n} points per GPU "
                               f"(2^{int(np.log2(max(1, tiles * n)))} points/GPU), reference test body per tile",
                   "tiles_per_gpu": tiles, "points_per_tile": n, "cells_per_tile": cells_per_tile,
                   "sharding": f"tiles round-robin over {world} GPU(s), all_gather of status words"},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                     "kernel": "h2e_run_tape<FP_BN256_FQ, false> (full expansion of the MSM window strands)",
                     "launch_ms": dom_ms, "algorithmic_bytes_per_launch": dom_bytes, "launches_per_step": dom_n,
                     "value_chain_ms": [float(x) for x in np.mean(np.array(launch_ms)[:, :, 0], axis=0)],
                     "expansion_ms": [float(x) for x in np.mean(np.array(launch_ms)[:, :, 1], axis=0)]},
    }
    # HBM traffic of the dominant kernel: PMC counters cannot be collected inside this process; the figure is the
    # committed rocprofv3 --pmc measurement of the same dispatch on the same workload (separate WRITE_SIZE /
    # FETCH_SIZE passes, KB -> bytes, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950).
    pmc = os.path.join(ROOT, "profiles", "hbm_pmc_latest.json")
    if tiles == 64 and n == 1024 and os.path.exists(pmc):
        with open(pmc) as f:
            m = json.load(f)
        # the file holds the sums over the window expansion's dispatches of one step; per launch like `achieved`
        out["roofline"]["traffic"] = 1e3 * (m["WRITE_SIZE_raw"] + 2.0 * m["FETCH_SIZE_raw"]) / dom_n
        out["roofline"]["traffic_source"] = "profiles/" + m.get("file", "hbm_pmc_latest.json")
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args.cpu_sample_points)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
