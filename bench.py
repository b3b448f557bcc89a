#!/usr/bin/env python3
"""Headline benchmark of the witness engine (BASELINE.json): one *step* = one pass of the hot path over one batch
of synthetic units, inputs resident in HBM -> every advice array resident in HBM, every unit's status word == 0.

    python bench.py --gpus N --steps K --warmup W [--workload msm|pairing_bn256|pairing_bls12_381]
                                                   (N > 1: launched by torch.distributed.run, one rank per GPU)

Workloads (units are independent: sharded over ranks with no data-path collective, one RCCL all_gather of the
per-unit records at the end of each step):
  msm               bn256 G1 select-chip MSM, 64 tiles x 1024 points per GPU = 2^16 points (configs[1]; 128 tiles per
                    GPU x 8 GPUs = configs[2]); each tile replays the reference's test body
                    (src/tests/native_scalar_ecc_chip.rs:34-47) in its own row space.  DEFAULT.
  pairing_bn256     64 x check_pairing([(a,b),(-a,b)]) (configs[3]; src/tests/native_scalar_pairing_chip.rs:67-97)
  pairing_bls12_381 16 x check_pairing([(ac,b),(-a,bc)]) (configs[4]; src/tests/general_scalar_pairing_chip.rs:74-105)

Steps are pipelined the way a streaming job runs them (a 2^20-point MSM is 1024 tiles through a ring of output
buffers): step k+1 is submitted (h2e_submit) into the other buffer of a ring of `--ring` (default 2) output-buffer
sets while step k's expansion is still streaming, so the value chain of one step runs under the expansion of the
previous one.  The timed region is bracketed by barrier + synchronize on both sides; `--ring 1` runs the steps
strictly one after the other (h2e_run).  Steps alternate between two input batches and the OR of every step's status
words must be 0.  Prints ONE JSON line on rank 0.
"""
import argparse
import concurrent.futures
import csv
import glob
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); measured float4-copy ceiling there: 6.29 TB/s
Q = 0x30644E72E131A029B85045B68181585D97816A916871CA8D3C208C16D87CFD47
DEFAULT_UNITS = {"msm": 64, "pairing_bn256": 64, "pairing_bls12_381": 16}


# ---------------------------------------------------------------------------------------------------------------
# CPU baseline: the oracle (C++ restatement = "port") on this box's host cores with the reference's own parallel
# structure: MSM - window-parallel inside a tile like the rayon region
# (src/circuit/ecc_chip.rs:317-343), tiles in parallel up to memory; pairing - single-threaded per instance (as in
# the reference), instances in parallel across cores.  Bounded sample of the same workload.
def cpu_baseline(workload, points):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib
    import psutil
    from halo2ecc_s_amd import synth
    oracle_lib.load()
    cores = os.cpu_count() or 1
    avail_gb = psutil.virtual_memory().available / 2**30
    if workload == "msm":
        per_unit_gb = 9.0 * points / 1024            # Records of a 1024-point tile: 7 GB resident
        in_flight = int(max(1, min(cores // 4 if cores >= 8 else 1, avail_gb * 0.8 // per_unit_gb, 32)))
        threads = max(1, cores // in_flight)
        inputs = [synth.msm_bn256_tile_inputs(points, tile=900 + t, cheap_points=True, with_expected=False)[0] for t in range(in_flight)]
        fn = lambda inp: oracle_lib.run_msm_bn256_tile(points, inp, threads=threads)   # noqa: E731
        what = f"{in_flight} x {points}-point bn256 MSM tiles (test body incl. assign_point) in parallel, {threads} threads each over the MSM windows"
    else:
        per_unit_gb = 4.0
        n_inst = DEFAULT_UNITS[workload]
        in_flight = int(max(1, min(cores, n_inst, avail_gb * 0.8 // per_unit_gb)))
        threads = 1
        gen = synth.pairing_check_bn256_inputs if workload == "pairing_bn256" else synth.pairing_check_bls12_381_inputs
        run = oracle_lib.run_pairing_check_bn256 if workload == "pairing_bn256" else oracle_lib.run_pairing_check_bls12_381
        inputs = [gen(instance=900 + k) for k in range(in_flight)]
        fn = run
        what = f"{in_flight} x {workload} check_pairing instances in parallel, single-threaded each (as in the reference)"

    def one(inp):
        r = fn(inp)
        cells, st = r.info.n_advice_cells, r.info.status
        r.close()
        return cells, st

    t0 = time.perf_counter()
    with concurrent.futures.ThreadPoolExecutor(in_flight) as ex:   # ctypes calls release the GIL
        res = list(ex.map(one, inputs))
    secs = time.perf_counter() - t0
    cells = sum(c for c, _ in res)
    out = {"value": cells / secs, "unit": "cells/s", "cores": min(cores, in_flight * threads), "kind": "port",
           "sample": f"{what}; {cells} advice cells in {secs:.1f} s wall; oracle C++ restatement; host has {cores} logical cores, "
                     f"{avail_gb:.0f} GiB free memory ({per_unit_gb:.0f} GiB per unit in flight)",
           "units_in_flight": in_flight, "threads_per_unit": threads, "units_per_s": in_flight / secs}
    if workload == "msm":
        out["points_per_s"] = points * in_flight / secs
    return out


# ---------------------------------------------------------------------------------------------------------------
# HBM traffic of the dominant kernel from the PMC counters, measured in this run: two child passes of this same
# script under rocprofv3 (`--pmc WRITE_SIZE`, `--pmc FETCH_SIZE`: the TCC has 4 counters, they do not fit one
# pass), one untimed step each, parsed as MI355X_MICROARCH.md prescribes (values in KB; FETCH_SIZE doubled on gfx950).
def measure_traffic(args):
    rocprof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(rocprof):
        return None, "rocprofv3 not found"
    tmp = tempfile.mkdtemp(prefix="h2e_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    got = {}
    try:
        for counter in ("WRITE_SIZE", "FETCH_SIZE"):
            d = os.path.join(tmp, counter)
            cmd = [rocprof, "--pmc", counter, "-d", d, "-o", "run", "--output-format", "csv", "--", sys.executable,
                   os.path.join(ROOT, "bench.py"), "--pmc-child", "--workload", args.workload, "--units", str(args.units),
                   "--points", str(args.points), "--steps", "1", "--warmup", "0", "--ring", "1", "--no-cpu-baseline"]
            r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=args.traffic_timeout)
            files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            if r.returncode != 0 or not files:
                return None, f"{counter} pass failed (rc {r.returncode}): {r.stderr[-300:]}"
            rows = []
            for row in csv.DictReader(open(files[0])):
                if "h2e_run_tape" in row["Kernel_Name"] and "false" in row["Kernel_Name"] and row["Counter_Name"] == counter:
                    rows.append((int(row["Grid_Size"]), int(row["Dispatch_Id"]), float(row["Counter_Value"])))
            if not rows:
                return None, f"{counter}: no dispatch of the expansion kernel in the counter file"
            got[counter] = rows
    except Exception as e:   # noqa: BLE001  (a failed measurement must not cost the bench line)
        return None, f"{type(e).__name__}: {e}"
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return got, None


def dominant_traffic(got, dom_n):
    """the child ran one step: the dominant launch's dispatches are the dom_n largest grids of the expansion kernel"""
    pick = lambda rows: [v for _, _, v in sorted(sorted(rows, key=lambda r: -r[0])[:dom_n], key=lambda r: r[1])]   # noqa: E731
    wr, rd = pick(got["WRITE_SIZE"]), pick(got["FETCH_SIZE"])
    per_launch = 1e3 * (sum(wr) + 2.0 * sum(rd)) / dom_n   # KB -> bytes; FETCH_SIZE doubled on gfx950 (MI355X_MICROARCH.md)
    return {"bytes_per_launch": per_launch, "launches": dom_n, "WRITE_SIZE_KB": wr, "FETCH_SIZE_KB_raw": rd}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--workload", default="msm", choices=sorted(DEFAULT_UNITS))
    ap.add_argument("--units", "--tiles", type=int, default=None, help="units per GPU: MSM tiles (64 x 1024 = 2^16 points) / pairing instances")
    ap.add_argument("--points", type=int, default=1024, help="points per MSM tile")
    ap.add_argument("--ring", type=int, default=None, help="output-buffer sets steps rotate through = runs in flight (default: 2 for the MSM - step k+1's value chain runs under "
                    "step k's expansion, 2 x 110 GB of arrays; 8 / 16 for the bn256 / bls12_381 pairing checks, whose 28-41 ms level-parallel value chains are latency; 1: h2e_run, no overlap)")
    ap.add_argument("--digest", action="store_true", help="consume every step's arrays with the on-device digest kernel (streaming-job mode, configs[2])")
    ap.add_argument("--job-tiles", type=int, default=None, help="run one MSM job of this many tiles over all ranks (2^20 points = 1024): steps = job_tiles / (units x gpus), digest on")
    ap.add_argument("--cpu-sample-points", type=int, default=1024)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--traffic", default="auto", choices=["auto", "off"], help="auto: measure the dominant kernel's HBM bytes with two rocprofv3 --pmc child passes (N=1 only)")
    ap.add_argument("--traffic-timeout", type=int, default=240)
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--no-check", action="store_true", help="A/B experiments with deliberately broken arithmetic")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL) on GPUs; gloo only to exercise the N>1 path on one GPU")
    ap.add_argument("--device", type=int, default=None, help="override the CUDA device index (default: LOCAL_RANK)")
    args = ap.parse_args()
    if args.units is None:
        args.units = DEFAULT_UNITS[args.workload]
    if args.ring is None:
        # MSM: two 110 GB buffer sets; 64 bn256 checks: 8 runs fill every CU (two instances per workgroup, one workgroup's value
        # slots per CU); 16 bls12_381 checks are 8 workgroups per run - latency: more runs in flight (6.9 -> 6.3 ms per step)
        args.ring = {"msm": 2, "pairing_bn256": 8, "pairing_bls12_381": 16}[args.workload]
    if args.job_tiles:
        args.digest = True
        args.steps = max(1, args.job_tiles // (args.units * max(1, args.gpus)))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world == 1:
        # `--gpus N` without a launcher: start one rank per GPU as a child job (never re-exec a process that touched the GPU)
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(29500 + os.getpid() % 1000), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.run(cmd).returncode)
    # pipelined runs use several HIP streams (caller's, expansion, fix-up, a chain and a side stream per job slot): more than
    # the 4 hardware queues a process gets by default, and streams that share a queue serialise
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "32" if args.ring > 8 else "16")
    if world > 1 and args.gpus != world:
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")

    import torch
    import torch.distributed as dist
    from halo2ecc_s_amd import Engine, Program, parallel, synth

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

    n, units = args.points, args.units
    if args.workload == "msm":
        make = lambda shape: Program.msm_bn256_tile(n, emit_shape=shape)   # noqa: E731
    elif args.workload == "pairing_bn256":
        make = lambda shape: Program.pairing_check_bn256(emit_shape=shape)   # noqa: E731
    else:
        make = lambda shape: Program.pairing_check_bls12_381(emit_shape=shape)   # noqa: E731
    prog = make(False)
    shape_prog = make(True)  # shape-only artefacts, once per shape (not timed)
    cells_per_unit = shape_prog.n_advice_cells
    launches = shape_prog.launches()
    shape_prog.close()
    # dominant kernel = the launch with the most cells (MSM: the window strands; pairing: the whole check)
    dom = max(range(len(launches)), key=lambda i: launches[i]["cells"])

    # HBM traffic of the dominant kernel: child passes under rocprofv3, before this process allocates its arrays
    traffic, traffic_err = None, "not measured"
    if args.traffic == "auto" and rank == 0 and world == 1 and not args.pmc_child:
        traffic, traffic_err = measure_traffic(args)

    eng = Engine(local_rank)
    ring = max(1, min(args.ring, eng.get_stat(4)))
    eng.set_option(4, ring)          # H2E_OPT_PIPELINE_DEPTH: as many job slots (workspaces, streams) as runs in flight
    # The first process that touches the HBM of a freshly booted box pays for it: without this throw-away allocate / fill / free
    # of (almost) the whole memory the steps of that process run 2 x slower than those of any later one (47 vs 24 ms, measured
    # with two bench runs in one gpurun call; exp/first_touch.py).  0.3 s, untimed, no effect on later processes.
    # (Skipped when ranks share a device - `--device`, the one-GPU test of the N > 1 path - where two ranks asking for "all free
    # memory" at the same moment would race; a failed attempt is not an error either.)
    if args.device is None or world == 1:
        try:
            free_b, _total = torch.cuda.mem_get_info(local_rank)
            scratch = torch.empty((int(free_b * 0.92) // 8,), dtype=torch.int64, device=dev)
            scratch.fill_(-1)
            torch.cuda.synchronize()
            del scratch
        except RuntimeError as e:   # out of memory: somebody else is using the device
            print(f"bench.py: first-touch pass skipped ({str(e).splitlines()[0]})", file=sys.stderr)
        torch.cuda.empty_cache()
    bufs = [eng.alloc(prog, units) for _ in range(ring)]   # (base, range, select, status) per ring slot
    out_refs = prog.outputs()
    L = 3

    # synthetic inputs, different per unit and per rank; two batches that the steps alternate between, so that a step
    # can never pass on data a previous step left behind (same inputs every step would hide a missing dependency)
    n_batches = 2
    batches = []
    for bi in range(n_batches):
        first = (bi * world + rank) * units
        if args.workload == "msm":
            ins = np.stack([synth.msm_bn256_tile_inputs(n, tile=first + t, cheap_points=True, with_expected=False)[0] for t in range(units)])
        elif args.workload == "pairing_bn256":
            ins = np.stack([synth.pairing_check_bn256_inputs(instance=first + t) for t in range(units)])
        else:
            ins = np.stack([synth.pairing_check_bls12_381_inputs(instance=first + t) for t in range(units)])
        d_in = eng.upload_inputs(prog, ins)
        if args.workload == "msm" and not args.pmc_child:
            # pass 0: learn each tile's MSM result, then feed it back as the `expected` input so that the in-circuit
            # ecc_assert_equal holds in every timed pass (the reference test computes it with the native library)
            base, rng, sel, status = bufs[0]
            eng.run(prog, d_in, base, rng, sel, status)
            torch.cuda.synchronize()
            exp = np.zeros((units, 3, 4), dtype=np.uint64)
            cells = {r: base[r & 0x3FFFFFF, (r >> 27) & 7].cpu().numpy().view(np.uint64) for r in out_refs}   # [half][inst][2]
            val = lambda r, t: sum(int(cells[r][k // 2, t, k % 2]) << (64 * k) for k in range(4))   # noqa: E731
            for t in range(units):
                xs = [val(r, t) for r in out_refs[0:L]]
                ys = [val(r, t) for r in out_refs[L + 1:2 * L + 1]]
                z = val(out_refs[2 * L + 2], t)
                x = sum(v << (108 * i) for i, v in enumerate(xs)) % Q
                y = sum(v << (108 * i) for i, v in enumerate(ys)) % Q
                if z:
                    x = y = 0
                exp[t] = synth.pack([x, y, z], 4)
            d_in[:, 4 * n + 6:4 * n + 9, :] = torch.from_numpy(exp.view(np.int64)).to(dev)
        batches.append(d_in)
    status_any = torch.zeros_like(bufs[0][3])   # OR of every step's status words
    step_no = [0]
    timing = [False]
    pending = []   # (job, ring slot) submitted and not yet waited for
    gathered_last = [None]
    my_units = parallel.shard_units(units * world, world, rank)   # global unit indices of this rank (round-robin)
    offsets = torch.tensor([prog.base_offset, prog.range_offset, prog.select_offset], dtype=torch.int64, device=dev)
    digests = [torch.zeros((3, units, 4), dtype=torch.int64, device=dev) for _ in range(ring)]
    digest_any = [None]

    def consume(slot):
        """what happens to a finished step's arrays: status check, optional on-device digest (the consumer of a streaming
        job: SURVEY 8d cfg 3), and for N > 1 the one collective of the path - an all_gather of the per-unit records
        {status, Offset, result point cells, 32-byte digest per advice array}"""
        base, rng, sel, status = bufs[slot]
        status_any.bitwise_or_(status)
        if args.digest:
            for region, arr in enumerate((base, rng, sel)):
                eng.digest(prog, region, arr, out=digests[slot][region])
            digest_any[0] = digests[slot]
        if world > 1 or args.digest:
            rec = parallel.unit_records(status, offsets, base, out_refs, digests[slot] if args.digest else None)
            gathered_last[0] = parallel.gather_unit_records(my_units, rec.to(coll_dev), units * world, world)

    launch_ms = []      # per timed step: (value chain ms, expansion ms) per launched segment, from the engine's HIP events

    def retire(job, slot):
        eng.wait(job)                                   # the current stream waits for every array of that step
        consume(slot)
        if timing[0]:
            # (waits on the host for that step only: the next one is already queued, the GPU stays busy)
            launch_ms.append(eng.job_launch_ms(job))

    def step():
        k = step_no[0]
        step_no[0] += 1
        slot = k % ring
        base, rng, sel, status = bufs[slot]
        if ring == 1:
            status.zero_()
            eng.run(prog, batches[k % n_batches], base, rng, sel, status)
            consume(slot)
            if timing[0]:
                torch.cuda.current_stream().synchronize()
                launch_ms.append(eng.last_run_launch_ms())
            return None
        while len(pending) >= ring:                     # the slot's previous step must have been consumed
            retire(*pending.pop(0))
        status.zero_()
        job = eng.submit(prog, batches[k % n_batches], base, rng, sel, status)
        pending.append((job, slot))
        while len(pending) > ring - 1:                  # consume the step before this one (its expansion overlaps our chain)
            retire(*pending.pop(0))
        return job

    def drain():
        while pending:
            retire(*pending.pop(0))

    for _ in range(args.warmup):
        step()
    drain()
    torch.cuda.synchronize()
    if args.warmup > 0 and not args.no_check and not args.pmc_child:
        assert int(status_any.abs().max()) == 0, f"unit status {status_any.cpu().numpy()}"

    eng.set_profiling(True)
    timing[0] = True
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    drain()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if args.pmc_child:
        return
    if not args.no_check:
        assert int(status_any.abs().max()) == 0, f"unit status {status_any.cpu().numpy()}"
    if world > 1:
        tmax = torch.tensor([elapsed], device=coll_dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    # per-launch times of every timed step (HIP events recorded by the engine on the launching streams)
    launch_ms = [ms for ms in launch_ms if len(ms) > dom]
    dom_n = eng.last_run_expansion_launches()[dom]
    if traffic:
        traffic = dominant_traffic(traffic, dom_n)
    total_cells = cells_per_unit * units * world * args.steps
    # a big expansion goes out as two back-to-back kernel launches over a prefix / the rest of its sub-ranges (h2e.h):
    # the events bracket both, so the per-launch figures are bracket / n and bytes / n
    dom_ms = float(np.mean([ms[dom][1] for ms in launch_ms])) / dom_n
    dom_bytes = 32.0 * launches[dom]["cells"] * units / dom_n
    achieved = dom_bytes / (dom_ms * 1e-3) / 1e9
    ms_per_step = 1e3 * elapsed / args.steps
    step_bytes = 32.0 * cells_per_unit * units
    if args.workload == "msm":
        desc = (f"bn256 G1 select-chip MSM witness, {units} tiles x {n} points per GPU "
                f"(2^{int(np.log2(max(1, units * n)))} points/GPU), reference test body per tile")
        kernel = "h2e_run_tape<FP_BN256_FQ, false> (full expansion of the MSM window strands)"
    else:
        desc = f"{units} x {args.workload} check_pairing (2 pairs, G2 constant) per GPU, reference test shape"
        kernel = ("h2e_run_tape<FP_BN256_FQ, false>" if args.workload == "pairing_bn256" else "h2e_run_tape<FP_BLS_FQ, false>") + " (full expansion of the pairing check)"
    out = {
        "metric": "witness_cells_per_sec",
        "value": total_cells / elapsed,
        "unit": "cells/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": ms_per_step,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u64",
        "data": "synthetic",
        "config": {"workload": desc, "units_per_gpu": units, "cells_per_unit": cells_per_unit,
                   "pipeline": f"ring of {ring} output-buffer sets, steps submitted with h2e_submit" if ring > 1 else "h2e_run, one step after the other",
                   "sharding": f"units round-robin over {world} GPU(s), all_gather of the per-unit records"},
        "whole_step": {"algorithmic_bytes": step_bytes, "achieved": step_bytes / (ms_per_step * 1e-3) / 1e9, "unit": "GB/s",
                       "frac": step_bytes / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic["bytes_per_launch"] if traffic else None,
                     "kernel": kernel, "launch_ms": dom_ms, "algorithmic_bytes_per_launch": dom_bytes, "launches_per_step": dom_n,
                     "value_chain_ms": [float(x) for x in np.mean(np.array([[a for a, _ in ms] for ms in launch_ms]), axis=0)],
                     "expansion_ms": [float(x) for x in np.mean(np.array([[b for _, b in ms] for ms in launch_ms]), axis=0)]},
    }
    if args.workload == "msm":
        out["msm_points_per_sec"] = n * units * world * args.steps / elapsed
        out["config"].update(tiles_per_gpu=units, points_per_tile=n, cells_per_tile=cells_per_unit,
                             points_note="per-tile-batch rate; the test body's assign_point / assign rows are part of every tile")
    if args.digest:
        out["config"]["consumer"] = "h2e_digest over the three advice arrays of every step (32 B per array and unit), inside the timed region"
        out["digest_sample"] = [int(x) & 0xFFFFFFFFFFFFFFFF for x in digest_any[0][0, 0].cpu().tolist()] if digest_any[0] is not None else None
    if gathered_last[0] is not None:
        g = gathered_last[0]
        out["gathered_records"] = {"shape": list(g.shape), "status_or": int(g[:, 0].abs().max())}
    if traffic:
        out["roofline"]["traffic_detail"] = traffic
        out["roofline"]["traffic_source"] = "two rocprofv3 --pmc child passes of this command (WRITE_SIZE + 2 x FETCH_SIZE, KB -> bytes), this run"
    else:
        out["roofline"]["traffic_note"] = traffic_err
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args.workload, args.cpu_sample_points)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
