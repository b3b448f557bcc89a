"""The boundary's threading contract (include/h2e.h "Threads"; SURVEY.md 8(b): re-entrant per h2e_ctx, calls on different
contexts may run concurrently, no global state in the data path) under contention.

The reference's analogue is test_one_line_mt (src/tests/base_chip.rs:47-100): ten threads writing disjoint row ranges of
cloned contexts (`unsafe impl Send`, native_scalar_ecc_chip.rs:92), then one MockProver run over the union.  Here:
  * several host threads hammer ONE context with different programs, each thread on its own HIP stream and its own arrays
    (the context's lock serialises the submissions, the job slots keep the runs' workspaces apart);
  * two contexts on one device run from two threads at the same time;
every result is compared cell for cell with the oracle afterwards.  ctypes releases the GIL for the duration of a C call, so
the threads really are inside libh2e.so together."""
import os
import threading

import numpy as np
import pytest

import oracle_lib
from halo2ecc_s_amd import Engine, Program, synth
from parity import compare_advice

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _jobs():
    """(name, program factory, inputs of instance k, oracle runner) - four different shapes, two field pairs"""
    n_a, n_b = 12, 5
    return [
        ("msm12", lambda: Program.msm_bn256_tile(n_a), lambda k: synth.msm_bn256_tile_inputs(n_a, tile=500 + k)[0],
         lambda inp: oracle_lib.run_msm_bn256_tile(n_a, inp)),
        ("msm5_no_select", lambda: Program.msm_bn256_tile(n_b, with_select=False), lambda k: synth.msm_bn256_tile_inputs(n_b, tile=520 + k)[0],
         lambda inp: oracle_lib.run_msm_bn256_tile(n_b, inp, with_select=False)),
        ("integer_chip_fp1", lambda: Program.integer_chip_st(1), lambda k: synth.integer_chip_st_inputs(1, seed_index=540 + k),
         lambda inp: oracle_lib.run_integer_chip_st(1, inp)),
        ("int_mul_fp0", lambda: Program.int_mul_batch(0, 9), lambda k: synth.int_mul_batch_inputs(0, 9, seed_index=560 + k),
         lambda inp: oracle_lib.run_int_mul_batch(0, 9, inp)),
    ]


def _worker(eng, job, rounds, n_inst, results, errors, barrier, tid):
    """one host thread: its own stream, its own arrays, `rounds` runs with different inputs each, alternating h2e_run and
    h2e_submit + h2e_wait; keeps the LAST round's arrays for the comparison"""
    try:
        t = eng.torch
        name, make, gen, _ = job
        prog = make()
        stream = t.cuda.Stream(device=eng.device)
        arrs = eng.alloc(prog, n_inst, fill=0xFF)
        barrier.wait(timeout=120)
        ins = None
        for r in range(rounds):
            ins = [gen(100 * tid + 10 * r + k) for k in range(n_inst)]
            d_in = eng.upload_inputs(prog, np.stack(ins))
            stream.wait_stream(t.cuda.current_stream())   # (the upload ran on this thread's current stream)
            if r % 2 == 0:
                eng.run(prog, d_in, *arrs, stream=stream)
            else:
                job_id = eng.submit(prog, d_in, *arrs, stream=stream)
                eng.wait(job_id, stream=stream)
            stream.synchronize()   # d_in may be freed / reused by the next round
        rows = tuple(eng.export(prog, region, a, stream=stream) for region, a in enumerate(arrs[:3]))
        stream.synchronize()
        results[tid] = (prog, ins, rows, arrs[3].cpu().numpy())
    except BaseException as e:   # noqa: BLE001  (reported by the main thread)
        errors.append((tid, repr(e)))
        try:
            barrier.abort()
        except Exception:   # noqa: BLE001
            pass


def _compare(job, res):
    prog, ins, (base, rng, sel), status = res
    assert (status == 0).all(), (job[0], status)
    for k in (0, len(ins) - 1):
        orun = job[3](ins[k])
        assert orun.info.status == 0, orun.error
        compare_advice(prog, orun, base, rng, sel, instance=k)
        orun.close()


def test_threads_on_one_context(engine, oracle):
    """four host threads, four different programs (two field pairs), one h2e_ctx"""
    jobs = _jobs()
    results, errors = {}, []
    barrier = threading.Barrier(len(jobs))
    runs_before = engine.get_stat(2)   # H2E_STAT_RUNS
    threads = [threading.Thread(target=_worker, args=(engine, job, 6, 3, results, errors, barrier, tid)) for tid, job in enumerate(jobs)]
    for th in threads:
        th.start()
    for th in threads:
        th.join(timeout=600)
        assert not th.is_alive(), "a thread hung inside the engine"
    assert not errors, errors
    assert engine.get_stat(2) == runs_before + len(jobs) * 6, "a run was lost or counted twice"
    for tid, job in enumerate(jobs):
        _compare(job, results[tid])


def test_two_contexts_on_one_device(engine, oracle):
    """two h2e_ctx on the same device, one thread each, running at the same time (plus the session's context idle beside them)"""
    jobs = _jobs()[:2]
    engines = [Engine(engine.device), Engine(engine.device)]
    results, errors = {}, []
    barrier = threading.Barrier(2)
    threads = [threading.Thread(target=_worker, args=(engines[tid], jobs[tid], 5, 4, results, errors, barrier, tid)) for tid in range(2)]
    for th in threads:
        th.start()
    for th in threads:
        th.join(timeout=600)
        assert not th.is_alive(), "a thread hung inside the engine"
    assert not errors, errors
    for tid in range(2):
        _compare(jobs[tid], results[tid])
    for e in engines:
        e.close()


def test_same_program_from_two_threads(engine, oracle):
    """ONE program handle run by two threads at once into different arrays (the job slots keep the workspaces apart; the
    program's device copy is made once under the context's lock)"""
    job = _jobs()[0]
    prog = job[1]()
    shared = (job[0], lambda: prog, job[2], job[3])
    results, errors = {}, []
    barrier = threading.Barrier(2)
    threads = [threading.Thread(target=_worker, args=(engine, shared, 4, 2, results, errors, barrier, tid)) for tid in range(2)]
    for th in threads:
        th.start()
    for th in threads:
        th.join(timeout=600)
        assert not th.is_alive(), "a thread hung inside the engine"
    assert not errors, errors
    for tid in range(2):
        _compare(shared, results[tid])


def test_pipeline_depth_warns_when_the_process_has_too_few_hardware_queues():
    """GPU_MAX_HW_QUEUES is read by the HIP runtime once, from the environment, before the first HIP call: the library cannot set it
    for a host that pipelines.  h2e_ctx_set_option(H2E_OPT_PIPELINE_DEPTH) therefore says so - h2e_last_warning(), one line on
    stderr, H2E_STAT_HW_QUEUES / _WANTED - instead of letting the streams serialise silently (1.8 instead of 0.7 ms per 8-check step).
    Child processes: the knob is per process."""
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from halo2ecc_s_amd import Engine\n"
            "from halo2ecc_s_amd.engine import OPT_PIPELINE_DEPTH, STAT_HW_QUEUES, STAT_HW_QUEUES_WANTED\n"
            "e = Engine(0)\n"
            "e.set_option(OPT_PIPELINE_DEPTH, 16)\n"
            "print('W16', repr(e.last_warning()), e.get_stat(STAT_HW_QUEUES), e.get_stat(STAT_HW_QUEUES_WANTED))\n"
            "e.set_option(OPT_PIPELINE_DEPTH, 1)\n"
            "print('W1', repr(e.last_warning()), e.get_stat(STAT_HW_QUEUES_WANTED))\n") % ROOT
    env = dict(os.environ)
    env.pop("GPU_MAX_HW_QUEUES", None)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-1500:]
    w16, w1 = [ln for ln in r.stdout.splitlines() if ln.startswith("W")]
    assert "GPU_MAX_HW_QUEUES >= 28" in w16 and w16.endswith(" 4 28") and "libh2e: warning" in r.stderr
    assert w1 == "W1 '' 1"
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(env, GPU_MAX_HW_QUEUES="28"), timeout=600)
    assert r.returncode == 0, r.stderr[-1500:]
    assert [ln for ln in r.stdout.splitlines() if ln.startswith("W16")] == ["W16 '' 28 28"] and "libh2e: warning" not in r.stderr
